/*
  kernels/encoder.hip.h -- canonical run encoder (RunBuffer + Run::write) with the block starts of BWT::build.
  Part of bwtm_kernels.hip.h (included there, inside namespace bwtm); gfx950 only.
*/
#pragma once

//------------------------------------------------------------------------------
// K4: canonical run encoder (RunBuffer + Run::write; utils.h:121-142, support.h:256-282).
//
// A "head" is a position whose symbol differs from its predecessor (position 0 is a head; a
// virtual head sits at position n).  Every head h > 0 is an EVENT: the maximal run
// [previous head, h) with symbol sym(h - 1) ends there and is encoded.  Events are encoded in
// order; a run shorter than 42 is always one byte, a longer run takes a number of bytes that
// depends on the byte offset modulo 64 (support.h:267-279).
//
//   tile    = 64 positions (one lane)      chunk = 64 tiles (one wave step)
//   segment = SEG_CHUNKS chunks, processed sequentially by one wave
//
//   k_enc_lasthead : last head of every segment (+1; 0 = none)  -> exclusive max-scan
//   k_enc_size     : for every segment, bytes emitted as a function of the start offset
//                    mod 64 (lane o evaluates hypothesis o)       -> folded by k_fold_*
//   k_enc_emit     : writes the bytes of every segment at its now known offset

constexpr int SEG_CHUNKS = 16;
constexpr u64 SEG_TILES = (u64)SEG_CHUNKS * 64;
constexpr u64 NONE = 0;   // "position + 1" encoding: 0 means no head

struct TileInfo
{
  u64 p0, p1, p2;    // planes of the tile
  u32 prev;          // symbol at tile_base - 1
  u64 H;             // heads (including position 0 and the virtual head at n)
  u64 E;             // events (H without position 0)
};

// Planes of tile T (64 positions) of the encoded index.
__device__ inline void load_tile(const uint4* recs, u64 nrecs, u64 T, u64& p0, u64& p1, u64& p2)
{
  u64 ch = 2 * T;                                 // 16-byte chunk index: record T >> 1, chunks 2 (T & 1) and + 1
  if(ch + 1 < 4 * nrecs)
  {
    uint4 a = recs[ch], b = recs[ch + 1];
    p0 = (u64)a.x | ((u64)b.x << 32); p1 = (u64)a.y | ((u64)b.y << 32); p2 = (u64)a.z | ((u64)b.z << 32);
  }
  else { p0 = p1 = p2 = 0; }
}

__device__ inline u32 symbol_at(const uint4* recs, u64 pos)
{
  const u32* words = (const u32*)recs;
  u64 wbase = (pos >> REC_SHIFT) * REC_WORDS + ((pos >> 5) & 3) * 4;
  u32 t = (u32)(pos & 31);
  return ((words[wbase] >> t) & 1u) | (((words[wbase + 1] >> t) & 1u) << 1) | (((words[wbase + 2] >> t) & 1u) << 2);
}

// Heads and events of one tile.  `prev` is the symbol at tile_base - 1 (ignored for tile 0).
__device__ inline void tile_heads(TileInfo& ti, u64 tile_base, u64 n)
{
  u64 q0 = (ti.p0 << 1) | (ti.prev & 1u), q1 = (ti.p1 << 1) | ((ti.prev >> 1) & 1u), q2 = (ti.p2 << 1) | ((ti.prev >> 2) & 1u);
  u64 D = (ti.p0 ^ q0) | (ti.p1 ^ q1) | (ti.p2 ^ q2);
  if(tile_base == 0) { D |= 1; }
  u64 valid = (n >= tile_base + 64 ? ~0ull : (n <= tile_base ? 0ull : ((1ull << (n - tile_base)) - 1)));
  D &= valid;
  if(n >= tile_base && n < tile_base + 64) { D |= 1ull << (n - tile_base); }
  ti.H = D;
  ti.E = (tile_base == 0 ? D & ~1ull : D);
}

// Symbol of the run that ends at in-tile bit t (the symbol at position tile_base + t - 1).
__device__ inline u32 event_symbol(const TileInfo& ti, u32 t)
{
  if(t == 0) { return ti.prev; }
  u32 s = t - 1;
  return (u32)((ti.p0 >> s) & 1) | ((u32)((ti.p1 >> s) & 1) << 1) | ((u32)((ti.p2 >> s) & 1) << 2);
}

// Loads the tiles of one chunk (lane = tile) and computes heads; `carry_prev` is the symbol
// before the chunk (wave-uniform).  Returns the symbol at the end of the chunk for the next one.
__device__ inline u32 chunk_tiles(const uint4* recs, u64 nrecs, u64 first_tile, u64 n, u32 carry_prev, TileInfo& ti)
{
  u64 T = first_tile + lane_id();
  load_tile(recs, nrecs, T, ti.p0, ti.p1, ti.p2);
  u32 last = (u32)((ti.p0 >> 63) & 1) | ((u32)((ti.p1 >> 63) & 1) << 1) | ((u32)((ti.p2 >> 63) & 1) << 2);
  u32 up = (u32)__shfl_up((int)last, 1, WAVE);
  ti.prev = (lane_id() == 0 ? carry_prev : up);
  tile_heads(ti, T << 6, n);
  return (u32)__shfl((int)last, WAVE - 1, WAVE);
}

// Maximum over the wave of a small value (DPP row shifts and broadcasts, like wave_incl_sum32); wave-uniform result.
__device__ inline u32 wave_max32(u32 v)
{
#define BWTM_MAX_DPP(ctrl, rows) { const u32 t = (u32)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xF, false); v = (t > v ? t : v); }
  BWTM_MAX_DPP(0x111, 0xF) BWTM_MAX_DPP(0x112, 0xF) BWTM_MAX_DPP(0x114, 0xF) BWTM_MAX_DPP(0x118, 0xF) BWTM_MAX_DPP(0x142, 0xA) BWTM_MAX_DPP(0x143, 0xC)
#undef BWTM_MAX_DPP
  return (u32)__builtin_amdgcn_readlane((int)v, WAVE - 1);
}

// Position of the k-th (0-based) set bit of x; k < popcount(x).
__device__ inline u32 select64(u64 x, u32 k)
{
  const u32 lo = (u32)x, clo = (u32)__builtin_popcount(lo);
  const bool hi = (k >= clo);
  u32 w = (hi ? (u32)(x >> 32) : lo), base = (hi ? 32u : 0u);
  k -= (hi ? clo : 0u);
#pragma unroll
  for(u32 sft = 16; sft != 0; sft >>= 1)
  {
    const u32 low = w & ((1u << sft) - 1u), c = (u32)__builtin_popcount(low);
    const bool up = (k >= c);
    w = (up ? w >> sft : low); k -= (up ? c : 0u); base += (up ? sft : 0u);
  }
  return base;
}

// LDS of k_enc_emit: one staging area per wave (4096 + 15 bytes of a chunk at most, 16-byte phase of the destination) and a DUMP region that
// absorbs the byte stores of the walk's idle trips: (address | ENC_DUMP) lies in [ENC_DUMP, ENC_DUMP + 4096 + 3] for every staging address.
constexpr u32 ENC_STAGE_STRIDE = 4352;                               // 4 waves: [0, 17184)
constexpr u32 ENC_DUMP = 0x7000;
constexpr u32 ENC_LDS_BYTES = 0x8000 + 16;
#ifndef BWTM_ENC_WALK_UNROLL
#define BWTM_ENC_WALK_UNROLL 4
#endif
constexpr u32 ENC_WALK = BWTM_ENC_WALK_UNROLL;                       // events per trip of the walk (a power of two)

// The one-byte events of one 32-bit half of a tile, in wave-uniform trips of ENC_WALK events (k_enc_emit's short-run path).  Every lane runs the
// same number of trips -- the wave's maximum -- so the loop is scalar (no execution-mask bookkeeping, which was a third of the old loop's
// instructions); a lane that has run out of events finds no bit (v_ffbl_b32 returns -1) and its byte goes to the dump region.
//   hh: event bits of the half; q0..q2: planes of the half; prev: in-half position of the previous head (negative: before the half);
//   sym: symbol of the run that is open at the start of the half; addr: LDS byte address of the half's first event
__device__ inline void walk_half(u32 hh, u32 q0, u32 q1, u32 q2, int prev, u32 sym, u32 addr, u32 trips)
{
  typedef __attribute__((address_space(3))) u8 lds_u8;
  for(u32 t = 0; t < trips; t += ENC_WALK)
  {
#pragma unroll
    for(u32 j = 0; j < ENC_WALK; j++)
    {
      u32 bb, byte, b0, b1, b2;
      asm("v_ffbl_b32 %0, %1" : "=v"(bb) : "v"(hh));                    // -1 when the half has no event left
      const u32 len1 = bb + ~(u32)prev;                                 // length - 1 <= 40
      asm("v_mad_u32_u24 %0, %1, 6, %2" : "=v"(byte) : "v"(len1), "v"(sym));     // sym + 6 (length - 1): Run::encodeBasic, support.h:231-234
      const u32 a = (bb & ENC_DUMP) | addr;
      ((lds_u8*)(uintptr_t)a)[j] = (u8)byte;
#ifdef BWTM_SLACK_ENC_EMIT
      { u32 slack = byte; valu_slack<BWTM_SLACK_ENC_EMIT>(slack); }
#endif
      b0 = __builtin_amdgcn_ubfe(q0, bb, 1u); b1 = __builtin_amdgcn_ubfe(q1, bb, 1u); b2 = __builtin_amdgcn_ubfe(q2, bb, 1u);
      asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(sym) : "v"(b1), "v"(b0));
      asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(sym) : "v"(b2), "v"(sym));
      prev = (int)bb;
      hh &= hh - 1;
    }
    addr += ENC_WALK;
  }
}

// The three segment kernels cover the segments [seg_first, seg_end) and index `recs` and the per-segment arrays by GLOBAL
// record / segment numbers: an output-range slice (one GPU's share of the result) passes the addresses of its local buffers
// minus the offsets of its first record / segment.  `carry` = (last head before the first segment of the launch) + 1 as
// known from the slices before (0 for the whole index): the run that is open at the slice boundary started there.
__global__ void __launch_bounds__(BLOCK_THREADS) k_enc_lasthead(const uint4* recs, u64 nrecs, u64 n, u64 ntiles, u64 seg_first, u64 seg_end, u64* lasthead)
{
  u64 seg = seg_first + (((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6);
  if(seg >= seg_end) { return; }
  const u64 first = seg * SEG_TILES;
  // From the segment's last chunk backwards: the first chunk that holds a head has the answer, and unless the segment ends inside a run
  // of more than 4096 positions that is the very first one looked at -- one chunk read per segment instead of sixteen (the forward
  // pass was a full read of the records, 0.8 ms at config 2, for a number that sits in the last tile of almost every segment).
  const u64 tiles_here = (ntiles - first < SEG_TILES ? ntiles - first : SEG_TILES);   // > 0: every segment of the launch has tiles
  u64 best = NONE;
  for(int k = (int)((tiles_here - 1) >> 6); k >= 0; k--)
  {
    const u64 ft = first + (u64)k * 64;
    const u32 carry = (ft == 0 ? 0u : symbol_at(recs, (ft << 6) - 1));
    TileInfo ti;
    (void)chunk_tiles(recs, nrecs, ft, n, carry, ti);
    const u64 T = ft + lane_id();
    const u64 mine = (ti.H != 0 && T < ntiles ? (T << 6) + (63 - (u64)__builtin_clzll(ti.H)) + 1 : NONE);
    const u64 m = shfl_u64(wave_incl_last(mine), WAVE - 1);                // the last head of the chunk
    if(m != NONE) { best = m; break; }
  }
  if(lane_id() == 0) { lasthead[seg] = best; }
}

// Per-lane event statistics of a tile: number of events and the LONG events among them (heads that end a
// run of >= 42: no other head among the 41 positions before them; at most two per tile).  `before` =
// (position of the last head before this tile) + 1.
// dep_len (round 6): != 0 when the tile's FIRST head ends a run of >= 83 = 2 * 42 - 1, its length.  Only such runs have a size that depends on the
// byte offset: a run of 42 .. 82 is two bytes wherever it starts (head + one extension byte, or -- when the head is the last byte of its
// block -- a run of 41 and a run of 1 .. 41, support.h:267-279), so the ordered walk over the long events (for_each_dep_event) only has to
// visit the others; a later head of a tile is never one of them (the head before it lies in the same 64 positions).
constexpr u64 DEP_RUN = 2 * MAX_RUN - 1;

__device__ inline void tile_event_stats(const TileInfo& ti, u64 tile_base, u64 before, u32& nev, u64& long_mask, u64& dep_len)
{
  nev = (u32)__builtin_popcountll(ti.E); long_mask = 0; dep_len = 0;
  const u64 H = ti.H;
  if(H == 0) { return; }
  // covered = OR of H << k for k = 1..41: positions that have a head among the 41 positions before them
  u64 s = H << 1;
  s |= s << 1; s |= s << 2; s |= s << 4;          // k = 1..8
  const u64 s9 = s | (H << 9);                     // k = 1..9
  s |= s << 8; s |= s << 16;                       // k = 1..32
  const u64 covered = s | (s9 << 32);              // k = 1..41
  const u64 later = H & (H - 1);                   // heads other than the first one of the tile
  long_mask = later & ~covered;
  const u64 pos = tile_base + (u32)__builtin_ctzll(H);   // first head: its run started before the tile
  if(pos > 0 && pos + 1 - before >= MAX_RUN) { long_mask |= H & (0 - H); if(pos + 1 - before >= DEP_RUN) { dep_len = pos + 1 - before; } }
}

// The offset-dependent long events of a chunk in position order: f(t, bytes, len) with t = tile (lane), bytes = bytes the chunk emits before the
// event when every such event before it is counted as ONE byte (pre_excl: the lane's exclusive prefix of its events + its two-byte long events),
// len = run length; all arguments wave-uniform.
template<class F>
__device__ inline void for_each_dep_event(u64 dep_len, u32 pre_excl, F&& f)
{
  u64 pending = __ballot(dep_len != 0);
  while(pending)
  {
    const int t = (int)__builtin_ctzll(pending); pending &= pending - 1;
    f((u32)t, (u32)__shfl((int)pre_excl, t, WAVE), shfl_u64(dep_len, t));
  }
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_enc_size(const uint4* recs, u64 nrecs, u64 n, u64 ntiles, u64 seg_first, u64 seg_end,
  const u64* prevhead, u64 head_carry, u32* table)
{
  u64 seg = seg_first + (((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6);
  if(seg >= seg_end) { return; }
  u64 first = seg * SEG_TILES;
  u32 carry = (first == 0 ? 0u : symbol_at(recs, (first << 6) - 1));
  u64 last = prevhead[seg];          // (last head before the segment) + 1, wave-uniform
  if(head_carry > last) { last = head_carry; }
  u64 acc = 0;                       // bytes emitted so far under hypothesis "start offset = lane"
  const u32 o = lane_id();
  for(int k = 0; k < SEG_CHUNKS; k++)
  {
    u64 ft = first + (u64)k * 64;
    if(ft >= ntiles) { break; }
    TileInfo ti;
    carry = chunk_tiles(recs, nrecs, ft, n, carry, ti);
    u64 T = ft + lane_id();
    if(T >= ntiles) { ti.H = 0; ti.E = 0; }
    u64 lh = (ti.H != 0 ? (T << 6) + (63 - (u64)__builtin_clzll(ti.H)) + 1 : NONE);
    u64 incl = wave_incl_last(lh);                                   // lh grows with the lane where it is set: the inclusive maximum is the last one so far
    u64 before = shfl_up_u64(incl, 1);
    if(lane_id() == 0) { before = NONE; }
    if(last > before) { before = last; }
    u32 nev; u64 long_mask, dep_len;
    tile_event_stats(ti, T << 6, before, nev, long_mask, dep_len);
    // Events shorter than 42 are one byte under every hypothesis and runs of 42 .. 82 two; only the longer ones are resolved in order
    // (round 6: on reads of a genome the ordered walk visited every run of >= 42, a tenth of the events, with three shuffles each).
    const u32 two = (u32)__builtin_popcountll(long_mask) - (dep_len != 0 ? 1u : 0u);
    const u32 pre = nev + two;
    const u32 pre_incl = wave_incl_sum32(pre);
    const u32 chunk_bytes = (u32)__builtin_amdgcn_readlane((int)pre_incl, WAVE - 1);
    u32 last_g = 0;
    for_each_dep_event(dep_len, pre_incl - pre, [&](u32, u32 g, u64 len)
    {
      acc += g - last_g;
      acc += long_run_bytes((u64)o + acc, len);
      last_g = g + 1;
    });
    acc += chunk_bytes - last_g;
    u64 m = shfl_u64(incl, WAVE - 1);
    if(m > last) { last = m; }
  }
  table[seg * 64 + o] = (u32)acc;
}

// Fold 1: composition of the segment tables of one group (lane o = start offset hypothesis).
// The folds are chains of dependent table reads (the next index is the offset the previous entry led to); round 5 stages the tables in LDS,
// FOLD_TILE at a time with coalesced loads, so that the chain runs at LDS latency instead of one L2 round trip per entry
// (k_fold_group / k_fold_top / k_fold_seg: 0.105 + 0.149 + 0.058 -> 0.03 + 0.03 + 0.03 ms per merge at config 2).
constexpr int FOLD_GROUP = 256;
constexpr int FOLD_TILE = 64;               // tables staged at a time: 64 x 64 entries

__global__ void __launch_bounds__(WAVE) k_fold_group(const u32* table, u64 nseg, u64* group_table)
{
  __shared__ u32 tile[FOLD_TILE * 64];
  u64 g = blockIdx.x;
  u64 s0 = g * FOLD_GROUP, s1 = s0 + FOLD_GROUP; if(s1 > nseg) { s1 = nseg; }
  u64 acc = 0; u32 o = lane_id();
  for(u64 t0 = s0; t0 < s1; t0 += FOLD_TILE)
  {
    const u32 cnt = (u32)(s1 - t0 < (u64)FOLD_TILE ? s1 - t0 : (u64)FOLD_TILE);
    for(u32 k = 0; k < cnt; k++) { tile[k * 64 + o] = table[(t0 + k) * 64 + o]; }       // independent, coalesced loads
    wave_sync_lds();
    for(u32 k = 0; k < cnt; k++) { acc += tile[k * 64 + ((o + (u32)acc) & 63)]; }
    wave_sync_lds();
  }
  group_table[g * 64 + o] = acc;
}

// Fold 1b (output-range slices): composition of all group tables of a slice: bytes the slice emits as a function of the offset
// it starts at (mod 64) -- what travels to the other GPUs.
__global__ void __launch_bounds__(WAVE) k_fold_slice(const u64* group_table, u64 ngroups, u64* slice_table)
{
  u64 acc = 0; u32 o = lane_id();
  for(u64 g = 0; g < ngroups; g++) { acc += group_table[g * 64 + ((o + acc) & 63)]; }
  slice_table[o] = acc;
}

// Fold 2: sequential pass over the groups from the byte offset `start` (0 for the whole index); group_base[ngroups] = end offset.
__global__ void __launch_bounds__(WAVE) k_fold_top(const u64* group_table, u64 ngroups, u64 start, u64* group_base)
{
  __shared__ u64 tile[FOLD_TILE * 64];
  __shared__ u64 bases[FOLD_TILE];
  const u32 o = lane_id();
  u64 off = start;                                                  // wave-uniform
  for(u64 t0 = 0; t0 < ngroups; t0 += FOLD_TILE)
  {
    const u32 cnt = (u32)(ngroups - t0 < (u64)FOLD_TILE ? ngroups - t0 : (u64)FOLD_TILE);
    for(u32 k = 0; k < cnt; k++) { tile[k * 64 + o] = group_table[(t0 + k) * 64 + o]; }
    wave_sync_lds();
    if(o == 0)
    {
      for(u32 k = 0; k < cnt; k++) { bases[k] = off; off += tile[k * 64 + (u32)(off & 63)]; }
    }
    wave_sync_lds();
    if(o < cnt) { group_base[t0 + o] = bases[o]; }
    off = shfl_u64(off, 0);
    wave_sync_lds();
  }
  if(o == 0) { group_base[ngroups] = off; }
}

// Fold 3: byte offset of every segment.
__global__ void __launch_bounds__(WAVE) k_fold_seg(const u32* table, u64 nseg, const u64* group_base, u64* seg_base)
{
  __shared__ u32 tile[FOLD_TILE * 64];
  __shared__ u64 bases[FOLD_TILE];
  const u32 o = lane_id();
  u64 g = blockIdx.x;
  u64 s0 = g * FOLD_GROUP, s1 = s0 + FOLD_GROUP; if(s1 > nseg) { s1 = nseg; }
  u64 off = group_base[g];
  for(u64 t0 = s0; t0 < s1; t0 += FOLD_TILE)
  {
    const u32 cnt = (u32)(s1 - t0 < (u64)FOLD_TILE ? s1 - t0 : (u64)FOLD_TILE);
    for(u32 k = 0; k < cnt; k++) { tile[k * 64 + o] = table[(t0 + k) * 64 + o]; }
    wave_sync_lds();
    if(o == 0)
    {
      for(u32 k = 0; k < cnt; k++) { bases[k] = off; off += tile[k * 64 + (u32)(off & 63)]; }
    }
    wave_sync_lds();
    if(o < cnt) { seg_base[t0 + o] = bases[o]; }
    off = shfl_u64(off, 0);
    wave_sync_lds();
  }
}

// Samples of the result (BWT::build, bwt.cpp:489-511) in the device's compact form: cum32[(c - 1) * stride + b] = occurrences of c (1..5)
// before the start p of block b MINUS the super table's entry of p's super block, i.e. exactly what a record stores for a position
// (header field + popcount): 20 bytes per block instead of the 48 of the six u64 arrays, no super-table access when it is produced, and
// samples[c].sum(b) = sup[8 (p >> 25) + c] + cum32[c - 1][b]  (c = 0: p minus the five).  k_cum_expand turns it into the u64 arrays on request.
//
// k_enc_emit produces the values from what its wave holds already (round 5; rounds 1 - 4 ran one rank query per block start, which
// re-fetched the record and the super row: 1.53 x the kernel's algorithmic read traffic, PMC in profiles/r04_pmc_per_kernel.txt).  Every block
// start p lies in a run that ENDS at a head h of the lane's own tile, so
//   rank_c(p) = rank_c(tile start) + #c in [tile start, h) - (c == symbol of the run ? h - p : 0)
// and rank_c(tile start) = (header of the segment's first record: ONE 64-byte read per 65 536 positions) + the symbol counts of the
// tiles before it (a running total over the chunks + a wave scan of the tiles' five popcounts, packed two to a register).

// Occurrences of the symbols 1..5 among the positions of a tile selected by the mask m.
__device__ inline void tile_counts(u64 p0, u64 p1, u64 p2, u64 m, u32& n1, u32& n2, u32& n3, u32& n4, u32& n5)
{
  const u64 only0 = p0 & ~p1 & m, only1 = ~p0 & p1 & m, both = p0 & p1 & m, none = ~(p0 | p1) & m;
  n1 = (u32)__builtin_popcountll(only0 & ~p2); n2 = (u32)__builtin_popcountll(only1 & ~p2); n3 = (u32)__builtin_popcountll(both & ~p2);
  n4 = (u32)__builtin_popcountll(none & p2);   n5 = (u32)__builtin_popcountll(only0 & p2);
}

// The five 25-bit fields of a record header (rec_header, bwtm_device.h) from its four words.
__device__ inline void header_fields(u32 h0, u32 h1, u32 h2, u32 h3, u32 f[6])
{
  const u64 lo = (u64)h0 | ((u64)h1 << 32), hi = (u64)h2 | ((u64)h3 << 32);
  f[0] = 0;
  f[1] = (u32)lo & FIELD_MASK; f[2] = (u32)(lo >> 25) & FIELD_MASK; f[3] = (u32)((lo >> 50) | (hi << 14)) & FIELD_MASK;
  f[4] = (u32)(hi >> 11) & FIELD_MASK; f[5] = (u32)(hi >> 36) & FIELD_MASK;
}

// The general form: one rank query on the records (the entry behind the last block; block starts whose run began in an earlier super
// block than the segment that encodes it; indexes whose samples are asked for without having been encoded here).
__device__ inline void block_cum32_query(const uint4* recs, u64 p, u32 r[6])
{
  const uint4* rec = recs + 4 * (p >> REC_SHIFT);
  const u32 j = (u32)(p & (REC_POS - 1));
  u32 n[6] = {0, 0, 0, 0, 0, 0}, h[4];
#pragma unroll
  for(u32 k = 0; k < 4; k++)
  {
    const uint4 ch = rec[k];
    const u32 mask = below_mask(j, k);
#pragma unroll
    for(u32 c = 1; c < 6; c++) { n[c] += (u32)__builtin_popcount(plane_match(ch.x, ch.y, ch.z, c) & mask); }
    h[k] = ch.w;
  }
  header_fields(h[0], h[1], h[2], h[3], r);
#pragma unroll
  for(u32 c = 1; c < 6; c++) { r[c] += n[c]; }
}

__device__ inline void store_cum32(u32* cum32, u64 stride, u64 block, const u32 r[6])
{
  cum32[0 * stride + block] = r[1]; cum32[1 * stride + block] = r[2]; cum32[2 * stride + block] = r[3];
  cum32[3 * stride + block] = r[4]; cum32[4 * stride + block] = r[5];
}

// cum32 for the blocks [first, first + count) from block_start and the records (count = 1 at first = nblocks: the entry behind the last block).
__global__ void __launch_bounds__(BLOCK_THREADS) k_block_cum32(const uint4* recs, const u64* block_start, u64 first, u64 count, u32* cum32, u64 stride)
{
  const u64 b = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(b >= count) { return; }
  u32 r[6]; block_cum32_query(recs, block_start[first + b], r);
  store_cum32(cum32, stride, first + b, r);
}

// samples[c] as BWT::build computes them: cum[c * stride + b] = occurrences of c before the start of block first + b, b in [0, count),
// expanded from the compact form (CumulativeArray::sum(first + b) of samples[c], support.h:338-343).
__global__ void __launch_bounds__(BLOCK_THREADS) k_cum_expand(const u64* sup, const u64* block_start, const u32* cum32, u64 cstride, u64 first, u64 count,
  u64* cum, u64 stride)
{
  const u64 b = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(b >= count) { return; }
  const u64 p = block_start[first + b];
  const u64* s = sup + (p >> SUPER_SHIFT) * SUP_STRIDE;
  u64 sum = 0;
#pragma unroll
  for(u32 c = 1; c < 6; c++) { const u64 r = s[c] + cum32[(u64)(c - 1) * cstride + first + b]; cum[c * stride + b] = r; sum += r; }
  cum[0 * stride + b] = p - sum;
}

// The launch covers the segments [seg_first, seg_end): the pipelined download copies the bytes of one range to the
// host while the next range is written.  CUM: also the samples' cumulative counts at every block start, in the compact form cum32 (above);
// whole-index launches only (`recs` is then the index's own record array).
template<bool CUM>
__global__ void __launch_bounds__(BLOCK_THREADS, 4) k_enc_emit(const uint4* recs, u64 nrecs, u64 n, u64 ntiles, u64 seg_first, u64 seg_end,
  const u64* prevhead, u64 head_carry, const u64* seg_base, u8* out, u64* block_start, u32* cum32, u64 cum_stride)
{
  static_assert(BLOCK_THREADS / WAVE * ENC_STAGE_STRIDE <= ENC_DUMP && ENC_STAGE_STRIDE >= 4096 + 32, "staging areas below the dump region");
  __shared__ __attribute__((aligned(16))) u8 lds_all[ENC_LDS_BYTES];   // the kernel's only LDS object: it starts at LDS address 0 (checked below)
  u64 seg = seg_first + (((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6);
  if(seg >= seg_end) { return; }
  u8* const stage_lds = lds_all + (threadIdx.x >> 6) * ENC_STAGE_STRIDE;
  const u32 stage_addr = (u32)(uintptr_t)(__attribute__((address_space(3))) u8*)stage_lds;      // LDS byte address of the wave's staging area
  if(stage_addr != (threadIdx.x >> 6) * ENC_STAGE_STRIDE) { __builtin_trap(); }                  // the dump-region trick needs lds_all at LDS address 0
  u64 first = seg * SEG_TILES;
  u32 carry = (first == 0 ? 0u : symbol_at(recs, (first << 6) - 1));
  u64 last = prevhead[seg];
  if(head_carry > last) { last = head_carry; }
  u64 off = seg_base[seg];           // wave-uniform byte offset
  u32 seg_head = (u32)(off & 15);    // bytes before the segment's first byte in its 16-byte group: they belong to the segment before
  // CUM: occurrences of 1..5 before the current chunk, relative to the super block of the segment (a segment is 2^16 positions and starts at
  // a multiple of that, so it lies inside one super block of 2^25): the header of the segment's first record, then a running total.
  u32 base1 = 0, base2 = 0, base3 = 0, base4 = 0, base5 = 0;          // wave-uniform (scalar registers)
  if(CUM)
  {
    const uint4* r0 = recs + 4 * (first >> 1);                          // tile T lies in record T >> 1
    u32 f[6]; header_fields(r0[0].w, r0[1].w, r0[2].w, r0[3].w, f);
    base1 = (u32)__builtin_amdgcn_readfirstlane((int)f[1]); base2 = (u32)__builtin_amdgcn_readfirstlane((int)f[2]);
    base3 = (u32)__builtin_amdgcn_readfirstlane((int)f[3]); base4 = (u32)__builtin_amdgcn_readfirstlane((int)f[4]);
    base5 = (u32)__builtin_amdgcn_readfirstlane((int)f[5]);
  }
  const u64 seg_super = (first << 6) >> SUPER_SHIFT;
  for(int k = 0; k < SEG_CHUNKS; k++)
  {
    u64 ft = first + (u64)k * 64;
    if(ft >= ntiles) { break; }
    // The bytes of a chunk leave as 16-byte stores at the 16-byte phase of their destination.  The group the chunk ends in is
    // completed by the next chunk of the same wave, so its bytes stay in LDS (moved to the front) instead of being stored one
    // by one and the next chunk's first group is whole; only the first group of a segment (shared with the segment before)
    // and the last one (shared with the next) are written bytewise.  (Per-chunk edge stores were ~10 % of the kernel's instructions.)
    const bool last_chunk = (k == SEG_CHUNKS - 1 || ft + 64 >= ntiles);
    auto flush_chunk = [&](u8* lds, u32 a, u32 total, u64 origin)      // lds[0, total) belongs at out + origin (16-byte aligned); [0, a) precedes this chunk
    {
      (void)a;
      __builtin_amdgcn_wave_barrier();
      u8* base = out + origin;
      const u32 head = seg_head;                                       // bytes of group 0 that are not this segment's (until group 0 has left)
      const u32 whole = (last_chunk ? total : (total & ~15u));         // bytes that leave now
      if(whole != 0) { seg_head = 0; }
      for(u32 j = lane_id(); j * 16 < whole; j += WAVE)
      {
        u32 lo = 16 * j, hi = lo + 16;
        if(lo >= head && hi <= whole) { *(uint4*)(base + lo) = *(const uint4*)(lds + lo); }
        else
        {
          u32 from = (lo > head ? lo : head), to = (hi < whole ? hi : whole);
          for(u32 t = from; t < to; t++) { base[t] = lds[t]; }
        }
      }
      __builtin_amdgcn_wave_barrier();
      if(!last_chunk && whole != 0)
      {
        const u32 r = total - whole;                                   // 0 .. 15 bytes stay for the next chunk
        u8 keep = 0;
        if(lane_id() < r) { keep = lds[whole + lane_id()]; }
        __builtin_amdgcn_wave_barrier();
        if(lane_id() < r) { lds[lane_id()] = keep; }
        __builtin_amdgcn_wave_barrier();
      }
    };
    TileInfo ti;
    // (Requesting the next chunk's planes here, while this one is worked on, is SLOWER: 4.78 -> 5.08 ms at config 2, round 5 -- the six
    // registers push the kernel over 128 and the compiler spills lane constants that every chunk then reloads.)
    carry = chunk_tiles(recs, nrecs, ft, n, carry, ti);
    u64 T = ft + lane_id();
    if(T >= ntiles) { ti.H = 0; ti.E = 0; }
    u64 lh = (ti.H != 0 ? (T << 6) + (63 - (u64)__builtin_clzll(ti.H)) + 1 : NONE);
    u64 incl = wave_incl_last(lh);                                   // lh grows with the lane where it is set: the inclusive maximum is the last one so far
    u64 before = shfl_up_u64(incl, 1);
    if(lane_id() == 0) { before = NONE; }
    if(last > before) { before = last; }
    u32 nev; u64 long_mask, dep_len;
    tile_event_stats(ti, T << 6, before, nev, long_mask, dep_len);
    u64 ev_incl = (u64)wave_incl_sum32(nev);
    u64 chunk_events = shfl_u64(ev_incl, WAVE - 1);
    bool slow = (__ballot(long_mask != 0) != 0);
    // CUM: occurrences of 1..5 before this lane's tile = running base + exclusive wave prefix of the tiles' counts (<= 4096 per wave: two per register)
    u32 ex12 = 0, ex34 = 0, ex5 = 0, tot12 = 0, tot34 = 0, tot5 = 0;
    if(CUM)
    {
      const u64 tb0 = T << 6;
      const u64 valid = (n >= tb0 + 64 ? ~0ull : (n <= tb0 ? 0ull : ((1ull << (n - tb0)) - 1)));
      u32 n1, n2, n3, n4, n5; tile_counts(ti.p0, ti.p1, ti.p2, valid, n1, n2, n3, n4, n5);
      const u32 own12 = n1 | (n2 << 16), own34 = n3 | (n4 << 16);
      const u32 in12 = wave_incl_sum32(own12), in34 = wave_incl_sum32(own34), in5 = wave_incl_sum32(n5);
      ex12 = in12 - own12; ex34 = in34 - own34; ex5 = in5 - n5;
      tot12 = (u32)__builtin_amdgcn_readlane((int)in12, WAVE - 1); tot34 = (u32)__builtin_amdgcn_readlane((int)in34, WAVE - 1);
      tot5 = (u32)__builtin_amdgcn_readlane((int)in5, WAVE - 1);       // added to the running base at the end of the chunk
    }
    // Counts before position tile_start + bit (bit = the in-tile position of a head, 0..63) minus, for the symbol `sym` of the run that ends
    // there, the part of that run behind position p: the five cum32 values of a block that starts at p (see above).
    auto store_cum_at = [&](u64 blk, u64 p, u32 bit, u32 sym)
    {
      const u64 tb0 = T << 6;
      u32 r[6];
      if((p >> SUPER_SHIFT) != seg_super) { block_cum32_query(recs, p, r); }       // the run began in an earlier super block: one rank query
      else
      {
        u32 n1, n2, n3, n4, n5; tile_counts(ti.p0, ti.p1, ti.p2, (bit == 0 ? 0ull : (~0ull >> (64 - bit))), n1, n2, n3, n4, n5);
        const u32 back = (u32)(tb0 + bit - p);                            // positions of the run between p and the head
        r[1] = base1 + (ex12 & 0xFFFF) + n1 - (sym == 1 ? back : 0u);
        r[2] = base2 + (ex12 >> 16) + n2 - (sym == 2 ? back : 0u);
        r[3] = base3 + (ex34 & 0xFFFF) + n3 - (sym == 3 ? back : 0u);
        r[4] = base4 + (ex34 >> 16) + n4 - (sym == 4 ? back : 0u);
        r[5] = base5 + ex5 + n5 - (sym == 5 ? back : 0u);
      }
      store_cum32(cum32, cum_stride, blk, r);
    };
    if(!slow)
    {
      // Every event is a run shorter than 42: one byte each, in position order.  The bytes are
      // staged in LDS at the same 16-byte phase as their destination and leave as 16-byte stores.
      u8* lds = stage_lds;
      const u32 a = (u32)(off & 15);
      const u32 idx0 = a + (u32)(ev_incl - nev);                          // staging index of the tile's first byte
      {
        // The run that ends at head b has the symbol found at the previous head and the length b - (previous head).  The tile is
        // walked as two 32-bit halves (walk_half): the run open at the start of the low half is the one that was open before the tile
        // (symbol ti.prev, begun at the last head before the tile) -- unless the tile starts at position 0, a head without an event --
        // and the run open at the start of the high half has the symbol at bit 31 and began at the last head of the low half, if any.
        const u64 tb = T << 6;
        const u32 h_lo = (u32)ti.H, e_lo = (u32)ti.E, e_hi = (u32)(ti.E >> 32);
        const bool at_zero = (tb == 0 && (h_lo & 1u) != 0);
        const int prev_lo = (at_zero ? 0 : (int)(u32)(before - 1 - tb));  // in-tile position of the last head before the tile's first event (negative: before the tile)
        const u32 sym_lo = (at_zero ? ((u32)(ti.p0 & 1) | ((u32)(ti.p1 & 1) << 1) | ((u32)(ti.p2 & 1) << 2)) : ti.prev);
        const int prev_hi = (h_lo != 0 ? 31 - (int)__builtin_clz(h_lo) : prev_lo) - 32;
        const u32 sym_hi = (u32)((ti.p0 >> 31) & 1) | ((u32)((ti.p1 >> 31) & 1) << 1) | ((u32)((ti.p2 >> 31) & 1) << 2);
        const u32 n_lo = (u32)__builtin_popcount(e_lo), n_hi = nev - n_lo;
        const u32 trips_lo = (wave_max32(n_lo) + (ENC_WALK - 1)) & ~(ENC_WALK - 1), trips_hi = (wave_max32(n_hi) + (ENC_WALK - 1)) & ~(ENC_WALK - 1);
        walk_half(e_lo, (u32)ti.p0, (u32)ti.p1, (u32)ti.p2, prev_lo, sym_lo, stage_addr + idx0, trips_lo);
        walk_half(e_hi, (u32)(ti.p0 >> 32), (u32)(ti.p1 >> 32), (u32)(ti.p2 >> 32), prev_hi, sym_hi, stage_addr + idx0 + n_lo, trips_hi);
        // A tile emits at most 64 bytes, so at most one of them opens a 64-byte block: event number `kopen` of the tile, if it has that many.
        // The run of that event starts at the head before it (found after the walk: tracking it inside cost two instructions per event).
        const u32 phase = (u32)((off - a) & (RLE_BLOCK - 1));            // byte (off - a + idx) opens a block iff ((phase + idx) & 63) == 0
        const u32 kopen = (0u - (phase + idx0)) & (u32)(RLE_BLOCK - 1);
        if(kopen < nev)
        {
          const u32 ebit = select64(ti.E, kopen);
          const u64 hb = ti.H & ((1ull << ebit) - 1);                     // heads before the event
          const int open_prev = (hb != 0 ? 63 - (int)__builtin_clzll(hb) : prev_lo);
          const u64 blk = (off - a + idx0 + kopen) >> 6, p = tb + (u64)(long long)open_prev;
          block_start[blk] = p;
          // the run that opens the block starts at a head of this tile (nothing of it lies behind p), or it is the run that ends at
          // the tile's first head and began before the tile (its symbol is the one before the tile)
          if(CUM) { store_cum_at(blk, p, (open_prev >= 0 ? (u32)open_prev : ebit), ti.prev); }
        }
      }
      flush_chunk(lds, a, a + (u32)chunk_events, off - a);
      off += chunk_events;
    }
    else
    {
      // Some runs of >= 42 end in this chunk.  The sizes of the longest ones depend on their byte offsets, so those are resolved in order (a
      // handful per chunk); every other event sits at (its event index + the extra bytes of the long events before it).
      // Round 6: a long event is always the FIRST event of its 32-position half of the tile (a head that ends a run of >= 42 has no head among
      // the 41 positions before it; if it is not the tile's first head, the head before it lies >= 42 positions back, in the low half, and it
      // lies in the high half itself).  So a half is [one long event] + one-byte events: the long event is written by the lane on its own and the
      // one-byte events go through walk_half like those of a chunk without long runs, from the state behind the long event.  Until then the lanes
      // walked ALL heads of a slow chunk one by one in a divergent loop of three cases: on reads of a genome -- a long run in nearly every
      // chunk -- k_enc_emit took 10 ms for a stream of 2 GB where the 7.6 GB of an iid stream take 4.7.
      u8* lds = stage_lds;
      const u32 a = (u32)(off & 15);
      const u64 origin = off - a;                                      // stream offset of lds[0]
      const u32 ev_excl = (u32)(ev_incl - nev);
      // extra bytes of all long events / of those in earlier tiles: one per run of 42 .. 82 (a wave scan), and what the ordered walk over the
      // longer runs adds (see tile_event_stats)
      const u32 two = (u32)__builtin_popcountll(long_mask) - (dep_len != 0 ? 1u : 0u);
      const u32 two_incl = wave_incl_sum32(two);
      u32 extra = (u32)__builtin_amdgcn_readlane((int)two_incl, WAVE - 1), shift = two_incl - two;
      {
        u32 dep_extra = 0;
        for_each_dep_event(dep_len, ev_excl + shift, [&](u32 t, u32 g, u64 len)
        {
          const u32 sz = (u32)long_run_bytes(off + g + dep_extra, len);
          if(lane_id() > t) { shift += sz - 1; }
          dep_extra += sz - 1;
        });
        extra += dep_extra;
      }
      const u64 tb = T << 6;
      const u32 idx0 = a + ev_excl + shift;                             // staging index of the tile's first byte
      const u32 h_lo = (u32)ti.H, e_lo = (u32)ti.E, e_hi = (u32)(ti.E >> 32);
      const u32 long_lo = (u32)long_mask, long_hi = (u32)(long_mask >> 32);
      const bool at_zero = (tb == 0 && (h_lo & 1u) != 0);
      const int prev_lo0 = (at_zero ? 0 : (int)(u32)(before - 1 - tb));  // as in the branch above
      int prev_lo = prev_lo0;
      u32 sym_lo = (at_zero ? ((u32)(ti.p0 & 1) | ((u32)(ti.p1 & 1) << 1) | ((u32)(ti.p2 & 1) << 2)) : ti.prev);
      int prev_hi = (h_lo != 0 ? 31 - (int)__builtin_clz(h_lo) : prev_lo0) - 32;
      u32 sym_hi = (u32)((ti.p0 >> 31) & 1) | ((u32)((ti.p1 >> 31) & 1) << 1) | ((u32)((ti.p2 >> 31) & 1) << 2);
      // The long event at in-tile bit b, whose run began at position prev1 - 1, written at staging index idx; returns its bytes.
      auto write_long = [&](u32 b, u64 prev1, u32 idx, bool dep) -> u32
      {
        const u32 run_sym = event_symbol(ti, b);
        const u64 len = tb + b + 1 - prev1;
        if(dep)
        {
          return (u32)long_run_write_cb(lds, origin + idx, run_sym, len, prev1 - 1, origin, [&](u64 blk, u64 p)
          {
            block_start[blk] = p;
            if(CUM) { store_cum_at(blk, p, b, run_sym); }
          });
        }
        // a run of 42 .. 82: head (basic length 42) + one extension byte, or -- when the head is the last byte of its block -- a run of 41
        // and a run of length - 41 that opens the next block (support.h:267-279; long_run_write_cb for these lengths, written out)
        const u64 at = origin + idx;
        const bool edge = ((at & (RLE_BLOCK - 1)) == RLE_BLOCK - 1);
        if((at & (RLE_BLOCK - 1)) == 0)
        {
          block_start[at >> 6] = prev1 - 1;
          if(CUM) { store_cum_at(at >> 6, prev1 - 1, b, run_sym); }
        }
        lds[idx] = (u8)(run_sym + 6 * (edge ? MAX_RUN - 2 : MAX_RUN - 1));
        lds[idx + 1] = (u8)(edge ? run_sym + 6 * (len - MAX_RUN) : len - MAX_RUN);
        if(edge)
        {
          block_start[(at + 1) >> 6] = prev1 - 1 + (MAX_RUN - 1);
          if(CUM) { store_cum_at((at + 1) >> 6, prev1 - 1 + (MAX_RUN - 1), b, run_sym); }
        }
        return 2u;
      };
      u32 w_lo = e_lo, w_hi = e_hi;                                    // the one-byte events of the halves
      u32 sz_lo = 0, sz_hi = 0;                                        // bytes of the halves' long events
      if(long_lo != 0)
      {
        // the tile's first head (and first event): its run began before the tile
        const u32 b = (u32)__builtin_ctz(long_lo);
        sz_lo = write_long(b, before, idx0, dep_len != 0);
        w_lo = e_lo & ~long_lo; prev_lo = (int)b;
        sym_lo = (u32)((ti.p0 >> b) & 1) | ((u32)((ti.p1 >> b) & 1) << 1) | ((u32)((ti.p2 >> b) & 1) << 2);
      }
      const u32 ns_lo = (u32)__builtin_popcount(w_lo);
      const u32 bytes_lo = sz_lo + ns_lo;
      if(long_hi != 0)
      {
        // the first event of the high half: its run began at the last head of the low half, or before the tile when that half has none
        const u32 b = 32u + (u32)__builtin_ctz(long_hi);
        const u64 prev1 = (h_lo != 0 ? tb + (u64)(31 - (int)__builtin_clz(h_lo)) + 1 : before);
        sz_hi = write_long(b, prev1, idx0 + bytes_lo, dep_len != 0 && h_lo == 0);
        w_hi = e_hi & ~long_hi; prev_hi = (int)b - 32;
        sym_hi = (u32)((ti.p0 >> b) & 1) | ((u32)((ti.p1 >> b) & 1) << 1) | ((u32)((ti.p2 >> b) & 1) << 2);
      }
      const u32 ns_hi = (u32)__builtin_popcount(w_hi);
      const u32 trips_lo = (wave_max32(ns_lo) + (ENC_WALK - 1)) & ~(ENC_WALK - 1), trips_hi = (wave_max32(ns_hi) + (ENC_WALK - 1)) & ~(ENC_WALK - 1);
      walk_half(w_lo, (u32)ti.p0, (u32)ti.p1, (u32)ti.p2, prev_lo, sym_lo, stage_addr + idx0 + sz_lo, trips_lo);
      walk_half(w_hi, (u32)(ti.p0 >> 32), (u32)(ti.p1 >> 32), (u32)(ti.p2 >> 32), prev_hi, sym_hi, stage_addr + idx0 + bytes_lo + sz_hi, trips_hi);
      // The one-byte events that open a 64-byte block (the long events have looked after theirs): the tile's bytes are
      // [long event of the low half][its one-byte events][long event of the high half][its one-byte events], at most two block starts among them.
      {
        const u32 tile_bytes = bytes_lo + sz_hi + ns_hi;
        const u32 phase = (u32)(origin & (RLE_BLOCK - 1));               // byte (origin + idx) opens a block iff ((phase + idx) & 63) == 0
        const u64 shorts = (u64)w_lo | ((u64)w_hi << 32);
        for(u32 k = (0u - (phase + idx0)) & (u32)(RLE_BLOCK - 1); k < tile_bytes; k += (u32)RLE_BLOCK)
        {
          u32 rank;
          if(k < sz_lo) { continue; }
          else if(k < bytes_lo) { rank = k - sz_lo; }
          else if(k < bytes_lo + sz_hi) { continue; }
          else { rank = k - sz_lo - sz_hi; }
          const u32 ebit = select64(shorts, rank);
          const u64 hb = ti.H & ((1ull << ebit) - 1);                     // heads before the event
          const int open_prev = (hb != 0 ? 63 - (int)__builtin_clzll(hb) : prev_lo0);
          const u64 blk = (origin + idx0 + k) >> 6, p = tb + (u64)(long long)open_prev;
          block_start[blk] = p;
          if(CUM) { store_cum_at(blk, p, (open_prev >= 0 ? (u32)open_prev : ebit), ti.prev); }
        }
      }
      flush_chunk(lds, a, a + (u32)chunk_events + extra, origin);
      off += chunk_events + extra;
    }
    u64 m = shfl_u64(incl, WAVE - 1);
    if(m > last) { last = m; }
    if(CUM) { base1 += tot12 & 0xFFFF; base2 += tot12 >> 16; base3 += tot34 & 0xFFFF; base4 += tot34 >> 16; base5 += tot5; }
  }
}

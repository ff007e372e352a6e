/*
  kernels/transcode.hip.h -- native byte stream / plain symbols -> device rank structure, block samples.
  Part of bwtm_kernels.hip.h (included there, inside namespace bwtm); gfx950 only.
*/
#pragma once

//------------------------------------------------------------------------------
// K0: native byte stream -> device rank structure (BWT::load + BWT::build, bwt.cpp:132-148, 476-512).
//
// The stream is cut into GROUPs of 62 consecutive 64-byte blocks; one wave owns one group and one
// lane decodes one block, so the byte stream is read exactly once per kernel with coalesced loads.
//
//   k_block_len   : positions per block, and per group of 62 blocks its positions and symbol counts (-> exclusive scans over the
//                   GROUPS: 1 / 62 of the blocks; the block starts -- the set bits of block_boundaries, bwt.cpp:496 -- are written by
//                   k_build_recs, whose waves add a scan of their own 64 block lengths to the group's start.  Until round 4 the lengths
//                   of ALL blocks were scanned: three passes over 8 bytes per block, 0.7 ms per merge at config 2)
//   k_build_sup   : absolute counts at the super boundaries
//   k_build_recs  : the records
//   k_block_cum   : cumulative symbol counts at the block starts (samples[c], bwt.cpp:489-511),
//                   read back from the finished rank structure
//
// Every full block of a stream written by Run::write encodes at least 64 positions (a run of k bytes
// is at least k long, support.h:256-282); k_block_len verifies this and the other kernels rely on it.

constexpr int GROUP = 62;                 // blocks owned by one wave; 2 more are staged as lookahead
constexpr int STAGE_WORDS = 17;           // LDS row stride of a staged block: conflict-free 32-bit reads
constexpr int STAGE_ROWS = 64;

__device__ inline void wave_sync_lds()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Stages blocks [first, first + count), count <= 64, into rows of STAGE_WORDS words (coalesced 16-byte loads).
// Bytes past the end of the stream are staged as zeros (= runs of one endmarker, which deposit no bits).
// The four loads of a lane are unconditional (from clamped chunk indexes) and leave together; with a predicate per load the
// compiler waited for each before it issued the next: four memory round trips in a row at the start of every wave.
__device__ inline void stage_blocks(const u8* data, u64 nbytes, u64 first, u32 count, u32* rows)
{
  const u32 lane = lane_id();
  const uint4* src = (const uint4*)(data + first * RLE_BLOCK);
  const u64 left = nbytes - first * RLE_BLOCK;                        // bytes of the stream from `first` on (> 0: the caller has blocks)
  const u64 chunks_avail = (left + 15) / 16;                          // the buffer is readable up to the next multiple of 16
  uint4 v[4];
#pragma unroll
  for(int k = 0; k < 4; k++)
  {
    const u64 g = (u64)k * 64 + lane;
    v[k] = src[g < chunks_avail ? g : chunks_avail - 1];
  }
#pragma unroll
  for(int k = 0; k < 4; k++)
  {
    const u32 g = (u32)k * 64 + lane;
    if(g < 4 * count)
    {
      uint4 x = v[k];
      if(g >= chunks_avail) { x = make_uint4(0, 0, 0, 0); }
      else if((u64)16 * g + 16 > left)                                // the chunk that holds the last byte
      {
        const u32 keep = (u32)(left - (u64)16 * g);                    // 1..15 bytes
        u32 m[4];
#pragma unroll
        for(u32 j = 0; j < 4; j++) { m[j] = (keep >= 4 * j + 4 ? ~0u : (keep <= 4 * j ? 0u : (1u << (8 * (keep - 4 * j))) - 1u)); }
        x.x &= m[0]; x.y &= m[1]; x.z &= m[2]; x.w &= m[3];
      }
      u32* dst = rows + (g >> 2) * STAGE_WORDS + (g & 3) * 4;
      dst[0] = x.x; dst[1] = x.y; dst[2] = x.z; dst[3] = x.w;
    }
  }
}

// Walks the runs of a staged block in order: on_short(sym, len) for runs of 1..41 (one byte, the common
// case, kept straight-line), on_long(sym, len) for runs with a varint extension (support.h:236-250).
// A byte-wise state machine without dynamic indexing.  CHECK_VALID: only the first `valid` bytes
// belong to the stream and a run cut off there is dropped; otherwise all 64 bytes are decoded (bytes
// staged past the end of the stream are zeros).
template<bool CHECK_VALID, class FS, class FL>
__device__ inline void for_each_run(const u32* row, u32 valid, FS&& on_short, FL&& on_long)
{
  u32 sym = 0, shift = 0; u64 len = 0; bool cont = false;
#pragma unroll 1
  for(int w = 0; w < 16; w++)
  {
    const u32 word = row[w];
    // bytes >= 246 (heads of runs with a varint extension): high bit set and low 7 bits >= 0x76
    const u32 long_heads = ((word & 0x7F7F7F7Fu) + 0x0A0A0A0Au) & word & 0x80808080u;
    if(!cont && long_heads == 0 && (!CHECK_VALID || (u32)(4 * w + 3) < valid))
    {
      // four one-byte runs: no state, no branches
#pragma unroll
      for(int k = 0; k < 4; k++)
      {
        const u32 byte = (word >> (8 * k)) & 0xFF;
        const u32 q = (byte * 171u) >> 10;                           // q = byte / 6, exact for byte < 256
        on_short(byte - 6 * q, q + 1);
      }
      continue;
    }
#pragma unroll
    for(int k = 0; k < 4; k++)
    {
      if(!CHECK_VALID || (u32)(4 * w + k) < valid)
      {
        u32 byte = (word >> (8 * k)) & 0xFF;
        if(cont)
        {
          len += (u64)(byte & 0x7F) << shift; shift += 7; cont = (byte & 0x80) != 0;
          if(!cont) { on_long(sym, len); }
        }
        else
        {
          u32 q = (byte * 171u) >> 10; sym = byte - 6 * q;
          if(q + 1 >= MAX_RUN) { len = q + 1; shift = 0; cont = true; }
          else { on_short(sym, q + 1); }
        }
      }
    }
  }
}

// blen[b] = positions encoded by block b; gcount[c * gstride + g] = occurrences of c in group g (c = 0..5), gcount[6 * gstride + g] = positions of group g.
// flags bit 0: a block other than the last one encodes fewer than 64 positions.
// The launch covers the groups [group_first, group_end): the pipelined upload runs one launch per H2D chunk.
// One-byte runs (the common case) are counted through a 256-entry LDS table: entry of byte v < 246 = (v / 6 + 1) << (10 (v % 6)),
// six 10-bit fields, one per symbol; a lane adds the entries of up to 16 bytes (16 x 41 < 1024) before it unpacks the fields.
// (The kernel was VALU-bound at ~12 instructions per byte; the table leaves ~5.)
__global__ void __launch_bounds__(BLOCK_THREADS) k_block_len(const u8* data, u64 nbytes, u64 nblocks, u64 group_first, u64 group_end,
  u64* blen, u64* gcount, u64 gstride, u32* flags)
{
  __shared__ u32 stage[BLOCK_THREADS / WAVE][STAGE_ROWS * STAGE_WORDS];
  __shared__ u64 tab[256];
  {
    const u32 v = threadIdx.x, q = (v * 171u) >> 10;
    tab[v] = (v < 246 ? (u64)(q + 1) << (10 * (v - 6 * q)) : 0ull);
  }
  __syncthreads();
  const u32 lane = lane_id(), wave = threadIdx.x >> 6;
  const u64 g = group_first + (u64)blockIdx.x * (BLOCK_THREADS / WAVE) + wave;
  if(g >= group_end) { return; }
  const u64 first = g * GROUP;
  const u32 nb = (nblocks > first ? (nblocks - first > (u64)GROUP ? (u32)GROUP : (u32)(nblocks - first)) : 0u);
  u32* rows = stage[wave];
  if(nb > 0) { stage_blocks(data, nbytes, first, nb, rows); }
  wave_sync_lds();
  u64 l0 = 0, l1 = 0, l2 = 0, l3 = 0, l4 = 0, l5 = 0;
  const u64 b = first + lane;
  if(lane < nb)
  {
    const u64 begin = b * RLE_BLOCK;
    const u32 valid = (nbytes - begin >= RLE_BLOCK ? (u32)RLE_BLOCK : (u32)(nbytes - begin));
    const u32* row = rows + lane * STAGE_WORDS;
    u64 acc = 0;
    u32 sym = 0, shift = 0; u64 len = 0; bool cont = false;
    auto add_long = [&](u32 s, u64 n)
    {
      l0 += (s == 0 ? n : 0); l1 += (s == 1 ? n : 0); l2 += (s == 2 ? n : 0);
      l3 += (s == 3 ? n : 0); l4 += (s == 4 ? n : 0); l5 += (s == 5 ? n : 0);
    };
    auto unpack = [&]()
    {
      l0 += acc & 1023; l1 += (acc >> 10) & 1023; l2 += (acc >> 20) & 1023;
      l3 += (acc >> 30) & 1023; l4 += (acc >> 40) & 1023; l5 += (acc >> 50) & 1023;
      acc = 0;
    };
#pragma unroll 1
    for(int w = 0; w < 16; w++)
    {
      const u32 word = row[w];
      // bytes >= 246 (heads of runs with a varint extension): high bit set and low 7 bits >= 0x76
      const u32 long_heads = ((word & 0x7F7F7F7Fu) + 0x0A0A0A0Au) & word & 0x80808080u;
      if(!cont && long_heads == 0 && (u32)(4 * w + 3) < valid)
      {
        acc += tab[word & 0xFF] + tab[(word >> 8) & 0xFF] + tab[(word >> 16) & 0xFF] + tab[word >> 24];
#ifdef BWTM_SLACK_BLOCK_LEN
        { u32 slack = word; valu_slack<BWTM_SLACK_BLOCK_LEN>(slack); }
#endif
      }
      else
      {
#pragma unroll
        for(int k = 0; k < 4; k++)
        {
          if((u32)(4 * w + k) < valid)
          {
            const u32 byte = (word >> (8 * k)) & 0xFF;
            if(cont)
            {
              len += (u64)(byte & 0x7F) << shift; shift += 7; cont = (byte & 0x80) != 0;
              if(!cont) { add_long(sym, len); }
            }
            else if(byte >= 246) { sym = byte - 6 * ((byte * 171u) >> 10); len = MAX_RUN; shift = 0; cont = true; }
            else { acc += tab[byte]; }
          }
        }
      }
      if((w & 3) == 3) { unpack(); }
    }
    const u64 total = l0 + l1 + l2 + l3 + l4 + l5;
    blen[b] = total;
    if(total < RLE_BLOCK && b + 1 < nblocks) { atomicOr(flags, 1u); }
  }
  // per-lane counts are at most 64 x 41 unless the block holds a long run: the DPP path almost always
  u64 t0 = wave_sum_mostly_small(l0), t1 = wave_sum_mostly_small(l1), t2 = wave_sum_mostly_small(l2), t3 = wave_sum_mostly_small(l3), t4 = wave_sum_mostly_small(l4), t5 = wave_sum_mostly_small(l5);
  if(lane == 0)
  {
    gcount[0 * gstride + g] = t0; gcount[1 * gstride + g] = t1; gcount[2 * gstride + g] = t2;
    gcount[3 * gstride + g] = t3; gcount[4 * gstride + g] = t4; gcount[5 * gstride + g] = t5;
    gcount[6 * gstride + g] = t0 + t1 + t2 + t3 + t4 + t5;
  }
}

// block_end[b] = block_start[b + 1] - 1 (the set bits of block_boundaries, bwt.cpp:496).
__global__ void __launch_bounds__(BLOCK_THREADS) k_block_end(const u64* block_start, u64 first, u64 count, u64* block_end)
{
  u64 b = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(b < count) { block_end[b] = block_start[first + b + 1] - 1; }
}

// Largest g in [0, ngroups) with gpos[g] <= p (p < n; gpos = exclusive scan of the groups' positions, gpos[ngroups] = n).
__device__ inline u64 find_group(const u64* gpos, u64 ngroups, u64 p)
{
  u64 lo = 0, hi = ngroups;           // invariant: gpos[lo] <= p < gpos[hi]
  while(hi - lo > 1)
  {
    u64 mid = (lo + hi) >> 1;
    if(gpos[mid] <= p) { lo = mid; } else { hi = mid; }
  }
  return lo;
}

// Start position of the lane's block of the 64 blocks from `first` on (the group's 62 and the two lookahead blocks): the group's start + an
// exclusive wave scan of the block lengths (the 32-bit DPP scan unless some block of the wave holds 2^25 positions or more).  len_out = the lane's own length.
__device__ inline u64 group_block_start(const u64* blen, u64 nblocks, u64 first, u64 group_start, u64& len_out)
{
  const u64 b = first + lane_id();
  const u64 len = (b < nblocks ? blen[b] : 0);
  len_out = len;
  u64 incl;
  if(__ballot((len >> 25) != 0) == 0) { incl = (u64)wave_incl_sum32((u32)len); }
  else { incl = wave_incl_sum(len); }
  return group_start + incl - len;
}

// Super table from the native stream: one wave per super.  The counts at position p are the counts at
// the start of p's group plus the runs of the group's blocks before p (one lane per block).
__global__ void __launch_bounds__(BLOCK_THREADS) k_build_sup(const u8* data, u64 nbytes, const u64* blen,
  const u64* gcum, u64 gstride, u64 nblocks, u64 ngroups, u64 n, u64* sup, u64 nsup)
{
  const u32 lane = lane_id();
  const u64 s = ((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6;
  if(s >= nsup) { return; }
  const u64 p = s << SUPER_SHIFT;
  u64 g = ngroups;                                  // column of the totals
  u64 c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
  if(p < n)
  {
    const u64* gpos = gcum + 6 * gstride;
    g = find_group(gpos, ngroups, p);               // wave-uniform
    u64 len;
    const u64 start = group_block_start(blen, nblocks, g * GROUP, gpos[g], len);
    const u64 blk = g * GROUP + lane;
    if(lane < (u32)GROUP && blk < nblocks && start <= p)
    {
      u64 pos = start, rle = blk * RLE_BLOCK;
      const u64 end = (nbytes - rle >= RLE_BLOCK ? rle + RLE_BLOCK : nbytes);
      while(rle < end && pos < p)
      {
        u32 sym; u64 len; run_decode(data, rle, sym, len);
        u64 take = (p - pos < len ? p - pos : len);
        c1 += (sym == 1 ? take : 0); c2 += (sym == 2 ? take : 0); c3 += (sym == 3 ? take : 0);
        c4 += (sym == 4 ? take : 0); c5 += (sym == 5 ? take : 0);
        pos += len;
      }
    }
  }
  c1 = wave_sum(c1); c2 = wave_sum(c2); c3 = wave_sum(c3); c4 = wave_sum(c4); c5 = wave_sum(c5);
  if(lane == 0)
  {
    u64* out = sup + s * SUP_STRIDE;
    out[0] = 0; out[6] = 0; out[7] = 0;
    out[1] = gcum[1 * gstride + g] + c1; out[2] = gcum[2 * gstride + g] + c2; out[3] = gcum[3 * gstride + g] + c3;
    out[4] = gcum[4 * gstride + g] + c4; out[5] = gcum[5 * gstride + g] + c5;
  }
}

// Records from the native stream.  The wave of group g owns the records that START inside the group's
// position range [S, E) (the last group also owns the rest); they may extend up to 127 positions into
// the next group, which the two lookahead blocks cover.  Positions are processed in windows of
// BR_WINDOW: every lane deposits the runs of its block into three LDS bit-planes (word-wise OR,
// accumulated in registers while consecutive runs stay inside one word), then lane r assembles record
// r of the window: planes from LDS, header = counts before the group + carried counts of the earlier
// windows + wave prefix of the records' own counts.  One window is the common case (a group of
// random-read BWT covers ~5300 positions); compressible streams take more windows over fewer bytes.
// The window is a template parameter: 8192 positions for streams around the iid density, larger windows
// (fewer waves per workgroup) for compressible streams whose groups cover more positions.
// FILL (for streams of long runs): the whole plane words in the middle of a long run are not deposited by the lane
// that decodes the run -- a serial loop of length / 32 steps on a mostly idle wave -- but queued in LDS and written
// by all 64 lanes after the decode pass.  It costs ~1 ms on read-like streams, hence a template flag.
// UNIFORM (for streams from ~3.5 positions per byte on, round 6): the blocks inside the window are deposited by straight-line code, the same for
// every byte class, instead of the word path / per-byte paths that serve read-like streams (see there).
constexpr u32 BR_FILLS = 256;

// (Before UNIFORM, round 6 measured a RUN-parallel deposit for the streams of longer runs -- every lane stores the window-relative end of each byte's run, then
// the group's 64 x 64 bytes are dealt out round-robin and every run ORs its own edge words into the planes: no accumulators, no divergent
// flushes.  It lost: 14.8 vs 12.1 ms at 300 x genome reads, 13.6 vs 8.2 at 30 x; four times the LDS atomics at 1.5 waves per SIMD.  So did
// smaller windows with more waves: 16 384 / 8 192 positions 17.2 / 21.0 vs 12.1 ms.  And so did a GATHER per record: pass 1 stores for every byte
// the position its run ends at (64 x 64 entries per wave in LDS), pass 2 gives every lane a record: a binary search for its first entry, then a
// walk that builds the record's twelve words in registers, one mask per run and word -- no atomics, no planes in LDS.  Bit-exact in all parity
// tests, and slower everywhere: 13.0 vs 12.3 ms at 300 x, 12.9 vs 8.2 at 30 x, 8.0 vs 1.8 on iid reads (2 x 20 M): the walk is a loop of
// dependent LDS reads whose trip count differs from lane to lane, at 2 waves per SIMD.  DESIGN_HISTORY.md, round 6.)
template<u32 BR_WINDOW, int WAVES, bool FILL, bool UNIFORM = false>
__global__ void __launch_bounds__(WAVES * WAVE) k_build_recs(const u8* data, u64 nbytes, const u64* blen, u64* block_start,
  const u64* gcum, u64 gstride, u64 nblocks, u64 ngroups, u64 n, const u64* sup, uint4* recs, u64 nrecs)
{
  constexpr u32 PW = BR_WINDOW / 32;                             // words per plane
  __shared__ u32 stage[WAVES][STAGE_ROWS * STAGE_WORDS];
  __shared__ uint4 planes[WAVES][3][BR_WINDOW / 128];
  __shared__ u32 fill_list[WAVES][FILL ? BR_FILLS : 1];
  __shared__ u32 fill_count[WAVES];
  const u32 lane = lane_id(), wave = threadIdx.x >> 6;
  const u64 g = (u64)blockIdx.x * WAVES + wave;
  if(g >= ngroups) { return; }
  const u64 first = g * GROUP;
  const bool last_group = (g + 1 == ngroups);
  const u32 nb = (nblocks > first ? (nblocks - first > (u64)STAGE_ROWS ? (u32)STAGE_ROWS : (u32)(nblocks - first)) : 0u);
  const u64* gpos = gcum + 6 * gstride;                          // start position of every group (gpos[ngroups] = n)
  const u64 S = gpos[g];
  // the starts of the wave's 64 blocks: written out for the group's own 62 (block_start: the set bits of block_boundaries, bwt.cpp:496, plus one)
  u64 blen_own;
  const u64 bstart_all = group_block_start(blen, nblocks, first, S, blen_own);
  if(lane < (u32)GROUP && first + lane < nblocks) { block_start[first + lane] = bstart_all; }
  if(last_group && lane == 0) { block_start[nblocks] = gpos[ngroups]; }
  const u64 q_lo = (S + REC_POS - 1) >> REC_SHIFT;
  const u64 q_hi = (last_group ? nrecs : (gpos[g + 1] + REC_POS - 1) >> REC_SHIFT);
  if(q_lo >= q_hi) { return; }                                  // wave-uniform: no record starts in this group
  u32* rows = stage[wave];
  if(nb > 0) { stage_blocks(data, nbytes, first, nb, rows); }
  const bool have = (lane < nb);
  const u64 b = first + lane;
  const u64 bstart = (have ? bstart_all : 0), bend = (have ? bstart_all + blen_own : 0);
  const u32 valid = (have ? (nbytes - b * RLE_BLOCK >= RLE_BLOCK ? (u32)RLE_BLOCK : (u32)(nbytes - b * RLE_BLOCK)) : 0u);
  u64 a1 = gcum[1 * gstride + g], a2 = gcum[2 * gstride + g], a3 = gcum[3 * gstride + g],
      a4 = gcum[4 * gstride + g], a5 = gcum[5 * gstride + g];   // counts before the first record of the window
  const u64 pos_end = ((q_hi << REC_SHIFT) < n ? (q_hi << REC_SHIFT) : n);
  u32* pl = (u32*)planes[wave];                                 // plane k: words [PW k, PW k + PW)
  for(u64 ws = S & ~(u64)(REC_POS - 1); (ws >> REC_SHIFT) < q_hi; ws += BR_WINDOW)
  {
#pragma unroll
    for(u32 k = 0; k < 3 * PW / WAVE; k++) { pl[k * 64 + lane] = 0; }
    if(FILL && lane == 0) { fill_count[wave] = 0; }
    wave_sync_lds();
    const u64 we = (ws + BR_WINDOW < pos_end ? ws + BR_WINDOW : pos_end);
    const bool inside = (have && bstart >= ws && bend <= ws + BR_WINDOW);
    // Queues `nwords` whole words from `word` on for the cooperative fill; returns the number of words taken over
    // (0 if the queue is full: the caller then deposits them itself).
    auto queue_fill = [&](u32 sym, u32 word, u32 nwords) -> u32
    {
      const u32 slot = atomicAdd(&fill_count[wave], 1u);
      if(slot >= BR_FILLS) { return 0; }
      fill_list[wave][slot] = word | ((nwords - 1) << 10) | (sym << 20);      // word < 1024, nwords <= 1024
      return nwords;
    };
    if(inside && UNIFORM)
    {
      // Streams of longer runs (round 6; reads of a genome at 300 x coverage: 5 positions per byte).  The loop below serves them badly: hardly a
      // word passes the test of the word path, and behind it the lanes split by byte class at every word -- words of four short runs, words with
      // a run of >= 32 or a varint -- so that the wave runs both bodies, with a flush check behind every byte and a loop of 32-bit appends for
      // every long run: 167 VALU instructions per byte position (PMC, profiles/r06_pmc_per_kernel_genome300.txt) where an iid stream takes 27.
      // Here every lane runs the same straight-line code for every byte -- the varint state machine as selects; a completed run (length 0 when
      // the byte completes none) gives its first min(length, room) bits to the current word, whole words of the rest are STORED (they lie
      // inside the lane's own run: nobody else writes them; >= 4 of them go to the cooperative fill), and the remainder opens the next word.
      u32 lo0 = 0, lo1 = 0, lo2 = 0;
      u32 fill = (u32)(bstart - ws) & 31u, wi = (u32)(bstart - ws) >> 5;
      u32 rsym = 0, rshift = 0, rlen = 0, cont = 0;                 // (a block inside the window: run lengths fit 32 bits)
      const u32* row = rows + lane * STAGE_WORDS;
#pragma unroll 1
      for(int w = 0; w < 16; w++)
      {
        const u32 word = row[w];
#pragma unroll
        for(int k = 0; k < 4; k++)
        {
          const u32 byte = (word >> (8 * k)) & 0xFF;
          const u32 q = (byte * 171u) >> 10, s = byte - 6 * q;
          const u32 more = byte >> 7;
          const u32 head_long = (cont == 0 && q + 1 >= MAX_RUN ? 1u : 0u);
          const u32 grown = rlen + ((byte & 0x7Fu) << (rshift & 31u));
          u32 len = (cont != 0 ? (more != 0 ? 0u : grown) : (head_long != 0 ? 0u : q + 1));
          const u32 sym = (cont != 0 ? rsym : s);
          rlen = (cont != 0 ? grown : q + 1); rshift = (cont != 0 ? rshift + 7 : 0u);
          rsym = sym;
          cont = (cont != 0 ? more : head_long);
          const u32 s0 = (u32)__builtin_amdgcn_sbfe((int)sym, 0, 1), s1 = (u32)__builtin_amdgcn_sbfe((int)sym, 1, 1), s2 = (u32)__builtin_amdgcn_sbfe((int)sym, 2, 1);
          // the first bits complete the current word (fill < 32)
          const u32 room = 32u - fill, take = (len < room ? len : room);
          const u32 upto = fill + take;                               // <= 32
          const u32 m = (upto >= 32u ? ~0u : (1u << upto) - 1u) & (~0u << fill);
          lo0 |= s0 & m; lo1 |= s1 & m; lo2 |= s2 & m;
          fill = upto; len -= take;
          if(fill == 32u)
          {
            atomicOr(&pl[wi], lo0); atomicOr(&pl[PW + wi], lo1); atomicOr(&pl[2 * PW + wi], lo2);       // edge words are shared with the neighbours
            lo0 = 0; lo1 = 0; lo2 = 0; fill = 0; wi++;
          }
          if(len >= 32u)
          {
            // whole words inside the run (fill == 0)
            u32 n = len >> 5;
            len &= 31u;
            if(sym != 0)
            {
              const u32 done = (FILL && n >= 4 ? queue_fill(sym, wi, n) : 0u);
              for(u32 j = done; j < n; j++) { pl[wi + j] = s0; pl[PW + wi + j] = s1; pl[2 * PW + wi + j] = s2; }
            }
            wi += n;
          }
          // what is left opens the next word (fill == 0 whenever len != 0)
          const u32 mt = (1u << len) - 1u;
          lo0 |= s0 & mt; lo1 |= s1 & mt; lo2 |= s2 & mt;
          fill += len;
        }
      }
      if(fill > 0 && wi < BR_WINDOW / 32) { atomicOr(&pl[wi], lo0); atomicOr(&pl[PW + wi], lo1); atomicOr(&pl[2 * PW + wi], lo2); }
    }
    else if(inside)
    {
      // The block lies inside the LDS window (the common case): its runs are appended to three bit
      // streams, one per plane, through a pair of 32-bit accumulators per plane (current word, spill-over) that release a
      // word whenever 32 bits are complete.  No clipping: bits past the last owned record are never read.
      // All arithmetic of the loop is 32-bit: the first version kept 64-bit accumulators, and its two v_lshlrev_b64 per byte
      // (quarter rate) and the select / or pairs on register pairs were half of the ~40 instructions per byte.
      u32 lo0 = 0, lo1 = 0, lo2 = 0, hi0 = 0, hi1 = 0, hi2 = 0;
      u32 fill = (u32)(bstart - ws) & 31u, wi = (u32)(bstart - ws) >> 5;
      auto flush = [&]()
      {
        atomicOr(&pl[wi], lo0); atomicOr(&pl[PW + wi], lo1); atomicOr(&pl[2 * PW + wi], lo2);   // edge words are shared with the neighbours
        lo0 = hi0; lo1 = hi1; lo2 = hi2; hi0 = 0; hi1 = 0; hi2 = 0; fill -= 32; wi++;
      };
      auto append = [&](u32 sym, u32 take)                      // 1 <= take <= 32, fill < 32
      {
        const u32 mask = (take >= 32 ? ~0u : (1u << take) - 1u);
        const u32 mlo = mask << fill, mhi = (mask >> 1) >> (31 - fill);
        const u32 s0 = 0u - (sym & 1u), s1 = 0u - ((sym >> 1) & 1u), s2 = 0u - ((sym >> 2) & 1u);
        lo0 |= s0 & mlo; lo1 |= s1 & mlo; lo2 |= s2 & mlo;
        hi0 |= s0 & mhi; hi1 |= s1 & mhi; hi2 |= s2 & mhi;
        fill += take;
        if(fill >= 32) { flush(); }
      };
      auto on_long = [&](u32 sym, u64 len)
      {
        u32 l = (u32)len;
        if(FILL)
        {
          if(fill != 0) { const u32 take = (l < 32 - fill ? l : 32 - fill); append(sym, take); l -= take; }   // completes the current word
          if(l >= 128)                                          // fill == 0 here: whole words follow
          {
            const u32 nfull = l >> 5;
            const u32 taken = (sym == 0 ? nfull : queue_fill(sym, wi, nfull));                     // endmarker runs leave the planes zero
            wi += taken; l -= 32 * taken;
          }
        }
        while(l > 0) { const u32 take = (l < 32 ? l : 32u); append(sym, take); l -= take; }
      };
      const u32* row = rows + lane * STAGE_WORDS;
      u32 rsym = 0, rshift = 0; u64 rlen = 0; bool cont = false;
      int w = 0;
#pragma unroll 1
      while(w < 16)
      {
#ifndef BWTM_BUILD_RECS_NO_WORD_PATH
        // Round 5.  The per-byte `if(fill >= 32) flush()` below is taken by SOME lane of the wave at nearly every byte (a lane completes a word
        // every ~24 bytes, 64 lanes), so the wave executed the flush body -- 3 LDS atomics + ~9 instructions -- ~0.9 times per byte on top of the
        // 22 of the straight-line code.  When every lane's four bytes are runs of at most 8 (bytes < 48: all but a few words of a read-like
        // stream), the word advances a lane by at most 32 positions, so a 64-bit accumulator per plane (lo = current word, hi = the next one)
        // takes all four runs -- deposited as one pattern per plane, built at offset 0 and shifted into place once -- and ONE flush check per
        // word suffices.  Wave-uniform test; any other word takes the per-byte path below.  The words of the word path run in a loop
        // of their own: as one of two branches of a common loop body the compiler shuffled the ~20 loop-carried registers between the two
        // branches' allocations at the end of every iteration (22 v_mov per word, a fifth of the path's instructions).
#pragma unroll 1
        for(; w < 16; w++)
        {
          const u32 word = row[w];
          const u32 over8 = (((word & 0x7F7F7F7Fu) + 0x50505050u) | word) & 0x80808080u;          // a byte >= 48
          if(__ballot(cont || over8 != 0) != 0) { break; }
#ifdef BWTM_SLACK_BUILD_RECS
          { u32 slack = word; valu_slack<BWTM_SLACK_BUILD_RECS>(slack); }
#endif
          // the four runs as one pattern per plane relative to the lane's current position (at most 32 bits), shifted into place once
          u32 pat0 = 0, pat1 = 0, pat2 = 0, adv = 0;
#pragma unroll
          for(int k = 0; k < 4; k++)
          {
            const u32 byte = (word >> (8 * k)) & 0xFF;
            const u32 q = (byte * 171u) >> 10;                           // byte / 6, exact for byte < 256
            const u32 sym = byte - 6 * q;
            const u32 m = ((2u << q) - 1u) << adv;                       // length q + 1 <= 8 at offset adv <= 24
            const u32 s0 = 0u - (sym & 1u), s1 = 0u - ((sym >> 1) & 1u), s2 = 0u - ((sym >> 2) & 1u);
            pat0 |= s0 & m; pat1 |= s1 & m; pat2 |= s2 & m;
            adv += q + 1;
          }
          const u64 w0 = (u64)pat0 << fill, w1 = (u64)pat1 << fill, w2 = (u64)pat2 << fill;      // fill < 32
          lo0 |= (u32)w0; lo1 |= (u32)w1; lo2 |= (u32)w2;
          hi0 |= (u32)(w0 >> 32); hi1 |= (u32)(w1 >> 32); hi2 |= (u32)(w2 >> 32);
          fill += adv;
          if(fill >= 32) { flush(); }
        }
        if(w >= 16) { break; }
#endif
        const u32 word = row[w];
        w++;
        // bytes >= 186 (runs of 32 and more, heads of runs with a varint extension): high bit set and low 7 bits >= 0x3A
        const u32 big = ((word & 0x7F7F7F7Fu) + 0x46464646u) & word & 0x80808080u;
        if(!cont && big == 0)
        {
          // four one-byte runs of 1 .. 31: straight-line, no per-byte checks
#ifdef BWTM_SLACK_BUILD_RECS
          { u32 slack = word; valu_slack<BWTM_SLACK_BUILD_RECS>(slack); }
#endif
#pragma unroll
          for(int k = 0; k < 4; k++)
          {
            const u32 byte = (word >> (8 * k)) & 0xFF;
            const u32 q = (byte * 171u) >> 10;                           // byte / 6, exact for byte < 256
            const u32 sym = byte - 6 * q, len = q + 1;
            const u32 mask = (1u << len) - 1u;
            const u32 mlo = mask << fill, mhi = (mask >> 1) >> (31 - fill);
            const u32 s0 = 0u - (sym & 1u), s1 = 0u - ((sym >> 1) & 1u), s2 = 0u - ((sym >> 2) & 1u);
            lo0 |= s0 & mlo; lo1 |= s1 & mlo; lo2 |= s2 & mlo;
            hi0 |= s0 & mhi; hi1 |= s1 & mhi; hi2 |= s2 & mhi;
            fill += len;
            if(fill >= 32) { flush(); }
          }
          continue;
        }
#pragma unroll
        for(int k = 0; k < 4; k++)
        {
          const u32 byte = (word >> (8 * k)) & 0xFF;
          if(cont)
          {
            rlen += (u64)(byte & 0x7F) << rshift; rshift += 7; cont = (byte & 0x80) != 0;
            if(!cont) { on_long(rsym, rlen); }
          }
          else
          {
            const u32 q = (byte * 171u) >> 10; rsym = byte - 6 * q;
            if(q + 1 >= MAX_RUN) { rlen = q + 1; rshift = 0; cont = true; }
            else { const u32 l = q + 1; append(rsym, (l < 32 ? l : 32u)); if(l > 32) { append(rsym, l - 32); } }
          }
        }
      }
      if(fill > 0 && wi < BR_WINDOW / 32) { atomicOr(&pl[wi], lo0); atomicOr(&pl[PW + wi], lo1); atomicOr(&pl[2 * PW + wi], lo2); }
    }
    else if(have && bstart < we && bend > ws)
    {
      // The block straddles a window edge: general path with clipping, word-wise OR.
      u32 cur = 0, acc0 = 0, acc1 = 0, acc2 = 0;
      auto deposit = [&](u32 sym, u32 a, u32 e)                 // window-relative positions [a, e)
      {
        while(a < e)
        {
          const u32 w = a >> 5;
          if(FILL && (a & 31) == 0 && e - a >= 128)
          {
            const u32 taken = queue_fill(sym, w, (e - a) >> 5);
            if(taken != 0) { a += 32 * taken; continue; }
          }
          if(w != cur)
          {
            if(acc0) { atomicOr(&pl[cur], acc0); } if(acc1) { atomicOr(&pl[PW + cur], acc1); } if(acc2) { atomicOr(&pl[2 * PW + cur], acc2); }
            cur = w; acc0 = 0; acc1 = 0; acc2 = 0;
          }
          const u32 stop = (e < ((w + 1) << 5) ? e : ((w + 1) << 5));
          const u32 count = stop - a;
          const u32 mask = (count == 32 ? ~0u : ((1u << count) - 1u) << (a & 31));
          acc0 |= (sym & 1 ? mask : 0u); acc1 |= (sym & 2 ? mask : 0u); acc2 |= (sym & 4 ? mask : 0u);
          a = stop;
        }
      };
      u64 pos = bstart;
      auto run = [&](u32 sym, u64 len)
      {
        const u64 from = pos, to = pos + len;
        pos = to;
        if(sym != 0 && to > ws && from < we) { deposit(sym, (from > ws ? (u32)(from - ws) : 0u), (to < we ? (u32)(to - ws) : (u32)(we - ws))); }
      };
      for_each_run<false>(rows + lane * STAGE_WORDS, valid, [&](u32 sym, u32 l) { run(sym, (u64)l); }, run);
      if(acc0) { atomicOr(&pl[cur], acc0); } if(acc1) { atomicOr(&pl[PW + cur], acc1); } if(acc2) { atomicOr(&pl[2 * PW + cur], acc2); }
    }
    wave_sync_lds();
    if(FILL)
    {
      const u32 nfills = (fill_count[wave] < BR_FILLS ? fill_count[wave] : BR_FILLS);
      for(u32 k = 0; k < nfills; k++)
      {
        const u32 entry = fill_list[wave][k];
        const u32 word = entry & 0x3FF, nwords = ((entry >> 10) & 0x3FF) + 1, sym = entry >> 20;
        for(u32 j = lane; j < nwords; j += WAVE)
        {
          if(sym & 1) { pl[word + j] = ~0u; } if(sym & 2) { pl[PW + word + j] = ~0u; } if(sym & 4) { pl[2 * PW + word + j] = ~0u; }
        }
      }
      if(nfills != 0) { wave_sync_lds(); }
    }
    // records rr * 64 + lane of the window
    for(u32 rr = 0; rr < BR_WINDOW / 8192; rr++)
    {
    const uint4 P0 = planes[wave][0][rr * 64 + lane], P1 = planes[wave][1][rr * 64 + lane], P2 = planes[wave][2][rr * 64 + lane];
    u32 n1 = 0, n2 = 0, n3 = 0, n4 = 0, n5 = 0;
#define BWTM_COUNT_WORD(f) \
    n1 += (u32)__builtin_popcount(plane_match(P0.f, P1.f, P2.f, 1)); n2 += (u32)__builtin_popcount(plane_match(P0.f, P1.f, P2.f, 2)); \
    n3 += (u32)__builtin_popcount(plane_match(P0.f, P1.f, P2.f, 3)); n4 += (u32)__builtin_popcount(plane_match(P0.f, P1.f, P2.f, 4)); \
    n5 += (u32)__builtin_popcount(plane_match(P0.f, P1.f, P2.f, 5));
    BWTM_COUNT_WORD(x) BWTM_COUNT_WORD(y) BWTM_COUNT_WORD(z) BWTM_COUNT_WORD(w)
#undef BWTM_COUNT_WORD
    const u32 own12 = n1 | (n2 << 16), own34 = n3 | (n4 << 16);                        // wave totals <= 8192 per field
    const u32 incl12 = wave_incl_sum32(own12), incl34 = wave_incl_sum32(own34), incl5 = wave_incl_sum32(n5);
    const u64 own14 = (u64)own12 | ((u64)own34 << 32), incl14 = (u64)incl12 | ((u64)incl34 << 32);
    const u64 before14 = incl14 - own14, before5 = (u64)(incl5 - n5);
    const u64 q = (ws >> REC_SHIFT) + rr * 64 + lane;
    if(q >= q_lo && q < q_hi)
    {
      const u64 p = q << REC_SHIFT;
      const u64* sp = sup + (p >> SUPER_SHIFT) * SUP_STRIDE;
      u32 rel[6]; u32 h[4];
      rel[0] = 0;
      rel[1] = (u32)(a1 + (before14 & 0xFFFF) - sp[1]); rel[2] = (u32)(a2 + ((before14 >> 16) & 0xFFFF) - sp[2]);
      rel[3] = (u32)(a3 + ((before14 >> 32) & 0xFFFF) - sp[3]); rel[4] = (u32)(a4 + (before14 >> 48) - sp[4]);
      rel[5] = (u32)(a5 + before5 - sp[5]);
      pack_header(rel, h);
      uint4* dst = recs + 4 * q;
      dst[0] = make_uint4(P0.x, P1.x, P2.x, h[0]);
      dst[1] = make_uint4(P0.y, P1.y, P2.y, h[1]);
      dst[2] = make_uint4(P0.z, P1.z, P2.z, h[2]);
      dst[3] = make_uint4(P0.w, P1.w, P2.w, h[3]);
    }
    const u64 tot14 = shfl_u64(incl14, WAVE - 1), tot5 = shfl_u64(incl5, WAVE - 1);
    a1 += tot14 & 0xFFFF; a2 += (tot14 >> 16) & 0xFFFF; a3 += (tot14 >> 32) & 0xFFFF; a4 += tot14 >> 48; a5 += tot5;
    }
    wave_sync_lds();                                             // the planes are cleared again by the next window
  }
}

// cum[c * stride + b] = occurrences of c before the start of block first + b, b in [0, count)
// (CumulativeArray::sum(first + b) of samples[c], support.h:338-343), from the rank structure.
__global__ void __launch_bounds__(BLOCK_THREADS) k_block_cum(IndexView x, const u64* block_start, u64 first, u64 count, u64* cum, u64 stride)
{
  u64 b = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(b >= count) { return; }
  u64 p = block_start[first + b];
  u64 r[6]; index_ranks(x, p, r);
  cum[0 * stride + b] = p - (r[1] + r[2] + r[3] + r[4] + r[5]);
  cum[1 * stride + b] = r[1]; cum[2 * stride + b] = r[2]; cum[3 * stride + b] = r[3]; cum[4 * stride + b] = r[4]; cum[5 * stride + b] = r[5];
}

// Compact form of the samples (what travels to the host when the caller asks for it: 12 or 24 bytes per block instead of 56).
// Per block the six FIELDS: positions in the block and occurrences of 1..5 in it (differences of the ranks at consecutive block
// starts), stored 16 or 32 bits wide; every 64th block an ANCHOR: its absolute start position and absolute counts of 1..5.
// block_end, samples[c].sum() of any block = its anchor + at most 63 fields.  k_block_field_max decides the width.
__device__ inline void block_fields(const IndexView& x, const u64* block_start, u64 b, u64 f[6], u64 at[6])
{
  const u64 p0 = block_start[b], p1 = block_start[b + 1];
  u64 r0[6], r1[6];
  index_ranks(x, p0, r0); index_ranks(x, p1, r1);
  f[0] = p1 - p0; at[0] = p0;
#pragma unroll
  for(int c = 1; c < 6; c++) { f[c] = r1[c] - r0[c]; at[c] = r0[c]; }
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_block_field_max(const u64* block_start, u64 nblocks, unsigned long long* out_max)
{
  // grid-stride: one atomic per wave of a few thousand waves (one per 64 blocks of the stream was two million atomics on ONE address,
  // 10 - 17 ms at config 2 -- more than the download of the fields it sizes)
  u64 m = 0;
  for(u64 b = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x; b < nblocks; b += (u64)gridDim.x * BLOCK_THREADS)
  {
    const u64 len = block_start[b + 1] - block_start[b];            // the block's length bounds its five counts: no rank query needed
    m = (len > m ? len : m);
  }
  m = wave_max(m);
  if(lane_id() == 0 && m > 0) { atomicMax(out_max, (unsigned long long)m); }
}

template<class T>
__global__ void __launch_bounds__(BLOCK_THREADS) k_block_fields(IndexView x, const u64* block_start, u64 first, u64 count, T* fields, u64 stride,
  u64* anchors, u64 anchor_stride)
{
  u64 k = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(k >= count) { return; }
  u64 f[6], at[6]; block_fields(x, block_start, first + k, f, at);
#pragma unroll
  for(int c = 0; c < 6; c++) { fields[c * stride + k] = (T)f[c]; }
  if(((first + k) & 63) == 0)
  {
#pragma unroll
    for(int c = 0; c < 6; c++) { anchors[c * anchor_stride + (k >> 6)] = at[c]; }       // `first` is a multiple of 64
  }
}

// The same for an output-range slice whose records start at position `slice_start`: a block whose opening run began before
// the slice (it belongs to the slice because the run ENDS there) is answered from the counts at the slice start and the
// symbol of that run (the symbol at slice_start - 1), without the records of the slices before.
__global__ void __launch_bounds__(BLOCK_THREADS) k_block_cum_slice(IndexView x, const u64* block_start, u64 first, u64 count, u64* cum, u64 stride,
  u64 slice_start, u32 run_symbol)
{
  u64 b = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(b >= count) { return; }
  const u64 p = block_start[first + b];
  u64 r[6]; index_ranks(x, (p < slice_start ? slice_start : p), r);
  if(p < slice_start && run_symbol != 0)
  {
    const u64 d = slice_start - p;
    r[1] -= (run_symbol == 1 ? d : 0); r[2] -= (run_symbol == 2 ? d : 0); r[3] -= (run_symbol == 3 ? d : 0);
    r[4] -= (run_symbol == 4 ? d : 0); r[5] -= (run_symbol == 5 ? d : 0);
  }
  cum[0 * stride + b] = p - (r[1] + r[2] + r[3] + r[4] + r[5]);
  cum[1 * stride + b] = r[1]; cum[2 * stride + b] = r[2]; cum[3 * stride + b] = r[3]; cum[4 * stride + b] = r[4]; cum[5 * stride + b] = r[5];
}

//------------------------------------------------------------------------------
// Plain symbols (one byte each) -> records.  k_sym_counts: per-record symbol counts
// (cnt[c * stride + q], c = 1..5 used); after an exclusive scan k_sym_recs writes the records.

__global__ void __launch_bounds__(BLOCK_THREADS) k_sym_counts(const u8* sym, u64 n, u64 nrecs, u64* cnt, u64 stride)
{
  u64 q = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(q >= nrecs) { return; }
  u64 p = q << REC_SHIFT;
  u32 c[6] = {0, 0, 0, 0, 0, 0};
  for(u32 t = 0; t < REC_POS && p + t < n; t++)
  {
    u32 s = sym[p + t];
    c[0] += (s == 0); c[1] += (s == 1); c[2] += (s == 2); c[3] += (s == 3); c[4] += (s == 4); c[5] += (s == 5);
  }
  for(int k = 0; k < 6; k++) { cnt[k * stride + q] = c[k]; }
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_sym_sup(const u64* cum, u64 stride, u64 nrecs, u64* sup, u64 nsup)
{
  u64 s = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(s >= nsup) { return; }
  u64 q = s << SUPER_REC_SHIFT; if(q > nrecs) { q = nrecs; }
  for(int c = 0; c < SUP_STRIDE; c++) { sup[s * SUP_STRIDE + c] = (c >= 1 && c < 6 ? cum[c * stride + q] : 0); }
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_sym_recs(const u8* sym, u64 n, const u64* cum, u64 stride,
  const u64* sup, uint4* recs, u64 nrecs)
{
  u64 q = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(q >= nrecs) { return; }
  u64 p = q << REC_SHIFT;
  u32 plane[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for(u32 k = 0; k < 4; k++)
  {
    u32 a = 0, b = 0, c = 0;
    for(u32 t = 0; t < 32; t++)
    {
      u64 pos = p + 32 * k + t;
      u32 s = (pos < n ? sym[pos] : 0);
      a |= (s & 1u) << t; b |= ((s >> 1) & 1u) << t; c |= ((s >> 2) & 1u) << t;
    }
    plane[0][k] = a; plane[1][k] = b; plane[2][k] = c;
  }
  const u64* s = sup + (p >> SUPER_SHIFT) * SUP_STRIDE;
  u32 rel[6]; u32 h[4];
  for(int c = 1; c < 6; c++) { rel[c] = (u32)(cum[c * stride + q] - s[c]); }
  pack_header(rel, h);
  uint4* dst = recs + 4 * q;
#pragma unroll
  for(u32 k = 0; k < 4; k++) { dst[k] = make_uint4(plane[0][k], plane[1][k], plane[2][k], h[k]); }
}

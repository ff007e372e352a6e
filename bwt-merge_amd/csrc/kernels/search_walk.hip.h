/*
  kernels/search_walk.hip.h -- the search as per-chain LF walks with partitioned emit (small shards, long sequences).
  Part of bwtm_kernels.hip.h (included there, inside namespace bwtm); gfx950 only.
*/
#pragma once

//------------------------------------------------------------------------------
// K1: the search.  Every lane walks LF over one sequence of B at a time:
//     i = j; r = m_A; emit(i, r); loop { c = BWT_B[i]; if c == 0 stop;
//     i = LF_B(i); r = LF_A(r, c); emit(i, r) }
// which yields the same multiset of ranks as the reverse-trie DFS of buildRA
// (fmi.cpp:272-334; single-position branch 296-303, which produces 93 % of the values, taken
// for every node).  emit sets bit i + r of the interleaving bitvector: the B position i is
// known, so the sorted rank array needs no sort at all.
// Per step one 64-byte record of B and one of A are fetched (both addresses are known at the
// top of the iteration, so the two HBM accesses overlap), plus two L2-resident super rows.

//------------------------------------------------------------------------------
// K1, walk form used by bwtm_search: FOUR lanes per chain.  A 64-byte record is four 16-byte chunks
// {plane0, plane1, plane2, header word} of 32 positions each, so lane q of a quad loads chunk q
// with ONE dwordx4: the quad's four loads fall into one 64-byte line and cost a single request in
// the vector memory pipeline (measured: 95 G records/s against 23 G records/s when one lane issues
// four loads, tools/microbench_gather.hip).  Every lane counts in its own 32 positions, extracts
// its slice of the 25-bit header field, contributes the super-table entry it loaded, and a
// quad-wide DPP butterfly adds the pieces, so all four lanes hold the next (i, r).

__device__ inline u32 dpp_quad(u32 v, int ctrl_xor1)
{
  // ctrl_xor1 != 0: lanes [1,0,3,2]; else lanes [2,3,0,1]
  return (ctrl_xor1 ? (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false)
                    : (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
}

__device__ inline u64 quad_sum_u64(u64 v)
{
  u64 t = ((u64)dpp_quad((u32)(v >> 32), 1) << 32) | dpp_quad((u32)v, 1);
  v += t;
  t = ((u64)dpp_quad((u32)(v >> 32), 0) << 32) | dpp_quad((u32)v, 0);
  return v + t;
}

__device__ inline u32 quad_or_u32(u32 v)
{
  v |= dpp_quad(v, 1);
  return v | dpp_quad(v, 0);
}

// This lane's share of rank(c) within a record: matches below position j in its 32 positions
// plus its slice of the header field of c.  `ch` = {plane0, plane1, plane2, header word} of chunk q.
__device__ inline u32 quad_rank_part(uint4 ch, u32 q, u32 c, u32 j)
{
  u32 part = (u32)__builtin_popcount(plane_match(ch.x, ch.y, ch.z, c) & below_mask(j, q));
  // field of c occupies header bits [s, s + 25); this lane holds header bits [32 q, 32 q + 32)
  int lo = (int)(FIELD_BITS * (c - 1)) - 32 * (int)q;          // field start relative to this lane's word
  u64 wide = (u64)ch.w << 32;                                   // word at bits [32, 64) of a 64-bit window
  int sh = lo + 32;                                             // shift of the window (may be out of range)
  u32 piece = (sh >= 0 && sh < 64 ? (u32)(wide >> sh) : 0u) & FIELD_MASK;
  // lo >= 0: (w >> lo); lo < 0: (w << -lo); |lo| >= 32 or field below the word: 0 by the range test / mask
  return part + piece;
}

//------------------------------------------------------------------------------
// K1 + K2, walk form used by bwtm_search (shards below 2^21 sequences, long sequences): search with a PARTITIONED EMIT.
//
// Scattered memory-side atomics cap at ~24 G/s on MI355X and queue behind the next step's loads
// (DESIGN.md 3.1), so the walk does not touch the bitvector.  Every emit p = i + r becomes a 32-bit
// entry that is radix-partitioned in two levels (this is the "radix sort" of the north star,
// reduced to what the interleave needs: which output positions come from B):
//
//   tile  = p >> 16                       (65 536 bits = 8 KiB of the bitvector: an LDS tile)
//   level 1 (inside the walk): bin = tile & 255, staged in LDS rings, flushed as full 64-byte
//            lines into per-workgroup chunks of the bin's region;  entry = (tile >> 8) << 16 | (p & 0xFFFF)
//   level 2 (k_part_count / k_part_offsets / k_part_scatter): counting sort of every bin by
//            sub = tile >> 8 into exact per-tile lists of 16-bit offsets
//   tiles   (k_tile_build): one workgroup per tile sets the bits in LDS and ORs the 8 KiB into
//            the bitvector with plain coalesced stores.
//
// Interleaving the tiles over the bins (bin = tile & 255) keeps the bins balanced whatever the
// distribution of B among A.  Ring overflow (a > 32-deep burst into one bin within four
// iterations) and region overflow fall back to an atomicOr on the bitvector, so the result is
// exact in every case.

constexpr int WB_THREADS   = 512;
constexpr int TILE_SHIFT   = 16;
constexpr u32 TILE_MASK    = (1u << TILE_SHIFT) - 1;
constexpr int L1_BITS      = 7;
constexpr int L1_BINS      = 1 << L1_BITS;
constexpr int L1_RING      = 64;
constexpr int L1_CHUNK     = 256;          // entries per chunk reservation (1 KiB)
constexpr int L1_FLUSH_EVERY = 8;
constexpr u32 L1_SENTINEL  = 0xFFFFFFFFu;
constexpr int WALK_ILP     = 4;            // chains per quad

struct EmitSink
{
  u32* l1;            // L1_BINS * subs regions of `cap` entries (region = bin * subs + sub)
  u64  cap;           // entries per region (multiple of L1_CHUNK)
  u32  subs;          // sub-regions per bin (1 for the walk; the frontier search spreads its
                      // reservations over 64 counters per bin: same-address atomics serialize)
  u64* gcount;        // entries reserved per region
  u32* bits;          // the bitvector (fallback path)
  u32* overflow;      // set when a region overflowed (diagnostic; the fallback keeps the result exact)
};

__device__ inline void sink_fallback(u32* bits, u64 p) { atomicOr(bits + (p >> 5), 1u << (p & 31)); }

// Exact fallback of the partitioned emit (also the first version of the search): one atomicOr on the
// bitvector per emit.
__global__ void __launch_bounds__(BLOCK_THREADS) k_lf_walk_quad(IndexView A, IndexView B, u64 seq_first, u64 seq_count, u32* bits)
{
  __shared__ u64 sC[16];
  if(threadIdx.x == 0)
  {
#pragma unroll
    for(int k = 0; k < 8; k++) { sC[k] = A.C[k]; sC[8 + k] = B.C[k]; }
  }
  __syncthreads();

  const u32 q = threadIdx.x & 3;
  const u64 stride = ((u64)gridDim.x * BLOCK_THREADS) >> 2;
  u64 next = ((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 2;
  u64 i = 0, r = 0;
  bool walking = false;
  while(true)
  {
    if(!walking)
    {
      if(next >= seq_count) { break; }
      i = seq_first + next; r = A.m;                          // fmi.cpp:286: trie root "$"
      next += stride; walking = true;
      if(q == 0) { sink_fallback(bits, i + r); }
    }
    const uint4 cb = B.recs[4 * (i >> REC_SHIFT) + q];
    const uint4 ca = A.recs[4 * (r >> REC_SHIFT) + q];
    const u64* sb = B.sup + (i >> SUPER_SHIFT) * SUP_STRIDE;
    const u64* sa = A.sup + (r >> SUPER_SHIFT) * SUP_STRIDE;
    const u64 sb_q = sb[1 + q], sb_5 = sb[5];
    const u64 sa_q = sa[1 + q], sa_5 = sa[5];

    const u32 jb = (u32)(i & (REC_POS - 1)), ja = (u32)(r & (REC_POS - 1));
    // BWT_B[i]: held by the lane whose 32 positions contain jb.
    const u32 t = jb & 31;
    u32 mine = ((cb.x >> t) & 1u) | (((cb.y >> t) & 1u) << 1) | (((cb.z >> t) & 1u) << 2);
    const u32 c = quad_or_u32((jb >> 5) == q ? mine : 0u);
    if(c == 0) { walking = false; continue; }                 // fmi.cpp:299: start of the sequence (quad-uniform)
    u64 pb = (u64)quad_rank_part(cb, q, c, jb) + (c == q + 1 ? sb_q : 0) + ((q == 0 && c == 5) ? sb_5 : 0);
    u64 pa = (u64)quad_rank_part(ca, q, c, ja) + (c == q + 1 ? sa_q : 0) + ((q == 0 && c == 5) ? sa_5 : 0);
    i = sC[8 + c] + quad_sum_u64(pb);                         // LF_B(i), utils.h:335-341
    r = sC[c] + quad_sum_u64(pa);                             // LF_A(r, c), utils.h:343-348
    if(q == 0) { sink_fallback(bits, i + r); }
  }
}

__device__ inline void sink_append(u32* bits, u32* ring, u32* tail, const u32* head, u64 p)
{
  (void)head;                                   // the ring always starts at slot 0 (see sink_flush_bin)
  u64 tile = p >> TILE_SHIFT;
  u32 b = (u32)tile & (L1_BINS - 1);
  u32 entry = ((u32)(tile >> L1_BITS) << TILE_SHIFT) | ((u32)p & TILE_MASK);
  u32 slot = atomicAdd(&tail[b], 1u);
  if(slot < (u32)L1_RING) { ring[b * L1_RING + slot] = entry; }
  else { sink_fallback(bits, p); }
}

// Flushes full 16-entry blocks of bin b (one thread per bin, between barriers).
__device__ __attribute__((noinline)) void sink_flush_bin(const EmitSink sink, u32* ring, u32* tail, u32* head, u64* chunk_pos, u32* chunk_left, u32 b, bool final)
{
  (void)head;
  u32 h = 0;
  u32 real = tail[b]; if(real > (u32)L1_RING) { real = L1_RING; }     // slots past the ring took the fallback
  while(real >= 16 || (final && real > 0))
  {
    u32 n = (real >= 16 ? 16u : real);
    uint4* src = (uint4*)(ring + b * L1_RING + h);
    uint4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
    if(n < 16)
    {
      u32 e[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
      for(u32 k = 0; k < 16; k++) { if(k >= n) { e[k] = L1_SENTINEL; } }
      v0 = make_uint4(e[0], e[1], e[2], e[3]); v1 = make_uint4(e[4], e[5], e[6], e[7]);
      v2 = make_uint4(e[8], e[9], e[10], e[11]); v3 = make_uint4(e[12], e[13], e[14], e[15]);
    }
    if(chunk_left[b] == 0)
    {
      u64 base = atomicAdd((unsigned long long*)&sink.gcount[b], (unsigned long long)L1_CHUNK);
      if(base + L1_CHUNK <= sink.cap) { chunk_pos[b] = (u64)b * sink.cap + base; chunk_left[b] = L1_CHUNK; }
      else { atomicOr(sink.overflow, 1u); }
    }
    if(chunk_left[b] != 0)
    {
      uint4* dst = (uint4*)(sink.l1 + chunk_pos[b]);
      dst[0] = v0; dst[1] = v1; dst[2] = v2; dst[3] = v3;
      chunk_pos[b] += 16; chunk_left[b] -= 16;
    }
    else
    {
      // region full: apply the block directly
      u32 e[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
      for(u32 k = 0; k < 16; k++)
      {
        if(e[k] != L1_SENTINEL) { sink_fallback(sink.bits, ((((u64)(e[k] >> TILE_SHIFT) << L1_BITS) | b) << TILE_SHIFT) | (e[k] & TILE_MASK)); }
      }
    }
    h += 16; real -= n;
  }
  if(final)
  {
    // pad the open chunk so that every reserved entry is either valid or a sentinel
    uint4 sv = make_uint4(L1_SENTINEL, L1_SENTINEL, L1_SENTINEL, L1_SENTINEL);
    uint4* dst = (uint4*)(sink.l1 + chunk_pos[b]);
    for(u32 k = 0; k < chunk_left[b] / 4; k++) { dst[k] = sv; }
    chunk_left[b] = 0;
    tail[b] = 0;
  }
  else
  {
    // keep the < 16 left-over entries at the front of the ring
    if(h != 0) { for(u32 k = 0; k < real; k++) { ring[b * L1_RING + k] = ring[b * L1_RING + h + k]; } }
    tail[b] = real;
  }
}

// LDS_SUP: both super tables are staged in dynamic LDS (5 u64 per super block: symbols 1..5),
// which removes two of the four distinct-line gathers per step (measured: 259 -> 172 ms).
// (Without the tables in LDS -- indexes beyond ~3 x 10^10 positions walked per chain -- the kernel carries eight more 64-bit super-table values per
// lane: at four waves per SIMD it spilled ten registers to scratch (round 5, tests/_build/bwtm_api.s); three waves per SIMD hold them.)
template<bool LDS_SUP>
__global__ void __launch_bounds__(WB_THREADS, (LDS_SUP ? 4 : 3)) k_lf_walk_binned(IndexView A, IndexView B, u64 seq_first, u64 seq_count, EmitSink sink, u32 nsup_a, u32 nsup_b)
{
  extern __shared__ u64 sup_lds[];            // LDS_SUP: [5 * nsup_a] for A, then [5 * nsup_b] for B, C already added
  __shared__ u64 sC[16];
  __shared__ u32 ring[L1_BINS * L1_RING];
  __shared__ u32 tail[L1_BINS], chunk_left[L1_BINS];
  __shared__ u64 chunk_pos[L1_BINS];
  if(threadIdx.x < L1_BINS) { tail[threadIdx.x] = 0; chunk_left[threadIdx.x] = 0; chunk_pos[threadIdx.x] = 0; }
  if(threadIdx.x == 0)
  {
#pragma unroll
    for(int k = 0; k < 8; k++) { sC[k] = A.C[k]; sC[8 + k] = B.C[k]; }
  }
  if(LDS_SUP)
  {
    for(u32 k = threadIdx.x; k < 5 * nsup_a; k += WB_THREADS) { sup_lds[k] = A.sup[(k / 5) * SUP_STRIDE + 1 + (k % 5)] + A.C[1 + (k % 5)]; }
    for(u32 k = threadIdx.x; k < 5 * nsup_b; k += WB_THREADS) { sup_lds[5 * nsup_a + k] = B.sup[(k / 5) * SUP_STRIDE + 1 + (k % 5)] + B.C[1 + (k % 5)]; }
  }
  const u64* lds_a = sup_lds; const u64* lds_b = sup_lds + 5 * nsup_a;
  __syncthreads();

  const u32 q = threadIdx.x & 3;
  const u64 stride = ((u64)gridDim.x * WB_THREADS) >> 2;
  u64 next = ((u64)blockIdx.x * WB_THREADS + threadIdx.x) >> 2;
  u64 i[WALK_ILP], r[WALK_ILP];
  bool walking[WALK_ILP];
  uint4 cb[WALK_ILP], ca[WALK_ILP];
  u64 sbq[WALK_ILP], saq[WALK_ILP];
#pragma unroll
  for(int s = 0; s < WALK_ILP; s++) { i[s] = 0; r[s] = 0; walking[s] = false; cb[s] = make_uint4(0, 0, 0, 0); ca[s] = cb[s]; sbq[s] = 0; saq[s] = 0; }

  // The four chains of a quad are software-pipelined: a chain's next records are requested right
  // after its step has been computed and are consumed one loop iteration later, i.e. behind the
  // steps of the other three chains.
  for(u32 it = 0; ; it++)
  {
    u64 pend = 0; bool have = false;          // lane q carries the emit of chain q
#pragma unroll
    for(int s = 0; s < WALK_ILP; s++)
    {
      u64 emit = 0; bool emitted = false;
      if(walking[s])
      {
        const u32 jb = (u32)(i[s] & (REC_POS - 1)), ja = (u32)(r[s] & (REC_POS - 1));
        const u32 t = jb & 31;
        u32 mine = ((cb[s].x >> t) & 1u) | (((cb[s].y >> t) & 1u) << 1) | (((cb[s].z >> t) & 1u) << 2);
        const u32 c = quad_or_u32((jb >> 5) == q ? mine : 0u);      // BWT_B[i]
        if(c == 0) { walking[s] = false; }                          // fmi.cpp:299: start of the sequence
        else
        {
          u64 pb = (u64)quad_rank_part(cb[s], q, c, jb);
          u64 pa = (u64)quad_rank_part(ca[s], q, c, ja);
          if(LDS_SUP)
          {
            if(q == 0)
            {
              pb += lds_b[5 * (u32)(i[s] >> SUPER_SHIFT) + (c - 1)];   // includes C_B[c]
              pa += lds_a[5 * (u32)(r[s] >> SUPER_SHIFT) + (c - 1)];   // includes C_A[c]
            }
          }
          else
          {
            pb += (c == q + 1 ? sbq[s] : 0); pa += (c == q + 1 ? saq[s] : 0);
            if(q == 0)
            {
              pb += sC[8 + c]; pa += sC[c];
              if(c == 5) { pb += B.sup[(i[s] >> SUPER_SHIFT) * SUP_STRIDE + 5]; pa += A.sup[(r[s] >> SUPER_SHIFT) * SUP_STRIDE + 5]; }
            }
          }
          i[s] = quad_sum_u64(pb);                                  // LF_B(i), utils.h:335-341
          r[s] = quad_sum_u64(pa);                                  // LF_A(r, c), utils.h:343-348
          emit = i[s] + r[s]; emitted = true;
        }
      }
      if(!walking[s] && next < seq_count)
      {
        i[s] = seq_first + next; r[s] = A.m;                        // fmi.cpp:286: trie root "$"
        next += stride; walking[s] = true;
        emit = i[s] + r[s]; emitted = true;
      }
      if(walking[s])
      {
        cb[s] = B.recs[4 * (i[s] >> REC_SHIFT) + q];
        ca[s] = A.recs[4 * (r[s] >> REC_SHIFT) + q];
        if(!LDS_SUP)
        {
          sbq[s] = B.sup[(i[s] >> SUPER_SHIFT) * SUP_STRIDE + 1 + q];
          saq[s] = A.sup[(r[s] >> SUPER_SHIFT) * SUP_STRIDE + 1 + q];
        }
      }
      if((u32)s == q) { pend = emit; have = emitted; }
    }
    if(have) { sink_append(sink.bits, ring, tail, nullptr, pend); }
    if((it & (L1_FLUSH_EVERY - 1)) == L1_FLUSH_EVERY - 1)
    {
      bool busy = (next < seq_count);
#pragma unroll
      for(int s = 0; s < WALK_ILP; s++) { busy = busy || walking[s]; }
      int any = __syncthreads_or(busy ? 1 : 0);
      if(threadIdx.x < L1_BINS) { sink_flush_bin(sink, ring, tail, nullptr, chunk_pos, chunk_left, threadIdx.x, !any); }
      __syncthreads();
      if(!any) { break; }
    }
  }
}

// Level 2, pass a: histogram of sub-bins for one slice of one bin.  slice_bin / slice_begin
// describe the slices (host-built); counts is [nslices][nsub].
constexpr int PART_THREADS = 1024;
constexpr u64 PART_SLICE = 1ull << 20;      // entries per slice

__global__ void __launch_bounds__(PART_THREADS) k_part_count(const u32* l1, u64 cap, const u64* gcount, const u32* slice_region,
  const u64* slice_begin, u32 nsub, u32* counts)
{
  extern __shared__ u32 hist[];
  for(u32 k = threadIdx.x; k < nsub; k += PART_THREADS) { hist[k] = 0; }
  __syncthreads();
  u32 region = slice_region[blockIdx.x];
  u64 begin = slice_begin[blockIdx.x];
  u64 end = begin + PART_SLICE; u64 total = gcount[region]; if(total > cap) { total = cap; } if(end > total) { end = total; }
  const u32* src = l1 + (u64)region * cap;
  for(u64 k0 = begin; k0 < end; k0 += PART_THREADS)
  {
    u64 k = k0 + threadIdx.x;
    u32 e = (k < end ? src[k] : L1_SENTINEL);
    bool valid = (e != L1_SENTINEL);
    u32 key = e >> TILE_SHIFT;
    // entries written by the frontier search arrive in long runs of one key: add them with one LDS atomic
    u64 vm = __ballot(valid);
    if(vm != 0)
    {
      u32 first = (u32)__builtin_ctzll(vm);
      u32 key0 = (u32)__shfl((int)key, (int)first, WAVE);
      u64 same = __ballot(valid && key == key0);
      if(same == vm) { if(lane_id() == first) { atomicAdd(&hist[key0], (u32)__builtin_popcountll(vm)); } }
      else if(valid) { atomicAdd(&hist[key], 1u); }
    }
  }
  __syncthreads();
  for(u32 k = threadIdx.x; k < nsub; k += PART_THREADS) { counts[(u64)blockIdx.x * nsub + k] = hist[k]; }
}

// Level 2, pass b: per (bin, sub) = tile: exclusive prefix of the slice counts (in place) and the
// tile total.  One thread per tile; slices of a bin are consecutive: [bin_slice0[b], bin_slice0[b + 1]).
__global__ void __launch_bounds__(BLOCK_THREADS) k_part_offsets(u32* counts, const u32* bin_slice0, u32 nsub, u64* tile_total)
{
  u64 id = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(id >= (u64)nsub * L1_BINS) { return; }
  u32 b = (u32)(id / nsub), sub = (u32)(id % nsub);
  u64 acc = 0;
  for(u32 sl = bin_slice0[b]; sl < bin_slice0[b + 1]; sl++)
  {
    u64 idx = (u64)sl * nsub + sub;
    u32 c = counts[idx]; counts[idx] = (u32)acc; acc += c;
  }
  tile_total[(u64)sub * L1_BINS + b] = acc;
}

// Level 2, pass c, direct form: scatter the 16-bit offsets of one slice to their tiles' lists (used when the tables of
// the sorted form below do not fit the LDS: outputs beyond ~1.6e10 positions).
__global__ void __launch_bounds__(PART_THREADS) k_part_scatter(const u32* l1, u64 cap, const u64* gcount, const u32* slice_region, u32 subs,
  const u64* slice_begin, u32 nsub, const u32* counts, const u64* tile_start, unsigned short* out)
{
  extern __shared__ u64 cursor[];
  u32 region = slice_region[blockIdx.x];
  u32 b = region / subs;
  for(u32 k = threadIdx.x; k < nsub; k += PART_THREADS)
  {
    cursor[k] = tile_start[(u64)k * L1_BINS + b] + counts[(u64)blockIdx.x * nsub + k];
  }
  __syncthreads();
  u64 begin = slice_begin[blockIdx.x];
  u64 end = begin + PART_SLICE; u64 total = gcount[region]; if(total > cap) { total = cap; } if(end > total) { end = total; }
  const u32* src = l1 + (u64)region * cap;
  for(u64 k = begin + threadIdx.x; k < end; k += PART_THREADS)
  {
    u32 e = src[k];
    if(e != L1_SENTINEL)
    {
      u64 pos = atomicAdd((unsigned long long*)&cursor[e >> TILE_SHIFT], 1ull);
      out[pos] = (unsigned short)(e & TILE_MASK);
    }
  }
}

// Level 2, pass c: LDS counting sort of 16 384-entry chunks, so that entries of
// the same tile leave the workgroup as contiguous runs (a wave store touches ~3 lines instead
// of 64).  Dynamic LDS: sorted[SORT_CHUNK] u32, hist[nsub] u32, offs[nsub] u32, cursor[nsub] u64.
constexpr int SORT_CHUNK = 16384;
constexpr int SORT_PER_THREAD = SORT_CHUNK / PART_THREADS;      // 16

__global__ void __launch_bounds__(PART_THREADS) k_part_scatter_sorted(const u32* l1, u64 cap, const u64* gcount, const u32* slice_region, u32 subs,
  const u64* slice_begin, u32 nsub, const u32* counts, const u64* tile_start, unsigned short* out)
{
  extern __shared__ u64 lds_raw[];
  u64* cursor = lds_raw;                                   // [nsub]
  u32* sorted = (u32*)(cursor + nsub);                     // [SORT_CHUNK]
  u32* hist = sorted + SORT_CHUNK;                         // [nsub]
  u32* offs = hist + nsub;                                 // [nsub + 1]
  __shared__ u32 wave_total[PART_THREADS / WAVE];

  const u32 region = slice_region[blockIdx.x];
  const u32 b = region / subs;
  for(u32 k = threadIdx.x; k < nsub; k += PART_THREADS)
  {
    cursor[k] = tile_start[(u64)k * L1_BINS + b] + counts[(u64)blockIdx.x * nsub + k];
  }
  u64 begin = slice_begin[blockIdx.x];
  u64 end = begin + PART_SLICE; u64 total = gcount[region]; if(total > cap) { total = cap; } if(end > total) { end = total; }
  const u32* src = l1 + (u64)region * cap;

  for(u64 chunk = begin; chunk < end; chunk += SORT_CHUNK)
  {
    for(u32 k = threadIdx.x; k < nsub; k += PART_THREADS) { hist[k] = 0; }
    __syncthreads();
    u32 e[SORT_PER_THREAD], rank[SORT_PER_THREAD];
#pragma unroll
    for(int k = 0; k < SORT_PER_THREAD; k++)
    {
      u64 idx = chunk + (u64)k * PART_THREADS + threadIdx.x;
      e[k] = (idx < end ? src[idx] : L1_SENTINEL);
      rank[k] = (e[k] != L1_SENTINEL ? atomicAdd(&hist[e[k] >> TILE_SHIFT], 1u) : 0u);
    }
    __syncthreads();
    // exclusive scan of hist -> offs (each thread owns a contiguous strip of sub-bins)
    const u32 strip = (nsub + PART_THREADS - 1) / PART_THREADS;
    u32 s0 = threadIdx.x * strip, s1 = s0 + strip; if(s1 > nsub) { s1 = nsub; } if(s0 > nsub) { s0 = nsub; }
    u32 mine = 0;
    for(u32 k = s0; k < s1; k++) { mine += hist[k]; }
    u64 incl = wave_incl_sum(mine);
    if(lane_id() == WAVE - 1) { wave_total[threadIdx.x >> 6] = (u32)incl; }
    __syncthreads();
    u32 base = 0;
    for(u32 w = 0; w < (threadIdx.x >> 6); w++) { base += wave_total[w]; }
    u32 run = base + (u32)incl - mine;
    for(u32 k = s0; k < s1; k++) { offs[k] = run; run += hist[k]; }
    if(threadIdx.x == PART_THREADS - 1) { offs[nsub] = run; }
    __syncthreads();
#pragma unroll
    for(int k = 0; k < SORT_PER_THREAD; k++)
    {
      if(e[k] != L1_SENTINEL) { sorted[offs[e[k] >> TILE_SHIFT] + rank[k]] = e[k]; }
    }
    __syncthreads();
    const u32 valid = offs[nsub];
    for(u32 p = threadIdx.x; p < valid; p += PART_THREADS)
    {
      u32 v = sorted[p]; u32 sub = v >> TILE_SHIFT;
      out[cursor[sub] + (p - offs[sub])] = (unsigned short)(v & TILE_MASK);
    }
    __syncthreads();
    for(u32 k = threadIdx.x; k < nsub; k += PART_THREADS) { cursor[k] += hist[k]; }
    __syncthreads();
  }
}

// Tiles: set the bits of one 65 536-bit tile in LDS, then OR the 8 KiB into the bitvector.
__global__ void __launch_bounds__(BLOCK_THREADS) k_tile_build(const unsigned short* lists, const u64* tile_start, u64 ntiles, u64* bits, u64 nwords)
{
  __shared__ u32 tile[1 << (TILE_SHIFT - 5)];           // 2048 x u32 = 8 KiB
  u64 t = blockIdx.x;
  if(t >= ntiles) { return; }
  u64 begin = tile_start[t], end = tile_start[t + 1];
  if(begin == end) { return; }
  for(u32 k = threadIdx.x; k < (1u << (TILE_SHIFT - 5)); k += BLOCK_THREADS) { tile[k] = 0; }
  __syncthreads();
  for(u64 k = begin + threadIdx.x; k < end; k += BLOCK_THREADS)
  {
    u32 off = lists[k];
    atomicOr(&tile[off >> 5], 1u << (off & 31));
  }
  __syncthreads();
  u64 w0 = t << (TILE_SHIFT - 6);
  for(u32 k = threadIdx.x; k < (1u << (TILE_SHIFT - 6)); k += BLOCK_THREADS)
  {
    u64 w = w0 + k;
    if(w < nwords) { bits[w] |= (u64)tile[2 * k] | ((u64)tile[2 * k + 1] << 32); }
  }
}

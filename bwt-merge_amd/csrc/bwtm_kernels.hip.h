/*
  bwtm_kernels.hip.h -- hand-written gfx950 (CDNA4, wave64) kernels of the rank-array /
  interleave path.  Included by bwtm_api.hip only.

  Kernel map (reference code each one replaces):
    k_block_len        BWT::build scan of the run stream            bwt.cpp:487-502
    k_build_sup/recs   native blocks -> device records ("transcode at upload", BWT::load)
    k_block_cum        samples[c] at the block starts               bwt.cpp:489-511
    k_sym_*            plain symbols -> device records (input tooling)
    k_frontier_*       buildRA + BWT::inverse_select + BWT::rank, level-synchronous form (product)
                                                                    fmi.cpp:272-334, bwt.cpp:318-341, 445-464
    k_bound_suffix_min, k_tile_build_frontier
                       mergeRA / RLArray / RankArray: the sorted emits of every step -> bitvector tiles
                                                                    fmi.cpp:139-257, support.h:396-638
    k_lf_walk*, k_part_*, k_tile_build
                       the same two stages in per-chain form (fallback: long sequences, > 40-bit coordinates)
    k_chunk_popc       RA finalize (prefix counts of the interleaving bitvector)
    k_interleave_*     mergeBWT                                     bwt.cpp:215-282
    k_enc_*            RunBuffer + Run::write, block starts of BWT::build
                                                                    utils.h:121-142, support.h:256-282, bwt.cpp:496
    k_fold_*           byte-offset carry of Run::write across segments (array.size() % 64, support.h:267)
    k_rank_batch, k_inverse_select_batch, k_extract, k_find_batch
                       BWT::rank / inverse_select / extract, FMI::find  bwt.cpp:318-464, fmi.h:195-221
*/
#pragma once

#include <hip/hip_runtime.h>
#include "bwtm_device.h"

namespace bwtm
{

constexpr int WAVE = 64;
constexpr int BLOCK_THREADS = 256;

// Read-only view of a device index for kernels.
struct IndexView
{
  const uint4* recs;     // 4 x uint4 per record
  const u64*   sup;      // SUP_STRIDE u64 per super
  u64 n;                 // positions
  u64 m;                 // sequences
  u64 nrecs;
  u64 C[8];              // C[c] = number of symbols smaller than c
};

//------------------------------------------------------------------------------
// Wave helpers (wave64: every shuffle spans 64 lanes).

__device__ inline u32 lane_id() { return threadIdx.x & 63u; }

__device__ inline u64 shfl_u64(u64 v, int src)
{
  u32 lo = (u32)__shfl((int)(u32)v, src, WAVE);
  u32 hi = (u32)__shfl((int)(u32)(v >> 32), src, WAVE);
  return ((u64)hi << 32) | lo;
}

__device__ inline u64 shfl_up_u64(u64 v, int delta)
{
  u32 lo = (u32)__shfl_up((int)(u32)v, delta, WAVE);
  u32 hi = (u32)__shfl_up((int)(u32)(v >> 32), delta, WAVE);
  return ((u64)hi << 32) | lo;
}

// Inclusive prefix sum over the wave.
__device__ inline u64 wave_incl_sum(u64 v)
{
#pragma unroll
  for(int d = 1; d < WAVE; d <<= 1)
  {
    u64 t = shfl_up_u64(v, d);
    if((int)lane_id() >= d) { v += t; }
  }
  return v;
}

__device__ inline u64 wave_incl_max(u64 v)
{
#pragma unroll
  for(int d = 1; d < WAVE; d <<= 1)
  {
    u64 t = shfl_up_u64(v, d);
    if((int)lane_id() >= d) { v = (t > v ? t : v); }
  }
  return v;
}

__device__ inline u64 wave_sum(u64 v)   { return shfl_u64(wave_incl_sum(v), WAVE - 1); }
__device__ inline u64 wave_max(u64 v)   { return shfl_u64(wave_incl_max(v), WAVE - 1); }

//------------------------------------------------------------------------------
// Record access.

__device__ inline void load_record(const uint4* recs, u64 q, u32 w[16])
{
  const uint4* p = recs + 4 * q;
  uint4 a = p[0], b = p[1], c = p[2], d = p[3];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
  w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
  w[12] = d.x; w[13] = d.y; w[14] = d.z; w[15] = d.w;
}

// rank(i, c) for c in 1..5 on the device structure (BWT::rank, bwt.cpp:318-341).
__device__ inline u64 index_rank(const IndexView& x, u64 i, u32 c)
{
  u32 w[16];
  load_record(x.recs, i >> REC_SHIFT, w);
  return x.sup[(i >> SUPER_SHIFT) * SUP_STRIDE + c] + rec_header(w, c) + rec_count(w, c, (u32)(i & (REC_POS - 1)));
}

// rank for all c in 1..5 at once (BWT::ranks, bwt.cpp:343-361); out[0] unused.
__device__ inline void index_ranks(const IndexView& x, u64 i, u64 out[6])
{
  u32 w[16];
  load_record(x.recs, i >> REC_SHIFT, w);
  const u64* s = x.sup + (i >> SUPER_SHIFT) * SUP_STRIDE;
  u32 j = (u32)(i & (REC_POS - 1));
#pragma unroll
  for(u32 c = 1; c < 6; c++) { out[c] = s[c] + rec_header(w, c) + rec_count(w, c, j); }
}

// 64-bit windows of the three bit-planes starting at sequence position pos (zero past the end).
__device__ inline void load_window(const IndexView& x, u64 pos, u64& p0, u64& p1, u64& p2)
{
  u64 wi = pos >> 5;                  // global 32-position word index: record wi >> 2, chunk wi & 3
  u32 sh = (u32)(pos & 31);
  u64 last = 4 * x.nrecs;             // number of chunks
  uint4 z = make_uint4(0, 0, 0, 0);
  uint4 a = (wi     < last ? x.recs[wi]     : z);
  uint4 b = (wi + 1 < last ? x.recs[wi + 1] : z);
  uint4 c = (wi + 2 < last ? x.recs[wi + 2] : z);
  u64 l0 = (u64)a.x | ((u64)b.x << 32), l1 = (u64)a.y | ((u64)b.y << 32), l2 = (u64)a.z | ((u64)b.z << 32);
  p0 = l0 >> sh; p1 = l1 >> sh; p2 = l2 >> sh;
  if(sh != 0)
  {
    p0 |= (u64)c.x << (64 - sh); p1 |= (u64)c.y << (64 - sh); p2 |= (u64)c.z << (64 - sh);
  }
}

//------------------------------------------------------------------------------
// Generic exclusive scans over u64 arrays (sum or max).  Three phases: per-tile reduce,
// scan of the tile totals (recursive on the host side), per-tile scan + carry.

constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK_THREADS * SCAN_ITEMS;   // 2048

template<int OP> __device__ inline u64 scan_op(u64 a, u64 b) { return (OP == 0 ? a + b : (a > b ? a : b)); }

template<int OP>
__device__ inline u64 block_reduce(u64 v, u64* lds)
{
  u64 w = (OP == 0 ? wave_sum(v) : wave_max(v));
  if(lane_id() == 0) { lds[threadIdx.x >> 6] = w; }
  __syncthreads();
  u64 r = lds[0];
  for(int k = 1; k < BLOCK_THREADS / WAVE; k++) { r = scan_op<OP>(r, lds[k]); }
  __syncthreads();
  return r;
}

// blockIdx.y selects one of several equally long arrays (stride elements apart): the six sample
// arrays of an index are scanned by one launch.
template<int OP>
__global__ void __launch_bounds__(BLOCK_THREADS) k_scan_reduce(const u64* in, u64* partial, u64 n, u64 stride, u64 partial_stride)
{
  __shared__ u64 lds[BLOCK_THREADS / WAVE];
  in += (u64)blockIdx.y * stride; partial += (u64)blockIdx.y * partial_stride;
  u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
  u64 acc = 0;
  for(int k = 0; k < SCAN_ITEMS; k++) { if(base + k < n) { acc = scan_op<OP>(acc, in[base + k]); } }
  u64 total = block_reduce<OP>(acc, lds);
  if(threadIdx.x == 0) { partial[blockIdx.x] = total; }
}

// Exclusive scan of one tile; carry[blockIdx.x] (may be null for a single tile) is added.
template<int OP>
__global__ void __launch_bounds__(BLOCK_THREADS) k_scan_apply(const u64* in, u64* out, const u64* carry, u64 n, u64 stride, u64 carry_stride)
{
  __shared__ u64 lds[BLOCK_THREADS / WAVE];
  in += (u64)blockIdx.y * stride; out += (u64)blockIdx.y * stride;
  if(carry) { carry += (u64)blockIdx.y * carry_stride; }
  u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
  u64 item[SCAN_ITEMS];
  u64 acc = 0;
  for(int k = 0; k < SCAN_ITEMS; k++)
  {
    item[k] = (base + k < n ? in[base + k] : 0);
    acc = scan_op<OP>(acc, item[k]);
  }
  u64 incl = (OP == 0 ? wave_incl_sum(acc) : wave_incl_max(acc));
  u64 wave_total = shfl_u64(incl, WAVE - 1);
  u64 excl = shfl_up_u64(incl, 1);
  if(lane_id() == 0) { excl = 0; }
  if(lane_id() == 0) { lds[threadIdx.x >> 6] = wave_total; }
  __syncthreads();
  u64 prefix = (carry ? carry[blockIdx.x] : 0);
  for(int k = 0; k < (int)(threadIdx.x >> 6); k++) { prefix = scan_op<OP>(prefix, lds[k]); }
  u64 run = scan_op<OP>(prefix, excl);
  for(int k = 0; k < SCAN_ITEMS; k++)
  {
    if(base + k < n) { out[base + k] = run; }
    run = scan_op<OP>(run, item[k]);
  }
}

//------------------------------------------------------------------------------
// K0: native byte stream -> device rank structure (BWT::load + BWT::build, bwt.cpp:132-148, 476-512).
//
// The stream is cut into GROUPs of 62 consecutive 64-byte blocks; one wave owns one group and one
// lane decodes one block, so the byte stream is read exactly once per kernel with coalesced loads.
//
//   k_block_len   : positions per block (-> exclusive scan = block_start, the set bits of
//                   block_boundaries, bwt.cpp:496) and symbol counts per group (-> exclusive scan)
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
__device__ inline void stage_blocks(const u8* data, u64 nbytes, u64 first, u32 count, u32* rows)
{
  const u32 lane = lane_id();
  const uint4* src = (const uint4*)(data + first * RLE_BLOCK);
  const u64 left = nbytes - first * RLE_BLOCK;                        // bytes of the stream from `first` on
  const u64 chunks_avail = (left + 15) / 16;                          // the buffer is readable up to the next multiple of 16
#pragma unroll
  for(int k = 0; k < 4; k++)
  {
    u32 g = (u32)k * 64 + lane;
    if(g < 4 * count)
    {
      uint4 v = (g < chunks_avail ? src[g] : make_uint4(0, 0, 0, 0));
      if((u64)16 * g + 16 > left && g < chunks_avail)                 // the chunk that holds the last byte
      {
        u32 keep = (u32)(left - (u64)16 * g);                          // 1..15 bytes
        u32 m[4];
#pragma unroll
        for(u32 j = 0; j < 4; j++) { m[j] = (keep >= 4 * j + 4 ? ~0u : (keep <= 4 * j ? 0u : (1u << (8 * (keep - 4 * j))) - 1u)); }
        v.x &= m[0]; v.y &= m[1]; v.z &= m[2]; v.w &= m[3];
      }
      u32* dst = rows + (g >> 2) * STAGE_WORDS + (g & 3) * 4;
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
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

// blen[b] = positions encoded by block b; gcount[c * gstride + g] = occurrences of c in group g.
// flags bit 0: a block other than the last one encodes fewer than 64 positions.
__global__ void __launch_bounds__(BLOCK_THREADS) k_block_len(const u8* data, u64 nbytes, u64 nblocks, u64 ngroups,
  u64* blen, u64* gcount, u64 gstride, u32* flags)
{
  __shared__ u32 stage[BLOCK_THREADS / WAVE][STAGE_ROWS * STAGE_WORDS];
  const u32 lane = lane_id(), wave = threadIdx.x >> 6;
  const u64 g = (u64)blockIdx.x * (BLOCK_THREADS / WAVE) + wave;
  if(g >= ngroups) { return; }
  const u64 first = g * GROUP;
  const u32 nb = (nblocks > first ? (nblocks - first > (u64)GROUP ? (u32)GROUP : (u32)(nblocks - first)) : 0u);
  u32* rows = stage[wave];
  if(nb > 0) { stage_blocks(data, nbytes, first, nb, rows); }
  wave_sync_lds();
  // Short runs (< 42) are counted in packed 16-bit fields (at most 64 * 41 per block), long ones in 64 bits.
  u64 packed03 = 0; u32 packed45 = 0;
  u64 l0 = 0, l1 = 0, l2 = 0, l3 = 0, l4 = 0, l5 = 0;
  const u64 b = first + lane;
  if(lane < nb)
  {
    u64 begin = b * RLE_BLOCK;
    u32 valid = (nbytes - begin >= RLE_BLOCK ? (u32)RLE_BLOCK : (u32)(nbytes - begin));
    for_each_run<true>(rows + lane * STAGE_WORDS, valid,
      [&](u32 sym, u32 l)
      {
        const u64 add = (u64)l << (16 * (sym & 3));                    // symbols 4 and 5 use fields 0 and 1 of packed45
        packed03 += (sym < 4 ? add : 0ull); packed45 += (sym < 4 ? 0u : (u32)add);
      },
      [&](u32 sym, u64 len)
      {
        l0 += (sym == 0 ? len : 0); l1 += (sym == 1 ? len : 0); l2 += (sym == 2 ? len : 0);
        l3 += (sym == 3 ? len : 0); l4 += (sym == 4 ? len : 0); l5 += (sym == 5 ? len : 0);
      });
    l0 += packed03 & 0xFFFF; l1 += (packed03 >> 16) & 0xFFFF; l2 += (packed03 >> 32) & 0xFFFF; l3 += packed03 >> 48;
    l4 += packed45 & 0xFFFF; l5 += packed45 >> 16;
    u64 total = l0 + l1 + l2 + l3 + l4 + l5;
    blen[b] = total;
    if(total < RLE_BLOCK && b + 1 < nblocks) { atomicOr(flags, 1u); }
  }
  u64 t0 = wave_sum(l0), t1 = wave_sum(l1), t2 = wave_sum(l2), t3 = wave_sum(l3), t4 = wave_sum(l4), t5 = wave_sum(l5);
  if(lane == 0)
  {
    gcount[0 * gstride + g] = t0; gcount[1 * gstride + g] = t1; gcount[2 * gstride + g] = t2;
    gcount[3 * gstride + g] = t3; gcount[4 * gstride + g] = t4; gcount[5 * gstride + g] = t5;
  }
}

// block_end[b] = block_start[b + 1] - 1 (the set bits of block_boundaries, bwt.cpp:496).
__global__ void __launch_bounds__(BLOCK_THREADS) k_block_end(const u64* block_start, u64 nblocks, u64* block_end)
{
  u64 b = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(b < nblocks) { block_end[b] = block_start[b + 1] - 1; }
}

// Largest b in [0, nblocks) with block_start[b] <= p (p < n).
__device__ inline u64 find_block(const u64* block_start, u64 nblocks, u64 p)
{
  u64 lo = 0, hi = nblocks;           // invariant: block_start[lo] <= p < block_start[hi]
  while(hi - lo > 1)
  {
    u64 mid = (lo + hi) >> 1;
    if(block_start[mid] <= p) { lo = mid; } else { hi = mid; }
  }
  return lo;
}

// Super table from the native stream: one wave per super.  The counts at position p are the counts at
// the start of p's group plus the runs of the group's blocks before p (one lane per block).
__global__ void __launch_bounds__(BLOCK_THREADS) k_build_sup(const u8* data, u64 nbytes, const u64* block_start,
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
    const u64 b = find_block(block_start, nblocks, p);   // wave-uniform
    g = b / GROUP;
    const u64 blk = g * GROUP + lane;
    if(lane < (u32)GROUP && blk <= b)
    {
      u64 pos = block_start[blk], rle = blk * RLE_BLOCK;
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
template<u32 BR_WINDOW, int WAVES>
__global__ void __launch_bounds__(WAVES * WAVE) k_build_recs(const u8* data, u64 nbytes, const u64* block_start,
  const u64* gcum, u64 gstride, u64 nblocks, u64 ngroups, u64 n, const u64* sup, uint4* recs, u64 nrecs)
{
  constexpr u32 PW = BR_WINDOW / 32;                             // words per plane
  __shared__ u32 stage[WAVES][STAGE_ROWS * STAGE_WORDS];
  __shared__ uint4 planes[WAVES][3][BR_WINDOW / 128];
  const u32 lane = lane_id(), wave = threadIdx.x >> 6;
  const u64 g = (u64)blockIdx.x * WAVES + wave;
  if(g >= ngroups) { return; }
  const u64 first = g * GROUP;
  const bool last_group = (g + 1 == ngroups);
  const u32 nb = (nblocks > first ? (nblocks - first > (u64)STAGE_ROWS ? (u32)STAGE_ROWS : (u32)(nblocks - first)) : 0u);
  const u64 S = (nb > 0 ? block_start[first] : 0);
  const u64 q_lo = (S + REC_POS - 1) >> REC_SHIFT;
  const u64 q_hi = (last_group ? nrecs : (block_start[first + GROUP] + REC_POS - 1) >> REC_SHIFT);
  if(q_lo >= q_hi) { return; }                                  // wave-uniform: no record starts in this group
  u32* rows = stage[wave];
  if(nb > 0) { stage_blocks(data, nbytes, first, nb, rows); }
  const bool have = (lane < nb);
  const u64 b = first + lane;
  const u64 bstart = (have ? block_start[b] : 0), bend = (have ? block_start[b + 1] : 0);
  const u32 valid = (have ? (nbytes - b * RLE_BLOCK >= RLE_BLOCK ? (u32)RLE_BLOCK : (u32)(nbytes - b * RLE_BLOCK)) : 0u);
  u64 a1 = gcum[1 * gstride + g], a2 = gcum[2 * gstride + g], a3 = gcum[3 * gstride + g],
      a4 = gcum[4 * gstride + g], a5 = gcum[5 * gstride + g];   // counts before the first record of the window
  const u64 pos_end = ((q_hi << REC_SHIFT) < n ? (q_hi << REC_SHIFT) : n);
  u32* pl = (u32*)planes[wave];                                 // plane k: words [PW k, PW k + PW)
  for(u64 ws = S & ~(u64)(REC_POS - 1); (ws >> REC_SHIFT) < q_hi; ws += BR_WINDOW)
  {
#pragma unroll
    for(u32 k = 0; k < 3 * PW / WAVE; k++) { pl[k * 64 + lane] = 0; }
    wave_sync_lds();
    const u64 we = (ws + BR_WINDOW < pos_end ? ws + BR_WINDOW : pos_end);
    const bool inside = (have && bstart >= ws && bend <= ws + BR_WINDOW);
    if(inside)
    {
      // The block lies inside the LDS window (the common case): its runs are appended to three bit
      // streams, one per plane, through 64-bit shift accumulators that release a word whenever 32 bits
      // are complete.  No clipping: bits past the last owned record are never read.
      u64 acc0 = 0, acc1 = 0, acc2 = 0;
      u32 fill = (u32)(bstart - ws) & 31u, wi = (u32)(bstart - ws) >> 5;
      auto append = [&](u32 sym, u32 take)                      // 1 <= take <= 32, fill < 32
      {
        const u64 v = ((1ull << take) - 1ull) << fill;
        acc0 |= (sym & 1 ? v : 0ull); acc1 |= (sym & 2 ? v : 0ull); acc2 |= (sym & 4 ? v : 0ull);
        fill += take;
        if(fill >= 32)
        {
          atomicOr(&pl[wi], (u32)acc0); atomicOr(&pl[PW + wi], (u32)acc1); atomicOr(&pl[2 * PW + wi], (u32)acc2);   // edge words are shared with the neighbours
          acc0 >>= 32; acc1 >>= 32; acc2 >>= 32; fill -= 32; wi++;
        }
      };
      for_each_run<false>(rows + lane * STAGE_WORDS, valid,
        [&](u32 sym, u32 l) { append(sym, (l < 32 ? l : 32u)); if(l > 32) { append(sym, l - 32); } },
        [&](u32 sym, u64 len) { u32 l = (u32)len; while(l > 0) { u32 take = (l < 32 ? l : 32u); append(sym, take); l -= take; } });
      if(fill > 0 && wi < BR_WINDOW / 32) { atomicOr(&pl[wi], (u32)acc0); atomicOr(&pl[PW + wi], (u32)acc1); atomicOr(&pl[2 * PW + wi], (u32)acc2); }
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
    const u64 own14 = (u64)n1 | ((u64)n2 << 16) | ((u64)n3 << 32) | ((u64)n4 << 48);   // wave totals <= 8192 per field
    const u64 incl14 = wave_incl_sum(own14), incl5 = wave_incl_sum((u64)n5);
    const u64 before14 = incl14 - own14, before5 = incl5 - n5;
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

// cum[c * stride + b] = occurrences of c before the start of block b, b in [0, nblocks]
// (CumulativeArray::sum(b) of samples[c], support.h:338-343), from the rank structure.
__global__ void __launch_bounds__(BLOCK_THREADS) k_block_cum(IndexView x, const u64* block_start, u64 count, u64* cum, u64 stride)
{
  u64 b = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(b >= count) { return; }
  u64 p = block_start[b];
  u64 r[6]; index_ranks(x, p, r);
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

//------------------------------------------------------------------------------
// Batch queries (BWT::rank, BWT::inverse_select, BWT::extract) -- used by the facade and tests.

__global__ void __launch_bounds__(BLOCK_THREADS) k_rank_batch(IndexView x, const u64* pos, const u8* comps, u64 count, u64* out)
{
  u64 k = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(k >= count) { return; }
  u64 i = pos[k]; if(i > x.n) { i = x.n; }              // bwt.cpp:322
  u32 c = comps[k];
  if(c >= 6) { out[k] = 0; return; }                     // bwt.cpp:321
  u64 r[6]; index_ranks(x, i, r);
  u64 rest = r[1] + r[2] + r[3] + r[4] + r[5];
  out[k] = (c == 0 ? i - rest : (c == 1 ? r[1] : (c == 2 ? r[2] : (c == 3 ? r[3] : (c == 4 ? r[4] : r[5])))));
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_inverse_select_batch(IndexView x, const u64* pos, u64 count, u64* out_rank, u8* out_comp)
{
  u64 k = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(k >= count) { return; }
  u64 i = pos[k];
  if(i >= x.n) { out_rank[k] = 0; out_comp[k] = 0; return; }   // bwt.cpp:449
  u32 w[16]; load_record(x.recs, i >> REC_SHIFT, w);
  u32 c = rec_symbol(w, (u32)(i & (REC_POS - 1)));
  u64 r[6]; index_ranks(x, i, r);
  out_comp[k] = (u8)c;
  u64 rest = r[1] + r[2] + r[3] + r[4] + r[5];
  out_rank[k] = (c == 0 ? i - rest : (c == 1 ? r[1] : (c == 2 ? r[2] : (c == 3 ? r[3] : (c == 4 ? r[4] : r[5])))));
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_extract(IndexView x, u64 first, u64 count, u8* out)
{
  u64 k = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(k >= count) { return; }
  u64 i = first + k;
  const u32* words = (const u32*)x.recs;
  u64 wbase = (i >> REC_SHIFT) * REC_WORDS + ((i >> 5) & 3) * 4;
  u32 t = (u32)(i & 31);
  out[k] = (u8)(((words[wbase] >> t) & 1u) | (((words[wbase + 1] >> t) & 1u) << 1) | (((words[wbase + 2] >> t) & 1u) << 2));
}

// Backward search of a batch of patterns (FMI::find, fmi.h:195-209): one lane per pattern.
// Patterns are comp values, concatenated; pattern k is text[offsets[k] .. offsets[k + 1]).
// Output: closed range [sp, ep] (empty when sp > ep, like Range::empty).
__global__ void __launch_bounds__(BLOCK_THREADS) k_find_batch(IndexView x, const u8* text, const u64* offsets, u64 count, u64* out_sp, u64* out_ep)
{
  u64 k = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(k >= count) { return; }
  u64 begin = offsets[k], end = offsets[k + 1];
  if(begin == end) { out_sp[k] = 0; out_ep[k] = x.n - 1; return; }              // fmi.h:198
  u64 pos = end - 1;
  u32 c = text[pos];
  if(c >= 6) { out_sp[k] = 1; out_ep[k] = 0; return; }
  u64 Cc[7];
#pragma unroll
  for(int j = 0; j < 7; j++) { Cc[j] = x.C[j]; }
  auto C_of = [&](u32 cc) { return (cc == 0 ? Cc[0] : (cc == 1 ? Cc[1] : (cc == 2 ? Cc[2] : (cc == 3 ? Cc[3] : (cc == 4 ? Cc[4] : (cc == 5 ? Cc[5] : Cc[6])))))); };
  u64 sp = C_of(c), ep = C_of(c + 1) - 1;                                         // charRange, utils.h:318-323
  while(sp + 1 <= ep + 1 && pos > begin)                                          // !Range::empty(range)
  {
    pos--;
    c = text[pos];
    if(c >= 6) { sp = 1; ep = 0; break; }
    u64 rs[6], re[6];
    index_ranks(x, sp, rs); index_ranks(x, (ep + 1 > x.n ? x.n : ep + 1), re);
    u64 a, b;
    if(c == 0) { a = sp - (rs[1] + rs[2] + rs[3] + rs[4] + rs[5]); b = (ep + 1) - (re[1] + re[2] + re[3] + re[4] + re[5]); }
    else { a = (c == 1 ? rs[1] : (c == 2 ? rs[2] : (c == 3 ? rs[3] : (c == 4 ? rs[4] : rs[5])))); b = (c == 1 ? re[1] : (c == 2 ? re[2] : (c == 3 ? re[3] : (c == 4 ? re[4] : re[5])))); }
    sp = C_of(c) + a; ep = C_of(c) + b - 1;                                       // LF(range, c), utils.h:350-355
  }
  out_sp[k] = sp; out_ep[k] = ep;
}

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

// EMIT: 0 = atomicOr into the bitvector (the product path); 1 = nothing, 2 = plain 8-byte store
// of r at scratch[i] (diagnostic builds for pricing the emit traffic; results are not a rank array).
template<int EMIT>
__device__ inline void walk_emit(u32* bits, u64 i, u64 r)
{
  if(EMIT == 0) { u64 p = i + r; atomicOr(bits + (p >> 5), 1u << (p & 31)); }
  else if(EMIT == 2) { ((u64*)bits)[i] = r; }
  else { asm volatile("" :: "v"((u32)r), "v"((u32)i)); }
}

template<int EMIT>
__global__ void __launch_bounds__(BLOCK_THREADS) k_lf_walk(IndexView A, IndexView B, u64 seq_first, u64 seq_count, u32* bits)
{
  __shared__ u64 sC[16];
  if(threadIdx.x == 0)
  {
#pragma unroll
    for(int k = 0; k < 8; k++) { sC[k] = A.C[k]; sC[8 + k] = B.C[k]; }
  }
  __syncthreads();

  const u64 stride = (u64)gridDim.x * BLOCK_THREADS;
  u64 next = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  u64 i = 0, r = 0;
  bool walking = false;
  while(true)
  {
    if(!walking)
    {
      if(next >= seq_count) { break; }
      i = seq_first + next; r = A.m;                          // fmi.cpp:286: trie root "$"
      next += stride; walking = true;
      walk_emit<EMIT>(bits, i, r);
    }
    u32 wb[16], wa[16];
    load_record(B.recs, i >> REC_SHIFT, wb);
    load_record(A.recs, r >> REC_SHIFT, wa);
    const u64* sb = B.sup + (i >> SUPER_SHIFT) * SUP_STRIDE;
    const u64* sa = A.sup + (r >> SUPER_SHIFT) * SUP_STRIDE;
    u64 sb1 = sb[1], sb2 = sb[2], sb3 = sb[3], sb4 = sb[4], sb5 = sb[5];
    u64 sa1 = sa[1], sa2 = sa[2], sa3 = sa[3], sa4 = sa[4], sa5 = sa[5];

    u32 jb = (u32)(i & (REC_POS - 1)), ja = (u32)(r & (REC_POS - 1));
    u32 c = rec_symbol(wb, jb);                               // BWT_B[i]
    if(c == 0) { walking = false; continue; }                 // fmi.cpp:299: start of the sequence
    u64 supb = (c == 1 ? sb1 : (c == 2 ? sb2 : (c == 3 ? sb3 : (c == 4 ? sb4 : sb5))));
    u64 supa = (c == 1 ? sa1 : (c == 2 ? sa2 : (c == 3 ? sa3 : (c == 4 ? sa4 : sa5))));
    i = sC[8 + c] + supb + rec_header(wb, c) + rec_count(wb, c, jb);   // LF_B(i), utils.h:335-341
    r = sC[c] + supa + rec_header(wa, c) + rec_count(wa, c, ja);       // LF_A(r, c), utils.h:343-348
    walk_emit<EMIT>(bits, i, r);
  }
}

//------------------------------------------------------------------------------
// K1, product form: FOUR lanes per chain.  A 64-byte record is four 16-byte chunks
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

// ABL (timing-only ablations, EMIT == 1): bit 0 = no super-table loads, bit 1 = no record load of A,
// bit 2 = no record load of B (the walk then follows a synthetic pseudo-random chain).
template<int EMIT, int ABL = 0>
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
  u32 steps = 0;
  bool walking = false;
  while(true)
  {
    if(!walking)
    {
      if(next >= seq_count) { break; }
      i = seq_first + next; r = A.m;                          // fmi.cpp:286: trie root "$"
      next += stride; walking = true;
      if(q == 0) { walk_emit<EMIT>(bits, i, r); }
    }
    uint4 cb = make_uint4((u32)i * 2654435761u, (u32)(i >> 7) * 40503u, (u32)i ^ 0x5bd1e995u, 0);
    if(!(ABL & 4)) { cb = B.recs[4 * (i >> REC_SHIFT) + q]; }
    uint4 ca = cb;
    if(!(ABL & 2)) { ca = A.recs[4 * (r >> REC_SHIFT) + q]; }
    const u64* sb = B.sup + (i >> SUPER_SHIFT) * SUP_STRIDE;
    const u64* sa = A.sup + (r >> SUPER_SHIFT) * SUP_STRIDE;
    const u64 sb_q = ((ABL & 1) ? (i >> 3) : sb[1 + q]), sb_5 = ((ABL & 1) ? 0 : sb[5]);
    const u64 sa_q = ((ABL & 1) ? (r >> 3) : sa[1 + q]), sa_5 = ((ABL & 1) ? 0 : sa[5]);

    const u32 jb = (u32)(i & (REC_POS - 1)), ja = (u32)(r & (REC_POS - 1));
    // BWT_B[i]: held by the lane whose 32 positions contain jb.
    const u32 t = jb & 31;
    u32 mine = ((cb.x >> t) & 1u) | (((cb.y >> t) & 1u) << 1) | (((cb.z >> t) & 1u) << 2);
    const u32 c = quad_or_u32((jb >> 5) == q ? mine : 0u);
    if(c == 0 && ABL == 0) { walking = false; continue; }     // fmi.cpp:299: start of the sequence (quad-uniform)
    if(ABL != 0) { if(++steps > 100) { walking = false; steps = 0; continue; } }
    u64 pb = (u64)quad_rank_part(cb, q, c, jb) + (c == q + 1 ? sb_q : 0) + ((q == 0 && c == 5) ? sb_5 : 0);
    u64 pa = (u64)quad_rank_part(ca, q, c, ja) + (c == q + 1 ? sa_q : 0) + ((q == 0 && c == 5) ? sa_5 : 0);
    i = sC[8 + c] + quad_sum_u64(pb);                         // LF_B(i), utils.h:335-341
    r = sC[c] + quad_sum_u64(pa);                             // LF_A(r, c), utils.h:343-348
    if(ABL != 0) { i = (i * 0x9E3779B97F4A7C15ULL >> 13) % B.n; r = (r * 0xBF58476D1CE4E5B9ULL >> 11) % (A.n + 1); }
    if(q == 0) { walk_emit<EMIT>(bits, i, r); }
  }
}

//------------------------------------------------------------------------------
// K1 + K2, product form: search with a PARTITIONED EMIT.
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
template<bool LDS_SUP>
__global__ void __launch_bounds__(WB_THREADS, 4) k_lf_walk_binned(IndexView A, IndexView B, u64 seq_first, u64 seq_count, EmitSink sink, u32 nsup_a, u32 nsup_b)
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

// K1, product form 3: COALESCED LOADS, ONE CHAIN PER LANE.
// The quad kernel above makes every lane of a quad repeat the chain arithmetic (9.6 wave
// instructions per LF step against 2.8 for one lane per chain), and the ablation shows ~106 ms of
// pure issue time at config 2.  Here a lane owns one chain again, but the records still arrive
// with quad-shaped loads: for j = 0..3 lane l fetches chunk (l & 3) of the record of chain
// (l >> 2) + 16 j (record index taken from that lane with a wave shuffle), the 64 records are
// written to a per-wave LDS tile (rows of 20 words: conflict-free 128-bit reads) and every lane
// reads its own row back.  Same number of distinct-line requests as the quad kernel, a third of
// the vector instructions.
constexpr int WL_THREADS = 1024;
constexpr int WL_ROW = 20;                 // words per staged record (16 + 4 padding)

template<bool LDS_SUP>
__global__ void __launch_bounds__(WL_THREADS, 4) k_lf_walk_lds(IndexView A, IndexView B, u64 seq_first, u64 seq_count, EmitSink sink, u32 nsup_a, u32 nsup_b)
{
  extern __shared__ u64 dyn_lds[];           // stage tiles, then the super tables
  __shared__ u64 sC[16];
  __shared__ u32 ring[L1_BINS * L1_RING];
  __shared__ u32 tail[L1_BINS], head[L1_BINS], chunk_left[L1_BINS];
  __shared__ u64 chunk_pos[L1_BINS];
  u32* stage_all = (u32*)dyn_lds;                                              // [waves][64][WL_ROW]
  u64* sup_lds = dyn_lds + (WL_THREADS / WAVE) * 64 * WL_ROW / 2;             // [5 nsup_a][5 nsup_b]
  if(threadIdx.x < L1_BINS) { tail[threadIdx.x] = 0; head[threadIdx.x] = 0; chunk_left[threadIdx.x] = 0; chunk_pos[threadIdx.x] = 0; }
  if(threadIdx.x == 0)
  {
#pragma unroll
    for(int k = 0; k < 8; k++) { sC[k] = A.C[k]; sC[8 + k] = B.C[k]; }
  }
  if(LDS_SUP)
  {
    for(u32 k = threadIdx.x; k < 5 * nsup_a; k += WL_THREADS) { sup_lds[k] = A.sup[(k / 5) * SUP_STRIDE + 1 + (k % 5)]; }
    for(u32 k = threadIdx.x; k < 5 * nsup_b; k += WL_THREADS) { sup_lds[5 * nsup_a + k] = B.sup[(k / 5) * SUP_STRIDE + 1 + (k % 5)]; }
  }
  const u64* lds_a = sup_lds; const u64* lds_b = sup_lds + 5 * nsup_a;
  __syncthreads();

  const u32 lane = lane_id();
  u32* tile = stage_all + (threadIdx.x >> 6) * 64 * WL_ROW;
  const u32 src_lane = lane >> 2, part = lane & 3;
  const u64 stride = (u64)gridDim.x * WL_THREADS;
  u64 next = (u64)blockIdx.x * WL_THREADS + threadIdx.x;
  u64 i = 0, r = 0;
  bool walking = false;

  for(u32 it = 0; ; it++)
  {
    if(!walking && next < seq_count)
    {
      i = seq_first + next; r = A.m;                              // fmi.cpp:286: trie root "$"
      next += stride; walking = true;
      sink_append(sink.bits, ring, tail, head, i + r);
    }
    // Record indexes (0 for idle lanes: any valid record).
    const u32 qb = (walking ? (u32)(i >> REC_SHIFT) : 0u), qa = (walking ? (u32)(r >> REC_SHIFT) : 0u);
    uint4 vb[4], va[4];
#pragma unroll
    for(int j = 0; j < 4; j++)
    {
      u32 ib = (u32)__shfl((int)qb, (int)src_lane + 16 * j, WAVE);
      u32 ia = (u32)__shfl((int)qa, (int)src_lane + 16 * j, WAVE);
      vb[j] = B.recs[4 * (u64)ib + part];
      va[j] = A.recs[4 * (u64)ia + part];
    }
    u32 wb[16], wa[16];
    // B records through the tile
#pragma unroll
    for(int j = 0; j < 4; j++) { *(uint4*)(tile + (src_lane + 16 * j) * WL_ROW + 4 * part) = vb[j]; }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for(int k = 0; k < 4; k++) { uint4 t = *(const uint4*)(tile + lane * WL_ROW + 4 * k); wb[4 * k] = t.x; wb[4 * k + 1] = t.y; wb[4 * k + 2] = t.z; wb[4 * k + 3] = t.w; }
    __builtin_amdgcn_wave_barrier();
    // A records through the same tile
#pragma unroll
    for(int j = 0; j < 4; j++) { *(uint4*)(tile + (src_lane + 16 * j) * WL_ROW + 4 * part) = va[j]; }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for(int k = 0; k < 4; k++) { uint4 t = *(const uint4*)(tile + lane * WL_ROW + 4 * k); wa[4 * k] = t.x; wa[4 * k + 1] = t.y; wa[4 * k + 2] = t.z; wa[4 * k + 3] = t.w; }
    __builtin_amdgcn_wave_barrier();

    if(walking)
    {
      const u32 jb = (u32)(i & (REC_POS - 1)), ja = (u32)(r & (REC_POS - 1));
      const u32 c = rec_symbol(wb, jb);                             // BWT_B[i]
      if(c == 0) { walking = false; }                               // fmi.cpp:299: start of the sequence
      else
      {
        u64 supb, supa;
        if(LDS_SUP)
        {
          supb = lds_b[5 * (u32)(i >> SUPER_SHIFT) + (c - 1)];
          supa = lds_a[5 * (u32)(r >> SUPER_SHIFT) + (c - 1)];
        }
        else
        {
          supb = B.sup[(i >> SUPER_SHIFT) * SUP_STRIDE + c];
          supa = A.sup[(r >> SUPER_SHIFT) * SUP_STRIDE + c];
        }
        i = sC[8 + c] + supb + rec_header(wb, c) + rec_count(wb, c, jb);   // LF_B(i), utils.h:335-341
        r = sC[c] + supa + rec_header(wa, c) + rec_count(wa, c, ja);       // LF_A(r, c), utils.h:343-348
        sink_append(sink.bits, ring, tail, head, i + r);
      }
    }
    if((it & (L1_FLUSH_EVERY - 1)) == L1_FLUSH_EVERY - 1)
    {
      int any = __syncthreads_or((walking || next < seq_count) ? 1 : 0);
      if(threadIdx.x < L1_BINS) { sink_flush_bin(sink, ring, tail, head, chunk_pos, chunk_left, threadIdx.x, !any); }
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

// Level 2, pass c: scatter the 16-bit offsets of one slice to their tiles' lists.
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

// Level 2, pass c, product form: LDS counting sort of 16 384-entry chunks, so that entries of
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

//------------------------------------------------------------------------------
// K1, level-synchronous form ("frontier search").
//
// All chains advance together, one LF step per launch, and the frontier F_t (the chains that are
// t steps from the end of their sequence) is kept SORTED BY SUFFIX.  Then both coordinates are
// monotone along the frontier -- i (rank among B's suffixes) strictly increasing, r (rank among
// A's suffixes) non-decreasing -- so the records of both indexes are read as ascending runs of cache lines
// instead of random gathers, and the emitted bit positions i + r are increasing as well.
// One LF step keeps the order inside a symbol class (LF is monotone for a fixed symbol) and the
// classes occupy disjoint, increasing ranges [C[c], C[c+1]), so F_{t+1} = stable 5-way split of
// F_t by c = BWT_B[i]: the "radix sort + segmented scan" of the north star, one digit per step.
//
// No data is moved for the split: a block of FR_BLOCK elements writes its survivors grouped by
// class into its own slot of the next buffer and records (length, physical start) per (class,
// block) SEGMENT; the logical order of F_{t+1} is (class, block), and an exclusive scan of the
// segment lengths (logical order) lets the next step map logical indexes to physical ones.
// The reference explores the same trie level by level implicitly (fmi.cpp:286-323: ranges of B
// with equal suffixes); here every sequence keeps its own element, which yields the same multiset
// of ranks.

constexpr int FR_BLOCK = 256;                  // elements (= threads) per block
constexpr int FR_SEGS = 31;                    // segment-table entries staged per block

struct FrontierView
{
  // Coordinates are 40-bit: the low words of (i, r) share one 8-byte entry, the high bytes one
  // 2-byte entry (10 bytes per element in two arrays).
  const uint2* lo; const unsigned short* hi;                             // current frontier (physical layout)
  uint2* lo_next; unsigned short* hi_next;                               // next frontier
  const u64* seg_prefix;                       // exclusive scan of seg_len (5 * nb_max + 1 entries); last = N_t
  const u64* seg_phys;                         // physical start of every segment
  const u32* first_seg;                        // per block: the segment that holds its first element (k_frontier_prep)
  u64* seg_len_next; u64* seg_phys_next;       // produced for the next step
  u64 nb_max;                                  // blocks per class in the segment tables
  // Dense emit of this step (EMIT == 0): the frontier is sorted, so are its bit positions p = i + r.
  unsigned short* emit16;                      // in-tile offsets p & 0xFFFF at emit_base[step] + logical index
  const u64* emit_base;                        // [steps + 1] running number of emits
  u64 emit_cap;                                // capacity of emit16; emits beyond it fall back to atomicOr
  u32* bits32;                                 // the bitvector (fallback path only)
  u32* bound_row;                              // this step's row of tile boundaries: [ntiles + 1], pre-set to ~0
  u64 step;
};

__global__ void __launch_bounds__(BLOCK_THREADS) k_frontier_init(uint2* lo, unsigned short* hi, u64* seg_len, u64* seg_phys, u64 nb_max,
  u64 seq_first, u64 count, u64 m_a)
{
  u64 g = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(g < count)
  {
    u64 i = seq_first + g;                                                // fmi.cpp:286: trie root "$"
    lo[g] = make_uint2((u32)i, (u32)m_a); hi[g] = (unsigned short)(((i >> 32) & 0xFF) | (((m_a >> 32) & 0xFF) << 8));
  }
  if(g < 5 * nb_max)
  {
    u64 cls = g / nb_max, b = g % nb_max;
    u64 begin = b * FR_BLOCK;
    seg_len[g] = (cls == 0 && begin < count ? (count - begin < (u64)FR_BLOCK ? count - begin : (u64)FR_BLOCK) : 0);
    seg_phys[g] = begin;
  }
  if(g == 5 * nb_max) { seg_len[g] = 0; }
}

// The records of the wave's elements are loaded per lane (four 16-byte loads of the lane's own
// record, all issued before the first use).  Along the sorted frontier consecutive lanes hit the
// same or neighbouring records, so the loads of a wave touch a short ascending run of cache lines;
// records without an element in this step (about a quarter of them at 100 bp) are never fetched.
// (Measured against fetching the whole spanned window with coalesced loads through LDS: 115 -> 105 ms
// per search at config 2.)
struct RecordFetch { uint4 v[4]; };

__device__ inline RecordFetch record_issue(const uint4* recs, u64 nrecs, u64 rec)
{
  RecordFetch rf;
  const uint4* p = recs + 4 * (rec < nrecs ? rec : nrecs - 1);
  rf.v[0] = p[0]; rf.v[1] = p[1]; rf.v[2] = p[2]; rf.v[3] = p[3];
  return rf;
}

__device__ inline void record_words(const RecordFetch& rf, u32 w[16])
{
#pragma unroll
  for(int k = 0; k < 4; k++) { w[4 * k] = rf.v[k].x; w[4 * k + 1] = rf.v[k].y; w[4 * k + 2] = rf.v[k].z; w[4 * k + 3] = rf.v[k].w; }
}

template<int EMIT>
__global__ void __launch_bounds__(FR_BLOCK, 8) k_frontier_step(IndexView A, IndexView B, FrontierView f)
{
  __shared__ u32 wave_cnt[FR_BLOCK / WAVE][6];
  __shared__ u64 s_prefix[FR_SEGS + 1], s_phys[FR_SEGS + 1];
  const u64 nseg = 5 * f.nb_max;
  const u64 N = f.seg_prefix[nseg];
  const u64 g0 = (u64)blockIdx.x * FR_BLOCK;
  const u32 lane = lane_id(), wave = threadIdx.x >> 6;

  // Blocks past the frontier only publish empty segments.
  if(g0 >= N)
  {
    if(threadIdx.x < 5) { f.seg_len_next[(u64)threadIdx.x * f.nb_max + blockIdx.x] = 0; f.seg_phys_next[(u64)threadIdx.x * f.nb_max + blockIdx.x] = g0; }
    if(blockIdx.x == 0 && threadIdx.x == 5) { f.seg_len_next[nseg] = 0; }
    return;
  }
  // The block's elements live in a handful of segments: stage their table entries in LDS.
  const u64 first_seg = f.first_seg[blockIdx.x];
  if(threadIdx.x <= FR_SEGS)
  {
    u64 sidx = first_seg + threadIdx.x; if(sidx > nseg) { sidx = nseg; }
    s_prefix[threadIdx.x] = f.seg_prefix[sidx];
    s_phys[threadIdx.x] = f.seg_phys[sidx < nseg ? sidx : nseg - 1];
  }
  __syncthreads();

  const u64 g = g0 + threadIdx.x;
  const bool active = (g < N);
  u64 i = 0, r = 0;
  if(active)
  {
    u64 phys;
    u32 k = 0;
    while(k < (u32)FR_SEGS && s_prefix[k + 1] <= g) { k++; }       // skips empty segments
    if(k < (u32)FR_SEGS) { phys = s_phys[k] + (g - s_prefix[k]); }
    else
    {
      u64 sgm = first_seg + FR_SEGS;                                 // rare: more segments than staged
      while(f.seg_prefix[sgm + 1] <= g) { sgm++; }
      phys = f.seg_phys[sgm] + (g - f.seg_prefix[sgm]);
    }
    uint2 l = f.lo[phys]; u32 h = f.hi[phys];
    i = (u64)l.x | ((u64)(h & 0xFF) << 32);
    r = (u64)l.y | ((u64)(h >> 8) << 32);
  }
  const u64 any_active = __ballot(active);
  u32 c = 0;
  u64 ni = 0, nr = 0;
  if(any_active != 0)
  {
    // idle lanes (a suffix of the wave) borrow the last active lane's coordinates
    const u32 last_lane = 63 - (u32)__builtin_clzll(any_active);
    const u64 li = shfl_u64(i, (int)last_lane), lr = shfl_u64(r, (int)last_lane);
    if(EMIT == 0)
    {
      // Dense emit + tile markers: bound_row[tile] = min(logical index of an element in the tile).
      // A lane marks when the previous lane lies in another tile; lane 0 of every wave always marks
      // (the true first element of the tile marks too and wins the minimum).  Tiles without
      // elements are filled in by k_bound_suffix_min.
      const u64 p = i + r;
      const u64 my_tile = p >> TILE_SHIFT;
      const u64 prev_tile = shfl_up_u64(my_tile, 1);
      if(active)
      {
        u64 slot = f.emit_base[f.step] + g;
        if(slot < f.emit_cap) { f.emit16[slot] = (unsigned short)(p & TILE_MASK); }
        else { sink_fallback(f.bits32, p); }                      // exact fallback; k_tile_build_frontier skips these slots
        if(lane == 0 || my_tile != prev_tile) { atomicMin(&f.bound_row[my_tile], (u32)g); }
      }
    }
    u32 wb[16];
    const u64 rec_b = (active ? i : li) >> REC_SHIFT, rec_a = (active ? r : lr) >> REC_SHIFT;
    RecordFetch fb = record_issue(B.recs, B.nrecs, rec_b);
    RecordFetch fa = record_issue(A.recs, A.nrecs, rec_a);         // in flight while B's record is used
    const u64 sup_b0 = shfl_u64(i, 0) >> SUPER_SHIFT, sup_a0 = shfl_u64(r, 0) >> SUPER_SHIFT;   // lane 0 is always active here
    const u64* row_b = B.sup + sup_b0 * SUP_STRIDE;                // wave-uniform addresses
    const u64* row_a = A.sup + sup_a0 * SUP_STRIDE;
    record_words(fb, wb);
    if(active) { c = rec_symbol(wb, (u32)(i & (REC_POS - 1))); }   // BWT_B[i]; 0 ends the chain (fmi.cpp:299)
    u64 supb = 0, supa = 0;
    if(active && c != 0)
    {
      // Super-table rows: the wave's coordinates are sorted, so nearly every lane needs the row of lane 0,
      // which was requested with scalar loads (row_b / row_a) together with the records.
      if((i >> SUPER_SHIFT) == sup_b0) { supb = (c == 1 ? row_b[1] : (c == 2 ? row_b[2] : (c == 3 ? row_b[3] : (c == 4 ? row_b[4] : row_b[5])))); }
      else { supb = B.sup[(i >> SUPER_SHIFT) * SUP_STRIDE + c]; }
      if((r >> SUPER_SHIFT) == sup_a0) { supa = (c == 1 ? row_a[1] : (c == 2 ? row_a[2] : (c == 3 ? row_a[3] : (c == 4 ? row_a[4] : row_a[5])))); }
      else { supa = A.sup[(r >> SUPER_SHIFT) * SUP_STRIDE + c]; }
      ni = rec_header(wb, c) + rec_count(wb, c, (u32)(i & (REC_POS - 1)));
    }
    u32 wa[16];
    record_words(fa, wa);
    if(active)
    {
      const u32 ja = (u32)(r & (REC_POS - 1));
      if(c != 0)
      {
        ni += supb;
        nr = supa + rec_header(wa, c) + rec_count(wa, c, ja);
        // C[c]: kernel arguments cannot be indexed dynamically without scratch, hence the selects
        u64 cb = (c == 1 ? B.C[1] : (c == 2 ? B.C[2] : (c == 3 ? B.C[3] : (c == 4 ? B.C[4] : B.C[5]))));
        u64 ca = (c == 1 ? A.C[1] : (c == 2 ? A.C[2] : (c == 3 ? A.C[3] : (c == 4 ? A.C[4] : A.C[5]))));
        ni += cb; nr += ca;                                     // LF_B(i), LF_A(r, c): utils.h:335-348
      }
    }
  }
  // Stable split by class inside the block.
  u32 my_rank = 0;
  u32 cnt_w[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for(u32 k = 1; k < 6; k++)
  {
    u64 m = __ballot(active && c == k);
    cnt_w[k] = (u32)__builtin_popcountll(m);
    if(c == k) { my_rank = (u32)__builtin_popcountll(m & ((1ull << lane) - 1)); }
  }
  if(lane == 0) { for(u32 k = 1; k < 6; k++) { wave_cnt[wave][k] = cnt_w[k]; } }
  // Raw barrier with an LDS-only wait: __syncthreads() would also drain vmcnt and expose the latency
  // of the emit reservation / stores that are still in flight (measured: +35 ms per search).
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  u32 class_base = 0, before_waves = 0;
#pragma unroll
  for(u32 k = 1; k < 6; k++)
  {
    u32 tot = 0, bw = 0;
    for(u32 w2 = 0; w2 < FR_BLOCK / WAVE; w2++) { u32 v = wave_cnt[w2][k]; if(w2 < wave) { bw += v; } tot += v; }
    if(k < c) { class_base += tot; }
    if(k == c) { before_waves = bw; }
    if(threadIdx.x == k - 1)
    {
      u32 base_k = 0;
      for(u32 k2 = 1; k2 < k; k2++) { for(u32 w2 = 0; w2 < FR_BLOCK / WAVE; w2++) { base_k += wave_cnt[w2][k2]; } }
      f.seg_len_next[(u64)(k - 1) * f.nb_max + blockIdx.x] = tot;
      f.seg_phys_next[(u64)(k - 1) * f.nb_max + blockIdx.x] = g0 + base_k;
    }
  }
  if(blockIdx.x == 0 && threadIdx.x == 5) { f.seg_len_next[nseg] = 0; }
  if(active && c != 0)
  {
    u64 dst = g0 + class_base + before_waves + my_rank;
    f.lo_next[dst] = make_uint2((u32)ni, (u32)nr);
    f.hi_next[dst] = (unsigned short)(((ni >> 32) & 0xFF) | (((nr >> 32) & 0xFF) << 8));
  }
}

// Per-step bookkeeping.  first_seg[b] = the segment that holds logical element b * FR_BLOCK: a segment
// has at most FR_BLOCK elements, so it covers at most one block boundary and every non-empty segment
// can publish "its" block directly (replaces a search of seg_prefix by every block of the step kernel).
// Dense emit: emit_base[t + 1] = emit_base[t] + N_t.
__global__ void __launch_bounds__(BLOCK_THREADS) k_frontier_prep(const u64* seg_prefix, u64 nseg, u32* first_seg, u64* emit_base, u64 step)
{
  u64 sgm = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(sgm == 0) { emit_base[step + 1] = emit_base[step] + seg_prefix[nseg]; }
  if(sgm >= nseg) { return; }
  u64 begin = seg_prefix[sgm], end = seg_prefix[sgm + 1];
  u64 b = (begin + FR_BLOCK - 1) / FR_BLOCK;
  if(b * FR_BLOCK < end) { first_seg[b] = (u32)sgm; }
}

// The same bookkeeping fused into the scan of the segment lengths (the path used when the segment table has at
// most FRONTIER_SCAN_TILES tiles): after k_scan_reduce has produced one total per 2048-entry tile, every block
// sums the totals before its tile itself, scans its tile, and publishes seg_prefix, first_seg and emit_base --
// two launches per step instead of four.
constexpr u64 FRONTIER_SCAN_TILES = 8192;

__global__ void __launch_bounds__(BLOCK_THREADS) k_frontier_scan(const u64* seg_len, const u64* partial, u64 nseg, u64* seg_prefix, u32* first_seg,
  u64* emit_base, u64 step)
{
  __shared__ u64 lds[BLOCK_THREADS / WAVE];
  const u64 n = nseg + 1;                                            // the entry after the last segment holds 0 and receives N_t
  u64 c = 0;
  for(u64 k = threadIdx.x; k < blockIdx.x; k += BLOCK_THREADS) { c += partial[k]; }
  const u64 carry = block_reduce<0>(c, lds);
  const u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
  u64 item[SCAN_ITEMS];
  u64 acc = 0;
  for(int k = 0; k < SCAN_ITEMS; k++) { item[k] = (base + k < n ? seg_len[base + k] : 0); acc += item[k]; }
  const u64 incl = wave_incl_sum(acc);
  const u64 wave_total = shfl_u64(incl, WAVE - 1);
  u64 excl = shfl_up_u64(incl, 1);
  if(lane_id() == 0) { excl = 0; }
  if(lane_id() == 0) { lds[threadIdx.x >> 6] = wave_total; }
  __syncthreads();
  u64 run = carry + excl;
  for(int k = 0; k < (int)(threadIdx.x >> 6); k++) { run += lds[k]; }
  for(int k = 0; k < SCAN_ITEMS; k++)
  {
    const u64 idx = base + k;
    if(idx < n)
    {
      seg_prefix[idx] = run;
      if(idx < nseg)
      {
        const u64 b = (run + FR_BLOCK - 1) / FR_BLOCK;
        if(b * FR_BLOCK < run + item[k]) { first_seg[b] = (u32)idx; }
      }
      else { emit_base[step + 1] = emit_base[step] + run; }
    }
    run += item[k];
  }
}

// Row t of the boundary table: bound[T] = logical index of the first element of step t whose bit
// position lies in tile >= T (suffix minimum over the markers; N_t past the last element).
__global__ void __launch_bounds__(BLOCK_THREADS) k_bound_suffix_min(u32* bound, u64 ntiles, const u64* emit_base, u64 nsteps)
{
  __shared__ u32 lds[BLOCK_THREADS];
  u64 t = blockIdx.x;
  if(t >= nsteps) { return; }
  u32* row = bound + t * (ntiles + 1);
  u32 running = (u32)(emit_base[t + 1] - emit_base[t]);        // N_t
  if(threadIdx.x == 0) { row[ntiles] = running; }
  for(u64 hi = ntiles; hi > 0; )
  {
    u64 lo = (hi > (u64)BLOCK_THREADS ? hi - BLOCK_THREADS : 0);
    u64 idx = lo + threadIdx.x;
    u32 v = (idx < hi ? row[idx] : 0xFFFFFFFFu);
    lds[threadIdx.x] = v;
    __syncthreads();
    // inclusive suffix min inside the chunk (Hillis-Steele over 256 entries)
    for(int d = 1; d < BLOCK_THREADS; d <<= 1)
    {
      u32 o = ((int)threadIdx.x + d < BLOCK_THREADS ? lds[threadIdx.x + d] : 0xFFFFFFFFu);
      __syncthreads();
      if(o < lds[threadIdx.x]) { lds[threadIdx.x] = o; }
      __syncthreads();
    }
    u32 m = lds[threadIdx.x]; if(running < m) { m = running; }
    if(idx < hi) { row[idx] = m; }
    u32 chunk_min = lds[0];
    __syncthreads();
    if(chunk_min < running) { running = chunk_min; }
    hi = lo;
  }
}

// Tiles from the dense per-step emits: tile T receives, from every step t, the contiguous run
// [bound[t][T], bound[t][T + 1]) of 16-bit offsets.  One workgroup per tile.
__global__ void __launch_bounds__(BLOCK_THREADS) k_tile_build_frontier(const unsigned short* emit16, const u64* emit_base, u64 emit_cap, const u32* bound,
  u64 ntiles, u64 nsteps, u64* bits, u64 nwords)
{
  __shared__ u32 tile[1 << (TILE_SHIFT - 5)];
  __shared__ u32 any;
  u64 T = blockIdx.x;
  if(T >= ntiles) { return; }
  for(u32 k = threadIdx.x; k < (1u << (TILE_SHIFT - 5)); k += BLOCK_THREADS) { tile[k] = 0; }
  if(threadIdx.x == 0) { any = 0; }
  __syncthreads();
  // Wave w takes the steps w, w + 4, ...  The run bounds of 64 of its steps are fetched at once (lane j
  // holds step w + 4 j) and handed out by shuffles, so that only the loads of the runs themselves are
  // dependent; four of those are in flight per lane.
  constexpr u32 NW = BLOCK_THREADS / WAVE;
  const u32 lane = lane_id(), wv = threadIdx.x >> 6;
  bool seen = false;
  for(u64 t0 = wv; t0 < nsteps; t0 += (u64)NW * WAVE)
  {
    const u64 tj = t0 + (u64)NW * lane;
    u32 my_lo = 0, my_hi = 0; u64 my_base = 0;
    if(tj < nsteps)
    {
      const u32* row = bound + tj * (ntiles + 1);
      my_lo = row[T]; my_hi = row[T + 1]; my_base = emit_base[tj];
    }
    const u64 left = (nsteps - t0 + NW - 1) / NW;                  // steps of this wave from t0 on
    const u32 cnt = (left < (u64)WAVE ? (u32)left : (u32)WAVE);
    for(u32 j = 0; j < cnt; j++)
    {
      const u32 lo = (u32)__shfl((int)my_lo, (int)j, WAVE), hi = (u32)__shfl((int)my_hi, (int)j, WAVE);
      const u64 base = shfl_u64(my_base, (int)j);
      seen |= (hi > lo);
      for(u32 k = lo + lane; k < hi; k += 4 * WAVE)
      {
        u32 off[4];
#pragma unroll
        for(u32 u = 0; u < 4; u++)
        {
          const u32 kk = k + u * WAVE;
          off[u] = (kk < hi && base + kk < emit_cap ? (u32)emit16[base + kk] : 0xFFFFFFFFu);
        }
#pragma unroll
        for(u32 u = 0; u < 4; u++) { if(off[u] != 0xFFFFFFFFu) { atomicOr(&tile[off[u] >> 5], 1u << (off[u] & 31)); } }
      }
    }
  }
  if(seen && lane == 0) { any = 1; }
  __syncthreads();
  if(any == 0) { return; }
  u64 w0 = T << (TILE_SHIFT - 6);
  for(u32 k = threadIdx.x; k < (1u << (TILE_SHIFT - 6)); k += BLOCK_THREADS)
  {
    u64 w = w0 + k;
    u64 v = (u64)tile[2 * k] | ((u64)tile[2 * k + 1] << 32);
    if(w < nwords && v != 0) { bits[w] |= v; }
  }
}

//------------------------------------------------------------------------------
// K2: rank-array finalize.  A chunk is 64 output records = 8192 bits = 128 words; one wave
// per chunk counts the set bits.  (An exclusive scan of the counts follows.)

constexpr int CHUNK_WORDS = 128;

__global__ void __launch_bounds__(BLOCK_THREADS) k_chunk_popc(const u64* bits, u64 nchunks, u64* cnt)
{
  u64 chunk = ((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6;
  if(chunk >= nchunks) { return; }
  const u64* w = bits + chunk * CHUNK_WORDS + 2 * lane_id();
  u64 v = (u64)__builtin_popcountll(w[0]) + (u64)__builtin_popcountll(w[1]);
  v = wave_sum(v);
  if(lane_id() == 0) { cnt[chunk] = v; }
}

// RA[i] for every B position (tests / facade): one wave per chunk, one lane per record.
__global__ void __launch_bounds__(BLOCK_THREADS) k_ra_extract(const u64* bits, const u64* chunk_base, u64 nchunks, u64 nb, u64* ra)
{
  u64 chunk = ((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6;
  if(chunk >= nchunks) { return; }
  u64 q = chunk * 64 + lane_id();
  u64 m0 = bits[2 * q], m1 = bits[2 * q + 1];
  u64 mine = (u64)__builtin_popcountll(m0) + (u64)__builtin_popcountll(m1);
  u64 incl = wave_incl_sum(mine);
  u64 i = chunk_base[chunk] + incl - mine;
  u64 base = q << REC_SHIFT;
  while(m0) { u32 t = (u32)__builtin_ctzll(m0); m0 &= m0 - 1; if(i < nb) { ra[i] = base + t - i; } i++; }
  while(m1) { u32 t = (u32)__builtin_ctzll(m1); m1 &= m1 - 1; if(i < nb) { ra[i] = base + 64 + t - i; } i++; }
}

//------------------------------------------------------------------------------
// K3: interleave (mergeBWT, bwt.cpp:215-282).  Output position p takes the next symbol of B
// when bit p of the interleaving bitvector is set and the next symbol of A otherwise, so
// an output record needs b_off = rank1(bits, 128 q) and a_off = 128 q - b_off, and its
// header is rank_A(a_off) + rank_B(b_off).  One lane per output record.

// Number of set bits before output record q, given the chunk bases.
__device__ inline u64 bits_before_record(const u64* bits, const u64* chunk_base, u64 q)
{
  u64 chunk = q >> 6;
  u64 b = chunk_base[chunk];
  for(u64 w = chunk * CHUNK_WORDS; w < 2 * q; w++) { b += (u64)__builtin_popcountll(bits[w]); }
  return b;
}

// Super table of the output: absolute counts at the start of every super.
__global__ void __launch_bounds__(BLOCK_THREADS) k_interleave_sup(IndexView A, IndexView B, const u64* bits, const u64* chunk_base,
  u64 n_out, u64* sup, u64 nsup)
{
  u64 s = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(s >= nsup) { return; }
  u64 q = s << SUPER_REC_SHIFT;
  u64 b_off = bits_before_record(bits, chunk_base, q);
  u64 a_off = (q << REC_SHIFT) - b_off;
  if(a_off > A.n) { a_off = A.n; }
  if(b_off > B.n) { b_off = B.n; }
  u64 ra[6], rb[6];
  index_ranks(A, a_off, ra); index_ranks(B, b_off, rb);
  for(int c = 0; c < SUP_STRIDE; c++) { sup[s * SUP_STRIDE + c] = (c >= 1 && c < 6 ? ra[c] + rb[c] : 0); }
  (void)n_out;
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_interleave(IndexView A, IndexView B, const u64* bits, const u64* chunk_base,
  u64 nchunks, const u64* sup_out, uint4* recs_out, u64 nrecs_out)
{
  u64 chunk = ((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6;
  if(chunk >= nchunks) { return; }
  u64 q = chunk * 64 + lane_id();
  u64 m0 = bits[2 * q], m1 = bits[2 * q + 1];
  u64 mine = (u64)__builtin_popcountll(m0) + (u64)__builtin_popcountll(m1);
  u64 incl = wave_incl_sum(mine);
  if(q >= nrecs_out) { return; }
  u64 b_off = chunk_base[chunk] + incl - mine;
  u64 a_off = (q << REC_SHIFT) - b_off;

  // Header: counts of symbols 1..5 before output position 128 q.
  u64 ra[6], rb[6];
  index_ranks(A, (a_off > A.n ? A.n : a_off), ra);
  index_ranks(B, (b_off > B.n ? B.n : b_off), rb);
  const u64* s = sup_out + (q >> SUPER_REC_SHIFT) * SUP_STRIDE;
  u32 rel[6]; u32 h[4];
  for(int c = 1; c < 6; c++) { rel[c] = (u32)(ra[c] + rb[c] - s[c]); }
  pack_header(rel, h);

  // Planes: two halves of 64 positions.
  u64 a0, a1, a2, b0, b1, b2, lo0, lo1, lo2, hi0, hi1, hi2;
  load_window(A, a_off, a0, a1, a2); load_window(B, b_off, b0, b1, b2);
  deposit64(m0, a0, a1, a2, b0, b1, b2, lo0, lo1, lo2);
  u64 nb0 = (u64)__builtin_popcountll(m0);
  load_window(A, a_off + 64 - nb0, a0, a1, a2); load_window(B, b_off + nb0, b0, b1, b2);
  deposit64(m1, a0, a1, a2, b0, b1, b2, hi0, hi1, hi2);

  uint4* dst = recs_out + 4 * q;
  dst[0] = make_uint4((u32)lo0, (u32)lo1, (u32)lo2, h[0]);
  dst[1] = make_uint4((u32)(lo0 >> 32), (u32)(lo1 >> 32), (u32)(lo2 >> 32), h[1]);
  dst[2] = make_uint4((u32)hi0, (u32)hi1, (u32)hi2, h[2]);
  dst[3] = make_uint4((u32)(hi0 >> 32), (u32)(hi1 >> 32), (u32)(hi2 >> 32), h[3]);
}

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

__global__ void __launch_bounds__(BLOCK_THREADS) k_enc_lasthead(const uint4* recs, u64 nrecs, u64 n, u64 ntiles, u64 nseg, u64* lasthead)
{
  u64 seg = ((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6;
  if(seg >= nseg) { return; }
  u64 first = seg * SEG_TILES;
  u32 carry = (first == 0 ? 0u : symbol_at(recs, (first << 6) - 1));
  u64 best = NONE;
  for(int k = 0; k < SEG_CHUNKS; k++)
  {
    u64 ft = first + (u64)k * 64;
    if(ft >= ntiles) { break; }
    TileInfo ti;
    carry = chunk_tiles(recs, nrecs, ft, n, carry, ti);
    u64 T = ft + lane_id();
    u64 mine = (ti.H != 0 && T < ntiles ? (T << 6) + (63 - (u64)__builtin_clzll(ti.H)) + 1 : NONE);
    u64 m = wave_max(mine);
    if(m > best) { best = m; }
  }
  if(lane_id() == 0) { lasthead[seg] = best; }
}

// Per-lane event statistics of a tile: number of events and the LONG events among them (heads that end a
// run of >= 42: no other head among the 41 positions before them; at most two per tile).  `before` =
// (position of the last head before this tile) + 1.
__device__ inline void tile_event_stats(const TileInfo& ti, u64 tile_base, u64 before, u32& nev, u64& long_mask)
{
  nev = (u32)__builtin_popcountll(ti.E); long_mask = 0;
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
  if(pos > 0 && pos + 1 - before >= MAX_RUN) { long_mask |= H & (0 - H); }
}

// The long events of a chunk in position order.  f(t, g, len): tile (lane) t, number of events of the
// chunk before this one, run length; all arguments are wave-uniform.  Every lane first works out its own
// (at most two) long events in parallel; the ordered walk then only broadcasts them.
template<class F>
__device__ inline void for_each_long_event(const TileInfo& ti, u64 first_tile, u64 before, u64 long_mask, u32 ev_excl, F&& f)
{
  u32 g0 = 0, g1 = 0; u64 len0 = 0, len1 = 0;
  if(long_mask != 0)
  {
    const u64 tb = (first_tile + lane_id()) << 6;
    u64 lm = long_mask;
#pragma unroll
    for(int k = 0; k < 2; k++)
    {
      if(lm != 0)
      {
        const u32 b = (u32)__builtin_ctzll(lm); lm &= lm - 1;
        const u64 below = (1ull << b) - 1;
        const u64 hb = ti.H & below;
        const u64 prev1 = (hb != 0 ? tb + (63 - (u64)__builtin_clzll(hb)) + 1 : before);     // (previous head) + 1
        const u32 g = ev_excl + (u32)__builtin_popcountll(ti.E & below);
        const u64 len = tb + b + 1 - prev1;
        if(k == 0) { g0 = g; len0 = len; } else { g1 = g; len1 = len; }
      }
    }
  }
  u64 pending = __ballot(long_mask != 0);
  while(pending)
  {
    const int t = (int)__builtin_ctzll(pending); pending &= pending - 1;
    f((u32)t, (u32)__shfl((int)g0, t, WAVE), shfl_u64(len0, t));
    const u64 second = shfl_u64(len1, t);
    if(second != 0) { f((u32)t, (u32)__shfl((int)g1, t, WAVE), second); }
  }
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_enc_size(const uint4* recs, u64 nrecs, u64 n, u64 ntiles, u64 nseg,
  const u64* prevhead, u32* table)
{
  u64 seg = ((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6;
  if(seg >= nseg) { return; }
  u64 first = seg * SEG_TILES;
  u32 carry = (first == 0 ? 0u : symbol_at(recs, (first << 6) - 1));
  u64 last = prevhead[seg];          // (last head before the segment) + 1, wave-uniform
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
    u64 incl = wave_incl_max(lh);
    u64 before = shfl_up_u64(incl, 1);
    if(lane_id() == 0) { before = NONE; }
    if(last > before) { before = last; }
    u32 nev; u64 long_mask;
    tile_event_stats(ti, T << 6, before, nev, long_mask);
    const u64 ev_incl = wave_incl_sum(nev);
    const u32 chunk_events = (u32)shfl_u64(ev_incl, WAVE - 1);
    // Events shorter than 42 are one byte under every hypothesis; only the long ones are resolved in order.
    u32 last_g = 0;
    for_each_long_event(ti, ft, before, long_mask, (u32)(ev_incl - nev), [&](u32, u32 g, u64 len)
    {
      acc += g - last_g;
      acc += long_run_bytes((u64)o + acc, len);
      last_g = g + 1;
    });
    acc += chunk_events - last_g;
    u64 m = shfl_u64(incl, WAVE - 1);
    if(m > last) { last = m; }
  }
  table[seg * 64 + o] = (u32)acc;
}

// Fold 1: composition of the segment tables of one group (lane o = start offset hypothesis).
constexpr int FOLD_GROUP = 256;

__global__ void __launch_bounds__(WAVE) k_fold_group(const u32* table, u64 nseg, u64* group_table)
{
  u64 g = blockIdx.x;
  u64 s0 = g * FOLD_GROUP, s1 = s0 + FOLD_GROUP; if(s1 > nseg) { s1 = nseg; }
  u64 acc = 0; u32 o = lane_id();
  for(u64 s = s0; s < s1; s++) { acc += table[s * 64 + ((o + acc) & 63)]; }
  group_table[g * 64 + o] = acc;
}

// Fold 2: sequential pass over the groups from offset 0; group_base[ngroups] = total bytes.
__global__ void __launch_bounds__(WAVE) k_fold_top(const u64* group_table, u64 ngroups, u64* group_base)
{
  if(threadIdx.x != 0) { return; }
  u64 off = 0;
  for(u64 g = 0; g < ngroups; g++) { group_base[g] = off; off += group_table[g * 64 + (off & 63)]; }
  group_base[ngroups] = off;
}

// Fold 3: byte offset of every segment.
__global__ void __launch_bounds__(WAVE) k_fold_seg(const u32* table, u64 nseg, const u64* group_base, u64* seg_base)
{
  if(threadIdx.x != 0) { return; }
  u64 g = blockIdx.x;
  u64 s0 = g * FOLD_GROUP, s1 = s0 + FOLD_GROUP; if(s1 > nseg) { s1 = nseg; }
  u64 off = group_base[g];
  for(u64 s = s0; s < s1; s++) { seg_base[s] = off; off += table[s * 64 + (off & 63)]; }
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_enc_emit(const uint4* recs, u64 nrecs, u64 n, u64 ntiles, u64 nseg,
  const u64* prevhead, const u64* seg_base, u8* out, u64* block_start)
{
  __shared__ __attribute__((aligned(16))) u8 stage[BLOCK_THREADS / WAVE][4096 + 32];
  u64 seg = ((u64)blockIdx.x * BLOCK_THREADS + threadIdx.x) >> 6;
  if(seg >= nseg) { return; }
  u64 first = seg * SEG_TILES;
  u32 carry = (first == 0 ? 0u : symbol_at(recs, (first << 6) - 1));
  u64 last = prevhead[seg];
  u64 off = seg_base[seg];           // wave-uniform byte offset
  for(int k = 0; k < SEG_CHUNKS; k++)
  {
    u64 ft = first + (u64)k * 64;
    if(ft >= ntiles) { break; }
    TileInfo ti;
    carry = chunk_tiles(recs, nrecs, ft, n, carry, ti);
    u64 T = ft + lane_id();
    if(T >= ntiles) { ti.H = 0; ti.E = 0; }
    u64 lh = (ti.H != 0 ? (T << 6) + (63 - (u64)__builtin_clzll(ti.H)) + 1 : NONE);
    u64 incl = wave_incl_max(lh);
    u64 before = shfl_up_u64(incl, 1);
    if(lane_id() == 0) { before = NONE; }
    if(last > before) { before = last; }
    u32 nev; u64 long_mask;
    tile_event_stats(ti, T << 6, before, nev, long_mask);
    u64 ev_incl = wave_incl_sum(nev);
    u64 chunk_events = shfl_u64(ev_incl, WAVE - 1);
    bool slow = (__ballot(long_mask != 0) != 0);
    if(!slow)
    {
      // Every event is a run shorter than 42: one byte each, in position order.  The bytes are
      // staged in LDS at the same 16-byte phase as their destination and leave as 16-byte stores.
      u8* lds = stage[threadIdx.x >> 6];
      const u32 a = (u32)(off & 15);
      u32 idx = a + (u32)(ev_incl - nev);
      if(ti.H != 0)
      {
        // Events of the tile in position order.  The run that ends at head b has the symbol found at the
        // previous head and the length b - (previous head); only the first head needs the 64-bit state
        // carried in from the tiles before.  The tile is walked as two 32-bit halves.
        const u64 tb = T << 6;
        const u32 phase = (u32)((off - a) & (RLE_BLOCK - 1));            // byte (off - a + idx) opens a block iff ((phase + idx) & 63) == 0
        const u32 b0 = (u32)__builtin_ctzll(ti.H);
        u64 h = ti.H;
        int prev_bit; u32 run_sym;
        if(tb + b0 == 0) { h &= h - 1; prev_bit = 0; run_sym = (u32)(ti.p0 & 1) | ((u32)(ti.p1 & 1) << 1) | ((u32)(ti.p2 & 1) << 2); }   // position 0 is a head without an event
        else { prev_bit = (int)b0 - (int)(u32)(tb + b0 + 1 - before); run_sym = event_symbol(ti, b0); }
#pragma unroll
        for(int half = 0; half < 2; half++)
        {
          u32 hh = (u32)(h >> (32 * half));
          const u32 q0 = (u32)(ti.p0 >> (32 * half)), q1 = (u32)(ti.p1 >> (32 * half)), q2 = (u32)(ti.p2 >> (32 * half));
          while(hh)
          {
            const u32 bb = (u32)__builtin_ctz(hh); hh &= hh - 1;
            const int bit = (int)bb + 32 * half;
            const u32 len = (u32)(bit - prev_bit);
            if(((phase + idx) & (u32)(RLE_BLOCK - 1)) == 0) { block_start[(off - a + idx) >> 6] = tb + (u64)(long long)prev_bit; }   // this run opens a block
            lds[idx++] = (u8)(run_sym + 6 * (len - 1));                  // Run::encodeBasic, support.h:231-234
            run_sym = ((q0 >> bb) & 1u) | (((q1 >> bb) & 1u) << 1) | (((q2 >> bb) & 1u) << 2);
            prev_bit = bit;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      const u32 total = a + (u32)chunk_events;
      u8* base = out + (off - a);                                      // 16-byte aligned
      for(u32 j = lane_id(); j * 16 < total; j += WAVE)
      {
        u32 lo = 16 * j, hi = lo + 16;
        if(lo >= a && hi <= total) { *(uint4*)(base + lo) = *(const uint4*)(lds + lo); }
        else
        {
          u32 from = (lo > a ? lo : a), to = (hi < total ? hi : total);
          for(u32 t = from; t < to; t++) { base[t] = lds[t]; }
        }
      }
      __builtin_amdgcn_wave_barrier();
      off += chunk_events;
    }
    else
    {
      // Some runs of >= 42 end in this chunk.  Their sizes depend on their byte offsets, so they are resolved
      // in order (a handful per chunk); every other event is one byte at (its event index + the extra bytes
      // of the long events before it), and all lanes write their tiles in parallel as above.
      u8* lds = stage[threadIdx.x >> 6];
      const u32 a = (u32)(off & 15);
      const u64 origin = off - a;                                      // stream offset of lds[0]
      const u32 ev_excl = (u32)(ev_incl - nev);
      u32 extra = 0, shift = 0;                                      // extra bytes of all long events / of those in earlier tiles
      for_each_long_event(ti, ft, before, long_mask, ev_excl, [&](u32 t, u32 g, u64 len)
      {
        const u32 sz = (u32)long_run_bytes(off + g + extra, len);
        if(lane_id() > t) { shift += sz - 1; }
        extra += sz - 1;
      });
      if(ti.H != 0)
      {
        const u64 tb = T << 6;
        u32 idx = a + ev_excl + shift;
        u64 h = ti.H;
        const u32 b0 = (u32)__builtin_ctzll(h);
        u64 prev1;                                                     // (previous head) + 1
        u32 run_sym;
        if(tb + b0 == 0) { h &= h - 1; prev1 = 1; run_sym = (u32)(ti.p0 & 1) | ((u32)(ti.p1 & 1) << 1) | ((u32)(ti.p2 & 1) << 2); }
        else { prev1 = before; run_sym = event_symbol(ti, b0); }
        while(h)
        {
          const u32 b = (u32)__builtin_ctzll(h); h &= h - 1;
          const u64 pos = tb + b, len = pos + 1 - prev1;
          if((long_mask >> b) & 1)
          {
            idx += (u32)long_run_write(lds, origin + idx, run_sym, len, block_start, prev1 - 1, origin);
          }
          else
          {
            if(((origin + idx) & (RLE_BLOCK - 1)) == 0) { block_start[(origin + idx) >> 6] = prev1 - 1; }
            lds[idx++] = (u8)(run_sym + 6 * (len - 1));
          }
          run_sym = (u32)((ti.p0 >> b) & 1) | ((u32)((ti.p1 >> b) & 1) << 1) | ((u32)((ti.p2 >> b) & 1) << 2);
          prev1 = pos + 1;
        }
      }
      __builtin_amdgcn_wave_barrier();
      const u32 total = a + (u32)chunk_events + extra;
      u8* base = out + origin;                                         // 16-byte aligned
      for(u32 j = lane_id(); j * 16 < total; j += WAVE)
      {
        u32 lo = 16 * j, hi = lo + 16;
        if(lo >= a && hi <= total) { *(uint4*)(base + lo) = *(const uint4*)(lds + lo); }
        else
        {
          u32 from = (lo > a ? lo : a), to = (hi < total ? hi : total);
          for(u32 t = from; t < to; t++) { base[t] = lds[t]; }
        }
      }
      __builtin_amdgcn_wave_barrier();
      off += chunk_events + extra;
    }
    u64 m = shfl_u64(incl, WAVE - 1);
    if(m > last) { last = m; }
  }
}

} // namespace bwtm

/*
  kernels/diagnostics.hip.h -- timing-only and A/B variants of the search kernels.  NOT part of the product build:
  compiled only with -DBWTM_DIAGNOSTICS (tools/build_variant.sh), selected through the bwtm_tune() keys walk_emit,
  walk_ablate, walk_kernel, walk_variant, scatter_kernel, walk_blocks.  Some of them produce results that are not a
  rank array (see the comments); DESIGN.md section 3.1 records what was measured with them.
  Part of bwtm_kernels.hip.h (included there, inside namespace bwtm); gfx950 only.
*/
#pragma once

// EMIT: 0 = atomicOr into the bitvector (first version; exact fallback of the partitioned emit); 1 = nothing, 2 = plain 8-byte store
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

// ABL (timing-only ablations, EMIT == 1): bit 0 = no super-table loads, bit 1 = no record load of A,
// bit 2 = no record load of B (the walk then follows a synthetic pseudo-random chain).
template<int EMIT, int ABL = 0>
__global__ void __launch_bounds__(BLOCK_THREADS) k_lf_walk_quad_diag(IndexView A, IndexView B, u64 seq_first, u64 seq_count, u32* bits)
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

// K1, experimental variant (walk_variant = 1, slower): COALESCED LOADS, ONE CHAIN PER LANE.
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

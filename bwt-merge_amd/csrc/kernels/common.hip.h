/*
  kernels/common.hip.h -- wave helpers, record access, generic scans.
  Part of bwtm_kernels.hip.h (included there, inside namespace bwtm); gfx950 only.
*/
#pragma once

//------------------------------------------------------------------------------
// Wave helpers (wave64: every shuffle spans 64 lanes).

__device__ inline u32 lane_id() { return threadIdx.x & 63u; }

// Issue-slack experiment (never in the product build): N extra dependent VALU instructions at a point of a kernel's hot loop, through
// tools/build_variant.sh <name> "" -DBWTM_SLACK_<KERNEL>=N.  The instructions feed nothing, so results are unchanged; a kernel whose
// time grows with N is bound by instruction issue, one whose time stays is bound by memory.
template<int N>
__device__ inline void valu_slack(u32& x)
{
#pragma unroll
  for(int k = 0; k < N; k++) { asm volatile("v_add_u32 %0, %0, 1" : "+v"(x)); }
}

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

// Inclusive prefix sum of 32-bit values with DPP row shifts and row broadcasts (no LDS crossbar: seven dependent VALU
// adds instead of twelve ds_bpermute round trips for a 64-bit shuffle scan).  For counts that fit 32 bits, also packed ones
// (e.g. two 16-bit fields) as long as no field overflows.
__device__ inline u32 wave_incl_sum32(u32 v)
{
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);      // row_shr:1
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);      // row_shr:2
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);      // row_shr:4
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);      // row_shr:8  -> scan inside every row of 16
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);      // row_bcast:15 into rows 1 and 3
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);      // row_bcast:31 into rows 2 and 3
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

// Inclusive "last nonzero so far" over the wave for values that increase with the lane wherever they are nonzero (positions
// of heads): lane L receives the value of the highest lane <= L that has one, 0 if none -- a ballot, a count of leading zeros
// and ONE 64-bit shuffle instead of the six shuffle-and-max steps of a generic inclusive max scan.
__device__ inline u64 wave_incl_last(u64 v)
{
  const u64 have = __ballot(v != 0) & ((2ull << lane_id()) - 1ull);        // lanes 0 .. L with a value
  const int src = 63 - (int)__builtin_clzll(have | 1ull);
  const u64 got = shfl_u64(v, src);
  return (have != 0 ? got : 0ull);
}

// Sum over the wave of values that are usually small: when every lane's value is below 2^25 (wave-uniform test) the sum fits
// 32 bits and takes the DPP scan (7 VALU instructions) instead of twelve ds_bpermute round trips for a 64-bit shuffle scan.
__device__ inline u64 wave_sum_mostly_small(u64 v)
{
  if(__ballot((v >> 25) != 0) == 0) { return (u64)(u32)__builtin_amdgcn_readlane((int)wave_incl_sum32((u32)v), WAVE - 1); }
  return wave_sum(v);
}
__device__ inline u64 wave_max(u64 v)   { return shfl_u64(wave_incl_max(v), WAVE - 1); }

//------------------------------------------------------------------------------
// Loads of memory that a kernel of ANOTHER GPU wrote before the last exchange (the parts' exported buffers of the merge over partitioned
// records: output coordinates, segment tables, node lists, boundary bits).  System scope: no cache of this GPU may answer with what the
// address held two steps ago (the buffers are reused every second step), whatever the runtime invalidates at a kernel's start.
// (The pointers come out of LDS-staged tables as generic ones: the casts say "global", so the loads are global_load ... sc0 sc1, not flat_load.)
typedef __attribute__((address_space(1))) const u64 peer_u64;
typedef __attribute__((address_space(1))) const unsigned short peer_u16;
__device__ inline u64 peer_load(const u64* p) { return __hip_atomic_load((peer_u64*)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline uint2 peer_load(const uint2* p) { const u64 v = peer_load((const u64*)p); return make_uint2((u32)v, (u32)(v >> 32)); }
__device__ inline unsigned short peer_load(const unsigned short* p) { return __hip_atomic_load((peer_u16*)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

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
  // unconditional loads from clamped indexes (a predicated load becomes a branch and the compiler then waits for every load
  // before it issues the next one: eight memory round trips in a row in a kernel that is launched once per LF step)
  u64 item[SCAN_ITEMS];
#pragma unroll
  for(int k = 0; k < SCAN_ITEMS; k++) { item[k] = in[base + k < n ? base + k : n - 1]; }
  u64 acc = 0;
#pragma unroll
  for(int k = 0; k < SCAN_ITEMS; k++) { if(base + k < n) { acc = scan_op<OP>(acc, item[k]); } }
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
#pragma unroll
  for(int k = 0; k < SCAN_ITEMS; k++) { item[k] = in[base + k < n ? base + k : n - 1]; }          // see k_scan_reduce
#pragma unroll
  for(int k = 0; k < SCAN_ITEMS; k++)
  {
    if(base + k >= n) { item[k] = 0; }
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

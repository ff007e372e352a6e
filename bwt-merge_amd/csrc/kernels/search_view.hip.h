/*
  kernels/search_view.hip.h -- the search view: two bit-planes + exceptions, 160 positions per 64 bytes (layout: bwtm_device.h),
  derived from the 64-byte records of an index.  Part of bwtm_kernels.hip.h (included there, inside namespace bwtm); gfx950 only.

  Only the frontier search reads it (k_frontier_step<.., VIEW = true>): a full LF step streams both rank structures once and is bound by
  those bytes alone, so 0.4 instead of 0.5 bytes per base is ~17 % of a step's traffic.  Everything else (node phase, walk, interleave,
  encoder, queries) keeps the ordinary records, which also serve the rare view records that overflow their exception slots.
*/
#pragma once

// vsup[8 s + c] = #c in [0, s * VIEW_SUPER_POS).
__global__ void __launch_bounds__(BLOCK_THREADS) k_view_sup(IndexView x, u64* vsup, u64 nvsup)
{
  const u64 s = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(s >= nvsup) { return; }
  u64 p = s * VIEW_SUPER_POS; if(p > x.n) { p = x.n; }
  u64 r[6]; index_ranks(x, p, r);
  u64* out = vsup + s * SUP_STRIDE;
  out[0] = 0; out[6] = 0; out[7] = 0;
  out[1] = r[1]; out[2] = r[2]; out[3] = r[3]; out[4] = r[4]; out[5] = r[5];
}

// One lane per view record: its five 32-position words are exactly five 16-byte chunks of the ordinary records (160 = 5 x 32).
__global__ void __launch_bounds__(BLOCK_THREADS) k_view_build(IndexView x, const u64* vsup, uint4* view, u64 nview)
{
  const u64 q = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(q >= nview) { return; }
  const u64 p0 = q * VIEW_POS;
  u32 v[16];
  u64 exc = 0; u32 nexc = 0;
  const u64 nchunks = 4 * x.nrecs;
#pragma unroll
  for(u32 k = 0; k < VIEW_WORDS; k++)
  {
    const u64 ch = 5 * q + k, pos = p0 + 32 * k;
    uint4 c = make_uint4(0, 0, 0, 0);
    if(ch < nchunks) { c = x.recs[ch]; }
    const u32 valid = (pos + 32 <= x.n ? 0xFFFFFFFFu : (pos >= x.n ? 0u : ((1u << (u32)(x.n - pos)) - 1u)));
    const u32 E = valid & (~(c.x | c.y | c.z) | (c.z & c.x));            // endmarkers and N
    v[k] = ~c.x & ~E & valid;
    v[VIEW_WORDS + k] = ((c.y & c.x) | c.z) & ~E & valid;
    u32 e = E;
    while(e)
    {
      const u32 b = (u32)__builtin_ctz(e); e &= e - 1;
      if(nexc < VIEW_EXC_SLOTS) { exc |= ((u64)(32 * k + b) << (8 * nexc)) | ((u64)((c.z >> b) & 1u) << (56 + nexc)); }
      nexc++;
    }
  }
  for(u32 k = (nexc < VIEW_EXC_SLOTS ? nexc : VIEW_EXC_SLOTS); k < VIEW_EXC_SLOTS; k++) { exc |= (u64)VIEW_EXC_EMPTY << (8 * k); }
  u64 r[6]; index_ranks(x, (p0 < x.n ? p0 : x.n), r);
  const u64* base = vsup + (q >> VIEW_SUPER_SHIFT) * SUP_STRIDE;
  u32 rel[6] = {0, (u32)(r[1] - base[1]), (u32)(r[2] - base[2]), (u32)(r[3] - base[3]), (u32)(r[4] - base[4]), (u32)(r[5] - base[5])};
  u32 h[4]; pack_header(rel, h);
  if(nexc > VIEW_EXC_SLOTS) { h[3] |= 1u << VIEW_OVERFLOW_BIT; }
  v[10] = h[0]; v[11] = h[1]; v[12] = h[2]; v[13] = h[3];
  v[14] = (u32)exc; v[15] = (u32)(exc >> 32);
  uint4* out = view + 4 * q;
  out[0] = make_uint4(v[0], v[1], v[2], v[3]); out[1] = make_uint4(v[4], v[5], v[6], v[7]);
  out[2] = make_uint4(v[8], v[9], v[10], v[11]); out[3] = make_uint4(v[12], v[13], v[14], v[15]);
}

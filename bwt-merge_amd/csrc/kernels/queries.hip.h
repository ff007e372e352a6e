/*
  kernels/queries.hip.h -- batch forms of BWT::rank / inverse_select / extract and FMI::find.
  Part of bwtm_kernels.hip.h (included there, inside namespace bwtm); gfx950 only.
*/
#pragma once

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

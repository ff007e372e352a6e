// Test-only shim: exposes the pure host+device helpers of bwtm_device.h to the CPU test-suite
// (compiled with g++; no HIP runtime involved).
#include <cstring>
#include "../bwt-merge_amd/csrc/bwtm_device.h"
#include "../bwt-merge_amd/csrc/bwtm_bitmerge.h"
#include "../bwt-merge_amd/csrc/bwtm_view.h"       // the experimental search view's record helpers (pure functions)
using namespace bwtm;
extern "C"
{
u64 shim_long_run_bytes(u64 offset, u64 length) { return long_run_bytes(offset, length); }
u64 shim_long_run_write(u8* out, u64 offset, u32 sym, u64 length) { return long_run_write(out, offset, sym, length); }
void shim_pack_header(const u32* rel, u32* h) { pack_header(rel, h); }
u32 shim_rec_header(const u32* w, u32 c) { return rec_header(w, c); }
u32 shim_rec_count(const u32* w, u32 c, u32 j) { return rec_count(w, c, j); }
u32 shim_rec_symbol(const u32* w, u32 j) { return rec_symbol(w, j); }
void shim_range_mask128(u32 from, u32 count, u64* lo, u64* hi) { range_mask128(from, count, *lo, *hi); }
// The encoder's sequential path (k_enc_emit): runs appended one after the other, block starts recorded.
u64 shim_encode_runs(u8* out, const u8* syms, const u64* lens, u64 count, u64* block_start)
{
  u64 off = 0, pos = 0;
  for(u64 k = 0; k < count; k++)
  {
    if(lens[k] < MAX_RUN)
    {
      if(off % RLE_BLOCK == 0) { block_start[off / RLE_BLOCK] = pos; }
      out[off++] = (u8)(syms[k] + 6 * (lens[k] - 1));
    }
    else { off += long_run_write(out, off, syms[k], lens[k], block_start, pos); }
    pos += lens[k];
  }
  return off;
}
// The search view (two planes + exceptions, 160 positions per record): a record built the way k_view_build does, and the queries.
int shim_view_build(const u8* sym, u32 count, const u32* rel, u32* v)
{
  for(int k = 0; k < 16; k++) { v[k] = 0; }
  u64 exc = 0; u32 nexc = 0;
  for(u32 p = 0; p < count; p++)
  {
    const u32 c = sym[p];
    if(c == 0 || c == 5)
    {
      if(nexc < VIEW_EXC_SLOTS) { exc |= ((u64)p << (8 * nexc)) | ((u64)(c == 5 ? 1 : 0) << (56 + nexc)); }
      nexc++;
    }
    else { v[p >> 5] |= ((c - 1) & 1u) << (p & 31); v[VIEW_WORDS + (p >> 5)] |= (((c - 1) >> 1) & 1u) << (p & 31); }
  }
  for(u32 k = (nexc < VIEW_EXC_SLOTS ? nexc : VIEW_EXC_SLOTS); k < VIEW_EXC_SLOTS; k++) { exc |= (u64)VIEW_EXC_EMPTY << (8 * k); }
  u32 h[4]; pack_header(rel, h);
  if(nexc > VIEW_EXC_SLOTS) { h[3] |= 1u << VIEW_OVERFLOW_BIT; }
  v[10] = h[0]; v[11] = h[1]; v[12] = h[2]; v[13] = h[3]; v[14] = (u32)exc; v[15] = (u32)(exc >> 32);
  return (int)nexc;
}
u32 shim_view_symbol(const u32* v, u32 j) { u32 below, below_n, at; view_exceptions(v[14], v[15], j, below, below_n, at); return view_symbol(v, j, at); }
u32 shim_view_count(const u32* v, u32 c, u32 j) { u32 below, below_n, at; view_exceptions(v[14], v[15], j, below, below_n, at); return view_count(v, c, j, below, below_n); }
u32 shim_view_header(const u32* v, u32 c) { return view_header(v, c); }
int shim_view_overflow(const u32* v) { return view_overflow(v) ? 1 : 0; }
// the bit merge k_interleave uses since round 4 (bwtm_bitmerge.h): one 32-bit mask word, three planes
void shim_bit_merge32(u32 mask, const u32* a, const u32* b, u32* o)
{
  const MergeMasks e = merge_masks(mask);
  for(int p = 0; p < 3; p++) { o[p] = bit_merge32(a[p], b[p], e); }
}
void shim_deposit64(u64 mask, const u64* a, const u64* b, u64* o) { deposit64(mask, a[0], a[1], a[2], b[0], b[1], b[2], o[0], o[1], o[2]); }
u64 shim_run_decode(const u8* data, u64 pos, u32* sym, u64* len) { run_decode(data, pos, *sym, *len); return pos; }
}

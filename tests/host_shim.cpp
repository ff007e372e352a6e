// Test-only shim: exposes the pure host+device helpers of bwtm_device.h to the CPU test-suite
// (compiled with g++; no HIP runtime involved).
#include <cstring>
#include "../bwt-merge_amd/csrc/bwtm_device.h"
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
void shim_deposit64(u64 mask, const u64* a, const u64* b, u64* o) { deposit64(mask, a[0], a[1], a[2], b[0], b[1], b[2], o[0], o[1], o[2]); }
u64 shim_run_decode(const u8* data, u64 pos, u32* sym, u64* len) { run_decode(data, pos, *sym, *len); return pos; }
}

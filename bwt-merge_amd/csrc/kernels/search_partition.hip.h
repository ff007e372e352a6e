// Partitioned search (experimental build): the frontier is cut at FIXED positions instead of into equal
// shares, so that GPU g only ever advances elements whose coordinates fall into its window of A's and of B's records.
//
// A cut is a pair (I, R) = (number of A's suffixes below some string w, number of B's suffixes below w): a suffix of B with rank
// r < R has i <= I, one with r >= R has i >= I (both ranks are monotone along the merged order), so the elements with r in
// [R_g, R_g+1) query A only inside [I_g, I_g+1] and B only inside [R_g, R_g+1).  The outputs of a step are sorted by r inside every
// (class, GPU) piece -- the inputs were, and LF is monotone inside a class -- so the elements of a piece that belong to GPU g are a
// contiguous range of it: below[c][k] = number of the piece's elements with r < R_k, found here by binary search through the
// exporting GPU's segment tables.  One thread per (class, cut); a prototype: every probe is a dependent chain of ~40 loads
// (a per-block sample of r would make it two).

constexpr u32 PARTITION_MAX_PARTS = 16;

__global__ void __launch_bounds__(BLOCK_THREADS) k_cut_counts(const uint2* lo, const unsigned short* hi, const u64* prefix, const u64* phys, u64 nbl,
  const u64* cuts, u32 ncuts, u64* below)
{
  const u32 t = threadIdx.x;
  if(t >= 5 * ncuts) { return; }
  const u32 c = t / ncuts, k = t - c * ncuts;
  const u64 cut = cuts[k];
  const u64 seg_first = (u64)c * nbl, seg_end = seg_first + nbl;
  const u64 x0 = prefix[seg_first], total = prefix[seg_end] - x0;
  u64 lo_x = 0, hi_x = total;                                        // elements [0, lo_x) are below the cut, [hi_x, total) are not
  while(lo_x < hi_x)
  {
    const u64 mid = (lo_x + hi_x) >> 1, x = x0 + mid;
    u64 lo_s = seg_first, hi_s = seg_end;                            // prefix[lo_s] <= x < prefix[hi_s]
    while(hi_s - lo_s > 1)
    {
      const u64 m = (lo_s + hi_s) >> 1;
      if(prefix[m] <= x) { lo_s = m; } else { hi_s = m; }
    }
    const u64 at = phys[lo_s] + (x - prefix[lo_s]);
    const u64 r = (u64)lo[at].x | (hi ? (u64)(hi[at] & 0xFF) << 32 : 0ull);      // the B coordinate (x; its high byte is the low byte of hi)
    if(r < cut) { lo_x = mid + 1; } else { hi_x = mid; }
  }
  below[t] = lo_x;
}

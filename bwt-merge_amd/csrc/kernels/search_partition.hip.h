// Partitioned search (experimental build): the frontier is cut at FIXED positions instead of into equal
// shares, so that GPU g only ever advances elements whose coordinates fall into its window of A's and of B's records.
//
// A cut is a pair (I, R) = (number of A's suffixes below some string w, number of B's suffixes below w): a suffix of B with rank
// i < R has at most I of A's suffixes below it, one with i >= R at least I (both ranks are monotone along the merged order), so the
// elements with B coordinate in [R_g, R_g+1) query B only inside that range and A only inside [I_g, I_g+1].  The outputs of a step are
// sorted by position inside every (class, GPU) piece -- the inputs were, and LF is monotone inside a class -- so the elements of a piece
// that belong to GPU g are a contiguous range of it.
//
// Per step, after the unchanged k_frontier_step:
//   k_compact_outputs   the GPU's outputs from the step kernel's physical layout (block by block, class runs back to back) into ONE dense
//                       array in logical order (class, block): the send buffer of the exchange -- peers read contiguous ranges of it
//                       instead of walking this GPU's segment tables across the fabric;
//   k_cut_search        below[c][k] = elements of class c with B coordinate < cut k: 5 (G - 1) binary searches on the dense array;
//   k_gather_dense      (on the receiving GPU) its elements from all peers' dense arrays: <= 5 G contiguous copies in class-major order,
//                       which is increasing position -- the input the step kernel expects.
// (Two earlier forms, measured at config 2 and G = 8 where a GPU's steps take 16.5 ms in all: counting by nested binary searches through
// the segment tables 7.5 ms and pulling through the peers' segment tables 14.6 ms; counting by a histogram pass with atomics 62 ms.)

constexpr u32 PARTITION_MAX_PARTS = 16;

// One thread per slot of the physical output layout: block b of the step kernel wrote its class runs back to back from slot b * FR_BLOCK
// on (k_frontier_step's epilogue; k_frontier_init for the roots), so a slot's class follows from the block's five segment lengths, and its
// logical index from the scanned segment table.
__global__ void __launch_bounds__(FR_BLOCK) k_compact_outputs(const uint2* lo, const unsigned short* hi, const u64* seg_len, const u64* seg_phys, const u64* prefix, u64 nbl,
  uint2* dense_lo, unsigned short* dense_hi)
{
  const u64 b = blockIdx.x;
  u32 len[5]; u32 total = 0;
#pragma unroll
  for(u32 c = 0; c < 5; c++) { len[c] = (u32)seg_len[(u64)c * nbl + b]; total += len[c]; }
  const u32 j = threadIdx.x;
  if(j >= total) { return; }
  u32 c = 0, base = 0;
#pragma unroll
  for(u32 k = 0; k < 4; k++) { if(c == k && j >= base + len[k]) { base += len[k]; c = k + 1; } }
  const u64 at = seg_phys[(u64)c * nbl + b] + (j - base);
  const u64 to = prefix[(u64)c * nbl + b] + (j - base);
  dense_lo[to] = lo[at];
  if(hi) { dense_hi[to] = hi[at]; }
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_cut_search(const uint2* dense_lo, const unsigned short* dense_hi, const u64* prefix, u64 nbl,
  const u64* cuts, u32 ncuts, u64* below)
{
  const u32 t = threadIdx.x;
  if(t >= 5 * ncuts) { return; }
  const u32 c = t / ncuts, k = t - c * ncuts;
  const u64 cut = cuts[k];
  const u64 first = prefix[(u64)c * nbl], end = prefix[(u64)(c + 1) * nbl];
  u64 lo_x = first, hi_x = end;                                      // elements [first, lo_x) are below the cut, [hi_x, end) are not
  while(lo_x < hi_x)
  {
    const u64 mid = (lo_x + hi_x) >> 1;
    const u64 pos = (u64)dense_lo[mid].x | (dense_hi ? (u64)(dense_hi[mid] & 0xFF) << 32 : 0ull);      // the B coordinate (x; its high byte is the low byte of hi)
    if(pos < cut) { lo_x = mid + 1; } else { hi_x = mid; }
  }
  below[t] = lo_x - first;
}

struct DensePiece
{
  const uint2* lo; const unsigned short* hi;     // the source GPU's dense outputs
  u64 src_first, count, dst_first;
};

__global__ void __launch_bounds__(BLOCK_THREADS) k_gather_dense(const DensePiece* pieces, u32 npieces, u64 n_in, uint2* lo_in, unsigned short* hi_in)
{
  const u64 j = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(j >= n_in) { return; }
  u32 q = 0;
  for(u32 k = 1; k < npieces; k++) { if(pieces[k].dst_first <= j) { q = k; } }      // pieces are sorted by dst_first; at most 5 G of them
  const DensePiece pc = pieces[q];
  const u64 at = pc.src_first + (j - pc.dst_first);
  lo_in[j] = pc.lo[at];
  if(hi_in) { hi_in[j] = pc.hi[at]; }
}

// (the node phase's routing kernels and the windows' interleave helper are product code since round 6: kernels/partition.hip.h)

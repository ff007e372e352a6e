/*
  kernels/partition.hip.h -- the merge over PARTITIONED records: what one part (GPU) runs around the unchanged
  search, interleave and encoder kernels.  Part of bwtm_kernels.hip.h (included there, inside namespace bwtm); gfx950 only.

  A cut is a pair (I, R) = (number of A's suffixes below some string w, number of B's suffixes below w).  Both ranks are monotone along
  the merged order, so the frontier elements with B coordinate in [R_g, R_g+1) query B only inside that range and A only inside
  [I_g, I_g+1]: part g holds one window of each index, its own range of the bitvector, and the elements that fall into its range.
  The outputs of a step are sorted by position inside every (class, part) piece -- the inputs were, and LF is monotone inside a class --
  so the elements of a piece that belong to part g are a contiguous range of it.

  Per LF step, on every part:
    k_frontier_step<.., PULL>  reads its input THROUGH a segment table whose entries point into the peers' output buffers (peer-mapped
                               device memory: xGMI loads on a node with several GPUs) -- no element is copied for the exchange;
    (scan of its own output table: k_frontier_scan1, as in the single-GPU search)
    k_cut_search_seg           where every cut falls in every class of its outputs, as (segment, offset inside it, elements below):
                               one workgroup per (class, cut), a 256-ary search in logical order through the scanned table;
    k_pull_tables              (on the receiving part) its next input's segment table: <= 5 x parts runs of the peers' tables, clipped at
                               the two cuts, every entry tagged with the part whose buffer it points into.
  The first levels run on trie NODES routed the same way (k_node_cut_search, k_gather_nodes).
*/
#pragma once

constexpr u32 PART_MAX = 16;                   // BWTM_MAX_PARTS
constexpr u32 PULL_SRC_SHIFT = PULL_SRC_SHIFT_; // a pulled segment's physical start carries the source part in its top byte (k_frontier_step<.., PULL>)
constexpr u32 PULL_ALL = 0xFFFFFFFFu;

// Where cut k falls in class c of a part's outputs: the first element at or above the cut lies in segment `seg` of the class (block number;
// = the number of blocks when every element is below), `off` elements of that segment lie below it, `below` elements of the class do.
struct CutEntry { u64 seg, off, below; };

// The B coordinate of logical element x of a scanned output table (entries e with prefix[e] <= x < prefix[e + 1]).
__device__ inline u64 cut_key(const uint2* lo, const unsigned short* hi, const u64* prefix, const u64* phys, const u32* first_seg, u64 nseg, u64 x, u64& entry)
{
  u64 e = first_seg[x >> 8];                                        // the entry that holds logical element (x >> 8) * 256
  u64 hi_e = (e + 32 < nseg ? e + 32 : nseg);                       // prefix[nseg] = all elements > x
  if(prefix[hi_e] <= x) { e = hi_e; hi_e = nseg; }                  // rare: beyond 32 entries (runs of empty segments between two classes)
  while(hi_e - e > 1)
  {
    const u64 mid = (e + hi_e) >> 1;
    if(prefix[mid] <= x) { e = mid; } else { hi_e = mid; }
  }
  entry = e;
  const u64 at = phys[e] + (x - prefix[e]);
  return (u64)lo[at].x | (hi ? (u64)(hi[at] & 0xFF) << 32 : 0ull);
}

// One workgroup per (class, cut 1 .. parts - 1): the smallest logical index of the class whose element is not below the cut.  One more
// workgroup writes the entries of cut 0 (nothing below) and of cut `parts` (everything below) of every class.
__global__ void __launch_bounds__(BLOCK_THREADS) k_cut_search_seg(const uint2* lo, const unsigned short* hi, const u64* prefix, const u64* phys, const u32* first_seg, u64 nb,
  const u64* cuts, u32 parts, u32 stride, CutEntry* out)
{
  __shared__ u32 s_cnt[BLOCK_THREADS / WAVE];
  const u32 ncuts = parts - 1;
  if(blockIdx.x == 5 * ncuts)
  {
    if(threadIdx.x < 5)
    {
      const u32 c = threadIdx.x;
      CutEntry z; z.seg = 0; z.off = 0; z.below = 0;
      CutEntry all; all.seg = nb; all.off = 0; all.below = prefix[(u64)(c + 1) * nb] - prefix[(u64)c * nb];
      out[c * stride] = z; out[c * stride + parts] = all;
    }
    return;
  }
  const u32 c = blockIdx.x / ncuts, k = blockIdx.x % ncuts + 1;
  const u64 cut = cuts[k];
  const u64 nseg = 5 * nb;
  const u64 first = prefix[(u64)c * nb], end = prefix[(u64)(c + 1) * nb];
  u64 a = first, b = end;                                           // the answer lies in [a, b]
  while(b > a)
  {
    const u64 step = (b - a + BLOCK_THREADS - 1) / BLOCK_THREADS;
    const u64 x = a + (u64)threadIdx.x * step;
    bool is_below = false;
    u64 e;
    if(x < b) { is_below = (cut_key(lo, hi, prefix, phys, first_seg, nseg, x, e) < cut); }
    const u64 m = __ballot(is_below);
    if(lane_id() == 0) { s_cnt[threadIdx.x >> 6] = (u32)__builtin_popcountll(m); }
    __syncthreads();
    u32 T = 0;
    for(int w = 0; w < BLOCK_THREADS / WAVE; w++) { T += s_cnt[w]; }
    __syncthreads();
    if(T == 0) { b = a; }                                           // the first probe (a itself) is at or above the cut
    else
    {
      const u64 last_below = a + (u64)(T - 1) * step;               // probes are monotone: exactly the first T are below
      a = last_below + 1;
      if(last_below + step < b) { b = last_below + step; }          // the next probe is at or above the cut
    }
  }
  if(threadIdx.x == 0)
  {
    CutEntry r; r.below = a - first;
    if(a >= end) { r.seg = nb; r.off = 0; }
    else
    {
      u64 e; (void)cut_key(lo, hi, prefix, phys, first_seg, nseg, a, e);
      r.seg = e - (u64)c * nb; r.off = a - prefix[e];
    }
    out[c * stride + k] = r;
  }
}

// A run of a peer's segment table inside this part's next input.
struct PullPiece
{
  u64 src_first;        // first entry of the run in the source's tables (class * its block count + first segment)
  u32 dst_first;        // where the run begins in this part's table
  u32 count;            // entries
  u32 src;              // source part
  u32 clip_first;       // elements of the run's first segment that lie below this part's lower cut
  u32 last_len;         // elements of the run's last segment below this part's upper cut, counted from the segment's start (PULL_ALL: all)
  u32 pad;
};

struct PullPlan
{
  PullPiece piece[5 * PART_MAX];
  const u64* seg_len[PART_MAX]; const u64* seg_phys[PART_MAX];      // the parts' output tables of this step (peer-mapped)
  u32 npieces, nseg;                                                // runs; entries of this part's table (the entry behind them is set to 0)
};

// The plan travels as a kernel ARGUMENT (2.8 KB of the 4 KB a launch may carry): no copy to the device ahead of the launch -- one stream
// operation less in every step (round 6: a step with nothing to do takes 59 us, every operation of it counts) -- and the runs are staged in
// LDS once per workgroup, so that a thread's look-up of its run is not one more dependent load from memory.
struct PullPieces { PullPiece piece[5 * PART_MAX]; u64 table[2][PART_MAX]; };

__device__ inline void stage_plan(const PullPlan& plan, PullPieces& s)
{
  if(threadIdx.x < plan.npieces) { s.piece[threadIdx.x] = plan.piece[threadIdx.x]; }
  if(threadIdx.x < PART_MAX) { s.table[0][threadIdx.x] = (u64)(uintptr_t)plan.seg_len[threadIdx.x]; s.table[1][threadIdx.x] = (u64)(uintptr_t)plan.seg_phys[threadIdx.x]; }
  __syncthreads();
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_pull_tables(const PullPlan plan, u64* seg_len, u64* seg_phys)
{
  __shared__ PullPieces sp;
  const u32 np = plan.npieces, nseg = plan.nseg;
  stage_plan(plan, sp);
  const u64 idx = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(idx == nseg) { seg_len[idx] = 0; seg_phys[idx] = 0; }
  if(idx >= nseg) { return; }
  u32 q = 0;
  for(u32 step = 64; step != 0; step >>= 1) { if(q + step < np && sp.piece[q + step].dst_first <= idx) { q += step; } }      // the runs are sorted by dst_first
  const PullPiece pc = sp.piece[q];
  const u32 j = (u32)idx - pc.dst_first;
  const u64 e = pc.src_first + j;
  u64 len = peer_load((const u64*)(uintptr_t)sp.table[0][pc.src] + e), at = peer_load((const u64*)(uintptr_t)sp.table[1][pc.src] + e);
  if(j + 1 == pc.count && pc.last_len != PULL_ALL && len > pc.last_len) { len = pc.last_len; }
  if(j == 0) { const u64 skip = (pc.clip_first < len ? pc.clip_first : len); len -= skip; at += skip; }
  seg_len[idx] = len;
  seg_phys[idx] = at | ((u64)pc.src << PULL_SRC_SHIFT);
}

// k_pull_tables and the scan of the pulled table (k_frontier_scan1) in ONE launch: every workgroup pulls the 2048 entries of its tile, publishes
// the tile's total as a tagged word and waits for the tiles before it, as k_frontier_scan1 does (at most FRONTIER_SCAN1_TILES tiles, all resident).
__global__ void __launch_bounds__(BLOCK_THREADS) k_pull_scan1(const PullPlan plan, unsigned long long* tile_total, u32 tag, u64* seg_phys, u64* seg_prefix, u32* first_seg,
  u64* emit_base, u64 step)
{
  __shared__ PullPieces sp;
  __shared__ u64 lds[BLOCK_THREADS / WAVE];
  const u32 np = plan.npieces;
  const u64 nseg = plan.nseg, n = nseg + 1;                           // the entry after the last segment holds 0 and receives N_t
  stage_plan(plan, sp);
  const u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
  u64 item[SCAN_ITEMS], at[SCAN_ITEMS];
  u64 acc = 0;
  // the thread's eight consecutive entries: the run of the first one by binary search, the following ones by stepping
  u32 q = 0;
  if(base < nseg) { for(u32 st = 64; st != 0; st >>= 1) { if(q + st < np && sp.piece[q + st].dst_first <= base) { q += st; } } }
  PullPiece pc = sp.piece[q];
#pragma unroll
  for(int k = 0; k < SCAN_ITEMS; k++)
  {
    const u64 idx = base + k;
    item[k] = 0; at[k] = 0;
    if(idx < nseg)
    {
      while(q + 1 < np && sp.piece[q + 1].dst_first <= idx) { q++; pc = sp.piece[q]; }
      const u32 j = (u32)idx - pc.dst_first;
      const u64 e = pc.src_first + j;
      u64 len = peer_load((const u64*)(uintptr_t)sp.table[0][pc.src] + e), a = peer_load((const u64*)(uintptr_t)sp.table[1][pc.src] + e);
      if(j + 1 == pc.count && pc.last_len != PULL_ALL && len > pc.last_len) { len = pc.last_len; }
      if(j == 0) { const u64 skip = (pc.clip_first < len ? pc.clip_first : len); len -= skip; a += skip; }
      item[k] = len; at[k] = a | ((u64)pc.src << PULL_SRC_SHIFT);
    }
    acc += item[k];
  }
  const u64 incl = wave_incl_sum(acc);
  const u64 wave_total = shfl_u64(incl, WAVE - 1);
  u64 excl = shfl_up_u64(incl, 1);
  if(lane_id() == 0) { excl = 0; }
  if(lane_id() == 0) { lds[threadIdx.x >> 6] = wave_total; }
  __syncthreads();
  u64 before_waves = 0, tile_sum = 0;
  for(int k = 0; k < BLOCK_THREADS / WAVE; k++) { if(k < (int)(threadIdx.x >> 6)) { before_waves += lds[k]; } tile_sum += lds[k]; }
  if(threadIdx.x == 0) { __hip_atomic_store(&tile_total[blockIdx.x], ((unsigned long long)tag << 32) | (unsigned long long)tile_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __syncthreads();                                                     // lds is reused below
  u64 c = 0;
  for(u64 k = threadIdx.x; k < blockIdx.x; k += BLOCK_THREADS)
  {
    unsigned long long w;
    do { w = __hip_atomic_load(&tile_total[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while((u32)(w >> 32) != tag);
    c += (u32)w;
  }
  const u64 carry = block_reduce<0>(c, lds);
  u64 run = carry + before_waves + excl;
  for(int k = 0; k < SCAN_ITEMS; k++)
  {
    const u64 idx = base + k;
    if(idx < n)
    {
      seg_prefix[idx] = run;
      if(idx < nseg)
      {
        seg_phys[idx] = at[k];
        const u64 b = (run + FR_BLOCK - 1) / FR_BLOCK;
        if(b * FR_BLOCK < run + item[k]) { first_seg[b] = (u32)idx; }
      }
      else { seg_phys[idx] = 0; emit_base[step + 1] = emit_base[step] + run; }
    }
    run += item[k];
  }
}

// ---- the node phase over partitioned records.  A level's nodes (sp, count, r) live on the GPU that owns sp; with cuts at k-mer boundaries a
// node never crosses a cut (the suffixes "x$" of a node x sort before every "x y...": a cut lies before or after all of them), so
// k_range_step runs on a window unchanged.  Its children come out symbol-major, i.e. sorted by sp, class after class: the children of class
// c that belong to GPU k are again a contiguous range, found here by binary search; a child that crosses a cut is reported (err).
__global__ void __launch_bounds__(BLOCK_THREADS) k_node_cut_search(const u64* sp, const u64* cnt, const u64* class_first /* 6 */, const u64* cuts, u32 ncuts, u64* below, u32* err)
{
  const u32 t = threadIdx.x;
  if(t >= 5 * ncuts) { return; }
  const u32 c = t / ncuts, k = t - c * ncuts;
  const u64 cut = cuts[k];
  const u64 first = class_first[c], end = class_first[c + 1];
  u64 lo_x = first, hi_x = end;
  while(lo_x < hi_x)
  {
    const u64 mid = (lo_x + hi_x) >> 1;
    if(sp[mid] < cut) { lo_x = mid + 1; } else { hi_x = mid; }
  }
  if(lo_x > first && cut != ~0ull && sp[lo_x - 1] + cnt[lo_x - 1] > cut) { atomicOr(err, 1u); }      // the node before the cut reaches across it
  below[t] = lo_x - first;
}

struct NodePiece
{
  const u64* sp; const u64* r; const u64* cnt;     // the source GPU's children
  u64 src_first, count, dst_first;
};

__global__ void __launch_bounds__(BLOCK_THREADS) k_gather_nodes(const NodePiece* pieces, u32 npieces, u64 n, u64* sp, u64* r, u64* cnt)
{
  const u64 j = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(j >= n) { return; }
  u32 q = 0;
  for(u32 k = 1; k < npieces; k++) { if(pieces[k].dst_first <= j) { q = k; } }
  const NodePiece pc = pieces[q];
  const u64 at = pc.src_first + (j - pc.dst_first);
  sp[j] = peer_load(&pc.sp[at]); r[j] = peer_load(&pc.r[at]); cnt[j] = peer_load(&pc.cnt[at]);
}

// dst |= the boundary row of another part (8 KiB in its exported buffers).
__global__ void __launch_bounds__(BLOCK_THREADS) k_bits_or_peer(u64* dst, const u64* src, u64 nwords)
{
  const u64 w = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(w < nwords) { const u64 v = peer_load(&src[w]); if(v != 0) { dst[w] |= v; } }
}

// ---- interleave of an output range from WINDOWS (bwtm_interleave_range on bwtm_x_index_window handles).  The super table of a slice only
// serves the slice's own records, and a row only has to lie at or below the counts of every record that refers to it (header fields are
// 25-bit offsets from it): the rows of the supers that BEGIN inside the range are the usual ones -- their positions lie inside the windows --
// and the row of the super the range begins in, whose own beginning belongs to another GPU, is taken at the range's first chunk instead.
// Rows of other supers stay zero: nothing of this slice refers to them.  (k_interleave_sup asks A and B at EVERY super of the output.)
__global__ void __launch_bounds__(BLOCK_THREADS) k_interleave_sup_window(IndexView A, IndexView B, const u64* chunk_base, u64* sup, u64 nsup, const u64* super_boff,
  u64 q_base, u64 q_end)
{
  const u64 s = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(s >= nsup) { return; }
  u64 q = s << SUPER_REC_SHIFT;
  bool have = (q >= q_base && q < q_end);
  u64 b_off = 0;
  if(have) { b_off = super_boff[s]; }
  else if(s == (q_base >> SUPER_REC_SHIFT) && q_base < q_end) { q = q_base; b_off = chunk_base[q_base >> 6]; have = true; }      // q_base is the first record of a chunk
  u64 ra[6] = {0, 0, 0, 0, 0, 0}, rb[6] = {0, 0, 0, 0, 0, 0};
  if(have)
  {
    u64 a_off = (q << REC_SHIFT) - b_off;
    if(a_off > A.n) { a_off = A.n; }
    if(b_off > B.n) { b_off = B.n; }
    index_ranks(A, a_off, ra); index_ranks(B, b_off, rb);
  }
  for(int c = 0; c < SUP_STRIDE; c++) { sup[s * SUP_STRIDE + c] = (c >= 1 && c < 6 ? ra[c] + rb[c] : 0); }
}

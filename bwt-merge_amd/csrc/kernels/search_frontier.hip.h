/*
  kernels/search_frontier.hip.h -- the search in level-synchronous form (what bwtm_search runs for shards of >= 2^21 sequences of a read collection).
  Part of bwtm_kernels.hip.h (included there, inside namespace bwtm); gfx950 only.
*/
#pragma once

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
constexpr u32 PULL_SRC_SHIFT_ = 56;            // = PULL_SRC_SHIFT of kernels/partition.hip.h (included after this file)
constexpr u64 PULL_PHYS_MASK_ = (1ull << PULL_SRC_SHIFT_) - 1;
constexpr int FR_SEGS = 31;                    // segment-table entries staged per block (a power of two minus one: k_frontier_step searches them in five steps)

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
  u32 block_base;                              // first block of this launch (0 unless the step is launched in slices)
  // PULL (the merge over partitioned records, kernels/partition.hip.h): the current frontier lies in the output buffers of up to 16 parts;
  // seg_phys carries the part in its top byte and these device arrays (16 entries each) hold the parts' buffers as this GPU maps them
  const uint2* const* src_lo; const unsigned short* const* src_hi;
  u64 nseg_in;                                 // PULL: entries of the pulled table (seg_prefix / seg_phys / first_seg); the outputs' tables still have 5 * nb_max
};

__global__ void __launch_bounds__(BLOCK_THREADS) k_frontier_init(uint2* lo, unsigned short* hi, u64* seg_len, u64* seg_phys, u64 nb_max,
  u64 seq_first, u64 count, u64 m_a)
{
  u64 g = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(g < count)
  {
    u64 i = seq_first + g;                                                // fmi.cpp:286: trie root "$"
    lo[g] = make_uint2((u32)i, (u32)m_a);
    if(hi) { hi[g] = (unsigned short)(((i >> 32) & 0xFF) | (((m_a >> 32) & 0xFF) << 8)); }      // no high bytes when both indexes are below 2^32 positions
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

// (Loads with the non-temporal hint are 33 - 40 % slower here, on thin frontiers too: profiles/r02_shard_search_time.log.)
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

// HI: the coordinates have high bytes (an index of 2^32 positions or more); otherwise the 2-byte array is neither read nor written.
// LF on the ordinary records, one 16-byte chunk at a time (the view's overflow path: rare, and it must not take sixteen registers from
// the common path).  want_symbol: c is first read from the record (the symbol at pos; 0 ends the chain and returns 0).
__device__ inline u64 ordinary_lf(const IndexView& X, u64 pos, u32& c, bool want_symbol)
{
  const uint4* p = X.recs + 4 * (pos >> REC_SHIFT);
  const u32 j = (u32)(pos & (REC_POS - 1));
  if(want_symbol)
  {
    const uint4 ch = p[j >> 5];
    const u32 t = j & 31;
    c = ((ch.x >> t) & 1u) | (((ch.y >> t) & 1u) << 1) | (((ch.z >> t) & 1u) << 2);
  }
  if(c == 0) { return 0; }
  u32 total = 0; u32 h[4];
#pragma unroll
  for(u32 k = 0; k < 4; k++)
  {
    const uint4 ch = p[k];
    total += (u32)__builtin_popcount(plane_match(ch.x, ch.y, ch.z, c) & below_mask(j, k));
    h[k] = ch.w;
  }
  const u32 sh = FIELD_BITS * (c - 1);
  const u64 lo = (u64)h[0] | ((u64)h[1] << 32), hi = (u64)h[2] | ((u64)h[3] << 32);
  u64 v;
  if(sh < 64) { v = lo >> sh; if(sh + FIELD_BITS > 64) { v |= hi << (64 - sh); } }
  else { v = hi >> (sh - 64); }
  return X.sup[(pos >> SUPER_SHIFT) * SUP_STRIDE + c] + ((u32)v & FIELD_MASK) + total;
}

// VIEW (instantiated only with -DBWTM_EXPERIMENTAL): the records come from the search view (160 positions per 64 bytes, bwtm_view.h); an
// element whose view record has overflowed its exception slots reads the ordinary record of its position instead.
// PULL: the elements are read from the parts' output buffers through a pulled segment table (k_pull_tables): one more LDS lookup per element.
template<int EMIT, bool HI, bool VIEW = false, bool PULL = false>
__global__ void __launch_bounds__(FR_BLOCK, (VIEW ? 6 : 8)) k_frontier_step(IndexView A, IndexView B, FrontierView f)
{
  __shared__ u32 wave_cnt[FR_BLOCK / WAVE][6];
  __shared__ u64 s_prefix[FR_SEGS + 1], s_phys[FR_SEGS + 1];
  __shared__ const uint2* s_src_lo[PULL ? 16 : 1];
  __shared__ const unsigned short* s_src_hi[PULL ? 16 : 1];
  const u64 nseg_out = 5 * f.nb_max;
  const u64 nseg = (PULL ? f.nseg_in : nseg_out);                 // entries of the table the frontier is read through
  const u64 N = f.seg_prefix[nseg];
  const u32 bid = blockIdx.x + f.block_base;                      // a launch may cover a range of the step's blocks (one slice of the frontier)
  const u64 g0 = (u64)bid * FR_BLOCK;
  const u32 lane = lane_id(), wave = threadIdx.x >> 6;

  // Blocks past the frontier only publish empty segments.
  if(g0 >= N)
  {
    if(threadIdx.x < 5) { f.seg_len_next[(u64)threadIdx.x * f.nb_max + bid] = 0; f.seg_phys_next[(u64)threadIdx.x * f.nb_max + bid] = g0; }
    if(bid == 0 && threadIdx.x == 5) { f.seg_len_next[nseg_out] = 0; }
    return;
  }
  // The block's elements live in a handful of segments: stage their table entries in LDS.
  const u64 first_seg = f.first_seg[bid];
  if(threadIdx.x <= FR_SEGS)
  {
    u64 sidx = first_seg + threadIdx.x; if(sidx > nseg) { sidx = nseg; }
    s_prefix[threadIdx.x] = f.seg_prefix[sidx];
    s_phys[threadIdx.x] = f.seg_phys[sidx < nseg ? sidx : nseg - 1];
  }
  if(PULL && threadIdx.x >= 64 && threadIdx.x < 80)               // (another wave than the one that stages the table entries)
  {
    s_src_lo[threadIdx.x - 64] = f.src_lo[threadIdx.x - 64];
    if(HI) { s_src_hi[threadIdx.x - 64] = f.src_hi[threadIdx.x - 64]; }
  }
  __syncthreads();

  const u64 g = g0 + threadIdx.x;
  const bool active = (g < N);
  u64 i = 0, r = 0;
  if(active)
  {
    u64 phys;
    // the staged entry that holds g: the largest k with s_prefix[k] <= g (empty segments repeat their neighbour's prefix).  Five
    // steps of a binary search: the linear walk was unrolled by the compiler into 31 compare-select steps on 64-bit values,
    // ~190 of the kernel's ~470 VALU instructions per wave.
    u32 k = 0;
#pragma unroll
    for(u32 step = 16; step != 0; step >>= 1) { if(s_prefix[k + step] <= g) { k += step; } }     // k + step <= 31 = FR_SEGS: inside the FR_SEGS + 1 staged entries
    if(k < (u32)FR_SEGS) { phys = s_phys[k] + (g - s_prefix[k]); }      // (PULL: the part in the top byte survives the addition)
    else
    {
      // Rare: the element lies beyond the staged entries.  Binary search of seg_prefix (non-decreasing) for the segment
      // with prefix <= g < next prefix: when many chains end at once (reads of mixed lengths), the blocks of every class
      // beyond the shrunken frontier are empty segments, tens of thousands in a row between two classes, and a linear
      // walk over them by the few lanes that straddle a class boundary made the whole launch 10 x slower (measured).
      u64 lo_s = first_seg + FR_SEGS, hi_s = nseg;                   // seg_prefix[lo_s] <= g < seg_prefix[hi_s] = N
      while(hi_s - lo_s > 1)
      {
        const u64 mid = (lo_s + hi_s) >> 1;
        if(f.seg_prefix[mid] <= g) { lo_s = mid; } else { hi_s = mid; }
      }
      const u64 sgm = lo_s;
      phys = f.seg_phys[sgm] + (g - f.seg_prefix[sgm]);
    }
    const uint2* lo_src = f.lo; const unsigned short* hi_src = f.hi;
    if(PULL)
    {
      const u32 src = (u32)(phys >> PULL_SRC_SHIFT_);
      phys &= PULL_PHYS_MASK_;
      lo_src = s_src_lo[src]; if(HI) { hi_src = s_src_hi[src]; }
    }
    uint2 l; u32 h = 0;
    if(PULL) { l = peer_load(&lo_src[phys]); if(HI) { h = (u32)peer_load(&hi_src[phys]); } }      // another GPU's kernel wrote them before the step's exchange
    else { l = lo_src[phys]; if(HI) { h = (u32)hi_src[phys]; } }
    __builtin_amdgcn_sched_barrier(0);      // both loads are issued before either is used (the scheduler otherwise waits for the high bytes before it issues the other load: one more round trip per wave)
    i = (u64)l.x | ((u64)(h & 0xFF) << 32);
    r = (u64)l.y | ((u64)(h >> 8) << 32);
  }
  const u64 any_active = __ballot(active);
  u32 tile_first = 0xFFFFFFFFu, tile_last = 0xFFFFFFFFu;           // tiles of the wave's first / last element (EMIT == 0)
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
      // A lane marks when the previous lane lies in another tile.  Lane 0 does not know its predecessor: it
      // marks after the partition barrier below, unless the previous wave of the block ended in the same tile
      // (the first wave of a block always marks; the true first element of the tile marks too and wins the
      // minimum).  Tiles without elements are filled in by k_bound_suffix_min.
      const u64 p = i + r;
      const u64 my_tile = p >> TILE_SHIFT;
      const u64 prev_tile = shfl_up_u64(my_tile, 1);
      if(active)
      {
        u64 slot = f.emit_base[f.step] + g;
        if(slot < f.emit_cap) { nt_store(&f.emit16[slot], (unsigned short)(p & TILE_MASK)); }
        else { sink_fallback(f.bits32, p); }                      // exact fallback; k_tile_build_frontier skips these slots
        if(lane != 0 && my_tile != prev_tile) { atomicMin(&f.bound_row[my_tile], (u32)g); }
      }
      tile_first = (u32)my_tile;                                  // meaningful in lane 0
      tile_last = (u32)shfl_u64(my_tile, (int)last_lane);
    }
#ifdef BWTM_EXPERIMENTAL
    if(VIEW)
    {
      // The same step on the view records.  Position p sits in view record p / 160 (a multiply-high), at p - 160 (p / 160).
      const u64 pi = (active ? i : li), pr = (active ? r : lr);
      const u64 qb = pi / VIEW_POS, qa = pr / VIEW_POS;
      const u32 jb = (u32)(pi - qb * VIEW_POS), ja = (u32)(pr - qa * VIEW_POS);
      RecordFetch fb = record_issue(B.view, B.nview, qb);
      RecordFetch fa = record_issue(A.view, A.nview, qa);
      // view-super rows of lane 0 (always active here) through scalar loads, requested together with the records
      const u64 vs_b0 = shfl_u64(qb, 0) >> VIEW_SUPER_SHIFT, vs_a0 = shfl_u64(qa, 0) >> VIEW_SUPER_SHIFT;
      const u64* vrow_b = B.vsup + vs_b0 * SUP_STRIDE;
      const u64* vrow_a = A.vsup + vs_a0 * SUP_STRIDE;
      bool over_b, over_a;
      {
        u32 wb[16];
        record_words(fb, wb);
        over_b = view_overflow(wb);
        if(active && !over_b)
        {
          u32 below, below_n, at;
          view_exceptions(wb[14], wb[15], jb, below, below_n, at);
          c = view_symbol(wb, jb, at);
          if(c != 0)
          {
            u64 sb;
            if((qb >> VIEW_SUPER_SHIFT) == vs_b0) { sb = (c == 1 ? vrow_b[1] : (c == 2 ? vrow_b[2] : (c == 3 ? vrow_b[3] : (c == 4 ? vrow_b[4] : vrow_b[5])))); }
            else { sb = B.vsup[(qb >> VIEW_SUPER_SHIFT) * SUP_STRIDE + c]; }
            ni = sb + view_header(wb, c) + view_count(wb, c, jb, below, below_n);
          }
        }
      }
      // rare: more than seven endmarkers / N among the 160 positions -- the ordinary record of the position answers
      if(__ballot(active && over_b) != 0) { if(active && over_b) { ni = ordinary_lf(B, i, c, true); } }
      {
        u32 wa[16];
        record_words(fa, wa);
        over_a = view_overflow(wa);
        if(active && c != 0 && !over_a)
        {
          u64 sa;
          if((qa >> VIEW_SUPER_SHIFT) == vs_a0) { sa = (c == 1 ? vrow_a[1] : (c == 2 ? vrow_a[2] : (c == 3 ? vrow_a[3] : (c == 4 ? vrow_a[4] : vrow_a[5])))); }
          else { sa = A.vsup[(qa >> VIEW_SUPER_SHIFT) * SUP_STRIDE + c]; }
          u32 below, below_n, at;
          view_exceptions(wa[14], wa[15], ja, below, below_n, at);
          nr = sa + view_header(wa, c) + view_count(wa, c, ja, below, below_n);
        }
      }
      if(__ballot(active && c != 0 && over_a) != 0) { if(active && c != 0 && over_a) { nr = ordinary_lf(A, r, c, false); } }
      if(active && c != 0)
      {
        const u64 cb = (c == 1 ? B.C[1] : (c == 2 ? B.C[2] : (c == 3 ? B.C[3] : (c == 4 ? B.C[4] : B.C[5]))));
        const u64 ca = (c == 1 ? A.C[1] : (c == 2 ? A.C[2] : (c == 3 ? A.C[3] : (c == 4 ? A.C[4] : A.C[5]))));
        ni += cb; nr += ca;                                       // LF_B(i), LF_A(r, c): utils.h:335-348
      }
    }
    else
#endif
    {
      u32 wb[16];
      const u64 rec_b = (active ? i : li) >> REC_SHIFT, rec_a = (active ? r : lr) >> REC_SHIFT;
      RecordFetch fb = record_issue(B.recs, B.nrecs, rec_b);
      RecordFetch fa = record_issue(A.recs, A.nrecs, rec_a);     // in flight while B's record is used
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
  if(lane == 0) { wave_cnt[wave][0] = tile_last; for(u32 k = 1; k < 6; k++) { wave_cnt[wave][k] = cnt_w[k]; } }
  // Raw barrier with an LDS-only wait: __syncthreads() would also drain vmcnt and expose the latency
  // of the emit reservation / stores that are still in flight (measured: +35 ms per search).
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // lane 0's tile marker is issued last: an atomic here would make the wave wait for it (vmcnt(0)) as soon as a register it reads is reused below
  const bool mark_first = (EMIT == 0 && lane == 0 && active && (wave == 0 || wave_cnt[wave - 1][0] != tile_first));
  // Positions inside the block: totals per class (all waves read the same LDS words), then the element's own store first;
  // the segment entries of the block are written by five lanes of wave 0 afterwards, behind a wave-uniform branch, so that the
  // other waves retire without passing any store inside divergent control flow (where the compiler waits for vmcnt(0)).
  u32 class_base = 0, before_waves = 0;
  u32 tot_k[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for(u32 k = 1; k < 6; k++)
  {
    u32 tot = 0, bw = 0;
    for(u32 w2 = 0; w2 < FR_BLOCK / WAVE; w2++) { u32 v = wave_cnt[w2][k]; if(w2 < wave) { bw += v; } tot += v; }
    if(k < c) { class_base += tot; }
    if(k == c) { before_waves = bw; }
    tot_k[k] = tot;
  }
  if(active && c != 0)
  {
    u64 dst = g0 + class_base + before_waves + my_rank;
    // streaming stores (nt_store): the next frontier and the emits are not read again by this launch; past the L2 they leave it to the records
    // (-2 to -3 ms of 92 per merge at config 2, profiles/r05_nt_stores_ab.txt; non-temporal LOADS of the coordinates on top: no change)
    nt_store(&f.lo_next[dst], make_uint2((u32)ni, (u32)nr));
    if(HI) { nt_store(&f.hi_next[dst], (unsigned short)(((ni >> 32) & 0xFF) | (((nr >> 32) & 0xFF) << 8))); }
  }
  if(wave == 0)
  {
    const u32 kk = threadIdx.x + 1;                                   // class of lanes 0 .. 4
    u32 tot = 0, base_k = 0;
#pragma unroll
    for(u32 k = 1; k < 6; k++) { if(k == kk) { tot = tot_k[k]; } if(k < kk) { base_k += tot_k[k]; } }
    if(kk < 6)
    {
      f.seg_len_next[(u64)(kk - 1) * f.nb_max + bid] = tot;
      f.seg_phys_next[(u64)(kk - 1) * f.nb_max + bid] = g0 + base_k;
    }
    if(bid == 0 && threadIdx.x == 5) { f.seg_len_next[nseg_out] = 0; }
  }
  if(mark_first) { atomicMin(&f.bound_row[tile_first], (u32)g); }
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
constexpr u64 FRONTIER_SCAN1_TILES = 1024;      // the one-launch form (k_frontier_scan1) below: all its workgroups must be co-resident

// host_n (may be null): page-locked host memory the frontier's size N_t is written to directly -- a separate 8-byte copy command between
// this kernel and the step kernel cost ~14 us of idle device per LF step (profiles/r03_config2_summary.md, gap table).
__global__ void __launch_bounds__(BLOCK_THREADS) k_frontier_scan(const u64* seg_len, const u64* partial, u64 nseg, u64* seg_prefix, u32* first_seg,
  u64* emit_base, u64 step, u64* host_n)
{
  __shared__ u64 lds[BLOCK_THREADS / WAVE];
  const u64 n = nseg + 1;                                            // the entry after the last segment holds 0 and receives N_t
  u64 c = 0;
  for(u64 k = threadIdx.x; k < blockIdx.x; k += BLOCK_THREADS) { c += partial[k]; }
  const u64 carry = block_reduce<0>(c, lds);
  const u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
  u64 item[SCAN_ITEMS];
  u64 acc = 0;
#pragma unroll
  for(int k = 0; k < SCAN_ITEMS; k++) { item[k] = seg_len[base + k < n ? base + k : n - 1]; }      // unconditional: see k_scan_reduce
#pragma unroll
  for(int k = 0; k < SCAN_ITEMS; k++) { if(base + k >= n) { item[k] = 0; } acc += item[k]; }
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
      else { emit_base[step + 1] = emit_base[step] + run; if(host_n) { *host_n = run; } }
    }
    run += item[k];
  }
}

// The same in ONE launch (round 5): every block first publishes the total of its own tile -- (tag << 32) | total in ONE 8-byte word, stored
// and polled at agent scope -- and then reads the words of all tiles before it, waiting for those that still carry the previous step's
// tag.  All blocks of the launch are resident together (at most FRONTIER_SCAN1_TILES tiles, one block each: four workgroups per CU where seven
// fit) and a block publishes before it waits, so nothing can wait for a block that has not started -- in whatever order blocks are dispatched.  `tag` changes with every step (never 0; the array is
// cleared once per search) and every tile rewrites its word in every step, so a stale word is always the previous step's.  Totals fit 32 bits:
// the frontier search is only used below 2^32 sequences per call.  Saves one launch and one kernel-to-kernel gap per LF step (~12 us of ~31).
__global__ void __launch_bounds__(BLOCK_THREADS) k_frontier_scan1(const u64* seg_len, unsigned long long* tile_total, u32 tag, u64 nseg, u64* seg_prefix, u32* first_seg,
  u64* emit_base, u64 step, u64* host_n)
{
  __shared__ u64 lds[BLOCK_THREADS / WAVE];
  const u64 n = nseg + 1;                                            // the entry after the last segment holds 0 and receives N_t
  const u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
  u64 item[SCAN_ITEMS];
  u64 acc = 0;
#pragma unroll
  for(int k = 0; k < SCAN_ITEMS; k++) { item[k] = seg_len[base + k < n ? base + k : n - 1]; }      // unconditional: see k_scan_reduce
#pragma unroll
  for(int k = 0; k < SCAN_ITEMS; k++) { if(base + k >= n) { item[k] = 0; } acc += item[k]; }
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
        const u64 b = (run + FR_BLOCK - 1) / FR_BLOCK;
        if(b * FR_BLOCK < run + item[k]) { first_seg[b] = (u32)idx; }
      }
      else { emit_base[step + 1] = emit_base[step] + run; if(host_n) { *host_n = run; } }
    }
    run += item[k];
  }
}

// Row t of the boundary table: bound[T] = logical index of the first element of step t whose bit
// position lies in tile >= T (suffix minimum over the markers; N_t past the last element).
// Two launches over (segment of BOUND_SEG tiles, step): the minimum of every segment, then every segment takes the minimum of
// the segments after it and finishes its own entries.  (One workgroup per row walked 600 chunks in sequence: 1.5 ms at
// config 2 on 104 of 256 CUs.)
constexpr u32 BOUND_SEG = 2048;

__global__ void __launch_bounds__(BLOCK_THREADS) k_bound_seg_min(const u32* bound, u64 ntiles, u64 nsegs, u32* segmin)
{
  __shared__ u32 lds[BLOCK_THREADS / WAVE];
  const u64 t = blockIdx.y, sg = blockIdx.x;
  const u32* row = bound + t * (ntiles + 1);
  u32 m = 0xFFFFFFFFu;
  for(u64 idx = sg * BOUND_SEG + threadIdx.x; idx < ntiles && idx < (sg + 1) * BOUND_SEG; idx += BLOCK_THREADS) { const u32 v = row[idx]; m = (v < m ? v : m); }
  m = 0xFFFFFFFFu - (u32)wave_max((u64)(0xFFFFFFFFu - m));
  if(lane_id() == 0) { lds[threadIdx.x >> 6] = m; }
  __syncthreads();
  if(threadIdx.x == 0)
  {
    for(int k = 1; k < BLOCK_THREADS / WAVE; k++) { m = (lds[k] < m ? lds[k] : m); }
    segmin[t * nsegs + sg] = m;
  }
}

__global__ void __launch_bounds__(BLOCK_THREADS) k_bound_suffix_min(u32* bound, u64 ntiles, u64 nsegs, const u32* segmin, const u64* emit_base)
{
  __shared__ u32 lds[BLOCK_THREADS];
  __shared__ u32 later[BLOCK_THREADS / WAVE];
  const u64 t = blockIdx.y, sg = blockIdx.x;
  u32* row = bound + t * (ntiles + 1);
  const u32 n_t = (u32)(emit_base[t + 1] - emit_base[t]);        // N_t
  if(sg + 1 == nsegs && threadIdx.x == 0) { row[ntiles] = n_t; }
  // minimum over the segments after this one
  u32 m = n_t;
  for(u64 k = sg + 1 + threadIdx.x; k < nsegs; k += BLOCK_THREADS) { const u32 v = segmin[t * nsegs + k]; m = (v < m ? v : m); }
  m = 0xFFFFFFFFu - (u32)wave_max((u64)(0xFFFFFFFFu - m));
  if(lane_id() == 0) { later[threadIdx.x >> 6] = m; }
  __syncthreads();
  u32 running = later[0];
  for(int k = 1; k < BLOCK_THREADS / WAVE; k++) { running = (later[k] < running ? later[k] : running); }
  const u64 seg_lo = sg * BOUND_SEG;
  u64 seg_hi = seg_lo + BOUND_SEG; if(seg_hi > ntiles) { seg_hi = ntiles; }
  for(u64 hi = seg_hi; hi > seg_lo; )
  {
    u64 lo = (hi - seg_lo > (u64)BLOCK_THREADS ? hi - BLOCK_THREADS : seg_lo);
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
    u32 mm = lds[threadIdx.x]; if(running < mm) { mm = running; }
    if(idx < hi) { row[idx] = mm; }
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
    // Software pipeline over the runs.  A run (~330 offsets at config 2) is ONE 16-byte load per lane: lane l takes the eight
    // offsets from the even element at or below the run's start + 8 l on (4-byte aligned; lanes past the end re-read the run's
    // last chunk, so no extra lines are fetched and no load is predicated).  The loads of run j + 1 are in flight while the
    // bits of run j are set: with unconditional loads and two alternating register sets the compiler counts them
    // (s_waitcnt vmcnt(1)) instead of draining right after the prefetch was issued, which the first version -- eight predicated
    // 2-byte loads per lane and run -- made it do.
    struct Run { u32 lo, hi; u64 base; };
    auto params = [&](u32 j) -> Run
    {
      Run r;
      r.lo = (u32)__shfl((int)my_lo, (int)j, WAVE); r.hi = (u32)__shfl((int)my_hi, (int)j, WAVE);
      r.base = shfl_u64(my_base, (int)j);
      return r;
    };
    const u32* emit32 = (const u32*)emit16;
    auto issue = [&](const Run& r, u32* v)
    {
      u64 A = (r.base + r.lo) & ~1ull;                                     // even element index: a dword boundary
      u64 last = r.base + r.hi; if(last > emit_cap) { last = emit_cap; }   // slots past the capacity took the fallback and are never read
      if(A > emit_cap) { A = emit_cap & ~1ull; }                           // (the buffer has 16 entries of padding)
      const u64 span = (r.hi > r.lo && last > A ? last - A : 1);           // elements from A to the end of the run
      const u64 cmax = (span - 1) >> 3;                                    // last 8-element chunk that holds a valid offset
      const u64 c = ((u64)lane < cmax ? (u64)lane : cmax);
      const u32* p = emit32 + (A >> 1) + 4 * c;
      v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
    };
    auto deposit = [&](const Run& r, const u32* v)
    {
      // element k of the lane's eight is valid iff first <= A + 8 lane + k < last: in 32-bit offsets from A (the indexes of a step
      // are 32-bit), ONE unsigned compare.  (Unconditional ORs -- a zero for an invalid slot -- were tried: 4.3 -> 4.5 ms with a
      // word of its own per invalid slot, 15.8 ms when the lanes past a run's end all hit the word of the run's last chunk.)
      const u64 A = (r.base + r.lo) & ~1ull;
      u64 last = r.base + r.hi; if(last > emit_cap) { last = emit_cap; }   // slots past the capacity took the fallback
      const u32 first_rel = (u32)(r.base + r.lo - A);                    // 0 or 1
      const u32 span = (last > A + first_rel ? (u32)(last - A) - first_rel : 0u);
      const u32 t = 8 * lane - first_rel;                               // wraps for lane 0 when first_rel = 1: then element 0 is invalid, as it should
#pragma unroll
      for(u32 q = 0; q < 8; q++)
      {
        const u32 o = (q & 1 ? v[q >> 1] >> 16 : v[q >> 1] & 0xFFFFu);
        if((t + q) < span) { atomicOr(&tile[o >> 5], 1u << (o & 31)); }
      }
    };
    auto rest = [&](const Run& r)                                          // a long run: offsets beyond the 512 elements from A on
    {
      const u64 A = (r.base + r.lo) & ~1ull;
      u64 last = r.base + r.hi; if(last > emit_cap) { last = emit_cap; }
      for(u64 e = A + 8 * WAVE + lane; e < last; e += WAVE) { const u32 o = emit16[e]; atomicOr(&tile[o >> 5], 1u << (o & 31)); }
    };
    if(cnt > 0)
    {
      u32 va[4], vb[4];
      Run ra = params(0);
      issue(ra, va);
      for(u32 j = 0; j < cnt; j += 2)
      {
        const Run rb = params(j + 1 < cnt ? j + 1 : cnt - 1);
        issue(rb, vb);
        seen |= (ra.hi > ra.lo);
        deposit(ra, va);
        const bool long_a = (ra.hi - ra.lo > 8 * WAVE - 2);
        const Run ra_done = ra;
        ra = params(j + 2 < cnt ? j + 2 : cnt - 1);
        issue(ra, va);
        if(j + 1 < cnt) { seen |= (rb.hi > rb.lo); deposit(rb, vb); }
        if(long_a) { rest(ra_done); }
        if(j + 1 < cnt && rb.hi - rb.lo > 8 * WAVE - 2) { rest(rb); }
      }
    }
  }
  if(seen && lane == 0) { any = 1; }
  __syncthreads();
  if(any == 0) { return; }
  // the tile is ORed into the bitvector: the four words of a thread are read together (unconditionally, from clamped indexes)
  // and then written -- as a loop of predicated read-modify-writes they were four memory round trips in a row
  const u64 w0 = T << (TILE_SHIFT - 6);
  constexpr u32 PER_THREAD = (1u << (TILE_SHIFT - 6)) / BLOCK_THREADS;
  u64 old[PER_THREAD];
#pragma unroll
  for(u32 r = 0; r < PER_THREAD; r++) { const u64 w = w0 + threadIdx.x + r * BLOCK_THREADS; old[r] = bits[w < nwords ? w : nwords - 1]; }
#pragma unroll
  for(u32 r = 0; r < PER_THREAD; r++)
  {
    const u32 k = threadIdx.x + r * BLOCK_THREADS;
    const u64 w = w0 + k;
    const u64 v = (u64)tile[2 * k] | ((u64)tile[2 * k + 1] << 32);
    if(w < nwords && v != 0) { bits[w] = old[r] | v; }
  }
}

#ifdef BWTM_EXPERIMENTAL
//------------------------------------------------------------------------------
// Sliced frontier (the dense multi-GPU form of the search).  The sorted frontier F_t is cut into G contiguous
// slices, one per GPU: a slice touches a contiguous range of B's AND of A's records (both coordinates are monotone along the
// frontier), so every GPU streams 1 / G of the records instead of a thinned 100 %.  The elements a GPU produces in a step belong
// anywhere in F_{t+1} (stable split by class over ALL GPUs): the logical order of F_{t+1} is (class, GPU, block), and a GPU's next
// slice is assembled from up to 5 G contiguous PIECES of its peers' outputs, read through their segment tables.
//
// k_frontier_gather pulls one GPU's slice into a contiguous local buffer (the layout k_frontier_init produces), after which the
// unchanged k_frontier_step runs on it.  (A production version would fold the gather into the step kernel's prologue.)
struct SlicePiece
{
  const uint2* lo; const unsigned short* hi;     // the source GPU's output coordinates (physical layout)
  const u64* prefix;                             // exclusive scan of the source's segment lengths (class-major, 5 nbl + 1 entries)
  const u64* phys;                               // physical start of every segment of the source
  u64 seg_first, seg_count;                      // the class's entries in those tables
  u64 src_first;                                 // index, inside the (class, GPU) piece, of the first element taken
  u64 count;                                     // elements taken
  u64 dst_first;                                 // where they go in this GPU's input
};

__global__ void __launch_bounds__(BLOCK_THREADS) k_frontier_gather(const SlicePiece* pieces, u32 npieces, u64 n_in, uint2* lo_in, unsigned short* hi_in)
{
  const u64 j = (u64)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if(j >= n_in) { return; }
  u32 q = 0;
  for(u32 k = 1; k < npieces; k++) { if(pieces[k].dst_first <= j) { q = k; } }      // pieces are sorted by dst_first; at most 5 G of them
  const SlicePiece pc = pieces[q];
  const u64 x = pc.prefix[pc.seg_first] + pc.src_first + (j - pc.dst_first);        // position in the source's scanned order
  u64 lo_s = pc.seg_first, hi_s = pc.seg_first + pc.seg_count;                       // prefix[lo_s] <= x < prefix[hi_s]
  while(hi_s - lo_s > 1)
  {
    const u64 mid = (lo_s + hi_s) >> 1;
    if(pc.prefix[mid] <= x) { lo_s = mid; } else { hi_s = mid; }
  }
  const u64 at = pc.phys[lo_s] + (x - pc.prefix[lo_s]);
  lo_in[j] = pc.lo[at];
  if(hi_in) { hi_in[j] = pc.hi[at]; }
}
#endif // BWTM_EXPERIMENTAL

/*
  api/search.hip.h -- the rank array: buildRA + mergeRA + RankArray (fmi.cpp:139-334, support.h:576-638)
  as the interleaving bitvector; dispatch between the level-synchronous frontier search and the per-chain
  walk; finalize; downloads.  Part of bwtm_api.hip.
*/
#pragma once

struct bwtm_ra
{
  bwtm_context* ctx = nullptr;
  bwtm_ra() : ctx(t_ctx) { if(ctx) { ctx->live_handles++; } }     // handles are created inside a Scope: t_ctx is their context
  ~bwtm_ra() { if(ctx) { ctx->live_handles--; } }
  bwtm_ra(const bwtm_ra&) = delete; bwtm_ra& operator=(const bwtm_ra&) = delete;
  u64 na = 0, nb = 0, n_out = 0;
  u64 nrecs_out = 0, nchunks = 0;
  DevBuf owned_bits;                  // nchunks * CHUNK_WORDS u64 words (unless caller-owned)
  void* bits_ptr = nullptr;
  template<class T> T* bits_as() const { return (T*)bits_ptr; }
  DevBuf chunk_base;                  // nchunks + 1 u64 (exclusive scan of chunk popcounts)
  bool finalized = false;
  u64 values = 0;
  // output-range form (bwtm_ra_finalize_range): bits and chunk_base are valid for the chunks of [range_first, range_last) and the one before
  bool ranged = false;
  u64 range_first = 0, range_last = 0;
  DevBuf range_rel;                   // chunk counts of the range, scanned (between bwtm_ra_range_counts and bwtm_ra_finalize_range)
  DevBuf super_boff;                  // set bits before every super block of the output (ranged form only)
  // A WINDOW of the bitvector (bwtm_ra_create_range; partitioned records): owned_bits holds the words
  // [win_word_first, win_word_first + win_words) only and bits_ptr is shifted back so that absolute word numbers address it unchanged.
  bool windowed = false;
  u64 win_word_first = 0, win_words = 0;
};

// Entry points that walk the whole bitvector refuse a window of one.
#define WHOLE_RA(ra, who) if((ra)->windowed) { return fail(BWTM_EINVAL, who ": a window of a rank array (bwtm_ra_create_range) only serves the merge over partitioned records and the output-range entry points"); }

namespace
{

void ra_destroy(bwtm_ra* ra)
{
  if(!ra) { return; }
  Scope scope(ra->ctx);
  // A caller-owned bitvector may be reused by the caller right away: drain the stream first.
  if(scope.rc == BWTM_OK && !ra->owned_bits.p) { (void)hipStreamSynchronize(CTX.stream); }
  delete ra;
}

// The exact fallback (also the first version of the search): one atomicOr on the bitvector per emit.  Used when the
// partition parameters do not fit (see search_partitioned) and by emit_path = 1.
int search_atomic(const bwtm_index* a, const bwtm_index* b, u64 seq_first, u64 count, bwtm_ra* ra)
{
  u64 blocks = div_up(count * 4, BLOCK_THREADS);
  u64 max_blocks = 256 * 8;
#ifdef BWTM_DIAGNOSTICS
  if(g_tune.walk_blocks > 0) { max_blocks = (u64)g_tune.walk_blocks; }
#endif
  if(blocks > max_blocks) { blocks = max_blocks; }
#ifdef BWTM_DIAGNOSTICS
  // Timing-only variants of the emit and of the loads (results are not a rank array unless walk_emit == 0).
  if(g_tune.walk_emit != 0 || g_tune.walk_kernel != 0)
  {
    const u64 lanes_per_chain = (g_tune.walk_kernel == 0 ? 4 : 1);
    blocks = div_up(count * lanes_per_chain, BLOCK_THREADS); if(blocks > max_blocks) { blocks = max_blocks; }
    DevBuf scratch;
    u32* target = ra->bits_as<u32>();
    if(g_tune.walk_emit == 2) { TRY(scratch.alloc(b->n * sizeof(u64) + 64)); target = scratch.as<u32>(); }
    if(g_tune.walk_kernel == 0)
    {
      if(g_tune.walk_emit == 1)
      {
        switch(g_tune.walk_ablate)
        {
          case 1: LAUNCH("lf_walk_noemit_nosup", (k_lf_walk_quad_diag<1, 1>), blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); break;
          case 2: LAUNCH("lf_walk_noemit_noA", (k_lf_walk_quad_diag<1, 2>), blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); break;
          case 3: LAUNCH("lf_walk_noemit_nosup_noA", (k_lf_walk_quad_diag<1, 3>), blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); break;
          case 7: LAUNCH("lf_walk_noemit_noloads", (k_lf_walk_quad_diag<1, 7>), blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); break;
          case 8: LAUNCH("lf_walk_noemit_synthetic", (k_lf_walk_quad_diag<1, 8>), blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); break;
          default: LAUNCH("lf_walk_noemit", (k_lf_walk_quad_diag<1, 0>), blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); break;
        }
      }
      else { LAUNCH("lf_walk_store", (k_lf_walk_quad_diag<2, 0>), blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); }
    }
    else
    {
      if(g_tune.walk_emit == 0)      { LAUNCH("lf_walk_lane", k_lf_walk<0>, blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); }
      else if(g_tune.walk_emit == 1) { LAUNCH("lf_walk_lane_noemit", k_lf_walk<1>, blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); }
      else                           { LAUNCH("lf_walk_lane_store", k_lf_walk<2>, blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); }
    }
    return BWTM_OK;
  }
#endif
  LAUNCH("lf_walk_atomic", k_lf_walk_quad, blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, ra->bits_as<u32>());
  return BWTM_OK;
}

// Level 2 of the emit partition + tile build: per-bin slices -> counts -> offsets -> LDS counting sort -> tiles ORed
// into the bitvector.
int partition_level2(DevBuf& l1, DevBuf& gcount, u64 cap, u64 nsub, u32 subs, bwtm_ra* ra)
{
  const u32 nregions = (u32)L1_BINS * subs;
  const u64 ntiles_pad = nsub * L1_BINS;
  const u64 nwords = ra->nchunks * CHUNK_WORDS;
  std::vector<u64> counts_host(nregions);
  HIP_TRY(hipMemcpyAsync(counts_host.data(), gcount.p, nregions * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));

  // Slices of at most PART_SLICE entries, each inside one region; the slices of a bin are consecutive.
  std::vector<u32> slice_bin, bin_slice0(L1_BINS + 1);
  std::vector<u64> slice_begin;
  u64 total_entries = 0;
  for(u32 bin = 0; bin < (u32)L1_BINS; bin++)
  {
    bin_slice0[bin] = (u32)slice_bin.size();
    for(u32 sub = 0; sub < subs; sub++)
    {
      u32 region = bin * subs + sub;
      u64 total = (counts_host[region] > cap ? cap : counts_host[region]);
      total_entries += total;
      for(u64 begin = 0; begin < total; begin += PART_SLICE) { slice_bin.push_back(region); slice_begin.push_back(begin); }
    }
  }
  bin_slice0[L1_BINS] = (u32)slice_bin.size();
  const u64 nslices = slice_bin.size();
  if(nslices == 0) { return BWTM_OK; }

  DevBuf d_slice_bin, d_slice_begin, d_bin_slice0, counts, tile_start, lists;
  TRY(d_slice_bin.alloc(nslices * sizeof(u32))); TRY(d_slice_begin.alloc(nslices * sizeof(u64))); TRY(d_bin_slice0.alloc((L1_BINS + 1) * sizeof(u32)));
  HIP_TRY(hipMemcpyAsync(d_slice_bin.p, slice_bin.data(), nslices * sizeof(u32), hipMemcpyHostToDevice, CTX.stream));
  HIP_TRY(hipMemcpyAsync(d_slice_begin.p, slice_begin.data(), nslices * sizeof(u64), hipMemcpyHostToDevice, CTX.stream));
  HIP_TRY(hipMemcpyAsync(d_bin_slice0.p, bin_slice0.data(), (L1_BINS + 1) * sizeof(u32), hipMemcpyHostToDevice, CTX.stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));         // the host vectors go out of scope at the end of the round
  TRY(counts.alloc(nslices * nsub * sizeof(u32)));
  TRY(tile_start.alloc((ntiles_pad + 1) * sizeof(u64), true));
  TRY(lists.alloc(total_entries * sizeof(unsigned short) + 64));

  LAUNCH_LDS("part_count", k_part_count, nslices, PART_THREADS, nsub * sizeof(u32), l1.as<const u32>(), cap, gcount.as<const u64>(),
    d_slice_bin.as<const u32>(), d_slice_begin.as<const u64>(), (u32)nsub, counts.as<u32>());
  LAUNCH("part_offsets", k_part_offsets, div_up(ntiles_pad, BLOCK_THREADS), BLOCK_THREADS, counts.as<u32>(), d_bin_slice0.as<const u32>(), (u32)nsub, tile_start.as<u64>());
  TRY(device_scan<0>(tile_start.as<u64>(), tile_start.as<u64>(), ntiles_pad + 1));
  const u64 sort_lds = nsub * sizeof(u64) + SORT_CHUNK * sizeof(u32) + (2 * nsub + 1) * sizeof(u32);
  bool direct = (sort_lds > 96 * 1024);               // the sorted form's tables do not fit the LDS (outputs beyond ~1.6e10 positions)
#ifdef BWTM_DIAGNOSTICS
  direct = direct || (g_tune.scatter_kernel != 0);
#endif
  if(direct)
  {
    LAUNCH_LDS("part_scatter_direct", k_part_scatter, nslices, PART_THREADS, nsub * sizeof(u64), l1.as<const u32>(), cap, gcount.as<const u64>(),
      d_slice_bin.as<const u32>(), subs, d_slice_begin.as<const u64>(), (u32)nsub, counts.as<const u32>(), tile_start.as<const u64>(), lists.as<unsigned short>());
  }
  else
  {
    LAUNCH_LDS("part_scatter", k_part_scatter_sorted, nslices, PART_THREADS, sort_lds, l1.as<const u32>(), cap, gcount.as<const u64>(),
      d_slice_bin.as<const u32>(), subs, d_slice_begin.as<const u64>(), (u32)nsub, counts.as<const u32>(), tile_start.as<const u64>(), lists.as<unsigned short>());
  }
  LAUNCH("tile_build", k_tile_build, ntiles_pad, BLOCK_THREADS, lists.as<const unsigned short>(), tile_start.as<const u64>(), ntiles_pad, ra->bits_as<u64>(), nwords);
  return BWTM_OK;
}

// Per-chain walk with partitioned emit, level-2 counting sort, tile build.
int search_partitioned(const bwtm_index* a, const bwtm_index* b, u64 seq_first, u64 count, bwtm_ra* ra)
{
  const u64 ntiles = div_up(ra->n_out + 1, 1ull << TILE_SHIFT);
  const u64 nsub = div_up(ntiles, L1_BINS);
  if(nsub > 8192) { return search_atomic(a, b, seq_first, count, ra); }       // LDS tables of level 2 would not fit

  // Rounds bound the temporary regions: emits of a round <= round_emits (estimated from the
  // average sequence length; the regions have slack and an exact fallback).
  const u64 per_seq = b->n / (b->m > 0 ? b->m : 1) + 1;
  u64 seqs_per_round = (u64)g_tune.round_emits / per_seq; if(seqs_per_round == 0) { seqs_per_round = 1; }
  const u64 nrounds = div_up(count, seqs_per_round);
  seqs_per_round = div_up(count, nrounds);

  for(u64 round = 0; round < nrounds; round++)
  {
    const u64 r_first = seq_first + round * seqs_per_round;
    u64 r_count = seqs_per_round; if(round * seqs_per_round + r_count > count) { r_count = count - round * seqs_per_round; }
    u64 blocks = div_up(r_count, (u64)(WB_THREADS / 4) * WALK_ILP);
    u64 max_blocks = 512;
#ifdef BWTM_DIAGNOSTICS
    if(g_tune.walk_blocks > 0) { max_blocks = (u64)g_tune.walk_blocks; }
#endif
    if(blocks > max_blocks) { blocks = max_blocks; }
    const u64 est = r_count * per_seq;
    u64 cap = est / L1_BINS + est / (4 * L1_BINS) + blocks * L1_CHUNK + (1ull << TILE_SHIFT);
    cap = div_up(cap, L1_CHUNK) * L1_CHUNK;
    if(g_tune.l1_cap > 0) { cap = div_up((u64)g_tune.l1_cap, L1_CHUNK) * L1_CHUNK; }      // tests: force region overflow

    DevBuf l1, gcount, overflow;
    TRY(l1.alloc((u64)L1_BINS * cap * sizeof(u32)));
    TRY(gcount.alloc(L1_BINS * sizeof(u64), true));
    TRY(overflow.alloc(64, true));
    EmitSink sink; sink.l1 = l1.as<u32>(); sink.cap = cap; sink.subs = 1; sink.gcount = gcount.as<u64>(); sink.bits = ra->bits_as<u32>(); sink.overflow = overflow.as<u32>();
    const u64 sup_bytes = 5 * (a->nsup + b->nsup) * sizeof(u64);
#ifdef BWTM_DIAGNOSTICS
    if(g_tune.walk_variant == 1 && a->nrecs < (1ull << 32) && b->nrecs < (1ull << 32))
    {
      // variant: coalesced loads + one chain per lane through an LDS transpose (measured slower, kept for A/B)
      const u64 stage_bytes = (u64)(WL_THREADS / WAVE) * 64 * WL_ROW * sizeof(u32);
      u64 wl_blocks = div_up(r_count, WL_THREADS); if(wl_blocks > 256) { wl_blocks = 256; }
      if(g_tune.walk_blocks > 0 && wl_blocks > (u64)g_tune.walk_blocks) { wl_blocks = g_tune.walk_blocks; }
      if(sup_bytes <= 40 * 1024)
      {
        LAUNCH_LDS("lf_walk_ldsT", k_lf_walk_lds<true>, wl_blocks, WL_THREADS, stage_bytes + sup_bytes, a->view(), b->view(), r_first, r_count, sink, (u32)a->nsup, (u32)b->nsup);
      }
      else
      {
        LAUNCH_LDS("lf_walk_ldsT", k_lf_walk_lds<false>, wl_blocks, WL_THREADS, stage_bytes, a->view(), b->view(), r_first, r_count, sink, (u32)a->nsup, (u32)b->nsup);
      }
    }
    else
#endif
    if(sup_bytes <= 40 * 1024)
    {
      LAUNCH_LDS("lf_walk", k_lf_walk_binned<true>, blocks, WB_THREADS, sup_bytes, a->view(), b->view(), r_first, r_count, sink, (u32)a->nsup, (u32)b->nsup);
    }
    else
    {
      LAUNCH_LDS("lf_walk", k_lf_walk_binned<false>, blocks, WB_THREADS, 0, a->view(), b->view(), r_first, r_count, sink, (u32)a->nsup, (u32)b->nsup);
    }

    TRY(partition_level2(l1, gcount, cap, nsub, 1, ra));
  }
  return BWTM_OK;
}

// Level-synchronous search (k_frontier_*): one launch per LF step over the sorted frontier; the emits of the steps of an
// EPOCH are written densely and turned into bitvector tiles when the epoch ends.
// tile_first: the rows of `bound` describe the tiles [tile_first, tile_first + ntiles) only (a part's window of the bitvector, api/pmerge.hip.h).
int frontier_flush(bwtm_ra* ra, DevBuf& emit16, u64 emit_cap, DevBuf& emit_base, DevBuf& bound, u64 ntiles, u64 nsteps, u64 tile_first = 0)
{
  if(nsteps == 0) { return BWTM_OK; }
  const u64 nsegs = div_up(ntiles, BOUND_SEG);
  DevBuf segmin; TRY(segmin.alloc(nsteps * nsegs * sizeof(u32)));
  LAUNCH2D("bound_seg_min", k_bound_seg_min, nsegs, nsteps, BLOCK_THREADS, bound.as<const u32>(), ntiles, nsegs, segmin.as<u32>());
  LAUNCH2D("bound_suffix_min", k_bound_suffix_min, nsegs, nsteps, BLOCK_THREADS, bound.as<u32>(), ntiles, nsegs, segmin.as<const u32>(), emit_base.as<const u64>());
  LAUNCH("tile_build", k_tile_build_frontier, ntiles, BLOCK_THREADS, emit16.as<const unsigned short>(), emit_base.as<const u64>(), emit_cap, bound.as<const u32>(),
    ntiles, nsteps, ra->bits_as<u64>() + (tile_first << (TILE_SHIFT - 6)), ra->nchunks * CHUNK_WORDS - (tile_first << (TILE_SHIFT - 6)));
  return BWTM_OK;
}

// The first levels of the search on trie NODES (k_range_*, fmi.cpp:286-323 as it is written): level t is the sorted list of
// (sp, count, r); a level is processed -- its runs of bits set, its children produced -- while it has at most `limit` nodes.
// Returns with N = 0 when every chain has ended (the whole search was done on nodes: collections of repeated reads), otherwise
// with the first level that was NOT processed in nodes[cur], to be expanded into frontier elements.
struct RangeLevels
{
  DevBuf sp[2], r[2], cnt[2], flags, pieces, npieces;
  u64 N = 0, alive = 0, levels = 0;
  u32 piece_cap = 0;
  int cur = 0;
  // stream ordered: the kernels queued so far still read the buffers, later allocations may take them over
  void release() { for(int k = 0; k < 2; k++) { sp[k].release(); r[k].release(); cnt[k].release(); } flags.release(); pieces.release(); npieces.release(); }
};

int range_phase(const bwtm_index* a, const bwtm_index* b, u64 seq_first, u64 count, bwtm_ra* ra, u64 limit, RangeLevels& L)
{
  const u64 cap = std::min<u64>(5 * limit, count) + 1;              // a level's children: at most five per node, at most one per sequence
  for(int k = 0; k < 2; k++) { TRY(L.sp[k].alloc(cap * sizeof(u64))); TRY(L.r[k].alloc(cap * sizeof(u64))); TRY(L.cnt[k].alloc(cap * sizeof(u64))); }
  TRY(L.flags.alloc((5 * limit + 1) * sizeof(u64)));
  L.piece_cap = (u32)std::min<u64>(count / 16 + 1024, 1ull << 24);
  TRY(L.pieces.alloc((u64)L.piece_cap * sizeof(RangePiece)));
  TRY(L.npieces.alloc(sizeof(u32), true));
  LAUNCH("range_init", k_range_init, 1, BLOCK_THREADS, L.sp[0].as<u64>(), L.r[0].as<u64>(), L.cnt[0].as<u64>(), seq_first, count, a->m);
  L.N = 1; L.alive = count; L.cur = 0; L.levels = 0;
  while(L.N > 0 && L.N <= limit)
  {
    const int c = L.cur;
    const u64 N = L.N, grid = div_up(N, BLOCK_THREADS);
    LAUNCH("range_step", k_range_step<false>, grid, BLOCK_THREADS, a->view(), b->view(), L.sp[c].as<const u64>(), L.r[c].as<const u64>(), L.cnt[c].as<const u64>(), N,
      L.flags.as<u64>(), (const u64*)nullptr, (u64*)nullptr, (u64*)nullptr, (u64*)nullptr, ra->bits_as<u32>(), L.pieces.as<RangePiece>(), L.npieces.as<u32>(), L.piece_cap);
    LAUNCH("range_emit", k_range_emit, 2048, BLOCK_THREADS, L.pieces.as<const RangePiece>(), L.npieces.as<const u32>(), L.piece_cap, ra->bits_as<u32>());
    HIP_TRY(hipMemsetAsync(L.npieces.p, 0, sizeof(u32), CTX.stream));
    TRY(device_scan<0>(L.flags.as<u64>(), L.flags.as<u64>(), 5 * N + 1));
    LAUNCH("range_children", k_range_step<true>, grid, BLOCK_THREADS, a->view(), b->view(), L.sp[c].as<const u64>(), L.r[c].as<const u64>(), L.cnt[c].as<const u64>(), N,
      (u64*)nullptr, L.flags.as<const u64>(), L.sp[1 - c].as<u64>(), L.r[1 - c].as<u64>(), L.cnt[1 - c].as<u64>(), (u32*)nullptr, (RangePiece*)nullptr, (u32*)nullptr, 0u);
    TRY(fetch_u64(L.flags.as<u64>() + 5 * N, 0));
    HIP_TRY(hipStreamSynchronize(CTX.stream));
    L.N = CTX.host_scratch[0];
    L.cur = 1 - c; L.levels++;
  }
  return BWTM_OK;
}

// Level `L.cur` of the node phase -> frontier elements in lo / hi (sized by the caller from `alive`).
int range_alive(RangeLevels& L, DevBuf& offsets)
{
  TRY(offsets.alloc((L.N + 1) * sizeof(u64)));
  HIP_TRY(hipMemcpyAsync(offsets.p, L.cnt[L.cur].p, L.N * sizeof(u64), hipMemcpyDeviceToDevice, CTX.stream));
  HIP_TRY(hipMemsetAsync(offsets.as<u64>() + L.N, 0, sizeof(u64), CTX.stream));
  TRY(device_scan<0>(offsets.as<u64>(), offsets.as<u64>(), L.N + 1));
  TRY(fetch_u64(offsets.as<u64>() + L.N, 0));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  L.alive = CTX.host_scratch[0];
  return BWTM_OK;
}

int range_expand(RangeLevels& L, const DevBuf& offsets, uint2* lo, unsigned short* hi)
{
  const int c = L.cur;
  LAUNCH("range_expand", k_range_expand, div_up(L.N, BLOCK_THREADS), BLOCK_THREADS, L.sp[c].as<const u64>(), L.r[c].as<const u64>(), L.cnt[c].as<const u64>(),
    offsets.as<const u64>(), L.N, lo, hi, L.pieces.as<RangePiece>(), L.npieces.as<u32>(), L.piece_cap);
  LAUNCH("range_expand_pieces", k_range_expand_pieces, 2048, BLOCK_THREADS, L.pieces.as<const RangePiece>(), L.npieces.as<const u32>(), L.piece_cap, lo, hi);
  return BWTM_OK;
}

constexpr u64 FRONTIER_MIN_SEQUENCES = 1ull << 21;

int search_frontier(const bwtm_index* a, const bwtm_index* b, u64 seq_first, u64 count, bwtm_ra* ra)
{
  if(a->n >= (1ull << 40) || b->n >= (1ull << 40) || count >= (1ull << 32)) { return search_partitioned(a, b, seq_first, count, ra); }
  // Node phase (range_ratio > 0): levels of at most count / range_ratio nodes are processed as trie nodes; `count` becomes the
  // number of sequences still alive at the first level that is not.
  RangeLevels levels;
  DevBuf node_offsets;
  const bool node_phase = (g_tune.range_ratio > 0);
  if(node_phase)
  {
    // The cap of 2^24 nodes bounds the level tables (7 arrays of min(5 x limit, count) + 1 words: 4.7 GB, released before the step loop
    // starts).  It only binds from ~1.3 x 10^8 sequences on, where the ratio rule would admit one more level; measured at 5 x 10^8
    // sequences with the cap lifted to 2^26 (round 4, profiles/r04_node_cap_at_target.txt): level 12 has 3.7 x 10^7 nodes there -- the
    // 4^12 strings over ACGT and about as many that hold one N -- and costs 5.7 ms of node kernels and scans for the 6.0 ms of
    // k_frontier_step it replaces, with 17.5 GB of tables instead of 4.7.  No gain: the cap stays.
    const u64 limit = std::max<u64>(1, std::min<u64>(count / (u64)g_tune.range_ratio, 1ull << 24));
    TRY(range_phase(a, b, seq_first, count, ra, limit, levels));
    if(levels.N == 0) { return BWTM_OK; }
    TRY(range_alive(levels, node_offsets));
    count = levels.alive;
    if(count == 0) { return BWTM_OK; }
  }
  const u64 ntiles = div_up(ra->n_out + 1, 1ull << TILE_SHIFT);
  const u64 nb_max = div_up(count, FR_BLOCK);
  const u64 nseg = 5 * nb_max;
  const u64 fcap = nb_max * FR_BLOCK;

  // Epochs bound the memory of the dense emits (2 bytes per emit) and of the tile boundary table (4 bytes per tile and
  // step): a step emits at most `count` values (the frontier only shrinks), so an epoch of k steps needs room for k * count
  // emits.  At config 2 (5e7 sequences of 101 symbols) one epoch holds the whole search; a 50 Gbase input takes seven.
  const u64 per_seq = b->n / (b->m > 0 ? b->m : 1) + 1;
  // Budget of the dense emits: half of the memory that is free now (pool included), between 16 and 64 GB, unless the caller fixed it.
  // Every epoch ends with a tile build that reads and rewrites the whole bitvector: at 2 x 50 Gbase a fixed 16 GB meant 7 epochs
  // (43.7 ms of tile builds per merge), while 130 GB were free during the search; the buffer is released before the result is allocated.
  u64 emit_budget = (u64)g_tune.emit_budget;
  if(emit_budget == 0)
  {
    emit_budget = 16ull << 30;
    size_t free_b = 0, total_b = 0;
    if(hipMemGetInfo(&free_b, &total_b) == hipSuccess) { emit_budget = std::min<u64>(std::max<u64>(emit_budget, ((u64)free_b + CTX.cached_bytes) / 2), 64ull << 30); }
    else { (void)hipGetLastError(); }
  }
  u64 emit_cap = emit_budget / sizeof(unsigned short);
  if(emit_cap > b->n + 64) { emit_cap = b->n + 64; }              // a search never emits more than one value per position of b
  if(emit_cap > 2 * count * per_seq + (1ull << 20)) { emit_cap = 2 * count * per_seq + (1ull << 20); }   // a shard of the sequences: far less
  if(emit_cap < count) { emit_cap = count; }                      // at least one step per epoch
  u64 EPOCH = std::max<u64>(1, (u64)g_tune.frontier_epoch);
  if(g_tune.l1_cap > 0) { emit_cap = (u64)g_tune.l1_cap; }       // tests: emits past the capacity take the exact atomicOr fallback
  else { EPOCH = std::max<u64>(1, std::min<u64>(EPOCH, emit_cap / std::max<u64>(count, 1))); }
  const u64 bound_budget = 2ull << 30;
  if(EPOCH * (ntiles + 1) * sizeof(u32) > bound_budget) { EPOCH = std::max<u64>(1, bound_budget / ((ntiles + 1) * sizeof(u32))); }

  // r may equal n_a and LF results stay below n: 32 bits hold every coordinate when both indexes are below 2^32 positions
  const bool wide = (a->n >= (1ull << 32) || b->n >= (1ull << 32));
  DevBuf lo[2], hi[2], seg_len[2], seg_phys[2], seg_prefix, emit16, emit_base, bound;
  for(int k = 0; k < 2; k++)
  {
    TRY(lo[k].alloc(fcap * 8)); if(wide) { TRY(hi[k].alloc(fcap * 2)); }
    TRY(seg_len[k].alloc((nseg + 1) * sizeof(u64), true)); TRY(seg_phys[k].alloc((nseg + 1) * sizeof(u64), true));
  }
  TRY(seg_prefix.alloc((nseg + 1) * sizeof(u64)));
  DevBuf first_seg; TRY(first_seg.alloc((nb_max + 1) * sizeof(u32)));
  // The host learns the frontier size with a delay of LOOK steps and never drains the stream for it: the size of step t travels
  // to page-locked memory behind step t's scan, and the host waits for it only before it queues step t + LOOK.  The frontier only
  // shrinks, so a stale size is a valid upper bound; the price is LOOK dead steps at the end (an empty step costs ~20 us).
  // (4 steps of lead when a step takes a millisecond: every step of lead is one dead launch pair, ~50 us, after the last chain has ended)
  const u32 LOOK = (count >= (1ull << 24) ? 4 : (count >= (1ull << 20) ? 8 : 32));
  struct Events
  {
    hipEvent_t ev[32]; u32 n = 0;
    ~Events() { for(u32 k = 0; k < n; k++) { (void)hipEventDestroy(ev[k]); } }
  } events;
  for(u32 k = 0; k < LOOK; k++) { HIP_TRY(hipEventCreateWithFlags(&events.ev[k], hipEventDisableTiming)); events.n = k + 1; }
  const u64 scan_tiles = div_up(nseg + 1, (u64)SCAN_TILE);
  DevBuf scan_partial; TRY(scan_partial.alloc(scan_tiles * sizeof(u64), true));      // cleared: k_frontier_scan1's tagged words
  TRY(emit16.alloc((emit_cap + 16) * sizeof(unsigned short)));     // k_tile_build_frontier reads 16-byte chunks
  TRY(emit_base.alloc((EPOCH + 1) * sizeof(u64), true));
  TRY(bound.alloc(EPOCH * (ntiles + 1) * sizeof(u32)));
  HIP_TRY(hipMemsetAsync(bound.p, 0xFF, EPOCH * (ntiles + 1) * sizeof(u32), CTX.stream));

  if(node_phase)
  {
    TRY(range_expand(levels, node_offsets, lo[0].as<uint2>(), hi[0].as<unsigned short>()));
    LAUNCH("frontier_init", k_frontier_init_tables, div_up(nseg + 1, BLOCK_THREADS), BLOCK_THREADS, seg_len[0].as<u64>(), seg_phys[0].as<u64>(), nb_max, count);
    levels.release(); node_offsets.release();                       // up to 10 GB at 5 x 10^8 sequences: not held through the step loop
  }
  else
  {
    u64 init_items = (fcap > nseg + 1 ? fcap : nseg + 1);
    LAUNCH("frontier_init", k_frontier_init, div_up(init_items, BLOCK_THREADS), BLOCK_THREADS, lo[0].as<uint2>(), hi[0].as<unsigned short>(),
      seg_len[0].as<u64>(), seg_phys[0].as<u64>(), nb_max, seq_first, count, a->m);
  }
  int cur = 0;
  u64 in_epoch = 0;
  u64 alive_bound = count, epoch_used = 0;                         // N_t <= alive_bound; emits reserved in this epoch so far
#ifdef BWTM_EXPERIMENTAL
  // The search view (two bit-planes + exceptions, 0.4 instead of 0.5 bytes per base): worth building when the search will stream the
  // indexes often enough -- long frontiers of many steps -- and memory allows; search_view: 0 = never, 1 = always (tests), 2 = by size.
  bool use_view = (g_tune.search_view == 1);
  if(g_tune.search_view == 2)
  {
    const u64 steps_left = (b->m > 0 ? b->n / b->m : 0);
    size_t free_b = 0, total_b = 0;
    const u64 need = (num_view_records(a->n) + num_view_records(b->n)) * 64;
    use_view = (count >= (1ull << 22) && steps_left >= 32 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && (u64)free_b + CTX.cached_bytes > need + (8ull << 30));
    (void)hipGetLastError();
  }
  if(use_view)
  {
    int rc_view = ensure_view(a);
    if(rc_view == BWTM_OK) { rc_view = ensure_view(b); }
    if(rc_view != BWTM_OK) { use_view = false; }                  // no room: the ordinary records do
  }
#endif
  // Blocks launched per step: the frontier only shrinks, and blocks past its end would only publish empty segments.  The grid
  // follows the (delayed) size when at least a quarter of it would be idle (reads of mixed lengths: after the short ones have ended);
  // the segment entries a buffer still holds from the wider grid that wrote it last are cleared then.
  u64 grid = nb_max;
  u64 written[2] = {nb_max, 0};                                    // blocks that wrote buffer k's entries when it last was the output
  u64* host_ring = nullptr;                                        // device-visible address of host_scratch[64 ..): the ring of frontier sizes
  HIP_TRY(hipHostGetDevicePointer((void**)&host_ring, CTX.host_scratch + 64, 0));
  for(u64 t = 0; t <= b->n; t++)
  {
    bool size_in_ring = false;
    if(t >= LOOK)
    {
      // before this step's scan may overwrite the slot: the size of step t - LOOK
      HIP_TRY(hipEventSynchronize(events.ev[t % LOOK]));
      alive_bound = CTX.host_scratch[64 + (u32)(t % LOOK)];
      if(alive_bound == 0) { break; }
    }
    // One launch when every tile's workgroup is resident at the same time whatever the order of dispatch (1024 workgroups of 256 threads:
    // four per CU, the kernel allows seven): a tile that waits for another can then never wait for one that has not started.  Larger tables
    // (above 5 x 10^7 sequences per call: the target size has 4769 tiles) take the two-launch form, 4 us more per step of 10 ms.
    if(scan_tiles <= CTX.scan1_tiles && g_tune.frontier_unfused == 0)
    {
      // scan of the segment lengths + per-step bookkeeping in one launch: the tiles exchange their totals through tagged words
      LAUNCH("frontier_scan", k_frontier_scan1, scan_tiles, BLOCK_THREADS, seg_len[cur].as<const u64>(), scan_partial.as<unsigned long long>(), (u32)((t & 0x7FFFFFFFull) + 1), nseg,
        seg_prefix.as<u64>(), first_seg.as<u32>(), emit_base.as<u64>(), in_epoch, host_ring + (t % LOOK));
      size_in_ring = true;
    }
    else if(scan_tiles <= FRONTIER_SCAN_TILES && g_tune.frontier_unfused != 1)
    {
      // the same in two launches (k_scan_reduce + k_frontier_scan): rounds 2 - 4, kept for comparison
      if(scan_tiles > 1)
      {
        LAUNCH("scan_reduce", k_scan_reduce<0>, scan_tiles, BLOCK_THREADS, seg_len[cur].as<const u64>(), scan_partial.as<u64>(), nseg + 1, (u64)0, scan_tiles);
      }
      LAUNCH("frontier_scan", k_frontier_scan, scan_tiles, BLOCK_THREADS, seg_len[cur].as<const u64>(), scan_partial.as<const u64>(), nseg,
        seg_prefix.as<u64>(), first_seg.as<u32>(), emit_base.as<u64>(), in_epoch, host_ring + (t % LOOK));
      size_in_ring = true;
    }
    else
    {
      TRY(device_scan<0>(seg_len[cur].as<u64>(), seg_prefix.as<u64>(), nseg + 1));
      LAUNCH("frontier_prep", k_frontier_prep, div_up(nseg, BLOCK_THREADS), BLOCK_THREADS, seg_prefix.as<const u64>(), nseg, first_seg.as<u32>(),
        emit_base.as<u64>(), in_epoch);
    }
    if(!size_in_ring) { TRY(fetch_u64(seg_prefix.as<u64>() + nseg, 64 + (u32)(t % LOOK))); HIP_TRY(hipEventRecord(events.ev[t % LOOK], CTX.stream)); }
    {
      const u64 need = std::max<u64>(1, div_up(alive_bound, FR_BLOCK));
      if(g_tune.frontier_parts <= 1 && need * 4 < grid * 3) { grid = need; }
      const int nxt = 1 - cur;
      if(written[nxt] > grid)
      {
        for(u64 c = 0; c < 5; c++) { HIP_TRY(hipMemsetAsync(seg_len[nxt].as<u64>() + c * nb_max + grid, 0, (written[nxt] - grid) * sizeof(u64), CTX.stream)); }
      }
      written[nxt] = grid;
    }
    FrontierView f;
    f.lo = lo[cur].as<const uint2>(); f.hi = hi[cur].as<const unsigned short>();
    f.lo_next = lo[1 - cur].as<uint2>(); f.hi_next = hi[1 - cur].as<unsigned short>();
    f.seg_prefix = seg_prefix.as<const u64>(); f.seg_phys = seg_phys[cur].as<const u64>(); f.first_seg = first_seg.as<const u32>();
    f.seg_len_next = seg_len[1 - cur].as<u64>(); f.seg_phys_next = seg_phys[1 - cur].as<u64>();
    f.nb_max = nb_max;
    f.emit16 = emit16.as<unsigned short>(); f.emit_base = emit_base.as<const u64>(); f.emit_cap = emit_cap; f.bits32 = ra->bits_as<u32>();
    f.bound_row = bound.as<u32>() + in_epoch * (ntiles + 1); f.step = in_epoch; f.block_base = 0; f.src_lo = nullptr; f.src_hi = nullptr; f.nseg_in = 0;
#ifdef BWTM_DIAGNOSTICS
    if(g_tune.walk_emit == 1 && wide) { LAUNCH("frontier_step_noemit", (k_frontier_step<1, true>), grid, FR_BLOCK, a->view(), b->view(), f); }
    else if(g_tune.walk_emit == 1) { LAUNCH("frontier_step_noemit", (k_frontier_step<1, false>), grid, FR_BLOCK, a->view(), b->view(), f); }
    else
#endif
    if(g_tune.frontier_parts > 1)
    {
      // Measurement of the dense multi-GPU model: the step as `frontier_parts` launches, each over a
      // contiguous slice of the sorted frontier -- what one GPU of a position-sliced search would run.  Same results.
      const u64 parts = (u64)g_tune.frontier_parts, per = div_up(nb_max, parts);
      for(u64 part = 0; part < parts && part * per < nb_max; part++)
      {
        f.block_base = (u32)(part * per);
        const u64 nb_part = std::min<u64>(per, nb_max - part * per);
        const char* label = (part == 0 ? "frontier_step_slice0" : "frontier_step_slices");
        if(wide) { LAUNCH(label, (k_frontier_step<0, true>), nb_part, FR_BLOCK, a->view(), b->view(), f); }
        else { LAUNCH(label, (k_frontier_step<0, false>), nb_part, FR_BLOCK, a->view(), b->view(), f); }
      }
    }
#ifdef BWTM_EXPERIMENTAL
    else if(use_view && wide) { LAUNCH("frontier_step", (k_frontier_step<0, true, true>), grid, FR_BLOCK, a->view(), b->view(), f); }
    else if(use_view) { LAUNCH("frontier_step", (k_frontier_step<0, false, true>), grid, FR_BLOCK, a->view(), b->view(), f); }
#endif
    else if(wide) { LAUNCH("frontier_step", (k_frontier_step<0, true>), grid, FR_BLOCK, a->view(), b->view(), f); }
    else { LAUNCH("frontier_step", (k_frontier_step<0, false>), grid, FR_BLOCK, a->view(), b->view(), f); }
    // the size of step t reaches the host behind step t's kernels: recorded AFTER the step kernel, so that reduce, scan and step
    // follow each other without another command between them
    if(size_in_ring) { HIP_TRY(hipEventRecord(events.ev[t % LOOK], CTX.stream)); }
    cur = 1 - cur;
    in_epoch++; epoch_used += alive_bound;
    // The epoch ends when the table is full or the next step might not fit (l1_cap: the tests want the overflow).
    if(in_epoch == EPOCH || (g_tune.l1_cap == 0 && epoch_used + alive_bound > emit_cap))
    {
      TRY(frontier_flush(ra, emit16, emit_cap, emit_base, bound, ntiles, in_epoch));
      HIP_TRY(hipMemsetAsync(bound.p, 0xFF, EPOCH * (ntiles + 1) * sizeof(u32), CTX.stream));
      HIP_TRY(hipMemsetAsync(emit_base.p, 0, (EPOCH + 1) * sizeof(u64), CTX.stream));
      in_epoch = 0; epoch_used = 0;
    }
  }
  TRY(frontier_flush(ra, emit16, emit_cap, emit_base, bound, ntiles, in_epoch));
  return BWTM_OK;
}

} // namespace

//------------------------------------------------------------------------------
// C ABI.

extern "C" uint64_t bwtm_ra_buffer_bytes(const bwtm_index* a, const bwtm_index* b)
{
  if(!a || !b) { return 0; }
  return div_up(num_records(a->n + b->n), 64) * CHUNK_WORDS * sizeof(u64);
}

extern "C" int bwtm_ra_create_on(const bwtm_index* a, const bwtm_index* b, void* device_buffer, uint64_t nbytes, bwtm_ra** out)
{
  if(!a || !b || !out) { return fail(BWTM_EINVAL, "bwtm_ra_create: null argument"); }
  if(a->ctx != b->ctx) { return fail(BWTM_EINVAL, "bwtm_ra_create: the two indexes live in different contexts"); }
  ENTER(a->ctx);
  bwtm_ra* ra = new bwtm_ra();
  ra->ctx = t_ctx;
  ra->na = a->n; ra->nb = b->n; ra->n_out = a->n + b->n;
  ra->nrecs_out = num_records(ra->n_out);
  ra->nchunks = div_up(ra->nrecs_out, 64);
  u64 need = ra->nchunks * CHUNK_WORDS * sizeof(u64);
  int rc = BWTM_OK;
  if(device_buffer)
  {
    if(nbytes < need) { rc = fail(BWTM_EINVAL, "bwtm_ra_create_on: buffer of %llu bytes, need %llu", (unsigned long long)nbytes, (unsigned long long)need); }
    ra->bits_ptr = device_buffer;
  }
  else
  {
    rc = ra->owned_bits.alloc(need, true);
    ra->bits_ptr = ra->owned_bits.p;
  }
  if(rc == BWTM_OK) { rc = ra->chunk_base.alloc((ra->nchunks + 1) * sizeof(u64), true); }
  if(rc != BWTM_OK) { delete ra; return rc; }
  *out = ra;
  return BWTM_OK;
}

extern "C" int bwtm_ra_create(const bwtm_index* a, const bwtm_index* b, bwtm_ra** out)
{
  return bwtm_ra_create_on(a, b, nullptr, 0, out);
}

extern "C" void bwtm_ra_free(bwtm_ra* ra) { ra_destroy(ra); }

extern "C" int bwtm_search(const bwtm_index* a, const bwtm_index* b, uint64_t seq_first, uint64_t seq_last, bwtm_ra* ra)
{
  if(!a || !b || !ra) { return fail(BWTM_EINVAL, "bwtm_search: null argument"); }
  if(a->ctx != ra->ctx || b->ctx != ra->ctx) { return fail(BWTM_EINVAL, "bwtm_search: handles of different contexts"); }
  ENTER(ra->ctx);
  WHOLE_RA(ra, "bwtm_search");
  WHOLE_INDEX(a, "bwtm_search"); WHOLE_INDEX(b, "bwtm_search");
  if(ra->na != a->n || ra->nb != b->n) { return fail(BWTM_EINVAL, "bwtm_search: rank array was created for other inputs"); }
  if(ra->finalized) { return fail(BWTM_EINVAL, "bwtm_search: rank array already finalized"); }
  if(b->m == 0 || seq_first > seq_last) { return BWTM_OK; }       // empty range (utils.h:80-83)
  if(seq_last >= b->m) { return fail(BWTM_EINVAL, "bwtm_search: sequence %llu out of range (%llu sequences)", (unsigned long long)seq_last, (unsigned long long)b->m); }
  u64 count = seq_last - seq_first + 1;
  // Two forms of the search.  The level-synchronous frontier search streams the rank structures once per LF step
  // (43 G steps/s on large read sets) but costs two to three launches per step, i.e. per symbol of the LONGEST
  // sequence; the per-chain walk does all steps in one launch at random-access speed (21-23 G steps/s).  Measured
  // crossover on MI355X: ~2-3 million sequences per call, whatever their length (both sides scale with it), so
  // small shards, small increments and collections of very long sequences take the walk.  search_algo: 0 = choose
  // by size, 1 = walk, 2 = frontier.
  const u64 avg_len = b->n / (b->m > 0 ? b->m : 1);
  const bool frontier_pays = (count >= FRONTIER_MIN_SEQUENCES && avg_len <= 4096);
  const bool want_frontier = (g_tune.search_algo == 2 || (g_tune.search_algo == 0 && frontier_pays));
#ifdef BWTM_DIAGNOSTICS
  if(g_tune.walk_kernel != 0 || (g_tune.walk_emit != 0 && !want_frontier)) { return search_atomic(a, b, seq_first, count, ra); }
#endif
  if(g_tune.emit_path != 0) { return search_atomic(a, b, seq_first, count, ra); }
  if(want_frontier) { return search_frontier(a, b, seq_first, count, ra); }
  return search_partitioned(a, b, seq_first, count, ra);
}

extern "C" int bwtm_ra_device_buffer(bwtm_ra* ra, void** device_ptr, uint64_t* nbytes)
{
  if(!ra || !device_ptr || !nbytes) { return fail(BWTM_EINVAL, "bwtm_ra_device_buffer: null argument"); }
  WHOLE_RA(ra, "bwtm_ra_device_buffer");
  ENTER(ra->ctx);
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  *device_ptr = ra->bits_ptr; *nbytes = ra->nchunks * CHUNK_WORDS * sizeof(u64);
  return BWTM_OK;
}

extern "C" int bwtm_ra_or_from(bwtm_ra* ra, const void* device_bits, uint64_t nbytes)
{
  if(!ra || !device_bits) { return fail(BWTM_EINVAL, "bwtm_ra_or_from: null argument"); }
  ENTER(ra->ctx);
  WHOLE_RA(ra, "bwtm_ra_or_from");
  if(ra->finalized) { return fail(BWTM_EINVAL, "bwtm_ra_or_from: rank array already finalized"); }
  const u64 nwords = ra->nchunks * CHUNK_WORDS;
  if(nbytes != nwords * sizeof(u64)) { return fail(BWTM_EINVAL, "bwtm_ra_or_from: buffer of %llu bytes, expected %llu", (unsigned long long)nbytes, (unsigned long long)(nwords * sizeof(u64))); }
  LAUNCH("bits_or", k_bits_or, div_up(nwords, BLOCK_THREADS), BLOCK_THREADS, ra->bits_as<u64>(), (const u64*)device_bits, nwords);
  HIP_TRY(hipStreamSynchronize(CTX.stream));          // the source buffer belongs to somebody else: done with it on return
  return BWTM_OK;
}

extern "C" int bwtm_ra_subset_check(bwtm_ra* part, bwtm_ra* whole, uint64_t* part_bits, uint64_t* words_outside)
{
  if(!part || !whole || !part_bits || !words_outside) { return fail(BWTM_EINVAL, "bwtm_ra_subset_check: null argument"); }
  if(part->ctx != whole->ctx || part->n_out != whole->n_out) { return fail(BWTM_EINVAL, "bwtm_ra_subset_check: rank arrays of different contexts or shapes"); }
  ENTER(part->ctx);
  WHOLE_RA(part, "bwtm_ra_subset_check"); WHOLE_RA(whole, "bwtm_ra_subset_check");
  const u64 nwords = part->nchunks * CHUNK_WORDS;
  DevBuf acc; TRY(acc.alloc(2 * sizeof(u64), true));
  LAUNCH("bits_subset", k_bits_subset, div_up(nwords, BLOCK_THREADS), BLOCK_THREADS, part->bits_as<const u64>(), whole->bits_as<const u64>(), nwords, acc.as<unsigned long long>());
  TRY(fetch_u64(acc.as<u64>(), 0, 2));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  *part_bits = CTX.host_scratch[0]; *words_outside = CTX.host_scratch[1];
  return BWTM_OK;
}

namespace
{
int ra_finalize(bwtm_ra* ra)
{
  LAUNCH("chunk_popc", k_chunk_popc, div_up(ra->nchunks * WAVE, BLOCK_THREADS), BLOCK_THREADS, ra->bits_as<const u64>(), ra->nchunks, ra->chunk_base.as<u64>());
  TRY(device_scan<0>(ra->chunk_base.as<u64>(), ra->chunk_base.as<u64>(), ra->nchunks + 1));
  TRY(fetch_u64(ra->chunk_base.as<u64>() + ra->nchunks, 0));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  ra->values = CTX.host_scratch[0];
  ra->finalized = true;
  return BWTM_OK;
}
} // namespace

extern "C" int bwtm_ra_finalize(bwtm_ra* ra)
{
  if(!ra) { return fail(BWTM_EINVAL, "null rank array"); }
  WHOLE_RA(ra, "bwtm_ra_finalize");
  ENTER(ra->ctx);
  return ra_finalize(ra);
}

extern "C" uint64_t bwtm_ra_values(const bwtm_ra* ra) { return ra ? ra->values : 0; }

extern "C" int bwtm_ra_download(bwtm_ra* ra, uint64_t* out, uint64_t capacity)
{
  if(!ra || !out) { return fail(BWTM_EINVAL, "bwtm_ra_download: null argument"); }
  ENTER(ra->ctx);
  WHOLE_RA(ra, "bwtm_ra_download");
  if(!ra->finalized) { return fail(BWTM_EINVAL, "bwtm_ra_download: rank array not finalized"); }
  if(capacity < ra->nb) { return fail(BWTM_EINVAL, "bwtm_ra_download: buffer too small"); }
  if(ra->nb == 0) { return BWTM_OK; }
  DevBuf d; TRY(d.alloc(ra->nb * sizeof(u64), true));
  LAUNCH("ra_extract", k_ra_extract, div_up(ra->nchunks * WAVE, BLOCK_THREADS), BLOCK_THREADS,
    ra->bits_as<const u64>(), ra->chunk_base.as<const u64>(), ra->nchunks, ra->nb, d.as<u64>());
  HIP_TRY(hipMemcpyAsync(out, d.p, ra->nb * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" int bwtm_ra_download_bits(bwtm_ra* ra, uint64_t* out_words, uint64_t capacity_words)
{
  if(!ra || !out_words) { return fail(BWTM_EINVAL, "bwtm_ra_download_bits: null argument"); }
  ENTER(ra->ctx);
  WHOLE_RA(ra, "bwtm_ra_download_bits");
  u64 words = div_up(ra->n_out, 64);
  if(capacity_words < words) { return fail(BWTM_EINVAL, "bwtm_ra_download_bits: buffer too small"); }
  if(words > 0) { HIP_TRY(hipMemcpyAsync(out_words, ra->bits_ptr, words * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream)); }
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" int bwtm_ra_download_runs(bwtm_ra* ra, uint64_t* ranks, uint64_t* counts, uint64_t capacity, uint64_t* nruns)
{
  if(!ra || !nruns) { return fail(BWTM_EINVAL, "bwtm_ra_download_runs: null argument"); }
  WHOLE_RA(ra, "bwtm_ra_download_runs");
  ENTER(ra->ctx);
  if(!ra->finalized) { return fail(BWTM_EINVAL, "bwtm_ra_download_runs: rank array not finalized"); }
  const u64 grid = div_up(ra->nchunks * WAVE, BLOCK_THREADS);
  DevBuf run_base; TRY(run_base.alloc((ra->nchunks + 1) * sizeof(u64), true));
  LAUNCH("ra_run_count", k_ra_run_count, grid, BLOCK_THREADS, ra->bits_as<const u64>(), ra->nchunks, run_base.as<u64>());
  TRY(device_scan<0>(run_base.as<u64>(), run_base.as<u64>(), ra->nchunks + 1));
  TRY(fetch_u64(run_base.as<u64>() + ra->nchunks, 0));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  const u64 total = CTX.host_scratch[0];
  *nruns = total;
  const u64 take = (total < capacity ? total : capacity);
  if(take == 0) { return BWTM_OK; }
  if(!ranks || !counts) { return fail(BWTM_EINVAL, "bwtm_ra_download_runs: null output with capacity > 0"); }
  // boff needs one entry more than is handed out when the output is truncated (the difference to the next run)
  DevBuf d_ranks, d_boff, d_counts;
  const u64 cap = std::min(total, take + 1);
  TRY(d_ranks.alloc(cap * sizeof(u64))); TRY(d_boff.alloc(cap * sizeof(u64))); TRY(d_counts.alloc(take * sizeof(u64)));
  LAUNCH("ra_run_write", k_ra_run_write, grid, BLOCK_THREADS, ra->bits_as<const u64>(), ra->chunk_base.as<const u64>(), run_base.as<const u64>(), ra->nchunks,
    d_ranks.as<u64>(), d_boff.as<u64>(), cap);
  LAUNCH("ra_run_diff", k_ra_run_diff, div_up(take, BLOCK_THREADS), BLOCK_THREADS, d_boff.as<const u64>(), cap, take, ra->nb, d_counts.as<u64>());
  HIP_TRY(hipMemcpyAsync(ranks, d_ranks.p, take * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipMemcpyAsync(counts, d_counts.p, take * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

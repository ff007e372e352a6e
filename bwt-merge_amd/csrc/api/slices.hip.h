/*
  api/slices.hip.h -- the result of a merge left SHARDED BY OUTPUT RANGE: every GPU interleaves and encodes only the
  output records [rec_first, rec_last) (mergeBWT, bwt.cpp:215-282, cut by output position), and the two things that
  cross a slice boundary travel as a few words between the GPUs:

    1. the run that is open at the boundary: (position of the last head before the slice) + 1 -- the maximal run that
       RunBuffer (utils.h:121-142) would still be extending there;
    2. Run::write's dependence on array.size() % 64 (support.h:256-282): every slice publishes the number of bytes it
       emits as a function of the byte offset it starts at (64 values, the composition of its segment tables); folding
       the tables of the slices before it gives a slice its exact byte offset.

  Needed for memory (BASELINE config 4: the result does not fit next to the replicated inputs) and for scaling (the
  interleave / encode / download tail shrinks with the number of GPUs).  Part of bwtm_api.hip.
*/
#pragma once

struct bwtm_slice
{
  bwtm_context* ctx = nullptr;
  bwtm_slice() : ctx(t_ctx) { if(ctx) { ctx->live_handles++; } }     // handles are created inside a Scope: t_ctx is their context
  ~bwtm_slice() { if(ctx) { ctx->live_handles--; } }
  bwtm_slice(const bwtm_slice&) = delete; bwtm_slice& operator=(const bwtm_slice&) = delete;
  u64 n = 0, m = 0;                    // the WHOLE merged index
  u64 C[8] = {};
  u64 nrecs_total = 0;
  u64 rec_first = 0, rec_last = 0;     // records of this slice
  u64 rec_halo = 0;                    // first record held (rec_first - 1, or 0)
  DevBuf recs;                         // records [rec_halo, rec_last)
  DevBuf sup; u64 nsup = 0;            // super table of the whole result (small)
  u64 seg_first = 0, seg_end = 0;      // encoder segments of this slice
  // encoder state
  DevBuf lasthead, table, group_table, group_base, seg_base;
  u64 ngroups = 0;
  int stage = 0;                       // 0 = interleaved, 1 = lasthead known, 2 = size table known, 3 = encoded
  u64 head_carry = 0;
  u32 halo_symbol = 0;                 // symbol at position 128 rec_first - 1
  u64 byte_first = 0, byte_end = 0;    // stream offsets of the slice's bytes
  DevBuf data;                         // bytes [byte_first & ~63, byte_end)
  u64 block_first = 0, nblocks = 0;    // blocks whose first byte lies in the slice: [block_first, block_first + nblocks)
  DevBuf block_start;                  // nblocks + 1 entries (the last one is filled in by download_samples)

  const uint4* recs_virtual() const { return recs.as<const uint4>() - 4 * rec_halo; }
  u64 pos_first() const { return rec_first << REC_SHIFT; }
  IndexView view() const
  {
    IndexView v;
    v.recs = recs_virtual(); v.sup = sup.as<const u64>();
    v.n = n; v.m = m; v.nrecs = nrecs_total;
    for(int c = 0; c < 8; c++) { v.C[c] = C[c]; }
#ifdef BWTM_EXPERIMENTAL
    v.view = nullptr; v.vsup = nullptr; v.nview = 0;
#endif
    return v;
  }
};

namespace
{

constexpr u64 SLICE_ALIGN = SEG_TILES / 2;        // records per encoder segment (512): slices are cut at segment boundaries

void slice_destroy(bwtm_slice* s)
{
  if(!s) { return; }
  Scope scope(s->ctx);
  delete s;
}

} // namespace

extern "C" uint64_t bwtm_merged_records(const bwtm_index* a, const bwtm_index* b)
{
  return (a && b ? num_records(a->n + b->n) : 0);
}

extern "C" int bwtm_slice_bounds(uint64_t nrecs, int parts, int part, uint64_t* rec_first, uint64_t* rec_last)
{
  if(parts <= 0 || part < 0 || part >= parts || !rec_first || !rec_last) { return fail(BWTM_EINVAL, "bwtm_slice_bounds: bad argument"); }
  // near-equal ranges of whole segments (getBounds over the segments, utils.cpp:169-187)
  const u64 nseg = div_up(nrecs, SLICE_ALIGN);
  const u64 lo = nseg * (u64)part / (u64)parts, hi = nseg * (u64)(part + 1) / (u64)parts;
  *rec_first = std::min(nrecs, lo * SLICE_ALIGN);
  *rec_last = (part + 1 == parts ? nrecs : std::min(nrecs, hi * SLICE_ALIGN));
  return BWTM_OK;
}

extern "C" int bwtm_slice_bounds_equal(uint64_t nrecs, int parts, int part, uint64_t* rec_first, uint64_t* rec_last, uint64_t* shard_bytes)
{
  if(parts <= 0 || part < 0 || part >= parts || !rec_first || !rec_last) { return fail(BWTM_EINVAL, "bwtm_slice_bounds_equal: bad argument"); }
  const u64 nseg = div_up(nrecs, SLICE_ALIGN);
  const u64 per = std::max<u64>(1, div_up(nseg, (u64)parts)) * SLICE_ALIGN;             // records per range
  *rec_first = std::min(nrecs, per * (u64)part);
  *rec_last = std::min(nrecs, per * (u64)(part + 1));
  if(shard_bytes) { *shard_bytes = per * (REC_POS / 8); }
  return BWTM_OK;
}

namespace
{
int check_range(const bwtm_ra* ra, u64 rec_first, u64 rec_last, const char* who)
{
  const u64 nrecs = ra->nrecs_out;
  if(rec_first > rec_last || rec_last > nrecs || (rec_first != rec_last && (rec_first % SLICE_ALIGN != 0 || (rec_last % SLICE_ALIGN != 0 && rec_last != nrecs))))
  {
    return fail(BWTM_EINVAL, "%s: [%llu, %llu) is not a range of whole %llu-record segments of %llu records", who,
      (unsigned long long)rec_first, (unsigned long long)rec_last, (unsigned long long)SLICE_ALIGN, (unsigned long long)nrecs);
  }
  return BWTM_OK;
}
} // namespace

extern "C" int bwtm_ra_range_counts(bwtm_ra* ra, uint64_t rec_first, uint64_t rec_last, uint64_t* ones, uint64_t* super_local, uint64_t* tail_words)
{
  if(!ra || !ones) { return fail(BWTM_EINVAL, "bwtm_ra_range_counts: null argument"); }
  ENTER(ra->ctx);
  TRY(check_range(ra, rec_first, rec_last, "bwtm_ra_range_counts"));
  if(ra->windowed && rec_first < rec_last)
  {
    // the chunks of the range and the one before it (the halo bwtm_ra_finalize_range installs) must lie inside the window
    const u64 w0 = (rec_first >= 64 ? (rec_first >> 6) - 1 : 0) * CHUNK_WORDS, w1 = div_up(rec_last, 64) * CHUNK_WORDS;
    if(w0 < ra->win_word_first || w1 > ra->win_word_first + ra->win_words) { return fail(BWTM_EINVAL, "bwtm_ra_range_counts: the records [%llu, %llu) reach outside the rank array's window", (unsigned long long)rec_first, (unsigned long long)rec_last); }
  }
  const u64 nsup = num_supers(ra->n_out);
  *ones = 0;
  if(super_local) { for(u64 k = 0; k < nsup; k++) { super_local[k] = 0; } }
  if(tail_words) { for(int k = 0; k < CHUNK_WORDS; k++) { tail_words[k] = 0; } }
  ra->range_first = rec_first; ra->range_last = rec_last; ra->finalized = false; ra->ranged = false;
  if(rec_first == rec_last) { ra->range_rel.release(); return BWTM_OK; }
  const u64 c0 = rec_first >> 6, c1 = div_up(rec_last, 64), nc = c1 - c0;          // rec_first is a multiple of 512: chunk aligned
  TRY(ra->range_rel.alloc((nc + 1) * sizeof(u64)));
  HIP_TRY(hipMemsetAsync(ra->range_rel.as<u64>() + nc, 0, sizeof(u64), CTX.stream));
  LAUNCH("chunk_popc", k_chunk_popc, div_up(nc * WAVE, BLOCK_THREADS), BLOCK_THREADS, ra->bits_as<const u64>() + c0 * CHUNK_WORDS, nc, ra->range_rel.as<u64>());
  TRY(device_scan<0>(ra->range_rel.as<u64>(), ra->range_rel.as<u64>(), nc + 1));
  DevBuf sl;
  if(super_local)
  {
    TRY(sl.alloc(nsup * sizeof(u64)));
    LAUNCH("super_local", k_super_local, div_up(nsup, BLOCK_THREADS), BLOCK_THREADS, ra->range_rel.as<const u64>(), c0, c1, sl.as<u64>(), nsup);
    HIP_TRY(hipMemcpyAsync(super_local, sl.p, nsup * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream));
  }
  if(tail_words) { HIP_TRY(hipMemcpyAsync(tail_words, ra->bits_as<const u64>() + (c1 - 1) * CHUNK_WORDS, CHUNK_WORDS * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream)); }
  TRY(fetch_u64(ra->range_rel.as<u64>() + nc, 0));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  *ones = CTX.host_scratch[0];
  return BWTM_OK;
}

extern "C" int bwtm_ra_finalize_range(bwtm_ra* ra, uint64_t rec_first, uint64_t rec_last, uint64_t ones_before, uint64_t ones_total,
  const uint64_t* super_boff, const uint64_t* halo_words)
{
  if(!ra || !super_boff) { return fail(BWTM_EINVAL, "bwtm_ra_finalize_range: null argument"); }
  ENTER(ra->ctx);
  TRY(check_range(ra, rec_first, rec_last, "bwtm_ra_finalize_range"));
  if(rec_first != ra->range_first || rec_last != ra->range_last || (rec_first != rec_last && !ra->range_rel.p))
  {
    return fail(BWTM_EINVAL, "bwtm_ra_finalize_range: call bwtm_ra_range_counts for the same range first");
  }
  const u64 nsup = num_supers(ra->n_out);
  TRY(ra->super_boff.alloc(nsup * sizeof(u64)));
  // pageable host arrays: staged synchronously by the runtime; a few KB
  HIP_TRY(hipMemcpyAsync(ra->super_boff.p, super_boff, nsup * sizeof(u64), hipMemcpyHostToDevice, CTX.stream));
  if(rec_first < rec_last)
  {
    const u64 c0 = rec_first >> 6, c1 = div_up(rec_last, 64), nc = c1 - c0;
    LAUNCH("add_offset", k_add_offset, div_up(nc + 1, BLOCK_THREADS), BLOCK_THREADS, ra->range_rel.as<u64>(), nc + 1, ones_before);
    HIP_TRY(hipMemcpyAsync(ra->chunk_base.as<u64>() + c0, ra->range_rel.p, (nc + 1) * sizeof(u64), hipMemcpyDeviceToDevice, CTX.stream));
    if(c0 > 0)
    {
      // the chunk before the range (it holds the one record the encoder looks back at): its bits come from the range before
      u64 halo_ones = 0;
      if(halo_words) { for(int k = 0; k < CHUNK_WORDS; k++) { halo_ones += (u64)__builtin_popcountll(halo_words[k]); } }
      if(halo_ones > ones_before) { return fail(BWTM_EINVAL, "bwtm_ra_finalize_range: the halo chunk holds more set bits than all earlier ranges"); }
      if(halo_words) { HIP_TRY(hipMemcpyAsync(ra->bits_as<u64>() + (c0 - 1) * CHUNK_WORDS, halo_words, CHUNK_WORDS * sizeof(u64), hipMemcpyHostToDevice, CTX.stream)); }
      else { HIP_TRY(hipMemsetAsync(ra->bits_as<u64>() + (c0 - 1) * CHUNK_WORDS, 0, CHUNK_WORDS * sizeof(u64), CTX.stream)); }
      CTX.host_scratch[61] = ones_before - halo_ones;
      HIP_TRY(hipMemcpyAsync(ra->chunk_base.as<u64>() + (c0 - 1), CTX.host_scratch + 61, sizeof(u64), hipMemcpyHostToDevice, CTX.stream));
    }
  }
  HIP_TRY(hipStreamSynchronize(CTX.stream));                       // the caller's arrays have been read
  ra->range_rel.release();
  ra->values = ones_total; ra->finalized = true; ra->ranged = true;
  return BWTM_OK;
}

extern "C" int bwtm_fold_offsets(const uint64_t* tables, int parts, uint64_t* offsets)
{
  if(!tables || !offsets || parts < 0) { return fail(BWTM_EINVAL, "bwtm_fold_offsets: bad argument"); }
  u64 off = 0;
  for(int g = 0; g < parts; g++) { offsets[g] = off; off += tables[(u64)g * 64 + (off & 63)]; }
  offsets[parts] = off;
  return BWTM_OK;
}

extern "C" int bwtm_interleave_range(const bwtm_index* a, const bwtm_index* b, bwtm_ra* ra, uint64_t rec_first, uint64_t rec_last, bwtm_slice** out)
{
  if(!a || !b || !ra || !out) { return fail(BWTM_EINVAL, "bwtm_interleave_range: null argument"); }
  if(a->ctx != ra->ctx || b->ctx != ra->ctx) { return fail(BWTM_EINVAL, "bwtm_interleave_range: handles of different contexts"); }
  ENTER(ra->ctx);
  TRY(check_interleave_args(a, b, ra, true));
  const u64 nrecs = ra->nrecs_out;
  TRY(check_range(ra, rec_first, rec_last, "bwtm_interleave_range"));
  if(ra->ranged && (rec_first != ra->range_first || rec_last != ra->range_last))
  {
    return fail(BWTM_EINVAL, "bwtm_interleave_range: the rank array was finalized for the records [%llu, %llu)", (unsigned long long)ra->range_first, (unsigned long long)ra->range_last);
  }
  bwtm_slice* s = new bwtm_slice();
  s->ctx = t_ctx;
  auto body = [&]() -> int
  {
    s->n = ra->n_out; s->m = a->m + b->m;
    for(int c = 0; c < 8; c++) { s->C[c] = a->C[c] + b->C[c]; }
    s->nrecs_total = nrecs; s->rec_first = rec_first; s->rec_last = rec_last;
    s->rec_halo = (rec_first > 0 ? rec_first - 1 : 0);
    if(rec_first < rec_last) { s->seg_first = rec_first / SLICE_ALIGN; s->seg_end = div_up(rec_last, SLICE_ALIGN); }   // an empty slice has no segments
    s->nsup = num_supers(s->n);
    TRY(s->recs.alloc((rec_last - s->rec_halo + 1) * 64));
    TRY(s->sup.alloc(s->nsup * SUP_STRIDE * sizeof(u64)));
    if(a->windowed || b->windowed)
    {
      // windows hold the records of this range only: the super rows the slice refers to, from inside the range (kernels/partition.hip.h)
      LAUNCH("interleave_sup", k_interleave_sup_window, div_up(s->nsup, BLOCK_THREADS), BLOCK_THREADS, a->view(), b->view(), ra->chunk_base.as<const u64>(),
        s->sup.as<u64>(), s->nsup, ra->super_boff.as<const u64>(), (s->rec_halo >> 6) << 6, rec_last);
    }
    else
    LAUNCH("interleave_sup", k_interleave_sup, div_up(s->nsup, BLOCK_THREADS), BLOCK_THREADS, a->view(), b->view(),
      ra->bits_as<const u64>(), ra->chunk_base.as<const u64>(), s->n, s->sup.as<u64>(), s->nsup, (ra->ranged ? ra->super_boff.as<const u64>() : (const u64*)nullptr));
    // an empty range (more GPUs than output chunks) interleaves nothing: its halo chunk was not installed by bwtm_ra_finalize_range either
    if(rec_first < rec_last)
    {
      const u64 c0 = s->rec_halo >> 6, c1 = div_up(rec_last, 64);
      TRY(interleave_chunks(a, b, ra, c0, c1, s->rec_halo, rec_last, s->sup.as<const u64>(), s->recs.as<uint4>() - 4 * s->rec_halo));
    }
    return BWTM_OK;
  };
  int rc = body();
  if(rc != BWTM_OK) { delete s; return rc; }
  *out = s;
  return BWTM_OK;
}

extern "C" void bwtm_slice_free(bwtm_slice* slice) { slice_destroy(slice); }

extern "C" int bwtm_slice_lasthead(bwtm_slice* s, uint64_t* lasthead)
{
  if(!s || !lasthead) { return fail(BWTM_EINVAL, "bwtm_slice_lasthead: null argument"); }
  ENTER(s->ctx);
  const u64 nseg = s->seg_end - s->seg_first;
  *lasthead = 0;
  s->stage = 1;
  if(nseg == 0) { return BWTM_OK; }
  const u64 ntiles = (s->n >> 6) + 1;
  TRY(s->lasthead.alloc((nseg + 1) * sizeof(u64)));
  HIP_TRY(hipMemsetAsync(s->lasthead.as<u64>() + nseg, 0, sizeof(u64), CTX.stream));
  LAUNCH("enc_lasthead", k_enc_lasthead, div_up(nseg * WAVE, BLOCK_THREADS), BLOCK_THREADS, s->recs_virtual(), s->nrecs_total, s->n, ntiles,
    s->seg_first, s->seg_end, s->lasthead.as<u64>() - s->seg_first);
  // exclusive max-scan over nseg + 1 entries: entry k = (last head of the slice before its segment k) + 1, the extra entry = the slice's own
  TRY(device_scan<1>(s->lasthead.as<u64>(), s->lasthead.as<u64>(), nseg + 1));
  TRY(fetch_u64(s->lasthead.as<u64>() + nseg, 0));
  // the symbol before the slice (for the samples of blocks opened by a run that began in an earlier slice)
  DevBuf sym; TRY(sym.alloc(8));
  if(s->rec_first > 0)
  {
    LAUNCH("extract", k_extract, 1, BLOCK_THREADS, s->view(), s->pos_first() - 1, (u64)1, sym.as<u8>());
    CTX.host_scratch[1] = 0;
    HIP_TRY(hipMemcpyAsync(CTX.host_scratch + 1, sym.p, 1, hipMemcpyDeviceToHost, CTX.stream));
  }
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  *lasthead = CTX.host_scratch[0];
  s->halo_symbol = (s->rec_first > 0 ? (u32)(CTX.host_scratch[1] & 0xFF) : 0u);
  return BWTM_OK;
}

extern "C" int bwtm_slice_size_table(bwtm_slice* s, uint64_t heads_before, uint64_t* table)
{
  if(!s || !table) { return fail(BWTM_EINVAL, "bwtm_slice_size_table: null argument"); }
  ENTER(s->ctx);
  if(s->stage < 1) { return fail(BWTM_EINVAL, "bwtm_slice_size_table: call bwtm_slice_lasthead first"); }
  const u64 nseg = s->seg_end - s->seg_first;
  s->head_carry = heads_before;
  s->stage = 2;
  for(int o = 0; o < 64; o++) { table[o] = 0; }
  if(nseg == 0) { return BWTM_OK; }
  const u64 ntiles = (s->n >> 6) + 1;
  s->ngroups = div_up(nseg, FOLD_GROUP);
  TRY(s->table.alloc(nseg * 64 * sizeof(u32)));
  TRY(s->group_table.alloc(s->ngroups * 64 * sizeof(u64)));
  DevBuf slice_table; TRY(slice_table.alloc(64 * sizeof(u64)));
  LAUNCH("enc_size", k_enc_size, div_up(nseg * WAVE, BLOCK_THREADS), BLOCK_THREADS, s->recs_virtual(), s->nrecs_total, s->n, ntiles, s->seg_first, s->seg_end,
    s->lasthead.as<const u64>() - s->seg_first, heads_before, s->table.as<u32>() - s->seg_first * 64);
  LAUNCH("fold_group", k_fold_group, s->ngroups, WAVE, s->table.as<const u32>(), nseg, s->group_table.as<u64>());
  LAUNCH("fold_slice", k_fold_slice, 1, WAVE, s->group_table.as<const u64>(), s->ngroups, slice_table.as<u64>());
  HIP_TRY(hipMemcpyAsync(table, slice_table.p, 64 * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" int bwtm_slice_encode(bwtm_slice* s, uint64_t byte_offset)
{
  if(!s) { return fail(BWTM_EINVAL, "bwtm_slice_encode: null argument"); }
  ENTER(s->ctx);
  if(s->stage < 2) { return fail(BWTM_EINVAL, "bwtm_slice_encode: call bwtm_slice_size_table first"); }
  const u64 nseg = s->seg_end - s->seg_first;
  s->byte_first = byte_offset; s->byte_end = byte_offset;
  s->stage = 3;
  if(nseg > 0)
  {
    const u64 ntiles = (s->n >> 6) + 1;
    TRY(s->group_base.alloc((s->ngroups + 1) * sizeof(u64)));
    TRY(s->seg_base.alloc(nseg * sizeof(u64)));
    LAUNCH("fold_top", k_fold_top, 1, WAVE, s->group_table.as<const u64>(), s->ngroups, byte_offset, s->group_base.as<u64>());
    LAUNCH("fold_seg", k_fold_seg, s->ngroups, WAVE, s->table.as<const u32>(), nseg, s->group_base.as<const u64>(), s->seg_base.as<u64>());
    TRY(fetch_u64(s->group_base.as<u64>() + s->ngroups, 0));
    HIP_TRY(hipStreamSynchronize(CTX.stream));
    s->byte_end = CTX.host_scratch[0];
    const u64 base = byte_offset & ~(u64)(RLE_BLOCK - 1);
    s->block_first = div_up(byte_offset, RLE_BLOCK);
    s->nblocks = div_up(s->byte_end, RLE_BLOCK) - s->block_first;
    TRY(alloc_native(s->data, s->byte_end - base));
    TRY(s->block_start.alloc((s->nblocks + 1) * sizeof(u64), true));
    LAUNCH("enc_emit", k_enc_emit<false>, div_up(nseg * WAVE, BLOCK_THREADS), BLOCK_THREADS, s->recs_virtual(), s->nrecs_total, s->n, ntiles, s->seg_first, s->seg_end,
      s->lasthead.as<const u64>() - s->seg_first, s->head_carry, s->seg_base.as<const u64>() - s->seg_first, s->data.as<u8>() - base,
      s->block_start.as<u64>() - s->block_first, (u32*)nullptr, (u64)0);
  }
  else
  {
    s->block_first = div_up(byte_offset, RLE_BLOCK); s->nblocks = 0;
    TRY(s->block_start.alloc(sizeof(u64), true));
  }
  s->table.release(); s->group_table.release(); s->group_base.release(); s->seg_base.release(); s->lasthead.release();
  return BWTM_OK;
}

extern "C" uint64_t bwtm_slice_byte_first(const bwtm_slice* s)  { return s ? s->byte_first : 0; }
extern "C" uint64_t bwtm_slice_bytes(const bwtm_slice* s)       { return s ? s->byte_end - s->byte_first : 0; }
extern "C" uint64_t bwtm_slice_block_first(const bwtm_slice* s) { return s ? s->block_first : 0; }
extern "C" uint64_t bwtm_slice_blocks(const bwtm_slice* s)      { return s ? s->nblocks : 0; }

extern "C" int bwtm_slice_first_block_start(bwtm_slice* s, uint64_t* position)
{
  if(!s || !position) { return fail(BWTM_EINVAL, "bwtm_slice_first_block_start: null argument"); }
  ENTER(s->ctx);
  if(s->stage < 3) { return fail(BWTM_EINVAL, "bwtm_slice_first_block_start: slice not encoded"); }
  *position = ~0ull;                                 // no block starts in this slice
  if(s->nblocks == 0) { return BWTM_OK; }
  TRY(fetch_u64(s->block_start.as<u64>(), 0));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  *position = CTX.host_scratch[0];
  return BWTM_OK;
}

extern "C" int bwtm_slice_download_data(bwtm_slice* s, uint8_t* out, uint64_t capacity)
{
  if(!s || (!out && capacity > 0)) { return fail(BWTM_EINVAL, "bwtm_slice_download_data: null argument"); }
  ENTER(s->ctx);
  if(s->stage < 3) { return fail(BWTM_EINVAL, "bwtm_slice_download_data: slice not encoded"); }
  const u64 bytes = s->byte_end - s->byte_first;
  if(capacity < bytes) { return fail(BWTM_EINVAL, "bwtm_slice_download_data: buffer too small"); }
  if(bytes > 0)
  {
    const u64 base = s->byte_first & ~(u64)(RLE_BLOCK - 1);
    HIP_TRY(hipMemcpyAsync(out, s->data.as<u8>() + (s->byte_first - base), bytes, hipMemcpyDeviceToHost, CTX.stream));
  }
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" int bwtm_slice_download_samples(bwtm_slice* s, uint64_t next_block_start, uint64_t* block_end, uint64_t* cum)
{
  if(!s) { return fail(BWTM_EINVAL, "bwtm_slice_download_samples: null argument"); }
  ENTER(s->ctx);
  if(s->stage < 3) { return fail(BWTM_EINVAL, "bwtm_slice_download_samples: slice not encoded"); }
  if(s->nblocks == 0) { return BWTM_OK; }
  if(!block_end || !cum) { return fail(BWTM_EINVAL, "bwtm_slice_download_samples: null argument"); }
  const u64 nb = s->nblocks;
  CTX.host_scratch[62] = next_block_start;
  HIP_TRY(hipMemcpyAsync(s->block_start.as<u64>() + nb, CTX.host_scratch + 62, sizeof(u64), hipMemcpyHostToDevice, CTX.stream));
  DevBuf be, dc;
  TRY(be.alloc(nb * sizeof(u64))); TRY(dc.alloc(6 * nb * sizeof(u64)));
  LAUNCH("block_end", k_block_end, div_up(nb, BLOCK_THREADS), BLOCK_THREADS, s->block_start.as<const u64>(), (u64)0, nb, be.as<u64>());
  LAUNCH("block_cum", k_block_cum_slice, div_up(nb, BLOCK_THREADS), BLOCK_THREADS, s->view(), s->block_start.as<const u64>(), (u64)0, nb, dc.as<u64>(), nb,
    s->pos_first(), s->halo_symbol);
  HIP_TRY(hipMemcpyAsync(block_end, be.p, nb * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipMemcpyAsync(cum, dc.p, 6 * nb * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" int bwtm_slice_extract(bwtm_slice* s, uint64_t first, uint64_t count, uint8_t* out)
{
  if(!s || !out) { return fail(BWTM_EINVAL, "bwtm_slice_extract: null argument"); }
  ENTER(s->ctx);
  const u64 lo = s->pos_first(), hi = std::min(s->n, s->rec_last << REC_SHIFT);
  if(first < lo || first + count > hi) { return fail(BWTM_EINVAL, "bwtm_slice_extract: range outside the slice"); }
  if(count == 0) { return BWTM_OK; }
  const u64 piece = 1ull << 30;                                    // fewer than 2^32 threads per launch (bwtm_extract)
  DevBuf d; TRY(d.alloc(std::min(count, piece)));
  for(u64 done = 0; done < count; done += piece)
  {
    const u64 n = std::min(piece, count - done);
    LAUNCH("extract", k_extract, div_up(n, BLOCK_THREADS), BLOCK_THREADS, s->view(), first + done, n, d.as<u8>());
    HIP_TRY(hipMemcpyAsync(out + done, d.p, n, hipMemcpyDeviceToHost, CTX.stream));
    HIP_TRY(hipStreamSynchronize(CTX.stream));
  }
  return BWTM_OK;
}

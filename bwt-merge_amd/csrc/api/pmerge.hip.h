/*
  api/pmerge.hip.h -- FMI::FMI(a, b, parameters) (fmi.cpp:336-369) over PARTITIONED records: one PART per GPU, nothing replicated
  (DESIGN.md section 6.3).  Part of bwtm_api.hip, after api/group.hip.h (the parts' shared control block and exported arenas).

  The reference hands blocks of b's sequences to threads that share both indexes (fmi.cpp:351-358).  Sequence blocks on GPUs mean
  replicated records and a thinned frontier on every GPU; here the merged ORDER is cut instead.  A cut (I_g, R_g) = (suffixes of a below
  w_g, suffixes of b below w_g) for a k-mer w_g; part g owns
      b's records of [R_g, R_g+1) and a's of [I_g, I_g+1] (+ a margin of two encoder segments), transcoded from ITS share of the native bytes,
      the frontier elements and trie nodes whose b coordinate lies in [R_g, R_g+1),
      the bitvector words and the output records of [I_g + R_g, I_g+1 + R_g+1) (rounded to encoder segments).
  Every rank query of a part is local; what crosses between parts is the frontier's elements (10 bytes each per LF step, read by the
  consumer's step kernel straight out of the producer's output buffer), 8 KiB of boundary bits per pair, and a few hundred bytes per step.

    bwtm_index_upload_window   a window of an index from the 64-byte blocks that cover it (BWT::load + BWT::build of a share, bwt.cpp:132-148, 476-512)
    bwtm_ra_create_range       a part's range of the interleaving bitvector
    bwtm_partition_cuts_host   the cuts: sp(c w) = C[c] + rank_c(sp(w)) on host-resident inputs (BWT::rank, bwt.cpp:318-341, over their samples)
    bwtm_part_create / _upload / _search / _finish
                               the merge as one part sees it; _search and _finish are collective over the group
*/
#pragma once

//------------------------------------------------------------------------------
// Windows.

namespace
{

// A window of an index from its own share of the native bytes: `data` = the whole 64-byte blocks [b0, b1) of the stream (host memory, or
// device memory read in place when on_device), first_position = the position block b0 begins at, counts_before[c] = occurrences of c before it.
int index_upload_window(const u8* data, u64 nbytes, u64 first_position, const u64 counts_before[6], u64 bases, u64 sequences, const u64 C[7], bool on_device, bwtm_index** out)
{
  if(!out || !data || nbytes == 0 || !counts_before || !C) { return fail(BWTM_EINVAL, "bwtm_index_upload_window: null argument"); }
  u64 before = 0; for(int c = 0; c < 6; c++) { before += counts_before[c]; }
  if(before != first_position || first_position > bases) { return fail(BWTM_EINVAL, "bwtm_index_upload_window: the counts before the bytes add up to %llu, their first position is %llu", (unsigned long long)before, (unsigned long long)first_position); }
  if(on_device && ((uintptr_t)data & 15) != 0) { return fail(BWTM_EINVAL, "bwtm_index_upload_window: a device share must be 16-byte aligned (whole 64-byte blocks of a 16-byte aligned stream)"); }
  bwtm_index* x = new bwtm_index();
  x->ctx = t_ctx; x->nbytes = nbytes;
  auto body = [&]() -> int
  {
    if(on_device) { x->borrowed = data; }
    else { TRY(alloc_native(x->data, nbytes)); }
    // the product's upload pipeline on the share: block lengths and group counts, their scans; the stream's verdict and totals come back
    int rc = upload_queue(x, on_device ? nullptr : data);
    if(rc == BWTM_OK) { rc = upload_scan(x, 0); }
    hipError_t e1 = hipStreamSynchronize(CTX.copy_stream), e2 = hipStreamSynchronize(CTX.stream);
    if(rc != BWTM_OK) { return rc; }
    if(e1 != hipSuccess || e2 != hipSuccess) { return fail(BWTM_ENODEV, "upload failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); }
    const u32 flags = (u32)CTX.host_scratch[6];
    x->flags.release();
    if(flags & 1u) { return fail(BWTM_EINVAL, "not a canonical run-length stream: a full 64-byte block encodes fewer than 64 positions"); }
    u64 held = 0; for(int c = 0; c < 6; c++) { held += CTX.host_scratch[c]; }
    const u64 end_position = first_position + held;
    if(end_position > bases) { return fail(BWTM_EINVAL, "bwtm_index_upload_window: the bytes decode to positions [%llu, %llu) of an index of %llu", (unsigned long long)first_position, (unsigned long long)end_position, (unsigned long long)bases); }
    // absolute positions and counts: the scanned group tables start at the share's first block
    const u64 gstride = x->ngroups + 1;
    for(u32 c = 0; c < 7; c++)
    {
      const u64 v = (c < 6 ? counts_before[c] : first_position);
      if(v != 0) { LAUNCH("add_offset", k_add_offset, div_up(gstride, BLOCK_THREADS), BLOCK_THREADS, x->gcum.as<u64>() + (u64)c * gstride, gstride, v); }
    }
    x->n = bases; x->m = sequences;
    for(int c = 0; c < 7; c++) { x->C[c] = C[c]; }
    x->C[7] = x->C[6];
    x->nrecs = num_records(bases); x->nsup = num_supers(bases);
    // records that begin inside the share (k_build_recs: a group owns the records that START in it; the last group owns the rest, here up to
    // the record the share ends in, whose tail is only valid when the share holds the end of the index)
    const u64 q0 = (first_position + REC_POS - 1) >> REC_SHIFT, q_end = (end_position >> REC_SHIFT) + 1;
    if(q0 >= q_end) { return fail(BWTM_EINVAL, "bwtm_index_upload_window: the bytes hold no whole record"); }
    x->windowed = true; x->win_first = q0; x->win_count = q_end - q0;
    TRY(x->recs.alloc(x->win_count * 64));
    TRY(x->sup.alloc(x->nsup * SUP_STRIDE * sizeof(u64)));
    // super rows: k_build_sup gives a super that begins before the share the counts at the share's first position -- at or below the counts
    // of every record of the window, which is all a row has to be (header fields are offsets from it)
    LAUNCH("build_sup", k_build_sup, div_up(x->nsup * WAVE, BLOCK_THREADS), BLOCK_THREADS,
      x->native_bytes(), x->nbytes, x->blen.as<const u64>(), x->gcum.as<const u64>(), gstride, x->nblocks, x->ngroups, end_position, x->sup.as<u64>(), x->nsup);
    uint4* shifted = (uint4*)((char*)x->recs.p - (q0 << 6));
    TRY(build_records(x, held, end_position, shifted, q_end));
    HIP_TRY(hipStreamSynchronize(CTX.stream));                       // a borrowed share may go back to its owner
    x->blen.release(); x->block_start.release(); x->gcum.release(); x->data.release();          // a window keeps its records and super rows only
    x->borrowed = nullptr;
    x->has_native = false; x->nbytes = 0; x->nblocks = 0; x->ngroups = 0;
    // the record the share begins in and the one it ends in are incomplete unless they are the index's own first / last
    if(end_position < bases) { x->win_count -= 1; }
    if(x->win_count == 0) { return fail(BWTM_EINVAL, "bwtm_index_upload_window: the bytes hold no whole record"); }
    return BWTM_OK;
  };
  int rc = body();
  if(rc != BWTM_OK) { (void)hipStreamSynchronize(CTX.stream); x->borrowed = nullptr; delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

int ra_create_range(const bwtm_index* a, const bwtm_index* b, u64 pos_first, u64 pos_last, bwtm_ra** out)
{
  bwtm_ra* ra = new bwtm_ra();
  ra->ctx = t_ctx;
  ra->na = a->n; ra->nb = b->n; ra->n_out = a->n + b->n;
  ra->nrecs_out = num_records(ra->n_out);
  ra->nchunks = div_up(ra->nrecs_out, 64);
  const u64 tile_words = 1ull << (TILE_SHIFT - 6), nwords = ra->nchunks * CHUNK_WORDS;
  const u64 t0 = (pos_first >> TILE_SHIFT), t1 = div_up(std::min<u64>(pos_last, ra->n_out) + 1, 1ull << TILE_SHIFT);
  const u64 w0 = (t0 > 0 ? t0 - 1 : 0) * tile_words, w1 = std::min<u64>(nwords, (t1 + 1) * tile_words);
  ra->windowed = true; ra->win_word_first = w0; ra->win_words = (w1 > w0 ? w1 - w0 : tile_words);
  int rc = ra->owned_bits.alloc(ra->win_words * sizeof(u64), true);
  if(rc == BWTM_OK) { ra->bits_ptr = (char*)ra->owned_bits.p - w0 * sizeof(u64); }
  if(rc == BWTM_OK) { rc = ra->chunk_base.alloc((ra->nchunks + 1) * sizeof(u64), true); }
  if(rc != BWTM_OK) { delete ra; return rc; }
  *out = ra;
  return BWTM_OK;
}

// Scan of a segment table (seg_len[0 .. nseg], the last entry 0): seg_prefix, first_seg (the entry that holds every 256th element),
// emit_base[step + 1] = emit_base[step] + elements.  `tiles` = a cleared buffer of at least ceil((nseg + 1) / 2048) words that only this
// table's scans use; tag = a number that changes with every call on it (never 0).
int frontier_table_scan(const u64* seg_len, u64 nseg, u64* seg_prefix, u32* first_seg, u64* emit_base, u64 step, DevBuf& tiles, u32 tag)
{
  const u64 scan_tiles = div_up(nseg + 1, (u64)SCAN_TILE);
  if(scan_tiles <= CTX.scan1_tiles && g_tune.frontier_unfused == 0)
  {
    LAUNCH("frontier_scan", k_frontier_scan1, scan_tiles, BLOCK_THREADS, seg_len, tiles.as<unsigned long long>(), tag, nseg, seg_prefix, first_seg, emit_base, step, (u64*)nullptr);
  }
  else if(scan_tiles <= FRONTIER_SCAN_TILES && g_tune.frontier_unfused != 1)
  {
    LAUNCH("scan_reduce", k_scan_reduce<0>, scan_tiles, BLOCK_THREADS, seg_len, tiles.as<u64>(), nseg + 1, (u64)0, scan_tiles);
    LAUNCH("frontier_scan", k_frontier_scan, scan_tiles, BLOCK_THREADS, seg_len, tiles.as<const u64>(), nseg, seg_prefix, first_seg, emit_base, step, (u64*)nullptr);
    HIP_TRY(hipMemsetAsync(tiles.p, 0, scan_tiles * sizeof(u64), CTX.stream));      // the one-launch form may use the buffer next (tagged words)
  }
  else
  {
    TRY(device_scan<0>(seg_len, seg_prefix, nseg + 1));
    LAUNCH("frontier_prep", k_frontier_prep, std::max<u64>(1, div_up(nseg, BLOCK_THREADS)), BLOCK_THREADS, (const u64*)seg_prefix, nseg, first_seg, emit_base, step);
  }
  return BWTM_OK;
}

} // namespace

extern "C" int bwtm_index_upload_window(const uint8_t* data, uint64_t nbytes, uint64_t first_position, const uint64_t counts_before[6],
  uint64_t bases, uint64_t sequences, const uint64_t C[7], int on_device, bwtm_index** out)
{
  ENTER(nullptr);
  return index_upload_window(data, nbytes, first_position, counts_before, bases, sequences, C, on_device != 0, out);
}

extern "C" uint64_t bwtm_index_record_bytes(const bwtm_index* x)
{
  if(!x) { return 0; }
  return (x->windowed ? x->win_count : x->nrecs) * 64;
}

extern "C" int bwtm_ra_create_range(const bwtm_index* a, const bwtm_index* b, uint64_t pos_first, uint64_t pos_last, bwtm_ra** out)
{
  if(!a || !b || !out || pos_first > pos_last) { return fail(BWTM_EINVAL, "bwtm_ra_create_range: bad argument"); }
  if(a->ctx != b->ctx) { return fail(BWTM_EINVAL, "bwtm_ra_create_range: the two indexes live in different contexts"); }
  ENTER(a->ctx);
  return ra_create_range(a, b, pos_first, pos_last, out);
}

extern "C" uint64_t bwtm_ra_bytes(const bwtm_ra* ra)
{
  if(!ra) { return 0; }
  return (ra->windowed ? ra->win_words : ra->nchunks * CHUNK_WORDS) * sizeof(u64);
}

//------------------------------------------------------------------------------
// Cuts from host-resident inputs: the reference's loaded FMI answers rank queries on the host (BWT::rank, bwt.cpp:318-341: block by
// block_rank, counts from the samples, a scan of the block's runs); so does this, over the arrays BWT::build leaves (bwt.cpp:476-512).

namespace
{

struct HostRank
{
  const bwtm_host_index* x;
  // block_start[k]: the position block k begins at = sum over c of cum[c][k]
  u64 start(u64 k) const { u64 p = 0; for(int c = 0; c < 6; c++) { p += x->cum[(u64)c * (x->blocks + 1) + k]; } return p; }
  // occurrences of c in [0, i)
  u64 rank(u64 i, u32 c) const
  {
    if(i >= x->bases) { return x->cum[(u64)c * (x->blocks + 1) + x->blocks]; }
    u64 lo = 0, hi = x->blocks;                                     // the last block that begins at or before i
    while(hi - lo > 1) { const u64 mid = (lo + hi) >> 1; if(start(mid) <= i) { lo = mid; } else { hi = mid; } }
    u64 pos = start(lo), r = x->cum[(u64)c * (x->blocks + 1) + lo];
    const u8* p = x->data + lo * RLE_BLOCK; const u8* end = x->data + std::min<u64>(x->nbytes, (lo + 1) * RLE_BLOCK);
    while(p < end && pos < i)
    {
      // Run::read, support.h:244-250
      const u32 byte = *p++;
      const u32 sym = byte % SIGMA; u64 len = byte / SIGMA + 1;
      if(len >= MAX_RUN) { u64 v = 0; u32 sh = 0; while(p < end) { const u32 b2 = *p++; v |= (u64)(b2 & 0x7F) << sh; sh += 7; if(!(b2 & 0x80)) { break; } } len += v; }
      const u64 take = std::min<u64>(len, i - pos);
      if(sym == c) { r += take; }
      pos += len;
    }
    return r;
  }
};

} // namespace

extern "C" int bwtm_partition_cuts_host(const bwtm_host_index* a, const bwtm_host_index* b, int parts, int kmer, uint64_t* cut_a, uint64_t* cut_b)
{
  if(!a || !b || !cut_a || !cut_b || parts < 1 || parts > (int)PART_MAX) { return fail(BWTM_EINVAL, "bwtm_partition_cuts_host: bad argument"); }
  if(!a->data || !a->cum || !b->data || !b->cum) { return fail(BWTM_EINVAL, "bwtm_partition_cuts_host: the inputs' bytes and cumulative sample arrays are needed"); }
  if(kmer <= 0) { kmer = (parts <= 8 ? 4 : 5); }
  if(kmer > 8) { return fail(BWTM_EINVAL, "bwtm_partition_cuts_host: at most 8-mers"); }
  cut_a[0] = 0; cut_b[0] = 0; cut_a[parts] = a->bases; cut_b[parts] = b->bases;
  if(parts == 1) { return BWTM_OK; }
  // insertion points of all 5^k k-mers in lexicographic order: sp(c w) = C[c] + rank_c(sp(w)) (utils.h:335-355), c-major keeps the order
  auto points = [&](const bwtm_host_index* x) -> std::vector<u64>
  {
    HostRank R{x};
    std::vector<u64> sp(1, 0);
    for(int round = 0; round < kmer; round++)
    {
      std::vector<u64> next(5 * sp.size());
      for(u32 c = 1; c <= 5; c++) { for(size_t j = 0; j < sp.size(); j++) { next[(c - 1) * sp.size() + j] = x->C[c] + R.rank(sp[j], c); } }
      sp.swap(next);
    }
    return sp;
  };
  const std::vector<u64> pa = points(a), pb = points(b);
  const double total = (double)a->bases + (double)b->bases;
  for(int g = 1; g < parts; g++)
  {
    size_t best = 0; double dist = -1;
    for(size_t j = 0; j < pa.size(); j++)
    {
      const double d = std::abs((double)pa[j] + (double)pb[j] - total * g / parts);
      if(dist < 0 || d < dist) { dist = d; best = j; }
    }
    cut_a[g] = std::max(cut_a[g - 1], pa[best]); cut_b[g] = std::max(cut_b[g - 1], pb[best]);
  }
  return BWTM_OK;
}

/* The 64-byte blocks of a native stream that cover the records of the positions [pos_first, pos_last]: pure host arithmetic on the samples. */
extern "C" int bwtm_window_blocks(const bwtm_host_index* x, uint64_t pos_first, uint64_t pos_last, uint64_t* block_first, uint64_t* block_end,
  uint64_t* first_position, uint64_t counts_before[6])
{
  if(!x || !x->cum || !block_first || !block_end || !first_position || !counts_before || pos_first > pos_last) { return fail(BWTM_EINVAL, "bwtm_window_blocks: bad argument"); }
  HostRank R{x};
  const u64 nb = x->blocks;
  if(nb == 0) { return fail(BWTM_EINVAL, "bwtm_window_blocks: the index has no blocks"); }
  const u64 first = pos_first & ~(u64)127, end = std::min<u64>(x->bases, (pos_last | 127) + 1);
  u64 l = 0, r = nb;                                                // last block that begins at or before `first`
  while(r - l > 1) { const u64 mid = (l + r) / 2; if(R.start(mid) <= first) { l = mid; } else { r = mid; } }
  const u64 b0 = l;
  l = b0; r = nb;                                                   // first block that begins at or after `end` (nb: none)
  while(l < r) { const u64 mid = (l + r) / 2; if(R.start(mid) >= end) { r = mid; } else { l = mid + 1; } }
  *block_first = b0; *block_end = std::max<u64>(l, b0 + 1);
  *first_position = R.start(b0);
  for(int c = 0; c < 6; c++) { counts_before[c] = x->cum[(u64)c * (nb + 1) + b0]; }
  return BWTM_OK;
}

//------------------------------------------------------------------------------
// One part of a merge.

namespace
{

constexpr u64 PART_MARGIN = 2 * 65536;             // positions a part reads beyond its cuts in the second half: one encoder segment + the halo chunk
constexpr u64 BOUNDARY_BYTES = 65536 / 8;          // one encoder segment of bitvector

struct StepInfo                                    // what a part tells the others about its outputs of a step
{
  u64 nb;                                          // blocks (= stride of its tables' classes)
  u64 error;                                       // a part that cannot go on says so here: everybody stops at the same step
  CutEntry cut[5][PART_MAX + 1];
};

struct NodeInfo { u64 class_first[6]; u64 below[5][PART_MAX + 1]; u64 error; };

struct PartLayout                                  // offsets of a part's exported arrays inside its arena
{
  u64 lo[2], hi[2], seg_len[2], seg_phys[2], node_sp[2], node_r[2], node_cnt[2], boundary, bytes;
  u64 cap, node_cap, wide;
};

} // namespace

struct bwtm_part
{
  bwtm_context* ctx = nullptr;
  bwtm_group* grp = nullptr;
  int part = 0, parts = 1;
  u64 na = 0, nb = 0, ma = 0, mb = 0;
  u64 Ca[8] = {}, Cb[8] = {};
  u64 cut_a[PART_MAX + 1] = {}, cut_b[PART_MAX + 1] = {};
  bwtm_index* win[2] = {nullptr, nullptr};
  bwtm_ra* ra = nullptr;
  bool searched = false;
  PartLayout lay = {};                             // of this merge's exported buffers (bytes == 0: the part has not searched)
  bwtm_part_info info = {};
  bwtm_part() : ctx(t_ctx) { if(ctx) { ctx->live_handles++; } }
  ~bwtm_part() { if(ctx) { ctx->live_handles--; } }
  bwtm_part(const bwtm_part&) = delete; bwtm_part& operator=(const bwtm_part&) = delete;
  u64 out_pos(int g) const { return cut_a[g] + cut_b[g]; }
  u64 out_seg(int g) const { return (g == 0 ? 0 : out_pos(g) >> 16); }
};

extern "C" int bwtm_part_create(bwtm_group* group, const bwtm_index_header* a, const bwtm_index_header* b, const uint64_t* cut_a, const uint64_t* cut_b, bwtm_part** out)
{
  if(!group || !a || !b || !cut_a || !cut_b || !out) { return fail(BWTM_EINVAL, "bwtm_part_create: null argument"); }
  ENTER(nullptr);
  const int parts = group->parts;
  if(cut_a[0] != 0 || cut_b[0] != 0 || cut_a[parts] != a->bases || cut_b[parts] != b->bases) { return fail(BWTM_EINVAL, "bwtm_part_create: the cuts begin at 0 and end at the inputs' sizes"); }
  for(int g = 0; g < parts; g++) { if(cut_a[g] > cut_a[g + 1] || cut_b[g] > cut_b[g + 1]) { return fail(BWTM_EINVAL, "bwtm_part_create: cuts must not decrease"); } }
  if(a->bases >= (1ull << 40) || b->bases >= (1ull << 40)) { return fail(BWTM_EINVAL, "bwtm_part_create: coordinates do not fit 40 bits"); }
  bwtm_part* P = new bwtm_part();
  P->grp = group; P->part = group->part; P->parts = parts;
  P->na = a->bases; P->nb = b->bases; P->ma = a->sequences; P->mb = b->sequences;
  for(int c = 0; c < 7; c++) { P->Ca[c] = a->C[c]; P->Cb[c] = b->C[c]; }
  P->Ca[7] = P->Ca[6]; P->Cb[7] = P->Cb[6];
  for(int g = 0; g <= parts; g++) { P->cut_a[g] = cut_a[g]; P->cut_b[g] = cut_b[g]; }
  *out = P;
  return BWTM_OK;
}

extern "C" void bwtm_part_free(bwtm_part* P)
{
  if(!P) { return; }
  if(P->ra) { bwtm_ra_free(P->ra); }
  for(int k = 0; k < 2; k++) { if(P->win[k]) { bwtm_index_free(P->win[k]); } }
  Scope scope(P->ctx);
  delete P;
}

extern "C" int bwtm_part_window(const bwtm_part* P, int which, uint64_t* pos_first, uint64_t* pos_last)
{
  if(!P || !pos_first || !pos_last || which < 0 || which > 1) { return fail(BWTM_EINVAL, "bwtm_part_window: bad argument"); }
  const u64* cut = (which == 0 ? P->cut_a : P->cut_b);
  const u64 n = (which == 0 ? P->na : P->nb);
  *pos_first = (cut[P->part] > PART_MARGIN ? cut[P->part] - PART_MARGIN : 0);
  *pos_last = std::min<u64>(n, cut[P->part + 1] + PART_MARGIN);
  return BWTM_OK;
}

extern "C" int bwtm_part_upload(bwtm_part* P, int which, const uint8_t* data, uint64_t nbytes, uint64_t first_position, const uint64_t counts_before[6], int on_device)
{
  if(!P || which < 0 || which > 1) { return fail(BWTM_EINVAL, "bwtm_part_upload: bad argument"); }
  ENTER(P->ctx);
  if(P->win[which]) { bwtm_index_free(P->win[which]); P->win[which] = nullptr; }
  const bool is_a = (which == 0);
  TURN(P->grp);
  TRY(index_upload_window(data, nbytes, first_position, counts_before, is_a ? P->na : P->nb, is_a ? P->ma : P->mb, is_a ? P->Ca : P->Cb, on_device != 0, &P->win[which]));
  // the window must hold the records of the part's range and its margins
  u64 lo, hi; (void)bwtm_part_window(P, which, &lo, &hi);
  const bwtm_index* w = P->win[which];
  const u64 q0 = lo >> REC_SHIFT, q1 = std::min<u64>(hi >> REC_SHIFT, w->nrecs - 1);
  if(w->win_first > q0 || w->win_first + w->win_count <= q1)
  {
    return fail(BWTM_EINVAL, "bwtm_part_upload: the bytes hold the records [%llu, %llu), the part needs [%llu, %llu]", (unsigned long long)w->win_first,
      (unsigned long long)(w->win_first + w->win_count), (unsigned long long)q0, (unsigned long long)q1);
  }
  P->info.record_bytes += w->win_count * 64;
  return BWTM_OK;
}

namespace
{

// What the search of one part holds besides its windows.
struct PartSearch
{
  bwtm_part* P; bwtm_group* G; int g, parts;
  const bwtm_index* A; const bwtm_index* B; bwtm_ra* ra;
  bool wide = false;
  PartLayout lay, peer_lay[PART_MAX];
  char* arena = nullptr; char* peer[PART_MAX] = {};
  u64 cap = 0, nbl_cap = 0, fcap = 0, node_cap = 0;
  // local
  DevBuf seg_len_in, seg_phys_in, seg_prefix_in, first_seg_in, tiles_in; u64 seg_in_cap = 0;
  DevBuf out_prefix, out_first_seg, tiles_out, dummy_emit;
  DevBuf cuts_dev, srcs;
  DevBuf emit16, emit_base, bound; u64 emit_cap = 0, EPOCH = 1, in_epoch = 0, epoch_used = 0, ntiles = 0, tile_first = 0;      // (the tiles of the part's window of the bitvector)
  // page-locked staging
  PullPlan* plan_host[2] = {nullptr, nullptr}; CutEntry* cut_host = nullptr; u64* small_host = nullptr;
  CutEntry* cut_mapped = nullptr;                  // cut_host as the device addresses it
  NodePiece* node_pieces_host = nullptr;           // (all five: pieces of the group's page-locked block)
  u64 step_limit = 0;                              // elements a step may hold (= cap unless a test lowers it)
  std::string own_error;                           // what this part said when it ran out of room (it learns of its own stop from the exchange like everybody)
  u32 tag_in = 1, tag_out = 1;
  template<class T> T* mine(u64 off) const { return (T*)(arena + off); }
  template<class T> T* theirs(int h, u64 off) const { return (T*)(peer[h] + off); }
};

u64 align256(u64 x) { return (x + 255) / 256 * 256; }

PartLayout part_layout(u64 cap, u64 node_cap, bool wide, int parts)
{
  PartLayout L = {};
  const u64 nbl = div_up(cap, (u64)FR_BLOCK), fcap = nbl * FR_BLOCK, nseg = 5 * nbl;
  u64 off = 0;
  auto take = [&](u64 bytes) { const u64 at = off; off += align256(bytes); return at; };
  for(int k = 0; k < 2; k++) { L.lo[k] = take(fcap * 8); L.hi[k] = (wide ? take(fcap * 2) : 0); L.seg_len[k] = take((nseg + 1) * 8); L.seg_phys[k] = take((nseg + 1) * 8); }
  for(int k = 0; k < 2; k++) { L.node_sp[k] = take(node_cap * 8); L.node_r[k] = take(node_cap * 8); L.node_cnt[k] = take(node_cap * 8); }
  L.boundary = take((u64)parts * BOUNDARY_BYTES);
  L.bytes = off; L.cap = cap; L.node_cap = node_cap; L.wide = wide ? 1 : 0;
  return L;
}

int search_setup(PartSearch& S, bool node_phase)
{
  bwtm_part* P = S.P;
  const u64 m = P->mb;
  S.wide = (P->na >= (1ull << 32) || P->nb >= (1ull << 32));
  // A part holds ~m / parts elements (the cuts balance positions): twice that and some slack, at most everything.  Without the node phase the
  // search begins with ALL roots on the part that owns the "$" suffixes (k-mer cuts: the first).
  S.cap = (S.parts == 1 || !node_phase ? m + 1 : std::min<u64>(m + 1, 2 * (m / S.parts) + 65536));
  if(g_tune.part_capacity > 0 && (u64)g_tune.part_capacity < S.cap) { S.cap = (u64)g_tune.part_capacity; }      // tests: a part that runs out of room
  S.step_limit = (g_tune.part_capacity < 0 ? std::min<u64>(S.cap, (u64)(-g_tune.part_capacity)) : S.cap);
  if(S.cap >= (1ull << 32)) { return fail(BWTM_EINVAL, "bwtm_part_search: %llu elements per part do not fit the 32-bit indexes of a step", (unsigned long long)S.cap); }
  S.nbl_cap = div_up(S.cap, (u64)FR_BLOCK); S.fcap = S.nbl_cap * FR_BLOCK;
  const u64 limit = (g_tune.range_ratio > 0 ? std::max<u64>(1, std::min<u64>(m / (u64)g_tune.range_ratio, 1ull << 24)) : 0);
  const u64 level_max = std::min<u64>(5 * std::max<u64>(limit, 1), m) + 1;           // children of the last level processed as nodes, over all parts
  S.node_cap = (S.parts == 1 ? level_max : std::min<u64>(level_max, 2 * (level_max / S.parts) + 65536));
  S.lay = part_layout(S.cap, S.node_cap, S.wide, S.parts);
  P->lay = S.lay;
  void* arena = nullptr;
  TRY(group_arena(S.G, S.lay.bytes, &arena));
  S.arena = (char*)arena;
  // tables and counters of the exported buffers start empty
  for(int k = 0; k < 2; k++)
  {
    HIP_TRY(hipMemsetAsync(S.arena + S.lay.seg_len[k], 0, (5 * S.nbl_cap + 1) * 8, CTX.stream));
    HIP_TRY(hipMemsetAsync(S.arena + S.lay.seg_phys[k], 0, (5 * S.nbl_cap + 1) * 8, CTX.stream));
  }
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  TRY(group_allgather(S.G, &S.lay, sizeof(PartLayout), S.peer_lay));
  for(int h = 0; h < S.parts; h++) { void* p = nullptr; TRY(group_peer_arena(S.G, h, &p)); S.peer[h] = (char*)p; }
  // local tables
  S.seg_in_cap = 2 * 5 * S.nbl_cap + 1024;
  TRY(S.seg_len_in.alloc((S.seg_in_cap + 1) * 8)); TRY(S.seg_phys_in.alloc((S.seg_in_cap + 1) * 8)); TRY(S.seg_prefix_in.alloc((S.seg_in_cap + 1) * 8));
  TRY(S.first_seg_in.alloc((S.nbl_cap + 2) * sizeof(u32)));
  TRY(S.tiles_in.alloc(std::max<u64>(div_up(S.seg_in_cap + 1, (u64)SCAN_TILE), 1) * 8, true));
  TRY(S.out_prefix.alloc((5 * S.nbl_cap + 1) * 8)); TRY(S.out_first_seg.alloc((S.nbl_cap + 2) * sizeof(u32)));
  TRY(S.tiles_out.alloc(std::max<u64>(div_up(5 * S.nbl_cap + 1, (u64)SCAN_TILE), 1) * 8, true));
  TRY(S.dummy_emit.alloc(2 * 8, true));
  TRY(S.cuts_dev.alloc((S.parts + 1) * 8));
  TRY(S.srcs.alloc(4 * PART_MAX * sizeof(void*)));
  static_assert(sizeof(PullPlan) <= 4096 && 5 * (PART_MAX + 1) * sizeof(CutEntry) <= 4096 && 5 * PART_MAX * sizeof(NodePiece) <= 4096 && GROUP_PINNED_BYTES >= 5 * 4096 + 2048, "the group's page-locked block");
  char* pinned = nullptr;
  TRY(group_pinned(S.G, &pinned));
  for(int k = 0; k < 2; k++) { S.plan_host[k] = (PullPlan*)(pinned + 4096 * k); }
  S.cut_host = (CutEntry*)(pinned + 8192); S.small_host = (u64*)(pinned + 12288); S.node_pieces_host = (NodePiece*)(pinned + 16384);
  // The cut search stores its triples straight into page-locked host memory (2 KB of posted writes): a copy command behind the kernel costs more
  // idle device than the transfer (the search's frontier size travels the same way, api/search.hip.h).  The plan goes the other way as a copy:
  // every workgroup of the pull kernels reads it, and 500 workgroups fetching it over PCIe take longer than one command.
  std::memset(S.cut_host, 0, 5 * (PART_MAX + 1) * sizeof(CutEntry));
  HIP_TRY(hipHostGetDevicePointer((void**)&S.cut_mapped, S.cut_host, 0));
  // the cuts and the parts' buffers, as this GPU addresses them: [parity][lo | hi][part]
  for(int k = 0; k <= S.parts; k++) { S.small_host[k] = (k == S.parts ? ~0ull : P->cut_b[k]); }
  HIP_TRY(hipMemcpyAsync(S.cuts_dev.p, S.small_host, (S.parts + 1) * 8, hipMemcpyHostToDevice, CTX.stream));
  void** sp = (void**)(S.small_host + 32);
  for(int par = 0; par < 2; par++)
  {
    for(u32 h = 0; h < PART_MAX; h++)
    {
      const int src = ((int)h < S.parts ? (int)h : S.g);
      sp[(2 * par + 0) * PART_MAX + h] = S.peer[src] + S.peer_lay[src].lo[par];
      sp[(2 * par + 1) * PART_MAX + h] = (S.wide ? S.peer[src] + S.peer_lay[src].hi[par] : nullptr);
    }
  }
  HIP_TRY(hipMemcpyAsync(S.srcs.p, sp, 4 * PART_MAX * sizeof(void*), hipMemcpyHostToDevice, CTX.stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  // dense emits of an epoch of steps, as in search_frontier()
  // the tile bookkeeping of the emits covers the part's own window of the bitvector, not the whole output
  S.tile_first = S.ra->win_word_first >> (TILE_SHIFT - 6);
  S.ntiles = div_up(S.ra->win_words, 1ull << (TILE_SHIFT - 6));
  const u64 per_seq = P->nb / (m > 0 ? m : 1) + 1;
  S.emit_cap = std::min<u64>((u64)(g_tune.emit_budget > 0 ? g_tune.emit_budget : (16ll << 30)) / sizeof(unsigned short), 2 * S.cap * per_seq + (1ull << 20));
  if(S.emit_cap < S.cap) { S.emit_cap = S.cap; }
  S.EPOCH = std::max<u64>(1, std::min<u64>((u64)std::max<long long>(1, g_tune.frontier_epoch), S.emit_cap / S.cap));
  const u64 bound_budget = 2ull << 30;
  if(S.EPOCH * (S.ntiles + 1) * sizeof(u32) > bound_budget) { S.EPOCH = std::max<u64>(1, bound_budget / ((S.ntiles + 1) * sizeof(u32))); }
  TRY(S.emit16.alloc((S.emit_cap + 16) * sizeof(unsigned short)));
  TRY(S.emit_base.alloc((S.EPOCH + 1) * sizeof(u64), true));
  TRY(S.bound.alloc(S.EPOCH * (S.ntiles + 1) * sizeof(u32)));
  HIP_TRY(hipMemsetAsync(S.bound.p, 0xFF, S.EPOCH * (S.ntiles + 1) * sizeof(u32), CTX.stream));
  return BWTM_OK;
}

// The part's outputs of a step (in its exported buffers of parity `par`, `nb` blocks): their scan and where the cuts fall; StepInfo of all parts.
int publish_outputs(PartSearch& S, int par, u64 nb, u64 error, StepInfo* all)
{
  const u64 nseg = 5 * nb;
  {
  TURN(S.G);
  TRY(frontier_table_scan(S.mine<const u64>(S.lay.seg_len[par]), nseg, S.out_prefix.as<u64>(), S.out_first_seg.as<u32>(), S.dummy_emit.as<u64>(), 0, S.tiles_out, S.tag_out));
  S.tag_out = (S.tag_out == 0x7FFFFFFFu ? 1 : S.tag_out + 1);
  LAUNCH("cut_search", k_cut_search_seg, 5 * (u64)(S.parts - 1) + 1, BLOCK_THREADS, S.mine<const uint2>(S.lay.lo[par]), (S.wide ? S.mine<const unsigned short>(S.lay.hi[par]) : nullptr),
    S.out_prefix.as<const u64>(), S.mine<const u64>(S.lay.seg_phys[par]), S.out_first_seg.as<const u32>(), nb, S.cuts_dev.as<const u64>(), (u32)S.parts, (u32)(PART_MAX + 1),
    S.cut_mapped);
  HIP_TRY(hipStreamSynchronize(CTX.stream));                         // this part's step is complete: its outputs may be read, its inputs overwritten
  }
  StepInfo mine;
  mine.nb = nb; mine.error = error;
  std::memcpy(mine.cut, S.cut_host, sizeof(mine.cut));
  TRY(group_allgather(S.G, &mine, sizeof(StepInfo), all));
  if(all[S.g].error != 0) { return fail(BWTM_ENOMEM, "%s", S.own_error.c_str()); }
  for(int h = 0; h < S.parts; h++) { if(all[h].error != 0) { return fail(BWTM_EPEER, "part %d ran out of room and stopped the search", h); } }
  return BWTM_OK;
}

// This part's next input from all parts' StepInfo: the runs of their tables between its two cuts, class after class, source after source.
void plan_input(const StepInfo* all, int g, int parts, PullPlan* plan, u64& n_in, u64& nseg_in)
{
  u32 np = 0; n_in = 0; nseg_in = 0;
  for(u32 c = 0; c < 5; c++)
  {
    for(int h = 0; h < parts; h++)
    {
      const CutEntry& lo = all[h].cut[c][g]; const CutEntry& hi = all[h].cut[c][g + 1];
      if(hi.below <= lo.below) { continue; }
      const u64 last = (hi.off > 0 ? hi.seg : hi.seg - 1);           // hi.below > lo.below: segment lo.seg holds an element of the run, so last >= lo.seg
      PullPiece pc;
      pc.src_first = (u64)c * all[h].nb + lo.seg; pc.dst_first = (u32)nseg_in; pc.count = (u32)(last - lo.seg + 1); pc.src = (u32)h;
      pc.clip_first = (u32)lo.off; pc.last_len = (hi.off > 0 ? (u32)hi.off : PULL_ALL); pc.pad = 0;
      plan->piece[np++] = pc;
      n_in += hi.below - lo.below; nseg_in += pc.count;
    }
  }
  plan->npieces = np; plan->nseg = (u32)nseg_in;
}

int part_flush(PartSearch& S)
{
  TRY(frontier_flush(S.ra, S.emit16, S.emit_cap, S.emit_base, S.bound, S.ntiles, S.in_epoch, S.tile_first));
  HIP_TRY(hipMemsetAsync(S.bound.p, 0xFF, S.EPOCH * (S.ntiles + 1) * sizeof(u32), CTX.stream));
  HIP_TRY(hipMemsetAsync(S.emit_base.p, 0, (S.EPOCH + 1) * sizeof(u64), CTX.stream));
  S.in_epoch = 0; S.epoch_used = 0;
  return BWTM_OK;
}

// The first levels on trie NODES (fmi.cpp:286-323), every node on the part that owns its range, the children routed by position.
// Returns with the part's share of the first level that is NOT processed as nodes expanded into its exported buffers of parity 0
// (nb_out blocks); done = every chain has ended.
int node_phase(PartSearch& S, u64 root_first, u64 root_count, u64 limit, u64& nb_out, bool& done)
{
  bwtm_part* P = S.P;
  const u64 ncap = S.node_cap;
  DevBuf sp, r, cnt, flags, pieces, npieces, class_first, below, err, gather_pieces, offsets;
  TRY(sp.alloc(ncap * 8)); TRY(r.alloc(ncap * 8)); TRY(cnt.alloc(ncap * 8));
  TRY(flags.alloc((5 * ncap + 1) * 8));
  const u32 piece_cap = (u32)std::min<u64>(S.cap / 16 + 1024, 1ull << 24);
  TRY(pieces.alloc((u64)piece_cap * sizeof(RangePiece))); TRY(npieces.alloc(sizeof(u32), true));
  TRY(class_first.alloc(6 * 8)); TRY(below.alloc(5ull * (S.parts + 1) * 8, true)); TRY(err.alloc(sizeof(u32), true));
  TRY(gather_pieces.alloc(5ull * PART_MAX * sizeof(NodePiece)));
  NodePiece* host_pieces = S.node_pieces_host;
  const u32 ncuts = (u32)S.parts + 1;
  u64 N = 0;
  if(root_count > 0) { LAUNCH("range_init", k_range_init, 1, BLOCK_THREADS, sp.as<u64>(), r.as<u64>(), cnt.as<u64>(), root_first, root_count, P->ma); N = 1; }
  u64 level_nodes = 1, lvl = 0;
  std::vector<NodeInfo> infos(S.parts);
  while(level_nodes > 0 && level_nodes <= limit)
  {
    const int par = (int)(lvl & 1);
    u64* csp = S.mine<u64>(S.lay.node_sp[par]); u64* cr = S.mine<u64>(S.lay.node_r[par]); u64* ccnt = S.mine<u64>(S.lay.node_cnt[par]);
    NodeInfo mine = {};
    {
    TURN(S.G);
    if(N > 0)
    {
      const u64 grid = div_up(N, BLOCK_THREADS);
      LAUNCH("range_step", k_range_step<false>, grid, BLOCK_THREADS, S.A->view(), S.B->view(), sp.as<const u64>(), r.as<const u64>(), cnt.as<const u64>(), N,
        flags.as<u64>(), (const u64*)nullptr, (u64*)nullptr, (u64*)nullptr, (u64*)nullptr, S.ra->bits_as<u32>(), pieces.as<RangePiece>(), npieces.as<u32>(), piece_cap);
      LAUNCH("range_emit", k_range_emit, 2048, BLOCK_THREADS, pieces.as<const RangePiece>(), npieces.as<const u32>(), piece_cap, S.ra->bits_as<u32>());
      HIP_TRY(hipMemsetAsync(npieces.p, 0, sizeof(u32), CTX.stream));
      TRY(device_scan<0>(flags.as<u64>(), flags.as<u64>(), 5 * N + 1));
      TRY(fetch_u64(flags.as<u64>() + 5 * N, 0));
      HIP_TRY(hipStreamSynchronize(CTX.stream));
      const u64 children = CTX.host_scratch[0];
      if(children > ncap) { mine.error = 1; (void)fail(BWTM_ENOMEM, "bwtm_part_search: %llu nodes of part %d have %llu children, capacity %llu", (unsigned long long)N, S.g, (unsigned long long)children, (unsigned long long)ncap); }
      else
      {
        LAUNCH("range_children", k_range_step<true>, grid, BLOCK_THREADS, S.A->view(), S.B->view(), sp.as<const u64>(), r.as<const u64>(), cnt.as<const u64>(), N,
          (u64*)nullptr, flags.as<const u64>(), csp, cr, ccnt, (u32*)nullptr, (RangePiece*)nullptr, (u32*)nullptr, 0u);
        // class c's children are [flags[(c - 1) N], flags[c N]): the six boundaries, then the cut points inside every class
        for(u32 c = 0; c <= 5; c++) { HIP_TRY(hipMemcpyAsync(class_first.as<u64>() + c, flags.as<u64>() + (u64)c * N, 8, hipMemcpyDeviceToDevice, CTX.stream)); }
        HIP_TRY(hipMemsetAsync(err.p, 0, sizeof(u32), CTX.stream));
        LAUNCH("cut_counts", k_node_cut_search, 1, BLOCK_THREADS, (const u64*)csp, (const u64*)ccnt, class_first.as<const u64>(), S.cuts_dev.as<const u64>(), ncuts, below.as<u64>(), err.as<u32>());
        HIP_TRY(hipMemcpyAsync(S.small_host + 128, below.p, 5ull * ncuts * 8, hipMemcpyDeviceToHost, CTX.stream));
        TRY(fetch_u64(class_first.as<u64>(), 96, 6));
        S.small_host[127] = 0;
        HIP_TRY(hipMemcpyAsync(S.small_host + 127, err.p, sizeof(u32), hipMemcpyDeviceToHost, CTX.stream));
        HIP_TRY(hipStreamSynchronize(CTX.stream));
        if((u32)S.small_host[127] != 0) { mine.error = 2; (void)fail(BWTM_EINVAL, "bwtm_part_search: a trie node crosses a cut (cuts must be k-mer boundaries of the merged order)"); }
        for(u32 c = 0; c <= 5; c++) { mine.class_first[c] = CTX.host_scratch[96 + c]; }
        for(u32 c = 0; c < 5; c++)
        {
          const u64 total = mine.class_first[c + 1] - mine.class_first[c];
          for(u32 k = 0; k <= (u32)S.parts; k++) { mine.below[c][k] = (k == 0 ? 0 : (k < (u32)S.parts ? S.small_host[128 + c * ncuts + k] : total)); }
        }
      }
    }
    else { HIP_TRY(hipStreamSynchronize(CTX.stream)); }
    }
    TRY(group_allgather(S.G, &mine, sizeof(NodeInfo), infos.data()));
    for(int h = 0; h < S.parts; h++) { if(infos[h].error != 0) { return (h == S.g ? (mine.error == 1 ? BWTM_ENOMEM : BWTM_EINVAL) : fail(BWTM_EPEER, "part %d stopped the node phase", h)); } }
    // this part's nodes of the next level: from every part's children, class after class
    u32 np = 0; u64 n = 0; level_nodes = 0;
    for(int h = 0; h < S.parts; h++) { level_nodes += infos[h].class_first[5]; }
    for(u32 c = 0; c < 5; c++)
    {
      for(int h = 0; h < S.parts; h++)
      {
        const u64 lo_x = infos[h].below[c][S.g], hi_x = infos[h].below[c][S.g + 1];
        if(lo_x >= hi_x) { continue; }
        NodePiece pc;
        pc.sp = S.theirs<const u64>(h, S.peer_lay[h].node_sp[par]); pc.r = S.theirs<const u64>(h, S.peer_lay[h].node_r[par]); pc.cnt = S.theirs<const u64>(h, S.peer_lay[h].node_cnt[par]);
        pc.src_first = infos[h].class_first[c] + lo_x; pc.count = hi_x - lo_x; pc.dst_first = n;
        host_pieces[np++] = pc; n += hi_x - lo_x;
      }
    }
    if(n > ncap) { return fail(BWTM_ENOMEM, "bwtm_part_search: %llu nodes fall into part %d's range, capacity %llu", (unsigned long long)n, S.g, (unsigned long long)ncap); }
    N = n;
    TURN(S.G);
    if(n > 0)
    {
      HIP_TRY(hipMemcpyAsync(gather_pieces.p, host_pieces, (u64)np * sizeof(NodePiece), hipMemcpyHostToDevice, CTX.stream));
      LAUNCH("nodes_gather", k_gather_nodes, div_up(n, BLOCK_THREADS), BLOCK_THREADS, gather_pieces.as<const NodePiece>(), np, n, sp.as<u64>(), r.as<u64>(), cnt.as<u64>());
      HIP_TRY(hipStreamSynchronize(CTX.stream));                     // host_pieces is rewritten by the next level
    }
    lvl++;
  }
  S.P->info.node_levels = lvl;
  done = (level_nodes == 0);
  nb_out = 1;
  if(done) { return BWTM_OK; }
  // expand: the part's nodes -> its elements, as the outputs of a step (contiguous, class 0)
  TURN(S.G);
  u64 alive = 0;
  if(N > 0)
  {
    TRY(offsets.alloc((N + 1) * 8));
    HIP_TRY(hipMemcpyAsync(offsets.p, cnt.p, N * 8, hipMemcpyDeviceToDevice, CTX.stream));
    HIP_TRY(hipMemsetAsync(offsets.as<u64>() + N, 0, 8, CTX.stream));
    TRY(device_scan<0>(offsets.as<u64>(), offsets.as<u64>(), N + 1));
    TRY(fetch_u64(offsets.as<u64>() + N, 0));
    HIP_TRY(hipStreamSynchronize(CTX.stream));
    alive = CTX.host_scratch[0];
  }
  if(alive > S.cap) { return fail(BWTM_ENOMEM, "bwtm_part_search: the nodes of part %d stand for %llu sequences, capacity %llu", S.g, (unsigned long long)alive, (unsigned long long)S.cap); }
  nb_out = std::max<u64>(1, div_up(alive, (u64)FR_BLOCK));
  if(N > 0)
  {
    LAUNCH("range_expand", k_range_expand, div_up(N, BLOCK_THREADS), BLOCK_THREADS, sp.as<const u64>(), r.as<const u64>(), cnt.as<const u64>(), offsets.as<const u64>(), N,
      S.mine<uint2>(S.lay.lo[0]), (S.wide ? S.mine<unsigned short>(S.lay.hi[0]) : nullptr), pieces.as<RangePiece>(), npieces.as<u32>(), piece_cap);
    LAUNCH("range_expand_pieces", k_range_expand_pieces, 2048, BLOCK_THREADS, pieces.as<const RangePiece>(), npieces.as<const u32>(), piece_cap,
      S.mine<uint2>(S.lay.lo[0]), (S.wide ? S.mine<unsigned short>(S.lay.hi[0]) : nullptr));
  }
  LAUNCH("frontier_init", k_frontier_init_tables, div_up(5 * nb_out + 1, BLOCK_THREADS), BLOCK_THREADS, S.mine<u64>(S.lay.seg_len[0]), S.mine<u64>(S.lay.seg_phys[0]), nb_out, alive);
  return BWTM_OK;
}

int part_search(PartSearch& S)
{
  bwtm_part* P = S.P;
  const u64 m = P->mb;
  const u64 root_first = std::min<u64>(P->cut_b[S.g], m), root_last = std::min<u64>(P->cut_b[S.g + 1], m);
  int owners = 0;
  for(int h = 0; h < S.parts; h++) { if(std::min<u64>(P->cut_b[h + 1], m) > std::min<u64>(P->cut_b[h], m)) { owners++; } }
  const u64 limit = (g_tune.range_ratio > 0 ? std::max<u64>(1, std::min<u64>(m / (u64)g_tune.range_ratio, 1ull << 24)) : 0);
  const bool use_nodes = (limit >= 1 && owners == 1);               // the root "$" must lie on one part
  TRY(search_setup(S, use_nodes));
  u64 nb_out = 1;
  int par = 0;
  if(use_nodes)
  {
    bool done = false;
    TRY(node_phase(S, root_first, root_last - root_first, limit, nb_out, done));
    if(done) { HIP_TRY(hipStreamSynchronize(CTX.stream)); return BWTM_OK; }
  }
  else
  {
    // elements from the roots on (fmi.cpp:286): the sequences of this part's range, in class 0
    const u64 count = root_last - root_first;
    if(count > S.cap) { return fail(BWTM_ENOMEM, "bwtm_part_search: %llu roots fall into part %d's range, capacity %llu", (unsigned long long)count, S.g, (unsigned long long)S.cap); }
    nb_out = std::max<u64>(1, div_up(count, (u64)FR_BLOCK));
    const u64 items = std::max<u64>(nb_out * FR_BLOCK, 5 * nb_out + 1);
    TURN(S.G);
    LAUNCH("frontier_init", k_frontier_init, div_up(items, BLOCK_THREADS), BLOCK_THREADS, S.mine<uint2>(S.lay.lo[0]), (S.wide ? S.mine<unsigned short>(S.lay.hi[0]) : nullptr),
      S.mine<u64>(S.lay.seg_len[0]), S.mine<u64>(S.lay.seg_phys[0]), nb_out, root_first, count, P->ma);
  }
  std::vector<StepInfo> all(S.parts);
  u64 pending_error = 0;
  for(u64 t = 0; ; t++)
  {
    TRY(publish_outputs(S, par, nb_out, pending_error, all.data()));
    u64 total = 0;
    for(int h = 0; h < S.parts; h++) { for(u32 c = 0; c < 5; c++) { total += all[h].cut[c][S.parts].below; } }
    if(total == 0) { break; }
    PullPlan* plan = S.plan_host[par];
    u64 n_in = 0, nseg_in = 0;
    plan_input(all.data(), S.g, S.parts, plan, n_in, nseg_in);
    for(u32 h = 0; h < PART_MAX; h++)
    {
      const int src = ((int)h < S.parts ? (int)h : S.g);
      plan->seg_len[h] = S.theirs<const u64>(src, S.peer_lay[src].seg_len[par]); plan->seg_phys[h] = S.theirs<const u64>(src, S.peer_lay[src].seg_phys[par]);
    }
    if(n_in > S.step_limit)
    {
      // everybody learns it at the next exchange and stops there; this part idles through the step
      (void)fail(BWTM_ENOMEM, "bwtm_part_search: %llu elements fall into part %d's range in step %llu, capacity %llu", (unsigned long long)n_in, S.g, (unsigned long long)t, (unsigned long long)S.step_limit);
      S.own_error = g_error;
      pending_error = 3; n_in = 0; nseg_in = 0; plan->npieces = 0; plan->nseg = 0;
    }
    if(nseg_in > S.seg_in_cap)
    {
      S.seg_in_cap = nseg_in + nseg_in / 4;
      TRY(S.seg_len_in.alloc((S.seg_in_cap + 1) * 8)); TRY(S.seg_phys_in.alloc((S.seg_in_cap + 1) * 8)); TRY(S.seg_prefix_in.alloc((S.seg_in_cap + 1) * 8));
      TRY(S.tiles_in.alloc(div_up(S.seg_in_cap + 1, (u64)SCAN_TILE) * 8, true));
    }
    TURN(S.G);
    if(div_up(nseg_in + 1, (u64)SCAN_TILE) <= CTX.pull_scan1_tiles && g_tune.frontier_unfused == 0)
    {
      // the pulled table and its scan in one launch
      LAUNCH("pull_scan", k_pull_scan1, div_up(nseg_in + 1, (u64)SCAN_TILE), BLOCK_THREADS, *plan, S.tiles_in.as<unsigned long long>(), S.tag_in,
        S.seg_phys_in.as<u64>(), S.seg_prefix_in.as<u64>(), S.first_seg_in.as<u32>(), S.emit_base.as<u64>(), S.in_epoch);
    }
    else
    {
      LAUNCH("pull_tables", k_pull_tables, div_up(nseg_in + 1, BLOCK_THREADS), BLOCK_THREADS, *plan, S.seg_len_in.as<u64>(), S.seg_phys_in.as<u64>());
      TRY(frontier_table_scan(S.seg_len_in.as<const u64>(), nseg_in, S.seg_prefix_in.as<u64>(), S.first_seg_in.as<u32>(), S.emit_base.as<u64>(), S.in_epoch, S.tiles_in, S.tag_in));
    }
    S.tag_in = (S.tag_in == 0x7FFFFFFFu ? 1 : S.tag_in + 1);
    const u64 grid = std::max<u64>(1, div_up(n_in, (u64)FR_BLOCK));
    FrontierView f;
    f.lo = nullptr; f.hi = nullptr;
    f.lo_next = S.mine<uint2>(S.lay.lo[1 - par]); f.hi_next = (S.wide ? S.mine<unsigned short>(S.lay.hi[1 - par]) : nullptr);
    f.seg_prefix = S.seg_prefix_in.as<const u64>(); f.seg_phys = S.seg_phys_in.as<const u64>(); f.first_seg = S.first_seg_in.as<const u32>();
    f.seg_len_next = S.mine<u64>(S.lay.seg_len[1 - par]); f.seg_phys_next = S.mine<u64>(S.lay.seg_phys[1 - par]);
    f.nb_max = grid;
    f.emit16 = S.emit16.as<unsigned short>(); f.emit_base = S.emit_base.as<const u64>(); f.emit_cap = S.emit_cap; f.bits32 = S.ra->bits_as<u32>();
    f.bound_row = S.bound.as<u32>() + S.in_epoch * (S.ntiles + 1) - S.tile_first; f.step = S.in_epoch; f.block_base = 0;      // indexed by absolute tile numbers
    f.src_lo = S.srcs.as<const uint2* const>() + (2 * par + 0) * PART_MAX; f.src_hi = S.srcs.as<const unsigned short* const>() + (2 * par + 1) * PART_MAX;
    f.nseg_in = nseg_in;
    if(S.wide) { LAUNCH("frontier_step", (k_frontier_step<0, true, false, true>), grid, FR_BLOCK, S.A->view(), S.B->view(), f); }
    else { LAUNCH("frontier_step", (k_frontier_step<0, false, false, true>), grid, FR_BLOCK, S.A->view(), S.B->view(), f); }
    nb_out = grid; par = 1 - par;
    S.in_epoch++; S.epoch_used += n_in;
    P->info.steps = t + 1; P->info.elements += n_in; if(n_in > P->info.largest) { P->info.largest = n_in; }
    P->info.pulled_bytes += n_in * (S.wide ? 10 : 8) + nseg_in * 16;
    if(S.in_epoch == S.EPOCH || S.epoch_used + S.cap > S.emit_cap) { TRY(part_flush(S)); }
  }
  TURN(S.G);
  TRY(frontier_flush(S.ra, S.emit16, S.emit_cap, S.emit_base, S.bound, S.ntiles, S.in_epoch, S.tile_first));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

} // namespace

extern "C" int bwtm_part_search(bwtm_part* P)
{
  if(!P) { return fail(BWTM_EINVAL, "bwtm_part_search: null argument"); }
  ENTER(P->ctx);
  if(!P->win[0] || !P->win[1]) { group_abort(P->grp); return fail(BWTM_EINVAL, "bwtm_part_search: upload both windows first (bwtm_part_upload)"); }
  if(P->ra) { bwtm_ra_free(P->ra); P->ra = nullptr; }
  int rc = ra_create_range(P->win[0], P->win[1], P->out_pos(P->part), P->out_pos(P->part + 1), &P->ra);
  if(rc == BWTM_OK && P->mb > 0)
  {
    const double t0 = group_now(); const double w0 = P->grp->wait_seconds;
    PartSearch S;
    S.P = P; S.G = P->grp; S.g = P->part; S.parts = P->parts; S.A = P->win[0]; S.B = P->win[1]; S.ra = P->ra;
    rc = part_search(S);
    if(rc != BWTM_OK) { (void)hipStreamSynchronize(CTX.stream); }
    P->info.ms_search = (group_now() - t0) * 1e3; P->info.ms_search_wait = (P->grp->wait_seconds - w0) * 1e3;
  }
  if(rc != BWTM_OK) { group_abort(P->grp); return rc; }
  P->info.bitvector_bytes = P->ra->win_words * 8;
  P->searched = true;
  return BWTM_OK;
}

namespace
{

int part_finish(bwtm_part* P, bwtm_slice** out, u64* byte_offset, u64* total_bytes, u64* next_block_start)
{
  bwtm_group* G = P->grp;
  const int g = P->part, parts = P->parts;
  bwtm_ra* ra = P->ra;
  const u64 nrecs = ra->nrecs_out;
  std::vector<u64> rec_first(parts), rec_last(parts);
  for(int h = 0; h < parts; h++)
  {
    rec_first[h] = std::min<u64>(nrecs, P->out_seg(h) * 512);
    rec_last[h] = (h + 1 == parts ? nrecs : std::min<u64>(nrecs, P->out_seg(h + 1) * 512));
  }
  // 1. what crosses a boundary: the bits of the parts before it inside a part's first segment (the ranges are the cuts rounded DOWN to segments)
  if(parts > 1)
  {
    void* arena = nullptr; PartLayout lay_all[PART_MAX];
    // the rows live in this merge's exported block (a merge without sequences has not searched: a block of its own)
    PartLayout mine_lay = (P->lay.bytes > 0 ? P->lay : part_layout(1, 1, false, parts));
    TRY(group_arena(G, mine_lay.bytes, &arena));
    TRY(group_allgather(G, &mine_lay, sizeof(PartLayout), lay_all));
    auto touches = [&](int h, int k)                                 // part h has bits inside part k's first segment (h < k)
    {
      const u64 lo = std::max<u64>(P->out_seg(k) << 16, P->out_pos(h)), hi = std::min<u64>(P->out_pos(k), P->out_pos(h + 1));
      return lo < hi;
    };
    const u64 nwords = ra->nchunks * CHUNK_WORDS, seg_words = BOUNDARY_BYTES / 8;
    {
    TURN(G);
    for(int k = g + 1; k < parts; k++)
    {
      u64* row = (u64*)((char*)arena + mine_lay.boundary + (u64)k * BOUNDARY_BYTES);
      HIP_TRY(hipMemsetAsync(row, 0, BOUNDARY_BYTES, CTX.stream));
      if(!touches(g, k)) { continue; }
      const u64 w0 = (P->out_seg(k) << 16) >> 6, w1 = std::min<u64>(nwords, w0 + seg_words);
      const u64 h0 = std::max<u64>(w0, ra->win_word_first), h1 = std::min<u64>(w1, ra->win_word_first + ra->win_words);
      if(h1 > h0) { HIP_TRY(hipMemcpyAsync(row + (h0 - w0), ra->bits_as<const u64>() + h0, (h1 - h0) * 8, hipMemcpyDeviceToDevice, CTX.stream)); }
    }
    HIP_TRY(hipStreamSynchronize(CTX.stream));
    }
    TRY(group_barrier(G));
    TURN(G);
    for(int h = 0; h < g; h++)
    {
      if(!touches(h, g)) { continue; }
      void* pa = nullptr; TRY(group_peer_arena(G, h, &pa));
      const u64* row = (const u64*)((const char*)pa + lay_all[h].boundary + (u64)g * BOUNDARY_BYTES);
      const u64 w0 = (P->out_seg(g) << 16) >> 6, w1 = std::min<u64>(nwords, w0 + seg_words);
      const u64 h0 = std::max<u64>(w0, ra->win_word_first), h1 = std::min<u64>(w1, ra->win_word_first + ra->win_words);
      if(h1 > h0) { LAUNCH("bits_or", k_bits_or_peer, div_up(h1 - h0, BLOCK_THREADS), BLOCK_THREADS, ra->bits_as<u64>() + h0, row + (h0 - w0), h1 - h0); }
    }
    HIP_TRY(hipStreamSynchronize(CTX.stream));
    P->info.boundary_bytes = BOUNDARY_BYTES;
  }
  // 2. the small exchange of the ranges: set bits of every range, local offsets of the supers that start in it, its last chunk of bits
  const u64 nsup = num_supers(ra->n_out);
  std::vector<u64> mine(1 + CHUNK_WORDS + nsup, 0), everyone((u64)parts * (1 + CHUNK_WORDS + nsup), 0);
  { TURN(G); TRY(bwtm_ra_range_counts(ra, rec_first[g], rec_last[g], &mine[0], mine.data() + 1 + CHUNK_WORDS, mine.data() + 1)); }
  TRY(group_allgather(G, mine.data(), mine.size() * 8, everyone.data()));
  const u64 stride = mine.size();
  u64 before = 0, total = 0;
  for(int h = 0; h < parts; h++) { if(h < g) { before += everyone[h * stride]; } total += everyone[h * stride]; }
  if(total != P->nb) { return fail(BWTM_EINVAL, "bwtm_part_finish: the parts' ranges hold %llu set bits, b has %llu positions", (unsigned long long)total, (unsigned long long)P->nb); }
  std::vector<u64> super_boff(nsup, 0);
  for(u64 sb = 0; sb < nsup; sb++)
  {
    const u64 q = sb << SUPER_REC_SHIFT;                             // the super's first record
    u64 prefix = 0;
    for(int h = 0; h < parts; h++)
    {
      if(q >= rec_first[h] && q < rec_last[h]) { super_boff[sb] = prefix + everyone[h * stride + 1 + CHUNK_WORDS + sb]; break; }
      prefix += everyone[h * stride];
    }
  }
  const u64* halo = nullptr;
  for(int h = g; h-- > 0; ) { if(rec_last[h] > rec_first[h]) { halo = everyone.data() + h * stride + 1; break; } }
  // 3. this part's range of the output; the encoder's two carries (api/slices.hip.h)
  bwtm_slice* slice = nullptr;
  u64 head = 0;
  int rc = BWTM_OK;
  {
    TURN(G);
    TRY(bwtm_ra_finalize_range(ra, rec_first[g], rec_last[g], before, total, super_boff.data(), halo));
    TRY(bwtm_interleave_range(P->win[0], P->win[1], ra, rec_first[g], rec_last[g], &slice));
    bwtm_ra_free(P->ra); P->ra = nullptr;
    for(int k = 0; k < 2; k++) { bwtm_index_free(P->win[k]); P->win[k] = nullptr; }
    rc = bwtm_slice_lasthead(slice, &head);
  }
  auto fail_slice = [&](int rc2) { bwtm_slice_free(slice); return rc2; };
  if(rc != BWTM_OK) { return fail_slice(rc); }
  std::vector<u64> heads(parts, 0);
  rc = group_allgather(G, &head, 8, heads.data());
  if(rc != BWTM_OK) { return fail_slice(rc); }
  u64 head_before = 0;
  for(int h = 0; h < g; h++) { head_before = std::max(head_before, heads[h]); }
  u64 table[64]; std::vector<u64> tables((u64)parts * 64, 0), offsets(parts + 1, 0);
  { Turn turn_(G); rc = (turn_.rc != BWTM_OK ? turn_.rc : bwtm_slice_size_table(slice, head_before, table)); }
  if(rc == BWTM_OK) { rc = group_allgather(G, table, sizeof(table), tables.data()); }
  if(rc == BWTM_OK) { rc = bwtm_fold_offsets(tables.data(), parts, offsets.data()); }
  u64 first_start = ~0ull;
  if(rc == BWTM_OK)
  {
    Turn turn_(G);
    rc = (turn_.rc != BWTM_OK ? turn_.rc : bwtm_slice_encode(slice, offsets[g]));
    if(rc == BWTM_OK) { rc = bwtm_slice_first_block_start(slice, &first_start); }
  }
  if(rc != BWTM_OK) { return fail_slice(rc); }
  std::vector<u64> starts(parts, ~0ull);
  rc = group_allgather(G, &first_start, 8, starts.data());
  if(rc != BWTM_OK) { return fail_slice(rc); }
  u64 next = P->na + P->nb;
  for(int h = parts; h-- > g + 1; ) { if(starts[h] != ~0ull) { next = starts[h]; } }
  *out = slice; *byte_offset = offsets[g]; *total_bytes = offsets[parts]; *next_block_start = next;
  return BWTM_OK;
}

} // namespace

extern "C" int bwtm_part_finish(bwtm_part* P, bwtm_slice** out, uint64_t* byte_offset, uint64_t* total_bytes, uint64_t* next_block_start)
{
  if(!P || !out || !byte_offset || !total_bytes || !next_block_start) { return fail(BWTM_EINVAL, "bwtm_part_finish: null argument"); }
  ENTER(P->ctx);
  if(!P->searched || !P->ra) { group_abort(P->grp); return fail(BWTM_EINVAL, "bwtm_part_finish: call bwtm_part_search first"); }
  const double t0 = group_now();
  int rc = part_finish(P, out, byte_offset, total_bytes, next_block_start);
  if(rc != BWTM_OK) { group_abort(P->grp); return rc; }
  P->info.ms_finish = (group_now() - t0) * 1e3;
  P->searched = false;
  return BWTM_OK;
}

extern "C" int bwtm_part_stats(const bwtm_part* P, bwtm_part_info* info)
{
  if(!P || !info) { return fail(BWTM_EINVAL, "bwtm_part_stats: null argument"); }
  *info = P->info;
  return BWTM_OK;
}

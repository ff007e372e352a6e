/*
  api/fslice.hip.h -- one GPU's state of the SLICED frontier search: the dense multi-GPU form of buildRA (fmi.cpp:272-334) in which
  every GPU advances a contiguous slice of the sorted frontier (kernels/search_frontier.hip.h, k_frontier_gather).  Part of bwtm_api.hip.

  A functional prototype: results are exact (tests/test_gpu_sliced.py against the oracle, contexts of one GPU standing in for GPUs);
  the sequence-sharded search (bwtm_search on a block of sequences per GPU) stays the default until a multi-GPU measurement decides.
  Per step the caller (one thread or process per GPU) runs
      bwtm_fslice_gather(fs, views of ALL GPUs, first, last)   -- pull this GPU's slice [first, last) of the global frontier
      <barrier: every GPU has gathered>
      bwtm_fslice_advance(fs)                                   -- one LF step on the slice; the outputs replace the old ones
      bwtm_fslice_export(fs, &view)                             -- synchronizes; pointers + per-class totals for the next step
      <barrier: every GPU has exported>
  Views hold raw device pointers: valid for contexts of one device, and across devices when the peers can access each other's
  memory (the exported buffers come from hipMalloc, not from the pool's device-local mapped blocks).
*/
#pragma once

struct bwtm_fslice
{
  bwtm_context* ctx = nullptr;
  bwtm_fslice() : ctx(t_ctx) { if(ctx) { ctx->live_handles++; } }
  ~bwtm_fslice()
  {
    if(ctx) { ctx->live_handles--; }
    for(void* p : exported) { if(p) { (void)hipFree(p); } }
    if(host_pieces) { (void)hipHostFree(host_pieces); }
    if(host_below) { (void)hipHostFree(host_below); }
    if(host_dense_pieces) { (void)hipHostFree(host_dense_pieces); }
    if(host_node_pieces) { (void)hipHostFree(host_node_pieces); }
  }
  bwtm_fslice(const bwtm_fslice&) = delete; bwtm_fslice& operator=(const bwtm_fslice&) = delete;
  const bwtm_index* a = nullptr; const bwtm_index* b = nullptr; bwtm_ra* ra = nullptr;
  u64 cap = 0, nbl = 0, nseg = 0, ntiles = 0;
  bool wide = false;
  u64 nb_out = 1;                         // blocks of the last outputs' segment tables (class c's entries start at c * nb_out): the tables are laid out for the step's
                                          // own size, not for the capacity, so that the per-step scans and the peers' lookups shrink with the frontier
  // exported to the peers (hipMalloc): this GPU's outputs of the last step
  uint2* lo_out = nullptr; unsigned short* hi_out = nullptr; u64* out_prefix = nullptr; u64* seg_phys_out = nullptr;
  std::vector<void*> exported;
  u64 totals[5] = {0, 0, 0, 0, 0};
  // local
  DevBuf lo_in, hi_in, seg_len_in, seg_phys_in, seg_prefix_in, first_seg, scan_partial, seg_len_out, pieces;
  SlicePiece* host_pieces = nullptr; u32 max_pieces = 0;
  u64 n_in = 0;
  DevBuf emit16, emit_base, bound; u64 emit_cap = 0, EPOCH = 1, in_epoch = 0, epoch_used = 0;
  // fixed cuts (bwtm_fslice_set_cuts): the cuts on the device, the counts below them of the last outputs (device and page-locked host)
  DevBuf cuts, below, dense_pieces; u64* host_below = nullptr; u32 ncuts = 0;
  uint2* dense_lo = nullptr; unsigned short* dense_hi = nullptr;     // exported (hipMalloc): the outputs in logical order, the send buffer of the exchange
  DensePiece* host_dense_pieces = nullptr;
  // node phase over partitioned records (bwtm_fslice_nodes_*): this level's nodes [0], their children [1] (exported, hipMalloc)
  u64* node_sp[2] = {nullptr, nullptr}; u64* node_r[2] = {nullptr, nullptr}; u64* node_cnt[2] = {nullptr, nullptr};
  u64 node_cap = 0, nodes = 0;
  DevBuf node_flags, node_pieces, node_npieces, node_class_first, node_err, node_gather_pieces;
  u32 node_piece_cap = 0;
  NodePiece* host_node_pieces = nullptr;
};

namespace
{

int fslice_scan_outputs(bwtm_fslice* fs)
{
  TRY(device_scan<0>(fs->seg_len_out.as<u64>(), fs->out_prefix, 5 * fs->nb_out + 1));
  for(u32 c = 0; c <= 5; c++) { TRY(fetch_u64(fs->out_prefix + (u64)c * fs->nb_out, 96 + c)); }     // class boundaries -> totals (bwtm_fslice_export)
  if(fs->ncuts > 0)
  {
    LAUNCH("compact_outputs", k_compact_outputs, fs->nb_out, FR_BLOCK, (const uint2*)fs->lo_out, (const unsigned short*)fs->hi_out, fs->seg_len_out.as<const u64>(), (const u64*)fs->seg_phys_out,
      (const u64*)fs->out_prefix, fs->nb_out, fs->dense_lo, fs->dense_hi);
    LAUNCH("cut_counts", k_cut_search, 1, BLOCK_THREADS, (const uint2*)fs->dense_lo, (const unsigned short*)fs->dense_hi, (const u64*)fs->out_prefix, fs->nb_out,
      fs->cuts.as<const u64>(), fs->ncuts, fs->below.as<u64>());
    HIP_TRY(hipMemcpyAsync(fs->host_below, fs->below.p, 5ull * fs->ncuts * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream));
  }
  return BWTM_OK;
}

template<class T> int fslice_export_alloc(bwtm_fslice* fs, T*& p, u64 count)
{
  void* q = nullptr;
  HIP_TRY(hipMalloc(&q, std::max<u64>(count, 1) * sizeof(T)));
  fs->exported.push_back(q);
  p = (T*)q;
  return BWTM_OK;
}

} // namespace

extern "C" int bwtm_fslice_create(const bwtm_index* a, const bwtm_index* b, bwtm_ra* ra, uint64_t capacity, int parts, bwtm_fslice** out)
{
  if(!a || !b || !ra || !out || parts < 1 || capacity == 0) { return fail(BWTM_EINVAL, "bwtm_fslice_create: bad argument"); }
  if(a->ctx != ra->ctx || b->ctx != ra->ctx) { return fail(BWTM_EINVAL, "bwtm_fslice_create: handles of different contexts"); }
  ENTER(ra->ctx);
  if(a->n >= (1ull << 40) || b->n >= (1ull << 40) || capacity >= (1ull << 32)) { return fail(BWTM_EINVAL, "bwtm_fslice_create: coordinates do not fit 40 bits"); }
  bwtm_fslice* fs = new bwtm_fslice();
  fs->ctx = t_ctx; fs->a = a; fs->b = b; fs->ra = ra;
  auto body = [&]() -> int
  {
    fs->cap = capacity; fs->nbl = div_up(capacity, FR_BLOCK); fs->nseg = 5 * fs->nbl;
    fs->ntiles = div_up(ra->n_out + 1, 1ull << TILE_SHIFT);
    fs->wide = (a->n >= (1ull << 32) || b->n >= (1ull << 32));
    const u64 fcap = fs->nbl * FR_BLOCK;
    TRY(fslice_export_alloc(fs, fs->lo_out, fcap));
    if(fs->wide) { TRY(fslice_export_alloc(fs, fs->hi_out, fcap)); }
    TRY(fslice_export_alloc(fs, fs->out_prefix, fs->nseg + 1));
    TRY(fslice_export_alloc(fs, fs->seg_phys_out, fs->nseg + 1));
    TRY(fs->lo_in.alloc(fcap * 8)); if(fs->wide) { TRY(fs->hi_in.alloc(fcap * 2)); }
    TRY(fs->seg_len_in.alloc((fs->nseg + 1) * sizeof(u64), true)); TRY(fs->seg_phys_in.alloc((fs->nseg + 1) * sizeof(u64), true));
    TRY(fs->seg_prefix_in.alloc((fs->nseg + 1) * sizeof(u64)));
    TRY(fs->first_seg.alloc((fs->nbl + 1) * sizeof(u32)));
    TRY(fs->scan_partial.alloc(div_up(fs->nseg + 1, (u64)SCAN_TILE) * sizeof(u64)));
    TRY(fs->seg_len_out.alloc((fs->nseg + 1) * sizeof(u64), true));
    fs->max_pieces = (u32)(5 * parts);
    TRY(fs->pieces.alloc((u64)fs->max_pieces * sizeof(SlicePiece)));
    HIP_TRY(hipHostMalloc((void**)&fs->host_pieces, (u64)fs->max_pieces * sizeof(SlicePiece), hipHostMallocDefault));
    // dense emits of an epoch of steps, as in search_frontier()
    const u64 per_seq = b->n / (b->m > 0 ? b->m : 1) + 1;
    fs->emit_cap = std::min<u64>((u64)(g_tune.emit_budget > 0 ? g_tune.emit_budget : (16ll << 30)) / sizeof(unsigned short), 2 * capacity * per_seq + (1ull << 20));
    if(fs->emit_cap < capacity) { fs->emit_cap = capacity; }
    fs->EPOCH = std::max<u64>(1, std::min<u64>((u64)std::max<long long>(1, g_tune.frontier_epoch), fs->emit_cap / capacity));
    const u64 bound_budget = 2ull << 30;
    if(fs->EPOCH * (fs->ntiles + 1) * sizeof(u32) > bound_budget) { fs->EPOCH = std::max<u64>(1, bound_budget / ((fs->ntiles + 1) * sizeof(u32))); }
    TRY(fs->emit16.alloc((fs->emit_cap + 16) * sizeof(unsigned short)));
    TRY(fs->emit_base.alloc((fs->EPOCH + 1) * sizeof(u64), true));
    TRY(fs->bound.alloc(fs->EPOCH * (fs->ntiles + 1) * sizeof(u32)));
    HIP_TRY(hipMemsetAsync(fs->bound.p, 0xFF, fs->EPOCH * (fs->ntiles + 1) * sizeof(u32), CTX.stream));
    return BWTM_OK;
  };
  int rc = body();
  if(rc != BWTM_OK) { delete fs; return rc; }
  *out = fs;
  return BWTM_OK;
}

extern "C" void bwtm_fslice_free(bwtm_fslice* fs)
{
  if(!fs) { return; }
  Scope scope(fs->ctx);
  if(scope.rc == BWTM_OK) { (void)hipStreamSynchronize(CTX.stream); }
  delete fs;
}

/* The outputs of "step -1": this GPU's block of b's sequences as the roots of their chains (fmi.cpp:286), in class 0. */
extern "C" int bwtm_fslice_seed(bwtm_fslice* fs, uint64_t seq_first, uint64_t count)
{
  if(!fs) { return fail(BWTM_EINVAL, "bwtm_fslice_seed: null argument"); }
  ENTER(fs->ctx);
  if(count > fs->cap) { return fail(BWTM_EINVAL, "bwtm_fslice_seed: %llu sequences for a capacity of %llu", (unsigned long long)count, (unsigned long long)fs->cap); }
  if(count > 0 && seq_first + count > fs->b->m) { return fail(BWTM_EINVAL, "bwtm_fslice_seed: sequences out of range"); }
  fs->nb_out = std::max<u64>(1, div_up(count, (u64)FR_BLOCK));
  const u64 items = std::max<u64>(fs->nb_out * FR_BLOCK, 5 * fs->nb_out + 1);
  LAUNCH("frontier_init", k_frontier_init, div_up(items, BLOCK_THREADS), BLOCK_THREADS, fs->lo_out, fs->hi_out, fs->seg_len_out.as<u64>(), fs->seg_phys_out, fs->nb_out,
    seq_first, count, fs->a->m);
  TRY(fslice_scan_outputs(fs));
  return BWTM_OK;
}

extern "C" int bwtm_fslice_export(bwtm_fslice* fs, bwtm_fslice_view* view)
{
  if(!fs || !view) { return fail(BWTM_EINVAL, "bwtm_fslice_export: null argument"); }
  ENTER(fs->ctx);
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  for(u32 c = 0; c < 5; c++) { fs->totals[c] = CTX.host_scratch[96 + c + 1] - CTX.host_scratch[96 + c]; view->totals[c] = fs->totals[c]; }
  view->lo = fs->lo_out; view->hi = fs->hi_out; view->prefix = fs->out_prefix; view->phys = fs->seg_phys_out; view->blocks = fs->nb_out;
  for(u32 c = 0; c < 5; c++)
  {
    for(u32 k = 0; k <= BWTM_X_MAX_PARTS; k++) { view->below[c][k] = (k + 1 < fs->ncuts ? fs->host_below[c * fs->ncuts + k] : fs->totals[c]); }      // past the last cut: everything
    if(fs->ncuts > 0) { view->below[c][0] = 0; }                                                                                            // R_0 = 0
  }
  view->dense_lo = fs->dense_lo; view->dense_hi = fs->dense_hi;
  for(u32 c = 0; c <= 5; c++) { view->class_first[c] = CTX.host_scratch[96 + c]; }
  return BWTM_OK;
}

extern "C" int bwtm_fslice_gather(bwtm_fslice* fs, const bwtm_fslice_view* views, int parts, uint64_t first, uint64_t last)
{
  if(!fs || !views || parts < 1 || first > last) { return fail(BWTM_EINVAL, "bwtm_fslice_gather: bad argument"); }
  ENTER(fs->ctx);
  if(last - first > fs->cap) { return fail(BWTM_EINVAL, "bwtm_fslice_gather: slice of %llu elements, capacity %llu", (unsigned long long)(last - first), (unsigned long long)fs->cap); }
  if((u32)(5 * parts) > fs->max_pieces) { return fail(BWTM_EINVAL, "bwtm_fslice_gather: more parts than the slice was created for"); }
  // The global logical order of the frontier is (class, GPU, block): walk the (class, GPU) pieces and keep what falls into [first, last).
  u32 np = 0; u64 base = 0;
  for(u32 c = 0; c < 5; c++)
  {
    for(int h = 0; h < parts; h++)
    {
      const u64 len = views[h].totals[c];
      const u64 lo_g = std::max<u64>(base, first), hi_g = std::min<u64>(base + len, last);
      if(lo_g < hi_g)
      {
        SlicePiece pc;
        pc.lo = (const uint2*)views[h].lo; pc.hi = (const unsigned short*)views[h].hi;
        pc.prefix = (const u64*)views[h].prefix; pc.phys = (const u64*)views[h].phys;
        pc.seg_first = (u64)c * views[h].blocks; pc.seg_count = views[h].blocks;
        pc.src_first = lo_g - base; pc.count = hi_g - lo_g; pc.dst_first = lo_g - first;
        fs->host_pieces[np++] = pc;
      }
      base += len;
    }
  }
  fs->n_in = last - first;
  if(fs->n_in == 0) { return BWTM_OK; }
  if(last > base) { return fail(BWTM_EINVAL, "bwtm_fslice_gather: the slice ends at %llu, the frontier holds %llu elements", (unsigned long long)last, (unsigned long long)base); }
  HIP_TRY(hipMemcpyAsync(fs->pieces.p, fs->host_pieces, (u64)np * sizeof(SlicePiece), hipMemcpyHostToDevice, CTX.stream));
  LAUNCH("frontier_gather", k_frontier_gather, div_up(fs->n_in, BLOCK_THREADS), BLOCK_THREADS, fs->pieces.as<const SlicePiece>(), np, fs->n_in,
    fs->lo_in.as<uint2>(), fs->hi_in.as<unsigned short>());
  HIP_TRY(hipStreamSynchronize(CTX.stream));                       // the peers may overwrite their outputs once every GPU has returned from here
  return BWTM_OK;
}

extern "C" int bwtm_fslice_advance(bwtm_fslice* fs)
{
  if(!fs) { return fail(BWTM_EINVAL, "bwtm_fslice_advance: null argument"); }
  ENTER(fs->ctx);
  const u64 nbl = std::max<u64>(1, div_up(fs->n_in, (u64)FR_BLOCK)), nseg = 5 * nbl;       // this step's tables: sized for its input, not for the capacity
  fs->nb_out = nbl;
  const u64 scan_tiles = div_up(nseg + 1, (u64)SCAN_TILE);
  // the gathered slice in the layout of a fresh frontier: contiguous, class-0 segments
  LAUNCH("frontier_init", k_frontier_init_tables, div_up(nseg + 1, BLOCK_THREADS), BLOCK_THREADS, fs->seg_len_in.as<u64>(), fs->seg_phys_in.as<u64>(), nbl, fs->n_in);
  if(scan_tiles > 1)
  {
    LAUNCH("scan_reduce", k_scan_reduce<0>, scan_tiles, BLOCK_THREADS, fs->seg_len_in.as<const u64>(), fs->scan_partial.as<u64>(), nseg + 1, (u64)0, scan_tiles);
  }
  LAUNCH("frontier_scan", k_frontier_scan, scan_tiles, BLOCK_THREADS, fs->seg_len_in.as<const u64>(), fs->scan_partial.as<const u64>(), nseg,
    fs->seg_prefix_in.as<u64>(), fs->first_seg.as<u32>(), fs->emit_base.as<u64>(), fs->in_epoch, (u64*)nullptr);
  FrontierView f;
  f.lo = fs->lo_in.as<const uint2>(); f.hi = fs->hi_in.as<const unsigned short>();
  f.lo_next = fs->lo_out; f.hi_next = fs->hi_out;
  f.seg_prefix = fs->seg_prefix_in.as<const u64>(); f.seg_phys = fs->seg_phys_in.as<const u64>(); f.first_seg = fs->first_seg.as<const u32>();
  f.seg_len_next = fs->seg_len_out.as<u64>(); f.seg_phys_next = fs->seg_phys_out;
  f.nb_max = nbl;
  f.emit16 = fs->emit16.as<unsigned short>(); f.emit_base = fs->emit_base.as<const u64>(); f.emit_cap = fs->emit_cap; f.bits32 = fs->ra->bits_as<u32>();
  f.bound_row = fs->bound.as<u32>() + fs->in_epoch * (fs->ntiles + 1); f.step = fs->in_epoch; f.block_base = 0;
  if(fs->wide) { LAUNCH("frontier_step", (k_frontier_step<0, true>), nbl, FR_BLOCK, fs->a->view(), fs->b->view(), f); }
  else { LAUNCH("frontier_step", (k_frontier_step<0, false>), nbl, FR_BLOCK, fs->a->view(), fs->b->view(), f); }
  fs->in_epoch++; fs->epoch_used += fs->n_in;
  if(fs->in_epoch == fs->EPOCH || fs->epoch_used + fs->cap > fs->emit_cap)
  {
    TRY(frontier_flush(fs->ra, fs->emit16, fs->emit_cap, fs->emit_base, fs->bound, fs->ntiles, fs->in_epoch));
    HIP_TRY(hipMemsetAsync(fs->bound.p, 0xFF, fs->EPOCH * (fs->ntiles + 1) * sizeof(u32), CTX.stream));
    HIP_TRY(hipMemsetAsync(fs->emit_base.p, 0, (fs->EPOCH + 1) * sizeof(u64), CTX.stream));
    fs->in_epoch = 0; fs->epoch_used = 0;
  }
  TRY(fslice_scan_outputs(fs));
  return BWTM_OK;
}

extern "C" int bwtm_fslice_finish(bwtm_fslice* fs)
{
  if(!fs) { return fail(BWTM_EINVAL, "bwtm_fslice_finish: null argument"); }
  ENTER(fs->ctx);
  TRY(frontier_flush(fs->ra, fs->emit16, fs->emit_cap, fs->emit_base, fs->bound, fs->ntiles, fs->in_epoch));
  fs->in_epoch = 0; fs->epoch_used = 0;
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" int bwtm_x_device_scan(const uint64_t* in, uint64_t* out, uint64_t n, uint64_t narrays, int op)
{
  if(!in || !out || n == 0 || narrays == 0 || (op != 0 && op != 1)) { return fail(BWTM_EINVAL, "bwtm_x_device_scan: bad argument"); }
  ENTER(nullptr);
  DevBuf buf; TRY(buf.alloc(n * narrays * sizeof(u64)));
  HIP_TRY(hipMemcpyAsync(buf.p, in, n * narrays * sizeof(u64), hipMemcpyHostToDevice, CTX.stream));
  if(op == 0) { TRY(device_scan_multi<0>(buf.as<u64>(), buf.as<u64>(), n, narrays, n)); }
  else { TRY(device_scan_multi<1>(buf.as<u64>(), buf.as<u64>(), n, narrays, n)); }
  HIP_TRY(hipMemcpyAsync(out, buf.p, n * narrays * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" int bwtm_fslice_set_cuts(bwtm_fslice* fs, const uint64_t* r_cuts, int parts)
{
  if(!fs || !r_cuts || parts < 1 || parts > BWTM_X_MAX_PARTS) { return fail(BWTM_EINVAL, "bwtm_fslice_set_cuts: bad argument (at most %d parts)", BWTM_X_MAX_PARTS); }
  for(int k = 0; k < parts; k++) { if(r_cuts[k] > r_cuts[k + 1] && k + 1 < parts) { return fail(BWTM_EINVAL, "bwtm_fslice_set_cuts: cuts must not decrease"); } }
  if(r_cuts[0] != 0) { return fail(BWTM_EINVAL, "bwtm_fslice_set_cuts: the first cut is 0"); }
  ENTER(fs->ctx);
  if((u32)(5 * parts) > fs->max_pieces) { return fail(BWTM_EINVAL, "bwtm_fslice_set_cuts: more parts than the slice was created for"); }
  fs->ncuts = (u32)parts + 1;
  TRY(fs->cuts.alloc(fs->ncuts * sizeof(u64))); TRY(fs->below.alloc(5ull * fs->ncuts * sizeof(u64), true));
  if(!fs->host_below) { HIP_TRY(hipHostMalloc((void**)&fs->host_below, 5ull * (BWTM_X_MAX_PARTS + 1) * sizeof(u64), hipHostMallocDefault)); }
  if(!fs->dense_lo)
  {
    const u64 fcap = fs->nbl * FR_BLOCK;
    TRY(fslice_export_alloc(fs, fs->dense_lo, fcap));
    if(fs->wide) { TRY(fslice_export_alloc(fs, fs->dense_hi, fcap)); }
    TRY(fs->dense_pieces.alloc((u64)fs->max_pieces * sizeof(DensePiece)));
    HIP_TRY(hipHostMalloc((void**)&fs->host_dense_pieces, (u64)fs->max_pieces * sizeof(DensePiece), hipHostMallocDefault));
  }
  std::memset(fs->host_below, 0, 5ull * (BWTM_X_MAX_PARTS + 1) * sizeof(u64));
  u64 host_cuts[BWTM_X_MAX_PARTS + 1];
  for(int k = 0; k <= parts; k++) { host_cuts[k] = (k == parts ? ~0ull : r_cuts[k]); }
  HIP_TRY(hipMemcpyAsync(fs->cuts.p, host_cuts, fs->ncuts * sizeof(u64), hipMemcpyHostToDevice, CTX.stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));                       // host_cuts lives on this stack
  return BWTM_OK;
}

extern "C" int bwtm_fslice_gather_cut(bwtm_fslice* fs, const bwtm_fslice_view* views, int parts, int part)
{
  if(!fs || !views || parts < 1 || part < 0 || part >= parts) { return fail(BWTM_EINVAL, "bwtm_fslice_gather_cut: bad argument"); }
  ENTER(fs->ctx);
  if(fs->ncuts != (u32)parts + 1) { return fail(BWTM_EINVAL, "bwtm_fslice_gather_cut: bwtm_fslice_set_cuts was not called for %d parts", parts); }
  // The global order of the frontier is (class, GPU, block) and it is sorted by position: this GPU's elements are, in every (class, GPU)
  // piece, the range between the piece's counts below this GPU's two cuts.
  u32 np = 0; u64 n_in = 0;
  for(u32 c = 0; c < 5; c++)
  {
    for(int h = 0; h < parts; h++)
    {
      const u64 lo_x = views[h].below[c][part], hi_x = views[h].below[c][part + 1];
      if(lo_x > hi_x || hi_x > views[h].totals[c]) { return fail(BWTM_EINVAL, "bwtm_fslice_gather_cut: counts of GPU %d, class %u are not monotone", h, c); }
      if(lo_x < hi_x)
      {
        DensePiece pc;
        pc.lo = (const uint2*)views[h].dense_lo; pc.hi = (const unsigned short*)views[h].dense_hi;
        pc.src_first = views[h].class_first[c] + lo_x; pc.count = hi_x - lo_x; pc.dst_first = n_in;
        fs->host_dense_pieces[np++] = pc;
        n_in += hi_x - lo_x;
      }
    }
  }
  if(n_in > fs->cap) { return fail(BWTM_EINVAL, "bwtm_fslice_gather_cut: %llu elements fall into this GPU's range, capacity %llu", (unsigned long long)n_in, (unsigned long long)fs->cap); }
  fs->n_in = n_in;
  if(n_in == 0) { return BWTM_OK; }
  HIP_TRY(hipMemcpyAsync(fs->dense_pieces.p, fs->host_dense_pieces, (u64)np * sizeof(DensePiece), hipMemcpyHostToDevice, CTX.stream));
  LAUNCH("frontier_gather", k_gather_dense, div_up(fs->n_in, BLOCK_THREADS), BLOCK_THREADS, fs->dense_pieces.as<const DensePiece>(), np, fs->n_in,
    fs->lo_in.as<uint2>(), fs->hi_in.as<unsigned short>());
  HIP_TRY(hipStreamSynchronize(CTX.stream));                       // the peers may overwrite their outputs once every GPU has returned from here
  return BWTM_OK;
}

extern "C" int bwtm_x_index_window(const bwtm_index* whole, uint64_t pos_first, uint64_t pos_last, bwtm_index** out)
{
  if(!whole || !out || pos_first > pos_last) { return fail(BWTM_EINVAL, "bwtm_x_index_window: bad argument"); }
  ENTER(whole->ctx);
  WHOLE_INDEX(whole, "bwtm_x_index_window");
  if(whole->nrecs == 0 || !whole->recs.p) { return fail(BWTM_EINVAL, "bwtm_x_index_window: the index holds no records"); }
  const u64 q0 = std::min<u64>(pos_first >> REC_SHIFT, whole->nrecs - 1), q1 = std::min<u64>(pos_last >> REC_SHIFT, whole->nrecs - 1);
  bwtm_index* w = new bwtm_index();
  auto body = [&]() -> int
  {
    w->n = whole->n; w->m = whole->m; w->nrecs = whole->nrecs; w->nsup = whole->nsup;
    for(int c = 0; c < 8; c++) { w->C[c] = whole->C[c]; }
    w->windowed = true; w->win_first = q0; w->win_count = q1 - q0 + 1;
    TRY(w->recs.alloc(w->win_count * 64));
    HIP_TRY(hipMemcpyAsync(w->recs.p, (const char*)whole->recs.p + (q0 << 6), w->win_count * 64, hipMemcpyDeviceToDevice, CTX.stream));
    TRY(w->sup.alloc(whole->nsup * SUP_STRIDE * sizeof(u64)));
    HIP_TRY(hipMemcpyAsync(w->sup.p, whole->sup.p, whole->nsup * SUP_STRIDE * sizeof(u64), hipMemcpyDeviceToDevice, CTX.stream));
    return BWTM_OK;
  };
  int rc = body();
  if(rc != BWTM_OK) { delete w; return rc; }
  *out = w;
  return BWTM_OK;
}

extern "C" uint64_t bwtm_x_index_record_bytes(const bwtm_index* x)
{
  if(!x) { return 0; }
  return (x->windowed ? x->win_count : x->nrecs) * 64;
}

extern "C" int bwtm_fslice_nodes_begin(bwtm_fslice* fs, uint64_t seq_first, uint64_t count, uint64_t node_capacity)
{
  if(!fs || node_capacity == 0) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_begin: bad argument"); }
  ENTER(fs->ctx);
  if(fs->ncuts == 0) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_begin: call bwtm_fslice_set_cuts first"); }
  if(count > 0 && seq_first + count > fs->b->m) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_begin: sequences out of range"); }
  if(!fs->node_sp[0])
  {
    fs->node_cap = node_capacity;
    for(int k = 0; k < 2; k++)
    {
      TRY(fslice_export_alloc(fs, fs->node_sp[k], node_capacity)); TRY(fslice_export_alloc(fs, fs->node_r[k], node_capacity)); TRY(fslice_export_alloc(fs, fs->node_cnt[k], node_capacity));
    }
    TRY(fs->node_flags.alloc((5 * node_capacity + 1) * sizeof(u64)));
    fs->node_piece_cap = (u32)std::min<u64>(fs->cap / 16 + 1024, 1ull << 24);
    TRY(fs->node_pieces.alloc((u64)fs->node_piece_cap * sizeof(RangePiece)));
    TRY(fs->node_npieces.alloc(sizeof(u32), true));
    TRY(fs->node_class_first.alloc(6 * sizeof(u64)));
    TRY(fs->node_err.alloc(sizeof(u32), true));
    TRY(fs->node_gather_pieces.alloc((u64)fs->max_pieces * sizeof(NodePiece)));
    HIP_TRY(hipHostMalloc((void**)&fs->host_node_pieces, (u64)fs->max_pieces * sizeof(NodePiece), hipHostMallocDefault));
  }
  else if(node_capacity > fs->node_cap) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_begin: the node buffers were created for %llu nodes", (unsigned long long)fs->node_cap); }
  fs->nodes = 0;
  if(count > 0)
  {
    LAUNCH("range_init", k_range_init, 1, BLOCK_THREADS, fs->node_sp[0], fs->node_r[0], fs->node_cnt[0], seq_first, count, fs->a->m);
    fs->nodes = 1;
  }
  return BWTM_OK;
}

extern "C" int bwtm_fslice_nodes_step(bwtm_fslice* fs, bwtm_fslice_nodes_view* view)
{
  if(!fs || !view) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_step: null argument"); }
  ENTER(fs->ctx);
  if(!fs->node_sp[0]) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_step: call bwtm_fslice_nodes_begin first"); }
  const u64 N = fs->nodes;
  std::memset(view, 0, sizeof(*view));
  view->sp = fs->node_sp[1]; view->r = fs->node_r[1]; view->count = fs->node_cnt[1];
  if(N == 0) { return BWTM_OK; }
  if(N > fs->node_cap) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_step: %llu nodes, capacity %llu", (unsigned long long)N, (unsigned long long)fs->node_cap); }
  const u64 grid = div_up(N, BLOCK_THREADS);
  u64* flags = fs->node_flags.as<u64>();
  LAUNCH("range_step", k_range_step<false>, grid, BLOCK_THREADS, fs->a->view(), fs->b->view(), (const u64*)fs->node_sp[0], (const u64*)fs->node_r[0], (const u64*)fs->node_cnt[0], N,
    flags, (const u64*)nullptr, (u64*)nullptr, (u64*)nullptr, (u64*)nullptr, fs->ra->bits_as<u32>(), fs->node_pieces.as<RangePiece>(), fs->node_npieces.as<u32>(), fs->node_piece_cap);
  LAUNCH("range_emit", k_range_emit, 2048, BLOCK_THREADS, fs->node_pieces.as<const RangePiece>(), fs->node_npieces.as<const u32>(), fs->node_piece_cap, fs->ra->bits_as<u32>());
  HIP_TRY(hipMemsetAsync(fs->node_npieces.p, 0, sizeof(u32), CTX.stream));
  TRY(device_scan<0>(flags, flags, 5 * N + 1));
  TRY(fetch_u64(flags + 5 * N, 0));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  if(CTX.host_scratch[0] > fs->node_cap) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_step: %llu nodes have %llu children, capacity %llu", (unsigned long long)N, (unsigned long long)CTX.host_scratch[0], (unsigned long long)fs->node_cap); }
  LAUNCH("range_children", k_range_step<true>, grid, BLOCK_THREADS, fs->a->view(), fs->b->view(), (const u64*)fs->node_sp[0], (const u64*)fs->node_r[0], (const u64*)fs->node_cnt[0], N,
    (u64*)nullptr, (const u64*)flags, fs->node_sp[1], fs->node_r[1], fs->node_cnt[1], (u32*)nullptr, (RangePiece*)nullptr, (u32*)nullptr, 0u);
  // class c's children are [flags[(c - 1) N], flags[c N]): the six boundaries, then the cut points inside every class
  u64* cf = fs->node_class_first.as<u64>();
  for(u32 c = 0; c <= 5; c++) { HIP_TRY(hipMemcpyAsync(cf + c, flags + (u64)c * N, sizeof(u64), hipMemcpyDeviceToDevice, CTX.stream)); }
  HIP_TRY(hipMemsetAsync(fs->node_err.p, 0, sizeof(u32), CTX.stream));
  LAUNCH("cut_counts", k_node_cut_search, 1, BLOCK_THREADS, (const u64*)fs->node_sp[1], (const u64*)fs->node_cnt[1], (const u64*)cf, fs->cuts.as<const u64>(), fs->ncuts, fs->below.as<u64>(),
    fs->node_err.as<u32>());
  HIP_TRY(hipMemcpyAsync(fs->host_below, fs->below.p, 5ull * fs->ncuts * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream));
  TRY(fetch_u64(cf, 96, 6));
  HIP_TRY(hipMemcpyAsync(CTX.host_scratch + 104, fs->node_err.p, sizeof(u32), hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  if((u32)CTX.host_scratch[104] != 0) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_step: a node crosses a cut (cuts must be k-mer boundaries of the merged order)"); }
  for(u32 c = 0; c <= 5; c++) { view->class_first[c] = CTX.host_scratch[96 + c]; }
  for(u32 c = 0; c < 5; c++)
  {
    const u64 total = view->class_first[c + 1] - view->class_first[c];
    for(u32 k = 0; k <= BWTM_X_MAX_PARTS; k++) { view->below[c][k] = (k + 1 < fs->ncuts ? fs->host_below[c * fs->ncuts + k] : total); }
    view->below[c][0] = 0;
  }
  return BWTM_OK;
}

extern "C" int bwtm_fslice_nodes_gather(bwtm_fslice* fs, const bwtm_fslice_nodes_view* views, int parts, int part)
{
  if(!fs || !views || parts < 1 || part < 0 || part >= parts) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_gather: bad argument"); }
  ENTER(fs->ctx);
  if(fs->ncuts != (u32)parts + 1 || !fs->node_sp[0]) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_gather: cuts / node buffers were not set up for %d parts", parts); }
  u32 np = 0; u64 n = 0;
  for(u32 c = 0; c < 5; c++)
  {
    for(int h = 0; h < parts; h++)
    {
      const u64 lo_x = views[h].below[c][part], hi_x = views[h].below[c][part + 1];
      if(lo_x > hi_x || hi_x > views[h].class_first[c + 1] - views[h].class_first[c]) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_gather: counts of GPU %d, class %u are not monotone", h, c); }
      if(lo_x < hi_x)
      {
        NodePiece pc;
        pc.sp = (const u64*)views[h].sp; pc.r = (const u64*)views[h].r; pc.cnt = (const u64*)views[h].count;
        pc.src_first = views[h].class_first[c] + lo_x; pc.count = hi_x - lo_x; pc.dst_first = n;
        fs->host_node_pieces[np++] = pc;
        n += hi_x - lo_x;
      }
    }
  }
  if(n > fs->node_cap) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_gather: %llu nodes fall into this GPU's range, capacity %llu", (unsigned long long)n, (unsigned long long)fs->node_cap); }
  fs->nodes = n;
  if(n == 0) { return BWTM_OK; }
  HIP_TRY(hipMemcpyAsync(fs->node_gather_pieces.p, fs->host_node_pieces, (u64)np * sizeof(NodePiece), hipMemcpyHostToDevice, CTX.stream));
  LAUNCH("nodes_gather", k_gather_nodes, div_up(n, BLOCK_THREADS), BLOCK_THREADS, fs->node_gather_pieces.as<const NodePiece>(), np, n, fs->node_sp[0], fs->node_r[0], fs->node_cnt[0]);
  HIP_TRY(hipStreamSynchronize(CTX.stream));                       // the peers may overwrite their children once every GPU has returned from here
  return BWTM_OK;
}

extern "C" int bwtm_fslice_nodes_expand(bwtm_fslice* fs)
{
  if(!fs) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_expand: null argument"); }
  ENTER(fs->ctx);
  if(!fs->node_sp[0]) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_expand: call bwtm_fslice_nodes_begin first"); }
  const u64 N = fs->nodes;
  u64 alive = 0;
  DevBuf offsets;
  if(N > 0)
  {
    TRY(offsets.alloc((N + 1) * sizeof(u64)));
    HIP_TRY(hipMemcpyAsync(offsets.p, fs->node_cnt[0], N * sizeof(u64), hipMemcpyDeviceToDevice, CTX.stream));
    HIP_TRY(hipMemsetAsync(offsets.as<u64>() + N, 0, sizeof(u64), CTX.stream));
    TRY(device_scan<0>(offsets.as<u64>(), offsets.as<u64>(), N + 1));
    TRY(fetch_u64(offsets.as<u64>() + N, 0));
    HIP_TRY(hipStreamSynchronize(CTX.stream));
    alive = CTX.host_scratch[0];
  }
  if(alive > fs->cap) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_expand: the nodes stand for %llu sequences, capacity %llu", (unsigned long long)alive, (unsigned long long)fs->cap); }
  // the elements as the outputs of a step: contiguous, class 0 (the layout k_frontier_init produces for the roots)
  fs->nb_out = std::max<u64>(1, div_up(alive, (u64)FR_BLOCK));
  if(N > 0)
  {
    LAUNCH("range_expand", k_range_expand, div_up(N, BLOCK_THREADS), BLOCK_THREADS, (const u64*)fs->node_sp[0], (const u64*)fs->node_r[0], (const u64*)fs->node_cnt[0],
      offsets.as<const u64>(), N, fs->lo_out, fs->hi_out, fs->node_pieces.as<RangePiece>(), fs->node_npieces.as<u32>(), fs->node_piece_cap);
    LAUNCH("range_expand_pieces", k_range_expand_pieces, 2048, BLOCK_THREADS, fs->node_pieces.as<const RangePiece>(), fs->node_npieces.as<const u32>(), fs->node_piece_cap, fs->lo_out, fs->hi_out);
    HIP_TRY(hipMemsetAsync(fs->node_npieces.p, 0, sizeof(u32), CTX.stream));
  }
  LAUNCH("frontier_init", k_frontier_init_tables, div_up(5 * fs->nb_out + 1, BLOCK_THREADS), BLOCK_THREADS, fs->seg_len_out.as<u64>(), fs->seg_phys_out, fs->nb_out, alive);
  fs->nodes = 0;
  TRY(fslice_scan_outputs(fs));
  return BWTM_OK;
}

extern "C" int bwtm_x_index_upload_window(const uint8_t* data, uint64_t nbytes, uint64_t first_position, const uint64_t counts_before[6],
  uint64_t bases, uint64_t sequences, const uint64_t C[7], bwtm_index** out)
{
  ENTER(nullptr);
  if(!out || !data || nbytes == 0 || !counts_before || !C) { return fail(BWTM_EINVAL, "bwtm_x_index_upload_window: null argument"); }
  u64 before = 0; for(int c = 0; c < 6; c++) { before += counts_before[c]; }
  if(before != first_position || first_position > bases) { return fail(BWTM_EINVAL, "bwtm_x_index_upload_window: the counts before the bytes add up to %llu, their first position is %llu", (unsigned long long)before, (unsigned long long)first_position); }
  bwtm_index* x = new bwtm_index();
  x->ctx = t_ctx; x->nbytes = nbytes;
  auto body = [&]() -> int
  {
    TRY(alloc_native(x->data, nbytes));
    // the product's upload pipeline on the share: block lengths and group counts, their scans; the stream's verdict and totals come back
    int rc = upload_queue(x, data);
    if(rc == BWTM_OK) { rc = upload_scan(x, 0); }
    hipError_t e1 = hipStreamSynchronize(CTX.copy_stream), e2 = hipStreamSynchronize(CTX.stream);
    if(rc != BWTM_OK) { return rc; }
    if(e1 != hipSuccess || e2 != hipSuccess) { return fail(BWTM_ENODEV, "upload failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); }
    const u32 flags = (u32)CTX.host_scratch[6];
    x->flags.release();
    if(flags & 1u) { return fail(BWTM_EINVAL, "not a canonical run-length stream: a full 64-byte block encodes fewer than 64 positions"); }
    u64 held = 0; for(int c = 0; c < 6; c++) { held += CTX.host_scratch[c]; }
    const u64 end_position = first_position + held;
    if(end_position > bases) { return fail(BWTM_EINVAL, "bwtm_x_index_upload_window: the bytes decode to positions [%llu, %llu) of an index of %llu", (unsigned long long)first_position, (unsigned long long)end_position, (unsigned long long)bases); }
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
    if(q0 >= q_end) { return fail(BWTM_EINVAL, "bwtm_x_index_upload_window: the bytes hold no whole record"); }
    x->windowed = true; x->win_first = q0; x->win_count = q_end - q0;
    TRY(x->recs.alloc(x->win_count * 64));
    TRY(x->sup.alloc(x->nsup * SUP_STRIDE * sizeof(u64)));
    // super rows: k_build_sup gives a super that begins before the share the counts at the share's first position -- at or below the counts
    // of every record of the window, which is all a row has to be (header fields are offsets from it)
    LAUNCH("build_sup", k_build_sup, div_up(x->nsup * WAVE, BLOCK_THREADS), BLOCK_THREADS,
      x->native_bytes(), x->nbytes, x->blen.as<const u64>(), x->gcum.as<const u64>(), gstride, x->nblocks, x->ngroups, end_position, x->sup.as<u64>(), x->nsup);
    uint4* shifted = (uint4*)((char*)x->recs.p - (q0 << 6));
    const u64 per_group = held / x->ngroups;
    const bool long_runs = (x->nblocks > 0 && held / x->nblocks > 400);
#define BUILD_RECS_W(W, WAVES, FILL) LAUNCH("build_recs", (k_build_recs<W, WAVES, FILL>), div_up(x->ngroups, WAVES), WAVES * WAVE, \
    x->native_bytes(), x->nbytes, x->blen.as<const u64>(), x->block_start.as<u64>(), x->gcum.as<const u64>(), gstride, x->nblocks, x->ngroups, end_position, \
    x->sup.as<const u64>(), shifted, q_end)
    if(per_group <= 6500) { BUILD_RECS_W(8192, 4, false); }
    else if(per_group <= 14000) { BUILD_RECS_W(16384, 4, false); }
    else if(!long_runs) { BUILD_RECS_W(32768, 2, false); }
    else { BUILD_RECS_W(32768, 2, true); }
#undef BUILD_RECS_W
    HIP_TRY(hipStreamSynchronize(CTX.stream));
    x->blen.release(); x->block_start.release(); x->gcum.release(); x->data.release();          // a window keeps its records and super rows only
    x->has_native = false; x->nbytes = 0; x->nblocks = 0; x->ngroups = 0;
    // the record the share begins in and the one it ends in are incomplete unless they are the index's own first / last
    if(end_position < bases) { x->win_count -= 1; }
    if(x->win_count == 0) { return fail(BWTM_EINVAL, "bwtm_x_index_upload_window: the bytes hold no whole record"); }
    return BWTM_OK;
  };
  int rc = body();
  if(rc != BWTM_OK) { (void)hipStreamSynchronize(CTX.stream); delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

extern "C" int bwtm_x_ra_create_range(const bwtm_index* a, const bwtm_index* b, uint64_t pos_first, uint64_t pos_last, bwtm_ra** out)
{
  if(!a || !b || !out || pos_first > pos_last) { return fail(BWTM_EINVAL, "bwtm_x_ra_create_range: bad argument"); }
  if(a->ctx != b->ctx) { return fail(BWTM_EINVAL, "bwtm_x_ra_create_range: the two indexes live in different contexts"); }
  ENTER(a->ctx);
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

extern "C" int bwtm_x_ra_or_range(bwtm_ra* dst, const bwtm_ra* src, uint64_t pos_first, uint64_t pos_last)
{
  if(!dst || !src || pos_first > pos_last) { return fail(BWTM_EINVAL, "bwtm_x_ra_or_range: bad argument"); }
  ENTER(dst->ctx);
  if(dst->n_out != src->n_out) { return fail(BWTM_EINVAL, "bwtm_x_ra_or_range: rank arrays of different shapes"); }
  if(dst->finalized) { return fail(BWTM_EINVAL, "bwtm_x_ra_or_range: rank array already finalized"); }
  if(pos_first == pos_last) { return BWTM_OK; }
  const u64 nwords = dst->nchunks * CHUNK_WORDS;
  const u64 w0 = pos_first >> 6, w1 = std::min<u64>(nwords, div_up(pos_last, 64));
  auto holds = [&](const bwtm_ra* r) { return !r->windowed || (w0 >= r->win_word_first && w1 <= r->win_word_first + r->win_words); };
  if(!holds(dst) || !holds(src)) { return fail(BWTM_EINVAL, "bwtm_x_ra_or_range: the positions [%llu, %llu) reach outside a rank array's window", (unsigned long long)pos_first, (unsigned long long)pos_last); }
  // the source lives in another context of this device or on a peer: its stream must have finished writing (the caller's barrier)
  LAUNCH("bits_or", k_bits_or, div_up(w1 - w0, BLOCK_THREADS), BLOCK_THREADS, dst->bits_as<u64>() + w0, src->bits_as<const u64>() + w0, w1 - w0);
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" uint64_t bwtm_x_ra_bytes(const bwtm_ra* ra)
{
  if(!ra) { return 0; }
  return (ra->windowed ? ra->win_words : ra->nchunks * CHUNK_WORDS) * sizeof(u64);
}

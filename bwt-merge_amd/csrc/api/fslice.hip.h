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

  The same object also carries the state of the search over PARTITIONED records (fixed cuts, routed node phase): api/partition.hip.h.
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
  uint2* recv_lo = nullptr; unsigned short* recv_hi = nullptr;       // (hipMalloc, so that a collective of another library may write them) where an exchange between processes delivers the next input
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
  f.bound_row = fs->bound.as<u32>() + fs->in_epoch * (fs->ntiles + 1); f.step = fs->in_epoch; f.block_base = 0; f.src_lo = nullptr; f.src_hi = nullptr; f.nseg_in = 0;
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

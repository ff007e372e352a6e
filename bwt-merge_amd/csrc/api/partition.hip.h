/*
  api/partition.hip.h -- the merge over PARTITIONED records (only with -DBWTM_EXPERIMENTAL, include/bwtm_experimental.h).
  Part of bwtm_api.hip, after api/fslice.hip.h whose per-GPU state (bwtm_fslice) it extends.

    windows          bwtm_x_index_window (cut from whole records), bwtm_x_index_upload_window (transcoded from a share of the native bytes),
                     bwtm_x_ra_create_range / bwtm_x_ra_or_range (a part's range of the bitvector): handles whose device arrays hold a range only
                     and are addressed by absolute record / word numbers through a base pointer shifted back by the range's start
    fixed cuts       bwtm_fslice_set_cuts, bwtm_fslice_gather_cut: elements routed to the GPU that owns their position
                     (kernels/search_partition.hip.h: k_compact_outputs, k_cut_search, k_gather_dense)
    node phase       bwtm_fslice_nodes_*: the first levels on trie nodes, routed the same way (k_node_cut_search, k_gather_nodes)
  The second half of a partitioned merge needs no entry point of its own: bwtm_ra_range_counts / _finalize_range / bwtm_interleave_range /
  bwtm_slice_* take the window handles (api/slices.hip.h, k_interleave_sup_window).
  Driver and tests: bwt-merge_amd/experimental.py (partition_cuts, search_partitioned), tests/experimental/test_gpu_partitioned.py,
  tools/partitioned_scale.py.
*/
#pragma once

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

extern "C" int bwtm_fslice_input_buffers(bwtm_fslice* fs, void** lo, void** hi, uint64_t* capacity)
{
  if(!fs || !lo || !hi || !capacity) { return fail(BWTM_EINVAL, "bwtm_fslice_input_buffers: null argument"); }
  ENTER(fs->ctx);
  if(!fs->recv_lo)
  {
    // plain hipMalloc blocks like everything a peer or a collective touches (the pool's large blocks are mapped for their own device only)
    const u64 fcap = fs->nbl * FR_BLOCK;
    TRY(fslice_export_alloc(fs, fs->recv_lo, fcap));
    if(fs->wide) { TRY(fslice_export_alloc(fs, fs->recv_hi, fcap)); }
  }
  *lo = fs->recv_lo; *hi = fs->recv_hi; *capacity = fs->cap;
  return BWTM_OK;
}

extern "C" int bwtm_fslice_set_input(bwtm_fslice* fs, uint64_t count)
{
  if(!fs) { return fail(BWTM_EINVAL, "bwtm_fslice_set_input: null argument"); }
  ENTER(fs->ctx);
  if(!fs->recv_lo) { return fail(BWTM_EINVAL, "bwtm_fslice_set_input: call bwtm_fslice_input_buffers first"); }
  if(count > fs->cap) { return fail(BWTM_EINVAL, "bwtm_fslice_set_input: %llu elements, capacity %llu", (unsigned long long)count, (unsigned long long)fs->cap); }
  // the caller has waited for its exchange: the elements are in the receive buffers (a production version would step on them in place)
  if(count > 0)
  {
    HIP_TRY(hipMemcpyAsync(fs->lo_in.p, fs->recv_lo, count * sizeof(uint2), hipMemcpyDeviceToDevice, CTX.stream));
    if(fs->wide) { HIP_TRY(hipMemcpyAsync(fs->hi_in.p, fs->recv_hi, count * sizeof(unsigned short), hipMemcpyDeviceToDevice, CTX.stream)); }
  }
  fs->n_in = count;
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

extern "C" uint64_t bwtm_x_index_record_bytes(const bwtm_index* x) { return bwtm_index_record_bytes(x); }

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

extern "C" int bwtm_fslice_nodes_input_buffers(bwtm_fslice* fs, void** sp, void** r, void** count, uint64_t* capacity)
{
  if(!fs || !sp || !r || !count || !capacity) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_input_buffers: null argument"); }
  ENTER(fs->ctx);
  if(!fs->node_sp[0]) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_input_buffers: call bwtm_fslice_nodes_begin first"); }
  HIP_TRY(hipStreamSynchronize(CTX.stream));                       // this level's step has read them
  *sp = fs->node_sp[0]; *r = fs->node_r[0]; *count = fs->node_cnt[0]; *capacity = fs->node_cap;      // plain hipMalloc blocks
  return BWTM_OK;
}

extern "C" int bwtm_fslice_nodes_set_input(bwtm_fslice* fs, uint64_t nodes)
{
  if(!fs) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_set_input: null argument"); }
  if(!fs->node_sp[0] || nodes > fs->node_cap) { return fail(BWTM_EINVAL, "bwtm_fslice_nodes_set_input: %llu nodes, capacity %llu", (unsigned long long)nodes, (unsigned long long)fs->node_cap); }
  fs->nodes = nodes;
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
  return index_upload_window(data, nbytes, first_position, counts_before, bases, sequences, C, false, out);     // product code since round 6 (api/pmerge.hip.h)
}

extern "C" int bwtm_x_ra_create_range(const bwtm_index* a, const bwtm_index* b, uint64_t pos_first, uint64_t pos_last, bwtm_ra** out)
{
  return bwtm_ra_create_range(a, b, pos_first, pos_last, out);                                                    // product code since round 6
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

namespace
{
// the words [w0, w1) of a rank array that the handle holds: [h0, h1)
void ra_held_words(const bwtm_ra* ra, u64 w0, u64 w1, u64& h0, u64& h1)
{
  const u64 lo = (ra->windowed ? ra->win_word_first : 0), hi = (ra->windowed ? ra->win_word_first + ra->win_words : ra->nchunks * CHUNK_WORDS);
  h0 = std::max(w0, lo); h1 = std::min(w1, hi);
  if(h0 > h1) { h0 = h1 = w0; }
}
} // namespace

extern "C" int bwtm_x_ra_read_words(const bwtm_ra* ra, uint64_t pos_first, uint64_t pos_last, void* device_out)
{
  if(!ra || !device_out || pos_first > pos_last) { return fail(BWTM_EINVAL, "bwtm_x_ra_read_words: bad argument"); }
  ENTER(ra->ctx);
  const u64 w0 = pos_first >> 6, w1 = div_up(pos_last, 64);
  if(w1 == w0) { return BWTM_OK; }
  u64 h0, h1; ra_held_words(ra, w0, w1, h0, h1);
  HIP_TRY(hipMemsetAsync(device_out, 0, (w1 - w0) * sizeof(u64), CTX.stream));
  if(h1 > h0) { HIP_TRY(hipMemcpyAsync((u64*)device_out + (h0 - w0), ra->bits_as<const u64>() + h0, (h1 - h0) * sizeof(u64), hipMemcpyDeviceToDevice, CTX.stream)); }
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" int bwtm_x_ra_or_words(bwtm_ra* ra, uint64_t pos_first, uint64_t pos_last, const void* device_in)
{
  if(!ra || !device_in || pos_first > pos_last) { return fail(BWTM_EINVAL, "bwtm_x_ra_or_words: bad argument"); }
  ENTER(ra->ctx);
  if(ra->finalized) { return fail(BWTM_EINVAL, "bwtm_x_ra_or_words: rank array already finalized"); }
  const u64 w0 = pos_first >> 6, w1 = div_up(pos_last, 64);
  u64 h0, h1; ra_held_words(ra, w0, w1, h0, h1);
  if(h1 > h0) { LAUNCH("bits_or", k_bits_or, div_up(h1 - h0, BLOCK_THREADS), BLOCK_THREADS, ra->bits_as<u64>() + h0, (const u64*)device_in + (h0 - w0), h1 - h0); }
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" uint64_t bwtm_x_ra_bytes(const bwtm_ra* ra) { return bwtm_ra_bytes(ra); }

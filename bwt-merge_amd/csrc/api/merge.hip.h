/*
  api/merge.hip.h -- interleave (BWT::BWT(a, b, ra), bwt.cpp:286-314) and the whole path
  (FMI::FMI(a, b, parameters), fmi.cpp:336-369): device-resident, consuming, and host-to-host forms.
  Part of bwtm_api.hip.
*/
#pragma once

// An input whose native bytes are on their way to the device (queued on the copy stream behind the uploads of the merge that
// was running when it was announced): what bwtm_merge_host_pipelined hands back for the NEXT merge of a chain.
struct bwtm_upload
{
  bwtm_index* x = nullptr;            // native buffer + sample arrays allocated, nothing decoded yet
  UploadEvents events;                // one per chunk
  bwtm_host_input host;               // the caller's descriptor (validated when the upload is consumed)
  u64 C_copy[8] = {};                 // host.C points here when the caller gave an alphabet (the caller's array may be gone by then)
  void set(const bwtm_host_input& in)
  {
    host = in;
    if(in.C) { for(int c = 0; c <= 6; c++) { C_copy[c] = in.C[c]; } host.C = C_copy; }
  }
  bwtm_upload() {}
  bwtm_upload(const bwtm_upload&) = delete; bwtm_upload& operator=(const bwtm_upload&) = delete;
  ~bwtm_upload() { delete x; }
};

namespace
{

// The records [q_lo, q_hi) of the chunks [c0, c1) of the output (recs_out is indexed by the global record number): the counts at the
// chunk starts first (k_interleave_base), then one workgroup per chunk.
int interleave_chunks(const bwtm_index* a, const bwtm_index* b, bwtm_ra* ra, u64 c0, u64 c1, u64 q_lo, u64 q_hi, const u64* sup_out, uint4* recs_out)
{
  if(c1 <= c0) { return BWTM_OK; }
  const u64 count = c1 - c0;
  // one workgroup of 256 threads per chunk, fewer than 2^32 threads per grid: beyond 2^23 chunks (69 * 10^9 positions) in halves
  if(count > (1ull << 23))
  {
    const u64 mid = c0 + count / 2;
    TRY(interleave_chunks(a, b, ra, c0, mid, q_lo, q_hi, sup_out, recs_out));
    return interleave_chunks(a, b, ra, mid, c1, q_lo, q_hi, sup_out, recs_out);
  }
  DevBuf base_rel; TRY(base_rel.alloc(5 * count * sizeof(u32)));
  LAUNCH("interleave_base", k_interleave_base, div_up(count, BLOCK_THREADS), BLOCK_THREADS, a->view(), b->view(), ra->chunk_base.as<const u64>(), c0, c1,
    sup_out, base_rel.as<u32>(), count);
  LAUNCH("interleave", k_interleave, count, BLOCK_THREADS, a->view(), b->view(),
    ra->bits_as<const u64>(), ra->chunk_base.as<const u64>(), c0, c1, q_lo, q_hi, base_rel.as<const u32>(), count, recs_out);
  return BWTM_OK;                                                    // base_rel returns to the pool in stream order
}

int interleave_impl(const bwtm_index* a, const bwtm_index* b, bwtm_ra* ra, bwtm_index* x)
{
  x->n = ra->n_out; x->m = a->m + b->m;                           // bwt.cpp:305-306
  for(int c = 0; c < 8; c++) { x->C[c] = a->C[c] + b->C[c]; }     // fmi.cpp:367-368
  x->nrecs = ra->nrecs_out; x->nsup = num_supers(x->n);
  TRY(x->recs.alloc(x->nrecs * 64));
  TRY(x->sup.alloc(x->nsup * SUP_STRIDE * sizeof(u64)));
  LAUNCH("interleave_sup", k_interleave_sup, div_up(x->nsup, BLOCK_THREADS), BLOCK_THREADS, a->view(), b->view(),
    ra->bits_as<const u64>(), ra->chunk_base.as<const u64>(), x->n, x->sup.as<u64>(), x->nsup, (const u64*)nullptr);
  TRY(interleave_chunks(a, b, ra, 0, ra->nchunks, 0, x->nrecs, x->sup.as<const u64>(), x->recs.as<uint4>()));
  return BWTM_OK;
}

int check_interleave_args(const bwtm_index* a, const bwtm_index* b, const bwtm_ra* ra, bool allow_ranged = false)
{
  if(!allow_ranged) { WHOLE_INDEX(a, "bwtm_interleave"); WHOLE_INDEX(b, "bwtm_interleave"); }
  if((a->windowed || b->windowed) && !ra->ranged) { return fail(BWTM_EINVAL, "bwtm_interleave_range: windows of indexes need a rank array finalized for the range (bwtm_ra_finalize_range)"); }
  if(!ra->finalized) { return fail(BWTM_EINVAL, "bwtm_interleave: rank array not finalized"); }
  if(ra->ranged && !allow_ranged) { return fail(BWTM_EINVAL, "bwtm_interleave: the rank array was finalized for an output range (bwtm_interleave_range takes it)"); }
  if(ra->na != a->n || ra->nb != b->n) { return fail(BWTM_EINVAL, "bwtm_interleave: rank array was created for other inputs"); }
  if(ra->values != b->n) { return fail(BWTM_EINVAL, "bwtm_interleave: rank array holds %llu values, expected %llu", (unsigned long long)ra->values, (unsigned long long)b->n); }
  return BWTM_OK;
}

// A rank array for a and b inside the current scope (no nested entry point).
int ra_make(const bwtm_index* a, const bwtm_index* b, bwtm_ra** out)
{
  bwtm_ra* ra = new bwtm_ra();
  ra->ctx = t_ctx;
  ra->na = a->n; ra->nb = b->n; ra->n_out = a->n + b->n;
  ra->nrecs_out = num_records(ra->n_out);
  ra->nchunks = div_up(ra->nrecs_out, 64);
  int rc = ra->owned_bits.alloc(ra->nchunks * CHUNK_WORDS * sizeof(u64), true);
  ra->bits_ptr = ra->owned_bits.p;
  if(rc == BWTM_OK) { rc = ra->chunk_base.alloc((ra->nchunks + 1) * sizeof(u64), true); }
  if(rc != BWTM_OK) { delete ra; return rc; }
  *out = ra;
  return BWTM_OK;
}

// search + finalize + interleave; with `consume` the inputs are deleted as soon as the interleave is queued (their
// buffers return to the pool in stream order, so the encoder that follows can reuse the memory).
int merge_records(bwtm_index*& a, bwtm_index*& b, bool consume, bwtm_index** out)
{
  bwtm_ra* ra = nullptr;
  int rc = ra_make(a, b, &ra);
  if(rc != BWTM_OK)
  {
    if(consume) { delete a; delete b; a = nullptr; b = nullptr; }   // "destroying them" holds on every exit path
    return rc;
  }
  if(b->m > 0) { rc = bwtm_search(a, b, 0, b->m - 1, ra); }
  if(rc == BWTM_OK) { rc = ra_finalize(ra); }
  if(rc == BWTM_OK) { rc = check_interleave_args(a, b, ra); }
  bwtm_index* x = nullptr;
  if(rc == BWTM_OK)
  {
    x = new bwtm_index(); x->ctx = t_ctx;
    rc = interleave_impl(a, b, ra, x);
  }
  delete ra;
  if(consume) { delete a; delete b; a = nullptr; b = nullptr; }
  if(rc != BWTM_OK) { delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// b comes from the host (b_host) or is an upload announced during the previous merge (b_pending, consumed).  `next` (optional)
// announces the input of the merge after this one: its copies are queued on the copy stream right behind this merge's own, so
// they run under this merge's search, and the pending upload is handed back in *next_out.
int merge_host_impl(bwtm_index* a_dev, const bwtm_host_input* a_host, const bwtm_host_input* b_host, bwtm_upload* b_pending,
  const bwtm_host_input* next, bwtm_upload** next_out, bwtm_alloc_fn alloc, void* user, int want_samples, bwtm_host_output* out, bwtm_index** keep)
{
  const double t0 = now_ms();
  bwtm_index* a = a_dev;
  bwtm_index* b = nullptr;
  bwtm_index* x = nullptr;
  bwtm_upload* pending_next = nullptr;
  bwtm_host_input b_desc;
  if(b_pending) { b = b_pending->x; b_pending->x = nullptr; b_desc = b_pending->host; b_host = &b_desc; }
  else { b = new bwtm_index(); }
  auto body = [&]() -> int
  {
    // The copies of both inputs are queued first (copy stream: b's chunks, then a's).  b's decode pass and scans run on the compute
    // stream while b's later chunks arrive; the host then waits for b's scan results only -- a's bytes are still on the link --
    // VALIDATES b's header against the stream, and only then queues b's transcode, which sizes its output from the header
    // (a wrong `bases` or a non-canonical stream must never reach k_build_recs).  a follows the same way.
    UploadEvents ev_a, ev_b_own;
    UploadEvents& ev_b = (b_pending ? b_pending->events : ev_b_own);
    if(!b_pending)
    {
      b->ctx = t_ctx; b->nbytes = b_host->nbytes; b->n = b_host->bases; b->m = b_host->sequences;
      TRY(alloc_native(b->data, b_host->nbytes));
      TRY(upload_prepare(b));
    }
    else if(b->ctx != t_ctx) { return fail(BWTM_EINVAL, "bwtm_merge_host_pipelined: the pending upload lives in another context"); }
    if(next)
    {
      pending_next = new bwtm_upload();
      pending_next->set(*next);
      bwtm_index* nx = new bwtm_index();
      pending_next->x = nx;
      nx->ctx = t_ctx; nx->nbytes = next->nbytes; nx->n = next->bases; nx->m = next->sequences;
      TRY(alloc_native(nx->data, next->nbytes));
      TRY(upload_prepare(nx));
    }
    if(a_host)
    {
      a = new bwtm_index();
      a->ctx = t_ctx; a->nbytes = a_host->nbytes; a->n = a_host->bases; a->m = a_host->sequences;
      TRY(alloc_native(a->data, a_host->nbytes));
      TRY(upload_prepare(a));
    }
    TRY(fork_copy_stream());                                        // recycled blocks may have queued users on the compute stream
    if(!b_pending) { TRY(upload_copies(b, (b_host->nbytes > 0 ? b_host->data : (const u8*)""), ev_b)); }
    if(a_host) { TRY(upload_copies(a, (a_host->nbytes > 0 ? a_host->data : (const u8*)""), ev_a)); }
    if(next) { TRY(upload_copies(pending_next->x, (next->nbytes > 0 ? next->data : (const u8*)""), pending_next->events)); }   // under this merge's search
    TRY(upload_decode(b, &ev_b));
    TRY(upload_scan(b, 8));
    hipEvent_t b_scanned = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&b_scanned, hipEventDisableTiming));
    hipError_t e = hipEventRecord(b_scanned, CTX.stream);
    if(e == hipSuccess) { e = hipEventSynchronize(b_scanned); }
    (void)hipEventDestroy(b_scanned);
    if(e != hipSuccess) { return fail(BWTM_ENODEV, "upload failed: %s", hipGetErrorString(e)); }
    TRY(upload_validate(b, b_host->sequences, b_host->bases, b_host->C, 8));
    TRY(transcode(b));
    if(a_host)
    {
      TRY(upload_decode(a, &ev_a));
      TRY(upload_scan(a, 16));
    }
    // (the compute stream has waited for every chunk of a and b; the copy stream itself may still be busy with `next`)
    HIP_TRY(hipStreamSynchronize(CTX.stream));
    if(a_host)
    {
      TRY(upload_validate(a, a_host->sequences, a_host->bases, a_host->C, 16));
      TRY(transcode(a));
    }
    if(a->ctx != t_ctx) { return fail(BWTM_EINVAL, "bwtm_merge_host_chained: the index lives in another context"); }
    // the native bytes of the inputs are not needed any more (BlockArray::clearUntil, bwt.cpp:224-225)
    TRY(bwtm_index_drop_native(a)); TRY(bwtm_index_drop_native(b));
    const double t1 = now_ms();
    out->ms_upload = t1 - t0;

    TRY(merge_records(a, b, true, &x));                           // a and b are gone after this
    const double t2 = now_ms();
    out->ms_search = t2 - t1;

    out->sequences = x->m; out->bases = x->n;
    for(int c = 0; c <= 6; c++) { out->C[c] = x->C[c]; }
    out->data = nullptr; out->nbytes = 0; out->blocks = 0; out->block_end = nullptr; out->cum = nullptr;
    out->sample_width = 0; out->fields = nullptr; out->anchors = nullptr;
    if(want_samples == BWTM_RESULT_ON_DEVICE)
    {
      // an intermediate result of a chain: the next merge's first input, never encoded or downloaded (bwt_merge.cpp:167-173)
      HIP_TRY(hipStreamSynchronize(CTX.stream));
      out->ms_interleave = now_ms() - t2; out->ms_encode_download = 0; out->ms_samples = 0; out->ms_total = now_ms() - t0;
      return BWTM_OK;
    }
    std::unique_ptr<EncodePlan> plan_holder(new EncodePlan());
    EncodePlan& plan = *plan_holder;
    if(x->n > 0) { TRY(encode_size(x, plan)); }
    const double t3 = now_ms();
    out->ms_interleave = t3 - t2;
    const u64 total = plan.total;
    out->nbytes = total; out->blocks = div_up(total, RLE_BLOCK);
    out->data = (u8*)alloc(user, BWTM_BUF_DATA, total);
    if(!out->data && total > 0) { return fail(BWTM_ENOMEM, "bwtm_merge_host: the caller's allocator returned no buffer for %llu bytes", (unsigned long long)total); }
    if(x->n > 0) { TRY(encode_emit(x, plan, out->data)); }
    else { TRY(encode_blocking(x)); }
    plan_holder.reset();                                           // the size tables return to the pool
    int width = 8;
    if(want_samples == BWTM_SAMPLES_COMPACT)
    {
      // after the data has left: queued behind the encoder it cost ~10 ms of the download (its small result copy and the pinned
      // allocations of the caller share the link with the 128-MiB chunks); on its own it takes 0.2 ms
      HIP_TRY(hipStreamSynchronize(CTX.copy_stream));
      TRY(samples_width(x, &width));
    }
    if(want_samples && width == 8)
    {
      out->block_end = (u64*)alloc(user, BWTM_BUF_BLOCK_END, out->blocks * sizeof(u64));
      out->cum = (u64*)alloc(user, BWTM_BUF_CUM, 6 * (out->blocks + 1) * sizeof(u64));
      if((!out->block_end && out->blocks > 0) || !out->cum) { return fail(BWTM_ENOMEM, "bwtm_merge_host: the caller's allocator returned no buffer for the samples"); }
    }
    else if(want_samples)
    {
      out->fields = alloc(user, BWTM_BUF_FIELDS, 6 * out->blocks * (u64)width);
      out->anchors = (u64*)alloc(user, BWTM_BUF_ANCHORS, 6 * div_up(out->blocks, 64) * sizeof(u64));
      if(out->blocks > 0 && (!out->fields || !out->anchors)) { return fail(BWTM_ENOMEM, "bwtm_merge_host: the caller's allocator returned no buffer for the samples"); }
    }
    if(want_samples) { out->sample_width = width; }
    HIP_TRY(hipStreamSynchronize(CTX.copy_stream));
    HIP_TRY(hipStreamSynchronize(CTX.stream));
    const double t4 = now_ms();
    out->ms_encode_download = t4 - t3;
    if(want_samples && width == 8) { TRY(download_samples(x, out->block_end, out->cum)); }
    else if(want_samples && width == 1) { TRY(download_samples_compact<u8>(x, (u8*)out->fields, out->anchors)); }
    else if(want_samples && width == 2) { TRY(download_samples_compact<unsigned short>(x, (unsigned short*)out->fields, out->anchors)); }
    else if(want_samples) { TRY(download_samples_compact<u32>(x, (u32*)out->fields, out->anchors)); }
    const double t5 = now_ms();
    out->ms_samples = t5 - t4;
    out->ms_total = t5 - t0;
    return BWTM_OK;
  };
  int rc = body();
  delete b_pending;                                                 // consumed (its index was taken over above)
  if(rc != BWTM_OK)
  {
    (void)hipStreamSynchronize(CTX.copy_stream); (void)hipStreamSynchronize(CTX.stream);   // nothing may touch the caller's buffers after the return
    delete a; delete b; delete x; delete pending_next;
    return rc;
  }
  if(next_out) { *next_out = pending_next; }
  if(keep) { rc = bwtm_index_drop_native(x); *keep = x; }
  else { delete x; }
  return rc;
}

} // namespace

//------------------------------------------------------------------------------
// C ABI.

extern "C" int bwtm_interleave(const bwtm_index* a, const bwtm_index* b, bwtm_ra* ra, bwtm_index** out)
{
  if(!a || !b || !ra || !out) { return fail(BWTM_EINVAL, "bwtm_interleave: null argument"); }
  if(a->ctx != ra->ctx || b->ctx != ra->ctx) { return fail(BWTM_EINVAL, "bwtm_interleave: handles of different contexts"); }
  ENTER(ra->ctx);
  TRY(check_interleave_args(a, b, ra));
  bwtm_index* x = new bwtm_index();
  x->ctx = t_ctx;
  int rc = interleave_impl(a, b, ra, x);
  if(rc != BWTM_OK) { delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

extern "C" int bwtm_merge(const bwtm_index* a, const bwtm_index* b, bwtm_index** out)
{
  if(!a || !b || !out) { return fail(BWTM_EINVAL, "bwtm_merge: null argument"); }
  if(a->ctx != b->ctx) { return fail(BWTM_EINVAL, "bwtm_merge: the two indexes live in different contexts"); }
  WHOLE_INDEX(a, "bwtm_merge"); WHOLE_INDEX(b, "bwtm_merge");
  ENTER(a->ctx);
  bwtm_index* aa = const_cast<bwtm_index*>(a); bwtm_index* bb = const_cast<bwtm_index*>(b);
  bwtm_index* x = nullptr;
  TRY(merge_records(aa, bb, false, &x));
  int rc = encode_blocking(x);
  if(rc != BWTM_OK) { delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

extern "C" int bwtm_merge_consume(bwtm_index* a, bwtm_index* b, bwtm_index** out)
{
  if(!a || !b || !out) { bwtm_index_free(a); bwtm_index_free(b); return fail(BWTM_EINVAL, "bwtm_merge_consume: null argument"); }
  if(a->ctx != b->ctx) { bwtm_index_free(a); bwtm_index_free(b); return fail(BWTM_EINVAL, "bwtm_merge_consume: the two indexes live in different contexts"); }
  Scope scope_(a->ctx);
  if(scope_.rc != BWTM_OK) { bwtm_index_free(a); bwtm_index_free(b); return scope_.rc; }      // the inputs are freed also when the call fails
  int rc = bwtm_index_drop_native(a);
  if(rc == BWTM_OK) { rc = bwtm_index_drop_native(b); }
  bwtm_index* x = nullptr;
  if(rc == BWTM_OK) { rc = merge_records(a, b, true, &x); }
  delete a; delete b;                                              // no-ops when merge_records consumed them
  if(rc == BWTM_OK) { rc = encode_blocking(x); }
  if(rc != BWTM_OK) { delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

extern "C" int bwtm_merge_host(const bwtm_host_input* a, const bwtm_host_input* b, bwtm_alloc_fn alloc, void* user,
  int want_samples, bwtm_host_output* out, bwtm_index** keep)
{
  if(!a || !b || !alloc || !out || (a->nbytes > 0 && !a->data) || (b->nbytes > 0 && !b->data)) { return fail(BWTM_EINVAL, "bwtm_merge_host: null argument"); }
  ENTER(nullptr);
  return merge_host_impl(nullptr, a, b, nullptr, nullptr, nullptr, alloc, user, want_samples, out, keep);
}

extern "C" int bwtm_merge_host_chained(bwtm_index* a, const bwtm_host_input* b, bwtm_alloc_fn alloc, void* user,
  int want_samples, bwtm_host_output* out, bwtm_index** keep)
{
  if(!a || !b || !alloc || !out || (b->nbytes > 0 && !b->data)) { bwtm_index_free(a); return fail(BWTM_EINVAL, "bwtm_merge_host_chained: null argument"); }
  ENTER(a->ctx);
  return merge_host_impl(a, nullptr, b, nullptr, nullptr, nullptr, alloc, user, want_samples, out, keep);
}

extern "C" int bwtm_merge_host_pipelined(bwtm_index* a_device, const bwtm_host_input* a_host, const bwtm_host_input* b_host, bwtm_upload* b_pending,
  const bwtm_host_input* next, bwtm_upload** next_pending, bwtm_alloc_fn alloc, void* user, int want_samples, bwtm_host_output* out, bwtm_index** keep)
{
  const bool bad = (!alloc || !out || ((a_device != nullptr) == (a_host != nullptr)) || ((b_host != nullptr) == (b_pending != nullptr)) || (next && !next_pending) ||
    (a_host && a_host->nbytes > 0 && !a_host->data) || (b_host && b_host->nbytes > 0 && !b_host->data) || (next && next->nbytes > 0 && !next->data));
  if(bad || (want_samples == BWTM_RESULT_ON_DEVICE && !keep))
  {
    bwtm_index_free(a_device); bwtm_upload_free(b_pending);
    return fail(BWTM_EINVAL, "bwtm_merge_host_pipelined: exactly one form of each input is required (and `keep` for a result that stays on the device)");
  }
  if(next_pending) { *next_pending = nullptr; }
  Scope scope_(a_device ? a_device->ctx : (b_pending && b_pending->x ? b_pending->x->ctx : nullptr));
  if(scope_.rc != BWTM_OK) { bwtm_index_free(a_device); bwtm_upload_free(b_pending); return scope_.rc; }      // consumed on every exit path
  return merge_host_impl(a_device, a_host, b_host, b_pending, next, next_pending, alloc, user, want_samples, out, keep);
}

extern "C" int bwtm_upload_begin(const bwtm_host_input* in, bwtm_upload** out)
{
  if(!in || !out || (in->nbytes > 0 && !in->data)) { return fail(BWTM_EINVAL, "bwtm_upload_begin: null argument"); }
  ENTER(nullptr);
  bwtm_upload* u = new bwtm_upload();
  u->set(*in);
  bwtm_index* x = new bwtm_index();
  u->x = x;
  x->ctx = t_ctx; x->nbytes = in->nbytes; x->n = in->bases; x->m = in->sequences;
  int rc = alloc_native(x->data, in->nbytes);
  if(rc == BWTM_OK) { rc = upload_prepare(x); }
  if(rc == BWTM_OK) { rc = fork_copy_stream(); }
  if(rc == BWTM_OK) { rc = upload_copies(x, (in->nbytes > 0 ? in->data : (const u8*)""), u->events); }
  if(rc != BWTM_OK) { (void)hipStreamSynchronize(CTX.copy_stream); delete u; return rc; }
  *out = u;
  return BWTM_OK;
}

extern "C" int bwtm_upload_finish(bwtm_upload* upload, bwtm_index** out)
{
  if(!upload || !out || !upload->x) { bwtm_upload_free(upload); return fail(BWTM_EINVAL, "bwtm_upload_finish: null argument"); }
  ENTER(upload->x->ctx);
  bwtm_index* x = upload->x;
  int rc = upload_decode(x, &upload->events);
  if(rc == BWTM_OK) { rc = upload_scan(x, 0); }
  hipError_t e = hipStreamSynchronize(CTX.stream);                  // has waited for every chunk: the caller's buffer is free again
  if(rc == BWTM_OK && e != hipSuccess) { rc = fail(BWTM_ENODEV, "upload failed: %s", hipGetErrorString(e)); }
  if(rc != BWTM_OK) { (void)hipStreamSynchronize(CTX.copy_stream); }
  if(rc == BWTM_OK) { rc = upload_validate(x, upload->host.sequences, upload->host.bases, upload->host.C, 0); }
  if(rc == BWTM_OK) { rc = transcode(x); }
  if(rc == BWTM_OK) { upload->x = nullptr; *out = x; }
  delete upload;
  return rc;
}

extern "C" void bwtm_upload_free(bwtm_upload* upload)
{
  if(!upload) { return; }
  Scope scope(upload->x ? upload->x->ctx : nullptr);
  // queued copies still read the caller's buffer: drain the copy stream, or the whole device when the context cannot be entered
  if(scope.rc == BWTM_OK) { (void)hipStreamSynchronize(CTX.copy_stream); }
  else { (void)hipDeviceSynchronize(); }
  delete upload;
}

/*
  api/index.hip.h -- the device index: BWT::load + BWT::build (pipelined upload, transcode), the canonical
  encoder with the pipelined download, samples, batch queries.  Part of bwtm_api.hip.
*/
#pragma once

//------------------------------------------------------------------------------
// Handle.

struct bwtm_index
{
  bwtm_context* ctx = nullptr;
  bwtm_index() : ctx(t_ctx) { if(ctx) { ctx->live_handles++; } }     // handles are created inside a Scope: t_ctx is their context
  ~bwtm_index() { if(ctx) { ctx->live_handles--; } }
  bwtm_index(const bwtm_index&) = delete; bwtm_index& operator=(const bwtm_index&) = delete;
  u64 n = 0, m = 0;
  u64 C[8] = {};
  DevBuf recs; u64 nrecs = 0;        // device rank structure
  DevBuf sup;  u64 nsup = 0;
  // Native form (present after upload or encode):
  bool has_native = false;
  DevBuf data; u64 nbytes = 0; u64 nblocks = 0;
  const void* borrowed = nullptr;     // caller-owned native bytes (bwtm_index_from_device_borrowed) instead of `data`
  const u8* native_bytes() const { return borrowed ? (const u8*)borrowed : data.as<const u8>(); }
  DevBuf block_start;                 // nblocks + 1 u64
  DevBuf gcum; u64 ngroups = 0;       // 7 x (ngroups + 1) u64: cumulative symbol counts (rows 0..5) and positions (row 6) at the starts of the 62-block groups
  DevBuf blen;                        // nblocks u64: positions per block, between the first decode pass and the transcode
  DevBuf cum32;                       // 5 x (nblocks + 1) u32: cumulative counts of 1..5 at the block starts, relative to the super block of the
                                      // start (kernels/encoder.hip.h); written by the encoder when it fits the budget, else answered by the records
  DevBuf flags;                       // k_block_len's verdict on the stream (read by upload_validate)
#ifdef BWTM_EXPERIMENTAL
  mutable DevBuf sview, vsup;         // the search view (built on demand by the frontier search, dropped with the records)
  mutable u64 nview = 0;
  mutable bool view_ready = false;    // set after both kernels that fill the view were queued
#endif
  // A WINDOW of an index's records (bwtm_index_upload_window; the merge over partitioned records): `recs` holds the
  // records [win_first, win_first + win_count) only, and view() hands the kernels a base pointer shifted back by win_first records, so that
  // absolute record numbers address it unchanged.  Positions outside the window must never be queried: only bwtm_part_* and the
  // output-range entry points take such a handle.
  u64 win_first = 0, win_count = 0;
  bool windowed = false;

  IndexView view() const
  {
    IndexView v;
    v.recs = recs.as<const uint4>(); v.sup = sup.as<const u64>();
    v.n = n; v.m = m; v.nrecs = nrecs;
    for(int c = 0; c < 8; c++) { v.C[c] = C[c]; }
#ifdef BWTM_EXPERIMENTAL
    v.view = (view_ready ? sview.as<const uint4>() : nullptr); v.vsup = (view_ready ? vsup.as<const u64>() : nullptr); v.nview = (view_ready ? nview : 0);
#endif
    if(windowed) { v.recs = (const uint4*)((const char*)recs.p - (win_first << 6)); }        // 64 bytes per record
    return v;
  }
};

// Entry points that query arbitrary positions refuse a window of an index.
#define WHOLE_INDEX(x, who) if((x)->windowed) { return fail(BWTM_EINVAL, who ": a window of an index (bwtm_index_upload_window) only serves the merge over partitioned records"); }

namespace
{

// Frees a handle inside its own context (its buffers return to that context's pool).
void index_destroy(bwtm_index* x)
{
  if(!x) { return; }
  Scope scope(x->ctx);
  if(scope.rc == BWTM_OK && x->borrowed) { (void)hipStreamSynchronize(CTX.stream); }   // queued readers of the caller's buffer
  delete x;                                         // buffers return to the pool (stream ordered)
}

//------------------------------------------------------------------------------
// Upload: native bytes -> block_start, gcum (first decode pass) -> validation -> records (transcode).

// Buffer for a native byte stream: 16 zero bytes of padding keep the last partial block readable with
// 16-byte loads.  Only the padding is cleared; the stream itself is written by the caller.
int alloc_native(DevBuf& buf, u64 nbytes)
{
  TRY(buf.alloc(nbytes + 16));
  HIP_TRY(hipMemsetAsync((u8*)buf.p + nbytes, 0, 16, CTX.stream));
  return BWTM_OK;
}

// Step 1.  Three parts, so that a caller can queue the copies of several inputs before any decode pass (bwtm_merge_host):
//   upload_prepare   allocates the sample arrays
//   upload_copies    (host sources only) copies the bytes into x->data in chunks on the copy stream, one event per chunk
//   upload_decode    queues the first decode pass (k_block_len) chunk by chunk on the compute stream; the pass over chunk k waits
//                    for chunk k's event only, so it runs while chunk k + 1 is in flight
struct UploadEvents
{
  std::vector<hipEvent_t> ev;
  u64 groups_per_chunk = 0;              // the chunking the copies were queued with (the upload_chunk knob may change before they are consumed)
  UploadEvents() {}
  UploadEvents(const UploadEvents&) = delete; UploadEvents& operator=(const UploadEvents&) = delete;
  ~UploadEvents() { for(hipEvent_t e : ev) { (void)hipEventDestroy(e); } }
};

u64 upload_groups_per_chunk(const bwtm_index* x, bool from_host)
{
  const u64 group_bytes = (u64)GROUP * RLE_BLOCK;
  return (from_host ? std::max<u64>(1, (u64)g_tune.upload_chunk / group_bytes) : x->ngroups);
}

int upload_prepare(bwtm_index* x)
{
  x->nblocks = div_up(x->nbytes, RLE_BLOCK);
  x->ngroups = std::max<u64>(1, div_up(x->nblocks, (u64)GROUP));
  const u64 gstride = x->ngroups + 1;
  x->cum32.release();
  TRY(x->block_start.alloc((x->nblocks + 1) * sizeof(u64)));
  TRY(x->blen.alloc(std::max<u64>(1, x->nblocks) * sizeof(u64)));
  TRY(x->gcum.alloc(7 * gstride * sizeof(u64)));
  TRY(x->flags.alloc(sizeof(u32), true));
  // the kernel fills the columns [0, ngroups); the extra column of each exclusive scan is zeroed here
  // (plain 1-D calls: the 2-D memset / memcpy entry points of the runtime reject the pool's mapped blocks)
  for(u64 c = 0; c < 7; c++) { HIP_TRY(hipMemsetAsync(x->gcum.as<u64>() + c * gstride + x->ngroups, 0, sizeof(u64), CTX.stream)); }
  if(x->nblocks == 0) { HIP_TRY(hipMemsetAsync(x->block_start.p, 0, sizeof(u64), CTX.stream)); }       // the empty stream: no group writes the entry behind the last block
  return BWTM_OK;
}

// The caller has forked the copy stream (fork_copy_stream: x->data may be a recycled block with queued users).
int upload_copies(bwtm_index* x, const u8* host_src, UploadEvents& events)
{
  const u64 group_bytes = (u64)GROUP * RLE_BLOCK;
  const u64 groups_per_chunk = upload_groups_per_chunk(x, true);
  events.groups_per_chunk = groups_per_chunk;
  for(u64 g0 = 0; g0 < x->ngroups; g0 += groups_per_chunk)
  {
    const u64 g1 = std::min(x->ngroups, g0 + groups_per_chunk);
    const u64 from = g0 * group_bytes, to = std::min(x->nbytes, g1 * group_bytes);
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if(e == hipSuccess) { events.ev.push_back(ev); }
    if(e == hipSuccess && to > from) { e = hipMemcpyAsync((u8*)x->data.p + from, host_src + from, to - from, hipMemcpyHostToDevice, CTX.copy_stream); }
    if(e == hipSuccess) { e = hipEventRecord(ev, CTX.copy_stream); }
    if(e != hipSuccess)
    {
      (void)hipStreamSynchronize(CTX.copy_stream);                   // the caller's buffer must not be read after the call returns
      return fail(BWTM_ENODEV, "H2D copy failed: %s", hipGetErrorString(e));
    }
  }
  return BWTM_OK;
}

int upload_decode(bwtm_index* x, const UploadEvents* events)
{
  const u64 gstride = x->ngroups + 1;
  const u64 groups_per_chunk = (events && events->groups_per_chunk > 0 ? events->groups_per_chunk : upload_groups_per_chunk(x, events != nullptr));
  if(events && events->ev.size() != div_up(x->ngroups, groups_per_chunk)) { return fail(BWTM_EINVAL, "upload: %zu chunk events for %llu chunks", events->ev.size(), (unsigned long long)div_up(x->ngroups, groups_per_chunk)); }
  u64 chunk = 0;
  for(u64 g0 = 0; g0 < x->ngroups; g0 += groups_per_chunk, chunk++)
  {
    const u64 g1 = std::min(x->ngroups, g0 + groups_per_chunk);
    if(events) { HIP_TRY(hipStreamWaitEvent(CTX.stream, events->ev[chunk], 0)); }
    LAUNCH("block_len", k_block_len, div_up(g1 - g0, BLOCK_THREADS / WAVE), BLOCK_THREADS,
      x->native_bytes(), x->nbytes, x->nblocks, g0, g1, x->blen.as<u64>(), x->gcum.as<u64>(), gstride, x->flags.as<u32>());
  }
  return BWTM_OK;
}

int upload_queue(bwtm_index* x, const u8* host_src)
{
  TRY(upload_prepare(x));
  if(!host_src) { return upload_decode(x, nullptr); }
  UploadEvents events;
  TRY(fork_copy_stream());
  TRY(upload_copies(x, host_src, events));
  int rc = upload_decode(x, &events);
  if(rc != BWTM_OK) { (void)hipStreamSynchronize(CTX.copy_stream); }
  return rc;
}

// Step 2.  Scans; the stream's verdict and symbol totals travel to host_scratch[slot .. slot + 6].
int upload_scan(bwtm_index* x, u32 slot)
{
  const u64 gstride = x->ngroups + 1;
  TRY(device_scan_multi<0>(x->gcum.as<u64>(), x->gcum.as<u64>(), gstride, 7, gstride));      // six symbol counts + the positions, per group
  for(u32 c = 0; c < 6; c++) { TRY(fetch_u64(x->gcum.as<u64>() + c * gstride + x->ngroups, slot + c)); }
  CTX.host_scratch[slot + 6] = 0;                         // the copy below fills the low 32 bits
  HIP_TRY(hipMemcpyAsync(CTX.host_scratch + slot + 6, x->flags.p, sizeof(u32), hipMemcpyDeviceToHost, CTX.stream));
  return BWTM_OK;
}

// Step 3 (after a synchronisation).  Validates the header against the stream and derives C (Alphabet(counts),
// support.cpp:84-91); a caller-supplied C must agree with the counts.
int upload_validate(bwtm_index* x, u64 sequences, u64 bases, const u64* C, u32 slot)
{
  const u64* totals = CTX.host_scratch + slot;
  const u32 flags = (u32)CTX.host_scratch[slot + 6];
  x->flags.release();
  if(flags & 1u) { return fail(BWTM_EINVAL, "not a canonical run-length stream: a full 64-byte block encodes fewer than 64 positions"); }
  u64 sum = 0; for(int c = 0; c < 6; c++) { sum += totals[c]; }
  if(sum != bases) { return fail(BWTM_EINVAL, "native stream decodes to %llu positions, header says %llu", (unsigned long long)sum, (unsigned long long)bases); }
  if(totals[0] != sequences) { return fail(BWTM_EINVAL, "native stream holds %llu endmarkers, header says %llu sequences", (unsigned long long)totals[0], (unsigned long long)sequences); }
  x->n = bases; x->m = sequences;
  x->C[0] = 0;
  for(int c = 0; c < 6; c++) { x->C[c + 1] = x->C[c] + totals[c]; }
  x->C[7] = x->C[6];
  if(C)
  {
    for(int c = 0; c <= 6; c++)
    {
      if(C[c] != x->C[c]) { return fail(BWTM_EINVAL, "alphabet mismatch: C[%d] = %llu, the stream's symbol counts give %llu", c, (unsigned long long)C[c], (unsigned long long)x->C[c]); }
    }
  }
  x->has_native = true;
  return BWTM_OK;
}

// The records of an index (or of a window of it: `held` positions from the share's first one up to n_end, records addressed through a shifted
// pointer) from its native bytes, block lengths and scanned group tables.  One wave per group; the LDS window is sized to the positions a group
// covers on average (iid reads: ~5300).  BWTM_TUNE=recs_window=... picks the window, recs_uniform=1 / -1 forces / forbids the straight-line deposit.
int build_records(bwtm_index* x, u64 held, u64 n_end, uint4* recs, u64 nrecs)
{
  const u64 gstride = x->ngroups + 1;
  const u64 per_group = held / x->ngroups;
  const bool long_runs = (x->nblocks > 0 && held / x->nblocks > 400);        // > ~6 positions per byte: cooperative fill of long runs pays
#define BUILD_RECS(W, WAVES, FILL, UNIFORM) LAUNCH("build_recs", (k_build_recs<W, WAVES, FILL, UNIFORM>), div_up(x->ngroups, WAVES), WAVES * WAVE, \
    x->native_bytes(), x->nbytes, x->blen.as<const u64>(), x->block_start.as<u64>(), x->gcum.as<const u64>(), gstride, x->nblocks, x->ngroups, n_end, \
    x->sup.as<const u64>(), recs, nrecs)
  const u64 window = (g_tune.recs_window != 0 ? (u64)g_tune.recs_window : (per_group <= 6500 ? 8192 : (per_group <= 14000 ? 16384 : 32768)));
  // the straight-line deposit for streams of longer runs (kernels/transcode.hip.h): from ~3.5 positions per byte on
  const bool uniform = (g_tune.recs_uniform != 0 ? g_tune.recs_uniform > 0 : window == 32768);
  if(window == 8192) { if(uniform) { BUILD_RECS(8192, 4, false, true); } else { BUILD_RECS(8192, 4, false, false); } }
  else if(window == 16384) { if(uniform) { BUILD_RECS(16384, 4, false, true); } else { BUILD_RECS(16384, 4, false, false); } }
  else if(!long_runs) { if(uniform) { BUILD_RECS(32768, 2, false, true); } else { BUILD_RECS(32768, 2, false, false); } }
  else { if(uniform) { BUILD_RECS(32768, 2, true, true); } else { BUILD_RECS(32768, 2, true, false); } }
#undef BUILD_RECS
  return BWTM_OK;
}

// Step 4.  Records + super table from the native stream (needs x->n; C is not used by the kernels).
int transcode(bwtm_index* x)
{
  x->nrecs = num_records(x->n); x->nsup = num_supers(x->n);
  TRY(x->recs.alloc(x->nrecs * 64));
  TRY(x->sup.alloc(x->nsup * SUP_STRIDE * sizeof(u64)));
  const u64 gstride = x->ngroups + 1;
  LAUNCH("build_sup", k_build_sup, div_up(x->nsup * WAVE, BLOCK_THREADS), BLOCK_THREADS,   // one wave per super
    x->native_bytes(), x->nbytes, x->blen.as<const u64>(), x->gcum.as<const u64>(), gstride, x->nblocks, x->ngroups, x->n,
    x->sup.as<u64>(), x->nsup);
  TRY(build_records(x, x->n, x->n, x->recs.as<uint4>(), x->nrecs));
  x->blen.release();                                             // (stream ordered) the block starts are in block_start now
  return BWTM_OK;
}

// The whole upload of one index, blocking (bwtm_index_upload / _from_device*).
int upload_blocking(bwtm_index* x, const u8* host_src, u64 sequences, u64 bases, const u64* C)
{
  x->n = bases; x->m = sequences;
  int rc = upload_queue(x, host_src);
  if(rc == BWTM_OK) { rc = upload_scan(x, 0); }
  hipError_t e1 = hipStreamSynchronize(CTX.copy_stream), e2 = hipStreamSynchronize(CTX.stream);   // also on failure: the caller's buffer is free again
  if(rc != BWTM_OK) { return rc; }
  if(e1 != hipSuccess || e2 != hipSuccess) { return fail(BWTM_ENODEV, "upload failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); }
  TRY(upload_validate(x, sequences, bases, C, 0));
  TRY(transcode(x));
  return BWTM_OK;
}

// Bytes of the compact samples (cum32) of an index of `blocks` blocks: what the eager_cum_budget knob is compared with.
u64 cum32_bytes(u64 blocks) { return 5 * (blocks + 1) * sizeof(u32); }

// samples[c] at the block starts (bwt.cpp:489-511) in the compact form, from block_start and the rank structure.
int ensure_block_cum(bwtm_index* x)
{
  if(x->cum32.p) { return BWTM_OK; }
  const u64 stride = x->nblocks + 1;
  TRY(x->cum32.alloc(cum32_bytes(x->nblocks)));
  LAUNCH("block_cum", k_block_cum32, div_up(stride, BLOCK_THREADS), BLOCK_THREADS,
    x->recs.as<const uint4>(), x->block_start.as<const u64>(), (u64)0, stride, x->cum32.as<u32>(), stride);
  return BWTM_OK;
}

#ifdef BWTM_EXPERIMENTAL
// The search view of an index (bwtm_view.h, kernels/search_view.hip.h): built when the frontier search first wants it, kept with
// the records (a chained merge searches the same first input again), released with the handle.  A launch that fails leaves no
// half-built view behind: the buffers are released and the search goes on with the ordinary records.
int ensure_view(const bwtm_index* x)
{
  if(x->view_ready) { return BWTM_OK; }
  x->nview = num_view_records(x->n);
  const u64 nvsup = num_view_supers(x->n);
  auto build = [&]() -> int
  {
    TRY(x->vsup.alloc(nvsup * SUP_STRIDE * sizeof(u64)));
    TRY(x->sview.alloc(x->nview * 64));
    IndexView v = x->view();
    LAUNCH("view_sup", k_view_sup, div_up(nvsup, BLOCK_THREADS), BLOCK_THREADS, v, x->vsup.as<u64>(), nvsup);
    LAUNCH("view_build", k_view_build, div_up(x->nview, BLOCK_THREADS), BLOCK_THREADS, v, x->vsup.as<const u64>(), x->sview.as<uint4>(), x->nview);
    return BWTM_OK;
  };
  const int rc = build();
  if(rc != BWTM_OK) { x->sview.release(); x->vsup.release(); return rc; }
  x->view_ready = true;
  return BWTM_OK;
}
#endif

//------------------------------------------------------------------------------
// Encoder: records -> native bytes (RunBuffer + Run::write) + block starts (BWT::build).

struct EncodePlan
{
  u64 ntiles = 0, nseg = 0, ngroups = 0, total = 0;
  DevBuf lasthead, table, group_table, group_base, seg_base;
  std::vector<u64> group_base_host;       // byte offset at which every fold group of segments starts (+ total)
};

// Size pass: the byte offset of every segment (and with it the size of the stream).  Synchronises.
int encode_size(bwtm_index* x, EncodePlan& plan)
{
  plan.ntiles = (x->n >> 6) + 1;
  plan.nseg = div_up(plan.ntiles, SEG_TILES);
  plan.ngroups = div_up(plan.nseg, FOLD_GROUP);
  const u64 nseg = plan.nseg, ngroups = plan.ngroups;
  TRY(plan.lasthead.alloc(nseg * sizeof(u64)));
  TRY(plan.table.alloc(nseg * 64 * sizeof(u32)));
  TRY(plan.group_table.alloc(ngroups * 64 * sizeof(u64)));
  TRY(plan.group_base.alloc((ngroups + 1) * sizeof(u64)));
  TRY(plan.seg_base.alloc(nseg * sizeof(u64)));
  const u64 wave_grid = div_up(nseg * WAVE, BLOCK_THREADS);
  LAUNCH("enc_lasthead", k_enc_lasthead, wave_grid, BLOCK_THREADS, x->recs.as<const uint4>(), x->nrecs, x->n, plan.ntiles, (u64)0, nseg, plan.lasthead.as<u64>());
  TRY(device_scan<1>(plan.lasthead.as<u64>(), plan.lasthead.as<u64>(), nseg));       // -> (last head before the segment) + 1
  LAUNCH("enc_size", k_enc_size, wave_grid, BLOCK_THREADS, x->recs.as<const uint4>(), x->nrecs, x->n, plan.ntiles, (u64)0, nseg,
    plan.lasthead.as<const u64>(), (u64)0, plan.table.as<u32>());
  LAUNCH("fold_group", k_fold_group, ngroups, WAVE, plan.table.as<const u32>(), nseg, plan.group_table.as<u64>());
  LAUNCH("fold_top", k_fold_top, 1, WAVE, plan.group_table.as<const u64>(), ngroups, (u64)0, plan.group_base.as<u64>());
  LAUNCH("fold_seg", k_fold_seg, ngroups, WAVE, plan.table.as<const u32>(), nseg, plan.group_base.as<const u64>(), plan.seg_base.as<u64>());
  plan.group_base_host.resize(ngroups + 1);
  HIP_TRY(hipMemcpyAsync(plan.group_base_host.data(), plan.group_base.p, (ngroups + 1) * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  plan.total = plan.group_base_host[ngroups];
  return BWTM_OK;
}

// Emit pass.  With `host_out` the bytes of every finished range of segments are copied to the host on the copy stream
// while the next range is being written (the caller joins the copy stream).
int encode_emit(bwtm_index* x, EncodePlan& plan, u8* host_out, bool with_cum = false)
{
  const u64 total = plan.total;
  x->nbytes = total;
  x->nblocks = div_up(total, RLE_BLOCK);
  TRY(alloc_native(x->data, total));
  x->gcum.release(); x->ngroups = 0; x->cum32.release();
  const u64 cum_stride = x->nblocks + 1;
  if(with_cum) { TRY(x->cum32.alloc(cum32_bytes(x->nblocks))); }
  // k_enc_emit records the position at which every 64-byte block starts; the entry after the last block is n
  TRY(x->block_start.alloc((x->nblocks + 1) * sizeof(u64)));
  CTX.host_scratch[63] = x->n;
  HIP_TRY(hipMemcpyAsync(x->block_start.as<u64>() + x->nblocks, CTX.host_scratch + 63, sizeof(u64), hipMemcpyHostToDevice, CTX.stream));
  hipEvent_t ev = nullptr;
  if(host_out) { HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); }
  int rc = BWTM_OK;
  for(u64 g0 = 0; g0 < plan.ngroups && rc == BWTM_OK; )
  {
    u64 g1 = g0 + 1;
    if(host_out) { while(g1 < plan.ngroups && plan.group_base_host[g1] - plan.group_base_host[g0] < (u64)g_tune.download_chunk) { g1++; } }
    else { g1 = plan.ngroups; }
    const u64 s0 = g0 * FOLD_GROUP, s1 = std::min(plan.nseg, g1 * FOLD_GROUP);
    auto launch = [&]() -> int
    {
      if(with_cum)
      {
        LAUNCH("enc_emit", k_enc_emit<true>, div_up((s1 - s0) * WAVE, BLOCK_THREADS), BLOCK_THREADS, x->recs.as<const uint4>(), x->nrecs, x->n, plan.ntiles, s0, s1,
          plan.lasthead.as<const u64>(), (u64)0, plan.seg_base.as<const u64>(), x->data.as<u8>(), x->block_start.as<u64>(), x->cum32.as<u32>(), cum_stride);
      }
      else
      {
        LAUNCH("enc_emit", k_enc_emit<false>, div_up((s1 - s0) * WAVE, BLOCK_THREADS), BLOCK_THREADS, x->recs.as<const uint4>(), x->nrecs, x->n, plan.ntiles, s0, s1,
          plan.lasthead.as<const u64>(), (u64)0, plan.seg_base.as<const u64>(), x->data.as<u8>(), x->block_start.as<u64>(), (u32*)nullptr, (u64)0);
      }
      return BWTM_OK;
    };
    rc = launch();
    if(rc == BWTM_OK && host_out)
    {
      const u64 from = plan.group_base_host[g0], to = plan.group_base_host[g1];
      hipError_t e = hipEventRecord(ev, CTX.stream);
      if(e == hipSuccess) { e = hipStreamWaitEvent(CTX.copy_stream, ev, 0); }
      if(e == hipSuccess && to > from) { e = hipMemcpyAsync(host_out + from, x->data.as<u8>() + from, to - from, hipMemcpyDeviceToHost, CTX.copy_stream); }
      if(e != hipSuccess) { rc = fail(BWTM_ENODEV, "D2H copy failed: %s", hipGetErrorString(e)); }
    }
    g0 = g1;
  }
  if(ev) { (void)hipEventDestroy(ev); }
  if(rc == BWTM_OK && with_cum)
  {
    // the entry behind the last block: the counts at n (every block's own entry was stored by the lane that opened it)
    auto last = [&]() -> int
    {
      LAUNCH("block_cum", k_block_cum32, 1, BLOCK_THREADS, x->recs.as<const uint4>(), x->block_start.as<const u64>(), x->nblocks, (u64)1, x->cum32.as<u32>(), cum_stride);
      return BWTM_OK;
    };
    rc = last();
  }
  x->has_native = true;
  return rc;
}

int encode_blocking(bwtm_index* x)
{
  if(x->has_native) { return BWTM_OK; }
  x->nbytes = 0; x->nblocks = 0;
  if(x->n > 0)
  {
    EncodePlan plan;
    TRY(encode_size(x, plan));
    // BWT::build, bwt.cpp:476-512: the samples of the new stream.  block_start is always there; the cumulative counts are answered by
    // the rank structure and materialized -- by the emit pass itself, in the compact form cum32 -- only when they are small enough to sit
    // next to everything else (a 2 x 50 Gbase result has 24 GB of them: produced in chunks when downloaded).
    const u64 blocks = div_up(plan.total, RLE_BLOCK);
    TRY(encode_emit(x, plan, nullptr, cum32_bytes(blocks) <= (u64)g_tune.eager_cum_budget));
    return BWTM_OK;
  }
  else
  {
    TRY(alloc_native(x->data, 0));
    TRY(x->block_start.alloc(sizeof(u64), true));
    x->gcum.release(); x->ngroups = 0; x->cum32.release();
    x->has_native = true;
  }
  if(cum32_bytes(x->nblocks) <= (u64)g_tune.eager_cum_budget) { TRY(ensure_block_cum(x)); }      // the empty index: one entry
  return BWTM_OK;
}

// Samples to the host in the form BWT::build computes them; chunked through two staging buffers so that the six
// cumulative arrays never have to be resident as a whole.  Uses the copy stream; returns after everything arrived.
int download_samples(bwtm_index* x, u64* block_end, u64* cum)
{
  const u64 stride = x->nblocks + 1;
  const u64 CH = 1ull << 22;                                  // blocks per chunk: 7 x 32 MiB of staging per buffer
  DevBuf stage[2];
  hipEvent_t filled[2] = {nullptr, nullptr}, drained[2] = {nullptr, nullptr};
  int rc = BWTM_OK;
  auto body = [&]() -> int
  {
    for(int k = 0; k < 2; k++)
    {
      TRY(stage[k].alloc(7 * CH * sizeof(u64)));
      HIP_TRY(hipEventCreateWithFlags(&filled[k], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&drained[k], hipEventDisableTiming));
    }
    u64 round = 0;
    for(u64 b0 = 0; b0 < stride; b0 += CH, round++)
    {
      const int k = (int)(round & 1);
      const u64 cnt = std::min(CH, stride - b0);               // entries of cum in this chunk
      const u64 nbe = (b0 + cnt <= x->nblocks ? cnt : x->nblocks - b0);   // entries of block_end
      u64* st = stage[k].as<u64>();
      if(round >= 2) { HIP_TRY(hipStreamWaitEvent(CTX.stream, drained[k], 0)); }
      if(nbe > 0) { LAUNCH("block_end", k_block_end, div_up(nbe, BLOCK_THREADS), BLOCK_THREADS, x->block_start.as<const u64>(), b0, nbe, st); }
      if(x->cum32.p)                                                     // the compact form written by the encoder, expanded chunk by chunk
      {
        LAUNCH("cum_expand", k_cum_expand, div_up(cnt, BLOCK_THREADS), BLOCK_THREADS, x->sup.as<const u64>(), x->block_start.as<const u64>(), x->cum32.as<const u32>(), stride,
          b0, cnt, st + CH, CH);
      }
      else { LAUNCH("block_cum", k_block_cum, div_up(cnt, BLOCK_THREADS), BLOCK_THREADS, x->view(), x->block_start.as<const u64>(), b0, cnt, st + CH, CH); }
      HIP_TRY(hipEventRecord(filled[k], CTX.stream));
      HIP_TRY(hipStreamWaitEvent(CTX.copy_stream, filled[k], 0));
      if(nbe > 0) { HIP_TRY(hipMemcpyAsync(block_end + b0, st, nbe * sizeof(u64), hipMemcpyDeviceToHost, CTX.copy_stream)); }
      for(u64 c = 0; c < 6; c++)
      {
        HIP_TRY(hipMemcpyAsync(cum + c * stride + b0, st + CH + c * CH, cnt * sizeof(u64), hipMemcpyDeviceToHost, CTX.copy_stream));
      }
      HIP_TRY(hipEventRecord(drained[k], CTX.copy_stream));
    }
    return BWTM_OK;
  };
  rc = body();
  hipError_t e1 = hipStreamSynchronize(CTX.copy_stream), e2 = hipStreamSynchronize(CTX.stream);
  for(int k = 0; k < 2; k++) { if(filled[k]) { (void)hipEventDestroy(filled[k]); } if(drained[k]) { (void)hipEventDestroy(drained[k]); } }
  if(rc != BWTM_OK) { return rc; }
  if(e1 != hipSuccess || e2 != hipSuccess) { return fail(BWTM_ENODEV, "sample download failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); }
  return BWTM_OK;
}

// Width of the compact sample fields: 1, 2 or 4 bytes when every block encodes fewer than 2^8 - 1 / 2^16 - 1 / 2^32 - 1 positions, else 8
// (= use the full arrays).  Synchronises.
int samples_width(bwtm_index* x, int* width)
{
  *width = 1;
  if(x->nblocks == 0) { return BWTM_OK; }
  DevBuf m; TRY(m.alloc(sizeof(u64), true));
  LAUNCH("block_field_max", k_block_field_max, std::min<u64>(div_up(x->nblocks, BLOCK_THREADS), 2048), BLOCK_THREADS, x->block_start.as<const u64>(), x->nblocks,
    m.as<unsigned long long>());
  TRY(fetch_u64(m.as<u64>(), 0));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  const u64 mx = CTX.host_scratch[0];
  *width = (mx < 0xFFull ? 1 : (mx < 0xFFFFull ? 2 : (mx < 0xFFFFFFFFull ? 4 : 8)));
  return BWTM_OK;
}

// Compact samples to the host, chunked through two staging buffers like download_samples.
// A chunk is 32 MiB per field row (2^25 blocks of 1-byte fields): the staging buffers are then large enough to be MAPPED blocks of the
// pool, and a result of 10^9 blocks travels in ~40 chunks of 12 copies.  The first version staged 4 Mi blocks per chunk in buffers of
// 24 MiB + 3 MiB, i.e. small blocks that come from hipMalloc: at 2 x 50 Gbase, with the device nearly full, those allocations (and 3420
// copies of 0.5 - 4 MiB) made the phase take 586 ms for 8.4 GB, four times the link's time.
template<class T>
int download_samples_compact(bwtm_index* x, T* fields, u64* anchors)
{
  const u64 nb = x->nblocks, nanch = div_up(nb, 64);
  if(nb == 0) { return BWTM_OK; }
  const u64 CH = std::min<u64>((32ull << 20) / sizeof(T), div_up(nb, 64) * 64);      // blocks per chunk (a multiple of 64)
  const u64 field_bytes = 6 * CH * sizeof(T);                                         // a multiple of 64 bytes: the anchors behind them stay aligned
  DevBuf stage[2];
  hipEvent_t filled[2] = {nullptr, nullptr}, drained[2] = {nullptr, nullptr};
  auto body = [&]() -> int
  {
    for(int k = 0; k < 2 && (k == 0 || nb > CH); k++)
    {
      TRY(stage[k].alloc(field_bytes + 6 * (CH / 64) * sizeof(u64)));
      HIP_TRY(hipEventCreateWithFlags(&filled[k], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&drained[k], hipEventDisableTiming));
    }
    u64 round = 0;
    for(u64 b0 = 0; b0 < nb; b0 += CH, round++)
    {
      const int k = (int)(round & 1);
      const u64 cnt = std::min(CH, nb - b0), na = div_up(cnt, 64);
      T* sf = stage[k].as<T>();
      u64* sa = (u64*)(stage[k].as<u8>() + field_bytes);
      if(round >= 2) { HIP_TRY(hipStreamWaitEvent(CTX.stream, drained[k], 0)); }
      LAUNCH("block_fields", k_block_fields<T>, div_up(cnt, BLOCK_THREADS), BLOCK_THREADS, x->view(), x->block_start.as<const u64>(), b0, cnt,
        sf, CH, sa, CH / 64);
      HIP_TRY(hipEventRecord(filled[k], CTX.stream));
      HIP_TRY(hipStreamWaitEvent(CTX.copy_stream, filled[k], 0));
      for(u64 c = 0; c < 6; c++)
      {
        HIP_TRY(hipMemcpyAsync(fields + c * nb + b0, sf + c * CH, cnt * sizeof(T), hipMemcpyDeviceToHost, CTX.copy_stream));
        HIP_TRY(hipMemcpyAsync(anchors + c * nanch + b0 / 64, sa + c * (CH / 64), na * sizeof(u64), hipMemcpyDeviceToHost, CTX.copy_stream));
      }
      HIP_TRY(hipEventRecord(drained[k], CTX.copy_stream));
    }
    return BWTM_OK;
  };
  int rc = body();
  hipError_t e1 = hipStreamSynchronize(CTX.copy_stream), e2 = hipStreamSynchronize(CTX.stream);
  for(int k = 0; k < 2; k++) { if(filled[k]) { (void)hipEventDestroy(filled[k]); } if(drained[k]) { (void)hipEventDestroy(drained[k]); } }
  if(rc != BWTM_OK) { return rc; }
  if(e1 != hipSuccess || e2 != hipSuccess) { return fail(BWTM_ENODEV, "sample download failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); }
  return BWTM_OK;
}

} // namespace

//------------------------------------------------------------------------------
// C ABI.

extern "C" int bwtm_index_upload(const uint8_t* data, uint64_t nbytes, uint64_t sequences, uint64_t bases,
  const uint64_t* C, bwtm_index** out)
{
  ENTER(nullptr);
  if(!out || (nbytes > 0 && !data)) { return fail(BWTM_EINVAL, "bwtm_index_upload: null argument"); }
  bwtm_index* x = new bwtm_index();
  x->ctx = t_ctx; x->nbytes = nbytes;
  int rc = alloc_native(x->data, nbytes);
  if(rc == BWTM_OK) { rc = upload_blocking(x, (nbytes > 0 ? data : nullptr), sequences, bases, C); }
  if(rc != BWTM_OK) { delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

extern "C" int bwtm_index_from_device(const void* device_data, uint64_t nbytes, uint64_t sequences, uint64_t bases,
  const uint64_t* C, bwtm_index** out)
{
  ENTER(nullptr);
  if(!out || (nbytes > 0 && !device_data)) { return fail(BWTM_EINVAL, "bwtm_index_from_device: null argument"); }
  bwtm_index* x = new bwtm_index();
  x->ctx = t_ctx; x->nbytes = nbytes;
  int rc = alloc_native(x->data, nbytes);
  if(rc == BWTM_OK && nbytes > 0)
  {
    hipError_t e = hipMemcpyAsync(x->data.p, device_data, nbytes, hipMemcpyDeviceToDevice, CTX.stream);
    if(e != hipSuccess) { rc = fail(BWTM_ENODEV, "D2D copy failed: %s", hipGetErrorString(e)); }
  }
  if(rc == BWTM_OK) { rc = upload_blocking(x, nullptr, sequences, bases, C); }
  if(rc != BWTM_OK) { (void)hipStreamSynchronize(CTX.stream); delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

extern "C" int bwtm_index_from_device_borrowed(const void* device_data, uint64_t nbytes, uint64_t sequences, uint64_t bases,
  const uint64_t* C, bwtm_index** out)
{
  ENTER(nullptr);
  if(!out || !device_data) { return fail(BWTM_EINVAL, "bwtm_index_from_device_borrowed: null argument"); }
  if(((uintptr_t)device_data & 15) != 0) { return fail(BWTM_EINVAL, "bwtm_index_from_device_borrowed: the buffer must be 16-byte aligned"); }
  bwtm_index* x = new bwtm_index();
  x->ctx = t_ctx; x->nbytes = nbytes; x->borrowed = device_data;
  int rc = upload_blocking(x, nullptr, sequences, bases, C);
  if(rc != BWTM_OK) { (void)hipStreamSynchronize(CTX.stream); delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

namespace
{

// Plain symbols on the device (one byte per position, 0 = endmarker) -> records.  Synchronizes the compute stream.
int index_from_symbols(const u8* device_symbols, u64 bases, bwtm_index* x)
{
  x->n = bases;
  x->nrecs = num_records(bases); x->nsup = num_supers(bases);
  u64 stride = x->nrecs + 1;
  DevBuf cnt; TRY(cnt.alloc(6 * stride * sizeof(u64), true));
  LAUNCH("sym_counts", k_sym_counts, div_up(x->nrecs, BLOCK_THREADS), BLOCK_THREADS,
    device_symbols, bases, x->nrecs, cnt.as<u64>(), stride);
  TRY(device_scan_multi<0>(cnt.as<u64>(), cnt.as<u64>(), stride, 6, stride));
  for(u32 c = 0; c < 6; c++) { TRY(fetch_u64(cnt.as<u64>() + c * stride + x->nrecs, c)); }
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  const u64* totals = CTX.host_scratch;
  x->m = totals[0];
  x->C[0] = 0; for(int c = 0; c < 6; c++) { x->C[c + 1] = x->C[c] + totals[c]; } x->C[7] = x->C[6];
  TRY(x->recs.alloc(x->nrecs * 64));
  TRY(x->sup.alloc(x->nsup * SUP_STRIDE * sizeof(u64)));
  LAUNCH("sym_sup", k_sym_sup, div_up(x->nsup, BLOCK_THREADS), BLOCK_THREADS, cnt.as<const u64>(), stride, x->nrecs, x->sup.as<u64>(), x->nsup);
  LAUNCH("sym_recs", k_sym_recs, div_up(x->nrecs, BLOCK_THREADS), BLOCK_THREADS,
    device_symbols, bases, cnt.as<const u64>(), stride, x->sup.as<const u64>(), x->recs.as<uint4>(), x->nrecs);
  HIP_TRY(hipStreamSynchronize(CTX.stream));     // the caller may release `device_symbols` on return
  return BWTM_OK;
}

} // namespace

extern "C" int bwtm_index_from_symbols_device(const void* device_symbols, uint64_t bases, bwtm_index** out)
{
  ENTER(nullptr);
  if(!out || (bases > 0 && !device_symbols)) { return fail(BWTM_EINVAL, "bwtm_index_from_symbols_device: null argument"); }
  bwtm_index* x = new bwtm_index();
  x->ctx = t_ctx;
  int rc = index_from_symbols((const u8*)device_symbols, bases, x);
  if(rc != BWTM_OK) { (void)hipStreamSynchronize(CTX.stream); delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

extern "C" void bwtm_index_free(bwtm_index* index) { index_destroy(index); }

extern "C" uint64_t bwtm_index_bases(const bwtm_index* x)     { return x ? x->n : 0; }
extern "C" uint64_t bwtm_index_sequences(const bwtm_index* x) { return x ? x->m : 0; }
extern "C" uint64_t bwtm_index_bytes(const bwtm_index* x)     { return (x && x->has_native) ? x->nbytes : 0; }
extern "C" uint64_t bwtm_index_blocks(const bwtm_index* x)    { return (x && x->has_native) ? x->nblocks : 0; }
extern "C" void bwtm_index_C(const bwtm_index* x, uint64_t* C) { for(int c = 0; c <= 6; c++) { C[c] = x->C[c]; } }

extern "C" int bwtm_index_drop_native(bwtm_index* x)
{
  if(!x) { return fail(BWTM_EINVAL, "null index"); }
  ENTER(x->ctx);
  if(x->borrowed) { HIP_TRY(hipStreamSynchronize(CTX.stream)); x->borrowed = nullptr; }
  x->data.release(); x->cum32.release(); x->gcum.release(); x->block_start.release(); x->blen.release();
  x->has_native = false; x->nbytes = 0; x->nblocks = 0;
  return BWTM_OK;
}

extern "C" int bwtm_index_encode(bwtm_index* x)
{
  if(!x) { return fail(BWTM_EINVAL, "null index"); }
  WHOLE_INDEX(x, "bwtm_index_encode");
  ENTER(x->ctx);
  return encode_blocking(x);
}

extern "C" int bwtm_index_device_data(bwtm_index* x, void** device_ptr, uint64_t* nbytes)
{
  if(!x || !device_ptr || !nbytes) { return fail(BWTM_EINVAL, "bwtm_index_device_data: null argument"); }
  ENTER(x->ctx);
  if(!x->has_native) { return fail(BWTM_EINVAL, "index has no native byte stream (call bwtm_index_encode first)"); }
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  *device_ptr = (void*)x->native_bytes(); *nbytes = x->nbytes;
  return BWTM_OK;
}

extern "C" int bwtm_index_download_data(bwtm_index* x, uint8_t* out, uint64_t capacity)
{
  if(!x) { return fail(BWTM_EINVAL, "null index"); }
  ENTER(x->ctx);
  if(!x->has_native) { return fail(BWTM_EINVAL, "index has no native byte stream (call bwtm_index_encode first)"); }
  if(capacity < x->nbytes) { return fail(BWTM_EINVAL, "buffer too small: %llu < %llu", (unsigned long long)capacity, (unsigned long long)x->nbytes); }
  if(x->nbytes > 0) { HIP_TRY(hipMemcpyAsync(out, x->native_bytes(), x->nbytes, hipMemcpyDeviceToHost, CTX.stream)); }
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" int bwtm_index_download_samples(bwtm_index* x, uint64_t* block_end, uint64_t* cum)
{
  if(!x) { return fail(BWTM_EINVAL, "null index"); }
  ENTER(x->ctx);
  if(!x->has_native) { return fail(BWTM_EINVAL, "index has no native samples (call bwtm_index_encode first)"); }
  return download_samples(x, block_end, cum);
}

extern "C" int bwtm_index_samples_width(bwtm_index* x, int* width)
{
  if(!x || !width) { return fail(BWTM_EINVAL, "bwtm_index_samples_width: null argument"); }
  ENTER(x->ctx);
  if(!x->has_native) { return fail(BWTM_EINVAL, "index has no native samples (call bwtm_index_encode first)"); }
  return samples_width(x, width);
}

extern "C" int bwtm_index_download_samples_compact(bwtm_index* x, int width, void* fields, uint64_t* anchors)
{
  if(!x) { return fail(BWTM_EINVAL, "null index"); }
  ENTER(x->ctx);
  if(!x->has_native) { return fail(BWTM_EINVAL, "index has no native samples (call bwtm_index_encode first)"); }
  if(x->nblocks > 0 && (!fields || !anchors)) { return fail(BWTM_EINVAL, "bwtm_index_download_samples_compact: null argument"); }
  int need = 0; TRY(samples_width(x, &need));
  if(width != 1 && width != 2 && width != 4) { return fail(BWTM_EINVAL, "bwtm_index_download_samples_compact: width must be 1, 2 or 4"); }
  if(width < need) { return fail(BWTM_EINVAL, "bwtm_index_download_samples_compact: a block encodes too many positions for %d-byte fields (need %d)", width, need); }
  if(width == 1) { return download_samples_compact<u8>(x, (u8*)fields, anchors); }
  return (width == 2 ? download_samples_compact<unsigned short>(x, (unsigned short*)fields, anchors) : download_samples_compact<u32>(x, (u32*)fields, anchors));
}

extern "C" int bwtm_rank_batch(const bwtm_index* x, const uint64_t* positions, const uint8_t* comps, uint64_t count, uint64_t* out_ranks)
{
  if(!x || !positions || !comps || !out_ranks) { return fail(BWTM_EINVAL, "bwtm_rank_batch: null argument"); }
  WHOLE_INDEX(x, "bwtm_rank_batch");
  ENTER(x->ctx);
  if(count == 0) { return BWTM_OK; }
  DevBuf dp, dc, dr;
  TRY(dp.alloc(count * 8)); TRY(dc.alloc(count)); TRY(dr.alloc(count * 8));
  HIP_TRY(hipMemcpyAsync(dp.p, positions, count * 8, hipMemcpyHostToDevice, CTX.stream));
  HIP_TRY(hipMemcpyAsync(dc.p, comps, count, hipMemcpyHostToDevice, CTX.stream));
  LAUNCH("rank_batch", k_rank_batch, div_up(count, BLOCK_THREADS), BLOCK_THREADS, x->view(), dp.as<const u64>(), dc.as<const u8>(), count, dr.as<u64>());
  HIP_TRY(hipMemcpyAsync(out_ranks, dr.p, count * 8, hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" int bwtm_inverse_select_batch(const bwtm_index* x, const uint64_t* positions, uint64_t count, uint64_t* out_ranks, uint8_t* out_comps)
{
  if(!x || !positions || !out_ranks || !out_comps) { return fail(BWTM_EINVAL, "bwtm_inverse_select_batch: null argument"); }
  WHOLE_INDEX(x, "bwtm_inverse_select_batch");
  ENTER(x->ctx);
  if(count == 0) { return BWTM_OK; }
  DevBuf dp, dc, dr;
  TRY(dp.alloc(count * 8)); TRY(dc.alloc(count)); TRY(dr.alloc(count * 8));
  HIP_TRY(hipMemcpyAsync(dp.p, positions, count * 8, hipMemcpyHostToDevice, CTX.stream));
  LAUNCH("inverse_select_batch", k_inverse_select_batch, div_up(count, BLOCK_THREADS), BLOCK_THREADS, x->view(), dp.as<const u64>(), count, dr.as<u64>(), dc.as<u8>());
  HIP_TRY(hipMemcpyAsync(out_ranks, dr.p, count * 8, hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipMemcpyAsync(out_comps, dc.p, count, hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" int bwtm_find_batch(const bwtm_index* x, const uint8_t* patterns, const uint64_t* offsets, uint64_t count, uint64_t* out_sp, uint64_t* out_ep)
{
  if(!x || !offsets || !out_sp || !out_ep) { return fail(BWTM_EINVAL, "bwtm_find_batch: null argument"); }
  WHOLE_INDEX(x, "bwtm_find_batch");
  ENTER(x->ctx);
  if(count == 0) { return BWTM_OK; }
  u64 total = offsets[count];
  if(total > 0 && !patterns) { return fail(BWTM_EINVAL, "bwtm_find_batch: null pattern text"); }
  DevBuf dt, doff, dsp, dep;
  TRY(dt.alloc(total + 16)); TRY(doff.alloc((count + 1) * 8)); TRY(dsp.alloc(count * 8)); TRY(dep.alloc(count * 8));
  if(total > 0) { HIP_TRY(hipMemcpyAsync(dt.p, patterns, total, hipMemcpyHostToDevice, CTX.stream)); }
  HIP_TRY(hipMemcpyAsync(doff.p, offsets, (count + 1) * 8, hipMemcpyHostToDevice, CTX.stream));
  LAUNCH("find_batch", k_find_batch, div_up(count, BLOCK_THREADS), BLOCK_THREADS, x->view(), dt.as<const u8>(), doff.as<const u64>(), count, dsp.as<u64>(), dep.as<u64>());
  HIP_TRY(hipMemcpyAsync(out_sp, dsp.p, count * 8, hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipMemcpyAsync(out_ep, dep.p, count * 8, hipMemcpyDeviceToHost, CTX.stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" int bwtm_extract(const bwtm_index* x, uint64_t first, uint64_t count, uint8_t* out)
{
  if(!x || !out) { return fail(BWTM_EINVAL, "bwtm_extract: null argument"); }
  WHOLE_INDEX(x, "bwtm_extract");
  ENTER(x->ctx);
  if(first + count > x->n) { return fail(BWTM_EINVAL, "bwtm_extract: range past the end"); }   // bwt.h:137
  if(count == 0) { return BWTM_OK; }
  // one thread per position, and a HIP grid holds fewer than 2^32 threads: pieces of 2^30 positions (round 5: a single launch over 5.05 * 10^9
  // positions returned zeros behind the first 2^32 -- tools/scale_cli.sh wrote config 2's inputs that way and merged endmarkers)
  const u64 piece = 1ull << 30;
  DevBuf d; TRY(d.alloc(std::min(count, piece)));
  for(u64 done = 0; done < count; done += piece)
  {
    const u64 n = std::min(piece, count - done);
    LAUNCH("extract", k_extract, div_up(n, BLOCK_THREADS), BLOCK_THREADS, x->view(), first + done, n, d.as<u8>());
    HIP_TRY(hipMemcpyAsync(out + done, d.p, n, hipMemcpyDeviceToHost, CTX.stream));
    HIP_TRY(hipStreamSynchronize(CTX.stream));                     // the piece's buffer is reused
  }
  return BWTM_OK;
}

/*
  api/context.hip.h -- contexts (device + streams + memory pool), errors, launch macros, device buffers,
  per-kernel profiling, generic scans.  Part of bwtm_api.hip.
*/
#pragma once

//------------------------------------------------------------------------------
// Errors.

namespace
{

thread_local std::string g_error;

int fail(int code, const char* fmt, ...)
{
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  g_error = buf;
  return code;
}

inline u64 div_up(u64 a, u64 b) { return (a + b - 1) / b; }

} // namespace

//------------------------------------------------------------------------------
// Context.  One HIP device, a compute stream (all kernels), a copy stream (chunked H2D / D2H that
// overlaps with kernels) and a memory pool.  All work that touches pooled buffers is ordered on the
// compute stream, so a block released by a handle may be handed to the next allocation immediately
// (stream order protects it); work on the copy stream is always joined back into the compute stream
// (or the host) before the buffers it touches are released.  Blocks return to the driver only in
// bwtm_trim() or when hipMalloc runs out of memory: a repeated merge of the same shape performs no
// hipMalloc / hipFree at all (both cost milliseconds per GB and serialise with the device).

struct bwtm_context
{
  int device = -1;
  hipStream_t stream = nullptr;            // compute
  hipStream_t copy_stream = nullptr;       // H2D / D2H
  std::recursive_mutex mu;                 // calls on one context are serialized
  bool is_default = false;
  std::atomic<long long> live_handles{0};  // indexes, rank arrays and slices that live in this context

  // per-kernel profiling (bwtm_profile_*)
  bool profiling = false;
  std::string profile_only;                // when not empty: only launches of this name are bracketed (bwtm_profile_only)
  bool profile_open = false;               // the launch in progress is bracketed
  struct Pending { const char* name; hipEvent_t start, stop; };
  std::vector<Pending> pending;
  std::map<std::string, std::pair<double, uint64_t>> totals;
  std::vector<const char*> order;

  // memory pool: released blocks by size (exact-size reuse is the fast path)
  std::multimap<u64, void*> free_blocks;
  u64 cached_bytes = 0, held_bytes = 0, peak_bytes = 0;     // held = physical memory obtained from the driver and not yet returned
  // large blocks live in a reserved virtual address range and are backed by pooled physical chunks (see the pool below)
  struct VBlock { u64 bytes = 0; std::vector<hipMemGenericAllocationHandle_t> chunks; hipEvent_t released = nullptr; };
  bool vmm = false;
  std::vector<hipMemGenericAllocationHandle_t> free_chunks; // physical chunks that are not mapped anywhere
  std::map<void*, VBlock> vblocks;                          // every mapped block, in use or released
  u64 device_total = 0;
  u64 va_fallbacks = 0;                                     // large blocks that came from hipMalloc because no address range was left

  // small page-locked scratch for results read back by the host (a pageable destination would make
  // hipMemcpyAsync stage and block)
  u64* host_scratch = nullptr;             // 128 u64: [0, 32) call results, [32, 64) upload / encode / slices, [64, 96) the search's size ring
  // The one-launch scans (k_frontier_scan1, k_pull_scan1) let a tile wait for the tiles before it: all their workgroups must be resident at the
  // same time.  The limits come from THIS device (CUs x workgroups of the kernel per CU, capped at FRONTIER_SCAN1_TILES); larger tables take the
  // two-launch form.  (A device that is shared with other processes delays such a kernel, it cannot deadlock it: the others' kernels end.)
  u64 scan1_tiles = 0, pull_scan1_tiles = 0;
};

namespace
{

// Knobs (bwtm_tune): process-wide, read at the start of a call.  None of the product knobs changes results.
struct Tuning
{
  long long search_algo = 0;      // 0 = by size (frontier search for large shards, per-chain walk for small ones), 1 = walk, 2 = frontier
  long long frontier_unfused = 0; // 0 = scan + bookkeeping in one launch per step; 1 = generic scan + k_frontier_prep (the path of segment tables with > 8192 tiles); 2 = two launches (rounds 2 - 4)
  long long l1_cap = 0;           // tests: entries per level-1 region / emit capacity (0 = sized from the input): forces the exact fallbacks
  long long emit_path = 0;        // 0 = partitioned emit (default), 1 = atomicOr on the bitvector (the exact fallback, first version)
  long long round_emits = 1ll << 33;      // upper bound of emits partitioned per round of the walk (bounds the temporary regions)
  long long emit_budget = 0;              // bytes of dense emits the frontier search keeps before it builds tiles (one epoch); 0 = half of the free memory, 16 - 64 GB
  long long frontier_epoch = 512;         // upper bound of steps per epoch (tests use small values)
  long long eager_cum_budget = 16ll << 30; // bwtm_index_encode materializes the samples' cumulative counts when they take at most this many bytes
  long long upload_chunk = 256ll << 20;   // bytes per H2D chunk of the pipelined upload (64 MiB: 141.8 ms for 7.64 GB, 256 MiB and 1 GiB: 140.3)
  long long download_chunk = 128ll << 20; // approximate bytes per D2H chunk of the pipelined download
  long long range_ratio = 8;              // frontier search: levels with at most (sequences / range_ratio) trie nodes are processed as nodes (k_range_*); 0 = never, 1 = as long as possible
#ifdef BWTM_EXPERIMENTAL
  long long search_view = 0;              // the frontier search reads the two-plane search view: 0 = never (default: it saves 14 % of the HBM reads and no time), 1 = always, 2 = by size
#endif
  long long frontier_parts = 0;           // > 1: every step of the frontier search as this many launches over slices of the frontier (a measurement, same results)
  long long part_capacity = 0;            // tests: elements a part of the partitioned merge can hold in a step (0 = 2 m / parts + slack): forces the out-of-room
                                          // paths; negative: only the element steps are held to |value| (the roots and the expansion of the node levels are not)
  long long recs_uniform = 0;             // k_build_recs: the straight-line deposit for streams of longer runs: 1 always, -1 never, 0 = by the stream's density
  long long recs_window = 0;              // k_build_recs: positions per LDS window (8192 / 16384 / 32768); 0 = by the stream's density (A/B measurements)
  long long ingest_verify = 0;            // 1 = the builder checks every leaf's suffix order against the reads (one extra pass of gathers per leaf)
#ifdef BWTM_DIAGNOSTICS
  long long walk_emit = 0;       // 0 = real emit; 1 / 2 timing-only variants of the emit (see diagnostics.hip.h)
  long long walk_blocks = 0;     // grid size override for the walk kernels (0 = default)
  long long walk_kernel = 0;     // 0 = four lanes per chain (default), 1 = one lane per chain (first version)
  long long walk_ablate = 0;     // timing-only ablations of the no-emit quad kernel (tools/walk_experiments.py)
  long long walk_variant = 0;    // 1 = LDS-transposed one chain per lane
  long long scatter_kernel = 0;  // 1 = direct scattered stores in level 2 of the partition (first version)
#endif
};
Tuning g_tune;

std::mutex g_registry_mu;
std::map<int, bwtm_context*> g_default_ctx;          // device -> default context
thread_local bwtm_context* t_bound = nullptr;        // the calling thread's context (bwtm_init / make_current)
thread_local bwtm_context* t_ctx = nullptr;          // the context of the API call in progress

#define CTX (*t_ctx)

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess) { \
  return fail(e_ == hipErrorOutOfMemory ? BWTM_ENOMEM : BWTM_ENODEV, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } } while(0)

#define TRY(expr) do { int rc_ = (expr); if(rc_ != BWTM_OK) { return rc_; } } while(0)

void vmm_setup(bwtm_context* c);
void pool_trim(bwtm_context* c);
void apply_env_tuning();

int context_setup(bwtm_context* c, int device)
{
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if(e != hipSuccess || count <= 0) { return fail(BWTM_ENODEV, "no HIP device available (%s)", hipGetErrorString(e)); }
  if(device < 0 || device >= count) { return fail(BWTM_EINVAL, "device %d out of range (%d devices)", device, count); }
  HIP_TRY(hipSetDevice(device));
  apply_env_tuning();
  c->device = device;
  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  HIP_TRY(hipHostMalloc((void**)&c->host_scratch, 128 * sizeof(u64), hipHostMallocDefault));
  vmm_setup(c);
  // Kernels that take more than the default 64 KiB of dynamic LDS (a per-device attribute).
  {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    int per_cu_scan = 0, per_cu_pull = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_scan, (const void*)k_frontier_scan1, BLOCK_THREADS, 0));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_pull, (const void*)k_pull_scan1, BLOCK_THREADS, 0));
    c->scan1_tiles = std::min<u64>(FRONTIER_SCAN1_TILES, (u64)std::max(per_cu_scan, 0) * (u64)std::max(prop.multiProcessorCount, 0));
    c->pull_scan1_tiles = std::min<u64>(FRONTIER_SCAN1_TILES, (u64)std::max(per_cu_pull, 0) * (u64)std::max(prop.multiProcessorCount, 0));
  }
  HIP_TRY(hipFuncSetAttribute((const void*)k_part_scatter_sorted, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIP_TRY(hipFuncSetAttribute((const void*)k_lf_walk_binned<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024));
#ifdef BWTM_DIAGNOSTICS
  HIP_TRY(hipFuncSetAttribute((const void*)k_lf_walk_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 122 * 1024));
  HIP_TRY(hipFuncSetAttribute((const void*)k_lf_walk_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 82 * 1024));
#endif
  return BWTM_OK;
}

void context_teardown(bwtm_context* c)
{
  if(c->device < 0) { return; }
  (void)hipSetDevice(c->device);
  if(c->stream) { (void)hipStreamSynchronize(c->stream); }
  if(c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); }
  for(auto& p : c->pending) { (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop); }
  c->pending.clear();
  pool_trim(c);
  if(c->host_scratch) { (void)hipHostFree(c->host_scratch); }
  if(c->stream) { (void)hipStreamDestroy(c->stream); }
  if(c->copy_stream) { (void)hipStreamDestroy(c->copy_stream); }
}

// The default context of a device (created on first use).
int default_context(int device, bwtm_context** out)
{
  std::lock_guard<std::mutex> lock(g_registry_mu);
  auto it = g_default_ctx.find(device);
  if(it != g_default_ctx.end()) { *out = it->second; return BWTM_OK; }
  bwtm_context* c = new bwtm_context();
  c->is_default = true;
  int rc = context_setup(c, device);
  if(rc != BWTM_OK) { context_teardown(c); delete c; return rc; }
  g_default_ctx[device] = c;
  *out = c;
  return BWTM_OK;
}

// Every entry point runs inside a Scope: it resolves the context (the handle's, or the thread's), takes its
// lock, makes its device current for the calling thread and publishes it as t_ctx for the helpers below.
struct Scope
{
  bwtm_context* prev;
  bwtm_context* ctx = nullptr;
  int rc = BWTM_OK;
  explicit Scope(bwtm_context* wanted)
  {
    prev = t_ctx;
    if(!wanted) { wanted = t_bound; }
    if(!wanted) { rc = default_context(0, &wanted); if(rc != BWTM_OK) { return; } }
    ctx = wanted;
    ctx->mu.lock();
    hipError_t e = hipSetDevice(ctx->device);
    if(e != hipSuccess) { rc = fail(BWTM_ENODEV, "hipSetDevice(%d) failed: %s", ctx->device, hipGetErrorString(e)); }
    t_ctx = ctx;
  }
  ~Scope()
  {
    if(ctx)
    {
      t_ctx = prev;
      // a nested scope may have switched the thread to another device: the rest of the outer call must run on its own
      if(prev && prev->device != ctx->device) { (void)hipSetDevice(prev->device); }
      ctx->mu.unlock();
    }
  }
  Scope(const Scope&) = delete; Scope& operator=(const Scope&) = delete;
};

#define ENTER(wanted) Scope scope_(wanted); if(scope_.rc != BWTM_OK) { return scope_.rc; }

//------------------------------------------------------------------------------
// Profiling: every launch bracketed by events on the compute stream (only when enabled).

void profile_begin(const char* name)
{
  CTX.profile_open = (CTX.profiling && (CTX.profile_only.empty() || CTX.profile_only.find(std::string(",") + name + ",") != std::string::npos));
  if(!CTX.profile_open) { return; }
  bwtm_context::Pending p; p.name = name;
  (void)hipEventCreate(&p.start); (void)hipEventCreate(&p.stop);
  (void)hipEventRecord(p.start, CTX.stream);
  CTX.pending.push_back(p);
}

void profile_end()
{
  if(!CTX.profile_open) { return; }
  (void)hipEventRecord(CTX.pending.back().stop, CTX.stream);
}

void profile_collect()
{
  if(CTX.pending.empty()) { return; }
  (void)hipStreamSynchronize(CTX.stream);
  for(auto& p : CTX.pending)
  {
    float ms = 0; (void)hipEventElapsedTime(&ms, p.start, p.stop);
    auto it = CTX.totals.find(p.name);
    if(it == CTX.totals.end()) { CTX.totals[p.name] = std::make_pair((double)ms, (uint64_t)1); CTX.order.push_back(p.name); }
    else { it->second.first += ms; it->second.second += 1; }
    (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop);
  }
  CTX.pending.clear();
}

// A HIP grid holds fewer than 2^32 threads in x, and a launch beyond that does not fail: it covers a part of the work (round 5: bwtm_extract of
// 5 * 10^9 positions returned zeros behind the first 2^32).  The grid is checked as a 64-bit number before it is narrowed, and refused loudly.
#define LAUNCH_CFG(name, kernel, gridx, gridy, block, lds, ...) do { \
  const unsigned long long gx_ = (unsigned long long)(gridx); \
  if(gx_ * (unsigned long long)(block) >= (1ull << 32)) { return fail(BWTM_EINVAL, "launch of %s: %llu workgroups of %u threads do not fit one grid", name, gx_, (unsigned)(block)); } \
  profile_begin(name); \
  hipLaunchKernelGGL(kernel, dim3((unsigned)gx_, (unsigned)(gridy)), dim3((unsigned)(block)), (unsigned)(lds), CTX.stream, __VA_ARGS__); \
  profile_end(); \
  hipError_t le_ = hipGetLastError(); \
  if(le_ != hipSuccess) { return fail(BWTM_ENODEV, "launch of %s failed: %s", name, hipGetErrorString(le_)); } } while(0)

#define LAUNCH(name, kernel, grid, block, ...) LAUNCH_CFG(name, kernel, grid, 1, block, 0, __VA_ARGS__)
#define LAUNCH_LDS(name, kernel, grid, block, lds, ...) LAUNCH_CFG(name, kernel, grid, 1, block, lds, __VA_ARGS__)
#define LAUNCH2D(name, kernel, gridx, gridy, block, ...) LAUNCH_CFG(name, kernel, gridx, gridy, block, 0, __VA_ARGS__)

//------------------------------------------------------------------------------
// Pool.
//
// Small blocks come from hipMalloc and are reused by size.  Blocks of VMM_MIN bytes and more live in a virtual address
// range reserved once per context and are backed by physical chunks of VMM_CHUNK bytes (hipMemCreate / hipMemMap): a
// released block keeps its mapping and is handed out again when a block of (nearly) the same size is asked for -- the
// steady state of a repeated merge, no driver call at all -- but when nothing fits, the chunks of released blocks of ANY
// size are unmapped and mapped into the new block.  Memory therefore moves between buffers of different sizes (the
// 38 GB native inputs of one phase become part of the 76 GB native output of a later one) without hipFree / hipMalloc:
// on MI355X a hipMalloc that has to wait for deferred frees takes seconds (measured at 2 x 50 Gbase: 4.7 s of a 6.1 s
// merge), mapping 32 GiB of pooled chunks takes about a millisecond (tools/microbench_vmm.hip).
// A released block may still be in use by kernels queued before its release; exact-size reuse relies on stream order
// (all work runs on the context's compute stream), unmapping waits for the event recorded at the release.
// ADDRESSES ARE NEVER REUSED: on this ROCm release a virtual address that is unmapped and mapped again onto other physical
// memory keeps its old translation in the shader TLBs (tools/vmm_stress.hip: kernels write through the stale mapping while
// copy engines see the new one; neither hipDeviceSynchronize nor freeing and re-reserving the range helps), whereas fresh
// addresses always work.  Blocks therefore take their addresses from a process-wide bump allocator over address ranges
// reserved 8 TiB at a time (64 TiB can be reserved); when that is exhausted -- after the order of a hundred merges of
// 2 x 50 Gbase, which consume addresses on every recycling -- the pool falls back to plain hipMalloc blocks.

// Chunk size and the smallest block that takes this path; BWTM_POOL_VMM_CHUNK / BWTM_POOL_VMM_MIN (bytes) override them
// (the test-suite runs once with small values so that every buffer of every test goes through map / unmap).
// 1-GiB chunks since round 4: at 2 x 50 Gbase a merge remaps ~300 GB of blocks (inputs' records -> result records -> native result ...), and
// with 128-MiB chunks the map / unmap calls took 45 ms per merge with the GPU idle (a 76.5 GB block: 14.5 ms); with 1-GiB chunks 7 ms
// (2.5 ms), 1202 -> 1149 ms per merge.  Blocks between VMM_MIN and 1 GiB round up to a whole chunk.
u64 VMM_CHUNK = 1024ull << 20;
u64 VMM_MIN = 512ull << 20;        // (128 MiB until round 5: a 130-MiB block then took a whole 1-GiB chunk; blocks below half a chunk now come from hipMalloc and are reused by size)

u64 pool_round(u64 n)
{
  if(n < 256) { n = 256; }
  u64 g = (n >= (64ull << 20) ? (2ull << 20) : (n >= (1ull << 20) ? (64ull << 10) : 256ull));
  return (n + g - 1) / g * g;
}

// BWTM_TRACE=1 in the environment: slow paths of the pool report to stderr.
bool trace_enabled()
{
  static const bool on = (std::getenv("BWTM_TRACE") != nullptr);
  return on;
}

double trace_now()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct AddressSpace
{
  std::mutex mu;
  char* segment = nullptr; u64 segment_size = 0, used = 0;
  u64 reserved_total = 0;
  bool exhausted = false;
};
AddressSpace g_va;
u64 VA_SEGMENT = 8ull << 40, VA_LIMIT = 64ull << 40;        // BWTM_POOL_VA_SEGMENT / BWTM_POOL_VA_LIMIT (bytes) override them: the tests exhaust a small range

void vmm_setup(bwtm_context* c)
{
  const char* env = std::getenv("BWTM_POOL_VMM");
  if(env && env[0] == '0') { return; }
  if(const char* v = std::getenv("BWTM_POOL_VMM_CHUNK")) { u64 x = std::strtoull(v, nullptr, 10); if(x >= (2ull << 20)) { VMM_CHUNK = x / (2ull << 20) * (2ull << 20); } }
  if(const char* v = std::getenv("BWTM_POOL_VMM_MIN")) { u64 x = std::strtoull(v, nullptr, 10); if(x >= 4096) { VMM_MIN = x; } }
  if(const char* v = std::getenv("BWTM_POOL_VA_SEGMENT")) { u64 x = std::strtoull(v, nullptr, 10); if(x >= VMM_CHUNK) { VA_SEGMENT = x / VMM_CHUNK * VMM_CHUNK; } }
  if(const char* v = std::getenv("BWTM_POOL_VA_LIMIT")) { u64 x = std::strtoull(v, nullptr, 10); if(x >= VMM_CHUNK) { VA_LIMIT = x; } }
  size_t free_b = 0, total_b = 0;
  if(hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
  c->device_total = total_b;
  c->vmm = true;
}

// A fresh address range of `bytes` (a multiple of the chunk size); nullptr when no more addresses can be had.
char* va_take(u64 bytes)
{
  std::lock_guard<std::mutex> lock(g_va.mu);
  if(g_va.exhausted) { return nullptr; }
  if(!g_va.segment || g_va.used + bytes > g_va.segment_size)
  {
    // the rest of the old segment is abandoned; reservations are never freed (a freed range comes back at the same address)
    u64 want = std::max(VA_SEGMENT, bytes);
    void* va = nullptr;
    if(g_va.reserved_total + want > VA_LIMIT || hipMemAddressReserve(&va, want, VMM_CHUNK, nullptr, 0) != hipSuccess)
    {
      (void)hipGetLastError();
      g_va.exhausted = true;
      if(trace_enabled()) { fprintf(stderr, "[bwtm] address space for mapped blocks exhausted after %.1f TiB: falling back to hipMalloc blocks\n", g_va.reserved_total / 1099511627776.0); }
      return nullptr;
    }
    g_va.segment = (char*)va; g_va.segment_size = want; g_va.used = 0; g_va.reserved_total += want;
  }
  char* p = g_va.segment + g_va.used;
  g_va.used += bytes;
  return p;
}

// Unmaps a released block and returns its chunks to the chunk pool (the caller has made sure the GPU is done with it).
void vmm_unmap(bwtm_context* c, void* p)
{
  auto it = c->vblocks.find(p);
  if(it == c->vblocks.end()) { return; }
  (void)hipMemUnmap(p, it->second.bytes);
  for(auto h : it->second.chunks) { c->free_chunks.push_back(h); }
  if(it->second.released) { (void)hipEventDestroy(it->second.released); }
  c->vblocks.erase(it);                                       // the address range is not used again
}

// Moves the chunks of released blocks into the chunk pool until it holds `need` chunks.  wait = false: only blocks whose
// release event has completed; wait = true: synchronises the compute stream first (then all of them have).
void vmm_harvest(bwtm_context* c, u64 need, bool wait)
{
  if(c->free_chunks.size() >= need) { return; }
  if(wait && c->stream) { (void)hipStreamSynchronize(c->stream); }
  for(auto it = c->free_blocks.end(); it != c->free_blocks.begin() && c->free_chunks.size() < need; )     // largest first
  {
    --it;
    auto vb = c->vblocks.find(it->second);
    if(vb == c->vblocks.end()) { continue; }                  // a small hipMalloc block
    // without an event (its creation failed) nothing proves that the GPU is done with the block: only a drained stream does
    if(!wait && (!vb->second.released || hipEventQuery(vb->second.released) != hipSuccess)) { (void)hipGetLastError(); continue; }
    void* p = it->second;
    c->cached_bytes -= it->first;
    it = c->free_blocks.erase(it);
    vmm_unmap(c, p);
  }
}

hipError_t vmm_alloc(bwtm_context* c, u64 n, void** p)
{
  const u64 need = n / VMM_CHUNK;
  const double t0 = trace_now();
  u64 created = 0;
  // Below 40 % of the device's memory the pool simply grows; above it, idle blocks are recycled before new chunks are created.
  const bool recycle_first = (c->held_bytes + n > c->device_total / 10 * 4);
  if(recycle_first) { vmm_harvest(c, need, false); }
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = c->device;
  while(c->free_chunks.size() < need)
  {
    hipMemGenericAllocationHandle_t h;
    if(hipMemCreate(&h, VMM_CHUNK, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
    c->free_chunks.push_back(h); created++;
    c->held_bytes += VMM_CHUNK; if(c->held_bytes > c->peak_bytes) { c->peak_bytes = c->held_bytes; }
  }
  if(c->free_chunks.size() < need) { vmm_harvest(c, need, false); }
  if(c->free_chunks.size() < need) { vmm_harvest(c, need, true); }
  if(c->free_chunks.size() < need) { return hipErrorOutOfMemory; }
  char* base = va_take(n);
  if(!base) { return hipErrorOutOfMemory; }
  bwtm_context::VBlock vb; vb.bytes = n;
  hipError_t e = hipSuccess;
  for(u64 k = 0; k < need && e == hipSuccess; k++)
  {
    hipMemGenericAllocationHandle_t h = c->free_chunks.back();
    e = hipMemMap(base + k * VMM_CHUNK, VMM_CHUNK, 0, h, 0);
    if(e == hipSuccess) { c->free_chunks.pop_back(); vb.chunks.push_back(h); }
  }
  if(e == hipSuccess)
  {
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = c->device; acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(base, n, &acc, 1);
  }
  if(e != hipSuccess)
  {
    (void)hipGetLastError();
    if(!vb.chunks.empty()) { (void)hipMemUnmap(base, vb.chunks.size() * VMM_CHUNK); }
    for(auto h : vb.chunks) { c->free_chunks.push_back(h); }
    return e;
  }
  c->vblocks[base] = std::move(vb);
  *p = base;
  if(trace_enabled() && (trace_now() - t0 > 1.0 || n >= (4ull << 30)))
  {
    fprintf(stderr, "[bwtm] map %.2f GB: %.2f ms (%llu new chunks; held %.1f GB, cached %.1f GB)\n", n / 1e9, trace_now() - t0, (unsigned long long)created,
      c->held_bytes / 1e9, c->cached_bytes / 1e9);
  }
  return hipSuccess;
}

void pool_trim(bwtm_context* c)
{
  const double t0 = trace_now();
  if(c->stream) { (void)hipStreamSynchronize(c->stream); }
  const u64 bytes = c->cached_bytes; const size_t blocks = c->free_blocks.size();
  for(auto& kv : c->free_blocks)
  {
    if(c->vblocks.count(kv.second)) { vmm_unmap(c, kv.second); }
    else { (void)hipFree(kv.second); c->held_bytes -= kv.first; }
  }
  c->free_blocks.clear(); c->cached_bytes = 0;
  for(auto h : c->free_chunks) { (void)hipMemRelease(h); c->held_bytes -= VMM_CHUNK; }
  c->free_chunks.clear();
  if(trace_enabled()) { fprintf(stderr, "[bwtm] trim: %zu blocks, %.2f GB, %.1f ms\n", blocks, bytes / 1e9, trace_now() - t0); }
}

hipError_t pool_get(bwtm_context* c, u64 n, void** p, u64* actual)
{
  n = pool_round(n);
  const bool large = (c->vmm && n >= VMM_MIN);
  if(large) { n = (n + VMM_CHUNK - 1) / VMM_CHUNK * VMM_CHUNK; }
  auto it = c->free_blocks.lower_bound(n);
  if(it != c->free_blocks.end() && it->first <= n + n / 8 && (!c->vmm || (c->vblocks.count(it->second) != 0) == large))
  {
    *p = it->second; *actual = it->first; c->cached_bytes -= it->first; c->free_blocks.erase(it);
    return hipSuccess;
  }
  *actual = n;
  if(large)
  {
    hipError_t ve = vmm_alloc(c, n, p);
    if(ve == hipErrorOutOfMemory && !g_va.exhausted && !c->free_blocks.empty())
    {
      pool_trim(c);                       // cached small hipMalloc blocks (and idle chunks) go back to the driver, then once more
      ve = vmm_alloc(c, n, p);
    }
    if(ve == hipSuccess || !g_va.exhausted) { return ve; }
    // no addresses left for mapped blocks: give the pooled chunks back to the driver and go on with hipMalloc blocks
    pool_trim(c);
    c->vmm = false;
  }
  hipError_t e = hipMalloc(p, n);
  if(e != hipSuccess) { (void)hipGetLastError(); pool_trim(c); e = hipMalloc(p, n); }
  if(e == hipSuccess) { c->held_bytes += n; if(c->held_bytes > c->peak_bytes) { c->peak_bytes = c->held_bytes; } }
  if(e == hipSuccess && g_va.exhausted && n >= VMM_MIN) { c->va_fallbacks++; }      // a large block that should have been a mapped one
  return e;
}

void pool_put(bwtm_context* c, void* p, u64 n)
{
  auto vb = c->vblocks.find(p);
  if(vb != c->vblocks.end())
  {
    // kernels queued before this point may still use the block: whoever unmaps it waits for this event
    if(!vb->second.released) { (void)hipEventCreateWithFlags(&vb->second.released, hipEventDisableTiming); }
    if(vb->second.released) { (void)hipEventRecord(vb->second.released, c->stream); }
  }
  c->free_blocks.insert(std::make_pair(n, p)); c->cached_bytes += n;
}

// RAII device buffer (pooled).  Released into the pool of the context it came from; the release must happen
// inside a Scope of that context (handles enter their own context before they are destroyed).
struct DevBuf
{
  void* p = nullptr; u64 bytes = 0; bwtm_context* owner = nullptr;
  DevBuf() {}
  DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() { if(p) { pool_put(owner, p, bytes); p = nullptr; bytes = 0; owner = nullptr; } }
  int alloc(u64 n, bool zero = false)
  {
    release();
    if(n == 0) { n = 8; }
    hipError_t e = pool_get(t_ctx, n, &p, &bytes);
    if(e != hipSuccess) { p = nullptr; bytes = 0; return fail(BWTM_ENOMEM, "hipMalloc(%llu bytes) failed: %s", (unsigned long long)n, hipGetErrorString(e)); }
    owner = t_ctx;
    if(zero) { e = hipMemsetAsync(p, 0, n, CTX.stream); if(e != hipSuccess) { return fail(BWTM_ENODEV, "hipMemsetAsync failed: %s", hipGetErrorString(e)); } }
    return BWTM_OK;
  }
  template<class T> T* as() const { return (T*)p; }
  void swap(DevBuf& o) { std::swap(p, o.p); std::swap(bytes, o.bytes); std::swap(owner, o.owner); }
};

// Small results the host needs: copied into the context's page-locked scratch (slot..slot+count-1), valid after the next
// synchronisation of the compute stream.
int fetch_u64(const u64* device_src, u32 slot, u32 count = 1)
{
  HIP_TRY(hipMemcpyAsync(CTX.host_scratch + slot, device_src, count * sizeof(u64), hipMemcpyDeviceToHost, CTX.stream));
  return BWTM_OK;
}

// The copy stream waits for everything queued so far on the compute stream.
int fork_copy_stream()
{
  hipEvent_t ev;
  HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, CTX.stream);
  if(e == hipSuccess) { e = hipStreamWaitEvent(CTX.copy_stream, ev, 0); }
  (void)hipEventDestroy(ev);
  if(e != hipSuccess) { return fail(BWTM_ENODEV, "forking the copy stream failed: %s", hipGetErrorString(e)); }
  return BWTM_OK;
}

//------------------------------------------------------------------------------
// Exclusive scan of `narrays` arrays of n u64 items each, `stride` items apart (in place allowed).
// OP 0 = sum, 1 = max.

template<int OP>
int device_scan_multi(const u64* in, u64* out, u64 n, u64 narrays, u64 stride)
{
  if(n == 0 || narrays == 0) { return BWTM_OK; }
  u64 tiles = div_up(n, SCAN_TILE);
  if(tiles == 1)
  {
    LAUNCH2D("scan_apply", k_scan_apply<OP>, 1, narrays, BLOCK_THREADS, in, out, (const u64*)nullptr, n, stride, (u64)0);
    return BWTM_OK;
  }
  DevBuf partial; TRY(partial.alloc(tiles * narrays * sizeof(u64)));
  LAUNCH2D("scan_reduce", k_scan_reduce<OP>, tiles, narrays, BLOCK_THREADS, in, partial.as<u64>(), n, stride, tiles);
  TRY(device_scan_multi<OP>(partial.as<u64>(), partial.as<u64>(), tiles, narrays, tiles));
  LAUNCH2D("scan_apply", k_scan_apply<OP>, tiles, narrays, BLOCK_THREADS, in, out, (const u64*)partial.as<u64>(), n, stride, tiles);
  return BWTM_OK;                                   // `partial` returns to the pool (stream ordered)
}

template<int OP>
int device_scan(const u64* in, u64* out, u64 n) { return device_scan_multi<OP>(in, out, n, 1, 0); }

} // namespace

//------------------------------------------------------------------------------
// Library entry points.

extern "C" int bwtm_init(int device)
{
  bwtm_context* c = nullptr;
  TRY(default_context(device, &c));
  t_bound = c;
  return BWTM_OK;
}

extern "C" int bwtm_context_create(int device, bwtm_context** out)
{
  if(!out) { return fail(BWTM_EINVAL, "bwtm_context_create: null argument"); }
  bwtm_context* c = new bwtm_context();
  int rc = context_setup(c, device);
  if(rc != BWTM_OK) { context_teardown(c); delete c; return rc; }
  *out = c;
  return BWTM_OK;
}

extern "C" int bwtm_context_make_current(bwtm_context* context)
{
  t_bound = context;
  return BWTM_OK;
}

extern "C" void bwtm_context_destroy(bwtm_context* context)
{
  if(!context || context->is_default) { return; }
  if(t_bound == context) { t_bound = nullptr; }
  if(context->live_handles.load() > 0)
  {
    // its handles would dangle: leave the context (and their memory) alone rather than free it under them
    fail(BWTM_EINVAL, "bwtm_context_destroy: %lld handles of the context are still alive; context not destroyed", context->live_handles.load());
    return;
  }
  context_teardown(context);
  delete context;
}

extern "C" const char* bwtm_last_error(void) { return g_error.c_str(); }

namespace
{
int tune_set(const char* key, long long value)
{
  std::string k(key);
  if(k == "search_algo") { g_tune.search_algo = value; }
  else if(k == "frontier_unfused") { g_tune.frontier_unfused = value; }
  else if(k == "l1_cap") { g_tune.l1_cap = value; }
  else if(k == "emit_path") { g_tune.emit_path = value; }
  else if(k == "round_emits") { g_tune.round_emits = (value > 0 ? value : 1); }
  else if(k == "emit_budget") { g_tune.emit_budget = (value > 0 ? value : 0); }
  else if(k == "frontier_epoch") { g_tune.frontier_epoch = (value > 0 ? value : 512); }
  else if(k == "range_ratio") { g_tune.range_ratio = (value >= 0 ? value : 8); }
#ifdef BWTM_EXPERIMENTAL
  else if(k == "search_view") { g_tune.search_view = (value >= 0 && value <= 2 ? value : 0); }
#endif
  else if(k == "frontier_parts") { g_tune.frontier_parts = (value > 0 ? value : 0); }
  else if(k == "ingest_verify") { g_tune.ingest_verify = (value != 0); }
  else if(k == "part_capacity") { g_tune.part_capacity = value; }
  else if(k == "recs_uniform") { g_tune.recs_uniform = (value > 0 ? 1 : (value < 0 ? -1 : 0)); }
  else if(k == "recs_window") { g_tune.recs_window = (value == 8192 || value == 16384 || value == 32768 ? value : 0); }
  else if(k == "eager_cum_budget") { g_tune.eager_cum_budget = (value > 0 ? value : (16ll << 30)); }
  else if(k == "upload_chunk") { g_tune.upload_chunk = (value > 0 ? value : (256ll << 20)); }
  else if(k == "download_chunk") { g_tune.download_chunk = (value > 0 ? value : (128ll << 20)); }
#ifdef BWTM_DIAGNOSTICS
  else if(k == "walk_emit") { g_tune.walk_emit = value; }
  else if(k == "walk_blocks") { g_tune.walk_blocks = value; }
  else if(k == "walk_kernel") { g_tune.walk_kernel = value; }
  else if(k == "walk_ablate") { g_tune.walk_ablate = value; }
  else if(k == "walk_variant") { g_tune.walk_variant = value; }
  else if(k == "scatter_kernel") { g_tune.scatter_kernel = value; }
#endif
  else { return fail(BWTM_EINVAL, "bwtm_tune: unknown key %s", key); }
  return BWTM_OK;
}

// BWTM_TUNE="key=value,key=value" in the environment: knobs applied ONCE, before the first bwtm_tune() call takes effect or the first
// context is set up, whichever comes first -- so a bwtm_tune() call always overrides the environment, also when it is made before any
// GPU call (whole test-suites run with another default this way, e.g. BWTM_TUNE=range_ratio=1).
void apply_env_tuning()
{
  static std::mutex mu;
  static bool done = false;
  std::lock_guard<std::mutex> lock(mu);            // a concurrent caller waits until the environment has been applied
  if(done) { return; }
  done = true;
  const char* env = std::getenv("BWTM_TUNE");
  if(!env) { return; }
  std::string all(env);
  size_t pos = 0;
  while(pos < all.size())
  {
    size_t end = all.find(',', pos); if(end == std::string::npos) { end = all.size(); }
    const std::string item = all.substr(pos, end - pos);
    const size_t eq = item.find('=');
    if(eq != std::string::npos)
    {
      if(tune_set(item.substr(0, eq).c_str(), std::atoll(item.c_str() + eq + 1)) != BWTM_OK) { fprintf(stderr, "[bwtm] BWTM_TUNE: %s\n", g_error.c_str()); }
    }
    pos = end + 1;
  }
}
} // namespace

extern "C" int bwtm_tune(const char* key, long long value)
{
  if(!key) { return fail(BWTM_EINVAL, "bwtm_tune: null key"); }
  apply_env_tuning();
  return tune_set(key, value);
}

extern "C" int bwtm_trim(void)
{
  ENTER(nullptr);
  pool_trim(t_ctx);
  return BWTM_OK;
}

extern "C" int bwtm_synchronize(void)
{
  ENTER(nullptr);
  HIP_TRY(hipStreamSynchronize(CTX.copy_stream));
  HIP_TRY(hipStreamSynchronize(CTX.stream));
  return BWTM_OK;
}

extern "C" uint64_t bwtm_device_bytes_peak(int reset)
{
  Scope scope_(nullptr);
  if(scope_.rc != BWTM_OK) { return 0; }
  u64 peak = CTX.peak_bytes;
  if(reset) { CTX.peak_bytes = CTX.held_bytes; }
  return peak;
}

extern "C" int bwtm_pool_stats(bwtm_pool_info* info)
{
  if(!info) { return fail(BWTM_EINVAL, "bwtm_pool_stats: null argument"); }
  ENTER(nullptr);
  info->held_bytes = CTX.held_bytes; info->cached_bytes = CTX.cached_bytes; info->peak_bytes = CTX.peak_bytes;
  info->mapped_blocks = CTX.vblocks.size(); info->hipmalloc_fallbacks = CTX.va_fallbacks;
  {
    std::lock_guard<std::mutex> lock(g_va.mu);
    info->address_bytes_reserved = g_va.reserved_total; info->address_space_exhausted = (g_va.exhausted ? 1 : 0);
  }
  return BWTM_OK;
}

extern "C" int bwtm_host_alloc(uint64_t nbytes, void** out)
{
  ENTER(nullptr);
  if(!out) { return fail(BWTM_EINVAL, "bwtm_host_alloc: null argument"); }
  hipError_t e = hipHostMalloc(out, nbytes > 0 ? nbytes : 8, hipHostMallocDefault);
  if(e != hipSuccess) { (void)hipGetLastError(); return fail(BWTM_ENOMEM, "hipHostMalloc(%llu bytes) failed: %s", (unsigned long long)nbytes, hipGetErrorString(e)); }
  return BWTM_OK;
}

extern "C" void bwtm_host_free(void* p)
{
  if(p) { (void)hipHostFree(p); }
}

//------------------------------------------------------------------------------
// Measurement.

extern "C" int bwtm_profile_enable(int on)
{
  ENTER(nullptr);
  profile_collect();
  CTX.profiling = (on != 0);
  return BWTM_OK;
}

extern "C" int bwtm_profile_only(const char* name)
{
  ENTER(nullptr);
  profile_collect();
  CTX.profile_only = (name && name[0] ? std::string(",") + name + "," : std::string());     // a comma-separated list of names
  return BWTM_OK;
}

extern "C" int bwtm_profile_reset(void)
{
  ENTER(nullptr);
  profile_collect();
  CTX.totals.clear(); CTX.order.clear();
  return BWTM_OK;
}

extern "C" int bwtm_profile_read(const char** names, double* total_ms, uint64_t* launches, int capacity)
{
  Scope scope_(nullptr);
  if(scope_.rc != BWTM_OK) { return 0; }
  profile_collect();
  int k = 0;
  for(const char* name : CTX.order)
  {
    if(k < capacity)
    {
      auto& t = CTX.totals[name];
      names[k] = name; total_ms[k] = t.first; launches[k] = t.second;
    }
    k++;
  }
  return k;
}

/*
  bwtm_api.hip -- implementation of the C ABI declared in include/bwtm.h: host-side
  orchestration of the gfx950 kernels in bwtm_kernels.hip.h.  No CPU fallback exists: every
  entry point fails with BWTM_ENODEV when no HIP device is usable.
*/
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/bwtm.h"
#include "bwtm_kernels.hip.h"

using namespace bwtm;

//------------------------------------------------------------------------------
// Context, errors, profiling.

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

struct Context
{
  bool ready = false;
  int device = -1;
  hipStream_t stream = nullptr;
  bool profiling = false;
  struct Pending { const char* name; hipEvent_t start, stop; };
  std::vector<Pending> pending;
  std::map<std::string, std::pair<double, uint64_t>> totals;
  std::vector<const char*> order;
};

Context g_ctx;

// Diagnostic knobs (bwtm_tune): never change results unless documented as timing-only.
struct Tuning
{
  long long walk_emit = 0;       // 0 = real emit; 1 / 2 timing-only variants of the emit (see k_lf_walk)
  long long walk_blocks = 0;     // grid size override for k_lf_walk (0 = default)
  long long walk_kernel = 0;     // 0 = four lanes per chain (default), 1 = one lane per chain (first version, kept for A/B)
  long long walk_ablate = 0;     // timing-only ablations of the no-emit quad kernel (tools/walk_experiments.py)
  long long search_algo = 0;     // 0 = by size (frontier search for large shards, per-chain walk for small ones), 1 = walk, 2 = frontier
  long long frontier_unfused = 0; // 1 = generic scan + k_frontier_prep per step (the path of segment tables with > 8192 tiles)
  long long l1_cap = 0;          // tests only: entries per level-1 region (0 = sized from the input)
  long long walk_variant = 0;    // 0 = four lanes per chain, four pipelined chains per quad (default); 1 = LDS-transposed one chain per lane
  long long scatter_kernel = 0;  // 0 = LDS counting sort (default), 1 = direct scattered stores (first version)
  long long emit_path = 0;       // 0 = partitioned emit (default), 1 = atomicOr on the bitvector (first version, also the fallback)
  long long round_emits = 1ll << 33;   // upper bound of emits partitioned per round (bounds the temporary regions)
};
Tuning g_tune;

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess) { \
  return fail(e_ == hipErrorOutOfMemory ? BWTM_ENOMEM : BWTM_ENODEV, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } } while(0)

#define TRY(expr) do { int rc_ = (expr); if(rc_ != BWTM_OK) { return rc_; } } while(0)

int ensure_ready()
{
  if(g_ctx.ready) { return BWTM_OK; }
  return bwtm_init(0);
}

void profile_begin(const char* name)
{
  if(!g_ctx.profiling) { return; }
  Context::Pending p; p.name = name;
  (void)hipEventCreate(&p.start); (void)hipEventCreate(&p.stop);
  (void)hipEventRecord(p.start, g_ctx.stream);
  g_ctx.pending.push_back(p);
}

void profile_end()
{
  if(!g_ctx.profiling) { return; }
  (void)hipEventRecord(g_ctx.pending.back().stop, g_ctx.stream);
}

void profile_collect()
{
  if(g_ctx.pending.empty()) { return; }
  (void)hipStreamSynchronize(g_ctx.stream);
  for(auto& p : g_ctx.pending)
  {
    float ms = 0; (void)hipEventElapsedTime(&ms, p.start, p.stop);
    auto it = g_ctx.totals.find(p.name);
    if(it == g_ctx.totals.end()) { g_ctx.totals[p.name] = std::make_pair((double)ms, (uint64_t)1); g_ctx.order.push_back(p.name); }
    else { it->second.first += ms; it->second.second += 1; }
    (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop);
  }
  g_ctx.pending.clear();
}

#define LAUNCH(name, kernel, grid, block, ...) do { \
  profile_begin(name); \
  hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3((unsigned)(block)), 0, g_ctx.stream, __VA_ARGS__); \
  profile_end(); \
  hipError_t le_ = hipGetLastError(); \
  if(le_ != hipSuccess) { return fail(BWTM_ENODEV, "launch of %s failed: %s", name, hipGetErrorString(le_)); } } while(0)

#define LAUNCH_LDS(name, kernel, grid, block, lds, ...) do { \
  profile_begin(name); \
  hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3((unsigned)(block)), (unsigned)(lds), g_ctx.stream, __VA_ARGS__); \
  profile_end(); \
  hipError_t le_ = hipGetLastError(); \
  if(le_ != hipSuccess) { return fail(BWTM_ENODEV, "launch of %s failed: %s", name, hipGetErrorString(le_)); } } while(0)

inline u64 div_up(u64 a, u64 b) { return (a + b - 1) / b; }

// Device memory pool.  All work of the library runs on ONE stream, so a block released by a
// handle may be handed to the next allocation immediately (stream order protects it); blocks
// return to the driver only in bwtm_trim() or when hipMalloc runs out of memory.  A repeated
// merge of the same shape therefore performs no hipMalloc / hipFree at all (both cost
// milliseconds per GB and serialise with the device).
struct Pool
{
  std::multimap<u64, void*> free_blocks;
  u64 cached_bytes = 0;

  static u64 round_size(u64 n)
  {
    if(n < 256) { n = 256; }
    u64 g = (n >= (64ull << 20) ? (2ull << 20) : (n >= (1ull << 20) ? (64ull << 10) : 256ull));
    return (n + g - 1) / g * g;
  }
  void trim()
  {
    if(g_ctx.stream) { (void)hipStreamSynchronize(g_ctx.stream); }
    for(auto& kv : free_blocks) { (void)hipFree(kv.second); }
    free_blocks.clear(); cached_bytes = 0;
  }
  hipError_t get(u64 n, void** p, u64* actual)
  {
    n = round_size(n);
    auto it = free_blocks.lower_bound(n);
    if(it != free_blocks.end() && it->first <= n + n / 8)
    {
      *p = it->second; *actual = it->first; cached_bytes -= it->first; free_blocks.erase(it);
      return hipSuccess;
    }
    hipError_t e = hipMalloc(p, n);
    if(e != hipSuccess) { (void)hipGetLastError(); trim(); e = hipMalloc(p, n); }
    *actual = n;
    return e;
  }
  void put(void* p, u64 n) { free_blocks.insert(std::make_pair(n, p)); cached_bytes += n; }
};

Pool g_pool;

// RAII device buffer (pooled).
struct DevBuf
{
  void* p = nullptr; u64 bytes = 0;
  DevBuf() {}
  DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() { if(p) { g_pool.put(p, bytes); p = nullptr; bytes = 0; } }
  int alloc(u64 n, bool zero = false)
  {
    release();
    if(n == 0) { n = 8; }
    hipError_t e = g_pool.get(n, &p, &bytes);
    if(e != hipSuccess) { p = nullptr; bytes = 0; return fail(BWTM_ENOMEM, "hipMalloc(%llu bytes) failed: %s", (unsigned long long)n, hipGetErrorString(e)); }
    if(zero) { e = hipMemsetAsync(p, 0, n, g_ctx.stream); if(e != hipSuccess) { return fail(BWTM_ENODEV, "hipMemsetAsync failed: %s", hipGetErrorString(e)); } }
    return BWTM_OK;
  }
  template<class T> T* as() const { return (T*)p; }
  void swap(DevBuf& o) { std::swap(p, o.p); std::swap(bytes, o.bytes); }
};

#define LAUNCH2D(name, kernel, gridx, gridy, block, ...) do { \
  profile_begin(name); \
  hipLaunchKernelGGL(kernel, dim3((unsigned)(gridx), (unsigned)(gridy)), dim3((unsigned)(block)), 0, g_ctx.stream, __VA_ARGS__); \
  profile_end(); \
  hipError_t le_ = hipGetLastError(); \
  if(le_ != hipSuccess) { return fail(BWTM_ENODEV, "launch of %s failed: %s", name, hipGetErrorString(le_)); } } while(0)

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
// Handles.

struct bwtm_index
{
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
  DevBuf gcum; u64 ngroups = 0;       // 6 x (ngroups + 1) u64: cumulative symbol counts at the starts of the 62-block groups
  DevBuf cum;                         // 6 x (nblocks + 1) u64: cumulative symbol counts at block starts (built on demand)

  IndexView view() const
  {
    IndexView v;
    v.recs = recs.as<const uint4>(); v.sup = sup.as<const u64>();
    v.n = n; v.m = m; v.nrecs = nrecs;
    for(int c = 0; c < 8; c++) { v.C[c] = C[c]; }
    return v;
  }
};

struct bwtm_ra
{
  u64 na = 0, nb = 0, n_out = 0;
  u64 nrecs_out = 0, nchunks = 0;
  DevBuf owned_bits;                  // nchunks * CHUNK_WORDS u64 words (unless caller-owned)
  void* bits_ptr = nullptr;
  template<class T> T* bits_as() const { return (T*)bits_ptr; }
  DevBuf chunk_base;                  // nchunks + 1 u64 (exclusive scan of chunk popcounts)
  bool finalized = false;
  u64 values = 0;
};

//------------------------------------------------------------------------------
// Internal pipeline stages.

namespace
{

// Buffer for a native byte stream: 16 zero bytes of padding keep the last partial block readable with
// 16-byte loads.  Only the padding is cleared; the stream itself is written by the caller.
int alloc_native(DevBuf& buf, u64 nbytes)
{
  TRY(buf.alloc(nbytes + 16));
  HIP_TRY(hipMemsetAsync((u8*)buf.p + nbytes, 0, 16, g_ctx.stream));
  return BWTM_OK;
}

// Scan of a native byte stream: block_start (positions before every block) and gcum (symbol counts
// before every 62-block group).  `stream_flags` (optional) receives the k_block_len flags after the
// next stream synchronisation.
int native_samples(bwtm_index* x, u32* stream_flags)
{
  x->nblocks = div_up(x->nbytes, RLE_BLOCK);
  x->ngroups = std::max<u64>(1, div_up(x->nblocks, (u64)GROUP));
  const u64 gstride = x->ngroups + 1;
  x->cum.release();
  TRY(x->block_start.alloc((x->nblocks + 1) * sizeof(u64)));
  TRY(x->gcum.alloc(6 * gstride * sizeof(u64)));
  DevBuf flags; TRY(flags.alloc(sizeof(u32), true));
  // the kernel fills columns [0, nblocks) / [0, ngroups); the extra column of each exclusive scan is zeroed here
  HIP_TRY(hipMemsetAsync(x->block_start.as<u64>() + x->nblocks, 0, sizeof(u64), g_ctx.stream));
  HIP_TRY(hipMemset2DAsync(x->gcum.as<u64>() + x->ngroups, gstride * sizeof(u64), 0, sizeof(u64), 6, g_ctx.stream));
  LAUNCH("block_len", k_block_len, div_up(x->ngroups, BLOCK_THREADS / WAVE), BLOCK_THREADS,
    x->native_bytes(), x->nbytes, x->nblocks, x->ngroups, x->block_start.as<u64>(), x->gcum.as<u64>(), gstride, flags.as<u32>());
  TRY(device_scan<0>(x->block_start.as<u64>(), x->block_start.as<u64>(), x->nblocks + 1));
  TRY(device_scan_multi<0>(x->gcum.as<u64>(), x->gcum.as<u64>(), gstride, 6, gstride));
  if(stream_flags) { HIP_TRY(hipMemcpyAsync(stream_flags, flags.p, sizeof(u32), hipMemcpyDeviceToHost, g_ctx.stream)); }
  return BWTM_OK;
}

// samples[c] at the block starts (bwt.cpp:489-511), from block_start and the rank structure.
int ensure_block_cum(bwtm_index* x)
{
  if(x->cum.p) { return BWTM_OK; }
  const u64 stride = x->nblocks + 1;
  TRY(x->cum.alloc(6 * stride * sizeof(u64)));
  LAUNCH("block_cum", k_block_cum, div_up(stride, BLOCK_THREADS), BLOCK_THREADS,
    x->view(), x->block_start.as<const u64>(), stride, x->cum.as<u64>(), stride);
  return BWTM_OK;
}

// Records + super table from the native stream.
int transcode(bwtm_index* x)
{
  x->nrecs = num_records(x->n); x->nsup = num_supers(x->n);
  TRY(x->recs.alloc(x->nrecs * 64));
  TRY(x->sup.alloc(x->nsup * SUP_STRIDE * sizeof(u64)));
  const u64 gstride = x->ngroups + 1;
  LAUNCH("build_sup", k_build_sup, div_up(x->nsup * WAVE, BLOCK_THREADS), BLOCK_THREADS,   // one wave per super
    x->native_bytes(), x->nbytes, x->block_start.as<const u64>(), x->gcum.as<const u64>(), gstride, x->nblocks, x->ngroups, x->n,
    x->sup.as<u64>(), x->nsup);
  // one wave per group; LDS window sized to the positions a group covers on average (iid reads: ~5300)
  const u64 per_group = x->n / x->ngroups;
  const bool long_runs = (x->nblocks > 0 && x->n / x->nblocks > 400);        // > ~6 positions per byte: cooperative fill of long runs pays
#define BUILD_RECS(W, WAVES, FILL) LAUNCH("build_recs", (k_build_recs<W, WAVES, FILL>), div_up(x->ngroups, WAVES), WAVES * WAVE, \
    x->native_bytes(), x->nbytes, x->block_start.as<const u64>(), x->gcum.as<const u64>(), gstride, x->nblocks, x->ngroups, x->n, \
    x->sup.as<const u64>(), x->recs.as<uint4>(), x->nrecs)
  if(per_group <= 6500) { BUILD_RECS(8192, 4, false); }
  else if(per_group <= 14000) { BUILD_RECS(16384, 4, false); }
  else if(!long_runs) { BUILD_RECS(32768, 2, false); }
  else { BUILD_RECS(32768, 2, true); }
#undef BUILD_RECS
  return BWTM_OK;
}

int finish_native_index(bwtm_index* x, uint64_t sequences, uint64_t bases, const uint64_t* C)
{
  x->n = bases; x->m = sequences;
  u32 flags = 0;
  TRY(native_samples(x, &flags));
  // Validate the header against the stream and derive C (Alphabet(counts), support.cpp:84-91).
  const u64 gstride = x->ngroups + 1;
  u64 totals[6];
  HIP_TRY(hipMemcpy2DAsync(totals, sizeof(u64), x->gcum.as<u64>() + x->ngroups, gstride * sizeof(u64), sizeof(u64), 6,
    hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  if(flags & 1u) { return fail(BWTM_EINVAL, "not a canonical run-length stream: a full 64-byte block encodes fewer than 64 positions"); }
  u64 sum = 0; for(int c = 0; c < 6; c++) { sum += totals[c]; }
  if(sum != bases) { return fail(BWTM_EINVAL, "native stream decodes to %llu positions, header says %llu", (unsigned long long)sum, (unsigned long long)bases); }
  if(totals[0] != sequences) { return fail(BWTM_EINVAL, "native stream holds %llu endmarkers, header says %llu sequences", (unsigned long long)totals[0], (unsigned long long)sequences); }
  x->C[0] = 0;
  for(int c = 0; c < 6; c++) { x->C[c + 1] = x->C[c] + totals[c]; }
  x->C[7] = x->C[6];
  if(C) { for(int c = 0; c <= 6; c++) { x->C[c] = C[c]; } }
  x->has_native = true;
  TRY(transcode(x));
  return BWTM_OK;
}

} // namespace

//------------------------------------------------------------------------------
// Library.

extern "C" int bwtm_init(int device)
{
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if(e != hipSuccess || count <= 0) { return fail(BWTM_ENODEV, "no HIP device available (%s)", hipGetErrorString(e)); }
  if(device < 0 || device >= count) { return fail(BWTM_EINVAL, "device %d out of range (%d devices)", device, count); }
  HIP_TRY(hipSetDevice(device));
  if(g_ctx.ready && g_ctx.device == device) { return BWTM_OK; }
  if(g_ctx.stream) { (void)hipStreamDestroy(g_ctx.stream); g_ctx.stream = nullptr; }
  HIP_TRY(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
  g_ctx.device = device; g_ctx.ready = true;
  // Kernels that take more than the default 64 KiB of dynamic LDS.
  HIP_TRY(hipFuncSetAttribute((const void*)k_part_scatter_sorted, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIP_TRY(hipFuncSetAttribute((const void*)k_lf_walk_binned<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024));
  HIP_TRY(hipFuncSetAttribute((const void*)k_lf_walk_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 122 * 1024));
  HIP_TRY(hipFuncSetAttribute((const void*)k_lf_walk_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 82 * 1024));
  return BWTM_OK;
}

extern "C" const char* bwtm_last_error(void) { return g_error.c_str(); }

extern "C" int bwtm_tune(const char* key, long long value)
{
  if(!key) { return fail(BWTM_EINVAL, "bwtm_tune: null key"); }
  std::string k(key);
  if(k == "walk_emit") { g_tune.walk_emit = value; }
  else if(k == "walk_blocks") { g_tune.walk_blocks = value; }
  else if(k == "walk_kernel") { g_tune.walk_kernel = value; }
  else if(k == "emit_path") { g_tune.emit_path = value; }
  else if(k == "walk_ablate") { g_tune.walk_ablate = value; }
  else if(k == "scatter_kernel") { g_tune.scatter_kernel = value; }
  else if(k == "walk_variant") { g_tune.walk_variant = value; }
  else if(k == "l1_cap") { g_tune.l1_cap = value; }
  else if(k == "search_algo") { g_tune.search_algo = value; }
  else if(k == "frontier_unfused") { g_tune.frontier_unfused = value; }
  else if(k == "round_emits") { g_tune.round_emits = (value > 0 ? value : 1); }
  else { return fail(BWTM_EINVAL, "bwtm_tune: unknown key %s", key); }
  return BWTM_OK;
}

extern "C" int bwtm_trim(void)
{
  g_pool.trim();
  return BWTM_OK;
}

extern "C" int bwtm_synchronize(void)
{
  TRY(ensure_ready());
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return BWTM_OK;
}

//------------------------------------------------------------------------------
// Index.

extern "C" int bwtm_index_upload(const uint8_t* data, uint64_t nbytes, uint64_t sequences, uint64_t bases,
  const uint64_t* C, bwtm_index** out)
{
  TRY(ensure_ready());
  if(!out || (nbytes > 0 && !data)) { return fail(BWTM_EINVAL, "bwtm_index_upload: null argument"); }
  bwtm_index* x = new bwtm_index();
  x->nbytes = nbytes;
  int rc = alloc_native(x->data, nbytes);
  if(rc == BWTM_OK && nbytes > 0)
  {
    hipError_t e = hipMemcpyAsync(x->data.p, data, nbytes, hipMemcpyHostToDevice, g_ctx.stream);
    if(e != hipSuccess) { rc = fail(BWTM_ENODEV, "H2D copy failed: %s", hipGetErrorString(e)); }
  }
  if(rc == BWTM_OK) { rc = finish_native_index(x, sequences, bases, C); }
  if(rc != BWTM_OK) { delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

extern "C" int bwtm_index_from_device(const void* device_data, uint64_t nbytes, uint64_t sequences, uint64_t bases,
  const uint64_t* C, bwtm_index** out)
{
  TRY(ensure_ready());
  if(!out || (nbytes > 0 && !device_data)) { return fail(BWTM_EINVAL, "bwtm_index_from_device: null argument"); }
  bwtm_index* x = new bwtm_index();
  x->nbytes = nbytes;
  int rc = alloc_native(x->data, nbytes);
  if(rc == BWTM_OK && nbytes > 0)
  {
    hipError_t e = hipMemcpyAsync(x->data.p, device_data, nbytes, hipMemcpyDeviceToDevice, g_ctx.stream);
    if(e != hipSuccess) { rc = fail(BWTM_ENODEV, "D2D copy failed: %s", hipGetErrorString(e)); }
  }
  if(rc == BWTM_OK) { rc = finish_native_index(x, sequences, bases, C); }
  if(rc != BWTM_OK) { delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

extern "C" int bwtm_index_from_device_borrowed(const void* device_data, uint64_t nbytes, uint64_t sequences, uint64_t bases,
  const uint64_t* C, bwtm_index** out)
{
  TRY(ensure_ready());
  if(!out || !device_data) { return fail(BWTM_EINVAL, "bwtm_index_from_device_borrowed: null argument"); }
  if(((uintptr_t)device_data & 15) != 0) { return fail(BWTM_EINVAL, "bwtm_index_from_device_borrowed: the buffer must be 16-byte aligned"); }
  bwtm_index* x = new bwtm_index();
  x->nbytes = nbytes; x->borrowed = device_data;
  int rc = finish_native_index(x, sequences, bases, C);
  if(rc != BWTM_OK) { delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

extern "C" int bwtm_index_from_symbols_device(const void* device_symbols, uint64_t bases, bwtm_index** out)
{
  TRY(ensure_ready());
  if(!out || (bases > 0 && !device_symbols)) { return fail(BWTM_EINVAL, "bwtm_index_from_symbols_device: null argument"); }
  bwtm_index* x = new bwtm_index();
  auto body = [&]() -> int
  {
    x->n = bases;
    x->nrecs = num_records(bases); x->nsup = num_supers(bases);
    u64 stride = x->nrecs + 1;
    DevBuf cnt; TRY(cnt.alloc(6 * stride * sizeof(u64), true));
    LAUNCH("sym_counts", k_sym_counts, div_up(x->nrecs, BLOCK_THREADS), BLOCK_THREADS,
      (const u8*)device_symbols, bases, x->nrecs, cnt.as<u64>(), stride);
    TRY(device_scan_multi<0>(cnt.as<u64>(), cnt.as<u64>(), stride, 6, stride));
    u64 totals[6];
    for(int c = 0; c < 6; c++)
    {
      HIP_TRY(hipMemcpyAsync(&totals[c], cnt.as<u64>() + c * stride + x->nrecs, sizeof(u64), hipMemcpyDeviceToHost, g_ctx.stream));
    }
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    x->m = totals[0];
    x->C[0] = 0; for(int c = 0; c < 6; c++) { x->C[c + 1] = x->C[c] + totals[c]; } x->C[7] = x->C[6];
    TRY(x->recs.alloc(x->nrecs * 64));
    TRY(x->sup.alloc(x->nsup * SUP_STRIDE * sizeof(u64)));
    LAUNCH("sym_sup", k_sym_sup, div_up(x->nsup, BLOCK_THREADS), BLOCK_THREADS, cnt.as<const u64>(), stride, x->nrecs, x->sup.as<u64>(), x->nsup);
    LAUNCH("sym_recs", k_sym_recs, div_up(x->nrecs, BLOCK_THREADS), BLOCK_THREADS,
      (const u8*)device_symbols, bases, cnt.as<const u64>(), stride, x->sup.as<const u64>(), x->recs.as<uint4>(), x->nrecs);
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));     // the caller may release `device_symbols` on return
    return BWTM_OK;
  };
  int rc = body();
  if(rc != BWTM_OK) { delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

extern "C" void bwtm_index_free(bwtm_index* index)
{
  if(index && index->borrowed) { (void)hipStreamSynchronize(g_ctx.stream); }   // queued readers of the caller's buffer
  delete index;                                     // buffers return to the pool (stream ordered)
}

extern "C" uint64_t bwtm_index_bases(const bwtm_index* x)     { return x ? x->n : 0; }
extern "C" uint64_t bwtm_index_sequences(const bwtm_index* x) { return x ? x->m : 0; }
extern "C" uint64_t bwtm_index_bytes(const bwtm_index* x)     { return (x && x->has_native) ? x->nbytes : 0; }
extern "C" uint64_t bwtm_index_blocks(const bwtm_index* x)    { return (x && x->has_native) ? x->nblocks : 0; }
extern "C" void bwtm_index_C(const bwtm_index* x, uint64_t* C) { for(int c = 0; c <= 6; c++) { C[c] = x->C[c]; } }

extern "C" int bwtm_index_drop_native(bwtm_index* x)
{
  if(!x) { return fail(BWTM_EINVAL, "null index"); }
  if(x->borrowed) { HIP_TRY(hipStreamSynchronize(g_ctx.stream)); x->borrowed = nullptr; }
  x->data.release(); x->cum.release(); x->gcum.release(); x->block_start.release();
  x->has_native = false; x->nbytes = 0; x->nblocks = 0;
  return BWTM_OK;
}

extern "C" int bwtm_index_encode(bwtm_index* x)
{
  TRY(ensure_ready());
  if(!x) { return fail(BWTM_EINVAL, "null index"); }
  if(x->has_native) { return BWTM_OK; }
  x->nbytes = 0; x->nblocks = 0;
  if(x->n > 0)
  {
    u64 ntiles = (x->n >> 6) + 1;
    u64 nseg = div_up(ntiles, SEG_TILES);
    u64 ngroups = div_up(nseg, FOLD_GROUP);
    DevBuf lasthead, table, group_table, group_base, seg_base;
    TRY(lasthead.alloc(nseg * sizeof(u64)));
    TRY(table.alloc(nseg * 64 * sizeof(u32)));
    TRY(group_table.alloc(ngroups * 64 * sizeof(u64)));
    TRY(group_base.alloc((ngroups + 1) * sizeof(u64)));
    TRY(seg_base.alloc(nseg * sizeof(u64)));
    u64 wave_grid = div_up(nseg * WAVE, BLOCK_THREADS);
    LAUNCH("enc_lasthead", k_enc_lasthead, wave_grid, BLOCK_THREADS, x->recs.as<const uint4>(), x->nrecs, x->n, ntiles, nseg, lasthead.as<u64>());
    TRY(device_scan<1>(lasthead.as<u64>(), lasthead.as<u64>(), nseg));       // -> (last head before the segment) + 1
    LAUNCH("enc_size", k_enc_size, wave_grid, BLOCK_THREADS, x->recs.as<const uint4>(), x->nrecs, x->n, ntiles, nseg,
      lasthead.as<const u64>(), table.as<u32>());
    LAUNCH("fold_group", k_fold_group, ngroups, WAVE, table.as<const u32>(), nseg, group_table.as<u64>());
    LAUNCH("fold_top", k_fold_top, 1, WAVE, group_table.as<const u64>(), ngroups, group_base.as<u64>());
    LAUNCH("fold_seg", k_fold_seg, ngroups, WAVE, table.as<const u32>(), nseg, group_base.as<const u64>(), seg_base.as<u64>());
    u64 total = 0;
    HIP_TRY(hipMemcpyAsync(&total, group_base.as<u64>() + ngroups, sizeof(u64), hipMemcpyDeviceToHost, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    x->nbytes = total;
    x->nblocks = div_up(total, RLE_BLOCK);
    TRY(alloc_native(x->data, total));
    // k_enc_emit records the position at which every 64-byte block starts; the entry after the last block is n
    TRY(x->block_start.alloc((x->nblocks + 1) * sizeof(u64)));
    HIP_TRY(hipMemcpyAsync(x->block_start.as<u64>() + x->nblocks, &x->n, sizeof(u64), hipMemcpyHostToDevice, g_ctx.stream));
    LAUNCH("enc_emit", k_enc_emit, wave_grid, BLOCK_THREADS, x->recs.as<const uint4>(), x->nrecs, x->n, ntiles, nseg,
      lasthead.as<const u64>(), seg_base.as<const u64>(), x->data.as<u8>(), x->block_start.as<u64>());
  }
  else
  {
    TRY(alloc_native(x->data, 0));
    TRY(x->block_start.alloc(sizeof(u64), true));
  }
  x->gcum.release(); x->ngroups = 0; x->cum.release();
  TRY(ensure_block_cum(x));                     // BWT::build, bwt.cpp:476-512: samples of the new stream
  x->has_native = true;
  return BWTM_OK;
}

extern "C" int bwtm_index_device_data(bwtm_index* x, void** device_ptr, uint64_t* nbytes)
{
  TRY(ensure_ready());
  if(!x || !device_ptr || !nbytes) { return fail(BWTM_EINVAL, "bwtm_index_device_data: null argument"); }
  if(!x->has_native) { return fail(BWTM_EINVAL, "index has no native byte stream (call bwtm_index_encode first)"); }
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  *device_ptr = (void*)x->native_bytes(); *nbytes = x->nbytes;
  return BWTM_OK;
}

extern "C" int bwtm_index_download_data(bwtm_index* x, uint8_t* out, uint64_t capacity)
{
  TRY(ensure_ready());
  if(!x || !x->has_native) { return fail(BWTM_EINVAL, "index has no native byte stream (call bwtm_index_encode first)"); }
  if(capacity < x->nbytes) { return fail(BWTM_EINVAL, "buffer too small: %llu < %llu", (unsigned long long)capacity, (unsigned long long)x->nbytes); }
  if(x->nbytes > 0) { HIP_TRY(hipMemcpyAsync(out, x->native_bytes(), x->nbytes, hipMemcpyDeviceToHost, g_ctx.stream)); }
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return BWTM_OK;
}

extern "C" int bwtm_index_download_samples(bwtm_index* x, uint64_t* block_end, uint64_t* cum)
{
  TRY(ensure_ready());
  if(!x || !x->has_native) { return fail(BWTM_EINVAL, "index has no native samples (call bwtm_index_encode first)"); }
  u64 stride = x->nblocks + 1;
  TRY(ensure_block_cum(x));
  if(x->nblocks > 0)
  {
    DevBuf be; TRY(be.alloc(x->nblocks * sizeof(u64)));
    LAUNCH("block_end", k_block_end, div_up(x->nblocks, BLOCK_THREADS), BLOCK_THREADS, x->block_start.as<const u64>(), x->nblocks, be.as<u64>());
    HIP_TRY(hipMemcpyAsync(block_end, be.p, x->nblocks * sizeof(u64), hipMemcpyDeviceToHost, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  }
  HIP_TRY(hipMemcpyAsync(cum, x->cum.p, 6 * stride * sizeof(u64), hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return BWTM_OK;
}

extern "C" int bwtm_rank_batch(const bwtm_index* x, const uint64_t* positions, const uint8_t* comps, uint64_t count, uint64_t* out_ranks)
{
  TRY(ensure_ready());
  if(!x || !positions || !comps || !out_ranks) { return fail(BWTM_EINVAL, "bwtm_rank_batch: null argument"); }
  if(count == 0) { return BWTM_OK; }
  DevBuf dp, dc, dr;
  TRY(dp.alloc(count * 8)); TRY(dc.alloc(count)); TRY(dr.alloc(count * 8));
  HIP_TRY(hipMemcpyAsync(dp.p, positions, count * 8, hipMemcpyHostToDevice, g_ctx.stream));
  HIP_TRY(hipMemcpyAsync(dc.p, comps, count, hipMemcpyHostToDevice, g_ctx.stream));
  LAUNCH("rank_batch", k_rank_batch, div_up(count, BLOCK_THREADS), BLOCK_THREADS, x->view(), dp.as<const u64>(), dc.as<const u8>(), count, dr.as<u64>());
  HIP_TRY(hipMemcpyAsync(out_ranks, dr.p, count * 8, hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return BWTM_OK;
}

extern "C" int bwtm_inverse_select_batch(const bwtm_index* x, const uint64_t* positions, uint64_t count, uint64_t* out_ranks, uint8_t* out_comps)
{
  TRY(ensure_ready());
  if(!x || !positions || !out_ranks || !out_comps) { return fail(BWTM_EINVAL, "bwtm_inverse_select_batch: null argument"); }
  if(count == 0) { return BWTM_OK; }
  DevBuf dp, dc, dr;
  TRY(dp.alloc(count * 8)); TRY(dc.alloc(count)); TRY(dr.alloc(count * 8));
  HIP_TRY(hipMemcpyAsync(dp.p, positions, count * 8, hipMemcpyHostToDevice, g_ctx.stream));
  LAUNCH("inverse_select_batch", k_inverse_select_batch, div_up(count, BLOCK_THREADS), BLOCK_THREADS, x->view(), dp.as<const u64>(), count, dr.as<u64>(), dc.as<u8>());
  HIP_TRY(hipMemcpyAsync(out_ranks, dr.p, count * 8, hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipMemcpyAsync(out_comps, dc.p, count, hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return BWTM_OK;
}

extern "C" int bwtm_find_batch(const bwtm_index* x, const uint8_t* patterns, const uint64_t* offsets, uint64_t count, uint64_t* out_sp, uint64_t* out_ep)
{
  TRY(ensure_ready());
  if(!x || !offsets || !out_sp || !out_ep) { return fail(BWTM_EINVAL, "bwtm_find_batch: null argument"); }
  if(count == 0) { return BWTM_OK; }
  u64 total = offsets[count];
  if(total > 0 && !patterns) { return fail(BWTM_EINVAL, "bwtm_find_batch: null pattern text"); }
  DevBuf dt, doff, dsp, dep;
  TRY(dt.alloc(total + 16)); TRY(doff.alloc((count + 1) * 8)); TRY(dsp.alloc(count * 8)); TRY(dep.alloc(count * 8));
  if(total > 0) { HIP_TRY(hipMemcpyAsync(dt.p, patterns, total, hipMemcpyHostToDevice, g_ctx.stream)); }
  HIP_TRY(hipMemcpyAsync(doff.p, offsets, (count + 1) * 8, hipMemcpyHostToDevice, g_ctx.stream));
  LAUNCH("find_batch", k_find_batch, div_up(count, BLOCK_THREADS), BLOCK_THREADS, x->view(), dt.as<const u8>(), doff.as<const u64>(), count, dsp.as<u64>(), dep.as<u64>());
  HIP_TRY(hipMemcpyAsync(out_sp, dsp.p, count * 8, hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipMemcpyAsync(out_ep, dep.p, count * 8, hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return BWTM_OK;
}

extern "C" int bwtm_extract(const bwtm_index* x, uint64_t first, uint64_t count, uint8_t* out)
{
  TRY(ensure_ready());
  if(!x || !out) { return fail(BWTM_EINVAL, "bwtm_extract: null argument"); }
  if(first + count > x->n) { return fail(BWTM_EINVAL, "bwtm_extract: range past the end"); }   // bwt.h:137
  if(count == 0) { return BWTM_OK; }
  DevBuf d; TRY(d.alloc(count));
  LAUNCH("extract", k_extract, div_up(count, BLOCK_THREADS), BLOCK_THREADS, x->view(), first, count, d.as<u8>());
  HIP_TRY(hipMemcpyAsync(out, d.p, count, hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return BWTM_OK;
}

//------------------------------------------------------------------------------
// Rank array.

extern "C" uint64_t bwtm_ra_buffer_bytes(const bwtm_index* a, const bwtm_index* b)
{
  if(!a || !b) { return 0; }
  return div_up(num_records(a->n + b->n), 64) * CHUNK_WORDS * sizeof(u64);
}

extern "C" int bwtm_ra_create_on(const bwtm_index* a, const bwtm_index* b, void* device_buffer, uint64_t nbytes, bwtm_ra** out)
{
  TRY(ensure_ready());
  if(!a || !b || !out) { return fail(BWTM_EINVAL, "bwtm_ra_create: null argument"); }
  bwtm_ra* ra = new bwtm_ra();
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

extern "C" void bwtm_ra_free(bwtm_ra* ra)
{
  if(!ra) { return; }
  // A caller-owned bitvector may be reused by the caller right away: drain the stream first.
  if(!ra->owned_bits.p && g_ctx.stream) { (void)hipStreamSynchronize(g_ctx.stream); }
  delete ra;
}

namespace
{

// First version of the search (one atomicOr per emit); kept for A/B measurements and as the
// fallback when the partition parameters do not fit (see search_partitioned).
int search_atomic(const bwtm_index* a, const bwtm_index* b, u64 seq_first, u64 count, bwtm_ra* ra)
{
  const u64 lanes_per_chain = (g_tune.walk_kernel == 0 ? 4 : 1);
  u64 blocks = div_up(count * lanes_per_chain, BLOCK_THREADS);
  const u64 max_blocks = (g_tune.walk_blocks > 0 ? (u64)g_tune.walk_blocks : 256 * 8);
  if(blocks > max_blocks) { blocks = max_blocks; }
  DevBuf scratch;
  u32* target = ra->bits_as<u32>();
  if(g_tune.walk_emit == 2) { TRY(scratch.alloc(b->n * sizeof(u64) + 64)); target = scratch.as<u32>(); }
  if(g_tune.walk_kernel == 0)
  {
    if(g_tune.walk_emit == 0)      { LAUNCH("lf_walk_atomic", k_lf_walk_quad<0>, blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); }
    else if(g_tune.walk_emit == 1)
    {
      switch(g_tune.walk_ablate)
      {
        case 1: LAUNCH("lf_walk_noemit_nosup", (k_lf_walk_quad<1, 1>), blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); break;
        case 2: LAUNCH("lf_walk_noemit_noA", (k_lf_walk_quad<1, 2>), blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); break;
        case 3: LAUNCH("lf_walk_noemit_nosup_noA", (k_lf_walk_quad<1, 3>), blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); break;
        case 7: LAUNCH("lf_walk_noemit_noloads", (k_lf_walk_quad<1, 7>), blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); break;
        case 8: LAUNCH("lf_walk_noemit_synthetic", (k_lf_walk_quad<1, 8>), blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); break;
        default: LAUNCH("lf_walk_noemit", (k_lf_walk_quad<1, 0>), blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); break;
      }
    }
    else                           { LAUNCH("lf_walk_store", k_lf_walk_quad<2>, blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); }
  }
  else
  {
    if(g_tune.walk_emit == 0)      { LAUNCH("lf_walk_lane", k_lf_walk<0>, blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); }
    else if(g_tune.walk_emit == 1) { LAUNCH("lf_walk_lane_noemit", k_lf_walk<1>, blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); }
    else                           { LAUNCH("lf_walk_lane_store", k_lf_walk<2>, blocks, BLOCK_THREADS, a->view(), b->view(), seq_first, count, target); }
  }
  return BWTM_OK;
}

// Level 2 of the emit partition + tile build (shared by the walk and the frontier search):
// per-bin slices -> counts -> offsets -> LDS counting sort -> tiles ORed into the bitvector.
int partition_level2(DevBuf& l1, DevBuf& gcount, u64 cap, u64 nsub, u32 subs, bwtm_ra* ra)
{
  const u32 nregions = (u32)L1_BINS * subs;
  const u64 ntiles_pad = nsub * L1_BINS;
  const u64 nwords = ra->nchunks * CHUNK_WORDS;
  {
    std::vector<u64> counts_host(nregions);
    HIP_TRY(hipMemcpyAsync(counts_host.data(), gcount.p, nregions * sizeof(u64), hipMemcpyDeviceToHost, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));

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
    HIP_TRY(hipMemcpyAsync(d_slice_bin.p, slice_bin.data(), nslices * sizeof(u32), hipMemcpyHostToDevice, g_ctx.stream));
    HIP_TRY(hipMemcpyAsync(d_slice_begin.p, slice_begin.data(), nslices * sizeof(u64), hipMemcpyHostToDevice, g_ctx.stream));
    HIP_TRY(hipMemcpyAsync(d_bin_slice0.p, bin_slice0.data(), (L1_BINS + 1) * sizeof(u32), hipMemcpyHostToDevice, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));         // the host vectors go out of scope at the end of the round
    TRY(counts.alloc(nslices * nsub * sizeof(u32)));
    TRY(tile_start.alloc((ntiles_pad + 1) * sizeof(u64), true));
    TRY(lists.alloc(total_entries * sizeof(unsigned short) + 64));

    LAUNCH_LDS("part_count", k_part_count, nslices, PART_THREADS, nsub * sizeof(u32), l1.as<const u32>(), cap, gcount.as<const u64>(),
      d_slice_bin.as<const u32>(), d_slice_begin.as<const u64>(), (u32)nsub, counts.as<u32>());
    LAUNCH("part_offsets", k_part_offsets, div_up(ntiles_pad, BLOCK_THREADS), BLOCK_THREADS, counts.as<u32>(), d_bin_slice0.as<const u32>(), (u32)nsub, tile_start.as<u64>());
    TRY(device_scan<0>(tile_start.as<u64>(), tile_start.as<u64>(), ntiles_pad + 1));
    const u64 sort_lds = nsub * sizeof(u64) + SORT_CHUNK * sizeof(u32) + (2 * nsub + 1) * sizeof(u32);
    if(sort_lds <= 96 * 1024 && g_tune.scatter_kernel == 0)
    {
      LAUNCH_LDS("part_scatter", k_part_scatter_sorted, nslices, PART_THREADS, sort_lds, l1.as<const u32>(), cap, gcount.as<const u64>(),
        d_slice_bin.as<const u32>(), subs, d_slice_begin.as<const u64>(), (u32)nsub, counts.as<const u32>(), tile_start.as<const u64>(), lists.as<unsigned short>());
    }
    else
    {
      LAUNCH_LDS("part_scatter_direct", k_part_scatter, nslices, PART_THREADS, nsub * sizeof(u64), l1.as<const u32>(), cap, gcount.as<const u64>(),
        d_slice_bin.as<const u32>(), subs, d_slice_begin.as<const u64>(), (u32)nsub, counts.as<const u32>(), tile_start.as<const u64>(), lists.as<unsigned short>());
    }
    LAUNCH("tile_build", k_tile_build, ntiles_pad, BLOCK_THREADS, lists.as<const unsigned short>(), tile_start.as<const u64>(), ntiles_pad, ra->bits_as<u64>(), nwords);
  }
  return BWTM_OK;
}

// Per-chain walk with partitioned emit, level-2 counting sort, tile build (fallback of the frontier search).
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
    const u64 max_blocks = (g_tune.walk_blocks > 0 ? (u64)g_tune.walk_blocks : 512);
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
    const u64 stage_bytes = (u64)(WL_THREADS / WAVE) * 64 * WL_ROW * sizeof(u32);
    if(g_tune.walk_variant == 1 && a->nrecs < (1ull << 32) && b->nrecs < (1ull << 32))
    {
      // variant: coalesced loads + one chain per lane through an LDS transpose (measured slower, kept for A/B)
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
    else if(sup_bytes <= 40 * 1024)
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

// Level-synchronous search (k_frontier_*): one launch per LF step over the sorted frontier; the
// emits of every step are written densely and turned into bitvector tiles at the end of an epoch.
int frontier_flush(bwtm_ra* ra, DevBuf& emit16, u64 emit_cap, DevBuf& emit_base, DevBuf& bound, u64 ntiles, u64 nsteps)
{
  if(nsteps == 0) { return BWTM_OK; }
  LAUNCH("bound_suffix_min", k_bound_suffix_min, nsteps, BLOCK_THREADS, bound.as<u32>(), ntiles, emit_base.as<const u64>(), nsteps);
  LAUNCH("tile_build", k_tile_build_frontier, ntiles, BLOCK_THREADS, emit16.as<const unsigned short>(), emit_base.as<const u64>(), emit_cap, bound.as<const u32>(),
    ntiles, nsteps, ra->bits_as<u64>(), ra->nchunks * CHUNK_WORDS);
  return BWTM_OK;
}

constexpr u64 FRONTIER_MIN_SEQUENCES = 1ull << 21;

int search_frontier(const bwtm_index* a, const bwtm_index* b, u64 seq_first, u64 count, bwtm_ra* ra)
{
  if(a->n >= (1ull << 40) || b->n >= (1ull << 40) || count >= (1ull << 32)) { return search_partitioned(a, b, seq_first, count, ra); }
  const u64 ntiles = div_up(ra->n_out + 1, 1ull << TILE_SHIFT);
  const u64 EPOCH = 512;                                  // steps whose emits are kept before tiles are built
  const u64 nb_max = div_up(count, FR_BLOCK);
  const u64 nseg = 5 * nb_max;
  const u64 fcap = nb_max * FR_BLOCK;

  DevBuf lo[2], hi[2], seg_len[2], seg_phys[2], seg_prefix, emit16, emit_base, bound;
  for(int k = 0; k < 2; k++)
  {
    TRY(lo[k].alloc(fcap * 8)); TRY(hi[k].alloc(fcap * 2));
    TRY(seg_len[k].alloc((nseg + 1) * sizeof(u64), true)); TRY(seg_phys[k].alloc((nseg + 1) * sizeof(u64), true));
  }
  TRY(seg_prefix.alloc((nseg + 1) * sizeof(u64)));
  DevBuf first_seg; TRY(first_seg.alloc((nb_max + 1) * sizeof(u32)));
  // The host looks at the frontier size every few steps: a dead step costs little for a small frontier, a
  // synchronisation costs little next to a large one.
  const u64 check_every = (count >= (1ull << 20) ? 8 : 32);
  const u64 scan_tiles = div_up(nseg + 1, (u64)SCAN_TILE);
  DevBuf scan_partial; TRY(scan_partial.alloc(scan_tiles * sizeof(u64)));
  // An epoch emits at most one value per position of b; a shard of the sequences usually far less.
  // Emits past the capacity take the exact atomicOr fallback.
  const u64 per_seq = b->n / (b->m > 0 ? b->m : 1) + 1;
  u64 emit_cap = 2 * count * per_seq + (1ull << 20);
  if(emit_cap > b->n + 64) { emit_cap = b->n + 64; }
  if(g_tune.l1_cap > 0) { emit_cap = (u64)g_tune.l1_cap; }       // tests: force the fallback
  TRY(emit16.alloc(emit_cap * sizeof(unsigned short)));
  TRY(emit_base.alloc((EPOCH + 1) * sizeof(u64), true));
  TRY(bound.alloc(EPOCH * (ntiles + 1) * sizeof(u32)));
  HIP_TRY(hipMemsetAsync(bound.p, 0xFF, EPOCH * (ntiles + 1) * sizeof(u32), g_ctx.stream));

  u64 init_items = (fcap > nseg + 1 ? fcap : nseg + 1);
  LAUNCH("frontier_init", k_frontier_init, div_up(init_items, BLOCK_THREADS), BLOCK_THREADS, lo[0].as<uint2>(), hi[0].as<unsigned short>(),
    seg_len[0].as<u64>(), seg_phys[0].as<u64>(), nb_max, seq_first, count, a->m);
  int cur = 0;
  u64 in_epoch = 0;
  for(u64 t = 0; t <= b->n; t++)
  {
    if(scan_tiles <= FRONTIER_SCAN_TILES && g_tune.frontier_unfused == 0)
    {
      // scan of the segment lengths + per-step bookkeeping in two launches (k_frontier_scan)
      if(scan_tiles > 1)
      {
        LAUNCH("scan_reduce", k_scan_reduce<0>, scan_tiles, BLOCK_THREADS, seg_len[cur].as<const u64>(), scan_partial.as<u64>(), nseg + 1, (u64)0, scan_tiles);
      }
      LAUNCH("frontier_scan", k_frontier_scan, scan_tiles, BLOCK_THREADS, seg_len[cur].as<const u64>(), scan_partial.as<const u64>(), nseg,
        seg_prefix.as<u64>(), first_seg.as<u32>(), emit_base.as<u64>(), in_epoch);
    }
    else
    {
      TRY(device_scan<0>(seg_len[cur].as<u64>(), seg_prefix.as<u64>(), nseg + 1));
      LAUNCH("frontier_prep", k_frontier_prep, div_up(nseg, BLOCK_THREADS), BLOCK_THREADS, seg_prefix.as<const u64>(), nseg, first_seg.as<u32>(),
        emit_base.as<u64>(), in_epoch);
    }
    if(t % check_every == 0)
    {
      u64 alive = 0;
      HIP_TRY(hipMemcpyAsync(&alive, seg_prefix.as<u64>() + nseg, sizeof(u64), hipMemcpyDeviceToHost, g_ctx.stream));
      HIP_TRY(hipStreamSynchronize(g_ctx.stream));
      if(alive == 0) { break; }
    }
    FrontierView f;
    f.lo = lo[cur].as<const uint2>(); f.hi = hi[cur].as<const unsigned short>();
    f.lo_next = lo[1 - cur].as<uint2>(); f.hi_next = hi[1 - cur].as<unsigned short>();
    f.seg_prefix = seg_prefix.as<const u64>(); f.seg_phys = seg_phys[cur].as<const u64>(); f.first_seg = first_seg.as<const u32>();
    f.seg_len_next = seg_len[1 - cur].as<u64>(); f.seg_phys_next = seg_phys[1 - cur].as<u64>();
    f.nb_max = nb_max;
    f.emit16 = emit16.as<unsigned short>(); f.emit_base = emit_base.as<const u64>(); f.emit_cap = emit_cap; f.bits32 = ra->bits_as<u32>();
    f.bound_row = bound.as<u32>() + in_epoch * (ntiles + 1); f.step = in_epoch;
    if(g_tune.walk_emit == 1) { LAUNCH("frontier_step_noemit", k_frontier_step<1>, nb_max, FR_BLOCK, a->view(), b->view(), f); }
    else { LAUNCH("frontier_step", k_frontier_step<0>, nb_max, FR_BLOCK, a->view(), b->view(), f); }
    cur = 1 - cur;
    in_epoch++;
    if(in_epoch == EPOCH)
    {
      if(g_tune.walk_emit == 0) { TRY(frontier_flush(ra, emit16, emit_cap, emit_base, bound, ntiles, in_epoch)); }
      HIP_TRY(hipMemsetAsync(bound.p, 0xFF, EPOCH * (ntiles + 1) * sizeof(u32), g_ctx.stream));
      HIP_TRY(hipMemsetAsync(emit_base.p, 0, (EPOCH + 1) * sizeof(u64), g_ctx.stream));
      in_epoch = 0;
    }
  }
  if(g_tune.walk_emit == 0) { TRY(frontier_flush(ra, emit16, emit_cap, emit_base, bound, ntiles, in_epoch)); }
  return BWTM_OK;
}

} // namespace

extern "C" int bwtm_search(const bwtm_index* a, const bwtm_index* b, uint64_t seq_first, uint64_t seq_last, bwtm_ra* ra)
{
  TRY(ensure_ready());
  if(!a || !b || !ra) { return fail(BWTM_EINVAL, "bwtm_search: null argument"); }
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
  if(want_frontier && g_tune.emit_path == 0 && g_tune.walk_kernel == 0) { return search_frontier(a, b, seq_first, count, ra); }
  if(g_tune.emit_path == 0 && g_tune.walk_emit == 0 && g_tune.walk_kernel == 0) { return search_partitioned(a, b, seq_first, count, ra); }
  return search_atomic(a, b, seq_first, count, ra);
}

extern "C" int bwtm_ra_device_buffer(bwtm_ra* ra, void** device_ptr, uint64_t* nbytes)
{
  TRY(ensure_ready());
  if(!ra || !device_ptr || !nbytes) { return fail(BWTM_EINVAL, "bwtm_ra_device_buffer: null argument"); }
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  *device_ptr = ra->bits_ptr; *nbytes = ra->nchunks * CHUNK_WORDS * sizeof(u64);
  return BWTM_OK;
}

extern "C" int bwtm_ra_finalize(bwtm_ra* ra)
{
  TRY(ensure_ready());
  if(!ra) { return fail(BWTM_EINVAL, "null rank array"); }
  LAUNCH("chunk_popc", k_chunk_popc, div_up(ra->nchunks * WAVE, BLOCK_THREADS), BLOCK_THREADS, ra->bits_as<const u64>(), ra->nchunks, ra->chunk_base.as<u64>());
  TRY(device_scan<0>(ra->chunk_base.as<u64>(), ra->chunk_base.as<u64>(), ra->nchunks + 1));
  HIP_TRY(hipMemcpyAsync(&ra->values, ra->chunk_base.as<u64>() + ra->nchunks, sizeof(u64), hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  ra->finalized = true;
  return BWTM_OK;
}

extern "C" uint64_t bwtm_ra_values(const bwtm_ra* ra) { return ra ? ra->values : 0; }

extern "C" int bwtm_ra_download(bwtm_ra* ra, uint64_t* out, uint64_t capacity)
{
  TRY(ensure_ready());
  if(!ra || !out) { return fail(BWTM_EINVAL, "bwtm_ra_download: null argument"); }
  if(!ra->finalized) { return fail(BWTM_EINVAL, "bwtm_ra_download: rank array not finalized"); }
  if(capacity < ra->nb) { return fail(BWTM_EINVAL, "bwtm_ra_download: buffer too small"); }
  if(ra->nb == 0) { return BWTM_OK; }
  DevBuf d; TRY(d.alloc(ra->nb * sizeof(u64), true));
  LAUNCH("ra_extract", k_ra_extract, div_up(ra->nchunks * WAVE, BLOCK_THREADS), BLOCK_THREADS,
    ra->bits_as<const u64>(), ra->chunk_base.as<const u64>(), ra->nchunks, ra->nb, d.as<u64>());
  HIP_TRY(hipMemcpyAsync(out, d.p, ra->nb * sizeof(u64), hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return BWTM_OK;
}

extern "C" int bwtm_ra_download_bits(bwtm_ra* ra, uint64_t* out_words, uint64_t capacity_words)
{
  TRY(ensure_ready());
  if(!ra || !out_words) { return fail(BWTM_EINVAL, "bwtm_ra_download_bits: null argument"); }
  u64 words = div_up(ra->n_out, 64);
  if(capacity_words < words) { return fail(BWTM_EINVAL, "bwtm_ra_download_bits: buffer too small"); }
  if(words > 0) { HIP_TRY(hipMemcpyAsync(out_words, ra->bits_ptr, words * sizeof(u64), hipMemcpyDeviceToHost, g_ctx.stream)); }
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return BWTM_OK;
}

//------------------------------------------------------------------------------
// Interleave and the whole path.

extern "C" int bwtm_interleave(const bwtm_index* a, const bwtm_index* b, bwtm_ra* ra, bwtm_index** out)
{
  TRY(ensure_ready());
  if(!a || !b || !ra || !out) { return fail(BWTM_EINVAL, "bwtm_interleave: null argument"); }
  if(!ra->finalized) { return fail(BWTM_EINVAL, "bwtm_interleave: rank array not finalized"); }
  if(ra->na != a->n || ra->nb != b->n) { return fail(BWTM_EINVAL, "bwtm_interleave: rank array was created for other inputs"); }
  if(ra->values != b->n) { return fail(BWTM_EINVAL, "bwtm_interleave: rank array holds %llu values, expected %llu", (unsigned long long)ra->values, (unsigned long long)b->n); }
  bwtm_index* x = new bwtm_index();
  auto body = [&]() -> int
  {
    x->n = ra->n_out; x->m = a->m + b->m;                           // bwt.cpp:305-306
    for(int c = 0; c < 8; c++) { x->C[c] = a->C[c] + b->C[c]; }     // fmi.cpp:367-368
    x->nrecs = ra->nrecs_out; x->nsup = num_supers(x->n);
    TRY(x->recs.alloc(x->nrecs * 64));
    TRY(x->sup.alloc(x->nsup * SUP_STRIDE * sizeof(u64)));
    LAUNCH("interleave_sup", k_interleave_sup, div_up(x->nsup, BLOCK_THREADS), BLOCK_THREADS, a->view(), b->view(),
      ra->bits_as<const u64>(), ra->chunk_base.as<const u64>(), x->n, x->sup.as<u64>(), x->nsup);
    LAUNCH("interleave", k_interleave, div_up(ra->nchunks * WAVE, BLOCK_THREADS), BLOCK_THREADS, a->view(), b->view(),
      ra->bits_as<const u64>(), ra->chunk_base.as<const u64>(), ra->nchunks, x->sup.as<const u64>(), x->recs.as<uint4>(), x->nrecs);
    return BWTM_OK;
  };
  int rc = body();
  if(rc != BWTM_OK) { delete x; return rc; }
  *out = x;
  return BWTM_OK;
}

extern "C" int bwtm_merge(const bwtm_index* a, const bwtm_index* b, bwtm_index** out)
{
  TRY(ensure_ready());
  if(!a || !b || !out) { return fail(BWTM_EINVAL, "bwtm_merge: null argument"); }
  bwtm_ra* ra = nullptr;
  TRY(bwtm_ra_create(a, b, &ra));
  int rc = BWTM_OK;
  if(b->m > 0) { rc = bwtm_search(a, b, 0, b->m - 1, ra); }
  if(rc == BWTM_OK) { rc = bwtm_ra_finalize(ra); }
  bwtm_index* x = nullptr;
  if(rc == BWTM_OK) { rc = bwtm_interleave(a, b, ra, &x); }
  if(rc == BWTM_OK) { rc = bwtm_index_encode(x); }
  bwtm_ra_free(ra);
  if(rc != BWTM_OK) { bwtm_index_free(x); return rc; }
  *out = x;
  return BWTM_OK;
}

//------------------------------------------------------------------------------
// Measurement.

extern "C" int bwtm_profile_enable(int on)
{
  TRY(ensure_ready());
  profile_collect();
  g_ctx.profiling = (on != 0);
  return BWTM_OK;
}

extern "C" int bwtm_profile_reset(void)
{
  profile_collect();
  g_ctx.totals.clear(); g_ctx.order.clear();
  return BWTM_OK;
}

extern "C" int bwtm_profile_read(const char** names, double* total_ms, uint64_t* launches, int capacity)
{
  profile_collect();
  int k = 0;
  for(const char* name : g_ctx.order)
  {
    if(k < capacity)
    {
      auto& t = g_ctx.totals[name];
      names[k] = name; total_ms[k] = t.first; launches[k] = t.second;
    }
    k++;
  }
  return k;
}

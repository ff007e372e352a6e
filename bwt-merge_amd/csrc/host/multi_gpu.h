/*
  multi_gpu.h -- FMI::FMI(a, b, parameters) on several GPUs of one node from ONE process: one host thread per GPU,
  each bound to its own library context (the reference's ParallelLoop workers, fmi.cpp:351-358, become threads
  that own a GPU each).  Two designs, one function each over a shared MergeJob:

  PARTITIONED RECORDS (the default for two GPUs and more; mergePartitionedWorker; include/bwtm.h bwtm_group_* / bwtm_part_*):
    nothing is replicated.  The merged order is cut at k-mer boundaries (the host's own rank queries on its FMIs); thread g uploads only the
    64-byte blocks that cover its windows of a and b, and the library runs the part's merge -- transcode of the windows, the search in lock
    step with the other parts (their step kernels read each other's output buffers: peer access over xGMI), boundary bits, range finalize,
    interleave and encode of its range of the output.  No bulk exchange, no collective library.  When a part runs out of room (cuts that
    balance positions do not bound the elements of a skewed collection) the merge is repeated with sequence blocks.

  SEQUENCE BLOCKS (MultiGPUMode::SequenceBlocks, bwt_merge -B; mergeBlocksWorker):
    1. every thread uploads 1 / G of each input's native bytes to its GPU and receives the other parts from its peers
       (all-gather over xGMI in up to eight rounds, the H2D copy of round j + 1 under the all-gather of round j), then decodes and
       transcodes the complete copy (the indexes are replicated);
    2. thread g searches block g of b's sequences (getBounds, utils.cpp:169-187) into its own bitvector;
    3. ONE bulk exchange: reduce-scatter (sum == or, the bits are disjoint) of the bitvectors by equal OUTPUT RANGES -- RCCL over
       xGMI (ncclReduceScatter in place on the buffer bwtm_ra_device_buffer() exposes); the few words a range needs from the others
       cross the threads through shared host variables;
    4. every thread interleaves and encodes only ITS range of the output (bwtm_interleave_range / bwtm_slice_*).
  Both end the same way (downloadSlice): every thread downloads its slice straight into its place in the result's page-locked arrays.

  Devices may repeat (e.g. {0, 0}): then the "GPUs" are contexts of one GPU (the partitioned merge needs nothing else; sequence blocks use
  bwtm_ra_or_from() instead of RCCL and CLEAR the bits outside a thread's own range afterwards, as a reduce-scatter leaves them undefined),
  so that every path is tested on a one-GPU box.

  Buffers that RCCL touches cannot come from the library's pool (its mapped blocks are device-local): the staging buffer of the sharded
  upload and the bitvector of the sequence-block path are plain hipMalloc blocks, kept per device in a process-wide cache (DeviceBuffers).
*/
#ifndef BWTM_HOST_MULTI_GPU_H
#define BWTM_HOST_MULTI_GPU_H

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <mutex>
#include <set>
#include <thread>
#include <unistd.h>

#include "fmi.h"
#ifdef BWTM_EXPERIMENTAL
#include "bwtm_experimental.h"
#endif

#ifdef BWTM_WITH_RCCL
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#endif

namespace bwtmerge
{

// Reusable barrier for a fixed number of threads (C++17 has none).
class ThreadBarrier
{
public:
  explicit ThreadBarrier(size_type n) : threads(n), waiting(0), generation(0) {}
  void wait()
  {
    std::unique_lock<std::mutex> lock(mu);
    size_type gen = generation;
    if(++waiting == threads) { waiting = 0; generation++; cv.notify_all(); }
    else { cv.wait(lock, [&] { return gen != generation; }); }
  }
private:
  std::mutex mu; std::condition_variable cv;
  size_type threads, waiting, generation;
};

#ifdef BWTM_WITH_RCCL
// Per-device hipMalloc blocks that survive a merge (slot 0: staging of an input's native bytes, slot 2: the rank-array bitvector).
class DeviceBuffers
{
public:
  static DeviceBuffers& instance() { static DeviceBuffers b; return b; }
  // A block of at least `bytes` on `device` (the current device of the calling thread is set to it); reuses the cached block of the
  // slot when it is large enough and not more than twice as large.
  void* get(int device, int slot, size_t bytes)
  {
    std::lock_guard<std::mutex> lock(mu);
    Entry& e = entries[std::make_pair(device, slot)];
    if(e.p && e.bytes >= bytes && e.bytes <= 2 * bytes + (64u << 20)) { return e.p; }
    if(hipSetDevice(device) != hipSuccess) { return nullptr; }
    if(e.p) { (void)hipFree(e.p); e.p = nullptr; e.bytes = 0; }
    if(hipMalloc(&e.p, bytes) != hipSuccess) { (void)hipGetLastError(); e.p = nullptr; return nullptr; }
    e.bytes = bytes;
    return e.p;
  }
  // After use: blocks above an eighth of the device's memory go back to the driver, the others stay for the next merge.
  void done(int device, int slot)
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = entries.find(std::make_pair(device, slot));
    if(it == entries.end() || !it->second.p) { return; }
    size_t free_b = 0, total_b = 0;
    if(hipSetDevice(device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = 0; }
    if(total_b == 0 || it->second.bytes > total_b / 8) { (void)hipFree(it->second.p); entries.erase(it); }
  }
  void releaseAll()
  {
    std::lock_guard<std::mutex> lock(mu);
    for(auto& kv : entries) { if(kv.second.p && hipSetDevice(kv.first.first) == hipSuccess) { (void)hipFree(kv.second.p); } }
    entries.clear();
  }
private:
  struct Entry { void* p = nullptr; size_t bytes = 0; };
  std::mutex mu;
  std::map<std::pair<int, int>, Entry> entries;
};
#endif

#ifdef BWTM_WITH_RCCL
// RCCL communicators and one collective stream per device, created once per device list and kept for the process: a chain of merges
// (bwt_merge in1 in2 in3 in4 out) pays ncclCommInitAll -- hundreds of milliseconds on eight GPUs -- and the stream creation once,
// not per merge (until round 4 every merge created and destroyed both).  Thread g of a merge uses comm(g) / stream(g) only.
class CollectiveCache
{
public:
  static CollectiveCache& instance() { static CollectiveCache c; return c; }
  struct Set { std::vector<ncclComm_t> comms; std::vector<hipStream_t> streams; };
  // The set for `devices` (distinct, more than one); nullptr and a message on stderr when RCCL or a stream cannot be set up.
  const Set* get(const std::vector<int>& devices)
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = sets.find(devices);
    if(it != sets.end()) { return &it->second; }
    Set set;
    set.comms.assign(devices.size(), nullptr); set.streams.assign(devices.size(), nullptr);
    if(ncclCommInitAll(set.comms.data(), (int)devices.size(), devices.data()) != ncclSuccess) { std::cerr << "mergeMultiGPU(): ncclCommInitAll failed" << std::endl; return nullptr; }
    for(size_t g = 0; g < devices.size(); g++)
    {
      if(hipSetDevice(devices[g]) != hipSuccess || hipStreamCreateWithFlags(&set.streams[g], hipStreamNonBlocking) != hipSuccess)
      {
        std::cerr << "mergeMultiGPU(): cannot create the collective stream of GPU " << devices[g] << std::endl; return nullptr;
      }
    }
    return &sets.emplace(devices, std::move(set)).first->second;
  }
  void releaseAll()
  {
    std::lock_guard<std::mutex> lock(mu);
    for(auto& kv : sets)
    {
      for(hipStream_t st : kv.second.streams) { if(st) { (void)hipStreamDestroy(st); } }
      for(ncclComm_t c : kv.second.comms) { if(c) { ncclCommDestroy(c); } }
    }
    sets.clear();
  }
private:
  std::mutex mu;
  std::map<std::vector<int>, Set> sets;
};
#endif

struct MultiGPUTimes
{
  uint64_t host_bytes_gpu0 = 0;          // native bytes GPU 0 received from the host (sharded upload: 1 / G of both inputs)
  uint64_t exchange_bytes = 0;           // bytes of bitvector every GPU sends and receives in the reduce-scatter: (G - 1) / G of one bitvector
  double upload = 0, search = 0, exchange = 0, interleave_encode = 0, download = 0, total = 0;   // seconds, thread 0's view
};

enum class MultiGPUMode
{
  Auto,                // two GPUs and more: partitioned records, repeated with sequence blocks when a part runs out of room
  Partitioned,         // partitioned records or failure
  SequenceBlocks,      // replicated records, blocks of b's sequences, reduce-scatter of the bitvector by output range
  Sliced               // (builds with -DBWTM_EXPERIMENTAL only) replicated records, every GPU advances a slice of the sorted frontier
};

namespace detail
{

// What the threads of one merge share.
struct MergeJob
{
  MergeJob(FMI& a_, FMI& b_, const std::vector<int>& devices_, FMI& result_) :
    a(a_), b(b_), devices(devices_), result(result_), G(devices_.size()), barrier(devices_.size()),
    adata(a_.bwt.hostData()), bdata(b_.bwt.hostData()), ca(a_.alpha.C.begin(), a_.alpha.C.end()), cb(b_.alpha.C.begin(), b_.alpha.C.end())
  {
    distinct = (std::set<int>(devices.begin(), devices.end()).size() == G);
    merged = a.alpha;
    for(size_type c = 0; c <= merged.sigma; c++) { merged.C[c] += b.alpha.C[c]; }
    nsup = ((a.size() + b.size()) >> 25) + 1;
    offsets.assign(G + 1, 0); first_block_start.assign(G, ~(uint64_t)0); block_first.assign(G, 0); block_count.assign(G, 0); next_block_start.assign(G, 0);
    host_bytes_per_gpu.assign(G, 0);
    bits.assign(G, nullptr); bits_bytes.assign(G, 0); heads.assign(G, 0); tables.assign(G * 64, 0);
    range_first.assign(G, 0); range_last.assign(G, 0); range_ones.assign(G, 0); super_local.assign(G * nsup, 0); tails.assign(G * 128, 0);
    cut_a.assign(G + 1, 0); cut_b.assign(G + 1, 0); part_rc.assign(G, BWTM_OK); part_error.assign(G, std::string());
  }
  FMI& a; FMI& b; const std::vector<int>& devices; FMI& result;
  const size_type G;
  ThreadBarrier barrier;
  const BlockArray& adata; const BlockArray& bdata;                  // (the inputs' device copies are dropped before the threads start)
  std::vector<uint64_t> ca, cb;
  bool distinct = false;
  Alphabet merged;
  size_type nsup = 0;                                                // super blocks of the output's rank structure
  double t0 = 0;
  MultiGPUTimes local;
  // the slices: byte offsets (offsets[G] = the stream's size), blocks, first block starts
  std::vector<uint64_t> offsets, first_block_start, block_first, block_count, next_block_start, host_bytes_per_gpu;
  // sequence blocks / sliced search
  std::vector<range_type> blocks;
  std::vector<void*> bits; std::vector<uint64_t> bits_bytes, heads, tables, range_first, range_last, range_ones, super_local, tails;
  bool sliced = false;
#ifdef BWTM_WITH_RCCL
  std::vector<ncclComm_t> comms; std::vector<hipStream_t> coll_streams;
  std::vector<void*> staging_a, staging_b;
#endif
#ifdef BWTM_EXPERIMENTAL
  std::vector<bwtm_fslice_view> views;
#endif
  // partitioned records
  std::vector<uint64_t> cut_a, cut_b;
  std::string group_name;
  std::vector<int> part_rc; std::vector<std::string> part_error;
};

inline void die(const std::string& what) { std::cerr << "mergeMultiGPU(): " << what << std::endl; std::exit(EXIT_FAILURE); }

// The end of every design: thread g's encoded slice into its place in the result's page-locked arrays.  J.offsets, J.block_first / _count and
// J.next_block_start[g] are known; thread 0 sizes the arrays.
inline void downloadSlice(MergeJob& J, size_type g, bwtm_slice* slice)
{
  BWT& out = J.result.bwt;
  if(g == 0)
  {
    const size_type nbytes = J.offsets[J.G], nblocks = (nbytes + Run::BLOCK_SIZE - 1) / Run::BLOCK_SIZE;
    out.data.bytes.resizeUninitialized(nbytes);
    out.block_end.resizeUninitialized(nblocks);
    out.cum_stride = nblocks + 1;
    out.cum_flat.resizeUninitialized(BWT::SIGMA * out.cum_stride);
    for(size_type c = 0; c < BWT::SIGMA; c++) { out.cum_flat[c * out.cum_stride + nblocks] = J.merged.C[c + 1] - J.merged.C[c]; }
  }
  J.barrier.wait();
  const double t_d = readTimer();
  gpuCheck(bwtm_slice_download_data(slice, out.data.bytes.data() + J.offsets[g], bwtm_slice_bytes(slice)), "mergeMultiGPU()");
  if(J.block_count[g] > 0)
  {
    HostArray<size_type> cum_local(BWT::SIGMA * J.block_count[g]);
    gpuCheck(bwtm_slice_download_samples(slice, J.next_block_start[g], out.block_end.data() + J.block_first[g], cum_local.data()), "mergeMultiGPU()");
    for(size_type c = 0; c < BWT::SIGMA; c++)
    {
      std::memcpy(out.cum_flat.data() + c * out.cum_stride + J.block_first[g], cum_local.data() + c * J.block_count[g], J.block_count[g] * sizeof(size_type));
    }
  }
  bwtm_slice_free(slice);
  J.barrier.wait();
  if(g == 0) { J.local.download = readTimer() - t_d; }
}

//------------------------------------------------------------------------------
// Partitioned records.

// The cuts (I_g, R_g) = (suffixes of a below w_g, suffixes of b below w_g) for G - 1 of the 5^k k-mers, chosen so that the parts' shares of the
// output are as equal as the candidates allow; sp(c w) = C[c] + rank_c(sp(w)) on the host's indexes (BWT::rank of the facade, bwt.h).
inline void partitionCuts(MergeJob& J)
{
  const size_type G = J.G;
  J.cut_a[G] = J.a.size(); J.cut_b[G] = J.b.size();
  if(G == 1) { return; }
  const unsigned k = (G <= 8 ? 4 : 5);
  auto points = [&](const FMI& x) -> std::vector<uint64_t>
  {
    std::vector<uint64_t> sp(1, 0);
    for(unsigned round = 0; round < k; round++)
    {
      std::vector<uint64_t> next; next.reserve(5 * sp.size());
      for(comp_type c = 1; c <= 5; c++) { for(uint64_t p : sp) { next.push_back(x.alpha.C[c] + x.bwt.rank(p, c)); } }
      sp.swap(next);
    }
    return sp;
  };
  const std::vector<uint64_t> pa = points(J.a), pb = points(J.b);
  const double total = (double)J.a.size() + (double)J.b.size();
  for(size_type g = 1; g < G; g++)
  {
    size_type best = 0; double dist = -1;
    for(size_type j = 0; j < pa.size(); j++)
    {
      const double d = std::abs((double)pa[j] + (double)pb[j] - total * g / G);
      if(dist < 0 || d < dist) { dist = d; best = j; }
    }
    J.cut_a[g] = std::max(J.cut_a[g - 1], pa[best]); J.cut_b[g] = std::max(J.cut_b[g - 1], pb[best]);
  }
}

// Thread g of the partitioned merge.  Failures of the part calls are recorded (J.part_rc), not fatal: the caller may fall back.
inline void mergePartitionedWorker(MergeJob& J, size_type g)
{
  bwtm_context* ctx = nullptr;
  gpuCheck(bwtm_context_create(J.devices[g], &ctx), "mergeMultiGPU()");
  gpuCheck(bwtm_context_make_current(ctx), "mergeMultiGPU()");
  bwtm_group* group = nullptr; bwtm_part* part = nullptr; bwtm_slice* slice = nullptr;
  uint64_t offset = 0, total = 0, next = 0;
  auto step = [&](int rc) -> bool
  {
    if(rc == BWTM_OK) { return true; }
    if(J.part_rc[g] == BWTM_OK) { J.part_rc[g] = rc; J.part_error[g] = bwtm_last_error(); }
    if(group) { bwtm_group_abort(group); }
    return false;
  };
  auto blockStartOf = [](const BWT& x, size_type k) { uint64_t p = 0; for(size_type c = 0; c < BWT::SIGMA; c++) { p += x.cum(c, k); } return p; };
  // the blocks [b0, b1) of a native stream whose records cover the positions [lo, hi]: block starts from the samples (bwt.cpp:489-511)
  auto upload = [&](int which, const BWT& x, const BlockArray& data) -> int
  {
    uint64_t lo = 0, hi = 0;
    int rc = bwtm_part_window(part, which, &lo, &hi);
    if(rc != BWTM_OK) { return rc; }
    const size_type nb = x.blocks();
    const uint64_t first = lo & ~(uint64_t)127, end = std::min<uint64_t>(x.size(), (hi | 127) + 1);
    size_type l = 0, r = nb;                                         // last block that begins at or before `first`
    while(r - l > 1) { const size_type mid = (l + r) / 2; if(blockStartOf(x, mid) <= first) { l = mid; } else { r = mid; } }
    const size_type b0 = l;
    l = b0; r = nb;                                                   // first block that begins at or after `end` (nb: none)
    while(l < r) { const size_type mid = (l + r) / 2; if(blockStartOf(x, mid) >= end) { r = mid; } else { l = mid + 1; } }
    const size_type b1 = std::max<size_type>(l, b0 + 1);
    uint64_t before[6];
    for(size_type c = 0; c < 6; c++) { before[c] = x.cum(c, b0); }
    const uint64_t from = (uint64_t)b0 * Run::BLOCK_SIZE, to = std::min<uint64_t>((uint64_t)b1 * Run::BLOCK_SIZE, data.size());
    J.host_bytes_per_gpu[g] += to - from;
    return bwtm_part_upload(part, which, data.data() + from, to - from, blockStartOf(x, b0), before, 0);
  };
  bwtm_index_header ha, hb;
  ha.bases = J.a.size(); ha.sequences = J.a.sequences(); hb.bases = J.b.size(); hb.sequences = J.b.sequences();
  for(size_type c = 0; c <= 6; c++) { ha.C[c] = J.ca[c]; hb.C[c] = J.cb[c]; }
  bool ok = step(bwtm_group_create(J.G > 1 ? J.group_name.c_str() : nullptr, (int)g, (int)J.G, &group));
  ok = ok && step(bwtm_part_create(group, &ha, &hb, J.cut_a.data(), J.cut_b.data(), &part));
  ok = ok && step(upload(0, J.a.bwt, J.adata)) && step(upload(1, J.b.bwt, J.bdata));
  if(g == 0) { J.local.upload = readTimer() - J.t0; }
  ok = ok && step(bwtm_part_search(part));
  if(g == 0) { J.local.search = readTimer() - J.t0 - J.local.upload; }
  const double t_i = readTimer();
  ok = ok && step(bwtm_part_finish(part, &slice, &offset, &total, &next));
  if(ok)
  {
    J.offsets[g] = offset; J.offsets[J.G] = total; J.next_block_start[g] = next;
    J.block_first[g] = bwtm_slice_block_first(slice); J.block_count[g] = bwtm_slice_blocks(slice);
    if(g == 0)
    {
      bwtm_part_info info;
      if(bwtm_part_stats(part, &info) == BWTM_OK) { J.local.exchange_bytes = info.pulled_bytes; J.local.exchange = info.ms_search_wait / 1e3; }
      J.local.interleave_encode = readTimer() - t_i;
    }
  }
  if(part) { bwtm_part_free(part); }
  J.barrier.wait();                                                   // every thread has succeeded or recorded its failure
  bool all_ok = true;
  for(size_type h = 0; h < J.G; h++) { all_ok = all_ok && (J.part_rc[h] == BWTM_OK); }
  if(all_ok) { downloadSlice(J, g, slice); }
  else if(slice) { bwtm_slice_free(slice); }
  if(group) { bwtm_group_free(group); }
  if(g == 0 && all_ok) { warnIfPoolExhausted("mergeMultiGPU()"); }
  gpuCheck(bwtm_context_make_current(nullptr), "mergeMultiGPU()");
  bwtm_context_destroy(ctx);
}

//------------------------------------------------------------------------------
// Sequence blocks (and, in experimental builds, the sliced search over the same replicated records).

#ifdef BWTM_WITH_RCCL
// One input, sharded over the links: staging[g] = this GPU's full-size device buffer (hipMalloc: peers and RCCL may touch it).
inline bwtm_index* uploadSharded(MergeJob& J, const BlockArray& data, size_type sequences, size_type bases, const uint64_t* C, size_type g, std::vector<void*>& staging, int slot,
  ncclComm_t comm, hipStream_t stream)
{
  const size_type G = J.G;
  // The stream is cut into K x G pieces of `sub` bytes; piece (j, h) = bytes [(j G + h) sub, (j G + h + 1) sub) is uploaded by GPU h in round j.
  // The pieces of one round are contiguous, so round j's all-gather is ONE in-place ncclAllGather over [j G sub, (j + 1) G sub) -- and
  // while it runs over xGMI, this GPU's piece of round j + 1 crosses PCIe.
  const uint64_t nbytes = data.size();
  uint64_t K = std::max<uint64_t>(1, std::min<uint64_t>(8, nbytes / (G * (64ull << 20))));
  if(const char* v = std::getenv("BWTM_SHARDED_UPLOAD_ROUNDS")) { const long r = std::atol(v); if(r >= 1 && r <= 64) { K = (uint64_t)r; } }   // tests: several rounds on small inputs
  const uint64_t sub = ((nbytes + G * K - 1) / (G * K) + 255) / 256 * 256;  // equal pieces (the collective wants them), 256-byte aligned
  const uint64_t staged = sub * G * K + 16;
  auto check = [&](hipError_t e, const char* what) { if(e != hipSuccess) { die(std::string(what) + ": " + hipGetErrorString(e)); } };
  check(hipSetDevice(J.devices[g]), "hipSetDevice");
  // contexts of one GPU share the device: every thread needs its own block there, so only distinct devices use the cache
  if(J.distinct) { staging[g] = DeviceBuffers::instance().get(J.devices[g], slot, staged); if(!staging[g]) { check(hipErrorOutOfMemory, "staging buffer"); } }
  else { check(hipMalloc(&staging[g], staged), "hipMalloc of the staging buffer"); }
  check(hipMemset((char*)staging[g] + nbytes, 0, staged - nbytes), "hipMemset");               // readable zeros behind the stream
  auto piece = [&](uint64_t j, uint64_t h, uint64_t& off, uint64_t& len)
  {
    off = std::min<uint64_t>((j * G + h) * sub, nbytes); len = std::min<uint64_t>(sub, nbytes - off);
  };
  if(comm)
  {
    hipStream_t copy = nullptr; hipEvent_t arrived = nullptr;
    check(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking), "hipStreamCreate");
    check(hipEventCreateWithFlags(&arrived, hipEventDisableTiming), "hipEventCreate");
    check(hipDeviceSynchronize(), "hipDeviceSynchronize");                                     // the memset above ran on the null stream
    J.barrier.wait();                                                                         // every GPU's buffer is ready to receive
    for(uint64_t j = 0; j < K; j++)
    {
      uint64_t off, len; piece(j, g, off, len);
      if(len > 0) { check(hipMemcpyAsync((char*)staging[g] + off, data.data() + off, len, hipMemcpyHostToDevice, copy), "H2D copy of this GPU's piece"); J.host_bytes_per_gpu[g] += len; }
      check(hipEventRecord(arrived, copy), "hipEventRecord");
      check(hipStreamWaitEvent(stream, arrived, 0), "hipStreamWaitEvent");                     // round j's collective waits for round j's piece only
      if(ncclAllGather((char*)staging[g] + (j * G + g) * sub, (char*)staging[g] + j * G * sub, sub, ncclUint8, comm, stream) != ncclSuccess) { die("ncclAllGather failed"); }
    }
    check(hipStreamSynchronize(stream), "all-gather"); check(hipStreamSynchronize(copy), "H2D copies");
    (void)hipEventDestroy(arrived); (void)hipStreamDestroy(copy);
    check(hipMemset((char*)staging[g] + nbytes, 0, staged - nbytes), "hipMemset");             // the padding of the last pieces travelled too
    check(hipDeviceSynchronize(), "hipDeviceSynchronize");
  }
  else
  {
    for(uint64_t j = 0; j < K; j++)
    {
      uint64_t off, len; piece(j, g, off, len);
      if(len > 0) { check(hipMemcpy((char*)staging[g] + off, data.data() + off, len, hipMemcpyHostToDevice), "H2D copy of this GPU's piece"); J.host_bytes_per_gpu[g] += len; }
    }
    J.barrier.wait();                                                                         // every piece is on its device
    for(uint64_t j = 0; j < K; j++)
    {
      for(size_type h = 0; h < G; h++)
      {
        uint64_t off, len; piece(j, h, off, len);
        if(h != g && len > 0) { check(hipMemcpy((char*)staging[g] + off, (const char*)staging[h] + off, len, hipMemcpyDeviceToDevice), "device-to-device copy of a peer's piece"); }
      }
    }
    // a device-to-device hipMemcpy returns before the copy has run (it is ordered on the null stream only), and the library decodes the
    // buffer on a stream of its own: without this the first decode pass raced the copies on inputs of a few gigabytes
    check(hipDeviceSynchronize(), "hipDeviceSynchronize");
  }
  J.barrier.wait();                                                       // nobody reads a peer's buffer any more
  bwtm_index* x = nullptr;
  gpuCheck(bwtm_index_from_device_borrowed(staging[g], nbytes, sequences, bases, C, &x), "mergeMultiGPU()");
  gpuCheck(bwtm_index_drop_native(x), "mergeMultiGPU()");                 // synchronizes: the staging buffer is free again
  if(J.distinct) { DeviceBuffers::instance().done(J.devices[g], slot); } else { check(hipFree(staging[g]), "hipFree"); }
  staging[g] = nullptr;
  return x;
}
#endif

// Step 1: both inputs, whole, as device rank structures of thread g's GPU.
inline void uploadReplicated(MergeJob& J, size_type g, bwtm_index** A, bwtm_index** B)
{
#ifdef BWTM_WITH_RCCL
  if(J.G > 1)
  {
    // Sharded upload: this GPU's PCIe link carries only 1 / G of each input's native bytes; the other parts arrive from the peers
    // (all-gather over xGMI, or device-to-device copies between contexts of one GPU); every GPU then decodes and transcodes its copy.
    *A = uploadSharded(J, J.adata, J.a.sequences(), J.a.size(), J.ca.data(), g, J.staging_a, 0, (J.distinct ? J.comms[g] : nullptr), (J.distinct ? J.coll_streams[g] : nullptr));
    *B = uploadSharded(J, J.bdata, J.b.sequences(), J.b.size(), J.cb.data(), g, J.staging_b, 0, (J.distinct ? J.comms[g] : nullptr), (J.distinct ? J.coll_streams[g] : nullptr));   // one cached block serves both
    return;
  }
#endif
  gpuCheck(bwtm_index_upload(J.adata.data(), J.adata.size(), J.a.sequences(), J.a.size(), J.ca.data(), A), "mergeMultiGPU()");
  gpuCheck(bwtm_index_drop_native(*A), "mergeMultiGPU()");
  gpuCheck(bwtm_index_upload(J.bdata.data(), J.bdata.size(), J.b.sequences(), J.b.size(), J.cb.data(), B), "mergeMultiGPU()");
  gpuCheck(bwtm_index_drop_native(*B), "mergeMultiGPU()");
}

// Step 3: every GPU receives the union of all shards' bits inside ITS output range [g, g + 1) * shard_bytes.
inline void exchangeBitvector(MergeJob& J, size_type g, bwtm_ra* ra, uint64_t shard_bytes)
{
  if(J.G == 1) { return; }
  if(J.distinct)
  {
#ifdef BWTM_WITH_RCCL
    hipStream_t stream = J.coll_streams[g];
    if(hipSetDevice(J.devices[g]) != hipSuccess) { die("hipSetDevice failed"); }
    if(ncclReduceScatter(J.bits[g], (char*)J.bits[g] + g * shard_bytes, shard_bytes / sizeof(uint64_t), ncclUint64, ncclSum, J.comms[g], stream) != ncclSuccess) { die("ncclReduceScatter failed"); }
    if(hipStreamSynchronize(stream) != hipSuccess) { die("the reduce-scatter failed"); }
#endif
  }
  else
  {
    // contexts of one GPU: thread 0 collects all shards, everybody takes the union from it -- and then forgets what lies outside
    // its own range, as after a reduce-scatter
    if(g == 0) { for(size_type h = 1; h < J.G; h++) { gpuCheck(bwtm_ra_or_from(ra, J.bits[h], J.bits_bytes[h]), "mergeMultiGPU()"); } }
    J.barrier.wait();
    if(g != 0) { gpuCheck(bwtm_ra_or_from(ra, J.bits[0], J.bits_bytes[0]), "mergeMultiGPU()"); }
    J.barrier.wait();
#ifdef BWTM_WITH_RCCL
    // (the share a reduce-scatter would have delivered; the null stream does not order with the library's streams, hence the device-wide synchronisation)
    const uint64_t lo = std::min<uint64_t>(g * shard_bytes, J.bits_bytes[g]), hi = std::min<uint64_t>((g + 1) * shard_bytes, J.bits_bytes[g]);
    if(hipSetDevice(J.devices[g]) != hipSuccess || hipMemset(J.bits[g], 0xA5, lo) != hipSuccess ||
       hipMemset((char*)J.bits[g] + hi, 0xA5, J.bits_bytes[g] - hi) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { die("hipMemset failed"); }
#endif
  }
  J.barrier.wait();
}

// Step 4: thread g's range of the output from the whole indexes and a bitvector that is complete inside the range.
inline bwtm_slice* interleaveEncodeRange(MergeJob& J, size_type g, bwtm_index* A, bwtm_index* B, bwtm_ra* ra, uint64_t rec_first, uint64_t rec_last)
{
  const size_type G = J.G, nsup = J.nsup;
  // The small exchange: set bits of every range, local offsets of the supers that start in it, its last chunk of bits.
  gpuCheck(bwtm_ra_range_counts(ra, rec_first, rec_last, &J.range_ones[g], J.super_local.data() + g * nsup, J.tails.data() + g * 128), "mergeMultiGPU()");
  J.barrier.wait();
  uint64_t before = 0, total = 0;
  for(size_type h = 0; h < G; h++) { if(h < g) { before += J.range_ones[h]; } total += J.range_ones[h]; }
  std::vector<uint64_t> super_boff(nsup, 0);
  for(size_type sb = 0; sb < nsup; sb++)
  {
    const uint64_t q = (uint64_t)sb << 18;                           // the super's first record
    uint64_t prefix = 0;
    for(size_type h = 0; h < G; h++)
    {
      if(q >= J.range_first[h] && q < J.range_last[h]) { super_boff[sb] = prefix + J.super_local[h * nsup + sb]; break; }
      prefix += J.range_ones[h];
    }
  }
  const uint64_t* halo = nullptr;
  for(size_type h = g; h-- > 0; ) { if(J.range_last[h] > J.range_first[h]) { halo = J.tails.data() + h * 128; break; } }
  gpuCheck(bwtm_ra_finalize_range(ra, rec_first, rec_last, before, total, super_boff.data(), halo), "mergeMultiGPU()");
  bwtm_slice* slice = nullptr;
  gpuCheck(bwtm_interleave_range(A, B, ra, rec_first, rec_last, &slice), "mergeMultiGPU()");
  bwtm_ra_free(ra); bwtm_index_free(A); bwtm_index_free(B);
  // the encoder's two carries (open run, byte offset mod 64) cross the threads through shared host variables
  gpuCheck(bwtm_slice_lasthead(slice, &J.heads[g]), "mergeMultiGPU()");
  J.barrier.wait();
  uint64_t head_before = 0;
  for(size_type h = 0; h < g; h++) { head_before = std::max(head_before, J.heads[h]); }
  gpuCheck(bwtm_slice_size_table(slice, head_before, J.tables.data() + 64 * g), "mergeMultiGPU()");
  J.barrier.wait();
  if(g == 0) { gpuCheck(bwtm_fold_offsets(J.tables.data(), (int)G, J.offsets.data()), "mergeMultiGPU()"); }
  J.barrier.wait();
  gpuCheck(bwtm_slice_encode(slice, J.offsets[g]), "mergeMultiGPU()");
  gpuCheck(bwtm_slice_first_block_start(slice, &J.first_block_start[g]), "mergeMultiGPU()");
  J.block_first[g] = bwtm_slice_block_first(slice); J.block_count[g] = bwtm_slice_blocks(slice);
  J.barrier.wait();
  uint64_t next = J.a.size() + J.b.size();
  for(size_type h = G; h-- > g + 1; ) { if(J.first_block_start[h] != ~(uint64_t)0) { next = J.first_block_start[h]; } }
  J.next_block_start[g] = next;
  return slice;
}

#ifdef BWTM_EXPERIMENTAL
// The sliced frontier search (bwtm_fslice_*, include/bwtm_experimental.h): GPU g advances slice g of the sorted frontier and pulls its next
// slice from all GPUs' outputs -- every GPU then streams 1 / G of both rank structures per LF step instead of a thinned 100 %.
inline void searchSliced(MergeJob& J, size_type g, bwtm_index* A, bwtm_index* B, bwtm_ra* ra)
{
  const size_type G = J.G;
  const uint64_t capacity = (J.b.sequences() + G - 1) / G + 1;
  bwtm_fslice* fs = nullptr;
  gpuCheck(bwtm_fslice_create(A, B, ra, capacity, (int)G, &fs), "mergeMultiGPU()");
  gpuCheck(bwtm_fslice_seed(fs, (g < J.blocks.size() ? J.blocks[g].first : 0), (g < J.blocks.size() ? J.blocks[g].second - J.blocks[g].first + 1 : 0)), "mergeMultiGPU()");
  gpuCheck(bwtm_fslice_export(fs, &J.views[g]), "mergeMultiGPU()");
  J.barrier.wait();
  while(true)
  {
    uint64_t total = 0;
    for(size_type h = 0; h < G; h++) { for(int c = 0; c < 5; c++) { total += J.views[h].totals[c]; } }
    if(total == 0) { break; }
    const uint64_t per = (total + G - 1) / G, first = std::min<uint64_t>(total, g * per), last = std::min<uint64_t>(total, (g + 1) * per);
    gpuCheck(bwtm_fslice_gather(fs, J.views.data(), (int)G, first, last), "mergeMultiGPU()");
    J.barrier.wait();                                                // every GPU has pulled its slice: the outputs may be overwritten
    gpuCheck(bwtm_fslice_advance(fs), "mergeMultiGPU()");
    gpuCheck(bwtm_fslice_export(fs, &J.views[g]), "mergeMultiGPU()");
    J.barrier.wait();                                                // every GPU's new outputs and totals are visible
  }
  gpuCheck(bwtm_fslice_finish(fs), "mergeMultiGPU()");
  bwtm_fslice_free(fs);
}
#endif

// Thread g of the merge over replicated records.
inline void mergeBlocksWorker(MergeJob& J, size_type g)
{
  const size_type G = J.G;
  bwtm_context* ctx = nullptr;
  gpuCheck(bwtm_context_create(J.devices[g], &ctx), "mergeMultiGPU()");
  gpuCheck(bwtm_context_make_current(ctx), "mergeMultiGPU()");
  bwtm_index *A = nullptr, *B = nullptr; bwtm_ra* ra = nullptr;
  uploadReplicated(J, g, &A, &B);
  if(g == 0) { J.local.upload = readTimer() - J.t0; }

  // Equal output ranges (the collective wants equal shares): range g = records [rec_first, rec_last), shard_bytes of bitvector each.
  uint64_t rec_first = 0, rec_last = 0, shard_bytes = 0;
  gpuCheck(bwtm_slice_bounds_equal(bwtm_merged_records(A, B), (int)G, (int)g, &rec_first, &rec_last, &shard_bytes), "mergeMultiGPU()");
  J.range_first[g] = rec_first; J.range_last[g] = rec_last;
  // Across devices the bitvector is handed to ncclReduceScatter, which may let a peer GPU read or write the buffer directly: the library's
  // pooled blocks are mapped for their own device only, so this buffer is a cached hipMalloc block.
  bool shared_bits = false;
  if(J.distinct && G > 1)
  {
#ifdef BWTM_WITH_RCCL
    const uint64_t need = G * shard_bytes;                           // >= bwtm_ra_buffer_bytes(A, B): zero words behind the bitvector
    void* p = DeviceBuffers::instance().get(J.devices[g], 2, need);
    if(!p || hipSetDevice(J.devices[g]) != hipSuccess || hipMemset(p, 0, need) != hipSuccess) { die("cannot allocate the rank-array bitvector"); }
    gpuCheck(bwtm_ra_create_on(A, B, p, need, &ra), "mergeMultiGPU()");
    shared_bits = true;
#endif
  }
  else { gpuCheck(bwtm_ra_create(A, B, &ra), "mergeMultiGPU()"); }
#ifdef BWTM_EXPERIMENTAL
  if(J.sliced && J.b.sequences() > 0) { searchSliced(J, g, A, B, ra); }
  else
#endif
  if(g < J.blocks.size()) { gpuCheck(bwtm_search(A, B, J.blocks[g].first, J.blocks[g].second, ra), "mergeMultiGPU()"); }
  gpuCheck(bwtm_ra_device_buffer(ra, &J.bits[g], &J.bits_bytes[g]), "mergeMultiGPU()");      // synchronizes: the search is done
  if(g == 0) { J.local.search = readTimer() - J.t0 - J.local.upload; }

  J.barrier.wait();
  const double t_x = readTimer();
  exchangeBitvector(J, g, ra, shard_bytes);
  if(g == 0) { J.local.exchange = readTimer() - t_x; J.local.exchange_bytes = (G > 1 ? (G - 1) * shard_bytes : 0); }

  const double t_i = readTimer();
  bwtm_slice* slice = interleaveEncodeRange(J, g, A, B, ra, rec_first, rec_last);
#ifdef BWTM_WITH_RCCL
  if(shared_bits) { DeviceBuffers::instance().done(J.devices[g], 2); }
#else
  (void)shared_bits;
#endif
  if(g == 0) { J.local.interleave_encode = readTimer() - t_i; }
  downloadSlice(J, g, slice);
  if(g == 0) { warnIfPoolExhausted("mergeMultiGPU()"); }
  gpuCheck(bwtm_context_make_current(nullptr), "mergeMultiGPU()");
  bwtm_context_destroy(ctx);
}

template<class Worker>
inline void runThreads(MergeJob& J, Worker worker)
{
  std::vector<std::thread> threads;
  for(size_type g = 1; g < J.G; g++) { threads.emplace_back(worker, std::ref(J), g); }
  worker(J, 0);
  for(std::thread& t : threads) { t.join(); }
}

} // namespace detail

// Merges a and b (both consumed) into `result` using the given devices (which may repeat: contexts of one GPU), one host thread per device.
inline void mergeMultiGPU(FMI& a, FMI& b, const std::vector<int>& devices, FMI& result, MultiGPUTimes* times = nullptr, MultiGPUMode mode = MultiGPUMode::Auto)
{
  if(a.alpha != b.alpha)
  {
    std::cerr << "FMI::FMI(): Cannot merge BWTs with different alphabets" << std::endl;
    std::exit(EXIT_FAILURE);
  }
  if(devices.empty()) { detail::die("no devices"); }
  if(devices.size() > BWTM_MAX_PARTS && mode != MultiGPUMode::SequenceBlocks) { detail::die("at most 16 GPUs"); }
#ifndef BWTM_EXPERIMENTAL
  if(mode == MultiGPUMode::Sliced) { detail::die("the sliced search is not part of this build"); }
#endif
  (void)a.bwt.hostData(); (void)b.bwt.hostData();
  a.bwt.dropDevice(); b.bwt.dropDevice();
  detail::MergeJob J(a, b, devices, result);
  J.t0 = readTimer();
  const uint64_t input_bytes = J.adata.size() + J.bdata.size();
  bool done = false;

  if(mode == MultiGPUMode::Auto || mode == MultiGPUMode::Partitioned)
  {
    static std::atomic<unsigned> counter{0};
    J.group_name = "/bwtm-" + std::to_string((unsigned long)getpid()) + "-" + std::to_string(counter.fetch_add(1));
    detail::partitionCuts(J);
    detail::runThreads(J, detail::mergePartitionedWorker);
    done = true;
    bool capacity_only = true;
    for(size_type g = 0; g < J.G; g++)
    {
      if(J.part_rc[g] == BWTM_OK) { continue; }
      done = false;
      if(J.part_rc[g] != BWTM_ENOMEM && J.part_rc[g] != BWTM_EPEER) { capacity_only = false; }
    }
    if(!done)
    {
      std::string why;
      for(size_type g = 0; g < J.G; g++) { if(J.part_rc[g] != BWTM_OK && J.part_rc[g] != BWTM_EPEER) { why = J.part_error[g]; break; } }
      if(why.empty()) { for(size_type g = 0; g < J.G; g++) { if(J.part_rc[g] != BWTM_OK) { why = J.part_error[g]; break; } } }
      if(mode == MultiGPUMode::Partitioned || !capacity_only) { detail::die("the merge over partitioned records failed: " + why); }
      std::cerr << "mergeMultiGPU(): the merge over partitioned records ran out of room (" << why << "); repeating it with sequence blocks" << std::endl;
      J.local = MultiGPUTimes(); J.t0 = readTimer();
      std::fill(J.host_bytes_per_gpu.begin(), J.host_bytes_per_gpu.end(), 0);
    }
  }
  if(!done)
  {
    J.sliced = (mode == MultiGPUMode::Sliced);
    J.blocks = (b.sequences() > 0 ? getBounds(range_type(0, b.sequences() - 1), J.G) : std::vector<range_type>());
#ifdef BWTM_WITH_RCCL
    J.comms.assign(J.G, nullptr); J.coll_streams.assign(J.G, nullptr); J.staging_a.assign(J.G, nullptr); J.staging_b.assign(J.G, nullptr);
    if(J.distinct && J.G > 1)
    {
      const CollectiveCache::Set* set = CollectiveCache::instance().get(devices);       // created by the first merge on these devices, reused afterwards
      if(!set) { std::exit(EXIT_FAILURE); }
      J.comms = set->comms; J.coll_streams = set->streams;
    }
#else
    if(J.distinct && J.G > 1) { detail::die("built without RCCL: sequence blocks cannot combine rank arrays across devices"); }
#endif
#if defined(BWTM_WITH_RCCL) && defined(BWTM_EXPERIMENTAL)
    J.views.resize(J.G);
    if(J.distinct && J.G > 1 && J.sliced)
    {
      // the sliced search reads its peers' frontier buffers directly (plain hipMalloc memory, bwtm_fslice_export): peers must be mapped
      for(size_type i = 0; i < J.G; i++)
      {
        for(size_type j = 0; j < J.G; j++)
        {
          if(i == j) { continue; }
          int can = 0;
          if(hipSetDevice(devices[i]) != hipSuccess || hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess || !can) { detail::die("the sliced search needs peer access between the GPUs"); }
          hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
          if(e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { detail::die(std::string("hipDeviceEnablePeerAccess failed: ") + hipGetErrorString(e)); }
          (void)hipGetLastError();
        }
      }
    }
#elif defined(BWTM_EXPERIMENTAL)
    J.views.resize(J.G);
#endif
    detail::runThreads(J, detail::mergeBlocksWorker);
  }

  BWT& out = result.bwt;
  out.header.sequences = a.sequences() + b.sequences();
  out.header.bases = a.size() + b.size();
  out.header.setOrder(a.bwt.header.order());
  out.adoptHost(nullptr, out.block_end.size());
  result.alpha = J.merged;
  a.bwt.clear(); b.bwt.clear();
  J.local.total = readTimer() - J.t0;
  J.local.host_bytes_gpu0 = (J.G > 1 ? J.host_bytes_per_gpu[0] : input_bytes);
  if(J.G == 1 && J.host_bytes_per_gpu[0] > 0) { J.local.host_bytes_gpu0 = J.host_bytes_per_gpu[0]; }
  if(times) { *times = J.local; }
}

} // namespace bwtmerge

#endif // BWTM_HOST_MULTI_GPU_H

/*
  multi_gpu.h -- FMI::FMI(a, b, parameters) on several GPUs of one node from ONE process: one host thread per GPU,
  each bound to its own library context (the reference's ParallelLoop workers, fmi.cpp:351-358, become threads
  that own a GPU each).

    1. every thread uploads 1 / G of each input's native bytes to its GPU and receives the other parts from its peers
       (all-gather over xGMI: the PCIe links carry every byte once instead of G times; in up to eight rounds, the H2D copy of round
       j + 1 under the all-gather of round j), then decodes and transcodes the complete copy (the indexes are replicated: every LF
       chain touches arbitrary positions of both);
    2. thread g searches block g of b's sequences (getBounds, utils.cpp:169-187) into its own bitvector;
    3. ONE bulk exchange: reduce-scatter (sum == or, the bits are disjoint) of the bitvectors by equal OUTPUT RANGES -- RCCL over
       xGMI, called directly (ncclReduceScatter in place on the buffer bwtm_ra_device_buffer() exposes): every GPU receives only the
       range it will interleave, half the bytes of an all-reduce; the few words a range needs from the others (set bits before
       it, offsets of the output's super blocks, the chunk of bits before it: bwtm_ra_range_counts / bwtm_ra_finalize_range)
       cross the threads through shared host variables;
    4. every thread interleaves and encodes only ITS range of the output (bwtm_interleave_range / bwtm_slice_*);
       the two encoder carries (open run, byte offset mod 64) cross the threads through shared host variables;
    5. every thread downloads its slice straight into its place in the result's page-locked arrays
       (eight D2H streams, each 1 / G of the output).

  Devices may repeat (e.g. {0, 0}): then the "GPUs" are contexts of one GPU and step 3 uses bwtm_ra_or_from()
  instead of RCCL, after which every thread CLEARS the bits outside its own range -- what a reduce-scatter leaves undefined --
  so that the range logic is tested on a one-GPU box.

  Buffers that RCCL or a peer GPU touches cannot come from the library's pool (its mapped blocks are device-local): the staging
  buffer of the sharded upload and the bitvector are plain hipMalloc blocks, kept per device in a process-wide cache
  (DeviceBuffers) and reused by the next merge of a chain instead of being allocated and freed per merge -- a hipMalloc that has
  to wait for deferred frees takes seconds on MI355X.  Blocks above an eighth of the device's memory (the
  native bytes of a 200 Gbase input) are released as soon as their phase is over: holding them would not leave room for the records.
*/
#ifndef BWTM_HOST_MULTI_GPU_H
#define BWTM_HOST_MULTI_GPU_H

#include <condition_variable>
#include <cstdlib>
#include <map>
#include <mutex>
#include <set>
#include <thread>

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

// Merges a and b (both consumed) into `result` using the given devices.
// sliced = false: GPU g searches block g of b's sequences (bwtm_search).  sliced = true (only in builds with -DBWTM_EXPERIMENTAL against
// libbwtm_experimental.so): the sliced frontier search (bwtm_fslice_*, include/bwtm_experimental.h): GPU g advances slice g of the sorted
// frontier and pulls its next slice from all GPUs' outputs -- every GPU then streams 1 / G of both rank structures per LF step
// instead of a thinned 100 % (contexts of one GPU, or devices with peer access).
// partitioned = true (same builds only): nothing is replicated -- every GPU transcodes one window of each input from its own
// share of the native bytes (the blocks the inputs' samples name for a position range), holds its own range of the bitvector, and the frontier's
// nodes and elements travel to the GPU that owns their position (cuts at k-mer boundaries, found with the host's rank queries); the ranges of
// the output follow the cuts, so there is no bulk exchange afterwards, only the earlier parts' bits inside a part's first segment (8 KiB).
inline void mergeMultiGPU(FMI& a, FMI& b, const std::vector<int>& devices, FMI& result, MultiGPUTimes* times = nullptr, bool sliced = false, bool partitioned = false)
{
  if(a.alpha != b.alpha)
  {
    std::cerr << "FMI::FMI(): Cannot merge BWTs with different alphabets" << std::endl;
    std::exit(EXIT_FAILURE);
  }
  const size_type G = devices.size();
  if(G == 0) { std::cerr << "mergeMultiGPU(): no devices" << std::endl; std::exit(EXIT_FAILURE); }
  const bool distinct = (std::set<int>(devices.begin(), devices.end()).size() == G);

  Alphabet merged = a.alpha;
  for(size_type c = 0; c <= merged.sigma; c++) { merged.C[c] += b.alpha.C[c]; }
  const BlockArray& adata = a.bwt.hostData(); const BlockArray& bdata = b.bwt.hostData();
  a.bwt.dropDevice(); b.bwt.dropDevice();
  const uint64_t input_bytes = adata.size() + bdata.size();
  std::vector<uint64_t> ca(a.alpha.C.begin(), a.alpha.C.end()), cb(b.alpha.C.begin(), b.alpha.C.end());
  std::vector<range_type> blocks = (b.sequences() > 0 ? getBounds(range_type(0, b.sequences() - 1), G) : std::vector<range_type>());

#ifdef BWTM_WITH_RCCL
  std::vector<ncclComm_t> comms(G, nullptr);
  std::vector<hipStream_t> coll_streams(G, nullptr);
  if(distinct && G > 1)
  {
    const CollectiveCache::Set* set = CollectiveCache::instance().get(devices);       // created by the first merge on these devices, reused afterwards
    if(!set) { std::exit(EXIT_FAILURE); }
    comms = set->comms; coll_streams = set->streams;
  }
#else
  if(distinct && G > 1) { std::cerr << "mergeMultiGPU(): built without RCCL, cannot combine rank arrays across devices" << std::endl; std::exit(EXIT_FAILURE); }
#endif

#ifndef BWTM_EXPERIMENTAL
  if(sliced || partitioned) { std::cerr << "mergeMultiGPU(): the sliced / partitioned search is not part of this build" << std::endl; std::exit(EXIT_FAILURE); }
#endif
#if defined(BWTM_EXPERIMENTAL) && !defined(BWTM_WITH_RCCL)
  if(partitioned) { std::cerr << "mergeMultiGPU(): the partitioned merge needs the HIP runtime headers (built without RCCL)" << std::endl; std::exit(EXIT_FAILURE); }
#endif
#if defined(BWTM_WITH_RCCL) && defined(BWTM_EXPERIMENTAL)
  if(distinct && G > 1 && (sliced || partitioned))
  {
    // the sliced search reads its peers' frontier buffers directly (plain hipMalloc memory, bwtm_fslice_export): peers must be mapped
    for(size_type i = 0; i < G; i++)
    {
      for(size_type j = 0; j < G; j++)
      {
        if(i == j) { continue; }
        int can = 0;
        if(hipSetDevice(devices[i]) != hipSuccess || hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess || !can)
        {
          std::cerr << "mergeMultiGPU(): GPU " << devices[i] << " cannot access GPU " << devices[j] << ": the sliced search needs peer access" << std::endl;
          std::exit(EXIT_FAILURE);
        }
        hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
        if(e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { std::cerr << "mergeMultiGPU(): hipDeviceEnablePeerAccess failed: " << hipGetErrorString(e) << std::endl; std::exit(EXIT_FAILURE); }
        (void)hipGetLastError();
      }
    }
  }
#endif

  // What the threads share.
  ThreadBarrier barrier(G);
  std::vector<void*> bits(G, nullptr); std::vector<uint64_t> bits_bytes(G, 0);
  std::vector<uint64_t> heads(G, 0), tables(G * 64, 0), offsets(G + 1, 0), first_block_start(G, ~(uint64_t)0);
  std::vector<uint64_t> block_first(G, 0), block_count(G, 0);
  const size_type nsup = ((a.size() + b.size()) >> 25) + 1;                     // super blocks of the output's rank structure
  std::vector<uint64_t> range_first(G, 0), range_last(G, 0), range_ones(G, 0), super_local(G * nsup, 0), tails(G * 128, 0);
#ifdef BWTM_EXPERIMENTAL
  std::vector<bwtm_fslice_view> views(G);
  std::vector<bwtm_fslice_nodes_view> nviews(G);
  std::vector<bwtm_ra*> part_ra(G, nullptr);
  std::vector<void*> boundary_stage(G, nullptr);
  // Partitioned records: the cuts (I_g, R_g) = (suffixes of a below w_g, suffixes of b below w_g) for G - 1 of the 5^k k-mers, chosen so that the parts'
  // shares of the output are as equal as the candidates allow; sp(c w) = C[c] + rank_c(sp(w)) on the host's indexes (support.h / bwt.h queries).
  std::vector<uint64_t> cut_a(G + 1, 0), cut_b(G + 1, 0);
  if(partitioned)
  {
    cut_a[G] = a.size(); cut_b[G] = b.size();
    auto points = [&](const FMI& x, unsigned k) -> std::vector<uint64_t>
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
    if(G > 1)
    {
      const unsigned k = 5;
      const std::vector<uint64_t> pa = points(a, k), pb = points(b, k);
      const double total = (double)a.size() + (double)b.size();
      for(size_type g = 1; g < G; g++)
      {
        size_type best = 0; double dist = -1;
        for(size_type j = 0; j < pa.size(); j++)
        {
          const double d = std::abs((double)pa[j] + (double)pb[j] - total * g / G);
          if(dist < 0 || d < dist) { dist = d; best = j; }
        }
        cut_a[g] = std::max(cut_a[g - 1], pa[best]); cut_b[g] = std::max(cut_b[g - 1], pb[best]);
      }
    }
  }
  // the blocks [b0, b1) of a native stream whose records cover the positions [lo, hi]: block starts from the samples (bwt.cpp:489-511)
  auto blockStartOf = [](const BWT& x, size_type k) { uint64_t p = 0; for(size_type c = 0; c < BWT::SIGMA; c++) { p += x.cum(c, k); } return p; };
  auto blocksFor = [&](const BWT& x, uint64_t lo, uint64_t hi, size_type& b0, size_type& b1)
  {
    const size_type nb = x.blocks();
    const uint64_t first = lo & ~(uint64_t)127, end = std::min<uint64_t>(x.size(), (hi | 127) + 1);
    size_type l = 0, r = nb;                                   // last block that begins at or before `first`
    while(r - l > 1) { const size_type mid = (l + r) / 2; if(blockStartOf(x, mid) <= first) { l = mid; } else { r = mid; } }
    b0 = l;
    l = b0; r = nb;                                             // first block that begins at or after `end` (nb: none)
    while(l < r) { const size_type mid = (l + r) / 2; if(blockStartOf(x, mid) >= end) { r = mid; } else { l = mid + 1; } }
    b1 = std::max<size_type>(l, b0 + 1);
  };
#endif
  BWT& out = result.bwt;
  double t0 = readTimer();
  MultiGPUTimes local;

#ifdef BWTM_WITH_RCCL
  // One input, sharded over the links: staging[g] = this GPU's full-size device buffer (hipMalloc: peers and RCCL may touch it).
  std::vector<void*> staging_a(G, nullptr), staging_b(G, nullptr);
  std::vector<uint64_t> host_bytes_per_gpu(G, 0);
  auto uploadSharded = [&](const BlockArray& data, size_type sequences, size_type bases, const uint64_t* C, size_type g, std::vector<void*>& staging, int slot, ncclComm_t comm, hipStream_t stream) -> bwtm_index*
  {
    // The stream is cut into K x G pieces of `sub` bytes; piece (j, h) = bytes [(j G + h) sub, (j G + h + 1) sub) is uploaded by GPU h in round j.
    // The pieces of one round are contiguous, so round j's all-gather is ONE in-place ncclAllGather over [j G sub, (j + 1) G sub) -- and
    // while it runs over xGMI, this GPU's piece of round j + 1 crosses PCIe (round 5: until then the whole 1 / G part was copied with one
    // synchronous hipMemcpy and the all-gather started only after every GPU had finished; small inputs still take one round).
    const uint64_t nbytes = data.size();
    uint64_t K = std::max<uint64_t>(1, std::min<uint64_t>(8, nbytes / (G * (64ull << 20))));
    if(const char* v = std::getenv("BWTM_SHARDED_UPLOAD_ROUNDS")) { const long r = std::atol(v); if(r >= 1 && r <= 64) { K = (uint64_t)r; } }   // tests: several rounds on small inputs
    const uint64_t sub = ((nbytes + G * K - 1) / (G * K) + 255) / 256 * 256;  // equal pieces (the collective wants them), 256-byte aligned
    const uint64_t staged = sub * G * K + 16;
    auto check = [&](hipError_t e, const char* what) { if(e != hipSuccess) { std::cerr << "mergeMultiGPU(): " << what << ": " << hipGetErrorString(e) << std::endl; std::exit(EXIT_FAILURE); } };
    check(hipSetDevice(devices[g]), "hipSetDevice");
    // contexts of one GPU share the device: every thread needs its own block there, so only distinct devices use the cache
    if(distinct) { staging[g] = DeviceBuffers::instance().get(devices[g], slot, staged); if(!staging[g]) { check(hipErrorOutOfMemory, "staging buffer"); } }
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
      barrier.wait();                                                                           // every GPU's buffer is ready to receive
      for(uint64_t j = 0; j < K; j++)
      {
        uint64_t off, len; piece(j, g, off, len);
        if(len > 0) { check(hipMemcpyAsync((char*)staging[g] + off, data.data() + off, len, hipMemcpyHostToDevice, copy), "H2D copy of this GPU's piece"); host_bytes_per_gpu[g] += len; }
        check(hipEventRecord(arrived, copy), "hipEventRecord");
        check(hipStreamWaitEvent(stream, arrived, 0), "hipStreamWaitEvent");                     // round j's collective waits for round j's piece only
        if(ncclAllGather((char*)staging[g] + (j * G + g) * sub, (char*)staging[g] + j * G * sub, sub, ncclUint8, comm, stream) != ncclSuccess)
        {
          std::cerr << "mergeMultiGPU(): ncclAllGather failed" << std::endl; std::exit(EXIT_FAILURE);
        }
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
        if(len > 0) { check(hipMemcpy((char*)staging[g] + off, data.data() + off, len, hipMemcpyHostToDevice), "H2D copy of this GPU's piece"); host_bytes_per_gpu[g] += len; }
      }
      barrier.wait();                                                                           // every piece is on its device
      for(uint64_t j = 0; j < K; j++)
      {
        for(size_type h = 0; h < G; h++)
        {
          uint64_t off, len; piece(j, h, off, len);
          if(h != g && len > 0) { check(hipMemcpy((char*)staging[g] + off, (const char*)staging[h] + off, len, hipMemcpyDeviceToDevice), "device-to-device copy of a peer's piece"); }
        }
      }
      // a device-to-device hipMemcpy returns before the copy has run (it is ordered on the null stream only), and the library decodes the
      // buffer on a stream of its own: without this the first decode pass raced the copies on inputs of a few gigabytes (round 5:
      // "native stream decodes to 1897126518 positions, header says 2020000000" with -g 0,0 at 2 x 2 Gbase; small inputs never showed it)
      check(hipDeviceSynchronize(), "hipDeviceSynchronize");
    }
    barrier.wait();                                                         // nobody reads a peer's buffer any more
    bwtm_index* x = nullptr;
    gpuCheck(bwtm_index_from_device_borrowed(staging[g], nbytes, sequences, bases, C, &x), "mergeMultiGPU()");
    gpuCheck(bwtm_index_drop_native(x), "mergeMultiGPU()");                 // synchronizes: the staging buffer is free again
    if(distinct) { DeviceBuffers::instance().done(devices[g], slot); } else { check(hipFree(staging[g]), "hipFree"); }
    staging[g] = nullptr;
    return x;
  };
#endif

  auto worker = [&](size_type g)
  {
    bwtm_context* ctx = nullptr;
    gpuCheck(bwtm_context_create(devices[g], &ctx), "mergeMultiGPU()");
    gpuCheck(bwtm_context_make_current(ctx), "mergeMultiGPU()");
    bwtm_index *A = nullptr, *B = nullptr; bwtm_ra* ra = nullptr; bwtm_slice* slice = nullptr;
#ifdef BWTM_EXPERIMENTAL
    const uint64_t PART_MARGIN = 2 * 65536;                        // positions a part reads beyond its cuts in the merge's second half: a segment + the halo chunk
    if(partitioned)
    {
      auto window = [&](const BWT& x, const BlockArray& data, uint64_t lo, uint64_t hi, const uint64_t* C, bwtm_index** w)
      {
        size_type b0 = 0, b1 = 0;
        blocksFor(x, lo, hi, b0, b1);
        uint64_t before[6];
        for(size_type c = 0; c < 6; c++) { before[c] = x.cum(c, b0); }
        const uint64_t from = (uint64_t)b0 * Run::BLOCK_SIZE, to = std::min<uint64_t>((uint64_t)b1 * Run::BLOCK_SIZE, data.size());
        gpuCheck(bwtm_x_index_upload_window(data.data() + from, to - from, blockStartOf(x, b0), before, x.size(), x.sequences(), C, w), "mergeMultiGPU()");
#ifdef BWTM_WITH_RCCL
        host_bytes_per_gpu[g] += to - from;
#endif
      };
      window(a.bwt, adata, (cut_a[g] > PART_MARGIN ? cut_a[g] - PART_MARGIN : 0), std::min<uint64_t>(a.size(), cut_a[g + 1] + PART_MARGIN), ca.data(), &A);
      window(b.bwt, bdata, (cut_b[g] > PART_MARGIN ? cut_b[g] - PART_MARGIN : 0), std::min<uint64_t>(b.size(), cut_b[g + 1] + PART_MARGIN), cb.data(), &B);
    }
    else
#endif
#ifdef BWTM_WITH_RCCL
    if(G > 1)
    {
      // Sharded upload: this GPU's PCIe link carries only 1 / G of each input's native bytes; the other parts arrive from the
      // peers (all-gather over xGMI, or device-to-device copies between contexts of one GPU); every GPU then decodes and
      // transcodes its complete device copy (BWT::load, ~9 ms per 5 Gbase input).
      A = uploadSharded(adata, a.sequences(), a.size(), ca.data(), g, staging_a, 0, (distinct ? comms[g] : nullptr), (distinct ? coll_streams[g] : nullptr));
      B = uploadSharded(bdata, b.sequences(), b.size(), cb.data(), g, staging_b, 0, (distinct ? comms[g] : nullptr), (distinct ? coll_streams[g] : nullptr));      // the inputs are staged one after the other: one cached block serves both
    }
    else
#endif
    {
      gpuCheck(bwtm_index_upload(adata.data(), adata.size(), a.sequences(), a.size(), ca.data(), &A), "mergeMultiGPU()");
      gpuCheck(bwtm_index_drop_native(A), "mergeMultiGPU()");
      gpuCheck(bwtm_index_upload(bdata.data(), bdata.size(), b.sequences(), b.size(), cb.data(), &B), "mergeMultiGPU()");
      gpuCheck(bwtm_index_drop_native(B), "mergeMultiGPU()");
    }
    if(g == 0) { local.upload = readTimer() - t0; }

    // Equal output ranges (the collective wants equal shares): range g = records [rec_first, rec_last), shard_bytes of bitvector each.
    uint64_t rec_first = 0, rec_last = 0, shard_bytes = 0;
    gpuCheck(bwtm_slice_bounds_equal(bwtm_merged_records(A, B), (int)G, (int)g, &rec_first, &rec_last, &shard_bytes), "mergeMultiGPU()");
#ifdef BWTM_EXPERIMENTAL
    // partitioned: the ranges follow the cuts, rounded down to the encoder's 65 536-position segments (512 records)
    auto partPosition = [&](size_type h) { return cut_a[h] + cut_b[h]; };
    auto partSegment = [&](size_type h) { return (h == 0 ? (uint64_t)0 : partPosition(h) >> 16); };
    if(partitioned)
    {
      const uint64_t nrecs = bwtm_merged_records(A, B);
      rec_first = std::min<uint64_t>(nrecs, partSegment(g) * 512);
      rec_last = (g + 1 == G ? nrecs : std::min<uint64_t>(nrecs, partSegment(g + 1) * 512));
      shard_bytes = 0;
    }
#endif
    range_first[g] = rec_first; range_last[g] = rec_last;
    // Across devices the bitvector is handed to ncclReduceScatter, which (all ranks in one process) may let a peer GPU read or write the
    // buffer directly: the library's pooled blocks are mapped for their own device only, so this buffer is a cached hipMalloc block.
    void* shared_bits = nullptr;
#ifdef BWTM_EXPERIMENTAL
    if(partitioned)
    {
      gpuCheck(bwtm_x_ra_create_range(A, B, partPosition(g), partPosition(g + 1), &ra), "mergeMultiGPU()");
      part_ra[g] = ra;
    }
    else
#endif
    if(distinct && G > 1)
    {
#ifdef BWTM_WITH_RCCL
      const uint64_t need = G * shard_bytes;                       // >= bwtm_ra_buffer_bytes(A, B): zero words behind the bitvector
      shared_bits = DeviceBuffers::instance().get(devices[g], 2, need);
      if(!shared_bits || hipSetDevice(devices[g]) != hipSuccess || hipMemset(shared_bits, 0, need) != hipSuccess)
      {
        std::cerr << "mergeMultiGPU(): cannot allocate the rank-array bitvector" << std::endl; std::exit(EXIT_FAILURE);
      }
      gpuCheck(bwtm_ra_create_on(A, B, shared_bits, need, &ra), "mergeMultiGPU()");
#endif
    }
    else { gpuCheck(bwtm_ra_create(A, B, &ra), "mergeMultiGPU()"); }
#ifdef BWTM_EXPERIMENTAL
    if(partitioned && b.sequences() > 0)
    {
      const uint64_t m = b.sequences(), limit = m / 8;
      const bool roots_on_one = (G == 1 || cut_b[1] >= m);        // k-mer cuts: all roots lie below the first cut
      const bool nodes = (limit >= 1 && roots_on_one);
      // with the node phase a part holds ~m / G elements; small inputs (and a search from the roots on) may put everything on one part
      const uint64_t capacity = (nodes && m >= (1u << 20) ? (uint64_t)(1.5 * m / G) + 65536 : m + 1);
      const uint64_t node_capacity = (m >= (1u << 20) ? (uint64_t)(1.5 * std::min<uint64_t>(5 * limit, m) / G) + 65536 : std::min<uint64_t>(5 * std::max<uint64_t>(limit, 1), m) + 1);
      bwtm_fslice* fs = nullptr;
      gpuCheck(bwtm_fslice_create(A, B, ra, capacity, (int)G, &fs), "mergeMultiGPU()");
      gpuCheck(bwtm_fslice_set_cuts(fs, cut_b.data(), (int)G), "mergeMultiGPU()");
      const uint64_t root_first = std::min<uint64_t>(cut_b[g], m), root_last = std::min<uint64_t>(cut_b[g + 1], m);
      if(nodes)
      {
        gpuCheck(bwtm_fslice_nodes_begin(fs, root_first, root_last - root_first, node_capacity), "mergeMultiGPU()");
        uint64_t level = 1;
        while(level > 0 && level <= limit)
        {
          gpuCheck(bwtm_fslice_nodes_step(fs, &nviews[g]), "mergeMultiGPU()");
          barrier.wait();                                          // every GPU's children and their counts are visible
          gpuCheck(bwtm_fslice_nodes_gather(fs, nviews.data(), (int)G, (int)g), "mergeMultiGPU()");
          level = 0;
          for(size_type h = 0; h < G; h++) { level += nviews[h].class_first[5]; }
          barrier.wait();                                          // every GPU has taken its nodes: the children may be overwritten
        }
        gpuCheck(bwtm_fslice_nodes_expand(fs), "mergeMultiGPU()");
      }
      else { gpuCheck(bwtm_fslice_seed(fs, root_first, root_last - root_first), "mergeMultiGPU()"); }
      gpuCheck(bwtm_fslice_export(fs, &views[g]), "mergeMultiGPU()");
      barrier.wait();
      while(true)
      {
        uint64_t total = 0;
        for(size_type h = 0; h < G; h++) { for(int c = 0; c < 5; c++) { total += views[h].totals[c]; } }
        if(total == 0) { break; }
        gpuCheck(bwtm_fslice_gather_cut(fs, views.data(), (int)G, (int)g), "mergeMultiGPU()");
        barrier.wait();                                            // every GPU has pulled its elements: the send buffers may be overwritten
        gpuCheck(bwtm_fslice_advance(fs), "mergeMultiGPU()");
        gpuCheck(bwtm_fslice_export(fs, &views[g]), "mergeMultiGPU()");
        barrier.wait();
      }
      gpuCheck(bwtm_fslice_finish(fs), "mergeMultiGPU()");
      bwtm_fslice_free(fs);
    }
    else if(partitioned) { }
    else if(sliced && b.sequences() > 0)
    {
      const uint64_t capacity = (b.sequences() + G - 1) / G + 1;
      bwtm_fslice* fs = nullptr;
      gpuCheck(bwtm_fslice_create(A, B, ra, capacity, (int)G, &fs), "mergeMultiGPU()");
      gpuCheck(bwtm_fslice_seed(fs, (g < blocks.size() ? blocks[g].first : 0), (g < blocks.size() ? blocks[g].second - blocks[g].first + 1 : 0)), "mergeMultiGPU()");
      gpuCheck(bwtm_fslice_export(fs, &views[g]), "mergeMultiGPU()");
      barrier.wait();
      while(true)
      {
        uint64_t total = 0;
        for(size_type h = 0; h < G; h++) { for(int c = 0; c < 5; c++) { total += views[h].totals[c]; } }
        if(total == 0) { break; }
        const uint64_t per = (total + G - 1) / G, first = std::min<uint64_t>(total, g * per), last = std::min<uint64_t>(total, (g + 1) * per);
        gpuCheck(bwtm_fslice_gather(fs, views.data(), (int)G, first, last), "mergeMultiGPU()");
        barrier.wait();                                            // every GPU has pulled its slice: the outputs may be overwritten
        gpuCheck(bwtm_fslice_advance(fs), "mergeMultiGPU()");
        gpuCheck(bwtm_fslice_export(fs, &views[g]), "mergeMultiGPU()");
        barrier.wait();                                            // every GPU's new outputs and totals are visible
      }
      gpuCheck(bwtm_fslice_finish(fs), "mergeMultiGPU()");
      bwtm_fslice_free(fs);
    }
    else
#endif
    if(g < blocks.size()) { gpuCheck(bwtm_search(A, B, blocks[g].first, blocks[g].second, ra), "mergeMultiGPU()"); }
    if(partitioned) { gpuCheck(bwtm_synchronize(), "mergeMultiGPU()"); }
    else { gpuCheck(bwtm_ra_device_buffer(ra, &bits[g], &bits_bytes[g]), "mergeMultiGPU()"); }      // synchronizes: the search is done
    if(g == 0) { local.search = readTimer() - t0 - local.upload; }

    // The one bulk exchange: every GPU receives the union of all shards inside ITS output range.
    barrier.wait();
    double t_x = readTimer();
#if defined(BWTM_EXPERIMENTAL) && defined(BWTM_WITH_RCCL)
    if(partitioned && G > 1)
    {
      // What crosses a boundary: the bits of the parts before it inside a part's first segment.  Every part writes its share of every later
      // part's first segment into a plain hipMalloc block (8 KiB per pair: peers can read it), the owner ORs them in.
      const uint64_t seg_bytes = 65536 / 8;
      auto hipOk = [&](hipError_t e, const char* what) { if(e != hipSuccess) { std::cerr << "mergeMultiGPU(): " << what << ": " << hipGetErrorString(e) << std::endl; std::exit(EXIT_FAILURE); } };
      hipOk(hipSetDevice(devices[g]), "hipSetDevice");
      hipOk(hipMalloc(&boundary_stage[g], G * seg_bytes), "hipMalloc of the boundary block");
      auto touches = [&](size_type h, size_type k)                 // part h has bits inside part k's first segment (h < k)
      {
        const uint64_t lo = std::max<uint64_t>(partSegment(k) << 16, partPosition(h)), hi = std::min<uint64_t>(partPosition(k), partPosition(h + 1));
        return lo < hi;
      };
      for(size_type k = g + 1; k < G; k++)
      {
        if(touches(g, k)) { gpuCheck(bwtm_x_ra_read_words(ra, partSegment(k) << 16, (partSegment(k) << 16) + 65536, (char*)boundary_stage[g] + k * seg_bytes), "mergeMultiGPU()"); }
      }
      barrier.wait();
      for(size_type h = 0; h < g; h++)
      {
        if(touches(h, g)) { gpuCheck(bwtm_x_ra_or_words(ra, partSegment(g) << 16, (partSegment(g) << 16) + 65536, (const char*)boundary_stage[h] + g * seg_bytes), "mergeMultiGPU()"); }
      }
      barrier.wait();
      hipOk(hipFree(boundary_stage[g]), "hipFree"); boundary_stage[g] = nullptr;
      if(g == 0) { local.exchange_bytes = seg_bytes; }
    }
    else
#endif
    if(G > 1)
    {
      if(distinct)
      {
#ifdef BWTM_WITH_RCCL
        hipStream_t stream = coll_streams[g];
        if(hipSetDevice(devices[g]) != hipSuccess) { std::cerr << "mergeMultiGPU(): hipSetDevice failed" << std::endl; std::exit(EXIT_FAILURE); }
        if(ncclReduceScatter(bits[g], (char*)bits[g] + g * shard_bytes, shard_bytes / sizeof(uint64_t), ncclUint64, ncclSum, comms[g], stream) != ncclSuccess)
        {
          std::cerr << "mergeMultiGPU(): ncclReduceScatter failed" << std::endl; std::exit(EXIT_FAILURE);
        }
        if(hipStreamSynchronize(stream) != hipSuccess) { std::cerr << "mergeMultiGPU(): the reduce-scatter failed" << std::endl; std::exit(EXIT_FAILURE); }
#endif
      }
      else
      {
        // contexts of one GPU: thread 0 collects all shards, everybody takes the union from it -- and then forgets what lies outside
        // its own range, as after a reduce-scatter
        if(g == 0) { for(size_type h = 1; h < G; h++) { gpuCheck(bwtm_ra_or_from(ra, bits[h], bits_bytes[h]), "mergeMultiGPU()"); } }
        barrier.wait();
        if(g != 0) { gpuCheck(bwtm_ra_or_from(ra, bits[0], bits_bytes[0]), "mergeMultiGPU()"); }
        barrier.wait();
#ifdef BWTM_WITH_RCCL
        // (the share a reduce-scatter would have delivered: [g, g + 1) * shard_bytes; the null stream does not order with the library's
        // streams, hence the device-wide synchronisation)
        const uint64_t lo = std::min<uint64_t>(g * shard_bytes, bits_bytes[g]), hi = std::min<uint64_t>((g + 1) * shard_bytes, bits_bytes[g]);
        if(hipSetDevice(devices[g]) != hipSuccess || hipMemset(bits[g], 0xA5, lo) != hipSuccess ||
           hipMemset((char*)bits[g] + hi, 0xA5, bits_bytes[g] - hi) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
        {
          std::cerr << "mergeMultiGPU(): hipMemset failed" << std::endl; std::exit(EXIT_FAILURE);
        }
#endif
      }
      barrier.wait();
    }
    // The small exchange: set bits of every range, local offsets of the supers that start in it, its last chunk of bits.
    gpuCheck(bwtm_ra_range_counts(ra, rec_first, rec_last, &range_ones[g], super_local.data() + g * nsup, tails.data() + g * 128), "mergeMultiGPU()");
    barrier.wait();
    if(g == 0) { local.exchange = readTimer() - t_x; if(!partitioned) { local.exchange_bytes = (G > 1 ? (G - 1) * shard_bytes : 0); } }

    // This thread's range of the output.
    double t_i = readTimer();
    {
      uint64_t before = 0, total = 0;
      for(size_type h = 0; h < G; h++) { if(h < g) { before += range_ones[h]; } total += range_ones[h]; }
      std::vector<uint64_t> super_boff(nsup, 0);
      for(size_type sb = 0; sb < nsup; sb++)
      {
        const uint64_t q = (uint64_t)sb << 18;                     // the super's first record
        uint64_t prefix = 0;
        for(size_type h = 0; h < G; h++)
        {
          if(q >= range_first[h] && q < range_last[h]) { super_boff[sb] = prefix + super_local[h * nsup + sb]; break; }
          prefix += range_ones[h];
        }
      }
      const uint64_t* halo = nullptr;
      for(size_type h = g; h-- > 0; ) { if(range_last[h] > range_first[h]) { halo = tails.data() + h * 128; break; } }
      gpuCheck(bwtm_ra_finalize_range(ra, rec_first, rec_last, before, total, super_boff.data(), halo), "mergeMultiGPU()");
    }
    gpuCheck(bwtm_interleave_range(A, B, ra, rec_first, rec_last, &slice), "mergeMultiGPU()");
    bwtm_ra_free(ra); bwtm_index_free(A); bwtm_index_free(B);
#ifdef BWTM_WITH_RCCL
    if(shared_bits) { DeviceBuffers::instance().done(devices[g], 2); }
#endif
    gpuCheck(bwtm_slice_lasthead(slice, &heads[g]), "mergeMultiGPU()");
    barrier.wait();
    uint64_t before = 0;
    for(size_type h = 0; h < g; h++) { before = std::max(before, heads[h]); }
    gpuCheck(bwtm_slice_size_table(slice, before, tables.data() + 64 * g), "mergeMultiGPU()");
    barrier.wait();
    if(g == 0) { gpuCheck(bwtm_fold_offsets(tables.data(), (int)G, offsets.data()), "mergeMultiGPU()"); }
    barrier.wait();
    gpuCheck(bwtm_slice_encode(slice, offsets[g]), "mergeMultiGPU()");
    gpuCheck(bwtm_slice_first_block_start(slice, &first_block_start[g]), "mergeMultiGPU()");
    block_first[g] = bwtm_slice_block_first(slice); block_count[g] = bwtm_slice_blocks(slice);
    if(g == 0)
    {
      // the result's arrays, sized now that the stream's length is known
      const size_type nbytes = offsets[G], nblocks = (nbytes + Run::BLOCK_SIZE - 1) / Run::BLOCK_SIZE;
      out.data.bytes.resizeUninitialized(nbytes);
      out.block_end.resizeUninitialized(nblocks);
      out.cum_stride = nblocks + 1;
      out.cum_flat.resizeUninitialized(BWT::SIGMA * out.cum_stride);
      for(size_type c = 0; c < BWT::SIGMA; c++) { out.cum_flat[c * out.cum_stride + nblocks] = merged.C[c + 1] - merged.C[c]; }
      local.interleave_encode = readTimer() - t_i;
    }
    barrier.wait();

    // Download: every slice into its place.
    double t_d = readTimer();
    gpuCheck(bwtm_slice_download_data(slice, out.data.bytes.data() + offsets[g], bwtm_slice_bytes(slice)), "mergeMultiGPU()");
    if(block_count[g] > 0)
    {
      uint64_t next = a.size() + b.size();
      for(size_type h = G; h-- > g + 1; ) { if(first_block_start[h] != ~(uint64_t)0) { next = first_block_start[h]; } }
      HostArray<size_type> cum_local(BWT::SIGMA * block_count[g]);
      gpuCheck(bwtm_slice_download_samples(slice, next, out.block_end.data() + block_first[g], cum_local.data()), "mergeMultiGPU()");
      for(size_type c = 0; c < BWT::SIGMA; c++)
      {
        std::memcpy(out.cum_flat.data() + c * out.cum_stride + block_first[g], cum_local.data() + c * block_count[g], block_count[g] * sizeof(size_type));
      }
    }
    bwtm_slice_free(slice);
    barrier.wait();
    if(g == 0) { local.download = readTimer() - t_d; }
    if(g == 0) { warnIfPoolExhausted("mergeMultiGPU()"); }
    gpuCheck(bwtm_context_make_current(nullptr), "mergeMultiGPU()");
    bwtm_context_destroy(ctx);
  };

  std::vector<std::thread> threads;
  for(size_type g = 1; g < G; g++) { threads.emplace_back(worker, g); }
  worker(0);
  for(std::thread& t : threads) { t.join(); }

  out.header.sequences = a.sequences() + b.sequences();
  out.header.bases = a.size() + b.size();
  out.header.setOrder(a.bwt.header.order());
  out.adoptHost(nullptr, out.block_end.size());
  result.alpha = merged;
  a.bwt.clear(); b.bwt.clear();
  local.total = readTimer() - t0;
  local.host_bytes_gpu0 = input_bytes;
#ifdef BWTM_WITH_RCCL
  if(G > 1) { local.host_bytes_gpu0 = host_bytes_per_gpu[0]; }
#endif
  if(times) { *times = local; }
}

} // namespace bwtmerge

#endif // BWTM_HOST_MULTI_GPU_H

/*
  api/group.hip.h -- the PARTS of one merge over partitioned records (api/pmerge.hip.h): up to 16 threads or processes, one per GPU, that
  meet in a block of shared host memory.  Part of bwtm_api.hip.

  The reference fans its search out to threads of one process that share everything (ParallelLoop, fmi.cpp:351-358, utils.cpp:189-218).
  Here a part owns a GPU and may live in a process of its own (bench.py's contract: one process per GPU), so what the parts share is
  explicit and small:
    * a control block in POSIX shared memory (shm_open; plain heap memory for a group of one): per-part sequence counters for the barrier,
      an abort flag, and two banks of per-part blobs for all-gathers of a few KB (the step's cut table, range counts, encoder carries);
    * every part's exported device ARENA (one hipMalloc block: its output buffers of the search, its node lists, its boundary bits), as a
      raw pointer for parts of the same process and a HIP IPC handle for the others; a part maps its peers' arenas once and re-maps one
      only when its generation changes (the arena was re-allocated for a larger merge).
  No collective library is involved: the elements of the frontier cross between GPUs as loads of peer-mapped memory inside the step
  kernel (xGMI on a node with several GPUs), everything else is a few hundred bytes per step through the control block.

  A part that fails sets the abort flag; the others leave their barrier with BWTM_EPEER instead of waiting for it, and every wait has a
  deadline (BWTM_GROUP_TIMEOUT seconds, default 300).
*/
#pragma once

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace
{

constexpr u32 GROUP_MAGIC = 0x62774731u;           // "bwG1"
constexpr u64 GROUP_BLOB = 16384;                  // bytes per part and bank of the all-gather area

struct GroupArena
{
  std::atomic<u64> generation;                     // 0 = nothing exported yet
  u64 pid, raw, bytes;
  int device;
  int has_handle;
  hipIpcMemHandle_t handle;
};

struct GroupShared
{
  std::atomic<u32> magic;
  u32 parts;
  std::atomic<u32> attached;
  std::atomic<u32> abort_flag;
  std::atomic<u64> seq[PART_MAX];                  // barriers passed by every part
  std::atomic<u64> ticket;                         // BWTM_GROUP_SERIAL: compute sections run one part at a time, in the order (section, part)
  GroupArena arena[PART_MAX];
  alignas(64) u8 blob[2][PART_MAX][GROUP_BLOB];
};

double group_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

double group_timeout()
{
  static const double t = []() { const char* v = std::getenv("BWTM_GROUP_TIMEOUT"); const double x = (v ? std::atof(v) : 0.0); return x > 0 ? x : 300.0; }();
  return t;
}

} // namespace

struct bwtm_group
{
  GroupShared* sh = nullptr;
  bool mapped = false;                             // shm mapping (false: heap, a group of one)
  int part = 0, parts = 1;
  u64 my_seq = 0;                                  // barriers this part has entered
  u64 gathers = 0;
  // the peers' arenas as this part maps them
  struct Peer { u64 generation = 0; void* ptr = nullptr; bool opened = false; };
  Peer peer[PART_MAX];
  // this part's arena (hipMalloc: the pool's mapped blocks cannot be exported), kept across merges
  void* arena = nullptr; u64 arena_bytes = 0; int arena_device = -1;
  // page-locked staging of this part's searches (the step's plan and cut tables, small vectors): allocated once, not per merge
  // (a hipHostMalloc takes a fraction of a millisecond; a merge of 1 / 8 of config 2 takes twenty)
  char* pinned = nullptr;
  double wait_seconds = 0;                         // time spent waiting for peers (statistics)
  // BWTM_GROUP_SERIAL=1 (measurements on ONE GPU whose contexts stand in for GPUs): every compute section of a part -- the kernels between two
  // exchanges -- runs alone on the device, so that a part's kernel times are what its own GPU would need (tools/parts_scale.py)
  bool serial = false; u64 sections = 0;
};

namespace
{

int group_wait(bwtm_group* g, u64 target)
{
  GroupShared* sh = g->sh;
  const double t0 = group_now();
  u32 spins = 0;
  for(int h = 0; h < g->parts; h++)
  {
    while(sh->seq[h].load(std::memory_order_acquire) < target)
    {
      if(sh->abort_flag.load(std::memory_order_relaxed) != 0) { return fail(BWTM_EPEER, "another part of the group has failed"); }
      if(++spins > 2000)
      {
        sched_yield();
        if((spins & 1023) == 0 && group_now() - t0 > group_timeout())
        {
          sh->abort_flag.store(1);
          return fail(BWTM_EPEER, "part %d waited %.0f s for part %d (BWTM_GROUP_TIMEOUT)", g->part, group_now() - t0, h);
        }
      }
      else { __builtin_ia32_pause(); }
    }
  }
  g->wait_seconds += group_now() - t0;
  return BWTM_OK;
}

int group_barrier(bwtm_group* g)
{
  if(g->parts == 1) { return BWTM_OK; }
  g->my_seq++;
  g->sh->seq[g->part].store(g->my_seq, std::memory_order_release);
  return group_wait(g, g->my_seq);
}

// all[h * nbytes ..] = part h's `mine`; any size (moved in pieces of GROUP_BLOB bytes, two banks: a part may already write the next piece
// while a slower one still reads this one, never the one after).
int group_allgather(bwtm_group* g, const void* mine, u64 nbytes, void* all)
{
  if(g->parts == 1) { if(all != mine && nbytes > 0) { std::memcpy(all, mine, nbytes); } return BWTM_OK; }
  for(u64 off = 0; off < nbytes || (nbytes == 0 && off == 0); off += GROUP_BLOB)
  {
    const u64 len = std::min<u64>(GROUP_BLOB, nbytes - off);
    const u32 bank = (u32)(g->gathers++ & 1);
    if(len > 0) { std::memcpy(g->sh->blob[bank][g->part], (const u8*)mine + off, len); }
    TRY(group_barrier(g));
    for(int h = 0; h < g->parts && len > 0; h++) { std::memcpy((u8*)all + (u64)h * nbytes + off, g->sh->blob[bank][h], len); }
    if(nbytes == 0) { break; }
  }
  return BWTM_OK;
}

void group_abort(bwtm_group* g) { if(g && g->sh) { g->sh->abort_flag.store(1); } }

// A compute section of a part: in serial mode it begins when all parts' sections before it (in the order (section, part)) have finished on
// the device, and its end drains the part's stream.  Every part must pass through the same sequence of sections.
struct Turn
{
  bwtm_group* g; int rc = BWTM_OK; bool held = false;
  explicit Turn(bwtm_group* g_) : g(g_)
  {
    if(!g->serial || g->parts == 1) { return; }
    const u64 mine = g->sections * (u64)g->parts + (u64)g->part;
    g->sections++;
    const double t0 = group_now();
    u32 spins = 0;
    while(g->sh->ticket.load(std::memory_order_acquire) != mine)
    {
      if(g->sh->abort_flag.load(std::memory_order_relaxed) != 0) { rc = fail(BWTM_EPEER, "another part of the group has failed"); return; }
      if(++spins > 2000) { sched_yield(); if((spins & 1023) == 0 && group_now() - t0 > group_timeout()) { g->sh->abort_flag.store(1); rc = fail(BWTM_EPEER, "part %d waited %.0f s for its turn", g->part, group_now() - t0); return; } }
      else { __builtin_ia32_pause(); }
    }
    held = true;
  }
  ~Turn() { if(held) { (void)hipStreamSynchronize(CTX.stream); g->sh->ticket.fetch_add(1, std::memory_order_release); } }
  Turn(const Turn&) = delete; Turn& operator=(const Turn&) = delete;
};
#define TURN(group) Turn turn_(group); if(turn_.rc != BWTM_OK) { return turn_.rc; }

constexpr u64 GROUP_PINNED_BYTES = 32768;

int group_pinned(bwtm_group* g, char** out)
{
  if(!g->pinned) { HIP_TRY(hipHostMalloc((void**)&g->pinned, GROUP_PINNED_BYTES, hipHostMallocDefault)); }
  *out = g->pinned;
  return BWTM_OK;
}

// This part's exported arena: at least `bytes`, zeroed where `zero_bytes` says (from its start); re-published when it had to grow.
int group_arena(bwtm_group* g, u64 bytes, void** out)
{
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  if(g->arena && g->arena_bytes >= bytes && g->arena_device == device) { *out = g->arena; return BWTM_OK; }
  if(g->arena)
  {
    // peers may still hold the old block mapped: it is only released here after every part has passed a barrier in the new merge's
    // set-up (bwtm_part_create calls this before its first all-gather; nothing of a finished merge reads a peer's arena any more)
    HIP_TRY(hipSetDevice(g->arena_device)); (void)hipFree(g->arena); HIP_TRY(hipSetDevice(device));
    g->arena = nullptr; g->arena_bytes = 0;
  }
  bytes = (bytes + (2ull << 20) - 1) / (2ull << 20) * (2ull << 20);
  hipError_t e = hipMalloc(&g->arena, bytes);
  if(e != hipSuccess) { (void)hipGetLastError(); pool_trim(t_ctx); e = hipMalloc(&g->arena, bytes); }
  if(e != hipSuccess) { (void)hipGetLastError(); g->arena = nullptr; return fail(BWTM_ENOMEM, "hipMalloc(%llu bytes) of the part's exported buffers failed: %s", (unsigned long long)bytes, hipGetErrorString(e)); }
  g->arena_bytes = bytes; g->arena_device = device;
  GroupArena& a = g->sh->arena[g->part];
  a.pid = (u64)getpid(); a.raw = (u64)(uintptr_t)g->arena; a.bytes = bytes; a.device = device; a.has_handle = 0;
  if(g->parts > 1)
  {
    hipError_t ie = hipIpcGetMemHandle(&a.handle, g->arena);
    if(ie == hipSuccess) { a.has_handle = 1; } else { (void)hipGetLastError(); }      // parts of this process do not need it; others will say so
  }
  a.generation.store(a.generation.load() + 1, std::memory_order_release);
  *out = g->arena;
  return BWTM_OK;
}

// Part h's arena as this part addresses it (after a barrier behind h's group_arena call).
int group_peer_arena(bwtm_group* g, int h, void** out)
{
  if(h == g->part) { *out = g->arena; return BWTM_OK; }
  GroupArena& a = g->sh->arena[h];
  const u64 gen = a.generation.load(std::memory_order_acquire);
  bwtm_group::Peer& p = g->peer[h];
  if(gen == 0) { return fail(BWTM_EPEER, "part %d has not exported its buffers", h); }
  if(p.generation != gen)
  {
    if(p.opened) { (void)hipIpcCloseMemHandle(p.ptr); p.opened = false; }
    p.ptr = nullptr;
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    if(a.pid == (u64)getpid())
    {
      // a part of this process: the same address space; another device needs peer access
      if(a.device != device)
      {
        int can = 0;
        HIP_TRY(hipDeviceCanAccessPeer(&can, device, a.device));
        if(!can) { return fail(BWTM_ENODEV, "GPU %d cannot access the memory of GPU %d: the merge over partitioned records needs peer access", device, a.device); }
        hipError_t e = hipDeviceEnablePeerAccess(a.device, 0);
        if(e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { return fail(BWTM_ENODEV, "hipDeviceEnablePeerAccess(%d) failed: %s", a.device, hipGetErrorString(e)); }
        (void)hipGetLastError();
      }
      p.ptr = (void*)(uintptr_t)a.raw;
    }
    else
    {
      if(!a.has_handle) { return fail(BWTM_ENODEV, "part %d (another process) could not export its buffers (hipIpcGetMemHandle)", h); }
      hipError_t e = hipIpcOpenMemHandle(&p.ptr, a.handle, hipIpcMemLazyEnablePeerAccess);
      if(e != hipSuccess) { (void)hipGetLastError(); p.ptr = nullptr; return fail(BWTM_ENODEV, "hipIpcOpenMemHandle of part %d's buffers failed: %s", h, hipGetErrorString(e)); }
      p.opened = true;
    }
    p.generation = gen;
  }
  *out = p.ptr;
  return BWTM_OK;
}

} // namespace

extern "C" int bwtm_group_create(const char* name, int part, int parts, bwtm_group** out)
{
  if(!out || parts < 1 || parts > (int)PART_MAX || part < 0 || part >= parts) { return fail(BWTM_EINVAL, "bwtm_group_create: bad argument (1 .. %u parts)", PART_MAX); }
  if(parts > 1 && (!name || name[0] != '/' || std::strlen(name) > 200)) { return fail(BWTM_EINVAL, "bwtm_group_create: a group of several parts needs a shared-memory name that begins with '/'"); }
  bwtm_group* g = new bwtm_group();
  g->part = part; g->parts = parts;
  if(parts == 1)
  {
    g->sh = new (std::calloc(1, sizeof(GroupShared))) GroupShared();
    g->sh->parts = 1; g->sh->magic.store(GROUP_MAGIC);
    *out = g;
    return BWTM_OK;
  }
  const double t0 = group_now();
  int fd = -1;
  if(part == 0)
  {
    (void)shm_unlink(name);                                        // a leftover of a process that died
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if(fd < 0 || ftruncate(fd, (off_t)sizeof(GroupShared)) != 0)
    {
      if(fd >= 0) { close(fd); (void)shm_unlink(name); }
      delete g;
      return fail(BWTM_ENOMEM, "bwtm_group_create: cannot create the shared memory %s", name);
    }
  }
  else
  {
    while(true)
    {
      fd = shm_open(name, O_RDWR, 0600);
      struct stat st;
      if(fd >= 0 && fstat(fd, &st) == 0 && (u64)st.st_size >= sizeof(GroupShared)) { break; }
      if(fd >= 0) { close(fd); fd = -1; }
      if(group_now() - t0 > group_timeout()) { delete g; return fail(BWTM_EPEER, "bwtm_group_create: part 0 did not create %s within %.0f s", name, group_timeout()); }
      usleep(1000);
    }
  }
  void* m = mmap(nullptr, sizeof(GroupShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if(m == MAP_FAILED) { if(part == 0) { (void)shm_unlink(name); } delete g; return fail(BWTM_ENOMEM, "bwtm_group_create: cannot map the shared memory %s", name); }
  g->sh = (GroupShared*)m; g->mapped = true;
  if(part == 0)
  {
    // a fresh shm object is zero-filled: counters, flags and generations start at 0
    g->sh->parts = (u32)parts;
    g->sh->magic.store(GROUP_MAGIC, std::memory_order_release);
  }
  else
  {
    while(g->sh->magic.load(std::memory_order_acquire) != GROUP_MAGIC)
    {
      if(group_now() - t0 > group_timeout()) { munmap(m, sizeof(GroupShared)); delete g; return fail(BWTM_EPEER, "bwtm_group_create: %s was never initialised", name); }
      usleep(200);
    }
    if(g->sh->parts != (u32)parts) { munmap(m, sizeof(GroupShared)); delete g; return fail(BWTM_EINVAL, "bwtm_group_create: %s was created for %u parts, not %d", name, g->sh->parts, parts); }
  }
  g->sh->attached.fetch_add(1);
  { const char* v = std::getenv("BWTM_GROUP_SERIAL"); g->serial = (v && v[0] == '1'); }
  int rc = group_barrier(g);                                         // everybody has the block mapped ...
  if(part == 0) { (void)shm_unlink(name); }                           // ... so its name can go: the memory lives as long as it is mapped
  if(rc != BWTM_OK) { munmap(m, sizeof(GroupShared)); delete g; return rc; }
  *out = g;
  return BWTM_OK;
}

extern "C" void bwtm_group_free(bwtm_group* g)
{
  if(!g) { return; }
  for(int h = 0; h < g->parts; h++) { if(g->peer[h].opened) { (void)hipIpcCloseMemHandle(g->peer[h].ptr); } }
  if(g->pinned) { (void)hipHostFree(g->pinned); }
  if(g->arena) { int dev = 0; (void)hipGetDevice(&dev); (void)hipSetDevice(g->arena_device); (void)hipFree(g->arena); (void)hipSetDevice(dev); }
  if(g->sh) { if(g->mapped) { munmap(g->sh, sizeof(GroupShared)); } else { std::free(g->sh); } }
  delete g;
}

extern "C" int bwtm_group_barrier(bwtm_group* g)
{
  if(!g) { return fail(BWTM_EINVAL, "bwtm_group_barrier: null argument"); }
  return group_barrier(g);
}

extern "C" int bwtm_group_allgather(bwtm_group* g, const void* mine, uint64_t nbytes, void* all)
{
  if(!g || (nbytes > 0 && (!mine || !all))) { return fail(BWTM_EINVAL, "bwtm_group_allgather: null argument"); }
  return group_allgather(g, mine, nbytes, all);
}

extern "C" void bwtm_group_abort(bwtm_group* g) { group_abort(g); }
extern "C" int bwtm_group_part(const bwtm_group* g) { return g ? g->part : -1; }
extern "C" int bwtm_group_parts(const bwtm_group* g) { return g ? g->parts : 0; }

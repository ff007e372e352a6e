// Microbenchmark (diagnostic, not product): what the MI355X memory system sustains for the
// access shapes of the search kernel.  Build: hipcc --offload-arch=gfx950 -O3 -o microbench_gather microbench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>
typedef uint64_t u64; typedef uint32_t u32;

__device__ inline u64 mix(u64 z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }

// MODE 0: lane reads 64 B (4 x 16 B) of a random 64-byte record.
// MODE 1: lane reads 128 B (8 x 16 B) of a random 128-byte line.
// MODE 2: lane reads only the first 16 B of a random 64-byte record.
// MODE 3: 4 lanes share one random 64-byte record (16 B each, one instruction per record).
// MODE 4: lane reads 64 B = lower half of a random 128-byte-aligned line.
template<int MODE>
__global__ void __launch_bounds__(256) k_gather(const uint4* table, u64 nunits, u64 iters, u64* sink)
{
  u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  u64 key = (MODE == 3 ? gid >> 2 : gid);
  u32 acc = 0;
  u64 state = mix(key * 0x9E3779B97F4A7C15ULL + 12345);   // independent stream per chain
  for(u64 it = 0; it < iters; it++)
  {
    u64 u = mix(state + acc) % nunits;      // dependent on the previous load (a chain)
    state = state * 6364136223846793005ULL + 1442695040888963407ULL;
    if(MODE == 0) { const uint4* p = table + 4 * u; uint4 a = p[0], b = p[1], c = p[2], d = p[3]; acc += a.x ^ b.y ^ c.z ^ d.w; }
    else if(MODE == 1) { const uint4* p = table + 8 * u; uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7]; acc += a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ f.y ^ g.z ^ h.w; }
    else if(MODE == 2) { const uint4* p = table + 4 * u; uint4 a = p[0]; acc += a.x; }
    else if(MODE == 3) { const uint4* p = table + 4 * u + (threadIdx.x & 3); uint4 a = p[0]; u32 v = a.x ^ a.y; v ^= __shfl_xor((int)v, 1); v ^= __shfl_xor((int)v, 2); acc += v; }
    else { const uint4* p = table + 8 * u; uint4 a = p[0], b = p[1], c = p[2], d = p[3]; acc += a.x ^ b.y ^ c.z ^ d.w; }
  }
  if(acc == 0x12345678) { sink[0] = acc; }
}

// Scattered emits: MODE 0 atomicOr on random 32-bit words, 1 plain 8-byte stores, 2 plain 4-byte stores.
template<int MODE>
__global__ void __launch_bounds__(256) k_scatter(u32* table, u64 nwords, u64 iters)
{
  u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  u64 state = mix(gid * 0x9E3779B97F4A7C15ULL + 777);
  for(u64 it = 0; it < iters; it++)
  {
    u64 z = mix(state); state = state * 6364136223846793005ULL + 1442695040888963407ULL;
    u64 w = z % nwords;
    if(MODE == 0) { atomicOr(table + w, 1u << (z >> 59)); }
    else if(MODE == 1) { ((u64*)table)[w >> 1] = z; }
    else { table[w] = (u32)z; }
  }
}

// Walk-like: per step two random 64-byte records (4 lanes per record) + one emit per chain.
// EMIT 0 none, 1 atomicOr (one lane of the quad), 2 plain 8-byte store (one lane of the quad).
template<int EMIT>
__global__ void __launch_bounds__(256) k_walklike(const uint4* ta, const uint4* tb, u64 nunits, u64 iters, u32* bits, u64 nwords, u64* sink)
{
  u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  u64 key = gid >> 2; u32 q = threadIdx.x & 3;
  u32 acc = 0;
  u64 state = mix(key * 0x9E3779B97F4A7C15ULL + 12345);   // independent stream per chain
  for(u64 it = 0; it < iters; it++)
  {
    u64 z = mix(state + acc); state = state * 6364136223846793005ULL + 1442695040888963407ULL;
    u64 ua = z % nunits, ub = (z >> 20) % nunits;
    uint4 a = ta[4 * ua + q], b = tb[4 * ub + q];
    u32 v = a.x ^ a.y ^ b.z ^ b.w; v ^= __shfl_xor((int)v, 1); v ^= __shfl_xor((int)v, 2); acc += v;
    if(EMIT != 0 && q == 0)
    {
      u64 w = mix(z + v) % nwords;
      if(EMIT == 1) { atomicOr(bits + w, 1u << (z >> 59)); } else { ((u64*)bits)[w >> 1] = z; }
    }
  }
  if(acc == 0x12345678) { sink[0] = acc; }
}

template<class F> float timeit(F f) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); f(); (void)hipDeviceSynchronize(); (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); return ms; }

int main()
{
  const u64 bytes = 5ull << 30;                 // 5 GiB table (two 2.5 GB indexes at config 2)
  uint4* table; (void)hipMalloc(&table, bytes); (void)hipMemset(table, 1, bytes);
  u64* sink; (void)hipMalloc(&sink, 64);
  const u64 iters = 2048;
  const char* names[] = {"64 B record, lane per record (4 x dwordx4)", "128 B line, lane per line (8 x dwordx4)", "16 B of a 64 B record, lane per record",
                         "64 B record, 4 lanes per record (1 x dwordx4)", "lower 64 B of a 128 B line, lane per line"};
  for(int blocks : {2048})
  {
    for(int mode = 0; mode < 5; mode++)
    {
      u64 threads = (u64)blocks * 256;
      u64 unit = (mode == 1 || mode == 4 ? 128 : 64);
      u64 nunits = bytes / unit;
      float ms = 0;
      switch(mode)
      {
        case 0: ms = timeit([&]() { hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(256), 0, 0, table, nunits, iters, sink); }); break;
        case 1: ms = timeit([&]() { hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, table, nunits, iters, sink); }); break;
        case 2: ms = timeit([&]() { hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(256), 0, 0, table, nunits, iters, sink); }); break;
        case 3: ms = timeit([&]() { hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(256), 0, 0, table, nunits, iters, sink); }); break;
        case 4: ms = timeit([&]() { hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(256), 0, 0, table, nunits, iters, sink); }); break;
      }
      double accesses = (double)(mode == 3 ? threads / 4 : threads) * iters;
      double useful = accesses * (mode == 1 ? 128 : (mode == 2 ? 16 : 64));
      printf("gather  blocks %5d  %-48s %8.2f ms  %7.2f G accesses/s  %7.1f GB/s useful\n", blocks, names[mode], ms, accesses / ms / 1e6, useful / ms / 1e6);
    }
  }
  const u64 sbytes = 1280ull << 20;             // 1.25 GiB bitvector (config 2)
  const char* snames[] = {"atomicOr u32 (random word)", "plain 8-byte store (random)", "plain 4-byte store (random)"};
  for(int mode = 0; mode < 3; mode++)
  {
    int blocks = 2048; u64 threads = (u64)blocks * 256; u64 it2 = 2048;
    float ms = 0;
    switch(mode)
    {
      case 0: ms = timeit([&]() { hipLaunchKernelGGL(k_scatter<0>, dim3(blocks), dim3(256), 0, 0, (u32*)table, sbytes / 4, it2); }); break;
      case 1: ms = timeit([&]() { hipLaunchKernelGGL(k_scatter<1>, dim3(blocks), dim3(256), 0, 0, (u32*)table, sbytes / 4, it2); }); break;
      case 2: ms = timeit([&]() { hipLaunchKernelGGL(k_scatter<2>, dim3(blocks), dim3(256), 0, 0, (u32*)table, sbytes / 4, it2); }); break;
    }
    double n = (double)threads * it2;
    printf("scatter %-48s %8.2f ms  %7.2f G ops/s\n", snames[mode], ms, n / ms / 1e6);
  }
  {
    u32* bits; (void)hipMalloc(&bits, sbytes); (void)hipMemset(bits, 0, sbytes);
    const uint4* ta = table; const uint4* tb = table + (bytes / 32);
    u64 nunits = bytes / 2 / 64;
    int blocks = 2048; u64 it3 = (getenv("MB_ITERS") ? strtoull(getenv("MB_ITERS"), 0, 10) : 1024); double n = (double)blocks * 256 / 4 * it3;
    const char* wn[] = {"walk-like quad gather x2, no emit", "walk-like quad gather x2 + atomicOr", "walk-like quad gather x2 + 8-byte store"};
    for(int mode = 0; mode < 3; mode++)
    {
      float ms = 0;
      if(mode == 0) ms = timeit([&]() { hipLaunchKernelGGL(k_walklike<0>, dim3(blocks), dim3(256), 0, 0, ta, tb, nunits, it3, bits, sbytes / 4, sink); });
      if(mode == 1) ms = timeit([&]() { hipLaunchKernelGGL(k_walklike<1>, dim3(blocks), dim3(256), 0, 0, ta, tb, nunits, it3, bits, sbytes / 4, sink); });
      if(mode == 2) ms = timeit([&]() { hipLaunchKernelGGL(k_walklike<2>, dim3(blocks), dim3(256), 0, 0, ta, tb, nunits, it3, bits, sbytes / 4, sink); });
      printf("%-58s %8.2f ms  %7.2f G steps/s\n", wn[mode], ms, n / ms / 1e6);
    }
  }
  return 0;
}

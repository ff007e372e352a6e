// Stress test of the virtual memory API as the library's pool uses it: physical chunks move between blocks, blocks are
// unmapped and mapped again (at a reused or at a fresh address), kernels and host copies see the right memory.
//   vmm_stress <reuse_va 0|1|2|3> <chunk bytes> <rounds>
//   reuse_va: 0 = fresh addresses every round, 1 = the same addresses every round, 2 = the address range is freed and
//   reserved again every round, 3 = like 1 with hipDeviceSynchronize between unmap and map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while(0)

__global__ void fill(unsigned long long* p, size_t n, unsigned long long tag) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if(i < n) { p[i] = tag + i; } }

int main(int argc, char** argv)
{
  const int reuse_va = (argc > 1 ? atoi(argv[1]) : 1);
  const size_t chunk = (argc > 2 ? strtoull(argv[2], nullptr, 10) : (2ull << 20));
  const int rounds = (argc > 3 ? atoi(argv[3]) : 200);
  CK(hipSetDevice(0));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t nchunks = 64;
  std::vector<hipMemGenericAllocationHandle_t> pool(nchunks);
  for(auto& h : pool) { CK(hipMemCreate(&h, chunk, &prop, 0)); }
  const size_t va_size = (reuse_va ? 4 * nchunks * chunk : (size_t)rounds * 8 * nchunks * chunk);
  void* va; CK(hipMemAddressReserve(&va, va_size, chunk, nullptr, 0));
  printf("reserved %zu bytes at %p\n", va_size, va);
  std::mt19937_64 rng(7);
  std::vector<unsigned long long> host(nchunks * chunk / 8);
  size_t bump = 0; int bad = 0;
  for(int r = 0; r < rounds; r++)
  {
    // two blocks A and B share the pool; their sizes and the assignment of chunks change every round
    std::shuffle(pool.begin(), pool.end(), rng);
    const size_t ka = 1 + rng() % (nchunks / 2), kb = 1 + rng() % (nchunks / 2);
    char* a = (char*)va + (reuse_va ? 0 : bump); bump += ka * chunk;
    char* b = (char*)va + (reuse_va ? 2 * nchunks * chunk : bump); bump += kb * chunk;
    for(size_t k = 0; k < ka; k++) { CK(hipMemMap(a + k * chunk, chunk, 0, pool[k], 0)); }
    CK(hipMemSetAccess(a, ka * chunk, &acc, 1));
    for(size_t k = 0; k < kb; k++) { CK(hipMemMap(b + k * chunk, chunk, 0, pool[ka + k], 0)); }
    CK(hipMemSetAccess(b, kb * chunk, &acc, 1));
    const size_t na = ka * chunk / 8, nb = kb * chunk / 8;
    const unsigned long long ta = (unsigned long long)r << 40, tb = ((unsigned long long)r << 40) | (1ull << 39);
    hipLaunchKernelGGL(fill, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s, (unsigned long long*)a, na, ta);
    hipLaunchKernelGGL(fill, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, (unsigned long long*)b, nb, tb);
    CK(hipMemsetAsync(b, 0, 64, s));                       // a small memset in between, like the pool's users do
    CK(hipMemcpyAsync(host.data(), a, na * 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    for(size_t i = 0; i < na; i++) { if(host[i] != ta + i) { if(bad < 5) { printf("round %d: A[%zu] = %llx, expected %llx\n", r, i, host[i], ta + i); } bad++; break; } }
    CK(hipMemcpyAsync(host.data(), b, nb * 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    for(size_t i = 8; i < nb; i++) { if(host[i] != tb + i) { if(bad < 5) { printf("round %d: B[%zu] = %llx, expected %llx\n", r, i, host[i], tb + i); } bad++; break; } }
    CK(hipMemUnmap(a, ka * chunk));
    CK(hipMemUnmap(b, kb * chunk));
    if(reuse_va == 3) { CK(hipDeviceSynchronize()); }
    if(reuse_va == 2)
    {
      void* old = va;
      CK(hipMemAddressFree(va, va_size));
      CK(hipMemAddressReserve(&va, va_size, chunk, nullptr, 0));
      if(r < 3) { printf("re-reserved: %p -> %p\n", old, va); }
    }
  }
  printf("reuse_va %d chunk %zu rounds %d: %d bad rounds\n", reuse_va, chunk, rounds, bad);
  return bad != 0;
}

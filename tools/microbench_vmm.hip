// Cost of getting / returning large device buffers: hipMalloc / hipFree against the virtual memory API
// (hipMemCreate once, then hipMemMap / hipMemUnmap of pooled physical chunks into a reserved address range).
// Decides how the library's pool recycles memory between buffers of different sizes.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while(0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void touch(unsigned long long* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if(i < n) { p[i] = i; } }

int main()
{
  CK(hipSetDevice(0));
  size_t free_b, total_b; CK(hipMemGetInfo(&free_b, &total_b));
  printf("device memory: %.1f GB free of %.1f GB\n", free_b / 1e9, total_b / 1e9);
  const size_t GB = 1ull << 30;
  for(size_t sz : {4 * GB, 32 * GB})
  {
    for(int rep = 0; rep < 2; rep++)
    {
      void* p; double t0 = now(); CK(hipMalloc(&p, sz)); double t1 = now();
      hipLaunchKernelGGL(touch, dim3((unsigned)(sz / 8 / 256)), dim3(256), 0, 0, (unsigned long long*)p, sz / 8); CK(hipDeviceSynchronize()); double t2 = now();
      CK(hipFree(p)); double t3 = now();
      printf("hipMalloc %3zu GiB: %.2f ms, touch %.2f ms, hipFree %.2f ms\n", sz / GB, t1 - t0, t2 - t1, t3 - t2);
    }
  }
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("VMM granularity: %zu bytes\n", gran);
  for(size_t chunk : {256ull << 20, 1ull << 30})
  {
    const size_t total = 32 * GB, n = total / chunk;
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    double t0 = now();
    for(size_t k = 0; k < n; k++) { CK(hipMemCreate(&h[k], chunk, &prop, 0)); }
    double t1 = now();
    void* va; CK(hipMemAddressReserve(&va, 2 * total, 0, nullptr, 0));
    double t2 = now();
    hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    for(int rep = 0; rep < 3; rep++)
    {
      char* base = (char*)va + (rep & 1) * total;
      double m0 = now();
      for(size_t k = 0; k < n; k++) { CK(hipMemMap(base + k * chunk, chunk, 0, h[k], 0)); }
      double m1 = now();
      CK(hipMemSetAccess(base, total, &acc, 1));
      double m2 = now();
      hipLaunchKernelGGL(touch, dim3((unsigned)(total / 8 / 256)), dim3(256), 0, 0, (unsigned long long*)base, total / 8); CK(hipDeviceSynchronize());
      double m3 = now();
      CK(hipMemUnmap(base, total));
      double m4 = now();
      printf("chunk %4zu MiB x %zu: map %.2f ms, set access %.2f ms, touch %.2f ms, unmap %.2f ms\n", chunk >> 20, n, m1 - m0, m2 - m1, m3 - m2, m4 - m3);
    }
    double t3 = now();
    for(size_t k = 0; k < n; k++) { CK(hipMemRelease(h[k])); }
    CK(hipMemAddressFree(va, 2 * total));
    double t4 = now();
    printf("chunk %4zu MiB: create all %.2f ms, reserve %.2f ms, release all %.2f ms\n", chunk >> 20, t1 - t0, t2 - t1, t4 - t3);
  }
  return 0;
}

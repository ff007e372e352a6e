// Does a HIP graph shrink the idle time between the kernels of the LF-step loop on this ROCm release?  (VERDICT r4 next #6.)
// The loop of bwtm_search is: [small scan kernel, ~10 us] -> [large streaming kernel, ~1 ms, ~200 k workgroups] x ~100, every kernel
// depending on the one before it.  The kernel trace of the product shows ~10.6 us of idle device behind every large kernel and ~5.8 us
// behind every small one (profiles/r05_config2_summary.md).  This program runs the same shape -- a copy kernel over a buffer of the size the step
// kernel moves, and a small kernel over 8 MB -- once as plain launches on one stream and once as ONE captured graph of all launches, and
// prints the time per pair of both forms and the sum of the kernels' own durations (from a pair of events around a single launch).
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_graph_gap tools/microbench_graph_gap.hip && tools/microbench_graph_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while(0)

__global__ void __launch_bounds__(256) k_large(const uint4* in, uint4* out, size_t n)
{
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if(i < n) { uint4 v = in[i]; v.x += 1; out[i] = v; }
}

__global__ void __launch_bounds__(256) k_small(const unsigned long long* in, unsigned long long* out, size_t n)
{
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long acc = 0;
  for(int k = 0; k < 8; k++) { size_t j = i * 8 + k; if(j < n) { acc += in[j]; } }
  if(i * 8 < n) { out[i] = acc; }
}

int main(int argc, char** argv)
{
  const int pairs = (argc > 1 ? std::atoi(argv[1]) : 100);
  const size_t large_bytes = (size_t)2600 << 20;             // read 2.6 GB + write 2.6 GB per launch: what a full step of config 2 moves
  const size_t nl = large_bytes / 16, ns = (size_t)1 << 20;  // small: 8 MB in
  uint4 *a, *b; unsigned long long *s, *t;
  CHECK(hipMalloc(&a, large_bytes)); CHECK(hipMalloc(&b, large_bytes)); CHECK(hipMalloc(&s, ns * 8)); CHECK(hipMalloc(&t, ns));
  CHECK(hipMemset(a, 1, large_bytes)); CHECK(hipMemset(s, 1, ns * 8));
  hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto launch_pair = [&](int k)
  {
    hipLaunchKernelGGL(k_small, dim3((unsigned)((ns / 8 + 255) / 256)), dim3(256), 0, st, s, t, ns);
    hipLaunchKernelGGL(k_large, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, (k & 1 ? b : a), (k & 1 ? a : b), nl);
  };
  float ms = 0;
  // the kernels alone
  for(int w = 0; w < 3; w++) { launch_pair(w); }
  CHECK(hipStreamSynchronize(st));
  float large_ms = 0, small_ms = 0;
  for(int r = 0; r < 5; r++)
  {
    CHECK(hipEventRecord(e0, st)); hipLaunchKernelGGL(k_large, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, a, b, nl); CHECK(hipEventRecord(e1, st));
    CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); large_ms += ms / 5;
    CHECK(hipEventRecord(e0, st)); hipLaunchKernelGGL(k_small, dim3((unsigned)((ns / 8 + 255) / 256)), dim3(256), 0, st, s, t, ns); CHECK(hipEventRecord(e1, st));
    CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); small_ms += ms / 5;
  }
  std::printf("kernels alone (events around one launch): large %.1f us, small %.1f us\n", large_ms * 1e3, small_ms * 1e3);
  // plain launches
  for(int round = 0; round < 3; round++)
  {
    CHECK(hipEventRecord(e0, st));
    for(int k = 0; k < pairs; k++) { launch_pair(k); }
    CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("stream launches: %d pairs in %.3f ms = %.1f us per pair (idle per pair: %.1f us)\n", pairs, ms, ms * 1e3 / pairs, ms * 1e3 / pairs - (large_ms + small_ms) * 1e3);
  }
  // one graph of all launches
  hipGraph_t graph; hipGraphExec_t exec;
  CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for(int k = 0; k < pairs; k++) { launch_pair(k); }
  CHECK(hipStreamEndCapture(st, &graph));
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  for(int round = 0; round < 3; round++)
  {
    CHECK(hipEventRecord(e0, st));
    CHECK(hipGraphLaunch(exec, st));
    CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("one graph:       %d pairs in %.3f ms = %.1f us per pair (idle per pair: %.1f us)\n", pairs, ms, ms * 1e3 / pairs, ms * 1e3 / pairs - (large_ms + small_ms) * 1e3);
  }
  // graphs of 8 pairs (what the product could build: the host learns the frontier's size with a delay of 8 steps)
  hipGraph_t g8; hipGraphExec_t e8;
  CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for(int k = 0; k < 8; k++) { launch_pair(k); }
  CHECK(hipStreamEndCapture(st, &g8));
  CHECK(hipGraphInstantiate(&e8, g8, nullptr, nullptr, 0));
  for(int round = 0; round < 3; round++)
  {
    CHECK(hipEventRecord(e0, st));
    for(int k = 0; k < pairs / 8; k++) { CHECK(hipGraphLaunch(e8, st)); }
    CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int done = pairs / 8 * 8;
    std::printf("graphs of 8:     %d pairs in %.3f ms = %.1f us per pair (idle per pair: %.1f us)\n", done, ms, ms * 1e3 / done, ms * 1e3 / done - (large_ms + small_ms) * 1e3);
  }
  return 0;
}

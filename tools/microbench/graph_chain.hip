// Microbenchmark: per-kernel cost of a dependent chain of tiny kernels -- plain stream launches, a captured hipGraph of the same
// chain, and a graph with two independent chains (fork/join through a second stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void tiny(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }

int main() {
  const int N = 200, n = 64 * 1024;
  float *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  auto now = [] { return std::chrono::steady_clock::now(); };
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    // (a) stream chain
    auto t0 = now();
    CK(hipEventRecord(e0, s1));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, s1, a, n);
    CK(hipEventRecord(e1, s1));
    auto t1 = now();
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream chain : %6.2f us/kernel GPU, %6.2f us/kernel host issue\n", ms * 1e3 / N, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
  }
  // (b) graph of one chain
  hipGraph_t g; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, s1, a, n);
  CK(hipStreamEndCapture(s1, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = now();
    CK(hipEventRecord(e0, s1)); CK(hipGraphLaunch(ex, s1)); CK(hipEventRecord(e1, s1));
    auto t1 = now();
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph chain  : %6.2f us/kernel GPU, %6.2f us/kernel host issue\n", ms * 1e3 / N, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
  }
  // (c) graph with two independent chains of N/2 + N/2... here N each (2N kernels total)
  hipGraph_t g2; hipGraphExec_t ex2;
  CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0));
  for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, s1, a, n); hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, s2, b, n); }
  CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));
  CK(hipStreamEndCapture(s1, &g2)); CK(hipGraphInstantiate(&ex2, g2, nullptr, nullptr, 0));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s1)); CK(hipGraphLaunch(ex2, s1)); CK(hipEventRecord(e1, s1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph 2 chains (2x%d kernels): %6.2f us per kernel PAIR GPU\n", N, ms * 1e3 / N);
  }
  // (d) two streams, no graph
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s1));
    CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0));
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, s1, a, n); hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, s2, b, n); }
    CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));
    CK(hipEventRecord(e1, s1));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("2 streams (2x%d kernels): %6.2f us per kernel PAIR GPU\n", N, ms * 1e3 / N);
  }
  return 0;
}

// Microbenchmark: does hipGraphLaunch block the host -- when the same executable graph is still running, or when the stream has
// earlier work pending?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void spin(float* p, int iters) { float v = p[threadIdx.x]; for (int i = 0; i < iters; ++i) v = v * 1.0001f + 1.f; p[threadIdx.x] = v; }
static double us(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
int main() {
  float* a; CK(hipMalloc(&a, 4096)); CK(hipMemset(a, 0, 4096));
  hipStream_t cap; CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  hipGraph_t g; hipGraphExec_t ex, ex2;
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed));
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, cap, a, 20000);      // ~10 us each -> ~2 ms graph
  CK(hipStreamEndCapture(cap, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0)); CK(hipGraphInstantiate(&ex2, g, nullptr, nullptr, 0));
  auto now = [] { return std::chrono::steady_clock::now(); };
  for (int rep = 0; rep < 3; ++rep) {
    for (hipStream_t st : {(hipStream_t)0, cap}) {
      CK(hipDeviceSynchronize());
      auto t0 = now(); CK(hipGraphLaunch(ex, st)); auto t1 = now(); CK(hipGraphLaunch(ex, st)); auto t2 = now(); CK(hipGraphLaunch(ex2, st)); auto t3 = now();
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, a, 10); auto t4 = now();
      CK(hipDeviceSynchronize()); auto t5 = now();
      printf("stream %s: launch #1 %8.1f us | same exec again %8.1f us | other exec %8.1f us | kernel after %6.1f us | drain %8.1f us\n",
             st ? "side" : "null", us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4), us(t4, t5));
    }
  }
  // a long plain kernel first, then the graph
  CK(hipDeviceSynchronize());
  auto t0 = now(); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, a, 4000000); auto t1 = now(); CK(hipGraphLaunch(ex, 0)); auto t2 = now();
  CK(hipDeviceSynchronize()); auto t3 = now();
  printf("long kernel launch %6.1f us, graph launch behind it %8.1f us, drain %8.1f us\n", us(t0, t1), us(t1, t2), us(t2, t3));
  return 0;
}

// Microbenchmark: does a chain of small dependent kernels overlap with one long atomic-bound kernel on a second stream?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void tiny(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
__global__ __launch_bounds__(256) void scatter(unsigned long long* buf, unsigned mask, int iters) {
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 1u;
  for (int i = 0; i < iters; ++i) { s = s * 1664525u + 1013904223u; atomicAdd(buf + ((s >> 8) & mask), 0x100000001ull); }
}
// a mid-size streaming kernel (like bn_apply on a 16 MB tensor)
__global__ void stream_k(const float4* a, float4* b, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { float4 v = a[i]; v.x += 1.f; b[i] = v; } }
int main() {
  const int n = 64 * 1024, N = 150;
  float* a; CK(hipMalloc(&a, n * 4)); CK(hipMemset(a, 0, n * 4));
  unsigned long long* buf; CK(hipMalloc(&buf, (size_t)8 << 22)); CK(hipMemset(buf, 0, (size_t)8 << 22));
  float4 *sa, *sb; const size_t sn = (size_t)1 << 20; CK(hipMalloc(&sa, sn * 16)); CK(hipMalloc(&sb, sn * 16)); CK(hipMemset(sa, 0, sn * 16));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  float ms;
  auto chain = [&](hipStream_t s) { for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny, dim3(n / 256), dim3(256), 0, s, a, n); if (i % 5 == 0) hipLaunchKernelGGL(stream_k, dim3(sn / 256), dim3(256), 0, s, sa, sb, sn); } };
  auto scat = [&](hipStream_t s) { hipLaunchKernelGGL(scatter, dim3(2048), dim3(256), 0, s, buf, (1u << 22) - 1, 28); };
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s1)); chain(s1); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); printf("chain alone      %7.1f us\n", ms * 1e3);
    CK(hipEventRecord(e0, s1)); scat(s1); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); printf("scatter alone    %7.1f us\n", ms * 1e3);
    CK(hipEventRecord(e0, s1)); scat(s1); chain(s1); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); printf("sequential       %7.1f us\n", ms * 1e3);
    CK(hipEventRecord(e0, s1)); CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0));
    scat(s2); chain(s1);
    CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0)); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); printf("two streams      %7.1f us\n", ms * 1e3);
  }
  return 0;
}

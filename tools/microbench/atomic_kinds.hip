// Microbenchmark: throughput of scattered global atomics by kind and access pattern (MI355X).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* buf, unsigned mask, int iters, unsigned seed) {
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + seed;
  const int lane = threadIdx.x & 63;
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    unsigned idx = (s >> 8) & mask & ~1u;                 // even float index (8-byte aligned)
    if (KIND == 0) atomicAdd(buf + idx, 1.0f);
    if (KIND == 1) { atomicAdd(buf + idx, 1.0f); atomicAdd(buf + idx + 1, 1.0f); }
    if (KIND == 2) atomicAdd(reinterpret_cast<unsigned long long*>(buf + idx), 0x0000000100000001ull);
    if (KIND == 3) atomicAdd(reinterpret_cast<unsigned*>(buf + idx), 1u);
    if (KIND == 4) {   // 16 lanes share one 128-B line (different words)
      unsigned base = __shfl(idx, lane & 48) & ~31u;
      atomicAdd(buf + base + (lane & 15) * 2, 1.0f);
    }
    if (KIND == 5) {
      h2 v = {(_Float16)1.f, (_Float16)1.f};
      __builtin_amdgcn_global_atomic_fadd_v2f16((h2 __attribute__((address_space(1)))*)(buf + idx), v);
    }
    if (KIND == 6) atomicAdd(reinterpret_cast<double*>(buf + idx), 1.0);
    if (KIND == 7) {   // all 64 lanes the same line, different words (half the line each 32 lanes)
      unsigned base = __shfl(idx, 0) & ~31u;
      atomicAdd(buf + base + (lane & 31), 1.0f);
    }
    if (KIND == 8) {   // returning atomic
      float r = atomicAdd(buf + idx, 1.0f);
      if (r == -1.f) buf[0] = r;
    }
    if (KIND == 9) {   // plain scattered 8-byte stores for comparison
      reinterpret_cast<float2*>(buf + idx)[0] = make_float2(1.f, 2.f);
    }
  }
}

int main() {
  const size_t n = (size_t)1 << 24;   // 64 MB
  float* buf; CK(hipMalloc(&buf, n * 4 + 256)); CK(hipMemset(buf, 0, n * 4 + 256));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 2048, iters = 128;
  const double total = (double)blocks * 256 * iters;
  const char* names[10] = {"f32 random", "f32 pair (2 atomics, adjacent)", "u64 random", "u32 random", "f32 16 lanes/line", "pk_add_f16 random",
                           "f64 random", "f32 64 lanes/line", "f32 random returning", "8-byte plain stores"};
  for (int lg = 20; lg <= 24; lg += 4)
    for (int v = 0; v < 10; ++v) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        const unsigned mask = (1u << lg) - 1;
        CK(hipEventRecord(a));
        switch (v) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u); break;
          case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u); break;
          case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u); break;
          case 8: hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u); break;
          case 9: hipLaunchKernelGGL(k<9>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u); break;
        }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
      }
      printf("set %6.1f MB  %-34s %8.2f G lane-ops/s  (%.3f ms)\n", (double)((1u << lg)) * 4 / 1e6, names[v], total / ms / 1e6, ms);
    }
  return 0;
}

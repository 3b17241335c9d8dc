// Microbenchmark: random fp32 atomic adds at agent scope (what atomicAdd emits) vs workgroup scope (executed in the XCD's L2),
// over working sets of different size; plus the XCC_ID of each workgroup to confirm the round-robin dispatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}

__global__ void xcc_kernel(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

template <int SCOPE, bool PARTITION>
__global__ __launch_bounds__(256) void atomics_kernel(float* buf, unsigned mask, int iters, unsigned seed) {
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + seed;
  const unsigned xcd = PARTITION ? xcc_id() : 0;
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    unsigned idx = (s >> 8) & mask;
    if (PARTITION) idx = (idx & ~7u) | xcd;        // entries owned by this XCD: idx % 8 == xcd  (8-float = 32 B granularity!)
    if (SCOPE == 0) atomicAdd(buf + idx, 1.0f);
    else __hip_atomic_fetch_add(buf + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

// partition by 128-byte line: line index % 8 == xcd
template <int SCOPE>
__global__ __launch_bounds__(256) void atomics_line_kernel(float* buf, unsigned mask, int iters, unsigned seed) {
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + seed;
  const unsigned xcd = xcc_id();
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    unsigned idx = (s >> 8) & mask;
    idx = (idx & ~(7u << 5)) | (xcd << 5);         // 32 floats per line
    if (SCOPE == 0) atomicAdd(buf + idx, 1.0f);
    else __hip_atomic_fetch_add(buf + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

int main() {
  unsigned* d_x; CK(hipMalloc(&d_x, 64 * 4));
  hipLaunchKernelGGL(xcc_kernel, dim3(64), dim3(64), 0, 0, d_x);
  unsigned hx[64]; CK(hipMemcpy(hx, d_x, 64 * 4, hipMemcpyDeviceToHost));
  printf("xcc of blocks 0..31:"); for (int i = 0; i < 32; ++i) printf(" %u", hx[i]); printf("\n");
  const size_t maxn = (size_t)1 << 26;   // 256 MB of floats
  float* buf; CK(hipMalloc(&buf, maxn * 4)); CK(hipMemset(buf, 0, maxn * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 2048, iters = 256;
  const double total = (double)blocks * 256 * iters;
  for (int lg = 16; lg <= 26; lg += 2) {
    const unsigned mask = (1u << lg) - 1;
    float ms[5];
    for (int v = 0; v < 5; ++v) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        if (v == 0) hipLaunchKernelGGL((atomics_kernel<0, false>), dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u);
        if (v == 1) hipLaunchKernelGGL((atomics_kernel<1, false>), dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u);
        if (v == 2) hipLaunchKernelGGL((atomics_kernel<1, true>), dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u);
        if (v == 3) hipLaunchKernelGGL((atomics_line_kernel<1>), dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u);
        if (v == 4) hipLaunchKernelGGL((atomics_line_kernel<0>), dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 1u);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms[v], a, b));
      }
    }
    printf("set %8.2f MB: agent %7.2f G/s | wg-scope(unsafe) %7.2f | wg-scope part-by-elem %7.2f | wg part-by-line %7.2f | agent part-by-line %7.2f\n",
           (double)(mask + 1) * 4 / 1e6, total / ms[0] / 1e6, total / ms[1] / 1e6, total / ms[2] / 1e6, total / ms[3] / 1e6, total / ms[4] / 1e6);
  }
  // correctness of the line-partitioned workgroup-scope variant: total must equal the number of adds
  CK(hipMemset(buf, 0, maxn * 4));
  const unsigned mask = (1u << 20) - 1;
  hipLaunchKernelGGL((atomics_line_kernel<1>), dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 7u);
  CK(hipDeviceSynchronize());
  std::vector<float> h((size_t)mask + 1);
  CK(hipMemcpy(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost));
  double sum = 0; for (float x : h) sum += x;
  printf("line-partitioned wg-scope sum %.0f expected %.0f\n", sum, total);
  CK(hipMemset(buf, 0, maxn * 4));
  hipLaunchKernelGGL((atomics_kernel<1, false>), dim3(blocks), dim3(256), 0, 0, buf, mask, iters, 7u);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost));
  sum = 0; for (float x : h) sum += x;
  printf("unpartitioned wg-scope sum %.0f expected %.0f (lost updates show the L2s are not coherent)\n", sum, total);
  return 0;
}

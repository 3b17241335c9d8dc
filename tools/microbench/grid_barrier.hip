// Microbenchmark: what does a producer -> grid-wide statistic -> consumer dependency cost on MI355X as (a) two kernels of a replayed
// hipGraph (the BatchNorm chain of the ResNet3D today: GEMM with statistics atomics, then an apply kernel) and (b) ONE kernel with an
// in-kernel grid barrier between the two phases (every workgroup resident: grid <= CUs x workgroups per CU)?
//   phase 1: every workgroup reads its 64 x 64 fp16 tile (8 KB), adds per-column partial sums to 64 fp32 accumulators with atomics
//   phase 2: every workgroup reads the 64 accumulators (system-scope loads) and rewrites its tile scaled by them
// The chain is repeated N times inside one graph; reported is the time per (phase 1 + phase 2) pair.  The spin of the barrier has an
// iteration cap (a workgroup that gives up sets a flag and the result is reported as invalid) so that a scheduling surprise cannot hang
// the GPU.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ float colsum_phase(const __half* tile, float* stats) {
  // 256 threads, tile 64 rows x 64 cols: thread t -> column t & 63, rows (t >> 6) * 16 ...
  const int c = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * 16;
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += __half2float(tile[(r0 + r) * 64 + c]);
  __shared__ float part[4][64];
  part[threadIdx.x >> 6][c] = s;
  __syncthreads();
  if (threadIdx.x < 64) atomicAdd(stats + c, part[0][c] + part[1][c] + part[2][c] + part[3][c]);
  return s;
}
__device__ __forceinline__ void apply_phase(__half* tile, const float* stats) {
  const int c = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * 16;
  const float k = 1.f / (1.f + fabsf(__hip_atomic_load(stats + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)));
  for (int r = 0; r < 16; ++r) tile[(r0 + r) * 64 + c] = __float2half(__half2float(tile[(r0 + r) * 64 + c]) * k + 0.001f);
}

__global__ __launch_bounds__(256) void k_phase1(__half* data, float* stats) { colsum_phase(data + (size_t)blockIdx.x * 4096, stats); }
__global__ __launch_bounds__(256) void k_phase2(__half* data, const float* stats) { apply_phase(data + (size_t)blockIdx.x * 4096, stats); }

// one kernel, grid barrier in between.  `counter` counts arrivals monotonically over the whole graph: launch number `epoch` (1-based)
// waits for epoch * gridDim.x arrivals.
__global__ __launch_bounds__(256) void k_fused(__half* data, float* stats, unsigned* counter, unsigned epoch, unsigned* gave_up) {
  __half* tile = data + (size_t)blockIdx.x * 4096;
  colsum_phase(tile, stats);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                                                            // the statistics atomics are ordered before the arrival
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = epoch * gridDim.x;
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 200000u) { *gave_up = 1u; break; }
    }
  }
  __syncthreads();
  apply_phase(tile, stats);
}

int main() {
  const int N = 100;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int grids[5] = {32, 64, 128, 256, 512};
  __half* data; CK(hipMalloc(&data, (size_t)512 * 4096 * 2)); CK(hipMemset(data, 0, (size_t)512 * 4096 * 2));
  float* stats; CK(hipMalloc(&stats, 64 * 4 * (N + 1)));
  unsigned* counter; CK(hipMalloc(&counter, 8));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%d CUs; chain of %d (statistics, apply) pairs in one hipGraph\n", prop.multiProcessorCount, N);
  for (int gi = 0; gi < 5; ++gi) {
    const int G = grids[gi];
    hipGraph_t g1, g2; hipGraphExec_t x1, x2;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) {
      hipLaunchKernelGGL(k_phase1, dim3(G), dim3(256), 0, s, data, stats + 64 * i);
      hipLaunchKernelGGL(k_phase2, dim3(G), dim3(256), 0, s, data, stats + 64 * i);
    }
    CK(hipStreamEndCapture(s, &g1)); CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i)
      hipLaunchKernelGGL(k_fused, dim3(G), dim3(256), 0, s, data, stats + 64 * i, counter, (unsigned)(i + 1), counter + 1);
    CK(hipStreamEndCapture(s, &g2)); CK(hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0));
    float best1 = 1e9f, best2 = 1e9f;
    unsigned gave_up = 0;
    for (int rep = 0; rep < 5; ++rep) {
      float ms;
      CK(hipMemsetAsync(stats, 0, 64 * 4 * (N + 1), s));
      CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(x1, s)); CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best1) best1 = ms;
      CK(hipMemsetAsync(stats, 0, 64 * 4 * (N + 1), s)); CK(hipMemsetAsync(counter, 0, 8, s));
      CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(x2, s)); CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best2) best2 = ms;
      unsigned h[2]; CK(hipMemcpy(h, counter, 8, hipMemcpyDeviceToHost));
      gave_up |= h[1];
    }
    printf("grid %3d workgroups: two kernels %6.2f us per pair | one kernel + grid barrier %6.2f us per pair%s\n", G, best1 * 1e3 / N, best2 * 1e3 / N,
           gave_up ? "   (INVALID: a workgroup gave up spinning)" : "");
    CK(hipGraphExecDestroy(x1)); CK(hipGraphExecDestroy(x2)); CK(hipGraphDestroy(g1)); CK(hipGraphDestroy(g2));
  }
  return 0;
}

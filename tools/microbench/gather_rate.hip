// Microbenchmark: how many lane-gathers per cycle does a CU's texture addresser sustain, by load width and by how the 64 addresses of
// an instruction fall into cache lines (MI355X)?  Each lane issues `iters` x 8 independent buffer loads at pseudo-random entries of a
// table (20 KB: L1-resident; 2 MB: L2-resident; 64 MB: Infinity Cache / HBM), as the hash-grid kernels do.
//   kind 0: 4-byte loads, every lane its own random entry
//   kind 1: 8-byte loads (dwordx2), random 8-byte-aligned entry pairs
//   kind 2: 4-byte loads, unaligned pairs: lane's entry and entry + 1 as TWO 4-byte loads (what a trilinear x-pair costs today)
//   kind 3: 8-byte loads at 4-byte-aligned (not 8-byte-aligned) entries: the x-pair as ONE load on a dense level
//   kind 4: 4-byte loads, 16 neighbouring lanes share a 64-byte line (coherent rays)
//   kind 5: 16-byte loads (dwordx4), random 16-byte-aligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k(const unsigned* table, unsigned mask, int iters, unsigned* out) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(table), 0, 0xFFFFFFFFu, 0x00020000);
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  const unsigned lane = threadIdx.x & 63;
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
    unsigned idx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s = s * 1664525u + 1013904223u;
      idx[j] = (s >> 8) & mask;
      if (KIND == 4) idx[j] = (__shfl(idx[j], lane & 48) & ~15u) + (lane & 15);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (KIND == 0 || KIND == 4) acc += __builtin_amdgcn_raw_buffer_load_b32(rs, idx[j] << 2, 0, 0);
      if (KIND == 1) { u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (idx[j] & ~1u) << 2, 0, 0); acc += v[0] + v[1]; }
      if (KIND == 2) { acc += __builtin_amdgcn_raw_buffer_load_b32(rs, idx[j] << 2, 0, 0); acc += __builtin_amdgcn_raw_buffer_load_b32(rs, (idx[j] << 2) + 4, 0, 0); }
      if (KIND == 3) { u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (idx[j] | 1u) << 2, 0, 0); acc += v[0] + v[1]; }
      if (KIND == 5) { u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (idx[j] & ~3u) << 2, 0, 0); acc += v[0] + v[1] + v[2] + v[3]; }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t n = (size_t)1 << 24;   // 64 MB of entries
  unsigned* buf; CK(hipMalloc(&buf, n * 4 + 256)); CK(hipMemset(buf, 1, n * 4 + 256));
  unsigned* out; CK(hipMalloc(&out, 256));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double mhz = prop.clockRate / 1e3;
  const int blocks = cus * 8, iters = 64;
  const char* names[6] = {"b32 random", "b64 aligned pair", "2 x b32 adjacent (today's x-pair)", "b64 at odd entry (x-pair as one load)", "b32 16 lanes/line",
                          "b128 aligned"};
  const int per_iter[6] = {8, 8, 16, 8, 8, 8};           // gather instructions per loop iteration
  printf("%d CUs at %.0f MHz; %d blocks x 256 threads x %d iterations\n", cus, mhz, blocks, iters);
  const int lgs[3] = {12, 19, 24};                       // 16 KB, 2 MB, 64 MB tables
  for (int li = 0; li < 3; ++li) {
    const unsigned mask = (1u << lgs[li]) - 1;
    for (int v = 0; v < 6; ++v) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        switch (v) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, out); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, out); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, out); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, out); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, out); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, out); break;
        }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
      }
      const double insts_per_cu = (double)blocks * 4 * iters * per_iter[v] / cus;      // wave-level gather instructions per CU
      const double cycles = ms * 1e-3 * mhz * 1e6;
      printf("table %6.0f KB  %-40s %8.1f us   %6.1f cycles per gather instruction per CU   %7.1f G lane-loads/s\n", (mask + 1) * 4.0 / 1024,
             names[v], ms * 1e3, cycles / insts_per_cu, (double)blocks * 256 * iters * per_iter[v] / (ms * 1e-3) / 1e9);
    }
  }
  return 0;
}

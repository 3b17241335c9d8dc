// Lane layout of v_mfma_f32_4x4x4_16b_f16 (16 independent 4x4x4 blocks per wave), found by probing with one-hot operands:
// for A one-hot at (lane la, element ka) and B one-hot at (lane lb, element kb), which (lane, register) of D becomes 1?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(int la, int ka, int lb, int kb, float* out) {
  half4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  if ((int)threadIdx.x == la) a[ka] = (_Float16)1.f;
  if ((int)threadIdx.x == lb) b[kb] = (_Float16)1.f;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[threadIdx.x * 4 + r] = c[r];
}
int main() {
  float* d; CK(hipMalloc(&d, 256 * 4));
  float h[256];
  const int cases[][4] = {{0,0,0,0},{1,0,0,0},{0,0,1,0},{2,1,3,1},{2,1,3,2},{5,2,6,2},{5,2,2,2},{63,3,60,3},{4,0,4,0},{17,1,18,1}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], c[2], c[3], d);
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("A one-hot (lane %2d, k %d), B one-hot (lane %2d, k %d) ->", c[0], c[1], c[2], c[3]);
    int n = 0;
    for (int i = 0; i < 256; ++i) if (h[i] != 0.f) { printf(" D[lane %d][reg %d] = %g", i / 4, i % 4, h[i]); ++n; }
    if (!n) printf(" nothing");
    printf("\n");
  }
  return 0;
}

// Check of ds_read_b64_tr_b16 (via __builtin_amdgcn_ds_read_tr16_b64_v4i16) against the mechanism the kernels assume:
// per 16-lane group, lane 4q+p supplies &tile[r0+q][c0+4p]; lane i receives tile[r0+e][c0+i] in element e.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short tile[64][64];
  for (int e = threadIdx.x; e < 64 * 64; e += 64) tile[e / 64][e % 64] = (short)e;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  for (int blk = 0; blk < 8; ++blk) {            // r0 = 8g + 4*(blk&1), c0 = 16*(blk>>1)
    const int r0 = 8 * g + 4 * (blk & 1), c0 = 16 * (blk >> 1);
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(&tile[r0 + q][c0 + 4 * p]));
    for (int e = 0; e < 4; ++e) out[(blk * 64 + l) * 4 + e] = v[e];
  }
}
int main() {
  short* d; hipMalloc(&d, 8 * 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[8 * 64 * 4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int blk = 0; blk < 8; ++blk) for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
    const int g = l >> 4, i = l & 15, r0 = 8 * g + 4 * (blk & 1), c0 = 16 * (blk >> 1);
    const short want = (short)((r0 + e) * 64 + c0 + i);
    if (h[(blk * 64 + l) * 4 + e] != want) { if (bad < 8) printf("blk %d lane %d e %d got %d want %d\n", blk, l, e, h[(blk * 64 + l) * 4 + e], want); ++bad; }
  }
  printf("tr_read mismatches: %d\n", bad);
  return bad != 0;
}

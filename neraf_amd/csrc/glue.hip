// Scalar plumbing of one training iteration as single launches.  The reference builds these out of scalar torch ops (loss factors,
// `functools.reduce(add, loss_dict.values())`, `grad_scaler.scale(loss)`, psnr, the camera regulariser: NerfactoModel.get_loss_dict /
// get_metrics_dict and Trainer.train_iteration [NS-recall], NeRAF_model.py:592-599); every such op is a dependent launch of >= 2 us on
// the stream, and a step had ~65 of them (tools/torch_glue.py).
#include "common.h"

namespace {

struct TermPtrs { const float* p[12]; int n; };

__global__ void loss_sum_scale_kernel(TermPtrs t, const float* __restrict__ scale, float* __restrict__ out2) {
  if (threadIdx.x != 0) return;
  float s = 0.f;                                   // left-to-right, as reduce(add, ...) does
  for (int i = 0; i < t.n; ++i) s += *t.p[i];
  out2[1] = s;
  out2[0] = scale ? s * (*scale) : s;
}

// out4 = {sums[0] * k[0], sums[1] * k[1], sums[2] * k[2], -10 log10(out4[0])}: the three vision losses from the kernels' raw sums
// (rgb squared error, distortion, interlevel) and the batch psnr
__global__ void vision_loss_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ k3, float* __restrict__ out4) {
  if (threadIdx.x != 0) return;
  const float mse = sums[0] * k3[0];
  out4[0] = mse; out4[1] = sums[1] * k3[1]; out4[2] = sums[2] * k3[2];
  out4[3] = -10.f * log10f(mse);
}

// d_rgb = u_rgb * up[0];  d_dens = u_dens * up[0] + u_dist * up[2];  up = (g_rgb, g_inter, g_dist) gathered from three device scalars
// (null = 0) and published for the interlevel backward; d_rays and sums4 zero-filled -- all the preparation of _VisionLossFn.backward
__global__ __launch_bounds__(256) void vision_bwd_prologue_kernel(const float* __restrict__ u_rgb, const float* __restrict__ u_dens,
                                                                  const float* __restrict__ u_dist, const float* g_rgb, const float* g_inter,
                                                                  const float* g_dist, size_t n, float* __restrict__ d_rgb,
                                                                  float* __restrict__ d_dens, float* __restrict__ up3,
                                                                  float* __restrict__ d_rays, size_t n_rays, float* __restrict__ sums4) {
  const float a = g_rgb ? *g_rgb : 0.f, b = g_inter ? *g_inter : 0.f, c = g_dist ? *g_dist : 0.f;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  if (tid == 0) { up3[0] = a; up3[1] = b; up3[2] = c; }
  if (tid < 4 && sums4) sums4[tid] = 0.f;
  for (size_t i = tid; i < n; i += nth) {
    d_dens[i] = u_dens[i] * a + u_dist[i] * c;
    const size_t j = 3 * i;
    d_rgb[j] = u_rgb[j] * a; d_rgb[j + 1] = u_rgb[j + 1] * a; d_rgb[j + 2] = u_rgb[j + 2] * a;
  }
  if (d_rays)
    for (size_t i = tid; i < n_rays; i += nth) d_rays[i] = 0.f;
}

}  // namespace

extern "C" int neraf_loss_sum_scale(neraf_ctx* ctx, const float* const* terms, int n, const float* scale, float* out2, neraf_stream_t stream) {
  if (!terms || n <= 0 || n > 12 || !out2) return neraf_fail(ctx, NERAF_EINVAL, "loss_sum_scale: 1..12 device scalars and an output pair");
  TermPtrs t{};
  t.n = n;
  for (int i = 0; i < n; ++i) { if (!terms[i]) return neraf_fail(ctx, NERAF_EINVAL, "loss_sum_scale: null term"); t.p[i] = terms[i]; }
  hipLaunchKernelGGL(loss_sum_scale_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, scale, out2);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_vision_loss_finalize(neraf_ctx* ctx, const float* sums, const float* k3, float* out4, neraf_stream_t stream) {
  if (!sums || !k3 || !out4) return neraf_fail(ctx, NERAF_EINVAL, "vision_loss_finalize: bad arguments");
  hipLaunchKernelGGL(vision_loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, k3, out4);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_vision_bwd_prologue(neraf_ctx* ctx, const float* u_rgb, const float* u_dens, const float* u_dist, const float* g_rgb,
                                         const float* g_inter, const float* g_dist, size_t n_samples, float* d_rgb, float* d_dens, float* up3,
                                         float* d_rays, size_t n_ray_floats, float* sums4, neraf_stream_t stream) {
  if (!u_rgb || !u_dens || !u_dist || !d_rgb || !d_dens || !up3 || n_samples == 0)
    return neraf_fail(ctx, NERAF_EINVAL, "vision_bwd_prologue: bad arguments");
  size_t blocks = (n_samples + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(vision_bwd_prologue_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, u_rgb, u_dens, u_dist, g_rgb, g_inter,
                     g_dist, n_samples, d_rgb, d_dens, up3, d_rays, n_ray_floats, sums4);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

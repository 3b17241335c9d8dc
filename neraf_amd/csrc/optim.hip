// Fused multi-tensor Adam for the joint training step (the reference trains its fields with torch Adam through nerfstudio's
// Optimizers, NeRAF_config.py:116-127, under a GradScaler): ONE launch updates every parameter tensor of an optimizer.
// The per-tensor table (parameter, gradient, moment pointers, element count, learning rate) and the workgroup -> (tensor,
// offset) map live in device memory owned by the host layer; the kernel un-scales the gradient by 1 / *grad_scale on the fly
// and leaves everything untouched when *found_inf != 0 (GradScaler semantics), so no separate un-scale pass writes gradients.
#include "common.h"

namespace {

struct AdamTensor { float* p; const float* g; float* m; float* v; long long numel; int group; int slot; };
struct AdamLrs { float lr[8]; };
// Optional second record per tensor: the same parameter is ALSO a member of an earlier optimizer whose update was deferred to this
// launch (the reference steps the radiance field with "fields" and then with "audio_fields", NeRAF_pipeline.py:487): moments and
// counter slot and parameter group of that earlier optimizer (slot indexes ITS step table, group ITS learning rates).  m == nullptr: no deferred update.
struct AdamDual { float* m; float* v; int group; int slot; };
static_assert(sizeof(AdamDual) == 24, "table layout is shared with neraf_amd/optim.py");
static_assert(sizeof(AdamTensor) == 48, "table layout is shared with neraf_amd/optim.py");

constexpr int kAdamChunk = 4096;       // elements per workgroup

// One counter record {t, 1 / (1 - b1^t), 1 / sqrt(1 - b2^t), -} per parameter TENSOR (slot = AdamTensor::slot), as torch.optim.Adam
// keeps a `step` per parameter and skips parameters without a gradient: tensors of one group need not receive their first gradient
// together (the reference's "audio_fields" group holds the radiance field, trained from step 0, and the NAcF / ResNet3D, whose
// gradients are None until start_step_audio -- NeRAF_pipeline.py:186, :487 -- and must start their bias correction at t = 1 then).
// Thread i advances the slot of the i-th tensor of the launch (unless a gradient was non-finite); the corrections in double as torch
// computes them on the host (1 - 0.999^t cancels catastrophically in fp32).
__global__ void adam_advance_step_kernel(float* __restrict__ step, const float* __restrict__ found_inf, double beta1, double beta2,
                                         const AdamTensor* __restrict__ table, int n_tensors) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tensors) return;
  float* s = step + 4 * table[i].slot;
  if (!(found_inf && found_inf[0] != 0.f)) s[0] += 1.f;
  const double t = (double)s[0];
  s[1] = (float)(1.0 / (1.0 - pow(beta1, t)));
  s[2] = (float)(1.0 / sqrt(1.0 - pow(beta2, t)));
}

// One Adam update of one element, every operation written out (explicit fused multiply-adds, no contraction left to the compiler) so
// that the plain path and the fused double update execute the SAME fp32 operation sequence and agree bit for bit.
__device__ __forceinline__ void adam_elem(float& p, float gr, float& m, float& v, float beta1, float beta2, float omb1, float omb2,
                                          float step_size, float inv_sqrt_bc2, float eps) {
  m = __fmaf_rn(beta1, m, __fmul_rn(omb1, gr));
  v = __fmaf_rn(beta2, v, __fmul_rn(__fmul_rn(omb2, gr), gr));
  const float denom = __fmaf_rn(__fsqrt_rn(v), inv_sqrt_bc2, eps);
  p = __fsub_rn(p, __fdiv_rn(__fmul_rn(step_size, m), denom));
}

// Up to kArgPtrs gradient pointers travel in the KERNEL ARGUMENTS (2 KiB of the 4 KiB argument segment): the gradient tensors of a
// step are new allocations, and a pointer column in device memory needs a host-to-device copy per optimizer and step in front of the
// launch.  A workgroup's tensor index is uniform, so the lookup is one scalar load from the argument segment.
constexpr int kArgPtrs = 256;
struct GradPtrArgs { unsigned long long p[kArgPtrs]; int n; };

template <bool ARGP>
__global__ __launch_bounds__(256) void fused_adam_kernel(const AdamTensor* __restrict__ table, const unsigned long long* __restrict__ g_ptrs,
                                                        GradPtrArgs gargs,
                                                        const int* __restrict__ blk_tensor,
                                                        const int* __restrict__ blk_chunk, AdamLrs lrs, float beta1, float beta2,
                                                        float omb1, float omb2, float eps,
                                                        const float* __restrict__ step, const float* __restrict__ grad_scale,
                                                        const float* __restrict__ found_inf, const AdamDual* __restrict__ dual,
                                                        const float* __restrict__ step0, const float* __restrict__ found_inf0, AdamLrs lrs0) {
  const bool skip = found_inf && found_inf[0] != 0.f;
  AdamTensor t = table[blk_tensor[blockIdx.x]];
  AdamDual d{};
  if (dual) d = dual[blk_tensor[blockIdx.x]];
  const bool first = d.m != nullptr && !(found_inf0 && found_inf0[0] != 0.f);     // the deferred update of the earlier optimizer
  if (skip && !first) return;
  if (ARGP) t.g = reinterpret_cast<const float*>(gargs.p[blk_tensor[blockIdx.x]]);
  else if (g_ptrs) t.g = reinterpret_cast<const float*>(g_ptrs[blk_tensor[blockIdx.x]]);
  const long long base = (long long)blk_chunk[blockIdx.x] * kAdamChunk;
  const float inv_scale = grad_scale ? 1.f / grad_scale[0] : 1.f;
  const float step_size = lrs.lr[t.group & 7] * step[4 * t.slot + 1], inv_sqrt_bc2 = step[4 * t.slot + 2];
  if (first) {
    // both updates of a doubly-stepped tensor in one pass over p and g: update 0 (the earlier optimizer's moments, counter and
    // learning rate), then update 1 on its result -- the same fp32 operations in the same order as two launches would perform,
    // p stays in a register between them instead of making a round trip through HBM
    const float ss0 = lrs0.lr[d.group & 7] * step0[4 * d.slot + 1], bc0 = step0[4 * d.slot + 2];
    const bool vec2 = ((reinterpret_cast<size_t>(t.p) | reinterpret_cast<size_t>(t.g) | reinterpret_cast<size_t>(t.m) | reinterpret_cast<size_t>(t.v) |
                        reinterpret_cast<size_t>(d.m) | reinterpret_cast<size_t>(d.v)) & 15) == 0;
#pragma unroll
    for (int it = 0; it < kAdamChunk / 1024; ++it) {
      const long long i = base + it * 1024 + threadIdx.x * 4;
      if (i >= t.numel) break;
      if (vec2 && i + 3 < t.numel) {
        f32x4 p = *reinterpret_cast<const f32x4*>(t.p + i), g = *reinterpret_cast<const f32x4*>(t.g + i);
        f32x4 m0 = *reinterpret_cast<const f32x4*>(d.m + i), v0 = *reinterpret_cast<const f32x4*>(d.v + i);
        f32x4 m1, v1;
        if (!skip) { m1 = *reinterpret_cast<const f32x4*>(t.m + i); v1 = *reinterpret_cast<const f32x4*>(t.v + i); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gr = __fmul_rn(g[r], inv_scale);
          float pe = p[r], a = m0[r], b = v0[r];
          adam_elem(pe, gr, a, b, beta1, beta2, omb1, omb2, ss0, bc0, eps);
          m0[r] = a; v0[r] = b;
          if (!skip) {
            a = m1[r]; b = v1[r];
            adam_elem(pe, gr, a, b, beta1, beta2, omb1, omb2, step_size, inv_sqrt_bc2, eps);
            m1[r] = a; v1[r] = b;
          }
          p[r] = pe;
        }
        *reinterpret_cast<f32x4*>(t.p + i) = p; *reinterpret_cast<f32x4*>(d.m + i) = m0; *reinterpret_cast<f32x4*>(d.v + i) = v0;
        if (!skip) { *reinterpret_cast<f32x4*>(t.m + i) = m1; *reinterpret_cast<f32x4*>(t.v + i) = v1; }
      } else {
        for (int r = 0; r < 4 && i + r < t.numel; ++r) {
          const float gr = __fmul_rn(t.g[i + r], inv_scale);
          float p = t.p[i + r], m0 = d.m[i + r], v0 = d.v[i + r];
          adam_elem(p, gr, m0, v0, beta1, beta2, omb1, omb2, ss0, bc0, eps);
          d.m[i + r] = m0; d.v[i + r] = v0;
          if (!skip) {
            float m1 = t.m[i + r], v1 = t.v[i + r];
            adam_elem(p, gr, m1, v1, beta1, beta2, omb1, omb2, step_size, inv_sqrt_bc2, eps);
            t.m[i + r] = m1; t.v[i + r] = v1;
          }
          t.p[i + r] = p;
        }
      }
    }
    return;
  }
  const bool vec = ((reinterpret_cast<size_t>(t.p) | reinterpret_cast<size_t>(t.g) | reinterpret_cast<size_t>(t.m) |
                     reinterpret_cast<size_t>(t.v)) & 15) == 0;
#pragma unroll
  for (int it = 0; it < kAdamChunk / 1024; ++it) {
    const long long i = base + it * 1024 + threadIdx.x * 4;
    if (i >= t.numel) break;
    if (vec && i + 3 < t.numel) {
      f32x4 p = *reinterpret_cast<const f32x4*>(t.p + i), g = *reinterpret_cast<const f32x4*>(t.g + i);
      f32x4 m = *reinterpret_cast<const f32x4*>(t.m + i), v = *reinterpret_cast<const f32x4*>(t.v + i);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pe = p[r], a = m[r], b = v[r];
        adam_elem(pe, __fmul_rn(g[r], inv_scale), a, b, beta1, beta2, omb1, omb2, step_size, inv_sqrt_bc2, eps);
        p[r] = pe; m[r] = a; v[r] = b;
      }
      *reinterpret_cast<f32x4*>(t.p + i) = p; *reinterpret_cast<f32x4*>(t.m + i) = m; *reinterpret_cast<f32x4*>(t.v + i) = v;
    } else {
      for (int r = 0; r < 4 && i + r < t.numel; ++r) {
        float p = t.p[i + r], m = t.m[i + r], v = t.v[i + r];
        adam_elem(p, __fmul_rn(t.g[i + r], inv_scale), m, v, beta1, beta2, omb1, omb2, step_size, inv_sqrt_bc2, eps);
        t.m[i + r] = m; t.v[i + r] = v; t.p[i + r] = p;
      }
    }
  }
}

// found[0] = 1 if any gradient element of the table's tensors is inf / nan (GradScaler's check, one launch for all tensors)
template <bool ARGP>
__global__ __launch_bounds__(256) void grads_nonfinite_kernel(const AdamTensor* __restrict__ table, const unsigned long long* __restrict__ g_ptrs,
                                                             GradPtrArgs gargs, const int* __restrict__ blk_tensor,
                                                             const int* __restrict__ blk_chunk, float* __restrict__ found) {
  const int ti = blk_tensor[blockIdx.x];
  const long long numel = table[ti].numel;
  const float* g = ARGP ? reinterpret_cast<const float*>(gargs.p[ti]) : (g_ptrs ? reinterpret_cast<const float*>(g_ptrs[ti]) : table[ti].g);
  const long long base = (long long)blk_chunk[blockIdx.x] * kAdamChunk;
  bool bad = false;
  const bool vec = (reinterpret_cast<size_t>(g) & 15) == 0;
#pragma unroll
  for (int it = 0; it < kAdamChunk / 1024; ++it) {
    const long long i = base + it * 1024 + threadIdx.x * 4;
    if (i >= numel) break;
    if (vec && i + 3 < numel) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
      for (int r = 0; r < 4; ++r) bad |= !(fabsf(v[r]) <= 3.4028234e38f);
    } else {
      for (int r = 0; r < 4 && i + r < numel; ++r) bad |= !(fabsf(g[i + r]) <= 3.4028234e38f);
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) found[0] = 1.f;
}

__global__ void set_f32_kernel(float* p, float v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = v; }

struct FoundPtrs { const float* p[8]; int n; };

// torch._amp_update_scale_ over the OR of up to 8 per-optimizer found_inf flags (GradScaler.update sums them first: one more launch)
__global__ void amp_update_scale_kernel(float* scale, int* growth_tracker, FoundPtrs f, float growth, float backoff, int interval, int clear) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float found = 0.f;
  for (int i = 0; i < f.n; ++i) {
    found += *f.p[i];
    if (clear) *const_cast<float*>(f.p[i]) = 0.f;       // consumed: the next step's checks find their flags cleared (no launch for it)
  }
  if (found != 0.f) {
    *scale = *scale * backoff;
    *growth_tracker = 0;
  } else {
    const int ok = *growth_tracker + 1;
    if (ok == interval) {
      const float ns = *scale * growth;
      if (fabsf(ns) <= 3.4028234e38f) *scale = ns;
      *growth_tracker = 0;
    } else {
      *growth_tracker = ok;
    }
  }
}

}  // namespace

static bool pack_grad_ptrs(const void* const* host, int n, GradPtrArgs* out) {
  if (!host || n <= 0 || n > kArgPtrs) return false;
  for (int i = 0; i < n; ++i) out->p[i] = (unsigned long long)(uintptr_t)host[i];
  out->n = n;
  return true;
}

extern "C" int neraf_grads_nonfinite(neraf_ctx* ctx, const void* table, const void* g_ptrs, const int* blk_tensor, const int* blk_chunk,
                                     int n_blocks, float* found_inf, const void* const* g_ptrs_host, int n_ptrs, int found_is_zero,
                                     neraf_stream_t stream) {
  if (!table || !blk_tensor || !blk_chunk || n_blocks <= 0 || !found_inf) return neraf_fail(ctx, NERAF_EINVAL, "grads_nonfinite: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (!found_is_zero) hipLaunchKernelGGL(set_f32_kernel, dim3(1), dim3(64), 0, st, found_inf, 0.f);
  GradPtrArgs ga{};
  if (pack_grad_ptrs(g_ptrs_host, n_ptrs, &ga))
    hipLaunchKernelGGL(grads_nonfinite_kernel<true>, dim3((unsigned)n_blocks), dim3(256), 0, st, (const AdamTensor*)table, nullptr, ga, blk_tensor,
                       blk_chunk, found_inf);
  else
    hipLaunchKernelGGL(grads_nonfinite_kernel<false>, dim3((unsigned)n_blocks), dim3(256), 0, st, (const AdamTensor*)table,
                       (const unsigned long long*)g_ptrs, ga, blk_tensor, blk_chunk, found_inf);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_amp_update_scale(neraf_ctx* ctx, float* scale, int32_t* growth_tracker, const float* const* found_infs, int n,
                                      double growth_factor, double backoff_factor, int growth_interval, int clear_flags,
                                      neraf_stream_t stream) {
  if (!scale || !growth_tracker || !found_infs || n < 1 || n > 8) return neraf_fail(ctx, NERAF_EINVAL, "amp_update_scale: 1..8 found_inf flags");
  FoundPtrs f{};
  f.n = n;
  for (int i = 0; i < n; ++i) { if (!found_infs[i]) return neraf_fail(ctx, NERAF_EINVAL, "amp_update_scale: null flag"); f.p[i] = found_infs[i]; }
  hipLaunchKernelGGL(amp_update_scale_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scale, (int*)growth_tracker, f, (float)growth_factor,
                     (float)backoff_factor, growth_interval, clear_flags);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_fused_adam_chunk(void) { return kAdamChunk; }

extern "C" int neraf_fused_adam(neraf_ctx* ctx, const void* table, const void* g_ptrs, const int* blk_tensor, const int* blk_chunk,
                                int n_blocks, const float* group_lr, int n_groups, int n_tensors, double beta1, double beta2, double eps,
                                float* step, const float* grad_scale, const float* found_inf, neraf_stream_t stream) {
  return neraf_fused_adam_dual(ctx, table, g_ptrs, blk_tensor, blk_chunk, n_blocks, group_lr, n_groups, n_tensors, beta1, beta2, eps, step, grad_scale,
                               found_inf, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, stream);
}

extern "C" int neraf_fused_adam_dual(neraf_ctx* ctx, const void* table, const void* g_ptrs, const int* blk_tensor, const int* blk_chunk,
                                     int n_blocks, const float* group_lr, int n_groups, int n_tensors, double beta1,
                                     double beta2, double eps, float* step, const float* grad_scale, const float* found_inf,
                                     const void* dual, const float* step0, const float* found_inf0, const float* group_lr0, int n_groups0,
                                     const void* const* g_ptrs_host, int n_ptrs, neraf_stream_t stream) {
  if (!table || n_blocks < 0 || (n_blocks > 0 && (!blk_tensor || !blk_chunk)) || !step || !group_lr || n_groups < 1 || n_groups > 8 || n_tensors < 0 ||
      (dual && (!step0 || !group_lr0 || n_groups0 < 1 || n_groups0 > 8)))
    return neraf_fail(ctx, NERAF_EINVAL, "fused_adam: bad arguments (1..8 parameter groups)");
  AdamLrs lrs{}, lrs0{};
  for (int i = 0; i < n_groups; ++i) lrs.lr[i] = group_lr[i];
  if (dual) for (int i = 0; i < n_groups0; ++i) lrs0.lr[i] = group_lr0[i];
  hipStream_t st = (hipStream_t)stream;
  // counters of EVERY record of the table advance here -- also those of tensors whose element update this optimizer defers to the
  // next optimizer's launch (their workgroups are simply absent from blk_tensor / blk_chunk)
  // n_tensors == 0: the counters were advanced by an earlier call over this table (the flush of a deferred update, neraf_amd/optim.py)
  if (n_tensors > 0)
    hipLaunchKernelGGL(adam_advance_step_kernel, dim3((unsigned)((n_tensors + 63) / 64)), dim3(64), 0, st, step, found_inf, beta1, beta2,
                       (const AdamTensor*)table, n_tensors);
  if (n_blocks > 0) {
    GradPtrArgs ga{};
    if (pack_grad_ptrs(g_ptrs_host, n_ptrs, &ga))
      hipLaunchKernelGGL(fused_adam_kernel<true>, dim3((unsigned)n_blocks), dim3(256), 0, st, (const AdamTensor*)table, nullptr, ga,
                         blk_tensor, blk_chunk, lrs, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, step,
                         grad_scale, found_inf, (const AdamDual*)dual, step0, found_inf0, lrs0);
    else
      hipLaunchKernelGGL(fused_adam_kernel<false>, dim3((unsigned)n_blocks), dim3(256), 0, st, (const AdamTensor*)table,
                         (const unsigned long long*)g_ptrs, ga, blk_tensor, blk_chunk, lrs, (float)beta1, (float)beta2, (float)(1.0 - beta1),
                         (float)(1.0 - beta2), (float)eps, step, grad_scale, found_inf, (const AdamDual*)dual, step0, found_inf0, lrs0);
  }
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

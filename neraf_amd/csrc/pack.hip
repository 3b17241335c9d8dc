// fp32 master parameters of the radiance half -> the fp16 copies its kernels read, in TWO launches for the whole model instead of
// ~15 small torch launches per optimizer step (three table.half(), torch.cat of the MLP weights + index gather + half() for the forward
// and the backward fragments, embedding.half(), proposal-weight cat + half()).  tiny-cuda-nn keeps the same pair (fp32 master, fp16
// working copy) inside its optimizer; here the host layer owns both and refreshes the copies once per parameter state.
#include "common.h"

namespace {

constexpr int kMaxSeg = 12;

// dst[i] = (half) src[i] for up to kMaxSeg contiguous segments (hash tables, embedding rows, proposal MLP weights)
struct CvtSegTable { int n; int begin[kMaxSeg + 1]; const float* src[kMaxSeg]; half_t* dst[kMaxSeg]; long long len[kMaxSeg]; };

__global__ __launch_bounds__(256) void cvt_f16_segments_kernel(CvtSegTable t) {
  int si = 0;
  while (si + 1 < t.n && (int)blockIdx.x >= t.begin[si + 1]) ++si;
  const long long base = ((long long)(blockIdx.x - t.begin[si]) * 256 + threadIdx.x) * 8;
  const long long len = t.len[si];
  if (base >= len) return;
  const float* s = t.src[si] + base;
  half_t* d = t.dst[si] + base;
  if (base + 8 <= len && ((reinterpret_cast<size_t>(s) & 15) == 0) && ((reinterpret_cast<size_t>(d) & 15) == 0)) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
    half8 h;
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = (half_t)a[j]; h[4 + j] = (half_t)b[j]; }
    *reinterpret_cast<half8*>(d) = h;
  } else {
    for (int j = 0; j < 8 && base + j < len; ++j) d[j] = (half_t)s[j];
  }
}

// dst[i] = (half) flat[index[i]] where `flat` is the virtual concatenation of up to 8 source tensors followed by one zero element
struct GatherTable { int nsrc; const float* src[8]; long long end[8]; const long long* index; half_t* dst; long long n; };

__global__ __launch_bounds__(256) void gather_f16_kernel(GatherTable t) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= t.n) return;
  const long long k = t.index[i];
  float v = 0.f;
  long long lo = 0;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s < t.nsrc) {
      if (k >= lo && k < t.end[s]) v = t.src[s][k - lo];
      lo = t.end[s];
    }
  }
  t.dst[i] = (half_t)v;
}

}  // namespace

extern "C" int neraf_cvt_f16_segments(neraf_ctx* ctx, const float* const* src, void* const* dst, const long long* len, int n,
                                      neraf_stream_t stream) {
  if (!src || !dst || !len || n <= 0 || n > kMaxSeg) return neraf_fail(ctx, NERAF_EINVAL, "cvt_f16_segments: 1..12 segments");
  CvtSegTable t{};
  t.n = n;
  for (int i = 0; i < n; ++i) {
    if (!src[i] || !dst[i] || len[i] <= 0) return neraf_fail(ctx, NERAF_EINVAL, "cvt_f16_segments: null segment");
    t.src[i] = src[i]; t.dst[i] = (half_t*)dst[i]; t.len[i] = len[i];
    t.begin[i + 1] = t.begin[i] + (int)((len[i] + 2047) / 2048);
  }
  hipLaunchKernelGGL(cvt_f16_segments_kernel, dim3((unsigned)t.begin[n]), dim3(256), 0, (hipStream_t)stream, t);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_gather_f16(neraf_ctx* ctx, const float* const* src, const long long* src_len, int nsrc, const long long* index,
                                void* dst, long long n, neraf_stream_t stream) {
  if (!src || !src_len || nsrc <= 0 || nsrc > 8 || !index || !dst || n <= 0) return neraf_fail(ctx, NERAF_EINVAL, "gather_f16: 1..8 sources");
  GatherTable t{};
  t.nsrc = nsrc; t.index = index; t.dst = (half_t*)dst; t.n = n;
  long long acc = 0;
  for (int i = 0; i < nsrc; ++i) { if (!src[i] || src_len[i] <= 0) return neraf_fail(ctx, NERAF_EINVAL, "gather_f16: null source"); acc += src_len[i]; t.src[i] = src[i]; t.end[i] = acc; }
  hipLaunchKernelGGL(gather_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

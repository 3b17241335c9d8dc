// NAcF (neural acoustic field) path: query prologue, weight packing, the MLP stack on the MFMA
// GEMM, its backward, and the STFT loss.  Follows NeRAF_model.py:531-566 (get_outputs),
// NeRAF_field.py:37-65 (NeRAFAudioSoundField) and NeRAF_evaluator.py:88-108 (STFTLoss).
//
// Layer-0 split: the first 1024 inputs of every row are the same ResNet3D feature vector
// (NeRAF_model.py:557-558), so  z0 = q . W0[:,1024:]^T + (W0[:,:1024] . feat + b0) -- one
// GEMV folded into the bias plus a K=163(->192) GEMM instead of a K=1187 GEMM.
#include "common.h"
#include <algorithm>

namespace {

constexpr int TRUNK[5] = {5096, 2048, 1024, 1024, 0};  // [4] = W (desc)
constexpr int KQ_PAD = 192;                            // 163 -> 192 (multiple of 64)
constexpr int QT_ROWS = 256;                           // q^T rows padded to a multiple of 128

struct Dims {
  int n[6];      // logical widths: trunk 0..4, heads (C*F)
  int np[6];     // padded to multiples of 128
  int k[6];      // logical input width per layer (layer 0: n_query for the split path)
  int kp[6];     // padded input width
  int kdense, kdense_p;  // dense layer 0: n_feat + n_query, padded to 128 multiple
};

Dims make_dims(const neraf_nacf_desc* d) {
  Dims D{};
  for (int i = 0; i < 4; ++i) D.n[i] = TRUNK[i];
  D.n[4] = d->W;
  D.n[5] = d->C * d->F;
  for (int i = 0; i < 6; ++i) D.np[i] = round_up(D.n[i], 128);
  D.k[0] = d->n_query; D.kp[0] = KQ_PAD;
  for (int i = 1; i < 6; ++i) { D.k[i] = D.n[i - 1]; D.kp[i] = D.np[i - 1]; }
  D.kdense = d->n_feat + d->n_query;
  D.kdense_p = round_up(D.kdense, 128);
  return D;
}

// ---- packed weight blob layout -----------------------------------------------------------
struct PackLayout {
  size_t w[6], wt[6];     // fp16 [np][kp] and [kp_as_rows(np_prev pad 128)][np]
  size_t w0d, w0dt;       // dense layer 0
  size_t bias[6];         // fp32 [np]
  size_t total;
};

PackLayout make_pack_layout(const neraf_nacf_desc* d, const Dims& D) {
  PackLayout L{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += round_up_sz(bytes, 256); return o; };
  for (int l = 0; l < 6; ++l) {
    L.w[l] = take((size_t)D.np[l] * D.kp[l] * 2);
    // transposed copy is the B operand of the dX GEMM: rows = input width padded to 128
    L.wt[l] = (l == 0) ? 0 : take((size_t)round_up(D.kp[l], 128) * D.np[l] * 2);
  }
  if (d->dense_l0) {
    L.w0d = take((size_t)D.np[0] * D.kdense_p * 2);
    L.w0dt = take((size_t)D.kdense_p * D.np[0] * 2);
  }
  for (int l = 0; l < 6; ++l) L.bias[l] = take((size_t)D.np[l] * 4);
  L.total = off;
  return L;
}

// ---- workspace layout ----------------------------------------------------------------------
struct WsLayout {
  int Mpad;
  size_t q, qT;            // fp16 [Mpad][192], [256][Mpad]
  size_t hd, hdT;          // dense input fp16 [Mpad][kdense_p], [kdense_p][Mpad]
  size_t h[5], hT[5];      // activations fp16 [Mpad][np], [np][Mpad]
  size_t bias0;            // fp32 [np0] effective layer-0 bias
  size_t dz[2], dzT[2];    // backward ping-pong fp16 [Mpad][5120] / [5120+128][Mpad]
  size_t colsum[6];        // fp32 [np]; deterministic mode: [Mpad / 32][np] slots (one per 32 rows, GemmParams::stat_det)
  size_t scale;            // fp32 [4]: {S, 1/S, amax bits, -} backward auto-scale
  size_t dfeat_part;       // deterministic mode: fp32 [64][n_feat] row-slice partials of d feat
  int col_slots;           // 1, or Mpad / 32 in deterministic mode
  size_t total;
};

WsLayout make_ws_layout(const neraf_nacf_desc* d, const Dims& D, int B, int training) {
  WsLayout L{};
  // rows padded to the tile height the GEMM dispatcher can use (csrc/gemm_f16.hip dispatch_tile: 128-row tiles need Mpad % 128, the
  // wide 256-row bodies Mpad % 256): NERAF_NACF_MALIGN (measurement knob)
  static const int kAlign = [] { const char* e = getenv("NERAF_NACF_MALIGN"); const int v = e ? atoi(e) : 128; return (v == 256 || v == 512) ? v : 128; }();
  L.Mpad = round_up(B, kAlign);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += round_up_sz(bytes, 256); return o; };
  const size_t M = L.Mpad;
  L.q = take(M * KQ_PAD * 2);
  L.qT = take((size_t)QT_ROWS * M * 2);
  if (d->dense_l0) {
    L.hd = take(M * D.kdense_p * 2);
    L.hdT = take((size_t)D.kdense_p * M * 2);
  }
  for (int l = 0; l < 5; ++l) {
    L.h[l] = take(M * D.np[l] * 2);
    L.hT[l] = training ? take((size_t)D.np[l] * M * 2) : 0;
  }
  L.bias0 = take((size_t)D.np[0] * 4);
  if (training) {
    int maxw = 0;
    for (int l = 0; l < 6; ++l) maxw = D.np[l] > maxw ? D.np[l] : maxw;
    for (int i = 0; i < 2; ++i) {
      L.dz[i] = take(M * maxw * 2);
      L.dzT[i] = take((size_t)(maxw + 128) * M * 2);
    }
    L.col_slots = neraf_deterministic() ? L.Mpad / 32 : 1;
    for (int l = 0; l < 6; ++l) L.colsum[l] = take((size_t)L.col_slots * D.np[l] * 4);
    L.scale = take(256);
    if (neraf_deterministic()) L.dfeat_part = take((size_t)64 * (d->n_feat > 0 ? d->n_feat : 1) * 4);
  }
  L.total = off;
  return L;
}

// ---- kernels ---------------------------------------------------------------------------------

// fp32 [R, Cc] (ld) -> fp16 padded [Rpad, Cpad] and/or its transpose [CpadT, RpadT]; zero fill outside.
// mode 1: value = src * (10 - aux^2/10) (tanh*10 backward, aux = forward output, same layout as src).
// Optional colsum (fp32 [>=Cpad]) of the written values.
__global__ __launch_bounds__(256) void cvt_pad_transpose_kernel(
    const float* __restrict__ src, const float* __restrict__ aux, const float* __restrict__ scale, int ld, int R, int Cc,
    int mode, half_t* __restrict__ dst, int ld_dst, int Rpad, int Cpad,
    half_t* __restrict__ dstT, int ld_dstT, int RpadT, int CpadT, float* __restrict__ colsum, int colsum_slot_stride) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const float sc = scale ? scale[0] : 1.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + i * 8, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < Cc) {
      v = src[(size_t)r * ld + c];
      if (mode == 1) {
        const float o = aux[(size_t)r * ld + c];
        v *= (10.f - o * o * 0.1f) * sc;
      }
    }
    tile[ty + i * 8][tx] = v;
    if (dst && r < Rpad && c < Cpad) dst[(size_t)r * ld_dst + c] = (half_t)v;
  }
  __syncthreads();
  if (dstT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + i * 8, r = r0 + tx;   // transposed: row index c, column index r
      if (c < CpadT && r < RpadT) dstT[(size_t)c * ld_dstT + r] = (half_t)tile[tx][ty + i * 8];
    }
  }
  if (colsum && ty == 0) {
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) s += (float)(half_t)tile[i][tx];
    const int c = c0 + tx;
    if (c < Cpad) {
      if (colsum_slot_stride > 0) colsum[(size_t)blockIdx.y * colsum_slot_stride + c] = s;      // deterministic mode: this row block's slot
      else atomicAdd(colsum + c, s);
    }
  }
}

__global__ void pad_copy_f32_kernel(const float* __restrict__ src, int n, float* __restrict__ dst, int npad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad) dst[i] = (i < n) ? src[i] : 0.f;
}

// ---- one-launch forms of the per-layer launches above (descriptor tables in the kernel arguments) ----------------------------
// Weight packing runs every training step; eight small launches cost more than their 80 MB of traffic.
struct CvtJob {
  const float* src; half_t* dst; half_t* dstT;
  int ld, R, Cc, ld_dst, Rpad, Cpad, ld_dstT, RpadT, CpadT;
  int tiles_x, tile_begin;
};
constexpr int kMaxCvtJobs = 10;
struct CvtTable { int n; int total; CvtJob j[kMaxCvtJobs]; };

__global__ __launch_bounds__(256) void cvt_pad_transpose_grouped_kernel(CvtTable t) {
  __shared__ float tile[32][33];
  int ji = 0;
  while (ji + 1 < t.n && (int)blockIdx.x >= t.j[ji + 1].tile_begin) ++ji;
  const CvtJob& J = t.j[ji];
  const int b = blockIdx.x - J.tile_begin;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = (b % J.tiles_x) * 32, r0 = (b / J.tiles_x) * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + i * 8, c = c0 + tx;
    float v = 0.f;
    if (r < J.R && c < J.Cc) v = J.src[(size_t)r * J.ld + c];
    tile[ty + i * 8][tx] = v;
    if (J.dst && r < J.Rpad && c < J.Cpad) J.dst[(size_t)r * J.ld_dst + c] = (half_t)v;
  }
  __syncthreads();
  if (J.dstT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + i * 8, r = r0 + tx;
      if (c < J.CpadT && r < J.RpadT) J.dstT[(size_t)c * J.ld_dstT + r] = (half_t)tile[tx][ty + i * 8];
    }
  }
}

// up to 16 segments: dst[i] = i < n ? src[i] * (mul ? mul[0] : 1) : 0 for i < npad
// nslots > 1 (deterministic mode): the source is [nslots][stride] per-row-block partial sums, added here in slot order
struct SegCopyTable { int n; int begin[17]; const float* src[16]; float* dst[16]; int len[16]; int npad[16]; int stride[16]; const float* mul;
                      int nslots; };

__global__ __launch_bounds__(256) void seg_copy_kernel(SegCopyTable t) {
  int si = 0;
  while (si + 1 < t.n && (int)blockIdx.x >= t.begin[si + 1]) ++si;
  const int i = (blockIdx.x - t.begin[si]) * 256 + threadIdx.x;
  if (i >= t.npad[si]) return;
  float v = 0.f;
  if (i < t.len[si]) {
    if (t.nslots > 1) {
      float a = 0.f, b = 0.f, c = 0.f, e = 0.f;
      int sl = 0;
      for (; sl + 4 <= t.nslots; sl += 4) {
        a += t.src[si][(size_t)sl * t.stride[si] + i]; b += t.src[si][(size_t)(sl + 1) * t.stride[si] + i];
        c += t.src[si][(size_t)(sl + 2) * t.stride[si] + i]; e += t.src[si][(size_t)(sl + 3) * t.stride[si] + i];
      }
      for (; sl < t.nslots; ++sl) a += t.src[si][(size_t)sl * t.stride[si] + i];
      v = (a + b) + (c + e);
    } else {
      v = t.src[si][i];
    }
    v *= t.mul ? t.mul[0] : 1.f;
  }
  t.dst[si][i] = v;
}

static void seg_copy_add(SegCopyTable& t, float* dst, const float* src, int n, int npad, int stride = 0) {
  const int k = t.n++;
  t.src[k] = src; t.dst[k] = dst; t.len[k] = n; t.npad[k] = npad; t.stride[k] = stride;
  t.begin[k + 1] = t.begin[k] + (npad + 255) / 256;
}

// bias0_eff[n] = b0[n] + sum_k W0[n][k] * feat[k]  (k < n_feat), one wave per output row.
__global__ __launch_bounds__(256) void feat_gemv_kernel(const float* __restrict__ W0, int ldw, const float* __restrict__ b0,
                                                       const float* __restrict__ feat, int n_feat, int N, int Npad,
                                                       float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= Npad) return;
  float s = 0.f;
  if (n < N) {
    const float* w = W0 + (size_t)n * ldw;
    for (int k = lane; k < n_feat; k += 64) s += w[k] * feat[k];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[n] = (n < N) ? s + b0[n] : 0.f;
}

// dfeat[k] = sum_n db0[n] * W0[n][k]; grid (ceil(n_feat/256), nsplit), atomics into zeroed dfeat.
__global__ __launch_bounds__(256) void dfeat_kernel(const float* __restrict__ W0, int ldw, const float* __restrict__ db0,
                                                   int N, int n_feat, float* __restrict__ dfeat, float* __restrict__ part) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int per = (N + gridDim.y - 1) / gridDim.y;
  const int n0 = blockIdx.y * per, n1 = min(N, n0 + per);
  if (k >= n_feat) return;
  // eight loads in flight per lane; 64 row splits: with 512 the kernel was 524 k atomics on 32 cache lines
  float s = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int n = n0;
  for (; n + 8 <= n1; n += 8) {
    float w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = W0[(size_t)(n + u) * ldw + k];
    s += db0[n] * w[0] + db0[n + 4] * w[4]; s1 += db0[n + 1] * w[1] + db0[n + 5] * w[5];
    s2 += db0[n + 2] * w[2] + db0[n + 6] * w[6]; s3 += db0[n + 3] * w[3] + db0[n + 7] * w[7];
  }
  for (; n < n1; ++n) s += db0[n] * W0[(size_t)n * ldw + k];
  // part != null (deterministic mode): row slice blockIdx.y stores its partial, dfeat_fold_kernel adds the slices in order
  if (part) part[(size_t)blockIdx.y * n_feat + k] = (s + s1) + (s2 + s3);
  else atomicAdd(dfeat + k, (s + s1) + (s2 + s3));
}

__global__ __launch_bounds__(256) void dfeat_fold_kernel(const float* __restrict__ part, int nsplit, int n_feat, float* __restrict__ dfeat) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n_feat) return;
  float a = 0.f, b = 0.f, c = 0.f, e = 0.f;
  int sl = 0;
  for (; sl + 4 <= nsplit; sl += 4) {
    a += part[(size_t)sl * n_feat + k]; b += part[(size_t)(sl + 1) * n_feat + k];
    c += part[(size_t)(sl + 2) * n_feat + k]; e += part[(size_t)(sl + 3) * n_feat + k];
  }
  for (; sl < nsplit; ++sl) a += part[(size_t)sl * n_feat + k];
  dfeat[k] = (a + b) + (c + e);
}

// dW0[:, :n_feat] = outer(db0, feat)
__global__ __launch_bounds__(256) void outer_kernel(const float* __restrict__ db0, const float* __restrict__ feat, int N,
                                                   int n_feat, float* __restrict__ dW0, int ldw, float* __restrict__ zero_nfeat) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (k < n_feat && n < N) dW0[(size_t)n * ldw + k] = db0[n] * feat[k];
  // row 0 of the grid also clears d feat, which the next launch (dfeat_kernel) accumulates into: no fill launch in between
  if (zero_nfeat && n == 0 && k < n_feat) zero_nfeat[k] = 0.f;
}

// Query prologue, NeRAF_model.py:533-551.  One thread per (row, column) of q[B,192]; `tmajor`
// selects the thread->element order so that either q (row-major) or q^T is written coalesced.
struct EncodeArgs {
  const int64_t* tq; const double* mic; const double* src; const double* rot;
  float aabb[6]; float inv_tmax; int B, Mpad;
  half_t* q; half_t* qT;
  int pstride;      // doubles between the pose rows of consecutive queries: 3, or 0 when every query shares row 0 (one RIR's T time queries)
};

__global__ __launch_bounds__(256) void encode_queries_kernel(EncodeArgs a, int tmajor) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  int row, col;
  if (tmajor) {
    if (idx >= (long)QT_ROWS * a.Mpad) return;
    col = (int)(idx / a.Mpad); row = (int)(idx % a.Mpad);
  } else {
    if (idx >= (long)a.Mpad * KQ_PAD) return;
    row = (int)(idx / KQ_PAD); col = (int)(idx % KQ_PAD);
  }
  float v = 0.f;
  if (row < a.B && col < 163) {
    const double TWO_PI = 6.283185307179586476925286766559;
    const double HALF_PI = 1.5707963267948966192313216916398;
    if (col < 21) {
      // time: float32 arithmetic as in the reference (time_query.float()/(T-1), NeRFEncoding in f32)
      const float t = (float)a.tq[row] * a.inv_tmax;
      if (col == 20) v = t;
      else {
        const int k = col % 10;
        const float f = exp2f((float)k * (8.0f / 9.0f));
        const float s = (6.283185307179586f * t) * f;
        v = (col < 10) ? (float)sin((double)s) : (float)sin((double)(s + 1.5707963267948966f));
      }
    } else if (col < 147) {
      const bool is_mic = col < 84;
      const double* P = (is_mic ? a.mic : a.src) + (size_t)row * a.pstride;
      const int c = is_mic ? col - 21 : col - 84;
      // SceneBox normalisation + in-box selector in float64 (poses are float64, NeRAF_dataset.py:129)
      double xn[3];
      bool inside = true;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        xn[d] = (P[d] - (double)a.aabb[d]) / ((double)a.aabb[3 + d] - (double)a.aabb[d]);
        inside = inside && (xn[d] > 0.0) && (xn[d] < 1.0);
      }
      if (!inside) { xn[0] = xn[1] = xn[2] = 0.0; }
      if (c >= 60) v = (float)xn[c - 60];
      else {
        const int cc = c < 30 ? c : c - 30;
        const int d = cc / 10, k = cc % 10;
        const double f = (double)exp2f((float)k * (8.0f / 9.0f));
        const double s = TWO_PI * xn[d] * f;
        v = (float)sin(c < 30 ? s : s + HALF_PI);
      }
    } else {
      // SH degree 4 on d = 2*rot - 1 (tiny-cuda-nn SphericalHarmonics, float)
      const double* R = a.rot + (size_t)row * a.pstride;
      const float x = (float)R[0] * 2.f - 1.f, y = (float)R[1] * 2.f - 1.f, z = (float)R[2] * 2.f - 1.f;
      const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
      switch (col - 147) {
        case 0: v = 0.28209479177387814f; break;
        case 1: v = -0.48860251190291987f * y; break;
        case 2: v = 0.48860251190291987f * z; break;
        case 3: v = -0.48860251190291987f * x; break;
        case 4: v = 1.0925484305920792f * xy; break;
        case 5: v = -1.0925484305920792f * yz; break;
        case 6: v = 0.94617469575755997f * z2 - 0.31539156525251999f; break;
        case 7: v = -1.0925484305920792f * xz; break;
        case 8: v = 0.54627421529603959f * x2 - 0.54627421529603959f * y2; break;
        case 9: v = 0.59004358992664352f * y * (-3.0f * x2 + y2); break;
        case 10: v = 2.8906114426405538f * xy * z; break;
        case 11: v = 0.45704579946446572f * y * (1.0f - 5.0f * z2); break;
        case 12: v = 0.3731763325901154f * z * (5.0f * z2 - 3.0f); break;
        case 13: v = 0.45704579946446572f * x * (1.0f - 5.0f * z2); break;
        case 14: v = 1.4453057213202769f * z * (x2 - y2); break;
        default: v = 0.59004358992664352f * x * (-x2 + 3.0f * y2); break;
      }
    }
  }
  if (tmajor) a.qT[(size_t)col * a.Mpad + row] = (half_t)v;
  else a.q[(size_t)row * KQ_PAD + col] = (half_t)v;
}

// Backward auto-scaling.  The gradient chain dz_l is carried in fp16; to keep it inside fp16's range for ANY
// upstream loss scale (with or without a GradScaler) the head gradient is multiplied by a power of two S chosen
// so that max|dz_5| lands in [256, 512); every fp32 output (dW, db, dfeat, dh) is multiplied back by 1/S.
__global__ __launch_bounds__(256) void amax_dz5_kernel(const float* __restrict__ dout, const float* __restrict__ out, size_t n,
                                                      unsigned* __restrict__ amax_bits) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float o = out[i];
    const float v = fabsf(dout[i] * (10.f - o * o * 0.1f));
    m = (v == v && v > m) ? v : m;   // NaNs are ignored here; they still propagate through the data path
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  // one same-address atomic per block (same-address atomics serialise at ~12 ns each)
  if (threadIdx.x == 0) atomicMax(amax_bits, __float_as_uint(fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]))));
}

__global__ void make_scale_kernel(float* __restrict__ scale) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float amax = __uint_as_float(reinterpret_cast<unsigned*>(scale)[2]);
    float S = 1.f;
    if (amax > 0.f && amax < 3.0e38f) {
      int e = 8 - (int)floorf(log2f(amax));
      e = e > 100 ? 100 : (e < -100 ? -100 : e);
      S = exp2f((float)e);
    }
    scale[0] = S;
    scale[1] = 1.f / S;
  }
}

__global__ void scale_copy_kernel(float* __restrict__ dst, const float* __restrict__ src, int n, const float* __restrict__ mul) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] * mul[0];
}

// ---- STFT loss -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_loss_sums_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            size_t n, int l1, float* __restrict__ part) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float xv = x[i], yv = y[i];
    const float xm = expf(xv) - 1e-3f, ym = expf(yv) - 1e-3f;
    const float dm = ym - xm, dl = xv - yv;
    s0 += dm * dm; s1 += ym * ym; s2 += l1 ? fabsf(dl) : dl * dl;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  __shared__ float sh[3][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh[0][w] = s0; sh[1][w] = s1; sh[2][w] = s2; }
  __syncthreads();
  // this workgroup's partial (no atomics): stft_sums_fold_kernel adds the <= 256 partials in a fixed order
  if (threadIdx.x < 3) part[(size_t)blockIdx.x * 4 + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

__global__ __launch_bounds__(64) void stft_sums_fold_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ sums) {
  const int lane = threadIdx.x;
  float v[3] = {0.f, 0.f, 0.f};
  for (int b = lane; b < nblocks; b += 64)
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] += part[(size_t)b * 4 + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);      // butterfly: every lane ends with the same fixed-order sum
  }
  if (lane < 3) sums[lane] = v[lane];
  if (lane == 3) sums[3] = 0.f;
}

__global__ void stft_loss_final_kernel(const float* __restrict__ sums, float inv_n, const float* __restrict__ weights,
                                       float* __restrict__ losses) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float sc = sqrtf(sums[0]) / sqrtf(sums[1]);   // NeRAF_evaluator.py:26
    const float mag = sums[2] * inv_n;                  // :51 / :53
    losses[0] = weights ? sc * weights[0] : sc;         // loss factors of NeRAF_model.py:592-599
    losses[1] = weights ? mag * weights[1] : mag;
  }
}

__global__ __launch_bounds__(256) void stft_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, size_t n,
                                                           size_t n_total, int l1, const float* __restrict__ sums, const float* g_sc,
                                                           const float* g_mag, const float* __restrict__ weights, float extra,
                                                           float* __restrict__ dx) {
  // upstream scalars: d total / d sc, d total / d mag (device scalars, null = 0) x loss factors x `extra` (data-parallel world size)
  const float w0 = (g_sc ? *g_sc : 0.f) * (weights ? weights[0] : 1.f) * extra;
  const float w1 = (g_mag ? *g_mag : 0.f) * (weights ? weights[1] : 1.f) * extra;
  const float inv_den = w0 / (sqrtf(sums[0]) * sqrtf(sums[1]));
  const float wm = w1 / (float)n_total;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float xv = x[i], yv = y[i];
    const float ex = expf(xv);
    const float xm = ex - 1e-3f, ym = expf(yv) - 1e-3f;
    const float dl = xv - yv;
    const float gm = l1 ? (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f)) : 2.f * dl;
    dx[i] = (xm - ym) * ex * inv_den + gm * wm;
  }
}

// ---- helpers ---------------------------------------------------------------------------------
int cvt_pad_transpose(neraf_ctx* ctx, hipStream_t st, const float* src, const float* aux, const float* scale, int ld, int R,
                      int Cc, int mode,
                      half_t* dst, int ld_dst, int Rpad, int Cpad, half_t* dstT, int ld_dstT, int RpadT, int CpadT,
                      float* colsum, int colsum_slot_stride = 0) {
  int rmax = Rpad, cmax = Cpad;
  if (dstT) { rmax = RpadT > rmax ? RpadT : rmax; cmax = CpadT > cmax ? CpadT : cmax; }
  if (!dst) { rmax = RpadT; cmax = CpadT; }
  dim3 grid((cmax + 31) / 32, (rmax + 31) / 32);
  hipLaunchKernelGGL(cvt_pad_transpose_kernel, grid, dim3(256), 0, st, src, aux, scale, ld, R, Cc, mode, dst, ld_dst,
                     dst ? Rpad : 0, dst ? Cpad : 0, dstT, ld_dstT, RpadT, CpadT, colsum, colsum_slot_stride);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

inline const float* Wptr(const float* const* w, int l) { return w[2 * l]; }
inline const float* Bptr(const float* const* w, int l) { return w[2 * l + 1]; }

}  // namespace

// =================================================================================================
extern "C" size_t neraf_nacf_packed_bytes(const neraf_nacf_desc* d) {
  const Dims D = make_dims(d);
  return make_pack_layout(d, D).total;
}

extern "C" size_t neraf_nacf_workspace_bytes(const neraf_nacf_desc* d, int B, int training) {
  const Dims D = make_dims(d);
  return make_ws_layout(d, D, B, training).total;
}

static int check_desc(neraf_ctx* ctx, const neraf_nacf_desc* d) {
  if (!d || d->n_query != 163 || d->W <= 0 || d->C < 1 || d->C > 8 || d->F <= 0 || d->n_feat < 0)
    return neraf_fail(ctx, NERAF_EINVAL, "nacf: unsupported descriptor (n_query must be 163)");
  return NERAF_OK;
}

extern "C" int neraf_nacf_pack_weights(neraf_ctx* ctx, const neraf_nacf_desc* d, const float* const* w, void* packed,
                                       neraf_stream_t stream) {
  if (int e = check_desc(ctx, d)) return e;
  if (d->C + 6 > kMaxCvtJobs || d->C + 5 > 16) return neraf_fail(ctx, NERAF_EINVAL, "nacf_pack_weights: too many heads");
  hipStream_t st = (hipStream_t)stream;
  const Dims D = make_dims(d);
  const PackLayout L = make_pack_layout(d, D);
  char* P = (char*)packed;
  const int ld0 = D.kdense;
  CvtTable t{};
  auto add = [&](const float* src, int ld, int R, int Cc, half_t* dst, int ld_dst, int Rpad, int Cpad, half_t* dstT, int ld_dstT,
                 int RpadT, int CpadT) {
    CvtJob& J = t.j[t.n++];
    J.src = src; J.dst = dst; J.dstT = dstT; J.ld = ld; J.R = R; J.Cc = Cc; J.ld_dst = ld_dst; J.Rpad = Rpad; J.Cpad = Cpad;
    J.ld_dstT = ld_dstT; J.RpadT = RpadT; J.CpadT = CpadT;
    const int rows = std::max(dst ? Rpad : 0, dstT ? RpadT : 0), cols = std::max(dst ? Cpad : 0, dstT ? CpadT : 0);
    J.tiles_x = (cols + 31) / 32;
    J.tile_begin = t.total;
    t.total += J.tiles_x * ((rows + 31) / 32);
  };
  // layer 0, query half: columns [n_feat, n_feat+163) of W0
  add(Wptr(w, 0) + d->n_feat, ld0, D.n[0], D.k[0], (half_t*)(P + L.w[0]), D.kp[0], D.np[0], D.kp[0], nullptr, 0, 0, 0);
  if (d->dense_l0)
    add(Wptr(w, 0), ld0, D.n[0], D.kdense, (half_t*)(P + L.w0d), D.kdense_p, D.np[0], D.kdense_p, (half_t*)(P + L.w0dt), D.np[0], D.np[0],
        D.kdense_p);
  for (int l = 1; l < 5; ++l)
    add(Wptr(w, l), D.k[l], D.n[l], D.k[l], (half_t*)(P + L.w[l]), D.kp[l], D.np[l], D.kp[l], (half_t*)(P + L.wt[l]), D.np[l], D.np[l],
        round_up(D.kp[l], 128));
  // heads: C tensors [F, W] stacked along rows; the last job also writes the zero rows / columns up to the padded extent
  const int kt5 = round_up(D.kp[5], 128);
  for (int c = 0; c < d->C; ++c) {
    const bool last = c == d->C - 1;
    const int rows = last ? D.np[5] - c * d->F : d->F;
    add(w[2 * (5 + c)], d->W, d->F, d->W, (half_t*)(P + L.w[5]) + (size_t)c * d->F * D.kp[5], D.kp[5], rows, D.kp[5],
        (half_t*)(P + L.wt[5]) + (size_t)c * d->F, D.np[5], rows, kt5);
  }
  hipLaunchKernelGGL(cvt_pad_transpose_grouped_kernel, dim3(t.total), dim3(256), 0, st, t);
  SegCopyTable sc{};
  for (int c = 0; c < d->C; ++c) {
    const bool last = c == d->C - 1;
    seg_copy_add(sc, (float*)(P + L.bias[5]) + (size_t)c * d->F, w[2 * (5 + c) + 1], d->F, last ? D.np[5] - c * d->F : d->F);
  }
  for (int l = 0; l < 5; ++l) seg_copy_add(sc, (float*)(P + L.bias[l]), Bptr(w, l), D.n[l], D.np[l]);
  hipLaunchKernelGGL(seg_copy_kernel, dim3(sc.begin[sc.n]), dim3(256), 0, st, sc);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_nacf_encode_queries(neraf_ctx* ctx, const neraf_nacf_desc* d, const int64_t* time_query,
                                         const double* mic_pose, const double* source_pose, const double* rot,
                                         const float* aabb_host, int max_len, int B, void* workspace, int training,
                                         neraf_stream_t stream) {
  return neraf_nacf_encode_queries_ex(ctx, d, time_query, mic_pose, source_pose, rot, B, aabb_host, max_len, B, workspace, training, stream);
}

extern "C" int neraf_nacf_encode_queries_ex(neraf_ctx* ctx, const neraf_nacf_desc* d, const int64_t* time_query,
                                            const double* mic_pose, const double* source_pose, const double* rot, int pose_rows,
                                            const float* aabb_host, int max_len, int B, void* workspace, int training,
                                            neraf_stream_t stream) {
  if (int e = check_desc(ctx, d)) return e;
  if (B <= 0 || max_len < 2) return neraf_fail(ctx, NERAF_EINVAL, "encode_queries: B>0 and max_len>=2 required");
  if (pose_rows != B && pose_rows != 1) return neraf_fail(ctx, NERAF_EINVAL, "encode_queries: pose_rows is B or 1");
  const Dims D = make_dims(d);
  const WsLayout L = make_ws_layout(d, D, B, training);
  EncodeArgs a{};
  a.tq = time_query; a.mic = mic_pose; a.src = source_pose; a.rot = rot;
  for (int i = 0; i < 6; ++i) a.aabb[i] = aabb_host[i];
  a.inv_tmax = 1.0f / (float)(max_len - 1.0);
  a.B = B; a.Mpad = L.Mpad; a.pstride = pose_rows == B ? 3 : 0;
  a.q = (half_t*)((char*)workspace + L.q);
  a.qT = (half_t*)((char*)workspace + L.qT);
  hipStream_t st = (hipStream_t)stream;
  long n = (long)L.Mpad * KQ_PAD;
  hipLaunchKernelGGL(encode_queries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, 0);
  if (training) {
    n = (long)QT_ROWS * L.Mpad;
    hipLaunchKernelGGL(encode_queries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, 1);
  }
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

// trunk layers 1..4 + heads, shared by the split and dense forward
static int nacf_fwd_tail(neraf_ctx* ctx, const neraf_nacf_desc* d, const Dims& D, const PackLayout& PL, const WsLayout& WL,
                         const char* P, char* ws, int B, float* out, int training, hipStream_t st) {
  for (int l = 1; l < 5; ++l) {
    GemmParams g{};
    g.A = (const half_t*)(ws + WL.h[l - 1]); g.lda = D.np[l - 1];
    g.B = (const half_t*)(P + PL.w[l]); g.ldb = D.kp[l];
    g.M = B; g.N = D.n[l]; g.K = D.kp[l]; g.Mpad = WL.Mpad; g.Npad = D.np[l]; g.alpha = 1.f;
    g.bias = (const float*)(P + PL.bias[l]); g.act = ACT_LEAKY;
    g.C16 = (half_t*)(ws + WL.h[l]); g.ldc16 = D.np[l];
    if (training) { g.C16T = (half_t*)(ws + WL.hT[l]); g.ldc16t = WL.Mpad; }
    if (int e = launch_gemm_f16(ctx, g, st)) return e;
  }
  GemmParams g{};
  g.A = (const half_t*)(ws + WL.h[4]); g.lda = D.np[4];
  g.B = (const half_t*)(P + PL.w[5]); g.ldb = D.kp[5];
  g.M = B; g.N = D.n[5]; g.K = D.kp[5]; g.Mpad = WL.Mpad; g.Npad = D.np[5]; g.alpha = 1.f;
  g.bias = (const float*)(P + PL.bias[5]); g.act = ACT_TANH10;
  g.C32 = out; g.ldc32 = D.n[5];
  return launch_gemm_f16(ctx, g, st);
}

extern "C" int neraf_nacf_fwd(neraf_ctx* ctx, const neraf_nacf_desc* d, const void* packed, const float* const* w,
                              const float* feat, int B, float* out, void* workspace, int training,
                              neraf_stream_t stream) {
  if (int e = check_desc(ctx, d)) return e;
  if (B <= 0) return neraf_fail(ctx, NERAF_EINVAL, "nacf_fwd: B must be positive");
  if (d->n_feat > 0 && !feat) return neraf_fail(ctx, NERAF_EINVAL, "nacf_fwd: feat is null");
  hipStream_t st = (hipStream_t)stream;
  const Dims D = make_dims(d);
  const PackLayout PL = make_pack_layout(d, D);
  const WsLayout WL = make_ws_layout(d, D, B, training);
  const char* P = (const char*)packed;
  char* ws = (char*)workspace;
  float* bias0 = (float*)(ws + WL.bias0);
  hipLaunchKernelGGL(feat_gemv_kernel, dim3(D.np[0] / 4), dim3(256), 0, st, Wptr(w, 0), D.kdense, Bptr(w, 0), feat,
                     d->n_feat, D.n[0], D.np[0], bias0);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  GemmParams g{};
  g.A = (const half_t*)(ws + WL.q); g.lda = KQ_PAD;
  g.B = (const half_t*)(P + PL.w[0]); g.ldb = D.kp[0];
  g.M = B; g.N = D.n[0]; g.K = D.kp[0]; g.Mpad = WL.Mpad; g.Npad = D.np[0]; g.alpha = 1.f;
  g.bias = bias0; g.act = ACT_LEAKY;
  g.C16 = (half_t*)(ws + WL.h[0]); g.ldc16 = D.np[0];
  if (training) { g.C16T = (half_t*)(ws + WL.hT[0]); g.ldc16t = WL.Mpad; }
  if (int e = launch_gemm_f16(ctx, g, st)) return e;
  return nacf_fwd_tail(ctx, d, D, PL, WL, P, ws, B, out, training, st);
}

extern "C" int neraf_nacf_fwd_dense(neraf_ctx* ctx, const neraf_nacf_desc* d, const void* packed, const float* h, int B,
                                    float* out, void* workspace, int training, neraf_stream_t stream) {
  if (int e = check_desc(ctx, d)) return e;
  if (!d->dense_l0) return neraf_fail(ctx, NERAF_EINVAL, "nacf_fwd_dense: descriptor was created without dense_l0");
  if (B <= 0) return neraf_fail(ctx, NERAF_EINVAL, "nacf_fwd_dense: B must be positive");
  hipStream_t st = (hipStream_t)stream;
  const Dims D = make_dims(d);
  const PackLayout PL = make_pack_layout(d, D);
  const WsLayout WL = make_ws_layout(d, D, B, training);
  const char* P = (const char*)packed;
  char* ws = (char*)workspace;
  if (int e = cvt_pad_transpose(ctx, st, h, nullptr, nullptr, D.kdense, B, D.kdense, 0, (half_t*)(ws + WL.hd), D.kdense_p, WL.Mpad,
                                D.kdense_p, training ? (half_t*)(ws + WL.hdT) : nullptr, WL.Mpad, WL.Mpad, D.kdense_p,
                                nullptr))
    return e;
  GemmParams g{};
  g.A = (const half_t*)(ws + WL.hd); g.lda = D.kdense_p;
  g.B = (const half_t*)(P + PL.w0d); g.ldb = D.kdense_p;
  g.M = B; g.N = D.n[0]; g.K = D.kdense_p; g.Mpad = WL.Mpad; g.Npad = D.np[0]; g.alpha = 1.f;
  g.bias = (const float*)(P + PL.bias[0]); g.act = ACT_LEAKY;
  g.C16 = (half_t*)(ws + WL.h[0]); g.ldc16 = D.np[0];
  if (training) { g.C16T = (half_t*)(ws + WL.hT[0]); g.ldc16t = WL.Mpad; }
  if (int e = launch_gemm_f16(ctx, g, st)) return e;
  return nacf_fwd_tail(ctx, d, D, PL, WL, P, ws, B, out, training, st);
}

// Backward through heads and trunk layers 4..1; leaves dz0 (+T) in ping-pong slot `*slot0` and
// the column sums (bias grads) in ws colsum[l].
static int nacf_bwd_body(neraf_ctx* ctx, const neraf_nacf_desc* d, const Dims& D, const PackLayout& PL, const WsLayout& WL,
                         const char* P, char* ws, int B, const float* out, const float* dout, float* const* grads,
                         int* slot0, hipStream_t st) {
  const int M = WL.Mpad;
  // the six column-sum rows and the scale block are adjacent in the workspace (make_ws_layout): one fill
  neraf_zero_async(st, ws + WL.colsum[0], WL.scale + 256 - WL.colsum[0]);
  // heads: dz5 = dout * (10 - out^2/10)  -> fp16 [Mpad][np5] and transposed [np5(+128)][Mpad]
  float* scale = (float*)(ws + WL.scale);
  const float* inv_scale = scale + 1;
  {
    const size_t n = (size_t)B * D.n[5];
    int blocks = (int)((n + 1023) / 1024);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(amax_dz5_kernel, dim3(blocks), dim3(256), 0, st, dout, out, n, reinterpret_cast<unsigned*>(scale) + 2);
    hipLaunchKernelGGL(make_scale_kernel, dim3(1), dim3(64), 0, st, scale);
  }
  // bias gradients = un-scaled column sums: collected here, copied out by ONE launch once the last sum is complete
  SegCopyTable outs{};
  outs.mul = inv_scale;
  const bool det = WL.col_slots > 1;          // deterministic mode: column sums in [Mpad / 32][np] slots, added in slot order on the way out
  outs.nslots = WL.col_slots;
  auto copy_out = [&](float* dst, const float* src, int n, int np) { seg_copy_add(outs, dst, src, n, n, np); };
  int cur = 0;
  half_t* dz = (half_t*)(ws + WL.dz[cur]);
  half_t* dzT = (half_t*)(ws + WL.dzT[cur]);
  if (int e = cvt_pad_transpose(ctx, st, dout, out, scale, D.n[5], B, D.n[5], 1, dz, D.np[5], M, D.np[5], dzT, M, M, D.np[5] + 128,
                                (float*)(ws + WL.colsum[5]), det ? D.np[5] : 0))
    return e;
  for (int c = 0; c < d->C; ++c) {
    // dWh_c [F, W] = dz5^T[cF:(c+1)F, :] . h4^T[W, :]^T
    GemmParams g{};
    g.A = dzT + (size_t)c * d->F * M; g.lda = M;
    g.B = (const half_t*)(ws + WL.hT[4]); g.ldb = M;
    g.M = d->F; g.N = D.n[4]; g.K = M; g.Mpad = round_up(d->F, 128); g.Npad = D.np[4]; g.alpha = 1.f;
    g.alpha_dev = inv_scale;
    g.C32 = grads[2 * (5 + c)]; g.ldc32 = D.n[4];
    if (int e = launch_gemm_f16(ctx, g, st)) return e;
    copy_out(grads[2 * (5 + c) + 1], (const float*)(ws + WL.colsum[5]) + (size_t)c * d->F, d->F, D.np[5]);
  }
  for (int l = 5; l >= 1; --l) {
    // dz_{l-1} = (dz_l . W_l) * leaky'(h_{l-1})   [Mpad, np_{l-1}]
    const int nxt = cur ^ 1;
    half_t* dzn = (half_t*)(ws + WL.dz[nxt]);
    half_t* dznT = (half_t*)(ws + WL.dzT[nxt]);
    GemmParams g{};
    g.A = (const half_t*)(ws + WL.dz[cur]); g.lda = D.np[l];
    g.B = (const half_t*)(P + PL.wt[l]); g.ldb = D.np[l];
    g.M = B; g.N = D.n[l - 1]; g.K = D.np[l]; g.Mpad = M; g.Npad = D.np[l - 1]; g.alpha = 1.f;
    g.lmask = (const half_t*)(ws + WL.h[l - 1]); g.ldmask = D.np[l - 1]; g.mask_slope = 0.1f;
    g.C16 = dzn; g.ldc16 = D.np[l - 1];
    g.C16T = dznT; g.ldc16t = M;
    g.colsum = (float*)(ws + WL.colsum[l - 1]);
    if (det) { g.stat_det = 1; g.stat_stride = D.np[l - 1]; }
    if (int e = launch_gemm_f16(ctx, g, st)) return e;
    cur = nxt;
    if (l - 1 >= 1) {
      // dW_{l-1} [n_{l-1}, k_{l-1}] = dz_{l-1}^T . h_{l-2}^T^T
      GemmParams w{};
      w.A = dznT; w.lda = M;
      w.B = (const half_t*)(ws + WL.hT[l - 2]); w.ldb = M;
      w.M = D.n[l - 1]; w.N = D.k[l - 1]; w.K = M; w.Mpad = D.np[l - 1]; w.Npad = D.kp[l - 1]; w.alpha = 1.f;
      w.alpha_dev = inv_scale;
      w.C32 = grads[2 * (l - 1)]; w.ldc32 = D.k[l - 1];
      if (int e = launch_gemm_f16(ctx, w, st)) return e;
      copy_out(grads[2 * (l - 1) + 1], (const float*)(ws + WL.colsum[l - 1]), D.n[l - 1], D.np[l - 1]);
    }
  }
  copy_out(grads[1], (const float*)(ws + WL.colsum[0]), D.n[0], D.np[0]);
  hipLaunchKernelGGL(seg_copy_kernel, dim3(outs.begin[outs.n]), dim3(256), 0, st, outs);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  *slot0 = cur;
  return NERAF_OK;
}

extern "C" int neraf_nacf_bwd(neraf_ctx* ctx, const neraf_nacf_desc* d, const void* packed, const float* const* w,
                              const float* feat, int B, const float* out, const float* dout, float* const* grads,
                              float* dfeat, void* workspace, neraf_stream_t stream) {
  if (int e = check_desc(ctx, d)) return e;
  if (B <= 0 || !out || !dout || !grads) return neraf_fail(ctx, NERAF_EINVAL, "nacf_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const Dims D = make_dims(d);
  const PackLayout PL = make_pack_layout(d, D);
  const WsLayout WL = make_ws_layout(d, D, B, 1);
  const char* P = (const char*)packed;
  char* ws = (char*)workspace;
  int s0 = 0;
  if (int e = nacf_bwd_body(ctx, d, D, PL, WL, P, ws, B, out, dout, grads, &s0, st)) return e;
  const int M = WL.Mpad;
  // dW0[:, n_feat:] = dz0^T . q^T^T   (N = 163, padded 256)
  GemmParams g{};
  g.A = (const half_t*)(ws + WL.dzT[s0]); g.lda = M;
  g.B = (const half_t*)(ws + WL.qT); g.ldb = M;
  g.M = D.n[0]; g.N = D.k[0]; g.K = M; g.Mpad = D.np[0]; g.Npad = QT_ROWS; g.alpha = 1.f;
  g.alpha_dev = (const float*)(ws + WL.scale) + 1;
  g.C32 = grads[0] + d->n_feat; g.ldc32 = D.kdense;
  if (int e = launch_gemm_f16(ctx, g, st)) return e;
  const float* db0 = grads[1];   // un-scaled layer-0 bias gradient
  if (d->n_feat > 0) {
    hipLaunchKernelGGL(outer_kernel, dim3((d->n_feat + 255) / 256, D.n[0]), dim3(256), 0, st, db0, feat, D.n[0], d->n_feat,
                       grads[0], D.kdense, dfeat);
    if (dfeat) {
      float* part = WL.col_slots > 1 ? (float*)(ws + WL.dfeat_part) : nullptr;
      hipLaunchKernelGGL(dfeat_kernel, dim3((d->n_feat + 255) / 256, 64), dim3(256), 0, st, Wptr(w, 0), D.kdense, db0,
                         D.n[0], d->n_feat, dfeat, part);
      if (part) hipLaunchKernelGGL(dfeat_fold_kernel, dim3((d->n_feat + 255) / 256), dim3(256), 0, st, part, 64, d->n_feat, dfeat);
    }
  }
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_nacf_bwd_dense(neraf_ctx* ctx, const neraf_nacf_desc* d, const void* packed, int B, const float* out,
                                    const float* dout, float* const* grads, float* dh, void* workspace,
                                    neraf_stream_t stream) {
  if (int e = check_desc(ctx, d)) return e;
  if (!d->dense_l0) return neraf_fail(ctx, NERAF_EINVAL, "nacf_bwd_dense: descriptor was created without dense_l0");
  if (B <= 0 || !out || !dout || !grads) return neraf_fail(ctx, NERAF_EINVAL, "nacf_bwd_dense: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const Dims D = make_dims(d);
  const PackLayout PL = make_pack_layout(d, D);
  const WsLayout WL = make_ws_layout(d, D, B, 1);
  const char* P = (const char*)packed;
  char* ws = (char*)workspace;
  int s0 = 0;
  if (int e = nacf_bwd_body(ctx, d, D, PL, WL, P, ws, B, out, dout, grads, &s0, st)) return e;
  const int M = WL.Mpad;
  GemmParams g{};
  g.A = (const half_t*)(ws + WL.dzT[s0]); g.lda = M;
  g.B = (const half_t*)(ws + WL.hdT); g.ldb = M;
  g.M = D.n[0]; g.N = D.kdense; g.K = M; g.Mpad = D.np[0]; g.Npad = D.kdense_p; g.alpha = 1.f;
  g.alpha_dev = (const float*)(ws + WL.scale) + 1;
  g.C32 = grads[0]; g.ldc32 = D.kdense;
  if (int e = launch_gemm_f16(ctx, g, st)) return e;
  if (dh) {
    GemmParams x{};
    x.A = (const half_t*)(ws + WL.dz[s0]); x.lda = D.np[0];
    x.B = (const half_t*)(P + PL.w0dt); x.ldb = D.np[0];
    x.M = B; x.N = D.kdense; x.K = D.np[0]; x.Mpad = M; x.Npad = D.kdense_p; x.alpha = 1.f;
    x.alpha_dev = (const float*)(ws + WL.scale) + 1;
    x.C32 = dh; x.ldc32 = D.kdense;
    if (int e = launch_gemm_f16(ctx, x, st)) return e;
  }
  return NERAF_OK;
}

extern "C" int neraf_stft_loss_sums(neraf_ctx* ctx, const float* pred, const float* gt, size_t n, int loss_type,
                                    float* sums, neraf_stream_t stream) {
  if (!pred || !gt || !sums || n == 0) return neraf_fail(ctx, NERAF_EINVAL, "stft_loss_sums: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  // sums[0..3] = the three sums (+ a zero); sums[4 .. 4 + 4 * 256) = per-workgroup partials: two launches, no atomics, no
  // clearing, and a fixed summation order (the loss and its gradient are bit-reproducible)
  int blocks = (int)((n + 1023) / 1024);
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(stft_loss_sums_kernel, dim3(blocks), dim3(256), 0, st, pred, gt, n, loss_type, sums + 4);
  hipLaunchKernelGGL(stft_sums_fold_kernel, dim3(1), dim3(64), 0, st, sums + 4, blocks, sums);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_stft_loss_finalize(neraf_ctx* ctx, const float* sums, size_t n_total, const float* weights, float* losses,
                                        neraf_stream_t stream) {
  if (!sums || !losses || n_total == 0) return neraf_fail(ctx, NERAF_EINVAL, "stft_loss_finalize: bad arguments");
  hipLaunchKernelGGL(stft_loss_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, 1.0f / (float)n_total, weights, losses);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_stft_loss_fwd(neraf_ctx* ctx, const float* pred, const float* gt, size_t n, int loss_type,
                                   float* sums, float* losses, neraf_stream_t stream) {
  if (int e = neraf_stft_loss_sums(ctx, pred, gt, n, loss_type, sums, stream)) return e;
  return neraf_stft_loss_finalize(ctx, sums, n, nullptr, losses, stream);
}

extern "C" int neraf_stft_loss_bwd(neraf_ctx* ctx, const float* pred, const float* gt, size_t n, size_t n_total,
                                   int loss_type, const float* sums, const float* g_sc, const float* g_mag, const float* weights,
                                   float extra, float* dpred, neraf_stream_t stream) {
  if (!pred || !gt || !sums || !dpred || n == 0 || n_total < n)
    return neraf_fail(ctx, NERAF_EINVAL, "stft_loss_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(stft_loss_bwd_kernel, dim3(blocks), dim3(256), 0, st, pred, gt, n, n_total, loss_type, sums, g_sc, g_mag, weights, extra, dpred);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

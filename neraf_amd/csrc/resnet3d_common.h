// Architecture walk and workspace layout of the ResNet3D scene encoder, shared by forward and backward.
#pragma once
#include "common.h"

namespace {

struct ConvSpec { int cin, cout, k, stride, pad, din, dout; };
struct BlockSpec { int conv[3]; int ds; int planes; };   // indices into the conv list (state-dict order); ds = -1 if none

struct Arch {
  int S;                       // input grid edge
  int nconv;
  ConvSpec conv[64];
  int nblock;
  BlockSpec block[16];
  int pooled;                  // edge after the stem max-pool
  int final_edge;              // edge of layer3's output
};

int make_arch(const neraf_resnet3d_desc* d, Arch* A) {
  if (!d || d->in_channels != 7 || d->n_features != 1024 || (d->grid_size != 128 && d->grid_size != 64)) return NERAF_EINVAL;
  A->S = d->grid_size;
  int n = 0;
  A->conv[n++] = ConvSpec{8, 64, 5, 2, 2, A->S, A->S / 2};             // stem: 7 (padded to 8) -> 64, NeRAF_resnet3d.py:120
  A->pooled = A->S / 4;                                                  // MaxPool3d(3, 2, 1), :123
  int edge = A->pooled, in_planes = 64, nb = 0;
  const int planes_l[3] = {64, 128, 256}, blocks_l[3] = {3, 4, 6}, stride_l[3] = {1, 2, 2};   // :124-126, resnet50 :237
  for (int li = 0; li < 3; ++li) {
    for (int b = 0; b < blocks_l[li]; ++b) {
      const int s = b == 0 ? stride_l[li] : 1, p = planes_l[li];
      BlockSpec B{};
      B.planes = p;
      B.conv[0] = n; A->conv[n++] = ConvSpec{in_planes, p, 1, 1, 0, edge, edge};             // :81
      B.conv[1] = n; A->conv[n++] = ConvSpec{p, p, 3, s, 1, edge, edge / s};                  // :83-84
      B.conv[2] = n; A->conv[n++] = ConvSpec{p, p * 4, 1, 1, 0, edge / s, edge / s};          // :86
      B.ds = -1;
      if (b == 0 && (s != 1 || in_planes != p * 4)) { B.ds = n; A->conv[n++] = ConvSpec{in_planes, p * 4, 1, s, 0, edge, edge / s}; }  // :169-174
      A->block[nb++] = B;
      in_planes = p * 4;
      edge /= s;
    }
  }
  A->nconv = n; A->nblock = nb; A->final_edge = edge;
  return NERAF_OK;
}

inline int conv_kpad(const ConvSpec& c) { return round_up(c.k * c.k * c.k * c.cin, 64); }
inline int conv_npad(const ConvSpec& c) { return c.cout == 64 ? 64 : round_up(c.cout, 128); }
__host__ __device__ inline size_t cube(int e) { return (size_t)e * e * e; }
inline size_t rows_pad(int e) { return round_up_sz(cube(e), 128); }

struct Layout {
  size_t w[64];                 // packed fp16 weights [npad][kpad]
  size_t packed_total;
  // workspace
  size_t zero_page, x0;         // zero page, NDHWC8 input
  size_t pre[64], stat[64];     // per conv: pre-BN output fp16 [rows_pad][cout], stats fp32 [2][cout]
  size_t act_pool;              // stem: pooled activation
  size_t a1[16], a2[16], out[16];   // per block post-activation tensors
  size_t splitk; size_t splitk_bytes;
  size_t stats_begin, stats_bytes;
  size_t total;
};

void make_layout(const Arch& A, Layout* L) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += round_up_sz(bytes, 256); return o; };
  for (int i = 0; i < A.nconv; ++i) L->w[i] = take((size_t)conv_npad(A.conv[i]) * conv_kpad(A.conv[i]) * 2);
  L->packed_total = off;
  off = 0;
  L->zero_page = take(256);
  L->x0 = take(cube(A.S) * 8 * 2);
  L->stats_begin = off;
  for (int i = 0; i < A.nconv; ++i) L->stat[i] = take((size_t)2 * round_up(A.conv[i].cout, 128) * 4);
  L->stats_bytes = off - L->stats_begin;
  for (int i = 0; i < A.nconv; ++i) L->pre[i] = take(rows_pad(A.conv[i].dout) * A.conv[i].cout * 2);
  L->act_pool = take(rows_pad(A.pooled) * 64 * 2);
  for (int b = 0; b < A.nblock; ++b) {
    const ConvSpec& c0 = A.conv[A.block[b].conv[0]]; const ConvSpec& c1 = A.conv[A.block[b].conv[1]];
    const ConvSpec& c2 = A.conv[A.block[b].conv[2]];
    L->a1[b] = take(rows_pad(c0.dout) * c0.cout * 2);
    L->a2[b] = take(rows_pad(c1.dout) * c1.cout * 2);
    L->out[b] = take(rows_pad(c2.dout) * c2.cout * 2);
  }
  L->splitk_bytes = (size_t)64 << 20;
  L->splitk = take(L->splitk_bytes);
  L->total = off;
}


struct BnSrc {
  const half_t* x;          // pre-BN conv output [rows][C]
  const float* stats;       // [2][Cpad]: sum, sum of squares (batch statistics) -- or null
  const float* gamma; const float* beta; const float* rmean; const float* rvar;
  int cpad;
};

__device__ __forceinline__ void bn_scale_shift(const BnSrc& s, int c, float inv_m, float& scale, float& shift) {
  float mean, var;
  if (s.stats) {
    mean = s.stats[c] * inv_m;
    var = fmaxf(s.stats[s.cpad + c] * inv_m - mean * mean, 0.f);    // biased variance, as nn.BatchNorm3d normalises with
  } else {
    mean = s.rmean[c]; var = s.rvar[c];
  }
  const float rstd = rsqrtf(var + 1e-5f);
  scale = s.gamma[c] * rstd;
  shift = s.beta[c] - mean * scale;
}


inline BnSrc bn_src(const half_t* x, const float* stats, const float* const* bn, int ci, int cout, int use_batch) {
  BnSrc s{};
  s.x = x; s.stats = use_batch ? stats : nullptr;
  s.gamma = bn[4 * ci + 0]; s.beta = bn[4 * ci + 1]; s.rmean = bn[4 * ci + 2]; s.rvar = bn[4 * ci + 3];
  s.cpad = round_up(cout, 128);
  return s;
}


}  // namespace

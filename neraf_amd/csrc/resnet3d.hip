// ResNet3D scene encoder (NeRAF_resnet3d.py:116-201; 'resnet50' truncated after layer3, N_features = 1024 -- or with layer4, 2048):
// voxel grid fp32 [7,S,S,S] -> N_features-vector.  Forward (train-mode batch statistics or eval-mode running
// statistics).  Activations are channels-last fp16 [D*H*W, C]; every Conv3d is the implicit-GEMM form of
// gemm_f16.hip (1x1x1 convs are plain GEMMs), whose epilogue also accumulates the per-channel sum and sum of
// squares that BatchNorm3d needs (batch = 1: statistics over the voxels, NeRAF_resnet3d.py:82-87 / SURVEY A4),
// so BN costs one element-wise pass that is fused with ReLU, the residual add and (stem) the 3^3 max-pool.
#include "resnet3d_common.h"
#include <algorithm>

namespace {

// ---- kernels ---------------------------------------------------------------------------------------------
// fp32 grid [7][S^3] (NeRAF_model.py:271-277) -> fp16 channels-last [S^3][8] (channel 7 = 0)
__global__ __launch_bounds__(256) void grid_to_ndhwc8_kernel(const float* __restrict__ grid, size_t nvox, half_t* __restrict__ out) {
  const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= nvox) return;
  half8 h;
#pragma unroll
  for (int c = 0; c < 7; ++c) h[c] = (half_t)grid[c * nvox + v];
  h[7] = (half_t)0.f;
  reinterpret_cast<half8*>(out)[v] = h;
}

// the same for the cells [*start, *start + n) only: the grid refresh rewrites one window of 4096 cells per step (NeRAF_model.py:395-404)
// and the converted image in the workspace is persistent; the window's first cell travels through device memory (graph replay)
__global__ __launch_bounds__(256) void grid_window_to_ndhwc8_kernel(const float* __restrict__ grid, size_t nvox, const unsigned long long* start,
                                                                   int n, half_t* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const size_t v = (size_t)*start + t;
  if (v >= nvox) return;
  half8 h;
#pragma unroll
  for (int c = 0; c < 7; ++c) h[c] = (half_t)grid[c * nvox + v];
  h[7] = (half_t)0.f;
  reinterpret_cast<half8*>(out)[v] = h;
}

__global__ void set_u64_fwd_kernel(unsigned long long* p, unsigned long long v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = v; }

struct RunTable {
  int n;
  int begin[kMaxConv + 1];                     // prefix of channel counts
  unsigned long long stat_off[kMaxConv];   // finalised mean / biased variance [2][cpad]
  int cpad[kMaxConv];
  float unbias[kMaxConv];
  float* rmean[kMaxConv]; float* rvar[kMaxConv];
  long long* nbt[kMaxConv];                // num_batches_tracked of each BatchNorm (or null): += 1 in the same launch
};

__global__ __launch_bounds__(256) void bn_update_running_all_kernel(RunTable t, const char* __restrict__ ws, float mom) {
  if (blockIdx.x == 0 && (int)threadIdx.x < t.n && t.nbt[threadIdx.x]) *t.nbt[threadIdx.x] += 1;    // BatchNorm3d.num_batches_tracked
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= t.begin[t.n]) return;
  int lo = 0, hi = t.n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t.begin[mid] <= idx) lo = mid; else hi = mid - 1; }
  const int i = lo, c = idx - t.begin[i];
  const float* fin = reinterpret_cast<const float*>(ws + t.stat_off[i]);
  const float mean = fin[c], var = fin[t.cpad[i] + c];
  t.rmean[i][c] = (1.f - mom) * t.rmean[i][c] + mom * mean;
  t.rvar[i][c] = (1.f - mom) * t.rvar[i][c] + mom * var * t.unbias[i];
}

// stem: out[32^3-like][64] = maxpool3(s2,p1)( relu(bn(x)) ), x pre-BN [din^3][64]
// arg (training): tap index (dz+1)*9 + (dy+1)*3 + (dx+1) of the FIRST maximum of each window, 255 when the maximum is not > 0 --
// the routing table of the backward (maxpool_bwd_gather_kernel)
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(BnSrc s, int din, int dout, size_t m_in, half_t* __restrict__ out,
                                                             unsigned char* __restrict__ arg) {
  __shared__ float sc[64], sh[64];
  if (threadIdx.x < 64) bn_scale_shift(s, threadIdx.x, 1.f / (float)m_in, sc[threadIdx.x], sh[threadIdx.x]);
  __syncthreads();
  const size_t total = cube(dout) * 8;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t vox = idx >> 3; const int c0 = (int)(idx & 7) * 8;
  const int x = (int)(vox % dout), y = (int)((vox / dout) % dout), z = (int)(vox / ((size_t)dout * dout));
  float best[8]; int at[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { best[j] = -3.0e38f; at[j] = 255; }
  for (int dz = -1; dz <= 1; ++dz) {
    const int iz = 2 * z + dz; if ((unsigned)iz >= (unsigned)din) continue;
    for (int dy = -1; dy <= 1; ++dy) {
      const int iy = 2 * y + dy; if ((unsigned)iy >= (unsigned)din) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int ix = 2 * x + dx; if ((unsigned)ix >= (unsigned)din) continue;
        const half8 v = *reinterpret_cast<const half8*>(s.x + (((size_t)iz * din + iy) * din + ix) * 64 + c0);
        const int tap = (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float a = fmaxf(fmaf((float)v[j], sc[c0 + j], sh[c0 + j]), 0.f);
          if (a > best[j]) { best[j] = a; at[j] = tap; }
        }
      }
    }
  }
  if (arg) {
    unsigned long long packed = 0ull;
#pragma unroll
    for (int j = 0; j < 8; ++j) packed |= (unsigned long long)(best[j] > 0.f ? at[j] : 255) << (8 * j);
    *reinterpret_cast<unsigned long long*>(arg + vox * 64 + c0) = packed;
  }
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (half_t)best[j];
  *reinterpret_cast<half8*>(out + vox * 64 + c0) = o;
}

// AvgPool3d over all remaining voxels (NeRAF_resnet3d.py:143/:149): feat[c] = mean_rows x[row][c].  One workgroup per 64 channels:
// 32 row groups x 8 channel chunks of threads take every 32nd row, a fixed tree over the groups finishes -- no atomics (the sum's
// order is fixed: the feature is bit-reproducible given the activations), no pre-zeroed output.
__global__ __launch_bounds__(256) void avgpool_kernel(const half_t* __restrict__ x, int M, int C, float* __restrict__ feat) {
  __shared__ float part[32][65];
  const int chunk = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + chunk * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = grp; r < M; r += 32) {
    const half8 v = *reinterpret_cast<const half8*>(x + (size_t)r * C + c0);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[grp][chunk * 8 + j] = s[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    float v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = part[k][threadIdx.x];
#pragma unroll
    for (int w = 1; w < 32; w <<= 1)
#pragma unroll
      for (int k = 0; k < 32; k += 2 * w) v[k] += v[k + w];
    feat[blockIdx.x * 64 + threadIdx.x] = v[0] * (1.f / (float)M);
  }
}

}  // namespace

// =================================================================================================
extern "C" int neraf_resnet3d_num_convs(const neraf_resnet3d_desc* d) {
  Arch A;
  if (make_arch(d, &A)) return -1;
  return A.nconv;
}

extern "C" size_t neraf_resnet3d_packed_bytes(const neraf_resnet3d_desc* d) {
  Arch A; Layout L;
  if (make_arch(d, &A)) return 0;
  make_layout(A, &L);
  return L.packed_total;
}

extern "C" size_t neraf_resnet3d_workspace_bytes(const neraf_resnet3d_desc* d) {
  Arch A; Layout L;
  if (make_arch(d, &A)) return 0;
  make_layout(A, &L);
  return L.total;
}

// algorithmic forward FLOPs of the encoder = sum over its convolutions of 2 dout^3 taps cin cout (real channels / taps;
// 94.72 GFLOP for the 7 x 128^3 grid).  Pure host arithmetic over the architecture table the launches are generated from.
extern "C" double neraf_resnet3d_forward_flops(const neraf_resnet3d_desc* d) {
  Arch A;
  if (make_arch(d, &A)) return -1.0;
  double f = 0.0;
  for (int i = 0; i < A.nconv; ++i) {
    const ConvSpec& c = A.conv[i];
    f += 2.0 * (double)cube(c.dout) * (c.k * c.k * c.k) * c.cin_real * c.cout;
  }
  return f;
}

// Test aid: byte offset / extent of a forward tensor inside the workspace (the gate-matched backward parity test reads the ReLU
// gates and the max-pool routing the forward actually used).
extern "C" int neraf_resnet3d_debug_locate(const neraf_resnet3d_desc* d, int kind, int index, size_t* offset, int* rows, int* cols) {
  Arch A; Layout L;
  if (make_arch(d, &A) || !offset || !rows || !cols) return NERAF_EINVAL;
  make_layout(A, &L);
  if (kind >= 0 && kind <= 2) {
    if (index < 0 || index >= A.nblock) return NERAF_EINVAL;
    const ConvSpec& c = A.conv[A.block[index].conv[kind]];
    *offset = kind == 0 ? L.a1[index] : kind == 1 ? L.a2[index] : L.out[index];
    *rows = (int)cube(c.dout); *cols = c.cout;
    return NERAF_OK;
  }
  if (kind == 7) { *offset = L.x0; *rows = (int)cube(A.S); *cols = 8; return NERAF_OK; }       // the converted input image [S^3][8] fp16
  if (kind == 3) { *offset = L.act_pool; *rows = (int)cube(A.pooled); *cols = 64; return NERAF_OK; }
  if (kind == 4) { *offset = L.pool_arg; *rows = (int)cube(A.pooled); *cols = 64; return NERAF_OK; }
  if (kind == 5 || kind == 6) {
    if (index < 0 || index >= A.nconv) return NERAF_EINVAL;
    const ConvSpec& c = A.conv[index];
    if (kind == 5) { *offset = L.pre[index]; *rows = (int)cube(c.dout); *cols = c.cout; }
    else { *offset = L.fin[index]; *rows = 2; *cols = round_up(c.cout, 128); }
    return NERAF_OK;
  }
  return NERAF_EINVAL;
}

extern "C" int neraf_resnet3d_pack_weights(neraf_ctx* ctx, const neraf_resnet3d_desc* d, const float* const* conv_w, void* packed,
                                           neraf_stream_t stream) {
  Arch A; Layout L;
  if (make_arch(d, &A) || !conv_w || !packed) return neraf_fail(ctx, NERAF_EINVAL, "resnet3d_pack_weights: bad arguments");
  make_layout(A, &L);
  hipStream_t st = (hipStream_t)stream;
  // un-padded convolutions go through the brick packer; the rest (the 7-channel stem) through the element-wise one
  PackTable t{};
  BrickTable bt{};
  unsigned long long acc = 0;
  int tiles = 0, max_run = 0;
  for (int i = 0; i < A.nconv; ++i) {
    const ConvSpec& c = A.conv[i];
    const int taps = c.k * c.k * c.k;
    if (brick_packable(c) && conv_kpad(c) == taps * c.cin && conv_npad(c) == c.cout) {
      const int j = bt.n++;
      int cob, cib;
      brick_shape(c, 0, &cob, &cib);
      bt.src[j] = conv_w[i]; bt.tile_begin[j] = tiles; bt.dst_off[j] = L.w[i];
      bt.cout[j] = c.cout; bt.cin[j] = c.cin; bt.taps[j] = taps; bt.cib[j] = cib; bt.cob[j] = cob;
      tiles += (c.cout / cob) * (c.cin / cib);
      max_run = std::max(max_run, cob * (cib * taps + 2));
      continue;
    }
    const int j = t.n++;
    t.src[j] = conv_w[i]; t.begin[j] = acc; t.dst_off[j] = L.w[i];
    t.cout[j] = c.cout; t.cin_real[j] = c.cin_real; t.cin[j] = c.cin; t.taps[j] = taps;
    t.kpad[j] = conv_kpad(c);
    acc += (unsigned long long)conv_npad(c) * conv_kpad(c);
  }
  t.begin[t.n] = acc;
  bt.tile_begin[bt.n] = tiles;
  if (t.n > 0) hipLaunchKernelGGL(pack_all_conv_weights_kernel, dim3((unsigned)((acc + 255) / 256)), dim3(256), 0, st, t, (char*)packed);
  if (bt.n > 0) {
    const size_t lds = brick_lds_bytes(max_run);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
      NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_bricks_kernel<0>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_lds = lds;
    }
    hipLaunchKernelGGL(pack_bricks_kernel<0>, dim3((unsigned)tiles), dim3(256), lds, st, bt, (char*)packed);
  }
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

static int resnet3d_fwd_body(neraf_ctx* ctx, const Arch& A, const Layout& L, const char* packed, const float* const* bn, const float* grid,
                             char* ws, float* feat, int use_batch_stats, int win_cells, hipStream_t st) {
  neraf_zero3_async(st, ws + L.zero_page, 256, ws + L.stats_begin, L.stats_bytes, feat, (size_t)A.n_features * sizeof(float));
  neraf_node(ctx, "neraf_zero3_kernel | statistics accumulators", 0.0, 0.0, 256.0 + (double)L.stats_bytes + 4096.0);
  const size_t nvox = cube(A.S);
  const bool training = use_batch_stats != 0;    // training forward: keep the max-pool routing for the backward
  if (win_cells > 0) {    // the rest of the image is the previous call's (the caller vouches for it)
    hipLaunchKernelGGL(grid_window_to_ndhwc8_kernel, dim3((unsigned)((win_cells + 255) / 256)), dim3(256), 0, st, grid, nvox,
                       reinterpret_cast<const unsigned long long*>(ws + L.win), win_cells, (half_t*)(ws + L.x0));
    neraf_node(ctx, "grid_window_to_ndhwc8_kernel | refresh window -> fp16 channels-last", 0.0, win_cells * 28.0, win_cells * 16.0);
  } else {
    hipLaunchKernelGGL(grid_to_ndhwc8_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, st, grid, nvox, (half_t*)(ws + L.x0));
    neraf_node(ctx, "grid_to_ndhwc8_kernel | grid -> fp16 channels-last", 0.0, (double)nvox * 28.0, (double)nvox * 16.0);
  }
  // stem: conv1 -> bn1 -> relu -> maxpool (NeRAF_resnet3d.py:185-188)
  if (int e = run_conv(ctx, st, A, L, 0, packed, ws, (const half_t*)(ws + L.x0))) return e;
  {
    const ConvSpec& c = A.conv[0];
    BnSrc s = bn_src_fwd(A, L, ws, bn, 0, use_batch_stats);
    const size_t total = cube(A.pooled) * 8;
    hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, s, c.dout, A.pooled,
                       cube(c.dout), (half_t*)(ws + L.act_pool), training ? (unsigned char*)(ws + L.pool_arg) : nullptr);
    NERAF_HIP_CHECK(ctx, hipGetLastError());
    neraf_node(ctx, "bn_relu_maxpool_kernel | stem BatchNorm + ReLU + 3x3x3/2 max-pool", 0.0, (double)cube(c.dout) * 64 * 2.0,
               (double)cube(A.pooled) * 64 * (training ? 3.0 : 2.0));
  }
  const half_t* x = (const half_t*)(ws + L.act_pool);
  for (int b = 0; b < A.nblock; ++b) {                                        // Bottleneck.forward, :92-113
    const BlockSpec& B = A.block[b];
    const int i0 = B.conv[0], i1 = B.conv[1], i2 = B.conv[2];
    const ConvSpec &c0 = A.conv[i0], &c1 = A.conv[i1], &c2 = A.conv[i2];
    if (int e = run_conv(ctx, st, A, L, i0, packed, ws, x)) return e;
    BnApplyArgs a{};
    a.a = bn_src_fwd(A, L, ws, bn, i0, use_batch_stats);
    a.M = (int)cube(c0.dout); a.Mpad = (int)rows_pad(c0.dout); a.C = c0.cout; a.relu = 1; a.out = (half_t*)(ws + L.a1[b]);
    if (int e = run_bn_apply(ctx, st, a)) return e;
    if (int e = run_conv(ctx, st, A, L, i1, packed, ws, (const half_t*)(ws + L.a1[b]))) return e;
    BnApplyArgs a2{};
    a2.a = bn_src_fwd(A, L, ws, bn, i1, use_batch_stats);
    a2.M = (int)cube(c1.dout); a2.Mpad = (int)rows_pad(c1.dout); a2.C = c1.cout; a2.relu = 1; a2.out = (half_t*)(ws + L.a2[b]);
    // MEASUREMENT ONLY (NERAF_SKIP_SMALL_BN=1, results are garbage): upper bound of what fusing the BatchNorm apply of a 1x1x1
    // consumer into that consumer's operand load could save -- the launch is simply dropped for the <= 4096-voxel layers
    static const int skip_small = [] { const char* e = getenv("NERAF_SKIP_SMALL_BN"); return e ? atoi(e) : 0; }();
    if (!(skip_small && cube(c1.dout) <= 4096))
    if (int e = run_bn_apply(ctx, st, a2)) return e;
    if (int e = run_conv(ctx, st, A, L, i2, packed, ws, (const half_t*)(ws + L.a2[b]))) return e;
    BnApplyArgs a3{};
    a3.a = bn_src_fwd(A, L, ws, bn, i2, use_batch_stats);
    if (B.ds >= 0) {
      if (int e = run_conv(ctx, st, A, L, B.ds, packed, ws, x)) return e;
      a3.r = bn_src_fwd(A, L, ws, bn, B.ds, use_batch_stats);
    } else {
      a3.res = x;
    }
    a3.M = (int)cube(c2.dout); a3.Mpad = (int)rows_pad(c2.dout); a3.C = c2.cout; a3.relu = 1; a3.out = (half_t*)(ws + L.out[b]);
    if (int e = run_bn_apply(ctx, st, a3)) return e;
    x = (const half_t*)(ws + L.out[b]);
  }
  {
    const int M = (int)cube(A.final_edge), C = A.n_features;
    hipLaunchKernelGGL(avgpool_kernel, dim3(C / 64), dim3(256), 0, st, x, M, C, feat);
    NERAF_HIP_CHECK(ctx, hipGetLastError());
    neraf_node(ctx, "avgpool_kernel | mean over the voxels", 0.0, (double)M * C * 2.0, C * 4.0);
  }
  return NERAF_OK;
}

extern "C" int neraf_resnet3d_fwd(neraf_ctx* ctx, const neraf_resnet3d_desc* d, const void* packed_, const float* const* bn,
                                  const float* grid, void* workspace, float* feat, int use_batch_stats, size_t win_start, int win_cells,
                                  neraf_stream_t stream) {
  Arch A; Layout L;
  if (make_arch(d, &A) || !packed_ || !bn || !grid || !workspace || !feat || win_cells < 0 ||
      (win_cells > 0 && win_start + (size_t)win_cells > cube(d->grid_size)))
    return neraf_fail(ctx, NERAF_EINVAL, "resnet3d_fwd: bad arguments (grid_size 64|128|256, in_channels 7, n_features 1024|2048, window inside the grid)");
  make_layout(A, &L);
  if (win_cells > 0)      // the window moves every step: its first cell travels through device memory, outside the captured sequence
    hipLaunchKernelGGL(set_u64_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned long long*>((char*)workspace + L.win), (unsigned long long)win_start);
  // ~107 launches of mostly 3-10 us kernels with arguments fixed by these values: replayed as one hipGraph
  ArgHash k;
  k.add(0x66776431u); k.add(d->grid_size); k.add(packed_); k.ptrs((const void* const*)bn, 4 * A.nconv); k.add(grid); k.add(workspace);
  k.add(feat); k.add(use_batch_stats); k.add(win_cells);
  return neraf_run_graphed(ctx, (hipStream_t)stream, k.h, [&](hipStream_t st) {
    return resnet3d_fwd_body(ctx, A, L, (const char*)packed_, bn, grid, (char*)workspace, feat, use_batch_stats, win_cells, st);
  });
}

extern "C" int neraf_resnet3d_update_running_stats(neraf_ctx* ctx, const neraf_resnet3d_desc* d, const void* workspace,
                                                   float* const* bn, float momentum, int64_t* const* num_batches_tracked,
                                                   neraf_stream_t stream) {
  Arch A; Layout L;
  if (make_arch(d, &A) || !workspace || !bn) return neraf_fail(ctx, NERAF_EINVAL, "resnet3d_update_running_stats: bad arguments");
  make_layout(A, &L);
  const char* ws = (const char*)workspace;
  RunTable t{};
  t.n = A.nconv;
  int acc = 0;
  for (int i = 0; i < A.nconv; ++i) {
    const ConvSpec& c = A.conv[i];
    const float m = (float)cube(c.dout);
    t.begin[i] = acc; t.stat_off[i] = L.fin[i]; t.cpad[i] = round_up(c.cout, 128);
    t.unbias[i] = m / (m - 1.f);
    t.rmean[i] = bn[4 * i + 2]; t.rvar[i] = bn[4 * i + 3];
    t.nbt[i] = num_batches_tracked ? (long long*)num_batches_tracked[i] : nullptr;
    acc += c.cout;
  }
  t.begin[A.nconv] = acc;
  hipLaunchKernelGGL(bn_update_running_all_kernel, dim3((acc + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, ws, momentum);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

// Architecture walk and workspace layout of the ResNet3D scene encoder, shared by forward and backward.
#pragma once
#include "common.h"

namespace {

constexpr int kMaxConv = 56;      // 43 conv / BatchNorm pairs at N_features = 1024, 53 with layer4 (tables in kernel arguments)
struct ConvSpec { int cin, cout, k, stride, pad, din, dout; int cin_real; };   // cin_real < cin only for the stem (7 of 8)
struct BlockSpec { int conv[3]; int ds; int planes; };   // indices into the conv list (state-dict order); ds = -1 if none

struct Arch {
  int S;                       // input grid edge
  int nconv;
  ConvSpec conv[64];
  int nblock;
  BlockSpec block[16];
  int pooled;                  // edge after the stem max-pool
  int final_edge;              // edge of the last layer's output (layer3; layer4 when N_features = 2048)
  int n_features;              // 1024 (layers 1-3) | 2048 (+ layer4), NeRAF_resnet3d.py:128-131
};

int make_arch(const neraf_resnet3d_desc* d, Arch* A) {
  // N_features 1024 | 2048 (NeRAF_resnet3d.py:128), grid 64^3 | 128^3 | 256^3 (grid_step 1/64 | 1/128 | 1/256, :138-156: the average pool
  // spans the last layer's whole output in every combination)
  if (!d || d->in_channels != 7 || (d->n_features != 1024 && d->n_features != 2048) ||
      (d->grid_size != 128 && d->grid_size != 64 && d->grid_size != 256)) return NERAF_EINVAL;
  A->S = d->grid_size;
  A->n_features = d->n_features;
  int n = 0;
  A->conv[n++] = ConvSpec{8, 64, 5, 2, 2, A->S, A->S / 2, 7};             // stem: 7 (padded to 8) -> 64, NeRAF_resnet3d.py:120
  A->pooled = A->S / 4;                                                  // MaxPool3d(3, 2, 1), :123
  int edge = A->pooled, in_planes = 64, nb = 0;
  const int planes_l[4] = {64, 128, 256, 512}, blocks_l[4] = {3, 4, 6, 3}, stride_l[4] = {1, 2, 2, 2};   // :124-126, :131, resnet50 :237
  const int n_layers = d->n_features == 2048 ? 4 : 3;
  for (int li = 0; li < n_layers; ++li) {
    for (int b = 0; b < blocks_l[li]; ++b) {
      const int s = b == 0 ? stride_l[li] : 1, p = planes_l[li];
      BlockSpec B{};
      B.planes = p;
      B.conv[0] = n; A->conv[n++] = ConvSpec{in_planes, p, 1, 1, 0, edge, edge, in_planes};             // :81
      B.conv[1] = n; A->conv[n++] = ConvSpec{p, p, 3, s, 1, edge, edge / s, p};                  // :83-84
      B.conv[2] = n; A->conv[n++] = ConvSpec{p, p * 4, 1, 1, 0, edge / s, edge / s, p};          // :86
      B.ds = -1;
      if (b == 0 && (s != 1 || in_planes != p * 4)) { B.ds = n; A->conv[n++] = ConvSpec{in_planes, p * 4, 1, s, 0, edge, edge / s, in_planes}; }  // :169-174
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

// BatchNorm batch statistics are accumulated by the conv GEMM's epilogue with one atomic per column per workgroup.  Atomics
// on one cache line serialise in the memory system (measured: the 32^3-voxel layers spent 60 of their 70 us there), so the
// accumulators are replicated: m-tile t adds into replica t % rep, replicas 4 KiB apart; consumers sum the replicas.
constexpr int kStatStride = 1024;          // floats between replicas (>= 2 * cpad for every layer)
inline int stat_rep(const ConvSpec& c) {
  const int tiles = (int)(rows_pad(c.dout) / 128);
  if (2 * round_up(c.cout, 128) > kStatStride) return 1;
  int r = 1;
  while (r < 16 && r * 8 < tiles) r <<= 1;
  return r;
}

inline size_t det_slots(const ConvSpec& c) { return rows_pad(c.dout) / 32; }

struct Layout {
  size_t w[64];                 // packed fp16 weights [npad][kpad]
  size_t packed_total;
  // workspace
  size_t zero_page, x0, win;    // zero page, NDHWC8 input, device scalar: first cell of the window to re-convert
  size_t pre[64], stat[64];     // per conv: pre-BN output fp16 [rows_pad][cout], stats fp32 [rep][2][cpad] (replica stride kStatStride)
  size_t fin[64];               // per conv: finalised batch statistics fp32 [2][cpad] = mean, biased variance (written by the BN pass)
  size_t act_pool;              // stem: pooled activation
  size_t pool_arg;              // stem: which of the 27 window taps held the maximum (uint8 [pooled^3][64], 255 = none > 0); training
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
  L->win = take(256);
  L->x0 = take(cube(A.S) * 8 * 2);
  L->stats_begin = off;
  for (int i = 0; i < A.nconv; ++i) {
    const int rep = stat_rep(A.conv[i]);
    if (neraf_deterministic())       // one slot [2][cpad] per 32 result rows (GemmParams::stat_det), summed in a fixed order by slot_sum_kernel
      L->stat[i] = take(det_slots(A.conv[i]) * 2 * round_up(A.conv[i].cout, 128) * 4);
    else
      L->stat[i] = take(rep > 1 ? (size_t)rep * kStatStride * 4 : (size_t)2 * round_up(A.conv[i].cout, 128) * 4);
  }
  L->stats_bytes = off - L->stats_begin;
  for (int i = 0; i < A.nconv; ++i) L->fin[i] = take((size_t)2 * round_up(A.conv[i].cout, 128) * 4);
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
  L->pool_arg = take(cube(A.pooled) * 64);
  L->total = off;
}


struct BnSrc {
  const half_t* x;          // pre-BN conv output [rows][C]
  const float* stats;       // [rep][2][Cpad]: sum, sum of squares (batch statistics, replicated accumulators) -- or null
  const float* fin_r;       // finalised [2][Cpad] mean / biased variance (backward and running-stat update read these) -- or null
  float* fin_w;             // forward BN pass: block 0 writes the finalised statistics here -- or null
  const float* gamma; const float* beta; const float* rmean; const float* rvar;
  int cpad, rep;
  int read_mode;            // how atomically produced accumulators are read (stat_ld): 2 system-scope loads (default), 1 atomic fetch-add of 0, 0 plain
};

// Reading one statistic accumulator.  The accumulators are produced by no-return fp32 atomics (GEMM epilogues, bn_bwd_reduce), and
// on MI355X float atomics execute at the MEMORY SIDE: they leave the issuing XCD's L2 as uncached requests and nothing stays in any
// L2 (MI355X_MICROARCH.md, "Global float atomics").  The consumer therefore reads them at the same point: mode 2 (the default) is a
// system-scope load (sc0 sc1: served past the L1 and the L2), mode 1 an atomic fetch-add of zero (a returning atomic at the memory
// side: exact by construction, but thousands of workgroups re-reading the same lines serialise there), mode 0 a plain load that
// relies on the kernel boundary alone.  Round 3 shipped mode 0 plus a never-taken re-read that merely changed code generation and
// happened to survive a second process on the GPU (profiles/r03_gpu_sharing_bisect.txt: single workgroups of bn_apply normalised
// with wrong tables, cause not understood); modes 1 and 2 measured 0 damaged forwards in the same bisect and do not depend on code
// placement.  NERAF_BN_STAT_READ selects the mode for A/B (tools/share_gpu_regression.sh re-checks the shared-GPU scenario).
__device__ __forceinline__ float stat_ld(const float* p, int mode) {
  if (mode == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (mode == 1) return atomicAdd(const_cast<float*>(p), 0.f);
  return *p;
}

inline int bn_stat_read_mode() {
  static const int m = [] { const char* e = getenv("NERAF_BN_STAT_READ"); const int v = e ? atoi(e) : 2; return (v < 0 || v > 2) ? 2 : v; }();
  return m;
}

__device__ __forceinline__ void bn_mean_var(const BnSrc& s, int c, float inv_m, float& mean, float& var) {
  if (s.fin_r) {
    mean = s.fin_r[c]; var = s.fin_r[s.cpad + c];
  } else if (s.stats) {
    // four independent chains: the loop is a latency chain of 2 * rep loads otherwise (rep is 1, 2, 4, 8 or 16)
    float a = 0.f, b = 0.f, a1 = 0.f, b1 = 0.f, a2 = 0.f, b2 = 0.f, a3 = 0.f, b3 = 0.f;
    const int md = s.read_mode;
    const float* q = s.stats + c;
    int r = 0;
    for (; r + 4 <= s.rep; r += 4) {
      a += stat_ld(q + r * kStatStride, md); b += stat_ld(q + r * kStatStride + s.cpad, md);
      a1 += stat_ld(q + (r + 1) * kStatStride, md); b1 += stat_ld(q + (r + 1) * kStatStride + s.cpad, md);
      a2 += stat_ld(q + (r + 2) * kStatStride, md); b2 += stat_ld(q + (r + 2) * kStatStride + s.cpad, md);
      a3 += stat_ld(q + (r + 3) * kStatStride, md); b3 += stat_ld(q + (r + 3) * kStatStride + s.cpad, md);
    }
    for (; r < s.rep; ++r) { a += stat_ld(q + r * kStatStride, md); b += stat_ld(q + r * kStatStride + s.cpad, md); }
    a = (a + a1) + (a2 + a3); b = (b + b1) + (b2 + b3);
    mean = a * inv_m;
    var = fmaxf(b * inv_m - mean * mean, 0.f);    // biased variance, as nn.BatchNorm3d normalises with
  } else {
    mean = s.rmean[c]; var = s.rvar[c];
  }
}

__device__ __forceinline__ void bn_scale_shift(const BnSrc& s, int c, float inv_m, float& scale, float& shift) {
  float mean, var;
  bn_mean_var(s, c, inv_m, mean, var);
  if (s.fin_w && blockIdx.x == 0) { s.fin_w[c] = mean; s.fin_w[s.cpad + c] = var; }
  const float rstd = rsqrtf(var + 1e-5f);
  scale = s.gamma[c] * rstd;
  shift = s.beta[c] - mean * scale;
}


// Deterministic mode: dst[which][c] = sum over the nslots slots of src[slot][which][c] in a FIXED order (16 interleaved partial sums
// per column, then a fixed tree), which = 0 / 1 the two statistics; to_meanvar: dst = {mean, biased variance} of a BatchNorm over
// 1 / inv_m rows instead of the raw sums.  grid = cpad / 64 column panels, 1024 threads.
__global__ __launch_bounds__(1024) void slot_sum_kernel(const float* __restrict__ src, int nslots, int stride, int cpad, float inv_m,
                                                       int to_meanvar, float* __restrict__ dst) {
  __shared__ float part[2][16][64];
  const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + col;
  float a = 0.f, b = 0.f;
  for (int sl = q; sl < nslots; sl += 16) { a += src[(size_t)sl * stride + c]; b += src[(size_t)sl * stride + cpad + c]; }
  part[0][q][col] = a; part[1][q][col] = b;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = part[which][k][col];
#pragma unroll
    for (int w = 1; w < 16; w <<= 1)
#pragma unroll
      for (int k = 0; k < 16; k += 2 * w) v[k] += v[k + w];
    part[which][0][col] = v[0];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const float s0 = part[0][0][col], s1 = part[1][0][col];
    if (to_meanvar) {
      const float mean = s0 * inv_m;
      dst[c] = mean; dst[cpad + c] = fmaxf(s1 * inv_m - mean * mean, 0.f);
    } else {
      dst[c] = s0; dst[cpad + c] = s1;
    }
  }
}

inline void run_slot_sum(hipStream_t st, const float* src, size_t nslots, int cpad, float inv_m, int to_meanvar, float* dst) {
  hipLaunchKernelGGL(slot_sum_kernel, dim3(cpad / 64), dim3(1024), 0, st, src, (int)nslots, 2 * cpad, cpad, inv_m, to_meanvar, dst);
}

// forward BN pass: reads the replicated accumulators of conv ci, block 0 publishes mean / variance to L.fin[ci]
// (deterministic mode: the statistics were finalised into L.fin[ci] by run_conv's slot_sum launch; every pass reads them there)
inline BnSrc bn_src_fwd(const Arch& A, const Layout& L, char* ws, const float* const* bn, int ci, int use_batch) {
  const ConvSpec& c = A.conv[ci];
  BnSrc s{};
  s.read_mode = bn_stat_read_mode();
  s.x = (const half_t*)(ws + L.pre[ci]);
  s.stats = use_batch ? (const float*)(ws + L.stat[ci]) : nullptr;
  s.fin_w = use_batch ? (float*)(ws + L.fin[ci]) : nullptr;
  if (use_batch && neraf_deterministic()) { s.stats = nullptr; s.fin_w = nullptr; s.fin_r = (const float*)(ws + L.fin[ci]); }
  s.gamma = bn[4 * ci + 0]; s.beta = bn[4 * ci + 1]; s.rmean = bn[4 * ci + 2]; s.rvar = bn[4 * ci + 3];
  s.cpad = round_up(c.cout, 128); s.rep = stat_rep(c);
  return s;
}

// backward: train-mode statistics as the forward BN pass published them
inline BnSrc bn_src_bwd(const Arch& A, const Layout& L, const char* ws, const float* const* bn, int ci) {
  const ConvSpec& c = A.conv[ci];
  BnSrc s{};
  s.x = (const half_t*)(ws + L.pre[ci]);
  s.fin_r = (const float*)(ws + L.fin[ci]);
  s.gamma = bn[4 * ci + 0]; s.beta = bn[4 * ci + 1]; s.rmean = bn[4 * ci + 2]; s.rvar = bn[4 * ci + 3];
  s.cpad = round_up(c.cout, 128); s.rep = 1;
  s.read_mode = bn_stat_read_mode();
  return s;
}


// one launch for all convolutions (43 | 53): table of per-conv descriptors in the kernel arguments
struct PackTable {
  int n;
  const float* src[kMaxConv];
  unsigned long long begin[kMaxConv + 1];      // prefix of packed element counts
  unsigned long long dst_off[kMaxConv];    // byte offset of the conv's packed matrix
  int cout[kMaxConv], cin_real[kMaxConv], cin[kMaxConv], taps[kMaxConv], kpad[kMaxConv];
};

__global__ __launch_bounds__(256) void pack_all_conv_weights_kernel(PackTable t, char* __restrict__ packed) {
  const unsigned long long idx = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= t.begin[t.n]) return;
  int lo = 0, hi = t.n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t.begin[mid] <= idx) lo = mid; else hi = mid - 1; }
  const int i = lo;
  const unsigned long long e = idx - t.begin[i];
  const int kpad = t.kpad[i], cin = t.cin[i];
  const int n = (int)(e / kpad), k = (int)(e % kpad);
  const int tap = k / cin, c = k % cin;
  float v = 0.f;
  if (n < t.cout[i] && tap < t.taps[i] && c < t.cin_real[i]) v = t.src[i][((size_t)n * t.cin_real[i] + c) * t.taps[i] + tap];
  reinterpret_cast<half_t*>(packed + t.dst_off[i])[e] = (half_t)v;
}

// Brick packer for the un-padded convolutions (cin, cout multiples of 64): a workgroup moves a [COB co][CIB ci][taps] brick through
// LDS.  The fp32 source [cout][cin][taps] is read in contiguous runs of CIB*taps floats per co row (16-byte loads), converted and
// parked in LDS in source order; the brick shape is chosen PER LAYOUT so that the destination is written in whole lines too:
//   MODE 0  forward layout   fp16  dst[co][tap*cin + ci]     CIB = cin: a brick is COB complete destination rows (contiguous)
//   MODE 1  dgrad layout     fp16  dst[ci][tap*cout + co]    COB = cout: every (ci, tap) run is a complete row segment of cout
// (the first version used [32 co][32 ci] bricks for both: 64-byte destination runs, 2 TB/s).  COB / CIB come from brick_shape().
struct BrickTable {
  int n;
  const float* src[kMaxConv];
  int tile_begin[kMaxConv + 1];                // prefix of (cout/COB)*(cin/CIB) bricks
  unsigned long long dst_off[kMaxConv];
  int cout[kMaxConv], cin[kMaxConv], taps[kMaxConv], cib[kMaxConv], cob[kMaxConv];
};

template <int MODE>
__global__ __launch_bounds__(256) void pack_bricks_kernel(BrickTable t, char* __restrict__ packed) {
  extern __shared__ unsigned short brick16[];          // [COB co][pitch]
  int lo = 0, hi = t.n - 1;
  const int bid = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t.tile_begin[mid] <= bid) lo = mid; else hi = mid - 1; }
  const int i = lo;
  const int tile = bid - t.tile_begin[i];
  const int taps = t.taps[i], cin = t.cin[i], cout = t.cout[i], cib = t.cib[i], cob = t.cob[i];
  const int ci_tiles = cin / cib;
  const int co0 = (tile / ci_tiles) * cob, ci0 = (tile % ci_tiles) * cib;
  const int run = cib * taps;                          // multiple of 4
  const int pitch = ((run >> 1) & 1) ? run : run + 2;  // odd number of dwords per row: column reads (MODE 1) hit distinct banks
  const float* src = t.src[i] + ((size_t)co0 * cin + ci0) * taps;
  {
    // 16-byte loads over the flat list of (co row, chunk) pairs, two in flight per thread (run and every row offset are multiples
    // of 4 floats); short runs (1x1x1 filters in MODE 1: 16 floats per co row) keep every lane busy this way
    const int run4 = run >> 2, total4 = cob * run4;
    const size_t row_stride = (size_t)cin * taps;
    auto put4 = [&](int e, const f32x4& v) {
      const int co_l = e / run4, i4 = e - co_l * run4;
      unsigned short* dst = brick16 + co_l * pitch + 4 * i4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const half_t h = (half_t)v[k];
        dst[k] = *reinterpret_cast<const unsigned short*>(&h);
      }
    };
    auto get4 = [&](int e) {
      const int co_l = e / run4, i4 = e - co_l * run4;
      return reinterpret_cast<const f32x4*>(src + (size_t)co_l * row_stride)[i4];
    };
    int e = threadIdx.x;
    for (; e + 768 < total4; e += 1024) {      // four loads in flight per lane: the kernel is bound by bytes in flight per CU
      const f32x4 v0 = get4(e), v1 = get4(e + 256), v2 = get4(e + 512), v3 = get4(e + 768);
      put4(e, v0); put4(e + 256, v1); put4(e + 512, v2); put4(e + 768, v3);
    }
    for (; e < total4; e += 256) put4(e, get4(e));
  }
  __syncthreads();
  // destination: 8 elements (16 bytes) per lane along the layout's fastest index
  if (MODE == 0) {
    // the brick is cob complete rows [tap][ci] of the destination (cib == cin): lanes run over (tap, ci / 8)
    uint4* out = reinterpret_cast<uint4*>(packed + t.dst_off[i]) + (((size_t)co0 * taps * cin) >> 3);
    const int c8 = cin >> 3, per_row = taps * c8, total = cob * per_row;
    for (int e = threadIdx.x; e < total; e += 256) {
      const int co_l = e / per_row, r = e - co_l * per_row, tap = r / c8, ci = (r - tap * c8) * 8;
      const unsigned short* sp = brick16 + co_l * pitch + ci * taps + tap;
      uint4 v;
      v.x = (unsigned)sp[0] | ((unsigned)sp[taps] << 16);
      v.y = (unsigned)sp[2 * taps] | ((unsigned)sp[3 * taps] << 16);
      v.z = (unsigned)sp[4 * taps] | ((unsigned)sp[5 * taps] << 16);
      v.w = (unsigned)sp[6 * taps] | ((unsigned)sp[7 * taps] << 16);
      out[e] = v;
    }
  } else {
    // runs of cout co (cob == cout) for every (ci, tap) of the brick: lanes run over (ci, tap, co / 8)
    uint4* out = reinterpret_cast<uint4*>(packed + t.dst_off[i]);
    const int c8 = cout >> 3, per_ci = taps * c8, total = cib * per_ci;
    for (int e = threadIdx.x; e < total; e += 256) {
      const int ci_l = e / per_ci, r = e - ci_l * per_ci, tap = r / c8, co = (r - tap * c8) * 8;
      const unsigned short* sp = brick16 + co * pitch + ci_l * taps + tap;
      uint4 v;
      v.x = (unsigned)sp[0] | ((unsigned)sp[pitch] << 16);
      v.y = (unsigned)sp[2 * pitch] | ((unsigned)sp[3 * pitch] << 16);
      v.z = (unsigned)sp[4 * pitch] | ((unsigned)sp[5 * pitch] << 16);
      v.w = (unsigned)sp[6 * pitch] | ((unsigned)sp[7 * pitch] << 16);
      out[(((size_t)(ci0 + ci_l) * taps * cout) >> 3) + r] = v;
    }
  }
}

// conv qualifies for the brick packer: no padding anywhere in either packed layout
inline bool brick_packable(const ConvSpec& c) {
  return c.cin == c.cin_real && (c.cin % 64) == 0 && (c.cout % 64) == 0 && (c.k == 1 || c.k == 3);
}
// brick [cob co][cib ci][taps] of at most ~28 KiB of 16-bit elements (four or five workgroups per CU: with two, 2.7 TB/s)
inline void brick_shape(const ConvSpec& c, int mode, int* cob, int* cib) {
  const int taps = c.k * c.k * c.k, budget = 14336;
  if (mode == 0) {
    *cib = c.cin;
    int b = budget / (c.cin * taps);
    b = b >= 32 ? 32 : (b >= 16 ? 16 : (b >= 8 ? 8 : (b >= 4 ? 4 : (b >= 2 ? 2 : 1))));
    while (c.cout % b) b >>= 1;
    *cob = b;
  } else {
    *cob = c.cout;
    int b = budget / (c.cout * taps);
    b = b >= 32 ? 32 : (b >= 16 ? 16 : (b >= 8 ? 8 : (b >= 4 ? 4 : (b >= 2 ? 2 : 1))));
    while (c.cin % b) b >>= 1;
    if (b < 4) b = 4;                 // source runs of whole 16-byte chunks (taps is odd for 3x3x3)
    *cib = b;
  }
}
inline size_t brick_lds_bytes(int max_elems) { return (size_t)max_elems * 2; }

// out = [relu]( bn(x) [+ residual | + bn_r(xr)] ) ; 8 channels (16 B) per thread; rows >= M are written as zeros
struct BnApplyArgs {
  BnSrc a; BnSrc r; const half_t* res;   // r.x != null: residual is bn_r(r.x); else res (may be null)
  int M, Mpad, C; int relu;
  half_t* out;
};

__global__ __launch_bounds__(256) void bn_apply_kernel(BnApplyArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];            // scale[C] shift[C] (+ scale_r[C] shift_r[C])
  float* sc = sm; float* sh = sm + p.C; float* scr = sm + 2 * p.C; float* shr = sm + 3 * p.C;
  const float inv_m = 1.f / (float)p.M;
  const int cpr = p.C >> 3;                 // 16-B chunks per row
  const size_t total = (size_t)p.Mpad * cpr;
  // the thread's first chunk (in the layers where the launch latency counts, its only one) is requested BEFORE the statistics the
  // table is built from: one memory round trip instead of two dependent ones
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  half8 x, rr;
  auto issue = [&](size_t i) {
    const size_t row = i / cpr; const int c0 = (int)(i % cpr) * 8;
    if (row < (size_t)p.M) {
      x = *reinterpret_cast<const half8*>(p.a.x + row * p.C + c0);
      if (p.r.x) rr = *reinterpret_cast<const half8*>(p.r.x + row * p.C + c0);
      else if (p.res) rr = *reinterpret_cast<const half8*>(p.res + row * p.C + c0);
    }
  };
  if (idx < total) issue(idx);
  for (int c = threadIdx.x; c < p.C; c += 256) {
    bn_scale_shift(p.a, c, inv_m, sc[c], sh[c]);
    if (p.r.x) bn_scale_shift(p.r, c, inv_m, scr[c], shr[c]);
  }
  __syncthreads();
  while (idx < total) {
    const size_t row = idx / cpr; const int c0 = (int)(idx % cpr) * 8;
    half8 o;
    if (row < (size_t)p.M) {
      // the 8 scale / shift values of this thread as two 16-byte LDS reads each: element-wise reads put the lanes of a wave 32 bytes
      // apart on every access, an 8-way bank conflict (SQ_LDS_BANK_CONFLICT was 83 % of the kernel's LDS cycles)
      float scv[8], shv[8], scrv[8], shrv[8];
      *reinterpret_cast<f32x4*>(scv) = *reinterpret_cast<const f32x4*>(sc + c0); *reinterpret_cast<f32x4*>(scv + 4) = *reinterpret_cast<const f32x4*>(sc + c0 + 4);
      *reinterpret_cast<f32x4*>(shv) = *reinterpret_cast<const f32x4*>(sh + c0); *reinterpret_cast<f32x4*>(shv + 4) = *reinterpret_cast<const f32x4*>(sh + c0 + 4);
      if (p.r.x) {
        *reinterpret_cast<f32x4*>(scrv) = *reinterpret_cast<const f32x4*>(scr + c0); *reinterpret_cast<f32x4*>(scrv + 4) = *reinterpret_cast<const f32x4*>(scr + c0 + 4);
        *reinterpret_cast<f32x4*>(shrv) = *reinterpret_cast<const f32x4*>(shr + c0); *reinterpret_cast<f32x4*>(shrv + 4) = *reinterpret_cast<const f32x4*>(shr + c0 + 4);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = fmaf((float)x[j], scv[j], shv[j]);
        if (p.r.x) v += fmaf((float)rr[j], scrv[j], shrv[j]);
        else if (p.res) v += (float)rr[j];
        if (p.relu) v = fmaxf(v, 0.f);
        o[j] = (half_t)v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)0.f;
    }
    *reinterpret_cast<half8*>(p.out + row * p.C + c0) = o;
    idx += (size_t)gridDim.x * 256;
    if (idx < total) issue(idx);
  }
}

inline int run_conv(neraf_ctx* ctx, hipStream_t st, const Arch& A, const Layout& L, int ci, const char* packed, char* ws,
             const half_t* input) {
  const ConvSpec& c = A.conv[ci];
  GemmParams g{};
  const int M = (int)cube(c.dout);
  g.A = input; g.lda = c.cin;
  g.B = (const half_t*)(packed + L.w[ci]); g.ldb = conv_kpad(c);
  g.M = M; g.N = c.cout; g.K = conv_kpad(c); g.Mpad = (int)rows_pad(c.dout); g.Npad = conv_npad(c); g.alpha = 1.f;
  g.tile_n = c.cout == 64 ? 64 : 0;
  g.alg_flops = 2.0 * (double)cube(c.dout) * (c.k * c.k * c.k) * c.cin_real * c.cout;       // un-padded taps x cin
  g.C16 = (half_t*)(ws + L.pre[ci]); g.ldc16 = c.cout;
  float* stats = (float*)(ws + L.stat[ci]);
  const int cpad = round_up(c.cout, 128);
  g.colsum = stats; g.colsumsq = stats + cpad;
  g.stat_rep = stat_rep(c); g.stat_stride = kStatStride;
  const bool det = neraf_deterministic();
  if (det) { g.stat_det = 1; g.stat_rep = 0; g.stat_stride = 2 * cpad; }
  g.splitk_ws = (float*)(ws + L.splitk); g.splitk_ws_bytes = L.splitk_bytes;
  if (c.k == 1 && c.stride == 1) {
    g.conv.loader = 0;
  } else {
    g.conv.loader = c.cin == 8 ? 2 : 1;
    g.conv.din = c.din; g.conv.dout = c.dout; g.conv.stride = c.stride; g.conv.pad = c.pad; g.conv.ksize = c.k; g.conv.cin = c.cin;
    g.conv.zero_page = (const half_t*)(ws + L.zero_page);
  }
  if (int e = launch_gemm_f16(ctx, g, st)) return e;
  if (det) {       // {mean, biased variance} of the batch into L.fin[ci], slots added in a fixed order
    run_slot_sum(st, stats, det_slots(c), cpad, 1.f / (float)M, 1, (float*)(ws + L.fin[ci]));
    NERAF_HIP_CHECK(ctx, hipGetLastError());
  }
  return NERAF_OK;
}

inline int run_bn_apply(neraf_ctx* ctx, hipStream_t st, const BnApplyArgs& a) {
  const size_t total = (size_t)a.Mpad * (a.C >> 3);
  long blocks = (long)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(256), (size_t)4 * a.C * sizeof(float), st, a);
  if (ctx && ctx->manifest) {
    char nm[96];
    snprintf(nm, sizeof(nm), "bn_apply_kernel | M=%d C=%d%s", a.M, a.C, a.r.x ? " + bn(downsample)" : (a.res ? " + residual" : ""));
    neraf_node(ctx, nm, 0.0, (double)a.M * a.C * 2.0 * ((a.r.x || a.res) ? 2 : 1), (double)a.Mpad * a.C * 2.0);
  }
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}


}  // namespace

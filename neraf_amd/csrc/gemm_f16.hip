// fp16 MFMA GEMM / implicit-GEMM convolution (NT form) with fused epilogue for gfx950 -- the contraction
// behind the NAcF MLP (NeRAF_field.py:49-58, forward, dX and dW) and the Conv3d layers of the ResNet3D scene
// encoder (NeRAF_resnet3d.py:81-86,120).
//
//   C[m][n] = epi( alpha * sum_k A[m][k] * B[n][k] )
//
// Design (MI355X): 256-thread workgroup = 4 waves (2x2), block tile BM x BN x 64, each wave a (BM/2)x(BN/2)
// sub-tile of v_mfma_f32_16x16x32_f16 fragments.  An NST-stage LDS ring is filled by LDS-DMA
// (global_load_lds_dwordx4, 16 B per lane, no VGPR round trip); NST-1 tiles stay in flight across the ONE raw
// s_barrier per K-step behind a counted s_waitcnt vmcnt(N) -- with one workgroup per CU this is what hides the
// load latency (the register-staged double buffer it replaced ran 3.1x slower, profiles/r01_a_gemm_*.log).
// The LDS image is [row][8 x 16-B chunk] with the chunk index XOR-swizzled by (row & 7): ds_read_b128 fragment
// reads are bank-conflict free while each DMA instruction still writes 1 KiB linearly (the swizzle is applied
// to the per-lane SOURCE address).  MFMA operand order is (W-fragment, X-fragment) so a lane's 4 accumulator
// registers run along n and the tile leaves the CU as whole 128-B rows through an LDS-staged epilogue.
// Workgroup ids are remapped so each XCD (blockIdx % 8) owns a compact patch of tiles (L2 re-use of panels).
// Implicit-GEMM convolution only changes where an A chunk comes from (a filter tap of an input voxel, or a
// zero page for padding).  Under-filled grids (few tiles, long K) are split along K into fp32 partial slabs that
// a second kernel reduces and finishes.
// Bodies in this file: gemm_f16_nt_pipe_kernel (4 waves; 128x128 / 128x64 / 64x64 tiles, plain or implicit-conv loader),
// gemm_f16_nt_wide_kernel (8 waves, ping-pong K loop; 256x160, 256x128, 128x128 tiles of plain GEMMs), the bf16 "TN" weight-gradient
// kernels wgrad_grouped_tn_kernel / wgrad_wide_tn_kernel over all convolutions of a backward pass, and their reducers.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BK = 64;

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_LEAKY) return v > 0.f ? v : 0.1f * v;
  if (act == ACT_TANH10) return 10.f * tanhf(v);
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}

// element traits: fp16 (forward paths, scaled gradient chains) or bf16 (deep gradient chains that need fp32's range)
template <bool BF> struct ET;
template <> struct ET<false> {
  typedef _Float16 s; typedef half8 v8; typedef half4 v4;
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct ET<true> {
  typedef __bf16 s; typedef bf16x8 v8; typedef bf16x4 v4;
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

template <int BM, int BN, int WAVES_M = 2>
struct Tile {
  static constexpr int NWAVES = WAVES_M * 2, NT = NWAVES * 64;
  static constexpr int WM = BM / WAVES_M, WN = BN / 2;
  static constexpr int FM = WM / 16, FN = WN / 16;
  static constexpr int A_CH = BM * 8 / NT, B_CH = BN * 8 / NT;
  static constexpr int STAGE_BYTES = (BM + BN) * 128;
  static constexpr int IMG16_LD = WN + 8;    // halfs
  static constexpr int IMG16T_LD = WM + 8;   // halfs
  static constexpr int IMG32_LD = WN + 4;    // floats
  // the fp32 image is staged in C32_PASSES row slices when a whole wave tile per wave would not fit the LDS
  static constexpr int C32_PASSES = (NWAVES * WM * IMG32_LD * 4 > 144 * 1024) ? 2 : 1;
  static constexpr int cmax(int a, int b) { return a > b ? a : b; }
  static constexpr int EPI_BYTES_WAVE = cmax(cmax(WM / C32_PASSES * IMG32_LD * 4, WM * IMG16_LD * 2), WN * IMG16T_LD * 2);
};

// Parity-class row order of a stride-2 transposed convolution (ConvGeom::tclass): GEMM row m -> (class, voxel coordinates).
// Tile t = m / BM holds rows of class t & 7; within a class the voxels (z>>1, y>>1, x>>1) run in raster order.
template <int BM>
__device__ __forceinline__ void tclass_coords(const ConvGeom& g, int m, int& z, int& y, int& x) {
  const int t = m / BM, cls = t & 7;
  const int r = (t >> 3) * BM + (m - t * BM);
  const int h = g.dout >> 1;
  const int x2 = r % h, y2 = (r / h) % h, z2 = r / (h * h);
  z = 2 * z2 + ((cls >> 2) & 1); y = 2 * y2 + ((cls >> 1) & 1); x = 2 * x2 + (cls & 1);
}
template <int BM, bool TC>
__device__ __forceinline__ int out_row(const GemmParams& p, int m) {
  if (!TC) return m;
  int z, y, x;
  tclass_coords<BM>(p.conv, m, z, y, x);
  return (z * p.conv.dout + y) * p.conv.dout + x;
}

// ------------------------------------------------------------------------------------------------------
// Epilogue of the GEMM kernel (the split-K reducer applies the same operations element-wise).
template <int BM, int BN, bool BF, int WAVES_M = 2, bool TC = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[Tile<BM, BN, WAVES_M>::FM][Tile<BM, BN, WAVES_M>::FN],
                                              char* smem, int bm, int bn, int lane, int wave, int M, int N, float* C32, int ldc32) {
  using T = Tile<BM, BN, WAVES_M>;
  using E = typename ET<BF>::s;
  using E4 = typename ET<BF>::v4;
  constexpr int WM = T::WM, WN = T::WN, FM = T::FM, FN = T::FN;
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fq = lane >> 4;
  // lane holds, for fragment (i,j): m = wm*WM + i*16 + (lane&15), n = wn*WN + j*16 + (lane>>4)*4 + r
  const int m_tile0 = bm * BM, n_tile0 = bn * BN;
  const float alpha = p.alpha_dev ? p.alpha * (*p.alpha_dev) : p.alpha;
  const int m_l = frow;          // + i*16
  const int n_l = fq * 4;        // + j*16 + r
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m_tile0 + wm * WM + i * 16 + m_l;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n0 = n_tile0 + wn * WN + j * 16 + n_l;
      f32x4 v = acc[i][j] * alpha;
      if (p.bias) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n0);
        v += b;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], p.act);
      if (p.lmask) {
        const E4 mk = *reinterpret_cast<const E4*>(reinterpret_cast<const E*>(p.lmask) + (size_t)m * p.ldmask + n0);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= ((float)mk[r] > 0.f) ? 1.f : p.mask_slope;
      }
      if (p.add16) {
        const E4 ad = *reinterpret_cast<const E4*>(reinterpret_cast<const E*>(p.add16) + (size_t)out_row<BM, TC>(p, m) * p.ldadd + n0);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)ad[r];
      }
      if (p.bnb_mask) {
        const E4 mk = *reinterpret_cast<const E4*>(reinterpret_cast<const E*>(p.bnb_mask) + (size_t)m * p.ldbnb + n0);
#pragma unroll
        for (int r = 0; r < 4; ++r) if (!((float)mk[r] > 0.f)) v[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (m >= M || (n0 + r) >= N) v[r] = 0.f;
      acc[i][j] = v;
    }
  }

  char* img = smem + wave * T::EPI_BYTES_WAVE;
  const int m_w0 = m_tile0 + wm * WM, n_w0 = n_tile0 + wn * WN;

  if (p.colsum || p.colsumsq) {
    // column sums over the workgroup's BM rows: over i and the 16 lanes (lane&15) sharing a column by shuffles, over the two
    // wm waves through LDS, then ONE atomic per column per workgroup into replica (bm % stat_rep)
    float* red = reinterpret_cast<float*>(smem);        // [wm][sum | sumsq][BN]
    static_assert(2 * BN <= T::NT, "one thread per (sum | sumsq, column)");
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bnb_x) {
        // BatchNorm-backward sums: sum g and sum g * xhat over the tile's rows (rows >= M hold g = 0)
        const int n0 = n_tile0 + wn * WN + j * 16 + n_l;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(p.bnb_fin + n0);
        f32x4 rs = *reinterpret_cast<const f32x4*>(p.bnb_fin + p.bnb_cpad + n0);
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[r] = rsqrtf(rs[r] + 1e-5f);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int m = m_tile0 + wm * WM + i * 16 + m_l;
          const half4 xv = *reinterpret_cast<const half4*>(p.bnb_x + (size_t)out_row<BM, TC>(p, m) * p.ldbnb + n0);
          s += acc[i][j];
#pragma unroll
          for (int r = 0; r < 4; ++r) s2[r] += acc[i][j][r] * (((float)xv[r] - mu[r]) * rs[r]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < FM; ++i) { s += acc[i][j]; s2 += acc[i][j] * acc[i][j]; }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = s[r], y = s2[r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { x += __shfl_xor(x, o); y += __shfl_xor(y, o); }
        if (frow == 0) {
          const int col = wn * WN + j * 16 + n_l + r;
          red[(wm * 2 + 0) * BN + col] = x;
          red[(wm * 2 + 1) * BN + col] = y;
        }
      }
    }
    __syncthreads();
    const int t = wave * 64 + lane;
    if (t < 2 * BN) {
      const int which = t / BN, col = t - which * BN;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) v += red[(w * 2 + which) * BN + col];
      float* dst = which ? p.colsumsq : p.colsum;
      if (p.stat_det) {                 // this workgroup alone holds rows [m_tile0, m_tile0 + BM): a plain store into their first slot
        if (dst) dst[(size_t)(m_tile0 >> 5) * p.stat_stride + n_tile0 + col] = v;
      } else {
        const int rep = p.stat_rep > 1 ? (bm & (p.stat_rep - 1)) * p.stat_stride : 0;
        if (dst) atomicAdd(dst + rep + n_tile0 + col, v);
      }
    }
    __syncthreads();
  }

  if (p.C16) {
    E* im = reinterpret_cast<E*>(img);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        E4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (E)acc[i][j][r];
        *reinterpret_cast<E4*>(im + (i * 16 + m_l) * T::IMG16_LD + j * 16 + n_l) = h;
      }
    __syncthreads();
    constexpr int CPR = WN / 8;          // 16-B chunks per row
    static_assert((WM * CPR) % 64 == 0 || WM * CPR < 64, "whole wave-instructions (or one partial one)");
#pragma unroll
    for (int it = 0; it < (WM * CPR + 63) / 64; ++it) {
      const int idx = it * 64 + lane, row = idx / CPR, ch = idx - row * CPR;
      if (WM * CPR < 64 && idx >= WM * CPR) break;
      const uint4 d = *reinterpret_cast<const uint4*>(im + row * T::IMG16_LD + ch * 8);
      *reinterpret_cast<uint4*>(p.C16 + (size_t)out_row<BM, TC>(p, m_w0 + row) * p.ldc16 + n_w0 + ch * 8) = d;
    }
    __syncthreads();
  }

  if (p.C16T) {
    E* im = reinterpret_cast<E*>(img);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          im[(j * 16 + n_l + r) * T::IMG16T_LD + i * 16 + m_l] = (E)acc[i][j][r];
    __syncthreads();
    constexpr int CPR = WM / 8;
    static_assert((WN * CPR) % 64 == 0 || WN * CPR < 64, "whole wave-instructions (or one partial one)");
#pragma unroll
    for (int it = 0; it < (WN * CPR + 63) / 64; ++it) {
      const int idx = it * 64 + lane, row = idx / CPR, ch = idx - row * CPR;   // row = n, chunk along m
      if (WN * CPR < 64 && idx >= WN * CPR) break;
      const uint4 d = *reinterpret_cast<const uint4*>(im + row * T::IMG16T_LD + ch * 8);
      *reinterpret_cast<uint4*>(p.C16T + (size_t)(n_w0 + row) * p.ldc16t + m_w0 + ch * 8) = d;
    }
    __syncthreads();
  }

  if (C32) {
    float* im = reinterpret_cast<float*>(img);
    constexpr int FMP = FM / T::C32_PASSES, ROWS = WM / T::C32_PASSES;
#pragma unroll
    for (int pass = 0; pass < T::C32_PASSES; ++pass) {
      if (pass) __syncthreads();
#pragma unroll
      for (int i = 0; i < FMP; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          *reinterpret_cast<f32x4*>(im + (i * 16 + m_l) * T::IMG32_LD + j * 16 + n_l) = acc[pass * FMP + i][j];
      __syncthreads();
      const int m_p0 = m_w0 + pass * ROWS;
      if (WN % 64 == 0) {
        // one float per lane, 64 consecutive columns per wave-instruction (256-B segments)
#pragma unroll 4
        for (int row = 0; row < ROWS; ++row) {
          const int m = m_p0 + row;
          if (m >= M) break;
#pragma unroll
          for (int c0 = 0; c0 < WN; c0 += 64) {
            const int n = n_w0 + c0 + lane;
            if (n < N) {
              float* dst = C32 + (size_t)m * ldc32 + n;
              *dst = im[row * T::IMG32_LD + c0 + lane] + (p.c32_beta ? *dst : 0.f);
            }
          }
        }
      } else {
        // 64 consecutive elements of the [ROWS][WN] slice per wave-instruction
        static_assert((ROWS * WN) % 64 == 0, "whole wave-instructions");
#pragma unroll 4
        for (int it = 0; it < ROWS * WN / 64; ++it) {
          const int idx = it * 64 + lane, row = idx / WN, col = idx - row * WN;
          const int m = m_p0 + row, n = n_w0 + col;
          if (m < M && n < N) {
            float* dst = C32 + (size_t)m * ldc32 + n;
            *dst = im[row * T::IMG32_LD + col] + (p.c32_beta ? *dst : 0.f);
          }
        }
      }
    }
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
// s_waitcnt vmcnt(min(after, MAXT) * LOADS): the counter is an immediate, so the (uniform) tile count selects among MAXT + 1 forms
template <int LOADS, int MAXT>
__device__ __forceinline__ void wait_tiles(int after) {
  static_assert(MAXT * LOADS <= 63, "vmcnt is a 6-bit counter");
  if constexpr (MAXT <= 0) { wait_vmcnt<0>(); }
  else {
    if (after >= MAXT) wait_vmcnt<MAXT * LOADS>();
    else wait_tiles<LOADS, MAXT - 1>(after);
  }
}

template <int BM, int BN, int NST>
struct PipeTile {
  using T = Tile<BM, BN>;
  static constexpr int LDS_BYTES =
      (NST * T::STAGE_BYTES > T::NWAVES * T::EPI_BYTES_WAVE) ? NST * T::STAGE_BYTES : T::NWAVES * T::EPI_BYTES_WAVE;
};

// LOADER: 0 plain GEMM, 1 conv tap-per-K-step (cin % 64 == 0), 2 conv tap-per-chunk (cin == 8).  KS = filter size.
// NW = 8 (round 6): the same tile, ring and epilogue on twice the waves -- wave pair (w, w + 4) owns one wave tile and splits every K-step's
// two 32-deep halves (half the fragment reads and MFMAs per wave and K-step), the 512 threads stage half the chunks each, and the pair's
// accumulators are added through the (idle) ring before the unchanged 4-wave epilogue.  For the K-loops that run one or two waves per
// SIMD behind a long per-K-step issue chain (the implicit-GEMM convolutions), as for wgrad_wide_tn_kernel (profiles/r06_wgrad_waves_ab.txt).
template <int BM, int BN, int NST, int LOADER, int KS, bool BF, bool TC = false, int NW = 4>
__global__ __launch_bounds__(NW * 64) void gemm_f16_nt_pipe_kernel(GemmParams p, int splits) {
  using T = Tile<BM, BN>;
  using E8 = typename ET<BF>::v8;
  constexpr int WM = T::WM, WN = T::WN, FM = T::FM, FN = T::FN;
  constexpr int KH = NW / 4, NTH = NW * 64;
  static_assert((NW == 4 || NW == 8) && T::A_CH % KH == 0 && T::B_CH % KH == 0, "4 waves, or 8 with whole chunks per thread");
  constexpr int A_CH = T::A_CH / KH, B_CH = T::B_CH / KH;      // 16-byte chunks a thread stages per K-step
  constexpr int LOADS = A_CH + B_CH;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave8 & 3, kh = wave8 >> 2;      // wave tile; NW == 8: which half of each K-step this wave multiplies
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware tile mapping (bijective for any tile count) + split-K slice
  const int tiles_m = p.Mpad / BM, tiles_n = p.Npad / BN;
  const int ntiles = tiles_m * tiles_n;
  const int ngroups = p.ngroups > 1 ? p.ngroups : 1;
  const int nblocks = ntiles * splits * ngroups;
  int bid = blockIdx.x;
  {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const half_t* Ag = p.A; const half_t* Bg0 = p.B;
  float* C32g = p.C32; int Mg = p.M, Ng = p.N, ldc32g = p.ldc32, grp = 0;
  if (ngroups > 1) {
    grp = bid / (ntiles * splits); bid -= grp * ntiles * splits;
    Ag = p.grp[grp].A; Bg0 = p.grp[grp].B; C32g = p.grp[grp].C32; Mg = p.grp[grp].M; Ng = p.grp[grp].N; ldc32g = p.grp[grp].ldc32;
  }
  const int split = bid % splits;      // the slices of one tile are neighbours -> same XCD, shared panels
  bid /= splits;
  constexpr int GROUP_M = 4;
  const int group = bid / (GROUP_M * tiles_n);
  const int first_m = group * GROUP_M;
  const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
  const int in_group = bid - group * GROUP_M * tiles_n;
  const int bm = first_m + in_group % gsz;
  const int bn = in_group / gsz;

  // parity-class rows of a stride-2 transposed convolution: this tile's class walks taps t0 + 2i per axis only
  constexpr bool tclass = LOADER == 1 && TC;     // compile-time: the tap decode below must not weigh on the ordinary conv loaders
  int t0z = 0, t0y = 0, t0x = 0, cy = KS, cx = KS;
  int nk_total = p.K / BK;
  if (tclass) {
    const int cls = bm & 7, padc = -p.conv.pad;
    t0z = (((cls >> 2) & 1) + padc) & 1; t0y = (((cls >> 1) & 1) + padc) & 1; t0x = ((cls & 1) + padc) & 1;
    const int cz = t0z < KS ? (KS - t0z + 1) >> 1 : 0;
    cy = t0y < KS ? (KS - t0y + 1) >> 1 : 0; cx = t0x < KS ? (KS - t0x + 1) >> 1 : 0;
    nk_total = cz * cy * cx * (p.conv.cin >> 6);
    // a class no tap reaches (strided 1x1x1): the result is add16 itself -- nothing to do when that is the destination
    if (nk_total == 0 && p.add16 == p.C16 && !p.C16T && !p.C32 && !p.colsum && !p.colsumsq && splits == 1) return;
  }
  const int per = (nk_total + splits - 1) / splits;
  const int k_begin = split * per;
  const int k_end = (k_begin + per) < nk_total ? (k_begin + per) : nk_total;
  const int nk = k_end > k_begin ? k_end - k_begin : 0;

  const half_t* Bg = Bg0 + (size_t)bn * BN * p.ldb;
  const half_t* b_src[B_CH];
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    const int c = i * NTH + tid, row = c >> 3, lc = (c & 7) ^ (row & 7);
    b_src[i] = Bg + (size_t)row * p.ldb + lc * 8;
  }
  // A side: plain rows, or output-voxel coordinates for the convolution loaders
  const half_t* a_src[A_CH];
  int az[A_CH], ay[A_CH], ax[A_CH], alc[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int c = i * NTH + tid, row = c >> 3, lc = (c & 7) ^ (row & 7);
    alc[i] = lc;
    az[i] = ay[i] = ax[i] = 0;
    a_src[i] = nullptr;
    if (LOADER == 0) {
      a_src[i] = Ag + (size_t)(bm * BM + row) * p.lda + lc * 8;
    } else {
      const int m = bm * BM + row;
      const int d = p.conv.dout;
      int x = m % d, y = (m / d) % d, z = m / (d * d);
      if (tclass) tclass_coords<BM>(p.conv, m, z, y, x);
      az[i] = z * p.conv.stride - p.conv.pad; ay[i] = y * p.conv.stride - p.conv.pad; ax[i] = x * p.conv.stride - p.conv.pad;
    }
  }
  // Loader 1, fast path (everything but a stride-2 transposed convolution in plain row order).  The K-loop of a 64..128-row tile
  // runs ONE wave per SIMD, so every instruction between the barrier and the MFMAs is exposed: the generic form below decodes the
  // tap with scalar divisions and builds each chunk's address with bounds tests, a branch and 64-bit multiplies (~150 dependent
  // instructions per K-step, more than the MFMAs take).  Here a chunk keeps a 32-bit element offset of its base voxel and a packed
  // per-axis validity mask (bit a of byte 0 / 1 / 2: tap offset a along z / y / x stays inside the source grid), and the K-step keeps
  // scalar tap counters: a chunk's source is one AND, one compare, one add and two selects.
  const bool fastc = LOADER == 1 && (tclass || p.conv.tstride != 2) && nk_total > 0;
  int c_off[A_CH];
  unsigned c_msk[A_CH];
  int jz = 0, jy = 0, jx = 0, jcb = 0;                    // tap counters / channel block of the NEXT K-step to issue (uniform)
  const int lim_y = tclass ? cy : KS, lim_x = tclass ? cx : KS;
  const int tsgn = (LOADER == 1 && p.conv.tflip) ? -1 : 1;
  if (fastc) {
    const int cpb = p.conv.cin >> 6, din = p.conv.din;
    int t = k_begin / cpb;
    jcb = k_begin - t * cpb;
    jx = t % lim_x; t /= lim_x; jy = t % lim_y; jz = t / lim_y;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      int bz = az[i], by = ay[i], bx = ax[i];
      if (tclass) { bz = (bz - t0z) >> 1; by = (by - t0y) >> 1; bx = (bx - t0x) >> 1; }   // tap t0 + 2j reaches source voxel b - j
      unsigned m = 0;
#pragma unroll
      for (int a = 0; a < KS; ++a) {
        m |= ((unsigned)(bz + tsgn * a) < (unsigned)din ? 1u : 0u) << a;
        m |= ((unsigned)(by + tsgn * a) < (unsigned)din ? 1u : 0u) << (8 + a);
        m |= ((unsigned)(bx + tsgn * a) < (unsigned)din ? 1u : 0u) << (16 + a);
      }
      c_msk[i] = m;
      c_off[i] = ((bz * din + by) * din + bx) * p.conv.cin + alc[i] * 8;
    }
  }

  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  auto issue = [&](int kt, int stage) {
    char* sa = smem + stage * T::STAGE_BYTES + wave8 * 1024;   // wave-uniform base; HW adds lane*16
    char* sb = sa + BM * 128;
    int kb = kt;                                               // K-step of the B panel (differs from kt for parity-class tiles)
    if (LOADER == 0) {
#pragma unroll
      for (int i = 0; i < A_CH; ++i)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[i] + (size_t)kt * BK), (lds_ptr_t)(sa + i * (NW * 1024)), 16, 0, 0);
    } else if (LOADER == 1 && fastc) {
      const int cpb = p.conv.cin >> 6, din = p.conv.din;
      const unsigned sel = (1u << jz) | (1u << (8 + jy)) | (1u << (16 + jx));
      const int toff = tsgn * ((jz * din + jy) * din + jx) * p.conv.cin + jcb * 64;
      const int rz = tclass ? t0z + 2 * jz : jz, ry = tclass ? t0y + 2 * jy : jy, rx = tclass ? t0x + 2 * jx : jx;
      kb = ((rz * KS + ry) * KS + rx) * cpb + jcb;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const half_t* src = (c_msk[i] & sel) == sel ? p.A + (c_off[i] + toff) : p.conv.zero_page;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(sa + i * (NW * 1024)), 16, 0, 0);
      }
      if (++jcb == cpb) { jcb = 0; if (++jx == lim_x) { jx = 0; if (++jy == lim_y) { jy = 0; ++jz; } } }
    } else if (LOADER == 1) {
      const int cpb = p.conv.cin >> 6;
      int tap = kt / cpb;
      const int cb = kt - tap * cpb;
      int dz = tap / (KS * KS), dy = (tap / KS) % KS, dx = tap % KS;
      if (tclass) {                       // tap = index into this class' list of contributing taps
        dx = t0x + 2 * (tap % cx); dy = t0y + 2 * ((tap / cx) % cy); dz = t0z + 2 * (tap / (cx * cy));
        tap = (dz * KS + dy) * KS + dx;
        kb = tap * cpb + cb;
      }
      const int din = p.conv.din;
      const int sdz = p.conv.tflip ? -dz : dz, sdy = p.conv.tflip ? -dy : dy, sdx = p.conv.tflip ? -dx : dx;
      const bool half_grid = p.conv.tstride == 2;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        int iz = az[i] + sdz, iy = ay[i] + sdy, ix = ax[i] + sdx;
        bool ok = true;
        if (half_grid) { ok = ((iz | iy | ix) & 1) == 0; iz >>= 1; iy >>= 1; ix >>= 1; }   // negatives stay negative -> rejected below
        ok = ok && (unsigned)iz < (unsigned)din && (unsigned)iy < (unsigned)din && (unsigned)ix < (unsigned)din;
        const half_t* src = ok ? p.A + ((size_t)(iz * din + iy) * din + ix) * p.conv.cin + cb * 64 + alc[i] * 8 : p.conv.zero_page;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(sa + i * (NW * 1024)), 16, 0, 0);
      }
    } else {
      const int din = p.conv.din;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const int tap = kt * 8 + alc[i];
        const int dz = tap / (KS * KS), dy = (tap / KS) % KS, dx = tap % KS;
        const int iz = az[i] + dz, iy = ay[i] + dy, ix = ax[i] + dx;
        const bool ok = tap < KS * KS * KS && (unsigned)iz < (unsigned)din && (unsigned)iy < (unsigned)din && (unsigned)ix < (unsigned)din;
        const half_t* src = ok ? p.A + ((size_t)(iz * din + iy) * din + ix) * 8 : p.conv.zero_page;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(sa + i * (NW * 1024)), 16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(b_src[i] + (size_t)kb * BK), (lds_ptr_t)(sb + i * (NW * 1024)), 16, 0, 0);
  };

  const int frow = lane & 15, fq = lane >> 4;
  const int a_row_off = (wm * WM + frow) * 128;
  const int b_row_off = (wn * WN + frow) * 128;
  int ch_off[2];
  ch_off[0] = ((0 + fq) ^ (frow & 7)) * 16;
  ch_off[1] = ((4 + fq) ^ (frow & 7)) * 16;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < nk) issue(k_begin + s0, s0);

  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // iteration kt:  s_waitcnt vmcnt(loads of the tiles issued after kt) ; s_barrier ; issue tile kt+NST-1 ; MFMA
    // tile kt must have landed; the tiles issued after it (at most NST - 2: tile kt + NST - 1 is issued below) may stay in flight
    const int after = nk - 1 - kt;
    wait_tiles<LOADS, NST - 2>(after);
    __builtin_amdgcn_s_barrier();     // also frees stage (kt-1)%NST for the DMA issued below
    if (kt + NST - 1 < nk) {
      int st2 = stage + NST - 1; if (st2 >= NST) st2 -= NST;
      issue(k_begin + kt + NST - 1, st2);
    }
    const char* sa = smem + stage * T::STAGE_BYTES;
    const char* sb = sa + BM * 128;
#pragma unroll
    for (int ks0 = 0; ks0 < 2 / KH; ++ks0) {
      const int ks = NW == 8 ? kh : ks0;
      E8 xa[FM], wb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        xa[i] = *reinterpret_cast<const E8*>(sa + a_row_off + i * 16 * 128 + ch_off[ks]);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        wb[j] = *reinterpret_cast<const E8*>(sb + b_row_off + j * 16 * 128 + ch_off[ks]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          // D[r = n][c = m] : W fragment is the MFMA "A" operand, X fragment the "B" operand
          acc[i][j] = ET<BF>::mfma(wb[j], xa[i], acc[i][j]);
    }
    if (++stage == NST) stage = 0;
  }
  __syncthreads();
  if constexpr (NW == 8) {
    // the pair's two partial products: wave kh == 1 parks its accumulators in the idle ring, wave kh == 0 adds them (fixed order)
    f32x4* pr = reinterpret_cast<f32x4*>(smem) + wave * (FM * FN * 64);
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) pr[(i * FN + j) * 64 + lane] = acc[i][j];
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] += pr[(i * FN + j) * 64 + lane];
    __syncthreads();                 // (the four remaining waves) before the epilogue re-uses the ring
  }

  if (splits > 1) {
    // raw fp32 partial slab [split][Mpad][Npad]; finished by splitk_reduce_kernel
    float* slab = p.splitk_ws + ((size_t)grp * splits + split) * p.Mpad * p.Npad;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int m = bm * BM + wm * WM + i * 16 + frow;
        const int n0 = bn * BN + wn * WN + j * 16 + fq * 4;
        *reinterpret_cast<f32x4*>(slab + (size_t)m * p.Npad + n0) = acc[i][j];
      }
    return;
  }
  gemm_epilogue<BM, BN, BF, 2, tclass>(p, acc, smem, bm, bn, lane, wave, Mg, Ng, C32g, ldc32g);
}

// ------------------------------------------------------------------------------------------------------
// Wide plain-GEMM body for the big NAcF contractions: 256 x BN x 64 tile, 512 threads = 8 waves as 4 (m) x 2 (n), each wave a
// 64 x BN/2 sub-tile.  What the 128x128 body is short of on these shapes is operand bytes per flop through the L2 -> LDS path
// (32 KiB per 2.1 MFLOP K-step; measured ~45-55 GB/s per CU whatever the staging depth): the wide tile moves 52 KiB per
// 5.2 MFLOP, and its two waves per SIMD overlap one wave's ds_reads and waits with the other's MFMAs.  BN = 160 exists because
// the NAcF hidden width 5096 pads to 5120 = 32 x 160: with M = 2048 that is exactly 8 x 32 = 256 tiles, one per CU, where
// 128-wide tiles would leave a quarter-filled second round.
// The stage image is (256 + BN) rows x 128 B (A rows, then B rows); one LDS-DMA wave-instruction fills 8 rows, wave w issues
// instructions w, w+8, ... so some waves issue one more than the others -- each wave counts its own in s_waitcnt vmcnt.
template <int BM, int BN, int NST, bool BF>
__global__ __launch_bounds__(512) void gemm_f16_nt_wide_kernel(GemmParams p, int splits) {
  constexpr int WAVES_M = 4;
  using T = Tile<BM, BN, WAVES_M>;
  using E8 = typename ET<BF>::v8;
  constexpr int WM = T::WM, WN = T::WN, FM = T::FM, FN = T::FN;
  constexpr int NINSTR = (BM + BN) / 8, LBASE = NINSTR / 8, LREM = NINSTR % 8, LMAX = LBASE + (LREM ? 1 : 0);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles_m = p.Mpad / BM, tiles_n = p.Npad / BN;
  const int nblocks = tiles_m * tiles_n * splits;
  int bid = blockIdx.x;
  {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = bid % splits;
  bid /= splits;
  constexpr int GROUP_M = 4;
  const int group = bid / (GROUP_M * tiles_n);
  const int first_m = group * GROUP_M;
  const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
  const int in_group = bid - group * GROUP_M * tiles_n;
  const int bm = first_m + in_group % gsz;
  const int bn = in_group / gsz;

  const int nk_total = p.K / BK;
  const int per = (nk_total + splits - 1) / splits;
  const int k_begin = split * per;
  const int k_end = (k_begin + per) < nk_total ? (k_begin + per) : nk_total;
  const int nk = k_end > k_begin ? k_end - k_begin : 0;

  const half_t* src[LMAX];
#pragma unroll
  for (int i = 0; i < LMAX; ++i) {
    int q = wave + 8 * i;
    if (q >= NINSTR) q = NINSTR - 1;                      // never issued
    const int row = q * 8 + (lane >> 3), lc = (lane & 7) ^ (lane >> 3);
    src[i] = (row < BM ? p.A + (size_t)(bm * BM + row) * p.lda : p.B + (size_t)(bn * BN + row - BM) * p.ldb) + lc * 8;
  }

  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

  const int frow = lane & 15, fq = lane >> 4;
  const int a_row_off = (wm * WM + frow) * 128;
  const int b_row_off = (BM + wn * WN + frow) * 128;
  int ch_off[2];
  ch_off[0] = ((0 + fq) ^ (frow & 7)) * 16;
  ch_off[1] = ((4 + fq) ^ (frow & 7)) * 16;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // K loop, "ping-pong": the two waves of a SIMD (wave w and w+4) run half a K-step apart, so one issues its LDS-DMA pieces and
  // fragment reads (memory phase) while the other owns the matrix pipe (compute phase); every phase ends in one s_barrier.
  //   waves 0-3:        mem(0) | comp(0) | mem(1) | comp(1) | ...
  //   waves 4-7:  ----- |  mem(0) | comp(0) | mem(1) | ...
  // mem(k) reads K-step k's fragments into registers, issues the pieces of K-step k+2 into the stage K-step k-1 used (its last
  // reads retired, lgkmcnt(0), before the barrier that closed the other group's mem(k-1)), then waits until only those pieces
  // are outstanding: K-step k+1 has landed for this wave before the barrier, one phase ahead of its first reader.
  static_assert(NST == 3, "ring depth the phase schedule below is written for");
  auto kloop = [&](auto lw) {
    constexpr int L = decltype(lw)::value;               // LDS-DMA instructions this wave issues per stage
    auto issue = [&](int kt, int stage) {
      char* s = smem + stage * T::STAGE_BYTES + wave * 1024;   // wave-uniform base; HW adds lane*16
#pragma unroll
      for (int i = 0; i < L; ++i)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[i] + (size_t)kt * BK), (lds_ptr_t)(s + i * 8192), 16, 0, 0);
    };
    if (nk > 0) issue(k_begin, 0);
    if (nk > 1) { issue(k_begin + 1, 1); wait_vmcnt<L>(); } else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (wave >= 4) __builtin_amdgcn_s_barrier();
    int stage = 0;                                         // stage of K-step k; K-step k+2 goes to stage - 1 (mod 3)
    for (int k = 0; k < nk; ++k) {
      E8 xa[2][FM], wb[2][FN];
      const char* st = smem + stage * T::STAGE_BYTES;
      if (k + 2 < nk) issue(k_begin + k + 2, stage == 0 ? 2 : stage - 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
          xa[ks][i] = *reinterpret_cast<const E8*>(st + a_row_off + i * 16 * 128 + ch_off[ks]);
#pragma unroll
        for (int j = 0; j < FN; ++j)
          wb[ks][j] = *reinterpret_cast<const E8*>(st + b_row_off + j * 16 * 128 + ch_off[ks]);
      }
      if (k + 2 < nk) wait_vmcnt<L>(); else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = ET<BF>::mfma(wb[ks][j], xa[ks][i], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      if (++stage == NST) stage = 0;
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();
  };
  if (LREM == 0 || wave < LREM) kloop(std::integral_constant<int, LMAX>{});
  else kloop(std::integral_constant<int, LBASE>{});
  __syncthreads();

  if (splits > 1) {
    float* slab = p.splitk_ws + (size_t)split * p.Mpad * p.Npad;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int m = bm * BM + wm * WM + i * 16 + frow;
        const int n0 = bn * BN + wn * WN + j * 16 + fq * 4;
        *reinterpret_cast<f32x4*>(slab + (size_t)m * p.Npad + n0) = acc[i][j];
      }
    return;
  }
  gemm_epilogue<BM, BN, BF, WAVES_M>(p, acc, smem, bm, bn, lane, wave, p.M, p.N, p.C32, p.ldc32);
}

// Split-K reducer: sums the partial slabs (one float4 per thread per slab, eight loads in flight) and applies the same
// epilogue element-wise on 32x32 tiles.
template <bool BF>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p, int splits) {
  using E = typename ET<BF>::s;
  using E4 = typename ET<BF>::v4;
  __shared__ float tile[32][33];
  const int t = threadIdx.x;
  const int row = t >> 3, c4 = (t & 7) * 4;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int grp = blockIdx.z;
  float* C32 = p.C32; int M = p.M, N = p.N, ldc32 = p.ldc32;
  if (p.ngroups > 1) { C32 = p.grp[grp].C32; M = p.grp[grp].M; N = p.grp[grp].N; ldc32 = p.grp[grp].ldc32; }
  const float alpha = p.alpha_dev ? p.alpha * (*p.alpha_dev) : p.alpha;
  const size_t slab = (size_t)p.Mpad * p.Npad;
  const int m = m0 + row, n = n0 + c4;
  const float* src = p.splitk_ws + (size_t)grp * splits * slab + (size_t)m * p.Npad + n;
  // every optional operand of the epilogue is requested BEFORE the slab sums (their addresses depend on the thread alone): behind the
  // uniform branches below each was a further dependent memory round trip of a 6 us kernel
  f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f}, mu = bias4, var = bias4;
  E4 mk_l, ad, mk_b;
  half4 xv;
  if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
  if (p.lmask) mk_l = *reinterpret_cast<const E4*>(reinterpret_cast<const E*>(p.lmask) + (size_t)m * p.ldmask + n);
  if (p.add16) ad = *reinterpret_cast<const E4*>(reinterpret_cast<const E*>(p.add16) + (size_t)m * p.ldadd + n);
  if (p.bnb_mask) mk_b = *reinterpret_cast<const E4*>(reinterpret_cast<const E*>(p.bnb_mask) + (size_t)m * p.ldbnb + n);
  if (p.bnb_x) {
    xv = *reinterpret_cast<const half4*>(p.bnb_x + (size_t)m * p.ldbnb + n);
    mu = *reinterpret_cast<const f32x4*>(p.bnb_fin + n);
    var = *reinterpret_cast<const f32x4*>(p.bnb_fin + p.bnb_cpad + n);
  }
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= splits; s += 8) {
    f32x4 a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(s + u) * slab);
#pragma unroll
    for (int u = 0; u < 8; ++u) v += a[u];
  }
  for (; s < splits; ++s) v += *reinterpret_cast<const f32x4*>(src + (size_t)s * slab);
  v *= alpha;
  if (p.bias) v += bias4;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], p.act);
  if (p.lmask) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= ((float)mk_l[r] > 0.f) ? 1.f : p.mask_slope;
  }
  if (p.add16) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += (float)ad[r];
  }
  if (p.bnb_mask) {
#pragma unroll
    for (int r = 0; r < 4; ++r) if (!((float)mk_b[r] > 0.f)) v[r] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (m >= M || (n + r) >= N) v[r] = 0.f;
    tile[row][c4 + r] = v[r];
  }
  __shared__ float tile2[32][33];                 // g * xhat (fused BatchNorm-backward sums): one coalesced 8-byte load per thread
  if (p.bnb_x) {
#pragma unroll
    for (int r = 0; r < 4; ++r) tile2[row][c4 + r] = v[r] * (((float)xv[r] - mu[r]) * rsqrtf(var[r] + 1e-5f));
  }
  if (p.C16) {
    E4 h;
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = (E)v[r];
    *reinterpret_cast<E4*>(reinterpret_cast<E*>(p.C16) + (size_t)m * p.ldc16 + n) = h;
  }
  if (C32 && m < M) {
    float* dst = C32 + (size_t)m * ldc32 + n;
    if (n + 3 < N && (reinterpret_cast<size_t>(dst) & 15) == 0) {
      f32x4 o = v;
      if (p.c32_beta) o += *reinterpret_cast<const f32x4*>(dst);
      *reinterpret_cast<f32x4*>(dst) = o;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (n + r < N) dst[r] = v[r] + (p.c32_beta ? dst[r] : 0.f);
    }
  }
  __syncthreads();
  const int tx = t & 31, ty = t >> 5;
  if (p.C16T) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nn = n0 + ty + i * 8, mm = m0 + tx;
      reinterpret_cast<E*>(p.C16T)[(size_t)nn * p.ldc16t + mm] = (E)tile[tx][ty + i * 8];
    }
  }
  if ((p.colsum || p.colsumsq) && ty < 2) {
    // wave 0, lanes 0-31: column sums; lanes 32-63: the second statistic (squares, or g * xhat for the fused BatchNorm backward)
    float cs = 0.f;
    if (ty == 0) {
#pragma unroll 8
      for (int i = 0; i < 32; ++i) cs += tile[i][tx];
    } else if (p.bnb_x) {
#pragma unroll 8
      for (int i = 0; i < 32; ++i) cs += tile2[i][tx];
    } else {
#pragma unroll 8
      for (int i = 0; i < 32; ++i) { const float x = tile[i][tx]; cs += x * x; }
    }
    float* dst = ty == 0 ? p.colsum : p.colsumsq;
    if (p.stat_det) {                   // rows [32 blockIdx.y, +32) belong to this workgroup: slot blockIdx.y, plain store
      if (dst) dst[(size_t)blockIdx.y * p.stat_stride + n0 + tx] = cs;
    } else {
      const int rep = p.stat_rep > 1 ? (blockIdx.y & (p.stat_rep - 1)) * p.stat_stride : 0;
      if (dst) atomicAdd(dst + rep + n0 + tx, cs);
    }
  }
}

// manifest records (neraf_manifest_enable) of one GEMM launch and, with split-K, of its reducer: operands once (an implicit-GEMM
// convolution reads its source tensor once, not once per tap), results once; the K-split slabs where they are written and read
static void gemm_manifest(neraf_ctx* ctx, const GemmParams& p, int loader, int splits, int bm, int bn, double flops, bool wide = false) {
  const double out_elems = (double)p.M * p.N;
  const double rA = loader == 0 ? (double)p.M * p.K * 2.0 : (double)p.conv.din * p.conv.din * p.conv.din * p.conv.cin * 2.0;
  const double rB = (double)p.N * p.K * 2.0;
  const double epi_r = out_elems * 2.0 * ((p.add16 ? 1 : 0) + (p.lmask ? 1 : 0) + (p.bnb_mask ? 1 : 0) + (p.bnb_x ? 1 : 0));
  const double epi_w = out_elems * (2.0 * ((p.C16 ? 1 : 0) + (p.C16T ? 1 : 0)) + (p.C32 ? 4.0 : 0.0));
  const double slabs = (double)splits * p.Mpad * p.Npad * 4.0;
  char nm[96];
  // "<rocprofv3 kernel-name prefix> | description": the tool matches the prefix against the trace
  snprintf(nm, sizeof(nm), "%s<%d, %d | %s M=%d N=%d K=%d%s", wide ? "gemm_f16_nt_wide_kernel" : "gemm_f16_nt_pipe_kernel", bm, bn,
           loader == 0 ? "plain" : (p.conv.tflip ? "dgrad" : "conv"), p.M, p.N, p.K, splits > 1 ? " splitK" : "");
  if (splits > 1) {
    neraf_node(ctx, nm, flops, rA + rB, slabs);
    neraf_node(ctx, "splitk_reduce_kernel | K-split slabs -> result", 0.0, slabs + epi_r, epi_w);
  } else {
    neraf_node(ctx, nm, flops, rA + rB + epi_r, epi_w);
  }
}

template <int BM, int BN, int NST, int LOADER, int KS, bool BF, bool TC = false, int NW = 4>
int launch_pipe(neraf_ctx* ctx, const GemmParams& p, int splits, hipStream_t stream) {
  using PT = PipeTile<BM, BN, NST>;
  static bool attr_set = false;
  if (!attr_set) {
    NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_nt_pipe_kernel<BM, BN, NST, LOADER, KS, BF, TC, NW>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, PT::LDS_BYTES));
    attr_set = true;
  }
  const int ntiles = (p.Mpad / BM) * (p.Npad / BN);
  const int ng = p.ngroups > 1 ? p.ngroups : 1;
  // executed: what the grid multiplies (padded K, zero-page taps of a convolution's faces); algorithmic: the operation's own count
  double flops = 2.0 * p.M * p.N * p.K;
  if (ng > 1) { flops = 0.0; for (int g = 0; g < ng; ++g) flops += 2.0 * p.grp[g].M * p.grp[g].N * p.K; }
  const double exec_flops = (LOADER != 0 && p.conv.tstride > 1) ? (p.alg_flops > 0.0 ? p.alg_flops : flops) : flops;   // parity-class dgrad walks only reaching taps
  if (p.alg_flops > 0.0) flops = p.alg_flops;
  constexpr int kid = LOADER == 2 ? PROF_CONV_STEM
                      : LOADER == 1 ? (BM * BN == 128 * 128 ? PROF_CONV128 : (BM == 128 ? PROF_CONV12864 : (BF ? PROF_CONV64_BF16 : PROF_CONV)))
                                    : (BM * BN == 128 * 128 ? PROF_GEMM128 : (BM == 128 ? PROF_GEMM12864 : (BF ? PROF_GEMM64_BF16 : PROF_GEMM64)));
  ProfScope prof(ctx, stream, kid, flops, exec_flops);
  hipLaunchKernelGGL((gemm_f16_nt_pipe_kernel<BM, BN, NST, LOADER, KS, BF, TC, NW>), dim3(ntiles * splits * ng), dim3(NW * 64), PT::LDS_BYTES,
                     stream, p, splits);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  if (ctx && ctx->manifest) gemm_manifest(ctx, p, LOADER, splits, BM, BN, flops);
  if (splits > 1) {
    hipLaunchKernelGGL(splitk_reduce_kernel<BF>, dim3(p.Npad / 32, p.Mpad / 32, ng), dim3(256), 0, stream, p, splits);
    NERAF_HIP_CHECK(ctx, hipGetLastError());
  }
  return NERAF_OK;
}


template <int BM, int BN, int NST, bool BF>
int launch_wide(neraf_ctx* ctx, const GemmParams& p, int splits, hipStream_t stream) {
  using T = Tile<BM, BN, 4>;
  constexpr int LDS_BYTES = (NST * T::STAGE_BYTES > T::NWAVES * T::EPI_BYTES_WAVE) ? NST * T::STAGE_BYTES : T::NWAVES * T::EPI_BYTES_WAVE;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_nt_wide_kernel<BM, BN, NST, BF>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set = true;
  }
  const int ntiles = (p.Mpad / BM) * (p.Npad / BN);
  ProfScope prof(ctx, stream, PROF_GEMM_WIDE, p.alg_flops > 0.0 ? p.alg_flops : 2.0 * p.M * p.N * p.K, 2.0 * p.M * p.N * p.K);
  hipLaunchKernelGGL((gemm_f16_nt_wide_kernel<BM, BN, NST, BF>), dim3(ntiles * splits), dim3(512), LDS_BYTES, stream, p, splits);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  if (ctx && ctx->manifest) gemm_manifest(ctx, p, 0, splits, BM, BN, p.alg_flops > 0.0 ? p.alg_flops : 2.0 * p.M * p.N * p.K, true);
  if (splits > 1) {
    hipLaunchKernelGGL(splitk_reduce_kernel<BF>, dim3(p.Npad / 32, p.Mpad / 32, 1), dim3(256), 0, stream, p, splits);
    NERAF_HIP_CHECK(ctx, hipGetLastError());
  }
  return NERAF_OK;
}

// ------------------------------------------------------------------------------------------------------
// "TN" weight-gradient GEMM (fp16, or bf16 for the exported test entry): C[m][n] = sum_k A[k][m] * B[row(k, n-tile)][n], 64x64 tile, K-step 64.
//   A = dY [voxel k][cout] , B = activation [voxel][cin] with the convolution's tap shift applied to the row index, so the
//   operand tiles are [k][64 x 16-bit] rows exactly as they lie in HBM (no transposed copies, no im2col): LDS-DMA fills the
//   same 4-stage ring, and both MFMA operands are read with ds_read_b64_tr_b16 (4 k-rows x 16 columns per 16-lane group,
//   delivered column-major).  The 32-byte segment index of a row is XORed with ((r>>1)&1) | ((r>>3)&1)<<1 on the DMA's
//   SOURCE side, which makes every transposed read bank-conflict free (8 rows x 32 B of a 32-lane half cover all 64 banks).
// Grouped form: the geometry is taken per workgroup from a descriptor table in the kernel arguments (runtime loader type /
// filter size), so that every convolution of the network shares one grid.
struct WgDesc {                       // 64 bytes
  const void* dy; const void* x; float* out;
  unsigned long long slab_off;        // floats, into the shared slab buffer (unused when splits == 1)
  int block_begin; int K;
  unsigned short cout, cin, cin_real, taps, din, tiles_n, splits;
  unsigned char dl, stride, ksize, loader; signed char pad; unsigned char mh, ai, r2, r3, r4;   // mh: wide kernel, 64-row halves of the M tile (1 | 2); ai: index of the item's factor in alpha_dev
};
static_assert(sizeof(WgDesc) == 64, "descriptor table must fit the 4 KiB kernel-argument segment");
constexpr int kMaxWg = 56;      // 43 weight gradients at N_features = 1024, 53 with layer4; 56 x 64 B + header < 4 KiB of kernel arguments
struct WgTable { int n; int total_blocks; const half_t* zero_page; const float* alpha_dev; float* slab; int order; WgDesc d[kMaxWg]; };

__device__ __forceinline__ int wg_find(const WgTable& t, int bid) {
  int lo = 0, hi = t.n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t.d[mid].block_begin <= bid) lo = mid; else hi = mid - 1; }
  return lo;
}

__device__ __forceinline__ size_t wg_out_index(const WgDesc& d, int m, int n) {   // n = tap * cin + c  ->  [cout][cin_real][taps]
  const int tap = n / d.cin, c = n - tap * d.cin;
  return ((size_t)m * d.cin_real + c) * d.taps + tap;
}

template <bool BF> struct TrRead;
template <> struct TrRead<true> {
  static __device__ __forceinline__ bf16x4 rd(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
  }
};
template <> struct TrRead<false> {
  typedef __fp16 fp16x4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ half4 rd(const char* p) {
    const fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)p);
    return __builtin_bit_cast(half4, v);
  }
};

template <int NST, bool BF>
__global__ __launch_bounds__(256) void wgrad_grouped_tn_kernel(WgTable t) {
  using E = typename ET<BF>::s; using E4 = typename ET<BF>::v4; using E8 = typename ET<BF>::v8;
  constexpr int BM = 64, BN = 64;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int gi = wg_find(t, blockIdx.x);
  const WgDesc& D = t.d[gi];
  int bid = blockIdx.x - D.block_begin;
  const int splits = D.splits, tiles_n = D.tiles_n, tiles_m = D.cout / BM;
  const int split = bid % splits;
  bid /= splits;
  const int bm = bid % tiles_m, bn = bid / tiles_m;
  const int nk_total = D.K / BK;
  const int per = (nk_total + splits - 1) / splits;
  const int k_begin = split * per;
  const int k_end = (k_begin + per) < nk_total ? (k_begin + per) : nk_total;
  const int nk = k_end > k_begin ? k_end - k_begin : 0;
  (void)tiles_n;

  int lrow[2], lcs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = i * 32 + (tid >> 3);
    lrow[i] = row;
    lcs[i] = (tid & 7) ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1);
  }
  const int cout = D.cout, cin = D.cin, KS = D.ksize, loader = D.loader, din = D.din, stride = D.stride, pad = D.pad;
  const E* Ab = reinterpret_cast<const E*>(D.dy) + bm * BM;
  const E* Bb = reinterpret_cast<const E*>(D.x);
  int tap_dz = 0, tap_dy = 0, tap_dx = 0, cb = 0;
  if (loader == 1) {
    const int n0 = bn * BN;
    const int tap = n0 / cin; cb = n0 - tap * cin;
    tap_dz = tap / (KS * KS); tap_dy = (tap / KS) % KS; tap_dx = tap % KS;
  }
  const int dl = D.dl, dmask = (1 << dl) - 1;
  const E* zero = reinterpret_cast<const E*>(t.zero_page);

  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  auto issue = [&](int kt, int stage) {
    char* sa = smem + stage * STAGE_BYTES + wave * 1024;
    char* sb = sa + BM * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = kt * BK + lrow[i];
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Ab + (size_t)m * cout + lcs[i] * 8), (lds_ptr_t)(sa + i * 4096), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = kt * BK + lrow[i];
      const E* src;
      if (loader == 0) {
        src = Bb + (size_t)m * cin + bn * BN + lcs[i] * 8;
      } else {
        const int x = m & dmask, y = (m >> dl) & dmask, z = m >> (2 * dl);
        int dz = tap_dz, dy = tap_dy, dx = tap_dx;
        if (loader == 2) {
          const int tap = bn * 8 + lcs[i];
          dz = tap / (KS * KS); dy = (tap / KS) % KS; dx = tap % KS;
          if (tap >= KS * KS * KS) dz = 1 << 20;
        }
        const int iz = z * stride - pad + dz, iy = y * stride - pad + dy, ix = x * stride - pad + dx;
        const bool ok = (unsigned)iz < (unsigned)din && (unsigned)iy < (unsigned)din && (unsigned)ix < (unsigned)din;
        const size_t vox = ((size_t)(iz * din + iy) * din + ix);
        src = ok ? (loader == 2 ? Bb + vox * 8 : Bb + vox * cin + cb + lcs[i] * 8) : zero;
      }
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(sb + i * 4096), 16, 0, 0);
    }
  };

  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int swz = (((q >> 1) & 1) | ((g & 1) << 1)) << 1;
  int a_off[2], b_off[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int ca = wm * 32 + f * 16 + 4 * pp, cbb = wn * 32 + f * 16 + 4 * pp;
    a_off[f] = (8 * g + q) * 128 + ((((ca >> 3)) ^ swz) << 4) + ((pp & 1) << 3);
    b_off[f] = (8 * g + q) * 128 + ((((cbb >> 3)) ^ swz) << 4) + ((pp & 1) << 3);
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < nk) issue(k_begin + s0, s0);
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int after = nk - 1 - kt;
    if (NST >= 4 && after >= 2) wait_vmcnt<8>();
    else if (NST >= 3 && after >= 1) wait_vmcnt<4>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + NST - 1 < nk) {
      int st2 = stage + NST - 1; if (st2 >= NST) st2 -= NST;
      issue(k_begin + kt + NST - 1, st2);
    }
    char* sa = smem + stage * STAGE_BYTES;
    char* sb = sa + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      E8 fa[2], fb[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const E4 a0 = TrRead<BF>::rd(sa + a_off[f] + ks * 4096);
        const E4 a1 = TrRead<BF>::rd(sa + a_off[f] + ks * 4096 + 512);
        const E4 b0 = TrRead<BF>::rd(sb + b_off[f] + ks * 4096);
        const E4 b1 = TrRead<BF>::rd(sb + b_off[f] + ks * 4096 + 512);
#pragma unroll
        for (int e = 0; e < 4; ++e) { fa[f][e] = a0[e]; fa[f][4 + e] = a1[e]; fb[f][e] = b0[e]; fb[f][4 + e] = b1[e]; }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = ET<BF>::mfma(fb[j], fa[i], acc[i][j]);
    }
    if (++stage == NST) stage = 0;
  }
  const int frow = lane & 15, fq = lane >> 4;
  const int N = D.taps * cin, Npad = tiles_n * BN;
  if (splits == 1 && D.taps == 1) {      // 1x1x1: [cout][cin] rows are contiguous; filters with taps go through the row reducer
    const float alpha = t.alpha_dev ? t.alpha_dev[D.ai] : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = bm * BM + wm * 32 + i * 16 + frow;
        const int n0 = bn * BN + wn * 32 + j * 16 + fq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + r;
          if (n < N && (n % cin) < D.cin_real) D.out[wg_out_index(D, m, n)] = acc[i][j][r] * alpha;
        }
      }
    return;
  }
  float* slab = t.slab + D.slab_off + (size_t)split * cout * Npad;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = bm * BM + wm * 32 + i * 16 + frow;
      const int n0 = bn * BN + wn * 32 + j * 16 + fq * 4;
      *reinterpret_cast<f32x4*>(slab + (size_t)m * Npad + n0) = acc[i][j];
    }
}

// ---- wide form: (64*MH) x (64*NH) tiles built from 64-column HALF tiles -------------------------------------------------------
// The 64x64 tile above stages 16 KiB per 0.52 MFLOP K-step and re-reads dY once per N tile (PMC: 1.57 GB fetched per step for
// 0.45 GB of operands).  Here a workgroup stages MH + NH half tiles ([64 k][64 columns] each, the layout, swizzle and
// transposed reads of the kernel above) and every wave owns one 64x64 block of the product -- (A half, B half) = (w >> 1, w & 1)
// for MH = NH = 2 (cout >= 128) or (0, w) for MH = 1, NH = 4 (the 64-filter convolutions, half of the FLOPs): 32 / 40 KiB per
// 2.1 MFLOP K-step, 16 transposed reads per 16 MFMAs instead of 16 per 8.
// NW = 8 (round 6): the SAME tile and LDS footprint on twice the waves -- the wave pair (w, w + 4) owns one 64x64 block of the product and
// splits every K-step's two 32-deep halves between them (each wave: 8 + 8 transposed reads, 16 MFMAs per K-step instead of 32 + 32), the
// 512 threads stage a row each instead of two, and the pair's accumulators are added through LDS before the epilogue.  Why: the kernel is
// bound by how many waves a SIMD has to interleave (profiles/r06_wgrad_nst_ab.txt: 412 us at one wave per SIMD, 264 at two, whatever
// the ring depth), and the LDS admits no third 4-wave workgroup; two 8-wave workgroups give every SIMD four waves.
template <int NST, bool BF, int NW = 4>
__global__ __launch_bounds__(NW * 64) void wgrad_wide_tn_kernel(WgTable t) {
  constexpr int NI = NW == 4 ? 2 : 1;               // staged rows per thread: rows i * 32 + (tid >> 3)
  using E = typename ET<BF>::s; using E4 = typename ET<BF>::v4; using E8 = typename ET<BF>::v8;
  constexpr int HALF = 64 * 128;
  constexpr int STAGE_BYTES = 5 * HALF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // t.order != 0 (NERAF_WGRAD_XCD, measurement only): XCD-aware orders.  Workgroups are dealt to the 8 XCDs round robin
  // (blockIdx % 8), each with its own L2; inside an item the N groups of one (row tile, K slice) become neighbours -- they read the
  // same dY rows and, for a 3x3x3 filter's taps, activation rows that are the same lines shifted by a voxel -- and land on ONE XCD,
  // either as 8 contiguous runs of the block list (1) or octet by octet (2).  Measured: 257.7 us as shipped, 258.9 us with octets,
  // 390 us with runs (the items' blocks are not equally long: one XCD finishes last) -- this kernel is not waiting on the fabric.
  int lb = blockIdx.x;
  if (t.order == 1) {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = lb & 7, idx = lb >> 3;
    lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  } else if (t.order == 2) {
    // octets: of every 64 consecutive hardware blocks (8 per XCD) XCD x takes the logical blocks [64 s + 8 x, 64 s + 8 x + 8)
    const int nb_full = (int)gridDim.x & ~63;
    if (lb < nb_full) { const int w = lb & 63; lb = (lb & ~63) + ((w & 7) << 3) + (w >> 3); }
  }
  const int gi = wg_find(t, lb);
  const WgDesc& D = t.d[gi];
  const int MH = D.mh, NH = MH == 2 ? 2 : 4;
  const int pw = wave & 3, kh = wave >> 2;          // the 64x64 block this wave works on; NW == 8: which half of each K-step it multiplies
  const int ha = MH == 2 ? (pw >> 1) : 0, hb = MH == 2 ? (pw & 1) : pw;
  int bid = lb - D.block_begin;
  const int splits = D.splits, n64 = D.tiles_n, tiles_m = D.cout / (64 * MH);
  int split, bm, bnw;
  if (t.order) {
    const int nbnw = (n64 + NH - 1) / NH;
    bnw = bid % nbnw; bid /= nbnw;
    bm = bid % tiles_m; split = bid / tiles_m;
  } else {
    split = bid % splits; bid /= splits;
    bm = bid % tiles_m; bnw = bid / tiles_m;
  }
  const int nk_total = D.K / BK;
  const int per = (nk_total + splits - 1) / splits;
  const int k_begin = split * per;
  const int k_end = (k_begin + per) < nk_total ? (k_begin + per) : nk_total;
  const int nk = k_end > k_begin ? k_end - k_begin : 0;

  int lrow[NI], lcs[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = i * 32 + (tid >> 3);
    lrow[i] = row;
    lcs[i] = (tid & 7) ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1);
  }
  const int cout = D.cout, cin = D.cin, KS = D.ksize, loader = D.loader, din = D.din, stride = D.stride, pad = D.pad;
  const E* Ab = reinterpret_cast<const E*>(D.dy) + (size_t)bm * MH * 64;
  const E* Bb = reinterpret_cast<const E*>(D.x);
  const int dl = D.dl, dmask = (1 << dl) - 1;
  const E* zero = reinterpret_cast<const E*>(t.zero_page);
  // per B half: filter tap and channel block of its 64 columns (loader 1), validity
  // Per B half (loader 1) or per (row, B half) chunk (loader 2: the tap is the lane's 16-byte chunk): the tap as a packed selector
  // (bit dz | bit 8+dy | bit 16+dx) and as an element offset from the row's base voxel.  A K-step then computes, per staged row, the
  // base voxel offset and a packed per-axis validity mask, and a chunk's source is one AND, one compare, one multiply-add and a select
  // -- this loop runs one wave per SIMD, and the generic bounds tests + 64-bit address products cost more than its 32 MFMAs.
  int h_cb[4];
  bool h_ok[4];
  unsigned h_sel[NI][4];
  int h_toff[NI][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int b64 = bnw * NH + j;
    h_ok[j] = j < NH && b64 < n64;
    h_cb[j] = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) { h_sel[i][j] = 0u; h_toff[i][j] = 0; }
    if (loader != 0 && h_ok[j]) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        int tap;
        if (loader == 1) { const int n0 = b64 * 64; tap = n0 / cin; h_cb[j] = n0 - tap * cin; }
        else tap = b64 * 8 + lcs[i];
        const int dz = tap / (KS * KS), dy = (tap / KS) % KS, dx = tap % KS;
        h_sel[i][j] = tap < KS * KS * KS ? (1u << dz) | (1u << (8 + dy)) | (1u << (16 + dx)) : 0x80000000u;   // bit 31 is never valid
        h_toff[i][j] = (dz * din + dy) * din + dx;
      }
    }
  }

  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int swz = (((q >> 1) & 1) | ((g & 1) << 1)) << 1;
  int f_off[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int c = f * 16 + 4 * pp;
    f_off[f] = (8 * g + q) * 128 + (((c >> 3) ^ swz) << 4) + ((pp & 1) << 3);
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto body = [&](auto mh_c) {
    constexpr int MHc = decltype(mh_c)::value, NHc = MHc == 2 ? 2 : 4, NP = NI * (MHc + NHc);   // LDS-DMA pieces per thread and stage
    auto issue = [&](int kt, int stage) {
      char* base = smem + stage * STAGE_BYTES + wave * 1024;       // (eight waves: waves 4-7 write what i == 1 writes for four)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int m = kt * BK + lrow[i];
#pragma unroll
        for (int h = 0; h < MHc; ++h)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(Ab + (size_t)m * cout + h * 64 + lcs[i] * 8), (lds_ptr_t)(base + h * HALF + i * 4096), 16, 0, 0);
        int vbase = 0;
        unsigned vmask = 0u;
        if (loader != 0) {
          const int x = m & dmask, y = (m >> dl) & dmask, z = m >> (2 * dl);
          const int vz = z * stride - pad, vy = y * stride - pad, vx = x * stride - pad;
          vbase = (vz * din + vy) * din + vx;
#pragma unroll
          for (int a = 0; a < 5; ++a)
            if (a < KS) {
              vmask |= ((unsigned)(vz + a) < (unsigned)din ? 1u : 0u) << a;
              vmask |= ((unsigned)(vy + a) < (unsigned)din ? 1u : 0u) << (8 + a);
              vmask |= ((unsigned)(vx + a) < (unsigned)din ? 1u : 0u) << (16 + a);
            }
        }
#pragma unroll
        for (int j = 0; j < NHc; ++j) {
          const E* src = zero;
          if (h_ok[j]) {
            if (loader == 0) {
              src = Bb + (size_t)m * cin + (bnw * NHc + j) * 64 + lcs[i] * 8;
            } else if ((vmask & h_sel[i][j]) == h_sel[i][j]) {
              const int vox = vbase + h_toff[i][j];
              src = loader == 2 ? Bb + vox * 8 : Bb + (vox * cin + h_cb[j] + lcs[i] * 8);
            }
          }
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(base + (MHc + j) * HALF + i * 4096), 16, 0, 0);
        }
      }
    };
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0)
      if (s0 < nk) issue(k_begin + s0, s0);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const int after = nk - 1 - kt;
      // stage kt must have landed; up to NST - 2 younger stages stay in flight across the barrier (and NST - 1 during the MFMAs)
      if (NST >= 4 && after >= 2) wait_vmcnt<2 * NP>();
      else if (NST >= 3 && after >= 1) wait_vmcnt<NP>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (kt + NST - 1 < nk) {
        int st2 = stage + NST - 1; if (st2 >= NST) st2 -= NST;
        issue(k_begin + kt + NST - 1, st2);
      }
      char* sa = smem + stage * STAGE_BYTES + ha * HALF;
      char* sb = smem + stage * STAGE_BYTES + (MHc + hb) * HALF;
#pragma unroll
      for (int ks0 = 0; ks0 < (NW == 8 ? 1 : 2); ++ks0) {
        const int ks = NW == 8 ? kh : ks0;
        E8 fa[4], fb[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const E4 a0 = TrRead<BF>::rd(sa + f_off[f] + ks * 4096);
          const E4 a1 = TrRead<BF>::rd(sa + f_off[f] + ks * 4096 + 512);
          const E4 b0 = TrRead<BF>::rd(sb + f_off[f] + ks * 4096);
          const E4 b1 = TrRead<BF>::rd(sb + f_off[f] + ks * 4096 + 512);
#pragma unroll
          for (int e = 0; e < 4; ++e) { fa[f][e] = a0[e]; fa[f][4 + e] = a1[e]; fb[f][e] = b0[e]; fb[f][4 + e] = b1[e]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = ET<BF>::mfma(fb[j], fa[i], acc[i][j]);
      }
      if (++stage == NST) stage = 0;
    }
  };
  static_assert(NST == 2 || NST == 3 || NST == 4, "ring depths the waits are written for");
  if (MH == 2) body(std::integral_constant<int, 2>{}); else body(std::integral_constant<int, 1>{});

  const int b64 = bnw * NH + hb;
  const int frow = lane & 15, fq = lane >> 4;
  if constexpr (NW == 8) {
    // the pair's two partial products: wave kh == 1 parks its accumulators in the (now idle) staging ring, wave kh == 0 adds them
    constexpr int PR_LD = 68;
    __syncthreads();                                 // every wave is done with the staging ring
    float* pr = reinterpret_cast<float*>(smem) + pw * (64 * PR_LD);
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4*>(pr + (i * 16 + frow) * PR_LD + j * 16 + fq * 4) = acc[i][j];
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] += *reinterpret_cast<const f32x4*>(pr + (i * 16 + frow) * PR_LD + j * 16 + fq * 4);
  }
  if (b64 >= n64) return;
  const int N = D.taps * cin, Npad = n64 * 64;
  const int m_base = (bm * MH + ha) * 64, n_base = b64 * 64;
  if (splits == 1 && D.taps == 1) {      // 1x1x1: [cout][cin] rows are contiguous; filters with taps go through the row reducer
    const float alpha = t.alpha_dev ? t.alpha_dev[D.ai] : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m_base + i * 16 + frow;
        const int n0 = n_base + j * 16 + fq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + r;
          if (n < N && (n % cin) < D.cin_real) D.out[wg_out_index(D, m, n)] = acc[i][j][r] * alpha;
        }
      }
    return;
  }
  // the 64x64 fp32 product leaves through LDS so that a store instruction writes four whole 256-byte row segments (straight from the
  // MFMA layout it wrote sixteen 64-byte pieces of sixteen rows: half-lines, 120 MB of them per step)
  float* slab = t.slab + D.slab_off + (size_t)split * cout * Npad;
  constexpr int IM_LD = 68;
  __syncthreads();                                   // every wave is done with the staging ring
  float* im = reinterpret_cast<float*>(smem) + pw * (64 * IM_LD);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f32x4*>(im + (i * 16 + frow) * IM_LD + j * 16 + fq * 4) = acc[i][j];
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 4 + (lane >> 4), col = (lane & 15) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(im + row * IM_LD + col);
    *reinterpret_cast<f32x4*>(slab + (size_t)(m_base + row) * Npad + n_base + col) = v;
  }
}

// one workgroup per 32x32 output tile of every split item: sums the slabs, scales, writes the PyTorch layout
struct WgRedTable { int n; const float* alpha_dev; const float* slab; int tile_begin[kMaxWg + 1]; unsigned char item[kMaxWg]; };

// rows of the result in the PyTorch layout [cout][cin_real][taps] are CONTIGUOUS over (ci, tap): a workgroup takes one filter row m
// and a block of CB input channels, sums the K-split slabs for all taps (coalesced runs of CB floats per tap), turns the
// [tap][ci] block through LDS and writes CB * taps consecutive floats.  (The first reducer, and the direct epilogue of un-split
// items, wrote element by element at a stride of `taps` floats: 4-byte stores to one line each -- 13.8 M of them per step for the
// 512-voxel layers alone; with every load, LDS read and MFMA of the weight-gradient kernel removed it still took 154 of its 300 us.)
__device__ __forceinline__ int wg_red_cb(const WgDesc& D) { return D.taps == 1 ? (D.cin < 256 ? D.cin : 256) : (D.taps > 32 ? D.cin : (D.cin < 128 ? D.cin : 128)); }

__global__ __launch_bounds__(256) void wgrad_grouped_reduce_kernel(WgTable t, WgRedTable r) {
  __shared__ float blk[3584];                       // [CB][taps + 1] (128 x 28, 8 x 126) or [256] for taps == 1
  int lo = 0, hi = r.n - 1;
  const int bid = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (r.tile_begin[mid] <= bid) lo = mid; else hi = mid - 1; }
  const WgDesc& D = t.d[r.item[lo]];
  const int tile = bid - r.tile_begin[lo];
  const int taps = D.taps, cin = D.cin, CB = wg_red_cb(D), cbs = cin / CB;
  const int m = tile / cbs, ci0 = (tile - m * cbs) * CB;
  const int Npad = D.tiles_n * 64;
  const size_t slab = (size_t)D.cout * Npad;
  const float* src = r.slab + D.slab_off + (size_t)m * Npad + ci0;
  const int splits = D.splits, pitch = taps + 1, q4 = CB >> 2, total4 = q4 * taps;      // 16-byte loads: four channels per thread
  for (int e = threadIdx.x; e < total4; e += 256) {
    const int tap = e / q4, c = (e - tap * q4) * 4;
    const float* p = src + tap * cin + c;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    int s2 = 0;
    for (; s2 + 8 <= splits; s2 += 8) {
      f32x4 a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(s2 + u) * slab);
#pragma unroll
      for (int u = 0; u < 8; ++u) v += a[u];
    }
    for (; s2 < splits; ++s2) v += *reinterpret_cast<const f32x4*>(p + (size_t)s2 * slab);
#pragma unroll
    for (int k = 0; k < 4; ++k) blk[taps == 1 ? c + k : (c + k) * pitch + tap] = v[k];
  }
  __syncthreads();
  const float alpha = r.alpha_dev ? r.alpha_dev[D.ai] : 1.f;
  const int creal = (D.cin_real - ci0) < CB ? (D.cin_real - ci0) : CB;          // the stem: 7 real channels of 8
  float* dst = D.out + ((size_t)m * D.cin_real + ci0) * taps;
  for (int j = threadIdx.x; j < creal * taps; j += 256) {
    const int c = j / taps, tap = j - c * taps;
    dst[j] = blk[taps == 1 ? c : c * pitch + tap] * alpha;
  }
}

// split-K only pays when the K loop is long: below this many K-steps the slab round trip + reducer launch cost more than the
// shorter chain saves (NERAF_SPLIT_MIN_K overrides, for measurements)
static const int kWide = [] { const char* e = getenv("NERAF_GEMM_WIDE"); return e ? atoi(e) : 1; }();
static const int kSplitMinK = [] { const char* e = getenv("NERAF_SPLIT_MIN_K"); return e ? atoi(e) : 32; }();

// NERAF_CONV_WAVES=8: the implicit-GEMM convolutions' 64x64 tiles on 8 waves per workgroup (gemm_f16_nt_pipe_kernel NW)
static bool conv_waves8() { static const bool v = [] { const char* e = getenv("NERAF_CONV_WAVES"); return e && atoi(e) == 8; }(); return v; }

template <int LOADER, int KS, bool BF>
int dispatch_tile(neraf_ctx* ctx, const GemmParams& p_in, hipStream_t stream) {
  GemmParams p = p_in;
  const int cus = ctx ? ctx->num_cus : 256;
  // stride-2 transposed convolution: parity-class row order on 64x64 tiles (see ConvGeom::tclass); heavy and light classes
  // alternate tile by tile, so every XCD's share of the grid holds the same mix
  static const int kTclass = [] { const char* e = getenv("NERAF_DGRAD_TCLASS"); return e ? atoi(e) : 1; }();
  p.conv.tclass = 0;
  if (LOADER == 1 && kTclass && p.conv.tstride == 2 && p.conv.tflip && (p.conv.dout & 1) == 0 && p.ngroups <= 1 &&
      p.Mpad == p.conv.dout * p.conv.dout * p.conv.dout && (p.Mpad % 512) == 0 && (p.Npad % 64) == 0 && !p.lmask && !p.C16T && !p.C32 &&
      !p.colsum && !p.colsumsq) {
    p.conv.tclass = 1;
    if constexpr (LOADER == 1) { if (conv_waves8()) return launch_pipe<64, 64, 4, LOADER, KS, BF, true, 8>(ctx, p, 1, stream); }
    return launch_pipe<64, 64, 4, LOADER, KS, BF, LOADER == 1>(ctx, p, 1, stream);
  }
  const int nk = p.K / BK;
  // tile choice: 128x128 when it fills the chip; 128x64 for 64-wide outputs; otherwise 64x64 (4x the workgroups)
  int bm = 128, bn = 128;
  const bool can128 = (p.Mpad % 128) == 0 && (p.Npad % 128) == 0;
  // 64-filter 3x3x3 convolutions (32^3 voxels): 512 tiles of 64x64, two workgroups per CU, run the 27-tap loop ~25 % faster than
  // 256 tiles of 128x64 with one (28.8 -> 21 us; a lone 4-wave workgroup has nothing to hide its LDS-DMA waits behind);
  // NERAF_CONV_TILE64=0 restores the 128x64 tile
  static const int kTile64 = [] { const char* e = getenv("NERAF_CONV_TILE64"); return e ? atoi(e) : 1; }();
  if (p.tile_n == 64 && (p.Mpad % 128) == 0 && !(kTile64 && LOADER == 1)) { bm = 128; bn = 64; }
  else if (!can128 || (p.Mpad / 128) * (p.Npad / 128) < cus) { bm = 64; bn = 64; }
  const int ng = p.ngroups > 1 ? p.ngroups : 1;
  const int ntiles = (p.Mpad / bm) * (p.Npad / bn) * ng;
  // split-K: only with scratch, when the grid under-fills the chip and every slice keeps >= 4 K-steps
  int splits = 1;
  // implicit-GEMM convolutions on >= 64 tiles (the 4096-voxel layers) are split to TWO workgroups per CU: a lone 4-wave workgroup has
  // nothing to hide its LDS-DMA waits behind (16.1 -> 13.3 us per 27-tap convolution; on 32 tiles -- 512 voxels -- the extra slabs
  // cost the reducer what the GEMM gains: profiles/r05_splitk_oversubscription_ab.txt).  NERAF_SPLIT_OVERSUB=1 restores one per CU.
  static const int kOversub = [] { const char* e = getenv("NERAF_SPLIT_OVERSUB"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 4 ? 4 : v); }();
  if (p.splitk_ws && ntiles * 2 <= cus && nk >= kSplitMinK) {
    splits = ((LOADER != 0 && ntiles >= 64) ? kOversub : 1) * cus / ntiles;
    if (splits > nk / 4) splits = nk / 4;
    if (splits > 128) splits = 128;
    while (splits >= 2 && (size_t)splits * ng * p.Mpad * p.Npad * 4 > p.splitk_ws_bytes) splits >>= 1;
    if (splits < 2) splits = 1;
  }
  if (LOADER == 0 && ng == 1 && kWide && (p.Mpad % 256) == 0) {
    // wide body where its tiles fill the chip: 256x160 (the 5096-wide NAcF layers), else 256x128
    if ((p.Npad % 160) == 0 && (p.Mpad / 256) * (p.Npad / 160) * 4 >= cus * 3) return launch_wide<256, 160, 3, BF>(ctx, p, 1, stream);
    if ((p.Npad % 128) == 0 && (p.Mpad / 256) * (p.Npad / 128) >= cus) return launch_wide<256, 128, 3, BF>(ctx, p, 1, stream);
  }
  // 128x128 tiles, about one per CU (too few for two 4-wave workgroups per CU to overlap each other): the 8-wave ping-pong body
  // on the same tile -- the dominant NAcF forward layer, 2048 x 2048 x 5120 = 256 tiles
  if (LOADER == 0 && ng == 1 && kWide && bm == 128 && bn == 128 && splits == 1 && ntiles < 2 * cus)
    return launch_wide<128, 128, 3, BF>(ctx, p, 1, stream);
  // staging depth: the 128-wide tiles run faster with two workgroups per CU (2 / 3 stages) than with one and a deep ring
  // plain GEMMs whose 64x64 tiling leaves half the chip or more idle (<= 128 tiles: the 512- and 4096-voxel 1x1x1 convolutions of
  // layers 2-3 and their dgrads).  A CU takes in operand tiles at ~50-64 GB/s whatever the body and however deep the ring (0.247 us
  // per 16 KiB K-step, cold or L2-warm operands alike), so the K loop is bound by how many CUs pull: 32x32 tiles on 4x the
  // workgroups run it at 0.164 us per K-step -- 512 x 256 x 1024: 7.9 -> 5.8 us, 4096 x 128 x 512: 5.9 -> 4.9 us
  // (tools/small_gemm_bench.py), step -0.05 ms.  NERAF_TILE32 = largest 64x64-tile count that still takes the 32x32 form (0: off).
  static const int kTile32 = [] { const char* e = getenv("NERAF_TILE32"); return e ? atoi(e) : 128; }();
  if constexpr (LOADER == 0) {
    if (kTile32 && ng == 1 && splits == 1 && bm == 64 && bn == 64 && ntiles <= kTile32 && !p.C16T)
      return launch_pipe<32, 32, 4, LOADER, KS, BF>(ctx, p, 1, stream);
  }
  if (bm == 128 && bn == 128) return launch_pipe<128, 128, 2, LOADER, KS, BF>(ctx, p, splits, stream);
  if (bm == 128 && bn == 64) return launch_pipe<128, 64, 3, LOADER, KS, BF>(ctx, p, splits, stream);
  if constexpr (LOADER == 1) { if (conv_waves8()) return launch_pipe<64, 64, 4, LOADER, KS, BF, false, 8>(ctx, p, splits, stream); }
  return launch_pipe<64, 64, 4, LOADER, KS, BF>(ctx, p, splits, stream);
}

}  // namespace

int launch_wgrad_grouped(neraf_ctx* ctx, const WgradItem* items, int n, const half_t* zero_page, float* slab_ws, size_t slab_bytes,
                         const float* alpha_dev, hipStream_t stream, bool bf16) {
  if (n <= 0 || n > kMaxWg || !items || !zero_page) return neraf_fail(ctx, NERAF_EINVAL, "wgrad_grouped: bad arguments");
  static const int nst = [] { const char* e = getenv("NERAF_WGRAD_NST"); const int v = e ? atoi(e) : 3; return v < 2 ? 2 : (v > 4 ? 4 : v); }();   // 3 stages = 48 KiB: three workgroups per CU (5.42 -> 5.34 ms/step against 4 stages)
  static const int rounds = [] { const char* e = getenv("NERAF_WGRAD_ROUNDS"); return e ? atoi(e) : 8; }();
  static const int wide = [] { const char* e = getenv("NERAF_WGRAD_WIDE"); return e ? atoi(e) : 1; }();        // 0: 64x64 tiles only
  static const int wide_nst = [] { const char* e = getenv("NERAF_WGRAD_WIDE_NST"); const int v = e ? atoi(e) : 2; return (v == 3 || v == 4) ? v : 2; }();
  static const int wide_rounds = [] { const char* e = getenv("NERAF_WGRAD_WIDE_ROUNDS"); return e ? atoi(e) : 4; }();
  static const int wide_waves = [] { const char* e = getenv("NERAF_WGRAD_WAVES"); return (e && atoi(e) == 4) ? 4 : 8; }();   // 4: round 5's form (A/B): 264 vs 226 us
  const int LDS_BYTES = nst * 128 * 128;
  static bool attr_set = false;
  if (!attr_set) {
    NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_grouped_tn_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * 128));
    NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_grouped_tn_kernel<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 128 * 128));
    NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_grouped_tn_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * 128));
    NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wide_tn_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 5 * 8192));
    NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wide_tn_kernel<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 5 * 8192));
    NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wide_tn_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 5 * 8192));
    NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wide_tn_kernel<2, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 5 * 8192));
    NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wide_tn_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 5 * 8192));
    NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_grouped_tn_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 128 * 128));
    attr_set = true;
  }
  const int cus = ctx ? ctx->num_cus : 256;
  WgTable t{};
  WgRedTable r{};
  static const int xcd_order = [] { const char* e = getenv("NERAF_WGRAD_XCD"); return e ? atoi(e) : 0; }();
  t.n = n; t.zero_page = zero_page; t.alpha_dev = alpha_dev; t.slab = slab_ws; t.order = wide ? xcd_order : 0;
  r.alpha_dev = alpha_dev; r.slab = slab_ws;
  // K-steps of work per item -> splits so that no workgroup runs more than ~target K-steps (the whole grid shares the chip)
  double total_steps = 0.0;
  for (int i = 0; i < n; ++i) {
    const WgradItem& it = items[i];
    const int taps = it.ksize * it.ksize * it.ksize;
    total_steps += (double)(it.cout / 64) * (round_up(taps * it.cin, 64) / 64) * (it.K / 64) / (wide ? 4.0 : 1.0);   // wide tiles: 4 halves' products each
  }
  int target = (int)(total_steps / (cus * (double)(wide ? wide_rounds : rounds))) + 1;       // K-steps per workgroup: ~8 (4 wide) rounds of the chip
  if (target < 16) target = 16;
  size_t slab_off = 0; int blocks = 0, red_tiles = 0, nred = 0;
  double flops = 0.0, exec_flops = 0.0;
  for (int i = 0; i < n; ++i) {
    const WgradItem& it = items[i];
    const int taps = it.ksize * it.ksize * it.ksize;
    if ((it.cout % 64) || (it.K % 64) || it.dout <= 0 || (it.dout & (it.dout - 1)) || (it.cin % 64 && it.cin != 8) || taps * it.cin > 65535)
      return neraf_fail(ctx, NERAF_EINVAL, "wgrad_grouped: cout % 64, K % 64, power-of-two output edge, cin % 64 (or the 8-channel stem)");
    WgDesc& d = t.d[i];
    d.dy = it.dy; d.x = it.x; d.out = it.out; d.K = it.K;
    if (it.alpha_idx < 0 || it.alpha_idx > 255) return neraf_fail(ctx, NERAF_EINVAL, "wgrad_grouped: alpha_idx out of range");
    d.ai = (unsigned char)it.alpha_idx;
    d.cout = (unsigned short)it.cout; d.cin = (unsigned short)it.cin; d.cin_real = (unsigned short)it.cin_real; d.taps = (unsigned short)taps;
    d.din = (unsigned short)it.din; d.stride = (unsigned char)it.stride; d.ksize = (unsigned char)it.ksize; d.pad = (signed char)it.pad;
    d.dl = (unsigned char)(31 - __builtin_clz((unsigned)it.dout));
    d.loader = (it.ksize == 1 && it.stride == 1) ? 0 : (it.cin == 8 ? 2 : 1);
    const int tiles_n = round_up(taps * it.cin, 64) / 64, tiles_m = it.cout / 64, nk = it.K / 64;
    d.tiles_n = (unsigned short)tiles_n;
    int splits = (nk + target - 1) / target;
    if (splits > nk / 4) splits = nk / 4;
    if (splits < 1) splits = 1;
    if (splits > 128) splits = 128;
    d.splits = (unsigned short)splits;
    d.block_begin = blocks;
    d.mh = (unsigned char)(it.cout % 128 == 0 ? 2 : 1);
    if (wide) blocks += (it.cout / (64 * d.mh)) * ((tiles_n + (d.mh == 2 ? 2 : 4) - 1) / (d.mh == 2 ? 2 : 4)) * splits;
    else blocks += tiles_m * tiles_n * splits;
    d.slab_off = slab_off;
    if (splits > 1 || taps > 1) {
      slab_off += (size_t)splits * it.cout * tiles_n * 64;
      r.item[nred] = (unsigned char)i; r.tile_begin[nred] = red_tiles; ++nred;
      const int cb = taps == 1 ? (it.cin < 256 ? it.cin : 256) : (taps > 32 ? it.cin : (it.cin < 128 ? it.cin : 128));      // wg_red_cb
      red_tiles += it.cout * (it.cin / cb);
    }
    flops += 2.0 * it.cout * taps * it.cin_real * (double)it.dout * it.dout * it.dout;       // algorithmic
    exec_flops += 2.0 * it.cout * (double)tiles_n * 64 * (double)it.K;                       // padded columns and voxel rows
  }
  if (slab_off * 4 > slab_bytes) return neraf_fail(ctx, NERAF_EINVAL, "wgrad_grouped: split-K scratch too small");
  t.total_blocks = blocks;
  r.n = nred; r.tile_begin[nred] = red_tiles;
  {
    ProfScope prof(ctx, stream, PROF_WGRAD, flops, exec_flops);
    if (bf16) {           // the exported bf16 test entry: one body of each form
      if (wide) hipLaunchKernelGGL((wgrad_wide_tn_kernel<2, true>), dim3(blocks), dim3(256), 2 * 5 * 8192, stream, t);
      else hipLaunchKernelGGL((wgrad_grouped_tn_kernel<3, true>), dim3(blocks), dim3(256), 3 * 128 * 128, stream, t);
    }
    else if (wide && wide_nst == 2 && wide_waves == 8) hipLaunchKernelGGL((wgrad_wide_tn_kernel<2, false, 8>), dim3(blocks), dim3(512), 2 * 5 * 8192, stream, t);
    else if (wide && wide_nst == 2) hipLaunchKernelGGL((wgrad_wide_tn_kernel<2, false>), dim3(blocks), dim3(256), 2 * 5 * 8192, stream, t);
    else if (wide && wide_nst == 4) hipLaunchKernelGGL((wgrad_wide_tn_kernel<4, false>), dim3(blocks), dim3(256), 4 * 5 * 8192, stream, t);
    else if (wide) hipLaunchKernelGGL((wgrad_wide_tn_kernel<3, false>), dim3(blocks), dim3(256), 3 * 5 * 8192, stream, t);
    else if (nst == 2) hipLaunchKernelGGL((wgrad_grouped_tn_kernel<2, false>), dim3(blocks), dim3(256), LDS_BYTES, stream, t);
    else if (nst == 3) hipLaunchKernelGGL((wgrad_grouped_tn_kernel<3, false>), dim3(blocks), dim3(256), LDS_BYTES, stream, t);
    else hipLaunchKernelGGL((wgrad_grouped_tn_kernel<4, false>), dim3(blocks), dim3(256), LDS_BYTES, stream, t);
  }
  if (nred > 0) hipLaunchKernelGGL(wgrad_grouped_reduce_kernel, dim3(red_tiles), dim3(256), 0, stream, t, r);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  if (ctx && ctx->manifest) {
    double rb = 0.0, wb = 0.0;
    for (int i = 0; i < n; ++i) {
      const WgradItem& it = items[i];
      const double taps = (double)it.ksize * it.ksize * it.ksize;
      rb += (double)it.K * it.cout * 2.0 + (double)it.din * it.din * it.din * it.cin * 2.0;       // dY once, the activation once
      wb += (double)it.cout * it.cin_real * taps * 4.0;
    }
    neraf_node(ctx, "wgrad_ | all 43 weight gradients, TN GEMM", flops, rb, nred > 0 ? (double)slab_off * 4.0 : wb);
    if (nred > 0) neraf_node(ctx, "wgrad_grouped_reduce_kernel | slabs -> [cout][cin][taps]", 0.0, (double)slab_off * 4.0, wb);
  }
  return NERAF_OK;
}


int launch_gemm_f16(neraf_ctx* ctx, const GemmParams& p, hipStream_t stream) {
  if (p.K <= 0 || (p.K % BK) != 0) return neraf_fail(ctx, NERAF_EINVAL, "gemm: K must be a positive multiple of 64");
  if ((p.Mpad % 64) != 0 || (p.Npad % 64) != 0 || p.M > p.Mpad || p.N > p.Npad || p.M <= 0 || p.N <= 0)
    return neraf_fail(ctx, NERAF_EINVAL, "gemm: Mpad/Npad must be multiples of 64 covering M/N");
  if (!p.stat_det && p.stat_rep > 1 && (p.stat_rep & (p.stat_rep - 1))) return neraf_fail(ctx, NERAF_EINVAL, "gemm: stat_rep must be a power of two");
  if (p.ngroups > 1 && (p.ngroups > 6 || p.conv.loader != 0 || p.C16 || p.C16T || p.colsum || p.colsumsq || p.lmask || p.add16))
    return neraf_fail(ctx, NERAF_EINVAL, "gemm: grouped launches are plain GEMMs with fp32 results only");
  if ((p.conv.loader == 0 && (p.lda % 8)) || (p.ldb % 8) || (p.C16 && (p.ldc16 % 8)) || (p.C16T && (p.ldc16t % 8)) ||
      (p.lmask && (p.ldmask % 4)))
    return neraf_fail(ctx, NERAF_EINVAL, "gemm: leading dimensions must keep 16-byte alignment");
  switch (p.conv.loader) {
    case 0: return p.bf16 ? dispatch_tile<0, 1, true>(ctx, p, stream) : dispatch_tile<0, 1, false>(ctx, p, stream);
    case 1:
      if ((p.conv.cin % 64) || !p.conv.zero_page) return neraf_fail(ctx, NERAF_EINVAL, "conv loader 1: cin % 64 and zero page");
      if (p.conv.ksize == 3) return p.bf16 ? dispatch_tile<1, 3, true>(ctx, p, stream) : dispatch_tile<1, 3, false>(ctx, p, stream);
      if (p.conv.ksize == 1) return p.bf16 ? dispatch_tile<1, 1, true>(ctx, p, stream) : dispatch_tile<1, 1, false>(ctx, p, stream);
      return neraf_fail(ctx, NERAF_EINVAL, "conv loader 1: ksize must be 1 or 3");
    case 2:
      if (p.conv.cin != 8 || p.conv.ksize != 5 || !p.conv.zero_page || p.bf16) return neraf_fail(ctx, NERAF_EINVAL, "conv loader 2: cin 8, ksize 5, fp16");
      return dispatch_tile<2, 5, false>(ctx, p, stream);
  }
  return neraf_fail(ctx, NERAF_EINVAL, "gemm: unknown loader");
}

extern "C" int neraf_gemm_f16(neraf_ctx* ctx, const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                              int Mpad, int Npad, float alpha, const float* bias, int act, void* C16, int ldc16,
                              void* C16T, int ldc16t, float* C32, int ldc32, neraf_stream_t stream) {
  GemmParams p{};
  p.A = (const half_t*)A; p.lda = lda; p.B = (const half_t*)B; p.ldb = ldb;
  p.M = M; p.N = N; p.K = K; p.Mpad = Mpad; p.Npad = Npad; p.alpha = alpha; p.bias = bias; p.act = act;
  p.C16 = (half_t*)C16; p.ldc16 = ldc16; p.C16T = (half_t*)C16T; p.ldc16t = ldc16t; p.C32 = C32; p.ldc32 = ldc32;
  return launch_gemm_f16(ctx, p, (hipStream_t)stream);
}

// Same contraction with bfloat16 operands/results (the deep gradient chains use it; exported for tests).
extern "C" int neraf_gemm_bf16(neraf_ctx* ctx, const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                               int Mpad, int Npad, float alpha, const float* bias, int act, void* C16, int ldc16,
                               void* C16T, int ldc16t, float* C32, int ldc32, neraf_stream_t stream) {
  GemmParams p{};
  p.bf16 = 1;
  p.A = (const half_t*)A; p.lda = lda; p.B = (const half_t*)B; p.ldb = ldb;
  p.M = M; p.N = N; p.K = K; p.Mpad = Mpad; p.Npad = Npad; p.alpha = alpha; p.bias = bias; p.act = act;
  p.C16 = (half_t*)C16; p.ldc16 = ldc16; p.C16T = (half_t*)C16T; p.ldc16t = ldc16t; p.C32 = C32; p.ldc32 = ldc32;
  return launch_gemm_f16(ctx, p, (hipStream_t)stream);
}

// "TN" form with bfloat16 operands, both stored K-major and dense: C32[m][n] = sum_k A[k][m] * B[k][n], A [K][M], B [K][N], C32 [M][N];
// M, N, K multiples of 64.  This is the plain (1x1x1-convolution) case of the grouped weight-gradient kernel, exported for tests.
extern "C" int neraf_gemm_bf16_tn(neraf_ctx* ctx, const void* A, const void* B, int M, int N, int K, float* C32, void* splitk_ws,
                                  size_t splitk_bytes, neraf_stream_t stream) {
  if (!A || !B || !C32 || M <= 0 || N <= 0 || K <= 0 || (M % 64) || (N % 64) || (K % 64) || !splitk_ws || splitk_bytes < 256)
    return neraf_fail(ctx, NERAF_EINVAL, "gemm_bf16_tn: M, N, K multiples of 64; scratch required");
  WgradItem it{};
  it.dy = A; it.x = B; it.out = C32;
  it.cout = M; it.cin = N; it.cin_real = N; it.ksize = 1; it.stride = 1; it.pad = 0; it.din = 1; it.dout = 1; it.K = K;
  // the first 256 bytes of the scratch serve as the zero page the loaders never touch in this plain form
  return launch_wgrad_grouped(ctx, &it, 1, (const half_t*)splitk_ws, (float*)((char*)splitk_ws + 256), splitk_bytes - 256, nullptr,
                              (hipStream_t)stream, true);
}

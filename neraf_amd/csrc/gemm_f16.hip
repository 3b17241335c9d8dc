// fp16 MFMA GEMM (NT form) with fused epilogue for gfx950 -- the contraction behind the NAcF
// MLP (NeRAF_field.py:49-58) and the 1x1x1 convolutions of the ResNet3D (NeRAF_resnet3d.py:81,86).
//
//   C[m][n] = epi( alpha * sum_k A[m][k] * B[n][k] )
//
// Design (MI355X): 256-thread workgroup = 4 waves (2x2), block tile BM x BN x 64, each wave a
// (BM/2)x(BN/2) sub-tile of v_mfma_f32_16x16x32_f16 fragments.  Global->LDS staging goes through
// registers (issue the next tile's 16-B loads before the MFMA phase, write them to the other LDS
// stage after it: one barrier per K-step).  The LDS image is [row][8 x 16-B chunk] with the chunk
// index XOR-swizzled by (row & 7) so that the ds_read_b128 fragment reads (16 rows x one chunk per
// lane group) are bank-conflict free, while the image stays lane-linear per staging instruction.
// MFMA operand order is (W-fragment, X-fragment) so each lane's 4 accumulator registers run along
// n: the epilogue packs them into one 8-byte LDS write and the tile leaves the CU as whole 128-B
// rows.  Workgroup ids are remapped so that each XCD (blockIdx % 8) owns a compact 2-D patch of
// tiles and re-uses its A/B panels out of its private 4 MiB L2.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BK = 64;

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_LEAKY) return v > 0.f ? v : 0.1f * v;
  if (act == ACT_TANH10) return 10.f * tanhf(v);
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}

template <int BM, int BN>
struct Tile {
  static constexpr int WM = BM / 2, WN = BN / 2;
  static constexpr int FM = WM / 16, FN = WN / 16;
  static constexpr int A_CH = BM * 8 / 256, B_CH = BN * 8 / 256;
  static constexpr int STAGE_BYTES = (BM + BN) * 128;
  // epilogue images, per wave
  static constexpr int IMG16_LD = WN + 8;    // halfs
  static constexpr int IMG16T_LD = WM + 8;   // halfs
  static constexpr int IMG32_LD = WN + 4;    // floats
  static constexpr int EPI_BYTES_WAVE = WM * IMG32_LD * 4;
  static constexpr int LDS_BYTES =
      (2 * STAGE_BYTES > 4 * EPI_BYTES_WAVE) ? 2 * STAGE_BYTES : 4 * EPI_BYTES_WAVE;
};

template <int BM, int BN>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[Tile<BM, BN>::FM][Tile<BM, BN>::FN],
                                              char* smem, int bm, int bn, int lane, int wave) {
  using T = Tile<BM, BN>;
  constexpr int WM = T::WM, WN = T::WN, FM = T::FM, FN = T::FN;
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fq = lane >> 4;
  // ------------------------------- epilogue -------------------------------------------------
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
        const half4 mk = *reinterpret_cast<const half4*>(p.lmask + (size_t)m * p.ldmask + n0);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= ((float)mk[r] > 0.f) ? 1.f : p.mask_slope;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (m >= p.M || (n0 + r) >= p.N) v[r] = 0.f;
      acc[i][j] = v;
    }
  }

  char* img = smem + wave * T::EPI_BYTES_WAVE;
  const int m_w0 = m_tile0 + wm * WM, n_w0 = n_tile0 + wn * WN;

  if (p.colsum) {
    // column sums over this wave's WM rows: reduce over i and over the 16 lanes (lane&15) that share a column
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < FM; ++i) s += acc[i][j];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = s[r];
        x += __shfl_xor(x, 1);
        x += __shfl_xor(x, 2);
        x += __shfl_xor(x, 4);
        x += __shfl_xor(x, 8);
        if (frow == 0) atomicAdd(p.colsum + n_w0 + j * 16 + n_l + r, x);
      }
    }
  }

  if (p.C16) {
    half_t* im = reinterpret_cast<half_t*>(img);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        half4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (half_t)acc[i][j][r];
        *reinterpret_cast<half4*>(im + (i * 16 + m_l) * T::IMG16_LD + j * 16 + n_l) = h;
      }
    __syncthreads();
    constexpr int CPR = WN / 8;          // 16-B chunks per row
    constexpr int RPI = 64 / CPR;        // rows per wave-instruction
#pragma unroll
    for (int it = 0; it < WM / RPI; ++it) {
      const int row = it * RPI + lane / CPR, ch = lane % CPR;
      const uint4 d = *reinterpret_cast<const uint4*>(im + row * T::IMG16_LD + ch * 8);
      *reinterpret_cast<uint4*>(p.C16 + (size_t)(m_w0 + row) * p.ldc16 + n_w0 + ch * 8) = d;
    }
    __syncthreads();
  }

  if (p.C16T) {
    half_t* im = reinterpret_cast<half_t*>(img);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          im[(j * 16 + n_l + r) * T::IMG16T_LD + i * 16 + m_l] = (half_t)acc[i][j][r];
    __syncthreads();
    constexpr int CPR = WM / 8;
    constexpr int RPI = 64 / CPR;
#pragma unroll
    for (int it = 0; it < WN / RPI; ++it) {
      const int row = it * RPI + lane / CPR, ch = lane % CPR;   // row = n, chunk along m
      const uint4 d = *reinterpret_cast<const uint4*>(im + row * T::IMG16T_LD + ch * 8);
      *reinterpret_cast<uint4*>(p.C16T + (size_t)(n_w0 + row) * p.ldc16t + m_w0 + ch * 8) = d;
    }
    __syncthreads();
  }

  if (p.C32) {
    float* im = reinterpret_cast<float*>(img);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        *reinterpret_cast<f32x4*>(im + (i * 16 + m_l) * T::IMG32_LD + j * 16 + n_l) = acc[i][j];
    __syncthreads();
    // one float per lane, 64 consecutive columns per wave-instruction (256-B segments)
    constexpr int RPI = 64 / WN > 0 ? 64 / WN : 1;   // rows per instruction when WN < 64
    if (WN >= 64) {
#pragma unroll 4
      for (int row = 0; row < WM; ++row) {
        const int m = m_w0 + row;
        if (m >= p.M) break;
#pragma unroll
        for (int c0 = 0; c0 < WN; c0 += 64) {
          const int n = n_w0 + c0 + lane;
          if (n < p.N) p.C32[(size_t)m * p.ldc32 + n] = im[row * T::IMG32_LD + c0 + lane];
        }
      }
    } else {
      for (int r0 = 0; r0 < WM; r0 += RPI) {
        const int row = r0 + lane / WN, col = lane % WN;
        const int m = m_w0 + row, n = n_w0 + col;
        if (m < p.M && n < p.N) p.C32[(size_t)m * p.ldc32 + n] = im[row * T::IMG32_LD + col];
      }
    }
  }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_f16_nt_kernel(GemmParams p) {
  using T = Tile<BM, BN>;
  constexpr int WM = T::WM, WN = T::WN, FM = T::FM, FN = T::FN;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware tile mapping: consecutive hardware block ids round-robin over the 8 XCDs, so
  // give XCD x the contiguous logical range [x*per, (x+1)*per) and walk that range in GROUP_M-row
  // column-major groups (compact 2-D patch -> few distinct A/B panels per L2).
  const int tiles_m = p.Mpad / BM, tiles_n = p.Npad / BN;
  const int ntiles = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective for any ntiles
  }
  constexpr int GROUP_M = 4;
  const int group = bid / (GROUP_M * tiles_n);
  const int first_m = group * GROUP_M;
  const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
  const int in_group = bid - group * GROUP_M * tiles_n;
  const int bm = first_m + in_group % gsz;
  const int bn = in_group / gsz;

  const half_t* Ag = p.A + (size_t)bm * BM * p.lda;
  const half_t* Bg = p.B + (size_t)bn * BN * p.ldb;

  // ---- per-thread staging descriptors: chunk c = i*256 + tid -> LDS byte c*16 (lane-linear);
  // physical chunk slot pc = c & 7 holds logical chunk pc ^ (row & 7).
  const half_t* a_src[T::A_CH];
  const half_t* b_src[T::B_CH];
#pragma unroll
  for (int i = 0; i < T::A_CH; ++i) {
    const int c = i * 256 + tid, row = c >> 3, lc = (c & 7) ^ (row & 7);
    a_src[i] = Ag + (size_t)row * p.lda + lc * 8;
  }
#pragma unroll
  for (int i = 0; i < T::B_CH; ++i) {
    const int c = i * 256 + tid, row = c >> 3, lc = (c & 7) ^ (row & 7);
    b_src[i] = Bg + (size_t)row * p.ldb + lc * 8;
  }
  uint4 ra[T::A_CH], rb[T::B_CH];

  auto g_load = [&](int kt) {
#pragma unroll
    for (int i = 0; i < T::A_CH; ++i) ra[i] = *reinterpret_cast<const uint4*>(a_src[i] + (size_t)kt * BK);
#pragma unroll
    for (int i = 0; i < T::B_CH; ++i) rb[i] = *reinterpret_cast<const uint4*>(b_src[i] + (size_t)kt * BK);
  };
  auto s_store = [&](int stage) {
    char* sa = smem + stage * T::STAGE_BYTES;
    char* sb = sa + BM * 128;
#pragma unroll
    for (int i = 0; i < T::A_CH; ++i) *reinterpret_cast<uint4*>(sa + (i * 256 + tid) * 16) = ra[i];
#pragma unroll
    for (int i = 0; i < T::B_CH; ++i) *reinterpret_cast<uint4*>(sb + (i * 256 + tid) * 16) = rb[i];
  };

  // fragment read offsets (bytes): row = sub-tile base + (lane & 15); chunk = ks*4 + (lane >> 4)
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

  const int nk = p.K / BK;
  g_load(0);
  s_store(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1) < nk;
    if (more) g_load(kt + 1);
    const char* sa = smem + (kt & 1) * T::STAGE_BYTES;
    const char* sb = sa + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 xa[FM], wb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        xa[i] = *reinterpret_cast<const half8*>(sa + a_row_off + i * 16 * 128 + ch_off[ks]);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        wb[j] = *reinterpret_cast<const half8*>(sb + b_row_off + j * 16 * 128 + ch_off[ks]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          // D[r = n][c = m] : W fragment is the MFMA "A" operand, X fragment the "B" operand
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], xa[i], acc[i][j], 0, 0, 0);
    }
    if (more) s_store((kt + 1) & 1);
    __syncthreads();
  }

  gemm_epilogue<BM, BN>(p, acc, smem, bm, bn, lane, wave);
}

template <int BM, int BN>
int launch_tile(neraf_ctx* ctx, const GemmParams& p, hipStream_t stream) {
  using T = Tile<BM, BN>;
  static bool attr_set = false;
  if (!attr_set) {
    NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_nt_kernel<BM, BN>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
    attr_set = true;
  }
  const int ntiles = (p.Mpad / BM) * (p.Npad / BN);
  ProfScope prof(ctx, stream, BM == 128 ? PROF_GEMM128 : PROF_GEMM64, 2.0 * p.M * p.N * p.K);
  hipLaunchKernelGGL((gemm_f16_nt_kernel<BM, BN>), dim3(ntiles), dim3(256), T::LDS_BYTES, stream, p);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}


// ---- v2: NST-stage LDS ring filled by LDS-DMA (global_load_lds_dwordx4), counted vmcnt ------------
// The v1 structure keeps one tile (32 KB) in flight per CU and is load-latency bound (~2.5 us per
// K-step measured).  Here NST-1 tiles stay in flight across the single raw s_barrier per K-step, which
// is what saturates the per-CU load path (MI355X guide: >= 64-72 KB in flight per CU).
//   iteration kt:  s_waitcnt vmcnt(tiles issued after kt) ; s_barrier ; issue tile kt+NST-1 ; MFMA tile kt
// The barrier at the top of iteration kt is also what frees stage (kt-1)%NST for the new DMA.
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

template <int BM, int BN, int NST>
struct PipeTile {
  using T = Tile<BM, BN>;
  static constexpr int LDS_BYTES =
      (NST * T::STAGE_BYTES > 4 * T::EPI_BYTES_WAVE) ? NST * T::STAGE_BYTES : 4 * T::EPI_BYTES_WAVE;
};

template <int BM, int BN, int NST>
__global__ __launch_bounds__(256) void gemm_f16_nt_pipe_kernel(GemmParams p) {
  using T = Tile<BM, BN>;
  constexpr int WM = T::WM, WN = T::WN, FM = T::FM, FN = T::FN;
  constexpr int LOADS = T::A_CH + T::B_CH;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles_m = p.Mpad / BM, tiles_n = p.Npad / BN;
  const int ntiles = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GROUP_M = 4;
  const int group = bid / (GROUP_M * tiles_n);
  const int first_m = group * GROUP_M;
  const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
  const int in_group = bid - group * GROUP_M * tiles_n;
  const int bm = first_m + in_group % gsz;
  const int bn = in_group / gsz;

  const half_t* Ag = p.A + (size_t)bm * BM * p.lda;
  const half_t* Bg = p.B + (size_t)bn * BN * p.ldb;

  const half_t* a_src[T::A_CH];
  const half_t* b_src[T::B_CH];
#pragma unroll
  for (int i = 0; i < T::A_CH; ++i) {
    const int c = i * 256 + tid, row = c >> 3, lc = (c & 7) ^ (row & 7);
    a_src[i] = Ag + (size_t)row * p.lda + lc * 8;
  }
#pragma unroll
  for (int i = 0; i < T::B_CH; ++i) {
    const int c = i * 256 + tid, row = c >> 3, lc = (c & 7) ^ (row & 7);
    b_src[i] = Bg + (size_t)row * p.ldb + lc * 8;
  }

  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  auto issue = [&](int kt, int stage) {
    char* sa = smem + stage * T::STAGE_BYTES + wave * 1024;   // wave-uniform base; HW adds lane*16
    char* sb = sa + BM * 128;
#pragma unroll
    for (int i = 0; i < T::A_CH; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a_src[i] + (size_t)kt * BK), (lds_ptr_t)(sa + i * 4096), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < T::B_CH; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(b_src[i] + (size_t)kt * BK), (lds_ptr_t)(sb + i * 4096), 16, 0, 0);
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

  const int nk = p.K / BK;
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < nk) issue(s0, s0);

  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int after = nk - 1 - kt;   // tiles issued after tile kt that may still be in flight
    if (NST >= 4 && after >= 2) {
      if (NST == 4) wait_vmcnt<2 * LOADS>(); else wait_vmcnt<(NST - 2) * LOADS>();
    } else if (after >= 1 && NST >= 3) {
      // (for NST > 4 the tail over-waits, which is only slower, never wrong)
      wait_vmcnt<LOADS>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (kt + NST - 1 < nk) {
      int st2 = stage + NST - 1; if (st2 >= NST) st2 -= NST;
      issue(kt + NST - 1, st2);
    }
    const char* sa = smem + stage * T::STAGE_BYTES;
    const char* sb = sa + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 xa[FM], wb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        xa[i] = *reinterpret_cast<const half8*>(sa + a_row_off + i * 16 * 128 + ch_off[ks]);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        wb[j] = *reinterpret_cast<const half8*>(sb + b_row_off + j * 16 * 128 + ch_off[ks]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], xa[i], acc[i][j], 0, 0, 0);
    }
    if (++stage == NST) stage = 0;
  }
  __syncthreads();
  gemm_epilogue<BM, BN>(p, acc, smem, bm, bn, lane, wave);
}

template <int BM, int BN, int NST>
int launch_pipe(neraf_ctx* ctx, const GemmParams& p, hipStream_t stream) {
  using PT = PipeTile<BM, BN, NST>;
  static bool attr_set = false;
  if (!attr_set) {
    NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_nt_pipe_kernel<BM, BN, NST>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, PT::LDS_BYTES));
    attr_set = true;
  }
  const int ntiles = (p.Mpad / BM) * (p.Npad / BN);
  ProfScope prof(ctx, stream, BM == 128 ? PROF_GEMM128 : PROF_GEMM64, 2.0 * p.M * p.N * p.K);
  hipLaunchKernelGGL((gemm_f16_nt_pipe_kernel<BM, BN, NST>), dim3(ntiles), dim3(256), PT::LDS_BYTES, stream, p);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

}  // namespace

int launch_gemm_f16(neraf_ctx* ctx, const GemmParams& p, hipStream_t stream) {
  if (p.K <= 0 || (p.K % BK) != 0) return neraf_fail(ctx, NERAF_EINVAL, "gemm: K must be a positive multiple of 64");
  if ((p.Mpad % 128) != 0 || (p.Npad % 128) != 0 || p.M > p.Mpad || p.N > p.Npad || p.M <= 0 || p.N <= 0)
    return neraf_fail(ctx, NERAF_EINVAL, "gemm: Mpad/Npad must be multiples of 128 covering M/N");
  if ((p.lda % 8) || (p.ldb % 8) || (p.C16 && (p.ldc16 % 8)) || (p.C16T && (p.ldc16t % 8)) || (p.lmask && (p.ldmask % 4)))
    return neraf_fail(ctx, NERAF_EINVAL, "gemm: leading dimensions must keep 16-byte alignment");
  // Tile choice: 128x128 when it fills the chip, otherwise 64x64 tiles (4x the workgroups) so that
  // the narrow layers of the MLP (N = 512..1024 at B = 2048) still cover all 256 CUs.
  const int cus = ctx ? ctx->num_cus : 256;
  const int tiles128 = (p.Mpad / 128) * (p.Npad / 128);
  // NERAF_GEMM_VARIANT (A/B switch, read once): 1 = v1 register-staged double buffer; 3/4 = LDS-DMA ring depth.
  static const int variant = [] { const char* e = getenv("NERAF_GEMM_VARIANT"); return e ? atoi(e) : 4; }();
  if (variant == 1) {
    if (tiles128 >= cus) return launch_tile<128, 128>(ctx, p, stream);
    return launch_tile<64, 64>(ctx, p, stream);
  }
  if (variant == 3) {
    if (tiles128 >= cus) return launch_pipe<128, 128, 3>(ctx, p, stream);
    return launch_pipe<64, 64, 3>(ctx, p, stream);
  }
  if (tiles128 >= cus) return launch_pipe<128, 128, 4>(ctx, p, stream);
  return launch_pipe<64, 64, 4>(ctx, p, stream);
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

// Radiance half of the hot path (forward): piecewise sampler, proposal density (hash grid + 16-wide MLP),
// PDF resampling, the fused nerfacto field query (16-level hash grid -> base MLP -> SH/appearance -> colour
// MLP on MFMA, entirely in registers), and the volume-render composite.
//
// Replaces, for NeRAFVisionModel (NeRAF_model.py:54-79) / NeRAFVisionFieldValue.forward (NeRAF_field.py:33-34)
// and the refresh query (NeRAF_model.py:333-350), what nerfstudio + tiny-cuda-nn execute: ProposalNetworkSampler,
// HashMLPDensityField, NerfactoField, RaySamples.get_weights, RGB/Depth/AccumulationRenderer.  Those sources are
// not in the reference tree; formulas follow oracle/vision.py (parity unpinned).
//
// MI355X mapping of the field query: one wavefront owns 16 sample points per step.  Lane l = (p = l & 15,
// q = l >> 4): the four lanes that share p split the 16 hash levels of point p (4 levels x 8 corners each,
// half2 gathers from the fp16 table, which is L2/Infinity-Cache resident), and their 8 interpolated
// features ARE the v_mfma_f32_16x16x32_f16 B-operand fragment of the first layer.  Every layer computes
// D[out][point] = W[out][k] . X[k][point], so a lane's 4 accumulator registers are 4 consecutive outputs of
// its own point; two accumulator blocks converted to fp16 form the next layer's B fragment directly (the
// k-order permutation this implies is baked into the packed weight fragments), so activations never touch
// LDS or HBM.  All 24 weight fragments (24 KB fp16) live in registers for the whole kernel.
#include "field_common.h"

namespace {

// ---- sampler level 0: UniformLinDispPiecewiseSampler with single jitter -----------------------------------
__global__ __launch_bounds__(256) void sample_uniform_kernel(int R, int S, float near, float far, const float* __restrict__ jitter,
                                                            unsigned long long jitter_seed, float* __restrict__ s_bins,
                                                            float* __restrict__ e_bins) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)R * (S + 1)) return;
  const int ray = (int)(idx / (S + 1)), i = (int)(idx % (S + 1));
  const float step = 1.0f / (float)S;
  float b = (float)i * step;
  if (i == S) b = 1.0f;
  if (jitter || jitter_seed) {
    // bins = lower + (upper-lower)*u ; lower/upper = neighbouring bin centres (ends clamp to 0 / 1)
    const float cl = i > 0 ? ((float)(i - 1) * step + b) * 0.5f : 0.f;      // centre below (or bins[0])
    const float bn = (i + 1 == S) ? 1.0f : (float)(i + 1) * step;
    const float cu = i < S ? (b + bn) * 0.5f : 1.0f;                         // centre above (or bins[-1])
    const float lower = i == 0 ? 0.f : cl;
    const float upper = i == S ? 1.f : cu;
    b = lower + (upper - lower) * (jitter ? jitter[ray] : jitter_u01(jitter_seed, ray));
  }
  const float sn = spacing_fn(near), sf = spacing_fn(far);
  s_bins[idx] = b;
  e_bins[idx] = spacing_inv(b * sf + (1.f - b) * sn);
}

// ---- proposal density: hash grid (<= 8 levels) + MLP 2L->16->1, VALU ----------------------------------------
struct PropArgs {
  GridLayout g;
  const unsigned* table;        // fp16x2 entries
  const half_t* w;              // [16][16] layer 0 (row = hidden unit, col = input; cols >= 2L ignored) then [16] layer 1 row 0
  const float* origins; const float* dirs; const float* e_bins;
  unsigned e_stride;   // floats between the bin-edge rows of consecutive rays: S + 1, or 0 when every ray shares row 0
  int R, S; float avg_density;
  float* density;
  FastDiv divS;    // division by S
  RayTiles tiles;  // frame kernel only: how a tile's 64 slots map to rays
  FastDiv divRuns; // frame kernel only: division by S / 16
};

// density of one sample point from the proposal network: NL hash-grid levels -> 2 NL fp16 features -> 16 (ReLU) -> 1 -> avg * exp
// (the sample's table offsets ride in the gathers' scalar offset; the 2 NL -> 16 layer is v_dot2c_f32_f16 on the fp16 feature pairs tcnn
// hands to its MLP: two exact fp16 x fp16 products per instruction, fp32 accumulation)
template <int NL>
__device__ __forceinline__ float proposal_point(const PropArgs& a, const unsigned (&w0h)[16][8], const float (&w1)[16], float x, float y, float z) {
  const bool sel = map_position(x, y, z, 0, nullptr);
  half2v enc[NL];
  LevelCell cell[NL];
  unsigned raw[NL][8];
#pragma unroll
  for (int l = 0; l < NL; ++l)
    if (NL == 5 || l < a.g.n_levels) {
      level_cell(x, y, z, a.g.scale[l], a.g.res[l], a.g.size[l], a.g.hashed[l], cell[l]);
      gather_corners<true, true>(a.table, a.g.offset[l], cell[l], raw[l], !a.g.hashed[l] && l + 1 < a.g.n_levels, a.g.size[l]);
    }
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    enc[l] = half2v{(half_t)0.f, (half_t)0.f};
    if (NL == 5 || l < a.g.n_levels) {
      float f0, f1;
      interpolate_level(cell[l], raw[l], f0, f1);
      enc[l] = half2v{(half_t)f0, (half_t)f1};               // tcnn hands fp16 features to the MLP
    }
  }
  // 2 NL -> 16 (ReLU) -> 1 on the matrix pipe: v_mfma_f32_4x4x4_16b_f16 is sixteen independent 4 x 4 x 4 products per wave with
  // A = (lane % 4 = row, 4 k-values per lane) and D = (lane % 4 = column, 4 rows per lane) inside each group of four lanes
  // (tools/microbench/mfma4_layout.hip).  Rows = the four SAMPLES of a lane quad (a lane's own features are its A row: no cross-lane
  // traffic to form the operands), columns = four hidden units, B = their weights (the same for every quad, read from LDS by
  // lane % 4).  Four column groups x ceil(2 NL / 4) k-steps = 12 MFMAs for NL = 5 instead of 80 v_dot2c; lane (quad, j) then holds
  // hidden units {j, 4 + j, 8 + j, 12 + j} of its quad's four samples, and a two-step transpose-reduce over the quad (DPP) leaves
  // every lane with the layer-2 sum of its own sample.
  constexpr int NS = (2 * NL + 3) / 4;
  typedef _Float16 half4v __attribute__((ext_vector_type(4)));
  half4v av[NS];
#pragma unroll
  for (int ks = 0; ks < NS; ++ks) {
    const half2v lo = enc[2 * ks], hi = (2 * ks + 1 < NL) ? enc[2 * ks + 1] : half2v{(half_t)0.f, (half_t)0.f};
    av[ks] = half4v{lo[0], lo[1], hi[0], hi[1]};
  }
  const unsigned j = threadIdx.x & 3u;
  const char* wrow = reinterpret_cast<const char*>(&w0h[0][0]) + j * 32u;      // row (4 g + j) of the [16][16] fp16 layer-0 matrix
  float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      const half4v bv = *reinterpret_cast<const half4v*>(wrow + g * 128 + ks * 8);
      acc = __builtin_amdgcn_mfma_f32_4x4x4f16(av[ks], bv, acc, 0, 0, 0);
    }
    const float wj = w1[4 * g + j];
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = fmaf(wj, fmaxf(acc[i], 0.f), p[i]);
  }
  // p[i] = this lane's four hidden units' share of sample i of the quad; sum over the quad's lanes, sample j stays in lane j
  const bool o1 = (j & 1u) != 0u, o2 = (j & 2u) != 0u;
  float k0 = o1 ? p[1] : p[0], k1 = o1 ? p[3] : p[2];
  const float s0 = o1 ? p[0] : p[1], s1 = o1 ? p[2] : p[3];
  k0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  k1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, false));
  float out = o2 ? k1 : k0;
  const float snd = o2 ? k0 : k1;
  out += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, snd), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  return sel ? a.avg_density * __expf(out) : 0.f;
}

// General form (training batches, any S): 64 consecutive samples per wave in ray-major order.  The sample's (ray, index) pair comes
// from a multiply-high division by S (no per-lane 64-bit division).
template <int NL>    // levels compiled in (5: both nerfacto proposal networks; 8: any n_levels <= 8)
__global__ __launch_bounds__(256) void proposal_density_kernel(PropArgs a) {
  __shared__ unsigned w0h[16][8];      // layer 0 as half2 pairs: [hidden unit][level] = columns (2 l, 2 l + 1)
  __shared__ float w1[16];
  if (threadIdx.x < 128) w0h[threadIdx.x >> 3][threadIdx.x & 7] = reinterpret_cast<const unsigned*>(a.w)[threadIdx.x];
  if (threadIdx.x < 16) w1[threadIdx.x] = (float)a.w[256 + threadIdx.x];
  __syncthreads();
  const unsigned S = (unsigned)a.S;
  unsigned n = blockIdx.x * 256u + threadIdx.x;
  const bool valid = n < (unsigned)a.R * S;
  if (!valid) n = (unsigned)a.R * S - 1u;            // the layer arithmetic runs on whole lane quads (MFMA + DPP): compute, do not store
  const unsigned ray = fastdiv(n, a.divS), s = n - ray * S;
  const unsigned eb = ray * a.e_stride + s;
  const float t = 0.5f * (a.e_bins[eb] + a.e_bins[eb + 1u]);
  const float x = fmaf(a.dirs[ray * 3u + 0u], t, a.origins[ray * 3u + 0u]);
  const float y = fmaf(a.dirs[ray * 3u + 1u], t, a.origins[ray * 3u + 1u]);
  const float z = fmaf(a.dirs[ray * 3u + 2u], t, a.origins[ray * 3u + 2u]);
  const float dens = proposal_point<NL>(a, w0h, w1, x, y, z);
  if (valid) a.density[n] = dens;
}

// Frame form (coherent rays: the chunks of a camera frame; S % 16 == 0).  A wave owns a TILE of 64 neighbouring rays (an 8 x 8 pixel
// tile, or 64 consecutive rays) and a RUN of 16 consecutive sample indices, and walks the run: at every step its 64 lanes are the
// same sample index of neighbouring rays, so a gather instruction touches a handful of cache lines.  What the walk adds over "one
// wave per sample index" (the first coherent form, 179 -> 135 us): the rays' origins / directions are loaded once per run, the 17 bin
// edges of a ray as one 68-byte segment (per-step loads of e_bins with lanes 4 (S + 1) bytes apart were 64 cache lines per instruction),
// and the 16 densities of a ray leave as ONE 64-byte line through an LDS transpose (per-step stores were 64 partial lines per
// instruction).  Measured by dropping them: stores 13 %, edge loads up to 18 % of the kernel (profiles/r04_frame_kernel_experiments.txt).
template <int NL>
__global__ __launch_bounds__(256) void proposal_density_frame_kernel(PropArgs a) {
  __shared__ unsigned w0h[16][8];
  __shared__ float w1[16];
  __shared__ float t_s[4][16][64];     // [wave][step][lane]: sample mid-points of the run, each replaced by the step's density once consumed
  if (threadIdx.x < 128) w0h[threadIdx.x >> 3][threadIdx.x & 7] = reinterpret_cast<const unsigned*>(a.w)[threadIdx.x];
  if (threadIdx.x < 16) w1[threadIdx.x] = (float)a.w[256 + threadIdx.x];
  __syncthreads();
  const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const unsigned task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wv);
  const unsigned S = (unsigned)a.S, runs = a.divRuns.d;
  const unsigned tile = fastdiv(task, a.divRuns), s0 = (task - tile * runs) * 16u;
  if (tile >= a.tiles.n_tiles) return;
  unsigned ray;
  if (!tile_ray<8>(a.tiles, tile, lane, (unsigned)a.R, ray)) ray = (unsigned)a.R - 1u;      // computes like its neighbours, stores nothing
  const float ox = a.origins[ray * 3u + 0u], oy = a.origins[ray * 3u + 1u], oz = a.origins[ray * 3u + 2u];
  const float dx = a.dirs[ray * 3u + 0u], dy = a.dirs[ray * 3u + 1u], dz = a.dirs[ray * 3u + 2u];
  const float* eb = a.e_bins + (size_t)ray * a.e_stride + s0;
  float e0 = eb[0];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float e1 = eb[i + 1];
    t_s[wv][i][lane] = 0.5f * (e0 + e1);
    e0 = e1;
  }
#pragma unroll 1
  for (int j = 0; j < 16; ++j) {
    asm volatile("" ::: "memory");       // the layer weights are re-read from LDS every step (hoisted out of the loop they cost 100 VGPRs)
    const float t = t_s[wv][j][lane];
    t_s[wv][j][lane] = proposal_point<NL>(a, w0h, w1, fmaf(dx, t, ox), fmaf(dy, t, oy), fmaf(dz, t, oz));
  }
  // a ray's 16 densities = one aligned 64-byte line of density[ray][s0 ...]: four lanes write it as four float4
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned slot = (lane >> 2) + 16u * i, part = lane & 3u;
    unsigned r;
    const bool ok = tile_ray<8>(a.tiles, tile, slot, (unsigned)a.R, r);
    const float4 v = make_float4(t_s[wv][4u * part][slot], t_s[wv][4u * part + 1u][slot], t_s[wv][4u * part + 2u][slot], t_s[wv][4u * part + 3u][slot]);
    if (ok) *reinterpret_cast<float4*>(a.density + (size_t)r * S + s0 + 4u * part) = v;
  }
}

// ---- stand-alone multiresolution hash encoding (tiny-cuda-nn HashGrid forward; SURVEY 8b op list) ------------------------------
// One thread per (point, level): x01 [N,3] already mapped to [0,1]^3 -> enc [N][2 L] fp32, the trilinear interpolation of the fp16
// table entries of each level (the fused kernels compute exactly this per lane and feed it to the MLP as fp16).
__global__ __launch_bounds__(256) void hash_encode_kernel(GridLayout g, const unsigned* __restrict__ table, const float* __restrict__ x01,
                                                         long N, float* __restrict__ enc) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * g.n_levels) return;
  const long n = idx / g.n_levels; const int l = (int)(idx % g.n_levels);
  float f0, f1;
  encode_level(table, x01[n * 3], x01[n * 3 + 1], x01[n * 3 + 2], g.scale[l], g.res[l], g.size[l], g.offset[l], g.hashed[l], f0, f1);
  enc[n * 2 * g.n_levels + 2 * l] = f0;
  enc[n * 2 * g.n_levels + 2 * l + 1] = f1;
}

// ---- weights + PDF resampling, one wavefront per ray ------------------------------------------------------
// RaySamples.get_weights then PDFSampler.generate_ray_samples (single jitter, include_original=False).
struct PdfArgs {
  const float* density; const float* s_bins; const float* e_bins;
  int R, S; float anneal; const float* jitter; int n_new; float near, far;
  float* weights; float* s_new; float* e_new;
  unsigned long long jitter_seed;   // jitter == null and seed != 0: the per-ray jitter is drawn in the kernel (jitter_u01)
  size_t bins_stride;               // floats between the input bin rows of consecutive rays: S + 1, or 0 (every ray shares row 0)
  // {min, max} over the batch of the NEW samples' first / last mid-points (the expected-depth clip range of the composite that
  // follows, DepthRenderer "expected" [NS-recall]) folded into the sampler: mm_mode 1 = this launch SEEDS mm[0..1] (and zeroes
  // mm_zero further words: the loss node's sums), 2 = this launch ACCUMULATES its rays' range (n_new + 1 <= 64).  Round 5 spent two
  // launches per composite on this (minmax_seed + steps_minmax: 22 x 17.7 us of an 11.15 ms frame).
  unsigned* mm; int mm_mode; int mm_zero;
};

constexpr int PDF_MAX_S = 256;

__global__ __launch_bounds__(256) void pdf_resample_kernel(PdfArgs a) {
  __shared__ float cdf_s[4][PDF_MAX_S + 1];
  __shared__ float bins_s[4][PDF_MAX_S + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int ray = blockIdx.x * 4 + wv;
  const bool active = ray < a.R;                 // tail waves run on the last ray and store nothing
  if (!active) ray = a.R - 1;
  const int S = a.S;
  const int per = (S + 63) / 64;                 // samples per lane, contiguous chunk [lane*per, ...)
  const float* dens = a.density + (size_t)ray * S;
  const float* eb = a.e_bins + (size_t)ray * a.bins_stride;
  const float* sb = a.s_bins + (size_t)ray * a.bins_stride;
  float* cdf = cdf_s[wv];
  float* bins = bins_s[wv];
  for (int i = lane; i <= S; i += 64) bins[i] = sb[i];
  // --- weights: w_i = (1 - exp(-dd_i)) * exp(-sum_{j<i} dd_j)
  float dd[4], loc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = lane * per + k;
    dd[k] = (k < per && i < S) ? (eb[i + 1] - eb[i]) * dens[i] : 0.f;
    loc += dd[k];
  }
  const float incl = wave_incl_scan(loc, lane);
  float run = wave_excl_from_incl(incl, lane);    // exclusive prefix of this lane's chunk
  float wts[4], wl = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = lane * per + k;
    float w = (1.f - __expf(-dd[k])) * __expf(-run);
    if (!(w == w) ) w = 0.f;                       // nan_to_num
    run += dd[k];
    wts[k] = w;
    if (k < per && i < S) {
      if (a.weights && active) a.weights[(size_t)ray * S + i] = w;
      // annealed weight + histogram padding 0.01
      const float wa = (a.anneal == 1.f ? w : powf(w, a.anneal)) + 0.01f;
      wts[k] = wa; wl += wa;
    } else wts[k] = 0.f;
  }
  // --- pdf / cdf
  float tot = wl;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
  const float padding = fmaxf(1e-5f - tot, 0.f);
  const float addw = padding / (float)S;
  const float wsum = tot + padding;
  float loc2 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = lane * per + k;
    if (k < per && i < S) { wts[k] = (wts[k] + addw) / wsum; loc2 += wts[k]; }
  }
  const float incl2 = wave_incl_scan(loc2, lane);
  float c = incl2 - loc2;
  if (lane == 0) cdf[0] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = lane * per + k;
    if (k < per && i < S) { c += wts[k]; cdf[i + 1] = fminf(1.f, c); }
  }
  // cdf / bins are per-wave arrays: the wave's own LDS operations execute in order, so no workgroup barrier is needed -- only that the
  // compiler keeps the stores above the loads below (a barrier here made the four waves of a block wait for the slowest ray)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // --- inverse-CDF sampling of n_new+1 bin edges
  const int nb = a.n_new + 1;
  const float sn = spacing_fn(a.near), sf = spacing_fn(a.far);
  float e_first = 0.f;
  for (int j = lane; j < nb; j += 64) {
    float u = (float)j * ((1.f - 1.f / (float)nb) / (float)(nb - 1));    // linspace(0, 1-1/nb, nb)
    if (j == nb - 1) u = 1.f - 1.f / (float)nb;
    u += a.jitter ? a.jitter[ray] / (float)nb : (a.jitter_seed ? jitter_u01(a.jitter_seed, ray) / (float)nb : 1.f / (2.f * (float)nb));
    // searchsorted(cdf, u, side='right'): first index with cdf[idx] > u, over S+1 entries
    int lo = 0, hi = S + 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] > u) hi = mid; else lo = mid + 1; }
    const int below = min(max(lo - 1, 0), S), above = min(max(lo, 0), S);
    const float c0 = cdf[below], c1 = cdf[above], b0 = bins[below], b1 = bins[above];
    float t = (u - c0) / (c1 - c0);
    if (!(t == t)) t = 0.f;                      // nan_to_num(nan=0); +-inf are absorbed by the clip
    t = fminf(fmaxf(t, 0.f), 1.f);
    const float b = b0 + t * (b1 - b0);
    const float ev = spacing_inv(b * sf + (1.f - b) * sn);
    e_first = ev;                                 // (nb <= 64 when the range is asked for: lane j holds edge j)
    if (active) {
      a.s_new[(size_t)ray * nb + j] = b;
      a.e_new[(size_t)ray * nb + j] = ev;
    }
  }
  if (a.mm_mode == 1) {
    // 64 replicas of the pair, 256 bytes apart; the loss node's sums (mm_zero words) behind them
    if (blockIdx.x == 0) {
      if (threadIdx.x < 64) { a.mm[threadIdx.x * 64] = 0x7f7fffffu; a.mm[threadIdx.x * 64 + 1] = 0u; }
      else if ((int)threadIdx.x < 64 + a.mm_zero) a.mm[64 * 64 + (threadIdx.x - 64)] = 0u;
    }
  } else if (a.mm_mode == 2) {
    // the mid-points steps_minmax_kernel forms, from the values just stored: 0.5 (e[0] + e[1]) and 0.5 (e[nb-2] + e[nb-1]).
    // One atomic pair per ray into replica (ray & 63): 32768 rays = 512 pairs per replica, each replica its own line (one line for
    // all of them serialised the launch: 8192 workgroups x 3.6 ns).  Positive floats order like their bit patterns.
    const float e_next = __shfl_down(e_first, 1);
    const float mid = 0.5f * (e_first + e_next);
    const float lo = __shfl(mid, 0), hi = __shfl(mid, nb - 2);
    if (lane == 0) {
      unsigned* slot = a.mm + (size_t)(ray & 63) * 64;
      atomicMin(slot, __float_as_uint(lo));
      atomicMax(slot + 1, __float_as_uint(hi));
    }
  }
}

// ---- fused nerfacto field query -----------------------------------------------------------------------------
struct FieldArgs {
  GridLayout g;                  // 16 levels
  const unsigned* table;         // fp16x2
  const half8* wfrag;            // 24 fragments x 64 lanes (packed by the host layer, see neraf_amd/vision.py)
  const half_t* emb;             // fp16 [n_emb][32]; row used = cam_idx (training) or row `avg_row` (eval mean)
  const float* origins; const float* dirs; const float* e_bins; const int* cam_idx;
  int R, S; int mode; float aabb[6]; float avg_density; int avg_row;
  float* rgb; float* density;
  half_t* enc_out;               // optional fp16 [N][32]: the interpolated encoding, lane order (sample, quarter, 8) -- the backward reads
                                 // it back instead of walking the hash table again at one wave per SIMD
  half_t* denc_out;              // optional fp16 [N][4][24]: d enc / d mapped position of each lane's 4 levels x 2 features x 3 axes
                                 // (the camera-pose edge of the backward)
  FastDiv divS;                  // division by S
  RayTiles tiles; FastDiv divRuns;   // frame kernel only (see proposal_density_frame_kernel)
};

constexpr int NFRAG = 24;   // base0: 0-3, base1: 4-5, head0: 6-13 (ob*2+s), head1: 14-21, head2: 22-23

// the lane's 4 levels of the point -> B fragment of the first layer (k = 8q + 2*li + f); all 32 gathers of the lane in flight before
// the first is consumed.  SAVE == 2 also returns d enc / d mapped position (24 halfs: [li][feature][axis]).
template <int SAVE>
__device__ __forceinline__ half8 field_encode(const unsigned* __restrict__ table, float x, float y, float z, bool sel, int q, const float* l_scale,
                                              const int* l_res, const unsigned* l_size, const unsigned* l_off, const int* l_hash, half8 (&dh)[3]) {
  half8 xin;
  LevelCell cell[4];
  unsigned raw[4][8];
#pragma unroll
  for (int li = 0; li < 4; ++li) {
    const int l = 4 * q + li;
    level_cell(x, y, z, l_scale[l], l_res[l], l_size[l], l_hash[l], cell[li]);
    gather_corners<false>(table, l_off[l], cell[li], raw[li]);
  }
  if (SAVE == 2) {
    half_t* dp = reinterpret_cast<half_t*>(dh);
#pragma unroll
    for (int li = 0; li < 4; ++li) {
      float f0, f1, d0[3], d1_[3];
      interpolate_level_grad(cell[li], raw[li], l_scale[4 * q + li], f0, f1, d0, d1_);
      xin[2 * li] = (half_t)f0; xin[2 * li + 1] = (half_t)f1;
#pragma unroll
      for (int k = 0; k < 3; ++k) { dp[li * 6 + k] = (half_t)(sel ? d0[k] : 0.f); dp[li * 6 + 3 + k] = (half_t)(sel ? d1_[k] : 0.f); }
    }
  } else {
#pragma unroll
    for (int li = 0; li < 4; ++li) {
      float f0, f1;
      interpolate_level(cell[li], raw[li], f0, f1);
      xin[2 * li] = (half_t)f0; xin[2 * li + 1] = (half_t)f1;
    }
  }
  return xin;
}

// base MLP 32 -> 64 (ReLU) -> 16, colour head [base out | SH | appearance embedding] -> 64 -> 64 -> 3 for the wave's 16 points.
// d2[0] of the q == 0 lanes is the density logit, d5[0..2] of the q == 0 lanes the colour logits.
#define wf(f) wfp[(f) * 64]
__device__ __forceinline__ void field_mlp(const half8* wfp, const half8 xin, int q, const float (&sh)[4], const half8 h1, f32x4& d2, f32x4& d5) {
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 d1[4];
#pragma unroll
  for (int ob = 0; ob < 4; ++ob) d1[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf(ob), xin, zero, 0, 0, 0);
  d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf(4), pack_relu(d1[0], d1[1], true), zero, 0, 0, 0);
  d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf(5), pack_relu(d1[2], d1[3], true), d2, 0, 0, 0);
  // colour head input: k-step 0 = [base out 4q..4q+3 | SH 4q..4q+3], k-step 1 = appearance embedding 8q..8q+7
  half8 h0;
#pragma unroll
  for (int r = 0; r < 4; ++r) { h0[r] = (half_t)d2[r]; h0[4 + r] = (half_t)sh[r]; }
  if (q == 0) h0[0] = (half_t)0.f;              // the density logit is not an input of the head (weight column is 0 too)
  f32x4 d3[4], d4[4];
#pragma unroll
  for (int ob = 0; ob < 4; ++ob) {
    d3[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf(6 + ob * 2), h0, zero, 0, 0, 0);
    d3[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf(7 + ob * 2), h1, d3[ob], 0, 0, 0);
  }
  const half8 a0 = pack_relu(d3[0], d3[1], true), a1 = pack_relu(d3[2], d3[3], true);
#pragma unroll
  for (int ob = 0; ob < 4; ++ob) {
    d4[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf(14 + ob * 2), a0, zero, 0, 0, 0);
    d4[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf(15 + ob * 2), a1, d4[ob], 0, 0, 0);
  }
  d5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf(22), pack_relu(d4[0], d4[1], true), zero, 0, 0, 0);
  d5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf(23), pack_relu(d4[2], d4[3], true), d5, 0, 0, 0);
}
#undef wf

// General form (training batches, the grid refresh, any S): a wave takes 16 consecutive samples in ray-major order per step.
template <int SAVE>   // 0: outputs only; 1: also the encoding; 2: the encoding and its position derivatives
__global__ __launch_bounds__(256, SAVE == 2 ? 2 : 3) void field_query_kernel(FieldArgs a) {
  __shared__ float l_scale[MAX_LEVELS];
  __shared__ int l_res[MAX_LEVELS];
  __shared__ unsigned l_size[MAX_LEVELS], l_off[MAX_LEVELS];
  __shared__ int l_hash[MAX_LEVELS];
  if (threadIdx.x < MAX_LEVELS) {
    const int l = threadIdx.x;
    l_scale[l] = a.g.scale[l]; l_res[l] = a.g.res[l]; l_size[l] = a.g.size[l]; l_off[l] = a.g.offset[l]; l_hash[l] = a.g.hashed[l];
  }
  const int lane = threadIdx.x & 63;
  const int p = lane & 15, q = lane >> 4;
  // the 24 weight fragments live in LDS (24 KiB per workgroup), not in 96 VGPRs: what hides the latency of the lane's 32 table
  // gathers is waves per SIMD
  __shared__ half8 wf_s[NFRAG * 64];
  for (int i = threadIdx.x; i < NFRAG * 64; i += 256) wf_s[i] = a.wfrag[i];
  __syncthreads();
  const half8* wfp = wf_s + lane;
  // 32-bit sample arithmetic (the entry point refuses R * S >= 2^31)
  const unsigned N = (unsigned)a.R * (unsigned)a.S, S = (unsigned)a.S;
  const unsigned ngroups = (N + 15u) / 16u, gstride = gridDim.x * 4u;
  for (unsigned grp = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6)); grp < ngroups; grp += gstride) {
    unsigned n = grp * 16u + (unsigned)p;
    const bool valid = n < N;
    if (!valid) n = N - 1u;                      // MFMA needs every lane: clamp, compute, do not store
    const unsigned ray = fastdiv(n, a.divS), s = n - ray * S;
    const unsigned eb = ray * (S + 1u) + s;
    const float t = 0.5f * (a.e_bins[eb] + a.e_bins[eb + 1u]);
    const float dx = a.dirs[ray * 3u + 0u], dy = a.dirs[ray * 3u + 1u], dz = a.dirs[ray * 3u + 2u];
    float x = fmaf(dx, t, a.origins[ray * 3u + 0u]);
    float y = fmaf(dy, t, a.origins[ray * 3u + 1u]);
    float z = fmaf(dz, t, a.origins[ray * 3u + 2u]);
    const bool sel = map_position(x, y, z, a.mode, a.aabb);
    half8 dh[3];
    const half8 xin = field_encode<SAVE>(a.table, x, y, z, sel, q, l_scale, l_res, l_size, l_off, l_hash, dh);
    if (SAVE == 2 && valid) {
      half8* dst = reinterpret_cast<half8*>(a.denc_out + ((size_t)n * 4 + q) * 24);
      dst[0] = dh[0]; dst[1] = dh[1]; dst[2] = dh[2];
    }
    if (SAVE >= 1 && valid) *reinterpret_cast<half8*>(a.enc_out + ((size_t)n * 4 + q) * 8) = xin;
    float sh[4];
    sh4_quarter(q, dx, dy, dz, sh);
    const int erow = a.avg_row >= 0 ? a.avg_row : a.cam_idx[ray];
    const half8 h1 = *reinterpret_cast<const half8*>(a.emb + (size_t)erow * 32 + 8 * q);
    f32x4 d2, d5;
    field_mlp(wfp, xin, q, sh, h1, d2, d5);
    if (q == 0 && valid) {
      // density = avg * trunc_exp(logit) * selector ; logit = base output 0 -> lane q == 0, register 0
      a.density[n] = sel ? a.avg_density * __expf(d2[0]) : 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) a.rgb[(size_t)n * 3 + c] = __builtin_amdgcn_rcpf(1.f + __expf(-d5[c]));   // v_rcp_f32: 1 ulp
    }
  }
}

// Frame form (coherent rays, S % 16 == 0; see proposal_density_frame_kernel): a wave owns a tile of 16 neighbouring rays (a 4 x 4 pixel
// tile, or 16 consecutive rays) and a run of 16 sample indices.  Per run and ray: origin, direction, SH of the direction and the
// appearance-embedding fragment once; the run's bin edges as one segment; its 16 densities as one 64-byte line and its 48 colour values
// as three, through LDS (per-step stores with lanes 4 S bytes apart were 16 partial lines per instruction, four instructions a step).
__global__ __launch_bounds__(256, 3) void field_query_frame_kernel(FieldArgs a) {
  __shared__ float l_scale[MAX_LEVELS];
  __shared__ int l_res[MAX_LEVELS];
  __shared__ unsigned l_size[MAX_LEVELS], l_off[MAX_LEVELS];
  __shared__ int l_hash[MAX_LEVELS];
  __shared__ half8 wf_s[NFRAG * 64];
  __shared__ float t_s[4][16][16];       // [wave][step][ray slot]
  __shared__ float den_s[4][16][17];     // [wave][ray slot][step] (+1: conflict-free column writes)
  __shared__ float rgb_s[4][16][49];     // [wave][ray slot][3 step + c]
  if (threadIdx.x < MAX_LEVELS) {
    const int l = threadIdx.x;
    l_scale[l] = a.g.scale[l]; l_res[l] = a.g.res[l]; l_size[l] = a.g.size[l]; l_off[l] = a.g.offset[l]; l_hash[l] = a.g.hashed[l];
  }
  for (int i = threadIdx.x; i < NFRAG * 64; i += 256) wf_s[i] = a.wfrag[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int p = lane & 15, q = lane >> 4;
  const half8* wfp = wf_s + lane;
  const unsigned S = (unsigned)a.S, runs = a.divRuns.d, ntasks = a.tiles.n_tiles * runs, gstride = gridDim.x * 4u;
  for (unsigned task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wv); task < ntasks; task += gstride) {
    const unsigned tile = fastdiv(task, a.divRuns), s0 = (task - tile * runs) * 16u;
    unsigned ray;
    const bool valid = tile_ray<4>(a.tiles, tile, (unsigned)p, (unsigned)a.R, ray);
    if (!valid) ray = (unsigned)a.R - 1u;
    const float ox = a.origins[ray * 3u + 0u], oy = a.origins[ray * 3u + 1u], oz = a.origins[ray * 3u + 2u];
    const float dx = a.dirs[ray * 3u + 0u], dy = a.dirs[ray * 3u + 1u], dz = a.dirs[ray * 3u + 2u];
    float sh[4];
    sh4_quarter(q, dx, dy, dz, sh);
    const int erow = a.avg_row >= 0 ? a.avg_row : a.cam_idx[ray];
    const half8 h1 = *reinterpret_cast<const half8*>(a.emb + (size_t)erow * 32 + 8 * q);
    {                                            // lane (p, q): edges 4q .. 4q+4 of ray p's run -> mid-points 4q .. 4q+3
      const float* eb = a.e_bins + (size_t)ray * (S + 1u) + s0 + 4u * (unsigned)q;
      float e0 = eb[0];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float e1 = eb[k + 1];
        t_s[wv][4 * q + k][p] = 0.5f * (e0 + e1);
        e0 = e1;
      }
    }
#pragma unroll 1
    for (int j = 0; j < 16; ++j) {
      const float t = t_s[wv][j][p];
      float x = fmaf(dx, t, ox), y = fmaf(dy, t, oy), z = fmaf(dz, t, oz);
      const bool sel = map_position(x, y, z, a.mode, a.aabb);
      half8 dh[3];
      const half8 xin = field_encode<0>(a.table, x, y, z, sel, q, l_scale, l_res, l_size, l_off, l_hash, dh);
      f32x4 d2, d5;
      field_mlp(wfp, xin, q, sh, h1, d2, d5);
      if (q == 0) {
        den_s[wv][p][j] = sel ? a.avg_density * __expf(d2[0]) : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb_s[wv][p][3 * j + c] = __builtin_amdgcn_rcpf(1.f + __expf(-d5[c]));
      }
    }
    if (valid) {                                 // lane (p, q): quarter q of ray p's density line and of its three colour lines
      const size_t n0 = (size_t)ray * S + s0;
      const float* ds = &den_s[wv][p][4 * q];
      *reinterpret_cast<float4*>(a.density + n0 + 4 * q) = make_float4(ds[0], ds[1], ds[2], ds[3]);
      const float* rs = &rgb_s[wv][p][12 * q];
      float4* dst = reinterpret_cast<float4*>(a.rgb + n0 * 3 + 12 * q);
#pragma unroll
      for (int k = 0; k < 3; ++k) dst[k] = make_float4(rs[4 * k], rs[4 * k + 1], rs[4 * k + 2], rs[4 * k + 3]);
    }
  }
}

// ---- weights + composite, one wavefront per ray (S <= 64) ---------------------------------------------------
struct CompArgs {
  const float* density; const float* rgb; const float* e_bins;
  int R, S, training;
  float* weights; float* rgb_out; float* depth; float* expected; float* acc;
  const unsigned* minmax;        // {bits(min step), bits(max step)} over the whole batch (expected-depth clip range)
  int mm_replicas;               // 1, or MM_REPLICAS partial pairs MM_STRIDE words apart (neraf_pdf_resample_mm): reduced by every wave
};
constexpr int MM_REPLICAS = 64, MM_STRIDE = 64;      // 64 pairs, 256 bytes apart: same-line atomics serialise in their L2 channel

// global min / max of the sample mid-points (positive floats order like their bit patterns)
// seeds the {min, max} pair of steps_minmax_kernel on the device (a host-side 8-byte copy would come from pageable memory:
// host-synchronous, and not capturable into a graph)
// words 2 .. 2 + extra_zero of the scratch are cleared by the same launch: the loss node that follows the composite in a training step
// accumulates its three sums there (one launch fewer than a fill of its own)
__global__ void minmax_seed_kernel(unsigned* __restrict__ mm, int extra_zero) {
  if (blockIdx.x) return;
  if (threadIdx.x < 2) mm[threadIdx.x] = threadIdx.x ? 0u : 0x7f7fffffu;
  else if ((int)threadIdx.x < 2 + extra_zero) mm[threadIdx.x] = 0u;
}

__global__ __launch_bounds__(256) void steps_minmax_kernel(const float* __restrict__ e_bins, int R, int S, unsigned* __restrict__ mm) {
  const int ray = blockIdx.x * 256 + threadIdx.x;
  float lo = 3.0e38f, hi = 0.f;
  if (ray < R) {
    const float* eb = e_bins + (size_t)ray * (S + 1);
    lo = 0.5f * (eb[0] + eb[1]); hi = 0.5f * (eb[S - 1] + eb[S]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
  if ((threadIdx.x & 63) == 0) { atomicMin(mm, __float_as_uint(lo)); atomicMax(mm + 1, __float_as_uint(hi)); }
}

__global__ __launch_bounds__(256) void composite_kernel(CompArgs a) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.R) return;
  const int S = a.S;
  const bool on = lane < S;
  const float* eb = a.e_bins + (size_t)ray * (S + 1);
  const float e0 = on ? eb[lane] : 0.f, e1 = on ? eb[lane + 1] : 0.f;
  const float dd = on ? (e1 - e0) * a.density[(size_t)ray * S + lane] : 0.f;
  const float incl = wave_incl_scan(dd, lane);
  float w = (1.f - __expf(-dd)) * __expf(-wave_excl_from_incl(incl, lane));
  if (!(w == w)) w = 0.f;
  if (!on) w = 0.f;
  if (on && a.weights) a.weights[(size_t)ray * S + lane] = w;
  const float step = 0.5f * (e0 + e1);
  float r = 0.f, g = 0.f, b = 0.f;
  if (on) { const float* c = a.rgb + ((size_t)ray * S + lane) * 3; r = c[0]; g = c[1]; b = c[2]; }
  const float cum = wave_incl_scan(w, lane);
  float sr = w * r, sg = w * g, sb = w * b, sw = w, sd = w * step;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sr += __shfl_xor(sr, o); sg += __shfl_xor(sg, o); sb += __shfl_xor(sb, o); sw += __shfl_xor(sw, o); sd += __shfl_xor(sd, o);
  }
  // background = last sample's colour
  const float lr = __shfl(r, S - 1), lg = __shfl(g, S - 1), lb = __shfl(b, S - 1);
  // median depth: first index with cumulative weight >= 0.5 (searchsorted side='left'), clamped to S-1
  const unsigned long long ge = __ballot(on && cum >= 0.5f);
  const int mi = ge ? (int)__ffsll((long long)ge) - 1 : S - 1;
  const float med = __shfl(step, mi);
  unsigned mm_lo = 0u, mm_hi = 0u;
  if (a.expected) {
    if (a.mm_replicas > 1) {       // lane r reads replica r (its own line, L1-resident after the CU's first wave); min / max of unsigned bit patterns
      mm_lo = a.minmax[lane * MM_STRIDE]; mm_hi = a.minmax[lane * MM_STRIDE + 1];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { mm_lo = min(mm_lo, (unsigned)__shfl_xor((int)mm_lo, o)); mm_hi = max(mm_hi, (unsigned)__shfl_xor((int)mm_hi, o)); }
    } else { mm_lo = a.minmax[0]; mm_hi = a.minmax[1]; }
  }
  if (lane == 0) {
    float cr = sr + lr * (1.f - sw), cg = sg + lg * (1.f - sw), cb = sb + lb * (1.f - sw);
    if (!a.training) { cr = fminf(fmaxf(cr, 0.f), 1.f); cg = fminf(fmaxf(cg, 0.f), 1.f); cb = fminf(fmaxf(cb, 0.f), 1.f); }
    // NeRAFVisionModel.get_outputs clips rgb to [0,1] (NeRAF_model.py:67)
    a.rgb_out[ray * 3 + 0] = fminf(fmaxf(cr, 0.f), 1.f);
    a.rgb_out[ray * 3 + 1] = fminf(fmaxf(cg, 0.f), 1.f);
    a.rgb_out[ray * 3 + 2] = fminf(fmaxf(cb, 0.f), 1.f);
    if (a.depth) a.depth[ray] = med;
    if (a.expected) a.expected[ray] = fminf(fmaxf(sd / (sw + 1e-10f), __uint_as_float(mm_lo)), __uint_as_float(mm_hi));
    if (a.acc) a.acc[ray] = sw;
  }
}

// ---- grid refresh epilogue (NeRAF_model.py:352-357, :386, :395-400) ---------------------------------------
// rgb [ndirs*n,3], density [ndirs*n] (direction-major, as the reference concatenates them :327-333) -> per-cell mean
// over the view directions, alpha = clip(1 - exp(-delta * density), 0, 1), written into channels 0..3 of the
// [7,S,S,S] grid at flat cells [start, start+n).  The refresh window is contiguous in the grid's x-major order
// (coordinates_to_render, :200-202), so the four index_put scatters are four contiguous slab writes.
__global__ __launch_bounds__(256) void grid_refresh_write_kernel(const float* __restrict__ rgb, const float* __restrict__ density,
                                                                int n, int ndirs, float delta, float* __restrict__ grid,
                                                                size_t nvox, size_t start) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float r = 0.f, g = 0.f, b = 0.f, d = 0.f;
  for (int j = 0; j < ndirs; ++j) {
    const size_t k = (size_t)j * n + i;
    r += rgb[k * 3 + 0]; g += rgb[k * 3 + 1]; b += rgb[k * 3 + 2]; d += density[k];
  }
  const float inv = 1.f / (float)ndirs;
  const float alpha = fminf(fmaxf(1.f - __expf(-delta * d * inv), 0.f), 1.f);
  grid[0 * nvox + start + i] = r * inv;
  grid[1 * nvox + start + i] = g * inv;
  grid[2 * nvox + start + i] = b * inv;
  grid[3 * nvox + start + i] = alpha;
}

// differentiable form of the same epilogue: vals [4][n] = (mean rgb, alpha) instead of the grid write, queries laid out
// cell-major (k = i*ndirs + j) or direction-major (k = j*n + i)
__global__ __launch_bounds__(256) void grid_refresh_vals_kernel(const float* __restrict__ rgb, const float* __restrict__ density,
                                                               int n, int ndirs, int cell_major, float delta, float* __restrict__ vals,
                                                               float* __restrict__ grid, size_t nvox, size_t start) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float r = 0.f, g = 0.f, b = 0.f, d = 0.f;
  for (int j = 0; j < ndirs; ++j) {
    const size_t k = cell_major ? (size_t)i * ndirs + j : (size_t)j * n + i;
    r += rgb[k * 3 + 0]; g += rgb[k * 3 + 1]; b += rgb[k * 3 + 2]; d += density[k];
  }
  const float inv = 1.f / (float)ndirs;
  const float alpha = fminf(fmaxf(1.f - expf(-delta * (d * inv)), 0.f), 1.f);
  vals[0 * (size_t)n + i] = r * inv;
  vals[1 * (size_t)n + i] = g * inv;
  vals[2 * (size_t)n + i] = b * inv;
  vals[3 * (size_t)n + i] = alpha;
  if (grid) {      // the slab write of the same values (NeRAF_model.py:395-400) in the same launch
    grid[0 * nvox + start + i] = r * inv;
    grid[1 * nvox + start + i] = g * inv;
    grid[2 * nvox + start + i] = b * inv;
    grid[3 * nvox + start + i] = alpha;
  }
}

// its backward: d rgb = dvals[0..2] / ndirs for every direction; d density = dvals[3] * delta * exp(-delta * mean) / ndirs where
// the clip is inactive (0 < alpha < 1), else 0
__global__ __launch_bounds__(256) void grid_refresh_vals_bwd_kernel(const float* __restrict__ dvals, const float* __restrict__ density,
                                                                   int n, int ndirs, int cell_major, float delta,
                                                                   float* __restrict__ d_rgb, float* __restrict__ d_density) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float d = 0.f;
  for (int j = 0; j < ndirs; ++j) d += density[cell_major ? (size_t)i * ndirs + j : (size_t)j * n + i];
  const float inv = 1.f / (float)ndirs;
  const float e = expf(-delta * (d * inv));
  const float alpha = fminf(fmaxf(1.f - e, 0.f), 1.f);
  const float dd = (alpha > 0.f && alpha < 1.f) ? dvals[3 * (size_t)n + i] * delta * e * inv : 0.f;
  const float dr = dvals[i] * inv, dg = dvals[(size_t)n + i] * inv, db = dvals[2 * (size_t)n + i] * inv;
  for (int j = 0; j < ndirs; ++j) {
    const size_t k = cell_major ? (size_t)i * ndirs + j : (size_t)j * n + i;
    d_rgb[k * 3 + 0] = dr; d_rgb[k * 3 + 1] = dg; d_rgb[k * 3 + 2] = db;
    d_density[k] = dd;
  }
}

// world positions of the refresh queries, cell-major: out[i*ndirs + j] = coords[i] * len + lo  (NeRAF_model.py:315, :327-333)
__global__ __launch_bounds__(256) void refresh_origins_kernel(const float* __restrict__ coords, int n, int ndirs, float lx, float ly, float lz,
                                                             float ox, float oy, float oz, float* __restrict__ out) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= (size_t)n * ndirs) return;
  const size_t i = k / ndirs;
  // separately rounded multiply and add, as the reference's `coords * lengths + aabb[0]` evaluates
  out[k * 3 + 0] = __fadd_rn(__fmul_rn(coords[i * 3 + 0], lx), ox);
  out[k * 3 + 1] = __fadd_rn(__fmul_rn(coords[i * 3 + 1], ly), oy);
  out[k * 3 + 2] = __fadd_rn(__fmul_rn(coords[i * 3 + 2], lz), oz);
}

}  // namespace

// =================================================================================================
extern "C" int neraf_grid_refresh_vals(neraf_ctx* ctx, const float* rgb, const float* density, int n, int ndirs, int cell_major,
                                       float delta, float* vals, float* grid, size_t nvox, size_t start, neraf_stream_t stream) {
  if (!rgb || !density || !vals || n <= 0 || ndirs <= 0 || (grid && start + (size_t)n > nvox))
    return neraf_fail(ctx, NERAF_EINVAL, "grid_refresh_vals: bad arguments");
  hipLaunchKernelGGL(grid_refresh_vals_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rgb, density, n, ndirs,
                     cell_major, delta, vals, grid, nvox, start);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_grid_refresh_vals_bwd(neraf_ctx* ctx, const float* dvals, const float* density, int n, int ndirs, int cell_major,
                                           float delta, float* d_rgb, float* d_density, neraf_stream_t stream) {
  if (!dvals || !density || !d_rgb || !d_density || n <= 0 || ndirs <= 0)
    return neraf_fail(ctx, NERAF_EINVAL, "grid_refresh_vals_bwd: bad arguments");
  hipLaunchKernelGGL(grid_refresh_vals_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dvals, density, n, ndirs,
                     cell_major, delta, d_rgb, d_density);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_refresh_origins(neraf_ctx* ctx, const float* coords, int n, int ndirs, const float* aabb_host, float* out,
                                     neraf_stream_t stream) {
  if (!coords || !aabb_host || !out || n <= 0 || ndirs <= 0) return neraf_fail(ctx, NERAF_EINVAL, "refresh_origins: bad arguments");
  const size_t total = (size_t)n * ndirs;
  hipLaunchKernelGGL(refresh_origins_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, coords, n, ndirs,
                     aabb_host[3] - aabb_host[0], aabb_host[4] - aabb_host[1], aabb_host[5] - aabb_host[2], aabb_host[0], aabb_host[1],
                     aabb_host[2], out);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_grid_refresh_write(neraf_ctx* ctx, const float* rgb, const float* density, int n, int ndirs, float delta,
                                        float* grid, size_t nvox, size_t start, neraf_stream_t stream) {
  if (!rgb || !density || !grid || n <= 0 || ndirs <= 0 || start + (size_t)n > nvox)
    return neraf_fail(ctx, NERAF_EINVAL, "grid_refresh_write: bad arguments");
  hipLaunchKernelGGL(grid_refresh_write_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rgb, density, n, ndirs,
                     delta, grid, nvox, start);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

// Host-side evaluation of the kernels' division-by-invariant (make_fastdiv + the arithmetic of fastdiv, with the multiply-high as a
// 64-bit product): lets a CPU test check the Granlund-Montgomery constants against n / d for any 32-bit n.
extern "C" uint32_t neraf_debug_fastdiv(uint32_t n, uint32_t d) {
  if (d == 0) return 0xFFFFFFFFu;
  const FastDiv f = make_fastdiv(d);
  const uint32_t t = (uint32_t)(((uint64_t)f.m * (uint64_t)n) >> 32);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

extern "C" int neraf_grid_layout(const neraf_grid_desc* g, float* scales, int* resolutions, uint32_t* sizes,
                                 uint32_t* offsets) {
  GridLayout L;
  if (int e = make_grid_layout(g, &L)) return e;
  for (int l = 0; l < L.n_levels; ++l) {
    if (scales) scales[l] = L.scale[l];
    if (resolutions) resolutions[l] = L.res[l];
    if (sizes) sizes[l] = L.size[l];
    if (offsets) offsets[l] = L.offset[l];
  }
  if (offsets) offsets[L.n_levels] = L.offset[L.n_levels];
  return NERAF_OK;
}

extern "C" int neraf_hash_encode(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const float* x01, long long n_points,
                                 float* enc, neraf_stream_t stream) {
  GridLayout L;
  if (make_grid_layout(g, &L)) return neraf_fail(ctx, NERAF_EINVAL, "hash_encode: bad grid descriptor");
  if (!table_f16 || !x01 || !enc || n_points <= 0) return neraf_fail(ctx, NERAF_EINVAL, "hash_encode: bad arguments");
  const long total = (long)n_points * L.n_levels;
  hipLaunchKernelGGL(hash_encode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, L,
                     (const unsigned*)table_f16, x01, (long)n_points, enc);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_sample_uniform(neraf_ctx* ctx, int R, int S, float near, float far, const float* jitter, uint64_t jitter_seed,
                                    float* s_bins, float* e_bins, neraf_stream_t stream) {
  if (R <= 0 || S <= 0 || !s_bins || !e_bins) return neraf_fail(ctx, NERAF_EINVAL, "sample_uniform: bad arguments");
  const long n = (long)R * (S + 1);
  hipLaunchKernelGGL(sample_uniform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, R, S, near, far,
                     jitter, (unsigned long long)jitter_seed, s_bins, e_bins);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_proposal_density(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* mlp_f16,
                                      const float* origins, const float* dirs, const float* e_bins, int R, int S,
                                      float avg_density, int coherent_rays, float* density, neraf_stream_t stream) {
  return neraf_proposal_density_ex(ctx, g, table_f16, mlp_f16, origins, dirs, e_bins, (int64_t)S + 1, R, S, avg_density, coherent_rays, density, stream);
}

extern "C" int neraf_proposal_density_ex(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* mlp_f16,
                                         const float* origins, const float* dirs, const float* e_bins, int64_t e_row_stride, int R, int S,
                                         float avg_density, int coherent_rays, float* density, neraf_stream_t stream) {
  PropArgs a{};
  if (e_row_stride != 0 && e_row_stride != (int64_t)S + 1) return neraf_fail(ctx, NERAF_EINVAL, "proposal_density: e_row_stride is S + 1 or 0");
  a.e_stride = (unsigned)e_row_stride;
  if (make_grid_layout(g, &a.g) || a.g.n_levels > 8) return neraf_fail(ctx, NERAF_EINVAL, "proposal_density: bad grid (<= 8 levels)");
  if (R <= 0 || S <= 0 || !table_f16 || !mlp_f16 || !origins || !dirs || !e_bins || !density)
    return neraf_fail(ctx, NERAF_EINVAL, "proposal_density: bad arguments");
  a.table = (const unsigned*)table_f16; a.w = (const half_t*)mlp_f16; a.origins = origins; a.dirs = dirs; a.e_bins = e_bins;
  a.R = R; a.S = S; a.avg_density = avg_density; a.density = density;
  const long n = (long)R * S;
  ProfScope prof(ctx, (hipStream_t)stream, PROF_PROP_DENSITY, (double)n * a.g.n_levels * 8 * 4);   // gathered table bytes
  if (coherent_rays > 1 && (coherent_rays % 8 || R % coherent_rays))
    return neraf_fail(ctx, NERAF_EINVAL, "proposal_density: coherent_rays = image width needs width % 8 == 0 and whole rows");
  if (n + 256 >= (1L << 31) || (long)(R + 64) * (S + 1) >= (1L << 31)) return neraf_fail(ctx, NERAF_EINVAL, "proposal_density: R * S must stay below 2^31");
  a.divS = make_fastdiv((unsigned)S);
  if (coherent_rays && S % 16 == 0) {          // frame form: (tile of 64 rays, run of 16 sample indices) per wave
    a.tiles = make_ray_tiles<8>(R, coherent_rays);
    a.divRuns = make_fastdiv((unsigned)(S / 16));
    const long tasks = (long)a.tiles.n_tiles * (S / 16);
    if (a.g.n_levels == 5) hipLaunchKernelGGL(proposal_density_frame_kernel<5>, dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(proposal_density_frame_kernel<8>, dim3((unsigned)((tasks + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  } else if (a.g.n_levels == 5) hipLaunchKernelGGL(proposal_density_kernel<5>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(proposal_density_kernel<8>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_pdf_resample(neraf_ctx* ctx, const float* density, const float* s_bins, const float* e_bins, int R, int S,
                                  float anneal, const float* jitter, uint64_t jitter_seed, int n_new, float near, float far,
                                  float* weights, float* s_new, float* e_new, neraf_stream_t stream) {
  return neraf_pdf_resample_ex(ctx, density, s_bins, e_bins, (int64_t)S + 1, R, S, anneal, jitter, jitter_seed, n_new, near, far, weights, s_new, e_new, stream);
}

extern "C" int neraf_pdf_resample_ex(neraf_ctx* ctx, const float* density, const float* s_bins, const float* e_bins, int64_t bins_row_stride,
                                     int R, int S, float anneal, const float* jitter, uint64_t jitter_seed, int n_new, float near, float far,
                                     float* weights, float* s_new, float* e_new, neraf_stream_t stream) {
  if (R <= 0 || S <= 0 || S > PDF_MAX_S || n_new <= 0 || !density || !s_bins || !e_bins || !s_new || !e_new)
    return neraf_fail(ctx, NERAF_EINVAL, "pdf_resample: bad arguments (S <= 256)");
  if (bins_row_stride != 0 && bins_row_stride != (int64_t)S + 1) return neraf_fail(ctx, NERAF_EINVAL, "pdf_resample: bins_row_stride is S + 1 or 0");
  return neraf_pdf_resample_mm(ctx, density, s_bins, e_bins, bins_row_stride, R, S, anneal, jitter, jitter_seed, n_new, near, far, weights, s_new, e_new,
                               nullptr, 0, 0, stream);
}

extern "C" int neraf_pdf_resample_mm(neraf_ctx* ctx, const float* density, const float* s_bins, const float* e_bins, int64_t bins_row_stride,
                                     int R, int S, float anneal, const float* jitter, uint64_t jitter_seed, int n_new, float near, float far,
                                     float* weights, float* s_new, float* e_new, void* minmax_scratch, size_t scratch_bytes, int minmax_mode,
                                     neraf_stream_t stream) {
  if (R <= 0 || S <= 0 || S > PDF_MAX_S || n_new <= 0 || !density || !s_bins || !e_bins || !s_new || !e_new)
    return neraf_fail(ctx, NERAF_EINVAL, "pdf_resample: bad arguments (S <= 256)");
  if (bins_row_stride != 0 && bins_row_stride != (int64_t)S + 1) return neraf_fail(ctx, NERAF_EINVAL, "pdf_resample: bins_row_stride is S + 1 or 0");
  constexpr size_t kMmBytes = (size_t)MM_REPLICAS * MM_STRIDE * 4;
  if (minmax_mode < 0 || minmax_mode > 2 || (minmax_mode && (!minmax_scratch || scratch_bytes < kMmBytes || scratch_bytes > kMmBytes + 4 * 60 || (scratch_bytes & 3))) ||
      (minmax_mode == 2 && (n_new + 1 > 64 || n_new < 2)))
    return neraf_fail(ctx, NERAF_EINVAL, "pdf_resample_mm: mode 0 / 1 (seed) / 2 (accumulate, 2 <= n_new <= 63), 16384..16624 scratch bytes");
  PdfArgs a{density, s_bins, e_bins, R, S, anneal, jitter, n_new, near, far, weights, s_new, e_new, (unsigned long long)jitter_seed, (size_t)bins_row_stride,
            (unsigned*)minmax_scratch, minmax_mode, minmax_mode == 1 ? (int)((scratch_bytes - kMmBytes) / 4) : 0};
  hipLaunchKernelGGL(pdf_resample_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

static int field_query_impl(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                            const void* emb_f16, const float* origins, const float* dirs, const float* e_bins,
                            const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host, float avg_density,
                            int avg_row, int coherent_rays, float* rgb, float* density, void* enc_out, void* denc_out,
                            neraf_stream_t stream);

extern "C" int neraf_field_query(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                                 const void* emb_f16, const float* origins, const float* dirs, const float* e_bins,
                                 const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host, float avg_density,
                                 int avg_row, int coherent_rays, float* rgb, float* density, neraf_stream_t stream) {
  return field_query_impl(ctx, g, table_f16, wfrag_f16, emb_f16, origins, dirs, e_bins, cam_idx, R, S, mode, aabb_host, avg_density,
                          avg_row, coherent_rays, rgb, density, nullptr, nullptr, stream);
}

extern "C" int neraf_field_query_train(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                                       const void* emb_f16, const float* origins, const float* dirs, const float* e_bins,
                                       const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host, float avg_density,
                                       int avg_row, float* rgb, float* density, void* enc_out, void* denc_out, neraf_stream_t stream) {
  if (!enc_out) return neraf_fail(ctx, NERAF_EINVAL, "field_query_train: enc_out required");
  return field_query_impl(ctx, g, table_f16, wfrag_f16, emb_f16, origins, dirs, e_bins, cam_idx, R, S, mode, aabb_host, avg_density,
                          avg_row, 0, rgb, density, enc_out, denc_out, stream);
}

static int field_query_impl(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                            const void* emb_f16, const float* origins, const float* dirs, const float* e_bins,
                            const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host, float avg_density,
                            int avg_row, int coherent_rays, float* rgb, float* density, void* enc_out, void* denc_out,
                            neraf_stream_t stream) {
  FieldArgs a{};
  if (make_grid_layout(g, &a.g) || a.g.n_levels != 16) return neraf_fail(ctx, NERAF_EINVAL, "field_query: grid must have 16 levels");
  if (R <= 0 || S <= 0 || !table_f16 || !wfrag_f16 || !emb_f16 || !origins || !dirs || !e_bins || !rgb || !density ||
      (avg_row < 0 && !cam_idx) || (mode != 0 && !aabb_host))
    return neraf_fail(ctx, NERAF_EINVAL, "field_query: bad arguments");
  a.table = (const unsigned*)table_f16; a.wfrag = (const half8*)wfrag_f16; a.emb = (const half_t*)emb_f16;
  a.origins = origins; a.dirs = dirs; a.e_bins = e_bins; a.cam_idx = cam_idx; a.R = R; a.S = S; a.mode = mode;
  for (int i = 0; i < 6; ++i) a.aabb[i] = aabb_host ? aabb_host[i] : 0.f;
  a.avg_density = avg_density; a.avg_row = avg_row; a.rgb = rgb; a.density = density;
  const long n = (long)R * S;
  if ((long)(R + 16) * (S + 1) >= (1L << 31)) return neraf_fail(ctx, NERAF_EINVAL, "field_query: R * S must stay below 2^31");
  a.divS = make_fastdiv((unsigned)S);
  if (coherent_rays > 1 && (coherent_rays % 8 || R % coherent_rays))
    return neraf_fail(ctx, NERAF_EINVAL, "field_query: coherent_rays = image width needs width % 8 == 0 and whole rows");
  const bool frame = coherent_rays && S % 16 == 0 && !enc_out && !denc_out;
  if (frame) { a.tiles = make_ray_tiles<4>(R, coherent_rays); a.divRuns = make_fastdiv((unsigned)(S / 16)); }
  const long groups = frame ? (long)a.tiles.n_tiles * (S / 16) : (n + 15) / 16;      // wave tasks
  long blocks = (groups + 3) / 4;
  const long cap = (long)(ctx ? ctx->num_cus : 256) * 8;
  if (blocks > cap) blocks = cap;
  ProfScope prof(ctx, (hipStream_t)stream, PROF_FIELD_QUERY, (double)n * 16 * 8 * 4);   // gathered table bytes
  a.enc_out = (half_t*)enc_out; a.denc_out = (half_t*)denc_out;
  if (frame) hipLaunchKernelGGL(field_query_frame_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else if (denc_out) hipLaunchKernelGGL(field_query_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else if (enc_out) hipLaunchKernelGGL(field_query_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(field_query_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_composite_mm(neraf_ctx* ctx, const float* density, const float* rgb, const float* e_bins, int R, int S,
                                  int training, float* weights, float* rgb_out, float* depth, float* expected, float* acc,
                                  const void* minmax, neraf_stream_t stream) {
  if (R <= 0 || S <= 0 || S > 64 || !density || !rgb || !e_bins || !rgb_out || (expected && !minmax))
    return neraf_fail(ctx, NERAF_EINVAL, "composite_mm: bad arguments (S <= 64; the {min, max} pair of neraf_pdf_resample_mm for expected depth)");
  CompArgs a{density, rgb, e_bins, R, S, training, weights, rgb_out, depth, expected, acc, (const unsigned*)minmax, MM_REPLICAS};
  hipLaunchKernelGGL(composite_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_composite(neraf_ctx* ctx, const float* density, const float* rgb, const float* e_bins, int R, int S,
                               int training, float* weights, float* rgb_out, float* depth, float* expected, float* acc,
                               void* scratch, size_t scratch_bytes, neraf_stream_t stream) {
  if (R <= 0 || S <= 0 || S > 64 || !density || !rgb || !e_bins || !rgb_out || (expected && (!scratch || scratch_bytes < 8)) ||
      scratch_bytes > 8 + 4 * 60 || (scratch_bytes & 3))
    return neraf_fail(ctx, NERAF_EINVAL, "composite: bad arguments (S <= 64; >= 8 scratch bytes needed for expected depth, <= 248)");
  unsigned* mm = (unsigned*)scratch;
  if (expected) {
    hipLaunchKernelGGL(minmax_seed_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, mm, (int)((scratch_bytes - 8) / 4));
    hipLaunchKernelGGL(steps_minmax_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, e_bins, R, S, mm);
  }
  CompArgs a{density, rgb, e_bins, R, S, training, weights, rgb_out, depth, expected, acc, mm, 1};
  hipLaunchKernelGGL(composite_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

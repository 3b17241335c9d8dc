// Radiance half, training: losses and backward.
//   render_loss kernel      : rgb MSE (on the clipped colour, NeRAF_model.py:67) + distortion loss on the final samples,
//                             and their gradients w.r.t. per-sample colour and density (through composite + get_weights)
//   interlevel kernel       : proposal "outer" histogram loss and its gradient w.r.t. the proposal densities
//   proposal_backward kernel: hash-grid + MLP(2L->16->1) backward (table gradient atomics, block-reduced weight grads)
//   field_backward kernel   : fused nerfacto field backward on MFMA: recompute forward in registers, chain dX through
//                             the transposed weight fragments, scatter hash-grid gradients, reduce appearance-embedding
//                             gradients, and dump per-layer (X, dY) in [feature][point] fp16 so that the five weight
//                             gradients are NT GEMMs (split-K over the ~2e5 points) on gemm_f16.hip.
// These replace autograd through nerfstudio's NerfactoModel losses (rgb MSE, interlevel_loss, distortion_loss) and
// tiny-cuda-nn's backward kernels [NS/TCNN-recall; oracle/vision.py is the unpinned checker].  Gradients do not flow
// through sample positions (PDFSampler detaches its bins) and the final weights are detached inside interlevel_loss.
#include "field_common.h"

namespace {

constexpr int NFRAG = 24;     // forward fragments (see field.hip)
constexpr int NFRAG_B = 28;   // backward: Th2 0-3, Th1 4-11 (ib*2+s), Th0a 12-13, Th0b 14-17 (ib*2+s), Tb1 18-21, Tb0 22-25 (rb*2+s), Th0sh 26-27 (SH inputs, ray gradient)

// ------------------------------------------------------------------------------------------------------------
struct RenderLossArgs {
  const float* density; const float* rgb_s; const float* e_bins; const float* s_bins; const float* gt;
  int R, S; float dist_mult;
  const float* up;         // device [3]: upstream gradients of {rgb_loss, interlevel_loss, distortion_loss}; null = values only
  float* d_rgb_s; float* d_density;     // outputs (may be null when up == null)
  float* d_density_dist;   // up == null and d_rgb_s != null: UNIT gradients -- d_rgb_s, d_density for d rgb_loss = 1 and, separately,
                           // d_density_dist for d distortion_loss = 1 (the caller combines them with the real upstream scalars)
  float* sums;             // device [>=2]: += sum (rgb-gt)^2 , += sum_rays distortion
};

__global__ __launch_bounds__(256) void render_loss_kernel(RenderLossArgs a) {
  const int lane = threadIdx.x & 63;
  int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool active = ray < a.R;
  if (!active) ray = a.R - 1;
  const int S = a.S;
  const bool on = lane < S;
  const float* eb = a.e_bins + (size_t)ray * (S + 1);
  const float* sb = a.s_bins + (size_t)ray * (S + 1);
  const float e0 = on ? eb[lane] : 0.f, e1 = on ? eb[lane + 1] : 0.f;
  const float s0 = on ? sb[lane] : 0.f, s1 = on ? sb[lane + 1] : 0.f;
  const float de = e1 - e0;
  const float dd = on ? de * a.density[(size_t)ray * S + lane] : 0.f;
  const float incl = wave_incl_scan(dd, lane);
  const float Ti = __expf(-wave_excl_from_incl(incl, lane)), ex = __expf(-dd);
  float w = (1.f - ex) * Ti;
  const bool bad = !(w == w);
  if (bad || !on) w = 0.f;
  float c[3] = {0.f, 0.f, 0.f};
  if (on) { const float* p = a.rgb_s + ((size_t)ray * S + lane) * 3; c[0] = p[0]; c[1] = p[1]; c[2] = p[2]; }
  float sw = w, sc[3] = {w * c[0], w * c[1], w * c[2]};
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sw += __shfl_xor(sw, o);
#pragma unroll
    for (int k = 0; k < 3; ++k) sc[k] += __shfl_xor(sc[k], o);
  }
  float cl[3], out[3], g[3], l_rgb = 0.f;
  const float inv_n = 1.f / (3.f * (float)a.R);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    cl[k] = __shfl(c[k], S - 1);
    out[k] = sc[k] + cl[k] * (1.f - sw);
    const float oc = fminf(fmaxf(out[k], 0.f), 1.f);
    const float diff = oc - a.gt[ray * 3 + k];
    l_rgb += diff * diff;
    g[k] = (out[k] > 0.f && out[k] < 1.f) ? 2.f * diff * inv_n : 0.f;     // d mean((clip(rgb)-gt)^2) / d rgb
  }
  // distortion (lossfun_distortion): sum_i w_i sum_j w_j |u_i-u_j| + sum_i w_i^2 (s_{i+1}-s_i)/3
  const float u = 0.5f * (s0 + s1), ds = s1 - s0;
  // inner_i = sum_j w_j |u_i - u_j|: the midpoints are sorted along the ray, so it splits into prefix / suffix sums of w and w*u
  // (two wave scans instead of an S-step shuffle loop)
  const float wu = w * u;
  const float w_inc = wave_incl_scan(w, lane), wu_inc = wave_incl_scan(wu, lane);
  const float w_tot = __shfl(w_inc, 63), wu_tot = __shfl(wu_inc, 63);
  const float w_lt = w_inc - w, wu_lt = wu_inc - wu, w_gt = w_tot - w_inc, wu_gt = wu_tot - wu_inc;
  const float inner = on ? (u * w_lt - wu_lt) + (wu_gt - u * w_gt) : 0.f;
  float l_dist = on ? (w * inner + w * w * ds * (1.f / 3.f)) : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) l_dist += __shfl_xor(l_dist, o);
  {
    // one atomic per workgroup: 4096 per-ray atomics on one cache line serialise at ~3.6 ns each (30 of this kernel's 110 us)
    __shared__ float red[4][2];
    if (lane == 0) { red[threadIdx.x >> 6][0] = active ? l_rgb : 0.f; red[threadIdx.x >> 6][1] = active ? l_dist : 0.f; }
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(a.sums + 0, red[0][0] + red[1][0] + red[2][0] + red[3][0]);
      atomicAdd(a.sums + 1, red[0][1] + red[1][1] + red[2][1] + red[3][1]);
    }
  }
  const bool unit = !a.up && a.d_rgb_s;
  if (!a.up && !unit) return;
  const float up_rgb = unit ? 1.f : a.up[0], up_dist = (unit ? 1.f : a.up[2]) * a.dist_mult / (float)a.R;
  // d loss / d w_i, kept apart for the two losses in unit mode
  float gw_r = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) gw_r += up_rgb * g[k] * (c[k] - cl[k]);
  float gw_d = up_dist * (2.f * inner + 2.f * w * ds * (1.f / 3.f));
  if (!on || bad) { gw_r = 0.f; gw_d = 0.f; }
  // d loss / d colour_i
  if (on && active) {
    float* o = a.d_rgb_s + ((size_t)ray * S + lane) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = up_rgb * g[k] * (w + (lane == S - 1 ? (1.f - sw) : 0.f));
  }
  // get_weights backward (linear in gw): d dd_i = gw_i T_i exp(-dd_i) - sum_{k>i} gw_k w_k ; d sigma_i = delta_i d dd_i
  auto weights_bwd = [&](float gw) {
    return de * (gw * Ti * ex - wave_suffix_excl_scan(gw * w, lane));
  };
  if (unit) {
    const float dr = weights_bwd(gw_r), dd2 = weights_bwd(gw_d);
    if (on && active) { a.d_density[(size_t)ray * S + lane] = dr; a.d_density_dist[(size_t)ray * S + lane] = dd2; }
  } else {
    const float dsum = weights_bwd(gw_r + gw_d);
    if (on && active) a.d_density[(size_t)ray * S + lane] = dsum;
  }
}

// ------------------------------------------------------------------------------------------------------------
struct InterlevelArgs {
  const float* c_bins; const float* w_fine; int S2;                      // final samples (detached)
  const float* p_bins; const float* p_ebins; const float* p_density; int Sp;   // proposal level
  int R; float mult;
  const float* up;            // device [3] or null (values only)
  float* d_density;           // [R,Sp]
  float* sums;                // sums[2] += sum_i lossfun_outer
};

constexpr int IL_MAX_SP = 256, IL_MAX_S2 = 64;

__global__ __launch_bounds__(256) void interlevel_kernel(InterlevelArgs a) {
  __shared__ float sp_s[4][IL_MAX_SP + 1], cy_s[4][IL_MAX_SP + 1], diff_s[4][IL_MAX_SP + 2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int ray = blockIdx.x * 4 + wv;
  const bool active = ray < a.R;
  if (!active) ray = a.R - 1;
  const int Sp = a.Sp, S2 = a.S2;
  const int per = (Sp + 63) / 64;
  float* sp = sp_s[wv]; float* cy = cy_s[wv]; float* diff = diff_s[wv];
  const float* pb = a.p_bins + (size_t)ray * (Sp + 1);
  const float* pe = a.p_ebins + (size_t)ray * (Sp + 1);
  const float* pd = a.p_density + (size_t)ray * Sp;
  for (int i = lane; i <= Sp; i += 64) { sp[i] = pb[i]; diff[i] = 0.f; }
  if (lane == 0) diff[Sp + 1] = 0.f;
  // proposal weights (recomputed) and their inclusive cumsum
  float dd[4], loc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = lane * per + k;
    dd[k] = (k < per && i < Sp) ? (pe[i + 1] - pe[i]) * pd[i] : 0.f;
    loc += dd[k];
  }
  const float incl = wave_incl_scan(loc, lane);
  float run = wave_excl_from_incl(incl, lane), wp[4], Tk[4], exk[4], wl = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    Tk[k] = __expf(-run); exk[k] = __expf(-dd[k]);
    float w = (1.f - exk[k]) * Tk[k];
    if (!(w == w)) w = 0.f;
    const int i = lane * per + k;
    if (!(k < per && i < Sp)) w = 0.f;
    run += dd[k];
    wp[k] = w; wl += w;
  }
  const float inclw = wave_incl_scan(wl, lane);
  float cw = inclw - wl;
  if (lane == 0) cy[0] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = lane * per + k;
    if (k < per && i < Sp) { cw += wp[k]; cy[i + 1] = cw; }
  }
  __syncthreads();
  // one fine interval per lane
  float loss = 0.f;
  if (lane < S2) {
    const float* cb = a.c_bins + (size_t)ray * (S2 + 1);
    const float c0 = cb[lane], c1 = cb[lane + 1], w = a.w_fine[(size_t)ray * S2 + lane];
    // idx_lo = searchsorted(starts = sp[0..Sp-1], c0, right) - 1, clamped to [0, Sp-1]
    int lo = 0, hi = Sp;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sp[mid] > c0) hi = mid; else lo = mid + 1; }
    const int ilo = min(max(lo - 1, 0), Sp - 1);
    // idx_hi = searchsorted(ends = sp[1..Sp], c1, right), clamped to [0, Sp-1]
    lo = 0; hi = Sp;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sp[mid + 1] > c1) hi = mid; else lo = mid + 1; }
    const int ihi = min(max(lo, 0), Sp - 1);
    const float w_outer = cy[ihi + 1] - cy[ilo];
    const float ex = fmaxf(w - w_outer, 0.f);
    loss = ex * ex / (w + 1e-7f);
    if (a.up) {   // (an empty range ihi < ilo yields the same signed contributions as torch's cy[hi+1] - cy[lo])
      const float h = -2.f * ex / (w + 1e-7f);
      atomicAdd(&diff[ilo], h);
      atomicAdd(&diff[ihi + 1], -h);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o);
  {
    __shared__ float redl[4];
    if (lane == 0) redl[wv] = active ? loss : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.sums + 2, redl[0] + redl[1] + redl[2] + redl[3]);
  }
  if (!a.up) return;
  __syncthreads();
  // d loss / d wp_k = prefix sum of the difference array; then get_weights backward
  const float scale = a.up[1] * a.mult / ((float)a.R * (float)S2);
  float dl = 0.f, dv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int i = lane * per + k; dv[k] = (k < per && i < Sp) ? diff[i] : 0.f; dl += dv[k]; }
  const float incd = wave_incl_scan(dl, lane);
  float rund = incd - dl, gw[4], gwl = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { rund += dv[k]; gw[k] = rund * scale; gwl += gw[k] * wp[k]; }
  float later = wave_suffix_excl_scan(gwl, lane);       // sum over the samples AFTER this one: later lanes, then later k of this lane
#pragma unroll
  for (int k = 3; k >= 0; --k) {
    const int i = lane * per + k;
    const float ddd = gw[k] * Tk[k] * exk[k] - later;
    later += gw[k] * wp[k];
    if (k < per && i < Sp && active) a.d_density[(size_t)ray * Sp + i] = (pe[i + 1] - pe[i]) * ddd;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Deterministic mode: d_ray[ray] += sum_s g6[ray][s] with one wavefront per ray, lanes striding over the samples and a fixed
// xor-tree over the lanes -- the same bits whatever order the producing waves ran in.
__global__ __launch_bounds__(256) void ray_fold_kernel(const float* __restrict__ g6, int R, int S, float* __restrict__ d_ray) {
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ray >= R) return;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = lane; s < S; s += 64) {
    const float* src = g6 + ((size_t)ray * S + s) * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] += src[k];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
  }
  if (lane < 6) {
    float v = acc[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) v = lane == k ? acc[k] : v;
    d_ray[(size_t)ray * 6 + lane] += v;
  }
}

struct PropBwdArgs {
  GridLayout g;
  const unsigned* table; const half_t* w;
  const float* origins; const float* dirs; const float* e_bins; const float* d_density;
  int R, S; float avg_density;
  float* table_grad;      // fp32 [rows,2], accumulated (caller zeroes)
  float* w_grad;          // fp32 [16*16 + 16], accumulated
  float* w_part;          // optional fp32 [blocks][272]: per-workgroup partials (folded by prop_wgrad_fold_kernel) instead of
                          // 272 atomics per workgroup on nine cache lines (~220 us of same-line serialisation at 2048 workgroups)
  float* d_ray;           // optional fp32 [R][6], ACCUMULATED: d loss / d (ray origin, ray direction)
  float* g6_out;          // deterministic mode (with d_ray): fp32 [N][6], the per-SAMPLE contributions, STORED; ray_fold_kernel adds each
                          // ray's S rows in a fixed order instead of waves adding to d_ray with float atomics in arrival order
  // packed two-pass table gradient (all four non-null): this kernel only STORES the per-sample encoding gradient, fp32 pairs
  // [level][npad], and each level's gradient mass sum_samples max(|g0|, |g1|) per workgroup; field_finalize_kernel turns the masses
  // into the power-of-two fixed-point scales, field_scatter_kernel adds both features of a table entry with ONE 64-bit integer
  // atomic (half the atomics of the fp32 pair path, and integer adds commute: the gradient becomes bit-reproducible),
  // field_unpack_grad_kernel converts in place
  float2* d32; long npad; float* t_part; float* one2;
};

__global__ __launch_bounds__(256) void proposal_backward_kernel(PropBwdArgs a) {
  __shared__ float w0[16][16], w1[16];
  __shared__ float s_dh[4][64][17], s_enc[4][64][17], s_hr[4][64][17];   // per wave: [point][16] (+1 pad)
  __shared__ float acc[4][16 * 16 + 16];      // per wave: every lane owns its entries, so no atomics and a fixed summation order
  for (int i = threadIdx.x; i < 256; i += 256) w0[i >> 4][i & 15] = (float)a.w[i];
  if (threadIdx.x < 16) w1[threadIdx.x] = (float)a.w[256 + threadIdx.x];
  for (int i = threadIdx.x; i < 4 * 272; i += 256) (&acc[0][0])[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long N = (long)a.R * a.S;
  const long nchunks = (N + 255) / 256;
  float tmass[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (a.d32 && blockIdx.x == 0 && threadIdx.x < 2) a.one2[threadIdx.x] = 1.f;
  for (long chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const long idx = chunk * 256 + threadIdx.x;
    const bool valid = idx < N;
    float enc[16], hpre[16], dh[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { enc[k] = 0.f; dh[k] = 0.f; hpre[k] = 0.f; }
    float x = 0.f, y = 0.f, z = 0.f, gs = 0.f;
    float xr = 0.f, yr = 0.f, zr = 0.f, tmid = 0.f; int ray_id = -1; bool sel_pt = false;
    float rgx = 0.f, rgy = 0.f, rgz = 0.f;                 // ray gradient: d loss / d mapped position
    if (valid) {
      const int ray = (int)(idx / a.S), s = (int)(idx % a.S);
      const float t = 0.5f * (a.e_bins[(size_t)ray * (a.S + 1) + s] + a.e_bins[(size_t)ray * (a.S + 1) + s + 1]);
      x = fmaf(a.dirs[ray * 3 + 0], t, a.origins[ray * 3 + 0]);
      y = fmaf(a.dirs[ray * 3 + 1], t, a.origins[ray * 3 + 1]);
      z = fmaf(a.dirs[ray * 3 + 2], t, a.origins[ray * 3 + 2]);
      xr = x; yr = y; zr = z; tmid = t; ray_id = ray;
      const bool sel = map_position(x, y, z, 0, nullptr);
      sel_pt = sel;
#pragma unroll
      for (int l = 0; l < 8; ++l)
        if (l < a.g.n_levels) {
          float f0, f1;
          encode_level(a.table, x, y, z, a.g.scale[l], a.g.res[l], a.g.size[l], a.g.offset[l], a.g.hashed[l], f0, f1);
          enc[2 * l] = (float)(half_t)f0; enc[2 * l + 1] = (float)(half_t)f1;
        }
      float out = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float h = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) h = fmaf(w0[j][k], enc[k], h);
        hpre[j] = h;
        out = fmaf(w1[j], fmaxf(h, 0.f), out);
      }
      // sigma = avg * trunc_exp(out) * sel  ->  d out = d sigma * avg * exp(min(out, 15)) * sel
      gs = sel ? a.d_density[idx] * a.avg_density * __expf(fminf(fmaxf(out, -15.f), 15.f)) : 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) dh[j] = hpre[j] > 0.f ? gs * w1[j] : 0.f;
    }
    // table gradients: every lane takes part in the segmented pre-reduction (64 consecutive samples per wave)
#pragma unroll
    for (int l = 0; l < 8; ++l)
      if (l < a.g.n_levels) {
        float g0 = 0.f, g1 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { g0 = fmaf(dh[j], w0[j][2 * l], g0); g1 = fmaf(dh[j], w0[j][2 * l + 1], g1); }
        if (a.d_ray && valid && sel_pt)
          encode_level_dpos(a.table, x, y, z, a.g.scale[l], a.g.res[l], a.g.size[l], a.g.offset[l], a.g.hashed[l], g0, g1, rgx, rgy, rgz);
        if (a.d32) {
          if (valid) { a.d32[(long)l * a.npad + idx] = make_float2(g0, g1); tmass[l] += fmaxf(fabsf(g0), fabsf(g1)); }
        } else {
          scatter_level_seg<64>(a.table_grad, x, y, z, a.g.scale[l], a.g.res[l], a.g.size[l], a.g.offset[l], a.g.hashed[l], g0, g1, lane);
        }
      }
    if (a.d_ray) {
      // camera-pose edge: the position gradient of this sample -> d (origin, direction) of its ray; the 64 samples of a wave lie
      // on one ray unless a ray boundary falls inside it (S = 256 / 96 samples per ray)
      map_position_jt(xr, yr, zr, rgx, rgy, rgz);
      float g6[6] = {rgx, rgy, rgz, tmid * rgx, tmid * rgy, tmid * rgz};
      const int ray0 = __shfl(ray_id, 0);
      if (a.g6_out) {
        if (valid) {
#pragma unroll
          for (int k = 0; k < 6; ++k) a.g6_out[(size_t)idx * 6 + k] = g6[k];
        }
      } else if (__all(ray_id == ray0 || !valid)) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          float v = valid ? g6[k] : 0.f;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
          if (lane == 0 && v != 0.f && ray0 >= 0) atomicAdd(a.d_ray + (size_t)ray0 * 6 + k, v);
        }
      } else if (valid) {
#pragma unroll
        for (int k = 0; k < 6; ++k) if (g6[k] != 0.f) atomicAdd(a.d_ray + (size_t)ray_id * 6 + k, g6[k]);
      }
    }
    // weight gradients: dW0[j][k] = sum_p dh[p][j] enc[p][k] ; dW1[j] = sum_p gs[p] relu(h[p][j]) -- per-wave LDS staging
#pragma unroll
    for (int k = 0; k < 16; ++k) { s_dh[wv][lane][k] = dh[k]; s_enc[wv][lane][k] = enc[k]; s_hr[wv][lane][k] = gs * fmaxf(hpre[k], 0.f); }
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    {
      const int j = lane >> 2, k0 = (lane & 3) * 4;     // 64 lanes x 4 outputs = the 256 entries of dW0
      float s[4] = {0.f, 0.f, 0.f, 0.f};
      for (int p = 0; p < 64; ++p) {
        const float d = s_dh[wv][p][j];
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = fmaf(d, s_enc[wv][p][k0 + q], s[q]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[wv][j * 16 + k0 + q] += s[q];
      if (lane < 16) {
        float t = 0.f;
        for (int p = 0; p < 64; ++p) t += s_hr[wv][p][lane];
        acc[wv][256 + lane] += t;
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < 272; i += 256) {
    const float v = (acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i]);
    if (a.w_part) a.w_part[(size_t)blockIdx.x * 272 + i] = v;
    else atomicAdd(a.w_grad + i, v);
  }
  if (a.d32) {
    // this workgroup's gradient mass per level -> t_part[block][16] (levels >= n_levels: 0), summed by field_finalize_kernel
    __syncthreads();
    float* red = &s_dh[0][0][0];                   // 4 x 64 x 17 floats, free now
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      float v = tmass[l];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) red[wv * 8 + l] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16)
      a.t_part[(size_t)blockIdx.x * 16 + threadIdx.x] = threadIdx.x < 8 ? (red[threadIdx.x] + red[8 + threadIdx.x]) + (red[16 + threadIdx.x] + red[24 + threadIdx.x]) : 0.f;
  }
}

// w_grad[i] += sum over workgroups of part[b][i], i < 272: 34 workgroups of 8 outputs x 32 row slices (five 64-thread workgroups
// walking 2048 rows each were a 60-80 us latency chain)
__global__ __launch_bounds__(256) void prop_wgrad_fold_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ w_grad,
                                                             float* __restrict__ w0_out, float* __restrict__ w1_out) {
  // w0_out / w1_out (both or neither): STORE the result in the parameters' own layouts -- w0 [16][16] and w1 [16][16] whose row 0 is
  // the used output row (rows 1..15 have no gradient: written as zeros) -- instead of adding into the flat w_grad[272]
  __shared__ float red[32][9];
  const int o = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + o;
  float t = 0.f, t1 = 0.f;
  if (i < 272) {
    int b = sl;
    for (; b + 32 < nblocks; b += 64) { t += part[(size_t)b * 272 + i]; t1 += part[(size_t)(b + 32) * 272 + i]; }
    for (; b < nblocks; b += 32) t += part[(size_t)b * 272 + i];
  }
  red[sl][o] = t + t1;
  __syncthreads();
  if (threadIdx.x < 8 && i < 272) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += red[r][threadIdx.x];
    const int k = blockIdx.x * 8 + threadIdx.x;
    if (w0_out) { if (k < 256) w0_out[k] = s; else w1_out[k - 256] = s; }
    else w_grad[k] += s;
  }
  if (w1_out && blockIdx.x == 0 && threadIdx.x >= 16) w1_out[threadIdx.x] = 0.f;      // rows 1..15 of w1 (240 floats)
}

// ------------------------------------------------------------------------------------------------------------
struct FieldBwdArgs {
  GridLayout g;
  const unsigned* table; const half8* wfrag; const half8* wfrag_b; const half_t* emb;
  const float* origins; const float* dirs; const float* e_bins; const int* cam_idx;
  int R, S; int mode; float aabb[6]; float avg_density; int avg_row;
  const float* d_rgb; const float* d_density;      // upstream [N,3], [N]
  const float* scale;                              // device {S, 1/S}: fp16 chain scaling
  float* emb_grad;                                 // fp32 [n_emb,32], accumulated (null in avg_row mode)
  half_t* dump; long npad;                         // fp16 [10][128][npad]: X_b0,dY_b0,X_b1,dY_b1,X_h0,dY_h0,X_h1,dY_h1,X_h2,dY_h2
  unsigned* d_enc;                                 // half2 [16 levels][npad]: scaled gradient w.r.t. the encoding (rows 64.. of slot 0)
  float* t_part;                                   // fp32 [blocks][16]: per-workgroup sums of max(|g0|,|g1|) per level (rows 64.. of slot 1)
  int emb_det;                                     // deterministic mode: e_part / e_part_row are PER 16-POINT GROUP ([npad / 16][32] + row, -1 = none),
                                                   // stored by the group (no atomics), folded per embedding row in group order
  float* e_part; int* e_part_row;                  // fp32 [blocks][32] + row: per-workgroup appearance-embedding gradient of the
                                                   // workgroup's leading embedding row (rows 64.. of slots 2 / 3)
  float* pos;                                      // optional fp32 [3][npad]: mapped sample positions for the owner scatter (slot 4, rows 64..)
  float* d_ray;                                    // optional fp32 [R][6], ACCUMULATED: d loss / d (ray origin, ray direction) -- the camera-pose optimizer's edge
  float* g6_out;                                   // deterministic mode (with d_ray): per-sample contributions fp32 [N][6], see PropBwdArgs
  const half_t* enc_in;                            // optional fp16 [N][32] saved by the forward (neraf_field_query_train): no table walk here
  const half_t* denc_in;                           // optional fp16 [N][4][24]: saved d enc / d position (with d_ray)
};

__device__ __forceinline__ void dump_block(half_t* base, long npad, long n, int row0, const f32x4& v, float m) {
  // rows row0..row0+3 of a [feat][npad] matrix, column n
#pragma unroll
  for (int r = 0; r < 4; ++r) base[(long)(row0 + r) * npad + n] = (half_t)(v[r] * m);
}

template <bool RAYGRAD, bool SAVED>
__global__ __launch_bounds__(256) void field_backward_kernel(FieldBwdArgs a) {
  __shared__ float l_scale[MAX_LEVELS];
  __shared__ int l_res[MAX_LEVELS];
  __shared__ unsigned l_size[MAX_LEVELS], l_off[MAX_LEVELS];
  __shared__ int l_hash[MAX_LEVELS];
  if (threadIdx.x < MAX_LEVELS) {
    const int l = threadIdx.x;
    l_scale[l] = a.g.scale[l]; l_res[l] = a.g.res[l]; l_size[l] = a.g.size[l]; l_off[l] = a.g.offset[l]; l_hash[l] = a.g.hashed[l];
  }
  __syncthreads();
  __shared__ float t_sum[4][16];
  __shared__ float e_acc[32];
  __shared__ int e_blk_s;
  if (threadIdx.x < 32) e_acc[threadIdx.x] = 0.f;
  if (threadIdx.x == 0) {
    long n0 = (long)blockIdx.x * 64; const long Nn = (long)a.R * a.S;
    if (n0 >= Nn) n0 = Nn - 1;
    e_blk_s = (a.avg_row < 0 && a.cam_idx) ? a.cam_idx[n0 / a.S] : -1;
  }
  __syncthreads();
  const int e_blk = e_blk_s;
  const int lane = threadIdx.x & 63;
  const int p = lane & 15, q = lane >> 4;
  // forward fragments in LDS too (24 KiB): held in registers they cost 96 VGPRs of a kernel that then fits one wave per SIMD only
  __shared__ half8 wf_s[NFRAG * 64];
  for (int i = threadIdx.x; i < NFRAG * 64; i += 256) wf_s[i] = a.wfrag[i];
  const half8* wfp = wf_s + lane;
#define wf(f) wfp[(f) * 64]
  // the 26 backward fragments live in LDS (26 KiB): each is read once per 16-point group, and a global (L1/L2) load in front
  // of every MFMA of the chain is a ~500-cycle dependency where the LDS read is ~64
  __shared__ half8 wb_s[NFRAG_B * 64];
  for (int i = threadIdx.x; i < NFRAG_B * 64; i += 256) wb_s[i] = a.wfrag_b[i];
  __syncthreads();
  const half8* wb = wb_s + lane;                 // fragment f at wb[f * 64]
  const float gscale = a.scale[0], inv_gscale = a.scale[1];
  float tl[4] = {0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const long N = (long)a.R * a.S;
  const long ngroups = a.npad / 16;              // covers the zero padding of the dumps as well
  const long gstride = (long)gridDim.x * 4;
  const long MAT = 128 * a.npad;                 // elements per dumped matrix
  for (long grp = (long)blockIdx.x * 4 + (threadIdx.x >> 6); grp < ngroups; grp += gstride) {
    const long ncol = grp * 16 + p;              // dump column
    long n = ncol;
    const bool valid = n < N;
    if (!valid) n = N - 1;
    const float vm = valid ? 1.f : 0.f;
    const int ray = (int)(n / a.S), s = (int)(n % a.S);
    const float t = 0.5f * (a.e_bins[(size_t)ray * (a.S + 1) + s] + a.e_bins[(size_t)ray * (a.S + 1) + s + 1]);
    const float dx = a.dirs[ray * 3 + 0], dy = a.dirs[ray * 3 + 1], dz = a.dirs[ray * 3 + 2];
    float x = fmaf(dx, t, a.origins[ray * 3 + 0]);
    float y = fmaf(dy, t, a.origins[ray * 3 + 1]);
    float z = fmaf(dz, t, a.origins[ray * 3 + 2]);
    const float xr = x, yr = y, zr = z;            // un-mapped position (ray gradient)
    const bool sel = map_position(x, y, z, a.mode, a.aabb);
    if (a.pos && q == 0) { a.pos[ncol] = x; a.pos[a.npad + ncol] = y; a.pos[2 * a.npad + ncol] = z; }
    // ---------------- forward recompute (identical to field_query_kernel) ----------------
    half8 xin;
    float dfe[RAYGRAD ? 4 : 1][2][3];                // RAYGRAD: d enc / d mapped position of this lane's four levels
    if (SAVED) {
      // the forward stored the encoding (and its position derivatives): two to five coalesced 16-byte loads instead of 32 gathers
      // whose latency a one-wave-per-SIMD kernel cannot hide (render batch: 210 -> see DESIGN.md)
      xin = *reinterpret_cast<const half8*>(a.enc_in + ((size_t)n * 4 + q) * 8);
      if (RAYGRAD) {
        half8 dh[3];
        const half8* src = reinterpret_cast<const half8*>(a.denc_in + ((size_t)n * 4 + q) * 24);
        dh[0] = src[0]; dh[1] = src[1]; dh[2] = src[2];
        const half_t* dp = reinterpret_cast<const half_t*>(dh);
#pragma unroll
        for (int li = 0; li < 4; ++li)
#pragma unroll
          for (int k = 0; k < 3; ++k) { dfe[RAYGRAD ? li : 0][0][k] = (float)dp[li * 6 + k]; dfe[RAYGRAD ? li : 0][1][k] = (float)dp[li * 6 + 3 + k]; }
      }
    } else {
#pragma unroll
      for (int li = 0; li < 4; ++li) {
        const int l = 4 * q + li;
        float f0, f1;
        if (RAYGRAD) encode_level_grad(a.table, x, y, z, l_scale[l], l_res[l], l_size[l], l_off[l], l_hash[l], f0, f1, dfe[RAYGRAD ? li : 0][0], dfe[RAYGRAD ? li : 0][1]);
        else encode_level(a.table, x, y, z, l_scale[l], l_res[l], l_size[l], l_off[l], l_hash[l], f0, f1);
        xin[2 * li] = (half_t)f0; xin[2 * li + 1] = (half_t)f1;
      }
    }
    f32x4 d1[4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) d1[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf(ob), xin, zero, 0, 0, 0);
    f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf(4), pack_relu(d1[0], d1[1], true), zero, 0, 0, 0);
    d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf(5), pack_relu(d1[2], d1[3], true), d2, 0, 0, 0);
    float sh[4];
    sh4_quarter(q, dx, dy, dz, sh);
    half8 h0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { h0[r] = (half_t)d2[r]; h0[4 + r] = (half_t)sh[r]; }
    const float logit = d2[0];
    if (q == 0) h0[0] = (half_t)0.f;
    const int erow = a.avg_row >= 0 ? a.avg_row : a.cam_idx[ray];
    const half8 h1 = *reinterpret_cast<const half8*>(a.emb + (size_t)erow * 32 + 8 * q);
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
    f32x4 d5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf(22), pack_relu(d4[0], d4[1], true), zero, 0, 0, 0);
    d5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf(23), pack_relu(d4[2], d4[3], true), d5, 0, 0, 0);
    // ---------------- backward ----------------
    f32x4 dy5 = zero;
    if (q == 0 && valid) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float o = __builtin_amdgcn_rcpf(1.f + __expf(-d5[c]));   // as the forward
        dy5[c] = a.d_rgb[(size_t)n * 3 + c] * o * (1.f - o) * gscale;
      }
    }
    half_t* D = a.dump;
    // X_h2 = relu(d4), dY_h2 = dy5
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      f32x4 r4; for (int r = 0; r < 4; ++r) r4[r] = fmaxf(d4[ob][r], 0.f);
      dump_block(D + 8 * MAT, a.npad, ncol, 16 * ob + 4 * q, r4, vm);
    }
    dump_block(D + 9 * MAT, a.npad, ncol, 4 * q, dy5, 1.f);
    // head2: dX = W_h2^T dy5
    f32x4 dy4[4];
    {
      const half8 b5 = pack_relu(dy5, zero, false);
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) {
        dy4[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[(0 + ib) * 64], b5, zero, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) dy4[ib][r] = d4[ib][r] > 0.f ? dy4[ib][r] : 0.f;
      }
    }
    // X_h1 = relu(d3), dY_h1 = dy4
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      f32x4 r3; for (int r = 0; r < 4; ++r) r3[r] = fmaxf(d3[ob][r], 0.f);
      dump_block(D + 6 * MAT, a.npad, ncol, 16 * ob + 4 * q, r3, vm);
      dump_block(D + 7 * MAT, a.npad, ncol, 16 * ob + 4 * q, dy4[ob], 1.f);
    }
    f32x4 dy3[4];
    {
      const half8 b0 = pack_relu(dy4[0], dy4[1], false), b1 = pack_relu(dy4[2], dy4[3], false);
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) {
        dy3[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[(4 + ib * 2) * 64], b0, zero, 0, 0, 0);
        dy3[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[(5 + ib * 2) * 64], b1, dy3[ib], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) dy3[ib][r] = d3[ib][r] > 0.f ? dy3[ib][r] : 0.f;
      }
    }
    // X_h0 (natural column order SH16 | geo15 | emb32 | 0), dY_h0 = dy3
    {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        D[4 * MAT + (long)(4 * q + r) * a.npad + ncol] = (half_t)(sh[r] * vm);
        const int tt = 4 * q + r;
        if (tt >= 1) D[4 * MAT + (long)(15 + tt) * a.npad + ncol] = (half_t)(d2[r] * vm);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) D[4 * MAT + (long)(31 + 8 * q + j) * a.npad + ncol] = (half_t)((float)h1[j] * vm);
      if (q == 0) D[4 * MAT + (long)63 * a.npad + ncol] = (half_t)0.f;
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) dump_block(D + 5 * MAT, a.npad, ncol, 16 * ob + 4 * q, dy3[ob], 1.f);
    }
    f32x4 dbase, demb[2], dsh = zero;
    {
      const half8 b0 = pack_relu(dy3[0], dy3[1], false), b1 = pack_relu(dy3[2], dy3[3], false);
      dbase = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[12 * 64], b0, zero, 0, 0, 0);
      dbase = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[13 * 64], b1, dbase, 0, 0, 0);
      if (RAYGRAD) {                               // d loss / d SH inputs 4q..4q+3 of this point
        dsh = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[26 * 64], b0, zero, 0, 0, 0);
        dsh = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[27 * 64], b1, dsh, 0, 0, 0);
      }
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
        demb[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[(14 + ib * 2) * 64], b0, zero, 0, 0, 0);
        demb[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[(15 + ib * 2) * 64], b1, demb[ib], 0, 0, 0);
      }
    }
    // appearance-embedding gradient: reduce over the 16 points of the group when they share the row; groups on the workgroup's
    // leading row accumulate in LDS (the refresh queries every sample with camera 0: 1.5e5 same-line global atomics otherwise)
    if (a.emb_grad && a.avg_row < 0) {
      const int e0 = __shfl(erow, lane & 48);                    // row of point 0 of this group (same q)
      const bool uniform = __all(e0 == erow || !valid);
      const bool to_lds = uniform && __all(e0 == e_blk);
      const bool det_store = a.emb_det && uniform;               // (a group that mixes rows -- S not a multiple of 16 -- keeps the atomics)
      if (a.emb_det && lane == 0) a.e_part_row[grp] = (uniform && __shfl(valid ? 1 : 0, 0)) ? e0 : -1;
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = valid ? demb[ib][r] * inv_gscale : 0.f;
          if (uniform) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
            if (det_store) {
              if (p == 0) a.e_part[grp * 32 + 16 * ib + 4 * q + r] = v;
            } else if (p == 0 && v != 0.f) {
              if (to_lds) atomicAdd(&e_acc[16 * ib + 4 * q + r], v);
              else atomicAdd(a.emb_grad + (size_t)e0 * 32 + 16 * ib + 4 * q + r, v);
            }
          } else if (valid && v != 0.f) {
            atomicAdd(a.emb_grad + (size_t)erow * 32 + 16 * ib + 4 * q + r, v);
          }
        }
    }
    // base output gradient: geo part from the head, logit part from d sigma (sigma = avg * exp(logit) * sel)
    f32x4 dy2 = dbase;
    if (q == 0) dy2[0] = (valid && sel) ? a.d_density[n] * a.avg_density * __expf(fminf(fmaxf(logit, -15.f), 15.f)) * gscale : 0.f;
    // X_b1 = relu(d1), dY_b1 = dy2
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      f32x4 r1; for (int r = 0; r < 4; ++r) r1[r] = fmaxf(d1[ob][r], 0.f);
      dump_block(D + 2 * MAT, a.npad, ncol, 16 * ob + 4 * q, r1, vm);
    }
    dump_block(D + 3 * MAT, a.npad, ncol, 4 * q, dy2, vm);
    f32x4 dy1[4];
    {
      const half8 b2 = pack_relu(dy2, zero, false);
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) {
        dy1[ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[(18 + ib) * 64], b2, zero, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) dy1[ib][r] = (d1[ib][r] > 0.f && valid) ? dy1[ib][r] : 0.f;
      }
    }
    // X_b0 = enc, dY_b0 = dy1
#pragma unroll
    for (int j = 0; j < 8; ++j) D[0 * MAT + (long)(8 * q + j) * a.npad + ncol] = (half_t)((float)xin[j] * vm);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) dump_block(D + 1 * MAT, a.npad, ncol, 16 * ob + 4 * q, dy1[ob], 1.f);
    // d enc: rows permuted so that this lane receives features 8q..8q+7
    f32x4 de[2];
    {
      const half8 b0 = pack_relu(dy1[0], dy1[1], false), b1 = pack_relu(dy1[2], dy1[3], false);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        de[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[(22 + rb * 2) * 64], b0, zero, 0, 0, 0);
        de[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[(23 + rb * 2) * 64], b1, de[rb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int li = 0; li < 4; ++li) {
      const int l = 4 * q + li;
      half2v gh;
      gh[0] = (half_t)(valid ? de[li >> 1][2 * (li & 1)] : 0.f);
      gh[1] = (half_t)(valid ? de[li >> 1][2 * (li & 1) + 1] : 0.f);
      a.d_enc[(long)l * a.npad + ncol] = *reinterpret_cast<const unsigned*>(&gh);
      tl[li] += fmaxf(fabsf((float)gh[0]), fabsf((float)gh[1]));
    }
    if (RAYGRAD) {
      // d loss / d (origin, direction) of this sample's ray: hash-grid input gradient of this lane's four levels (their
      // derivatives came out of the forward recompute's gathers), through the position map, plus the SH input gradient; summed over
      // the four lanes of a point and, when the 16 points of the group lie on one ray, over the group
      float px = 0.f, py = 0.f, pz = 0.f;
      if (valid && sel && a.mode == 0) {
#pragma unroll
        for (int li = 0; li < 4; ++li) {
          const float g0 = de[li >> 1][2 * (li & 1)], g1 = de[li >> 1][2 * (li & 1) + 1];
          px = fmaf(g0, dfe[RAYGRAD ? li : 0][0][0], fmaf(g1, dfe[RAYGRAD ? li : 0][1][0], px));
          py = fmaf(g0, dfe[RAYGRAD ? li : 0][0][1], fmaf(g1, dfe[RAYGRAD ? li : 0][1][1], py));
          pz = fmaf(g0, dfe[RAYGRAD ? li : 0][0][2], fmaf(g1, dfe[RAYGRAD ? li : 0][1][2], pz));
        }
      }
      px += __shfl_xor(px, 16); py += __shfl_xor(py, 16); pz += __shfl_xor(pz, 16);
      px += __shfl_xor(px, 32); py += __shfl_xor(py, 32); pz += __shfl_xor(pz, 32);
      map_position_jt(xr, yr, zr, px, py, pz);
      float sx = 0.f, sy = 0.f, sz = 0.f;
      if (valid) {
        const float g4[4] = {dsh[0], dsh[1], dsh[2], dsh[3]};
        sh4_quarter_dpos(q, dx, dy, dz, g4, sx, sy, sz);
      }
      sx += __shfl_xor(sx, 16); sy += __shfl_xor(sy, 16); sz += __shfl_xor(sz, 16);
      sx += __shfl_xor(sx, 32); sy += __shfl_xor(sy, 32); sz += __shfl_xor(sz, 32);
      float g6[6] = {px * inv_gscale, py * inv_gscale, pz * inv_gscale,
                     fmaf(t, px, sx) * inv_gscale, fmaf(t, py, sy) * inv_gscale, fmaf(t, pz, sz) * inv_gscale};
      const int ray0 = __shfl(ray, lane & 48);
      const bool one_ray = __all(ray == ray0 || !valid);
      if (a.g6_out) {
        if (valid && q == 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) a.g6_out[(size_t)n * 6 + k] = g6[k];
        }
      } else if (one_ray) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          float v = valid ? g6[k] : 0.f;
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
          g6[k] = v;
        }
        if (p == 0 && q == 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) if (g6[k] != 0.f) atomicAdd(a.d_ray + (size_t)ray0 * 6 + k, g6[k]);
        }
      } else if (valid && q == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) if (g6[k] != 0.f) atomicAdd(a.d_ray + (size_t)ray * 6 + k, g6[k]);
      }
    }
  }
  // per-level gradient mass of this workgroup (bounds every table entry's sum: the trilinear weights of a sample add to 1)
#pragma unroll
  for (int li = 0; li < 4; ++li) {
    float v = tl[li];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
    if (p == 0) t_sum[threadIdx.x >> 6][4 * q + li] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16)
    a.t_part[(long)blockIdx.x * 16 + threadIdx.x] = t_sum[0][threadIdx.x] + t_sum[1][threadIdx.x] + t_sum[2][threadIdx.x] + t_sum[3][threadIdx.x];
  if (!a.emb_det) {
    if (threadIdx.x < 32) a.e_part[(long)blockIdx.x * 32 + threadIdx.x] = e_acc[threadIdx.x];
    if (threadIdx.x == 0) a.e_part_row[blockIdx.x] = e_blk;
  }
#undef wf
}

// ---- hash-grid gradient scatter in packed fixed point ---------------------------------------------------------------------
// Scattered global atomics cost one L2 request per distinct cache line per instruction (MI355X: 21 G/s for fp32 adds at any
// scope or working-set size, 23.7 G/s for 64-bit integer adds, tools/microbench/atomic_kinds.hip), and the two features of a
// table entry are two fp32 atomics.  Here both features travel in ONE 64-bit integer atomic: each contribution is rounded to
// int32 fixed point with a per-level power-of-two scale F_l, the pair is packed as (q1 << 32) + q0, and the sum decodes exactly
// because F_l is chosen from the level's total gradient mass T_l (sum over samples of max|g|) so that no entry can leave
// int32: |sum| <= T_l F_l + n/2 < 2^31.  Integer adds commute, so the table gradient is bit-reproducible run to run.
// workgroups 0..15 turn level l's gradient mass into its fixed-point scale; workgroups 16.. fold the per-workgroup
// appearance-embedding partials into emb_grad (runs of equal rows are summed first: one atomic per element per run and slice)
__global__ __launch_bounds__(256) void field_finalize_kernel(const float* __restrict__ t_part, int nblocks, float* __restrict__ lvl,
                                                            const float* __restrict__ e_part, const int* __restrict__ e_part_row,
                                                            float* __restrict__ emb_grad) {
  __shared__ float red[4];
  if (blockIdx.x < 16) {
    const int l = blockIdx.x;
    float v = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += 256) v += t_part[(long)b * 16 + l];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float T = red[0] + red[1] + red[2] + red[3];
      float F = 1.f;
      if (T > 0.f && T < 3.0e38f) {
        int e = 29 - (int)ceilf(log2f(T));
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        F = exp2f((float)e);
      }
      lvl[l] = F; lvl[16 + l] = 1.f / F;
    }
    return;
  }
  if (!emb_grad) return;
  // workgroup 16 + j folds partials 64j .. 64j+63: eight sub-slices of eight, runs of equal rows summed in registers first
  const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int b0 = (blockIdx.x - 16) * 64 + sl * 8;
  int rows[8]; float vals[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int b = b0 + i;
    rows[i] = b < nblocks ? e_part_row[b] : -1;
    vals[i] = b < nblocks ? e_part[(long)b * 32 + c] : 0.f;
  }
  int cur = -1; float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (rows[i] != cur) {
      if (cur >= 0 && acc != 0.f) atomicAdd(emb_grad + (size_t)cur * 32 + c, acc);
      cur = rows[i]; acc = 0.f;
    }
    acc += vals[i];
  }
  if (cur >= 0 && acc != 0.f) atomicAdd(emb_grad + (size_t)cur * 32 + c, acc);
}

// Deterministic mode: emb_grad[row] (+)= sum of the group partials of that row, in group order.  One workgroup per embedding row; 8
// interleaved partial sums per element (threads = 8 x 32), then a fixed tree.
__global__ __launch_bounds__(256) void field_emb_fold_det_kernel(const float* __restrict__ e_part, const int* __restrict__ e_part_row,
                                                                long ngroups, float* __restrict__ emb_grad, int beta) {
  __shared__ float part[8][32];
  const int row = blockIdx.x, c = threadIdx.x & 31, q = threadIdx.x >> 5;
  float acc = 0.f;
  for (long g = q; g < ngroups; g += 8)
    if (e_part_row[g] == row) acc += e_part[g * 32 + c];
  part[q][c] = acc;
  __syncthreads();
  if (threadIdx.x < 32) {
    const float v = ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) + ((part[4][c] + part[5][c]) + (part[6][c] + part[7][c]));
    float* dst = emb_grad + (size_t)row * 32 + c;
    *dst = beta ? *dst + v : v;
  }
}

struct FieldScatterArgs {
  GridLayout g;
  const float* origins; const float* dirs; const float* e_bins;
  int R, S; int mode; float aabb[6];
  const unsigned* d_enc; long npad;
  const float* lvl;                    // F_l [16], 1/F_l [16]
  unsigned long long* acc;             // [rows]: packed fixed-point sums (the table_grad buffer, zero on entry)
  int l_end;                           // levels [0, l_end) are scattered here (the rest by the owner kernels below)
  const float2* d_red; int run;        // run > 1: rays [k*run, (k+1)*run) share their (single) sample position and d_red [16][npad_r]
  long npad_r;                         // holds the gradients already summed over each run (the grid refresh: 18 directions per cell)
  const float2* d32;                   // instead of d_enc: fp32 pairs [level][npad] (the proposal networks' two-pass table gradient)
};

// sums the encoding gradient over runs of rays that share a position: d_red[l][k] = sum_j d_enc[l][k*run + j]
__global__ __launch_bounds__(256) void field_runsum_kernel(const unsigned* __restrict__ d_enc, long npad, long nruns, int run,
                                                          float2* __restrict__ d_red, long npad_r) {
  const int l = blockIdx.y;
  for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nruns; k += (long)gridDim.x * 256) {
    float s0 = 0.f, s1 = 0.f;
    for (int j = 0; j < run; ++j) {
      const unsigned raw = d_enc[(long)l * npad + k * run + j];
      const half2v gh = *reinterpret_cast<const half2v*>(&raw);
      s0 += (float)gh[0]; s1 += (float)gh[1];
    }
    d_red[(long)l * npad_r + k] = make_float2(s0, s1);
  }
}

__device__ __forceinline__ long long shfl_up_i64(long long v, int o) {
  const int lo = __shfl_up((int)(v & 0xffffffffll), o), hi = __shfl_up((int)(v >> 32), o);
  return ((long long)hi << 32) | (unsigned)lo;
}

__global__ __launch_bounds__(256) void field_scatter_kernel(FieldScatterArgs a) {
  __shared__ float l_scale[MAX_LEVELS], l_fix[MAX_LEVELS];
  __shared__ int l_res[MAX_LEVELS];
  __shared__ unsigned l_size[MAX_LEVELS], l_off[MAX_LEVELS];
  __shared__ int l_hash[MAX_LEVELS];
  if (threadIdx.x < 16) {
    const int l = threadIdx.x;
    l_scale[l] = a.g.scale[l]; l_res[l] = a.g.res[l]; l_size[l] = a.g.size[l]; l_off[l] = a.g.offset[l]; l_hash[l] = a.g.hashed[l];
    l_fix[l] = a.lvl[l];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long N = a.run > 1 ? (long)a.R / a.run : (long)a.R * a.S;      // runs of rays (S == 1) or samples
  const long nwave = (N + 63) / 64;
  for (long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6); wv < nwave; wv += (long)gridDim.x * 4) {
    long n = wv * 64 + lane;
    const bool valid = n < N;
    if (!valid) n = N - 1;
    const int ray = a.run > 1 ? (int)(n * a.run) : (int)(n / a.S), sidx = a.run > 1 ? 0 : (int)(n % a.S);
    const float t = 0.5f * (a.e_bins[(size_t)ray * (a.S + 1) + sidx] + a.e_bins[(size_t)ray * (a.S + 1) + sidx + 1]);
    float x = fmaf(a.dirs[ray * 3 + 0], t, a.origins[ray * 3 + 0]);
    float y = fmaf(a.dirs[ray * 3 + 1], t, a.origins[ray * 3 + 1]);
    float z = fmaf(a.dirs[ray * 3 + 2], t, a.origins[ray * 3 + 2]);
    map_position(x, y, z, a.mode, a.aabb);
    // grid.y > 1: one level per workgroup row (few runs, many levels: the refresh has 4096 positions -- 64 waves -- and a wave's
    // 16 x 8 corner updates are a latency chain)
    const int l_first = gridDim.y > 1 ? (int)blockIdx.y : 0, l_last = gridDim.y > 1 ? l_first + 1 : a.l_end;
    for (int l = l_first; l < l_last; ++l) {
      const float F = l_fix[l];
      float g0, g1;
      if (a.run > 1) {
        const float2 gr = valid ? a.d_red[(long)l * a.npad_r + n] : make_float2(0.f, 0.f);
        g0 = gr.x * F; g1 = gr.y * F;
      } else if (a.d32) {
        const float2 gr = valid ? a.d32[(long)l * a.npad + n] : make_float2(0.f, 0.f);
        g0 = gr.x * F; g1 = gr.y * F;
      } else {
        const unsigned raw = valid ? a.d_enc[(long)l * a.npad + n] : 0u;
        const half2v gh = *reinterpret_cast<const half2v*>(&raw);
        g0 = (float)gh[0] * F; g1 = (float)gh[1] * F;
      }
      const float scale = l_scale[l];
      const int res = l_res[l]; const unsigned size = l_size[l], offset = l_off[l]; const int hashed = l_hash[l];
      const float px = fmaf(scale, x, 0.5f), py = fmaf(scale, y, 0.5f), pz = fmaf(scale, z, 0.5f);
      const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
      const float wx = px - flx, wy = py - fly, wz = pz - flz;
      const unsigned ix = (unsigned)(int)flx, iy = (unsigned)(int)fly, iz = (unsigned)(int)flz;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const unsigned cx = ix + (c & 1), cy = iy + ((c >> 1) & 1), cz = iz + ((c >> 2) & 1);
        const float w = ((c & 1) ? wx : 1.f - wx) * ((c & 2) ? wy : 1.f - wy) * ((c & 4) ? wz : 1.f - wz);
        unsigned idx;
        if (hashed) idx = (cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) & (size - 1u);
        else {
          idx = cx + cy * (unsigned)res + cz * (unsigned)res * (unsigned)res;
          if (idx >= size) idx -= size;
        }
        const int q0 = __float2int_rn(w * g0), q1 = __float2int_rn(w * g1);
        long long v = ((long long)q1 << 32) + (long long)q0;
        // segmented sum over runs of equal indices among the 64 consecutive samples (skipped when the wave has no run)
        const unsigned prev = __shfl_up(idx, 1);
        const unsigned next = __shfl_down(idx, 1);
        int head = (lane == 0 || prev != idx) ? 1 : 0;
        const bool tail = (lane == 63) || next != idx;
        if (__any(!head)) {
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const long long up = shfl_up_i64(v, o);
            const int hu = __shfl_up(head, o);
            if (lane >= o) {
              if (!head) v += up;
              head |= hu;
            }
          }
        }
        if (tail && v != 0) atomicAdd(a.acc + offset + idx, (unsigned long long)v);
      }
    }
  }
}

// ---- owner scatter (all 16 levels of a large batch) ---------------------------------------------------------------------------
// Scattered 8-byte atomics retire at ~21-24 G/s on this chip whatever their scope or footprint (tools/microbench/atomic_*), and a
// hashed level gives consecutive samples nothing to merge: 10 hashed levels x 8 corners x 196,608 samples cost ~0.5 ms.  Here a
// workgroup OWNS a 2^14-entry slice of one level's table in LDS (128 KiB of packed sums), finds the corner updates that fall
// into it, adds them with LDS atomics and writes the slice back with plain stores -- no global atomics, same integer sums.
//   * idx = (cx ^ cy*P1 ^ cz*P2) mod 2^k and cx < 2^14, so the slice (idx >> 14) depends on (cy, cz) only: a sample has 4
//     slice ids per level, and the two x-corners of a (cy, cz) pair always land in the same slice.
//   * a dense pre-pass ORs those 4 ids into a 32-bit slice mask per (level, sample); the owner's scan reads one dword per sample,
//     tests one bit, and pushes hits into a per-wave LDS queue; every 64 queued samples are expanded with all lanes busy (the
//     hit rate per lane is 1/8, expanding in place would run the expensive part at 8 of 64 lanes).
//   * dense levels take the same path with the slice of the linear index (8 corners tested one by one); the coarsest ones, whose
//     one or two slices every sample touches, are shared by rep[l] workgroups over sample ranges and flushed with global atomics.
constexpr int OWN_SLICE_LOG2 = 14;
constexpr int OWN_THREADS = 1024;
constexpr int OWN_QUEUE = 128;
constexpr unsigned HASH_P1 = 2654435761u, HASH_P2 = 805459861u;

struct FieldOwnerArgs {
  GridLayout g;
  const float* pos; long npad; long N;
  const unsigned* d_enc;
  unsigned* ids;                       // [16][npad]: bit s set = some corner of the sample falls into slice s of that level
  const float* lvl;
  unsigned long long* acc;
  int blk_begin[MAX_LEVELS + 1];       // owner workgroups of level l: [blk_begin[l], blk_begin[l+1]) = slices x rep[l]
  unsigned char rep[MAX_LEVELS];       // coarse levels have few slices that every sample hits: rep[l] workgroups share a slice, each
                                       // takes 1/rep[l] of the samples and adds its sums to the table with (few) global atomics
  // direct flush (dst != null): a workgroup that alone owns its slice (rep == 1) holds the slice's FINISHED sums, so it decodes them
  // and writes (beta == 0) or adds to (beta != 0) the fp32 gradient itself -- the accumulator is never touched for those levels and
  // field_unpack_grad_kernel skips them (it read 8 B, read + wrote 8 B of gradient and re-zeroed 8 B per table row: 60 us per step)
  float2* dst; int beta; const float* scale;
};

// corner index of a level (dense: x + y*res + z*res^2, wrapped once; hashed: tcnn's coherent prime hash)
__device__ __forceinline__ unsigned corner_index(unsigned cx, unsigned cy, unsigned cz, int res, unsigned size, int hashed) {
  if (hashed) return (cx ^ (cy * HASH_P1) ^ (cz * HASH_P2)) & (size - 1u);
  unsigned idx = cx + cy * (unsigned)res + cz * (unsigned)res * (unsigned)res;
  if (idx >= size) idx -= size;
  return idx;
}

__global__ __launch_bounds__(256) void field_slice_ids_kernel(FieldOwnerArgs a) {
  const int l = blockIdx.y;
  const float scale = a.g.scale[l];
  const int res = a.g.res[l]; const unsigned size = a.g.size[l]; const int hashed = a.g.hashed[l];
  const unsigned nsl_mask = (size >> OWN_SLICE_LOG2) - 1u;     // hashed levels only
  for (long n = (long)blockIdx.x * 256 + threadIdx.x; n < a.npad; n += (long)gridDim.x * 256) {
    unsigned word = 0u;
    if (n < a.N && a.d_enc[(long)l * a.npad + n] != 0u) {
      const float py = fmaf(scale, a.pos[a.npad + n], 0.5f), pz = fmaf(scale, a.pos[2 * a.npad + n], 0.5f);
      const unsigned iy = (unsigned)(int)floorf(py), iz = (unsigned)(int)floorf(pz);
      if (hashed) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned t = ((iy + (k & 1)) * HASH_P1) ^ ((iz + (k >> 1)) * HASH_P2);
          word |= 1u << ((t >> OWN_SLICE_LOG2) & nsl_mask);
        }
      } else {
        const unsigned ix = (unsigned)(int)floorf(fmaf(scale, a.pos[n], 0.5f));
#pragma unroll
        for (int c = 0; c < 8; ++c)
          word |= 1u << (corner_index(ix + (c & 1), iy + ((c >> 1) & 1), iz + ((c >> 2) & 1), res, size, 0) >> OWN_SLICE_LOG2);
      }
    }
    a.ids[(long)l * a.npad + n] = word;
  }
}

__global__ __launch_bounds__(OWN_THREADS) void field_scatter_owner_kernel(FieldOwnerArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long own[];     // [1 << OWN_SLICE_LOG2] sums, then the wave queues
  constexpr unsigned NENT = 1u << OWN_SLICE_LOG2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  volatile unsigned* queue = reinterpret_cast<volatile unsigned*>(own + NENT) + wave * OWN_QUEUE;
  int l = 0;
  while (l + 1 < a.g.n_levels && (int)blockIdx.x >= a.blk_begin[l + 1]) ++l;
  const int nrep = a.rep[l];
  const unsigned nsl = (unsigned)(a.blk_begin[l + 1] - a.blk_begin[l]) / (unsigned)nrep;
  const unsigned slice = (blockIdx.x - a.blk_begin[l]) % nsl, rep = (blockIdx.x - a.blk_begin[l]) / nsl;
  const long span = (((a.N + nrep - 1) / nrep) + 4095) / 4096 * 4096;      // whole scan steps: 16 waves x 256 samples
  const long n_begin = (long)rep * span, n_end = (n_begin + span) < a.N ? (n_begin + span) : a.N;
  for (unsigned i = tid; i < NENT; i += OWN_THREADS) own[i] = 0ull;
  __syncthreads();
  const float scale = a.g.scale[l], F = a.lvl[l];
  const int res = a.g.res[l]; const unsigned size = a.g.size[l]; const int hashed = a.g.hashed[l];
  const unsigned mask = size - 1u;
  const unsigned nsl_mask = (size >> OWN_SLICE_LOG2) - 1u;
  const uint4* ids = reinterpret_cast<const uint4*>(a.ids + (long)l * a.npad);
  const unsigned* genc = a.d_enc + (long)l * a.npad;
  const float* xs = a.pos; const float* ys = a.pos + a.npad; const float* zs = a.pos + 2 * a.npad;

  // expands queue entries [head, head + cnt) (cnt <= 64), one sample per lane: the corners that fall into this slice
  auto expand = [&](unsigned head, int cnt) {
    if (lane < cnt) {
      const long n = (long)queue[(head + lane) & (OWN_QUEUE - 1)];
      const unsigned raw = genc[n];
      const float x = xs[n], y = ys[n], z = zs[n];
      const half2v gh = *reinterpret_cast<const half2v*>(&raw);
      const float g0 = (float)gh[0] * F, g1 = (float)gh[1] * F;
      const float px = fmaf(scale, x, 0.5f), py = fmaf(scale, y, 0.5f), pz = fmaf(scale, z, 0.5f);
      const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
      const float wx = px - flx, wy = py - fly, wz = pz - flz;
      const unsigned ix = (unsigned)(int)flx, iy = (unsigned)(int)fly, iz = (unsigned)(int)flz;
      auto add = [&](unsigned idx, float w) {
        const int q0 = __float2int_rn(w * g0), q1 = __float2int_rn(w * g1);
        const long long v = ((long long)q1 << 32) + (long long)q0;
        if (v != 0) atomicAdd(&own[idx & (NENT - 1u)], (unsigned long long)v);
      };
      if (hashed) {
        // which of the sample's four (cy, cz) pairs fall into this slice: 1.09 of them on average (a queued sample has at least one).
        // The lanes pop their pairs round by round, so the two LDS atomics of the first round run with every hit lane active and a
        // second round only for the lanes that have a second pair -- as four predicated pairs the same updates took EIGHT atomic
        // instructions at a quarter of the lanes each, and the owner's expansion is priced by atomic instructions issued (round 6).
        unsigned m4 = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned t = ((iy + (k & 1)) * HASH_P1) ^ ((iz + (k >> 1)) * HASH_P2);
          if (((t >> OWN_SLICE_LOG2) & nsl_mask) == slice) m4 |= 1u << k;
        }
        while (m4) {
          const int k = __builtin_ctz(m4);
          m4 &= m4 - 1u;
          const int kb = k & 1, kc = k >> 1;
          const unsigned t = ((iy + kb) * HASH_P1) ^ ((iz + kc) * HASH_P2);
          const float wyv = kb ? wy : 1.f - wy, wzv = kc ? wz : 1.f - wz;
#pragma unroll
          for (int ka = 0; ka < 2; ++ka)
            add(((ix + ka) ^ t) & mask, ((ka ? wx : 1.f - wx) * wyv) * wzv);   // the product order of field_scatter_kernel
        }
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const unsigned idx = corner_index(ix + (c & 1), iy + ((c >> 1) & 1), iz + ((c >> 2) & 1), res, size, 0);
          if ((idx >> OWN_SLICE_LOG2) != slice) continue;
          add(idx, ((c & 1) ? wx : 1.f - wx) * ((c & 2) ? wy : 1.f - wy) * ((c & 4) ? wz : 1.f - wz));
        }
      }
    }
  };

  // scan: 4 consecutive samples per lane and step (one 16-B load, the next step's already in flight)
  unsigned head = 0; int qcount = 0;                      // wave-uniform
  const long step = (long)(OWN_THREADS / 64) * 256;
  long base = n_begin + (long)wave * 256;
  const uint4 none = make_uint4(0u, 0u, 0u, 0u);
  uint4 cur = (base < n_end && base + lane * 4 < a.npad) ? ids[(base >> 2) + lane] : none;
  for (; base < n_end; base += step) {
    const long nb = base + step;
    const uint4 nxt = (nb < n_end && nb + lane * 4 < a.npad) ? ids[(nb >> 2) + lane] : none;
    const unsigned wv[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool has = (wv[j] >> slice) & 1u;
      const unsigned long long b = __ballot(has);
      if (b == 0ull) continue;
      if (has) {
        const int at = qcount + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0));
        queue[(head + at) & (OWN_QUEUE - 1)] = (unsigned)(base + lane * 4 + j);
      }
      qcount += __popcll(b);
      __builtin_amdgcn_wave_barrier();
      if (qcount >= 64) { expand(head, 64); head += 64; qcount -= 64; __builtin_amdgcn_wave_barrier(); }
    }
    cur = nxt;
  }
  if (qcount > 0) expand(head, qcount);
  __syncthreads();
  const size_t e0 = (size_t)slice * NENT;
  unsigned long long* dst = a.acc + a.g.offset[l] + e0;
  if (nrep == 1 && a.dst) {
    // the same decode as field_unpack_grad_kernel, operation for operation (the two paths are compared bit for bit)
    const float m = a.lvl[16 + l] * a.scale[1];
    float2* out = a.dst + a.g.offset[l] + e0;
    for (unsigned i = tid; i < NENT; i += OWN_THREADS) {
      if (e0 + i >= size) continue;
      const long long v = (long long)own[i];
      if (v == 0) {
        if (!a.beta) out[i] = make_float2(0.f, 0.f);
        continue;
      }
      const int lo = (int)(v & 0xffffffffll);
      const int hi = (int)((v - (long long)lo) >> 32);
      float2 o = make_float2((float)lo * m, (float)hi * m);
      if (a.beta) { const float2 old = out[i]; o.x += old.x; o.y += old.y; }
      out[i] = o;
    }
  } else if (nrep == 1) {
    for (unsigned i = tid; i < NENT; i += OWN_THREADS)
      if (e0 + i < size) dst[i] = own[i];
  } else {
    for (unsigned i = tid; i < NENT; i += OWN_THREADS) {
      const unsigned long long v = own[i];
      if (v != 0ull && e0 + i < size) atomicAdd(dst + i, v);
    }
  }
}

// packed fixed point -> fp32 pair:  (lo, hi) * (1 / F_l) * (1 / S).  dst == acc: in place (the legacy form: the gradient buffer
// doubles as the accumulator).  dst != acc: the accumulator is a persistent scratch that this kernel leaves ZERO again for the next
// call (no fill launch per call), and dst is written (beta == 0: every entry, zeros included) or added to (beta != 0: entries
// with a contribution only) -- the second producer of a parameter's gradient in one backward pass adds in place instead of
// handing autograd a second tensor to sum (a 49 MB add per step for the radiance table: render batch + grid refresh)
__global__ __launch_bounds__(256) void field_unpack_grad_kernel(GridLayout g, const float* __restrict__ lvl, const float* __restrict__ scale,
                                                               unsigned long long* __restrict__ acc, float2* __restrict__ dst, int beta,
                                                               unsigned skip_levels = 0u) {
  const int l = blockIdx.y;
  if ((skip_levels >> l) & 1u) return;      // flushed by its owner workgroups (FieldOwnerArgs::dst)
  const float m = lvl[16 + l] * scale[1];
  const unsigned size = g.size[l], off = g.offset[l];
  const bool inplace = reinterpret_cast<void*>(dst) == reinterpret_cast<void*>(acc);
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < size; i += gridDim.x * 256) {
    const long long v = (long long)acc[off + i];
    if (v == 0) {                                           // bit pattern 0 is also fp32 (0, 0)
      if (!inplace && !beta) dst[off + i] = make_float2(0.f, 0.f);
      continue;
    }
    const int lo = (int)(v & 0xffffffffll);
    const int hi = (int)((v - (long long)lo) >> 32);
    float2 o = make_float2((float)lo * m, (float)hi * m);
    if (!inplace) {
      acc[off + i] = 0ull;
      if (beta) { const float2 old = dst[off + i]; o.x += old.x; o.y += old.y; }
    }
    dst[off + i] = o;
  }
}

// amax of the upstream gradients entering the fp16 chain -> power-of-two scale {S, 1/S}
__global__ __launch_bounds__(256) void field_amax_kernel(const float* __restrict__ d_rgb, const float* __restrict__ d_density,
                                                        const float* __restrict__ density, long N, float avg_density,
                                                        unsigned* __restrict__ amax_bits, float* __restrict__ zero_p, long zero_n) {
  // first launch of the backward: also clears the (small) embedding-gradient buffer the later kernels add into
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < zero_n; i += (long)gridDim.x * 256) zero_p[i] = 0.f;
  // |d logit| = |d sigma| * avg * exp(clamp(logit, -15, 15)): the forward density with the trunc_exp clamp applied to it (an
  // un-clamped density of 1e20 would push the scale so low that every clamped gradient underflows the fp16 chain)
  const float dlo = avg_density * 3.0590232e-7f, dhi = avg_density * 3269017.4f;       // avg * e^-15, avg * e^15
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) {
    float v = fmaxf(fabsf(d_rgb[i * 3]), fmaxf(fabsf(d_rgb[i * 3 + 1]), fabsf(d_rgb[i * 3 + 2]))) * 0.25f;   // sigmoid' <= 1/4
    v = fmaxf(v, fabsf(d_density[i]) * fminf(fmaxf(density[i], dlo), dhi));
    m = (v == v && v > m) ? v : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(amax_bits, __float_as_uint(fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]))));
}

__global__ void field_make_scale_kernel(float* __restrict__ scale) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float amax = __uint_as_float(reinterpret_cast<unsigned*>(scale)[2]);
    reinterpret_cast<unsigned*>(scale)[2] = 0u;      // consumed: the atomic-max word is zero again for the next call (the dump starts zeroed)
    float S = 1.f;
    if (amax > 0.f && amax < 3.0e38f) {
      int e = 6 - (int)floorf(log2f(amax));       // max |dY| of the last layer -> [64, 128)
      e = e > 100 ? 100 : (e < -100 ? -100 : e);
      S = exp2f((float)e);
    }
    scale[0] = S; scale[1] = 1.f / S;
  }
}

}  // namespace

// =================================================================================================
extern "C" int neraf_render_loss(neraf_ctx* ctx, const float* density, const float* rgb_s, const float* e_bins,
                                 const float* s_bins, const float* gt_rgb, int R, int S, float distortion_mult, const float* up3,
                                 float* d_rgb_s, float* d_density, float* d_density_dist, float* sums, neraf_stream_t stream) {
  if (R <= 0 || S <= 0 || S > 64 || !density || !rgb_s || !e_bins || !s_bins || !gt_rgb || !sums || (up3 && (!d_rgb_s || !d_density)) ||
      (!up3 && d_rgb_s && (!d_density || !d_density_dist)))
    return neraf_fail(ctx, NERAF_EINVAL, "render_loss: bad arguments (S <= 64)");
  RenderLossArgs a{density, rgb_s, e_bins, s_bins, gt_rgb, R, S, distortion_mult, up3, d_rgb_s, d_density, d_density_dist, sums};
  hipLaunchKernelGGL(render_loss_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_interlevel_loss(neraf_ctx* ctx, const float* c_bins, const float* w_fine, int S2, const float* p_bins,
                                     const float* p_ebins, const float* p_density, int Sp, int R, float mult, const float* up3,
                                     float* d_density, float* sums, neraf_stream_t stream) {
  if (R <= 0 || S2 <= 0 || S2 > IL_MAX_S2 || Sp <= 0 || Sp > IL_MAX_SP || !c_bins || !w_fine || !p_bins || !p_ebins || !p_density ||
      !sums || (up3 && !d_density))
    return neraf_fail(ctx, NERAF_EINVAL, "interlevel_loss: bad arguments (S2 <= 64, Sp <= 256)");
  InterlevelArgs a{c_bins, w_fine, S2, p_bins, p_ebins, p_density, Sp, R, mult, up3, d_density, sums};
  hipLaunchKernelGGL(interlevel_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

static int proposal_backward_impl(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* mlp_f16,
                                  const float* origins, const float* dirs, const float* e_bins, const float* d_density,
                                  int R, int S, float avg_density, float* table_grad, float* w_grad, void* scratch,
                                  size_t scratch_bytes, float* d_rays, float* w0_out, float* w1_out, void* acc_scratch,
                                  neraf_stream_t stream);

extern "C" size_t neraf_proposal_backward_scratch_bytes(int R, int S, int n_levels) {
  const size_t npad = ((size_t)R * S + 63) / 64 * 64;
  return (size_t)2048 * 272 * sizeof(float) + (size_t)2048 * 16 * sizeof(float) + 64 * sizeof(float) + (size_t)n_levels * npad * sizeof(float2) +
         (neraf_deterministic() ? npad * 6 * sizeof(float) : 0);      // deterministic mode: per-sample ray gradients (ray_fold_kernel)
}

extern "C" int neraf_proposal_backward(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* mlp_f16,
                                       const float* origins, const float* dirs, const float* e_bins, const float* d_density,
                                       int R, int S, float avg_density, float* table_grad, float* w_grad, void* scratch,
                                       size_t scratch_bytes, neraf_stream_t stream) {
  return proposal_backward_impl(ctx, g, table_f16, mlp_f16, origins, dirs, e_bins, d_density, R, S, avg_density, table_grad, w_grad,
                                scratch, scratch_bytes, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int neraf_proposal_backward_ex(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* mlp_f16,
                                          const float* origins, const float* dirs, const float* e_bins, const float* d_density,
                                          int R, int S, float avg_density, float* table_grad, float* w0_grad, float* w1_grad,
                                          void* scratch, size_t scratch_bytes, void* acc_scratch, float* d_rays,
                                          neraf_stream_t stream) {
  if (!w0_grad || !w1_grad || !acc_scratch || !scratch || scratch_bytes < neraf_proposal_backward_scratch_bytes(R, S, g ? g->n_levels : 0))
    return neraf_fail(ctx, NERAF_EINVAL, "proposal_backward_ex: w0_grad, w1_grad, acc_scratch and the full scratch are required");
  return proposal_backward_impl(ctx, g, table_f16, mlp_f16, origins, dirs, e_bins, d_density, R, S, avg_density, table_grad, w0_grad,
                                scratch, scratch_bytes, d_rays, w0_grad, w1_grad, acc_scratch, stream);
}

extern "C" int neraf_proposal_backward_rays(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* mlp_f16,
                                            const float* origins, const float* dirs, const float* e_bins, const float* d_density,
                                            int R, int S, float avg_density, float* table_grad, float* w_grad, void* scratch,
                                            size_t scratch_bytes, float* d_rays, neraf_stream_t stream) {
  if (!d_rays) return neraf_fail(ctx, NERAF_EINVAL, "proposal_backward_rays: d_rays required");
  return proposal_backward_impl(ctx, g, table_f16, mlp_f16, origins, dirs, e_bins, d_density, R, S, avg_density, table_grad, w_grad,
                                scratch, scratch_bytes, d_rays, nullptr, nullptr, nullptr, stream);
}

static int proposal_backward_impl(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* mlp_f16,
                                  const float* origins, const float* dirs, const float* e_bins, const float* d_density,
                                  int R, int S, float avg_density, float* table_grad, float* w_grad, void* scratch,
                                  size_t scratch_bytes, float* d_rays, float* w0_out, float* w1_out, void* acc_scratch,
                                  neraf_stream_t stream) {
  PropBwdArgs a{};
  if (make_grid_layout(g, &a.g) || a.g.n_levels > 8) return neraf_fail(ctx, NERAF_EINVAL, "proposal_backward: bad grid (<= 8 levels)");
  if (R <= 0 || S <= 0 || !table_f16 || !mlp_f16 || !origins || !dirs || !e_bins || !d_density || !table_grad || !w_grad)
    return neraf_fail(ctx, NERAF_EINVAL, "proposal_backward: bad arguments");
  a.table = (const unsigned*)table_f16; a.w = (const half_t*)mlp_f16; a.origins = origins; a.dirs = dirs; a.e_bins = e_bins;
  a.d_density = d_density; a.R = R; a.S = S; a.avg_density = avg_density; a.table_grad = table_grad; a.w_grad = w_grad; a.d_ray = d_rays;
  const long n = (long)R * S;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  a.w_part = (scratch && scratch_bytes >= (size_t)blocks * 272 * sizeof(float)) ? (float*)scratch : nullptr;
  // packed two-pass table gradient when the scratch holds it (neraf_proposal_backward_scratch_bytes); without the scratch: fp32 atomics
  // from the MLP-backward kernel (the NERAF_PROP_PACKED=0 toggle of rounds 2-4 is gone: with the persistent accumulator every caller
  // passes since round 4 it could only fail)
  constexpr int packed_on = 1;
  const long npad = (n + 63) / 64 * 64;
  float* lvl = nullptr;
  if (packed_on && a.w_part && scratch_bytes >= neraf_proposal_backward_scratch_bytes(R, S, a.g.n_levels)) {
    char* sp = (char*)scratch + (size_t)2048 * 272 * sizeof(float);
    a.t_part = (float*)sp; sp += (size_t)2048 * 16 * sizeof(float);
    lvl = (float*)sp; sp += 32 * sizeof(float);
    a.one2 = (float*)sp; sp += 32 * sizeof(float);
    a.d32 = (float2*)sp; a.npad = npad;
    sp += (size_t)a.g.n_levels * npad * sizeof(float2);
    if (neraf_deterministic() && d_rays) a.g6_out = (float*)sp;
  }
  hipStream_t st = (hipStream_t)stream;
  {
    ProfScope prof(ctx, st, PROF_PROP_BWD, (double)n * a.g.n_levels * 8 * 8);   // 8 bytes added per (sample, level, corner)
    hipLaunchKernelGGL(proposal_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    if (a.g6_out) hipLaunchKernelGGL(ray_fold_kernel, dim3((R + 3) / 4), dim3(256), 0, st, a.g6_out, R, S, d_rays);
    if (a.d32) {
      hipLaunchKernelGGL(field_finalize_kernel, dim3(16), dim3(256), 0, st, a.t_part, (int)blocks, lvl, nullptr, nullptr, nullptr);
      FieldScatterArgs sa{};
      sa.g = a.g; sa.origins = origins; sa.dirs = dirs; sa.e_bins = e_bins; sa.R = R; sa.S = S; sa.mode = 0;
      sa.npad = npad; sa.lvl = lvl; sa.acc = reinterpret_cast<unsigned long long*>(acc_scratch ? acc_scratch : (void*)table_grad);
      sa.l_end = a.g.n_levels; sa.run = 1;
      sa.d32 = a.d32;
      long sblocks = ((n + 63) / 64 + 3) / 4;
      const long cap = (long)(ctx ? ctx->num_cus : 256) * 8;
      if (sblocks > cap) sblocks = cap;
      hipLaunchKernelGGL(field_scatter_kernel, dim3((unsigned)sblocks, 1), dim3(256), 0, st, sa);
      unsigned maxsize = 0;
      for (int l = 0; l < a.g.n_levels; ++l) maxsize = a.g.size[l] > maxsize ? a.g.size[l] : maxsize;
      hipLaunchKernelGGL(field_unpack_grad_kernel, dim3((maxsize + 1023) / 1024, a.g.n_levels), dim3(256), 0, st, a.g, lvl, a.one2,
                         sa.acc, reinterpret_cast<float2*>(table_grad), 0);
    } else if (acc_scratch) return neraf_fail(ctx, NERAF_EINVAL, "proposal_backward_ex: packed table gradient unavailable");
  }
  if (a.w_part) hipLaunchKernelGGL(prop_wgrad_fold_kernel, dim3(34), dim3(256), 0, st, a.w_part, (int)blocks, w_grad, w0_out, w1_out);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" size_t neraf_field_backward_dump_bytes(int R, int S) {
  const long npad = ((long)R * S + 63) / 64 * 64;
  return (size_t)10 * 128 * npad * 2 + 256 + (neraf_deterministic() ? (size_t)npad * 6 * sizeof(float) : 0);   // + per-sample ray gradients
}

static int field_backward_impl(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                               const void* wfrag_bwd_f16, const void* emb_f16, const float* origins, const float* dirs,
                               const float* e_bins, const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host,
                               float avg_density, int avg_row, const float* density, const float* d_rgb,
                               const float* d_density, float* table_grad, float* emb_grad, float* const* w_grads, void* dump,
                               void* splitk_ws, size_t splitk_bytes, int pos_run, float* d_rays, const void* enc_in, const void* denc_in,
                               void* acc_scratch, int accumulate, int emb_rows, neraf_stream_t stream);

extern "C" int neraf_field_backward(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                                    const void* wfrag_bwd_f16, const void* emb_f16, const float* origins, const float* dirs,
                                    const float* e_bins, const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host,
                                    float avg_density, int avg_row, const float* density, const float* d_rgb, const float* d_density,
                                    float* table_grad, float* emb_grad, float* const* w_grads, void* dump, void* splitk_ws,
                                    size_t splitk_bytes, neraf_stream_t stream) {
  return field_backward_impl(ctx, g, table_f16, wfrag_f16, wfrag_bwd_f16, emb_f16, origins, dirs, e_bins, cam_idx, R, S, mode,
                             aabb_host, avg_density, avg_row, density, d_rgb, d_density, table_grad, emb_grad, w_grads, dump,
                             splitk_ws, splitk_bytes, 1, nullptr, nullptr, nullptr, nullptr, 0, 0, stream);
}

extern "C" int neraf_field_backward_runs(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                                         const void* wfrag_bwd_f16, const void* emb_f16, const float* origins, const float* dirs,
                                         const float* e_bins, const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host,
                                         float avg_density, int avg_row, const float* density, const float* d_rgb,
                                         const float* d_density, float* table_grad, float* emb_grad, float* const* w_grads, void* dump,
                                         void* splitk_ws, size_t splitk_bytes, int pos_run, neraf_stream_t stream) {
  return field_backward_impl(ctx, g, table_f16, wfrag_f16, wfrag_bwd_f16, emb_f16, origins, dirs, e_bins, cam_idx, R, S, mode,
                             aabb_host, avg_density, avg_row, density, d_rgb, d_density, table_grad, emb_grad, w_grads, dump,
                             splitk_ws, splitk_bytes, pos_run, nullptr, nullptr, nullptr, nullptr, 0, 0, stream);
}

extern "C" int neraf_field_backward_rays(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                                         const void* wfrag_bwd_f16, const void* emb_f16, const float* origins, const float* dirs,
                                         const float* e_bins, const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host,
                                         float avg_density, int avg_row, const float* density, const float* d_rgb,
                                         const float* d_density, float* table_grad, float* emb_grad, float* const* w_grads, void* dump,
                                         void* splitk_ws, size_t splitk_bytes, float* d_rays, neraf_stream_t stream) {
  if (!d_rays) return neraf_fail(ctx, NERAF_EINVAL, "field_backward_rays: d_rays required");
  return field_backward_impl(ctx, g, table_f16, wfrag_f16, wfrag_bwd_f16, emb_f16, origins, dirs, e_bins, cam_idx, R, S, mode,
                             aabb_host, avg_density, avg_row, density, d_rgb, d_density, table_grad, emb_grad, w_grads, dump,
                             splitk_ws, splitk_bytes, 1, d_rays, nullptr, nullptr, nullptr, 0, 0, stream);
}

extern "C" int neraf_field_backward_ex(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                                      const void* wfrag_bwd_f16, const void* emb_f16, const float* origins, const float* dirs,
                                      const float* e_bins, const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host,
                                      float avg_density, int avg_row, const float* density, const float* d_rgb,
                                      const float* d_density, float* table_grad, float* emb_grad, float* const* w_grads, void* dump,
                                      void* splitk_ws, size_t splitk_bytes, int pos_run, float* d_rays, const void* enc_saved,
                                      const void* denc_saved, void* acc_scratch, int accumulate, int emb_rows, neraf_stream_t stream) {
  if (d_rays && pos_run != 1) return neraf_fail(ctx, NERAF_EINVAL, "field_backward_ex: ray gradients need pos_run == 1");
  return field_backward_impl(ctx, g, table_f16, wfrag_f16, wfrag_bwd_f16, emb_f16, origins, dirs, e_bins, cam_idx, R, S, mode,
                             aabb_host, avg_density, avg_row, density, d_rgb, d_density, table_grad, emb_grad, w_grads, dump,
                             splitk_ws, splitk_bytes, pos_run, d_rays, enc_saved, denc_saved, acc_scratch, accumulate, emb_rows, stream);
}

static int field_backward_impl(neraf_ctx* ctx, const neraf_grid_desc* g, const void* table_f16, const void* wfrag_f16,
                               const void* wfrag_bwd_f16, const void* emb_f16, const float* origins, const float* dirs,
                               const float* e_bins, const int32_t* cam_idx, int R, int S, int mode, const float* aabb_host,
                               float avg_density, int avg_row, const float* density, const float* d_rgb,
                               const float* d_density, float* table_grad, float* emb_grad, float* const* w_grads, void* dump,
                               void* splitk_ws, size_t splitk_bytes, int pos_run, float* d_rays, const void* enc_in, const void* denc_in,
                               void* acc_scratch, int accumulate, int emb_rows, neraf_stream_t stream) {
  if (accumulate && !acc_scratch)
    return neraf_fail(ctx, NERAF_EINVAL, "field_backward: accumulate needs acc_scratch (the gradient buffer cannot double as the accumulator)");
  if (pos_run < 1 || (pos_run > 1 && (S != 1 || R % pos_run != 0)))
    return neraf_fail(ctx, NERAF_EINVAL, "field_backward_runs: pos_run > 1 needs S == 1 and R a multiple of pos_run");
  FieldBwdArgs a{};
  if (make_grid_layout(g, &a.g) || a.g.n_levels != 16) return neraf_fail(ctx, NERAF_EINVAL, "field_backward: grid must have 16 levels");
  if (R <= 0 || S <= 0 || !table_f16 || !wfrag_f16 || !wfrag_bwd_f16 || !emb_f16 || !origins || !dirs || !e_bins || !density ||
      !d_rgb || !d_density || !table_grad || !w_grads || !dump || (avg_row < 0 && !cam_idx) || (mode != 0 && !aabb_host))
    return neraf_fail(ctx, NERAF_EINVAL, "field_backward: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const long N = (long)R * S;
  const long npad = (N + 63) / 64 * 64;
  a.table = (const unsigned*)table_f16; a.wfrag = (const half8*)wfrag_f16; a.wfrag_b = (const half8*)wfrag_bwd_f16;
  a.emb = (const half_t*)emb_f16; a.origins = origins; a.dirs = dirs; a.e_bins = e_bins; a.cam_idx = cam_idx;
  a.R = R; a.S = S; a.mode = mode;
  for (int i = 0; i < 6; ++i) a.aabb[i] = aabb_host ? aabb_host[i] : 0.f;
  a.avg_density = avg_density; a.avg_row = avg_row; a.d_rgb = d_rgb; a.d_density = d_density;
  a.emb_grad = emb_grad;
  a.dump = (half_t*)dump; a.npad = npad;
  a.d_enc = reinterpret_cast<unsigned*>((half_t*)dump + (size_t)64 * npad);                       // slot 0, rows 64..127
  a.t_part = reinterpret_cast<float*>((half_t*)dump + (size_t)128 * npad + (size_t)64 * npad);    // slot 1, rows 64..
  a.e_part = reinterpret_cast<float*>((half_t*)dump + (size_t)2 * 128 * npad + (size_t)64 * npad);    // slot 2, rows 64..
  a.e_part_row = reinterpret_cast<int*>((half_t*)dump + (size_t)3 * 128 * npad + (size_t)64 * npad);  // slot 3, rows 64..
  float* scale = (float*)((char*)dump + (size_t)10 * 128 * npad * 2);
  a.scale = scale;
  if (neraf_deterministic() && d_rays) a.g6_out = (float*)((char*)dump + (size_t)10 * 128 * npad * 2 + 256);   // neraf_field_backward_dump_bytes
  // owner scatter (no global atomics) for a large batch: every level in <= 32 slices of 2^14 entries, hashed levels of
  // power-of-two size, and every coordinate below the slice size (the hashed slice id must not depend on x)
  constexpr int kOwnLds = (1 << OWN_SLICE_LOG2) * 8 + (OWN_THREADS / 64) * OWN_QUEUE * 4;
  const char* own_e = getenv("NERAF_FIELD_OWNER_SCATTER");      // read per call: the parity test flips it between two runs
  const int own_env = own_e ? atoi(own_e) : 1;
  bool use_owner = own_env && pos_run == 1 && (N >= 131072 || own_env == 2) && N < (1l << 30);   // 2 = also for small batches (tests)
  int own_blk[MAX_LEVELS + 1] = {0};
  unsigned char own_rep[MAX_LEVELS] = {0};
  for (int l = 0; l < 16 && use_owner; ++l) {
    const unsigned sz = a.g.size[l];
    const unsigned nsl = (sz + (1u << OWN_SLICE_LOG2) - 1u) >> OWN_SLICE_LOG2;
    use_owner = nsl <= 32 && a.g.res[l] + 1 < (1 << OWN_SLICE_LOG2) &&
                (!a.g.hashed[l] || ((sz & (sz - 1)) == 0 && sz >= (2u << OWN_SLICE_LOG2)));
    // a hashed slice sees ~1/8 of the samples (4 slice ids of 32); a dense level's sample touches 1-2 of its nsl slices
    int rep = a.g.hashed[l] ? 1 : (int)((10 + nsl / 2) / nsl);
    rep = rep < 1 ? 1 : (rep > 12 ? 12 : rep);
    own_rep[l] = (unsigned char)rep;
    own_blk[l + 1] = own_blk[l] + (int)nsl * rep;
  }
  if (use_owner) {
    static bool attr_set = false;
    if (!attr_set) {
      NERAF_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&field_scatter_owner_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kOwnLds));
      attr_set = true;
    }
    a.pos = reinterpret_cast<float*>((half_t*)dump + (size_t)4 * 128 * npad + (size_t)64 * npad);   // slot 4, rows 64..: pos [3][npad], ids
  }
  // scale[2] (the atomic-max word) is zero on entry: the dump is zero-initialised by the caller once, and field_make_scale_kernel
  // clears the word after reading it
  {
    long blocks = (N + 1023) / 1024; if (blocks > 256) blocks = 256;
    // with a separate accumulator the library owns the initial state of the outputs: a first producer (accumulate == 0) starts the
    // embedding gradient from zero here; the legacy form (acc_scratch == NULL) expects the caller to have zeroed both buffers
    a.emb_det = (neraf_deterministic() && acc_scratch && emb_grad && avg_row < 0 && emb_rows > 0) ? 1 : 0;
    const bool zero_emb = acc_scratch && !accumulate && emb_grad && avg_row < 0 && emb_rows > 0;
    hipLaunchKernelGGL(field_amax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_rgb, d_density, density, N, avg_density,
                       reinterpret_cast<unsigned*>(scale) + 2, zero_emb ? emb_grad : nullptr, zero_emb ? (long)emb_rows * 32 : 0l);
    hipLaunchKernelGGL(field_make_scale_kernel, dim3(1), dim3(64), 0, st, scale);
  }
  {
    const long groups = npad / 16;
    long blocks = (groups + 3) / 4;
    const long cap = (long)(ctx ? ctx->num_cus : 256) * 8;
    if (blocks > cap) blocks = cap;
    {
      ProfScope prof(ctx, st, PROF_FIELD_BWD, (double)N * 16 * 8 * 4);       // gathered fp16x2 bytes
      a.d_ray = d_rays; a.enc_in = (const half_t*)enc_in; a.denc_in = (const half_t*)denc_in;
      if (d_rays && enc_in && denc_in) hipLaunchKernelGGL((field_backward_kernel<true, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
      else if (d_rays) hipLaunchKernelGGL((field_backward_kernel<true, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);
      else if (enc_in) hipLaunchKernelGGL((field_backward_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
      else hipLaunchKernelGGL((field_backward_kernel<false, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);
      if (a.g6_out) hipLaunchKernelGGL(ray_fold_kernel, dim3((R + 3) / 4), dim3(256), 0, st, a.g6_out, R, S, d_rays);
    }
    NERAF_HIP_CHECK(ctx, hipGetLastError());
    // hash-grid gradient: per-level fixed-point scale from the gradient mass, packed 64-bit scatter, in-place unpack
    float* lvl = scale + 8;
    const bool fold_emb = avg_row < 0 && emb_grad && !a.emb_det;
    if (a.emb_det)
      hipLaunchKernelGGL(field_emb_fold_det_kernel, dim3((unsigned)emb_rows), dim3(256), 0, st, a.e_part, a.e_part_row, npad / 16, emb_grad, 1);
    hipLaunchKernelGGL(field_finalize_kernel, dim3(16 + (fold_emb ? (unsigned)((blocks + 63) / 64) : 0u)), dim3(256), 0, st, a.t_part,
                       (int)blocks, lvl, a.e_part, a.e_part_row, fold_emb ? emb_grad : nullptr);
    FieldScatterArgs sa{};
    sa.g = a.g; sa.origins = origins; sa.dirs = dirs; sa.e_bins = e_bins; sa.R = R; sa.S = S; sa.mode = mode;
    for (int i = 0; i < 6; ++i) sa.aabb[i] = a.aabb[i];
    sa.d_enc = a.d_enc; sa.npad = npad; sa.lvl = lvl;
    sa.acc = reinterpret_cast<unsigned long long*>(acc_scratch ? acc_scratch : (void*)table_grad);
    sa.l_end = 16;
    sa.run = pos_run;
    long sblocks = ((N + 63) / 64 + 3) / 4;
    if (pos_run > 1) {
      // runs of rays sharing one position (the grid refresh: 18 directions per cell): sum their encoding gradients first, then
      // scatter one update set per run -- 18x less index arithmetic and merging (scratch: slot 5, rows 64.. of the dump)
      const long nruns = N / pos_run;
      sa.npad_r = (nruns + 63) / 64 * 64;
      float2* d_red = reinterpret_cast<float2*>((half_t*)dump + (size_t)5 * 128 * npad + (size_t)64 * npad);
      sa.d_red = d_red;
      long rb = (nruns + 255) / 256; if (rb > 256) rb = 256;
      hipLaunchKernelGGL(field_runsum_kernel, dim3((unsigned)rb, 16), dim3(256), 0, st, a.d_enc, npad, nruns, pos_run, d_red, sa.npad_r);
      sblocks = ((nruns + 63) / 64 + 3) / 4;
    }
    if (sblocks > cap) sblocks = cap;
    unsigned direct_levels = 0u;
    {
      ProfScope prof(ctx, st, PROF_FIELD_SCATTER, (double)N * 16 * 8 * 8);   // one 8-byte update per (sample, level, corner) before merging
      if (!use_owner) hipLaunchKernelGGL(field_scatter_kernel, dim3((unsigned)sblocks, pos_run > 1 ? 16 : 1), dim3(256), 0, st, sa);
      else {
        FieldOwnerArgs oa{};
        oa.g = a.g; oa.pos = a.pos; oa.npad = npad; oa.N = N; oa.d_enc = a.d_enc; oa.lvl = lvl; oa.acc = sa.acc;
        oa.ids = reinterpret_cast<unsigned*>(a.pos + 3 * npad);
        for (int l = 0; l <= 16; ++l) oa.blk_begin[l] = own_blk[l];
        for (int l = 0; l < 16; ++l) oa.rep[l] = own_rep[l];
        static const int direct_on = [] { const char* e = getenv("NERAF_OWNER_DIRECT"); return e ? atoi(e) : 1; }();      // 0: A/B
        if (acc_scratch && direct_on) {
          oa.dst = reinterpret_cast<float2*>(table_grad); oa.beta = accumulate; oa.scale = scale;
          for (int l = 0; l < 16; ++l) if (own_rep[l] == 1) direct_levels |= 1u << l;
        }
        long iblocks = (npad + 255) / 256; if (iblocks > 512) iblocks = 512;
        hipLaunchKernelGGL(field_slice_ids_kernel, dim3((unsigned)iblocks, 16), dim3(256), 0, st, oa);
        hipLaunchKernelGGL(field_scatter_owner_kernel, dim3(own_blk[16]), dim3(OWN_THREADS), kOwnLds, st, oa);
      }
    }
    unsigned maxsize = 0;
    for (int l = 0; l < 16; ++l) maxsize = a.g.size[l] > maxsize ? a.g.size[l] : maxsize;
    hipLaunchKernelGGL(field_unpack_grad_kernel, dim3((maxsize + 1023) / 1024, 16), dim3(256), 0, st, a.g, lvl, scale,
                       sa.acc, reinterpret_cast<float2*>(table_grad), accumulate, direct_levels);
    NERAF_HIP_CHECK(ctx, hipGetLastError());
  }
  // weight gradients: dW_l [out,in] = (1/S) dY_l [out,N] . X_l [in,N]^T   (NT GEMM, K = points, split-K)
  // -- one grouped launch: five 64x64 output tiles, K split over the whole chip
  const int outs[5] = {64, 16, 64, 64, 16}, ins[5] = {32, 64, 64, 64, 64};
  GemmParams gp{};
  gp.lda = (int)npad; gp.ldb = (int)npad; gp.K = (int)npad; gp.Mpad = 64; gp.Npad = 64; gp.alpha = 1.f; gp.alpha_dev = scale + 1;
  gp.splitk_ws = (float*)splitk_ws; gp.splitk_ws_bytes = splitk_bytes;
  gp.ngroups = 5;
  gp.c32_beta = accumulate;
  for (int l = 0; l < 5; ++l) {
    gp.grp[l].A = (const half_t*)dump + (size_t)(2 * l + 1) * 128 * npad;
    gp.grp[l].B = (const half_t*)dump + (size_t)(2 * l) * 128 * npad;
    gp.grp[l].M = outs[l]; gp.grp[l].N = ins[l]; gp.grp[l].C32 = w_grads[l]; gp.grp[l].ldc32 = ins[l];
  }
  gp.A = gp.grp[0].A; gp.B = gp.grp[0].B; gp.M = 64; gp.N = 64; gp.C32 = w_grads[0]; gp.ldc32 = ins[0];
  if (int e = launch_gemm_f16(ctx, gp, st)) return e;
  return NERAF_OK;
}

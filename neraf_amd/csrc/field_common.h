// Device/host helpers shared by the radiance-half kernels (field.hip forward, field_bwd.hip backward).
#pragma once
#include "common.h"

namespace {

constexpr int MAX_LEVELS = 16;

struct GridLayout {
  int n_levels;
  float scale[MAX_LEVELS];
  int res[MAX_LEVELS];
  unsigned size[MAX_LEVELS];
  unsigned offset[MAX_LEVELS + 1];
  int hashed[MAX_LEVELS];
};

int make_grid_layout(const neraf_grid_desc* g, GridLayout* L) {
  if (!g || g->n_levels < 1 || g->n_levels > MAX_LEVELS || g->n_features != 2 || g->log2_hashmap_size < 4 ||
      g->log2_hashmap_size > 24 || g->base_res < 2 || g->max_res < g->base_res)
    return NERAF_EINVAL;
  L->n_levels = g->n_levels;
  const double growth = g->n_levels > 1 ? exp(log((double)g->max_res / g->base_res) / (g->n_levels - 1)) : 1.0;
  const float log2g = (float)log2(growth);
  const unsigned T = 1u << g->log2_hashmap_size;
  unsigned off = 0;
  for (int l = 0; l < g->n_levels; ++l) {
    const float scale = exp2f((float)l * log2g) * (float)g->base_res - 1.0f;   // tcnn grid_scale()
    const int res = (int)ceilf(scale) + 1;                                      // tcnn grid_resolution()
    unsigned long long n = (unsigned long long)res * res * res;
    n = (n + 7ull) / 8ull * 8ull;
    const unsigned sz = n > T ? T : (unsigned)n;
    L->scale[l] = scale; L->res[l] = res; L->size[l] = sz; L->offset[l] = off;
    L->hashed[l] = ((unsigned long long)res * res * res) > sz;
    off += sz;
  }
  L->offset[g->n_levels] = off;
  return NERAF_OK;
}

// n / d for any 32-bit n by a multiply-high, one subtraction, one addition and two shifts (Granlund & Montgomery 1994, fig. 4.1): the
// sample -> (ray, index) maps of the gather kernels divide by the runtime samples-per-ray; on a wave-uniform n this is five SCALAR
// instructions, on a per-lane n five vector ones (the compiler's 32-bit udiv is ~25, its 64-bit one far more).
struct FastDiv { unsigned d, m, sh1, sh2; };
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f{d, 0u, 0u, 0u};
  unsigned L = 0;
  while ((1ull << L) < d) ++L;                                                   // ceil(log2 d)
  f.m = (unsigned)(((1ull << 32) * ((1ull << L) - d)) / d + 1ull);
  f.sh1 = L < 1u ? L : 1u; f.sh2 = L > 0u ? L - 1u : 0u;
  return f;
}
__device__ __forceinline__ unsigned fastdiv(unsigned n, const FastDiv& f) {
  const unsigned t = __umulhi(f.m, n);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

// How the SIDE x SIDE slots of ray tile `tile` map to rays (the frame kernels of field.hip): with an image width, the tile is a
// SIDE x SIDE block of pixels of a row-major image (`rows` whole rows, `width` % 8 == 0); without, SIDE^2 consecutive rays.
struct RayTiles { int width, rows; FastDiv divTX; unsigned n_tiles; };
template <int SIDE>
inline RayTiles make_ray_tiles(int R, int width) {
  RayTiles t{};
  if (width > 1) {
    t.width = width; t.rows = R / width; t.divTX = make_fastdiv((unsigned)(width / SIDE));
    t.n_tiles = (unsigned)(width / SIDE) * (unsigned)((t.rows + SIDE - 1) / SIDE);
  } else t.n_tiles = (unsigned)((R + SIDE * SIDE - 1) / (SIDE * SIDE));
  return t;
}
template <int SIDE>
__device__ __forceinline__ bool tile_ray(const RayTiles& t, unsigned tile, unsigned slot, unsigned R, unsigned& ray) {
  if (t.width) {
    const unsigned ty = fastdiv(tile, t.divTX), tx = tile - ty * t.divTX.d;
    const unsigned row = ty * SIDE + slot / SIDE;
    ray = row * (unsigned)t.width + tx * SIDE + slot % SIDE;
    return row < (unsigned)t.rows;
  }
  ray = tile * (SIDE * SIDE) + slot;
  return ray < R;
}

__device__ __forceinline__ float spacing_fn(float x) { return x < 1.f ? 0.5f * x : 1.f - 1.f / (2.f * x); }
__device__ __forceinline__ float spacing_inv(float x) { return x < 0.5f ? 2.f * x : 1.f / (2.f - 2.f * x); }

// One uniform [0,1) draw per (seed, ray): the sampler's single jitter per ray and stage, generated where it is consumed (splitmix64
// finaliser over seed + golden-ratio * (ray + 1); the top 24 bits).  The host layer derives a fresh seed per call and stage.
__device__ __forceinline__ float jitter_u01(unsigned long long seed, int ray) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(ray + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// ---- shared point helpers ---------------------------------------------------------------------------------
__device__ __forceinline__ bool map_position(float& x, float& y, float& z, int mode, const float* aabb) {
  if (mode == 0) {   // SceneContraction(L-inf) then (x+2)/4
    const float mag = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    if (mag >= 1.f) {
      const float k = (2.f - 1.f / mag) / mag;
      x *= k; y *= k; z *= k;
    }
    x = (x + 2.f) * 0.25f; y = (y + 2.f) * 0.25f; z = (z + 2.f) * 0.25f;
  } else {           // SceneBox normalisation (spatial_distortion = None)
    x = (x - aabb[0]) / (aabb[3] - aabb[0]);
    y = (y - aabb[1]) / (aabb[4] - aabb[1]);
    z = (z - aabb[2]) / (aabb[5] - aabb[2]);
  }
  const bool sel = x > 0.f && x < 1.f && y > 0.f && y < 1.f && z > 0.f && z < 1.f;
  if (!sel) { x = 0.f; y = 0.f; z = 0.f; }
  return sel;
}

// ---- one hash-grid level: cell, corner indices, gathers ---------------------------------------------------------------------------
// The eight corner entries of the cell that contains (x,y,z) at one level.  Index arithmetic shared between the corners: the hashed
// index is (cx ^ cy P1 ^ cz P2) & (size - 1) and (cy + 1) P1 = cy P1 + P1 in uint32 arithmetic, so the four (y, z) corner hashes
// cost two multiplies and four xors for the cell, and a corner one xor + and (v_bitop3); the dense index is base + {0,1} + {0,res} +
// {0,res^2}.  Gathers are raw BUFFER loads (32-bit byte offset from a scalar resource for the table: no 64-bit address arithmetic
// per corner -- the generic-pointer form spent two v_mad_u64_u32 / v_lshl_add_u64 per corner on it).
struct LevelCell { float wx, wy, wz; unsigned off[8]; };   // off = BYTE offset of the corner's entry inside its level (index * 4)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const unsigned* __restrict__ table) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(table), 0, 0xFFFFFFFFu, 0x00020000);
}

__device__ __forceinline__ void level_cell(float x, float y, float z, float scale, int res, unsigned size, int hashed, LevelCell& c) {
  // p >= 0.5: the cell is the truncation of p and the in-cell weight its fractional part (v_fract_f32 == p - floor(p) exactly here)
  const float px = fmaf(scale, x, 0.5f), py = fmaf(scale, y, 0.5f), pz = fmaf(scale, z, 0.5f);
  c.wx = __builtin_amdgcn_fractf(px); c.wy = __builtin_amdgcn_fractf(py); c.wz = __builtin_amdgcn_fractf(pz);
  const unsigned ix = (unsigned)px, iy = (unsigned)py, iz = (unsigned)pz;
  if (hashed) {                     // hashed levels have size = 2^log2_T
    // ((x ^ t) & m) << 2 = ((x << 2) ^ (t << 2)) & (m << 2): the hash is taken in byte space, a corner is ONE v_bitop3
    const unsigned m4 = (size - 1u) << 2;
    const unsigned hy0 = iy * 2654435761u, hy1 = hy0 + 2654435761u;
    const unsigned hz0 = iz * 805459861u, hz1 = hz0 + 805459861u;
    const unsigned t00 = (hy0 ^ hz0) << 2, t10 = (hy1 ^ hz0) << 2, t01 = (hy0 ^ hz1) << 2, t11 = (hy1 ^ hz1) << 2;
    const unsigned x0 = ix << 2, x1 = x0 + 4u;
    c.off[0] = (x0 ^ t00) & m4; c.off[1] = (x1 ^ t00) & m4; c.off[2] = (x0 ^ t10) & m4; c.off[3] = (x1 ^ t10) & m4;
    c.off[4] = (x0 ^ t01) & m4; c.off[5] = (x1 ^ t01) & m4; c.off[6] = (x0 ^ t11) & m4; c.off[7] = (x1 ^ t11) & m4;
  } else {
    // dense level: tcnn's `index % size`; corners of the last cell reach res, so idx < res^3+res^2+res < 2*size
    const unsigned r4 = (unsigned)res << 2, r24 = r4 * (unsigned)res, size4 = size << 2;
    const unsigned base = (ix << 2) + iy * r4 + iz * r24;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned i = base + ((k & 1) ? 4u : 0u) + ((k & 2) ? r4 : 0u) + ((k & 4) ? r24 : 0u);
      c.off[k] = min(i, i - size4);          // i >= size4 ? i - size4 : i   (unsigned wrap makes i - size4 huge when i < size4)
    }
  }
}

// trilinear weight of corner k = (fx * fy) * fz with f = w or 1 - w per axis bit (the product order every consumer shares)
// (pairs of products as 2-wide vectors: v_pk_mul_f32 does two fp32 multiplies per instruction)
__device__ __forceinline__ void corner_weights(const LevelCell& c, float (&w)[8]) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const float fy0 = 1.f - c.wy, fz0 = 1.f - c.wz;
  const f32x2 fx = {1.f - c.wx, c.wx};
  const f32x2 xy0 = fx * fy0, xy1 = fx * c.wy;
  const f32x2 w01 = xy0 * fz0, w23 = xy1 * fz0, w45 = xy0 * c.wz, w67 = xy1 * c.wz;
  w[0] = w01[0]; w[1] = w01[1]; w[2] = w23[0]; w[3] = w23[1]; w[4] = w45[0]; w[5] = w45[1]; w[6] = w67[0]; w[7] = w67[1];
}

// What a gather costs (tools/microbench/gather_rate.hip, profiles/r04_gather_rate_microbench.txt): with lanes spread over many cache
// lines, its distinct lines (width and re-reads of the same lines are free).  On a DENSE level the x-neighbour of a corner is the next
// entry: PAIRS fetches the two with ONE 8-byte load (4 instructions per level instead of 8) -- worth 2-3 % on the frame's proposal
// kernel, nothing elsewhere, and only for kernels whose `pairs` flag is wave-uniform (a per-lane choice between the two forms is a
// divergent branch around every load: tried on hashed levels with even cell x, proposal density 128 -> 144 us, field query on frames
// 233 -> 405 us).  The one exception to "next entry" is an x0 corner that is the level's LAST entry (its neighbour wraps to entry 0:
// tcnn's `index % size`): a rare lane-level fix-up load; the 8-byte load then reads 4 bytes of the NEXT level, so the caller never
// asks for pairs on the table's last level.
// UNIFORM: `offset` (the level's first entry) is the same for every lane of the wave: it rides in the instruction's scalar offset;
// otherwise (a wave whose lanes hold different levels: the fused field kernels) it is added per lane.
template <bool UNIFORM, bool PAIRS = false>
__device__ __forceinline__ void gather_corners(const unsigned* __restrict__ table, unsigned offset, const LevelCell& c, unsigned (&raw)[8],
                                               bool pairs = false, unsigned size = 0) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t rs = table_rsrc(table);
  const unsigned off4 = offset << 2;
  if (PAIRS && UNIFORM && pairs) {
    const unsigned soff = __builtin_amdgcn_readfirstlane(off4), last = (size << 2) - 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, c.off[2 * j], soff, 0);
      raw[2 * j] = v[0]; raw[2 * j + 1] = v[1];
      if (c.off[2 * j] == last) raw[2 * j + 1] = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, soff, 0);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
    raw[k] = UNIFORM ? __builtin_amdgcn_raw_buffer_load_b32(rs, c.off[k], __builtin_amdgcn_readfirstlane(off4), 0)
                     : __builtin_amdgcn_raw_buffer_load_b32(rs, c.off[k] + off4, 0, 0);
}

// acc + w * (fp32 of the low / high fp16 of `raw`) in ONE instruction: v_fma_mix_f32 reads an fp16 operand in place (the conversion is
// exact, the result is the fused multiply-add of the converted value: bit-identical to fmaf(w, (float)h, acc))
__device__ __forceinline__ float fma_f16lo(float w, unsigned raw, float acc) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(raw), "v"(acc));
  return r;
}
__device__ __forceinline__ float fma_f16hi(float w, unsigned raw, float acc) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(raw), "v"(acc));
  return r;
}

// trilinear interpolation of a gathered cell; table entries are half2 (2 features).  Kernels that walk several levels per lane call
// level_cell + gather_corners for ALL of them first and interpolate afterwards: every gather of the lane is then in flight at once
// (written as one encode_level per level, the compiler waits for each level's eight loads before it starts the next level's).
__device__ __forceinline__ void interpolate_level(const LevelCell& c, const unsigned (&raw)[8], float& f0, float& f1) {
  float w[8];
  corner_weights(c, w);
  f0 = 0.f; f1 = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    f0 = fma_f16lo(w[k], raw[k], f0);
    f1 = fma_f16hi(w[k], raw[k], f1);
  }
}

// one hash-grid level, trilinear
template <bool UNIFORM = false>
__device__ __forceinline__ void encode_level(const unsigned* __restrict__ table, float x, float y, float z, float scale, int res,
                                             unsigned size, unsigned offset, int hashed, float& f0, float& f1) {
  LevelCell c;
  level_cell(x, y, z, scale, res, size, hashed, c);
  unsigned raw[8];
  gather_corners<UNIFORM>(table, offset, c, raw);
  interpolate_level(c, raw, f0, f1);
}

// interpolation that also returns d f0 / d(x,y,z) and d f1 / d(x,y,z) (in mapped [0,1] coordinates) from the SAME eight gathers: the
// field backward recomputes the encoding anyway.  Measured on the 196,608-sample render batch (the kernel runs at one wave per SIMD):
// baseline 210 us; a second gather pass inside the kernel +75 us; the position gradient as its own high-occupancy kernel on the
// stored d enc +103 us (a second full table walk); this form +27 us.
__device__ __forceinline__ void interpolate_level_grad(const LevelCell& c, const unsigned (&raw)[8], float scale, float& f0, float& f1,
                                                       float (&d0)[3], float (&d1)[3]) {
  const float wx = c.wx, wy = c.wy, wz = c.wz;
  f0 = 0.f; f1 = 0.f;
  float a0x = 0.f, a0y = 0.f, a0z = 0.f, a1x = 0.f, a1y = 0.f, a1z = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const half2v v = *reinterpret_cast<const half2v*>(&raw[k]);
    const float v0 = (float)v[0], v1 = (float)v[1];
    const float fx = (k & 1) ? wx : 1.f - wx, fy = (k & 2) ? wy : 1.f - wy, fz = (k & 4) ? wz : 1.f - wz;
    const float w = fx * fy * fz;
    f0 = fmaf(w, v0, f0); f1 = fmaf(w, v1, f1);
    const float gx = ((k & 1) ? 1.f : -1.f) * fy * fz, gy = ((k & 2) ? 1.f : -1.f) * fx * fz, gz = ((k & 4) ? 1.f : -1.f) * fx * fy;
    a0x = fmaf(gx, v0, a0x); a0y = fmaf(gy, v0, a0y); a0z = fmaf(gz, v0, a0z);
    a1x = fmaf(gx, v1, a1x); a1y = fmaf(gy, v1, a1y); a1z = fmaf(gz, v1, a1z);
  }
  d0[0] = scale * a0x; d0[1] = scale * a0y; d0[2] = scale * a0z;
  d1[0] = scale * a1x; d1[1] = scale * a1y; d1[2] = scale * a1z;
}

__device__ __forceinline__ void encode_level_grad(const unsigned* __restrict__ table, float x, float y, float z, float scale, int res,
                                                  unsigned size, unsigned offset, int hashed, float& f0, float& f1,
                                                  float (&d0)[3], float (&d1)[3]) {
  LevelCell c;
  level_cell(x, y, z, scale, res, size, hashed, c);
  unsigned raw[8];
  gather_corners<false>(table, offset, c, raw);
  interpolate_level_grad(c, raw, scale, f0, f1, d0, d1);
}

// d/d(x,y,z) of  g0 * f0 + g1 * f1  for one level (f = the trilinear interpolation of encode_level), ACCUMULATED into (dx,dy,dz):
// the hash-grid input gradient (tiny-cuda-nn computes it for its inputs; here it feeds the camera-pose optimizer).  The weights
// are products of (w | 1-w) per axis, so the derivative along an axis replaces that axis' factor by (+1 | -1) * scale.
__device__ __forceinline__ void encode_level_dpos(const unsigned* __restrict__ table, float x, float y, float z, float scale, int res,
                                                  unsigned size, unsigned offset, int hashed, float g0, float g1,
                                                  float& dx, float& dy, float& dz) {
  LevelCell c;
  level_cell(x, y, z, scale, res, size, hashed, c);
  unsigned raw[8];
  gather_corners<false>(table, offset, c, raw);
  const float wx = c.wx, wy = c.wy, wz = c.wz;
  float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const half2v v = *reinterpret_cast<const half2v*>(&raw[k]);
    const float val = g0 * (float)v[0] + g1 * (float)v[1];
    const float fx = (k & 1) ? wx : 1.f - wx, fy = (k & 2) ? wy : 1.f - wy, fz = (k & 4) ? wz : 1.f - wz;
    ax = fmaf(((k & 1) ? 1.f : -1.f) * fy * fz, val, ax);
    ay = fmaf(((k & 2) ? 1.f : -1.f) * fx * fz, val, ay);
    az = fmaf(((k & 4) ? 1.f : -1.f) * fx * fy, val, az);
  }
  dx = fmaf(scale, ax, dx); dy = fmaf(scale, ay, dy); dz = fmaf(scale, az, dz);
}

// transpose-Jacobian of map_position (mode 0: L-inf contraction, then (x+2)/4) applied to (gx,gy,gz), at the UNMAPPED point (x,y,z)
__device__ __forceinline__ void map_position_jt(float x, float y, float z, float& gx, float& gy, float& gz) {
  const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
  const float mag = fmaxf(ax, fmaxf(ay, az));
  if (mag >= 1.f) {
    const float inv = 1.f / mag;
    const float c = (2.f - inv) * inv;                       // (2 - 1/m) / m
    const float dc = (-2.f + 2.f * inv) * inv * inv;         // d c / d m = -2/m^2 + 2/m^3
    const float dot = x * gx + y * gy + z * gz;
    const int k = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    const float sk = (k == 0 ? x : (k == 1 ? y : z)) >= 0.f ? 1.f : -1.f;
    gx *= c; gy *= c; gz *= c;
    const float extra = sk * dc * dot;
    if (k == 0) gx += extra; else if (k == 1) gy += extra; else gz += extra;
  }
  gx *= 0.25f; gy *= 0.25f; gz *= 0.25f;
}

// d/d(dx,dy,dz) of  sum_r g[r] * sh[r]  for the four SH components of sh4_quarter(q, ...), ACCUMULATED into (ox,oy,oz)
__device__ __forceinline__ void sh4_quarter_dpos(int q, float dx, float dy, float dz, const float (&g)[4], float& ox, float& oy, float& oz) {
  const float x2 = dx * dx, y2 = dy * dy, z2 = dz * dz;
  if (q == 0) {
    oy += -0.48860251190291987f * g[1]; oz += 0.48860251190291987f * g[2]; ox += -0.48860251190291987f * g[3];
  } else if (q == 1) {
    ox += 1.0925484305920792f * dy * g[0];                 oy += 1.0925484305920792f * dx * g[0];
    oy += -1.0925484305920792f * dz * g[1];                oz += -1.0925484305920792f * dy * g[1];
    oz += 2.f * 0.94617469575755997f * dz * g[2];
    ox += -1.0925484305920792f * dz * g[3];                oz += -1.0925484305920792f * dx * g[3];
  } else if (q == 2) {
    ox += 2.f * 0.54627421529603959f * dx * g[0];          oy += -2.f * 0.54627421529603959f * dy * g[0];
    ox += 0.59004358992664352f * dy * (-6.f * dx) * g[1];  oy += 0.59004358992664352f * (-3.f * x2 + 3.f * y2) * g[1];
    ox += 2.8906114426405538f * dy * dz * g[2];            oy += 2.8906114426405538f * dx * dz * g[2];   oz += 2.8906114426405538f * dx * dy * g[2];
    oy += 0.45704579946446572f * (1.f - 5.f * z2) * g[3];  oz += 0.45704579946446572f * dy * (-10.f * dz) * g[3];
  } else {
    oz += 0.3731763325901154f * (15.f * z2 - 3.f) * g[0];
    ox += 0.45704579946446572f * (1.f - 5.f * z2) * g[1];  oz += 0.45704579946446572f * dx * (-10.f * dz) * g[1];
    ox += 1.4453057213202769f * dz * 2.f * dx * g[2];      oy += -1.4453057213202769f * dz * 2.f * dy * g[2];   oz += 1.4453057213202769f * (x2 - y2) * g[2];
    ox += 0.59004358992664352f * (-3.f * x2 + 3.f * y2) * g[3];   oy += 0.59004358992664352f * dx * 6.f * dy * g[3];
  }
}

__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float n = __shfl_up(v, o);
    if (lane >= o) v += n;
  }
  return v;
}

// EXCLUSIVE prefix from an inclusive one: the previous lane's inclusive value, not `incl - v`.  The optical depths of a trained
// field reach 1e15 and beyond at surfaces (densities of 1e14-1e18 after a few thousand iterations of the trajectory scene):
// `incl - v` then cancels catastrophically -- the sample BEHIND a surface sample got transmittance exp(-0) = 1 instead of 0, its
// weight came out 1 next to the surface sample's 1, and the colour / the gradient of those rays were garbage (found by
// tools/long_trajectory_curve.py against the CPU oracle on the same state; nerfstudio's get_weights takes the cumulative sum of
// the PREVIOUS samples directly, RaySamples.get_weights [NS-recall]).
__device__ __forceinline__ float wave_excl_from_incl(float incl, int lane) {
  const float prev = __shfl_up(incl, 1);
  return lane == 0 ? 0.f : prev;
}

// Sum over the LATER lanes (exclusive suffix), summed directly -- not `total - inclusive prefix`: behind an opaque sample every
// later term is exactly zero and so must this sum be (torch's cumsum backward is a reversed cumsum); the difference of two
// differently ordered fp32 sums leaves ~1e-7 of the total there, which the density's backward multiplies by up to e^15.
__device__ __forceinline__ float wave_suffix_excl_scan(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float n = __shfl_down(v, o);
    if (lane + o < 64) v += n;
  }
  const float next = __shfl_down(v, 1);
  return lane == 63 ? 0.f : next;
}


// gradient scatter of one level: table_grad[(offset+idx)*2 + f] += w_corner * g_f  (fp32 atomics, 8 corners)
__device__ __forceinline__ void scatter_level(float* __restrict__ tgrad, float x, float y, float z, float scale, int res,
                                              unsigned size, unsigned offset, int hashed, float g0, float g1) {
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
    float* t = tgrad + (size_t)(offset + idx) * 2;
    atomicAdd(t, w * g0);
    atomicAdd(t + 1, w * g1);
  }
}

// Same scatter with an in-register pre-reduction: the WIDTH consecutive lanes of a group hold consecutive samples of
// a ray, which at all but the finest levels fall into the same grid cell.  A segmented scan over runs of equal table
// indices leaves each run's sum in its last lane, and only that lane issues the atomics (global float atomics are the
// bottleneck of the backward: ~2e10 scattered adds/s chip-wide).  Merging equal indices is valid across rays too.
// Every lane of the group must call this (zero gradients for lanes without work).
template <int WIDTH>
__device__ __forceinline__ void scatter_level_seg(float* __restrict__ tgrad, float x, float y, float z, float scale, int res,
                                                  unsigned size, unsigned offset, int hashed, float g0, float g1, int pos) {
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
    float v0 = w * g0, v1 = w * g1;
    const unsigned prev = __shfl_up(idx, 1, WIDTH);
    const unsigned next = __shfl_down(idx, 1, WIDTH);
    int head = (pos == 0 || prev != idx) ? 1 : 0;
    const bool tail = (pos == WIDTH - 1) || next != idx;
#pragma unroll
    for (int o = 1; o < WIDTH; o <<= 1) {
      const float a0 = __shfl_up(v0, o, WIDTH), a1 = __shfl_up(v1, o, WIDTH);
      const int hu = __shfl_up(head, o, WIDTH);
      if (pos >= o) {
        if (!head) { v0 += a0; v1 += a1; }
        head |= hu;
      }
    }
    if (tail && (v0 != 0.f || v1 != 0.f)) {
      float* t = tgrad + (size_t)(offset + idx) * 2;
      atomicAdd(t, v0);
      atomicAdd(t + 1, v1);
    }
  }
}

// SH degree 4 components 4q..4q+3 of direction (dx,dy,dz) (tiny-cuda-nn constants), q = lane >> 4
__device__ __forceinline__ void sh4_quarter(int q, float dx, float dy, float dz, float (&sh)[4]) {
  const float xy = dx * dy, xz = dx * dz, yz = dy * dz, x2 = dx * dx, y2 = dy * dy, z2 = dz * dz;
  if (q == 0) {
    sh[0] = 0.28209479177387814f; sh[1] = -0.48860251190291987f * dy;
    sh[2] = 0.48860251190291987f * dz; sh[3] = -0.48860251190291987f * dx;
  } else if (q == 1) {
    sh[0] = 1.0925484305920792f * xy; sh[1] = -1.0925484305920792f * yz;
    sh[2] = 0.94617469575755997f * z2 - 0.31539156525251999f; sh[3] = -1.0925484305920792f * xz;
  } else if (q == 2) {
    sh[0] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2; sh[1] = 0.59004358992664352f * dy * (-3.0f * x2 + y2);
    sh[2] = 2.8906114426405538f * xy * dz; sh[3] = 0.45704579946446572f * dy * (1.0f - 5.0f * z2);
  } else {
    sh[0] = 0.3731763325901154f * dz * (5.0f * z2 - 3.0f); sh[1] = 0.45704579946446572f * dx * (1.0f - 5.0f * z2);
    sh[2] = 1.4453057213202769f * dz * (x2 - y2); sh[3] = 0.59004358992664352f * dx * (-x2 + 3.0f * y2);
  }
}

// two accumulator blocks -> one fp16 B fragment, ReLU applied AFTER the conversion on the packed pairs (round-to-nearest is monotonic
// and maps 0 to 0, so relu(round(x)) == round(relu(x)); v_pk_max_f16 with 0 returns 0 for a NaN like fmaxf): 4 conversions + 4
// packed maxima for 8 values instead of 8 canonicalising maxima + 4 conversions
__device__ __forceinline__ half8 pack_relu(const f32x4& a, const f32x4& b, bool relu) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 u;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const half2v lo = {(half_t)a[2 * r], (half_t)a[2 * r + 1]}, hi = {(half_t)b[2 * r], (half_t)b[2 * r + 1]};
    unsigned ul = __builtin_bit_cast(unsigned, lo), uh = __builtin_bit_cast(unsigned, hi);
    if (relu) {
      asm("v_pk_max_f16 %0, %1, 0" : "=v"(ul) : "v"(ul));
      asm("v_pk_max_f16 %0, %1, 0" : "=v"(uh) : "v"(uh));
    }
    u[r] = ul; u[2 + r] = uh;
  }
  return __builtin_bit_cast(half8, u);
}

}  // namespace

// SO3xR3 camera-pose refinement applied to a ray bundle (nerfstudio CameraOptimizer.apply_to_raybundle, configured at
// NeRAF_config.py:97): per ray  o' = o + t[cam],  d' = exp(w[cam]) d  with pose_adjustment[cam] = (t | w), exp by Rodrigues' formula
// with nerfstudio's small-angle guard (the squared angle is clamped at 1e-4 before the square root).  Forward and backward are one
// launch each; the PyTorch expression they replace (neraf_amd/cameras.py, kept as the reference of the tests) is ~35 small launches
// per training step once autograd has replayed it.
#include "common.h"

namespace {

struct Rod { float f1, f2, df1, df2, a; bool clamped; };

__device__ __forceinline__ Rod rodrigues(float wx, float wy, float wz) {
  Rod r;
  const float n = wx * wx + wy * wy + wz * wz;
  r.clamped = n < 1e-4f;
  const float a2 = r.clamped ? 1e-4f : n;
  const float a = sqrtf(a2), inv = 1.f / a;
  const float s = sinf(a), c = cosf(a);
  r.a = a;
  r.f1 = inv * s;
  r.f2 = inv * inv * (1.f - c);
  r.df1 = (c * a - s) * inv * inv;                          // d/da sin(a)/a
  r.df2 = (s * a - 2.f * (1.f - c)) * inv * inv * inv;      // d/da (1 - cos a)/a^2
  return r;
}

__device__ __forceinline__ void cross(float ax, float ay, float az, float bx, float by, float bz, float& cx, float& cy, float& cz) {
  cx = ay * bz - az * by; cy = az * bx - ax * bz; cz = ax * by - ay * bx;
}

__global__ __launch_bounds__(256) void camera_apply_kernel(const float* __restrict__ pose, const int* __restrict__ cam, const float* __restrict__ o,
                                                          const float* __restrict__ d, int R, float* __restrict__ o_out, float* __restrict__ d_out,
                                                          int n_cams, float w_t, float w_r, float* __restrict__ reg_out3) {
  // nerfstudio's CameraOptimizer.get_loss_dict / get_metrics_dict on the way (workgroup 0): reg_out3 = {mean|t| * trans_l2_penalty +
  // mean|w| * rot_l2_penalty (w_t, w_r carry the 1/n), |t|_F, |w|_F}
  if (reg_out3 && blockIdx.x == 0) {
    __shared__ float red[4][4];
    float a = 0.f, b = 0.f, a2 = 0.f, b2 = 0.f;
    for (int c = threadIdx.x; c < n_cams; c += 256) {
      const float* q = pose + (size_t)c * 6;
      const float t2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], r2 = q[3] * q[3] + q[4] * q[4] + q[5] * q[5];
      a += sqrtf(t2); b += sqrtf(r2); a2 += t2; b2 += r2;
    }
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) { a += __shfl_xor(a, o2); b += __shfl_xor(b, o2); a2 += __shfl_xor(a2, o2); b2 += __shfl_xor(b2, o2); }
    if ((threadIdx.x & 63) == 0) { float* r4 = red[threadIdx.x >> 6]; r4[0] = a; r4[1] = b; r4[2] = a2; r4[3] = b2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float s[4] = {0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < 4; ++w) for (int k = 0; k < 4; ++k) s[k] += red[w][k];
      reg_out3[0] = s[0] * w_t + s[1] * w_r; reg_out3[1] = sqrtf(s[2]); reg_out3[2] = sqrtf(s[3]);
    }
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R) return;
  const float* p = pose + (size_t)cam[i] * 6;
  const float wx = p[3], wy = p[4], wz = p[5];
  const Rod r = rodrigues(wx, wy, wz);
  const float dx = d[i * 3], dy = d[i * 3 + 1], dz = d[i * 3 + 2];
  float kx, ky, kz, k2x, k2y, k2z;
  cross(wx, wy, wz, dx, dy, dz, kx, ky, kz);                // K d
  cross(wx, wy, wz, kx, ky, kz, k2x, k2y, k2z);             // K^2 d
  d_out[i * 3] = dx + r.f1 * kx + r.f2 * k2x;
  d_out[i * 3 + 1] = dy + r.f1 * ky + r.f2 * k2y;
  d_out[i * 3 + 2] = dz + r.f1 * kz + r.f2 * k2z;
  o_out[i * 3] = o[i * 3] + p[0]; o_out[i * 3 + 1] = o[i * 3 + 1] + p[1]; o_out[i * 3 + 2] = o[i * 3 + 2] + p[2];
}

// d pose[cam] += (d_o | J_w^T d_d); runs of rays with the same camera inside a wave are summed first (the sampler draws rays image
// by image or at random: either way one atomic per run instead of one per ray)
// one ray's contribution to d pose[cam]: (d_o | J_w^T d_d)
__device__ __forceinline__ void ray_pose_grad(const float* __restrict__ pose, int c, const float* __restrict__ d, const float* __restrict__ g_o,
                                              const float* __restrict__ g_d, int gs, int i, float (&g)[6]) {
  const float* p = pose + (size_t)c * 6;
  const float wx = p[3], wy = p[4], wz = p[5];
  const Rod r = rodrigues(wx, wy, wz);
  const float dx = d[i * 3], dy = d[i * 3 + 1], dz = d[i * 3 + 2];
  const float ux = g_d[(size_t)i * gs], uy = g_d[(size_t)i * gs + 1], uz = g_d[(size_t)i * gs + 2];
  g[0] = g_o[(size_t)i * gs]; g[1] = g_o[(size_t)i * gs + 1]; g[2] = g_o[(size_t)i * gs + 2];
  float kx, ky, kz, k2x, k2y, k2z;
  cross(wx, wy, wz, dx, dy, dz, kx, ky, kz);
  cross(wx, wy, wz, kx, ky, kz, k2x, k2y, k2z);
  // radial term: (df1 K d + df2 K^2 d) . u * w_i / a   (zero inside the small-angle guard, where a is constant)
  const float radial = r.clamped ? 0.f : ((r.df1 * kx + r.df2 * k2x) * ux + (r.df1 * ky + r.df2 * k2y) * uy + (r.df1 * kz + r.df2 * k2z) * uz) / r.a;
  const float w[3] = {wx, wy, wz};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float ex = a == 0 ? 1.f : 0.f, ey = a == 1 ? 1.f : 0.f, ez = a == 2 ? 1.f : 0.f;
    float t1x, t1y, t1z, t2x, t2y, t2z, t3x, t3y, t3z;
    cross(ex, ey, ez, dx, dy, dz, t1x, t1y, t1z);          // e_a x d
    cross(ex, ey, ez, kx, ky, kz, t2x, t2y, t2z);          // e_a x (w x d)
    cross(wx, wy, wz, t1x, t1y, t1z, t3x, t3y, t3z);       // w x (e_a x d)
    g[3 + a] = r.f1 * (t1x * ux + t1y * uy + t1z * uz) + r.f2 * ((t2x + t3x) * ux + (t2y + t3y) * uy + (t2z + t3z) * uz) + radial * w[a];
  }
}

__global__ __launch_bounds__(256) void camera_apply_bwd_kernel(const float* __restrict__ pose, const int* __restrict__ cam, const float* __restrict__ d,
                                                              const float* __restrict__ g_o, const float* __restrict__ g_d, int gs, int R,
                                                              float* __restrict__ g_pose) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool valid = i < R;
  const int c = valid ? cam[i] : -1;
  float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (valid) ray_pose_grad(pose, c, d, g_o, g_d, gs, i, g);
  // segmented sum over runs of equal camera index among consecutive lanes
  const int prev = __shfl_up(c, 1), next = __shfl_down(c, 1);
  int head = (lane == 0 || prev != c) ? 1 : 0;
  const bool tail = (lane == 63) || next != c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float up[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) up[k] = __shfl_up(g[k], o);
    const int hu = __shfl_up(head, o);
    if (lane >= o) {
      if (!head) {
#pragma unroll
        for (int k = 0; k < 6; ++k) g[k] += up[k];
      }
      head |= hu;
    }
  }
  if (valid && tail) {
#pragma unroll
    for (int k = 0; k < 6; ++k) if (g[k] != 0.f) atomicAdd(g_pose + (size_t)c * 6 + k, g[k]);
  }
}

// Deterministic mode: one wavefront per camera walks ALL rays in index order (lane-strided), adds the contributions of its own
// rays and folds the lanes with a fixed xor-tree: no atomics, the same bits every run (R x n_cams index compares: 0.9 M at the
// bench shape).
__global__ __launch_bounds__(256) void camera_apply_bwd_det_kernel(const float* __restrict__ pose, const int* __restrict__ cam, const float* __restrict__ d,
                                                                  const float* __restrict__ g_o, const float* __restrict__ g_d, int gs, int R, int n_cams,
                                                                  float* __restrict__ g_pose) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= n_cams) return;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = lane; i < R; i += 64)
    if (cam[i] == c) {
      float g[6];
      ray_pose_grad(pose, c, d, g_o, g_d, gs, i, g);
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] += g[k];
    }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) g_pose[(size_t)c * 6 + k] += acc[k];
  }
}

// d pose = g_reg * d regulariser / d pose (the zero vector has subgradient 0, as torch's norm backward), or zero: initialises the
// buffer the ray-gradient kernel then adds into
__global__ void camera_reg_bwd_init_kernel(const float* __restrict__ pose, int n_cams, float w_t, float w_r, const float* __restrict__ g_reg,
                                           float* __restrict__ g_pose) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cams * 2) return;
  const float g = g_reg ? *g_reg : 0.f;
  const float* q = pose + (size_t)i * 3;
  const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  const float k = (g != 0.f && n > 0.f) ? g * ((i & 1) ? w_r : w_t) / n : 0.f;
  g_pose[(size_t)i * 3] = k * q[0]; g_pose[(size_t)i * 3 + 1] = k * q[1]; g_pose[(size_t)i * 3 + 2] = k * q[2];
}

// ---- camera -> rays (nerfstudio Cameras.generate_rays as Model.get_outputs_for_camera uses it, NeRAF_model.py:70-79; restated in
// neraf_amd/cameras.py [NS-recall]): pixel centre -> image-plane coordinates -> OpenCV radial-tangential undistortion by 10 Newton
// steps (camera_utils.radial_and_tangential_undistort: the step is zero where the Jacobian is singular) -> OpenGL camera frame
// (+x right, +y up, looking along -z) -> rotated by camera_to_world, normalised; origin = camera centre.  One thread per ray; a
// 684 x 1024 eval frame is one launch instead of ~150 element-wise torch launches and a 700k-batch 3x3 matmul.
struct RayGenArgs {
  const float* c2w;        // [n_cams,3,4]
  const float* fx; const float* fy; const float* cx; const float* cy;   // [n_cams] each
  const float* dist;       // [n_cams,6] = k1,k2,k3,k4,p1,p2 or null
  const long long* cam;    // [R] camera index per ray, or null: `cam_single` for every ray
  const float* coords;     // [R,2] = (row, col) in pixels, or null: pixel centres of a `width`-wide image, row-major
  int cam_single, R, width;
  float* origins; float* dirs; long long* cam_out;   // [R,3], [R,3], [R] (optional)
  int n_cams;
};

__global__ __launch_bounds__(256) void camera_rays_kernel(RayGenArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.R) return;
  const int c = a.cam ? (int)a.cam[i] : a.cam_single;
  if ((unsigned)c >= (unsigned)a.n_cams) {
    // a per-ray camera index outside the camera set (the CPU path raises IndexError): no out-of-bounds read -- the ray is NaN and
    // its camera index -1, which no consumer can mistake for a result
    const float nan = __builtin_nanf("");
    for (int k = 0; k < 3; ++k) { a.origins[3 * (size_t)i + k] = nan; a.dirs[3 * (size_t)i + k] = nan; }
    if (a.cam_out) a.cam_out[i] = -1;
    return;
  }
  float y, x;
  if (a.coords) { y = a.coords[2 * (size_t)i]; x = a.coords[2 * (size_t)i + 1]; }
  else { y = (float)(i / a.width) + 0.5f; x = (float)(i % a.width) + 0.5f; }
  const float xd = (x - a.cx[c]) / a.fx[c], yd = (y - a.cy[c]) / a.fy[c];
  float xc = xd, yc = yd;
  if (a.dist) {
    const float* q = a.dist + (size_t)c * 6;
    const float k1 = q[0], k2 = q[1], k3 = q[2], k4 = q[3], p1 = q[4], p2 = q[5];
    if (k1 != 0.f || k2 != 0.f || k3 != 0.f || k4 != 0.f || p1 != 0.f || p2 != 0.f) {
#pragma unroll 1
      for (int it = 0; it < 10; ++it) {
        const float r = xc * xc + yc * yc;
        const float d = 1.f + r * (k1 + r * (k2 + r * (k3 + r * k4)));
        const float fx = d * xc + 2.f * p1 * xc * yc + p2 * (r + 2.f * xc * xc) - xd;
        const float fy = d * yc + 2.f * p2 * xc * yc + p1 * (r + 2.f * yc * yc) - yd;
        const float d_r = k1 + r * (2.f * k2 + r * (3.f * k3 + r * 4.f * k4));
        const float d_x = 2.f * xc * d_r, d_y = 2.f * yc * d_r;
        const float fx_x = d + d_x * xc + 2.f * p1 * yc + 6.f * p2 * xc;
        const float fx_y = d_y * xc + 2.f * p1 * xc + 2.f * p2 * yc;
        const float fy_x = d_x * yc + 2.f * p2 * yc + 2.f * p1 * xc;
        const float fy_y = d + d_y * yc + 2.f * p2 * xc + 6.f * p1 * yc;
        const float den = fy_x * fx_y - fx_x * fy_y;
        const bool ok = fabsf(den) > 1e-9f;
        xc += ok ? (fx * fy_y - fy * fx_y) / den : 0.f;
        yc += ok ? (fy * fx_x - fx * fy_x) / den : 0.f;
      }
    }
  }
  const float* m = a.c2w + (size_t)c * 12;
  const float dx = xc, dy = -yc, dz = -1.f;
  float wx = m[0] * dx + m[1] * dy + m[2] * dz;
  float wy = m[4] * dx + m[5] * dy + m[6] * dz;
  float wz = m[8] * dx + m[9] * dy + m[10] * dz;
  const float inv = 1.f / sqrtf(wx * wx + wy * wy + wz * wz);
  float* o = a.origins + (size_t)i * 3; float* dd = a.dirs + (size_t)i * 3;
  o[0] = m[3]; o[1] = m[7]; o[2] = m[11];
  dd[0] = wx * inv; dd[1] = wy * inv; dd[2] = wz * inv;
  if (a.cam_out) a.cam_out[i] = c;
}

}  // namespace

extern "C" int neraf_camera_apply(neraf_ctx* ctx, const float* pose_adjustment, const int32_t* cam_idx, const float* origins,
                                  const float* dirs, int R, float* origins_out, float* dirs_out, int n_cams, float w_trans, float w_rot,
                                  float* reg_out3, neraf_stream_t stream) {
  if (!pose_adjustment || !cam_idx || !origins || !dirs || R <= 0 || !origins_out || !dirs_out || (reg_out3 && n_cams <= 0))
    return neraf_fail(ctx, NERAF_EINVAL, "camera_apply: bad arguments");
  hipLaunchKernelGGL(camera_apply_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, pose_adjustment, cam_idx, origins, dirs, R,
                     origins_out, dirs_out, n_cams, w_trans, w_rot, reg_out3);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_camera_apply_bwd(neraf_ctx* ctx, const float* pose_adjustment, const int32_t* cam_idx, const float* dirs,
                                      const float* d_origins, const float* d_dirs, int g_stride, int R, int n_cams, float w_trans,
                                      float w_rot, const float* g_reg, float* d_pose, neraf_stream_t stream) {
  if (!pose_adjustment || !cam_idx || !dirs || R <= 0 || !d_pose || n_cams <= 0 || g_stride < 3 || (!d_origins) != (!d_dirs))
    return neraf_fail(ctx, NERAF_EINVAL, "camera_apply_bwd: bad arguments");
  hipLaunchKernelGGL(camera_reg_bwd_init_kernel, dim3((n_cams * 2 + 255) / 256), dim3(256), 0, (hipStream_t)stream, pose_adjustment, n_cams,
                     w_trans, w_rot, g_reg, d_pose);
  if (d_origins && neraf_deterministic())
    hipLaunchKernelGGL(camera_apply_bwd_det_kernel, dim3((n_cams + 3) / 4), dim3(256), 0, (hipStream_t)stream, pose_adjustment, cam_idx, dirs,
                       d_origins, d_dirs, g_stride, R, n_cams, d_pose);
  else if (d_origins)
    hipLaunchKernelGGL(camera_apply_bwd_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, pose_adjustment, cam_idx, dirs, d_origins,
                       d_dirs, g_stride, R, d_pose);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

extern "C" int neraf_camera_rays(neraf_ctx* ctx, const float* c2w, const float* fx, const float* fy, const float* cx, const float* cy,
                                 const float* distortion, int n_cams,
                                 const int64_t* cam_idx, int cam_single, const float* coords, int R, int width, float* origins,
                                 float* dirs, int64_t* cam_out, neraf_stream_t stream) {
  if (!c2w || !fx || !fy || !cx || !cy || n_cams <= 0 || R <= 0 || !origins || !dirs || (!coords && width <= 0) ||
      (!cam_idx && (cam_single < 0 || cam_single >= n_cams)))
    return neraf_fail(ctx, NERAF_EINVAL, "camera_rays: bad arguments");
  RayGenArgs a{c2w, fx, fy, cx, cy, distortion, (const long long*)cam_idx, coords, cam_single, R, width, origins, dirs, (long long*)cam_out, n_cams};
  hipLaunchKernelGGL(camera_rays_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  NERAF_HIP_CHECK(ctx, hipGetLastError());
  return NERAF_OK;
}

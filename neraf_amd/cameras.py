"""Cameras -> ray bundles and the SO3xR3 camera-pose optimizer: the two nerfstudio pieces in front of
``NeRAFVisionModel.get_outputs`` (K1 of SURVEY.md 2.1) that the reference configures but does not contain.

  * ``Cameras.generate_rays`` is what ``Model.get_outputs_for_camera`` (called at NeRAF_model.py:70-79,
    NeRAF_pipeline.py:277, :325) does with a camera before chunking: pixel centres -> (undistorted) image-plane
    coordinates -> unit directions rotated by camera_to_world, origins = camera centre.  The RAF scenes ship OPENCV
    intrinsics with radial / tangential distortion (data/RAF/*/transforms.json: fl_x, fl_y, cx, cy, k1..k4, p1, p2,
    684 x 1024).
  * ``CameraOptimizer(mode="SO3xR3")`` is the ``camera_optimizer`` of NeRAF_config.py:94-98: a 6-vector per training
    camera (translation | so(3) log), applied to the ray bundle in training, with nerfacto's L2 regulariser.

Both are restated from nerfstudio's published behaviour [NS-recall] (its source is not under /root/reference): parity
unpinned, guarded by the property tests in tests/test_cameras.py.  The tensor expressions below are the CPU form (data
preparation, tests); on the device ``generate_rays`` is one HIP launch (csrc/camera.hip): as a tensor expression the 700k rays of
an eval frame cost ~150 small launches and a 700k-batch 3x3 matmul -- 12 ms of a 34 ms frame (profiles/r04_a_eval_kernel_stats.csv).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from .vision import RayBundle


def _undistort(xd: torch.Tensor, yd: torch.Tensor, dist: torch.Tensor, iters: int = 10):
    """Inverse of the OpenCV radial-tangential model by Newton iteration [NS-recall: camera_utils.radial_and_tangential_undistort,
    10 iterations, step clamped where the Jacobian is singular].  dist = (k1, k2, k3, k4, p1, p2)."""
    k1, k2, k3, k4, p1, p2 = [dist[..., i] for i in range(6)]
    x, y = xd.clone(), yd.clone()
    for _ in range(iters):
        r = x * x + y * y
        d = 1.0 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
        fx = d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x) - xd
        fy = d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y) - yd
        d_r = k1 + r * (2.0 * k2 + r * (3.0 * k3 + r * 4.0 * k4))
        d_x, d_y = 2.0 * x * d_r, 2.0 * y * d_r
        fx_x = d + d_x * x + 2.0 * p1 * y + 6.0 * p2 * x
        fx_y = d_y * x + 2.0 * p1 * x + 2.0 * p2 * y
        fy_x = d_x * y + 2.0 * p2 * y + 2.0 * p1 * x
        fy_y = d + d_y * y + 2.0 * p2 * x + 6.0 * p1 * y
        den = fy_x * fx_y - fx_x * fy_y
        ok = den.abs() > 1e-9
        sx = torch.where(ok, (fx * fy_y - fy * fx_y) / torch.where(ok, den, torch.ones_like(den)), torch.zeros_like(den))
        sy = torch.where(ok, (fy * fx_x - fx * fy_x) / torch.where(ok, den, torch.ones_like(den)), torch.zeros_like(den))
        x, y = x + sx, y + sy
    return x, y


def distort(x: torch.Tensor, y: torch.Tensor, dist: torch.Tensor):
    """Forward OpenCV model (tests: ``_undistort`` inverts it)."""
    k1, k2, k3, k4, p1, p2 = [dist[..., i] for i in range(6)]
    r = x * x + y * y
    d = 1.0 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
    return d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x), d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y)


class Cameras:
    """The subset of nerfstudio's ``Cameras`` the hot path's callers use: perspective cameras with optional OpenCV distortion.

    camera_to_worlds [N,3,4] (OpenGL convention: the camera looks along -z, +y up, as nerfstudio stores poses);
    fx, fy, cx, cy floats or [N]; distortion_params [N,6] = (k1,k2,k3,k4,p1,p2) or None."""

    def __init__(self, camera_to_worlds: torch.Tensor, fx, fy, cx, cy, width: int, height: int,
                 distortion_params: Optional[torch.Tensor] = None):
        c2w = torch.as_tensor(camera_to_worlds, dtype=torch.float32)
        if c2w.dim() == 2:
            c2w = c2w[None]
        self.camera_to_worlds = c2w[:, :3, :4].contiguous()
        n = self.camera_to_worlds.shape[0]

        def per_cam(v):
            return torch.as_tensor(v, dtype=torch.float32).reshape(-1).expand(n).clone()
        self.fx, self.fy, self.cx, self.cy = per_cam(fx), per_cam(fy), per_cam(cx), per_cam(cy)
        self.width, self.height = int(width), int(height)
        self.distortion_params = None if distortion_params is None else torch.as_tensor(distortion_params, dtype=torch.float32).reshape(-1, 6).expand(n, 6).clone()

    @property
    def size(self) -> int:
        return self.camera_to_worlds.shape[0]

    def __len__(self):
        return self.size

    @property
    def device(self):
        return self.camera_to_worlds.device

    def to(self, device):
        c = Cameras.__new__(Cameras)
        c.width, c.height = self.width, self.height
        for k in ("camera_to_worlds", "fx", "fy", "cx", "cy", "distortion_params"):
            v = getattr(self, k)
            setattr(c, k, None if v is None else v.to(device))
        return c

    def __getitem__(self, i):
        if isinstance(i, int):
            i = slice(i, i + 1)
        c = Cameras.__new__(Cameras)
        c.width, c.height = self.width, self.height
        for k in ("camera_to_worlds", "fx", "fy", "cx", "cy", "distortion_params"):
            v = getattr(self, k)
            setattr(c, k, None if v is None else v[i])
        return c

    def get_image_coords(self) -> torch.Tensor:
        """[H,W,2] pixel-centre coordinates (row + 0.5, col + 0.5) [NS-recall: Cameras.get_image_coords, pixel_offset 0.5]."""
        ys, xs = torch.meshgrid(torch.arange(self.height, device=self.device, dtype=torch.float32),
                                torch.arange(self.width, device=self.device, dtype=torch.float32), indexing="ij")
        return torch.stack([ys, xs], dim=-1) + 0.5

    def _generate_rays_hip(self, camera_indices, coords: Optional[torch.Tensor]) -> RayBundle:
        """``generate_rays`` on the device as ONE launch (csrc/camera.hip ``camera_rays_kernel``): a 684 x 1024 eval frame is 700,416
        rays -- as a tensor expression that is ~150 element-wise launches plus a 700k-batch 3x3 matmul (12 ms of a 34 ms frame)."""
        from . import _lib
        from .field import _dev_index, _stream_ptr
        lib = _lib.load()
        dev = self.device
        d = _dev_index(self.camera_to_worlds)
        c2w = self.camera_to_worlds.float().contiguous()                # contiguous fp32 views of a Cameras slice: no launches
        fx, fy, cx, cy = (t.float().contiguous() for t in (self.fx, self.fy, self.cx, self.cy))
        dist = self.distortion_params.float().contiguous() if self.distortion_params is not None else None
        cam_t, cam_single = None, 0
        if isinstance(camera_indices, int):
            cam_single = camera_indices
        else:
            host = camera_indices if not (isinstance(camera_indices, torch.Tensor) and camera_indices.is_cuda) else None
            if host is not None:            # indices that live on the host are validated here (the CPU path raises IndexError too);
                hv = torch.as_tensor(host).reshape(-1)      # device-resident ones by the kernel: out-of-range rays come back NaN
                if hv.numel() and (int(hv.min()) < 0 or int(hv.max()) >= self.size):
                    raise IndexError(f"camera index out of range for {self.size} cameras")
            cam_t = torch.as_tensor(camera_indices, device=dev).reshape(-1).long()
        if coords is None:
            R, width, co = self.height * self.width, self.width, None
        else:
            co = coords.reshape(-1, 2).to(dev).float().contiguous()
            R, width = co.shape[0], self.width
        if cam_t is not None:
            if cam_t.numel() not in (1, R):
                raise ValueError(f"camera_indices must hold 1 or {R} entries, got {cam_t.numel()}")
            cam_t = cam_t.expand(R).contiguous()
        o = torch.empty((R, 3), dtype=torch.float32, device=dev)
        dd = torch.empty((R, 3), dtype=torch.float32, device=dev)
        cam_out = torch.empty((R, 1), dtype=torch.int64, device=dev)
        _lib.check(lib.neraf_camera_rays(_lib.ctx(d), c2w.data_ptr(), fx.data_ptr(), fy.data_ptr(), cx.data_ptr(), cy.data_ptr(),
                                         dist.data_ptr() if dist is not None else None,
                                         self.size, cam_t.data_ptr() if cam_t is not None else None, int(cam_single),
                                         co.data_ptr() if co is not None else None, R, width, o.data_ptr(), dd.data_ptr(), cam_out.data_ptr(),
                                         _stream_ptr()), d)
        return RayBundle(origins=o, directions=dd, camera_indices=cam_out)

    def generate_rays(self, camera_indices, coords: Optional[torch.Tensor] = None,
                      camera_opt_to_camera: Optional[torch.Tensor] = None) -> RayBundle:
        """Rays of camera(s) ``camera_indices`` (int, or int tensor [R]) through ``coords`` [..., 2] = (row, col) in pixels
        (default: every pixel centre of one camera, row-major [H*W]).  Directions are unit vectors; ``camera_indices`` of the
        bundle is [R,1] as nerfstudio's.  ``camera_opt_to_camera`` [R,3,4] right-multiplies the poses (pose refinement)."""
        dev = self.device
        if dev.type == "cuda" and camera_opt_to_camera is None:
            return self._generate_rays_hip(camera_indices, coords)
        if coords is None:
            coords = self.get_image_coords().reshape(-1, 2)
        coords = coords.reshape(-1, 2).to(dev)
        R = coords.shape[0]
        if isinstance(camera_indices, int):
            cam = torch.full((R,), camera_indices, dtype=torch.long, device=dev)
        else:
            cam = torch.as_tensor(camera_indices, device=dev).reshape(-1).long().expand(R)
        y, x = coords[:, 0], coords[:, 1]
        xc = (x - self.cx[cam]) / self.fx[cam]
        yc = (y - self.cy[cam]) / self.fy[cam]
        if self.distortion_params is not None and bool((self.distortion_params[cam] != 0).any()):
            xc, yc = _undistort(xc, yc, self.distortion_params[cam])
        d_cam = torch.stack([xc, -yc, -torch.ones_like(xc)], dim=-1)          # OpenGL: +x right, +y up, looking along -z
        c2w = self.camera_to_worlds[cam]
        if camera_opt_to_camera is not None:
            c2w = multiply_poses(c2w, camera_opt_to_camera)
        d_world = torch.einsum("rij,rj->ri", c2w[:, :3, :3], d_cam)
        d_world = d_world / d_world.norm(dim=-1, keepdim=True)
        return RayBundle(origins=c2w[:, :3, 3].contiguous(), directions=d_world.contiguous(), camera_indices=cam[:, None])


def multiply_poses(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[..,3,4] x [..,3,4] as homogeneous 4x4 products [NS-recall: pose_utils.multiply]."""
    R = a[..., :3, :3] @ b[..., :3, :3]
    t = a[..., :3, 3:] + a[..., :3, :3] @ b[..., :3, 3:]
    return torch.cat([R, t], dim=-1)


def exp_map_SO3xR3(tangent: torch.Tensor) -> torch.Tensor:
    """[N,6] = (translation | rotation log) -> [N,3,4] with R = exp(so3) by Rodrigues and t copied [NS-recall: lie_groups]."""
    log_rot = tangent[:, 3:]
    nrms = (log_rot * log_rot).sum(dim=1)
    ang = torch.clamp(nrms, min=1e-4).sqrt()
    inv = 1.0 / ang
    f1, f2 = inv * ang.sin(), inv * inv * (1.0 - ang.cos())
    skew = torch.zeros((tangent.shape[0], 3, 3), dtype=tangent.dtype, device=tangent.device)
    skew[:, 0, 1], skew[:, 0, 2] = -log_rot[:, 2], log_rot[:, 1]
    skew[:, 1, 0], skew[:, 1, 2] = log_rot[:, 2], -log_rot[:, 0]
    skew[:, 2, 0], skew[:, 2, 1] = -log_rot[:, 1], log_rot[:, 0]
    R = f1[:, None, None] * skew + f2[:, None, None] * (skew @ skew) + torch.eye(3, dtype=tangent.dtype, device=tangent.device)[None]
    return torch.cat([R, tangent[:, :3, None]], dim=-1)


def _update_epoch() -> int:
    from . import optim
    return optim.UPDATE_EPOCH


class _CameraApplyFn(torch.autograd.Function):
    """``apply_to_raybundle`` on the device as ONE launch each way (csrc/camera.hip) instead of the ~35 small launches the PyTorch
    expression below costs per training step once autograd has replayed it.  Inputs: pose_adjustment [n,6], camera index [R] int32,
    origins / directions [R,3] fp32 (constants: the data manager's rays).  The same launch evaluates the pose regulariser and the
    two pose-norm metrics (``CameraOptimizer.get_loss_dict`` / ``get_metrics_dict``): outputs 3 and 4, the first differentiable; the
    backward writes d pose = regulariser gradient + ray pull-back in two launches (the ray gradients are read where they lie,
    e.g. as the two halves of the field backward's [R,6] buffer)."""

    @staticmethod
    def forward(ctx, pose: torch.Tensor, cam: torch.Tensor, o: torch.Tensor, d: torch.Tensor, w_t: float, w_r: float):
        from . import _lib
        from .field import _dev_index, _stream_ptr
        lib = _lib.load()
        dev = _dev_index(o)
        R = o.shape[0]
        o_out, d_out = torch.empty_like(o), torch.empty_like(d)
        reg = torch.empty(3, dtype=torch.float32, device=o.device)
        _lib.check(lib.neraf_camera_apply(_lib.ctx(dev), pose.data_ptr(), cam.data_ptr(), o.data_ptr(), d.data_ptr(), R, o_out.data_ptr(),
                                          d_out.data_ptr(), pose.shape[0], w_t, w_r, reg.data_ptr(), _stream_ptr()), dev)
        ctx.save_for_backward(pose, cam, d)
        ctx.dev, ctx.w = dev, (w_t, w_r)
        ctx.set_materialize_grads(False)
        norms = reg[1:]
        ctx.mark_non_differentiable(norms)
        return o_out, d_out, reg[0], norms

    @staticmethod
    def backward(ctx, g_o, g_d, g_reg, _g_norms):
        from . import _lib
        from .field import _stream_ptr
        pose, cam, d = ctx.saved_tensors
        lib = _lib.load()
        g_pose = torch.empty_like(pose)
        stride = 3
        if (g_o is None) != (g_d is None):
            z = torch.zeros_like(d)
            g_o, g_d = (g_o if g_o is not None else z), (g_d if g_d is not None else z)
        if g_o is not None:
            same = (g_o.dtype == torch.float32 and g_d.dtype == torch.float32 and g_o.stride(1) == 1 and g_d.stride(1) == 1
                    and g_o.stride(0) == g_d.stride(0) and g_o.stride(0) >= 3)
            if not same:
                g_o, g_d = g_o.contiguous().float(), g_d.contiguous().float()
            stride = g_o.stride(0)
        if g_reg is not None:
            g_reg = g_reg.float().reshape(())
        _lib.check(lib.neraf_camera_apply_bwd(_lib.ctx(ctx.dev), pose.data_ptr(), cam.data_ptr(), d.data_ptr(),
                                              g_o.data_ptr() if g_o is not None else None, g_d.data_ptr() if g_d is not None else None,
                                              stride, d.shape[0], pose.shape[0], ctx.w[0], ctx.w[1],
                                              g_reg.data_ptr() if g_reg is not None else None, g_pose.data_ptr(), _stream_ptr()), ctx.dev)
        return g_pose, None, None, None, None, None


class CameraOptimizer(nn.Module):
    """``CameraOptimizerConfig(mode="SO3xR3")`` (NeRAF_config.py:97) [NS-recall: cameras/camera_optimizers.py].

    ``pose_adjustment`` [num_cameras, 6] starts at zero; ``apply_to_raybundle`` moves origins by the translation part and
    rotates directions (training only); ``get_loss_dict`` adds nerfacto's regulariser
    ``mean |t| * trans_l2_penalty + mean |r| * rot_l2_penalty`` (1e-2, 1e-3).  The parameter lives in the ``camera_opt`` group
    (NeRAF_config.py:128-131: Adam lr 1e-3 -> 1e-4 over 5000 steps).

    ``pose_adjustment`` receives the regulariser's gradient and the photometric one: the vision loss node returns d loss / d (ray
    origin, ray direction) from the HIP backward kernels (neraf_field_backward_rays / neraf_proposal_backward_rays: hash-grid input
    gradient through the contraction, SH input gradient), and autograd carries it through ``apply_to_raybundle`` into the pose
    deltas."""

    def __init__(self, num_cameras: int, mode: str = "SO3xR3", trans_l2_penalty: float = 1e-2, rot_l2_penalty: float = 1e-3):
        super().__init__()
        if mode not in ("off", "SO3xR3"):
            raise NotImplementedError("camera optimizer modes: 'off', 'SO3xR3' (what NeRAF configures)")
        self.mode, self.num_cameras = mode, num_cameras
        self.trans_l2_penalty, self.rot_l2_penalty = trans_l2_penalty, rot_l2_penalty
        if mode != "off":
            self.pose_adjustment = nn.Parameter(torch.zeros((num_cameras, 6)))

    def forward(self, indices: torch.Tensor) -> torch.Tensor:
        """[R] camera indices -> [R,3,4] correction matrices (identity when off)."""
        idx = indices.reshape(-1).long()
        if self.mode == "off":
            eye = torch.eye(4, device=idx.device)[None, :3, :4]
            return eye.expand(idx.shape[0], 3, 4)
        return exp_map_SO3xR3(self.pose_adjustment[idx])

    def apply_to_raybundle(self, ray_bundle: RayBundle) -> RayBundle:
        if self.mode == "off" or ray_bundle.camera_indices is None:
            return ray_bundle
        if ray_bundle.origins.is_cuda and not ray_bundle.origins.requires_grad and not ray_bundle.directions.requires_grad:
            cam = ray_bundle.camera_indices_i32()
            o, d, reg, norms = _CameraApplyFn.apply(self.pose_adjustment, cam, ray_bundle.origins.float().contiguous(),
                                                    ray_bundle.directions.float().contiguous(),
                                                    self.trans_l2_penalty / self.num_cameras, self.rot_l2_penalty / self.num_cameras)
            # the regulariser and the pose norms of THIS parameter state came with the launch: get_loss_dict / get_metrics_dict of
            # the same iteration pick them up (consumed once each; without them they fall back to the torch expressions)
            self._fused = {"reg": reg, "norms": norms, "key": (self.pose_adjustment.data_ptr(), self.pose_adjustment._version, _update_epoch())}
            out = RayBundle(o, d, ray_bundle.camera_indices, ray_bundle.nears, ray_bundle.fars)
            out._cam_i32 = getattr(ray_bundle, "_cam_i32", None)
            return out
        corr = self(ray_bundle.camera_indices)
        origins = ray_bundle.origins + corr[:, :3, 3]
        directions = torch.bmm(corr[:, :3, :3], ray_bundle.directions[..., None]).squeeze(-1)
        return RayBundle(origins, directions, ray_bundle.camera_indices, ray_bundle.nears, ray_bundle.fars)

    def _take_fused(self, what: str):
        f = getattr(self, "_fused", None)
        if f is None or what not in f or f["key"] != (self.pose_adjustment.data_ptr(), self.pose_adjustment._version, _update_epoch()):
            return None
        return f.pop(what)

    def get_loss_dict(self, loss_dict: Dict[str, torch.Tensor]) -> None:
        if self.mode != "off":
            reg = self._take_fused("reg")
            if reg is not None:
                loss_dict["camera_opt_regularizer"] = reg
                return
            # mean |t| * trans_l2_penalty + mean |w| * rot_l2_penalty, as three launches: norms of the [n, 2, 3] view, weights, sum
            w = getattr(self, "_reg_w", None)
            if w is None or w.device != self.pose_adjustment.device:
                w = torch.tensor([self.trans_l2_penalty / self.num_cameras, self.rot_l2_penalty / self.num_cameras],
                                 device=self.pose_adjustment.device)
                self._reg_w = w
            n = torch.linalg.vector_norm(self.pose_adjustment.view(-1, 2, 3), dim=-1)
            loss_dict["camera_opt_regularizer"] = (n * w).sum()

    def get_metrics_dict(self, metrics_dict: Dict[str, torch.Tensor]) -> None:
        if self.mode != "off":
            norms = self._take_fused("norms")
            if norms is not None:
                metrics_dict["camera_opt_translation"], metrics_dict["camera_opt_rotation"] = norms[0], norms[1]
                return
            metrics_dict["camera_opt_translation"] = self.pose_adjustment[:, :3].norm()
            metrics_dict["camera_opt_rotation"] = self.pose_adjustment[:, 3:].norm()

    def get_param_groups(self, param_groups: dict) -> None:
        if self.mode != "off":
            param_groups["camera_opt"] = [self.pose_adjustment]

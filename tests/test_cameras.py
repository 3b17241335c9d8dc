"""CPU property tests of neraf_amd/cameras.py (camera -> rays, SO3xR3 pose refinement) and the image metrics of
neraf_amd/vision.py.  nerfstudio is not importable, so these restatements [NS-recall] are guarded by invariants."""
import math

import numpy as np
import torch

from neraf_amd.cameras import CameraOptimizer, Cameras, _undistort, distort, exp_map_SO3xR3, multiply_poses
from neraf_amd.datamanagers import synthetic_cameras
from neraf_amd.vision import RayBundle, psnr, ssim


def test_center_pixel_looks_along_minus_z_and_directions_are_unit():
    c2w = torch.eye(4)[:3, :4].clone()
    c2w[:, 3] = torch.tensor([0.3, -0.2, 0.1])
    cam = Cameras(c2w, 100.0, 100.0, 32.0, 24.0, 64, 48)
    rb = cam.generate_rays(0)
    assert rb.origins.shape == (48 * 64, 3) and rb.camera_indices.shape == (48 * 64, 1)
    np.testing.assert_allclose(rb.directions.norm(dim=-1).numpy(), 1.0, atol=1e-6)
    np.testing.assert_allclose(rb.origins.numpy(), np.tile([0.3, -0.2, 0.1], (48 * 64, 1)), atol=0)
    # the ray through the principal point (cx, cy) = pixel centre (31.5+0.5, 23.5+0.5)
    rbc = cam.generate_rays(0, torch.tensor([[24.0, 32.0]]))
    np.testing.assert_allclose(rbc.directions.numpy(), [[0.0, 0.0, -1.0]], atol=1e-7)
    # +col -> +x, +row -> -y (image rows grow downwards)
    d = cam.generate_rays(0, torch.tensor([[24.0, 42.0], [34.0, 32.0]])).directions
    assert d[0, 0] > 0 and abs(float(d[0, 1])) < 1e-7 and d[1, 1] < 0 and abs(float(d[1, 0])) < 1e-7
    # row-major pixel order: ray k = row k // W, col k % W
    k = 5 * 64 + 7
    expect = cam.generate_rays(0, torch.tensor([[5.5, 7.5]])).directions[0]
    np.testing.assert_allclose(rb.directions[k].numpy(), expect.numpy(), atol=1e-7)


def test_undistort_inverts_the_opencv_model_at_raf_intrinsics():
    cams = synthetic_cameras(1)
    dist = cams.distortion_params[0]
    x = torch.linspace(-0.95, 0.95, 41)[:, None].expand(41, 41).reshape(-1)       # the RAF frame spans |x| < 0.98, |y| < 1.46
    y = torch.linspace(-1.4, 1.4, 41)[None, :].expand(41, 41).reshape(-1)
    xd, yd = distort(x, y, dist[None].expand(x.shape[0], 6))
    xu, yu = _undistort(xd, yd, dist[None].expand(x.shape[0], 6))
    np.testing.assert_allclose(xu.numpy(), x.numpy(), atol=2e-6)
    np.testing.assert_allclose(yu.numpy(), y.numpy(), atol=2e-6)
    assert float((xd - x).abs().max()) > 1e-3          # the distortion is not a no-op at these coefficients


def test_full_frame_ray_count_matches_the_raf_frame():
    cams = synthetic_cameras(2)
    rb = cams[1].generate_rays(0)
    assert len(rb) == 684 * 1024 == 700416
    np.testing.assert_allclose(rb.directions.norm(dim=-1).numpy(), 1.0, atol=1e-5)
    fwd = -cams.camera_to_worlds[1, :, 2]
    assert float((rb.directions @ fwd).min()) > 0.3     # every pixel looks into the forward half-space


def test_exp_map_is_a_rotation_and_matches_rodrigues():
    torch.manual_seed(0)
    t = torch.randn(5, 6, dtype=torch.float64) * 0.3
    m = exp_map_SO3xR3(t)
    R = m[:, :, :3]
    np.testing.assert_allclose((R @ R.transpose(1, 2)).numpy(), np.tile(np.eye(3), (5, 1, 1)), atol=1e-12)
    np.testing.assert_allclose(torch.linalg.det(R).numpy(), 1.0, atol=1e-12)
    np.testing.assert_allclose(m[:, :, 3].numpy(), t[:, :3].numpy())
    # rotation by angle |w| about w/|w|: trace = 1 + 2 cos|w|
    ang = t[:, 3:].norm(dim=-1)
    np.testing.assert_allclose(torch.diagonal(R, dim1=1, dim2=2).sum(-1).numpy(), (1 + 2 * torch.cos(ang)).numpy(), atol=1e-10)
    ident = exp_map_SO3xR3(torch.zeros(2, 6))
    np.testing.assert_allclose(ident.numpy(), np.tile(np.eye(4)[:3], (2, 1, 1)), atol=1e-6)


def test_camera_optimizer_identity_at_init_regulariser_and_gradient():
    co = CameraOptimizer(4, mode="SO3xR3")
    o, d = torch.rand(6, 3), torch.nn.functional.normalize(torch.randn(6, 3), dim=-1)
    rb = RayBundle(o, d, torch.tensor([0, 1, 2, 3, 0, 1])[:, None])
    out = co.apply_to_raybundle(rb)
    np.testing.assert_allclose(out.origins.detach().numpy(), o.numpy(), atol=1e-6)
    np.testing.assert_allclose(out.directions.detach().numpy(), d.numpy(), atol=1e-6)
    with torch.no_grad():
        co.pose_adjustment[1] = torch.tensor([0.1, 0.0, -0.2, 0.0, 0.0, math.pi / 2])
    out = co.apply_to_raybundle(rb)
    np.testing.assert_allclose(out.origins[1].detach().numpy(), (o[1] + torch.tensor([0.1, 0.0, -0.2])).numpy(), atol=1e-6)
    exp_d = torch.stack([-d[1, 1], d[1, 0], d[1, 2]])       # +90 degrees about z
    np.testing.assert_allclose(out.directions[1].detach().numpy(), exp_d.numpy(), atol=1e-5)
    ld = {}
    co.get_loss_dict(ld)
    expect = (math.sqrt(0.05) / 4) * 1e-2 + (math.pi / 2 / 4) * 1e-3
    np.testing.assert_allclose(float(ld["camera_opt_regularizer"]), expect, rtol=1e-5)
    ld["camera_opt_regularizer"].backward()
    assert co.pose_adjustment.grad is not None and bool(torch.isfinite(co.pose_adjustment.grad).all())
    g = {}
    co.get_param_groups(g)
    assert list(g) == ["camera_opt"] and g["camera_opt"][0] is co.pose_adjustment
    off = CameraOptimizer(4, mode="off")
    assert off.apply_to_raybundle(rb) is rb and not list(off.parameters())


def test_multiply_poses_is_the_homogeneous_product():
    torch.manual_seed(1)
    a, b = exp_map_SO3xR3(torch.randn(3, 6) * 0.5), exp_map_SO3xR3(torch.randn(3, 6) * 0.5)
    h = lambda m: torch.cat([m, torch.tensor([0.0, 0, 0, 1]).expand(3, 1, 4)], dim=1)
    np.testing.assert_allclose(multiply_poses(a, b).numpy(), (h(a) @ h(b))[:, :3].numpy(), atol=1e-6)


def test_psnr_and_ssim_known_answers():
    a = torch.rand(40, 50, 3)
    np.testing.assert_allclose(float(psnr(a, a + 0.1)), 20.0, rtol=1e-5)        # MSE 0.01 -> 20 dB
    assert float(ssim(a, a)) > 0.9999
    assert float(ssim(a, torch.rand(40, 50, 3))) < 0.2
    b = (a * 0.8 + 0.1)
    assert float(ssim(a, b)) < float(ssim(a, a)) and float(ssim(a, b)) > 0.5

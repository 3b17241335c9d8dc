"""csrc/glue.hip and the GradScaler update kernel, called through the C ABI, against the torch expressions they replace
(Trainer.train_iteration's loss sum + grad_scaler.scale, NerfactoModel's loss normalisation + psnr, the fan-out of the upstream loss
gradients in the vision backward, torch._amp_update_scale_)."""
import ctypes as C

import numpy as np
import pytest
import torch

from neraf_amd import _lib, synth

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_loss_sum_scale_and_finalize():
    lib, h = _lib.load(), _lib.ctx(0)
    dev = torch.device("cuda:0")
    terms = [T(synth.uniform(f"glue.term{i}", (1,), 0.0, 2.0)).to(dev).reshape(()) for i in range(7)]
    scale = torch.tensor(65536.0, device=dev)
    out = torch.empty(2, device=dev)
    _lib.check(lib.neraf_loss_sum_scale(h, _lib.ptr_array(terms), len(terms), scale.data_ptr(), out.data_ptr(), _st()), 0)
    want = terms[0].clone()
    for t in terms[1:]:
        want = want + t                                       # functools.reduce(add, ...): left to right, fp32
    assert float(out[1]) == float(want) and float(out[0]) == float(want * scale)
    _lib.check(lib.neraf_loss_sum_scale(h, _lib.ptr_array(terms[:1]), 1, None, out.data_ptr(), _st()), 0)
    assert float(out[0]) == float(terms[0]) == float(out[1])
    assert lib.neraf_loss_sum_scale(h, _lib.ptr_array(terms), 13, None, out.data_ptr(), _st()) != 0           # more than 12 terms
    sums = torch.tensor([12.5, 3.0, 0.25, 99.0], device=dev)
    k3 = torch.tensor([1.0 / (3 * 4096), 0.002 / 4096, 1.0 / (4096 * 48)], device=dev)
    out4 = torch.empty(4, device=dev)
    _lib.check(lib.neraf_vision_loss_finalize(h, sums.data_ptr(), k3.data_ptr(), out4.data_ptr(), _st()), 0)
    np.testing.assert_allclose(out4[:3].cpu().numpy(), (sums[:3] * k3).cpu().numpy(), rtol=1e-7)
    np.testing.assert_allclose(float(out4[3]), float(-10.0 * torch.log10(sums[0] * k3[0])), rtol=1e-6)


@pytest.mark.parametrize("with_rays,with_inter", [(True, True), (False, False)])
def test_vision_bwd_prologue(with_rays, with_inter):
    lib, h = _lib.load(), _lib.ctx(0)
    dev = torch.device("cuda:0")
    R, S = 300, 48
    n = R * S
    u_rgb = T(synth.normal("glue.u_rgb", (n, 3))).to(dev)
    u_dens, u_dist = T(synth.normal("glue.u_dens", (n,))).to(dev), T(synth.normal("glue.u_dist", (n,))).to(dev)
    g_rgb, g_dist = torch.tensor(512.0, device=dev), torch.tensor(-3.25, device=dev)
    g_inter = torch.tensor(7.0, device=dev) if with_inter else None
    d_rgb, d_dens = torch.full((n, 3), float("nan"), device=dev), torch.full((n,), float("nan"), device=dev)
    up, sums = torch.full((3,), float("nan"), device=dev), torch.full((4,), float("nan"), device=dev)
    d_rays = torch.full((R, 6), float("nan"), device=dev) if with_rays else None
    _lib.check(lib.neraf_vision_bwd_prologue(h, u_rgb.data_ptr(), u_dens.data_ptr(), u_dist.data_ptr(), g_rgb.data_ptr(),
                                             g_inter.data_ptr() if g_inter is not None else None, g_dist.data_ptr(), n, d_rgb.data_ptr(),
                                             d_dens.data_ptr(), up.data_ptr(), d_rays.data_ptr() if with_rays else None,
                                             R * 6 if with_rays else 0, sums.data_ptr(), _st()), 0)
    torch.testing.assert_close(d_rgb, u_rgb * g_rgb, rtol=0, atol=0)
    torch.testing.assert_close(d_dens, torch.addcmul(u_dens * g_rgb, u_dist, g_dist), rtol=1e-6, atol=1e-6)
    assert up.tolist() == [512.0, 7.0 if with_inter else 0.0, -3.25] and sums.tolist() == [0.0] * 4
    if with_rays:
        assert float(d_rays.abs().max()) == 0.0


def test_amp_update_scale_matches_torch():
    lib, h = _lib.load(), _lib.ctx(0)
    dev = torch.device("cuda:0")
    sa, sb = torch.tensor(1024.0, device=dev), torch.tensor(1024.0, device=dev)
    ta, tb = torch.zeros((), dtype=torch.int32, device=dev), torch.zeros((), dtype=torch.int32, device=dev)
    pattern = [(0, 0), (0, 0), (0, 0), (1, 0), (0, 0), (0, 1), (0, 0), (0, 0), (0, 0), (1, 1), (0, 0)]
    for f1, f2 in pattern:
        fa = [torch.tensor([float(f1)], device=dev), torch.tensor([float(f2)], device=dev)]
        clear = int(f1 == 0)              # clear_flags: both flags are reset by the launch after it has read them
        _lib.check(lib.neraf_amp_update_scale(h, sa.data_ptr(), ta.data_ptr(), _lib.ptr_array(fa), 2, 2.0, 0.5, 3, clear, _st()), 0)
        torch._amp_update_scale_(sb, tb, torch.tensor([float(f1 + f2)], device=dev), 2.0, 0.5, 3)
        assert float(sa) == float(sb) and int(ta) == int(tb), (f1, f2, float(sa), float(sb))
        assert [float(f) for f in fa] == ([0.0, 0.0] if clear else [float(f1), float(f2)])
    # growth that would overflow keeps the scale (torch: only a finite product is adopted)
    big = torch.tensor(3.0e38, device=dev)
    trk = torch.tensor(2, dtype=torch.int32, device=dev)
    z = [torch.zeros(1, device=dev)]
    _lib.check(lib.neraf_amp_update_scale(h, big.data_ptr(), trk.data_ptr(), _lib.ptr_array(z), 1, 2.0, 0.5, 3, 0, _st()), 0)
    assert float(big) == pytest.approx(3.0e38) and int(trk) == 0


def test_host_f32_cache_survives_address_reuse():
    """_lib.host_f32 caches the host copy of an AABB per (address, version, numel).  The caching allocator hands a freed small
    tensor's address to the next one of the same size (version 0 again): the cache entry pins the storage, so a later model's box
    can never alias a dead one's key (round 6: in suite order the RAF model read the previous test's AABB)."""
    dev = torch.device("cuda:0")
    seen = set()
    for i in range(8):
        t = torch.full((2, 3), float(i + 1), device=dev)
        vals = list(_lib.host_f32(t))
        assert vals == [float(i + 1)] * 6, (i, vals)
        assert t.data_ptr() not in seen           # pinned: every live-keyed address is still owned by its entry
        seen.add(t.data_ptr())
        del t
    u = torch.zeros(6, device=dev)
    assert list(_lib.host_f32(u)) == [0.0] * 6
    u.add_(2.0)                                    # an in-place edit bumps the version: re-read
    assert list(_lib.host_f32(u)) == [2.0] * 6

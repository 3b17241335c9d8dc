"""Property tests for the UNPINNED radiance oracle (oracle/vision.py): nerfstudio / tiny-cuda-nn are not
available, so the restatement is guarded by invariants instead of golden vectors."""
import numpy as np
import torch

from neraf_amd import synth
from oracle import vision as V


def _params(scale=0.5, n=8):
    spec = V.NerfactoSpec()
    tot = (spec.prop_grids[0].total, spec.prop_grids[1].total, spec.main_grid.total)
    P = {k: torch.from_numpy(v) for k, v in synth.vision_params(tot, num_train_data=n, table_scale=scale).items()}
    return spec, P


def test_grid_spec_matches_tcnn_conventions():
    g = V.GridSpec(16, 16, 2048, 19)
    assert g.resolutions[0] == 16 and g.resolutions[-1] == 2048
    assert all(s % 8 == 0 for s in g.sizes) and max(g.sizes) == 1 << 19
    assert g.sizes[0] == 4096 and g.total == sum(g.sizes)
    p = V.GridSpec(5, 16, 128, 17)
    assert p.resolutions == [16, 27, 46, 77, 128]


def test_hash_encode_partition_of_unity_and_range():
    g = V.GridSpec(16, 16, 2048, 19)
    table = torch.ones(g.total, 2)
    x = torch.rand(257, 3)
    enc = V.hash_encode(x, table, g)
    assert enc.shape == (257, 32)
    np.testing.assert_allclose(enc.numpy(), 1.0, atol=1e-5)          # trilinear weights sum to one on every level
    # constant-per-level table -> level value is reproduced exactly (index stays inside its level)
    t2 = torch.zeros(g.total, 2)
    for l in range(16):
        t2[g.offsets[l]:g.offsets[l + 1]] = float(l + 1)
    enc = V.hash_encode(x, t2, g)
    np.testing.assert_allclose(enc[:, ::2].numpy(), np.arange(1, 17)[None, :].repeat(257, 0), atol=1e-4)


def test_contraction_and_spacing_inverse():
    x = torch.randn(1000, 3) * 5
    c = V.contract_linf(x)
    assert float(c.abs().amax()) < 2.0
    inside = x.abs().amax(-1) < 1
    np.testing.assert_array_equal(c[inside].numpy(), x[inside].numpy())
    t = torch.logspace(-2, 3, 100)
    np.testing.assert_allclose(V.spacing_fn_inv(V.spacing_fn(t)).numpy(), t.numpy(), rtol=2e-4)


def test_sampler_weights_and_render_invariants():
    spec, P = _params()
    rb = synth.ray_batch(64, num_train_data=8)
    out = V.nerfacto_forward(torch.from_numpy(rb["origins"]), torch.from_numpy(rb["directions"]),
                             torch.from_numpy(rb["camera_indices"]), P, spec, step=500, training=True,
                             jitters=[torch.from_numpy(j) for j in rb["jitters"]])
    for w, ray in zip(out["weights_list"], out["ray_samples_list"]):
        assert float(w.min()) >= 0 and float(w.sum(-1).max()) <= 1 + 1e-5         # sum of weights <= 1
        assert bool((ray.s_bins[:, 1:] >= ray.s_bins[:, :-1]).all())               # monotone bins
        assert bool((ray.e_bins[:, 1:] >= ray.e_bins[:, :-1]).all())
        assert float(ray.e_bins.min()) >= spec.near - 1e-6 and float(ray.e_bins.max()) <= spec.far * (1 + 1e-5)
    assert out["rgb"].shape == (64, 3) and float(out["rgb"].min()) >= 0 and float(out["rgb"].max()) <= 1
    assert out["weights_list"][0].shape == (64, 256) and out["weights_list"][1].shape == (64, 96)
    assert out["weights_list"][2].shape == (64, 48)
    ld = V.vision_loss_dict(out, torch.from_numpy(rb["rgb"]), spec)
    assert all(torch.isfinite(v) and v >= 0 for v in ld.values())


def test_distortion_loss_closed_form_single_interval():
    # one sample with weight w on [a,b]: inter term 0, intra term w^2 (b-a)/3
    t = torch.tensor([[0.2, 0.5]])
    w = torch.tensor([[0.7]])
    np.testing.assert_allclose(V.lossfun_distortion(t, w).item(), 0.49 * 0.3 / 3, rtol=1e-6)


def test_interlevel_loss_zero_when_proposal_bounds_fine_weights():
    # identical histograms -> the proposal upper-bounds the fine weights -> zero loss
    t = torch.linspace(0, 1, 9)[None]
    w = torch.full((1, 8), 0.1)
    assert float(V.lossfun_outer(t, w, t, w).sum()) == 0.0
    assert float(V.lossfun_outer(t, w, t, w * 0.5).sum()) > 0.0


def test_trunc_exp_backward_is_clamped_to_pm15():
    """nerfstudio's trunc_exp [NS-recall]: forward exp(x), backward g * exp(clamp(x, -15, 15)) -- the three places it lives (this
    oracle, field_bwd.hip's field and proposal backward) agree; test_gpu_vision_train drives logits past +-15 on the GPU."""
    x = torch.tensor([-40.0, -15.0, -3.0, 0.0, 3.0, 15.0, 20.0], dtype=torch.float64, requires_grad=True)
    y = V.trunc_exp(x)
    np.testing.assert_allclose(y.detach().numpy(), np.exp(x.detach().numpy()))
    y.sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), np.exp(np.clip(x.detach().numpy(), -15, 15)))


def test_gradcheck_fp64_of_the_differentiable_pieces():
    """fp64 finite-difference check (torch.autograd.gradcheck) of every differentiable piece of the unpinned oracle that the
    HIP backward kernels are compared with: trilinear hash encode w.r.t. the table, the bias-free MLP, get_weights, the
    composite, the distortion and interlevel ("outer") losses w.r.t. weights."""
    torch.manual_seed(0)
    g = V.GridSpec(3, 4, 16, 8)                       # tiny grid: 2 dense levels + 1 hashed one
    assert g.sizes[-1] < g.resolutions[-1] ** 3
    x = torch.rand(5, 3, dtype=torch.float64)
    table = torch.randn(g.total, 2, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda t: V.hash_encode(x, t, g), (table,), eps=1e-6, atol=1e-6)
    ws = [torch.randn(8, 6, dtype=torch.float64, requires_grad=True), torch.randn(3, 8, dtype=torch.float64, requires_grad=True)]
    xin = torch.randn(4, 6, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda a, b, c: V.tcnn_mlp(a, [b, c]), (xin, ws[0], ws[1]), eps=1e-6, atol=1e-6)
    dens = (torch.rand(3, 7, dtype=torch.float64) * 3).requires_grad_(True)
    deltas = torch.rand(3, 7, dtype=torch.float64) + 0.05
    assert torch.autograd.gradcheck(lambda d: V.get_weights(d, deltas), (dens,), eps=1e-6, atol=1e-6)
    # losses on a 2-level toy sample hierarchy
    R, S0, S1 = 2, 6, 4
    e0 = torch.sort(torch.rand(R, S0 + 1, dtype=torch.float64), -1).values
    e1 = torch.sort(torch.rand(R, S1 + 1, dtype=torch.float64), -1).values
    z = torch.zeros(R, 3, dtype=torch.float64)
    r0, r1 = V.RaySamples(z, z, e0, e0), V.RaySamples(z, z, e1, e1)
    w0 = torch.rand(R, S0, dtype=torch.float64).requires_grad_(True)
    w1 = torch.rand(R, S1, dtype=torch.float64).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda a: V.distortion_loss([a], [r1]), (w1,), eps=1e-6, atol=1e-6)
    assert torch.autograd.gradcheck(lambda a: V.interlevel_loss([a, w1.detach()], [r0, r1]), (w0,),
                                    eps=1e-6, atol=1e-6)
    rgb = torch.rand(R, S1, 3, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda c, w: V.render(r1, c, w, True)[0], (rgb, w1), eps=1e-6, atol=1e-6)


def test_hash_index_stays_inside_its_level_at_the_2pow19_boundary():
    """Hashed levels of the main grid (T = 2^19): every corner index is < 2^19 for coordinates at the extreme corners of the
    unit cube and for the finest level's largest integer coordinates; dense levels index below res^3 (rounded up to 8)."""
    g = V.GridSpec(16, 16, 2048, 19)
    marks = torch.zeros(g.total, 2)
    x = torch.cat([torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [1.0, 0.0, 1.0], [0.999999, 0.999999, 0.999999]]), torch.rand(4096, 3)])
    # a table whose value is the level number: if any index left its level the encoded value would mix levels
    for l in range(16):
        marks[g.offsets[l]:g.offsets[l + 1]] = float(l)
    enc = V.hash_encode(x, marks, g)
    np.testing.assert_allclose(enc[:, ::2].numpy(), np.arange(16)[None, :].repeat(x.shape[0], 0), atol=2e-4)
    assert max(g.sizes) == 1 << 19 and g.sizes[-1] == 1 << 19 and g.resolutions[-1] ** 3 > 1 << 19

"""Pin the CPU oracle (oracle/audio.py) against golden vectors produced by the imported reference
(tests/tools/gen_golden.py).  CPU-only; runs in seconds-to-a-minute."""
import numpy as np
import pytest
import torch

from neraf_amd import synth
from oracle import audio as O


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def stats(x):
    x = x.detach().double()
    return np.array([x.mean().item(), x.abs().mean().item(), x.pow(2).mean().sqrt().item()])


@pytest.mark.parametrize("C,Fq,tag", [(1, 513, "g2_nacf_raf"), (2, 257, "g2_nacf_ss")])
def test_g2_nacf_forward_backward(golden, C, Fq, tag):
    g = golden(tag)
    sd = {k: T(v).requires_grad_(True) for k, v in synth.nacf_state_dict(1187, 512, C, Fq).items()}
    h = T(synth.uniform("g2.h", (8, 1187), -1.0, 1.0)).requires_grad_(True)
    wout = T(synth.uniform("g2.wout", (8, C, Fq), -1.0, 1.0))
    y = O.nacf_forward(h, sd)
    assert y.shape == (8, C, Fq)
    np.testing.assert_allclose(y.detach().numpy(), g["out"], rtol=1e-5, atol=1e-5)
    (y * wout).sum().backward()
    np.testing.assert_allclose(h.grad.numpy(), g["dh"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd["soundfield.0.weight"].grad[:4, :8].numpy(), g["dw0_slab"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(stats(sd["soundfield.0.weight"].grad), g["dw0_stats"], rtol=1e-4)
    np.testing.assert_allclose(sd["soundfield.0.bias"].grad[:16].numpy(), g["db0"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(stats(sd["soundfield.4.weight"].grad), g["dw4_stats"], rtol=1e-4)
    np.testing.assert_allclose(sd["STFT_linear.0.weight"].grad[:4, :8].numpy(), g["dwh0_slab"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd[f"STFT_linear.{C-1}.bias"].grad.numpy(), g["dbh_last"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("C,Fq", [(1, 513), (2, 257)])
@pytest.mark.parametrize("lt", ["mse", "l1"])
def test_g3_stft_loss(golden, C, Fq, lt):
    g = golden("g3_stft_loss")
    x = T(synth.uniform(f"g3.x{C}", (8, C, Fq), -6.0, 2.0)).requires_grad_(True)
    y = T(synth.uniform(f"g3.y{C}", (8, C, Fq), -6.0, 2.0))
    sc, mag = O.stft_loss(x, y, lt)
    np.testing.assert_allclose(sc.item(), g[f"sc_{lt}_{C}"], rtol=1e-6)
    np.testing.assert_allclose(mag.item(), g[f"mag_{lt}_{C}"], rtol=1e-6)
    d = O.audio_loss_dict(x, y, "SC+SLMSE" if lt == "mse" else "SC+SLL1")
    (d["audio_sc_loss"] + d["audio_mag_loss"]).backward()
    np.testing.assert_allclose(x.grad.numpy(), g[f"dx_{lt}_{C}"], rtol=1e-5, atol=1e-12)


def _run_resnet(S, need_grad=True):
    sd = {k: T(v) for k, v in synth.resnet3d_state_dict(7).items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(need_grad)
    x = T(synth.uniform(f"g1.grid{S}", (1, 7, S, S, S), 0.0, 1.0)).requires_grad_(need_grad)
    return sd, x


def test_g1_resnet3d_64(golden):
    g = golden("g1_resnet3d_64")
    S = 64
    sd, x = _run_resnet(S)
    wsum = T(synth.uniform("g1.outw", (1024,), -1.0, 1.0))
    y, st = O.resnet3d_forward(x, sd, train=True, return_stages=True)
    assert y.shape == (1, 1024, 1, 1, 1)
    np.testing.assert_allclose(y.detach().flatten().numpy(), g["out_train"], rtol=2e-4, atol=2e-5)
    for k in ("conv1", "maxpool", "layer1", "layer2", "layer3"):
        np.testing.assert_allclose(stats(st[k]), g["stage_" + k], rtol=1e-4)
    np.testing.assert_allclose(st["conv1"][0, :8, 3, 5, :16].detach().numpy(), g["conv1_slab"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(st["layer3"][0, :16, 1, 2, :].detach().numpy(), g["layer3_slab"], rtol=2e-4, atol=2e-5)
    (y.flatten() * wsum).sum().backward()
    p = g["probe_idx"]
    np.testing.assert_allclose(x.grad[0, p[:, 0], p[:, 1], p[:, 2], p[:, 3]].numpy(), g["dx_probe"], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(stats(x.grad), g["dx_stats"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(sd["conv1.weight"].grad.numpy(), g["dw_conv1"], rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(stats(sd["layer1.0.conv2.weight"].grad), g["dw_l1_0_conv2_stats"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(stats(sd["layer3.5.conv3.weight"].grad), g["dw_l3_5_conv3_stats"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(sd["bn1.weight"].grad.numpy(), g["dgamma_bn1"], rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(sd["bn1.bias"].grad.numpy(), g["dbeta_bn1"], rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(sd["layer2.0.downsample.1.weight"].grad.numpy(), g["dgamma_l2_0_ds"], rtol=2e-3, atol=1e-3)
    with torch.no_grad():
        ye = O.resnet3d_forward(x, sd, train=False)
    np.testing.assert_allclose(ye.flatten().numpy(), g["out_eval"], rtol=2e-4, atol=2e-5)


def test_g1_resnet3d_128_forward(golden):
    """Full BASELINE size (7x128^3), forward only to stay within the CPU suite budget."""
    g = golden("g1_resnet3d_128")
    sd, x = _run_resnet(128, need_grad=False)
    with torch.no_grad():
        y, st = O.resnet3d_forward(x, sd, train=True, return_stages=True)
    np.testing.assert_allclose(y.flatten().numpy(), g["out_train"], rtol=2e-4, atol=2e-5)
    for k in ("conv1", "maxpool", "layer1", "layer2", "layer3"):
        np.testing.assert_allclose(stats(st[k]), g["stage_" + k], rtol=1e-4)


def _run_resnet_variant(S, N, need_grad):
    layers = (3, 4, 6, 3) if N == 2048 else (3, 4, 6)
    sd = {k: T(v) for k, v in synth.resnet3d_state_dict(7, layers=layers).items()}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(need_grad)
    x = T(synth.uniform(f"g1.grid{S}", (1, 7, S, S, S), 0.0, 1.0)).requires_grad_(need_grad)
    return sd, x, layers


def test_g1_resnet3d_64_2048(golden):
    """N_features = 2048 (layer4, NeRAF_resnet3d.py:131 / :193-195) on the 64^3 grid: forward, stages, backward and eval mode of the
    restatement against the reference module's own outputs (tests/tools/gen_golden.py g1v)."""
    g = golden("g1_resnet3d_64_2048")
    sd, x, layers = _run_resnet_variant(64, 2048, True)
    wsum = T(synth.uniform("g1.outw2048", (2048,), -1.0, 1.0))
    y, st = O.resnet3d_forward(x, sd, train=True, layers=layers, return_stages=True)
    assert y.shape == (1, 2048, 1, 1, 1)
    np.testing.assert_allclose(y.detach().flatten().numpy(), g["out_train"], rtol=2e-4, atol=2e-5)
    for k in ("conv1", "maxpool", "layer1", "layer2", "layer3", "layer4"):
        np.testing.assert_allclose(stats(st[k]), g["stage_" + k], rtol=1e-4)
    np.testing.assert_allclose(st["layer4"][0, :16, 1, 1, :].detach().numpy(), g["last_slab"], rtol=2e-4, atol=2e-5)
    (y.flatten() * wsum).sum().backward()
    p = g["probe_idx"]
    np.testing.assert_allclose(x.grad[0, p[:, 0], p[:, 1], p[:, 2], p[:, 3]].numpy(), g["dx_probe"], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(stats(x.grad), g["dx_stats"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(sd["conv1.weight"].grad.numpy(), g["dw_conv1"], rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(stats(sd["layer4.2.conv3.weight"].grad), g["dw_last_conv3_stats"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(sd["layer4.2.conv3.weight"].grad[:8, :8, 0, 0, 0].numpy(), g["dw_last_conv3_slab"], rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(stats(sd["layer4.0.conv2.weight"].grad), g["dw_last_0_conv2_stats"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(sd["layer4.0.downsample.1.weight"].grad.numpy(), g["dgamma_last_0_ds"], rtol=2e-3, atol=1e-3)
    with torch.no_grad():
        ye = O.resnet3d_forward(x, sd, train=False, layers=layers)
    np.testing.assert_allclose(ye.flatten().numpy(), g["out_eval"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("S,N", [(128, 2048), (256, 1024)])
def test_g1_resnet3d_variants_forward(golden, S, N):
    """Layer4 at the BASELINE grid and the 7 x 256^3 grid (grid_step 1/256, average pool 16, NeRAF_resnet3d.py:150-156): forward
    only (the 256^3 pass is 0.76 TFLOP of CPU convolutions)."""
    g = golden(f"g1_resnet3d_{S}_{N}")
    sd, x, layers = _run_resnet_variant(S, N, False)
    with torch.no_grad():
        y, st = O.resnet3d_forward(x, sd, train=True, layers=layers, return_stages=True)
    assert y.shape == (1, N, 1, 1, 1)
    np.testing.assert_allclose(y.flatten().numpy(), g["out_train"], rtol=2e-4, atol=2e-5)
    for k in ["conv1", "maxpool", "layer1", "layer2", "layer3"] + (["layer4"] if N == 2048 else []):
        np.testing.assert_allclose(stats(st[k]), g["stage_" + k], rtol=1e-4)


def _drive_refresh(gs, bs, steps, start=0):
    grid = O.reset_grid(gs)
    coords = O.coordinates_to_render(gs)
    dirs = O.fixed_viewing_directions()
    aabb = torch.tensor([[-3.5, -2.0, -4.5], [4.0, 2.5, 5.0]])
    cursor, cursors, first = start, [], None
    for _ in range(steps):
        s, n, cursor = O.refresh_window(cursor, bs, coords.shape[0])
        c01 = coords[s:s + n]
        ori = O.refresh_world_positions(c01, aabb)
        if first is None:
            first = ori
        rgbs, dens = [], []
        for j in range(dirs.shape[0]):
            r, d = synth.toy_field(ori, dirs[j].expand(n, -1))
            rgbs.append(r)
            dens.append(d)
        rgb = torch.stack(rgbs).mean(0)
        den = torch.stack(dens).mean(0)
        grid = O.grid_refresh_scatter(grid, c01, rgb, den, gs)
        cursors.append(cursor)
    return grid, cursors, first, dirs


def test_g4_grid_small(golden):
    g = golden("g4_grid")
    grid, cursors, first, dirs = _drive_refresh(1 / 16, 1500, 4)
    np.testing.assert_array_equal(np.array(cursors), g["cursors_s16"])
    np.testing.assert_allclose(dirs.numpy(), g["view_dirs"], rtol=0, atol=0)
    np.testing.assert_allclose(first[:6].numpy(), g["ori_first_s16"], rtol=1e-6)
    np.testing.assert_allclose(grid[:4].numpy(), g["grid_s16"], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(grid[4:, :2, :2, :].numpy(), g["grid_coords_s16"])


def test_g4_grid_full_size_wrap(golden):
    g = golden("g4_grid")
    grid, cursors, first, _ = _drive_refresh(1 / 128, 4096, 2, start=128 ** 3 - 4096 - 1000)
    np.testing.assert_array_equal(np.array(cursors), g["cursors_s128"])
    np.testing.assert_allclose(np.stack([stats(grid[c]) for c in range(7)]), g["grid_stats_s128"], rtol=1e-5, atol=1e-9)
    nz = (grid[3] != 0).nonzero()
    assert nz.shape[0] == int(g["grid_nnz_s128"])
    np.testing.assert_array_equal(nz[:4].numpy(), g["grid_first_nz_s128"])
    np.testing.assert_array_equal(nz[-4:].numpy(), g["grid_last_nz_s128"])
    np.testing.assert_allclose(grid[:4, 0, 0, :64].numpy(), g["grid_slab_s128"], rtol=1e-5, atol=1e-6)


# ---- unpinned encodings: property tests ----------------------------------
def test_nerf_encoding_properties():
    x = torch.rand(32, 3, dtype=torch.float64)
    e = O.nerf_encoding(x)
    assert e.shape == (32, 63)
    # sin / cos pairs: e[:, :30]^2 + e[:, 30:60]^2 == 1
    np.testing.assert_allclose((e[:, :30] ** 2 + e[:, 30:60] ** 2).numpy(), 1.0, atol=1e-9)
    np.testing.assert_array_equal(e[:, 60:].numpy(), x.numpy())
    assert O.nerf_encoding(torch.rand(5, 1)).shape == (5, 21)
    # frequency k of dim 0 is sin(2 pi x 2^(8k/9))
    k = 4
    # (frequencies are float32-rounded as in nerfstudio -> phase error <= 2*pi*2^8*6e-8)
    np.testing.assert_allclose(e[:, k].numpy(), np.sin(2 * np.pi * x[:, 0].numpy() * 2 ** (8 * k / 9)), atol=2e-4)


def test_sh4_orthonormal_on_sphere():
    # Monte-Carlo / quadrature check that the 16 functions are orthonormal on S^2.
    n_t, n_p = 64, 128
    ct, wt = np.polynomial.legendre.leggauss(n_t)
    ph = (np.arange(n_p) + 0.5) * 2 * np.pi / n_p
    CT, PH = np.meshgrid(ct, ph, indexing="ij")
    st = np.sqrt(1 - CT ** 2)
    d = np.stack([st * np.cos(PH), st * np.sin(PH), CT], -1).reshape(-1, 3)
    w = (wt[:, None] * np.ones_like(PH) * 2 * np.pi / n_p).reshape(-1)
    Y = O.sh4_encoding(torch.from_numpy((d + 1) / 2)).numpy()
    G = (Y * w[:, None]).T @ Y
    np.testing.assert_allclose(G, np.eye(16), atol=1e-6)


def test_audio_prologue_layout_and_selector():
    b = synth.audio_batch(64, 1, 513, 60)
    aabb = torch.from_numpy(synth.audio_aabb())
    q = O.audio_prologue(*[torch.from_numpy(b[k]) for k in ("time_query", "mic_pose", "source_pose", "rot")], aabb, 60)
    assert q.shape == (64, 163) and q.dtype == torch.float32
    # row 0 has the mic pushed outside the box -> normalised mic zeroed -> raw coords (last 3 of the 63) are 0
    assert torch.all(q[0, 21 + 60:21 + 63] == 0)
    assert torch.all(q[5, 21 + 60:21 + 63] != 0)
    np.testing.assert_allclose(q[:, 20].numpy(), b["time_query"] / 59.0, rtol=1e-6)


def test_gated_resnet3d_with_own_gates_is_the_pinned_forward():
    """oracle.audio.resnet3d_forward_gated (the parity target of the HIP backward) with the gates of its own forward reproduces the
    G1-pinned resnet3d_forward: same feature, same gradient for every parameter and for the grid (32^3 grid, seconds on CPU)."""
    from oracle import audio as O
    sd = {k: torch.from_numpy(v) for k, v in synth.resnet3d_state_dict(7).items()}
    S = 32
    x = torch.from_numpy(synth.uniform("gated.grid", (1, 7, S, S, S), 0.0, 1.0))
    w = torch.from_numpy(synth.uniform("g1.outw", (1024,), -1.0, 1.0))
    gates = O.resnet3d_gates_from_forward(x, sd)
    assert gates["pool_arg"].shape == (8 ** 3, 64) and gates["layer3.5.out"].shape == (1024, 2, 2, 2)
    res = []
    for gated in (True, False):
        sdg = {k: (v.clone().requires_grad_(True) if k.endswith("weight") or k.endswith("bias") else v) for k, v in sd.items()}
        xg = x.clone().requires_grad_(True)
        y = O.resnet3d_forward_gated(xg, sdg, gates) if gated else O.resnet3d_forward(xg, sdg, train=True)
        (y.flatten() * w).sum().backward()
        res.append((y.detach(), xg.grad, {k: v.grad for k, v in sdg.items() if v.requires_grad}))
    (ya, dxa, ga), (yb, dxb, gb) = res
    assert len(ga) == 129
    np.testing.assert_allclose(ya.numpy(), yb.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dxa.numpy(), dxb.numpy(), rtol=1e-5, atol=1e-9)
    for k in ga:
        np.testing.assert_allclose(ga[k].numpy(), gb[k].numpy(), rtol=1e-5, atol=1e-9, err_msg=k)

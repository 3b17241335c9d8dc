"""GPU tests of the NeRAFAudioModel mirror (neraf_amd/model.py): grid refresh (A3), end-to-end get_outputs (A1+A4+A5),
loss dict (A6), eval branch (A7), against the CPU oracle."""
import numpy as np
import pytest
import torch

from neraf_amd import synth

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def models():
    from neraf_amd.model import NeRAFAudioModel, NeRAFAudioModelConfig
    from neraf_amd.vision import NeRAFVisionModel
    from oracle import vision as V
    dev = torch.device("cuda:0")
    spec = V.NerfactoSpec()
    tot = (spec.prop_grids[0].total, spec.prop_grids[1].total, spec.main_grid.total)
    P = {k: T(v) for k, v in synth.vision_params(tot, num_train_data=210, table_scale=0.5).items()}
    vaabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    vm = NeRAFVisionModel(vaabb, 210)
    with torch.no_grad():
        f = vm.field.module
        f.table.copy_(P["field.table"])
        for k in ("base_w0", "base_w1", "head_w0", "head_w1", "head_w2", "embedding"):
            getattr(f, k).copy_(P["field." + k])
    cfg = NeRAFAudioModelConfig(dataset="RAF", grid_step=1 / 64)
    am = NeRAFAudioModel(cfg, T(synth.audio_aabb()))
    am.field.load_state_dict({k: T(v) for k, v in synth.nacf_state_dict(1187, 512, 1, 513).items()})
    am.resnet3d.backbone_net.load_state_dict({k: T(v) for k, v in synth.resnet3d_state_dict(7).items()})
    return vm.to(dev), am.to(dev), {k: v.half().float() for k, v in P.items()}, spec, V, dev


def test_state_dict_keys_match_reference_prefixes(models):
    _, am, *_ = models
    keys = set(am.state_dict().keys())
    for k in ("field.soundfield.0.weight", "field.STFT_linear.0.bias", "resnet3d.backbone_net.conv1.weight",
              "resnet3d.backbone_net.layer3.5.bn3.running_var", "resnet3d.backbone_net.layer2.0.downsample.1.weight", "grid"):
        assert k in keys, k
    assert am.max_len == 60 and am.mic_ch == 1 and am.field.in_size == 1187      # RAF: int(0.32*48000)//256 (NeRAF_model.py:128)


def test_grid_refresh_vs_oracle(models):
    """query_grid_one_batch (NeRAF_model.py:294-407): window walk incl. the partial last batch + wrap, 18-direction mean,
    alpha, slab writes -- against oracle.grid_refresh_scatter (pinned by G4) fed by the oracle field."""
    from oracle import audio as O
    vm, am, P16, spec, V, dev = models
    am.reset_grid()
    am.grid_batch_i = 64 ** 3 - 4096 - 1000
    gs = 1 / 64
    grid_o = O.reset_grid(gs)
    coords = O.coordinates_to_render(gs)
    dirs = O.fixed_viewing_directions()
    aabb = vm.field.module.aabb.cpu()
    cursor = am.grid_batch_i
    for _ in range(2):
        am.query_grid_one_batch(0, vm.field, renderer_rgb=vm.renderer_rgb, batch_size=4096)
        s, n, cursor = O.refresh_window(cursor, 4096, coords.shape[0])
        c01 = coords[s:s + n]
        ori = O.refresh_world_positions(c01, aabb)
        rgbs, dens = [], []
        for j in range(18):
            r, d = V.field_forward(ori, dirs[j].expand(n, -1), torch.zeros(n, dtype=torch.long), P16, spec, contract=False, aabb=aabb)
            rgbs.append(r); dens.append(d[:, None])
        grid_o = O.grid_refresh_scatter(grid_o, c01, torch.stack(rgbs).mean(0), torch.stack(dens).mean(0), gs)
        assert am.grid_batch_i == cursor
    assert vm.field.module.spatial_distortion == "linf"          # restored (NeRAF_model.py:407)
    g = am.grid.cpu()
    np.testing.assert_array_equal(g[4:].numpy(), grid_o[4:].numpy())
    assert float((g[:3] - grid_o[:3]).abs().max()) <= 4e-3
    np.testing.assert_allclose(g[3].numpy(), grid_o[3].numpy(), rtol=2e-2, atol=1e-7)
    assert int((g[3] != 0).sum()) == int((grid_o[3] != 0).sum()) == 5096


def test_get_outputs_loss_and_eval_branch(models):
    from oracle import audio as O
    vm, am, P16, spec, V, dev = models
    am.train()
    for m in am.resnet3d.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.momentum = 0.0
    B = 256
    b = {k: T(v) for k, v in synth.audio_batch(B, 1, 513, 60, tag="t.model").items()}
    y = am.get_outputs({k: v.to(dev) for k, v in b.items()})
    assert y.shape == (B, 1, 513)
    # oracle: ResNet3D (train-mode BN) on the same grid -> NAcF
    sdr = {k: T(v) for k, v in synth.resnet3d_state_dict(7).items()}
    sdn = {k: T(v) for k, v in synth.nacf_state_dict(1187, 512, 1, 513).items()}
    with torch.no_grad():
        feat = O.resnet3d_forward(am.grid.cpu().unsqueeze(0), sdr, train=True).flatten()
        yo = O.audio_get_outputs(b, feat, sdn, T(synth.audio_aabb()), 60)
    rel = float((y.detach().cpu() - yo).norm() / yo.norm())
    assert rel <= 1e-2, rel
    ld = am.get_loss_dict(y, {k: v.to(dev) for k, v in b.items()})
    lo = O.audio_loss_dict(yo, b["data"])
    np.testing.assert_allclose(ld["audio_sc_loss"].item(), lo["audio_sc_loss"].item(), rtol=2e-2)
    np.testing.assert_allclose(ld["audio_mag_loss"].item(), lo["audio_mag_loss"].item(), rtol=2e-2)
    (ld["audio_sc_loss"] + ld["audio_mag_loss"]).backward()
    assert am.field.soundfield[0].weight.grad is not None and bool(torch.isfinite(am.field.soundfield[0].weight.grad).all())
    # eval branch: T = 60 slices of one RIR
    am.eval()
    item = {"mic_pose": b["mic_pose"][5], "source_pose": b["source_pose"][5], "rot": b["rot"][5],
            "data": T(synth.uniform("t.model.gt", (1, 513, 60), -6.0, 1.0))}
    out = am.get_outputs_for_camera(None, None, batch_audio=item)
    assert out["raw_output"].shape == (60, 1, 513) and out["stft_ch_0"].shape == (513, 60, 1)
    assert out["comparison_ch_0"].shape == (513, 120, 1) and out["grid"].shape == (64, 64, 3) and out["grid_density"].shape == (64, 64, 1)
    with torch.no_grad():
        feat_e = O.resnet3d_forward(am.grid.cpu().unsqueeze(0), sdr, train=False).flatten()
        be = {"time_query": torch.arange(60), "mic_pose": item["mic_pose"].reshape(1, 3).expand(60, -1),
              "source_pose": item["source_pose"].reshape(1, 3).expand(60, -1), "rot": item["rot"].reshape(1, 3).expand(60, -1)}
        ye = O.audio_get_outputs(be, feat_e, sdn, T(synth.audio_aabb()), 60)
    rel = float((out["raw_output"].cpu() - ye).norm() / ye.norm())
    assert rel <= 1e-2, rel
    np.testing.assert_array_equal(out["stft_ch_0"][:, :, 0].numpy(), np.flip(out["raw_output"][:, 0, :].cpu().numpy().T, 0))
    # metric chain on the predicted RIR: Griffin-Lim on the GPU, T60 / EDT / C50 / RAF spectral error on the host
    n = 60 * 256
    tt = np.arange(n) / 48000.0
    wav = (synth.normal("t.model.wav", (1, n), 1.0, np.float64) * np.exp(-tt / 0.06)).astype(np.float32)
    mb = {"data": item["data"], "waveform": T(wav)}
    met = am.get_audio_metrics(out, mb, generator=torch.Generator(device=dev).manual_seed(0))
    assert set(met) == {"audio_T60", "audio_total_invalids_T60", "audio_stft_error", "audio_EDT", "audio_C50"}
    assert all(np.isfinite(v) for v in met.values())
    # the image half (NeRAF_model.py:763-803): viridis panels normalised by the ground truth's range, prediction | ground truth
    from matplotlib import cm
    met2, img = am.get_image_metrics_and_images(out, mb, generator=torch.Generator(device=dev).manual_seed(0))
    assert met2 == met and set(img) == {"comparison_ch_0", "grid", "grid_density"}
    assert img["comparison_ch_0"].shape == (513, 120, 3) and img["grid_density"].shape == (64, 64, 3)
    g = out["gt_ch_0"].numpy().squeeze().astype(np.float64)
    lo, hi = g.min(), g.max()
    np.testing.assert_array_equal(img["comparison_ch_0"][:, 60:].numpy(), cm.viridis((g - lo) / (hi - lo))[..., :3])
    pr = out["stft_ch_0"].numpy().squeeze().astype(np.float64)
    np.testing.assert_array_equal(img["comparison_ch_0"][:, :60].numpy(), cm.viridis((pr - lo) / (hi - lo))[..., :3])
    sm = am.get_metrics_dict(out["raw_output"].permute(1, 2, 0), {"data": item["data"]})
    assert set(sm) == {"audio_mag", "audio_spectral_loss"} and np.isfinite(float(sm["audio_mag"]))
    am.train()


def test_grid_refresh_and_outputs_on_the_256_cubed_grid(models):
    """grid_step = 1/256 (NeRAF_model.py:91, NeRAF_resnet3d.py:150-156): 16.7 M cells, 4096 per refresh (a full sweep = 4096 steps).
    The window walk incl. the wrap at the end of the grid, the slab writes and the cell-centre channels against the oracle scatter
    (pinned by G4) fed by the oracle field -- then the encoder consumes the 7 x 256^3 grid: get_outputs against oracle ResNet3D
    (eval mode) -> oracle NAcF."""
    from oracle import audio as O
    from neraf_amd.model import NeRAFAudioModel, NeRAFAudioModelConfig
    vm, _, P16, spec, V, dev = models
    gs, S = 1 / 256, 256
    am = NeRAFAudioModel(NeRAFAudioModelConfig(dataset="RAF", grid_step=gs), T(synth.audio_aabb()))
    sdn = {k: T(v) for k, v in synth.nacf_state_dict(1187, 512, 1, 513).items()}
    sdr = {k: T(v) for k, v in synth.resnet3d_state_dict(7).items()}
    am.field.load_state_dict(sdn)
    am.resnet3d.backbone_net.load_state_dict(sdr)
    am = am.to(dev)
    assert tuple(am.grid.shape) == (7, S, S, S)
    am.reset_grid()
    am.grid_batch_i = S ** 3 - 4096 - 1000
    grid_o = O.reset_grid(gs)
    coords = O.coordinates_to_render(gs)
    dirs = O.fixed_viewing_directions()
    aabb = vm.field.module.aabb.cpu()
    cursor = am.grid_batch_i
    for _ in range(2):
        am.query_grid_one_batch(0, vm.field, renderer_rgb=vm.renderer_rgb, batch_size=4096)
        s0, n, cursor = O.refresh_window(cursor, 4096, coords.shape[0])
        c01 = coords[s0:s0 + n]
        ori = O.refresh_world_positions(c01, aabb)
        rgbs, dens = [], []
        for j in range(18):
            r, d = V.field_forward(ori, dirs[j].expand(n, -1), torch.zeros(n, dtype=torch.long), P16, spec, contract=False, aabb=aabb)
            rgbs.append(r); dens.append(d[:, None])
        grid_o = O.grid_refresh_scatter(grid_o, c01, torch.stack(rgbs).mean(0), torch.stack(dens).mean(0), gs)
        assert am.grid_batch_i == cursor
    g = am.grid.cpu()
    np.testing.assert_array_equal(g[4:].numpy(), grid_o[4:].numpy())
    assert float((g[:3] - grid_o[:3]).abs().max()) <= 4e-3
    np.testing.assert_allclose(g[3].numpy(), grid_o[3].numpy(), rtol=2e-2, atol=1e-7)
    assert int((g[3] != 0).sum()) == int((grid_o[3] != 0).sum()) == 5096
    am.eval()
    B = 64
    b = {k: T(v) for k, v in synth.audio_batch(B, 1, 513, 60, tag="t.model256").items()}
    with torch.no_grad():
        y = am.get_outputs({k: v.to(dev) for k, v in b.items()})
        feat = O.resnet3d_forward(g.unsqueeze(0), sdr, train=False).flatten()
        yo = O.audio_get_outputs(b, feat, sdn, T(synth.audio_aabb()), 60)
    rel = float((y.cpu() - yo).norm() / yo.norm())
    print(f"256^3 grid: refresh == oracle scatter; eval chain vs oracle rel-L2 {rel:.2e}")
    assert rel <= 5e-3, rel


def test_audio_model_with_layer4_features():
    """NeRAFAudioModelConfig(N_features=2048) (NeRAF_model.py:92, :181-189): the encoder keeps resnet50's layer4, the NAcF's first
    layer takes 2048 + 163 inputs.  State-dict keys as the reference's; eval-mode outputs (running statistics: no small-batch
    amplification) against oracle ResNet3D -> oracle NAcF; train-mode outputs against the oracle NAcF on the feature the HIP encoder
    produced; loss + backward reach every parameter, layer4's and the 2211-column first layer's included."""
    from oracle import audio as O
    from neraf_amd.model import NeRAFAudioModel, NeRAFAudioModelConfig
    dev = torch.device("cuda:0")
    am = NeRAFAudioModel(NeRAFAudioModelConfig(dataset="RAF", grid_step=1 / 64, N_features=2048), T(synth.audio_aabb()))
    assert am.field.in_size == 2048 + 163 and am.resnet3d.backbone_net.N_features == 2048
    sdn = {k: T(v) for k, v in synth.nacf_state_dict(2211, 512, 1, 513).items()}
    sdr = {k: T(v) for k, v in synth.resnet3d_state_dict(7, layers=(3, 4, 6, 3)).items()}
    am.field.load_state_dict(sdn)
    am.resnet3d.backbone_net.load_state_dict(sdr, strict=True)
    keys = set(am.state_dict().keys())
    for k in ("resnet3d.backbone_net.layer4.0.downsample.1.running_var", "resnet3d.backbone_net.layer4.2.conv3.weight"):
        assert k in keys, k
    assert tuple(am.state_dict()["field.soundfield.0.weight"].shape) == (5096, 2211)
    am = am.to(dev)
    with torch.no_grad():
        am.grid.copy_(T(synth.uniform("t.model.grid2048", (7, 64, 64, 64), 0.0, 1.0)))
    am.mark_grid_written()
    B = 192
    b = {k: T(v) for k, v in synth.audio_batch(B, 1, 513, 60, tag="t.model2048").items()}
    bd = {k: v.to(dev) for k, v in b.items()}
    # eval mode: the whole chain against the oracle
    am.eval()
    with torch.no_grad():
        ye = am.get_outputs(bd)
        feat_o = O.resnet3d_forward(am.grid.cpu().unsqueeze(0), sdr, train=False, layers=(3, 4, 6, 3)).flatten()
        yo = O.audio_get_outputs(b, feat_o, sdn, T(synth.audio_aabb()), 60)
    assert ye.shape == (B, 1, 513)
    rel = float((ye.cpu() - yo).norm() / yo.norm())
    print(f"N_features = 2048, eval chain vs oracle: rel-L2 {rel:.2e}")
    assert rel <= 5e-3, rel
    # train mode: NAcF on the HIP encoder's own feature; loss; backward
    am.train()
    cap = {}
    orig = am.scene_feature
    am.scene_feature = lambda: cap.setdefault("f", orig())
    y = am.get_outputs(bd)
    am.scene_feature = orig
    with torch.no_grad():
        yo2 = O.audio_get_outputs(b, cap["f"].detach().flatten().cpu(), sdn, T(synth.audio_aabb()), 60)
    rel2 = float((y.detach().cpu() - yo2).norm() / yo2.norm())
    assert rel2 <= 5e-3, rel2
    ld = am.get_loss_dict(y, bd)
    lo = O.audio_loss_dict(yo2, b["data"])
    np.testing.assert_allclose(ld["audio_sc_loss"].item(), lo["audio_sc_loss"].item(), rtol=1e-2)
    (ld["audio_sc_loss"] + ld["audio_mag_loss"]).backward()
    torch.cuda.synchronize()
    n = 0
    for name, p_ in am.named_parameters():
        assert p_.grad is not None and bool(torch.isfinite(p_.grad).all()), name
        n += 1
    assert n == 159 + 12                      # 53 conv + 53 x 2 BatchNorm affine + 6 Linear x 2
    assert float(am.resnet3d.backbone_net.layer4[2].conv3.weight.grad.abs().max()) > 0
    assert float(am.field.soundfield[0].weight.grad[:, :2048].abs().max()) > 0


def test_refresh_gradient_edge_vs_oracle(models):
    """The autograd edge from the refreshed grid cells into the radiance field (NeRAF_model.py:395-400): for a random
    upstream d loss / d grid[0:4, window], gradients of the field parameters against autograd through the oracle's refresh
    (field_forward in AABB mode -> mean over the 18 directions -> alpha).  Tolerance 6e-2 relative L2 (as the field tests)."""
    from neraf_amd.model import _RefreshFn
    from oracle import audio as O
    vm, am, P16, spec, V, dev = models
    f = vm.field.module
    n, gs = 512, 1 / 64
    coords = O.coordinates_to_render(gs)[1000:1000 + n]
    dirs = O.fixed_viewing_directions()
    aabb = f.aabb.cpu()
    ori = O.refresh_world_positions(coords, aabb)
    dvals = T(synth.normal("t.edge.dvals", (4, n)))
    # oracle
    P = {k: v.clone().requires_grad_(True) for k, v in P16.items()}
    # the 18 directions as ONE oracle call over 18 n rows (rows are independent): one autograd pass into the dense 2^19 x 16 table
    # gradient instead of eighteen (the test took 97 s of host time on a slow box)
    r, d = V.field_forward(ori[None].expand(18, n, 3).reshape(-1, 3), dirs[:, None, :].expand(18, n, 3).reshape(-1, 3),
                           torch.zeros(18 * n, dtype=torch.long), P, spec, contract=False, aabb=aabb)
    rgb_m, den_m = r.reshape(18, n, 3).mean(0), d.reshape(18, n).mean(0)
    vals_o = torch.cat([rgb_m.t(), torch.clip(1 - torch.exp(-1e-2 * den_m), 0, 1)[None]], 0)
    (vals_o * dvals).sum().backward()
    # HIP
    old = f.spatial_distortion
    f.spatial_distortion = None
    for p in f.parameters():
        p.grad = None
    try:
        dirs_d = dirs.to(dev).contiguous()
        vals = _RefreshFn.apply(f, coords.float().to(dev).contiguous(), f.aabb, dirs_d, 18, 1e-2, am._refresh_consts(dirs_d, n),
                                None, *f.grad_params())
        assert float((vals.detach().cpu() - vals_o.detach()).abs().max()) <= 4e-3
    finally:
        f.spatial_distortion = old
    # as in a training step (NeRAF_model.py:302, :407): the contraction is switched back on BEFORE the backward pass runs -- the
    # backward must still map positions the way the forward did (round 5: it did not; the refresh's hash gradients went to the cells
    # of the contracted positions)
    assert f.spatial_distortion is not None
    (vals * dvals.to(dev)).sum().backward()
    for name, p in (("field.table", f.table), ("field.base_w0", f.base_w0), ("field.base_w1", f.base_w1), ("field.head_w0", f.head_w0),
                    ("field.head_w1", f.head_w1), ("field.head_w2", f.head_w2)):
        a, b = p.grad.double().cpu(), P[name].grad.double()
        assert float((a - b).norm() / b.norm()) <= 6e-2, name
    e = f.embedding.grad.double().cpu()
    assert float((e[0] - P["field.embedding"].grad[0].double()).norm() / P["field.embedding"].grad[0].double().norm()) <= 6e-2
    assert float(e[1:].abs().max()) == 0.0        # the refresh uses camera index 0 only


def test_joint_backward_reaches_all_parameter_groups(models):
    """One joint step: the audio loss reaches the NAcF MLP, the ResNet3D and (through the refreshed cells) the field."""
    vm, am, P16, spec, V, dev = models
    vm.train(); am.train()
    for p in list(vm.parameters()) + list(am.parameters()):
        p.grad = None
    am.query_grid_one_batch(0, vm.field, renderer_rgb=vm.renderer_rgb, batch_size=2048)
    b = {k: T(v).to(dev) for k, v in synth.audio_batch(128, 1, 513, 60, tag="t.joint").items()}
    y = am.get_outputs(b)
    ld = am.get_loss_dict(y, b)
    (ld["audio_sc_loss"] + ld["audio_mag_loss"]).backward()
    for name, p in (("nacf", am.field.soundfield[2].weight), ("resnet conv", am.resnet3d.backbone_net.layer2[1].conv2.weight),
                    ("resnet bn", am.resnet3d.backbone_net.bn1.weight), ("field table", vm.field.module.table),
                    ("field head", vm.field.module.head_w1)):
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0, name
    assert vm.proposal_networks[0].table.grad is None          # the audio loss does not touch the proposal networks


def test_metric_chain_recovers_t60_of_a_synthetic_exponential_rir():
    """SURVEY 8f rank 1, end to end on the device: a noise RIR with a known exponential decay (T60 = 6.91 tau) -> its magnitude STFT ->
    seeded Griffin-Lim on the GPU (rocFFT) -> RAFEvaluator.get_full_metrics against the true waveform.  The reconstruction keeps the
    decay: T60 error below 5 %, EDT error below 20 ms, C50 error below 1 dB, and the RAF spectral error of a perfect magnitude
    prediction is small; a prediction whose decay is 1.6 x slower must show up as a large T60 error."""
    from neraf_amd.evaluator import GriffinLim, RAFEvaluator, compute_t60, spectrogram
    dev = torch.device("cuda:0")
    fs, n_fft, win, hop, Tn = 48000, 1024, 512, 256, 60
    n = hop * (Tn - 1)
    tt = np.arange(n) / fs
    ev = RAFEvaluator(fs=fs)
    gl = GriffinLim(n_fft=n_fft, win_length=win, hop_length=hop, power=1).to(dev)

    def rir(tau, tag):
        return (synth.normal(tag, (1, n), 1.0, np.float64) * np.exp(-tt / tau)).astype(np.float32)
    wav_gt = rir(0.05, "t60.gt")
    mag_gt = spectrogram(T(wav_gt).to(dev), n_fft, win, hop).abs()[..., :Tn]                      # [1, 513, 60]
    assert mag_gt.shape == (1, 513, Tn)
    g = torch.Generator(device=dev).manual_seed(0)
    wav_rec = gl(mag_gt, generator=g).cpu().numpy()
    log_gt = torch.log(mag_gt + 1e-3).cpu().numpy()
    met = ev.get_full_metrics(mag_gt.cpu().numpy(), mag_gt.cpu().numpy(), wav_gt, wav_rec, wav_rec, log_gt, log_gt)
    t_true, _ = compute_t60(wav_gt, wav_gt, fs=fs, advanced=True)
    assert abs(float(t_true[0]) - 6.91 * 0.05) / (6.91 * 0.05) < 0.1          # the estimator itself sees the analytic decay
    assert met["audio_total_invalids_T60"] == 0.0
    assert met["audio_T60"] < 5.0, met
    assert met["audio_EDT"] < 0.02 and met["audio_C50"] < 1.0, met
    assert met["audio_stft_error"] < 0.5, met
    # sensitivity: a slower decay is reported
    wav_slow = rir(0.08, "t60.gt")
    mag_slow = spectrogram(T(wav_slow).to(dev), n_fft, win, hop).abs()[..., :Tn]
    rec_slow = gl(mag_slow, generator=torch.Generator(device=dev).manual_seed(0)).cpu().numpy()
    met2 = ev.get_full_metrics(mag_slow.cpu().numpy(), mag_gt.cpu().numpy(), wav_gt, rec_slow, wav_rec, torch.log(mag_slow + 1e-3).cpu().numpy(), log_gt)
    assert met2["audio_T60"] > 30.0, met2


def test_device_rir_bank_equals_cpu_tokenisation_and_reference_item_semantics():
    """SURVEY 8f rank 2 on the device: DeviceRIRBank built ON the GPU (rocFFT) equals the same tokenisation on the CPU
    (log(|STFT| + 1e-3), slice-major), and a sampled batch has the reference's item semantics (NeRAF_dataset.py:85-86, :127-130:
    flat index -> (rir, time slice), 'data' [B, C, F] = that slice of that RIR, poses / rot of that RIR, int64 time_query)."""
    from neraf_amd.data import DeviceRIRBank
    dev = torch.device("cuda:0")
    n_rir, n = 5, 256 * 59
    waves = T(np.stack([synth.normal(f"bank.rir{i}", (n,), 1.0, np.float64) * np.exp(-np.arange(n) / 48000.0 / 0.04) for i in range(n_rir)])).float()
    mic, src = T(synth.uniform("bank.mic", (n_rir, 3), -2, 2)).double(), T(synth.uniform("bank.src", (n_rir, 3), -2, 2)).double()
    rot = T(synth.uniform("bank.rot", (n_rir, 3), 0, 1)).double()
    cpu = DeviceRIRBank.from_waveforms(waves, 48000, 60, mic, src, rot)
    gpu = DeviceRIRBank.from_waveforms(waves, 48000, 60, mic, src, rot, device=dev)
    assert gpu.log_mag.device.type == "cuda" and tuple(gpu.log_mag.shape) == (n_rir, 60, 1, 513)
    np.testing.assert_allclose(gpu.log_mag.cpu().numpy(), cpu.log_mag.numpy(), rtol=0, atol=2e-3)
    assert float((gpu.log_mag.cpu() - cpu.log_mag).abs().mean()) < 1e-5
    g = torch.Generator(device=dev).manual_seed(3)
    b = gpu.next_train(512, generator=g)
    assert b["data"].shape == (512, 1, 513) and b["time_query"].dtype == torch.int64 and b["mic_pose"].dtype == torch.float64
    r, t = b["audio_idx"].cpu(), b["time_query"].cpu()
    assert int(t.max()) < 60 and int(r.max()) < n_rir
    np.testing.assert_array_equal(b["data"].cpu().numpy(), gpu.log_mag.cpu()[r, t].numpy())
    np.testing.assert_array_equal(b["mic_pose"].cpu().numpy(), mic[r].numpy())
    item = gpu.get_data(3 * 60 + 7)                                    # flat index -> (3, 7)
    assert item["audio_idx"] == 3 and item["time_query"] == 7 and torch.equal(item["data"], gpu.log_mag[3, 7])
    ev = gpu.get_data_eval(2)
    assert tuple(ev["data"].shape) == (1, 513, 60) and torch.equal(ev["data"][:, :, 11], gpu.log_mag[2, 11])


def test_window_only_grid_conversion_keeps_the_input_image_exact():
    """The ResNet3D host layer re-converts only the refreshed window of the grid into its persistent fp16 channels-last input image
    (neraf_resnet3d_fwd win_cells > 0) when the audio model vouches for everything else: after every step the image in the workspace
    must equal a full conversion of the current grid bit for bit -- across the cursor's wrap-around, after an in-place edit of the
    grid by somebody else (version counter), after a reset and after load_state_dict (all of which must fall back to the full pass)."""
    import ctypes as C
    from neraf_amd import _lib
    from neraf_amd import resnet3d as R3
    from neraf_amd.model import NeRAFAudioModel, NeRAFAudioModelConfig
    from neraf_amd.vision import NeRAFVisionModel
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    vm = NeRAFVisionModel(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 8).to(dev).train()
    am = NeRAFAudioModel(NeRAFAudioModelConfig(dataset="RAF", grid_step=1 / 64), T(synth.audio_aabb())).to(dev).train()
    bb = am.resnet3d.backbone_net
    lib = _lib.load()
    off, rows, cols = C.c_size_t(0), C.c_int(0), C.c_int(0)
    _lib.check(lib.neraf_resnet3d_debug_locate(C.byref(bb._desc), 7, 0, C.byref(off), C.byref(rows), C.byref(cols)), 0)
    assert (rows.value, cols.value) == (64 ** 3, 8)

    def image():
        return bb._ws[off.value:off.value + rows.value * 16].view(torch.float16).reshape(rows.value, 8).clone()

    def expect():
        g = am.grid.reshape(7, -1).t().half()
        return torch.cat([g, torch.zeros((g.shape[0], 1), dtype=torch.float16, device=dev)], dim=1)

    calls = []
    real = lib.neraf_resnet3d_fwd

    def spy(*a):
        calls.append(int(a[9]))                      # win_cells
        return real(*a)
    lib.neraf_resnet3d_fwd = spy
    try:
        n_cells = 64 ** 3
        bs = 100_000                                 # 3 windows per sweep: the third is partial (62,144 cells) and wraps the cursor
        for step in range(7):
            am.query_grid_one_batch(step, vm.field, renderer_rgb=vm.renderer_rgb, batch_size=bs)
            am.scene_feature()
            assert torch.equal(image(), expect()), step
        assert calls[0] == 0 and calls[1] == bs and calls[2] == n_cells - 2 * bs and calls[3] == bs, calls
        # somebody else edits the grid in place: noticed through the version counter -> full conversion
        with torch.no_grad():
            am.grid[0, 3, 3, 3] += 0.25
        am.query_grid_one_batch(7, vm.field, renderer_rgb=vm.renderer_rgb, batch_size=bs)
        am.scene_feature()
        assert calls[-1] == 0 and torch.equal(image(), expect())
        am.query_grid_one_batch(8, vm.field, renderer_rgb=vm.renderer_rgb, batch_size=bs)
        am.scene_feature()
        assert calls[-1] > 0 and torch.equal(image(), expect())
        # two refreshes between two features: a generation was missed -> full
        am.query_grid_one_batch(9, vm.field, renderer_rgb=vm.renderer_rgb, batch_size=bs)
        am.query_grid_one_batch(10, vm.field, renderer_rgb=vm.renderer_rgb, batch_size=bs)
        am.scene_feature()
        assert calls[-1] == 0 and torch.equal(image(), expect())
        # reset and load_state_dict are writes of unknown extent
        am.reset_grid()
        am.query_grid_one_batch(11, vm.field, renderer_rgb=vm.renderer_rgb, batch_size=bs)
        am.scene_feature()
        assert calls[-1] == 0 and torch.equal(image(), expect())
        sd = {k: v.clone() for k, v in am.state_dict().items()}
        sd["grid"] = sd["grid"] * 0.5
        am.load_state_dict(sd)
        am.query_grid_one_batch(12, vm.field, renderer_rgb=vm.renderer_rgb, batch_size=bs)
        am.scene_feature()
        assert calls[-1] == 0 and torch.equal(image(), expect())
        # the non-differentiable refresh (raw-pointer write, reported by the model) takes the window path too
        with torch.no_grad():
            am.query_grid_one_batch(13, vm.field, renderer_rgb=vm.renderer_rgb, batch_size=bs)
        am.scene_feature()
        assert calls[-1] > 0 and torch.equal(image(), expect())
    finally:
        lib.neraf_resnet3d_fwd = real

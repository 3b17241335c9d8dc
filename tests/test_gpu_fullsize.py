"""Parity at BASELINE.json's FULL sizes (configs[2] RAF joint step: 4096 rays + 2048 slices x 513 bins + 128^3 grid; configs[3]
per-GPU shape: 4096 rays + 808 slices x 2 x 257 bins, T = 101; configs[4]: full 684 x 1024 frames), where running the CPU oracle
on everything would take minutes.  Each test runs the HIP path at full size -- so the kernels take their full-size code paths:
LDS-owner hash-gradient scatter (>= 131072 samples), wide NAcF tiles, 22-chunk renders -- and compares with the oracle

  * on a strided subsample of per-ray / per-slice OUTPUTS (rays and slices are independent), and
  * on parameter GRADIENTS by restricting the upstream gradient to the subsample (the backward is linear in it), or in full where
    the oracle is fast enough (the NAcF MLP: 2048 slices take ~2 s on the host).

Tolerances as in the small-size tests they extend (test_gpu_vision.py, test_gpu_vision_train.py, test_gpu_nacf.py)."""
import copy

import numpy as np
import pytest
import torch

from neraf_amd import synth

pytestmark = pytest.mark.gpu
GRAD_REL_L2 = 6e-2


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _vision_P(vm):
    """The model's parameters as the oracle's dict, rounded to fp16 like the copies the kernels read."""
    f = vm.field.module
    P = {"field.table": f.table, "field.base_w0": f.base_w0, "field.base_w1": f.base_w1, "field.head_w0": f.head_w0,
         "field.head_w1": f.head_w1, "field.head_w2": f.head_w2, "field.embedding": f.embedding}
    for i, pn in enumerate(vm.proposal_networks):
        P[f"prop{i}.table"], P[f"prop{i}.w0"], P[f"prop{i}.w1"] = pn.table, pn.w0, pn.w1
    return {k: v.detach().half().float().cpu() for k, v in P.items()}


@pytest.fixture(scope="module")
def vm():
    from neraf_amd.vision import NeRAFVisionModel
    from oracle import vision as V
    dev = torch.device("cuda:0")
    spec = V.NerfactoSpec()
    tot = (spec.prop_grids[0].total, spec.prop_grids[1].total, spec.main_grid.total)
    P = {k: T(v) for k, v in synth.vision_params(tot, num_train_data=210, table_scale=0.5).items()}
    m = NeRAFVisionModel(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 210)
    with torch.no_grad():
        for i in range(2):
            m.proposal_networks[i].table.copy_(P[f"prop{i}.table"])
            m.proposal_networks[i].w0.copy_(P[f"prop{i}.w0"])
            m.proposal_networks[i].w1.copy_(P[f"prop{i}.w1"])
        f = m.field.module
        f.table.copy_(P["field.table"])
        for k in ("base_w0", "base_w1", "head_w0", "head_w1", "head_w2", "embedding"):
            getattr(f, k).copy_(P["field." + k])
    return m.to(dev), spec, V, dev


# ---- V3 / configs[4]: a full RAF frame -----------------------------------------------------------------------------------------
def test_full_frame_camera_render_crosses_chunks_and_matches_oracle(vm):
    """get_outputs_for_camera(camera, None, eval=True) (NeRAF_model.py:70-79) on one 684 x 1024 OPENCV camera: 700,416 rays =
    21 full chunks of 32,768 (NeRAF_config.py:95) + one of 12,288.  (1) the image equals the concatenation of direct per-chunk
    calls, bit for bit, for the first, a middle and the partial last chunk; (2) 300 pixels strided over the WHOLE frame (so from
    every chunk) equal the oracle's eval-mode render; (3) PSNR / image dict of get_image_metrics_and_images."""
    from neraf_amd.datamanagers import synthetic_cameras
    from neraf_amd.vision import RayBundle, psnr
    m, spec, V, dev = vm
    cams = synthetic_cameras(2)
    cam = cams[1]
    out = m.get_outputs_for_camera(cam, None, eval=True)
    H, W = 1024, 684
    assert out["rgb"].shape == (H, W, 3) and out["depth"].shape == (H, W, 1) and out["accumulation"].shape == (H, W, 1)
    assert float(out["rgb"].min()) >= 0.0 and float(out["rgb"].max()) <= 1.0 and m.training
    rb = cam.to(dev).generate_rays(0)
    n, chunk = len(rb), m.eval_num_rays_per_chunk
    assert n == 700416 and (n + chunk - 1) // chunk == 22
    flat = out["rgb"].reshape(-1, 3)
    m.eval()
    with torch.no_grad():
        for c in (0, 10, 21):
            sl = slice(c * chunk, min(n, (c + 1) * chunk))
            o = m.get_outputs(RayBundle(rb.origins[sl], rb.directions[sl], rb.camera_indices[sl]))
            assert torch.equal(o["rgb"], flat[sl]), c
            assert torch.equal(o["depth"], out["depth"].reshape(-1, 1)[sl])
    m.train()
    idx = torch.arange(0, n, n // 300, device=dev)
    P16 = _vision_P(m)
    ref = V.nerfacto_forward(rb.origins[idx].cpu(), rb.directions[idx].cpu(), rb.camera_indices[idx, 0].cpu(), P16, spec, training=False)
    assert float((flat[idx].cpu() - ref["rgb"]).abs().max()) <= 5e-3
    assert float((out["accumulation"].reshape(-1, 1)[idx].cpu() - ref["accumulation"]).abs().max()) <= 5e-3
    gt = torch.clip(out["rgb"] + 0.05, 0, 1)
    met, img = m.get_image_metrics_and_images(out, {"image": gt})
    np.testing.assert_allclose(met["psnr"], float(psnr(out["rgb"].cpu(), gt.cpu())), rtol=1e-5)
    assert 0.0 < met["ssim"] <= 1.0 and img["img"].shape == (H, 2 * W, 3)


# ---- V2 backward at full size: owner scatter vs oracle ---------------------------------------------------------------------------
def test_field_backward_full_size_owner_scatter_vs_oracle_on_masked_upstream(vm):
    """The fused field backward on the full render batch (4096 rays x 48 samples = 196,608 samples: the DEFAULT LDS-owner scatter
    path) with the upstream gradient non-zero on every 16th ray only: all seven parameter gradients must equal autograd through
    the oracle evaluated on those 256 rays alone."""
    from neraf_amd.vision import RayBundle
    m, spec, V, dev = vm
    R, S = 4096, 48
    rb = synth.ray_batch(R, tag="full.rays")
    bundle = RayBundle(T(rb["origins"]).to(dev), T(rb["directions"]).to(dev), T(rb["camera_indices"]).to(dev))
    m.train()
    m.update_to_step(300)
    out = m.get_outputs(bundle, jitters=[T(j).to(dev) for j in rb["jitters"]])
    st = out["_state"]
    e_bins = st["samples"][-1].e_bins
    assert e_bins.shape == (R, S + 1) and R * S >= 131072
    f = m.field.module
    sel = torch.arange(0, R, 16, device=dev)
    g = torch.Generator(device=dev).manual_seed(5)
    d_rgb = torch.zeros((R, S, 3), device=dev)
    d_den = torch.zeros((R, S), device=dev)
    d_rgb[sel] = (torch.rand((sel.numel(), S, 3), generator=g, device=dev) - 0.5) * 1e-2
    d_den[sel] = (torch.rand((sel.numel(), S), generator=g, device=dev) - 0.5) * 1e-4
    grads = f.backward_query(st["field_packed"], st["o"], st["d"], e_bins, st["cam"], st["dens"], d_rgb, d_den)
    # oracle on the selected rays
    P = {k: v.clone().requires_grad_(k.startswith("field.")) for k, v in _vision_P(m).items()}
    eb = e_bins[sel].cpu()
    mid = (eb[:, :-1] + eb[:, 1:]) / 2
    o_s, d_s = st["o"][sel].cpu(), st["d"][sel].cpu()
    pos = o_s[:, None, :] + d_s[:, None, :] * mid[..., None]
    cams = st["cam"].reshape(-1)[sel].cpu().long()[:, None].expand(-1, S)
    rgb_o, den_o = V.field_forward(pos, d_s[:, None, :].expand(-1, S, -1), cams, P, spec, training=True)
    # forward parity on the subsample first (density relative 2e-2, colours 4e-3: test_gpu_vision.py)
    np.testing.assert_allclose(st["dens"][sel].cpu().numpy(), den_o.detach().reshape(-1, S).numpy(), rtol=2e-2, atol=1e-7)
    assert float((st["rgb_s"][sel].cpu() - rgb_o.detach().reshape(-1, S, 3)).abs().max()) <= 4e-3
    ((rgb_o.reshape(-1, S, 3) * d_rgb[sel].cpu()).sum() + (den_o.reshape(-1, S) * d_den[sel].cpu()).sum()).backward()
    names = ["field.table", "field.base_w0", "field.base_w1", "field.head_w0", "field.head_w1", "field.head_w2", "field.embedding"]
    for name, gk in zip(names, grads):
        assert rel_l2(gk, P[name].grad) <= GRAD_REL_L2, (name, rel_l2(gk, P[name].grad))
    # rows of unselected cameras' embeddings and untouched table entries are exactly zero
    touched = torch.zeros(210, dtype=torch.bool)
    touched[st["cam"].reshape(-1)[sel].cpu().long()] = True
    assert float(grads[6][~touched.to(dev)].abs().max()) == 0.0


def test_trunc_exp_backward_clamps_at_plus_and_minus_15_like_the_oracle(vm):
    """Density logits driven far beyond +-15 (density row of the base MLP scaled to a logit spread of ~20): the gradient through avg_density * trunc_exp
    must use exp(clamp(logit, -15, 15)) [NS-recall], on both sides -- checked separately by routing the upstream density gradient to
    samples above +15 only, then to samples below -15 only (where an unclamped exp would give a gradient orders of magnitude smaller)."""
    m0, spec, V, dev = vm
    m = copy.deepcopy(m0)
    f = m.field.module
    R, S = 256, 48
    rb = synth.ray_batch(R, tag="clamp.rays")
    o, d, cam = T(rb["origins"]).to(dev), T(rb["directions"]).to(dev), T(rb["camera_indices"]).to(dev)
    g = torch.Generator(device=dev).manual_seed(9)
    e = torch.sort(torch.rand((R, S + 1), generator=g, device=dev) * 2.0 + 0.05, dim=1).values.contiguous()
    with torch.no_grad():                                   # scale the density row so that the logits have a standard deviation of ~20
        _, den0 = f.query(o, d, e, cam, packed=f.packed(with_average=False))
        lg = torch.log(den0[den0 > 0] / 0.01)
        f.base_w1[0] *= float(20.0 / lg.std())
    f.invalidate_packed()
    packed = f.packed(with_average=False)
    rgb, den = f.query(o, d, e, cam, packed=packed)
    P = {k: v.clone().requires_grad_(k.startswith("field.")) for k, v in _vision_P(m).items()}
    mid = (e[:, :-1] + e[:, 1:]).cpu() / 2
    pos = o.cpu()[:, None, :] + d.cpu()[:, None, :] * mid[..., None]
    cams = cam.cpu().long()[:, None].expand(-1, S)
    rgb_o, den_o = V.field_forward(pos, d.cpu()[:, None, :].expand(-1, S, -1), cams, P, spec, training=True)
    den_o = den_o.reshape(R, S)
    logit = torch.log(den_o.detach().double() / 0.01)
    hi, lo = logit > 16.0, (logit < -16.0) & (den_o.detach() > 0)
    assert int(hi.sum()) > 100 and int(lo.sum()) > 100, (int(hi.sum()), int(lo.sum()))
    for mask in (hi, lo):
        up = torch.where(mask, torch.rand((R, S)) + 0.5, torch.zeros(()))
        for p in P.values():
            p.grad = None
        (den_o * up).sum().backward(retain_graph=True)
        grads = f.backward_query(packed, o, d, e, cam, den, torch.zeros((R, S, 3), device=dev), up.to(dev).contiguous())
        for name, gk in (("field.table", grads[0]), ("field.base_w0", grads[1]), ("field.base_w1", grads[2])):
            r = rel_l2(gk, P[name].grad)
            assert r <= GRAD_REL_L2, (name, r, "upper" if mask is hi else "lower")
    # what the clamp does on the lower side: the true derivative is far smaller than the clamped one the reference back-propagates
    assert float(den_o.detach()[lo].max()) < 0.01 * np.exp(-16.0) * 1.0001


# ---- configs[2] / configs[3]: the joint step at full size ---------------------------------------------------------------------
@pytest.mark.parametrize("dataset,R,B,C_,F_,T_", [("raf", 4096, 2048, 1, 513, 60), ("soundspaces", 4096, 808, 2, 257, 101),
                                                   ("soundspaces", 32768, 6464, 2, 257, 101)])
def test_joint_step_full_size_vs_oracle(dataset, R, B, C_, F_, T_):
    """One NeRAFPipeline.get_train_loss_dict + backward at the benchmark's shapes (the bench's own JointStep object): configs[2]
    = 4096 rays + 2048 RAF slices, configs[3] per-GPU = 4096 rays + 808 SoundSpaces slices (2 x 257 head, T = 101), 128^3 grid, and
    configs[3] at its GLOBAL size in one process -- 32768 rays + 64 RIRs x T = 6464 slices (NeRAF_config.py:43-47, :87), the batch
    the reference's single process trains on: 12.6 M proposal + 1.57 M field samples, a 32768-cell refresh window, M = 6464 NAcF rows.
      * rendered colours of every 16th (global size: 128th) ray == oracle render of those rays with the same jitters; render
        invariants on ALL rays (colours and accumulation in [0, 1], finite positive depth);
      * the audio outputs of ALL slices == oracle NAcF on the feature the HIP ResNet3D produced; both audio losses == oracle;
      * NAcF parameter gradients (incl. layer 0's feature half and both heads) == oracle autograd on the full slice batch;
      * every parameter of both models receives a finite gradient; the rgb loss equals the mean over the HIP colours."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from neraf_amd.vision import NeRAFVisionModel
    from oracle import audio as O
    from oracle import vision as V
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    bench.C_, bench.F_, bench.T_ = C_, F_, T_
    try:
        js = bench.JointStep(dev, R, B, 1, dataset=dataset, rotate=1)
    finally:
        bench.C_, bench.F_, bench.T_ = 1, 513, 60
    vm_, am, pipe = js.vm, js.am, js.pipe
    rb = synth.ray_batch(R, tag="bench.rays.r0")
    jit = [T(j).to(dev) for j in rb["jitters"]]
    vm_.forward = lambda bundle: NeRAFVisionModel.get_outputs(vm_, bundle, jitters=jit)
    cap = {}
    orig_feat = am.scene_feature

    def rec():
        f = orig_feat()
        cap["feat"] = f.detach().clone()
        return f
    am.scene_feature = rec
    step = 20001
    vm_.update_to_step(step)
    for p in list(vm_.parameters()) + list(am.parameters()):
        p.grad = None
    outs, ld, _ = pipe.get_train_loss_dict(step)
    assert set(ld) >= {"rgb_loss", "interlevel_loss", "distortion_loss", "audio_sc_loss", "audio_mag_loss"}
    y_hip = None
    # the audio outputs are not returned by get_train_loss_dict: recompute them on the captured feature through the same field
    with torch.no_grad():
        b = js.batch
        y_hip = am.field.forward_queries(cap["feat"], b["time_query"], b["mic_pose"], b["source_pose"], b["rot"], am.aabb, am.max_len)
    sum(ld.values()).backward()
    torch.cuda.synchronize()
    # -- radiance: strided rays against the oracle
    spec = V.NerfactoSpec()
    P16 = _vision_P(vm_)
    sel = torch.arange(0, R, 16 * (R // 4096))
    ref = V.nerfacto_forward(T(rb["origins"])[sel], T(rb["directions"])[sel], T(rb["camera_indices"])[sel], P16, spec, step=step,
                             training=True, jitters=[T(j)[sel] for j in rb["jitters"]])
    assert float((outs["rgb"][sel.to(dev)].cpu() - ref["rgb"]).abs().max()) <= 5e-3
    np.testing.assert_allclose(float(ld["rgb_loss"]), float(((outs["rgb"] - js.gt["image"]) ** 2).mean()), rtol=1e-4)
    assert tuple(outs["rgb"].shape) == (R, 3) and bool(torch.isfinite(outs["rgb"]).all())
    assert float(outs["rgb"].min()) >= 0.0 and float(outs["rgb"].max()) <= 1.0
    acc = outs["accumulation"]
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-3
    assert bool(torch.isfinite(outs["depth"]).all()) and float(outs["depth"].min()) > 0.0
    # -- audio: all slices, oracle NAcF on the HIP feature
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in am.field.state_dict().items()}
    bc = {k: v.cpu() for k, v in js.batch.items()}
    feat = cap["feat"].cpu()
    yo = O.audio_get_outputs(bc, feat, sd, am.aabb.cpu(), T_)
    assert tuple(yo.shape) == (B, C_, F_)
    assert rel_l2(y_hip, yo) <= 3e-3 and float((y_hip.cpu() - yo.detach()).abs().max()) <= 0.05
    lo = O.audio_loss_dict(yo, bc["data"])
    for k in ("audio_sc_loss", "audio_mag_loss"):
        np.testing.assert_allclose(float(ld[k]), float(lo[k]), rtol=5e-3)
    (lo["audio_sc_loss"] + lo["audio_mag_loss"]).backward()
    hip = dict(am.field.named_parameters())
    for k in ["soundfield.0.weight", "soundfield.0.bias", "soundfield.1.weight", "soundfield.4.weight"] + \
             [f"STFT_linear.{c}.weight" for c in range(C_)] + [f"STFT_linear.{c}.bias" for c in range(C_)]:
        r = rel_l2(hip[k].grad, sd[k].grad)
        assert r <= 5e-2, (k, r)
    for name, p in list(vm_.named_parameters()) + list(am.named_parameters()):
        if name.startswith("proposal_networks") and not outs["_state"]["prop_updated"]:
            continue
        if "camera_optimizer" in name:
            continue
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name

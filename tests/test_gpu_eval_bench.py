"""The eval-render workload that ``bench.py --mode eval`` times (BASELINE configs[4]) produces the outputs of the parity-tested
paths: its frame is ``get_outputs_for_camera`` (held to the oracle's eval-mode render on pixels strided over the whole frame, as
tests/test_gpu_fullsize.py does for its own camera), its RIRs are the audio eval branch (held to the oracle's prologue + NAcF with
the HIP engine's scene feature), and the batched RIR call equals the per-RIR calls."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _vision_P(vm):
    """The model's parameters as the oracle's dict, rounded to fp16 like the copies the kernels read."""
    f = vm.field.module
    P = {"field.table": f.table, "field.base_w0": f.base_w0, "field.base_w1": f.base_w1, "field.head_w0": f.head_w0,
         "field.head_w1": f.head_w1, "field.head_w2": f.head_w2, "field.embedding": f.embedding}
    for i, pn in enumerate(vm.proposal_networks):
        P[f"prop{i}.table"], P[f"prop{i}.w0"], P[f"prop{i}.w1"] = pn.table, pn.w0, pn.w1
    return {k: v.detach().half().float().cpu() for k, v in P.items()}


@pytest.fixture(scope="module")
def er():
    import bench
    return bench.EvalRender(torch.device("cuda:0"), n_cams=2, n_items=8), bench


def test_eval_frame_is_the_chunked_camera_render_and_matches_the_oracle(er):
    from oracle import vision as V
    e, bench = er
    out = e.frame(1)
    H, W = 1024, 684
    assert (e.H, e.W) == (H, W) and e.rays_per_frame == 700416
    assert out["rgb"].shape == (H, W, 3) and out["depth"].shape == (H, W, 1)
    # the timed call IS the model's entry point: a second call on the same camera is bit-identical (no jitter in eval mode)
    again = e.vm.get_outputs_for_camera(e.cams[1], None, eval=True)
    assert torch.equal(out["rgb"], again["rgb"]) and torch.equal(out["accumulation"], again["accumulation"])
    # 300 pixels strided over the whole frame (every chunk) against the oracle's eval-mode render
    rb = e.cams[1].generate_rays(0)
    n = len(rb)
    idx = torch.arange(0, n, n // 300, device=rb.origins.device)
    spec = V.NerfactoSpec()
    ref = V.nerfacto_forward(rb.origins[idx].cpu(), rb.directions[idx].cpu(), rb.camera_indices[idx, 0].cpu(), _vision_P(e.vm), spec,
                             training=False)
    assert float((out["rgb"].reshape(-1, 3)[idx].cpu() - ref["rgb"]).abs().max()) <= 5e-3
    assert float((out["accumulation"].reshape(-1, 1)[idx].cpu() - ref["accumulation"]).abs().max()) <= 5e-3


def test_eval_rirs_match_the_oracle_and_the_batched_call_matches_the_per_rir_calls(er):
    from neraf_amd import synth
    from oracle import audio as O
    e, bench = er
    Tn, Cn, Fn = bench.T_, bench.C_, bench.F_
    feat = e.am.scene_feature().detach().cpu()
    sdn = {k: T(v) for k, v in synth.nacf_state_dict(1187, 512, Cn, Fn).items()}
    aabb = T(synth.audio_aabb())
    outs = []
    for k in range(3):
        o = e.rir(k)
        raw = o["raw_output"]
        assert raw.shape == (Tn, Cn, Fn)
        assert o["stft_ch_0"].shape == (Fn, Tn, 1) and o["gt_ch_0"].shape == (Fn, Tn, 1) and "grid" in o       # NeRAF_model.py:695-723
        it = e.items[k]
        one = {"time_query": torch.arange(Tn), "mic_pose": it["mic_pose"].cpu().expand(Tn, 3), "source_pose": it["source_pose"].cpu().expand(Tn, 3),
               "rot": it["rot"].cpu().expand(Tn, 3)}
        with torch.no_grad():
            ref = O.audio_get_outputs(one, feat, sdn, aabb, Tn)
        rel = float((raw.cpu() - ref).norm() / ref.norm())
        assert rel <= 3e-3, rel                                   # tests/test_gpu_nacf.py's forward tolerance
        # the panel is the flipped transposed slice of the raw output (:697-699)
        assert torch.equal(o["stft_ch_0"], torch.flip(raw[:, 0, :].transpose(0, 1).unsqueeze(-1).cpu(), [0]))
        outs.append(raw)
    b = e.batched_rirs(3)
    assert b.shape == (3, Tn, Cn, Fn)
    for k in range(3):
        rel = float((b[k] - outs[k]).norm() / outs[k].norm())
        assert rel <= 1e-3, rel                                   # same arithmetic, other GEMM tiles (fp32 accumulation order)


def test_measure_eval_reports_the_reference_keys(er):
    from neraf_amd import _lib
    e, bench = er
    lib, h = _lib.load(), _lib.ctx(0)
    m, fams = bench.measure_eval(e, 1, 1, 2, lib, h, 0, torch.cuda.synchronize, full=True)
    for k in ("rays_per_s", "fps", "bins_per_s", "fps_audio", "value", "ms_per_step", "batched_rirs"):
        assert k in m and (m[k] if not isinstance(m[k], dict) else True)
    assert m["chunks_per_frame"] == 22 and m["rays_per_frame"] == 700416 and m["bins_per_rir"] == bench.T_ * bench.C_ * bench.F_
    names = [f["kernel"] for f in fams]
    assert any(n.startswith("field_query_kernel") for n in names) and any(n.startswith("proposal_density_kernel") for n in names)
    fq = next(f for f in fams if f["kernel"].startswith("field_query_kernel"))
    # 22 launches per frame; algorithmic bytes = samples x 512 B
    assert abs(fq["launches_per_step"] - 22) < 1e-9
    np.testing.assert_allclose(fq["work_per_step"], 700416 * 48 * 512, rtol=1e-12)
    # every family is priced against the resource that bounds it: the proposal tables are L2-resident (bound "l2"), the main table is
    # not (bound "hbm"); no fraction above 1; the eval roofline's headline is the HBM-priced field query
    pd = next(f for f in fams if f["kernel"].startswith("proposal_density_kernel"))
    assert pd["bound"] == "l2" and pd["peak"] == bench.L2_PEAK_GBS and fq["bound"] == "hbm"
    assert all(0.0 < f["frac"] <= 1.0 for f in fams), [(f["kernel"], f["frac"]) for f in fams]
    rf = bench.eval_roofline(fams)
    assert rf["bound"] == "hbm" and rf["kernel"].startswith("field_query_kernel") and 0.0 < rf["frac"] <= 1.0


def test_camera_rays_kernel_equals_the_tensor_expression():
    """Cameras.generate_rays on the device is one HIP launch (csrc/camera.hip); the tensor expression of neraf_amd/cameras.py (the CPU
    form, property-tested in tests/test_cameras.py) is its reference: whole frame of one distorted RAF camera, and a pixel-sampler
    call (random cameras, fractional coordinates)."""
    from neraf_amd.datamanagers import synthetic_cameras
    cams = synthetic_cameras(3, tag="raygen")
    dev = torch.device("cuda:0")
    cg = cams.to(dev)
    ref = cams[2].generate_rays(0)                        # CPU tensors -> tensor expression
    out = cg[2].generate_rays(0)
    assert out.origins.shape == (1024 * 684, 3) and out.camera_indices.shape == (1024 * 684, 1) and out.camera_indices.dtype == torch.int64
    assert torch.equal(out.origins.cpu(), ref.origins)
    assert float((out.directions.cpu() - ref.directions).abs().max()) <= 2e-6
    np.testing.assert_allclose(out.directions.norm(dim=-1).cpu().numpy(), 1.0, atol=1e-6)
    assert int(out.camera_indices.max()) == 0
    g = torch.Generator().manual_seed(3)
    ci = torch.randint(0, 3, (5000,), generator=g)
    co = torch.rand((5000, 2), generator=g) * torch.tensor([1024.0, 684.0])
    ref = cams.generate_rays(ci, co)
    out = cg.generate_rays(ci.to(dev), co.to(dev))
    assert torch.equal(out.camera_indices.cpu(), ref.camera_indices)
    assert torch.equal(out.origins.cpu(), ref.origins)
    assert float((out.directions.cpu() - ref.directions).abs().max()) <= 2e-6

"""The training-loop oracle (oracle/trainer.py) and its committed fixture (tests/golden/g7_trajectory.npz): the fixture must be
reproducible from the committed code -- the first iterations of a fresh oracle run give the fixture's loss curves -- and the
trainer-level pieces it restates (scheduler, proposal-update schedule, BatchNorm running statistics, optimizer grouping) behave as
the reference's configuration says (NeRAF_config.py:115-132, NeRAF_pipeline.py:186, :487)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))


def test_exponential_decay_schedule_matches_the_product_config():
    from neraf_amd import config as C
    from oracle.trainer import reference_schedules
    cfg = C.default_optimizers(2000)
    fn = reference_schedules(2000)
    lr0 = {"proposal_networks": 1e-2, "fields": 1e-2, "audio_fields": 1e-4, "camera_opt": 1e-3}
    for name, f in fn.items():
        for k in (0, 1, 7, 1999, 2000, 2001, 5000, 199999, 400000):
            np.testing.assert_allclose(f(k), cfg[name]["scheduler"].lr_at(k, lr0[name]), rtol=1e-12, err_msg=f"{name} {k}")
    assert abs(fn["audio_fields"](0) - 1e-8) < 1e-12 and abs(fn["audio_fields"](2000) - 1e-4) < 1e-12      # warm-up ends at audio start


def test_bn_running_update_context_matches_torch_batchnorm3d():
    from oracle import audio as O
    g = torch.Generator().manual_seed(0)
    bn = torch.nn.BatchNorm3d(5)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g); bn.bias.uniform_(-0.5, 0.5, generator=g)
    sd = {"bn." + k: v.detach().clone() for k, v in bn.state_dict().items()}
    x = torch.randn((1, 5, 4, 4, 4), generator=g)
    bn.train()
    want = bn(x)
    y0 = O._bn(x, sd, "bn", True)                               # outside the context: running statistics untouched
    assert torch.equal(sd["bn.running_mean"], torch.zeros(5))
    with O.bn_running_update(0.1):
        y1 = O._bn(x, sd, "bn", True)
    torch.testing.assert_close(y0, want); torch.testing.assert_close(y1, want)
    torch.testing.assert_close(sd["bn.running_mean"], bn.running_mean)
    torch.testing.assert_close(sd["bn.running_var"], bn.running_var)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("scenario", ["g7_trajectory", "g8_trajectory_pose"])
def test_fixture_head_is_reproducible_from_the_committed_oracle(golden, scenario):
    """Four iterations of a fresh OracleTrainer on the scenario (two before the audio branch ... two with it would need iteration 7;
    the scenario starts audio at 6, so iterations 0-3 cover the radiance half, the refresh and the optimizer grouping) must give the
    fixture's loss-dict values; chaos needs ~100 iterations to show (trajectory_common.py), 4 agree to fp32 summation-order noise."""
    import trajectory_common as TC
    from oracle.trainer import OracleTrainer
    g = golden(scenario)
    cfg = TC.SCENARIOS[scenario]
    assert int(g["steps"]) == cfg["steps"] and int(g["start_step_audio"]) == cfg["start_step_audio"]
    assert int(g["camera_opt"]) == int(bool(cfg["camera_opt"]))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    P, sdn, sdr = TC.initial_weights()
    tr = OracleTrainer(P, sdn, sdr, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), TC.T(TC.synth.audio_aabb()), cfg["grid_step"], cfg["T"],
                       cfg["start_step_audio"], cfg["R"], num_cameras=cfg["n_cam"] if cfg.get("camera_opt") else 0)
    bank = TC.rir_bank(cfg["n_rir"], cfg["tag"] + ".train")
    keys = [str(k) for k in g["keys"]]
    n_it = 4 if scenario == "g7_trajectory" else 2            # the pose scenario shares everything but the pose deltas: two iterations
    for s in range(n_it):
        r = tr.train_iteration(s, TC.ray_batch(s), TC.audio_batch(s, bank))
        for j, k in enumerate(keys[:3]):
            np.testing.assert_allclose(r[k], g["curves"][s, j], rtol=2e-3, atol=1e-9, err_msg=f"iteration {s} {k}")
        assert r["proposal_updated"] == g["curves"][s, keys.index("proposal_updated")]
    # optimizer grouping: the field parameters are in "fields" and in "audio_fields" (NeRAF_pipeline.py:487); the NAcF / ResNet3D
    # members of "audio_fields" have not been stepped yet (their gradient is None before start_step_audio): torch's per-parameter step
    f = tr.P["field.table"]
    assert float(tr.opt["fields"].state[f]["step"]) == n_it and float(tr.opt["audio_fields"].state[f]["step"]) == n_it
    if cfg["camera_opt"]:
        assert set(tr.opt) == {"proposal_networks", "fields", "audio_fields", "camera_opt"} and float(tr.pose.abs().max()) > 0.0
    w = tr.sdn["soundfield.0.weight"]
    assert w not in tr.opt["audio_fields"].state or not tr.opt["audio_fields"].state[w]


@pytest.mark.timeout(600)
def test_precision_probes_perturb_one_rounding_source_each_and_leave_the_fp32_oracle_alone():
    """OracleTrainer(probe=...) -- the yardstick runs of fixture G9: one joint iteration (audio branch on) of the fp32 oracle and of its
    three probes from the same state.  Every probe moves the losses (it does something) by a rounding-sized amount (it does nothing
    else), the probes' hooks are gone afterwards (the pinned arithmetic of oracle/audio.py and oracle/vision.py is untouched: a second
    fp32 iteration reproduces the first bit for bit), and "resnet_grad_bf16" changes no FORWARD quantity at all."""
    import trajectory_common as TC
    from oracle import audio as O, vision as V
    from oracle.trainer import OracleTrainer
    cfg = TC.SCENARIOS["g9_long"]
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    bank = TC.rir_bank(cfg["n_rir"], cfg["tag"] + ".train")
    step = cfg["start_step_audio"] + 1

    def one(probe):
        P, sdn, sdr = TC.initial_weights()
        tr = OracleTrainer(P, sdn, sdr, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), TC.T(TC.synth.audio_aabb()), cfg["grid_step"], cfg["T"],
                           cfg["start_step_audio"], cfg["R"], probe=probe)
        r = tr.train_iteration(step, TC.ray_batch(step), TC.audio_batch(step, bank))
        # Adam's first moment after one step = 0.1 x the gradient: the observable of a backward-only probe (the first update itself is
        # lr x sign(g) whatever the magnitudes)
        return r, tr.opt["audio_fields"].state[tr.sdr["conv1.weight"]]["exp_avg"].detach().clone(), tr.P["field.table"].detach().clone()

    base, w0, t0 = one(None)
    again, w0b, t0b = one(None)
    same = lambda a, b: all(abs(a[k] - b[k]) <= 1e-5 * abs(b[k]) + 1e-12 for k in b)      # noqa: E731
    assert same(again, base)
    rel = lambda a, b: float((a - b).norm() / b.norm())      # noqa: E731
    assert rel(w0b, w0) <= 1e-4
    torch.testing.assert_close(t0b, t0, rtol=1e-3, atol=2e-6)
    assert V.PROBE_ACT is None and O.PROBE["act"] is None and O.PROBE["grad"] is None
    for probe in ("params16", "acts16", "resnet_grad_bf16"):
        r, w, t = one(probe)
        assert V.PROBE_ACT is None and O.PROBE["act"] is None and O.PROBE["grad"] is None, probe
        if probe == "resnet_grad_bf16":
            assert same(r, base), probe                                           # a backward-only perturbation: same forward, same losses
        else:
            assert any(r[k] != base[k] for k in ("rgb_loss", "audio_mag_loss")), probe
            for k in ("rgb_loss", "audio_mag_loss", "audio_sc_loss"):
                assert abs(r[k] - base[k]) <= 2e-2 * abs(base[k]), (probe, k, r[k], base[k])
        # the gradient of the encoder's first convolution: close, and different where the probe touches the encoder's own arithmetic
        # (bf16 gradients: 8 significant bits through 43 layers; fp16 storage points can flip ReLU gates: larger)
        dw = rel(w, w0)
        assert dw <= 0.8 and (probe == "params16" or dw > 1e-3), (probe, dw)

"""FusedAdam (one HIP launch for all tensors) against torch.optim.Adam on the same parameters and gradients.

Tolerance: both compute in fp32 with the same formula; the only difference is the order of fused multiply-adds, so parameters
must agree to 1e-6 relative / 1e-7 absolute after several steps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(5096, 163), (17,), (3, 5, 7), (1,), (4096 * 3 + 5,), (64, 64, 3, 3, 3)]
    return [torch.randn(s, generator=g).to(dev).requires_grad_(True) for s in shapes]


def test_fused_adam_matches_torch_adam_two_groups():
    from neraf_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    pa, pb = _params(dev, 0), _params(dev, 0)
    oa = FusedAdam([{"params": pa[:3], "lr": 1e-2}, {"params": pa[3:], "lr": 1e-4}], eps=1e-15)
    ob = torch.optim.Adam([{"params": pb[:3], "lr": 1e-2}, {"params": pb[3:], "lr": 1e-4}], eps=1e-15)
    g = torch.Generator().manual_seed(1)
    for it in range(6):
        for x, y in zip(pa, pb):
            gr = torch.randn(x.shape, generator=g).to(dev) * 10.0 ** float(torch.randint(-6, 2, (1,), generator=g))
            x.grad, y.grad = gr.clone(), gr.clone()
        if it == 3:                                    # schedulers change the learning rate every step
            oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 5e-3
        oa.step(); ob.step()
    for x, y in zip(pa, pb):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=1e-6, atol=1e-7)
    st = oa.state[pa[0]]
    assert float(st["step"]) == 6.0
    np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), ob.state[pb[0]]["exp_avg_sq"].cpu().numpy(), rtol=1e-6, atol=1e-12)


def test_fused_adam_under_grad_scaler_skips_on_inf_and_unscales():
    from neraf_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    pa, pb = _params(dev, 2), _params(dev, 2)
    oa = FusedAdam(pa, lr=1e-3, eps=1e-15)
    ob = torch.optim.Adam(pb, lr=1e-3, eps=1e-15)
    from neraf_amd.optim import GradScaler
    sa = GradScaler("cuda", init_scale=1024.0)              # one-launch non-finite check over the optimizer's tensor table
    sb = torch.amp.GradScaler("cuda", init_scale=1024.0)
    g = torch.Generator().manual_seed(3)
    for it in range(5):
        oa.zero_grad(set_to_none=True); ob.zero_grad(set_to_none=True)
        w = [torch.randn(x.shape, generator=g).to(dev) for x in pa]
        la = sum((x * wi).sum() for x, wi in zip(pa, w))
        lb = sum((y * wi).sum() for y, wi in zip(pb, w))
        sa.scale(la).backward(); sb.scale(lb).backward()
        if it == 2:                                    # a non-finite gradient: both must skip the step and halve the scale
            pa[1].grad[0] = float("inf"); pb[1].grad[0] = float("inf")
        sa.step(oa); sb.step(ob)
        sa.update(); sb.update()
        assert sa.get_scale() == sb.get_scale()
    for x, y in zip(pa, pb):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=1e-6, atol=1e-7)
    assert float(oa.state[pa[0]]["step"]) == 4.0


def test_grad_scaler_update_two_optimizers_growth_and_backoff():
    """neraf_amd.optim.GradScaler.step / .update (flags handed over as they are, scale update in one launch over both optimizers'
    flags) against torch.amp.GradScaler driving torch.optim.Adam: growth every 3 clean steps, back-off when EITHER optimizer saw a
    non-finite gradient, explicit unscale_() before step, parameters equal at the end."""
    from neraf_amd.optim import FusedAdam, GradScaler
    dev = torch.device("cuda:0")
    pa, pb = _params(dev, 5), _params(dev, 5)
    oa = [FusedAdam(pa[:3], lr=1e-3, eps=1e-15), FusedAdam(pa[3:], lr=2e-3, eps=1e-8)]
    ob = [torch.optim.Adam(pb[:3], lr=1e-3, eps=1e-15), torch.optim.Adam(pb[3:], lr=2e-3, eps=1e-8)]
    kw = dict(init_scale=256.0, growth_interval=3, growth_factor=2.0, backoff_factor=0.5)
    sa, sb = GradScaler("cuda", **kw), torch.amp.GradScaler("cuda", **kw)
    g = torch.Generator().manual_seed(7)
    scales = []
    for it in range(11):
        for o in oa + ob:
            o.zero_grad(set_to_none=True)
        w = [torch.randn(x.shape, generator=g).to(dev) for x in pa]
        sa.scale(sum((x * wi).sum() for x, wi in zip(pa, w))).backward()
        sb.scale(sum((y * wi).sum() for y, wi in zip(pb, w))).backward()
        if it == 4:                                     # second optimizer only
            pa[4].grad[7] = float("nan"); pb[4].grad[7] = float("nan")
        if it == 8:                                     # first optimizer only
            pa[0].grad[3, 3] = float("-inf"); pb[0].grad[3, 3] = float("-inf")
        if it == 6:                                     # explicit unscale_ (e.g. before gradient clipping)
            sa.unscale_(oa[0]); sb.unscale_(ob[0])
            np.testing.assert_allclose(pa[0].grad.cpu().numpy(), pb[0].grad.cpu().numpy(), rtol=1e-6)
        for x, y in zip(oa, ob):
            sa.step(x); sb.step(y)
        sa.update(); sb.update()
        assert sa.get_scale() == sb.get_scale(), (it, sa.get_scale(), sb.get_scale())
        assert int(sa._growth_tracker) == int(sb._growth_tracker)
        scales.append(sa.get_scale())
    assert max(scales) > 256.0 and min(scales) < max(scales)          # it both grew and backed off
    for o in oa + ob:
        o.zero_grad(set_to_none=True)
    sa.scale(pa[0].sum()).backward(); sb.scale(pb[0].sum()).backward()
    sa.step(oa[0]); sb.step(ob[0])
    with pytest.raises(RuntimeError):                   # a second step() before update(), as the inherited class
        sa.step(oa[0])
    for x, y in zip(pa, pb):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=1e-6, atol=1e-7)


def test_fused_adam_late_first_gradient_starts_bias_correction_at_one():
    """Advisor finding (round 2): the reference's "audio_fields" group holds the radiance field (gradients from iteration 0) AND the
    NAcF / ResNet3D (gradient None until start_step_audio, NeRAF_pipeline.py:186, :487).  torch.optim.Adam keeps `step` per
    parameter, so the late tensors start at t = 1; a per-group counter would start them at t = N + 1 (bc2 ~ 0.87 instead of 0.001 for
    N = 2000: first updates 3-6x too large).  Part of ONE group receives its first gradient 7 steps late; also across a
    state_dict round trip."""
    from neraf_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    pa, pb = _params(dev, 4), _params(dev, 4)
    oa = FusedAdam([{"params": pa, "lr": 1e-3}], eps=1e-15)
    ob = torch.optim.Adam([{"params": pb, "lr": 1e-3}], eps=1e-15)
    g = torch.Generator().manual_seed(5)
    late = {2, 3, 5}

    def run(oa, pa, n0, n1):
        for it in range(n0, n1):
            for i, (x, y) in enumerate(zip(pa, pb)):
                if i in late and it < 7:
                    x.grad = y.grad = None
                    continue
                gr = torch.randn(x.shape, generator=g).to(dev)
                x.grad, y.grad = gr.clone(), gr.clone()
            oa.step(); ob.step()
    run(oa, pa, 0, 9)
    for i, (x, y) in enumerate(zip(pa, pb)):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=2e-6, atol=1e-7, err_msg=str(i))
        assert float(oa.state[x]["step"]) == float(ob.state[y]["step"]) == (2.0 if i in late else 9.0)
    # save / load / continue: per-parameter steps survive
    sd = oa.state_dict()
    pc = [x.detach().clone().requires_grad_(True) for x in pa]
    oc = FusedAdam([{"params": pc, "lr": 1e-3}], eps=1e-15)
    oc.load_state_dict(sd)
    run(oc, pc, 9, 12)
    for i, (x, y) in enumerate(zip(pc, pb)):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=2e-6, atol=1e-7, err_msg=str(i))
        assert float(oc.state[x]["step"]) == float(ob.state[y]["step"])


def test_fused_double_update_of_shared_tensors_is_bit_identical_to_two_steps():
    """The radiance-field parameters belong to "fields" AND "audio_fields" (NeRAF_pipeline.py:487): the reference steps them twice per
    iteration.  ``first.fuse_shared_updates_into(second)`` applies both updates in the second optimizer's launch; parameters and all
    four moment tensors must equal the two-launch form BIT FOR BIT over several iterations, including an iteration where only the
    first optimizer's gradients are non-finite (its update is skipped, the second's is applied) and one where only the second's are;
    and both must agree with two torch.optim.Adam instances to fp32 rounding."""
    from neraf_amd.optim import FusedAdam, GradScaler
    dev = torch.device("cuda:0")

    def build(linked):
        ps = _params(dev, 7)
        shared, only0, only1 = ps[:3], ps[3:4], ps[4:]
        o0 = FusedAdam([{"params": only0, "lr": 1e-2}, {"params": shared, "lr": 5e-3}], eps=1e-15)
        o1 = FusedAdam([{"params": only1 + shared, "lr": 1e-4}], eps=1e-15)
        if linked:
            o0.fuse_shared_updates_into(o1)
        sc = GradScaler("cuda", init_scale=256.0)
        sc.scale(torch.zeros(1, device=dev))                 # initialises the device-side scale (the gradients below are pre-scaled)
        return ps, o0, o1, sc
    (pa, a0, a1, sa), (pb, b0, b1, sb) = build(True), build(False)
    pt = _params(dev, 7)
    t0 = torch.optim.Adam([{"params": pt[3:4], "lr": 1e-2}, {"params": pt[:3], "lr": 5e-3}], eps=1e-15)
    t1 = torch.optim.Adam([{"params": pt[4:] + pt[:3], "lr": 1e-4}], eps=1e-15)
    g = torch.Generator().manual_seed(11)
    for it in range(7):
        grads = [torch.randn(x.shape, generator=g).to(dev) * 256.0 for x in pa]
        if it == 2:
            grads[3][0] = float("inf")          # only the FIRST optimizer sees a non-finite gradient
        if it == 4:
            grads[5].view(-1)[3] = float("nan")  # only the SECOND
        for ps in (pa, pb):
            for x, gr in zip(ps, grads):
                x.grad = gr.clone()
        if it == 5:                              # learning rates move (schedulers)
            for o in (a0, b0):
                o.param_groups[1]["lr"] = 2e-3
            t0.param_groups[1]["lr"] = 2e-3
        for sc, o0, o1 in ((sa, a0, a1), (sb, b0, b1)):
            sc.step(o0); sc.step(o1); sc.update()
        if it not in (2, 4):
            for x, gr in zip(pt, grads):
                x.grad = gr / float(256.0 if it < 2 else (128.0 if it < 4 else 64.0))
            t0.step(); t1.step()
        elif it == 2:                            # first skipped, second applied
            for x, gr in zip(pt, grads):
                x.grad = gr / 256.0
            t1.step()
        else:
            for x, gr in zip(pt, grads):
                x.grad = gr / 128.0
            t0.step()
    assert sa.get_scale() == sb.get_scale() == 64.0
    for i, (x, y) in enumerate(zip(pa, pb)):
        assert torch.equal(x.detach(), y.detach()), i
    for oa, ob, ps_a, ps_b in ((a0, b0, pa, pb), (a1, b1, pa, pb)):
        for x, y in zip(ps_a, ps_b):
            if x in oa.state:
                assert torch.equal(oa.state[x]["exp_avg"], ob.state[y]["exp_avg"]) and torch.equal(oa.state[x]["exp_avg_sq"], ob.state[y]["exp_avg_sq"])
                assert float(oa.state[x]["step"]) == float(ob.state[y]["step"])
    for i, (x, y) in enumerate(zip(pa, pt)):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=3e-6, atol=2e-7, err_msg=str(i))
    # protocol: stepping the first optimizer twice without the second is an error, not a silently lost update
    for x in pa:
        x.grad = torch.ones_like(x)
    a0.step()
    with pytest.raises(RuntimeError, match="fused into"):
        a0.step()


def test_deferred_update_is_flushed_not_lost_when_the_second_optimizer_never_steps():
    """after ``first.fuse_shared_updates_into(second)`` the first optimizer's ``step`` defers the shared tensors' update to
    the second's launch.  If the second is not stepped that iteration (an exception, a caller stepping "fields" alone), the update must
    not be dropped: ``zero_grad`` (and ``state_dict`` / a state load) applies it with a plain launch over those tensors -- equal, bit
    for bit, to what an un-linked optimizer does.  And chains are refused: an optimizer is part of ONE pair."""
    from neraf_amd.optim import FusedAdam
    dev = torch.device("cuda:0")

    def build(linked):
        ps = _params(dev, 6)
        shared, only0, only1 = ps[:2], ps[2:4], ps[4:]
        o0 = FusedAdam([{"params": only0, "lr": 1e-2}, {"params": shared, "lr": 5e-3}], eps=1e-15)
        o1 = FusedAdam([{"params": only1 + shared, "lr": 1e-4}], eps=1e-15)
        if linked:
            o0.fuse_shared_updates_into(o1)
        return ps, o0, o1
    (pa, a0, a1), (pb, b0, b1) = build(True), build(False)
    g = torch.Generator().manual_seed(5)
    for it in range(3):
        grads = [torch.randn(x.shape, generator=g).to(dev) for x in pa]
        for ps in (pa, pb):
            for x, gr in zip(ps, grads):
                x.grad = gr.clone()
        a0.step(); b0.step()
        if it != 1:                        # iteration 1: the second optimizer never steps
            a1.step(); b1.step()
        if it == 1:
            assert not torch.equal(pa[0], pb[0])          # the shared update is still pending on the linked side ...
        a0.zero_grad(); a1.zero_grad(); b0.zero_grad(); b1.zero_grad()
        for x, y in zip(pa, pb):                           # ... and applied by zero_grad: both sides agree again, bit for bit
            assert torch.equal(x, y), it
    for o, q in ((a0, b0), (a1, b1)):
        for (px, sx), (py, sy) in zip(o.state.items(), q.state.items()):
            assert torch.equal(sx["exp_avg"], sy["exp_avg"]) and torch.equal(sx["exp_avg_sq"], sy["exp_avg_sq"])
            assert float(sx["step"]) == float(sy["step"])
    # chains are refused
    ps = _params(dev, 3)
    c0, c1, c2 = (FusedAdam([{"params": ps, "lr": 1e-3}], eps=1e-15) for _ in range(3))
    c0.fuse_shared_updates_into(c1)
    with pytest.raises(ValueError):
        c1.fuse_shared_updates_into(c2)

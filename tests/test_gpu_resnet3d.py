"""GPU parity of the HIP ResNet3D scene encoder (csrc/resnet3d.hip: implicit-GEMM MFMA convolutions + fused
BatchNorm statistics) against the golden vectors produced by the reference's own ResNet3D_helper (G1) and the
pinned oracle.

Tolerance: fp16 operands / fp16 stored activations with fp32 accumulation and fp32 BN statistics through 43
conv+BN layers; against the reference's fp32 outputs we require the 1024-d feature within relative L2 1e-2 and
every stage statistic (mean, abs-mean, rms) within 1e-2 relative.
"""
import numpy as np
import pytest
import torch

from neraf_amd import synth

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _model(dev, grid_step):
    from neraf_amd.resnet3d import ResNet3D_helper
    net = ResNet3D_helper(in_channels=7, backbone="resnet50", pretrained=False, grid_step=grid_step, N_features=1024)
    sd = {k: T(v) for k, v in synth.resnet3d_state_dict(7).items()}
    net.backbone_net.load_state_dict(sd, strict=True)      # identical keys to the reference module
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.momentum = 0.0                                # keep running stats fixed between the two passes (as in gen_golden)
    return net.to(dev)


def rel_l2(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("S,tag", [(64, "g1_resnet3d_64"), (128, "g1_resnet3d_128")])
def test_resnet3d_forward_vs_reference_golden(golden, S, tag):
    dev = torch.device("cuda:0")
    g = golden(tag)
    net = _model(dev, 1 / S)
    x = T(synth.uniform(f"g1.grid{S}", (1, 7, S, S, S), 0.0, 1.0)).to(dev)
    net.train()
    y = net(x)
    assert y.shape == (1, 1024, 1, 1, 1)
    assert rel_l2(y, T(g["out_train"])) <= 1e-2
    net.eval()
    with torch.no_grad():
        ye = net(x)
    assert rel_l2(ye, T(g["out_eval"])) <= 1e-2


def test_resnet3d_running_stats_update():
    """train-mode forward updates running_mean/var like nn.BatchNorm3d (momentum 0.1, unbiased variance)."""
    from oracle import audio as O
    dev = torch.device("cuda:0")
    net = _model(dev, 1 / 64)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.momentum = 0.1
    x = T(synth.uniform("g1.grid64", (1, 7, 64, 64, 64), 0.0, 1.0))
    bn1 = net.backbone_net.bn1
    rm0, rv0 = bn1.running_mean.clone().cpu(), bn1.running_var.clone().cpu()
    net.train()
    net(x.to(dev))
    sd = {k: T(v) for k, v in synth.resnet3d_state_dict(7).items()}
    c1 = torch.nn.functional.conv3d(x, sd["conv1.weight"], stride=2, padding=2)
    mean, var = c1.mean((0, 2, 3, 4)), c1.var((0, 2, 3, 4), unbiased=True)
    np.testing.assert_allclose(bn1.running_mean.cpu().numpy(), (0.9 * rm0 + 0.1 * mean).numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(bn1.running_var.cpu().numpy(), (0.9 * rv0 + 0.1 * var).numpy(), rtol=2e-3, atol=2e-4)
    assert int(bn1.num_batches_tracked) == 1


@pytest.mark.parametrize("cin,cin_real,cout,k,stride,pad,din", [
    (64, 64, 256, 1, 1, 0, 16),      # 1x1x1 (plain GEMM dgrad / wgrad)
    (64, 64, 64, 3, 1, 1, 16),       # 3x3x3 stride 1 (64-wide tile)
    (128, 128, 128, 3, 2, 1, 16),    # 3x3x3 stride 2: parity-filtered transposed-conv loader
    (256, 256, 512, 1, 2, 0, 16),    # strided 1x1x1 downsample
    (8, 7, 64, 5, 2, 2, 32),         # the 7->64 stem (wgrad only; its input gradient is covered by the grid test)
])
def test_conv_bn_relu_stage_backward(cin, cin_real, cout, k, stride, pad, din):
    """One conv -> BatchNorm(train) -> ReLU stage on random data against torch autograd (fp32 CPU).  This is the tight check
    of the backward kernels (BN backward with the ReLU gate, im2col^T + split-K wgrad, transposed-conv dgrad): a single
    stage is well conditioned, unlike the 43-layer random-weight network (see the next test).  Tolerance 3e-2 relative L2
    (bf16 gradient tensors, fp16 activations, ReLU gates from the fp16 forward)."""
    import ctypes as C
    from neraf_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7 + cin + cout + k)
    dout = (din + 2 * pad - k) // stride + 1
    x = torch.from_numpy(rng.normal(size=(din ** 3, cin)).astype(np.float32))
    x[:, cin_real:] = 0
    x = x.half().float()
    w = torch.from_numpy((rng.normal(size=(cout, cin_real, k, k, k)) / np.sqrt(cin_real * k ** 3)).astype(np.float32)).half().float()
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    beta = torch.from_numpy(rng.uniform(-0.3, 0.3, cout).astype(np.float32))
    g = torch.from_numpy(rng.normal(size=(dout ** 3, cout)).astype(np.float32))
    # reference
    xr = x[:, :cin_real].reshape(1, din, din, din, cin_real).permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    wr, gr, br = w.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    c = torch.nn.functional.conv3d(xr, wr, stride=stride, padding=pad)
    y = torch.relu(torch.nn.functional.batch_norm(c, None, None, gr, br, training=True, eps=1e-5))
    gy = g.reshape(1, dout, dout, dout, cout).permute(0, 4, 1, 2, 3)
    (y * gy).sum().backward()
    # HIP
    xd, wd, gd_, bd, gg = x.half().to(dev).contiguous(), w.to(dev).contiguous(), gamma.to(dev), beta.to(dev), g.to(dev).contiguous()
    yk = torch.empty((dout ** 3, cout), dtype=torch.float16, device=dev)
    dx = torch.zeros((din ** 3, cin), dtype=torch.float32, device=dev)
    dw = torch.empty_like(wd)
    dgam, dbet = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
    _lib.check(lib.neraf_debug_conv_bn_relu_stage(_lib.ctx(0), cin, cin_real, cout, k, stride, pad, din, xd.data_ptr(), wd.data_ptr(),
                                                  gd_.data_ptr(), bd.data_ptr(), gg.data_ptr(), yk.data_ptr(), dx.data_ptr(),
                                                  dw.data_ptr(), dgam.data_ptr(), dbet.data_ptr(),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    yref = y.detach()[0].permute(1, 2, 3, 0).reshape(-1, cout)
    assert rel_l2(yk.float(), yref) <= 3e-3
    tol = 3e-2
    assert rel_l2(dw, wr.grad) <= tol
    assert rel_l2(dgam, gr.grad) <= tol and rel_l2(dbet, br.grad) <= tol
    if cin % 64 == 0:
        dxr = xr.grad[0].permute(1, 2, 3, 0).reshape(-1, cin_real)
        assert rel_l2(dx, dxr) <= tol


def test_resnet3d_backward_full_chain(golden):
    """Whole encoder backward (G1, 64^3 grid).  The randomly initialised 43-layer BatchNorm network is chaotic under fp16
    rounding: the REFERENCE module under its own fp16 autocast (NeRAF_config.py:79) deviates from its fp32 gradients by
    0.41-0.71 relative L2 (tests/tools/amp_sensitivity_probe.py), because the pooled loss makes the last BatchNorm's input
    gradient a pure function of the ReLU gates and ~5% of the gates flip.  So the full chain is held to
      (a) the same gradient NORMS as the fp32 reference (within 10%), i.e. no lost / duplicated / mis-scaled branch,
      (b) a deviation from the reference's fp32 gradients no larger than the reference's own fp16-autocast deviation (<= 0.75),
      (c) exact structural facts (padding channel of the stem gets no gradient),
    while the kernels themselves are verified tightly stage by stage (test above)."""
    dev = torch.device("cuda:0")
    g = golden("g1_resnet3d_64")
    S = 64
    net = _model(dev, 1 / S)
    net.train()
    bb = net.backbone_net
    x = T(synth.uniform(f"g1.grid{S}", (1, 7, S, S, S), 0.0, 1.0)).to(dev)
    wsum = T(synth.uniform("g1.outw", (1024,), -1.0, 1.0)).to(dev)
    got = {}
    bb.grid_window = (0, S ** 3, 7)
    bb.grid_grad_sink = lambda d: got.__setitem__("dx", d.clone())
    y = net(x)
    (y.flatten() * wsum).sum().backward()
    bb.grid_window, bb.grid_grad_sink = None, None
    for p in bb.parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all())
    def norm_ratio(a, b):
        return float(a.double().cpu().norm() / T(b).double().norm())
    assert 0.9 <= norm_ratio(bb.conv1.weight.grad, g["dw_conv1"]) <= 1.1
    # d gamma of the first BatchNorm is a sum of signed terms over all 64^3 voxels behind 42 chaotic layers: its norm ratio spreads
    # 0.91 ... 1.07 from run to run (the forward's statistics are fp32 atomic sums; tools/resnet_chain_spread.py, 60 runs), the
    # other norms 1.01 ... 1.06 -- so this one bound is wider
    assert 0.8 <= norm_ratio(bb.bn1.weight.grad, g["dgamma_bn1"]) <= 1.2
    assert 0.9 <= norm_ratio(bb.layer2[0].downsample[1].weight.grad, g["dgamma_l2_0_ds"]) <= 1.1
    np.testing.assert_allclose(bb.layer1[0].conv2.weight.grad.double().pow(2).mean().sqrt().item(), g["dw_l1_0_conv2_stats"][2], rtol=0.1)
    np.testing.assert_allclose(bb.layer3[5].conv3.weight.grad.double().pow(2).mean().sqrt().item(), g["dw_l3_5_conv3_stats"][2], rtol=0.1)
    assert rel_l2(bb.conv1.weight.grad, T(g["dw_conv1"])) <= 0.75
    assert rel_l2(bb.bn1.bias.grad, T(g["dbeta_bn1"])) <= 0.75
    dx = got["dx"].reshape(7, S, S, S).cpu()
    np.testing.assert_allclose(dx.double().pow(2).mean().sqrt().item(), g["dx_stats"][2], rtol=0.1)
    pi = g["probe_idx"]
    assert rel_l2(dx[pi[:, 0], pi[:, 1], pi[:, 2], pi[:, 3]], T(g["dx_probe"])) <= 0.75


def test_graph_replay_after_interleaved_graph_matches_first_launch():
    """The forward / backward launch sequences are replayed as hipGraphs keyed by their arguments.  Regression for the one failure
    that mode showed: train step -> eval forward (a different graph, un-normalised activations with momentum 0) -> train step
    REPLAYED; with memset nodes inside the graphs the replay produced NaN gradients (un-zeroed BatchNorm accumulators)."""
    import ctypes as C
    from neraf_amd import _lib
    dev = torch.device("cuda:0")
    net = _model(dev, 1 / 64)
    bb = net.backbone_net
    for m in bb.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.momentum = 0.0
    x = T(synth.uniform("g1.grid64", (1, 7, 64, 64, 64), 0.0, 1.0)).to(dev)
    w = T(synth.normal("graph.w", (1024,))).to(dev)

    def train_step():
        net.train()
        for p in net.parameters():
            p.grad = None
        (net(x).flatten() * w).sum().backward()
        g = bb.layer2[1].conv2.weight.grad
        assert bool(torch.isfinite(g).all()) and bool(torch.isfinite(bb.bn1.weight.grad).all())
        return float(g.double().pow(2).mean().sqrt()), float(bb.conv1.weight.grad.double().pow(2).mean().sqrt())

    first = train_step()
    net.eval()
    with torch.no_grad():
        net(x)
    second = train_step()
    third = train_step()
    lib = _lib.load()
    cap, lau = C.c_int(), C.c_int()
    enabled = lib.neraf_graph_stats(_lib.ctx(0), C.byref(cap), C.byref(lau))
    if enabled:
        assert lau.value > cap.value            # at least one sequence was replayed
    for a, b in zip(first, second):
        np.testing.assert_allclose(b, a, rtol=0.2)
    for a, b in zip(first, third):
        np.testing.assert_allclose(b, a, rtol=0.2)

"""GPU parity tests (run with ``-m gpu`` on the MI355X box): libneraf_hip through the C ABI vs the
CPU oracle (oracle/audio.py, pinned to the imported reference by tests/golden/*) on identical
seeded inputs.

Tolerances (stated once, used below): the HIP path computes the NAcF GEMMs with fp16 operands and
fp32 accumulation (the reference trains under fp16 autocast, NeRAF_config.py:79), so against the
fp32 oracle we require
    * log-magnitude outputs (range +-10): relative L2 error <= 3e-3 and max |err| <= 0.05
    * gradients: relative L2 error <= 5e-2 per tensor.  (The bound is set by the LeakyReLU kink, not by
      the GEMMs: pre-activations within ~1e-3 of zero change sign between the fp16 forward and the fp32
      oracle, which flips that unit's derivative 1 <-> 0.1; ~0.1% of units flipping gives ~2.5% relative
      L2 error on a gradient tensor.  The reference's own fp16 autocast has the same property.  A wrong
      transpose/layer/mask would show as O(1) error.)
The raw GEMM on small-integer data must be bit-exact (products and sums exactly representable).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from neraf_amd import synth

pytestmark = pytest.mark.gpu

OUT_REL_L2, OUT_MAX_ABS, GRAD_REL_L2 = 3e-3, 0.05, 5e-2


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def T(a, dev=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dev) if dev is not None else t


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(2048, 2048, 256), (200, 300, 192), (1024, 256, 128), (2048, 5096, 192), (4096, 4096, 128)])
def test_gemm_bf16_exact_small_integers(dev, M, N, K):
    """bfloat16 instantiation (gradient chains): exact on small integers (8 significant bits suffice)."""
    from neraf_amd import _lib
    lib = _lib.load()
    Mp, Np = (M + 127) // 128 * 128, (N + 127) // 128 * 128
    rng = np.random.default_rng(99 + M + N)
    A = rng.integers(-3, 4, size=(Mp, K)).astype(np.float32)
    B = rng.integers(-2, 3, size=(Np, K)).astype(np.float32)
    ref = (A[:M].astype(np.float64) @ B[:N].astype(np.float64).T).astype(np.float32)
    Ad, Bd = T(A, dev).bfloat16(), T(B, dev).bfloat16()
    C16 = torch.full((Mp, Np), 7.0, dtype=torch.bfloat16, device=dev)
    C16T = torch.full((Np, Mp), 7.0, dtype=torch.bfloat16, device=dev)
    C32 = torch.full((M, N), 7.0, dtype=torch.float32, device=dev)
    _lib.check(lib.neraf_gemm_bf16(_lib.ctx(0), Ad.data_ptr(), K, Bd.data_ptr(), K, M, N, K, Mp, Np, 1.0, None, 0,
                                   C16.data_ptr(), Np, C16T.data_ptr(), Mp, C32.data_ptr(), N,
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(C32.cpu().numpy(), ref)
    full = np.zeros((Mp, Np), np.float32)
    full[:M, :N] = ref
    np.testing.assert_array_equal(C16.float().cpu().numpy(), T(full).bfloat16().float().numpy())
    np.testing.assert_array_equal(C16T.float().cpu().numpy(), T(full).bfloat16().float().numpy().T)


@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (256, 192, 4096), (64, 1024, 32768), (1024, 256, 512), (128, 64, 640)])
def test_gemm_bf16_tn_exact_small_integers(dev, M, N, K):
    """"TN" form (both operands K-major, read through ds_read_b64_tr_b16): C[m][n] = sum_k A[k][m] B[k][n], exact on small
    integers -- checks the transposed-read fragment maps, the source-side swizzle, the split-K slabs and the reducer of the
    grouped weight-gradient kernel in its plain single-matrix case."""
    from neraf_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(7 + M + N + K)
    A = rng.integers(-3, 4, size=(K, M)).astype(np.float32)
    B = rng.integers(-2, 3, size=(K, N)).astype(np.float32)
    ref = (A.astype(np.float64).T @ B.astype(np.float64)).astype(np.float32)
    Ad, Bd = T(A, dev).bfloat16(), T(B, dev).bfloat16()
    C32 = torch.full((M, N), 7.0, dtype=torch.float32, device=dev)
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=dev)
    _lib.check(lib.neraf_gemm_bf16_tn(_lib.ctx(0), Ad.data_ptr(), Bd.data_ptr(), M, N, K, C32.data_ptr(), ws.data_ptr(), ws.numel(),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(C32.cpu().numpy(), ref)


@pytest.mark.parametrize("M,N,K", [(2048, 2048, 256), (200, 300, 192), (2048, 513, 512), (130, 5096, 64),
                                   # the wide 8-wave bodies: 256x160 tiles (uneven LDS-DMA counts per wave), 256x128 tiles
                                   (2048, 5096, 2048), (2000, 5000, 128), (4096, 4096, 256), (5096, 5096, 64)])
@pytest.mark.parametrize("act", [0, 1])
def test_gemm_exact_small_integers(dev, M, N, K, act):
    """Asymmetric small-integer operands: every product/sum is exact in fp16 x fp16 -> fp32, so the MFMA
    fragment maps, the LDS swizzle, the tile remap and all three epilogue writers must agree bit for bit
    with an integer matmul."""
    from neraf_amd import _lib
    lib = _lib.load()
    Mp, Np = (M + 127) // 128 * 128, (N + 127) // 128 * 128
    rng = np.random.default_rng(1234 + M + N)
    A = rng.integers(-3, 4, size=(Mp, K)).astype(np.float32)
    B = rng.integers(-2, 3, size=(Np, K)).astype(np.float32)
    bias = rng.integers(-4, 5, size=(Np,)).astype(np.float32)
    ref = (A[:M].astype(np.float64) @ B[:N].astype(np.float64).T + bias[:N]).astype(np.float32)
    if act == 1:
        ref = np.where(ref > 0, ref, np.float32(0.1) * ref).astype(np.float32)  # one fp32 multiply, as the kernel
    Ad, Bd, bd = T(A, dev).half(), T(B, dev).half(), T(bias, dev)
    C16 = torch.full((Mp, Np), 7.0, dtype=torch.float16, device=dev)
    C16T = torch.full((Np, Mp), 7.0, dtype=torch.float16, device=dev)
    C32 = torch.full((M, N), 7.0, dtype=torch.float32, device=dev)
    rc = lib.neraf_gemm_f16(_lib.ctx(0), Ad.data_ptr(), K, Bd.data_ptr(), K, M, N, K, Mp, Np, 1.0, bd.data_ptr(), act,
                            C16.data_ptr(), Np, C16T.data_ptr(), Mp, C32.data_ptr(), N,
                            C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(C32.cpu().numpy(), ref.astype(np.float32))
    full = np.zeros((Mp, Np), np.float32)
    full[:M, :N] = ref
    exp16 = full.astype(np.float16)
    np.testing.assert_array_equal(C16.cpu().numpy(), exp16)
    np.testing.assert_array_equal(C16T.cpu().numpy(), exp16.T)


# ------------------------------------------------------------------------------------------------
def _make_field(C_, F_, dev):
    from neraf_amd.field import NeRAFAudioSoundField
    sd = synth.nacf_state_dict(1187, 512, C_, F_)
    f = NeRAFAudioSoundField(1187, 512, sound_rez=C_, N_frequencies=F_)
    f.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    return f.to(dev), {k: T(v) for k, v in sd.items()}


@pytest.mark.parametrize("C_,F_,tag", [(1, 513, "g2_nacf_raf"), (2, 257, "g2_nacf_ss")])
def test_dense_forward_backward_vs_golden(dev, golden, C_, F_, tag):
    """NeRAFAudioSoundField.forward(h) (NeRAF_field.py:47) against the reference's own outputs (G2)."""
    g = golden(tag)
    f, _ = _make_field(C_, F_, dev)
    h = T(synth.uniform("g2.h", (8, 1187), -1.0, 1.0), dev).requires_grad_(True)
    wout = T(synth.uniform("g2.wout", (8, C_, F_), -1.0, 1.0), dev)
    y = f(h)
    assert y.shape == (8, C_, F_) and y.dtype == torch.float32
    assert rel_l2(y, T(g["out"])) <= OUT_REL_L2
    assert float((y.detach().cpu() - T(g["out"])).abs().max()) <= OUT_MAX_ABS
    (y * wout).sum().backward()
    assert rel_l2(h.grad, T(g["dh"])) <= GRAD_REL_L2
    assert rel_l2(f.soundfield[0].weight.grad[:4, :8], T(g["dw0_slab"])) <= 2 * GRAD_REL_L2
    assert rel_l2(f.soundfield[0].bias.grad[:16], T(g["db0"])) <= GRAD_REL_L2
    assert rel_l2(f.STFT_linear[0].weight.grad[:4, :8], T(g["dwh0_slab"])) <= GRAD_REL_L2
    assert rel_l2(f.STFT_linear[C_ - 1].bias.grad, T(g["dbh_last"])) <= GRAD_REL_L2
    gw0 = f.soundfield[0].weight.grad.double().cpu()
    np.testing.assert_allclose(gw0.abs().mean().item(), g["dw0_stats"][1], rtol=2e-2)


@pytest.mark.parametrize("B,C_,F_,T_", [(2048, 1, 513, 60), (101, 2, 257, 101), (60, 1, 513, 60), (1, 1, 513, 60)])
def test_split_forward_backward_vs_oracle(dev, B, C_, F_, T_):
    """forward_queries (GPU prologue + layer-0 split) vs the oracle's get_outputs (NeRAF_model.py:531-566),
    including rows outside the audio AABB, at the BASELINE batch (2048 slices) and ragged sizes."""
    from oracle import audio as O
    f, sd = _make_field(C_, F_, dev)
    b = synth.audio_batch(B, C_, F_, T_, tag=f"t.split{B}")
    aabb = T(synth.audio_aabb())
    feat = T(synth.uniform("t.feat", (1024,), 0.0, 2.0))
    wout = T(synth.uniform(f"t.wout{B}", (B, C_, F_), -1.0, 1.0))
    # oracle (fp32 CPU)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    feat_o = feat.clone().requires_grad_(True)
    bt = {k: T(v) for k, v in b.items()}
    yo = O.audio_get_outputs(bt, feat_o, sdo, aabb, T_)
    (yo * wout).sum().backward()
    # HIP
    feat_d = feat.to(dev).requires_grad_(True)
    y = f.forward_queries(feat_d, bt["time_query"].to(dev), bt["mic_pose"].to(dev), bt["source_pose"].to(dev),
                          bt["rot"].to(dev), aabb, T_)
    assert y.shape == (B, C_, F_)
    assert rel_l2(y, yo) <= OUT_REL_L2
    assert float((y.detach().cpu() - yo.detach()).abs().max()) <= OUT_MAX_ABS
    (y * wout.to(dev)).sum().backward()
    assert rel_l2(feat_d.grad, feat_o.grad) <= GRAD_REL_L2
    for name, p in f.state_dict(keep_vars=True).items():
        assert rel_l2(p.grad, sdo[name].grad) <= GRAD_REL_L2, name


def test_shared_pose_row_equals_expanded_rows(dev):
    """The eval branch hands forward_queries ONE (microphone, source, orientation) for the T time queries of a RIR
    (neraf_nacf_encode_queries_ex, pose_rows = 1): the same bits as the reference's [T,3] expansions (NeRAF_model.py:676-678), with and
    without a graph being recorded."""
    f, _ = _make_field(1, 513, dev)
    T_ = 60
    b = synth.audio_batch(4, 1, 513, T_, tag="t.shared_pose")
    aabb = T(synth.audio_aabb())
    feat = T(synth.uniform("t.feat", (1024,), 0.0, 2.0)).to(dev)
    tq = torch.arange(T_, device=dev)
    mic, src, rot = (T(b[k][2]).to(dev).reshape(1, 3) for k in ("mic_pose", "source_pose", "rot"))
    with torch.no_grad():
        y1 = f.forward_queries(feat, tq, mic, src, rot, aabb, T_)
        yT = f.forward_queries(feat, tq, mic.expand(T_, -1), src.expand(T_, -1), rot.expand(T_, -1), aabb, T_)
    assert y1.shape == (T_, 1, 513) and torch.equal(y1, yT)
    yg = f.forward_queries(feat.clone().requires_grad_(True), tq, mic, src, rot, aabb, T_)          # through the autograd node
    assert torch.allclose(yg.detach(), y1, rtol=0, atol=2e-3)       # training-mode packing of the same weights
    with pytest.raises(ValueError):
        f.forward_queries(feat, tq, mic, src.expand(T_, -1), rot, aabb, T_)


def test_split_equals_dense(dev):
    """Layer-0 split + GPU encodings == dense path on h = cat[feat, oracle encodings] (NeRAF_model.py:560)."""
    from oracle import audio as O
    f, _ = _make_field(1, 513, dev)
    B, T_ = 300, 60
    b = {k: T(v) for k, v in synth.audio_batch(B, 1, 513, T_, tag="t.eq").items()}
    aabb = T(synth.audio_aabb())
    feat = T(synth.uniform("t.feat", (1024,), 0.0, 2.0))
    q = O.audio_prologue(b["time_query"], b["mic_pose"], b["source_pose"], b["rot"], aabb, T_)
    h = torch.cat([feat.expand(B, -1), q], dim=-1).to(dev)
    with torch.no_grad():
        yd = f(h)
        ys = f.forward_queries(feat.to(dev), b["time_query"].to(dev), b["mic_pose"].to(dev), b["source_pose"].to(dev),
                               b["rot"].to(dev), aabb, T_)
    assert rel_l2(ys, yd) <= 2e-3


@pytest.mark.parametrize("C_,F_", [(1, 513), (2, 257)])
@pytest.mark.parametrize("lt", ["mse", "l1"])
def test_stft_loss_vs_golden(dev, golden, C_, F_, lt):
    """STFTLoss (NeRAF_evaluator.py:88-108) + loss scaling (NeRAF_model.py:597-598) against G3."""
    from neraf_amd.losses import STFTLoss
    g = golden("g3_stft_loss")
    x = T(synth.uniform(f"g3.x{C_}", (8, C_, F_), -6.0, 2.0), dev).requires_grad_(True)
    y = T(synth.uniform(f"g3.y{C_}", (8, C_, F_), -6.0, 2.0), dev)
    d = STFTLoss(loss_type=lt)(x, y)
    np.testing.assert_allclose(d["audio_sc_loss"].item(), g[f"sc_{lt}_{C_}"], rtol=1e-5)
    np.testing.assert_allclose(d["audio_mag_loss"].item(), g[f"mag_{lt}_{C_}"], rtol=1e-5)
    (d["audio_sc_loss"] * 1e-1 * 1e-3 + d["audio_mag_loss"] * 1.0 * 1e-3).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"dx_{lt}_{C_}"], rtol=2e-4, atol=1e-10)


def test_stft_loss_full_batch_linearity(dev):
    """Full BASELINE size [2048,1,513]: the Frobenius sums are additive over a partition of the batch."""
    lib_in = synth.audio_batch(2048, 1, 513, 60, tag="t.loss")
    from neraf_amd.losses import STFTLoss
    y = T(lib_in["data"], dev)
    x = y + 0.3 * T(synth.normal("t.loss.noise", (2048, 1, 513)), dev)
    crit = STFTLoss("mse")
    full = crit(x, y)
    mag_parts = torch.stack([crit(x[i::4], y[i::4])["audio_mag_loss"] for i in range(4)]).mean()
    np.testing.assert_allclose(full["audio_mag_loss"].item(), mag_parts.item(), rtol=1e-5)
    from oracle import audio as O
    sc, mag = O.stft_loss(x.cpu(), y.cpu(), "mse")
    np.testing.assert_allclose(full["audio_sc_loss"].item(), sc.item(), rtol=1e-4)
    np.testing.assert_allclose(full["audio_mag_loss"].item(), mag.item(), rtol=1e-4)

"""CPU check of the MFMA weight-fragment packing used by csrc/field.hip: emulate the kernel's per-lane data flow
(v_mfma_f32_16x16x32 operand maps, accumulator blocks re-used as the next layer's B operand) in float64 and
compare with the plain MLP arithmetic of the oracle.  Exercises neraf_amd.vision._FRAG_INDEX only (no GPU)."""
import numpy as np

from neraf_amd import synth
from neraf_amd.vision import _FRAG_INDEX, _FRAG_INDEX_BWD


def mfma(afrag, bfrag, acc=None):
    """afrag/bfrag [64,8]: lane l=(i=l&15, q=l>>4) holds A[i][8q+j] / B[8q+j][i].  Returns D as per-lane regs [64,4]:
    lane (c=l&15, q) holds D[4q+r][c]."""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for l in range(64):
        i, q = l & 15, l >> 4
        A[i, 8 * q:8 * q + 8] = afrag[l]
        B[8 * q:8 * q + 8, i] = bfrag[l]
    D = A @ B
    out = np.zeros((64, 4))
    for l in range(64):
        c, q = l & 15, l >> 4
        out[l] = D[4 * q:4 * q + 4, c]
    return out if acc is None else out + acc


def pack(a, b, relu=True):
    h = np.concatenate([a, b], axis=1)
    return np.maximum(h, 0) if relu else h


def test_field_fragment_dataflow_matches_plain_mlp():
    P = synth.vision_params((8, 8, 8), num_train_data=4, table_scale=0.5)
    W = {k: P["field." + k].astype(np.float64) for k in ("base_w0", "base_w1", "head_w0", "head_w1", "head_w2")}
    flat = np.concatenate([W["base_w0"].ravel(), W["base_w1"].ravel(), W["head_w0"].ravel(), W["head_w1"].ravel(),
                           W["head_w2"].ravel(), np.zeros(1)])
    wf = flat[_FRAG_INDEX].reshape(24, 64, 8)
    rng = np.random.default_rng(0)
    enc = rng.normal(size=(16, 32))           # 16 points x 32 hash features
    sh = rng.normal(size=(16, 16))
    emb = rng.normal(size=(16, 32))
    # ---- emulate the kernel
    xin = np.zeros((64, 8)); h1 = np.zeros((64, 8)); shl = np.zeros((64, 4))
    for l in range(64):
        p, q = l & 15, l >> 4
        xin[l] = enc[p, 8 * q:8 * q + 8]
        h1[l] = emb[p, 8 * q:8 * q + 8]
        shl[l] = sh[p, 4 * q:4 * q + 4]
    d1 = [mfma(wf[ob], xin) for ob in range(4)]
    d2 = mfma(wf[4], pack(d1[0], d1[1]))
    d2 = mfma(wf[5], pack(d1[2], d1[3]), d2)
    h0 = np.concatenate([d2, shl], axis=1)
    for l in range(16):          # q == 0 lanes: element 0 is the density logit
        h0[l, 0] = 0.0
    d3 = [mfma(wf[7 + 2 * ob], h1, mfma(wf[6 + 2 * ob], h0)) for ob in range(4)]
    a0, a1 = pack(d3[0], d3[1]), pack(d3[2], d3[3])
    d4 = [mfma(wf[15 + 2 * ob], a1, mfma(wf[14 + 2 * ob], a0)) for ob in range(4)]
    d5 = mfma(wf[23], pack(d4[2], d4[3]), mfma(wf[22], pack(d4[0], d4[1])))
    logit_k = np.array([d2[p, 0] for p in range(16)])          # lane q=0, reg 0
    rgb_k = np.array([d5[p, :3] for p in range(16)])
    # ---- plain arithmetic (oracle.vision.field_forward's MLP part)
    hb = np.maximum(enc @ W["base_w0"].T, 0) @ W["base_w1"].T           # [16,16]
    hin = np.concatenate([sh, hb[:, 1:16], emb, np.zeros((16, 1))], axis=1)   # 63 -> 64
    x = np.maximum(hin @ W["head_w0"].T, 0)
    x = np.maximum(x @ W["head_w1"].T, 0)
    rgb = (x @ W["head_w2"].T)[:, :3]
    np.testing.assert_allclose(logit_k, hb[:, 0], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(rgb_k, rgb, rtol=1e-10, atol=1e-12)


def test_field_backward_fragment_dataflow_matches_autograd():
    """Same emulation for csrc/field_bwd.hip: dX chain through the transposed-weight fragments."""
    import torch
    P = synth.vision_params((8, 8, 8), num_train_data=4, table_scale=0.5)
    Wn = {k: P["field." + k].astype(np.float64) for k in ("base_w0", "base_w1", "head_w0", "head_w1", "head_w2")}
    flat = np.concatenate([Wn["base_w0"].ravel(), Wn["base_w1"].ravel(), Wn["head_w0"].ravel(), Wn["head_w1"].ravel(),
                           Wn["head_w2"].ravel(), np.zeros(1)])
    wf = flat[_FRAG_INDEX].reshape(24, 64, 8)
    wb = flat[_FRAG_INDEX_BWD].reshape(28, 64, 8)
    rng = np.random.default_rng(1)
    enc = rng.normal(size=(16, 32)); sh = rng.normal(size=(16, 16)); emb = rng.normal(size=(16, 32))
    g_rgb = rng.normal(size=(16, 3)); g_logit = rng.normal(size=(16,))
    # ---- autograd reference on the plain MLP
    t = {k: torch.tensor(v, requires_grad=True) for k, v in Wn.items()}
    enc_t = torch.tensor(enc, requires_grad=True); emb_t = torch.tensor(emb, requires_grad=True)
    hb = torch.relu(enc_t @ t["base_w0"].T) @ t["base_w1"].T
    hin = torch.cat([torch.tensor(sh), hb[:, 1:16], emb_t, torch.zeros(16, 1, dtype=torch.float64)], 1)
    x = torch.relu(torch.relu(hin @ t["head_w0"].T) @ t["head_w1"].T) @ t["head_w2"].T
    loss = (x[:, :3] * torch.tensor(g_rgb)).sum() + (hb[:, 0] * torch.tensor(g_logit)).sum()
    loss.backward()
    # ---- emulate the kernel: forward
    xin = np.zeros((64, 8)); h1 = np.zeros((64, 8)); shl = np.zeros((64, 4))
    for l in range(64):
        p, q = l & 15, l >> 4
        xin[l] = enc[p, 8 * q:8 * q + 8]; h1[l] = emb[p, 8 * q:8 * q + 8]; shl[l] = sh[p, 4 * q:4 * q + 4]
    d1 = [mfma(wf[ob], xin) for ob in range(4)]
    d2 = mfma(wf[5], pack(d1[2], d1[3]), mfma(wf[4], pack(d1[0], d1[1])))
    h0 = np.concatenate([d2, shl], axis=1)
    h0[:16, 0] = 0.0
    d3 = [mfma(wf[7 + 2 * ob], h1, mfma(wf[6 + 2 * ob], h0)) for ob in range(4)]
    d4 = [mfma(wf[15 + 2 * ob], pack(d3[2], d3[3]), mfma(wf[14 + 2 * ob], pack(d3[0], d3[1]))) for ob in range(4)]
    # ---- backward
    zero = np.zeros((64, 4))
    dy5 = np.zeros((64, 4))
    for p in range(16):
        dy5[p, :3] = g_rgb[p]                       # lanes q == 0
    dy4 = [mfma(wb[ib], pack(dy5, zero, relu=False)) * (d4[ib] > 0) for ib in range(4)]
    b0, b1 = pack(dy4[0], dy4[1], relu=False), pack(dy4[2], dy4[3], relu=False)
    dy3 = [mfma(wb[5 + ib * 2], b1, mfma(wb[4 + ib * 2], b0)) * (d3[ib] > 0) for ib in range(4)]
    b0, b1 = pack(dy3[0], dy3[1], relu=False), pack(dy3[2], dy3[3], relu=False)
    dbase = mfma(wb[13], b1, mfma(wb[12], b0))
    demb = [mfma(wb[15 + ib * 2], b1, mfma(wb[14 + ib * 2], b0)) for ib in range(2)]
    dy2 = dbase.copy()
    for p in range(16):
        dy2[p, 0] = g_logit[p]
    dy1 = [mfma(wb[18 + ib], pack(dy2, zero, relu=False)) * (d1[ib] > 0) for ib in range(4)]
    b0, b1 = pack(dy1[0], dy1[1], relu=False), pack(dy1[2], dy1[3], relu=False)
    de = [mfma(wb[23 + rb * 2], b1, mfma(wb[22 + rb * 2], b0)) for rb in range(2)]
    denc = np.zeros((16, 32)); demb_k = np.zeros((16, 32))
    for l in range(64):
        p, q = l & 15, l >> 4
        denc[p, 8 * q:8 * q + 4] = de[0][l]; denc[p, 8 * q + 4:8 * q + 8] = de[1][l]
        for ib in range(2):
            demb_k[p, 16 * ib + 4 * q:16 * ib + 4 * q + 4] = demb[ib][l]
    np.testing.assert_allclose(denc, enc_t.grad.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(demb_k, emb_t.grad.numpy(), rtol=1e-9, atol=1e-12)
    # weight gradients from the dumped (X, dY) pairs: dW = dY . X^T over points (natural feature order)
    def nat(blocks):            # D-layout blocks -> [feat, 16 points]
        out = np.zeros((16 * len(blocks), 16))
        for ob, b in enumerate(blocks):
            for l in range(64):
                p, q = l & 15, l >> 4
                out[16 * ob + 4 * q:16 * ob + 4 * q + 4, p] = b[l]
        return out
    dW_h2 = nat([dy5]) @ np.maximum(nat(d4), 0).T
    dW_h1 = nat(dy4) @ np.maximum(nat(d3), 0).T
    dW_b1 = nat([dy2]) @ np.maximum(nat(d1), 0).T
    dW_b0 = nat(dy1) @ enc
    np.testing.assert_allclose(dW_h2, t["head_w2"].grad.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(dW_h1, t["head_w1"].grad.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(dW_b1, t["base_w1"].grad.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(dW_b0.T.T, t["base_w0"].grad.numpy(), rtol=1e-9, atol=1e-12)
    Xh0 = np.concatenate([sh, nat([d2]).T[:, 1:16], emb, np.zeros((16, 1))], axis=1)      # natural column order
    np.testing.assert_allclose(nat(dy3) @ Xh0, t["head_w0"].grad.numpy(), rtol=1e-9, atol=1e-12)


def test_packed_fixed_point_pair_sums_decode_exactly():
    """field_scatter_kernel packs both features of a table entry as (q1 << 32) + q0 and adds them with ONE 64-bit integer
    atomic; field_unpack_grad_kernel decodes lo = int32(total), hi = (total - lo) >> 32.  Emulated here in int64: the decode is
    exact for any sign pattern as long as each per-feature sum stays inside int32 (what the per-level scale guarantees)."""
    rng = np.random.default_rng(5)
    for _ in range(50):
        n = int(rng.integers(1, 2000))
        q0 = rng.integers(-(2 ** 31 - 1) // n, (2 ** 31 - 1) // n + 1, size=n).astype(np.int64)
        q1 = rng.integers(-(2 ** 31 - 1) // n, (2 ** 31 - 1) // n + 1, size=n).astype(np.int64)
        packed = (q1 << 32) + q0                       # what each lane adds (two's complement, as the kernel does)
        total = np.int64(0)
        for v in rng.permutation(packed):              # any order: integer adds commute -> bit-reproducible
            total = np.int64(total + v)
        lo = np.int64(np.int32(total & 0xFFFFFFFF))
        hi = (total - lo) >> 32
        assert lo == q0.sum() and hi == q1.sum()


def test_fixed_point_level_scale_bound():
    """F_l = 2^(29 - ceil(log2 T_l)) with T_l = sum_n max(|g0|, |g1|): every entry's |sum| <= T_l F_l + n_contrib / 2 < 2^31."""
    rng = np.random.default_rng(6)
    g = rng.standard_normal((4096, 2)).astype(np.float32) * 10.0 ** rng.uniform(-6, 3, size=(4096, 1)).astype(np.float32)
    T = float(np.abs(g).max(axis=1).sum())
    F = 2.0 ** (29 - int(np.ceil(np.log2(T))))
    w = rng.dirichlet(np.ones(8), size=4096).astype(np.float32)          # trilinear weights of a sample add to 1
    q = np.rint(w[:, :, None] * g[:, None, :] * F).astype(np.int64)
    worst = np.abs(q).sum(axis=(0, 1)).max()                             # all contributions landing in one entry
    assert worst < 2 ** 31
    # resolution: the quantised total matches the float total to ~1e-6 of the gradient mass
    tot = (w[:, :, None] * g[:, None, :]).sum(axis=(0, 1))
    np.testing.assert_allclose(q.sum(axis=(0, 1)) / F, tot, atol=1e-6 * T)


def test_packed_copies_are_cached_per_parameter_state():
    """NerfactoField.packed() / packed_bwd() / HashMLPDensityField.packed() make their fp16 copies once per parameter state: the
    key is the fused optimizer's update counter (its kernels write through raw pointers) plus every parameter's storage pointer
    and version counter (torch-side in-place updates)."""
    import torch
    from neraf_amd import optim
    from neraf_amd.vision import HashMLPDensityField, NerfactoField
    f = NerfactoField(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_images=4)
    a = f.packed(with_average=False)
    b = f.packed(with_average=False)
    assert all(x is y for x, y in zip(a, b))                      # same objects: nothing was converted again
    c = f.packed(with_average=True)
    assert c[0] is a[0] and c[1] is a[1] and c[2].shape[0] == a[2].shape[0] + 1     # only the embedding rows differ
    wb = f.packed_bwd()
    assert f.packed_bwd() is wb
    with torch.no_grad():
        f.base_w0.mul_(2.0)                                       # torch-side update: version counter
    d = f.packed(with_average=False)
    assert d[1] is not a[1] and not torch.equal(d[1], a[1]) and f.packed_bwd() is not wb
    optim.UPDATE_EPOCH += 1                                       # what FusedAdam.step() does
    e = f.packed(with_average=False)
    assert e[0] is not d[0] and torch.equal(e[0], d[0])
    f.invalidate_packed()
    assert f.packed(with_average=False)[0] is not e[0]
    p = HashMLPDensityField.__new__(HashMLPDensityField)
    torch.nn.Module.__init__(p)
    p.table = torch.nn.Parameter(torch.rand(64, 2))
    p.w0 = torch.nn.Parameter(torch.rand(16, 16))
    p.w1 = torch.nn.Parameter(torch.rand(16, 16))
    t0 = p.packed()
    assert p.packed()[0] is t0[0]
    with torch.no_grad():
        p.table.add_(1.0)
    assert p.packed()[0] is not t0[0]

"""CPU check of the MFMA weight-fragment packing used by csrc/field.hip: emulate the kernel's per-lane data flow
(v_mfma_f32_16x16x32 operand maps, accumulator blocks re-used as the next layer's B operand) in float64 and
compare with the plain MLP arithmetic of the oracle.  Exercises neraf_amd.vision._FRAG_INDEX only (no GPU)."""
import numpy as np

from neraf_amd import synth
from neraf_amd.vision import _FRAG_INDEX


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

"""Size-independent properties at the BASELINE shapes (4096 rays x 48 samples, 2048 RIR slices), where the CPU oracle is too slow
to be the checker, and the C ABI's error behaviour.

* the hash-grid gradient is BIT-REPRODUCIBLE (packed fixed-point integer atomics) and scales EXACTLY with a power-of-two
  upstream factor (the chain's automatic scale and the per-level fixed-point scale are powers of two);
* rendering invariants of the full forward (colours in [0,1], weights in [0,1] summing to <= 1, finite depth);
* the NAcF backward is linear in the upstream gradient; STFT loss of a tensor against itself is exactly zero;
* bad arguments return NERAF_EINVAL with a message, never a crash."""
import ctypes as C

import numpy as np
import pytest
import torch

from neraf_amd import synth

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def vm():
    from neraf_amd.vision import NeRAFVisionModel
    dev = torch.device("cuda:0")
    m = NeRAFVisionModel(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 210).to(dev)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for p in [m.field.module.table] + [pn.table for pn in m.proposal_networks]:
            p.copy_((torch.rand(p.shape, generator=g) - 0.5).to(dev))
    return m, dev


def _field_backward(m, dev, scale):
    from neraf_amd.vision import RayBundle
    f = m.field.module
    rb = synth.ray_batch(4096, tag="prop.rays")
    o, d = T(rb["origins"]).to(dev), T(rb["directions"]).to(dev)
    cam = T(rb["camera_indices"]).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    e = torch.sort(torch.rand((4096, 49), generator=g, device=dev) * 4.0 + 0.05, dim=1).values.contiguous()
    packed = f.packed()
    rgb, den = f.query(o, d, e, cam, packed=packed)
    d_rgb = (torch.rand(rgb.shape, generator=g, device=dev) - 0.5) * 1e-3 * scale
    d_den = (torch.rand(den.shape, generator=g, device=dev) - 0.5) * 1e-5 * scale
    return f.backward_query(packed, o, d, e, cam, den, d_rgb.contiguous(), d_den.contiguous())


def test_table_gradient_is_bit_reproducible_and_scales_exactly(vm):
    m, dev = vm
    a = _field_backward(m, dev, 1.0)
    b = _field_backward(m, dev, 1.0)
    assert torch.equal(a[0], b[0])                                   # 2.5e7 atomics in arbitrary order, identical bits
    assert float(a[0].abs().max()) > 0 and bool(torch.isfinite(a[0]).all())
    c = _field_backward(m, dev, 4.0)
    assert torch.equal(c[0], 4.0 * a[0])                             # power-of-two factor: exact
    for x, y in zip(a[1:6], c[1:6]):                                 # MLP weight gradients: split-K fp32 sums, same order
        np.testing.assert_allclose(y.cpu().numpy(), 4.0 * x.cpu().numpy(), rtol=1e-5, atol=0)


def test_owner_scatter_equals_atomic_scatter_bit_for_bit(vm, monkeypatch):
    """The hashed levels' gradient is summed either by 64-bit global atomics or, for large batches, by workgroups that own a
    table slice in LDS (field_scatter_owner_kernel): the same fixed-point integers are added, so the results are identical."""
    m, dev = vm
    monkeypatch.setenv("NERAF_FIELD_OWNER_SCATTER", "0")
    a = _field_backward(m, dev, 1.0)
    monkeypatch.setenv("NERAF_FIELD_OWNER_SCATTER", "1")
    b = _field_backward(m, dev, 1.0)
    assert float(a[0].abs().max()) > 0
    assert torch.equal(a[0], b[0])
    for x, y in zip(a[1:6], b[1:6]):                                 # nothing else may change (fp32 sums: not bit-stable)
        np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=1e-4, atol=1e-7)


def test_full_size_render_invariants(vm):
    from neraf_amd.vision import RayBundle
    m, dev = vm
    rb = synth.ray_batch(4096, tag="prop.rays2")
    m.train()
    out = m.get_outputs(RayBundle(T(rb["origins"]).to(dev), T(rb["directions"]).to(dev), T(rb["camera_indices"]).to(dev)))
    rgb, acc, depth = out["rgb"], out["accumulation"], out["depth"]
    assert rgb.shape == (4096, 3) and float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0
    assert bool(torch.isfinite(depth).all()) and float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5
    for w in out["weights_list"]:
        assert float(w.min()) >= 0.0 and float(w.sum(dim=-2 if w.dim() == 3 else -1).max()) <= 1.0 + 1e-4
    s = out["ray_samples_list"][-1]
    assert s.e_bins.shape == (4096, 49) and bool((s.e_bins[:, 1:] >= s.e_bins[:, :-1]).all())      # sorted sample edges


def test_nacf_backward_is_linear_and_self_loss_is_zero():
    from neraf_amd.field import NeRAFAudioSoundField
    from neraf_amd.losses import STFTLoss
    dev = torch.device("cuda:0")
    f = NeRAFAudioSoundField(1187, 512, sound_rez=1, N_frequencies=513)
    f.load_state_dict({k: T(v) for k, v in synth.nacf_state_dict(1187, 512, 1, 513).items()})
    f = f.to(dev)
    b = {k: T(v).to(dev) for k, v in synth.audio_batch(2048, 1, 513, 60, tag="prop.audio").items()}
    feat = T(synth.normal("prop.feat", (1024,))).to(dev).requires_grad_(True)
    aabb = T(synth.audio_aabb()).to(dev)
    g = T(synth.normal("prop.g", (2048, 1, 513))).to(dev)

    def grads(mult):
        for p in f.parameters():
            p.grad = None
        feat.grad = None
        y = f.forward_queries(feat, b["time_query"], b["mic_pose"], b["source_pose"], b["rot"], aabb, 60)
        y.backward(g * mult)
        return [p.grad.clone() for p in f.parameters()] + [feat.grad.clone()], y.detach()

    g1, y = grads(1.0)
    g8, _ = grads(8.0)
    for a, c in zip(g1, g8):
        # power-of-two factor: the automatically scaled fp16 chain is identical; what differs is the order of the fp32 atomics of
        # the bias-gradient column sums (cancelling columns), i.e. a few 1e-5 of the tensor's scale
        np.testing.assert_allclose(c.cpu().numpy(), 8.0 * a.cpu().numpy(), rtol=2e-5, atol=2e-5 * 8.0 * float(a.abs().max()))
    ld = STFTLoss(loss_type="mse")(y, y.clone())
    assert float(ld["audio_sc_loss"]) == 0.0 and float(ld["audio_mag_loss"]) == 0.0


def test_bad_arguments_return_einval_with_message():
    from neraf_amd import _lib
    lib = _lib.load()
    h = _lib.ctx(0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.zeros(128 * 64, dtype=torch.float16, device="cuda:0")
    rc = lib.neraf_gemm_f16(h, x.data_ptr(), 64, x.data_ptr(), 64, 128, 128, 60, 128, 128, 1.0, None, 0, None, 0, None, 0, None, 0, st)
    assert rc != 0 and b"K must be" in lib.neraf_last_error(h)
    rc = lib.neraf_sample_uniform(h, 0, 48, 0.05, 1000.0, None, 0, None, None, st)
    assert rc != 0 and len(lib.neraf_last_error(h)) > 0
    d = _lib.ResnetDesc(96, 7, 1024)                                  # unsupported grid size
    assert lib.neraf_resnet3d_workspace_bytes(C.byref(d)) == 0 and lib.neraf_resnet3d_num_convs(C.byref(d)) == -1
    rc = lib.neraf_fused_adam(h, None, None, None, None, 0, None, 0, 0, 0.9, 0.999, 1e-8, None, None, None, st)
    assert rc != 0
    torch.cuda.synchronize()

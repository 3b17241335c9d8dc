"""GPU parity tests of the radiance-half forward kernels (csrc/field.hip through the C ABI) against the CPU
oracle (oracle/vision.py -- PARITY UNPINNED: nerfstudio / tiny-cuda-nn restated from recall, see its header).

Tolerances.  The HIP path keeps hash tables, MLP weights and inter-layer activations in fp16 with fp32
accumulation (tiny-cuda-nn's own precision); the oracle is evaluated in fp32 on the SAME fp16-rounded
parameters, so remaining differences are activation rounding and summation order:
    sampler bins            : |ds| <= 2e-6 for the uniform stage; resampled edges: median |ds| <= 2e-5
                              (inverse-CDF lookups amplify fp32 prefix-sum differences), relative 1e-4 euclidean
    proposal / field density: relative <= 2e-2 per element (exp of a logit with ~1e-3 abs error)
    colours                 : |d rgb| <= 4e-3
    rendered rgb            : |d| <= 5e-3 ; accumulation |d| <= 5e-3
"""
import ctypes as C

import numpy as np
import pytest
import torch

from neraf_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def setup(dev):
    from neraf_amd.vision import NeRAFVisionModel
    from oracle import vision as V
    spec = V.NerfactoSpec()
    tot = (spec.prop_grids[0].total, spec.prop_grids[1].total, spec.main_grid.total)
    P = {k: T(v) for k, v in synth.vision_params(tot, num_train_data=210, table_scale=0.5).items()}
    P16 = {k: v.half().float() for k, v in P.items()}        # what the kernels actually see
    aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    m = NeRAFVisionModel(aabb, 210)
    with torch.no_grad():
        for i in range(2):
            m.proposal_networks[i].table.copy_(P[f"prop{i}.table"])
            m.proposal_networks[i].w0.copy_(P[f"prop{i}.w0"])
            m.proposal_networks[i].w1.copy_(P[f"prop{i}.w1"])
        f = m.field.module
        assert f.table.shape == P["field.table"].shape
        f.table.copy_(P["field.table"])
        for k in ("base_w0", "base_w1", "head_w0", "head_w1", "head_w2", "embedding"):
            getattr(f, k).copy_(P["field." + k])
    return m.to(dev), P16, spec, V


def test_hash_encode_standalone_vs_oracle(dev, setup):
    """neraf_hash_encode (the stand-alone encoding of SURVEY 8b's op list) against oracle.vision.hash_encode on the three grids of the
    model: dense and hashed levels, positions on cell faces and at the box corners included."""
    from neraf_amd import _lib
    from neraf_amd.vision import hash_encode
    m, P16, spec, V = setup
    g = torch.Generator().manual_seed(3)
    x = torch.rand((4096, 3), generator=g)
    x[:64] = torch.randint(0, 17, (64, 3), generator=g).float() / 16.0          # faces / corners of the coarsest level, 0 and 1 included
    for gs, d, key in ((spec.main_grid, _lib.GridDesc(16, 16, 2048, 19, 2), "field.table"),
                       (spec.prop_grids[0], _lib.GridDesc(5, 16, 128, 17, 2), "prop0.table"),
                       (spec.prop_grids[1], _lib.GridDesc(5, 16, 256, 17, 2), "prop1.table")):
        ref = V.hash_encode(x, P16[key], gs)
        out = hash_encode(d, P16[key].to(dev), x.to(dev))
        assert out.shape == ref.shape == (4096, 2 * gs.n_levels)
        # the kernels form the cell coordinate with one rounding (fmaf(scale, x, 0.5), as tiny-cuda-nn does), the oracle with two
        # (x * scale + 0.5): at the finest levels the coordinate is ~2e3 and one fp32 ulp of it is 1.2e-4 of a cell, which moves an
        # interpolated value by up to that times the local table contrast (entries in +-0.5 here)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=3e-4)
        lvl0 = slice(0, 2)              # the coarsest level (coordinate < 17: ulp 1e-6) agrees to rounding
        np.testing.assert_allclose(out.cpu().numpy()[:, lvl0], ref.numpy()[:, lvl0], rtol=0, atol=2e-6)
    rc = _lib.load().neraf_hash_encode(_lib.ctx(0), None, None, None, 0, None, None)
    assert rc != 0


def test_grid_layout_matches_oracle(setup):
    from neraf_amd import _lib
    from neraf_amd.vision import grid_layout
    _, _, spec, V = setup
    for g, d in ((spec.main_grid, _lib.GridDesc(16, 16, 2048, 19, 2)), (spec.prop_grids[0], _lib.GridDesc(5, 16, 128, 17, 2)),
                 (spec.prop_grids[1], _lib.GridDesc(5, 16, 256, 17, 2))):
        sc, rs, sz, off = grid_layout(d)
        assert rs == g.resolutions and sz == g.sizes and off == g.offsets
        np.testing.assert_allclose(sc, g.scales, rtol=1e-6)


@pytest.mark.parametrize("jit", [True, False])
def test_sample_uniform(dev, setup, jit):
    from neraf_amd import _lib
    _, _, spec, V = setup
    lib = _lib.load()
    R, S = 300, 256
    rb = synth.ray_batch(R, tag="t.su")
    j = T(rb["jitters"][0]) if jit else None
    ref = V.sample_uniform(T(rb["origins"]), T(rb["directions"]), torch.full((R, 1), 0.05), torch.full((R, 1), 1000.0), S, j)
    s, e = torch.empty((R, S + 1), device=dev), torch.empty((R, S + 1), device=dev)
    jd = j.reshape(-1).to(dev) if jit else None
    _lib.check(lib.neraf_sample_uniform(_lib.ctx(0), R, S, 0.05, 1000.0, jd.data_ptr() if jit else None, 0, s.data_ptr(),
                                        e.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    np.testing.assert_allclose(s.cpu().numpy(), ref.s_bins.numpy(), atol=2e-6)
    np.testing.assert_allclose(e.cpu().numpy(), ref.e_bins.numpy(), rtol=2e-4)     # 1/(2-2s) amplifies near s=1


@pytest.mark.parametrize("i,S", [(0, 256), (1, 96), (0, 7), (1, 33)])
def test_proposal_density(dev, setup, i, S):
    m, P16, spec, V = setup
    R = 257
    rb = synth.ray_batch(R, tag="t.pd")
    o, d = T(rb["origins"]), T(rb["directions"])
    ray = V.sample_uniform(o, d, torch.full((R, 1), 0.05), torch.full((R, 1), 1000.0), S, T(rb["jitters"][0]))
    ref = V.proposal_density(ray.positions(), P16, i, spec)
    out = m.proposal_networks[i].density(o.to(dev), d.to(dev), ray.e_bins.to(dev).contiguous())
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-2, atol=1e-7)
    assert float(out.min()) >= 0.0
    # the coherent forms only re-order lanes (frame kernel when S % 16 == 0, the general order otherwise): same bits
    assert torch.equal(out, m.proposal_networks[i].density(o.to(dev), d.to(dev), ray.e_bins.to(dev).contiguous(), coherent_rays=1))


@pytest.mark.parametrize("S,n_new,anneal,jit", [(256, 96, 0.37, True), (96, 48, 1.0, True), (256, 96, 1.0, False)])
def test_pdf_resample(dev, setup, S, n_new, anneal, jit):
    from neraf_amd import _lib
    _, _, spec, V = setup
    lib = _lib.load()
    R = 203          # not a multiple of 4: exercises the tail waves
    rb = synth.ray_batch(R, tag="t.pdf")
    o, d = T(rb["origins"]), T(rb["directions"])
    near, far = torch.full((R, 1), 0.05), torch.full((R, 1), 1000.0)
    ray = V.sample_uniform(o, d, near, far, S, T(rb["jitters"][0]))
    dens = T(synth.uniform("t.pdf.dens", (R, S), 0.0, 1.0)) ** 4 * 0.5
    dens[:5] = 0.0                                                   # empty rays -> uniform resampling
    w_ref = V.get_weights(dens, ray.deltas)
    j = T(rb["jitters"][1]) if jit else None
    new = V.sample_pdf(ray, torch.pow(w_ref, anneal), n_new, near, far, j)
    w = torch.empty((R, S), device=dev)
    s_n, e_n = torch.empty((R, n_new + 1), device=dev), torch.empty((R, n_new + 1), device=dev)
    jd = j.reshape(-1).to(dev) if jit else None
    dens_d, sb_d, eb_d = dens.to(dev), ray.s_bins.to(dev).contiguous(), ray.e_bins.to(dev).contiguous()   # keep alive
    _lib.check(lib.neraf_pdf_resample(_lib.ctx(0), dens_d.data_ptr(), sb_d.data_ptr(),
                                      eb_d.data_ptr(), R, S, anneal, jd.data_ptr() if jit else None, 0,
                                      n_new, 0.05, 1000.0, w.data_ptr(), s_n.data_ptr(), e_n.data_ptr(),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    np.testing.assert_allclose(w.cpu().numpy(), w_ref.numpy(), rtol=2e-4, atol=1e-7)
    assert bool((s_n[:, 1:] >= s_n[:, :-1]).all())
    # bins: inverse-CDF lookups agree except where a CDF plateau makes the lookup ill-conditioned
    ds = (s_n.cpu() - new.s_bins).abs()
    # (ds = d(cdf) / pdf: fp32 prefix sums differ by ~1e-7 and flat pdf regions amplify that ~100x)
    assert float(ds.median()) <= 2e-5 and float((ds > 1e-3).float().mean()) <= 2e-3
    de = ((e_n.cpu() - new.e_bins).abs() / new.e_bins)
    assert float(de.median()) <= 1e-4


def test_shared_first_stage_bins_equal_per_ray_rows(dev, setup):
    """Row stride 0 (neraf_proposal_density_ex / neraf_pdf_resample_ex): one row of first-stage bin edges shared by every ray gives the
    bits of R identical rows -- the un-jittered frame render generates that single row."""
    from neraf_amd import _lib
    m, _, _, _ = setup
    lib = _lib.load()
    R, S, n_new = 203, 256, 96
    rb = synth.ray_batch(R, tag="t.shared")
    o, d = T(rb["origins"]).to(dev), T(rb["directions"]).to(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    s1, e1 = torch.empty((1, S + 1), device=dev), torch.empty((1, S + 1), device=dev)
    _lib.check(lib.neraf_sample_uniform(_lib.ctx(0), 1, S, 0.05, 1000.0, None, 0, s1.data_ptr(), e1.data_ptr(), st), 0)
    sR, eR = s1.expand(R, S + 1).contiguous(), e1.expand(R, S + 1).contiguous()
    pn = m.proposal_networks[0]
    for coherent in (0, 1):
        assert torch.equal(pn.density(o, d, e1, coherent_rays=coherent), pn.density(o, d, eR, coherent_rays=coherent))
    dens = pn.density(o, d, eR)
    outs = []
    for sb, eb, stride in ((sR, eR, S + 1), (s1, e1, 0)):
        w = torch.empty((R, S), device=dev)
        s_n, e_n = torch.empty((R, n_new + 1), device=dev), torch.empty((R, n_new + 1), device=dev)
        _lib.check(lib.neraf_pdf_resample_ex(_lib.ctx(0), dens.data_ptr(), sb.data_ptr(), eb.data_ptr(), stride, R, S, 0.7, None, 0, n_new,
                                             0.05, 1000.0, w.data_ptr(), s_n.data_ptr(), e_n.data_ptr(), st), 0)
        outs.append((w, s_n, e_n))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_weights_at_surface_densities(dev, setup):
    """Round-3 regression: optical depths of 1e9-1e17 (surface densities of a trained field).  get_weights needs the exclusive
    prefix sum of the PREVIOUS samples (RaySamples.get_weights [NS-recall]); inclusive-minus-self cancelled there and gave the sample
    behind a surface transmittance 1.  Both kernels that compute weights -- the PDF resampler and the compositor -- against the
    oracle on identical densities: rows with two adjacent huge samples, huge after moderate, moderate only, and an inf."""
    from neraf_amd import _lib
    _, _, spec, V = setup
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for S in (48, 96, 256):
        R = 67
        rb = synth.ray_batch(R, tag=f"t.surf{S}")
        o, d = T(rb["origins"]), T(rb["directions"])
        near, far = torch.full((R, 1), 0.05), torch.full((R, 1), 1000.0)
        ray = V.sample_uniform(o, d, near, far, S, T(rb["jitters"][0]))
        dens = T(synth.uniform(f"t.surf.dens{S}", (R, S), 0.0, 1.0)) ** 4 * 0.5
        k = S // 3
        dens[0::4, k] = 3.0e15; dens[0::4, k + 1] = 2.9e15; dens[0::4, k + 2] = 1.0e17      # noqa: E702 -- a surface: three huge neighbours
        dens[1::4, k] = 5.0e5; dens[1::4, k + 1] = 1.0e15                                     # noqa: E702 -- opaque, then 2e9 x denser: the second must get weight 0
        dens[2::4, k + 5] = float("inf")
        w_ref = V.get_weights(dens, ray.deltas)
        assert float(w_ref.sum(1).max()) <= 1.0 + 1e-5
        dens_d, sb_d, eb_d = dens.to(dev), ray.s_bins.to(dev).contiguous(), ray.e_bins.to(dev).contiguous()
        if S != 48:
            n_new = 96 if S == 256 else 48
            w = torch.empty((R, S), device=dev)
            s_n, e_n = torch.empty((R, n_new + 1), device=dev), torch.empty((R, n_new + 1), device=dev)
            j = T(rb["jitters"][1]).reshape(-1).to(dev)
            _lib.check(lib.neraf_pdf_resample(_lib.ctx(0), dens_d.data_ptr(), sb_d.data_ptr(), eb_d.data_ptr(), R, S, 1.0, j.data_ptr(), 0, n_new,
                                              0.05, 1000.0, w.data_ptr(), s_n.data_ptr(), e_n.data_ptr(), st))
            np.testing.assert_allclose(w.cpu().numpy(), w_ref.numpy(), rtol=2e-4, atol=1e-7)
            assert bool(torch.isfinite(s_n).all()) and bool((s_n[:, 1:] >= s_n[:, :-1]).all())
        else:
            rgb_s = T(synth.uniform("t.surf.rgb", (R, S, 3), 0.0, 1.0)).to(dev)
            w = torch.empty((R, S), device=dev)
            rgb, depth = torch.empty((R, 3), device=dev), torch.empty((R, 1), device=dev)
            expd, acc = torch.empty((R, 1), device=dev), torch.empty((R, 1), device=dev)
            scratch = torch.empty(2, dtype=torch.int32, device=dev)
            _lib.check(lib.neraf_composite(_lib.ctx(0), dens_d.data_ptr(), rgb_s.data_ptr(), eb_d.data_ptr(), R, S, 1, w.data_ptr(), rgb.data_ptr(),
                                           depth.data_ptr(), expd.data_ptr(), acc.data_ptr(), scratch.data_ptr(), 8, st))
            np.testing.assert_allclose(w.cpu().numpy(), w_ref.numpy(), rtol=2e-4, atol=1e-7)
            assert float(acc.max()) <= 1.0 + 1e-5 and bool(torch.isfinite(rgb).all())


@pytest.mark.parametrize("R", [5, 4096, 32768])
def test_sampler_folded_depth_range_equals_the_standalone_reduction(dev, R):
    """Round 6: the {min, max} step pair that clips the expected depth (DepthRenderer "expected" [NS-recall]) is formed by the sampler
    launches (neraf_pdf_resample_mm: mode 1 seeds + zeroes the loss sums, mode 2 accumulates the new samples' range) instead of two
    launches in front of every composite.  Same bits as the stand-alone reduction (neraf_composite's own seed + minmax launches): the
    pair itself, the words behind it zeroed, and the composite's five outputs through neraf_composite_mm vs neraf_composite."""
    from neraf_amd import _lib
    lib = _lib.load()
    h = _lib.ctx(0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    S0, S1 = 96, 48
    g = torch.Generator(device=dev).manual_seed(R)
    s0, e0 = torch.empty((R, S0 + 1), device=dev), torch.empty((R, S0 + 1), device=dev)
    _lib.check(lib.neraf_sample_uniform(h, R, S0, 0.05, 1000.0, None, 12345, s0.data_ptr(), e0.data_ptr(), st), 0)
    dens0 = (torch.rand((R, S0), generator=g, device=dev) ** 6 * 50.0).contiguous()
    NM = 64 * 64                                                      # 64 replicas of the pair, 256 bytes apart
    scratch = torch.full((NM + 4,), 7.0, device=dev)                  # garbage: the seeding launch owns the initial state
    s1, e1 = torch.empty((R, S1 + 1), device=dev), torch.empty((R, S1 + 1), device=dev)
    _lib.check(lib.neraf_pdf_resample_mm(h, dens0.data_ptr(), s0.data_ptr(), e0.data_ptr(), S0 + 1, R, S0, 1.0, None, 999, S1, 0.05, 1000.0,
                                         None, s1.data_ptr(), e1.data_ptr(), scratch.data_ptr(), (NM + 4) * 4, 1, st), 0)
    seeded = scratch.view(torch.int32).cpu()
    assert seeded[0:NM:64].tolist() == [0x7F7FFFFF] * 64 and seeded[1:NM:64].tolist() == [0] * 64 and seeded[NM:].tolist() == [0] * 4
    _lib.check(lib.neraf_pdf_resample_mm(h, dens0.data_ptr(), s0.data_ptr(), e0.data_ptr(), S0 + 1, R, S0, 1.0, None, 999, S1, 0.05, 1000.0,
                                         None, s1.data_ptr(), e1.data_ptr(), scratch.data_ptr(), (NM + 4) * 4, 2, st), 0)
    mid = 0.5 * (e1[:, :-1] + e1[:, 1:])
    lo_f, hi_f = scratch[0:NM:64].min(), scratch[1:NM:64].max()
    assert float(lo_f) == float(mid[:, 0].min()) and float(hi_f) == float(mid[:, -1].max())
    assert scratch[NM:].cpu().tolist() == [0.0] * 4
    dens = (torch.rand((R, S1), generator=g, device=dev) ** 8 * 3.0).contiguous()      # many rays with almost no mass: the clip acts
    dens[::3] = 0.0
    rgb_s = torch.rand((R, S1, 3), generator=g, device=dev).contiguous()
    outs = []
    for form in ("mm", "standalone"):
        w = torch.empty((R, S1), device=dev)
        rgb, depth = torch.empty((R, 3), device=dev), torch.empty((R, 1), device=dev)
        expd, acc = torch.empty((R, 1), device=dev), torch.empty((R, 1), device=dev)
        if form == "mm":
            _lib.check(lib.neraf_composite_mm(h, dens.data_ptr(), rgb_s.data_ptr(), e1.data_ptr(), R, S1, 0, w.data_ptr(), rgb.data_ptr(),
                                              depth.data_ptr(), expd.data_ptr(), acc.data_ptr(), scratch.data_ptr(), st), 0)
        else:
            own = torch.empty(2, dtype=torch.int32, device=dev)
            _lib.check(lib.neraf_composite(h, dens.data_ptr(), rgb_s.data_ptr(), e1.data_ptr(), R, S1, 0, w.data_ptr(), rgb.data_ptr(),
                                           depth.data_ptr(), expd.data_ptr(), acc.data_ptr(), own.data_ptr(), 8, st), 0)
            assert own.cpu().tolist() == torch.stack([lo_f, hi_f]).view(torch.int32).cpu().tolist()
        outs.append((w, rgb, depth, expd, acc))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    lo, hi = float(lo_f), float(hi_f)
    assert float(outs[0][3].min()) >= lo and float(outs[0][3].max()) <= hi and float((outs[0][3] == lo).sum()) >= R // 3


@pytest.mark.parametrize("mode,training", [("contract", True), ("contract", False), ("aabb", True)])
def test_field_query(dev, setup, mode, training):
    m, P16, spec, V = setup
    f = m.field.module
    R, S = (301, 48) if mode == "contract" else (301, 13)       # 13: a samples-per-ray count that is no power of two and no run of 16
    rb = synth.ray_batch(R, tag="t.fq")
    o, d = T(rb["origins"]), T(rb["directions"])
    cam = T(rb["camera_indices"])
    ray = V.sample_uniform(o, d, torch.full((R, 1), 0.05), torch.full((R, 1), 6.0), S, T(rb["jitters"][0]))
    pos = ray.positions()
    aabb = f.aabb.cpu()
    rgb_ref, den_ref = V.field_forward(pos, d[:, None, :].expand(-1, S, -1), cam[:, None].expand(-1, S), P16, spec,
                                       contract=(mode == "contract"), aabb=aabb, training=training)
    old = f.spatial_distortion
    f.spatial_distortion = "linf" if mode == "contract" else None
    try:
        rgb, den = f.query(o.to(dev), d.to(dev), ray.e_bins.to(dev).contiguous(), cam.to(dev), use_average_embedding=not training)
    finally:
        f.spatial_distortion = old
    np.testing.assert_allclose(den.cpu().numpy(), den_ref.numpy(), rtol=2e-2, atol=1e-7)
    assert float((rgb.cpu() - rgb_ref).abs().max()) <= 4e-3
    if mode == "aabb":
        assert float((den_ref == 0).float().mean()) > 0.05          # some samples fall outside the box -> selector


def test_field_forward_generic_frustums(dev, setup):
    """Field.forward(ray_samples) with zero-length frustums, as the grid refresh calls it (NeRAF_model.py:333-342)."""
    from neraf_amd.vision import Frustums, RaySamples, FieldHeadNames
    m, P16, spec, V = setup
    f = m.field.module
    n = 1000
    pos = T(synth.uniform("t.ff.pos", (n, 3), -0.9, 0.9))
    dirs = T(synth.normal("t.ff.dir", (n, 3)))
    z = torch.zeros(n, 1)
    f.spatial_distortion = None
    f.train()
    try:
        out = f(RaySamples(Frustums(pos.to(dev), dirs.to(dev), z.to(dev), z.to(dev)), torch.zeros((n, 1), dtype=torch.int32, device=dev)))
    finally:
        f.spatial_distortion = "linf"
    rgb_ref, den_ref = V.field_forward(pos, dirs, torch.zeros(n, dtype=torch.long), P16, spec, contract=False, aabb=f.aabb.cpu())
    assert out[FieldHeadNames.RGB].shape == (n, 3) and out[FieldHeadNames.DENSITY].shape == (n, 1)
    np.testing.assert_allclose(out[FieldHeadNames.DENSITY][:, 0].cpu().numpy(), den_ref.numpy(), rtol=2e-2, atol=1e-7)
    assert float((out[FieldHeadNames.RGB].cpu() - rgb_ref).abs().max()) <= 4e-3


@pytest.mark.parametrize("training", [True, False])
def test_full_forward_vs_oracle(dev, setup, training):
    from neraf_amd.vision import RayBundle
    m, P16, spec, V = setup
    R = 512
    rb = synth.ray_batch(R, tag="t.full")
    o, d, cam = T(rb["origins"]), T(rb["directions"]), T(rb["camera_indices"])
    jit = [T(j) for j in rb["jitters"]]
    ref = V.nerfacto_forward(o, d, cam, P16, spec, step=300, training=training, jitters=jit)
    m.train(training)
    m.update_to_step(300)
    out = m.get_outputs(RayBundle(o.to(dev), d.to(dev), cam.to(dev)), jitters=[j.to(dev) for j in jit] if training else None)
    m.train(True)
    assert float((out["rgb"].cpu() - ref["rgb"]).abs().max()) <= 5e-3
    assert float((out["accumulation"].cpu() - ref["accumulation"]).abs().max()) <= 5e-3
    rel = ((out["expected_depth"].cpu() - ref["expected_depth"]).abs() / ref["expected_depth"])
    assert float(rel.median()) <= 1e-3
    # median depth is an order statistic: it may flip to the neighbouring sample on a few rays
    agree = ((out["depth"].cpu() - ref["depth"]).abs() / ref["depth"] < 1e-3).float().mean()
    assert float(agree) >= 0.97
    if training:
        for w, wr in zip(out["weights_list"], ref["weights_list"]):
            assert float((w.cpu() - wr).abs().max()) <= 5e-3


@pytest.mark.parametrize("H,W", [(24, 32), (12, 64), (5, 40)])
def test_coherent_lane_orders_are_bit_identical(dev, setup, H, W):
    """coherent_rays only re-assigns samples to lanes (include/neraf_hip.h, neraf_proposal_density): 0 (ray-major), 1 (row segments of
    64 / 16 rays) and the image width (8 x 8 / 4 x 4 pixel tiles; 12 and 5 rows leave partial tiles) give the same bits for a small
    pinhole frame -- every sampler stage, both gather kernels, the composite."""
    from neraf_amd.vision import RayBundle
    m, P16, spec, V = setup
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    d = torch.stack([(xs - W / 2) / W, -(ys - H / 2) / W, -torch.ones_like(xs)], -1).reshape(-1, 3)
    d = d / d.norm(dim=-1, keepdim=True)
    o = torch.tensor([0.1, -0.05, 0.3]).expand(H * W, 3).contiguous()
    rb = RayBundle(o.to(dev), d.to(dev), torch.zeros((H * W, 1), dtype=torch.long, device=dev))
    m.eval()
    try:
        with torch.no_grad():
            outs = [m.get_outputs(rb, coherent_rays=c) for c in (0, 1, W)]
    finally:
        m.train(True)
    for k in ("rgb", "depth", "expected_depth", "accumulation"):
        assert torch.equal(outs[0][k], outs[1][k]), k
        assert torch.equal(outs[0][k], outs[2][k]), k
    assert float(outs[0]["accumulation"].max()) > 0


def test_full_size_batch_invariants(dev, setup):
    """BASELINE size: 4096 rays (NeRAF_config.py:87).  Size-independent properties instead of a CPU oracle run."""
    from neraf_amd.vision import RayBundle
    m, _, _, _ = setup
    R = 4096
    rb = synth.ray_batch(R, tag="t.big")
    m.train(True)
    out = m.get_outputs(RayBundle(T(rb["origins"]).to(dev), T(rb["directions"]).to(dev), T(rb["camera_indices"]).to(dev)))
    for w, rs in zip(out["weights_list"], out["ray_samples_list"]):
        assert bool(torch.isfinite(w).all()) and float(w.min()) >= 0 and float(w.sum(-1).max()) <= 1 + 1e-4
        assert bool((rs.e_bins[:, 1:] >= rs.e_bins[:, :-1]).all())
    assert float(out["rgb"].min()) >= 0 and float(out["rgb"].max()) <= 1
    np.testing.assert_allclose(out["accumulation"][:, 0].cpu().numpy(), out["weights_list"][-1].sum(-1).cpu().numpy(), rtol=1e-5, atol=1e-6)
    # linearity of the composite in colour: rendering with the same weights is affine in rgb_samples
    m.train(False)
    o2 = m.get_outputs_for_camera_ray_bundle(RayBundle(T(rb["origins"]).to(dev), T(rb["directions"]).to(dev), None))
    assert o2["rgb"].shape == (R, 3)
    m.train(True)

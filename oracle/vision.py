"""CPU ORACLE -- radiance half of the NeRAF hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  ``NeRAFVisionModel`` (NeRAF_model.py:54-79) inherits every line of its arithmetic
from nerfstudio's ``NerfactoModel`` / ``NerfactoField`` / ``ProposalNetworkSampler`` and from
tiny-cuda-nn's HashGrid / FullyFusedMLP / SphericalHarmonics.  Neither package is under
/root/reference nor installed here, and the reference has no tests or golden vectors, so this file
restates their *published* algorithms from recall (nerfstudio >= 0.3.0 as pinned by the reference's
pyproject.toml:6, tiny-cuda-nn 1.7 per README.md:45) and is guarded by property tests only
(tests/test_oracle_vision.py).  The reference call sites that anchor it:
NeRAF_model.py:54-60 (subclass), :65-68 (get_outputs + clip), :70-79 (get_outputs_for_camera),
:302-350 (field.forward on refresh frustums, renderer_rgb), NeRAF_config.py:94-98 (overrides:
eval_num_rays_per_chunk=32768, average_init_density=0.01, camera optimizer SO3xR3).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field as dc_field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .audio import sh4_encoding

Tensor = torch.Tensor
PRIMES = (1, 2654435761, 805459861)

# Precision-sensitivity probe (oracle/trainer.py ``OracleTrainer(probe=...)``; None = plain fp32): applied to the interpolated
# hash encodings and to every hidden / output activation of the fused MLPs -- the places where tiny-cuda-nn's half-precision
# FullyFusedMLP / HashGrid (and the HIP kernels) store 16-bit values.
PROBE_ACT = None
# Second probe: the GRADIENT arriving at the interpolated hash encodings rounded the way an fp16 backward chain stores it -- scaled by
# a power of two that puts the tensor's amax at 2^12, rounded to fp16 (values below 2^-24 of that flush to ZERO), un-scaled.  With
# Adam at eps = 1e-15 this is not a rounding-sized effect: a table row that only ever receives gradients below the flush threshold
# takes full lr-sized steps in fp32 and none at all in fp16 (tiny-cuda-nn's half-precision backward and the HIP engine alike).
PROBE_ENC_GRAD = None


def _pa(t: Tensor) -> Tensor:
    return PROBE_ACT(t) if PROBE_ACT is not None else t


class _EncGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return PROBE_ENC_GRAD(g) if PROBE_ENC_GRAD is not None else g


def fp16_scaled_round(g: Tensor) -> Tensor:
    amax = float(g.abs().max())
    if not (amax > 0.0 and math.isfinite(amax)):
        return g
    s = 2.0 ** (12 - math.floor(math.log2(amax)))
    return (g * s).half().float() / s


# ---------------------------------------------------------------------------
# tiny-cuda-nn multiresolution hash grid  [TCNN-recall]
# ---------------------------------------------------------------------------
@dataclass
class GridSpec:
    n_levels: int
    base_res: int
    max_res: int
    log2_hashmap_size: int
    n_features: int = 2
    scales: List[float] = dc_field(default_factory=list)
    resolutions: List[int] = dc_field(default_factory=list)
    sizes: List[int] = dc_field(default_factory=list)
    offsets: List[int] = dc_field(default_factory=list)

    def __post_init__(self):
        growth = math.exp(math.log(self.max_res / self.base_res) / (self.n_levels - 1)) if self.n_levels > 1 else 1.0
        log2_g = np.float32(math.log2(growth))
        off = 0
        T = 1 << self.log2_hashmap_size
        for l in range(self.n_levels):
            # grid_scale(): exp2f(level * log2_per_level_scale) * base_resolution - 1.0f   (float32 arithmetic)
            scale = np.float32(np.exp2(np.float32(l) * log2_g)) * np.float32(self.base_res) - np.float32(1.0)
            res = int(np.ceil(scale)) + 1                       # grid_resolution()
            n = res ** 3
            n = (n + 7) // 8 * 8                                # next_multiple(params_in_level, 8)
            n = min(n, T)
            self.scales.append(float(scale))
            self.resolutions.append(res)
            self.sizes.append(n)
            self.offsets.append(off)
            off += n
        self.offsets.append(off)

    @property
    def total(self) -> int:
        return self.offsets[-1]


def hash_encode(x01: Tensor, table: Tensor, spec: GridSpec) -> Tensor:
    """HashGrid forward, trilinear ("Linear" interpolation).  x01 [N,3] in [0,1]; table [total, F]."""
    outs = []
    N = x01.shape[0]
    for l in range(spec.n_levels):
        scale, res, size, off = spec.scales[l], spec.resolutions[l], spec.sizes[l], spec.offsets[l]
        pos = x01.float() * scale + 0.5                          # fmaf(scale, x, 0.5)
        fl = torch.floor(pos)
        frac = pos - fl
        g0 = fl.to(torch.int64)
        acc = torch.zeros((N, spec.n_features), dtype=table.dtype)
        hashed = size < res ** 3                                 # stride after 3 dims exceeds the level size
        for corner in range(8):
            w = torch.ones(N, dtype=frac.dtype)
            g = []
            for d in range(3):
                bit = (corner >> d) & 1
                w = w * (frac[:, d] if bit else (1.0 - frac[:, d]))
                g.append(g0[:, d] + bit)
            if hashed:
                idx = torch.zeros(N, dtype=torch.int64)
                for d in range(3):
                    idx = idx ^ ((g[d] * PRIMES[d]) & 0xFFFFFFFF)   # uint32 wrap-around multiply
            else:
                idx = g[0] + g[1] * res + g[2] * res * res
            idx = (idx & 0xFFFFFFFF) % size
            acc = acc + w[:, None].to(table.dtype) * table[off + idx]
        outs.append(acc)
    enc = _pa(torch.cat(outs, dim=-1))
    return _EncGrad.apply(enc) if PROBE_ENC_GRAD is not None else enc


def tcnn_mlp(x: Tensor, weights: List[Tensor]) -> Tensor:
    """FullyFusedMLP: bias-free Linear layers, ReLU between, no output activation.  weights[i] is [out,in];
    the input is zero-padded up to weights[0].shape[1] (tcnn pads inputs to a multiple of 16)."""
    if x.shape[-1] < weights[0].shape[1]:
        x = F.pad(x, (0, weights[0].shape[1] - x.shape[-1]))
    for i, w in enumerate(weights):
        x = x @ w.t()
        if i + 1 < len(weights):
            x = F.relu(x)
        x = _pa(x)
    return x


# ---------------------------------------------------------------------------
# nerfstudio pieces  [NS-recall]
# ---------------------------------------------------------------------------
def contract_linf(x: Tensor) -> Tensor:
    """SceneContraction(order=inf): x if |x|_inf <= 1 else (2 - 1/|x|) (x/|x|)."""
    mag = x.abs().amax(dim=-1, keepdim=True)
    safe = torch.clamp(mag, min=1e-30)
    return torch.where(mag < 1, x, (2 - 1 / safe) * (x / safe))


def spacing_fn(x: Tensor) -> Tensor:           # UniformLinDispPiecewiseSampler
    return torch.where(x < 1, x / 2, 1 - 1 / (2 * x))


def spacing_fn_inv(x: Tensor) -> Tensor:
    return torch.where(x < 0.5, 2 * x, 1 / (2 - 2 * x))


@dataclass
class RaySamples:
    origins: Tensor          # [R,3]
    directions: Tensor       # [R,3]
    s_bins: Tensor           # [R,S+1] normalised-spacing bin edges
    e_bins: Tensor           # [R,S+1] euclidean bin edges
    camera_indices: Optional[Tensor] = None   # [R]

    @property
    def starts(self): return self.e_bins[:, :-1]
    @property
    def ends(self): return self.e_bins[:, 1:]
    @property
    def deltas(self): return self.ends - self.starts

    def positions(self) -> Tensor:             # Frustums.get_positions: o + d * (start+end)/2
        mid = (self.starts + self.ends) / 2
        return self.origins[:, None, :] + self.directions[:, None, :] * mid[..., None]


def s_to_euclid(s: Tensor, near: Tensor, far: Tensor) -> Tensor:
    s_near, s_far = spacing_fn(near), spacing_fn(far)
    return spacing_fn_inv(s * s_far + (1 - s) * s_near)


def sample_uniform(origins, directions, near, far, n: int, jitter: Optional[Tensor], cam=None) -> RaySamples:
    """SpacedSampler.generate_ray_samples with single_jitter (jitter [R,1] in [0,1) or None = eval)."""
    R = origins.shape[0]
    bins = torch.linspace(0.0, 1.0, n + 1)[None, :].expand(R, -1)
    if jitter is not None:
        centers = (bins[:, 1:] + bins[:, :-1]) / 2.0
        upper = torch.cat([centers, bins[:, -1:]], -1)
        lower = torch.cat([bins[:, :1], centers], -1)
        bins = lower + (upper - lower) * jitter
    e = s_to_euclid(bins, near, far)
    return RaySamples(origins, directions, bins, e, cam)


def get_weights(density: Tensor, deltas: Tensor) -> Tensor:
    """RaySamples.get_weights: alpha = 1-exp(-sigma delta); T = exp(-exclusive cumsum); w = alpha T."""
    dd = deltas * density
    alphas = 1 - torch.exp(-dd)
    trans = torch.cumsum(dd[:, :-1], dim=-1)
    trans = torch.cat([torch.zeros_like(trans[:, :1]), trans], dim=-1)
    w = alphas * torch.exp(-trans)
    return torch.nan_to_num(w)


def sample_pdf(prev: RaySamples, weights: Tensor, n: int, near, far, jitter: Optional[Tensor],
               histogram_padding: float = 0.01, eps: float = 1e-5) -> RaySamples:
    """PDFSampler.generate_ray_samples (include_original=False, single_jitter)."""
    R = weights.shape[0]
    num_bins = n + 1
    w = weights + histogram_padding
    wsum = w.sum(-1, keepdim=True)
    padding = torch.relu(eps - wsum)
    w = w + padding / w.shape[-1]
    wsum = wsum + padding
    pdf = w / wsum
    cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], dim=-1)
    if jitter is not None:
        u = torch.linspace(0.0, 1.0 - 1.0 / num_bins, num_bins)[None, :].expand(R, -1)
        u = u + jitter / num_bins
    else:
        u = torch.linspace(0.0, 1.0 - 1.0 / num_bins, num_bins) + 1.0 / (2 * num_bins)
        u = u[None, :].expand(R, -1)
    u = u.contiguous()
    existing = prev.s_bins
    inds = torch.searchsorted(cdf.contiguous(), u, side="right")
    below = torch.clamp(inds - 1, 0, existing.shape[-1] - 1)
    above = torch.clamp(inds, 0, existing.shape[-1] - 1)
    cdf_g0, bins_g0 = torch.gather(cdf, -1, below), torch.gather(existing, -1, below)
    cdf_g1, bins_g1 = torch.gather(cdf, -1, above), torch.gather(existing, -1, above)
    t = torch.clip(torch.nan_to_num((u - cdf_g0) / (cdf_g1 - cdf_g0), 0), 0, 1)
    bins = (bins_g0 + t * (bins_g1 - bins_g0)).detach()
    return RaySamples(prev.origins, prev.directions, bins, s_to_euclid(bins, near, far), prev.camera_indices)


# ---------------------------------------------------------------------------
# model parameters and forward
# ---------------------------------------------------------------------------
@dataclass
class NerfactoSpec:
    """nerfacto defaults [NS-recall] with the NeRAF overrides (NeRAF_config.py:94-98)."""
    num_proposal_samples: Tuple[int, int] = (256, 96)
    num_nerf_samples: int = 48
    near: float = 0.05
    far: float = 1000.0
    average_init_density: float = 0.01
    prop_grids: Tuple[GridSpec, GridSpec] = dc_field(default_factory=lambda: (GridSpec(5, 16, 128, 17), GridSpec(5, 16, 256, 17)))
    main_grid: GridSpec = dc_field(default_factory=lambda: GridSpec(16, 16, 2048, 19))
    hidden_dim: int = 64
    geo_feat_dim: int = 15
    appearance_dim: int = 32
    prop_hidden: int = 16
    anneal_slope: float = 10.0
    anneal_iters: int = 1000
    interlevel_mult: float = 1.0
    distortion_mult: float = 0.002


class _TruncExp(torch.autograd.Function):
    """nerfstudio's ``trunc_exp`` [NS-recall: field_components/activations.py]: forward ``exp(x)``, backward
    ``g * exp(clamp(x, -15, 15))`` -- the density activation of NerfactoField and HashMLPDensityField."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


def trunc_exp(x: Tensor) -> Tensor:
    return _TruncExp.apply(x)


def anneal_value(step: int, spec: NerfactoSpec) -> float:
    x = float(np.clip(step / spec.anneal_iters, 0, 1))
    b = spec.anneal_slope
    return (b * x) / ((b - 1) * x + 1)


def proposal_density(pos: Tensor, P: Dict[str, Tensor], i: int, spec: NerfactoSpec, contract: bool = True) -> Tensor:
    """HashMLPDensityField.density_fn: contraction -> (x+2)/4 -> selector -> hash+MLP(10->16->1) -> avg*exp."""
    shp = pos.shape[:-1]
    x = pos.reshape(-1, 3)
    x = (contract_linf(x) + 2.0) / 4.0 if contract else x
    sel = ((x > 0.0) & (x < 1.0)).all(-1)
    x = x * sel[:, None]
    enc = hash_encode(x, P[f"prop{i}.table"], spec.prop_grids[i])
    out = tcnn_mlp(enc, [P[f"prop{i}.w0"], P[f"prop{i}.w1"]])[:, :1]
    dens = spec.average_init_density * trunc_exp(out) * sel[:, None]
    return dens.reshape(*shp)


def field_forward(pos: Tensor, dirs: Tensor, cam_idx: Optional[Tensor], P: Dict[str, Tensor], spec: NerfactoSpec,
                  contract: bool = True, aabb: Optional[Tensor] = None, training: bool = True):
    """NerfactoField.get_density + get_outputs.  pos/dirs [...,3].  With ``contract=False`` positions are
    normalised by ``aabb`` instead (spatial_distortion=None, as set for the grid refresh NeRAF_model.py:302)."""
    shp = pos.shape[:-1]
    x = pos.reshape(-1, 3)
    if contract:
        x = (contract_linf(x) + 2.0) / 4.0
    else:
        x = (x - aabb[0]) / (aabb[1] - aabb[0])
    sel = ((x > 0.0) & (x < 1.0)).all(-1)
    x = x * sel[:, None]
    enc = hash_encode(x, P["field.table"], spec.main_grid)
    h = tcnn_mlp(enc, [P["field.base_w0"], P["field.base_w1"]])          # [N,16]
    density = spec.average_init_density * trunc_exp(h[:, :1]) * sel[:, None]
    geo = h[:, 1:1 + spec.geo_feat_dim]
    d = sh4_encoding((dirs.reshape(-1, 3) + 1.0) / 2.0)
    if training:
        emb = P["field.embedding"][cam_idx.reshape(-1).long()]
    else:
        emb = P["field.embedding"].mean(0)[None, :].expand(x.shape[0], -1)   # use_average_appearance_embedding
    hin = torch.cat([d, geo, emb], dim=-1)                                # 16 + 15 + 32 = 63
    rgb = torch.sigmoid(tcnn_mlp(hin, [P["field.head_w0"], P["field.head_w1"], P["field.head_w2"]])[:, :3])
    return rgb.reshape(*shp, 3), density.reshape(*shp)


def render(ray: RaySamples, rgb: Tensor, weights: Tensor, training: bool):
    """RGBRenderer('last_sample') + DepthRenderer(median / expected) + AccumulationRenderer."""
    acc = weights.sum(-1, keepdim=True)
    comp = (weights[..., None] * rgb).sum(-2) + rgb[:, -1, :] * (1.0 - acc)
    if not training:
        comp = torch.clamp(comp, 0.0, 1.0)
    steps = (ray.starts + ray.ends) / 2
    cum = torch.cumsum(weights, dim=-1)
    idx = torch.searchsorted(cum.contiguous(), torch.full_like(cum[:, :1], 0.5), side="left")
    idx = torch.clamp(idx, 0, steps.shape[-1] - 1)
    median = torch.gather(steps, -1, idx)
    expected = (weights * steps).sum(-1, keepdim=True) / (acc + 1e-10)
    expected = torch.clip(expected, steps.min(), steps.max())
    return comp, median, expected, acc


def nerfacto_forward(origins, directions, cam_idx, P, spec: NerfactoSpec, step: int = 100000, training: bool = True,
                     jitters: Optional[List[Tensor]] = None):
    """NerfactoModel.get_outputs followed by NeRAFVisionModel's clip (NeRAF_model.py:65-68)."""
    R = origins.shape[0]
    near = torch.full((R, 1), spec.near)
    far = torch.full((R, 1), spec.far)
    jit = jitters if (training and jitters is not None) else [None, None, None]
    anneal = anneal_value(step, spec) if training else 1.0
    weights_list, samples_list = [], []
    ray = sample_uniform(origins, directions, near, far, spec.num_proposal_samples[0], jit[0], cam_idx)
    for i in range(2):
        dens = proposal_density(ray.positions(), P, i, spec)
        w = get_weights(dens, ray.deltas)
        weights_list.append(w)
        samples_list.append(ray)
        n_next = spec.num_proposal_samples[1] if i == 0 else spec.num_nerf_samples
        ray = sample_pdf(ray, torch.pow(w.detach(), anneal), n_next, near, far, jit[i + 1])
    S = spec.num_nerf_samples
    pos = ray.positions()
    dirs = directions[:, None, :].expand(-1, S, -1)
    cams = cam_idx[:, None].expand(-1, S) if cam_idx is not None else None
    rgb_s, dens = field_forward(pos, dirs, cams, P, spec, training=training)
    w = get_weights(dens, ray.deltas)
    weights_list.append(w)
    samples_list.append(ray)
    rgb, depth, expected, acc = render(ray, rgb_s, w, training)
    out = {"rgb": torch.clip(rgb, 0, 1), "depth": depth, "expected_depth": expected, "accumulation": acc,
           "weights_list": weights_list, "ray_samples_list": samples_list, "rgb_samples": rgb_s, "density": dens}
    return out


# ---------------------------------------------------------------------------
# losses (V4)  [NS-recall]
# ---------------------------------------------------------------------------
def _outer(t0_starts, t0_ends, t1_starts, t1_ends, y1):
    cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
    idx_lo = torch.searchsorted(t1_starts.contiguous(), t0_starts.contiguous(), side="right") - 1
    idx_lo = torch.clamp(idx_lo, min=0, max=y1.shape[-1] - 1)
    idx_hi = torch.searchsorted(t1_ends.contiguous(), t0_ends.contiguous(), side="right")
    idx_hi = torch.clamp(idx_hi, min=0, max=y1.shape[-1] - 1)
    cy1_lo = torch.take_along_dim(cy1[..., :-1], idx_lo, dim=-1)
    cy1_hi = torch.take_along_dim(cy1[..., 1:], idx_hi, dim=-1)
    return cy1_hi - cy1_lo


def lossfun_outer(t, w, t_env, w_env, eps: float = 1e-7):
    w_outer = _outer(t[..., :-1], t[..., 1:], t_env[..., :-1], t_env[..., 1:], w_env)
    return torch.clip(w - w_outer, min=0) ** 2 / (w + eps)


def interlevel_loss(weights_list, samples_list):
    c = samples_list[-1].s_bins.detach()
    w = weights_list[-1].detach()
    loss = 0.0
    for ray, wp in zip(samples_list[:-1], weights_list[:-1]):
        loss = loss + torch.mean(lossfun_outer(c, w, ray.s_bins, wp))
    return loss


def lossfun_distortion(t, w):
    ut = (t[..., 1:] + t[..., :-1]) / 2
    dut = torch.abs(ut[..., :, None] - ut[..., None, :])
    loss_inter = torch.sum(w * torch.sum(w[..., None, :] * dut, dim=-1), dim=-1)
    loss_intra = torch.sum(w ** 2 * (t[..., 1:] - t[..., :-1]), dim=-1) / 3
    return loss_inter + loss_intra


def distortion_loss(weights_list, samples_list):
    return torch.mean(lossfun_distortion(samples_list[-1].s_bins, weights_list[-1]))


def vision_loss_dict(out, gt_rgb, spec: NerfactoSpec):
    d = {"rgb_loss": F.mse_loss(gt_rgb, out["rgb"])}
    d["interlevel_loss"] = spec.interlevel_mult * interlevel_loss(out["weights_list"], out["ray_samples_list"])
    d["distortion_loss"] = spec.distortion_mult * distortion_loss(out["weights_list"], out["ray_samples_list"])
    return d

"""Host-side mirror of the radiance half of the reference (``NeRAFVisionModel``, NeRAF_model.py:54-79,
and the field it wraps, NeRAF_field.py:27-34) on libneraf_hip.

The reference inherits this half from nerfstudio's ``NerfactoModel`` and runs it on tiny-cuda-nn; here
the same stages are hand-written HIP kernels behind the C ABI (include/neraf_hip.h, "Radiance half"):
piecewise sampler -> proposal density x2 -> PDF resampling x2 -> fused field query -> composite.

Training: ``get_outputs`` (forward) + ``get_loss_dict`` (rgb MSE, interlevel x1.0, distortion x0.002 -- the
nerfacto loss dict, V4) are differentiable end to end: the three losses come out of one autograd node
(``_VisionLossFn``) whose backward runs the HIP loss-gradient, proposal-backward and fused field-backward
kernels and returns gradients for the hash tables, the MLP weights and the appearance embedding.  The audio
loss reaches the same parameters through the voxel grid (``model._RefreshFn``, NeRAF_model.py:395-400).  With the
camera optimizer on (neraf_amd/cameras.py, NeRAF_config.py:97) the same node also returns d loss / d (ray origin,
ray direction) -- hash-grid input gradient through the contraction, SH input gradient -- which autograd carries
into the SO3xR3 pose deltas.

No fallback: every method raises if the HIP library or the GPU is missing.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import os

import torch
import torch.nn as nn

from . import _lib
from .field import _dev_index, _stream_ptr


# ---------------------------------------------------------------------------------------------------
@dataclass
class RayBundle:
    """The fields of nerfstudio's RayBundle that the hot path reads."""
    origins: torch.Tensor            # [R,3] fp32
    directions: torch.Tensor         # [R,3] fp32 (unit)
    camera_indices: Optional[torch.Tensor] = None   # [R] or [R,1] int
    nears: Optional[torch.Tensor] = None
    fars: Optional[torch.Tensor] = None

    def __len__(self):
        return self.origins.shape[0]

    def camera_indices_i32(self) -> Optional[torch.Tensor]:
        """``camera_indices`` as the flat int32 vector the kernels index with, converted ONCE per bundle (nerfstudio hands over int64
        [R,1]); a resident bundle (bench, tests) or one that already carries int32 indices costs no launch at all."""
        ci = self.camera_indices
        if ci is None:
            return None
        key = (ci.data_ptr(), ci._version, ci.dtype)
        c = getattr(self, "_cam_i32", None)
        if c is None or c[0] != key:
            c = (key, ci.reshape(-1).to(torch.int32).contiguous())
            self._cam_i32 = c
        return c[1]


@dataclass
class Frustums:
    origins: torch.Tensor
    directions: torch.Tensor
    starts: torch.Tensor
    ends: torch.Tensor
    pixel_area: Optional[torch.Tensor] = None

    def get_positions(self):
        return self.origins + self.directions * (self.starts + self.ends) / 2


@dataclass
class RaySamples:
    """Structured samples of R rays: S+1 bin edges per ray in normalised spacing and euclidean distance."""
    frustums: Optional[Frustums]
    camera_indices: Optional[torch.Tensor] = None
    s_bins: Optional[torch.Tensor] = None     # [R,S+1]
    e_bins: Optional[torch.Tensor] = None     # [R,S+1]

    def to(self, device):
        return self


class FieldHeadNames:
    RGB = "rgb"
    DENSITY = "density"


def grid_layout(desc: _lib.GridDesc):
    lib = _lib.load()
    L = desc.n_levels
    sc, rs = (C.c_float * L)(), (C.c_int * L)()
    sz, off = (C.c_uint32 * L)(), (C.c_uint32 * (L + 1))()
    rc = lib.neraf_grid_layout(C.byref(desc), sc, rs, sz, off)
    if rc != 0:
        raise ValueError("unsupported hash-grid descriptor")
    return list(sc), list(rs), list(sz), list(off)


def hash_encode(desc: _lib.GridDesc, table: torch.Tensor, x01: torch.Tensor) -> torch.Tensor:
    """tiny-cuda-nn's HashGrid forward on its own (``neraf_hash_encode``): positions ``x01`` [N,3] in [0,1]^3 and a table [rows,2]
    (rounded to fp16 as the kernels read it) -> [N, 2 * n_levels] fp32.  The fused field / proposal kernels contain the same
    arithmetic; this entry exists for callers that want the encoding itself (SURVEY 8b op list) and as its stand-alone parity point."""
    lib = _lib.load()
    dev = _dev_index(x01)
    x = x01.reshape(-1, 3).float().contiguous()
    t16 = table.detach().half().contiguous()
    out = torch.empty((x.shape[0], 2 * desc.n_levels), dtype=torch.float32, device=x.device)
    _lib.check(lib.neraf_hash_encode(_lib.ctx(dev), C.byref(desc), t16.data_ptr(), x.data_ptr(), x.shape[0], out.data_ptr(), _stream_ptr()), dev)
    return out


# ---- MFMA weight-fragment packing of the field MLPs ----------------------------------------------------
def _dperm(s: int, q: int, j: int) -> int:
    """k index held by (lane quarter q, element j) of k-step s when the B operand is built from two
    accumulator blocks of the previous layer (see csrc/field.hip header)."""
    return 16 * (2 * s) + 4 * q + j if j < 4 else 16 * (2 * s + 1) + 4 * q + (j - 4)


def _build_frag_index() -> np.ndarray:
    sizes = {"b0": (64, 32), "b1": (16, 64), "h0": (64, 64), "h1": (64, 64), "h2": (16, 64)}
    base, off = {}, 0
    for k, (o, i) in sizes.items():
        base[k] = off
        off += o * i
    ZERO = off
    idx = np.full((24, 64, 8), ZERO, np.int64)

    def put(f, l, j, name, out, col):
        idx[f, l, j] = base[name] + out * sizes[name][1] + col

    for l in range(64):
        row, q = l & 15, l >> 4
        for j in range(8):
            for ob in range(4):
                put(ob, l, j, "b0", 16 * ob + row, 8 * q + j)
            for s in range(2):
                put(4 + s, l, j, "b1", row, _dperm(s, q, j))
                put(22 + s, l, j, "h2", row, _dperm(s, q, j))
                for ob in range(4):
                    put(14 + ob * 2 + s, l, j, "h1", 16 * ob + row, _dperm(s, q, j))
            for ob in range(4):
                out = 16 * ob + row
                # k-step 0: [base output 4q+j (geo feature 4q+j-1; the density logit has no weight) | SH 4q+j-4]
                if j < 4:
                    t = 4 * q + j
                    if t >= 1:
                        put(6 + ob * 2, l, j, "h0", out, 16 + (t - 1))
                else:
                    put(6 + ob * 2, l, j, "h0", out, 4 * q + (j - 4))
                # k-step 1: appearance embedding 8q+j -> column 31 + 8q + j of [SH16 | geo15 | emb32]
                put(7 + ob * 2, l, j, "h0", out, 31 + 8 * q + j)
    return idx.reshape(-1)


_FRAG_INDEX = _build_frag_index()


def _build_frag_index_bwd() -> np.ndarray:
    """Transposed-weight fragments of the backward chain (csrc/field_bwd.hip): the MFMA A operand is W^T restricted to
    16 input rows, the B operand is built from accumulator blocks of dY exactly as in the forward."""
    sizes = {"b0": (64, 32), "b1": (16, 64), "h0": (64, 64), "h1": (64, 64), "h2": (16, 64)}
    base, off = {}, 0
    for k, (o, i) in sizes.items():
        base[k] = off
        off += o * i
    ZERO = off
    idx = np.full((28, 64, 8), ZERO, np.int64)

    def W(name, out, col):
        return base[name] + out * sizes[name][1] + col

    for l in range(64):
        rho, q = l & 15, l >> 4
        for j in range(8):
            for ib in range(4):
                if j < 4:                                   # 16-output layers: k = out 4q+j, upper half of the k-step is zero
                    idx[0 + ib, l, j] = W("h2", 4 * q + j, 16 * ib + rho)
                    idx[18 + ib, l, j] = W("b1", 4 * q + j, 16 * ib + rho)
                for s in range(2):
                    idx[4 + ib * 2 + s, l, j] = W("h1", _dperm(s, q, j), 16 * ib + rho)
            for s in range(2):
                if rho >= 1:                                # row t = base output index; t = 0 is the density logit (no weight)
                    idx[12 + s, l, j] = W("h0", _dperm(s, q, j), 15 + rho)
                for ib in range(2):
                    idx[14 + ib * 2 + s, l, j] = W("h0", _dperm(s, q, j), 31 + 16 * ib + rho)
                for rb in range(2):                          # rows permuted: lane quarter q' = rho>>2 receives features 8q'..8q'+7
                    feat = 8 * (rho >> 2) + 4 * rb + (rho & 3)
                    idx[22 + rb * 2 + s, l, j] = W("b0", _dperm(s, q, j), feat)
                idx[26 + s, l, j] = W("h0", _dperm(s, q, j), rho)   # SH input columns 0..15 of head layer 0 (ray-direction gradient)
    return idx.reshape(-1)


_FRAG_INDEX_BWD = _build_frag_index_bwd()


def _param_key(params):
    """Identity of a parameter state: the fused optimizer's update counter (its kernels write through raw pointers and do not
    touch tensor version counters) plus storage pointer and version counter of every tensor (torch-side updates)."""
    from . import optim
    return (optim.UPDATE_EPOCH,) + tuple((p.data_ptr(), p._version) for p in params)


class HashMLPDensityField(nn.Module):
    """Proposal network: 5-level hash grid + MLP(10 -> 16 -> 1), tcnn layout (bias-free, padded to 16)."""

    def __init__(self, max_res: int, log2_hashmap_size: int = 17, num_levels: int = 5, base_res: int = 16,
                 average_init_density: float = 0.01):
        super().__init__()
        self.desc = _lib.GridDesc(num_levels, base_res, max_res, log2_hashmap_size, 2)
        _, _, _, off = grid_layout(self.desc)
        self.table = nn.Parameter(torch.empty(off[-1], 2).uniform_(-1e-4, 1e-4))
        self.w0 = nn.Parameter(torch.empty(16, 16).uniform_(-0.43, 0.43))
        self.w1 = nn.Parameter(torch.empty(16, 16).uniform_(-0.43, 0.43))
        self.average_init_density = average_init_density

    def acc_scratch(self, device) -> torch.Tensor:
        """Persistent fixed-point accumulator of the table gradient, zero between calls (see NerfactoField.acc_scratch)."""
        a = getattr(self, "_acc_scratch", None)
        if a is None or a.device != torch.device(device) or a.numel() != self.table.shape[0]:
            a = torch.zeros(self.table.shape[0], dtype=torch.int64, device=device)
            self._acc_scratch = a
        return a

    def packed(self):
        """fp16 copies for the kernels, cached until the parameters change (see ``_param_key``)."""
        key = _param_key((self.table, self.w0, self.w1))
        c = getattr(self, "_pack_cache", None)
        if c is None or c[0] != key:
            c = (key, (self.table.detach().half().contiguous(),
                       torch.cat([self.w0.detach().reshape(-1), self.w1.detach()[0]]).half().contiguous()))
            self._pack_cache = c
        return c[1]

    def density(self, origins, directions, e_bins, packed=None, coherent_rays: int = 0):
        """``e_bins`` [R, S+1], or [1, S+1] when every ray shares the same edges (the first sampler stage without jitter)."""
        lib = _lib.load()
        dev = _dev_index(origins)
        R, S = origins.shape[0], e_bins.shape[1] - 1
        shared = e_bins.shape[0] == 1 and R > 1
        tab, w = packed if packed is not None else self.packed()
        out = torch.empty((R, S), dtype=torch.float32, device=origins.device)
        _lib.check(lib.neraf_proposal_density_ex(_lib.ctx(dev), C.byref(self.desc), tab.data_ptr(), w.data_ptr(),
                                                 origins.data_ptr(), directions.data_ptr(), e_bins.data_ptr(), 0 if shared else S + 1, R, S,
                                                 self.average_init_density, int(coherent_rays), out.data_ptr(), _stream_ptr()), dev)
        return out


class NerfactoField(nn.Module):
    """The nerfacto radiance field: 16-level hash grid -> MLP(32->64->16) -> density / geo features;
    SH(dir) + geo + appearance embedding -> MLP(63->64->64->3) -> sigmoid.  ``forward`` keeps the
    ``Field`` signature the reference calls (NeRAF_model.py:339-342)."""

    def __init__(self, aabb: torch.Tensor, num_images: int, average_init_density: float = 0.01, spatial_distortion="linf"):
        super().__init__()
        self.register_buffer("aabb", aabb.float())
        self.spatial_distortion = spatial_distortion          # "linf" scene contraction, or None (NeRAF_model.py:302)
        self.desc = _lib.GridDesc(16, 16, 2048, 19, 2)
        _, _, _, off = grid_layout(self.desc)
        self.table = nn.Parameter(torch.empty(off[-1], 2).uniform_(-1e-4, 1e-4))

        def xav(o, i):
            b = (6.0 / (o + i)) ** 0.5
            return nn.Parameter(torch.empty(o, i).uniform_(-b, b))
        self.base_w0, self.base_w1 = xav(64, 32), xav(16, 64)
        self.head_w0, self.head_w1, self.head_w2 = xav(64, 64), xav(64, 64), xav(16, 64)
        self.embedding = nn.Parameter(torch.randn(num_images, 32))
        self.average_init_density = average_init_density
        self.register_buffer("_frag_index", torch.from_numpy(_FRAG_INDEX), persistent=False)
        self.register_buffer("_frag_index_bwd", torch.from_numpy(_FRAG_INDEX_BWD), persistent=False)

    def _flat_weights(self):
        return torch.cat([self.base_w0.detach().reshape(-1), self.base_w1.detach().reshape(-1), self.head_w0.detach().reshape(-1),
                          self.head_w1.detach().reshape(-1), self.head_w2.detach().reshape(-1),
                          torch.zeros(1, device=self.table.device)])

    def _cache(self):
        """fp16 copies of the parameters are made once per parameter state: a training step queries the field three times
        (render, grid refresh, backward) between two optimizer steps, and every conversion is a handful of small launches."""
        key = _param_key(self.grad_params())
        c = getattr(self, "_pack_cache", None)
        if c is None or c["key"] != key:
            c = {"key": key}
            self._pack_cache = c
        return c

    def invalidate_packed(self):
        """Drop the cached fp16 copies (needed only after a parameter edit that bypasses tensor version counters, e.g. ``p.data``)."""
        self._pack_cache = None

    def packed(self, with_average: bool = True):
        """fp16 copies for the kernels: (table, 24 weight fragments, embedding rows).  ``with_average`` appends the mean embedding
        as an extra row (eval / camera-less queries use it); training queries index real rows only and skip the reduction."""
        c = self._cache()
        if "table" not in c:
            c["table"] = self.table.detach().half().contiguous()
            c["wfrag"] = self._flat_weights()[self._frag_index].half().contiguous()
        k = "emb_avg" if with_average else "emb"
        if k not in c:
            e = self.embedding.detach()
            c[k] = (torch.cat([e, e.mean(0, keepdim=True)], 0) if with_average else e).half().contiguous()
        return c["table"], c["wfrag"], c[k]

    def packed_bwd(self):
        c = self._cache()
        if "wfrag_bwd" not in c:
            c["wfrag_bwd"] = self._flat_weights()[self._frag_index_bwd].half().contiguous()
        return c["wfrag_bwd"]

    def dump_buffer(self, R: int, S: int, device) -> torch.Tensor:
        """Persistent (X, dY) scratch of the field backward, one per (R, S) shape; zeroed ONCE (padding rows stay zero)."""
        if not hasattr(self, "_dumps"):
            self._dumps = {}
        key = (R, S, str(device))
        if key not in self._dumps:
            self._dumps[key] = torch.zeros(_lib.load().neraf_field_backward_dump_bytes(R, S), dtype=torch.uint8, device=device)
        return self._dumps[key]

    def splitk_buffer(self, device) -> torch.Tensor:
        """Split-K scratch of the weight-gradient GEMMs: one per (device, stream) -- a caller that runs two backward passes of this
        field on two streams must not share it."""
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        if not isinstance(getattr(self, "_splitk", None), dict):
            self._splitk = {}
        if key not in self._splitk:
            self._splitk[key] = torch.empty(4 << 20, dtype=torch.float32, device=device)
        return self._splitk[key]

    def acc_scratch(self, device) -> torch.Tensor:
        """Persistent 64-bit fixed-point accumulator of the hash-table gradient (8 bytes per table row): zeroed ONCE here; every
        backward leaves it zero again (csrc/field_bwd.hip ``field_unpack_grad_kernel``), so no call pays a 49 MB fill."""
        a = getattr(self, "_acc_scratch", None)
        if a is None or a.device != torch.device(device) or a.numel() != self.table.shape[0]:
            a = torch.zeros(self.table.shape[0], dtype=torch.int64, device=device)
            self._acc_scratch = a
        return a

    def backward_query(self, packed, origins, directions, e_bins, camera_indices, density, d_rgb, d_density, pos_run: int = 1,
                       d_rays: Optional[torch.Tensor] = None, saved=None, in_autograd: bool = False, accumulate_into=None,
                       contract: Optional[bool] = None):
        """Gradients of the field parameters for upstream d_rgb [R,S,3] / d_density [R,S] of a structured query.
        Returns [d table, d base_w0, d base_w1, d head_w0, d head_w1, d head_w2, d embedding].  ``pos_run`` > 1: runs of that many
        consecutive rays share their single sample position (the grid refresh), see neraf_field_backward_runs.

        ``in_autograd`` (set by the autograd nodes, which call this from inside a backward pass with every field parameter
        requiring a gradient): the FIRST call of a backward pass writes fresh tensors and returns them; every LATER call of the same
        pass (NeRAF: the render batch's node after the grid refresh's, NeRAF_model.py:395-400) ADDS into those tensors in place
        (``accumulate`` of neraf_field_backward_ex) and returns Nones -- autograd then receives ONE gradient per parameter instead of
        two to sum (seven add launches per step, one of them over the 49 MB table gradient).  Only raw pointers are remembered
        between the calls (a reference would stop AccumulateGrad from adopting the tensors without a copy); they stay valid because
        autograd holds the first contribution in the parameter's input buffer until every producer of the pass has run.  An
        end-of-pass engine callback forgets them.  ``accumulate_into`` = a previous call's seven tensors: the explicit form of the same
        thing for direct callers (adds into them, returns them).

        ``contract``: the position mapping the FORWARD of this query used (scene contraction, or the plain box of the grid refresh,
        NeRAF_model.py:302-407, which switches ``spatial_distortion`` off only for the duration of its forward call); None = the
        module's current setting."""
        lib = _lib.load()
        dev = _dev_index(origins)
        device = origins.device
        R, S = e_bins.shape[0], e_bins.shape[1] - 1
        tab, wfrag, emb = packed
        wfrag_b = self.packed_bwd()
        first = getattr(self, "_pass_ptrs", None) if in_autograd else None
        if accumulate_into is not None:
            first = (accumulate_into[0].data_ptr(), accumulate_into[6].data_ptr(), [t.data_ptr() for t in accumulate_into[1:6]])
        if first is None:
            g_table = torch.empty_like(self.table)
            g_emb = torch.empty_like(self.embedding) if camera_indices is not None else torch.zeros_like(self.embedding)
            g_w = [torch.empty_like(p) for p in (self.base_w0, self.base_w1, self.head_w0, self.head_w1, self.head_w2)]
            ptrs = (g_table.data_ptr(), g_emb.data_ptr(), [t.data_ptr() for t in g_w])
            if in_autograd:
                self._pass_ptrs = ptrs
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward_pass)
            accumulate = 0
        else:
            ptrs, accumulate = first, 1
            if camera_indices is None:
                raise RuntimeError("a later producer of the pass must index real embedding rows (camera indices)")
        w_ptrs = (C.c_void_p * 5)(*ptrs[2])
        dump = self.dump_buffer(R, S, device)
        splitk = self.splitk_buffer(device)
        cam = camera_indices.reshape(-1).to(torch.int32).contiguous() if camera_indices is not None else None
        if contract is None:
            contract = self.spatial_distortion is not None
        mode = 0 if contract else 1
        ab = _lib.host_f32(self.aabb)
        common = (_lib.ctx(dev), C.byref(self.desc), tab.data_ptr(), wfrag.data_ptr(), wfrag_b.data_ptr(),
                  emb.data_ptr(), origins.data_ptr(), directions.data_ptr(), e_bins.data_ptr(),
                  cam.data_ptr() if cam is not None else None, R, S, mode, ab, self.average_init_density,
                  -1 if cam is not None else self.embedding.shape[0], density.data_ptr(),
                  d_rgb.data_ptr(), d_density.data_ptr(), ptrs[0],
                  ptrs[1] if cam is not None else None, w_ptrs, dump.data_ptr(),
                  splitk.data_ptr(), splitk.numel() * 4)
        # d_rays: camera-pose edge, also accumulate d loss / d (origin, direction) per ray (fp32 [R,6]); saved = (enc, denc | None) as
        # stored by query(save=...) for this batch and parameter state: no second walk of the hash table
        if d_rays is not None and pos_run != 1:
            raise ValueError("ray gradients are for structured ray batches (pos_run == 1)")
        enc, denc = saved if saved is not None else (None, None)
        _lib.check(lib.neraf_field_backward_ex(*common, int(pos_run), d_rays.data_ptr() if d_rays is not None else None,
                                               enc.data_ptr() if enc is not None else None,
                                               denc.data_ptr() if (denc is not None and d_rays is not None) else None,
                                               self.acc_scratch(device).data_ptr(), accumulate, self.embedding.shape[0], _stream_ptr()), dev)
        if accumulate_into is not None:
            return list(accumulate_into)
        if first is not None:
            return [None] * 7
        return [g_table] + g_w + [g_emb]

    def _end_of_backward_pass(self):
        self._pass_ptrs = None

    def grad_params(self):
        return [self.table, self.base_w0, self.base_w1, self.head_w0, self.head_w1, self.head_w2, self.embedding]

    def query(self, origins, directions, e_bins, camera_indices=None, use_average_embedding: bool = False, packed=None, save: int = 0,
              coherent_rays: bool = False):
        """Structured query: R rays x S samples.  Returns (rgb [R,S,3], density [R,S]); with ``save`` = 1 / 2 also
        ``(enc fp16 [R*S,32], denc fp16 [R*S,4,24] | None)``: the interpolated encoding (and its position derivatives) for
        ``backward_query(saved=...)``.  ``coherent_rays``: consecutive rays are neighbouring pixels of one camera (inference only;
        include/neraf_hip.h, neraf_proposal_density)."""
        lib = _lib.load()
        dev = _dev_index(origins)
        R, S = e_bins.shape[0], e_bins.shape[1] - 1
        if torch.is_grad_enabled():
            # a forward under autograd opens a new pass: pointers a FAILED backward left behind (the engine skips its end-of-pass
            # callbacks when a node raises) must not be accumulated into by the next one
            self._pass_ptrs = None
        tab, wfrag, emb = packed if packed is not None else self.packed()
        rgb = torch.empty((R, S, 3), dtype=torch.float32, device=origins.device)
        den = torch.empty((R, S), dtype=torch.float32, device=origins.device)
        avg_row = self.embedding.shape[0] if (use_average_embedding or camera_indices is None) else -1
        if avg_row >= 0 and emb.shape[0] <= avg_row:
            raise ValueError("this query uses the average appearance embedding: pack with with_average=True")
        cam = camera_indices.reshape(-1).to(torch.int32).contiguous() if camera_indices is not None else None
        mode = 0 if self.spatial_distortion is not None else 1
        ab = _lib.host_f32(self.aabb)
        args = (_lib.ctx(dev), C.byref(self.desc), tab.data_ptr(), wfrag.data_ptr(), emb.data_ptr(),
                origins.data_ptr(), directions.data_ptr(), e_bins.data_ptr(),
                cam.data_ptr() if cam is not None else None, R, S, mode, ab,
                self.average_init_density, avg_row)
        if save:
            enc = torch.empty((R * S, 32), dtype=torch.float16, device=origins.device)
            denc = torch.empty((R * S, 4, 24), dtype=torch.float16, device=origins.device) if save >= 2 else None
            _lib.check(lib.neraf_field_query_train(*args, rgb.data_ptr(), den.data_ptr(), enc.data_ptr(),
                                                   denc.data_ptr() if denc is not None else None, _stream_ptr()), dev)
            return rgb, den, (enc, denc)
        _lib.check(lib.neraf_field_query(*args, int(coherent_rays), rgb.data_ptr(), den.data_ptr(), _stream_ptr()), dev)
        return rgb, den

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False) -> Dict[str, torch.Tensor]:
        """Generic frustum query (one sample per frustum), e.g. the refresh call NeRAF_model.py:339."""
        f = ray_samples.frustums
        o = f.origins.reshape(-1, 3).float().contiguous()
        d = f.directions.reshape(-1, 3).float().contiguous()
        e = torch.cat([f.starts.reshape(-1, 1), f.ends.reshape(-1, 1)], dim=-1).float().contiguous()
        cam = ray_samples.camera_indices
        rgb, den = self.query(o, d, e, cam, use_average_embedding=not self.training)
        shp = f.origins.shape[:-1]
        return {FieldHeadNames.RGB: rgb.reshape(*shp, 3), FieldHeadNames.DENSITY: den.reshape(*shp, 1)}


class RGBRenderer:
    """``renderer_rgb`` as used by the refresh (NeRAF_model.py:344-350): background = last sample."""

    def __call__(self, rgb: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
        comp = torch.sum(weights * rgb, dim=-2)
        acc = torch.sum(weights, dim=-2)
        return comp + rgb[..., -1, :] * (1.0 - acc)


_MINMAX_FOLD = __import__("os").environ.get("NERAF_MINMAX_FOLD", "1") != "0"


class _VisionLossFn(torch.autograd.Function):
    """{rgb_loss, interlevel_loss, distortion_loss} as one autograd node over the radiance parameters."""

    @staticmethod
    def forward(ctx, model: "NeRAFVisionModel", st: dict, gt: torch.Tensor, ray_o: torch.Tensor, ray_d: torch.Tensor,
                *params: torch.Tensor):
        """``ray_o`` / ``ray_d`` are the bundle's origins / directions as the camera optimizer produced them: when they carry a graph
        (pose refinement on) the backward also returns d loss / d (origin, direction) per ray."""
        lib = _lib.load()
        dev = _dev_index(gt)
        h, stream = _lib.ctx(dev), _stream_ptr()
        fine = st["samples"][-1]
        R, S2 = st["dens"].shape
        sums = st.pop("loss_sums", None)             # zeroed by the composite's seeding launch; one use (a second loss call: own fill)
        if sums is None:
            sums = torch.zeros(4, dtype=torch.float32, device=gt.device)
        # values and UNIT gradients in one pass (the backward only scales them by the upstream scalars)
        need_grad = any(ctx.needs_input_grad[3:])
        f32 = dict(dtype=torch.float32, device=gt.device)
        unit = (torch.empty((R, S2, 3), **f32), torch.empty((R, S2), **f32), torch.empty((R, S2), **f32)) if need_grad else None
        _lib.check(lib.neraf_render_loss(h, st["dens"].data_ptr(), st["rgb_s"].data_ptr(), fine.e_bins.data_ptr(),
                                         fine.s_bins.data_ptr(), gt.data_ptr(), R, S2, model.distortion_loss_mult, None,
                                         unit[0].data_ptr() if unit else None, unit[1].data_ptr() if unit else None,
                                         unit[2].data_ptr() if unit else None, sums.data_ptr(), stream), dev)
        ctx.unit = unit
        for i in range(2):
            ps = st["samples"][i]
            Sp = ps.e_bins.shape[1] - 1
            _lib.check(lib.neraf_interlevel_loss(h, fine.s_bins.data_ptr(), st["w_fine"].data_ptr(), S2, ps.s_bins.data_ptr(),
                                                 ps.e_bins.data_ptr(), st["prop_dens"][i].data_ptr(), Sp, R, model.interlevel_loss_mult,
                                                 None, None, sums.data_ptr(), stream), dev)
        ctx.model, ctx.st, ctx.gt, ctx.dev = model, st, gt, dev
        ctx.set_materialize_grads(False)            # unused loss terms arrive as None, not as zero tensors (a fill launch each)
        ctx.n_params = len(params)
        ctx.need_rays = bool(ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        # {sum (rgb-gt)^2, sum distortion, sum outer} -> the three means and the batch psnr in one launch
        key = (R, S2, model.distortion_loss_mult, model.interlevel_loss_mult, str(gt.device))
        sc = getattr(model, "_loss_scale", None)
        if sc is None or sc[0] != key:
            sc = (key, torch.tensor([1.0 / (3.0 * R), model.distortion_loss_mult / R, model.interlevel_loss_mult / (R * S2)], **f32))
            model._loss_scale = sc
        out4 = torch.empty(4, **f32)
        _lib.check(lib.neraf_vision_loss_finalize(h, sums.data_ptr(), sc[1].data_ptr(), out4.data_ptr(), stream), dev)
        psnr_v = out4[3]
        ctx.mark_non_differentiable(psnr_v)
        return out4[0], out4[2], out4[1], psnr_v

    @staticmethod
    def backward(ctx, g_rgb, g_inter, g_dist, _g_psnr=None):
        lib = _lib.load()
        model, st, gt, dev = ctx.model, ctx.st, ctx.gt, ctx.dev
        h, stream = _lib.ctx(dev), _stream_ptr()
        device = gt.device
        f32 = dict(dtype=torch.float32, device=device)
        fine = st["samples"][-1]
        R, S2 = st["dens"].shape
        field = model.field.module
        u_rgb, u_dens, u_dens_dist = ctx.unit
        # one launch: upstream scalars gathered, unit gradients scaled (d rgb = u_rgb g_rgb, d density = u_dens g_rgb + u_dist g_dist),
        # the ray-gradient buffer and the interlevel sums zeroed
        scal = [(g.float().reshape(()) if g is not None else None) for g in (g_rgb, g_inter, g_dist)]
        up, sums = torch.empty(3, **f32), torch.empty(4, **f32)
        d_rgb_s, d_dens = torch.empty_like(u_rgb), torch.empty_like(u_dens)
        d_rays = torch.empty((R, 6), **f32) if ctx.need_rays else None
        _lib.check(lib.neraf_vision_bwd_prologue(h, u_rgb.data_ptr(), u_dens.data_ptr(), u_dens_dist.data_ptr(),
                                                 *[(g.data_ptr() if g is not None else None) for g in scal], R * S2, d_rgb_s.data_ptr(),
                                                 d_dens.data_ptr(), up.data_ptr(), d_rays.data_ptr() if d_rays is not None else None,
                                                 R * 6 if d_rays is not None else 0, sums.data_ptr(), stream), dev)
        # ---- main field (+ the camera-pose edge: d loss / d (origin, direction) per ray)
        grads = field.backward_query(st["field_packed"], st["o"], st["d"], fine.e_bins, st["cam"], st["dens"], d_rgb_s, d_dens,
                                     d_rays=d_rays, saved=st.get("field_saved"), in_autograd=all(ctx.needs_input_grad[5:12]),
                                     contract=st.get("contract"))
        ray_grads = (d_rays[:, :3], d_rays[:, 3:]) if ctx.need_rays else (None, None)
        # ---- proposal networks (interlevel loss); densities were computed under no_grad when not `updated`
        if not st["prop_updated"]:
            return (None, None, None, *ray_grads, *grads, *([None] * (3 * len(model.proposal_networks))))
        for i, pn in enumerate(model.proposal_networks):
            ps = st["samples"][i]
            Sp = ps.e_bins.shape[1] - 1
            d_pd = torch.empty((R, Sp), **f32)
            _lib.check(lib.neraf_interlevel_loss(h, fine.s_bins.data_ptr(), st["w_fine"].data_ptr(), S2, ps.s_bins.data_ptr(),
                                                 ps.e_bins.data_ptr(), st["prop_dens"][i].data_ptr(), Sp, R, model.interlevel_loss_mult,
                                                 up.data_ptr(), d_pd.data_ptr(), sums.data_ptr(), stream), dev)
            ptab, pw = st["prop_packed"][i]
            # outputs in the parameters' own layouts, written (not accumulated) by the kernels: nothing to zero, nothing to copy
            g_pt, g_w0, g_w1 = torch.empty_like(pn.table), torch.empty_like(pn.w0), torch.empty_like(pn.w1)
            # per-workgroup weight-gradient partials + the two-pass table gradient's per-sample encoding gradients and level masses
            scratch = torch.empty((int(lib.neraf_proposal_backward_scratch_bytes(R, Sp, pn.desc.n_levels)) + 3) // 4, **f32)
            _lib.check(lib.neraf_proposal_backward_ex(h, C.byref(pn.desc), ptab.data_ptr(), pw.data_ptr(), st["o"].data_ptr(),
                                                      st["d"].data_ptr(), ps.e_bins.data_ptr(), d_pd.data_ptr(), R, Sp,
                                                      pn.average_init_density, g_pt.data_ptr(), g_w0.data_ptr(), g_w1.data_ptr(),
                                                      scratch.data_ptr(), scratch.numel() * 4, pn.acc_scratch(device).data_ptr(),
                                                      d_rays.data_ptr() if ctx.need_rays else None, stream), dev)
            grads += [g_pt, g_w0, g_w1]
        return (None, None, None, *ray_grads, *grads)


class NeRAFVisionModel(nn.Module):
    """Drop-in surface of ``NeRAFVisionModel`` (NeRAF_model.py:54-79) with nerfacto defaults and the NeRAF
    overrides (NeRAF_config.py:94-98)."""

    num_proposal_samples_per_ray = (256, 96)
    num_nerf_samples_per_ray = 48
    near_plane, far_plane = 0.05, 1000.0
    eval_num_rays_per_chunk = 1 << 15
    proposal_weights_anneal_slope, proposal_weights_anneal_max_num_iters = 10.0, 1000
    proposal_update_every, proposal_warmup = 5, 5000          # nerfacto defaults [NS-recall]

    def __init__(self, aabb=None, num_train_data: int = None, average_init_density: float = 0.01, *, scene_box=None, config=None,
                 **kwargs):
        """Two call forms: ``NeRAFVisionModel(aabb [2,3], num_train_data)`` and nerfstudio's ``Model.__init__(config, scene_box,
        num_train_data, **kwargs)`` as ``NeRAFVisionModelConfig.setup`` calls it (NeRAF_pipeline.py:125-132; metadata, device,
        grad_scaler and seed_points are accepted and unused, as in NerfactoModel)."""
        super().__init__()
        from .field import NeRAFVisionFieldValue
        if aabb is not None and not isinstance(aabb, torch.Tensor) and hasattr(aabb, "eval_num_rays_per_chunk"):
            config, aabb = aabb, None                             # positional (config, scene_box, num_train_data)
            if num_train_data is not None and not isinstance(num_train_data, int):
                scene_box, num_train_data = num_train_data, kwargs.pop("num_train_data", None)
        if aabb is None:
            if scene_box is None:
                raise ValueError("NeRAFVisionModel needs the scene box (aabb=... or scene_box=...)")
            aabb = scene_box.aabb
        if num_train_data is None:
            raise ValueError("NeRAFVisionModel needs num_train_data (size of the appearance embedding)")
        self.config, self.scene_box, self.num_train_data = config, scene_box, num_train_data
        if config is not None:                                    # NeRAF_config.py:94-98 + nerfacto defaults
            for k in ("num_proposal_samples_per_ray", "num_nerf_samples_per_ray", "near_plane", "far_plane", "eval_num_rays_per_chunk",
                      "proposal_weights_anneal_slope", "proposal_weights_anneal_max_num_iters", "proposal_update_every",
                      "proposal_warmup", "interlevel_loss_mult", "distortion_loss_mult"):
                setattr(self, k, getattr(config, k))
            average_init_density = config.average_init_density
        aabb = torch.as_tensor(aabb).float()
        self.field = NeRAFVisionFieldValue(NerfactoField(aabb, num_train_data, average_init_density))   # NeRAF_model.py:61
        self.proposal_networks = nn.ModuleList([HashMLPDensityField(128, average_init_density=average_init_density),
                                                HashMLPDensityField(256, average_init_density=average_init_density)])
        self.renderer_rgb = RGBRenderer()
        from .cameras import CameraOptimizer
        cam_cfg = getattr(config, "camera_optimizer", None) if config is not None else None
        self.camera_optimizer = (cam_cfg.setup(num_cameras=num_train_data) if cam_cfg is not None
                                 else CameraOptimizer(num_train_data, mode="off"))
        self.audio_model = None
        self.step = 0
        self._steps_since_update = 0
        # optional ``jitter_fn(step, num_rays, device) -> 3 tensors [R]``: replaces torch.rand as the source of the sampler's single
        # jitter per ray and stage in training (reproducible runs; the trajectory parity test feeds the oracle's values)
        self.jitter_fn = None

    @property
    def device(self):
        return self.field.module.table.device

    def update_to_step(self, step: int):
        """ProposalNetworkSampler.step_cb [NS-recall]: called once per training iteration."""
        self.step = step
        self._steps_since_update += 1

    def _refresh_packs(self):
        """fp16 working copies of EVERY radiance parameter in two launches (csrc/pack.hip) when the parameter state changed since
        the last call -- i.e. once per optimizer step -- stored in the per-module caches that ``packed()`` / ``packed_bwd()`` consult.
        (The per-module paths they fall back to do the same with ~15 small torch launches.)"""
        f = self.field.module
        fkey = _param_key(f.grad_params())
        pkeys = [_param_key((pn.table, pn.w0, pn.w1)) for pn in self.proposal_networks]
        fc = getattr(f, "_pack_cache", None)
        if fc is not None and fc.get("key") == fkey and "wfrag_bwd" in fc and "emb" in fc and \
                all(getattr(pn, "_pack_cache", None) is not None and pn._pack_cache[0] == k for pn, k in zip(self.proposal_networks, pkeys)):
            return
        lib = _lib.load()
        dev_t = f.table.device
        dev = _dev_index(f.table)
        h16 = dict(dtype=torch.float16, device=dev_t)
        table16 = torch.empty(f.table.shape, **h16)
        emb16 = torch.empty(f.embedding.shape, **h16)
        srcs, dsts, lens = [f.table, f.embedding], [table16, emb16], [f.table.numel(), f.embedding.numel()]
        prop = []
        for pn in self.proposal_networks:
            t16 = torch.empty(pn.table.shape, **h16)
            w16 = torch.empty(16 * 16 + 16, **h16)
            prop.append((t16, w16))
            srcs += [pn.table, pn.w0, pn.w1]                      # w1: only its first row (the used output) follows w0
            dsts += [t16, w16, w16[256:]]
            lens += [pn.table.numel(), 256, 16]
        n = len(srcs)
        _lib.check(lib.neraf_cvt_f16_segments(_lib.ctx(dev), _lib.ptr_array([s_.detach() for s_ in srcs]), _lib.ptr_array(dsts),
                                              (C.c_longlong * n)(*lens), n, _stream_ptr()), dev)
        ws = [f.base_w0, f.base_w1, f.head_w0, f.head_w1, f.head_w2]
        if getattr(f, "_frag_index_all", None) is None or f._frag_index_all.device != dev_t:
            f._frag_index_all = torch.cat([f._frag_index, f._frag_index_bwd]).contiguous()
        nf, nb = f._frag_index.numel(), f._frag_index_bwd.numel()
        frag = torch.empty(nf + nb, **h16)
        _lib.check(lib.neraf_gather_f16(_lib.ctx(dev), _lib.ptr_array([w.detach() for w in ws]), (C.c_longlong * 5)(*[w.numel() for w in ws]), 5,
                                        f._frag_index_all.data_ptr(), frag.data_ptr(), nf + nb, _stream_ptr()), dev)
        f._pack_cache = {"key": fkey, "table": table16, "wfrag": frag[:nf], "wfrag_bwd": frag[nf:], "emb": emb16}
        for pn, k, pk in zip(self.proposal_networks, pkeys, prop):
            pn._pack_cache = (k, pk)

    def _next_jitter_seed(self) -> int:
        """Next value of a splitmix64 stream (never 0: 0 means "no jitter" to the kernels)."""
        M = (1 << 64) - 1
        st = getattr(self, "_jitter_state", None)
        if st is None:
            # the stream starts from (torch seed, data-parallel rank, training step): ranks that share the torch seed draw different
            # jitters, and a run resumed at step k does not replay the jitters of step 0
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            st = (torch.initial_seed() * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019 + rank * 0xD1B54A32D192ED03 +
                  int(self.step) * 0x8CB92BA72F3D8DD7) & M
        st = (st + 0x9E3779B97F4A7C15) & M
        self._jitter_state = st
        z = st
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
        return z or 1

    def _proposal_updated(self) -> bool:
        """ProposalNetworkSampler's schedule [NS-recall]: the proposal networks receive gradients on every step during
        warm-up and then only when more than ``update_sched(step)`` (-> proposal_update_every) steps have passed."""
        sched = float(np.clip(np.interp(self.step, [0, self.proposal_warmup], [0, self.proposal_update_every]), 1,
                              self.proposal_update_every))
        updated = self._steps_since_update > sched or self.step < 10
        if updated:
            self._steps_since_update = 0
        return updated

    def _anneal(self) -> float:
        if not self.training:
            return 1.0
        x = min(max(self.step / self.proposal_weights_anneal_max_num_iters, 0.0), 1.0)
        b = self.proposal_weights_anneal_slope
        return (b * x) / ((b - 1) * x + 1)

    def get_outputs(self, ray_bundle: RayBundle, jitters: Optional[List[torch.Tensor]] = None, coherent_rays: int = 0,
                    _out: Optional[Dict[str, torch.Tensor]] = None):
        """NerfactoModel.get_outputs + rgb clip (NeRAF_model.py:65-68).  ``jitters`` (3 tensors [R]) override the
        training-time single jitter of the three sampling stages (tests pass the oracle's values).  ``coherent_rays``: the bundle is
        a run of neighbouring pixels of one camera (the chunks of ``get_outputs_for_camera``): the gather kernels then put
        neighbouring RAYS, not neighbouring samples of a ray, into a wavefront (same results, fewer cache lines per instruction).
        ``_out``: preallocated rgb / depth / expected_depth / accumulation rows to write into (the chunked frame render)."""
        lib = _lib.load()
        # eval renders use the mean appearance embedding: the per-ray camera indices are not read (their int32 conversion was one torch
        # launch per 32768-ray chunk: 22 x 5 us of a frame)
        cam32 = ray_bundle.camera_indices_i32() if self.training else None
        if self.training:
            ray_bundle = self.camera_optimizer.apply_to_raybundle(ray_bundle)     # NerfactoModel.get_outputs [NS-recall]
        ray_o, ray_d = ray_bundle.origins, ray_bundle.directions      # carry the camera optimizer's graph when it is on
        o = ray_o.detach().float().contiguous()
        d = ray_d.detach().float().contiguous()
        dev = _dev_index(o)
        h, st = _lib.ctx(dev), _stream_ptr()
        R = o.shape[0]
        near, far = self.near_plane, self.far_plane
        seeds = [0, 0, 0]
        if self.training and jitters is None:
            if self.jitter_fn is not None:
                jitters = self.jitter_fn(self.step, R, o.device)
            else:
                # the single jitter per ray and stage is drawn INSIDE the sampler kernels from (seed, ray): three seeds per call from a
                # host-side splitmix64 stream started at torch.initial_seed() -- reproducible under torch.manual_seed, no launch
                seeds = [self._next_jitter_seed() for _ in range(3)]
        jit = [j.reshape(-1).float().contiguous() if j is not None else None for j in (jitters or [None] * 3)]
        jp = [j.data_ptr() if j is not None else None for j in jit]
        S0, S1 = self.num_proposal_samples_per_ray
        S2 = self.num_nerf_samples_per_ray
        f32 = dict(dtype=torch.float32, device=o.device)
        # without jitter every ray gets the same first-stage edges: a whole-frame render generates ONE row and reads it with stride 0
        R0 = 1 if (_out is not None and jp[0] is None and seeds[0] == 0) else R
        s0, e0 = torch.empty((R0, S0 + 1), **f32), torch.empty((R0, S0 + 1), **f32)
        _lib.check(lib.neraf_sample_uniform(h, R0, S0, near, far, jp[0], seeds[0], s0.data_ptr(), e0.data_ptr(), st), dev)
        anneal = self._anneal()
        prop_updated = self._proposal_updated() if self.training else False
        weights_list, samples_list = [], []
        s_prev, e_prev = s0, e0
        # {min step, max step} of the expected-depth clip, formed by the two sampler launches themselves (first: seeds the pair and
        # zeroes the loss node's four sums behind it; second: accumulates its rays' range) -- the composite is then ONE launch
        fold = _MINMAX_FOLD
        n_mm = 64 * 64 if fold else 2           # folded: 64 replicas of the pair, 256 bytes apart (include/neraf_hip.h, neraf_pdf_resample_mm)
        scratch = torch.empty(n_mm + (4 if self.training else 0), dtype=torch.float32, device=o.device)
        if self.training:
            self._refresh_packs()
        prop_packed = [pn.packed() for pn in self.proposal_networks]
        prop_dens = []
        for i, S_next in enumerate((S1, S2)):
            dens = self.proposal_networks[i].density(o, d, e_prev, packed=prop_packed[i], coherent_rays=0 if self.training else int(coherent_rays))
            prop_dens.append(dens)
            S_cur = e_prev.shape[1] - 1
            # the proposal levels' rendering weights only feed the interlevel loss: a whole-frame render (_out given) does not ask for them
            w = torch.empty((R, S_cur), **f32) if _out is None else None
            s_n, e_n = torch.empty((R, S_next + 1), **f32), torch.empty((R, S_next + 1), **f32)
            _lib.check(lib.neraf_pdf_resample_mm(h, dens.data_ptr(), s_prev.data_ptr(), e_prev.data_ptr(),
                                                 0 if (e_prev.shape[0] == 1 and R > 1) else S_cur + 1, R, S_cur, anneal,
                                                 jp[i + 1], seeds[i + 1], S_next, near, far, w.data_ptr() if w is not None else None,
                                                 s_n.data_ptr(), e_n.data_ptr(), scratch.data_ptr(), scratch.numel() * 4, (i + 1) if fold else 0, st), dev)
            weights_list.append(w)
            samples_list.append(RaySamples(None, ray_bundle.camera_indices, s_prev, e_prev))
            s_prev, e_prev = s_n, e_n
        field = self.field.module
        field_packed = field.packed(with_average=not self.training or cam32 is None)
        saved = None
        if self.training and torch.is_grad_enabled():
            # keep the encoding for the backward; with trainable poses (the rays carry the camera optimizer's graph) its derivatives too
            rgb_s, dens, saved = field.query(o, d, e_prev, cam32, use_average_embedding=False, packed=field_packed,
                                             save=2 if (ray_o.requires_grad or ray_d.requires_grad) else 1)
        else:
            rgb_s, dens = field.query(o, d, e_prev, cam32, use_average_embedding=not self.training,
                                      packed=field_packed, coherent_rays=0 if self.training else int(coherent_rays))
        w = torch.empty((R, S2), **f32)
        if _out is not None:
            rgb, depth, expd, acc = _out["rgb"], _out["depth"], _out["expected_depth"], _out["accumulation"]
        else:
            rgb, depth = torch.empty((R, 3), **f32), torch.empty((R, 1), **f32)
            expd, acc = torch.empty((R, 1), **f32), torch.empty((R, 1), **f32)
        if fold:
            _lib.check(lib.neraf_composite_mm(h, dens.data_ptr(), rgb_s.data_ptr(), e_prev.data_ptr(), R, S2, int(self.training),
                                              w.data_ptr(), rgb.data_ptr(), depth.data_ptr(), expd.data_ptr(), acc.data_ptr(),
                                              scratch.data_ptr(), st), dev)
        else:       # round 5's form (A/B: NERAF_MINMAX_FOLD=0): the composite's own seeding + reduction launches
            _lib.check(lib.neraf_composite(h, dens.data_ptr(), rgb_s.data_ptr(), e_prev.data_ptr(), R, S2, int(self.training),
                                           w.data_ptr(), rgb.data_ptr(), depth.data_ptr(), expd.data_ptr(), acc.data_ptr(),
                                           scratch.data_ptr(), scratch.numel() * 4, st), dev)
        weights_list.append(w)
        samples_list.append(RaySamples(None, ray_bundle.camera_indices, s_prev, e_prev))
        out = {"rgb": rgb, "accumulation": acc, "depth": depth, "expected_depth": expd}
        if self.training:
            out["weights_list"] = weights_list
            out["ray_samples_list"] = samples_list
            # everything the fused loss/backward node needs (same packed fp16 parameter copies as the forward used)
            out["_state"] = dict(o=o, d=d, ray_o=ray_o, ray_d=ray_d, cam=cam32, samples=samples_list, prop_dens=prop_dens,
                                 prop_packed=prop_packed, field_packed=field_packed, field_saved=saved, rgb_s=rgb_s, dens=dens, w_fine=w,
                                 prop_updated=prop_updated, loss_sums=scratch[n_mm:n_mm + 4], contract=field.spatial_distortion is not None)
        out["rgb_samples"], out["density"] = rgb_s, dens
        return out

    # ---- V4: losses + backward ---------------------------------------------------------------------------
    interlevel_loss_mult, distortion_loss_mult = 1.0, 0.002

    def loss_params(self) -> List[torch.Tensor]:
        f = self.field.module
        ps = [f.table, f.base_w0, f.base_w1, f.head_w0, f.head_w1, f.head_w2, f.embedding]
        for pn in self.proposal_networks:
            ps += [pn.table, pn.w0, pn.w1]
        return ps

    def get_metrics_dict(self, outputs, batch):
        """NerfactoModel.get_metrics_dict [NS-recall]: psnr of the batch (+ camera-optimizer norms in training).  In training the
        psnr is filled in by ``get_loss_dict`` from the rgb loss it computes anyway (psnr = -10 log10(mse), written by the launch that
        finalises the losses); outside training it is computed here."""
        m: Dict[str, torch.Tensor] = {}
        if self.training:
            self.camera_optimizer.get_metrics_dict(m)
        elif batch is not None and ("image" in batch or "rgb" in batch):
            gt = (batch["image"] if "image" in batch else batch["rgb"]).to(outputs["rgb"].device).float()
            m["psnr"] = psnr(outputs["rgb"].detach(), gt)
        return m

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, torch.Tensor]:
        """NerfactoModel.get_loss_dict: rgb MSE (on the clipped colour, NeRAF_model.py:67) + interlevel + distortion
        (+ the camera-optimizer regulariser)."""
        if "_state" not in outputs:
            raise RuntimeError("get_loss_dict needs the outputs of a training-mode get_outputs call")
        gt = (batch["image"] if "image" in batch else batch["rgb"]).to(outputs["rgb"].device).float().contiguous()
        st = outputs["_state"]
        rgb_l, inter, dist, psnr_v = _VisionLossFn.apply(self, st, gt, st["ray_o"], st["ray_d"], *self.loss_params())
        d = {"rgb_loss": rgb_l, "interlevel_loss": inter, "distortion_loss": dist}
        self.camera_optimizer.get_loss_dict(d)
        if metrics_dict is not None and "psnr" not in metrics_dict:
            metrics_dict["psnr"] = psnr_v                      # -10 log10(mse), from the loss-finalising launch
        return d

    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        """NerfactoModel.get_param_groups [NS-recall]: {"proposal_networks", "fields", "camera_opt"}."""
        g = {"proposal_networks": list(self.proposal_networks.parameters()), "fields": list(self.field.parameters())}
        self.camera_optimizer.get_param_groups(g)
        return g

    def get_training_callbacks(self, training_callback_attributes=None):
        """nerfacto registers the proposal sampler's step callbacks (anneal, update schedule) [NS-recall]; here the Trainer calls
        ``update_to_step(step)`` once per iteration instead (neraf_amd/pipeline.py::train_iteration)."""
        return []

    forward = get_outputs

    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, ray_bundle: RayBundle, coherent_rays: int = 0):
        """Full-image render in chunks of eval_num_rays_per_chunk rays (NeRAF_config.py:95), as nerfstudio's
        Model.get_outputs_for_camera_ray_bundle does; NeRAFVisionModel.get_outputs_for_camera (NeRAF_model.py:70-79)
        then clips rgb (already applied by the composite kernel).  Every chunk writes its rows of the frame-sized outputs directly
        (no concatenation); ``coherent_rays``: the bundle is one camera's pixels in row-major order (see ``get_outputs``) -- 1 / True,
        or the image WIDTH: chunks that are whole rows of a width divisible by 8 then walk 8 x 8 / 4 x 4 pixel tiles instead of row
        segments (include/neraf_hip.h, neraf_proposal_density)."""
        R = len(ray_bundle)
        W = int(coherent_rays)
        f32 = dict(dtype=torch.float32, device=ray_bundle.origins.device)
        full = {"rgb": torch.empty((R, 3), **f32), "accumulation": torch.empty((R, 1), **f32), "depth": torch.empty((R, 1), **f32),
                "expected_depth": torch.empty((R, 1), **f32)}
        was = self.training
        self.eval()
        try:
            for i in range(0, R, self.eval_num_rays_per_chunk):
                sl = slice(i, min(R, i + self.eval_num_rays_per_chunk))
                rb = RayBundle(ray_bundle.origins[sl], ray_bundle.directions[sl],
                               ray_bundle.camera_indices[sl] if ray_bundle.camera_indices is not None else None)
                n = sl.stop - sl.start
                tiles = W > 1 and W % 8 == 0 and i % W == 0 and n % W == 0
                self.get_outputs(rb, coherent_rays=W if tiles else min(W, 1), _out={k: v[sl] for k, v in full.items()})
        finally:
            self.train(was)
        return full

    @torch.no_grad()
    def get_outputs_for_camera(self, camera, obb_box=None, eval: bool = False):
        """NeRAFVisionModel.get_outputs_for_camera(camera, obb_box, eval) (NeRAF_model.py:70-79): rays of every pixel of ``camera``
        (a one-camera ``neraf_amd.cameras.Cameras``) -> chunked render -> image-shaped outputs [H,W,.], rgb clipped to [0,1].
        ``eval=False`` is the viewer branch that additionally renders the audio model's outputs (:73-77): UI code, out of scope."""
        if obb_box is not None:
            raise NotImplementedError("oriented-box cropping is viewer functionality (the reference always passes None)")
        if not eval and self.audio_model is not None and getattr(self.audio_model, "viewer_enabled", False):
            raise NotImplementedError("viewer branch (NeRAF_model.py:73-77) is UI code, out of scope (SURVEY.md row 15)")
        cam = camera.to(self.device)
        rb = cam.generate_rays(0)
        H, W = cam.height, cam.width
        # row-major pixels of one camera; NERAF_PIXEL_TILES=0 (measurement toggle): row segments instead of pixel tiles
        out = self.get_outputs_for_camera_ray_bundle(rb, coherent_rays=int(W) if os.environ.get("NERAF_PIXEL_TILES", "1") != "0" else 1)
        # rgb is already clipped to [0,1] by the composite kernel (:78; csrc/field.hip composite_kernel)
        return {k: v.reshape(H, W, -1) for k, v in out.items()}

    @torch.no_grad()
    def get_image_metrics_and_images(self, outputs, batch):
        """NerfactoModel.get_image_metrics_and_images [NS-recall]: PSNR and SSIM of the rendered frame against ``batch['image']``
        [H,W,3], and the side-by-side image.  LPIPS needs pretrained network weights that cannot be obtained offline: omitted."""
        gt = batch["image"].to(outputs["rgb"].device).float()
        pred = outputs["rgb"]
        metrics = {"psnr": float(psnr(pred, gt)), "ssim": float(ssim(pred, gt))}
        images = {"img": torch.cat([gt, pred], dim=1), "accumulation": outputs["accumulation"], "depth": outputs["depth"]}
        return metrics, images


def psnr(pred: torch.Tensor, gt: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """10 log10(range^2 / MSE) over all elements (torchmetrics PeakSignalNoiseRatio(data_range=1.0) as nerfacto builds it)."""
    mse = torch.mean((pred.float() - gt.float()) ** 2)
    return 10.0 * torch.log10(data_range ** 2 / mse)


def ssim(pred: torch.Tensor, gt: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """Structural similarity with torchmetrics' defaults (11x11 gaussian window, sigma 1.5, k1 0.01, k2 0.03), images [H,W,3]."""
    import torch.nn.functional as F
    x = pred.float().permute(2, 0, 1)[None]
    y = gt.float().permute(2, 0, 1)[None]
    ax = torch.arange(11, dtype=torch.float32, device=x.device) - 5
    g = torch.exp(-(ax ** 2) / (2 * 1.5 ** 2))
    g = (g / g.sum())
    win = (g[:, None] * g[None, :])[None, None].expand(3, 1, 11, 11)
    pad = 5
    xp, yp = F.pad(x, (pad,) * 4, mode="reflect"), F.pad(y, (pad,) * 4, mode="reflect")
    mu_x, mu_y = F.conv2d(xp, win, groups=3), F.conv2d(yp, win, groups=3)
    sxx = F.conv2d(xp * xp, win, groups=3) - mu_x ** 2
    syy = F.conv2d(yp * yp, win, groups=3) - mu_y ** 2
    sxy = F.conv2d(xp * yp, win, groups=3) - mu_x * mu_y
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * mu_x * mu_y + c1) * (2 * sxy + c2)) / ((mu_x ** 2 + mu_y ** 2 + c1) * (sxx + syy + c2))
    return s[..., pad:-pad, pad:-pad].mean() if min(s.shape[-2:]) > 2 * pad else s.mean()

"""Host-side mirror of ``NeRAF/NeRAF_field.py`` for the MI355X engine.

``NeRAFAudioSoundField`` keeps the reference's constructor signature, sub-module names and
therefore its state-dict keys (``soundfield.{0-4}.{weight,bias}``, ``STFT_linear.{c}.{weight,bias}``;
NeRAF_field.py:39-45), so reference checkpoints load with ``load_state_dict``.  The arithmetic is
not PyTorch: ``forward`` runs the fused fp16-MFMA MLP of libneraf_hip through the C ABI
(include/neraf_hip.h) and there is no fallback when the library or the GPU is missing.

Two entry points:
  * ``forward(h)``            -- exactly NeRAF_field.py:47 (dense ``h [B, in_size]``).
  * ``forward_queries(...)``  -- what NeRAFAudioModel.get_outputs needs (NeRAF_model.py:531-564):
    raw batch fields + the shared 1024-d grid feature; the query encodings are computed on the GPU
    and the feature half of layer 0 is folded into its bias (layer-0 split).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib

N_QUERY = 163  # 21 (time) + 63 (mic) + 63 (source) + 16 (SH rot)  NeRAF_model.py:169-171


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream_ptr() -> C.c_void_p:
    """The current torch stream of the current device as the ABI's ``neraf_stream_t``.  ``torch.cuda.current_stream()`` builds a
    Stream object per call (~10 us; a training step asks a few hundred times, one audio eval call twice): the raw-handle query torch
    itself uses is ~1 us."""
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_index(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise RuntimeError("neraf_amd ops need CUDA(HIP) tensors: the NeRAF hot path has no CPU fallback")
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


class _NacfSplitFn(torch.autograd.Function):
    """out = NAcF(feat, encoded queries) ; grads for feat and the 2*(5+C) parameters."""

    @staticmethod
    def run(field: "NeRAFAudioSoundField", feat: torch.Tensor, ws: torch.Tensor, B: int, training: bool, params):
        """The forward launch itself; ``forward_queries`` calls it directly when no graph is recorded (the eval branch: one call per
        RIR, where ``Function.apply`` is a third of the host time of the call)."""
        lib = _lib.load()
        dev = _dev_index(feat)
        packed = field._packed(params, training)
        out = torch.empty((B, field.sound_rez, field.N_frequencies), dtype=torch.float32, device=feat.device)
        _lib.check(lib.neraf_nacf_fwd(_lib.ctx(dev), C.byref(field._desc), packed.data_ptr(), _lib.ptr_array(params), feat.data_ptr(), B,
                                      out.data_ptr(), ws.data_ptr(), int(training), _stream_ptr()), dev)
        return out, packed, dev

    @staticmethod
    def forward(ctx, field: "NeRAFAudioSoundField", feat: torch.Tensor, ws: torch.Tensor, B: int, training: bool,
                *params: torch.Tensor):
        out, packed, dev = _NacfSplitFn.run(field, feat, ws, B, training, params)
        ctx.field, ctx.B, ctx.dev = field, B, dev
        # optional hand-off: the producer of `feat` names the buffer it wants d loss / d feat in (ResNet3D.dfeat_buffer)
        gb = getattr(feat, "_neraf_grad_buffer", None)
        ctx.dfeat_out = gb if (gb is not None and gb.shape == feat.shape and gb.dtype == feat.dtype and gb.device == feat.device) else None
        if ctx.dfeat_out is not None:
            # ONE consumer per forward may write its gradient into the producer's buffer: a second NAcF call on the same feature would
            # overwrite the first one's gradient before autograd sums them -- it gets a tensor of its own
            feat._neraf_grad_buffer = None
        ctx.save_for_backward(feat, ws, out, packed, *params)
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        feat, ws, out, packed, *params = ctx.saved_tensors
        field: NeRAFAudioSoundField = ctx.field
        lib = _lib.load()
        dout = dout.contiguous().float()
        grads = [torch.empty_like(p) for p in params]
        dfeat = ctx.dfeat_out if ctx.dfeat_out is not None else torch.empty_like(feat)
        _lib.check(lib.neraf_nacf_bwd(_lib.ctx(ctx.dev), C.byref(field._desc), packed.data_ptr(), _lib.ptr_array(params),
                                      feat.data_ptr(), ctx.B, out.data_ptr(), dout.data_ptr(), _lib.ptr_array(grads),
                                      dfeat.data_ptr(), ws.data_ptr(), _stream_ptr()), ctx.dev)
        return (None, dfeat, None, None, None, *grads)


class _NacfDenseFn(torch.autograd.Function):
    """out = NeRAFAudioSoundField.forward(h) with h dense [B, in_size]."""

    @staticmethod
    def forward(ctx, field: "NeRAFAudioSoundField", h_in: torch.Tensor, training: bool, *params: torch.Tensor):
        lib = _lib.load()
        dev = _dev_index(h_in)
        B = h_in.shape[0]
        packed = field._packed(params, training, dense=True)
        ws = field._workspace(B, training, h_in.device, dense=True)
        out = torch.empty((B, field.sound_rez, field.N_frequencies), dtype=torch.float32, device=h_in.device)
        _lib.check(lib.neraf_nacf_fwd_dense(_lib.ctx(dev), C.byref(field._desc_dense), packed.data_ptr(), h_in.data_ptr(), B,
                                            out.data_ptr(), ws.data_ptr(), int(training), _stream_ptr()), dev)
        ctx.field, ctx.B, ctx.dev = field, B, dev
        ctx.need_dh = h_in.requires_grad
        ctx.save_for_backward(ws, out, packed, *params)
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        ws, out, packed, *params = ctx.saved_tensors
        field: NeRAFAudioSoundField = ctx.field
        lib = _lib.load()
        dout = dout.contiguous().float()
        grads = [torch.empty_like(p) for p in params]
        dh = torch.empty((ctx.B, field.in_size), dtype=torch.float32, device=dout.device) if ctx.need_dh else None
        _lib.check(lib.neraf_nacf_bwd_dense(_lib.ctx(ctx.dev), C.byref(field._desc_dense), packed.data_ptr(), ctx.B,
                                            out.data_ptr(), dout.data_ptr(), _lib.ptr_array(grads),
                                            dh.data_ptr() if dh is not None else None, ws.data_ptr(), _stream_ptr()),
                   ctx.dev)
        return (None, dh, None, *grads)


class NeRAFAudioSoundField(nn.Module):
    """Drop-in for ``NeRAFAudioSoundField`` (NeRAF_field.py:37-65) running on libneraf_hip."""

    def __init__(self, in_size: int, W: int, sound_rez: int = 2, N_frequencies: int = 257):
        super().__init__()
        # Parameter containers identical to the reference (NeRAF_field.py:41-45) -> same state-dict keys/init.
        self.soundfield = nn.ModuleList(
            [nn.Linear(in_size, 5096), nn.Linear(5096, 2048), nn.Linear(2048, 1024), nn.Linear(1024, 1024), nn.Linear(1024, W)])
        self.STFT_linear = nn.ModuleList([nn.Linear(W, N_frequencies) for _ in range(sound_rez)])
        if in_size < N_QUERY:
            raise ValueError(f"in_size must be >= {N_QUERY} (grid features + {N_QUERY} encoded query dims)")
        self.in_size, self.W, self.sound_rez, self.N_frequencies = in_size, W, sound_rez, N_frequencies
        # two layouts: split (layer-0 feature half folded into the bias) and dense (full K=in_size layer 0)
        self._desc = _lib.NacfDesc(in_size - N_QUERY, N_QUERY, W, sound_rez, N_frequencies, 0)
        self._desc_dense = _lib.NacfDesc(in_size - N_QUERY, N_QUERY, W, sound_rez, N_frequencies, 1)
        self._packed_buf = {False: None, True: None}
        self._packed_key = {False: None, True: None}

    # -- parameters in the C-ABI order (state-dict order) -------------------------------------
    def flat_params(self) -> List[torch.Tensor]:
        ps: List[torch.Tensor] = []
        for lin in list(self.soundfield) + list(self.STFT_linear):
            ps += [lin.weight, lin.bias]
        return ps

    def _packed(self, params: Sequence[torch.Tensor], training: bool = True, dense: bool = False) -> torch.Tensor:
        """fp16 MFMA-layout copy of the fp32 master weights.

        Training forwards ALWAYS re-pack: fused optimizers (torch.optim.Adam(fused=True)) update the
        parameters in place without bumping ``_version``, so a version-keyed cache would silently go
        stale.  Inference re-packs only when a parameter's (data_ptr, _version) changed; call
        ``invalidate_packed()`` after out-of-band weight edits."""
        from . import optim
        # UPDATE_EPOCH: FusedAdam writes parameters through raw pointers without bumping tensor versions
        key = (optim.UPDATE_EPOCH,) + tuple((p.data_ptr(), p._version) for p in params)
        dev = params[0].device
        desc = self._desc_dense if dense else self._desc
        cur = self._packed_buf[dense]
        if training or cur is None or cur.device != dev or key != self._packed_key[dense]:
            lib = _lib.load()
            d = _dev_index(params[0])
            nbytes = lib.neraf_nacf_packed_bytes(C.byref(desc))
            # a fresh buffer each re-pack: earlier ones may still be referenced by a pending backward
            buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            for p in params:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("NAcF parameters must be contiguous float32")
            _lib.check(lib.neraf_nacf_pack_weights(_lib.ctx(d), C.byref(desc), _lib.ptr_array(params),
                                                   buf.data_ptr(), _stream_ptr()), d)
            self._packed_buf[dense], self._packed_key[dense] = buf, key
        return self._packed_buf[dense]

    def invalidate_packed(self) -> None:
        self._packed_buf = {False: None, True: None}
        self._packed_key = {False: None, True: None}

    def _workspace(self, B: int, training: bool, device, dense: bool = False) -> torch.Tensor:
        desc = self._desc_dense if dense else self._desc
        nbytes = _lib.load().neraf_nacf_workspace_bytes(C.byref(desc), B, int(training))
        return torch.empty(nbytes, dtype=torch.uint8, device=device)

    # -- NeRAF_field.py:47 -----------------------------------------------------------------------
    def forward(self, h: torch.Tensor) -> torch.Tensor:
        if h.dim() != 2 or h.shape[1] != self.in_size:
            raise ValueError(f"h must be [B, {self.in_size}]")
        h = h.contiguous().float()
        training = torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in self.parameters()))
        return _NacfDenseFn.apply(self, h, training, *self.flat_params())

    # -- NeRAF_model.py:531-564 ------------------------------------------------------------------
    def forward_queries(self, feat: torch.Tensor, time_query: torch.Tensor, mic_pose: torch.Tensor,
                        source_pose: torch.Tensor, rot: torch.Tensor, aabb: torch.Tensor, max_len: int) -> torch.Tensor:
        """feat [n_feat] fp32 (ResNet3D output, flattened); batch fields as produced by the audio
        datamanager (NeRAF_dataset.py:129-130): time_query int64 [B], poses / rot float64 [B,3]."""
        lib = _lib.load()
        dev = _dev_index(mic_pose)
        B = int(time_query.shape[0])
        gb = getattr(feat, "_neraf_grad_buffer", None)
        if gb is not None:
            feat._neraf_grad_buffer = None    # one consumer per forward (see _NacfSplitFn.forward): a second call gets its own tensor
        feat = feat.reshape(-1).float().contiguous()
        if gb is not None:                    # the view made above is a new tensor object: carry the producer's hand-off along
            feat._neraf_grad_buffer = gb
        if feat.numel() != self._desc.n_feat:
            raise ValueError(f"feat must have {self._desc.n_feat} elements")
        tq = time_query.to(torch.int64).contiguous()
        # poses: one row per query [B,3], or ONE row [1,3] / [3] shared by all queries (the eval branch: T time queries of one RIR)
        shared = mic_pose.numel() == 3 and B > 1
        mic = mic_pose.to(torch.float64).contiguous()
        src = source_pose.to(torch.float64).contiguous()
        r = rot.to(torch.float64).contiguous()
        if (src.numel() == 3) != (mic.numel() == 3) or (r.numel() == 3) != (mic.numel() == 3):
            raise ValueError("mic_pose, source_pose and rot are all per query [B,3] or all shared [1,3]")
        training = torch.is_grad_enabled() and (feat.requires_grad or any(p.requires_grad for p in self.parameters()))
        ws = self._workspace(B, training, mic.device)
        ab = _lib.host_f32(aabb)
        _lib.check(lib.neraf_nacf_encode_queries_ex(_lib.ctx(dev), C.byref(self._desc), tq.data_ptr(), mic.data_ptr(),
                                                    src.data_ptr(), r.data_ptr(), 1 if shared else B, ab, int(max_len), B, ws.data_ptr(),
                                                    int(training), _stream_ptr()), dev)
        if not training:
            return _NacfSplitFn.run(self, feat, ws, B, False, self.flat_params())[0]
        return _NacfSplitFn.apply(self, feat, ws, B, training, *self.flat_params())


class NeRAFVisionFieldValue(nn.Module):
    """Pass-through wrapper that provides the ``.module`` attribute path (NeRAF_field.py:27-34)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, ray_samples, compute_normals=False):
        return self.module(ray_samples, compute_normals=compute_normals)

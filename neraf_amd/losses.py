"""Host-side mirror of the training loss in ``NeRAF/NeRAF_evaluator.py`` (STFTLoss :76-108,
SpectralConvergenceLoss :8-26, LogSTFTMagnitudeLoss :29-53) on libneraf_hip.

``STFTLoss(loss_type)(x_log, y_log)`` returns ``{'audio_sc_loss', 'audio_mag_loss'}`` exactly
like the reference; both reductions run in one fused HIP kernel and the backward is one
element-wise kernel.  No fallback without the HIP library / GPU.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .field import _dev_index, _stream_ptr


class _StftLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_log: torch.Tensor, y_log: torch.Tensor, loss_type: int, group=None, weights=None):
        """``weights``: optional device tensor [2] multiplied into (sc, mag) here -- the reference scales its two audio losses by
        constants right after computing them (NeRAF_model.py:592-599); done inside the node that is one kernel forward and one
        backward instead of four each."""
        lib = _lib.load()
        dev = _dev_index(x_log)
        x = x_log.contiguous().float()
        y = y_log.contiguous().float()
        if x.shape != y.shape:
            raise ValueError("STFTLoss: prediction / target shape mismatch")
        buf = torch.empty(4 + 4 * 256, dtype=torch.float32, device=x.device)      # NERAF_STFT_SUMS_FLOATS: the 4 sums + per-workgroup partials
        sums = buf[:4]
        losses = torch.empty(2, dtype=torch.float32, device=x.device)
        n_total = x.numel()
        world = 1
        _lib.check(lib.neraf_stft_loss_sums(_lib.ctx(dev), x.data_ptr(), y.data_ptr(), x.numel(), loss_type,
                                            sums.data_ptr(), _stream_ptr()), dev)
        if group is not None:
            # data parallel: the Frobenius ratio / mean are over the GLOBAL batch (NeRAF_evaluator.py:26)
            from .parallel import allreduce_loss_sums
            import torch.distributed as dist
            n_total = allreduce_loss_sums(sums, x.numel(), group=group if group is not True else None, uniform_shards=True)
            world = dist.get_world_size(group if group is not True else None)
        _lib.check(lib.neraf_stft_loss_finalize(_lib.ctx(dev), sums.data_ptr(), n_total,
                                                weights.data_ptr() if weights is not None else None, losses.data_ptr(), _stream_ptr()), dev)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, y, sums)
        ctx.loss_type, ctx.dev, ctx.n_total, ctx.weights, ctx.world = loss_type, dev, n_total, weights, world
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_sc: torch.Tensor, g_mag: torch.Tensor):
        x, y, sums = ctx.saved_tensors
        lib = _lib.load()
        dx = torch.empty_like(x)
        # upstream scalars (GradScaler scale), the loss factors and the world size are combined inside the kernel: no host sync, no
        # scalar launches.  Data parallel: the loss value is already the GLOBAL one (sums all-reduced in the forward), so this rank's dx is
        # dL_global/dx_local, and the sum over ranks of the resulting parameter gradients is dL_global/dtheta.  The gradient reducer
        # AVERAGES over ranks (right for the per-rank-mean radiance losses): the factor `world` makes the averaged gradient of the
        # audio loss the single-process global-batch gradient, not 1/world of it.
        g_sc = g_sc.float().reshape(()) if g_sc is not None else None
        g_mag = g_mag.float().reshape(()) if g_mag is not None else None
        _lib.check(lib.neraf_stft_loss_bwd(_lib.ctx(ctx.dev), x.data_ptr(), y.data_ptr(), x.numel(), ctx.n_total, ctx.loss_type,
                                           sums.data_ptr(), g_sc.data_ptr() if g_sc is not None else None,
                                           g_mag.data_ptr() if g_mag is not None else None,
                                           ctx.weights.data_ptr() if ctx.weights is not None else None, float(ctx.world), dx.data_ptr(),
                                           _stream_ptr()), ctx.dev)
        return dx, None, None, None, None


class SpectralConvergenceLoss(nn.Module):
    """NeRAF_evaluator.py:8-26 (takes magnitudes).  Kept for API parity; STFTLoss uses the fused kernel."""

    def forward(self, x_mag, y_mag):
        sc, _ = _StftLossFn.apply(torch.log(x_mag + 1e-3), torch.log(y_mag + 1e-3), 0)
        return sc


class LogSTFTMagnitudeLoss(nn.Module):
    """NeRAF_evaluator.py:29-53."""

    def __init__(self, loss_type="l1"):
        super().__init__()
        self.loss_type = loss_type

    def forward(self, x_log, y_log):
        _, mag = _StftLossFn.apply(x_log, y_log, 1 if self.loss_type == "l1" else 0)
        return mag


class STFTLoss(nn.Module):
    """NeRAF_evaluator.py:76-108."""

    def __init__(self, loss_type="l1", process_group=None):
        """``process_group``: None = single process; True = default group; or a group handle.  With a
        group, per-rank Frobenius sums are all-reduced (RCCL) so the loss equals the reference's on the
        concatenated global batch."""
        super().__init__()
        if loss_type not in ("l1", "mse"):
            raise ValueError("loss_type must be 'l1' or 'mse'")
        self.loss_type = loss_type
        self.process_group = process_group

    def forward(self, x_log, y_log, weights=None):
        """``weights`` (optional, device tensor [2]): factors applied to (sc, mag) inside the fused node."""
        sc, mag = _StftLossFn.apply(x_log, y_log, 1 if self.loss_type == "l1" else 0, self.process_group, weights)
        return {"audio_sc_loss": sc, "audio_mag_loss": mag}

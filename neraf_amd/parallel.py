"""Data-parallel plumbing for the NeRAF hot path: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) across xGMI.  The reference has no multi-GPU support at all
(NeRAF_pipeline.py:153-157 raises); this is the build's own design:

  * rays and RIR slices shard contiguously over ranks (``shard_range``); weights / voxel grid are replicated;
  * parameter gradients are all-reduced in a few large flat buckets (xGMI rings are per-link bound, so few
    large messages beat many small ones) -- ``allreduce_gradients``;
  * the spectral-convergence loss is a ratio of Frobenius norms over the GLOBAL batch
    (NeRAF_evaluator.py:26), so the per-rank partial sums are all-reduced before the ratio is taken --
    ``allreduce_loss_sums`` (used by neraf_amd.losses.STFTLoss(process_group=...)).

Everything here is backend-agnostic torch.distributed code; the CPU test-suite runs it with gloo, world 2.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n items for ``rank``; sizes differ by at most one, every item covered once."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_gradients(params: Iterable[torch.Tensor], world: Optional[int] = None, group=None,
                        bucket_bytes: int = 64 << 20, average: bool = True) -> int:
    """Sum (or average) ``p.grad`` over the group in flat buckets of about ``bucket_bytes``.  Returns #buckets."""
    if world is None:
        world = dist.get_world_size(group)
    grads: List[torch.Tensor] = [p.grad for p in params if p.grad is not None]
    buckets, cur, cur_bytes = [], [], 0
    for g in grads:
        nb = g.numel() * g.element_size()
        if cur and (cur_bytes + nb > bucket_bytes or g.dtype != cur[0].dtype):
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(g)
        cur_bytes += nb
    if cur:
        buckets.append(cur)
    for b in buckets:
        flat = torch._utils._flatten_dense_tensors(b)
        dist.all_reduce(flat, group=group)
        if average:
            flat.div_(world)
        for g, s in zip(b, torch._utils._unflatten_dense_tensors(flat, b)):
            g.copy_(s)
    return len(buckets)


def allreduce_loss_sums(sums: torch.Tensor, n_local: int, group=None, uniform_shards: bool = False) -> int:
    """In-place all-reduce of the STFT-loss partial sums {sum (ymag-xmag)^2, sum ymag^2, sum |x-y|^p, -};
    returns the global element count.  With ``uniform_shards`` every rank is known to hold ``n_local`` bins and
    the count needs no communication (and no host sync); otherwise the counts are all-reduced too."""
    dist.all_reduce(sums, group=group)
    if uniform_shards:
        return n_local * dist.get_world_size(group)
    t = torch.tensor([n_local], dtype=torch.int64, device=sums.device)
    dist.all_reduce(t, group=group)
    return int(t.item())


def finalize_stft_loss(sums: torch.Tensor, n_total: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """{sc, mag} from (global) sums -- the arithmetic of neraf_stft_loss_finalize, for host-side checks."""
    return torch.sqrt(sums[0]) / torch.sqrt(sums[1]), sums[2] / n_total

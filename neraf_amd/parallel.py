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


def _coalesce(grads: List[torch.Tensor]) -> Tuple[List[Tuple[torch.Tensor, List[torch.Tensor]]], List[torch.Tensor]]:
    """Split gradient tensors into (a) maximal runs that tile one contiguous range of a shared storage -- e.g. the ResNet3D
    gradients, which are views of one flat buffer -- returned as (flat view over the run, members), reducible IN PLACE with one
    collective and no copies, and (b) the rest."""
    by_storage = {}
    for g in grads:
        if g.is_contiguous():
            by_storage.setdefault(g.untyped_storage().data_ptr(), []).append(g)
    runs, rest, used = [], [], set()
    for members in by_storage.values():
        members.sort(key=lambda t: t.storage_offset())
        i = 0
        while i < len(members):
            j, end = i, members[i].storage_offset() + members[i].numel()
            while j + 1 < len(members) and members[j + 1].dtype == members[i].dtype and members[j + 1].storage_offset() == end:
                j += 1
                end += members[j].numel()
            if j > i:
                first = members[i]
                n = end - first.storage_offset()
                runs.append((first.as_strided((n,), (1,), first.storage_offset()), members[i:j + 1]))
                used.update(id(t) for t in members[i:j + 1])
            i = j + 1
    rest = [g for g in grads if id(g) not in used]
    return runs, rest


class GradientReducer:
    """Overlapped data-parallel gradient averaging for the joint step.

    The parameters are given as GROUPS in the order the backward pass finishes them (NAcF, ResNet3D, radiance field, proposal
    networks).  A post-accumulate hook counts the gradients of each group; the moment a group is complete its all-reduce is
    launched asynchronously, so the collectives of the audio branch (~150 MB) travel over xGMI while the radiance backward is
    still computing.  Few, large, in-place messages: runs of gradients that are views of one flat buffer are reduced as one
    tensor without copies, tensors above ``direct_bytes`` individually, and only the small remainder through a flattened bucket.
    ``finish()`` (before the optimizer step) waits, writes the bucketed remainder back and re-arms the hooks.  Groups whose
    parameters received no gradient this step (the proposal networks between their update steps) send nothing -- the schedule
    is the same on every rank."""

    def __init__(self, groups: List[List[torch.nn.Parameter]], group=None, direct_bytes: int = 1 << 20, overlap: bool = True,
                 compress_bytes: Optional[int] = None):
        """``compress_bytes`` (default off, EXPERIMENTAL; ``NERAF_DP_COMPRESS_MB`` in the pipeline): gradient tensors of at least that many
        bytes travel as bfloat16 -- the 49 MB radiance hash-table gradient is the one collective of the step the backward cannot hide
        (it is produced last), and half the bytes are half the exposed time on the per-link-bound xGMI rings.  bfloat16 keeps fp32's
        range (the gradients are GradScaler-scaled).  Precision: every rank rounds its contribution to 8 significant bits on the way
        in, and a ring all-reduce on a bfloat16 tensor ACCUMULATES in bfloat16 -- world - 1 further roundings, so the error of the
        average grows with the world size.  Replicas stay bit-identical (an all-reduce returns the same bits on every
        rank).  Tested on gloo at world 2 only, unmeasured on RCCL at any world size: not for production runs until it is."""
        self.compress_bytes = compress_bytes
        self.groups = [list(g) for g in groups]
        self.pg = group
        self.world = dist.get_world_size(group)
        self.direct_bytes = direct_bytes
        self.overlap = overlap
        self._avg = dist.get_backend(group) == "nccl"       # RCCL averages in the collective; gloo has no AVG: sum, then divide
        self._count = [0] * len(self.groups)
        self._launched = [False] * len(self.groups)
        self._pending = []       # (work handle, tensor to divide or None, [(grad, source slice)] to copy back)
        self._hooks = []
        # the hooks act only while armed: a training loop that owns the reducer (NeRAFPipeline.train_iteration) arms it for its own
        # backward pass and disarms it after finish(), so that a backward OUTSIDE the loop (an evaluation of gradients, a test) does
        # not launch asynchronous in-place collectives that nobody waits for.  Stand-alone use: armed from construction.
        self.armed = True
        if overlap:
            for gi, params in enumerate(self.groups):
                for p in params:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(gi)))

    def _make_hook(self, gi: int):
        def hook(_param):
            if not self.armed:
                return
            self._count[gi] += 1
            if self._count[gi] == len(self.groups[gi]) and not self._launched[gi]:
                self._launch(gi)
        return hook

    def notify_group(self, gi: int):
        """For engines that assign a group's gradients themselves instead of routing them through autograd (the ResNet3D
        backward): the group's gradients are final, launch its collectives now."""
        if self.armed and not self._launched[gi]:
            self._launch(gi)

    def _launch(self, gi: int):
        self._launched[gi] = True
        grads = [p.grad for p in self.groups[gi] if p.grad is not None]
        if not grads:
            return
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        runs, rest = _coalesce(grads)
        small = []
        for flat, _members in runs:
            self._pending.append((dist.all_reduce(flat, op=op, group=self.pg, async_op=True), None if self._avg else flat, []))
        for g in rest:
            nbytes = g.numel() * g.element_size()
            if self.compress_bytes is not None and g.dtype == torch.float32 and g.is_contiguous() and nbytes >= self.compress_bytes:
                g16 = g.to(torch.bfloat16)                   # the collective moves half the bytes; finish() widens the result back into g
                self._pending.append((dist.all_reduce(g16, op=op, group=self.pg, async_op=True), None if self._avg else g16, [(g, g16)]))
            elif g.is_contiguous() and nbytes >= self.direct_bytes:
                self._pending.append((dist.all_reduce(g, op=op, group=self.pg, async_op=True), None if self._avg else g, []))
            else:
                small.append(g)
        by_dtype = {}
        for g in small:
            by_dtype.setdefault(g.dtype, []).append(g)
        for bucket in by_dtype.values():
            flat = torch._utils._flatten_dense_tensors(bucket)
            back = list(zip(bucket, torch._utils._unflatten_dense_tensors(flat, bucket)))
            self._pending.append((dist.all_reduce(flat, op=op, group=self.pg, async_op=True), None if self._avg else flat, back))

    def finish(self) -> int:
        """Launch whatever has not been launched (incomplete groups, or everything when ``overlap`` is off), wait for all
        collectives, finalise, and re-arm.  Returns the number of collectives of this step."""
        for gi in range(len(self.groups)):
            if not self._launched[gi]:
                self._launch(gi)
        n = len(self._pending)
        for work, div, back in self._pending:
            work.wait()
            if div is not None:
                div.div_(self.world)
            for g, src in back:
                g.copy_(src)
        self._pending = []
        self._count = [0] * len(self.groups)
        self._launched = [False] * len(self.groups)
        return n

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


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


class _GatherShards(torch.autograd.Function):
    """Differentiable assembly of a tensor whose LAST dimension is sharded over the ranks: rank r holds ``local`` = full[..., lo:hi].

    Forward: every rank writes its shard into zeros and the buffers are summed (x + 0 = x exactly, so the result is bit-identical
    on every rank whatever the reduction order -- an all-gather that tolerates uneven shards).

    Backward, under this package's data-parallel convention (``GradientReducer`` AVERAGES parameter gradients over ranks, and every
    rank's upstream gradient is world x its share of the global one, see ``_StftLossFn.backward``): the parameters behind
    ``local`` receive on rank r ``sum_over_ranks(d)[..., lo:hi]``, so that the average over ranks of the resulting parameter
    gradients equals the single-process gradient through the full tensor."""

    @staticmethod
    def forward(ctx, local: torch.Tensor, lo: int, hi: int, total: int, group):
        full = torch.zeros(local.shape[:-1] + (total,), dtype=local.dtype, device=local.device)
        full[..., lo:hi] = local
        dist.all_reduce(full, group=group)
        ctx.lo, ctx.hi, ctx.group = lo, hi, group
        return full

    @staticmethod
    def backward(ctx, d: torch.Tensor):
        d = d.contiguous().clone()
        dist.all_reduce(d, group=ctx.group)
        return d[..., ctx.lo:ctx.hi].contiguous(), None, None, None, None


def gather_shards(local: torch.Tensor, lo: int, hi: int, total: int, group=None) -> torch.Tensor:
    """See ``_GatherShards``: full[..., total] from per-rank shards [..., lo:hi]; differentiable; identical on every rank."""
    return _GatherShards.apply(local, lo, hi, total, group)

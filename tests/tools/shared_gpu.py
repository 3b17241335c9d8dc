"""Two-rank GPU tests on a box with ONE GPU share it between two processes.  Round 3 found kernels of two processes resident on one
MI355X disturbing each other's BatchNorm statistics (profiles/r03_gpu_sharing_bisect.txt) and wrapped these tests in three attempts and
an XFAIL.  Round 4: the accumulators are read where the producing atomics executed (csrc/resnet3d_common.h ``stat_ld``), and the
wrapper is gone from the default path -- ONE attempt, a failure is a failure.  ``NERAF_SHARED_GPU_RETRIES=n`` (n > 1) restores the
retry loop for diagnosing a box (it still FAILS when every attempt fails; nothing turns into XFAIL)."""
import os

import torch


def one_gpu_per_rank(world: int = 2) -> bool:
    return torch.cuda.device_count() >= world


def rank_device(rank: int, world: int = 2) -> int:
    return rank if one_gpu_per_rank(world) else 0


def attempts_for_shared_gpu(attempt):
    """attempt(i) runs the ranks and asserts; returns whatever it returns."""
    n = int(os.environ.get("NERAF_SHARED_GPU_RETRIES", "1"))
    if one_gpu_per_rank() or n <= 1:
        return attempt(0)
    last = None
    for i in range(n):
        try:
            return attempt(i)
        except AssertionError as e:      # noqa: PERF203 -- diagnosis aid only
            last = e
            print(f"[shared GPU] attempt {i + 1} of {n} failed: {str(e)[:600]}")
    raise last

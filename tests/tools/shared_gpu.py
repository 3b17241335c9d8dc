"""Two-rank GPU tests on a box with ONE GPU share it between two processes.  Round 3 found that on this pool kernels of two processes
resident on one MI355X at the same time can disturb each other (profiles/r03_gpu_sharing_bisect.txt: single
workgroups read a small buffer an earlier kernel of their own stream wrote as it was BEFORE that kernel; never seen with one process per
GPU; strongly box-dependent -- 0 of 20 runs on most boxes, 10 of 12 on one).  That is not what these tests are about, so on a
one-GPU box they get a few attempts, and if every attempt is disturbed they report XFAIL with the last failure instead of FAIL.
With two GPUs (one per rank -- the deployment model) there is one attempt and a failure is a failure."""
import pytest
import torch


def one_gpu_per_rank(world: int = 2) -> bool:
    return torch.cuda.device_count() >= world


def rank_device(rank: int, world: int = 2) -> int:
    return rank if one_gpu_per_rank(world) else 0


def attempts_for_shared_gpu(attempt, attempts: int = 3):
    """attempt(i) runs the ranks and asserts; returns whatever it returns."""
    if one_gpu_per_rank():
        return attempt(0)
    last = None
    for i in range(attempts):
        try:
            return attempt(i)
        except AssertionError as e:      # noqa: PERF203 -- a disturbed run; try again
            last = e
            print(f"[shared GPU] attempt {i + 1} of {attempts} failed: {str(e)[:600]}")
    pytest.xfail(f"all {attempts} attempts on a SHARED GPU were disturbed (known effect of two processes on one GPU of this pool, "
                 f"DESIGN.md section 6; tools/share_gpu_regression.sh): {str(last)[:1500]}")

"""Worker of tests/test_gpu_dp2.py::test_rccl_world_size_one_...: the data-parallel plumbing on the REAL backend ("nccl" = RCCL on
ROCm) with a group of ONE rank -- what a one-GPU box can execute of it: ReduceOp.AVG inside the collective, the in-place all-reduce of
the ResNet3D's flat gradient buffer, the hook order of the overlapped reducer, the sharded refresh assembled by ``gather_shards`` and
the STFT loss on all-reduced sums.  The same iterations are run first WITHOUT any group; with one rank every collective is the
identity, so the two runs must agree (bit for bit under NERAF_DETERMINISTIC=1, to rounding of the atomics' order otherwise).

    python tests/tools/nccl1_worker.py <out.json>
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def run(world_group: bool, dev):
    import bench
    torch.manual_seed(0)
    js = bench.JointStep(dev, 512, 256, 1, rotate=2, camera_opt=False)     # no ray-gradient atomics: the run is bit-reproducible
    n_coll = []
    if world_group:
        js.am.process_group = True
        js.am.dp_single_rank_collectives = True
        js.am.criterion.process_group = True
        red = js.pipe.attach_gradient_reducer()
        orig = red.finish
        red.finish = lambda: n_coll.append(orig()) or n_coll[-1]
    losses = []
    for _ in range(3):
        js.i += 1
        loss, ld = js.pipe.train_iteration(js.i, js.optimizers, js.scaler)
        losses.append({k: float(v) for k, v in ld.items()})
        if world_group:
            assert js.am._dp_world() is not None
    torch.cuda.synchronize()
    params = {n: p for n, p in list(js.vm.named_parameters()) + [("am." + n, p) for n, p in js.am.named_parameters()]}
    return {"losses": losses, "digest": {n: digest(p) for n, p in params.items()}, "grid": digest(js.am.grid),
            "scale": js.scaler.get_scale(), "collectives_per_step": n_coll}, params


def main():
    out_path = sys.argv[1]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    a, pa = run(False, dev)
    pa = {n: p.detach().clone() for n, p in pa.items()}
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # RCCL
    assert dist.get_backend() == "nccl"
    t = torch.arange(8, dtype=torch.float32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.AVG)                                   # the op the reducer uses: must exist in this RCCL build
    assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
    b, pb = run(True, dev)
    rel = {}
    for n in pa:
        d = (pb[n].detach().double() - pa[n].double()).norm() / (pa[n].double().norm() + 1e-30)
        rel[n] = float(d)
    dist.destroy_process_group()
    json.dump({"plain": a, "nccl": b, "rel": rel, "deterministic": os.environ.get("NERAF_DETERMINISTIC", "0")}, open(out_path, "w"))


if __name__ == "__main__":
    main()

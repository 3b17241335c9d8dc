#!/usr/bin/env python3
"""Longer run of the bench workload (synthetic batch, fixed): the losses must fall, the GradScaler scale must stay put (no
skipped steps), every parameter must stay finite."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
js = bench.JointStep(torch.device("cuda:0"), 4096, 2048, 1, rotate=1)
hist = []
for i in range(n):
    loss, ld = js.pipe.train_iteration(js.i + 1, js.optimizers, js.scaler); js.i += 1
    if i % 50 == 0 or i == n - 1:
        hist.append((i, float(loss), {k: float(v) for k, v in ld.items()}, js.scaler.get_scale()))
for h in hist:
    print(f"step {h[0]:4d} loss {h[1]:.6f} scale {h[3]:.0f} " + " ".join(f"{k}={v:.3e}" for k, v in h[2].items()))
bad = [n_ for n_, p in list(js.vm.named_parameters()) + list(js.am.named_parameters()) if not bool(torch.isfinite(p).all())]
print("non-finite parameters:", bad)
print("optimizer steps taken per parameter group:", [[float(o.group_steps(gi).max()) for gi in range(len(o.param_groups))] for o in js.optimizers], "of", n)

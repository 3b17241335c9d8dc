#!/usr/bin/env python3
"""Which torch operators (not this package's HIP kernels) still launch device work inside one joint training step, and from where:
torch.profiler over 3 steps of the bench workload, aten ops that own a device kernel grouped by the innermost neraf_amd / bench frame."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench

js = bench.JointStep(torch.device("cuda:0"), 4096, 2048, 1)
for _ in range(30):
    js.step()
torch.cuda.synchronize()
N = 6
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(N):
        js.step()
    torch.cuda.synchronize()

evs = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.kernels), key=lambda e: e.time_range.start)
# one step = N-th part of the list; print the last step in launch order: aten leaf ops with shapes, own kernels by name only
leaf = []
for e in evs:
    if any(c.kernels for c in e.cpu_children):
        continue
    leaf.append(e)
per = len(leaf) // N
tot = 0.0
n_aten = 0
for e in leaf[-per:]:
    k = e.kernels[0]
    if e.name.startswith("aten::"):
        n_aten += len(e.kernels)
        tot += sum(x.duration for x in e.kernels)
        site = next((fr for fr in (e.stack or []) if "neraf_amd" in fr or "bench.py" in fr), "")
        print(f"  ATEN {e.name:22s} {str(e.input_shapes)[:56]:56s} {sum(x.duration for x in e.kernels):5.1f} us  {site[-70:]}")
    else:
        print(f"{e.name[:60]:60s} -> {k.name[:70]}")
print("aten launches in the step:", n_aten, "sum", tot, "us")

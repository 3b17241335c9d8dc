#!/usr/bin/env python3
"""Per-step host issue time and GPU time (events) of the first N steps of the bench workload: shows warm-up transients
(hipGraph captures, allocator growth, optimizer plan builds) that a short --warmup would put inside the timed region."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from neraf_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
js = bench.JointStep(torch.device("cuda:0"), 4096, 2048, 1)
lib = _lib.load()


def captures():
    c, l = C.c_int(0), C.c_int(0)
    lib.neraf_graph_stats(_lib.ctx(0), C.byref(c), C.byref(l))
    return c.value


ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host, marks = [], []
torch.cuda.synchronize()
ev[0].record()
for i in range(n):
    t = time.perf_counter()
    js.step()
    host.append(1e3 * (time.perf_counter() - t))
    marks.append((captures(), torch.cuda.memory_reserved() >> 20, torch.cuda.memory_stats().get("num_device_alloc", 0)))
    ev[i + 1].record()
torch.cuda.synchronize()
gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for i in range(0, n, 10):
    print(f"steps {i:3d}-{i+9:3d}: host " + " ".join(f"{h:5.1f}" for h in host[i:i + 10]) + "  | gpu " + " ".join(f"{g:5.1f}" for g in gpu[i:i + 10]))
prev = None
for i, m in enumerate(marks):
    if m != prev:
        print("step", i, "graph captures", m[0], "reserved MiB", m[1], "device allocs", m[2])
        prev = m

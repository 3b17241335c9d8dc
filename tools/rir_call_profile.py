#!/usr/bin/env python3
"""Where the ~210 us of ONE audio eval call (get_outputs_for_camera(None, None, batch): NeRAF_model.py:610-728) go: cProfile over 300
calls on the bench's eval model, plus the same call with the device work removed from the critical path (no .cpu())."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
er = bench.EvalRender(dev)
for k in range(20):
    er.rir(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(300):
    er.rir(k % 32)
torch.cuda.synchronize()
print("us per call", (time.perf_counter() - t0) / 300 * 1e6)
pr = cProfile.Profile()
pr.enable()
for k in range(300):
    er.rir(k % 32)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)

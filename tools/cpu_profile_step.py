#!/usr/bin/env python3
"""Host-side cost of one training step: cProfile over N steps of bench.JointStep with the GPU queue kept async
(one synchronise at the end), printed by cumulative and by own time."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    torch.cuda.set_device(0)
    js = bench.JointStep(torch.device("cuda:0"), 4096, 2048, 1)
    for _ in range(8):
        js.step()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        js.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host issue time per step {1e3*(t1-t0)/n:.3f} ms ; wall per step incl. drain {1e3*(t2-t0)/n:.3f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        js.step()
    pr.disable()
    torch.cuda.synchronize()
    for key in ("cumulative", "tottime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
        print(s.getvalue()[:9000])

if __name__ == "__main__":
    main()

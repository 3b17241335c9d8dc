#!/usr/bin/env python3
"""Platform check, no code of this package involved: a dependent chain of plain torch kernels (a += 1 alternating with a strided
read-modify-write, result known exactly) run by two processes on ONE GPU at the same time.  A wrong result means kernels of one
stream did not execute in order under GPU sharing.    python tools/share_gpu_torch_chain.py [--pair]"""
import argparse, os, subprocess, sys, time
ap = argparse.ArgumentParser()
ap.add_argument("--pair", action="store_true"); ap.add_argument("--tag", default="A"); ap.add_argument("--seconds", type=float, default=25.0)
a = ap.parse_args()
if a.pair:
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--tag", t, "--seconds", str(a.seconds)]) for t in ("A", "B")]
    sys.exit(max(p.wait() for p in ps))
import torch
dev = torch.device("cuda:0")
N, K = 1 << 19, 64
x = torch.zeros(N, device=dev)
y = torch.zeros(N, device=dev)
bad = rounds = 0
t0 = time.time()
while time.time() - t0 < a.seconds:
    x.zero_(); y.zero_()
    for i in range(K):
        x += 1.0                    # kernel 2i
        y.copy_(x.flip(0))          # kernel 2i+1 reads ALL of x (reversed): an early start sees a mix of i and i+1
        x += y - (i + 1.0)          # kernel 2i+2: adds 0 if y == i+1 everywhere
    torch.cuda.synchronize()
    rounds += 1
    if not bool((x == float(K)).all()):
        bad += 1
        print(f"[{a.tag}] round {rounds}: {int((x != float(K)).sum())} of {N} elements wrong (min {float(x.min())} max {float(x.max())})", flush=True)
print(f"[{a.tag}] {bad} of {rounds} rounds wrong", flush=True)

#!/usr/bin/env python3
"""Does GPU sharing with another process disturb the radiance half?  Evaluates the SAME radiance step (fixed weights, rays, jitters) N
times and compares every output and gradient with the first evaluation: deterministic kernels must agree bit for bit (hash-table
gradients are integer sums), the others to fp32 summation noise.  Run alone and next to a second GPU process (`--noise` starts
one: a matmul loop) -- two independent training processes sharing a GPU were seen to corrupt each other's radiance training.

    python tools/contention_race_probe.py [--noise] [--iters 300] [--rays 256]"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

ap = argparse.ArgumentParser()
ap.add_argument("--noise", action="store_true")
ap.add_argument("--noise-only", action="store_true")
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--rays", type=int, default=256)
a = ap.parse_args()

import torch

if a.noise_only:
    x = torch.randn(4096, 4096, device="cuda:0", dtype=torch.float16)
    t0 = time.time()
    while time.time() - t0 < 170:
        for _ in range(20):
            y = x @ x
        torch.cuda.synchronize()
    sys.exit(0)

noise = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--noise-only"]) if a.noise else None
import numpy as np
import trajectory_common as TC
from neraf_amd.vision import NeRAFVisionModel, RayBundle

dev = torch.device("cuda:0")
cfg = dict(TC.CFG); cfg["R"] = a.rays
vm = NeRAFVisionModel(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), cfg["n_cam"])
P, _, _ = TC.initial_weights((vm.proposal_networks[0].table.shape[0], vm.proposal_networks[1].table.shape[0], vm.field.module.table.shape[0]))
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for i in range(2):
        vm.proposal_networks[i].table.copy_((torch.rand(P[f"prop{i}.table"].shape, generator=g) - 0.5) * 0.5)
        vm.proposal_networks[i].w0.copy_(P[f"prop{i}.w0"]); vm.proposal_networks[i].w1.copy_(P[f"prop{i}.w1"])
    f = vm.field.module
    f.table.copy_((torch.rand(P["field.table"].shape, generator=g) - 0.5) * 0.5)
    for k in ("base_w0", "base_w1", "head_w0", "head_w1", "head_w2", "embedding"):
        getattr(f, k).copy_(P["field." + k])
vm.to(dev).train()
TC.CFG["R"] = a.rays
b = TC.ray_batch(3)
bundle = RayBundle(b["origins"].to(dev), b["directions"].to(dev), b["camera_indices"].to(dev))
jit = [j.reshape(-1).to(dev) for j in b["jitters"]]
gt = {"image": b["rgb"].to(dev)}
names = ["field.table", "base_w0", "base_w1", "head_w0", "head_w1", "head_w2", "embedding", "p0.table", "p0.w0", "p0.w1", "p1.table", "p1.w0", "p1.w1"]


def evaluate():
    vm.update_to_step(5)                      # step < 10: the proposal networks are updated on every step
    vm.zero_grad(set_to_none=True)
    out = vm.get_outputs(bundle, jitters=jit)
    ld = vm.get_loss_dict(out, gt, {})
    (ld["rgb_loss"] + ld["interlevel_loss"] + ld["distortion_loss"]).backward()
    res = {"rgb": out["rgb"].detach().clone(), "density": out["density"].detach().clone(), "w0": out["weights_list"][0].clone(),
           "losses": torch.stack([ld[k].detach() for k in ("rgb_loss", "interlevel_loss", "distortion_loss")])}
    for n, p in zip(names, vm.loss_params()):
        res["g." + n] = p.grad.detach().clone() if p.grad is not None else torch.zeros(1, device=dev)
    return res


ref = evaluate()
torch.cuda.synchronize()
bad = {}
for it in range(a.iters):
    r = evaluate()
    for k in ref:
        if not torch.equal(r[k], ref[k]):
            d = float((r[k].double() - ref[k].double()).abs().max())
            rel = d / (float(ref[k].double().abs().max()) + 1e-30)
            e = bad.setdefault(k, [0, 0.0])
            e[0] += 1; e[1] = max(e[1], rel)
            if rel > 1e-3:
                print(f"iteration {it}: {k} differs: max |d| {d:.3e} (rel to max {rel:.2e})", flush=True)
torch.cuda.synchronize()
print("noise process:", "on" if a.noise else "off")
for k in ref:
    print(f"  {k:14s} mismatching evaluations {bad.get(k, [0, 0])[0]:4d} / {a.iters}, worst relative deviation {bad.get(k, [0, 0.0])[1]:.2e}")
if noise is not None:
    noise.kill()

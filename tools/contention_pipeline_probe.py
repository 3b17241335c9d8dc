#!/usr/bin/env python3
"""Full-pipeline version of contention_race_probe.py: after 20 training iterations of the trajectory scenario, evaluates the SAME
iteration's loss dict + backward (no optimizer step; grid and refresh cursor restored) N times and reports, per parameter tensor, the
worst relative deviation of its gradient from the first evaluation.  `--pair` runs two such processes at once on the one GPU."""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
ap = argparse.ArgumentParser()
ap.add_argument("--pair", action="store_true")
ap.add_argument("--iters", type=int, default=150)
ap.add_argument("--tag", default="A")
a = ap.parse_args()
if a.pair:
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--iters", str(a.iters), "--tag", t]) for t in ("A", "B")]
    sys.exit(max(p.wait() for p in ps))
import numpy as np, torch
import trajectory_common as TC
dev = torch.device("cuda:0"); torch.manual_seed(0)
curves, img, stft, pipe, evb = TC.run_hip_trajectory(dev, steps=20)
vm, am = pipe.model, pipe.audio_model
params = [(n, p) for n, p in list(vm.named_parameters()) + list(am.named_parameters())]
grid0, cur0 = am.grid.detach().clone(), am.grid_batch_i
scaler = torch.amp.GradScaler("cuda", enabled=False)
STEP = 25


def evaluate():
    with torch.no_grad():
        am.grid.copy_(grid0)
    am.mark_grid_written(); am.grid_batch_i = cur0
    vm.update_to_step(STEP)
    for _, p in params:
        p.grad = None
    _, ld, _ = pipe.get_train_loss_dict(STEP)
    sum(ld.values()).backward()
    torch.cuda.synchronize()
    out = {"loss." + k: v.detach().clone() for k, v in ld.items()}
    for n, p in params:
        out[n] = p.grad.detach().clone() if p.grad is not None else None
    return out


ref = evaluate()
worst = {}
for it in range(a.iters):
    r = evaluate()
    for k, v in r.items():
        if v is None or ref[k] is None:
            continue
        den = float(ref[k].double().norm()) + 1e-30
        rel = float((v.double() - ref[k].double()).norm()) / den
        if rel > worst.get(k, 0.0):
            worst[k] = rel
        if rel > 0.2 and "resnet3d" not in k:      # the encoder's gradients move 20-45 % between evaluations on their own (ReLU-gate chaos)
            print(f"[{a.tag}] evaluation {it}: {k} deviates by rel-L2 {rel:.3f}", flush=True)
for name, sel in (("encoder (resnet3d)", lambda k: "resnet3d" in k), ("everything else", lambda k: "resnet3d" not in k)):
    top = sorted(((k, v) for k, v in worst.items() if sel(k)), key=lambda kv: -kv[1])[:8]
    print(f"[{a.tag}] worst relative deviations over {a.iters} evaluations, {name}:", [(k[-34:], round(v, 5)) for k, v in top], flush=True)

#!/usr/bin/env python3
"""ResNet3D forward (train-mode BatchNorm) of one fixed grid, repeated: relative deviation of the 1024-feature from the first
evaluation.  Alone the only source is the order of the BatchNorm-statistic atomics (1e-3 level through flipped ReLU gates); `--pair`
runs two such processes at once on the one GPU."""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--pair", action="store_true"); ap.add_argument("--iters", type=int, default=400); ap.add_argument("--tag", default="A")
ap.add_argument("--size", type=int, default=64); ap.add_argument("--graphs", default=None)
a = ap.parse_args()
if a.graphs is not None:
    os.environ["NERAF_GRAPHS"] = a.graphs
if a.pair:
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--iters", str(a.iters), "--tag", t, "--size", str(a.size)] + (["--graphs", a.graphs] if a.graphs else [])) for t in ("A", "B")]
    sys.exit(max(p.wait() for p in ps))
import numpy as np, torch
from neraf_amd import synth
from neraf_amd.resnet3d import ResNet3D_helper
dev = torch.device("cuda:0")
net = ResNet3D_helper(in_channels=7, backbone="resnet50", pretrained=False, grid_step=1.0 / a.size, N_features=1024)
net.backbone_net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.resnet3d_state_dict(7).items()})
net.to(dev).train()
x = torch.from_numpy(synth.uniform("probe.grid", (1, 7, a.size, a.size, a.size), 0.0, 1.0)).to(dev)
import ctypes as C
from neraf_amd import _lib
lib = _lib.load()
bb = net.backbone_net


def ws_tensor(kind, index, dtype):
    off, rows, cols = C.c_size_t(), C.c_int(), C.c_int()
    assert lib.neraf_resnet3d_debug_locate(C.byref(bb._desc), kind, index, C.byref(off), C.byref(rows), C.byref(cols)) == 0
    nbytes = rows.value * cols.value * torch.empty((), dtype=dtype).element_size()
    return bb._ws[off.value:off.value + nbytes].view(dtype).reshape(rows.value, cols.value)


def snapshot():
    """pre-BN outputs of all 43 convolutions (kind 5) and their published batch statistics (kind 6)"""
    return [ws_tensor(5, i, torch.float16).clone() for i in range(43)], [ws_tensor(6, i, torch.float32).clone() for i in range(43)]


def snapshot_acts():
    """post-activation tensors a1 / a2 / out of the 13 blocks (kinds 0, 1, 2)"""
    return [[ws_tensor(k, b, torch.float16).clone() for k in (0, 1, 2)] for b in range(13)]


reported = 0
with torch.no_grad():
    ref = net(x).flatten().clone()
    ref_pre, ref_stat = snapshot()
    ref_act = snapshot_acts()
    devs = []
    for it in range(a.iters):
        f = net(x).flatten()
        d = float((f - ref).norm() / ref.norm())
        devs.append(d)
        if d > 0.05:
            print(f"[{a.tag}] evaluation {it}: feature deviates by rel-L2 {d:.3f}", flush=True)
            if reported < 3:
                reported += 1
                pre, stat = snapshot()
                rel = [float((p.float() - r.float()).norm() / (r.float().norm() + 1e-30)) for p, r in zip(pre, ref_pre)]
                rels = [float((p - r).norm() / (r.norm() + 1e-30)) for p, r in zip(stat, ref_stat)]
                act = snapshot_acts()
                arel = [[round(float((p.float() - r.float()).norm() / (r.float().norm() + 1e-30)), 4) for p, r in zip(pb, rb)] for pb, rb in zip(act, ref_act)]
                print(f"[{a.tag}]   activations (a1, a2, out) per block: {arel}", flush=True)
                first = next((i for i, v in enumerate(rel) if v > 0.02), None)
                firsts = next((i for i, v in enumerate(rels) if v > 0.02), None)
                print(f"[{a.tag}]   first conv whose pre-BN output deviates > 2e-2: {first}; first BN whose statistics deviate: {firsts}; "
                      f"pre-BN deviations conv 0..{min(43, (first or 0) + 6)}: {[round(v, 4) for v in rel[:(first or 0) + 6]]}; stats: {[round(v, 4) for v in rels[:(firsts or 0) + 4]]}", flush=True)
devs = np.array(devs)
print(f"[{a.tag}] {a.size}^3 feature deviation over {a.iters} forwards: median {np.median(devs):.2e} p99 {np.quantile(devs, 0.99):.2e} max {devs.max():.2e}", flush=True)

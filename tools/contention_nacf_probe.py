#!/usr/bin/env python3
"""NAcF field forward + STFT loss + backward of one fixed batch, repeated: relative deviation of every parameter gradient from the
first evaluation (alone: 1e-6 level, the order of the bias-gradient atomics).  `--pair` runs two such processes on the one GPU.
What tests/test_gpu_dp2.py saw under GPU sharing: the layer-0 weight gradient (db0 (x) feat, outer_kernel reading the bias gradient
that seg_copy_kernel wrote two launches earlier) off by 50-100 % in about one run of four."""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--pair", action="store_true"); ap.add_argument("--iters", type=int, default=300); ap.add_argument("--tag", default="A")
ap.add_argument("--batch", type=int, default=96)
a = ap.parse_args()
if a.pair:
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--iters", str(a.iters), "--tag", t, "--batch", str(a.batch)]) for t in ("A", "B")]
    sys.exit(max(p.wait() for p in ps))
import numpy as np, torch
from neraf_amd import synth
from neraf_amd.field import NeRAFAudioSoundField
from neraf_amd.losses import STFTLoss
dev = torch.device("cuda:0")
C_, F_, T_ = 1, 513, 60
torch.manual_seed(0)
field = NeRAFAudioSoundField(in_size=1024 + 163, W=512, sound_rez=C_, N_frequencies=F_).to(dev)
full = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth.audio_batch(a.batch, C_, F_, T_, tag="dp2.audio").items()}
feat = torch.from_numpy(synth.uniform("dp2.feat", (1024,), 0.0, 2.0)).to(dev)
aabb = torch.from_numpy(np.asarray(synth.audio_aabb(), dtype=np.float32)).to(dev)
loss_fn = STFTLoss("mse")
names = [n for n, _ in field.named_parameters()]


def grads():
    for p in field.parameters():
        p.grad = None
    y = field.forward_queries(feat, full["time_query"], full["mic_pose"], full["source_pose"], full["rot"], aabb, T_)
    l = loss_fn(y, full["data"])
    (l["audio_sc_loss"] * 1e-4 + l["audio_mag_loss"] * 1e-3).backward()
    return [p.grad.detach().clone() for p in field.parameters()]


ref = grads()
bad = 0; worst = [0.0] * len(ref)
for it in range(a.iters):
    g = grads()
    rel = [float((x - r).norm() / (r.norm() + 1e-30)) for x, r in zip(g, ref)]
    worst = [max(w, v) for w, v in zip(worst, rel)]
    if max(rel) > 1e-3:
        bad += 1
        if bad <= 5:
            k = int(np.argmax(rel))
            extra = ""
            if k == 0:      # split the layer-0 weight gradient into its outer-product part and its GEMM part
                nf = 1024
                extra = (f"; columns < {nf} (db0 (x) feat): {float((g[0][:, :nf] - ref[0][:, :nf]).norm() / ref[0][:, :nf].norm()):.3e}, "
                         f"columns >= {nf} (GEMM): {float((g[0][:, nf:] - ref[0][:, nf:]).norm() / ref[0][:, nf:].norm()):.3e}; "
                         f"rows of the outer part that are wrong: {int(((g[0][:, :nf] - ref[0][:, :nf]).abs().amax(1) > 1e-3 * ref[0][:, :nf].abs().max()).sum())} of {g[0].shape[0]}")
            print(f"[{a.tag}] evaluation {it}: {names[k]} deviates by {rel[k]:.3e}{extra}", flush=True)
print(f"[{a.tag}] {bad} of {a.iters} backward passes with a gradient off by more than 1e-3; worst per parameter: "
      + ", ".join(f"{n} {w:.1e}" for n, w in zip(names, worst) if w > 1e-5), flush=True)

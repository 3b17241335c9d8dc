#!/usr/bin/env python3
"""CPU oracle (fp32), radiance half only, on the trajectory scene with R rays per iteration: held-out PSNR against ground truth every
--every iterations.  Counterpart of `tools/long_trajectory_curve.py --rays R --start-audio 1000000`: does the held-out quality fall
with the ray batch size in the REFERENCE arithmetic too?     python tools/oracle_batch_size_check.py R steps [--every N]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
ap = argparse.ArgumentParser()
ap.add_argument("rays", type=int); ap.add_argument("steps", type=int); ap.add_argument("--every", type=int, default=100); ap.add_argument("--threads", type=int, default=4)
a = ap.parse_args()
import numpy as np, torch
torch.set_num_threads(a.threads)
import trajectory_common as TC
from neraf_amd import synth
from oracle.trainer import OracleTrainer
cfg = TC.CFG
cfg.update(R=a.rays, start_step_audio=10 ** 9)
P, sdn, sdr = TC.initial_weights()
tr = OracleTrainer(P, sdn, sdr, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), TC.T(synth.audio_aabb()), cfg["grid_step"], cfg["T"], cfg["start_step_audio"], cfg["R"])
ev = synth.trajectory_eval_camera(*cfg["eval_hw"], tag=cfg["tag"])
t0 = time.time()
for s in range(a.steps):
    res = tr.train_iteration(s, TC.ray_batch(s), None)
    if (s + 1) % a.every == 0:
        img = tr.render(TC.T(ev["origins"]), TC.T(ev["directions"])).reshape(*cfg["eval_hw"], 3).numpy()
        print(f"oracle R={a.rays} iteration {s + 1}: rgb {res['rgb_loss']:.5f}  held-out PSNR {TC.psnr(img, np.asarray(ev['image'])):.2f} dB  ({time.time() - t0:.0f} s)", flush=True)

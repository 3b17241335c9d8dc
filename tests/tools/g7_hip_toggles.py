#!/usr/bin/env python3
"""Which part of the HIP engine owns the distance of its G7 image from the fp32 oracle's (34 dB, against 38-39 dB for every 16-bit rounding
source the oracle can emulate, profiles/r05_g7_gap_attribution.txt)?  The deterministic G7 run repeated with one engine knob changed at a
time; prints PSNR(run image, oracle image) and the batch-statistics STFT rel-L2.     python tests/tools/g7_hip_toggles.py"""
import os, subprocess, sys, tempfile
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(__file__))
import numpy as np
import trajectory_common as TC
g = np.load(os.path.join(ROOT, "tests", "golden", "g7_trajectory.npz"))
tmp = tempfile.mkdtemp()
variants = [("baseline (deterministic)", {}), ("GradScaler scale 2^12", {"NERAF_TRAJ_INIT_SCALE": "4096"}), ("GradScaler scale 2^20", {"NERAF_TRAJ_INIT_SCALE": "1048576"}),
            ("hash gradient by global atomics", {"NERAF_FIELD_OWNER_SCATTER": "0"}),
            ("default mode (fp32 atomics everywhere)", {"NERAF_DETERMINISTIC": "0"})]
for name, envx in variants:
    out = os.path.join(tmp, "run.npz")
    env = dict(os.environ, NERAF_DETERMINISTIC="1", HSA_ENABLE_IPC_MODE_LEGACY="0"); env.update(envx)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "trajectory_worker.py"), "g7_trajectory", out], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if p.returncode:
        print(f"{name:42s} FAILED: {p.stdout.decode(errors='replace')[-300:]}"); continue
    r = np.load(out)
    print(f"{name:42s} PSNR(run, oracle) {TC.psnr(r['image'], g['image']):6.2f} dB   PSNR vs GT {TC.psnr(r['image'], g['gt_image']):6.2f}   "
          f"STFT rel-L2 (batch stats) {TC.rel_l2(r['stft_batch_stats'], g['stft_batch_stats']):.4f}", flush=True)

#!/usr/bin/env python3
""""HIP deviates from the fp32 oracle 2.7x (G7, in MSE) more than 16-bit parameter rounding explains, and nothing says which
component owns the difference".  The G7 scenario (100 iterations) re-run in the CPU oracle with one rounding source switched on at a
time -- and with all of them -- against the committed fp32 run of tests/golden/g7_trajectory.npz:

    for p in acts16 resnet_grad_bf16 all16; do python tests/tools/gen_trajectory.py --scenario g7_trajectory --part $p --threads 8 --parts-dir DIR; done
    python tests/tools/g7_gap_attribution.py DIR [hip_run.npz]        ->  profiles/r05_g7_gap_attribution.txt

Prints, per probe, PSNR(probe image, oracle image) and the rel-L2 of the held-out log-STFTs (batch-statistics branch and eval branch)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(__file__))
import numpy as np
import trajectory_common as TC

g = np.load(os.path.join(ROOT, "tests", "golden", "g7_trajectory.npz"))
rows = [("params16 (fixture)", g["probe_image"], g["probe_stft_batch_stats"], g["probe_stft"])]
for name in ("acts16", "resnet_grad_bf16", "all16", "order", "enc_grad16", "all16+enc_grad16"):
    f = os.path.join(sys.argv[1], f"g7_trajectory.{name}.npz")
    if os.path.exists(f):
        p = np.load(f)
        rows.append((name, p["image"], p["stft_batch_stats"], p["stft"]))
hip = np.load(sys.argv[2]) if len(sys.argv) > 2 else None       # a HIP run of the scenario (image, stft_*): how far is IT from each probe?
print(f"{'probe':22s} {'PSNR(probe, oracle) dB':>24s} {'STFT rel-L2 batch stats':>26s} {'STFT rel-L2 eval branch':>26s}" + ("   PSNR(HIP, probe) dB" if hip is not None else ""))
if hip is not None:
    print(f"{'HIP run':22s} {TC.psnr(hip['image'], g['image']):24.2f} {TC.rel_l2(hip['stft_batch_stats'], g['stft_batch_stats']):26.4f} "
          f"{TC.rel_l2(hip['stft_eval'], g['stft']):26.4f}")
for name, img, sbs, sev in rows:
    print(f"{name:22s} {TC.psnr(img, g['image']):24.2f} {TC.rel_l2(sbs, g['stft_batch_stats']):26.4f} {TC.rel_l2(sev, g['stft']):26.4f}" +
          (f" {TC.psnr(hip['image'], img):21.2f}" if hip is not None else ""))

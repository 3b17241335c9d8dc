#!/usr/bin/env python3
"""The yardstick of the G9 gates (tests/test_gpu_trajectory.py::test_long_trajectory_metric_parity): T60 / EDT / C50 errors against
ground truth of the fp32 oracle's and of its precision probes' held-out predictions in tests/golden/g9_long.npz, through the
package's evaluator (seeded Griffin-Lim on the GPU), and the spread max |probe - oracle| per metric.  Needs the GPU (the evaluator),
not the HIP training pipeline: run it BEFORE the first HIP run of the scenario and write the spread into the test.

    python tests/tools/g9_probe_spread.py [scenario = g9_long | g10_long_pose]
"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(__file__))
import numpy as np, torch
import trajectory_common as TC
from neraf_amd import synth
from neraf_amd.model import NeRAFAudioModel, NeRAFAudioModelConfig

SC = sys.argv[1] if len(sys.argv) > 1 else "g9_long"
g = np.load(os.path.join(ROOT, "tests", "golden", SC + ".npz"))
cfg = TC.SCENARIOS[SC]
dev = torch.device("cuda:0")
am = NeRAFAudioModel(NeRAFAudioModelConfig(dataset="RAF", grid_step=1 / 64), TC.T(synth.audio_aabb())).to(dev)
evb = TC.rir_bank(cfg["n_rir_eval"], cfg["tag"] + ".eval")
probes = [str(p) for p in g["probes"]]
pre = lambda n: "probe_" if n == "params16" else f"probe_{n}_"
stfts = {"oracle": g["stft"], **{n: np.asarray(g[pre(n) + "stft"], np.float32) for n in probes}}
images = {"oracle": g["image"], **{n: g[pre(n) + "image"] for n in probes}}
m = TC.metric_table(am, stfts, evb, gt_image=g["gt_image"], images=images)
for name, row in m.items():
    print(f"{name:18s} PSNR {row['psnr_vs_gt_db']:6.2f} dB  T60 {row['audio_T60']:7.3f} %  EDT {row['audio_EDT']:.4f} s  C50 {row['audio_C50']:.3f} dB  "
          f"STFT rel-L2 vs GT {row['stft_rel_l2_vs_gt']:.4f}")
for k in ("psnr_vs_gt_db", "audio_T60", "audio_EDT", "audio_C50"):
    print(f"spread {k}: {max(abs(m[n][k] - m['oracle'][k]) for n in probes):.6f}")

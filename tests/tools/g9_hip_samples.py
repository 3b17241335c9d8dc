#!/usr/bin/env python3
"""How far apart do two HIP runs of the G9 scenario land?  N default-mode runs (fp32 atomics: summation order differs run to run) and
one deterministic run, each in a fresh process, through the same evaluator as tests/test_gpu_trajectory.py -- next to the oracle and
its probes.  The spread between the HIP runs is the chaos of the system itself, seen from the engine's side.

    python tests/tools/g9_hip_samples.py [N=3] [scenario = g9_long | g10_long_pose]
"""
import os, subprocess, sys, tempfile
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(__file__))
import numpy as np, torch
import trajectory_common as TC
from neraf_amd import synth
from neraf_amd.model import NeRAFAudioModel, NeRAFAudioModelConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
SC = sys.argv[2] if len(sys.argv) > 2 else "g9_long"
g = np.load(os.path.join(ROOT, "tests", "golden", SC + ".npz"))
cfg = TC.SCENARIOS[SC]
runs = {}
tmp = tempfile.mkdtemp()
for name, det in [("hip deterministic", "1")] + [(f"hip default #{i + 1}", "0") for i in range(n)]:
    out = os.path.join(tmp, name.replace(" ", "_").replace("#", "") + ".npz")
    env = dict(os.environ, NERAF_DETERMINISTIC=det, HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "trajectory_worker.py"), SC, out], env=env, cwd=ROOT, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    runs[name] = dict(np.load(out))
dev = torch.device("cuda:0")
am = NeRAFAudioModel(NeRAFAudioModelConfig(dataset="RAF", grid_step=1 / 64), TC.T(synth.audio_aabb())).to(dev)
evb = TC.rir_bank(cfg["n_rir_eval"], cfg["tag"] + ".eval")
probes = [str(p) for p in g["probes"]]
pre = lambda k: "probe_" if k == "params16" else f"probe_{k}_"
stfts = {**{k: v["stft_eval"] for k, v in runs.items()}, "oracle": g["stft"], **{k: np.asarray(g[pre(k) + "stft"], np.float32) for k in probes}}
images = {**{k: v["image"] for k, v in runs.items()}, "oracle": g["image"], **{k: g[pre(k) + "image"] for k in probes}}
m = TC.metric_table(am, stfts, evb, gt_image=g["gt_image"], images=images)
for name, row in m.items():
    print(f"{name:20s} PSNR {row['psnr_vs_gt_db']:6.2f} dB  T60 {row['audio_T60']:7.3f} %  EDT {row['audio_EDT']:.4f} s  C50 {row['audio_C50']:.3f} dB  "
          f"STFT rel-L2 vs GT {row['stft_rel_l2_vs_gt']:.4f}")

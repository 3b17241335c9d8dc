#!/usr/bin/env python3
"""BASELINE configs[4] at scale, one GPU: NeRAFPipeline.get_average_eval_image_metrics (NeRAF_pipeline.py:291-436) over a whole
synthetic test split -- F full-resolution RAF frames (684 x 1024, 22 chunks of 32,768 rays each) and R held-out RIRs, each RIR through
the audio model's eval branch (T = 60 time queries) AND the evaluator's metric chain (seeded Griffin-Lim, T60 / EDT / C50, STFT and
envelope errors), the reference's own throughput keys next to wall times.  The datasets do not ship: images and RIRs are synthetic
(device-resident, like the bench); the weights are random, so the metric VALUES mean nothing -- what is measured is the loop.

    python tools/full_eval_run.py [F=42] [R=1024]     # RAF Empty + Furnished: 2 x 21 test views at nerfstudio's 10 % split
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from neraf_amd import config as C
from neraf_amd.datamanagers import SyntheticAudioDataManager, SyntheticVisionDataManager

F = int(sys.argv[1]) if len(sys.argv) > 1 else 42
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
torch.manual_seed(0)
m = C.make_method("RAF", "FurnishedRoom", datamanager=SyntheticVisionDataManager(4, F, 684, 1024, 4096),
                  audio_datamanager=SyntheticAudioDataManager(4, R, batch_size=2048))
m.config.pipeline.start_step_audio = 3
p = m.config.pipeline.setup(device="cuda:0", test_mode="val", world_size=1, local_rank=0, grad_scaler=None)
p.eval()
# warm-up: one frame + a few RIRs worth of kernels / plans / FFT plans (a 2-frame, 8-RIR pipeline would need other managers: the first
# call below simply runs twice and the second is the one reported)
res = {}
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    met = p.get_average_eval_image_metrics(step=10)
    torch.cuda.synchronize(); res[rep] = (time.time() - t0, met)
wall, met = res[1]
keys = {k: (float(v) if np.isscalar(v) or getattr(v, "ndim", 1) == 0 else None) for k, v in met.items()}
out = {"frames": F, "rays_per_frame": 684 * 1024, "rirs": R, "wall_s_first_call": round(res[0][0], 3), "wall_s": round(wall, 3),
       "metrics": {k: v for k, v in keys.items() if v is not None}}
print(json.dumps(out))

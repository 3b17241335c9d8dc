import os, sys, numpy as np, torch
sys.path.insert(0, "tests/tools"); sys.path.insert(0, ".")
import trajectory_common as TC
TC.CFG["R"] = 256; TC.CFG["B"] = 64
cfg = dict(TC.CFG)
dev = torch.device("cuda:0"); torch.manual_seed(0)
curves, img, stft, pipe, evb = TC.run_hip_trajectory(dev, cfg=cfg)
from neraf_amd import synth
ev = synth.trajectory_eval_camera(*cfg["eval_hw"], tag=cfg["tag"])
print("single-process R=256: held-out PSNR vs GT %.2f  rgb tail %.5f" % (TC.psnr(img, ev["image"]), float(np.nanmean(curves[-20:, 0]))))

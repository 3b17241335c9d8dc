"""Worker of tests/test_gpu_trajectory.py: ONE single-process training run of a trajectory scenario in a fresh process, so that the
test can choose the library's summation mode (NERAF_DETERMINISTIC, read once per process by libneraf_hip) without touching the
rest of the GPU suite, which runs in the default (atomics) mode.

    python tests/tools/trajectory_worker.py <scenario> <out.npz>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import torch


def main():
    scenario, out_path = sys.argv[1], sys.argv[2]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import trajectory_common as TC
    torch.manual_seed(0)
    cfg = TC.SCENARIOS[scenario]
    fx = os.path.join(ROOT, "tests", "golden", scenario + ".npz")
    if os.path.exists(fx):          # evaluate the held-out RIRs the committed fixture holds (the bank is prefix-stable: RIR i is RIR i for any count)
        cfg = dict(cfg, n_rir_eval=int(np.load(fx)["stft"].shape[0]))
    # the 100-iteration fixtures stay below the GradScaler's growth interval (asserted); the long one runs through it like a real run
    curves, img, stft, pipe, evb = TC.run_hip_trajectory(dev, cfg=cfg, fixed_scale=cfg["steps"] <= 100)
    extra = {}
    if TC.SCENARIOS[scenario]["camera_opt"]:
        extra["pose"] = pipe.model.camera_optimizer.pose_adjustment.detach().cpu().double().numpy()
    tab = pipe.model.field.module.table.detach().cpu().numpy()
    w1 = pipe.audio_model.field.soundfield[1].weight.detach().cpu().numpy()
    cv = pipe.audio_model.resnet3d.backbone_net.layer3[5].conv3.weight.detach().cpu().numpy()
    np.savez(out_path, curves=curves, image=img, stft_eval=stft["eval"], stft_batch_stats=stft["batch_stats"], table=tab, nacf_w1=w1,
             conv=cv, deterministic=int(os.environ.get("NERAF_DETERMINISTIC", "0")), **extra)


if __name__ == "__main__":
    main()

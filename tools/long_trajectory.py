#!/usr/bin/env python3
"""The trajectory scenario of tests/test_gpu_trajectory.py (G7: box room, 12 cameras, 16 training RIRs, 512 rays + 128 RIR slices per
iteration) trained for LONGER than the 100 iterations a CPU oracle can follow: held-out PSNR and held-out RIR errors against GROUND
TRUTH after N iterations, for several N (each a fresh run from the fixture's initial weights).  No oracle here -- this shows that the
joint step keeps learning through the GradScaler's growth intervals and the schedulers, and what the eval branch (BatchNorm running
statistics) does once those have converged.      python tools/long_trajectory.py 100 1000 3000"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np, torch
import trajectory_common as TC
from neraf_amd import synth
dev = torch.device("cuda:0")
cfg = TC.CFG
for n in [int(a) for a in sys.argv[1:]] or [100, 1000, 3000]:
    t0 = time.time()
    curves, img, stft, pipe, evb = TC.run_hip_trajectory(dev, steps=n, fixed_scale=False)
    dt = time.time() - t0
    am = pipe.audio_model if hasattr(pipe, "audio_model") else pipe.model.audio_model
    ev = synth.trajectory_eval_camera(*cfg["eval_hw"], tag=cfg["tag"])
    gt_img = np.asarray(ev["image"])
    gt_stft = evb["log_mag"].numpy()           # as the fixture's gt_stft
    line = (f"{n:6d} iterations ({dt:5.1f} s): held-out PSNR vs GT {TC.psnr(img, gt_img):6.2f} dB; rgb loss tail {np.nanmean(curves[-20:, 0]):.5f}; "
            f"STFT rel-L2 vs GT: batch statistics {TC.rel_l2(stft['batch_stats'], gt_stft):.4f}, eval branch {TC.rel_l2(stft['eval'], gt_stft):.4f}; "
            f"eval vs batch-statistics branch {TC.rel_l2(stft['eval'], stft['batch_stats']):.4f}")
    ms = [TC.audio_metrics(am, stft["batch_stats"][i], evb, i) for i in range(cfg["n_rir_eval"])]
    keys = [k for k in ms[0] if any(t in k.lower() for t in ("t60", "edt", "c50"))]
    line += "; " + ", ".join(f"{k} {np.mean([float(m[k]) for m in ms]):.3f}" for k in keys)
    # did anything overflow on the way?  the GradScaler starts at 65536, doubles every 2000 clean steps and halves on a skipped one; the
    # ResNet3D backward's fp16 chain reports the scale groups that were not finite in its last recording pass
    import ctypes as C
    from neraf_amd import _lib
    bb = am.resnet3d.backbone_net
    e, amx, info = (C.c_int32 * 41)(), (C.c_float * 41)(), (C.c_int32 * 2)()
    _lib.load().neraf_resnet3d_bwd_chain_state(_lib.ctx(0), C.byref(bb._desc), bb._bws.data_ptr(), e, amx, 41, info,
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
    expect = 65536.0 * 2.0 ** (n // 2000)
    line += (f"; GradScaler scale {pipe._trajectory_scaler.get_scale():.0f} (no skipped step: {expect:.0f}); fp16 chain: {info[1]} passes, "
             f"unsettled groups {info[0]}, exponents {min(e)} .. {max(e)}")
    print(line, flush=True)

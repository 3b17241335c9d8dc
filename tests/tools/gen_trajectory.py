#!/usr/bin/env python3
"""Generate tests/golden/g7_trajectory.npz: the CPU oracle (oracle/trainer.py, fp32) trained for CFG['steps'] iterations on the
trajectory scenario (tests/tools/trajectory_common.py), then evaluated on a held-out camera and two held-out RIRs.

    python tests/tools/gen_trajectory.py [--scenario g7_trajectory|g8_trajectory_pose] [--threads 8] [--probe]
    python tests/tools/gen_trajectory.py --scenario g9_long --part main|params16|acts16|resnet_grad_bf16 --parts-dir DIR   (one run, one file)
    python tests/tools/gen_trajectory.py --scenario g9_long --merge --parts-dir DIR                                         (parts -> fixture)

Stored: per-iteration loss-dict curves, the rendered held-out image + its analytic ground truth, the predicted log-magnitude STFTs
[T, C, F] of the held-out RIRs + their ground truth, scalar summaries.  ``--probe`` additionally trains the SAME oracle with its
radiance and NAcF parameters rounded to fp16 in every forward (straight-through) and stores that run's outputs next to the fp32
ones: the distance between the two oracle runs is the sensitivity of this training trajectory to 16-bit parameter rounding alone --
the band inside which an fp16 engine (the HIP path here, tiny-cuda-nn + AMP in the reference) can be expected to land.
Runs only in the build container (minutes of CPU); the GPU test reads the committed fixture."""
import argparse
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(__file__))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import trajectory_common as TC  # noqa: E402
from oracle.trainer import OracleTrainer  # noqa: E402


def run(fp16_params, log, cfg):
    """``fp16_params``: False (the fp32 oracle), True (== "params16") or the name of a probe of OracleTrainer."""
    probe = "params16" if fp16_params is True else (fp16_params or None)
    fp16_params = probe is not None
    P, sdn, sdr = TC.initial_weights()
    tr = OracleTrainer(P, sdn, sdr, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), TC.T(TC.synth.audio_aabb()), cfg["grid_step"], cfg["T"],
                       cfg["start_step_audio"], cfg["R"], probe=probe,
                       num_cameras=cfg["n_cam"] if cfg.get("camera_opt") else 0)
    bank = TC.rir_bank(cfg["n_rir"], cfg["tag"] + ".train")
    keys = ["rgb_loss", "interlevel_loss", "distortion_loss", "audio_sc_loss", "audio_mag_loss", "loss", "proposal_updated"]
    curves = np.full((cfg["steps"], len(keys)), np.nan, np.float64)
    t0 = time.time()
    for s in range(cfg["steps"]):
        r = tr.train_iteration(s, TC.ray_batch(s), TC.audio_batch(s, bank))
        curves[s] = [r.get(k, np.nan) for k in keys]
        if s % 10 == 0 or s == cfg["steps"] - 1:
            log(f"[{probe or 'fp32'}] step {s} {time.time() - t0:.0f}s " +
                " ".join(f"{k}={r[k]:.5f}" for k in keys[:5] if k in r))
    ev = TC.synth.trajectory_eval_camera(*cfg["eval_hw"], tag=cfg["tag"])
    img = tr.render(TC.T(ev["origins"]), TC.T(ev["directions"])).reshape(*cfg["eval_hw"], 3).numpy()
    evb = TC.rir_bank(cfg["n_rir_eval"], cfg["tag"] + ".eval")
    stft = np.stack([tr.predict_rir(evb["mic_pose"][i], evb["source_pose"][i], evb["rot"][i]).numpy() for i in range(cfg["n_rir_eval"])])
    # the same queries with the encoder's BatchNorms on the grid's own statistics (as in training): separates what the running
    # statistics add -- 43 exponential averages with a ~10-iteration memory over weights that move every iteration -- from the rest
    stft_bs = np.stack([tr.predict_rir(evb["mic_pose"][i], evb["source_pose"][i], evb["rot"][i], batch_stats=True).numpy()
                        for i in range(cfg["n_rir_eval"])])
    log(f"  held-out PSNR vs ground truth {TC.psnr(img, ev['image']):.2f} dB; STFT rel-L2 vs ground truth "
        f"{float(np.linalg.norm(stft - evb['log_mag'].numpy()) / np.linalg.norm(evb['log_mag'].numpy())):.4f}")
    extra = {"pose": tr.pose.detach().numpy().copy()} if cfg.get("camera_opt") else {}
    if cfg.get("save_state"):      # the trained oracle (not committed): more held-out RIRs / views can be evaluated later without re-training
        torch.save({"P": {k: v.detach() for k, v in tr.P.items()}, "sdn": {k: v.detach() for k, v in tr.sdn.items()},
                    "sdr": {k: v.detach() for k, v in tr.sdr.items()}, "grid": tr.grid, "cursor": tr.cursor,
                    "pose": tr.pose.detach() if cfg.get("camera_opt") else None}, cfg["save_state"])
    return {**extra, "curves": curves, "image": img.astype(np.float32), "stft": stft.astype(np.float32), "stft_batch_stats": stft_bs.astype(np.float32),
            "keys": np.array(keys),
            "gt_image": ev["image"], "gt_stft": evb["log_mag"].numpy()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--probe", action="store_true")
    ap.add_argument("--scenario", default="g7_trajectory", choices=sorted(TC.SCENARIOS))
    ap.add_argument("--out", default=None)
    ap.add_argument("--part-name", default=None, help="file name of the part (default: --part); e.g. 'order' for a second fp32 run on another thread count")
    ap.add_argument("--part", default=None, choices=["main", "params16", "acts16", "resnet_grad_bf16", "all16", "enc_grad16", "all16+enc_grad16"],
                    help="run ONE oracle variant and write DIR/<scenario>.<part>.npz (long scenarios: one process per variant)")
    ap.add_argument("--parts-dir", default="/tmp/trajectory_parts")
    ap.add_argument("--merge", action="store_true", help="assemble the fixture from the part files in --parts-dir")
    ap.add_argument("--steps", type=int, default=None, help="override the scenario's iteration count (experiments only)")
    ap.add_argument("--save-state", action="store_true", help="--part runs: also write DIR/<scenario>.<part>.state.pt (the trained oracle)")
    a = ap.parse_args()
    if a.out is None:
        a.out = os.path.join(ROOT, "tests", "golden", a.scenario + ".npz")
    torch.set_num_threads(a.threads)
    log = lambda m: print(m, flush=True)      # noqa: E731
    cfg = dict(TC.SCENARIOS[a.scenario])
    if a.steps is not None:
        cfg["steps"] = a.steps
    if a.part is not None:
        os.makedirs(a.parts_dir, exist_ok=True)
        if a.save_state:
            cfg["save_state"] = os.path.join(a.parts_dir, f"{a.scenario}.{a.part_name or a.part}.state.pt")
        r = run(False if a.part == "main" else a.part, log, cfg)
        f = os.path.join(a.parts_dir, f"{a.scenario}.{a.part_name or a.part}.npz")
        np.savez_compressed(f, **r)
        log(f"wrote {f}")
        return
    if a.merge:
        m = np.load(os.path.join(a.parts_dir, f"{a.scenario}.main.npz"))
        out = {"steps": cfg["steps"], "R": cfg["R"], "B": cfg["B"], "start_step_audio": cfg["start_step_audio"],
               "camera_opt": int(bool(cfg.get("camera_opt"))), "keys": m["keys"], "curves": m["curves"], "image": m["image"],
               "stft": m["stft"], "stft_batch_stats": m["stft_batch_stats"], "gt_image": m["gt_image"],
               "gt_stft": m["gt_stft"].astype(np.float32)}
        if cfg["steps"] > 100 and m["stft"].shape[0] > 8:      # 16 held-out RIRs: the batch-statistics diagnostic is not part of the long fixture
            del out["stft_batch_stats"]
        probes = []
        for name in ("params16", "acts16", "resnet_grad_bf16", "all16", "order", "order2"):      # order / order2: the fp32 oracle on other thread counts
            f = os.path.join(a.parts_dir, f"{a.scenario}.{name}.npz")
            if not os.path.exists(f):
                continue
            p = np.load(f)
            probes.append(name)
            pre = "probe_" if name == "params16" else f"probe_{name}_"      # "probe_*" = the fp16-parameter probe, as in G7 / G8
            # 16 held-out RIRs: the probes' predictions as float16 (log-magnitudes in [-7, 3]: 5e-4 relative, three orders below what two
            # runs differ by); consumers widen them to float32 before the evaluator
            st = p["stft"].astype(np.float16) if (cfg["steps"] > 100 and p["stft"].shape[0] > 8) else p["stft"]
            out.update({pre + "curves": p["curves"].astype(np.float32) if cfg["steps"] > 100 and p["stft"].shape[0] > 8 else p["curves"],
                        pre + "image": p["image"], pre + "stft": st})
            if cfg["steps"] <= 100:       # the short fixtures compare the batch-statistics branch too; the long one gates on the eval branch
                out[pre + "stft_batch_stats"] = p["stft_batch_stats"]
            log(f"{name} vs fp32 oracle: image PSNR {TC.psnr(p['image'], m['image']):.2f} dB, STFT rel-L2 "
                f"{float(np.linalg.norm(p['stft'] - m['stft']) / np.linalg.norm(m['stft'])):.4f}")
        out["probes"] = np.array(probes)
        np.savez_compressed(a.out, **out)
        log(f"wrote {a.out} ({os.path.getsize(a.out) / 1e3:.0f} kB)")
        return
    main_run = run(False, log, cfg)
    out = {"steps": cfg["steps"], "R": cfg["R"], "B": cfg["B"], "start_step_audio": cfg["start_step_audio"],
           "camera_opt": int(bool(cfg.get("camera_opt"))), "keys": main_run["keys"],
           "curves": main_run["curves"], "image": main_run["image"], "stft": main_run["stft"],
           "stft_batch_stats": main_run["stft_batch_stats"], "gt_image": main_run["gt_image"],
           "gt_stft": main_run["gt_stft"].astype(np.float32)}
    if "pose" in main_run:
        out["pose"] = main_run["pose"]
    if a.probe:
        p = run(True, log, cfg)
        if "pose" in p:
            out["probe_pose"] = p["pose"]
        out.update({"probe_curves": p["curves"], "probe_image": p["image"], "probe_stft": p["stft"],
                    "probe_stft_batch_stats": p["stft_batch_stats"]})
        log(f"probe vs fp32 oracle: image PSNR {TC.psnr(p['image'], main_run['image']):.2f} dB, STFT rel-L2 "
            f"{float(np.linalg.norm(p['stft'] - main_run['stft']) / np.linalg.norm(main_run['stft'])):.4f}")
    np.savez_compressed(a.out, **out)
    log(f"wrote {a.out} ({os.path.getsize(a.out) / 1e3:.0f} kB)")


if __name__ == "__main__":
    main()

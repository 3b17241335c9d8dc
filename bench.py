#!/usr/bin/env python3
"""NeRAF hot-path benchmark on MI355X (contract: see the task brief / DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--dataset raf|soundspaces] [--rays R --slices B]
    python bench.py --mode eval [--steps K --warmup W --rirs M]         (BASELINE configs[4]: full eval render, see EvalRender)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A *step* is one pass of NeRAFPipeline.get_train_loss_dict (NeRAF_pipeline.py:166-222) + backward + optimizer steps over one
synthetic batch already resident in HBM, at the RAF FurnishedRoom training shape (4096 rays + 2048 RIR STFT slices of 513 bins,
NeRAF_config.py:57,87); ``config.workload`` states exactly which stages are inside the timed region.  One process per GPU; with
``--gpus N`` and no launcher in the environment the script starts the N ranks itself (before anything touches the GPU).
``--scaling weak`` (default): every rank owns --rays / --slices of its own; ``--scaling strong``: --rays / --slices are the GLOBAL
batch, split contiguously over the ranks.  Gradients and STFT-loss sums are all-reduced over RCCL.

Rank 0 prints ONE COMPACT JSON line (< 6 KB, asserted: ``compact_line``): whole-job field-samples/s (and rays/s, bins/s),
``roofline`` for the kernel family with the largest share of the step (HIP-event durations recorded inside the library over an
instrumented replay of the same steps), ``cpu_baseline`` (the CPU oracle on this host, bounded sample of the same workload),
``eval_render`` and ``parity`` in short.  The FULL record (every kernel family, every window, the notes) goes to the file the
line names under ``detail`` (gpurun_out/bench_detail_*.json) -- round 5 printed it all on stdout, 22.7 KB, and the driver could
not read the line.
"""
import argparse
import ctypes as C
import gc
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
HBM_PEAK_GBS = 8000.0       # HBM3E spec, same guide
L2_PEAK_GBS = 34500.0       # same guide, "L2 (per XCD)": 4 MiB per XCD, ~34.5 TB/s aggregate
# What bounds each byte-priced kernel family (ids: csrc/common.h PROF_*).  The proposal networks' hash tables are 1.5-1.7 MB: every
# XCD's 4 MiB L2 holds them whole (counter traffic 38.6 MB per frame launch against 896 MB of gathered entries, L2 hit 0.92), so
# their gathered bytes are priced against the aggregate L2 rate, not against HBM (round 4 printed frac 1.04 of 8 TB/s for them); the
# 24 MB main table is served from the Infinity Cache / HBM and stays priced against HBM.
BYTE_FAMILY_BOUND = {2: "l2", 5: "l2", 3: "hbm", 6: "hbm", 7: "hbm"}
LINE_LIMIT = 6144           # bytes of the one stdout line (the driver reads a bounded tail of stdout)
# The committed profiles the line cites, by NAME (round 5 took sorted(glob)[-1]: a stale file could silently become the evidence).
# Updated by hand when tools/gpu_profile.sh / tools/gpu_pmc.sh produce a new set for a new build.
PROFILE_REFS = {"train_stats": "r06_d_joint_step_kernel_stats.csv", "train_pmc": "r06_d_pmc_traffic.json",
                "eval_stats": "r06_d_eval_kernel_stats.csv", "eval_pmc": "r06_d_eval_pmc_traffic.json"}
DTYPE = "f16"      # every 16-bit tensor of the step is fp16 (the ResNet3D backward's gradient chain too since round 5: per-group power-of-two scales)
PRIME_STEPS = 8             # untimed set-up steps before the --warmup steps (see main)
CLOCK_STEPS = 120           # further untimed steps (~0.5 s) in the full run only: five consecutive 30-step windows of a fresh process read
                            # 4.40 / 4.48 / 4.40 / 4.32 / 4.29 ms -- the first ~100 steps run before clocks and caches settle
NACF_DENSE_FLOP_PER_SLICE_FWD = 40_836_464  # SURVEY.md 8(d), RAF head (C*F = 513)
RESNET_FWD_GFLOP = 94.72
C_, F_, T_ = 1, 513, 60


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=4096, help="rays per step: per GPU (weak) or global (strong) (NeRAF_config.py:87)")
    ap.add_argument("--slices", type=int, default=2048, help="RIR STFT slices per step: per GPU (weak) or global (strong) (NeRAF_config.py:57)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed windows of exactly --steps steps each, back to back (the headline value is the MEDIAN window; all of "
                         "them, min and max are reported)")
    ap.add_argument("--parity", choices=("off", "g9", "all"), default="g9",
                    help="in-process trajectory-parity run(s) after the timed region: g9 (default; one 1000-iteration HIP training on fixture "
                         "tests/golden/g9_long.npz, ~16 s), all (+ g10_long_pose: camera optimizer on), off")
    ap.add_argument("--no-parity", action="store_true", help="same as --parity off")
    ap.add_argument("--detail", default=None, help="where the full record goes (default gpurun_out/bench_detail_<mode>_n<N>.json)")
    ap.add_argument("--plain", action="store_true",
                    help="priming + warm-up + timed steps only (no second regime, no instrumented replay, no CPU baseline): the form that "
                         "runs under rocprofv3, so that its per-kernel totals divide by exactly PRIME_STEPS + warmup + steps")
    ap.add_argument("--mode", choices=("train", "eval"), default="train",
                    help="train (default): the joint training step, BASELINE's metric; eval: the no-grad eval render of configs[4] -- a step is "
                         "one 684x1024 frame through get_outputs_for_camera (22 chunks of 32768 rays) + --rirs RIRs through the audio eval branch")
    ap.add_argument("--rirs", type=int, default=32, help="eval mode: RIRs evaluated per step (one get_outputs_for_camera call each)")
    ap.add_argument("--rotate", type=int, default=16,
                    help="train mode: number of DISTINCT resident ray / slice batches the steps cycle through (1 = the same batch every step)")
    ap.add_argument("--no-eval-line", action="store_true", help="train mode: skip the short eval-render measurement (key eval_render)")
    ap.add_argument("--grid", type=int, choices=(64, 128, 256), default=128,
                    help="voxel-grid edge of the scene encoder (grid_step = 1/edge; the reference's and the metric's configuration: 128)")
    ap.add_argument("--n-features", type=int, choices=(1024, 2048), default=1024,
                    help="scene-feature size: 1024, or 2048 = resnet50 with layer4 (NeRAF_resnet3d.py:128-131); the metric's configuration: 1024")
    ap.add_argument("--dataset", choices=("raf", "soundspaces"), default="raf",
                    help="audio head shape: raf = 1 x 513 bins, T = 60 (BASELINE configs[1..2], the default and the metric's config); "
                         "soundspaces = 2 x 257 bins, T = 101 (configs[3]: globally 32768 rays + 6464 slices, i.e. per GPU 4096 + 808)")
    return ap.parse_args()


# ---- the stdout line ------------------------------------------------------------------------------------------------------------
def _sig(x, n=6):
    """Floats to n significant digits (bytes of the line), containers recursively; everything else as is."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float(f"{x:.{n}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_launch_us", "launches_per_step",
              "whole_step_mfma_frac", "rocprof_reference", "above_peak")


def _compact_roofline(r, n_families=6):
    if not isinstance(r, dict):
        return r
    out = _pick(r, _ROOF_KEYS)
    if "kernel" in out:
        out["kernel"] = _short(out["kernel"], 96)
    fams = sorted(r.get("all_kernel_families") or [], key=lambda k: -k.get("ms_per_step", 0.0))[:n_families]
    if fams:      # the largest families of the step, each with its own fraction of its own peak (all of them: the detail file)
        out["families"] = [{"k": _short(k["kernel"], 56), "ms": k["ms_per_step"], "bound": k["bound"], "frac": k["frac"]} for k in fams]
    return out


def compact_line(full: dict, detail: str = None, limit: int = LINE_LIMIT) -> str:
    """The ONE stdout line, built from the full record: the contract's keys, ``roofline`` / ``cpu_baseline`` / ``eval_render`` /
    ``parity`` reduced to their figures, ``detail`` = the file holding the full record.  Always shorter than ``limit`` bytes
    (optional parts are dropped in a fixed order if it is not; the contract's keys never are) -- asserted here and by
    tests/test_bench_line.py on canned records."""
    top = ("metric", "mode", "plain", "value", "unit", "n_gpus", "ranks_seen", "backend", "steps", "warmup", "ms_per_step", "ms_per_step_min",
           "ms_per_step_max", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "rays_per_s", "bins_per_s", "fps", "fps_audio",
           "ms_per_frame", "us_per_rir", "priming_steps", "steps_executed_in_process", "rirs_per_step")
    out = _pick(full, top)
    cfg = full.get("config") or {}
    out["config"] = {"workload": _short(cfg.get("workload", ""), 200),
                     **_pick(cfg, ("rays_per_gpu", "slices_per_gpu", "global_rays", "global_slices", "rays_per_frame", "rirs_per_step",
                                   "bins_per_rir", "parallelism", "distinct_resident_batches"))}
    if (cfg.get("encoder_grid", 128), cfg.get("encoder_features", 1024)) != (128, 1024):
        out["config"].update(_pick(cfg, ("encoder_grid", "encoder_features")))
    if "roofline" in full:
        out["roofline"] = _compact_roofline(full["roofline"])
    if isinstance(full.get("cpu_baseline"), dict):
        cb = full["cpu_baseline"]
        out["cpu_baseline"] = {**_pick(cb, ("value", "unit", "cores", "kind")), "sample": _short(cb.get("sample_short") or cb.get("sample", ""), 240)}
    ev = full.get("eval_render")
    if isinstance(ev, dict):
        e = _pick(ev, ("ms_per_step", "value", "ms_per_frame", "rays_per_s", "bins_per_s", "us_per_rir", "rirs_per_step", "error"))
        if isinstance(ev.get("roofline"), dict):
            e["roofline"] = _pick(ev["roofline"], ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "above_peak"))
        if isinstance(ev.get("cpu_baseline"), dict):
            e["cpu_baseline"] = _pick(ev["cpu_baseline"], ("value", "unit", "cores", "kind"))
        out["eval_render"] = e
    par = full.get("parity")
    if isinstance(par, dict):
        out["parity"] = _pick(par, ("fixture", "steps", "psnr_db", "psnr_db_oracle", "t60_err_pct", "t60_err_pct_oracle", "edt_err_s",
                                    "edt_err_s_oracle", "c50_err_db", "c50_err_db_oracle", "inside", "outside", "mode", "rule", "error"))
    if "hip_graphs" in full:
        out["hip_graphs"] = _pick(full["hip_graphs"], ("enabled", "captures", "launches"))
    out["detail"] = detail
    out = _sig(out)
    # never longer than `limit`: optional parts go first, in this order
    for drop in (None, ("roofline", "families"), ("eval_render", "cpu_baseline"), ("hip_graphs",), ("cpu_baseline", "sample"),
                 ("eval_render", "roofline"), ("parity",), ("eval_render",)):
        if drop is not None:
            d = out
            for k in drop[:-1]:
                d = d.get(k) if isinstance(d, dict) else None
            if isinstance(d, dict):
                d.pop(drop[-1], None)
        line = json.dumps(out, separators=(",", ":"))
        if len(line.encode()) < limit:
            break
    assert len(line.encode()) < limit, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step"):
        assert k in out, k
    return line


def emit(full: dict, a, mode: str, world: int):
    """Write the full record to the detail file, print the compact line (the LAST thing on stdout)."""
    path = a.detail or os.path.join(ROOT, "gpurun_out", f"bench_detail_{mode}_n{world}.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f)
        shown = os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path
    except OSError as e:             # a read-only tree must not cost the line
        import tempfile
        path = os.path.join(tempfile.gettempdir(), f"bench_detail_{mode}_n{world}.json")
        try:
            with open(path, "w") as f:
                json.dump(full, f)
            shown = path
        except OSError:
            shown = None
    sys.stdout.flush()
    print(compact_line(full, shown), flush=True)


def _spawn_ranks(a):
    """``python bench.py --gpus N`` without a launcher: start the N ranks with torch.distributed.run as a CHILD process (nothing in
    this process has touched the GPU yet -- only argparse and the standard library are loaded) and relay its output and exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def T(a):
    import numpy as np
    import torch
    return torch.from_numpy(np.ascontiguousarray(a))


class JointStep:
    """cfg3 (RAF FurnishedRoom joint) training step, NeRAFPipeline.get_train_loss_dict order (NeRAF_pipeline.py:175-199):
       1. NeRAFVisionModel.get_outputs on the ray batch (sampler, 2 proposal nets, 2 PDF resamplings, fused field
          query, composite) + get_loss_dict (rgb MSE, interlevel, distortion)
       2. audio_model.query_grid_one_batch: 4096 cells x 18 directions through the field, mean, slab write (data parallel: every
          rank queries 1/world of the cells and the shares are assembled)
       3. ResNet3D(7x128^3 grid) -> 1024 feature (train-mode BatchNorm)
       4. audio get_outputs (GPU prologue + NAcF MLP) -> STFT loss
       5. ONE backward over the summed loss dict: NAcF (all grads + d/d feature) -> ResNet3D backward (dgrad/wgrad GEMMs,
          BatchNorm backward) -> grid-window gradient -> refresh backward into the field; radiance half (loss grads,
          proposal backward, fused field backward, weight-grad GEMMs) -> [RCCL all-reduce] -> GradScaler + fused Adam on
          the radiance parameters (lr 1e-2) and the audio parameters (NAcF + ResNet3D + field, lr 1e-4) -> scheduler steps.

    ``R`` / ``B`` are THIS rank's rays / slices; ``tag_rank`` selects which synthetic shard it holds."""

    def __init__(self, dev, R, B, world, dataset="raf", start_step=20000, camera_opt=True, rotate=16, grid=128, n_features=1024):
        import torch
        from neraf_amd import synth
        from neraf_amd.config import NeRAFVisionModelConfig, CameraOptimizerConfig, SceneBox
        from neraf_amd.model import NeRAFAudioModel, NeRAFAudioModelConfig
        from neraf_amd.vision import RayBundle
        self.dev, self.R, self.B, self.world = dev, R, B, world
        rank = int(os.environ.get("RANK", "0"))
        # the vision model as NeRAF_config.py:94-98 configures it: nerfacto defaults, 32768-ray eval chunks, average_init_density 0.01,
        # camera_optimizer SO3xR3 (pose deltas of the 210 training cameras are applied to every ray bundle and trained)
        vcfg = NeRAFVisionModelConfig(camera_optimizer=CameraOptimizerConfig(mode="SO3xR3" if camera_opt else "off"))
        self.vm = vcfg.setup(scene_box=SceneBox(torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])), num_train_data=210, metadata={},
                             device=dev, grad_scaler=None, seed_points=None).to(dev)
        with torch.no_grad():      # trained-like table magnitudes (synthetic, same on every rank)
            g = torch.Generator(device="cpu").manual_seed(0)
            for p in [self.vm.field.module.table] + [pn.table for pn in self.vm.proposal_networks]:
                p.copy_((torch.rand(p.shape, generator=g) - 0.5).to(dev))
        # grid / n_features: the reference's defaults (NeRAF_config.py:102-103: 1/128, 1024) are the benchmark's; the other values its
        # constructor accepts (256^3 grid, layer4 -> 2048 features) are measurable with --grid / --n-features
        cfg = (NeRAFAudioModelConfig(dataset="RAF", grid_step=1 / grid, N_features=n_features) if dataset == "raf" else
               NeRAFAudioModelConfig(dataset="SoundSpaces", grid_step=1 / grid, N_features=n_features, max_len=T_, N_freq_stft=F_))
        self.am = NeRAFAudioModel(cfg, T(synth.audio_aabb()), process_group=True if world > 1 else None)
        self.am.field.load_state_dict({k: T(v) for k, v in synth.nacf_state_dict(n_features + 163, 512, C_, F_).items()})
        self.am.resnet3d.backbone_net.load_state_dict(
            {k: T(v) for k, v in synth.resnet3d_state_dict(7, layers=(3, 4, 6, 3) if n_features == 2048 else (3, 4, 6)).items()})
        self.am.to(dev)
        self.vm.train(); self.am.train()
        # `rotate` DISTINCT resident batches, cycled step by step: the reference draws a fresh ray batch and fresh slices every step
        # (NeRAF_pipeline.py:175, :187), so consecutive steps gather other hash-table rows and scatter into other gradient rows.
        # Batch k of rank r is a pure function of (k, r): tags "bench.rays.r{r}[.k]" / "bench.r{r}[.k]" (k = 0 keeps the round-3 tags).
        bundles, gts, batches = [], [], []
        for k in range(max(int(rotate), 1)):
            sfx = "" if k == 0 else f".{k}"
            rb = synth.ray_batch(R, tag=f"bench.rays.r{rank}{sfx}")
            bundles.append(RayBundle(T(rb["origins"]).to(dev), T(rb["directions"]).to(dev), T(rb["camera_indices"]).to(dev)))
            gts.append({"image": T(rb["rgb"]).to(dev)})
            batches.append({kk: T(v).to(dev) for kk, v in synth.audio_batch(B, C_, F_, T_, tag=f"bench.r{rank}{sfx}").items()})
        self.bundle, self.gt, self.batch = bundles[0], gts[0], batches[0]
        # the reference's pipeline object: get_train_loss_dict (NeRAF_pipeline.py:166-222) inside Trainer.train_iteration, with its
        # parameter groups / optimizers / schedulers (NeRAF_config.py:115-132; the field parameters are in "fields" AND
        # "audio_fields", :487)
        from neraf_amd.datamanagers import RotatingBatchDataManager
        from neraf_amd.pipeline import NeRAFPipeline
        self.dm = RotatingBatchDataManager(bundles, gts, R)
        self.adm = RotatingBatchDataManager([None] * len(batches), batches)
        self.pipe = NeRAFPipeline(self.vm, self.am, datamanager=self.dm, audio_datamanager=self.adm, start_step_audio=2000,
                                  world_size=world, local_rank=rank)
        self.opt_wrapper, self.scaler = self.pipe.make_optimizers(init_scale=65536.0, with_schedulers=True)
        self.optimizers = self.opt_wrapper.steppers
        if world > 1:
            # gradient averaging overlapped with the backward pass; every group is all-reduced, the ResNet3D's too: that keeps the
            # replicas bit-identical although BatchNorm statistics are summed with order-dependent fp32 atomics
            self.pipe.attach_gradient_reducer()
        self.i = start_step   # 20000: steady-state regime of the 400k-iteration schedule (anneal done, proposal nets every 6th step)

    def samples_per_step(self):
        return self.R + self.B * C_ * F_

    def pin_batch(self, i):
        """None: rotate through the resident batches (default); i: serve batch i every step (the round-3 fixed-batch form)."""
        self.dm.pin(i)
        self.adm.pin(i)

    def step(self):
        self.i += 1
        loss, _ = self.pipe.train_iteration(self.i, self.opt_wrapper, self.scaler)
        return loss


class EvalRender:
    """BASELINE configs[4] ("full eval render") on one GPU: what NeRAFPipeline.get_average_eval_image_metrics (NeRAF_pipeline.py:291-436)
    runs per eval item, without the metric code around it.  One STEP =
      * one 684 x 1024 RAF frame (data/RAF/*/transforms.json: 700,416 rays) through NeRAFVisionModel.get_outputs_for_camera(camera, None,
        eval=True) (NeRAF_model.py:70-79): camera -> rays -> 22 chunks of 32,768 rays (NeRAF_config.py:95), each chunk sampler -> proposal
        density x2 -> PDF resampling x2 -> fused field query -> composite, no grad, mean appearance embedding, rgb clipped;
      * `rirs` RIRs through NeRAFAudioModel.get_outputs_for_camera(None, None, batch_audio) (NeRAF_model.py:648-728), one call each as the
        reference's loop does (:355-362): T time queries -> prologue -> NAcF MLP -> [T,C,F] log-magnitudes + the per-channel panels,
        with the ResNet3D scene feature computed ONCE and cached (the grid is static in eval; the reference recomputes it per RIR).
    ``batched_rirs`` evaluates N RIRs as ONE N*T-row field call (NeRAFAudioModel.get_outputs_for_rirs) -- the engine-native form."""

    def __init__(self, dev, dataset="raf", n_cams=4, n_items=64):
        import torch
        from neraf_amd import synth
        from neraf_amd.config import NeRAFVisionModelConfig, CameraOptimizerConfig, SceneBox
        from neraf_amd.datamanagers import synthetic_cameras
        from neraf_amd.model import NeRAFAudioModel, NeRAFAudioModelConfig
        self.dev = dev
        rank = int(os.environ.get("RANK", "0"))
        vcfg = NeRAFVisionModelConfig(camera_optimizer=CameraOptimizerConfig(mode="SO3xR3"))
        self.vm = vcfg.setup(scene_box=SceneBox(torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])), num_train_data=210, metadata={},
                             device=dev, grad_scaler=None, seed_points=None).to(dev)
        with torch.no_grad():      # trained-like table magnitudes (synthetic, the training bench's)
            g = torch.Generator(device="cpu").manual_seed(0)
            for p in [self.vm.field.module.table] + [pn.table for pn in self.vm.proposal_networks]:
                p.copy_((torch.rand(p.shape, generator=g) - 0.5).to(dev))
        cfg = (NeRAFAudioModelConfig(dataset="RAF", grid_step=1 / 128) if dataset == "raf" else
               NeRAFAudioModelConfig(dataset="SoundSpaces", grid_step=1 / 128, max_len=T_, N_freq_stft=F_))
        self.am = NeRAFAudioModel(cfg, T(synth.audio_aabb()))
        self.am.field.load_state_dict({k: T(v) for k, v in synth.nacf_state_dict(1187, 512, C_, F_).items()})
        self.am.resnet3d.backbone_net.load_state_dict({k: T(v) for k, v in synth.resnet3d_state_dict(7).items()})
        self.am.to(dev)
        self.vm.eval(); self.am.eval()
        self.cams = synthetic_cameras(n_cams, tag=f"bench.eval.cams.r{rank}").to(dev)       # RAF intrinsics incl. OPENCV distortion
        self.H, self.W = self.cams.height, self.cams.width
        ab = synth.audio_batch(n_items, C_, F_, T_, tag=f"bench.eval.r{rank}", outside_frac=0.0)
        gt = T(synth.uniform(f"bench.eval.gt.r{rank}", (C_, F_, T_), -6.0, 1.0)).to(dev)       # eval item layout [C,F,T] (NeRAF_dataset.py:180-181)
        self.items = [{"mic_pose": T(ab["mic_pose"][i]).to(dev), "source_pose": T(ab["source_pose"][i]).to(dev),
                       "rot": T(ab["rot"][i]).to(dev), "data": gt} for i in range(n_items)]
        self.mic = T(ab["mic_pose"]).to(dev)
        self.src = T(ab["source_pose"]).to(dev)
        self.rot = T(ab["rot"]).to(dev)
        self.i = 0

    rays_per_frame = property(lambda self: self.H * self.W)
    bins_per_rir = property(lambda self: T_ * C_ * F_)

    def frame(self, k=None):
        k = self.i if k is None else k
        return self.vm.get_outputs_for_camera(self.cams[k % self.cams.size], None, eval=True)

    def rir(self, k):
        return self.am.get_outputs_for_camera(None, None, batch_audio=self.items[k % len(self.items)])

    def step(self, rirs):
        out = self.frame()
        for j in range(rirs):
            self.rir(self.i * rirs + j)
        self.i += 1
        return out

    def batched_rirs(self, n):
        idx = [(k % len(self.items)) for k in range(n)]
        return self.am.get_outputs_for_rirs(self.mic[idx], self.src[idx], self.rot[idx])


def eval_cpu_baseline(n_rays=4096):
    """The CPU oracle's eval forward on this host, bounded: `n_rays` rays strided over one 684x1024 frame through the eval-mode
    nerfacto forward (no jitter, mean embedding), ONE RIR (T rows) through the prologue + NAcF with the feature given, and the
    ResNet3D eval forward (BatchNorm on running statistics) once -- the reference recomputes that per RIR (NeRAF_model.py:680-684)."""
    import torch
    from neraf_amd import synth
    from neraf_amd.datamanagers import synthetic_cameras
    from oracle import audio as O
    from oracle import vision as V
    ncores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(ncores)
    spec = V.NerfactoSpec()
    tot = (spec.prop_grids[0].total, spec.prop_grids[1].total, spec.main_grid.total)
    P = {k: T(v) for k, v in synth.vision_params(tot, num_train_data=210, table_scale=0.5).items()}
    sdn = {k: T(v) for k, v in synth.nacf_state_dict(1187, 512, C_, F_).items()}
    sdr = {k: T(v) for k, v in synth.resnet3d_state_dict(7).items()}
    aabb = T(synth.audio_aabb())
    cam = synthetic_cameras(1, tag="bench.eval.cams.r0")
    rb = cam.generate_rays(0)
    n = len(rb)
    idx = torch.arange(0, n, max(n // n_rays, 1))[:n_rays]

    def once(fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    def render(ix):
        with torch.no_grad():
            V.nerfacto_forward(rb.origins[ix], rb.directions[ix], rb.camera_indices[ix, 0], P, spec, training=False)
    render(idx[:64])
    t_rays = once(lambda: render(idx))
    grid = O.reset_grid(1 / 128)
    box = {}

    def resnet():
        with torch.no_grad():
            box["feat"] = O.resnet3d_forward(grid.unsqueeze(0), sdr, train=False).flatten()
    t_resnet = once(resnet)
    ab = synth.audio_batch(1, C_, F_, T_, tag="bench.eval.r0", outside_frac=0.0)
    one = {"time_query": torch.arange(T_), "mic_pose": T(ab["mic_pose"]).expand(T_, 3), "source_pose": T(ab["source_pose"]).expand(T_, 3),
           "rot": T(ab["rot"]).expand(T_, 3)}

    def rir():
        with torch.no_grad():
            O.audio_get_outputs(one, box["feat"], sdn, aabb, T_)
    rir()
    t_rir = once(rir)
    bins = T_ * C_ * F_
    frame_s = t_rays * n / len(idx)
    return {"value": (n + bins) / (frame_s + t_rir), "unit": "field-samples/s", "cores": ncores, "kind": "port",
            "rays_per_s": len(idx) / t_rays, "bins_per_s_cached_feature": bins / t_rir, "bins_per_s_resnet_per_rir": bins / (t_rir + t_resnet),
            "seconds": {"rays_sample": round(t_rays, 3), "resnet3d_eval_fwd": round(t_resnet, 3), "one_rir": round(t_rir, 4)},
            "sample_short": "%d rays strided over one 684x1024 frame (time scaled to the frame) + 1 RIR with cached scene feature; torch-CPU fp32 "
                            "oracle, %d threads" % (len(idx), ncores),
            "sample": ("%d rays strided over one 684x1024 frame through the oracle's eval-mode nerfacto forward (1 warm-up at 64 rays + 1 timed), "
                       "one RIR = %d time queries through prologue + NAcF (1 warm-up + 1 timed) with the scene feature given, the ResNet3D eval "
                       "forward on 7x128^3 once; value = (frame rays + one RIR's bins) / (rays time scaled to the %d-ray frame + one RIR, cached "
                       "feature); torch-CPU fp32 oracle, %d threads") % (len(idx), T_, n, ncores)}


def _prof_families(lib, h, local, nprof, ref_rows):
    """Per-kernel-family records of the library's event profiler (neraf_prof_enable) after `nprof` instrumented steps."""
    from neraf_amd import _lib
    fams = []
    kid = 0
    while lib.neraf_prof_kernel_name(kid):
        ms, n, w, ex = C.c_double(), C.c_int(), C.c_double(), C.c_double()
        _lib.check(lib.neraf_prof_summary_ex(h, kid, C.byref(ms), C.byref(n), C.byref(w), C.byref(ex)), local)
        if n.value:
            name = lib.neraf_prof_kernel_name(kid).decode()
            is_bytes = kid in BYTE_FAMILY_BOUND     # gather / scatter kernels are priced in gathered bytes
            bound = BYTE_FAMILY_BOUND.get(kid, "mfma")
            peak = {"hbm": HBM_PEAK_GBS, "l2": L2_PEAK_GBS, "mfma": MFMA_PEAK_TFLOPS}[bound]
            rate = w.value / (ms.value * 1e-3) / (1e9 if is_bytes else 1e12) if ms.value > 0 else 0.0
            fam = {"kernel": name, "bound": bound, "launches_per_step": n.value / nprof,
                   "avg_us": ms.value * 1e3 / n.value, "ms_per_step": ms.value / nprof, "achieved": rate,
                   "unit": "GB/s" if is_bytes else "TFLOP/s", "peak": peak, "frac": rate / peak, "work_per_launch": w.value / n.value,
                   # `work` / `frac` are ALGORITHMIC (a conv, its dgrad and its wgrad = 2 dout^3 taps cin cout each, real
                   # channels and taps); `executed` is what the grid multiplied (padded K / channels / voxel rows, zero-page taps)
                   "executed_per_launch": ex.value / n.value, "executed_frac": (ex.value / (ms.value * 1e-3) / (1e9 if is_bytes else 1e12) / peak) if ms.value > 0 else 0.0,
                   "work_per_step": w.value / nprof}
            if ref_rows:
                rx = _family_regex(name)
                calls = sum(c for k, (c, _) in ref_rows.items() if rx.search(k))
                tot = sum(t for k, (_, t) in ref_rows.items() if rx.search(k))
                if calls:
                    fam["rocprof_avg_us"] = tot / calls / 1e3
                    fam["rocprof_frac"] = (w.value / n.value) / (tot / calls * 1e-9) / (1e9 if is_bytes else 1e12) / peak
            fams.append(fam)
        kid += 1
    return fams


GATHER_CEILING_GBS = 8600.0   # MI355X_MICROARCH.md "Indexed rows": 38 MB table, uniformly random rows served from the Infinity Cache


def measure_eval(er, steps, warmup, rirs, lib, h, local, sync, full=True):
    """Time `steps` eval steps (see EvalRender) after `warmup`; returns the measurement dict shared by `--mode eval` and the
    `eval_render` key of the training line."""
    import torch
    for _ in range(max(warmup, 1)):
        er.step(rirs)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        er.step(rirs)
    sync()
    el = time.perf_counter() - t0
    # the two halves on their own (the reference reports them separately: num_rays_per_sec / fps, num_rays_per_sec_audio / fps_audio)
    sync(); t1 = time.perf_counter()
    for _ in range(steps):
        er.frame()
        er.i += 1
    sync(); t_frames = time.perf_counter() - t1
    n_r = max(steps * rirs, 8)
    sync(); t2 = time.perf_counter()
    for k in range(n_r):
        er.rir(k)
    sync(); t_rirs = time.perf_counter() - t2
    nb = 32
    er.batched_rirs(nb)
    reps = max(steps, 4)
    sync(); t3 = time.perf_counter()
    for _ in range(reps):
        er.batched_rirs(nb)
    sync(); t_batched = time.perf_counter() - t3
    rays, bins = er.rays_per_frame, er.bins_per_rir
    out = {"ms_per_step": el / steps * 1e3, "value": (rays + rirs * bins) * steps / el,
           "rays_per_s": rays * steps / t_frames, "fps": steps / t_frames, "ms_per_frame": t_frames / steps * 1e3,
           "bins_per_s": bins * n_r / t_rirs, "fps_audio": n_r / t_rirs, "us_per_rir": t_rirs / n_r * 1e6,
           "batched_rirs": {"rirs_per_call": nb, "rows_per_call": nb * T_, "bins_per_s": bins * nb * reps / t_batched,
                            "rirs_per_s": nb * reps / t_batched, "us_per_rir": t_batched / (nb * reps) * 1e6,
                            "note": "NeRAFAudioModel.get_outputs_for_rirs: N RIRs as one N*T-row field call (device tensor out, no per-RIR panels)"},
           "rays_per_frame": rays, "chunks_per_frame": (rays + er.vm.eval_num_rays_per_chunk - 1) // er.vm.eval_num_rays_per_chunk,
           "bins_per_rir": bins, "rirs_per_step": rirs}
    if not full:
        return out, None
    # instrumented replay: per-kernel-family HIP-event durations over whole frames (and the RIRs of a step)
    lib.neraf_prof_enable(h, 1)
    nprof = min(steps, 4)
    for _ in range(nprof):
        er.step(rirs)
    torch.cuda.synchronize()
    fams = _prof_families(lib, h, local, nprof, {})
    lib.neraf_prof_enable(h, 0)
    return out, fams


def eval_roofline(fams, dataset="raf", world=1):
    """``roofline`` object of the eval render from the instrumented families: the committed counter traffic / rocprofv3 averages
    attached, the family with the largest share of the eval step as the headline (the frame form of the field query)."""
    pmc_file = os.path.join(ROOT, "profiles", PROFILE_REFS["eval_pmc"])
    pmc = json.load(open(pmc_file))["families"] if os.path.exists(pmc_file) and dataset == "raf" and world == 1 else {}
    ref_name, ref_rows = _rocprof_reference("eval_stats")
    for k in fams:
        t = pmc.get(k["kernel"])
        k["traffic"] = t["hbm_bytes_per_launch"] if t else None
        if k["bound"] == "hbm":
            k["frac_of_gather_ceiling"] = k["achieved"] / GATHER_CEILING_GBS
        rx = _family_regex(k["kernel"])
        calls = sum(c for n_, (c, _) in ref_rows.items() if rx.search(n_))
        tot = sum(t_ for n_, (_, t_) in ref_rows.items() if rx.search(n_))
        if calls:
            k["rocprof_avg_us"] = tot / calls / 1e3
        # HIP-event timing noise around a few-microsecond launch can read above a guide figure (round 4: 1.04 of the aggregate L2 rate);
        # flagged here, asserted only in tests/test_gpu_eval_bench.py (an exception here would cost the whole line)
        if k["frac"] > 1.0:
            k["above_peak"] = True
    if not fams:
        return None
    byte_fams = [k for k in fams if k["bound"] == "hbm"] or fams
    dom = max(byte_fams, key=lambda k: k["ms_per_step"])
    fq = next((k for k in fams if k["kernel"].startswith("field_query_kernel")), None)
    return {"bound": dom["bound"], "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"],
            "frac": dom["frac"], "traffic": dom.get("traffic"),
            "frac_of_gather_ceiling": dom.get("frac_of_gather_ceiling"), "gather_ceiling_gbs": GATHER_CEILING_GBS,
            "avg_launch_us": dom["avg_us"], "launches_per_step": dom["launches_per_step"],
            "algorithmic_work_per_launch": dom["work_per_launch"],
            "selection": "the HBM-priced family with the largest share of the eval step among the instrumented families",
            "durations": "HIP events on the launch stream, as recorded",
            "traffic_source": PROFILE_REFS["eval_pmc"] if pmc and dom.get("traffic") is not None else None,
            "rocprof_reference": ref_name, "above_peak": any(k.get("above_peak") for k in fams),
            "field_query_inference": fq,
            "algorithmic_bytes": "proposal density: samples x 5 levels x 8 corners x 4 B = 160 B/sample (352 samples/ray), tables L2-resident: "
                                 "priced against the aggregate L2 rate; field query: samples x 16 x 8 x 4 B = 512 B/sample (48 samples/ray), "
                                 "priced against HBM -- SURVEY 8(d): 80,896 B/ray",
            "all_kernel_families": fams}


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def cpu_baseline(R, B):
    """The CPU oracle (torch fp32; audio half pinned to the reference on G1-G5, radiance half unpinned) timed on this host: TWO
    complete steps at the FULL batch (mean) -- every stage the GPU step contains, at its full size, forward + backward + torch Adam --
    timed stage by stage after a warm-up of the same code path at 1/64 of the batch (thread-pool start-up, first-touch
    allocations; the per-step-constant ResNet3D forward + backward has no smaller size and is run twice, second run timed).  No
    extrapolation.  A full step is ~9-17 s on 64 threads, so SURVEY 8(d)'s 3 warm-up + >= 10 timed steps (~4 minutes) do not fit the
    ~20-30 s of host time a default run may spend here; `sample` says exactly what was timed.  Forward-only figures come from a
    second, no-grad pass of the same stages."""
    import torch
    from neraf_amd import synth
    from oracle import audio as O
    from oracle import vision as V
    ncores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(ncores)
    spec = V.NerfactoSpec()
    tot = (spec.prop_grids[0].total, spec.prop_grids[1].total, spec.main_grid.total)
    P = {k: T(v).requires_grad_(True) for k, v in synth.vision_params(tot, num_train_data=210, table_scale=0.5).items()}
    sdn = {k: T(v).requires_grad_(True) for k, v in synth.nacf_state_dict(1187, 512, C_, F_).items()}
    sdr = {k: (T(v).requires_grad_(True) if k.endswith("weight") or k.endswith("bias") else T(v)) for k, v in synth.resnet3d_state_dict(7).items()}
    aabb = T(synth.audio_aabb())
    grid = O.reset_grid(1 / 128)
    dirs = O.fixed_viewing_directions()
    vaabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    opt_v = torch.optim.Adam([v for v in P.values()], lr=1e-2, eps=1e-15)
    opt_a = torch.optim.Adam(list(sdn.values()) + [v for v in sdr.values() if v.requires_grad], lr=1e-4, eps=1e-15)

    def once(fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    def vision(n, train):
        rb = synth.ray_batch(n, tag="bench.rays.r0")

        def run():
            with torch.set_grad_enabled(train):
                ov = V.nerfacto_forward(T(rb["origins"]), T(rb["directions"]), T(rb["camera_indices"]), P, spec, training=True,
                                        jitters=[T(j) for j in rb["jitters"]])
                lv = V.vision_loss_dict(ov, T(rb["rgb"]), spec)
                if train:
                    opt_v.zero_grad(set_to_none=True)
                    (lv["rgb_loss"] + lv["interlevel_loss"] + lv["distortion_loss"]).backward()
                    opt_v.step()
        return run

    def refresh(n, train):
        coords = O.coordinates_to_render(1 / 128)[:n]

        def run():
            with torch.set_grad_enabled(train):
                ori = O.refresh_world_positions(coords, vaabb)
                # the 18 directions of every cell in ONE batched field call (direction-major, NeRAF_model.py:327-333)
                a, d_ = V.field_forward(ori.repeat(18, 1), dirs.repeat_interleave(n, dim=0), torch.zeros(18 * n, dtype=torch.long), P, spec,
                                        contract=False, aabb=vaabb)
                rgb_m, den_m = a.reshape(18, n, 3).mean(0), d_.reshape(18, n, 1).mean(0)
                O.grid_refresh_scatter(grid, coords, rgb_m.detach(), den_m.detach(), 1 / 128)
                if train:      # the audio loss reaches the field through these cells (NeRAF_model.py:395-400); unit upstream gradient
                    opt_v.zero_grad(set_to_none=True)
                    (rgb_m.sum() + den_m.sum()).backward()
        return run

    feat_box = {}

    def resnet(train):
        def run():
            with torch.set_grad_enabled(train):
                f = O.resnet3d_forward(grid.unsqueeze(0), sdr, train=True).flatten()
                feat_box["feat"] = f.detach()
                if train:
                    opt_a.zero_grad(set_to_none=True)
                    f.sum().backward()
        return run

    def audio(n, train):
        ab = {k: T(v) for k, v in synth.audio_batch(n, C_, F_, T_, tag="bench.r0").items()}

        def run():
            with torch.set_grad_enabled(train):
                f = feat_box["feat"].clone().requires_grad_(train)
                y = O.audio_get_outputs(ab, f, sdn, aabb, T_)
                l = O.audio_loss_dict(y, ab["data"])
                if train:
                    opt_a.zero_grad(set_to_none=True)
                    (l["audio_sc_loss"] + l["audio_mag_loss"]).backward()
                    opt_a.step()
        return run

    t, spread = {}, {}
    for name, make, n_full in (("vision", vision, R), ("refresh", refresh, R), ("audio", audio, B)):
        if name == "audio":      # the NAcF stage consumes the feature: the ResNet3D comes first, as in the step
            resnet(True)()                                           # warm-up (first-touch of 87 M activation elements + autograd graph)
            both = [once(resnet(True)) for _ in range(2)]
            spread["resnet3d"] = [round(v_, 3) for v_ in both]
            t["resnet3d_train"] = sum(both) / 2
            t["resnet3d_fwd"] = once(resnet(False))
        make(max(n_full // 64, 8), True)()                           # warm-up of this stage's code path
        both = [once(make(n_full, True)) for _ in range(2)]          # the training form twice: two complete steps, mean reported
        spread[name] = [round(v_, 3) for v_ in both]
        t[name + "_train"] = sum(both) / 2
        t[name + "_fwd"] = once(make(n_full, False))
    step_train = t["vision_train"] + t["refresh_train"] + t["resnet3d_train"] + t["audio_train"]
    step_fwd = t["vision_fwd"] + t["refresh_fwd"] + t["resnet3d_fwd"] + t["audio_fwd"]
    bins = B * C_ * F_
    return {"value": (R + bins) / step_train, "unit": "field-samples/s", "cores": ncores, "kind": "port",
            "rays_per_s_train": R / (t["vision_train"] + t["refresh_train"]), "rays_per_s_fwd": R / (t["vision_fwd"] + t["refresh_fwd"]),
            "bins_per_s_train": bins / (t["resnet3d_train"] + t["audio_train"]), "bins_per_s_fwd": bins / (t["resnet3d_fwd"] + t["audio_fwd"]),
            "field_samples_per_s_fwd": (R + bins) / step_fwd,
            "stage_seconds_full_step": {k: round(v_, 3) for k, v_ in t.items()},
            "train_stage_seconds_each_of_the_two_steps": spread,
            "extrapolated": False,
            "sample_short": ("2 full training steps (mean) of the torch-CPU fp32 oracle at the full batch (%d rays + %d slices; radiance, grid refresh, "
                             "ResNet3D, NAcF, backward, Adam), %.1f s each, %d threads; no extrapolation") % (R, B, step_train, ncores),
            "sample": ("TWO full-batch training steps (mean), every stage timed at its full size after a warm-up of the same code path at 1/64 of the "
                       "batch (ResNet3D: run twice, second timed): radiance step at %d rays (sampler, 2 proposal nets, field, composite, "
                       "3 losses, backward, torch Adam lr 1e-2), grid refresh at %d cells x 18 directions (forward + backward with a unit "
                       "upstream gradient), ResNet3D forward + backward on the full 7x128^3 grid, NAcF + STFT loss at %d slices (forward + "
                       "backward + torch Adam lr 1e-4 over NAcF + ResNet3D); forward-only figures from a second no-grad pass; measured "
                       "full step %.1f s train / %.1f s forward; torch-CPU fp32 oracle, %d threads; SURVEY 8(d)'s 3 + 10 repetitions "
                       "would take ~4 minutes") % (R, R, B, step_train, step_fwd, ncores)}


def trajectory_parity(dev, scenario="g9_long", floor_family=None):
    """BASELINE.json's metric, second half ("PSNR & T60 err vs ref"): an in-process training run of the HIP pipeline on the trajectory
    scenario G9 (tests/tools/trajectory_common.py: 1000 iterations of 512 rays + 128 RIR slices on the box-room scene, 64^3 grid, audio
    from iteration 6, the reference's optimizer groups and schedulers) next to the CPU oracle's runs from the same weights on the same
    batches, whose held-out predictions are the committed fixture tests/golden/g9_long.npz (tests/tools/gen_trajectory.py; ~2 h of
    CPU each).  1000 iterations is where the metrics mean something (T60 error ~10 % instead of ~650 % after 100) and far beyond the
    horizon inside which two runs of this chaotic system stay tensor-comparable, so the comparison is metric against metric: held-out
    PSNR and T60 / EDT / C50 errors against GROUND TRUTH, HIP next to the oracle family (the fp32 oracle and its precision / summation-
    order probes) through the frozen rule trajectory_common.GATE_RULE.  tests/test_gpu_trajectory.py asserts the same on the
    deterministic run and on the median of five default-mode runs; this is ONE default-mode run, reported."""
    import numpy as np
    fx = os.path.join(ROOT, "tests", "golden", scenario + ".npz")
    if not os.path.exists(fx):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import trajectory_common as TC
    g = np.load(fx)
    cfg = dict(TC.SCENARIOS[scenario], n_rir_eval=int(g["stft"].shape[0]))
    t0 = time.perf_counter()
    curves, img, stft, pipe, evb = TC.run_hip_trajectory(dev, cfg=cfg, fixed_scale=False)
    train_s = time.perf_counter() - t0
    probes = [str(p_) for p_ in g["probes"]] if "probes" in g else []
    pre = lambda n: "probe_" if n == "params16" else f"probe_{n}_"      # noqa: E731
    stfts = {"hip": stft["eval"], "oracle": g["stft"], **{n: np.asarray(g[pre(n) + "stft"], np.float32) for n in probes}}
    images = {"hip": img, "oracle": g["image"], **{n: g[pre(n) + "image"] for n in probes}}
    m = TC.metric_table(pipe.audio_model, stfts, evb, gt_image=g["gt_image"], images=images)
    keys = {"psnr_db": "psnr_vs_gt_db", "t60_err_pct": "audio_T60", "edt_err_s": "audio_EDT", "c50_err_db": "audio_C50",
            "stft_rel_l2_vs_gt": "stft_rel_l2_vs_gt"}
    out = {"steps": int(g["steps"]), "scenario": scenario, "held_out_rirs": int(g["stft"].shape[0]), "camera_optimizer": bool(cfg.get("camera_opt")),
           "fixture": "tests/golden/" + scenario + ".npz", "hip_training_seconds": round(train_s, 1), "mode": "default (fp32 atomics), one run"}
    for k, mk in keys.items():
        out[k] = m["hip"].get(mk)
        out[k + "_oracle"] = m["oracle"].get(mk)
    family = ["oracle"] + probes
    out["oracle_family"] = {mk: [m[n][mk] for n in family] for mk in TC.GATE_METRICS}
    out["oracle_family_names"] = family
    out["gates"] = TC.gate_table(m, family, floor_family=floor_family)
    out["inside"] = all(v["inside"] for v in out["gates"].values())
    # which metric left its gate and on which side ("better": an error against ground truth LOWER than the gate's better-side bound)
    short = {"psnr_vs_gt_db": "psnr_db", "audio_T60": "t60_err_pct", "audio_EDT": "edt_err_s", "audio_C50": "c50_err_db"}
    out["outside"] = {short[k]: ("better" if (v["hip"] > v["high"]) == TC.GATE_METRICS[k] else "worse") for k, v in out["gates"].items() if not v["inside"]}
    out["rule"] = TC.GATE_RULE
    tail = slice(int(g["steps"]) - 50, int(g["steps"]))
    names = [str(k_) for k_ in g["keys"]][:5]
    out["loss_tails"] = {n: {"hip": float(np.nanmean(curves[tail, j])), "oracle": float(np.nanmean(np.asarray(g["curves"])[tail, j]))}
                         for j, n in enumerate(names)}
    out["note"] = ("every *_err_* is the error against ground truth through the eval branch (BatchNorm on running statistics, "
                   "NeRAF_model.py:648-728) and the evaluator (NeRAF_evaluator.py:131-190, seeded Griffin-Lim), mean over the held-out RIRs; "
                   "*_oracle = the fp32 CPU oracle's; oracle_family = the fp32 oracle and its probes, in oracle_family_names order")
    return out


def _rocprof_reference(which="train_stats"):
    """Average kernel durations from the committed rocprofv3 --kernel-trace --stats summary of this command that PROFILE_REFS names
    (profiles/), so that the HIP-event durations in ``roofline`` can be cross-checked without re-running the profiler."""
    import csv
    f_ = os.path.join(ROOT, "profiles", PROFILE_REFS[which])
    if not os.path.exists(f_):
        return None, {}
    rows = {}
    with open(f_) as f:
        for r in csv.DictReader(f):
            rows[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return PROFILE_REFS[which], rows


def _family_regex(pattern: str):
    """neraf_prof_kernel_name patterns ('a<64, 64, *, 160|128> | b + c') -> regex over rocprof kernel names."""
    alts = []
    for part in re.split(r"\s*[|+]\s*(?![^<]*>)", pattern):
        part = part.strip()
        if not part:
            continue
        rx = re.escape(part).replace(r"\*", r"[^,>]+").replace(r"\|", "|")
        rx = re.sub(r"(\d+)\|(\d+)", r"(?:\1|\2)", rx)
        if rx.endswith(">"):                      # kernels may carry further trailing template arguments (e.g. the parity-class flag)
            rx = rx[:-1] + r"(?:, [^,>]+)*>"
        alts.append(rx)
    return re.compile("(?:" + "|".join(alts) + ")")


def run_eval_mode(a, dev, rank, local, world):
    """`bench.py --mode eval`: BASELINE configs[4] (full eval render) as a throughput line -- see EvalRender for what a step is.
    N > 1: frames and RIRs are independent, every rank renders its own (the pipeline deals eval items round-robin, SURVEY 8e "Eval"):
    no collective in the loop, weak scaling."""
    import torch
    from neraf_amd import _lib
    lib = _lib.load()
    h = _lib.ctx(local)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()
    er = EvalRender(dev, dataset=a.dataset)
    for _ in range(2):                       # plans, allocator pools, the cached ResNet3D feature
        er.step(a.rirs)
    torch.cuda.synchronize()
    gc.collect()
    gc.freeze()
    if a.plain:
        for _ in range(a.warmup):
            er.step(a.rirs)
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            er.step(a.rirs)
        sync()
        el = time.perf_counter() - t0
        if rank == 0:
            print(json.dumps({"metric": "field-samples/sec (rays + RIR STFT bins), eval render", "mode": "eval", "plain": True,
                              "value": (er.rays_per_frame + a.rirs * er.bins_per_rir) * a.steps * world / el, "unit": "field-samples/s",
                              "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
                              "steps_executed_in_process": 2 + a.warmup + a.steps, "rirs_per_step": a.rirs}))
        return
    m, fams = measure_eval(er, a.steps, a.warmup, a.rirs, lib, h, local, sync)
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([m["ms_per_step"]], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        m["ms_per_step"] = float(tt.item())
        m["value"] = (er.rays_per_frame + a.rirs * er.bins_per_rir) * world / (m["ms_per_step"] * 1e-3)
    if rank != 0:
        return
    out = {"metric": "field-samples/sec (rays + RIR STFT bins), eval render", "mode": "eval",
           "value": m["value"], "unit": "field-samples/s", "n_gpus": world, "ranks_seen": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
           "data": "synthetic",
           "rays_per_s": m["rays_per_s"] * world, "fps": m["fps"] * world, "bins_per_s": m["bins_per_s"] * world,
           "fps_audio": m["fps_audio"] * world, "ms_per_frame": m["ms_per_frame"], "us_per_rir": m["us_per_rir"],
           "batched_rirs": m["batched_rirs"],
           "config": {"workload": "BASELINE configs[4] (full eval render) per GPU, no grad: a step = one 684x1024 RAF frame (22 chunks of 32768 rays) + "
                                  "%d RIRs through the audio eval branch, scene feature cached" % a.rirs,
                      "workload_detail": ("BASELINE configs[4] (full eval render) per GPU, no grad: a step = one 684x1024 RAF frame (700,416 rays) through "
                                   "NeRAFVisionModel.get_outputs_for_camera(camera, None, eval=True) -- OPENCV camera -> rays -> %d chunks of 32,768 "
                                   "(sampler, 2 proposal nets, 2 PDF resamplings, fused field query with the mean appearance embedding, "
                                   "composite) -> [H,W,.] images, rgb clipped -- plus %d RIRs through NeRAFAudioModel.get_outputs_for_camera(None, "
                                   "None, batch_audio), one call each (T = %d time queries -> prologue -> NAcF MLP -> [T,%d,%d] + the "
                                   "per-channel panels copied to the host as the reference builds them), ResNet3D scene feature cached (static "
                                   "grid).  rays_per_s / bins_per_s time each half alone (the reference's num_rays_per_sec / "
                                   "num_rays_per_sec_audio, NeRAF_pipeline.py:341-344, :384-387, without its metric code).  Random-init "
                                   "weights, trained-like hash-table magnitudes; cameras with the RAF intrinsics incl. distortion.")
                                  % (m["chunks_per_frame"], a.rirs, T_, C_, F_),
                      "rays_per_frame": m["rays_per_frame"], "rirs_per_step": a.rirs, "bins_per_rir": m["bins_per_rir"],
                      "parallelism": f"dp{world} (eval items dealt round-robin, no collective)"}}
    if fams:
        out["roofline"] = eval_roofline(fams, a.dataset, world)
    if not a.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = eval_cpu_baseline()
    emit(out, a, "eval", world)


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_spawn_ranks(a))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    import numpy as np  # noqa: F401
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    # NERAF_BENCH_SHARE_GPU=1 (test aid): several ranks share the visible GPUs and talk over gloo -- exercises the multi-rank
    # code path (sharding, overlapped gradient reducer, global loss sums) on a 1-GPU box; never a measurement.
    share = os.environ.get("NERAF_BENCH_SHARE_GPU") == "1"
    if share:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # RCCL
        ranks_seen, backend = dist.get_world_size(), dist.get_backend()
    else:
        ranks_seen, backend = 1, None
    from neraf_amd import _lib
    from neraf_amd.parallel import shard_range

    if a.dataset == "soundspaces":
        global C_, F_, T_
        C_, F_, T_ = 2, 257, 101
    if a.scaling == "strong":
        lo, hi = shard_range(a.rays, rank, world)
        R_local = hi - lo
        lo, hi = shard_range(a.slices, rank, world)
        B_local = hi - lo
        R_global, B_global = a.rays, a.slices
    else:
        R_local, B_local = a.rays, a.slices
        R_global, B_global = a.rays * world, a.slices * world
    if a.mode == "eval" and (a.grid, a.n_features) != (128, 1024):
        raise SystemExit("--grid / --n-features select the training step's encoder; the eval-render line is measured on the metric's configuration (128^3, 1024)")
    if a.mode == "eval":
        run_eval_mode(a, dev, rank, local, world)
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    st = JointStep(dev, R_local, B_local, world, dataset=a.dataset, rotate=a.rotate, grid=a.grid, n_features=a.n_features)
    # Setup, before the W warm-up steps: the first steps of a run build the optimizer launch plans (the step with the first
    # proposal-network update builds a second one), capture the ResNet3D hipGraphs and grow the allocator pools --
    # tools/step_trace.py shows them as 10-400 ms steps -- and a full Python garbage collection over the module graph costs
    # 50-80 ms wherever it falls (with W = 5 it fell inside the timed region of some runs and not of others: 6.3 vs 8 ms/step
    # for the same kernels).  PRIME_STEPS untimed steps run first, then the survivors are moved out of the collector's reach.
    for _ in range(PRIME_STEPS + (0 if a.plain else CLOCK_STEPS)):
        st.step()
    torch.cuda.synchronize()
    gc.collect()
    gc.freeze()
    for _ in range(a.warmup):
        st.step()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    def timed_steps(k):
        sync()
        t0 = time.perf_counter()
        for _ in range(k):
            st.step()
        sync()
        el = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    elapsed = timed_steps(a.steps)
    # the same window again, `--repeats` times: 20 steps are 87 ms -- one sample; the spread says how much to trust it
    repeat_ms = [elapsed / a.steps * 1e3] + ([timed_steps(a.steps) / a.steps * 1e3 for _ in range(max(a.repeats - 1, 0))] if not a.plain else [])

    if a.plain:
        if rank == 0:
            print(json.dumps({"metric": "field-samples/sec (rays + RIR STFT bins)", "value": (R_global + B_global * C_ * F_) * a.steps / elapsed,
                              "unit": "field-samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "priming_steps": PRIME_STEPS,
                              "ms_per_step": elapsed / a.steps * 1e3, "scaling": a.scaling, "plain": True, "ranks_seen": ranks_seen,
                              "backend": backend, "dtype": DTYPE, "data": "synthetic", "higher_is_better": True,
                              "rays_per_s": R_global * a.steps / elapsed, "bins_per_s": B_global * C_ * F_ * a.steps / elapsed,
                              "config": {"rays_per_gpu": R_local, "slices_per_gpu": B_local, "global_rays": R_global,
                                         "global_slices": B_global, "parallelism": f"dp{world}", "dataset": a.dataset},
                              "steps_executed_in_process": PRIME_STEPS + a.warmup + a.steps}))
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return

    # ---- the round-3 form of the same step, for the record: ONE resident batch served every step (identical hash-table rows and
    # scatter targets each step) -- two windows alternated with two rotating ones, so that clock drift does not read as a difference
    fixed_ms, rot_ms = [], []
    if a.rotate > 1:
        for _ in range(2):
            st.pin_batch(0)
            st.step()
            fixed_ms.append(timed_steps(a.steps) / a.steps * 1e3)
            st.pin_batch(None)
            st.step()
            rot_ms.append(timed_steps(a.steps) / a.steps * 1e3)

    # ---- the other proposal-update regime (not the headline value): the first 5000 iterations back-propagate through the proposal
    # networks on EVERY step (ProposalNetworkSampler's warm-up schedule); the steady state above does so every 6th step
    steady_i = st.i
    st.i = 2500          # audio branch running (start_step_audio = 2000), proposal warm-up not over (5000)
    for _ in range(3):
        st.step()
    early = timed_steps(min(a.steps, 12))
    early_ms = early / min(a.steps, 12) * 1e3
    st.i = steady_i
    for _ in range(2):
        st.step()

    # ---- the per-rank REPLICATED part of a data-parallel step: ResNet3D forward + backward on the (replicated) grid
    torch.cuda.synchronize()
    net = st.am.resnet3d
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w1024 = torch.ones(a.n_features, device=dev)
    for it in range(6):
        if it == 1:
            ev0.record()
        f = net(st.am.grid.unsqueeze(0)).flatten()
        (f * w1024).sum().backward()
    ev1.record()
    torch.cuda.synchronize()
    resnet_ms = ev0.elapsed_time(ev1) / 5
    # SURVEY 8(d) figures for THIS encoder configuration (the module constants at the default 128^3 / 1024)
    _fl = _lib.load().neraf_resnet3d_forward_flops
    _fl.restype = C.c_double
    resnet_fwd_gflop = RESNET_FWD_GFLOP if (a.grid, a.n_features) == (128, 1024) else _fl(C.byref(net.backbone_net._desc)) / 1e9
    nacf_flop_slice = NACF_DENSE_FLOP_PER_SLICE_FWD + 2 * (a.n_features - 1024) * 5096

    # ---- instrumented replay (not timed): per-kernel-family HIP-event durations, as recorded (no overhead subtraction: an event
    # pair around a few-microsecond kernel reads ~3 us more than rocprofv3's kernel duration, so small kernels are UNDER-stated)
    lib = _lib.load()
    h = _lib.ctx(local)
    lib.neraf_prof_enable(h, 1)
    nprof = min(a.steps, 12)
    for _ in range(nprof):
        st.step()
    torch.cuda.synchronize()
    ref_name, ref_rows = _rocprof_reference()
    fams = _prof_families(lib, h, local, nprof, ref_rows)
    lib.neraf_prof_enable(h, 0)
    sync()

    eval_line = None
    if not a.no_eval_line and world == 1 and a.dataset == "raf":
        # configs[4] in short (the full line: `bench.py --mode eval`): 3 frames + 3 x 16 RIRs after one warm-up step
        try:
            er = EvalRender(dev, dataset=a.dataset)
            eval_line, efams = measure_eval(er, 3, 1, 16, lib, h, local, sync, full=True)
            eval_line["note"] = ("BASELINE configs[4] shape on one GPU, short form (3 steps of one 684x1024 frame in 22 chunks + 16 RIRs through "
                                 "the eval branch with the cached scene feature) with the eval render's own roofline (instrumented replay of "
                                 "3 more steps) and cpu_baseline; `bench.py --mode eval` is the same measurement over more steps")
            if rank == 0:
                eval_line["roofline"] = eval_roofline(efams or [], a.dataset, world)
                if not a.no_cpu_baseline:
                    eval_line["cpu_baseline"] = eval_cpu_baseline()
            del er
        except Exception as e:                          # a side measurement must not take the training line down
            eval_line = {"error": repr(e)}

    if rank == 0:
        bins = B_global * C_ * F_
        samples = (R_global + bins) * a.steps
        # HBM bytes per launch from the committed PMC passes of this same command (rocprofv3 cannot wrap itself from inside):
        # tools/gpu_pmc.sh -> profiles/*_pmc_traffic.json (FETCH_SIZE doubled as the gfx950 guide prescribes, + WRITE_SIZE)
        pmc_file = os.path.join(ROOT, "profiles", PROFILE_REFS["train_pmc"])
        default_shape = a.rays == 4096 and a.slices == 2048 and a.dataset == "raf" and world == 1
        pmc = json.load(open(pmc_file))["families"] if os.path.exists(pmc_file) and default_shape else {}
        for k in fams:
            t = pmc.get(k["kernel"])
            k["traffic"] = t["hbm_bytes_per_launch"] if t else None
        # the headline family is priced against HBM or the matrix pipe (the contract's two bounds); the L2-priced proposal families stay in
        # the family list with their own fractions
        head = [k for k in fams if k["bound"] in ("hbm", "mfma")] or fams
        dom = max(head, key=lambda k: k["ms_per_step"]) if head else None
        # `value`: the MEDIAN of the `--repeats` back-to-back windows of exactly --steps steps each (every window bracketed by barrier +
        # device sync, max over ranks); window 0 alone was the headline until round 4 -- 20 steps are 84 ms, one sample
        ms_step = _median(repeat_ms)
        elapsed = ms_step * 1e-3 * a.steps
        out = {
            "metric": "field-samples/sec (rays + RIR STFT bins)",
            "value": samples / elapsed,
            "unit": "field-samples/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen, "backend": backend,
            "steps": a.steps,
            "warmup": a.warmup,
            "priming_steps": PRIME_STEPS + CLOCK_STEPS,
            "ms_per_step": ms_step,
            "repeat_windows": {"n": len(repeat_ms), "steps_each": a.steps, "ms_per_step": [round(v, 4) for v in repeat_ms],
                               "median": _median(repeat_ms), "min": min(repeat_ms), "max": max(repeat_ms),
                               "note": "value / ms_per_step are the MEDIAN window; every window is exactly --steps steps, back to back"},
            "ms_per_step_median": _median(repeat_ms), "ms_per_step_min": min(repeat_ms), "ms_per_step_max": max(repeat_ms),
            "batches": {"distinct_resident_batches": a.rotate, "rotating_ms_per_step": rot_ms, "fixed_batch_ms_per_step": fixed_ms,
                        "fixed_minus_rotating_ms": (sum(fixed_ms) / len(fixed_ms) - sum(rot_ms) / len(rot_ms)) if fixed_ms else None,
                        "note": "the timed steps cycle through `distinct_resident_batches` different ray bundles / slice batches (the "
                                "reference draws a fresh batch per step, NeRAF_pipeline.py:175, :187); fixed_batch = the round-3 form, "
                                "batch 0 every step, in windows alternated with rotating ones"},
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": DTYPE,
            "data": "synthetic",
            "rays_per_s": R_global * a.steps / elapsed,
            "bins_per_s": bins * a.steps / elapsed,
            "steps_per_s": a.steps / elapsed,
            "regimes": {"steady_state_ms_per_step": ms_step,
                        "iterations_2000_to_5000_ms_per_step": early_ms,
                        "note": "value is the steady state of the 400k-iteration schedule (step 20000: proposal networks back-propagated every "
                                "6th step, nerfacto's schedule); before iteration 5000 they are back-propagated (almost) every step; before iteration 2000 the audio branch is off"},
            "replicated_per_rank": {"resnet3d_fwd_bwd_ms": resnet_ms,
                                    "note": "data parallel: the ResNet3D forward + backward runs on every rank (the grid and its weights are "
                                            "replicated); rays, RIR slices and the grid-refresh cells are sharded"},
            "config": {
                "workload": (("RAF FurnishedRoom joint training step (BASELINE configs[2]): %d rays + %d RIR slices x 513 bins per GPU" if a.dataset == "raf"
                              else "SoundSpaces joint training step (BASELINE configs[3] head): %d rays + %d RIR slices x 2 x 257 bins per GPU")
                             % (R_local, B_local)) + "; radiance + grid refresh + ResNet3D + NAcF + STFT loss, backward, GradScaler + fused Adam; batches resident"
                            + ("" if (a.grid, a.n_features) == (128, 1024) else f"; NON-DEFAULT encoder: {a.grid}^3 grid, {a.n_features} features"),
                "workload_detail": (("RAF FurnishedRoom joint step (BASELINE configs[2] shape: %d rays + %d RIR slices x 513 bins per GPU): " if a.dataset == "raf"
                              else "SoundSpaces joint step (BASELINE configs[3] head shape: %d rays + %d RIR slices x 2 x 257 bins per GPU): ") +
                             "radiance forward (camera-pose deltas, sampler, 2 proposal nets, 2 PDF resamplings, fused field query, composite) + rgb/"
                             "interlevel/distortion losses -> grid refresh (%d cells x 18 dirs%s) -> ResNet3D forward on the 7x128^3 "
                             "grid -> audio prologue + NAcF MLP -> STFT loss -> one backward (NAcF -> ResNet3D -> refreshed grid cells -> "
                             "field; radiance losses -> proposal nets + fused field backward + weight-grad GEMMs) -> %sGradScaler + "
                             "fused Adam: proposal_networks + fields (lr 1e-2 -> 1e-4 @200k), then audio_fields = NAcF + ResNet3D + fields again "
                             "(lr 1e-4 -> 1e-8, 2000 warm-up), as NeRAF_pipeline.py:487 / NeRAF_config.py:115-132 group and schedule them -> "
                             "scheduler steps; the camera optimizer (SO3xR3, NeRAF_config.py:97) is on: pose deltas applied to the ray bundle, "
                             "their photometric + regulariser gradients computed, camera_opt Adam group stepped (lr 1e-3 -> 1e-4 @5k).  "
                             "The steps cycle through %d distinct resident batches.  Not inside: data loading (batches are resident).")
                             % (R_local, B_local, 4096, ", sharded over the ranks" if world > 1 else "", "RCCL all-reduce -> " if world > 1 else "",
                                a.rotate),
                "rays_per_gpu": R_local, "slices_per_gpu": B_local, "global_rays": R_global, "global_slices": B_global,
                "parallelism": f"dp{world}", "distinct_resident_batches": a.rotate, "encoder_grid": a.grid, "encoder_features": a.n_features,
                "repeat_windows_ms_per_step": {"n": len(repeat_ms), "median": _median(repeat_ms), "min": min(repeat_ms), "max": max(repeat_ms)},
            },
        }
        if dom:
            out["roofline"] = {"bound": dom["bound"], "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": dom["peak"],
                               "unit": dom["unit"], "frac": dom["frac"], "traffic": dom["traffic"],
                               "traffic_source": PROFILE_REFS["train_pmc"] if pmc and dom["traffic"] is not None else None,
                               "avg_launch_us": dom["avg_us"], "launches_per_step": dom["launches_per_step"],
                               "algorithmic_work_per_launch": dom["work_per_launch"],
                               "selection": "family with the largest share of the step (ms_per_step) among the instrumented families",
                               "durations": "HIP events on the launch stream, as recorded (no overhead subtraction)",
                               "rocprof_reference": ref_name,
                               "dense_equiv_gflop_per_step": {"nacf_fwd_bwd": 3 * nacf_flop_slice * B_local / 1e9,
                                                              "resnet3d_fwd_bwd": 3 * resnet_fwd_gflop},
                               "instrumented_gflop_per_step": {
                                   "algorithmic": sum(k["work_per_step"] for k in fams if k["bound"] == "mfma") / 1e9,
                                   "executed": sum(k["executed_per_launch"] * k["launches_per_step"] for k in fams if k["bound"] == "mfma") / 1e9,
                                   "note": "sum over the MFMA-priced families; algorithmic = ResNet3D 3 x 94.72 - 29.36 (the stem's input "
                                           "gradient is a per-cell kernel, not a GEMM) + the NAcF GEMMs as executed (layer-0 split: the "
                                           "1024 shared inputs are a GEMV, so less than the dense-equivalent 3 x 40.84 MFLOP/slice) + the "
                                           "radiance field's weight-gradient GEMMs"},
                               "whole_step_mfma_frac": ((3 * nacf_flop_slice * B_local / 1e9 + 3 * resnet_fwd_gflop) / 1e3)
                                                       / (ms_step * 1e-3) / MFMA_PEAK_TFLOPS,
                               "all_kernel_families": fams}
        g_cap, g_launch = C.c_int(), C.c_int()
        out["hip_graphs"] = {"enabled": bool(lib.neraf_graph_stats(h, C.byref(g_cap), C.byref(g_launch))), "captures": g_cap.value,
                             "launches": g_launch.value}
        if eval_line is not None:
            out["eval_render"] = eval_line
        parity = "off" if a.no_parity else a.parity
        if parity != "off" and world == 1:
            try:
                out["parity"] = trajectory_parity(dev)
                if parity == "all":     # the same comparison in the reference's configuration (camera optimizer on); detail file only
                    p10 = trajectory_parity(dev, "g10_long_pose", floor_family=(out["parity"] or {}).get("oracle_family"))
                    if p10 is not None:
                        out["parity_camera_optimizer_on"] = p10
            except Exception as e:                      # a measurement aid must not take the bench line down
                out.setdefault("parity", {"error": repr(e)})
        if not a.no_cpu_baseline and world == 1:      # the host baseline is reported by the single-GPU run only
            out["cpu_baseline"] = cpu_baseline(a.rays, a.slices)
        emit(out, a, "train", world)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""NeRAF hot-path benchmark on MI355X (contract: see the task brief / DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A *step* is one pass of NeRAFPipeline.get_train_loss_dict (NeRAF_pipeline.py:166-222) over one synthetic
batch already resident in HBM, at the RAF FurnishedRoom training shape (4096 rays + 2048 RIR STFT slices of
513 bins, NeRAF_config.py:57,87); ``config.workload`` states exactly which stages are inside the timed region.
One process per GPU; for N > 1 every rank owns its own shard of rays and slices (weak scaling) and gradients
and loss sums are all-reduced over RCCL.

Rank 0 prints ONE JSON line: whole-job field-samples/s, plus ``roofline`` for the dominant kernel family
(HIP-event durations recorded inside the library over an instrumented replay of the same steps) and
``cpu_baseline`` (the CPU oracle on this host, bounded sample of the same workload).
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
PRIME_STEPS = 8             # untimed set-up steps before the --warmup steps (see main)
HBM_PEAK_GBS = 8000.0       # HBM3E spec, same guide
NACF_DENSE_FLOP_PER_SLICE_FWD = 40_836_464  # SURVEY.md 8(d), RAF head (C*F = 513)
RESNET_FWD_GFLOP = 94.72
C_, F_, T_ = 1, 513, 60


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=4096, help="rays per GPU per step (NeRAF_config.py:87)")
    ap.add_argument("--slices", type=int, default=2048, help="RIR STFT slices per GPU per step (NeRAF_config.py:57)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dataset", choices=("raf", "soundspaces"), default="raf",
                    help="audio head shape: raf = 1 x 513 bins, T = 60 (BASELINE configs[1..2], the default and the metric's config); "
                         "soundspaces = 2 x 257 bins, T = 101 (configs[3]: per GPU 4096 rays + 808 slices)")
    return ap.parse_args()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class JointStep:
    """cfg3 (RAF FurnishedRoom joint) training step, NeRAFPipeline.get_train_loss_dict order (NeRAF_pipeline.py:175-199):
       1. NeRAFVisionModel.get_outputs on the ray batch (sampler, 2 proposal nets, 2 PDF resamplings, fused field
          query, composite) + get_loss_dict (rgb MSE, interlevel, distortion)
       2. audio_model.query_grid_one_batch: 4096 cells x 18 directions through the field, mean, slab write
       3. ResNet3D(7x128^3 grid) -> 1024 feature (train-mode BatchNorm)
       4. audio get_outputs (GPU prologue + NAcF MLP) -> STFT loss
       5. ONE backward over the summed loss dict: NAcF (all grads + d/d feature) -> ResNet3D backward (dgrad/wgrad GEMMs,
          BatchNorm backward) -> grid-window gradient -> refresh backward into the field; radiance half (loss grads,
          proposal backward, fused field backward, weight-grad GEMMs) -> [RCCL all-reduce] -> GradScaler + fused Adam on
          the radiance parameters (lr 1e-2) and the audio parameters (NAcF + ResNet3D, lr 1e-4)."""

    def __init__(self, dev, R, B, world, dataset="raf"):
        from neraf_amd import synth
        from neraf_amd.model import NeRAFAudioModel, NeRAFAudioModelConfig
        from neraf_amd.vision import NeRAFVisionModel, RayBundle
        self.dev, self.R, self.B, self.world = dev, R, B, world
        rank = int(os.environ.get("RANK", "0"))
        self.vm = NeRAFVisionModel(torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]]), 210).to(dev)
        with torch.no_grad():      # trained-like table magnitudes (synthetic, same on every rank)
            g = torch.Generator(device="cpu").manual_seed(0)
            for p in [self.vm.field.module.table] + [pn.table for pn in self.vm.proposal_networks]:
                p.copy_((torch.rand(p.shape, generator=g) - 0.5).to(dev))
        cfg = (NeRAFAudioModelConfig(dataset="RAF", grid_step=1 / 128) if dataset == "raf" else
               NeRAFAudioModelConfig(dataset="SoundSpaces", grid_step=1 / 128, max_len=T_, N_freq_stft=F_))
        self.am = NeRAFAudioModel(cfg, T(synth.audio_aabb()), process_group=True if world > 1 else None)
        self.am.field.load_state_dict({k: T(v) for k, v in synth.nacf_state_dict(1187, 512, C_, F_).items()})
        self.am.resnet3d.backbone_net.load_state_dict({k: T(v) for k, v in synth.resnet3d_state_dict(7).items()})
        self.am.to(dev)
        self.vm.train(); self.am.train()
        rb = synth.ray_batch(R, tag=f"bench.rays.r{rank}")
        self.bundle = RayBundle(T(rb["origins"]).to(dev), T(rb["directions"]).to(dev), T(rb["camera_indices"]).to(dev))
        self.batch = {k: T(v).to(dev) for k, v in synth.audio_batch(B, C_, F_, T_, tag=f"bench.r{rank}").items()}
        self.gt = {"image": T(rb["rgb"]).to(dev)}
        # the reference's pipeline object: get_train_loss_dict (NeRAF_pipeline.py:166-222) inside Trainer.train_iteration, with its
        # parameter groups / optimizers (NeRAF_config.py:115-127; the field parameters are in "fields" AND "audio_fields", :487)
        from neraf_amd.pipeline import FixedBatchDataManager, NeRAFPipeline
        self.pipe = NeRAFPipeline(self.vm, self.am, datamanager=FixedBatchDataManager(self.bundle, self.gt, R),
                                  audio_datamanager=FixedBatchDataManager(None, self.batch), start_step_audio=2000, world_size=world)
        self.optimizers, self.scaler = self.pipe.make_optimizers(init_scale=65536.0)
        if world > 1:
            # gradient averaging overlapped with the backward pass; every group is all-reduced, the ResNet3D's too (its forward
            # accumulates BatchNorm statistics with fp32 atomics, so per-rank gradients differ in the last bits and the chaotic
            # encoder amplifies that: only an all-reduce keeps the replicas identical)
            self.pipe.attach_gradient_reducer()
        self.i = 20000      # steady-state regime of the 400k-iteration schedule: anneal done, proposal nets updated every 6th step

    def samples_per_step(self):
        return self.R + self.B * C_ * F_

    def step(self):
        self.i += 1
        loss, _ = self.pipe.train_iteration(self.i, self.optimizers, self.scaler)
        return loss


def cpu_baseline(R, B):
    """The CPU oracle (torch fp32; audio half pinned to the reference on G1-G5, radiance half unpinned) timed on this
    host for the SAME stages.  A full step costs ~1 min of CPU, so the ray/slice-proportional stages are timed on a
    1/8 sample and scaled by 8 while the per-step-constant ResNet3D forward is timed in full; the figure reported is
    samples_per_step / extrapolated step time."""
    from neraf_amd import synth
    from oracle import audio as O
    from oracle import vision as V
    ncores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(ncores)
    frac = 8
    r, b = R // frac, B // frac
    spec = V.NerfactoSpec()
    tot = (spec.prop_grids[0].total, spec.prop_grids[1].total, spec.main_grid.total)
    P = {k: T(v) for k, v in synth.vision_params(tot, num_train_data=210, table_scale=0.5).items()}
    rb = synth.ray_batch(r, tag="bench.rays.r0")
    sdn = {k: T(v).requires_grad_(True) for k, v in synth.nacf_state_dict(1187, 512, C_, F_).items()}
    sdr = {k: T(v) for k, v in synth.resnet3d_state_dict(7).items()}
    ab = {k: T(v) for k, v in synth.audio_batch(b, C_, F_, T_, tag="bench.r0").items()}
    aabb = T(synth.audio_aabb())
    grid = O.reset_grid(1 / 128)
    coords = O.coordinates_to_render(1 / 128)[:r]
    dirs = O.fixed_viewing_directions()
    vaabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    opt = torch.optim.Adam(list(sdn.values()), lr=1e-4, eps=1e-15)
    t = {}
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    t0 = time.perf_counter()
    ov = V.nerfacto_forward(T(rb["origins"]), T(rb["directions"]), T(rb["camera_indices"]), Pg, spec, training=True,
                            jitters=[T(j) for j in rb["jitters"]])
    lv = V.vision_loss_dict(ov, T(rb["rgb"]), spec)
    (lv["rgb_loss"] + lv["interlevel_loss"] + lv["distortion_loss"]).backward()
    t["vision_train"] = (time.perf_counter() - t0) * frac
    del Pg, ov, lv
    with torch.no_grad():
        t0 = time.perf_counter()
        ori = O.refresh_world_positions(coords, vaabb)
        rg, dn = [], []
        for j in range(18):
            a, d_ = V.field_forward(ori, dirs[j].expand(r, -1), torch.zeros(r, dtype=torch.long), P, spec, contract=False, aabb=vaabb)
            rg.append(a); dn.append(d_[:, None])
        grid = O.grid_refresh_scatter(grid, coords, torch.stack(rg).mean(0), torch.stack(dn).mean(0), 1 / 128)
        t["refresh"] = (time.perf_counter() - t0) * frac
        O.resnet3d_forward(grid.unsqueeze(0), sdr, train=True)   # warm-up
        t0 = time.perf_counter()
        feat = O.resnet3d_forward(grid.unsqueeze(0), sdr, train=True).flatten()
        t["resnet3d_fwd"] = time.perf_counter() - t0

    def audio():
        opt.zero_grad(set_to_none=True)
        f = feat.clone().requires_grad_(True)
        y = O.audio_get_outputs(ab, f, sdn, aabb, T_)
        l = O.audio_loss_dict(y, ab["data"])
        (l["audio_sc_loss"] + l["audio_mag_loss"]).backward()
        opt.step()
    audio()
    t0 = time.perf_counter()
    audio()
    t["audio_train"] = (time.perf_counter() - t0) * frac
    step_s = sum(t.values())
    return {"value": (R + B * C_ * F_) / step_s, "unit": "field-samples/s", "cores": ncores, "kind": "port",
            "sample": ("1/8 sample (%d rays, %d refresh cells, %d slices) scaled x8 + one full ResNet3D forward; extrapolated "
                       "step %.1f s = %s; torch-CPU fp32 oracle") % (r, r, b, step_s, {k: round(v, 2) for k, v in t.items()})}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
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
    from neraf_amd import _lib

    if a.dataset == "soundspaces":
        global C_, F_, T_
        C_, F_, T_ = 2, 257, 101
    st = JointStep(dev, a.rays, a.slices, world, dataset=a.dataset)
    # Setup, before the W warm-up steps: the first steps of a run build the optimizer launch plans (the step with the first
    # proposal-network update builds a second one), capture the ResNet3D hipGraphs and grow the allocator pools --
    # tools/step_trace.py shows them as 10-400 ms steps -- and a full Python garbage collection over the module graph costs
    # 50-80 ms wherever it falls (with W = 5 it fell inside the timed region of some runs and not of others: 6.3 vs 8 ms/step
    # for the same kernels).  PRIME_STEPS untimed steps run first, then the survivors are moved out of the collector's reach.
    for _ in range(PRIME_STEPS):
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

    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st.step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- instrumented replay (not timed): per-kernel-family HIP-event durations
    lib = _lib.load()
    h = _lib.ctx(local)
    # what an event pair measures around nothing: subtracted per launch below (a 10 us kernel would read 13 us otherwise)
    ov = C.c_double()
    word = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.neraf_prof_event_overhead(h, word.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(ov)), local)
    lib.neraf_prof_enable(h, 1)
    nprof = min(a.steps, 10)
    for _ in range(nprof):
        st.step()
    torch.cuda.synchronize()
    fams = []
    kid = 0
    while lib.neraf_prof_kernel_name(kid):
        ms, n, w = C.c_double(), C.c_int(), C.c_double()
        _lib.check(lib.neraf_prof_summary(h, kid, C.byref(ms), C.byref(n), C.byref(w)), local)
        if n.value:
            raw_ms = ms.value
            ms.value = max(raw_ms - n.value * ov.value, 0.25 * raw_ms)
            is_bytes = kid in (2, 3, 5, 6, 7)     # gather / scatter kernels are priced in bytes against HBM (ids: csrc/common.h PROF_*)
            rate = w.value / (ms.value * 1e-3) / (1e9 if is_bytes else 1e12) if ms.value > 0 else 0.0
            fams.append({"kernel": lib.neraf_prof_kernel_name(kid).decode(), "bound": "hbm" if is_bytes else "mfma",
                         "launches_per_step": n.value / nprof, "avg_us": ms.value * 1e3 / n.value,
                         "ms_per_step": ms.value / nprof, "achieved": rate, "unit": "GB/s" if is_bytes else "TFLOP/s",
                         "work_per_launch": w.value / n.value, "avg_us_with_event_overhead": raw_ms * 1e3 / n.value})
        kid += 1
    lib.neraf_prof_enable(h, 0)
    sync()

    if rank == 0:
        samples = st.samples_per_step() * world * a.steps
        # HBM bytes per launch from the committed PMC passes of this same command (rocprofv3 cannot wrap itself from inside):
        # tools/gpu_pmc.sh -> profiles/*_pmc_traffic.json (FETCH_SIZE doubled as the gfx950 guide prescribes, + WRITE_SIZE)
        import glob
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
        pmc = json.load(open(pmc_files[-1]))["families"] if pmc_files and a.rays == 4096 and a.slices == 2048 and a.dataset == "raf" else {}
        for k in fams:
            t = pmc.get(k["kernel"])
            k["traffic"] = t["hbm_bytes_per_launch"] if t else None
        dom = max(fams, key=lambda k: k["ms_per_step"]) if fams else None
        out = {
            "metric": "field-samples/sec (rays + RIR STFT bins)",
            "value": samples / elapsed,
            "unit": "field-samples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "priming_steps": PRIME_STEPS,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {
                "workload": (("RAF FurnishedRoom joint step (BASELINE configs[2] shape: %d rays + %d RIR slices x 513 bins per GPU): " if a.dataset == "raf"
                              else "SoundSpaces joint step (BASELINE configs[3] head shape: %d rays + %d RIR slices x 2 x 257 bins per GPU): ") +
                             "radiance forward (sampler, 2 proposal nets, 2 PDF resamplings, fused field query, composite) + rgb/"
                             "interlevel/distortion losses -> grid refresh (%d cells x 18 dirs) -> ResNet3D forward on the 7x128^3 "
                             "grid -> audio prologue + NAcF MLP -> STFT loss -> one backward (NAcF -> ResNet3D -> refreshed grid cells -> "
                             "field; radiance losses -> proposal nets + fused field backward + weight-grad GEMMs) -> %sGradScaler + "
                             "fused Adam: proposal_networks + fields (lr 1e-2), then audio_fields = NAcF + ResNet3D + fields again (lr 1e-4), as NeRAF_pipeline.py:487 groups them.  Not modelled: camera-pose optimizer "
                             "(nerfstudio CameraOptimizer), data loading.")
                             % (a.rays, a.slices, a.rays, "RCCL all-reduce -> " if world > 1 else ""),
                "rays_per_gpu": a.rays, "slices_per_gpu": a.slices, "parallelism": f"dp{world}",
            },
        }
        if dom:
            peak = HBM_PEAK_GBS if dom["bound"] == "hbm" else MFMA_PEAK_TFLOPS
            out["roofline"] = {"bound": dom["bound"], "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": peak,
                               "unit": dom["unit"], "frac": dom["achieved"] / peak, "traffic": dom["traffic"],
                               "traffic_source": os.path.basename(pmc_files[-1]) if pmc and dom["traffic"] is not None else None,
                               "avg_launch_us": dom["avg_us"], "launches_per_step": dom["launches_per_step"],
                               "algorithmic_work_per_launch": dom["work_per_launch"],
                               "dense_equiv_gflop_per_step": {"nacf_fwd_bwd": 3 * NACF_DENSE_FLOP_PER_SLICE_FWD * a.slices / 1e9,
                                                              "resnet3d_fwd": RESNET_FWD_GFLOP},
                               "event_pair_overhead_us": ov.value * 1e3, "all_kernel_families": fams}
        g_cap, g_launch = C.c_int(), C.c_int()
        out["hip_graphs"] = {"enabled": bool(lib.neraf_graph_stats(h, C.byref(g_cap), C.byref(g_launch))), "captures": g_cap.value,
                             "launches": g_launch.value}
        if not a.no_cpu_baseline and world == 1:      # the host baseline is reported by the single-GPU run only
            out["cpu_baseline"] = cpu_baseline(a.rays, a.slices)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

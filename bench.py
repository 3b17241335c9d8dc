#!/usr/bin/env python3
"""NeRAF hot-path benchmark on MI355X (contract: see the task brief / DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A *step* is one training pass of the hot path over one synthetic batch (already resident in HBM):
``config.workload`` names exactly which stages are inside the timed region.  One process per GPU;
for N > 1 each rank owns its own shard of RIR slices (weak scaling: per-GPU batch fixed) and the
gradients are all-reduced over RCCL before the optimizer step.

Rank 0 prints ONE JSON line: whole-job field-samples/s (+ ``roofline`` for the dominant kernel,
measured with HIP events inside the library over an instrumented replay of the same steps, and
``cpu_baseline`` = the CPU oracle timed on this host on a bounded sample of the same workload).
"""
import argparse
import ctypes as C
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
NACF_DENSE_FLOP_PER_SLICE_FWD = 40_836_464  # SURVEY.md 8(d), RAF head (C*F = 513)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--slices", type=int, default=2048, help="RIR STFT slices per GPU per step (NeRAF_config.py:57)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    return ap.parse_args()


class AudioBranchStep:
    """RAF FurnishedRoom audio branch of NeRAFPipeline.get_train_loss_dict (NeRAF_pipeline.py:186-199):
    next audio batch (synthetic, resident) -> NeRAFAudioModel.get_outputs (query prologue + NAcF MLP,
    NeRAF_model.py:531-566) -> STFTLoss + scaling (:584-600) -> backward (all NAcF parameter grads +
    d/d(grid feature)) -> [RCCL grad all-reduce] -> GradScaler + Adam step (NeRAF_config.py:124-127)."""

    C_, F_, T_ = 1, 513, 60

    def __init__(self, dev, B, world):
        from neraf_amd import synth
        from neraf_amd.field import NeRAFAudioSoundField
        from neraf_amd.losses import STFTLoss
        self.dev, self.B, self.world = dev, B, world
        rank = int(os.environ.get("RANK", "0"))
        sd = {k: torch.from_numpy(v) for k, v in synth.nacf_state_dict(1187, 512, self.C_, self.F_).items()}
        self.field = NeRAFAudioSoundField(1187, 512, sound_rez=self.C_, N_frequencies=self.F_)
        self.field.load_state_dict(sd)
        self.field.to(dev)
        b = synth.audio_batch(B, self.C_, self.F_, self.T_, tag=f"bench.r{rank}")
        self.batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in b.items()}
        self.aabb = torch.from_numpy(synth.audio_aabb())
        # stand-in for the ResNet3D scene feature until that stage is inside the step (requires grad: dfeat is computed)
        self.feat = torch.from_numpy(synth.uniform("bench.feat", (1024,), 0.0, 2.0)).to(dev).requires_grad_(True)
        self.crit = STFTLoss("mse", process_group=True if world > 1 else None)
        self.params = list(self.field.parameters())
        try:
            self.opt = torch.optim.Adam(self.params, lr=1e-4, eps=1e-15, fused=True)
        except Exception:
            self.opt = torch.optim.Adam(self.params, lr=1e-4, eps=1e-15, foreach=True)
        self.scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
        self.flat = None

    def samples_per_step(self):
        return self.B * self.C_ * self.F_

    def step(self):
        bt = self.batch
        self.opt.zero_grad(set_to_none=True)
        self.feat.grad = None
        y = self.field.forward_queries(self.feat, bt["time_query"], bt["mic_pose"], bt["source_pose"], bt["rot"],
                                       self.aabb, self.T_)
        d = self.crit(y, bt["data"])
        loss = d["audio_sc_loss"] * 1e-1 * 1e-3 + d["audio_mag_loss"] * 1.0 * 1e-3   # NeRAF_model.py:597-598
        self.scaler.scale(loss).backward()
        if self.world > 1:
            from neraf_amd.parallel import allreduce_gradients
            allreduce_gradients(self.params, self.world)
        self.scaler.step(self.opt)
        self.scaler.update()
        return loss


def cpu_baseline(B, seconds):
    """The oracle (torch-CPU restatement, pinned to the reference on G1-G5) on the same workload:
    prologue + NAcF fwd + STFT loss + backward + Adam, all host cores."""
    from neraf_amd import synth
    from oracle import audio as O
    ncores = os.cpu_count() or 1
    torch.set_num_threads(ncores)
    C_, F_, T_ = AudioBranchStep.C_, AudioBranchStep.F_, AudioBranchStep.T_
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.nacf_state_dict(1187, 512, C_, F_).items()}
    b = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.audio_batch(B, C_, F_, T_, tag="bench.r0").items()}
    aabb = torch.from_numpy(synth.audio_aabb())
    feat = torch.from_numpy(synth.uniform("bench.feat", (1024,), 0.0, 2.0)).requires_grad_(True)
    opt = torch.optim.Adam(list(sd.values()), lr=1e-4, eps=1e-15)

    def one():
        opt.zero_grad(set_to_none=True)
        feat.grad = None
        y = O.audio_get_outputs(b, feat, sd, aabb, T_)
        l = O.audio_loss_dict(y, b["data"])
        (l["audio_sc_loss"] + l["audio_mag_loss"]).backward()
        opt.step()

    one()  # warm-up
    t0 = time.perf_counter()
    one()
    t1 = time.perf_counter() - t0
    n = max(2, min(12, int(seconds / max(t1, 1e-3))))
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    return {"value": B * C_ * F_ / med, "unit": "field-samples/s", "cores": ncores, "kind": "port",
            "sample": f"{n} steps of the same {B}-slice audio-branch step (median {med*1e3:.0f} ms/step), torch-CPU fp32 oracle"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL
    from neraf_amd import _lib

    st = AudioBranchStep(dev, a.slices, world)
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

    # ---- instrumented replay (not timed): per-kernel HIP-event durations of the dominant kernel
    lib = _lib.load()
    h = _lib.ctx(local)
    lib.neraf_prof_enable(h, 1)
    nprof = min(a.steps, 10)
    for _ in range(nprof):
        st.step()
    torch.cuda.synchronize()
    kernels = []
    kid = 0
    while lib.neraf_prof_kernel_name(kid):
        ms, n, w = C.c_double(), C.c_int(), C.c_double()
        _lib.check(lib.neraf_prof_summary(h, kid, C.byref(ms), C.byref(n), C.byref(w)), local)
        if n.value:
            kernels.append({"kernel": lib.neraf_prof_kernel_name(kid).decode(), "launches_per_step": n.value / nprof,
                            "avg_us": ms.value * 1e3 / n.value, "ms_per_step": ms.value / nprof,
                            "tflops": w.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0,
                            "gflop_per_launch": w.value / n.value / 1e9})
        kid += 1
    lib.neraf_prof_enable(h, 0)
    sync()

    if rank == 0:
        samples = st.samples_per_step() * world * a.steps
        dom = max(kernels, key=lambda k: k["ms_per_step"]) if kernels else None
        out = {
            "metric": "field-samples/sec (rays + RIR STFT bins)",
            "value": samples / elapsed,
            "unit": "field-samples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {
                "workload": ("RAF FurnishedRoom audio-branch training step (cfg3 audio half): %d RIR STFT slices x 513 bins "
                             "per GPU; GPU query prologue -> NAcF MLP (layer-0 split) -> STFT loss -> backward (all NAcF grads "
                             "+ d/d grid-feature) -> %sGradScaler+Adam; rays=0 and ResNet3D/grid refresh NOT yet inside the step"
                             % (a.slices, "RCCL grad all-reduce -> " if world > 1 else "")),
                "slices_per_gpu": a.slices, "rays_per_gpu": 0, "parallelism": f"dp{world}",
            },
        }
        if dom:
            out["roofline"] = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"], "peak": MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": dom["tflops"] / MFMA_PEAK_TFLOPS, "traffic": None,
                               "avg_launch_us": dom["avg_us"], "launches_per_step": dom["launches_per_step"],
                               "gflop_per_launch_executed": dom["gflop_per_launch"],
                               "dense_equiv_gflop_per_step": 3 * NACF_DENSE_FLOP_PER_SLICE_FWD * a.slices / 1e9,
                               "all_kernels": kernels}
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.slices, a.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

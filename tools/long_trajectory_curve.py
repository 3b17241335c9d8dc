#!/usr/bin/env python3
"""Loss curve + GradScaler scale of the G7 trajectory scenario over a long run, printed every --every iterations.
    python tools/long_trajectory_curve.py [steps] [--no-growth] [--every N]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
ap = argparse.ArgumentParser()
ap.add_argument("steps", type=int, nargs="?", default=3000); ap.add_argument("--no-growth", action="store_true")
ap.add_argument("--every", type=int, default=250); ap.add_argument("--no-audio", action="store_true")
ap.add_argument("--camera-opt", action="store_true", help="scenario G8: camera optimizer SO3xR3 on")
ap.add_argument("--rays", type=int, default=0); ap.add_argument("--start-audio", type=int, default=-1)
ap.add_argument("--full-size", action="store_true", help="the bench's shapes on this scene: 4096 rays + 2048 RIR slices per iteration, 128^3 grid, audio branch from iteration 2000 (the reference's start_step_audio)")
ap.add_argument("--debug-from", type=int, default=-1, help="from this iteration on: report the first iterations whose GradScaler scale drops, with the non-finite gradients")
a = ap.parse_args()
import numpy as np, torch
import trajectory_common as TC
from neraf_amd import config as Cfg, synth
from neraf_amd.model import NeRAFAudioModel, NeRAFAudioModelConfig
from neraf_amd.pipeline import NeRAFPipeline
from neraf_amd.vision import NeRAFVisionModel, RayBundle
dev = torch.device("cuda:0"); cfg = TC.CFG; T = TC.T
if a.full_size:
    cfg.update(R=4096, B=2048, grid_step=1 / 128, start_step_audio=2000)      # in place: trajectory_common's helpers read this dict
if a.rays:
    cfg.update(R=a.rays)
if a.start_audio >= 0:
    cfg.update(start_step_audio=a.start_audio)
if a.camera_opt:
    vm = Cfg.NeRAFVisionModelConfig(camera_optimizer=Cfg.CameraOptimizerConfig(mode="SO3xR3")).setup(
        scene_box=Cfg.SceneBox(torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), num_train_data=cfg["n_cam"], metadata={}, device=dev, grad_scaler=None, seed_points=None)
else:
    vm = NeRAFVisionModel(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), cfg["n_cam"])
P, sdn, sdr = TC.initial_weights((vm.proposal_networks[0].table.shape[0], vm.proposal_networks[1].table.shape[0], vm.field.module.table.shape[0]))
with torch.no_grad():
    for i in range(2):
        vm.proposal_networks[i].table.copy_(P[f"prop{i}.table"]); vm.proposal_networks[i].w0.copy_(P[f"prop{i}.w0"]); vm.proposal_networks[i].w1.copy_(P[f"prop{i}.w1"])
    f = vm.field.module
    f.table.copy_(P["field.table"])
    for k in ("base_w0", "base_w1", "head_w0", "head_w1", "head_w2", "embedding"):
        getattr(f, k).copy_(P["field." + k])
am = NeRAFAudioModel(NeRAFAudioModelConfig(dataset="RAF", grid_step=cfg["grid_step"]), T(synth.audio_aabb()))
am.field.load_state_dict(sdn); am.resnet3d.backbone_net.load_state_dict(sdr)
vm.to(dev).train(); am.to(dev).train()
bank = TC.rir_bank(cfg["n_rir"], cfg["tag"] + ".train")
rays = TC._StepRays(dev)
vm.jitter_fn = lambda step, R, device: rays.jit[step]
start_audio = 10 ** 9 if a.no_audio else cfg["start_step_audio"]
pipe = NeRAFPipeline(vm, am, datamanager=rays, audio_datamanager=TC._StepSlices(bank, dev), start_step_audio=start_audio, world_size=1, local_rank=0)
opts, scaler = pipe.make_optimizers(init_scale=65536.0, optimizers_config=Cfg.default_optimizers(start_audio), with_schedulers=True)
if a.no_growth:
    scaler.set_growth_interval(10 ** 9)
ev = synth.trajectory_eval_camera(*cfg["eval_hw"], tag=cfg["tag"])
acc = []
prev_scale, reported = 65536.0 * 4, 0
hist, ema, jumps = [], None, 0
def vision_P():
    f = vm.field.module
    P = {"field.table": f.table, **{"field." + k: getattr(f, k) for k in ("base_w0", "base_w1", "head_w0", "head_w1", "head_w2", "embedding")}}
    for i in range(2):
        pn = vm.proposal_networks[i]
        P.update({f"prop{i}.table": pn.table, f"prop{i}.w0": pn.w0, f"prop{i}.w1": pn.w1})
    return P


g_ema, oracle_done = None, 0
for s in range(a.steps):
    pre = {k: v.detach().clone() for k, v in vision_P().items()} if (a.debug_from >= 0 and s >= a.debug_from and oracle_done < 2) else None
    _, ld = pipe.train_iteration(s, opts, scaler)
    acc.append([float(ld[k]) if k in ld else float("nan") for k in ("rgb_loss", "interlevel_loss", "distortion_loss", "audio_mag_loss")])
    rays.jit.pop(s, None)
    if a.debug_from >= 0 and s >= a.debug_from:
        # per-tensor gradient amax (unscaled by the GradScaler's scale as of this step) and parameter amax of the radiance model
        sc_now = scaler.get_scale()
        row = {n: (float(p.grad.abs().max()) / max(prev_scale if prev_scale < 1e9 else 65536.0, 1e-30), float(p.detach().abs().max()))
               for n, p in vm.named_parameters() if p.grad is not None}
        hist.append((s, float(ld["rgb_loss"]), float(ld["interlevel_loss"]), row))
        hist[:] = hist[-6:]
        ema = float(ld["rgb_loss"]) if ema is None else 0.98 * ema + 0.02 * float(ld["rgb_loss"])
        if jumps < 2 and float(ld["rgb_loss"]) > 8 * ema:
            jumps += 1
            print(f"iteration {s}: rgb loss {float(ld['rgb_loss']):.5f} against a running mean of {ema:.5f}; the last iterations (gradient amax / parameter amax per tensor):", flush=True)
            for (ss, r, il, rw) in hist:
                print(f"iteration {ss}:   rgb {r:.5f} interlevel {il:.5f}  " + "  ".join(f"{n.split('.')[-1] if 'proposal' not in n else 'p' + n.split('.')[1] + '.' + n.split('.')[-1]}: {g:.2e}/{pm:.2f}" for n, (g, pm) in rw.items()), flush=True)
        gt_now = row.get("field.module.table", (0.0, 0.0))[0]
        if pre is not None and g_ema is not None and gt_now > 200 * g_ema and oracle_done < 2:
            # the same step in the CPU oracle, from the parameters as they were BEFORE this iteration (radiance losses only)
            oracle_done += 1
            from oracle import vision as V
            used = prev_scale if prev_scale < 1e9 else 65536.0
            updated = vm._steps_since_update == 0
            Pl = {k: v.detach().cpu().float().clone().requires_grad_(True) for k, v in pre.items()}
            rb = TC.ray_batch(s)
            out = V.nerfacto_forward(rb["origins"], rb["directions"], rb["camera_indices"], Pl, V.NerfactoSpec(), step=s, training=True, jitters=rb["jitters"])
            if not updated:
                out["weights_list"] = [w.detach() for w in out["weights_list"][:-1]] + [out["weights_list"][-1]]
            ldo = V.vision_loss_dict(out, rb["rgb"], V.NerfactoSpec())
            sum(ldo.values()).backward()
            cur = vision_P()
            msg = []
            for k in ("field.table", "field.base_w0", "field.base_w1", "field.head_w2"):
                gh = (cur[k].grad.detach().cpu().float() / used).flatten(); go = Pl[k].grad.flatten()
                msg.append(f"{k}: HIP amax {float(gh.abs().max()):.3e} oracle amax {float(go.abs().max()):.3e} cosine {float((gh @ go) / (gh.norm() * go.norm() + 1e-30)):.4f}")
            # the forward of the same state in both: where do they part?
            with torch.no_grad():
                for k, v in vision_P().items():
                    v.copy_(pre[k])
                jit = [j.to(dev) for j in rb["jitters"]]
                oh = vm.get_outputs(RayBundle(rb["origins"].to(dev), rb["directions"].to(dev), rb["camera_indices"].to(dev)), jitters=jit)
            drgb = (oh["rgb"].cpu() - out["rgb"].detach()).abs().amax(1)
            worst = torch.argsort(drgb, descending=True)[:4]
            sb_h = [x.s_bins.cpu() for x in oh["ray_samples_list"]]; sb_o = [x.s_bins.detach() for x in out["ray_samples_list"]]
            dh, do = oh["density"].cpu().reshape(sb_h[-1].shape[0], -1), out["density"].detach().reshape(sb_o[-1].shape[0], -1)
            wh, wo = [w.cpu() for w in oh["weights_list"]], [w.detach() for w in out["weights_list"]]
            print(f"iteration {s}:   forward of that state: max |rgb HIP - oracle| {float(drgb.max()):.4f} (rays over 0.01: {int((drgb > 0.01).sum())} of {drgb.numel()}); "
                  f"density max HIP {float(dh.max()):.4g} oracle {float(do.max()):.4g}; non-finite HIP densities {int((~torch.isfinite(dh)).sum())}; "
                  f"sample bins max |HIP - oracle|: level0 {float((sb_h[0] - sb_o[0]).abs().max()):.2e} level1 {float((sb_h[1] - sb_o[1]).abs().max()):.2e} fine {float((sb_h[2] - sb_o[2]).abs().max()):.2e}; "
                  f"weights max |diff|: {[round(float((a_ - b_).abs().max()), 4) for a_, b_ in zip(wh, wo)]}", flush=True)
            for r_ in worst.tolist():
                print(f"iteration {s}:     ray {r_}: rgb HIP {[round(float(v), 4) for v in oh['rgb'][r_].cpu()]} oracle {[round(float(v), 4) for v in out['rgb'][r_].detach()]} target {[round(float(v), 4) for v in rb['rgb'][r_]]}; "
                      f"density max HIP {float(dh[r_].max()):.4g} oracle {float(do[r_].max()):.4g}; fine-bin max diff {float((sb_h[2][r_] - sb_o[2][r_]).abs().max()):.3e}; "
                      f"proposal weights max diff {float((wh[0][r_] - wo[0][r_]).abs().max()):.4f} / {float((wh[1][r_] - wo[1][r_]).abs().max()):.4f}; fine weights max diff {float((wh[2][r_] - wo[2][r_]).abs().max()):.4f}", flush=True)
            break
            print(f"iteration {s}: gradient spike (table amax {gt_now:.3e} against a running mean of {g_ema:.3e}); proposal networks updated: {updated}; "
                  f"oracle losses { {k: round(float(v), 6) for k, v in ldo.items()} } vs HIP {float(ld['rgb_loss']):.6f} / {float(ld['interlevel_loss']):.6f} / {float(ld['distortion_loss']):.6f}; "
                  + "; ".join(msg), flush=True)
        g_ema = gt_now if g_ema is None else (0.95 * g_ema + 0.05 * min(gt_now, 10 * g_ema))
        sc = scaler.get_scale()
        if sc < prev_scale and reported < 4:
            reported += 1
            named = [("vision." + n, p) for n, p in vm.named_parameters()] + [("audio." + n, p) for n, p in am.named_parameters()]
            bad = [(n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for n, p in named if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
            none = [n for n, p in named if p.grad is None]
            badp = [n for n, p in named if not bool(torch.isfinite(p).all())]
            print(f"iteration {s}: scale {prev_scale:.0f} -> {sc:.0f}; losses { {k: float(v) for k, v in ld.items()} }; non-finite gradients: {bad[:8]}; "
                  f"parameters without a gradient: {len(none)} {none[:6]}; non-finite PARAMETERS: {badp[:8]}; proposal updated this step: {getattr(vm, '_steps_since_update', None)}", flush=True)
        prev_scale = sc
    if (s + 1) % a.every == 0:
        m = np.nanmean(np.asarray(acc), 0); acc = []
        vm.eval()
        img = vm.get_outputs_for_camera_ray_bundle(RayBundle(T(ev["origins"]).to(dev), T(ev["directions"]).to(dev), None))["rgb"].reshape(*cfg["eval_hw"], 3).cpu().numpy()
        vm.train()
        print(f"iteration {s + 1:6d}: rgb {m[0]:.5f} interlevel {m[1]:.5f} distortion {m[2]:.5f} audio_mag {m[3]:.5f}  scale {scaler.get_scale():.0f}  "
              f"held-out PSNR {TC.psnr(img, np.asarray(ev['image'])):.2f} dB  |field table| max {float(vm.field.module.table.abs().max()):.3f}"
              + (f"  |pose deltas| {float(vm.camera_optimizer.pose_adjustment.detach().norm()):.4f}" if a.camera_opt else ""), flush=True)

#!/usr/bin/env python3
"""Diagnostic: d loss / d (camera pose deltas) of the HIP pipeline against the CPU oracle at the trajectory scenario's initial state
(identical weights, iteration-0 and iteration-k batches; pose deltas set to a small non-zero value so that the rotation part has a
gradient).  Prints cosine similarity and rel-L2 of the 12 x 6 pose gradient for the photometric losses alone and in total."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np
import torch

import trajectory_common as TC
from neraf_amd import config as Cfg
from neraf_amd.vision import RayBundle
from oracle import vision as V
from oracle.trainer import exp_map_so3xr3

dev = torch.device("cuda:0")
cfg = TC.SCENARIOS["g8_trajectory_pose"]
vcfg = Cfg.NeRAFVisionModelConfig(camera_optimizer=Cfg.CameraOptimizerConfig(mode="SO3xR3"))
vm = vcfg.setup(scene_box=Cfg.SceneBox(torch.tensor([[-1.0, -1, -1], [1, 1, 1]])), num_train_data=cfg["n_cam"], metadata={}, device=dev,
                grad_scaler=None, seed_points=None)
P, _, _ = TC.initial_weights((vm.proposal_networks[0].table.shape[0], vm.proposal_networks[1].table.shape[0], vm.field.module.table.shape[0]))
# trained-like magnitudes: the initial tables are U(-1e-4, 1e-4), where every gradient is ~0
g = torch.Generator().manual_seed(0)
for k in list(P):
    if k.endswith("table"):
        P[k] = (torch.rand(P[k].shape, generator=g) - 0.5) * 0.5
with torch.no_grad():
    for i in range(2):
        vm.proposal_networks[i].table.copy_(P[f"prop{i}.table"]); vm.proposal_networks[i].w0.copy_(P[f"prop{i}.w0"]); vm.proposal_networks[i].w1.copy_(P[f"prop{i}.w1"])
    f = vm.field.module
    f.table.copy_(P["field.table"])
    for k in ("base_w0", "base_w1", "head_w0", "head_w1", "head_w2", "embedding"):
        getattr(f, k).copy_(P["field." + k])
vm.to(dev).train()
pose0 = (torch.rand((cfg["n_cam"], 6), generator=g) - 0.5) * 0.02
with torch.no_grad():
    vm.camera_optimizer.pose_adjustment.copy_(pose0.to(dev))
spec = V.NerfactoSpec()
P16 = {k: v.half().float() for k, v in P.items()}
for step in (0, 50):
    b = TC.ray_batch(step)
    vm.update_to_step(step + 200)
    vm.zero_grad(set_to_none=True)
    jit = [j.reshape(-1).to(dev) for j in b["jitters"]]
    out = vm.get_outputs(RayBundle(b["origins"].to(dev), b["directions"].to(dev), b["camera_indices"].to(dev)), jitters=jit)
    ld = vm.get_loss_dict(out, {"image": b["rgb"].to(dev)}, {})
    (ld["rgb_loss"] + ld["interlevel_loss"] + ld["distortion_loss"]).backward()
    gh = vm.camera_optimizer.pose_adjustment.grad.detach().cpu().double()
    pose = pose0.clone().requires_grad_(True)
    corr = exp_map_so3xr3(pose[b["camera_indices"].long()])
    o = b["origins"] + corr[:, :3, 3]
    d = torch.bmm(corr[:, :3, :3], b["directions"][..., None]).squeeze(-1)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P16.items()}
    oo = V.nerfacto_forward(o, d, b["camera_indices"], Pg, spec, step=step + 200, training=True, jitters=b["jitters"])
    lo = V.vision_loss_dict(oo, b["rgb"], spec)
    (lo["rgb_loss"] + lo["interlevel_loss"] + lo["distortion_loss"]).backward()
    go = pose.grad.double()
    cos = float((gh * go).sum() / (gh.norm() * go.norm()))
    print(f"step {step}: |g_hip| {float(gh.norm()):.4e} |g_oracle| {float(go.norm()):.4e} cosine {cos:.4f} rel-L2 {float((gh - go).norm() / go.norm()):.3e}")
    print("   per-camera cosine:", [round(float((gh[c] * go[c]).sum() / (gh[c].norm() * go[c].norm() + 1e-30)), 3) for c in range(cfg["n_cam"])])
    print("   translation part rel-L2 %.3e rotation part rel-L2 %.3e" % (float((gh[:, :3] - go[:, :3]).norm() / go[:, :3].norm()), float((gh[:, 3:] - go[:, 3:]).norm() / go[:, 3:].norm())))
    print("   losses hip", {k: float(v) for k, v in ld.items() if "camera" not in k}, "oracle", {k: float(v) for k, v in lo.items()})

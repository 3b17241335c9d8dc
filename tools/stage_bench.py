#!/usr/bin/env python3
"""Per-stage timings of the hot path on the MI355X (torch events on the launch stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neraf_amd import synth
from neraf_amd.model import NeRAFAudioModel, NeRAFAudioModelConfig
from neraf_amd.vision import NeRAFVisionModel, RayBundle
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
vm = NeRAFVisionModel(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 210).to(dev)
with torch.no_grad():
    vm.field.module.table.uniform_(-0.5, 0.5)
    for p in vm.proposal_networks: p.table.uniform_(-0.5, 0.5)
am = NeRAFAudioModel(NeRAFAudioModelConfig(dataset="RAF", grid_step=1 / 128), T(synth.audio_aabb())).to(dev)
rb = synth.ray_batch(4096)
bundle = RayBundle(T(rb["origins"]).to(dev), T(rb["directions"]).to(dev), T(rb["camera_indices"]).to(dev))
b = {k: T(v).to(dev) for k, v in synth.audio_batch(2048, 1, 513, 60).items()}
def timeit(name, fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / n * 1e3:9.1f} us")
vm.train(); am.train()
timeit("vision forward 4096 rays (train)", lambda: vm.get_outputs(bundle))
timeit("grid refresh 4096 cells x 18 dirs", lambda: am.query_grid_one_batch(0, vm.field, vm.renderer_rgb, 4096))
timeit("resnet3d forward 7x128^3 (train)", lambda: am.scene_feature())
timeit("audio get_outputs B=2048 (train)", lambda: am.get_outputs(b))
am.eval()
timeit("resnet3d forward (eval, cached)", lambda: am.scene_feature())
vm.eval()
big = synth.ray_batch(32768, tag="big")
bb = RayBundle(T(big["origins"]).to(dev), T(big["directions"]).to(dev), None)
timeit("vision eval chunk 32768 rays", lambda: vm.get_outputs(bb), n=10)

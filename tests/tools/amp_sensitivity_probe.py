#!/usr/bin/env python3
"""Container-only probe (imports /root/reference): how far do the reference ResNet3D's OWN gradients move when it runs in
its own training precision (fp16 autocast, NeRAF_config.py:79) instead of fp32?  Prints relative L2 deviations.

Measured here (64^3 grid, synthetic weights, CPU autocast): feature 4.7e-3; weight gradients 0.41 (layer3.5.conv3),
0.55 (layer3.0.conv2), 0.60 (layer2.0.conv2), 0.61 (layer1.0.conv2), 0.61 (conv1), 0.71 (bn1.weight).  The HIP engine's
deviations from the fp32 golden are the same figures, which is why tests/test_gpu_resnet3d.py checks gradients against the
rounding-matched oracle (oracle.audio.resnet3d_forward_gated(fp16_storage=True) with the engine's gates) instead."""
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import gen_golden as g
import torch

from neraf_amd import synth

if __name__ == "__main__":
    r3, *_ = g.import_reference()
    S = 64

    def run(amp):
        net = r3.ResNet3D_helper(in_channels=7, backbone="resnet50", grid_step=1 / S, N_features=1024)
        net.backbone_net.load_state_dict({k: g.t(v) for k, v in synth.resnet3d_state_dict(7).items()})
        net.train()
        x = g.t(synth.uniform(f"g1.grid{S}", (1, 7, S, S, S), 0.0, 1.0))
        wsum = g.t(synth.uniform("g1.outw", (1024,), -1.0, 1.0))
        if amp:
            with torch.autocast("cpu", dtype=amp):
                y = net(x)
        else:
            y = net(x)
        (y.float().flatten() * wsum).sum().backward()
        return y.detach().float().flatten(), {n: p.grad.clone() for n, p in net.backbone_net.named_parameters()}

    y0, g0 = run(None)
    y1, g1 = run(torch.float16)
    print("feature rel-L2 (fp16 autocast vs fp32):", float((y1 - y0).norm() / y0.norm()))
    for n in ("layer3.5.conv3.weight", "layer3.0.conv2.weight", "layer2.0.conv2.weight", "layer1.0.conv2.weight", "conv1.weight", "bn1.weight"):
        print(f"  grad {n}: {float((g1[n].float() - g0[n]).norm() / g0[n].norm()):.3f}")

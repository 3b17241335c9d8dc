#!/usr/bin/env python3
"""ResNet3D forward + backward alone, N iterations (run under `rocprofv3 --kernel-trace`): tools/resnet_trace_summary.py then folds
the dispatch timeline by position in the launch sequence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neraf_amd import synth
from neraf_amd.resnet3d import ResNet3D_helper
dev = torch.device("cuda:0")
net = ResNet3D_helper(in_channels=7, backbone="resnet50", pretrained=False, grid_step=1 / 128, N_features=1024)
net.backbone_net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.resnet3d_state_dict(7).items()})
net.to(dev).train()
x = torch.from_numpy(synth.uniform("g1.grid128", (1, 7, 128, 128, 128), 0.0, 1.0)).to(dev)
w = torch.ones(1024, device=dev)
bb = net.backbone_net
bb.grid_window = (0, 4096, 4)
bb.grid_grad_sink = lambda d: None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for it in range(n):
    (net(x).flatten() * w).sum().backward()
torch.cuda.synchronize()
print("done", n)

#!/usr/bin/env python3
"""Un-profiled cost of every launch in the ResNet3D forward / backward sequences: replay only the first n nodes of the captured hipGraph
(NERAF_GRAPH_TRUNC, csrc/common.h) for n = 0..N and difference the iteration times.  rocprofv3's kernel trace cannot give this: it reports
>= 4.7 us for any kernel of a replayed graph.  Usage: graph_prefix_times.py [S=128] [iters=20] [stride=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neraf_amd import synth
from neraf_amd.resnet3d import ResNet3D_helper

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
net = ResNet3D_helper(in_channels=7, backbone="resnet50", pretrained=False, grid_step=1 / S, N_features=1024)
net.backbone_net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.resnet3d_state_dict(7).items()})
net.to(dev).train()
x = torch.from_numpy(synth.uniform(f"g1.grid{S}", (1, 7, S, S, S), 0.0, 1.0)).to(dev)
w = torch.ones(1024, device=dev)
bb = net.backbone_net
bb.grid_window = (0, 4096, 4)
bb.grid_grad_sink = lambda d: None


def run(nf, nb):
    """ms per iteration with the forward graph cut to nf nodes and the backward graph to nb (None = whole)."""
    def it():
        if nf is None: os.environ.pop("NERAF_GRAPH_TRUNC", None)
        else: os.environ["NERAF_GRAPH_TRUNC"] = str(nf)
        y = net(x)
        if nb is None: os.environ.pop("NERAF_GRAPH_TRUNC", None)
        else: os.environ["NERAF_GRAPH_TRUNC"] = str(nb)
        (y.flatten() * w).sum().backward()
    for _ in range(3):
        it()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        it()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


full = run(None, None)
print(f"S={S}: whole forward + backward {full:.1f} us per iteration")
for name, total, mk in (("forward", 140, lambda n: (n, 0)), ("backward", 320, lambda n: (None, n))):
    prev, last_n = None, 0
    base = run(*mk(0))
    print(f"--- {name}: prefix n -> iteration us (minus n=0: {base:.1f}), increment per node")
    prev = base
    flat = 0
    for n in range(stride, total + 1, stride):
        t = run(*mk(n))
        print(f"{name} {n:4d} {t - base:9.1f} {(t - prev) / stride:7.2f}")
        flat = flat + 1 if abs(t - prev) < 0.3 else 0
        prev = t
        if flat >= 6:
            break

"""Run-to-run spread of the full-chain ResNet3D backward statistics that tests/test_gpu_resnet3d.py::test_resnet3d_backward_full_chain
asserts (the forward's BatchNorm statistics are summed with fp32 atomics, and the randomly initialised network amplifies their
last bit -- DESIGN.md "Chaos")."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_resnet3d as t
from neraf_amd import synth
T = t.T
dev = torch.device("cuda:0")
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g1_resnet3d_64.npz")))
S = 64
net = t._model(dev, 1 / S); net.train(); bb = net.backbone_net
x = T(synth.uniform(f"g1.grid{S}", (1, 7, S, S, S), 0.0, 1.0)).to(dev)
wsum = T(synth.uniform("g1.outw", (1024,), -1.0, 1.0)).to(dev)
rows = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    for p in bb.parameters(): p.grad = None
    got = {}
    bb.grid_window = (0, S ** 3, 7); bb.grid_grad_sink = lambda d: got.__setitem__("dx", d.clone())
    y = net(x); (y.flatten() * wsum).sum().backward()
    bb.grid_window, bb.grid_grad_sink = None, None
    nr = lambda a, b: float(a.double().cpu().norm() / T(b).double().norm())
    dx = got["dx"].reshape(7, S, S, S).cpu()
    pi = g["probe_idx"]
    rows.append([nr(bb.conv1.weight.grad, g["dw_conv1"]), nr(bb.bn1.weight.grad, g["dgamma_bn1"]),
                 nr(bb.layer2[0].downsample[1].weight.grad, g["dgamma_l2_0_ds"]),
                 bb.layer1[0].conv2.weight.grad.double().pow(2).mean().sqrt().item() / g["dw_l1_0_conv2_stats"][2],
                 bb.layer3[5].conv3.weight.grad.double().pow(2).mean().sqrt().item() / g["dw_l3_5_conv3_stats"][2],
                 t.rel_l2(bb.conv1.weight.grad, T(g["dw_conv1"])), t.rel_l2(bb.bn1.bias.grad, T(g["dbeta_bn1"])),
                 dx.double().pow(2).mean().sqrt().item() / g["dx_stats"][2],
                 t.rel_l2(dx[pi[:, 0], pi[:, 1], pi[:, 2], pi[:, 3]], T(g["dx_probe"]))])
a = np.array(rows)
names = ["norm conv1.w", "norm bn1.gamma", "norm l2.0.ds.gamma", "rms l1.0.conv2.w", "rms l3.5.conv3.w", "relL2 conv1.w", "relL2 bn1.beta", "rms dx", "relL2 dx probe"]
for i, n in enumerate(names):
    print(f"{n:20s} min {a[:, i].min():.4f} mean {a[:, i].mean():.4f} max {a[:, i].max():.4f}")

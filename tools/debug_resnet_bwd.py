import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neraf_amd import synth
from neraf_amd.resnet3d import ResNet3D_helper
from oracle import audio as O
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
dev = torch.device("cuda:0")
S = int(os.environ.get("DBG_S", "64"))
net = ResNet3D_helper(7, "resnet50", False, 1 / S, 1024)
sd = {k: T(v) for k, v in synth.resnet3d_state_dict(7).items()}
net.backbone_net.load_state_dict(sd)
for m in net.modules():
    if isinstance(m, torch.nn.BatchNorm3d): m.momentum = 0.0
net = net.to(dev).train()
x = T(synth.uniform(f"g1.grid{S}", (1, 7, S, S, S), 0.0, 1.0))
wsum = T(synth.uniform("g1.outw", (1024,), -1.0, 1.0))
y = net(x.to(dev)); (y.flatten() * wsum.to(dev)).sum().backward()
sdo = {k: v.clone() for k, v in sd.items()}
for k, v in sdo.items():
    if v.is_floating_point() and "running" not in k: v.requires_grad_(True)
torch.set_num_threads(32)
yo = O.resnet3d_forward_fp16_storage(x, sdo); (yo.flatten() * wsum).sum().backward()
print('feature rel', float((y.detach().cpu().flatten()-yo.detach().flatten()).norm()/yo.norm()))
names = [n for n, p in net.backbone_net.named_parameters()]
for n in reversed(names):
    p = dict(net.backbone_net.named_parameters())[n]
    a, b = p.grad.double().cpu(), sdo[n].grad.double()
    print(f"{n:40s} rel {float((a-b).norm()/(b.norm()+1e-30)):.3e}  |ref| {float(b.norm()):.3e}")

import os, sys, ctypes as C
os.environ["NERAF_RESNET_BWD_STOP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neraf_amd import synth, _lib
from neraf_amd.resnet3d import ResNet3D_helper
from oracle import audio as O
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
dev = torch.device("cuda:0")
S = 64
net = ResNet3D_helper(7, "resnet50", False, 1 / S, 1024)
sd = {k: T(v) for k, v in synth.resnet3d_state_dict(7).items()}
net.backbone_net.load_state_dict(sd)
for m in net.modules():
    if isinstance(m, torch.nn.BatchNorm3d): m.momentum = 0.0
net = net.to(dev).train()
x = T(synth.uniform(f"g1.grid{S}", (1, 7, S, S, S), 0.0, 1.0))
wsum = T(synth.uniform("g1.outw", (1024,), -1.0, 1.0))
y = net(x.to(dev)); (y.flatten() * wsum.to(dev)).sum().backward()
torch.cuda.synchronize()
bb = net.backbone_net
lib = _lib.load()
off = (C.c_size_t * 10)()
lib.neraf_resnet3d_bwd_debug_offsets(C.byref(bb._desc), off)
bws = bb._bws
def view(o, rows, cols):
    return bws[o:o + rows * cols * 2].view(torch.bfloat16).reshape(rows, cols).float().cpu()
scale = bws[0:16].view(torch.float32).cpu()
print("scale", scale)
sdo = {k: v.clone() for k, v in sd.items()}
for k, v in sdo.items():
    if v.is_floating_point() and "running" not in k: v.requires_grad_(True)
torch.set_num_threads(32)
taps = {}
yo = O.resnet3d_forward(x, sdo, train=True, taps=taps); (yo.flatten() * wsum).sum().backward()
def cl(t):  # [1,C,D,H,W] -> [M,C]
    return t[0].permute(1, 2, 3, 0).reshape(-1, t.shape[1])
S_ = float(scale[0])
dy2 = view(off[4], 128, 1024)[:64] / S_
ref = cl(taps["layer3.5.c3"].grad)
print("dy2 (grad wrt pre-bn3 conv3 out) rel", float((dy2 - ref).norm() / ref.norm()), float(ref.norm()), float(dy2.norm()))
da = view(off[7], 128, 256)[:64] / S_
ref = cl(taps["layer3.5.a2"].grad)
print("d a2 rel", float((da - ref).norm() / ref.norm()))
g = view(off[0], 128, 1024)[:64] / S_
print("g rows", g[0, :4], wsum[:4] / 64)
dyT = view(off[8], 1024, 128)[:, :64] / S_
print("dyT vs dy2^T", float((dyT - dy2.T).abs().max()))
xT = view(off[9], 256, 128)[:, :64]
a2 = cl(taps["layer3.5.a2"].detach())
print("xT vs a2^T rel", float((xT - a2.T).norm() / a2.norm()))
gw = bb.layer3[5].conv3.weight.grad[:, :, 0, 0, 0].cpu()
print("dW conv3 rel", float((gw - sdo["layer3.5.conv3.weight"].grad[:, :, 0, 0, 0]).norm() / sdo["layer3.5.conv3.weight"].grad.norm()))
print("dW from oracle operands:", float(((cl(taps["layer3.5.c3"].grad).T @ a2) - sdo["layer3.5.conv3.weight"].grad[:, :, 0, 0, 0]).norm()))
print("dW from GPU operands  :", float(((dy2.T @ xT.T) - gw).norm() / gw.norm()))
# --- which BN-backward term is off?
c3 = cl(taps["layer3.5.c3"].detach()).double()
outo = None
gam = sd["layer3.5.bn3.weight"].double()
mean = c3.mean(0); var = c3.var(0, unbiased=False); rstd = 1 / torch.sqrt(var + 1e-5)
xh = (c3 - mean) * rstd
# oracle block output: recompute mask from oracle forward (relu(bn3 + res) > 0) via grad trick: mask = (grad wrt bn3 output != 0)
dfeat = (wsum / 64).double()
ref = cl(taps["layer3.5.c3"].grad).double()
# infer dy from ref: dy - mean(dy) - xh*mean(dy*xh) = ref/(gam*rstd); try candidate masks from GPU 'out'
print("ref stats", float(ref.norm()))
for name, cand in (("k1*dy_nomask", gam * rstd * dfeat[None, :].expand(64, -1)),):
    print(name, float((dy2.double() - cand).norm() / ref.norm()))
d = dy2.double() / (gam * rstd)       # = dy - k2 - xh*k3 (GPU)
r = ref / (gam * rstd)                # oracle
print("col-mean of GPU d (should be 0):", float(d.mean(0).abs().max()), " oracle:", float(r.mean(0).abs().max()))
print("col <d,xh> GPU (should be 0):", float((d * xh).mean(0).abs().max()), " oracle:", float((r * xh).mean(0).abs().max()))
print("corr(d, r) per column (first 8):", [round(float(torch.corrcoef(torch.stack([d[:, c], r[:, c]]))[0, 1]), 3) for c in range(8)])

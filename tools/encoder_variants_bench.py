#!/usr/bin/env python3
"""The scene encoder alone in every configuration the reference's constructor accepts (NeRAF_resnet3d.py:128-156): N_features 1024 |
2048 (layer4) on the 64^3 | 128^3 | 256^3 grid.  Training call (forward with batch statistics + backward with a 4096-cell refresh
window, as the joint step issues it), timed with HIP events on the current stream over N iterations after warm-up (both graphs
captured, chain calibrated).  Algorithmic FLOPs from the library's own architecture table (neraf_resnet3d_forward_flops; backward =
2 x forward: dgrad + wgrad).

    python tools/encoder_variants_bench.py [N=20]  > profiles/<tag>_encoder_variants.txt
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neraf_amd import _lib, synth
from neraf_amd.resnet3d import ResNet3D_helper

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = tuple(int(v) for v in sys.argv[2].split("x")) if len(sys.argv) > 2 else None      # e.g. 256x1024: that configuration alone (under rocprofv3)
dev = torch.device("cuda:0")
lib = _lib.load()
lib.neraf_resnet3d_forward_flops.restype = C.c_double
print(f"{'grid':>6s} {'N_feat':>6s} {'convs':>5s} {'fwd GFLOP':>10s} {'fwd ms':>8s} {'fwd+bwd ms':>10s} {'TFLOP/s':>8s} {'of 2.5 PF':>9s} {'workspaces GB':>13s}")
for S, N in ((64, 1024), (64, 2048), (128, 1024), (128, 2048), (256, 1024), (256, 2048)):
    if only and (S, N) != only:
        continue
    layers = (3, 4, 6, 3) if N == 2048 else (3, 4, 6)
    net = ResNet3D_helper(in_channels=7, backbone="resnet50", pretrained=False, grid_step=1 / S, N_features=N)
    net.backbone_net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.resnet3d_state_dict(7, layers=layers).items()})
    net.to(dev).train()
    bb = net.backbone_net
    x = torch.rand(1, 7, S, S, S, device=dev)
    w = torch.ones(N, device=dev)
    bb.grid_window = (0, 4096, 4)
    bb.grid_grad_sink = lambda d: None
    for _ in range(6):
        (net(x).flatten() * w).sum().backward()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    with torch.no_grad():
        for _ in range(n):
            net(x)
    e[1].record()
    for _ in range(n):
        (net(x).flatten() * w).sum().backward()
    e[2].record()
    torch.cuda.synchronize()
    fwd_ms, both_ms = e[0].elapsed_time(e[1]) / n, e[1].elapsed_time(e[2]) / n
    gf = lib.neraf_resnet3d_forward_flops(C.byref(bb._desc)) / 1e9
    tf = 3 * gf / both_ms
    ws = (bb._ws.numel() + bb._bws.numel() + bb._packed.numel() + bb._packed_t.numel()) / 1e9
    print(f"{S:4d}^3 {N:6d} {len(bb.conv_bn_pairs()):5d} {gf:10.2f} {fwd_ms:8.3f} {both_ms:10.3f} {tf:8.1f} {tf / 2500:9.3f} {ws:13.2f}")
    del net, bb, x
    torch.cuda.empty_cache()

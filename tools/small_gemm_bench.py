#!/usr/bin/env python3
"""Per-launch cost of the fp16 GEMM on the ResNet3D layer-2/3 shapes inside a replayed graph (dependent chain, rotating weight
buffers so that B is never cache-warm), next to an empty kernel chain: what a 512..4096-voxel 1x1x1 convolution costs un-profiled."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neraf_amd import _lib
lib = _lib.load(); h = _lib.ctx(0)
dev = torch.device("cuda:0")
shapes = [(512, 256, 1024), (512, 1024, 256), (512, 256, 6912), (4096, 128, 512), (4096, 512, 128), (4096, 128, 3456),
          (32768, 64, 256), (32768, 256, 64), (32768, 64, 1728)]
NB = int(os.environ.get("NB_BUFFERS", "24"))      # distinct weight buffers in the chain (1: the same B every launch, L2-warm)
NL = 24
side = torch.cuda.Stream()
for M, N, K in shapes:
    A = (torch.rand(M, K, device=dev) - 0.5).half()
    Bs = [(torch.rand(N, K, device=dev) - 0.5).half() for _ in range(NB)]
    bias = torch.zeros(N, device=dev)
    Cs = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(2)]
    def chain(st):
        for i in range(NL):
            _lib.check(lib.neraf_gemm_f16(h, A.data_ptr(), K, Bs[i % NB].data_ptr(), K, M, N, K, M, N, 1.0, bias.data_ptr(), 0,
                                          Cs[i & 1].data_ptr(), N, None, 0, None, 0, C.c_void_p(st)))
    with torch.cuda.stream(side):
        chain(side.cuda_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            chain(torch.cuda.current_stream().cuda_stream)
    for _ in range(3): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    R = 20
    for _ in range(R): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (R * NL)
    print(f"M{M:6d} N{N:5d} K{K:5d}: {us:7.2f} us per launch   {2.0*M*N*K/us/1e6:7.1f} TF/s   weights {N*K*2/1e6:.2f} MB", flush=True)

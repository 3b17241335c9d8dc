#!/usr/bin/env python3
"""Neighbour process for the GPU-sharing bisect: back-to-back launches of ONE fp16 GEMM shape of this package for --seconds."""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("shape", type=int, nargs=3); ap.add_argument("--seconds", type=float, default=50.0); ap.add_argument("--torch", action="store_true")
a = ap.parse_args()
import torch
from neraf_amd import _lib
lib = _lib.load(); h = _lib.ctx(0); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = a.shape
A = (torch.rand(M, K, device=dev) - 0.5).half(); B = (torch.rand(N, K, device=dev) - 0.5).half()
bias = torch.zeros(N, device=dev); Cs = torch.empty(M, N, dtype=torch.float16, device=dev)
t0 = time.time(); n = 0
while time.time() - t0 < a.seconds:
    for _ in range(200):
        if a.torch:
            torch.matmul(A, B.t(), out=Cs)
        else:
            _lib.check(lib.neraf_gemm_f16(h, A.data_ptr(), K, B.data_ptr(), K, M, N, K, M, N, 1.0, bias.data_ptr(), 0, Cs.data_ptr(), N, None, 0, None, 0, st))
    torch.cuda.synchronize(); n += 200
print(f"aggressor {M}x{N}x{K}: {n} launches", flush=True)

#!/usr/bin/env python3
"""The fp16 GEMM / conv kernels alone under GPU sharing: fixed operands, repeated launches, outputs compared bit for bit with the first
(these kernels have no atomics when no statistics are requested: they are deterministic).  `--pair` runs two processes at once."""
import argparse, ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--pair", action="store_true"); ap.add_argument("--iters", type=int, default=3000); ap.add_argument("--tag", default="A")
a = ap.parse_args()
if a.pair:
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--iters", str(a.iters), "--tag", t]) for t in ("A", "B")]
    sys.exit(max(p.wait() for p in ps))
import torch
from neraf_amd import _lib
lib = _lib.load(); h = _lib.ctx(0); dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [(512, 256, 1024), (512, 1024, 256), (4096, 128, 512), (4096, 512, 128), (32768, 64, 256), (32768, 256, 64), (2048, 1024, 2048), (2048, 5120, 2048)]
for M, N, K in shapes:
    A = (torch.rand(M, K, device=dev) - 0.5).half(); B = (torch.rand(N, K, device=dev) - 0.5).half()
    bias = torch.zeros(N, device=dev)
    Cs = torch.empty(M, N, dtype=torch.float16, device=dev)
    def run():
        _lib.check(lib.neraf_gemm_f16(h, A.data_ptr(), K, B.data_ptr(), K, M, N, K, M, N, 1.0, bias.data_ptr(), 0, Cs.data_ptr(), N, None, 0, None, 0, st))
    run(); torch.cuda.synchronize(); ref = Cs.clone()
    bad = 0; worst = 0.0
    n = max(200, a.iters * 512 * 256 // (M * N) // 4)
    for it in range(n):
        Cs.zero_(); run()
        if it % 16 == 15 or it == n - 1:
            pass
        if not torch.equal(Cs, ref):
            bad += 1; worst = max(worst, float((Cs.float() - ref.float()).abs().max()))
    print(f"[{a.tag}] gemm {M}x{N}x{K}: {bad} of {n} launches differ (max |d| {worst:.3e})", flush=True)

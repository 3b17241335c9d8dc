#!/usr/bin/env python3
"""Micro-benchmark of libneraf_hip's fp16 GEMM on the NAcF shapes (A/B of NERAF_GEMM_VARIANT)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neraf_amd import _lib
lib = _lib.load(); h = _lib.ctx(0)
dev = torch.device("cuda:0")
import sys
if len(sys.argv) > 1 and sys.argv[1] == "m6464":
    shapes_override = [(6464, 5096, 192, 1), (6464, 2048, 5120, 1), (6464, 1024, 2048, 1), (6464, 1024, 1024, 1), (6464, 512, 1024, 1),
                       (6464, 5096, 2048, 1), (5096, 2048, 6464, 4), (2048, 1024, 6464, 4), (8192, 8192, 8192, 1)]
else:
    shapes_override = None
shapes = [  # (M, N, K, outputs)  outputs: 1=C16, 2=C16T, 4=C32
    (2048, 5096, 192, 3), (2048, 2048, 5120, 3), (2048, 1024, 2048, 3), (2048, 1024, 1024, 3), (2048, 512, 1024, 3),
    (2048, 513, 512, 4), (2048, 5096, 2048, 3), (2048, 5096, 2048, 4), (1024, 2048, 2048, 4), (5096, 163, 2048, 4),
    (4096, 4096, 4096, 1),
]
if shapes_override:
    shapes = shapes_override
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
print("variant", os.environ.get("NERAF_GEMM_VARIANT", "default"))
for M, N, K, o in shapes:
    Mp, Np = (M + 127) // 128 * 128, (N + 127) // 128 * 128
    A = (torch.rand(Mp, K, device=dev) - 0.5).half(); B = (torch.rand(Np, K, device=dev) - 0.5).half()
    bias = torch.zeros(Np, device=dev)
    C16 = torch.empty(Mp, Np, dtype=torch.float16, device=dev) if o & 1 else None
    C16T = torch.empty(Np, Mp, dtype=torch.float16, device=dev) if o & 2 else None
    C32 = torch.empty(M, N, device=dev) if o & 4 else None
    def run():
        _lib.check(lib.neraf_gemm_f16(h, A.data_ptr(), K, B.data_ptr(), K, M, N, K, Mp, Np, 1.0, bias.data_ptr(), 1,
                                      C16.data_ptr() if C16 is not None else None, Np,
                                      C16T.data_ptr() if C16T is not None else None, Mp,
                                      C32.data_ptr() if C32 is not None else None, N, st))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    n = 20
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    lib_us = float("nan")
    if os.environ.get("NERAF_GEMM_BENCH_LIB"):      # hipBLASLt through torch, as an orientation figure only
        Bt = B.t()
        for _ in range(3): torch.matmul(A, Bt)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): torch.matmul(A, Bt)
        e1.record(); torch.cuda.synchronize()
        lib_us = e0.elapsed_time(e1) * 1e3 / n
    print(f"M{M:5d} N{N:5d} K{K:5d} out{o}: {us:8.1f} us  {2.0*M*N*K/us/1e6:8.1f} TF/s   (library fp16 matmul, padded shape: {lib_us:8.1f} us)")

#!/usr/bin/env python3
"""VERDICT r3 #5, "K-major operand" lead of the weight-gradient GEMM, as a measured A/B on the shapes where it needs no tap shifts
(the 1x1x1 convolutions: dW [cout][cin] = sum_v dY[v][cout] X[v][cin]) and on the flattened shapes of the tapped ones:

  TN  (what ships)   neraf_gemm_bf16_tn: operands voxel-major as the producers leave them, transposed by ds_read_b64_tr_b16
  NT  (the lead)     neraf_gemm_bf16 on K-MAJOR copies dY^T [cout][V], X^T [cin][V] (what the BatchNorm kernels would have to write
                     in addition), fp32 result through the same split-K path
  T   the cost of producing the two K-major copies, as a bf16 transpose of each operand (a fused producer write cannot be cheaper
      than the bytes: listed separately)

Each form replayed in a graph-free loop of 20 launches with rotating buffers; microseconds per launch and TFLOP/s."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neraf_amd import _lib
lib = _lib.load(); h = _lib.ctx(0)
dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)      # noqa: E731
# (cout, N = taps * cin as the kernel sees it, V voxels): 1x1x1 convolutions of layers 1-3 and the tapped ones flattened
shapes = [("layer1 conv3 1x1x1", 256, 64, 32768), ("layer1 conv1 1x1x1", 64, 256, 32768), ("layer2 conv3 1x1x1", 512, 128, 4096),
          ("layer3 conv3 1x1x1", 1024, 256, 512), ("layer3 conv1 1x1x1", 256, 1024, 512),
          ("layer1 conv2 3x3x3 (flattened, no shift)", 64, 1728, 32768), ("layer2 conv2 3x3x3 (flattened)", 128, 3456, 4096),
          ("layer3 conv2 3x3x3 (flattened)", 256, 6912, 512)]
ws = torch.empty(192 << 20, dtype=torch.uint8, device=dev)
NB = 4


def timed(fn, reps=20):
    fn(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


tot = {"tn": 0.0, "nt": 0.0, "tr": 0.0}
for name, M, N, V in shapes:
    Np = (N + 127) // 128 * 128
    Mp = (M + 127) // 128 * 128
    dy = [(torch.rand(V, M, device=dev) - 0.5).bfloat16() for _ in range(NB)]
    x = [(torch.rand(V, N, device=dev) - 0.5).bfloat16() for _ in range(NB)]
    dyT = [torch.zeros(Mp, V, dtype=torch.bfloat16, device=dev) for _ in range(NB)]
    xT = [torch.zeros(Np, V, dtype=torch.bfloat16, device=dev) for _ in range(NB)]
    for i in range(NB):
        dyT[i][:M] = dy[i].t(); xT[i][:N] = x[i].t()
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    out2 = torch.empty(M, N, dtype=torch.float32, device=dev)

    def tn(i):
        _lib.check(lib.neraf_gemm_bf16_tn(h, dy[i % NB].data_ptr(), x[i % NB].data_ptr(), M, N, V, out.data_ptr(), ws.data_ptr(), ws.numel(), st()))

    def nt(i):
        _lib.check(lib.neraf_gemm_bf16(h, dyT[i % NB].data_ptr(), V, xT[i % NB].data_ptr(), V, M, N, V, Mp, Np, 1.0, None, 0, None, 0, None, 0,
                                       out2.data_ptr(), N, st()))

    def tr(i):
        dyT[i % NB][:M].copy_(dy[i % NB].t()); xT[i % NB][:N].copy_(x[i % NB].t())
    ok = (M % 64 == 0 and N % 64 == 0 and V % 64 == 0)
    t_tn = timed(tn) if ok else float("nan")
    t_nt = timed(nt)
    t_tr = timed(tr)
    tn(0); nt(0); torch.cuda.synchronize()
    rel = float((out - out2).norm() / out.norm()) if ok else float("nan")
    fl = 2.0 * M * N * V
    tot["tn"] += t_tn; tot["nt"] += t_nt; tot["tr"] += t_tr
    print(f"{name:44s} M{M:5d} N{N:5d} V{V:6d}: TN {t_tn:7.1f} us ({fl/t_tn/1e6:6.0f} TF)   NT on K-major {t_nt:7.1f} us ({fl/t_nt/1e6:6.0f} TF)"
          f"   + transposes {t_tr:6.1f} us ({(M+N)*V*4/1e6:6.1f} MB moved)   rel {rel:.1e}", flush=True)
print(f"sum: TN {tot['tn']:.1f} us   NT {tot['nt']:.1f} us   transposes {tot['tr']:.1f} us")

#!/usr/bin/env python3
"""NAcF alone (A1 + A5 + A6 forward, backward with all weight gradients and d feature) at a given number of RIR slices: wall time per
forward + backward and the library's per-family HIP-event records -- the GEMM families' achieved fraction of the dense MFMA peak at
M = 2048 (configs[2]) and M = 6464 (configs[3] global).      python tools/nacf_bench.py [B ...] [--ss]
Environment knobs are read by the library at first use (one process per variant): NERAF_NACF_MALIGN, NERAF_GEMM_WIDE, ..."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from neraf_amd import _lib, synth
from neraf_amd.field import NeRAFAudioSoundField
from neraf_amd.losses import STFTLoss

lib = _lib.load(); h = _lib.ctx(0)
dev = torch.device("cuda:0")
ss = "--ss" in sys.argv
Bs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [2048, 6464]
C_, F_, T_ = (2, 257, 101) if ss else (1, 513, 60)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
field = NeRAFAudioSoundField(1187, 512, sound_rez=C_, N_frequencies=F_)
field.load_state_dict({k: T(v) for k, v in synth.nacf_state_dict(1187, 512, C_, F_).items()})
field.to(dev)
aabb = T(synth.audio_aabb())
crit = STFTLoss("mse")
flop_slice = 2 * (1187 * 5096 + 5096 * 2048 + 2048 * 1024 + 1024 * 1024 + 1024 * 512 + C_ * 512 * F_)
flop_exec = 2 * (163 * 5096 + 5096 * 2048 + 2048 * 1024 + 1024 * 1024 + 1024 * 512 + C_ * 512 * F_)
print(f"head {C_}x{F_}, env " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("NERAF_")))
for B in Bs:
    b = {k: T(v).to(dev) for k, v in synth.audio_batch(B, C_, F_, T_, tag="nacf.bench").items()}
    feat = T(synth.uniform("nacf.bench.feat", (1024,), 0.0, 2.0)).to(dev).requires_grad_(True)

    def step():
        for p in field.parameters():
            p.grad = None
        feat.grad = None
        y = field.forward_queries(feat, b["time_query"], b["mic_pose"], b["source_pose"], b["rot"], aabb, T_)
        l = crit(y, b["data"])
        (l["audio_sc_loss"] * 1e-4 + l["audio_mag_loss"] * 1e-3).backward()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    lib.neraf_prof_enable(h, 1)
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    rows, kid, tot_ms, tot_w = [], 0, 0.0, 0.0
    while lib.neraf_prof_kernel_name(kid):
        m_, n_, w_, e_ = C.c_double(), C.c_int(), C.c_double(), C.c_double()
        _lib.check(lib.neraf_prof_summary_ex(h, kid, C.byref(m_), C.byref(n_), C.byref(w_), C.byref(e_)), 0)
        if n_.value:
            rows.append((lib.neraf_prof_kernel_name(kid).decode(), m_.value / 8, n_.value / 8, w_.value / 8, e_.value / 8))
            if "gemm" in rows[-1][0] or "wgrad" in rows[-1][0]:
                tot_ms += m_.value / 8; tot_w += w_.value / 8
        kid += 1
    lib.neraf_prof_enable(h, 0)
    print(f"B = {B}: {ms:.3f} ms per forward + backward (wall, host-paced);  dense-equivalent 3 x {flop_slice / 1e6:.2f} MFLOP/slice = "
          f"{3 * flop_slice * B / 1e9:.1f} GF -> {3 * flop_slice * B / ms / 1e9:.0f} TF/s = {3 * flop_slice * B / ms / 1e9 / 2500:.3f} of peak (wall)")
    for name, m_, n_, w_, e_ in sorted(rows, key=lambda r: -r[1]):
        print(f"    {name[:60]:60s} {m_ * 1e3:8.1f} us {n_:5.1f} launches  {w_ / 1e9:8.2f} GF  {w_ / m_ / 1e9 / 2500 if m_ else 0:.3f} of peak")
    print(f"    GEMM families together: {tot_ms * 1e3:.1f} us, {tot_w / 1e9:.1f} GF executed -> {tot_w / tot_ms / 1e9:.0f} TF/s = {tot_w / tot_ms / 1e9 / 2500:.3f} of the dense peak; "
          f"dense-equivalent {3 * flop_slice * B / 1e9:.1f} GF -> {3 * flop_slice * B / tot_ms / 1e9 / 2500:.3f}")

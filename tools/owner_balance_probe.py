#!/usr/bin/env python3
"""Hits per owner workgroup of the hash-gradient scatter: after one field backward of a bench-shaped ray batch, read the slice bit planes the
pre-pass left in the dump buffer (csrc/field_bwd.hip FieldOwnerArgs::planes) and count the set bits of every (level, slice) plane -- the
number of samples the owner of that slice expands.  The owner kernel ends when its slowest workgroup does.   python tools/owner_balance_probe.py [rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

R = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device("cuda:0")
js = bench.JointStep(dev, R, 256, 1, rotate=1)
for _ in range(3):
    js.step()
torch.cuda.synchronize()
f = js.vm.field.module
S = 48
npad = (R * S + 63) // 64 * 64
dump = f.dump_buffer(R, S, dev)
half = 2
off = (4 * 128 * npad + 64 * npad) * half + 3 * npad * 4          # slot 4, rows 64..: pos [3][npad] fp32, then the planes
wpp = npad // 64
planes = dump[off:off + 16 * 32 * wpp * 8].view(torch.int64).reshape(16, 32, wpp)
# popcount of int64 words
x = planes.view(torch.uint8)
lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int32, device=dev)
cnt = lut[x.long()].reshape(16, 32, -1).sum(-1).cpu().numpy()
N = R * S
print(f"{R} rays = {N} samples; hits per (level, slice) plane:")
for l in range(16):
    c = cnt[l]
    nz = c[c > 0]
    print(f"  level {l:2d}: slices used {len(nz):2d}  mean {nz.mean():9.0f}  max {nz.max():9d}  max/mean {nz.max() / nz.mean():5.2f}  total/N {c.sum() / N:5.2f}")
allnz = cnt[cnt > 0]
print(f"all planes: {len(allnz)} workgroups' worth, mean {allnz.mean():.0f}, max {allnz.max()}, max/mean {allnz.max() / allnz.mean():.2f}; "
      f"sum {allnz.sum()} = {allnz.sum() / N:.1f} x N")

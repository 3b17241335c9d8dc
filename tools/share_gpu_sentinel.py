#!/usr/bin/env python3
"""GPU-sharing bisect: this process fills --mib MiB of device memory with a pattern and then only RE-READS it (a torch compare
kernel) every 50 ms.  A mismatch that is there on the next read as well = something wrote into this process' memory; one that is
gone on the next read = the read itself returned foreign data."""
import argparse, time
ap = argparse.ArgumentParser()
ap.add_argument("--mib", type=int, default=1024); ap.add_argument("--seconds", type=float, default=30.0)
a = ap.parse_args()
import torch
dev = torch.device("cuda:0")
n = (a.mib << 20) // 4
x = torch.full((n,), 0x5A5A5A5A, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
t0 = time.time(); reads = bad_reads = 0; sticky = 0; last = None
while time.time() - t0 < a.seconds:
    wrong = (x != 0x5A5A5A5A)
    k = int(wrong.sum()); reads += 1
    if k:
        bad_reads += 1
        idx = wrong.nonzero().flatten()
        pages = torch.unique(idx // 1024)
        again = int((x != 0x5A5A5A5A).sum())
        sticky += again > 0
        if bad_reads <= 5:
            print(f"read {reads}: {k} wrong words in {len(pages)} 4-KiB pages (first pages {pages[:6].tolist()}, sample value {int(x[idx[0]]):#x}); "
                  f"re-read: {again} wrong", flush=True)
        if again:
            x.fill_(0x5A5A5A5A)
    time.sleep(0.01)
print(f"sentinel {a.mib} MiB: {bad_reads} of {reads} reads saw foreign data, {sticky} of them still there on the re-read", flush=True)

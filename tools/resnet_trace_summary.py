#!/usr/bin/env python3
"""Fold a rocprofv3 kernel trace of tools/resnet_trace.py by position in the per-iteration launch sequence: prints every launch with
its mean duration and the gap to its predecessor, and totals per phase."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n_iter = int(sys.argv[2])
names = [r["Kernel_Name"] for r in rows]
# sequence length: launches per iteration in steady state (take the last n_iter-2 iterations)
per = None
for L in range(100, 700):
    tail = names[-L * 3:]
    if len(tail) == 3 * L and tail[:L] == tail[L:2 * L] == tail[2 * L:]:
        per = L
        break
print("launches per iteration:", per)
its = 4
seq = rows[-per * its:]
dur = collections.defaultdict(float); gap = collections.defaultdict(float)
for k in range(its):
    for i in range(per):
        r = seq[k * per + i]
        dur[i] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / its
        if i > 0:
            gap[i] += (int(r["Start_Timestamp"]) - int(seq[k * per + i - 1]["End_Timestamp"])) / its
tot_d = sum(dur.values()) / 1e3; tot_g = sum(gap.values()) / 1e3
print(f"sum of kernel durations {tot_d:.1f} us, sum of gaps {tot_g:.1f} us, iteration {(int(seq[per-1]['End_Timestamp'])-int(seq[0]['Start_Timestamp']))/1e3:.1f} us")
for i in range(per):
    nm = seq[i]["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{i:4d} {dur[i]/1e3:8.2f} us  gap {gap[i]/1e3:6.2f}  {nm}")

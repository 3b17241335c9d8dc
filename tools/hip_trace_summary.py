#!/usr/bin/env python3
"""Summary of a rocprofv3 --hip-runtime-trace of bench.py: per-step host time of the HIP calls inside the timed region, the
blocking ones listed (tools/gpu_hip_trace.sh)."""
import collections, csv, glob, os, sys
f = sorted(glob.glob("gpurun_out/hiptrace/**/*hip_api_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
syncs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Function"] == "hipDeviceSynchronize"]
a, b = syncs[0][1], syncs[1][0]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
print(f"timed region: host loop {(b-a)/1e6/steps:.3f} ms/step + drain {(syncs[1][1]-syncs[1][0])/1e6/steps:.3f} ms/step")
agg = collections.defaultdict(lambda: [0, 0, 0])
for r in rows:
    s = int(r["Start_Timestamp"])
    if a <= s < b:
        d = int(r["End_Timestamp"]) - s
        x = agg[r["Function"]]; x[0] += 1; x[1] += d; x[2] = max(x[2], d)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  {k:24s} {v[0]/steps:7.1f}/step {v[1]/steps/1e3:8.1f} us/step  max {v[2]/1e3:9.1f} us")
for k in ("hipMalloc", "hipFree", "hipHostMalloc", "hipStreamSynchronize"):
    if k in agg: print(f"  {k}: {agg[k][0]} calls in the timed region")

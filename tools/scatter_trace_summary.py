#!/usr/bin/env python3
"""Per-call durations of the hash-grid gradient kernels in the newest rocprofv3 kernel trace under gpurun_out/prof."""
import collections, csv, glob, os
f = sorted(glob.glob('gpurun_out/prof/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)[-1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'field_scatter' in n or 'field_slice' in n or 'field_unpack' in n:
        d[n.split('::')[1].split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    print(f"{k:32s} {len(v):4d} calls, last: {[round(x) for x in v[-8:]]}")

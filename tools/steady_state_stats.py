#!/usr/bin/env python3
"""Per-kernel statistics of the STEADY-STATE steps only, from a rocprofv3 kernel trace of `bench.py --plain`: the process-wide
`*_kernel_stats.csv` also counts set-up (weight initialisation, priming, graph capture), whose torch fills / copies are not part of a
step.  The timed steps are the last thing the --plain process runs, so the window is the last `steps` x (their mean length) of GPU
activity; the window is found from the dispatches of one once-per-step kernel.      steady_state_stats.py <trace.csv> <steps> <out.csv>"""
import collections, csv, sys

trace, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = []
for r in csv.DictReader(open(trace)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marker = "wgrad_wide_tn_kernel"                      # exactly one launch per training step
starts = [s for s, _, n in rows if marker in n]
if len(starts) < steps + 1:
    sys.exit(f"only {len(starts)} steps in the trace")
# a step runs from just after the previous step's optimizer to its own: cut at the start of the first kernel after the
# (steps+1)-th last marker's step ended is not observable, so the window is [start of the marker `steps` steps before the last, last
# marker start): exactly `steps` step-lengths of consecutive kernels
t0, t1 = starts[-steps - 1], starts[-1]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    if t0 <= s < t1:
        agg[n][0] += 1; agg[n][1] += e - s
tot = sum(v[1] for v in agg.values())
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "CallsPerStep", "TotalDurationNs", "AverageNs", "NsPerStep", "Percentage"])
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n, c, round(c / steps, 3), t, round(t / c, 1), round(t / steps, 1), round(100.0 * t / tot, 3)])
torch_like = sum(c for n, (c, _) in agg.items() if "at::native" in n or "rocclr" in n or "elementwise_kernel" in n)
print(f"{steps} steady-state steps, window {(t1 - t0) / steps / 1e6:.3f} ms per step, kernel time {tot / steps / 1e6:.3f} ms per step, "
      f"{sum(c for c, _ in agg.values()) / steps:.1f} launches per step, torch / rocclr launches in the window: {torch_like}")

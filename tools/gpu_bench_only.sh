#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== bench" ; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
for k in d['roofline']['all_kernel_families']:
    print(f\"  {k['kernel'][:50]:50s} {k['launches_per_step']:6.1f}/step avg {k['avg_us']:8.1f} us  {k['ms_per_step']:6.3f} ms/step  {k['achieved']:8.1f} {k['unit']}\")
"
echo "== rocprof" ; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | sort | tail -1); echo "stats: $f"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = max(int(r["Calls"]) for r in rows if "grid_to_ndhwc8" in r["Name"])    # one launch per step (priming + warm-up + timed + replay)
print(f"total GPU ms per step ({steps} steps):", tot / steps / 1e6)
for r in rows[:28]:
    print(f'{r["Name"][:90]:90s} calls/step {int(r["Calls"])/steps:6.1f} avg {float(r["AverageNs"])/1e3:8.1f} us  {float(r["TotalDurationNs"])/steps/1e6:6.3f} ms/step')
PY

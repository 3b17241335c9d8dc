#!/bin/bash
# rocprofv3 kernel stats of the eval-render bench (bench.py --mode eval --plain): per-kernel us per STEP (frame + RIRs)
#   tools/gpu_eval_trace.sh [tag] [steps] [rirs]
TAG=${1:-r04_eval}
STEPS=${2:-10}
RIRS=${3:-32}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out $R/profiles
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ks && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $R/bench.py --mode eval --steps $STEPS --warmup 2 --rirs $RIRS --plain > /tmp/ks.log 2>&1
echo "rc=$?"; tail -2 /tmp/ks.log
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
head -101 $f > $R/profiles/${TAG}_kernel_stats.csv
grep -h '"metric"' /tmp/ks.log | tail -1 > $R/profiles/${TAG}_bench_under_rocprof.json
python3 - "$f" $((STEPS + 4)) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step: {tot/n/1e3:.1f} us over {n} steps")
for r in rows[:45]:
    print(f"{float(r['TotalDurationNs'])/n/1e3:9.1f} us/step {int(r['Calls'])/n:7.1f} calls/step avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Name'].replace('(anonymous namespace)::','')[:110]}")
PY
mkdir -p $R/gpurun_out/profiles_out && cp $R/profiles/${TAG}_* $R/gpurun_out/profiles_out/

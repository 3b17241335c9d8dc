#!/bin/bash
# rocprofv3 kernel stats of a short bench run, rows matching a regex:  tools/gpu_kernel_stats.sh 'pack|cvt' [steps]
PAT=${1:-.}
STEPS=${2:-12}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ks && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $R/bench.py --steps $STEPS --warmup 3 --plain > /tmp/ks.log 2>&1
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$PAT" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
pat = re.compile(sys.argv[2])
for r in rows:
    if pat.search(r["Name"]):
        print(f"{float(r['TotalDurationNs'])/tot*100:5.2f}% calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Name'].replace('(anonymous namespace)::','')[:110]}")
PY

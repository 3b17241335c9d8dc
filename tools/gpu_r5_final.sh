#!/bin/bash
# round 5, evidence session: full GPU suite, the default bench line, then the profile sets (training + eval) -> profiles/<tag>_*
TAG=${1:-r05_b}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15
echo "== bench default"; timeout 1200 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo rc=$?
python - $TAG <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/{sys.argv[1]}_bench_default.json").read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'dtype', d['dtype'], 'windows', d['repeat_windows']['ms_per_step'])
print('roofline', {k: d['roofline'].get(k) for k in ('kernel', 'bound', 'achieved', 'frac', 'traffic')})
e = d.get('eval_render', {})
print('eval ms/frame', e.get('ms_per_frame'), {k: (e.get('roofline') or {}).get(k) for k in ('kernel', 'bound', 'achieved', 'frac', 'traffic')})
print('parity', json.dumps(d.get('parity'))[:1500])
print('cpu', {k: d.get('cpu_baseline', {}).get(k) for k in ('value', 'cores')})
PY
bash tools/gpu_profile.sh $TAG 2>&1 | tail -30
bash tools/gpu_profile.sh $TAG eval 2>&1 | tail -12
cp gpurun_out/${TAG}_bench_default.json gpurun_out/profiles_out/ 2>/dev/null

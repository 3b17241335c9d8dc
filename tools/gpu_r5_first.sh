#!/bin/bash
# round 5, first call: GPU parity tests of the unchanged kernels + the default bench line (eval roofline / cpu_baseline now inside it)
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
echo "== bench default"; timeout 900 python bench.py --no-parity > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err; echo rc=$?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_bench_default.json").read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'dtype', d['dtype'], 'windows', d['repeat_windows']['ms_per_step'])
e = d.get('eval_render', {})
print('eval ms/frame', e.get('ms_per_frame'), 'roofline', {k: e.get('roofline', {}).get(k) for k in ('kernel', 'bound', 'achieved', 'frac', 'traffic')})
for k in (e.get('roofline') or {}).get('all_kernel_families', []):
    print(f"  {k['kernel'][:60]:60s} {k['bound']:5s} avg {k['avg_us']:8.1f} us frac {k['frac']:.3f}")
print('eval cpu', {k: e.get('cpu_baseline', {}).get(k) for k in ('value', 'cores')})
PY

#!/bin/bash
# GPU parity tests, then the bench with its per-kernel breakdown (no rocprof)
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
echo "== bench" ; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_line.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_line.json").read())
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
for k in d['roofline']['all_kernel_families']:
    print(f"  {k['kernel'][:50]:50s} {k['launches_per_step']:6.1f}/step avg {k['avg_us']:8.1f} us  {k['ms_per_step']:6.3f} ms/step  {k['achieved']:8.1f} {k['unit']}")
PY

#!/bin/bash
# owner-scatter batch change: parity (owner vs global atomics, bit for bit) + the cfg3 / cfg2 steps
export TMPDIR=/tmp
python -m pytest tests/test_gpu_properties.py tests/test_gpu_vision_train.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_autograd_audit.py -q -m gpu -x 2>&1 | tail -4
for args in "--dataset soundspaces --rays 32768 --slices 6464 --rotate 4 --steps 10" "--steps 30"; do
  python bench.py $args --warmup 3 --parity off --no-eval-line --no-cpu-baseline --detail gpurun_out/r06_owner_detail.json > gpurun_out/r06_owner_line.json 2>/dev/null
  python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_owner_detail.json'))
print(d['config']['rays_per_gpu'], d['config']['slices_per_gpu'], 'ms_per_step', round(d['ms_per_step'],4), d['repeat_windows']['ms_per_step'])
for k in sorted(d['roofline']['all_kernel_families'], key=lambda k:-k['ms_per_step']):
    if k['bound']!='mfma': print(f"   {k['kernel'][:70]:70s} {k['ms_per_step']*1e3:8.1f} us {k['launches_per_step']:5.1f}x{k['avg_us']:8.1f}  frac {k['frac']:.3f}")
PY
done

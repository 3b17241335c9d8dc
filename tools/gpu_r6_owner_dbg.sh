#!/bin/bash
# what the owner scatter spends its time on (measurement only, results garbage for dbg != 0): 0 = full, 1 = scan + queue pushes without expanding hits, 2 = loads + loop only
for dbg in 0 1 2; do for args in "--dataset soundspaces --rays 32768 --slices 6464 --rotate 4 --steps 8" "--steps 20"; do
  NERAF_OWNER_DEBUG=$dbg python bench.py $args --warmup 3 --parity off --no-eval-line --no-cpu-baseline --repeats 2 --detail gpurun_out/own_dbg.json > /dev/null 2>&1
  python - $dbg <<'PY'
import json, sys
d=json.load(open('gpurun_out/own_dbg.json'))
f=[k for k in d['roofline']['all_kernel_families'] if k['kernel'].startswith('field_scatter')][0]
print(f"dbg={sys.argv[1]} rays {d['config']['rays_per_gpu']:6d}  scatter family {f['ms_per_step']*1e3:8.1f} us/step  step {d['ms_per_step']:.3f} ms")
PY
done; done

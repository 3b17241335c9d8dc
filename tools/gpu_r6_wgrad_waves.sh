#!/bin/bash
# wgrad_wide_tn: 4 waves per workgroup (round 5) vs 8 (round 6: wave pairs split each K-step), same LDS, same tiles -- same box, alternating
NERAF_WGRAD_WAVES=8 python -m pytest tests/test_gpu_resnet3d.py -q -m gpu -k "gate_matched or norms or stage_backward or chain" 2>&1 | tail -2
for round in 1 2 3; do for w in 4 8; do
  NERAF_WGRAD_WAVES=$w python bench.py --steps 20 --warmup 3 --parity off --no-eval-line --no-cpu-baseline --repeats 3 --detail gpurun_out/wg_detail.json > /dev/null 2>&1
  python - $w <<'PY'
import json, sys
d = json.load(open('gpurun_out/wg_detail.json'))
f = [k for k in d['roofline']['all_kernel_families'] if k['kernel'].startswith('wgrad_wide')][0]
print(f"waves={sys.argv[1]}  ms_per_step {d['ms_per_step']:.4f}  wgrad_wide_tn {f['avg_us']:.1f} us  frac {f['frac']:.3f}  resnet fwd+bwd {d['replicated_per_rank']['resnet3d_fwd_bwd_ms']:.4f}")
PY
done; done

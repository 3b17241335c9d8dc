#!/bin/bash
# ResNet3D weight-gradient kernel (wgrad_wide_tn): LDS ring depth 2 (shipped: two workgroups per CU, one stage in flight each), 3, 4 (one
# workgroup per CU, two / three stages in flight) -- same box, alternating; gradients checked by tests/test_gpu_resnet3d.py
for round in 1 2; do for nst in 2 3 4; do
  NERAF_WGRAD_WIDE_NST=$nst python bench.py --steps 20 --warmup 3 --parity off --no-eval-line --no-cpu-baseline --repeats 3 --detail gpurun_out/wg_detail.json > /dev/null 2>&1
  python - $nst <<'PY'
import json, sys
d = json.load(open('gpurun_out/wg_detail.json'))
f = [k for k in d['roofline']['all_kernel_families'] if k['kernel'].startswith('wgrad_wide')][0]
print(f"NST={sys.argv[1]}  ms_per_step {d['ms_per_step']:.4f}  wgrad_wide_tn {f['avg_us']:.1f} us  frac {f['frac']:.3f}  resnet fwd+bwd {d['replicated_per_rank']['resnet3d_fwd_bwd_ms']:.4f}")
PY
done; done
NERAF_WGRAD_WIDE_NST=4 python -m pytest tests/test_gpu_resnet3d.py -q -m gpu -k "gate_matched or norms or stage_backward" 2>&1 | tail -2

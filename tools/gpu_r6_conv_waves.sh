#!/bin/bash
# implicit-GEMM convolutions on 64x64 tiles: 4 waves per workgroup (round 5) vs 8 (wave pairs split each K-step) -- same box, alternating
NERAF_CONV_WAVES=8 python -m pytest tests/test_gpu_resnet3d.py -q -m gpu 2>&1 | tail -2
for round in 1 2 3; do for w in 4 8; do
  NERAF_CONV_WAVES=$w python bench.py --steps 20 --warmup 3 --parity off --no-eval-line --no-cpu-baseline --repeats 3 --detail gpurun_out/cw_detail.json > /dev/null 2>&1
  python - $w <<'PY'
import json, sys
d = json.load(open('gpurun_out/cw_detail.json'))
f = {k['kernel']: k for k in d['roofline']['all_kernel_families']}
c = f.get('gemm_f16_nt_pipe_kernel<64, 64, 4, 1, *, false>')
print(f"conv waves={sys.argv[1]}  ms_per_step {d['ms_per_step']:.4f}  conv family {c['ms_per_step']*1e3:.1f} us ({c['launches_per_step']:.0f} x {c['avg_us']:.1f})  resnet fwd+bwd {d['replicated_per_rank']['resnet3d_fwd_bwd_ms']:.4f}")
PY
done; done

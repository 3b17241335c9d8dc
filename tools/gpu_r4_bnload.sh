#!/bin/bash
# BatchNorm kernels with their loads hoisted above the statistics barriers: ResNet3D parity tests, then the training line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_resnet3d.py tests/test_gpu_model.py -q 2>&1 | tail -3
for i in 1 2; do
timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-eval-line > gpurun_out/train_line.json 2> gpurun_out/train_line.err; echo rc=$?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/train_line.json").read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'repeat', d['repeat_windows']['ms_per_step'], 'resnet3d fwd+bwd', d['replicated_per_rank'].get('resnet3d_fwd_bwd_ms'))
PY
done

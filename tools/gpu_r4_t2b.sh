#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_optim.py tests/test_gpu_pipeline.py tests/test_gpu_dp2.py tests/test_gpu_eval_loop.py -x -q 2>&1 | tail -5
echo ==== TORCH GLUE; timeout 600 python tools/torch_glue.py 2>&1 | grep "ATEN\|aten launches" | tail -10
echo ==== bench; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-eval-line > gpurun_out/train_line.json 2> gpurun_out/train_line.err; echo rc=$?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/train_line.json").read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'repeat', d['repeat_windows']['ms_per_step'], 'fixed/rot', d['batches']['fixed_batch_ms_per_step'], d['batches']['rotating_ms_per_step'])
PY

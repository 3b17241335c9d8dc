#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_vision.py tests/test_gpu_vision_train.py tests/test_gpu_eval_bench.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_eval_loop.py -q 2>&1 | tail -4
echo "== eval line"; timeout 900 python bench.py --mode eval --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/eval_line.json 2> gpurun_out/eval_line.err; echo rc=$?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/eval_line.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "rays_per_s", "ms_per_frame", "bins_per_s", "us_per_rir")}, d["batched_rirs"]["us_per_rir"])
for k in d["roofline"]["all_kernel_families"]:
    print(f"  {k['kernel'][:50]:50s} {k['launches_per_step']:6.1f}/step avg {k['avg_us']:8.1f} us  {k['ms_per_step']:6.3f} ms/step  {k['achieved']:8.1f} {k['unit']}")
PY
echo "== train line"; timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-eval-line > gpurun_out/train_line.json 2> gpurun_out/train_line.err; echo rc=$?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/train_line.json").read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'repeat', d['repeat_windows']['ms_per_step'])
for k in d['roofline']['all_kernel_families']:
    if 'field' in k['kernel'] or 'proposal' in k['kernel']:
        print(f"  {k['kernel'][:50]:50s} {k['launches_per_step']:6.1f}/step avg {k['avg_us']:8.1f} us  {k['ms_per_step']:6.3f} ms/step  {k['achieved']:8.1f} {k['unit']}")
PY

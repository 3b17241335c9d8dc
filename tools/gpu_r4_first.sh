#!/bin/bash
# round 4, first GPU visit: the new eval-bench tests, the eval line, the training line with rotating batches
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest eval bench"; timeout 900 python -m pytest tests/test_gpu_eval_bench.py -x -q 2>&1 | tail -8
echo "== eval line"; timeout 900 python bench.py --mode eval --steps 10 --warmup 2 > gpurun_out/eval_line.json 2> gpurun_out/eval_line.err; echo rc=$?
tail -3 gpurun_out/eval_line.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/eval_line.json").read().strip().splitlines()[-1])
    for k in ("value", "ms_per_step", "rays_per_s", "fps", "ms_per_frame", "bins_per_s", "fps_audio", "us_per_rir", "batched_rirs"):
        print(k, d.get(k))
    for k in d["roofline"]["all_kernel_families"]:
        print(f"  {k['kernel'][:50]:50s} {k['launches_per_step']:6.1f}/step avg {k['avg_us']:8.1f} us  {k['ms_per_step']:6.3f} ms/step  {k['achieved']:8.1f} {k['unit']}")
    print(json.dumps(d.get("cpu_baseline"))[:600])
except Exception as e:
    print("eval line unreadable", e)
PY
echo "== train line"; timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/train_line.json 2> gpurun_out/train_line.err; echo rc=$?
tail -3 gpurun_out/train_line.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/train_line.json").read().strip().splitlines()[-1])
    print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'repeat', d['repeat_windows']['ms_per_step'])
    print('batches', d['batches'])
    print('eval_render', json.dumps(d.get('eval_render'))[:900])
    print('cpu', json.dumps(d.get('cpu_baseline'))[:900])
    for k in d['roofline']['all_kernel_families']:
        print(f"  {k['kernel'][:50]:50s} {k['launches_per_step']:6.1f}/step avg {k['avg_us']:8.1f} us  {k['ms_per_step']:6.3f} ms/step  {k['achieved']:8.1f} {k['unit']}")
except Exception as e:
    print("train line unreadable", e)
PY

#!/bin/bash
# full GPU test suite, then the eval line and the training line (short forms)
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
echo "== eval line"; timeout 900 python bench.py --mode eval --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/eval_line.json 2> gpurun_out/eval_line.err; echo rc=$?
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/eval_line.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "rays_per_s", "ms_per_frame", "bins_per_s", "us_per_rir")}, d["batched_rirs"]["us_per_rir"])
    for k in d["roofline"]["all_kernel_families"]:
        print(f"  {k['kernel'][:50]:50s} {k['launches_per_step']:6.1f}/step avg {k['avg_us']:8.1f} us  {k['ms_per_step']:6.3f} ms/step  {k['achieved']:8.1f} {k['unit']}")
except Exception as e:
    print("eval line unreadable", e)
PY
echo "== train line"; timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/train_line.json 2> gpurun_out/train_line.err; echo rc=$?
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/train_line.json").read().strip().splitlines()[-1])
    print('ms_per_step', d['ms_per_step'], 'repeat', d['repeat_windows']['ms_per_step'], 'fixed/rot', d['batches']['fixed_batch_ms_per_step'], d['batches']['rotating_ms_per_step'])
    print('eval_render', {k: d['eval_render'].get(k) for k in ("ms_per_frame", "us_per_rir", "rays_per_s", "bins_per_s")})
except Exception as e:
    print("train line unreadable", e)
PY

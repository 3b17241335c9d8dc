#!/bin/bash
# A/B of the ray-tile orders of the two eval gather kernels (NERAF_RAY_TILES bit 0: proposal density, bit 1: field query)
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_eval_bench.py -x -q 2>&1 | tail -3
for v in 0 1 2 3 0 3; do
  NERAF_RAY_TILES=$v timeout 600 python bench.py --mode eval --steps 10 --warmup 2 --no-cpu-baseline > /tmp/e.json 2>/tmp/e.err || tail -3 /tmp/e.err
  python - $v <<'PY'
import json, sys
d = json.loads(open("/tmp/e.json").read().strip().splitlines()[-1])
f = {k["kernel"][:22]: k for k in d["roofline"]["all_kernel_families"]}
print("RAY_TILES", sys.argv[1], "ms/frame %.2f" % d["ms_per_frame"], "us/rir %.1f" % d["us_per_rir"],
      "| prop avg %.1f us" % f["proposal_density_kerne"]["avg_us"], "field avg %.1f us" % f["field_query_kernel"]["avg_us"])
PY
done

#!/bin/bash
# pixel-tile vs row-segment lane order of the frame render: lane-order test, then the eval line under NERAF_PIXEL_TILES = 1 / 0
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vision.py tests/test_gpu_eval_bench.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_eval_loop.py -q 2>&1 | tail -4
for v in 1 0 1 0; do
  NERAF_PIXEL_TILES=$v timeout 600 python bench.py --mode eval --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/tiles_line.json 2> gpurun_out/tiles_line.err
  python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/tiles_line.json").read().strip().splitlines()[-1])
fam = {k["kernel"]: k for k in d["roofline"]["all_kernel_families"]}
print(f"NERAF_PIXEL_TILES={sys.argv[1]}  ms_per_frame {d['ms_per_frame']:.3f}  field {fam['field_query_kernel | field_query_frame_kernel']['avg_us']:.1f} us  proposal {fam['proposal_density_kernel | proposal_density_frame_kernel']['avg_us']:.1f} us")
PY
done

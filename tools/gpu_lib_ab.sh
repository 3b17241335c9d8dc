#!/bin/bash
# same-box A/B of the training line between the in-tree library and variant builds: tools/gpu_lib_ab.sh variants/libA.so variants/libB.so ...
# (alternating, two rounds; NERAF_HIP_LIB selects the library)
export TMPDIR=/tmp
mkdir -p gpurun_out
for round in 1 2; do
for v in "" "$@"; do
  if [ -n "$v" ]; then export NERAF_HIP_LIB=$PWD/$v; else unset NERAF_HIP_LIB; fi
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-eval-line > gpurun_out/ab_line.json 2> gpurun_out/ab_line.err
  python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_line.json").read().strip().splitlines()[-1])
w = d['repeat_windows']['ms_per_step']
print(f"{sys.argv[1] or 'in-tree':28s} ms_per_step {d['ms_per_step']:.4f}  median {sorted(w)[len(w)//2]:.4f}  windows {w}  resnet3d {d['replicated_per_rank'].get('resnet3d_fwd_bwd_ms'):.4f}")
PY
done
done

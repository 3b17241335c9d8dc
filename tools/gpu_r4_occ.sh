#!/bin/bash
# A/B of libraries built with another occupancy target of field_query_kernel (variants/libneraf_fq{2,4}.so; in-tree = 3): eval line
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "" variants/libneraf_fq2.so variants/libneraf_fq4.so "" variants/libneraf_fq4.so; do
  if [ -n "$v" ]; then export NERAF_HIP_LIB=$PWD/$v; else unset NERAF_HIP_LIB; fi
  timeout 600 python bench.py --mode eval --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/occ_line.json 2> gpurun_out/occ_line.err
  python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/occ_line.json").read().strip().splitlines()[-1])
fam = {k["kernel"]: k for k in d["roofline"]["all_kernel_families"]}
print(f"{sys.argv[1] or 'in-tree (3)':32s} ms_per_frame {d['ms_per_frame']:.3f}  field {fam['field_query_kernel']['avg_us']:.1f} us  proposal {fam['proposal_density_kernel']['avg_us']:.1f} us")
PY
done

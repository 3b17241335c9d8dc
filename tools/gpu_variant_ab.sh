#!/bin/bash
# A/B of environment-selected kernel variants on one box: bench.py per setting, kernel-family times from its HIP-event scopes.
#   tools/gpu_variant_ab.sh "" "NERAF_GEMM_NST128=2" "NERAF_GEMM_NST128=2 NERAF_GEMM_NST64=2"
export TMPDIR=/tmp
for v in "$@"; do
  echo "== variant: [$v]"
  env $v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
fam = {k['kernel'][:44]: k['ms_per_step'] for k in d['roofline']['all_kernel_families']}
print('  ms_per_step %.3f   ' % d['ms_per_step'] + '  '.join('%s=%.3f' % (k[24:44] if k.startswith('gemm_f16_nt_pipe') else k[:22], v) for k, v in fam.items()))
"
done

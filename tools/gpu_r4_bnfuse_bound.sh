#!/bin/bash
# upper bound of the BatchNorm-apply fusion: ResNet3D forward + backward alone with the candidate launches DROPPED
export TMPDIR=/tmp
for rep in 1 2 3; do for m in 0 1 2; do
  NERAF_SKIP_SMALL_BN=$m timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-eval-line 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  skip $m resnet3d fwd+bwd %.4f ms   step %.4f ms' % (d['replicated_per_rank']['resnet3d_fwd_bwd_ms'], d['ms_per_step']))"
done; done

#!/bin/bash
# NAcF narrow-layer weight gradients: 64x64 tiles un-split (round 5) vs 128x128 tiles K-split over the idle CUs (NERAF_SPLIT128_MIN_K = fewest K-steps)
for k in 0 32 64; do
  echo "=== NERAF_SPLIT128_MIN_K=$k"
  NERAF_SPLIT128_MIN_K=$k python tools/nacf_bench.py 2048 6464 2>&1 | grep -v amdgpu.ids
done
NERAF_SPLIT128_MIN_K=32 python -m pytest tests/test_gpu_nacf.py tests/test_gpu_fullsize.py -q -m gpu -k "nacf or joint_step" 2>&1 | tail -2

#!/bin/bash
# round 6, second call: new tests (fp16-chain recovery, configs[3] at global size) + the profile set of configs[3] global
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest"; timeout 1500 python -m pytest tests/test_gpu_resnet3d.py tests/test_gpu_fullsize.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -40
echo "== cfg3 global profile set"
BENCH_ARGS="--dataset soundspaces --rays 32768 --slices 6464 --rotate 4" STEPS=20 bash tools/gpu_profile.sh r06_cfg3_global 2>&1 | tail -70

#!/bin/bash
# Everything that showed the GPU-sharing damage of round 3 (profiles/r03_gpu_sharing_bisect.txt), on the current build:
# a ResNet3D probe next to the GEMM neighbour, paired ResNet3D probes, pairs of independent training runs, the two-rank tests repeated.
echo "== one ResNet3D probe (clean reference first) next to the 2048 x 1024 x 2048 GEMM neighbour"
LIBS=" " tools/share_gpu_ab_libs.sh
echo "== paired ResNet3D probes (train mode both)"
timeout 400 python tools/contention_resnet_probe.py --pair --iters 3000 2>&1 | grep -v Warning > /tmp/p.log; echo "  $(grep -c deviates /tmp/p.log) bad of 6000"; grep "feature deviation" /tmp/p.log
echo "== pairs of independent training runs"
timeout 900 tools/share_gpu_check.sh
echo "== two-rank tests, repeated"
for i in 1 2 3 4; do timeout 600 python -m pytest tests/test_gpu_dp2.py tests/test_gpu_trajectory.py -q -k "two_ranks or data_parallel" 2>&1 | tail -1; done

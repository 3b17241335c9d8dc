#!/bin/bash
# round 5: the fp16 gradient chain of the ResNet3D backward -- its tests, the superposition / gate-matched tests under the round-4
# (bf16 chain) library kept as variants/libneraf_r4.so, then a same-box A/B of the training line against it, with amax recorded on
# every pass (NERAF_CHAIN_AMAX_PERIOD=1) and never (NERAF_CHAIN_AMAX=0, measurement only).  Log -> gpurun_out/r5_fp16_chain_ab.txt
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_fp16_chain_ab.txt
{
echo "== resnet3d tests (in-tree, fp16 chain)"; timeout 1200 python -m pytest tests/test_gpu_resnet3d.py -q -s 2>&1 | grep -E "gate-matched|all-fp32|chain|linearity|stage|passed|failed|Error"
echo "== superposition + gate-matched under variants/libneraf_r4.so (bf16 chain)"
NERAF_HIP_LIB=$PWD/variants/libneraf_r4.so timeout 1200 python -m pytest tests/test_gpu_resnet3d.py -q -s -k "linearity or gate_matched" 2>&1 | grep -E "gate-matched|all-fp32|chain|linearity|passed|failed|Error"
echo "== A/B (default: amax every 4th pass)"; bash tools/gpu_lib_ab.sh variants/libneraf_r4.so 2>&1 | tail -4
echo "== amax every pass"; NERAF_CHAIN_AMAX_PERIOD=1 bash tools/gpu_lib_ab.sh 2>&1 | tail -2
echo "== amax never"; NERAF_CHAIN_AMAX=0 bash tools/gpu_lib_ab.sh 2>&1 | tail -2
echo "== trajectory"; timeout 1500 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
} 2>&1 | tee $L | tail -40

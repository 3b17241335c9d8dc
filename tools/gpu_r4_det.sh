#!/bin/bash
export TMPDIR=/tmp
echo "== RCCL world-1 test alone (verbose)"; timeout 900 python -m pytest tests/test_gpu_dp2.py -x -q -k rccl 2>&1 | tail -30
echo "== trajectory tests"; timeout 2400 python -m pytest tests/test_gpu_trajectory.py -x -q -s 2>&1 | grep -v "^$" | tail -40

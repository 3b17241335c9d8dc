#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" ; timeout 1200 python -m pytest tests -q -m gpu --timeout=900 -x 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log

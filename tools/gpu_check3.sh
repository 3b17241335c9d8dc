#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" ; timeout 1200 python -m pytest tests -q -m gpu --timeout=900 -x 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
echo "== stages"; timeout 600 python tools/stage_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stage_bench.log
echo "== rocprof stages"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_stage -- python3 $GRAFT_REPO_ROOT/tools/stage_bench.py > $GRAFT_REPO_ROOT/gpurun_out/rocprof_stage.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_stage -name "*kernel_stats.csv" | head -1); echo "stats: $f"; head -24 "$f" | cut -c1-170

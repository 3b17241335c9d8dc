#!/bin/bash
# HIP runtime API trace of the bench (which host calls block, and for how long)
mkdir -p gpurun_out/hiptrace
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --hip-runtime-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hiptrace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/hiptrace.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/hiptrace.log | cut -c1-200
f=$(find gpurun_out/hiptrace -name "*hip_api_stats.csv" | sort | tail -1); echo "$f"; head -25 "$f" | cut -c1-150

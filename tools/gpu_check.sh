#!/bin/bash
# One GPU-box session: tests, smoke, bench, rocprof.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" ; timeout 1200 python -m pytest tests -q -m gpu --timeout=900 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
echo "== stages"; timeout 600 python tools/stage_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stage_bench.log
echo "== bench" ; timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -2 | tee gpurun_out/bench.log
echo "== rocprof" ; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/rocprof.log | cut -c1-300
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | sort | tail -1); echo "stats: $f"; head -22 "$f" | cut -c1-160

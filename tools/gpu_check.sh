#!/bin/bash
# One GPU-box session: tests, smoke, bench, rocprof.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 1 3 4; do NERAF_GEMM_VARIANT=$v timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_bench_v$v.log; done
echo "== pytest -m gpu" ; timeout 900 python -m pytest tests -q -m gpu --timeout=600 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
echo "== pytest -m gpu (variant 1)" ; NERAF_GEMM_VARIANT=1 timeout 900 python -m pytest tests -q -m gpu --timeout=600 -k gemm 2>&1 | tail -5
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench" ; timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "== rocprof" ; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/rocprof.log
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); echo "stats: $f"; head -12 "$f" | cut -c1-200

#!/bin/bash
# HBM traffic counters for the bench step: two separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950),
# kernel trace only -- no runtime/sys tracing next to --pmc.
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc/$c.log 2>&1
  echo "$c rc=$?"
done
cd $GRAFT_REPO_ROOT
find gpurun_out/pmc -name "*.csv" | head; for f in $(find gpurun_out/pmc -name "*counter_collection.csv"); do echo $f; head -3 $f; wc -l $f; done

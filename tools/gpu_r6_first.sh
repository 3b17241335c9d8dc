#!/bin/bash
# round 6, first call: the new compact bench line (default run) + BASELINE configs[3] at its GLOBAL size on one GPU (32768 rays + 6464 slices)
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== bench default"; timeout 1200 python bench.py > gpurun_out/r06_a_bench_default.json 2> gpurun_out/r06_a_bench_default.err; echo rc=$?
tail -c 7000 gpurun_out/r06_a_bench_default.json; wc -c gpurun_out/r06_a_bench_default.json
tail -5 gpurun_out/r06_a_bench_default.err
echo "== cfg3 global (plain)"; timeout 900 python bench.py --dataset soundspaces --rays 32768 --slices 6464 --steps 10 --warmup 3 --plain --rotate 4 > gpurun_out/r06_a_cfg3_plain.json 2> gpurun_out/r06_a_cfg3_plain.err; echo rc=$?
cat gpurun_out/r06_a_cfg3_plain.json; tail -15 gpurun_out/r06_a_cfg3_plain.err
echo "== cfg3 global (full line, no parity / eval / cpu)"; timeout 900 python bench.py --dataset soundspaces --rays 32768 --slices 6464 --steps 10 --warmup 3 --rotate 4 --parity off --no-eval-line --no-cpu-baseline --detail gpurun_out/r06_a_cfg3_detail.json > gpurun_out/r06_a_cfg3_line.json 2> gpurun_out/r06_a_cfg3_line.err; echo rc=$?
cat gpurun_out/r06_a_cfg3_line.json; tail -5 gpurun_out/r06_a_cfg3_line.err

#!/bin/bash
# Final measurement set of round 6 on the GPU box: train + eval + configs[3]-global profiles, then the default bench line.
cd $GRAFT_REPO_ROOT
bash tools/gpu_profile.sh r06_d 2>&1 | tail -12
bash tools/gpu_profile.sh r06_d eval 2>&1 | tail -8
cd $GRAFT_REPO_ROOT
BENCH_ARGS="--dataset soundspaces --rays 32768 --slices 6464 --rotate 4" STEPS=20 bash tools/gpu_profile.sh r06_d_cfg3_global 2>&1 | tail -8
cd $GRAFT_REPO_ROOT
python3 bench.py --detail gpurun_out/profiles_out/r06_d_bench_detail.json > gpurun_out/profiles_out/r06_d_bench_default.json 2> gpurun_out/bench_default.err
wc -c gpurun_out/profiles_out/r06_d_bench_default.json

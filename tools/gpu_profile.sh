#!/bin/bash
# Full measurement set of one round for the bench command, on the GPU box:  tools/gpu_profile.sh <tag>
#   1. rocprofv3 --kernel-trace --stats (bench.py --plain: 8 + 3 + 40 = 51 steps in the process) -> profiles/<tag>_joint_step_kernel_stats.csv
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes)         -> profiles/<tag>_pmc_traffic.json (+ per-kernel CSVs)
#   3. --pmc SQ matrix-pipe / wait counters, --pmc SQ LDS + TCC -> profiles/<tag>_pmc_sq_by_kernel.csv
# Counter passes use --kernel-trace only (no runtime / sys tracing next to --pmc); the program after `--` is python3 itself.
#   tools/gpu_profile.sh <tag> eval   profiles `bench.py --mode eval --plain` instead (BASELINE configs[4]) -> profiles/<tag>_eval_*
#   BENCH_ARGS="--dataset soundspaces --rays 32768 --slices 6464 --rotate 4" tools/gpu_profile.sh r06_cfg3_global   profiles another shape
#   (profiles/<tag>_* as above; tools/pmc_summary.py prices the families by the same per-launch work the library's profiler reports)
TAG=${1:-r02_x}
MODE=${2:-train}
R=$GRAFT_REPO_ROOT
if [ "$MODE" = "eval" ]; then MARGS="--mode eval"; STEPS=10; WARM=2; PSTEPS=2; PWARM=1; NAME=eval; else MARGS=""; STEPS=${STEPS:-40}; WARM=3; PSTEPS=4; PWARM=2; NAME=joint_step; fi
MARGS="$MARGS $BENCH_ARGS"
mkdir -p $R/gpurun_out/prof $R/gpurun_out/pmc $R/profiles
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/$TAG -- python3 $R/bench.py $MARGS --steps $STEPS --warmup $WARM --plain > $R/gpurun_out/prof/$TAG.log 2>&1
echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$c -- python3 $R/bench.py $MARGS --steps $PSTEPS --warmup $PWARM --plain > $R/gpurun_out/pmc/$c.log 2>&1
  echo "$c rc=$?"
done
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc/SQ1 -- python3 $R/bench.py $MARGS --steps $PSTEPS --warmup $PWARM --plain > $R/gpurun_out/pmc/SQ1.log 2>&1
echo "SQ1 rc=$?"
timeout 900 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES TCC_HIT TCC_MISS --kernel-trace --output-format csv -d $R/gpurun_out/pmc/SQ2 -- python3 $R/bench.py $MARGS --steps $PSTEPS --warmup $PWARM --plain > $R/gpurun_out/pmc/SQ2.log 2>&1
echo "SQ2 rc=$?"
cd $R
f=$(find gpurun_out/prof/$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -121 $f > profiles/${TAG}_${NAME}_kernel_stats.csv && echo "stats -> profiles/${TAG}_${NAME}_kernel_stats.csv"
if [ "$MODE" != "eval" ]; then     # the same statistics over the timed (steady-state) steps alone
  t=$(find gpurun_out/prof/$TAG -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 tools/steady_state_stats.py $t $STEPS profiles/${TAG}_${NAME}_steady_state_kernel_stats.csv | tee profiles/${TAG}_${NAME}_steady_state_summary.txt
fi
tail -1 gpurun_out/prof/$TAG.log | head -c 400 > /dev/null
grep -h '"metric"' gpurun_out/prof/$TAG.log | tail -1 > profiles/${TAG}_${NAME}_bench_under_rocprof.json
if [ "$MODE" = "eval" ]; then PT=${TAG}_eval; else PT=$TAG; fi
python3 tools/pmc_summary.py $PT 2>&1 | tail -20
python3 tools/pmc_sq_summary.py $PT 2>&1 | tail -40
# the raw traces are hundreds of MB: keep the summaries only (gpurun copies back at most 64 MiB)
rm -rf gpurun_out/prof gpurun_out/pmc
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_* gpurun_out/profiles_out/

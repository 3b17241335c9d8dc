#!/bin/bash
# Alternating A/B of bench.py --plain under different values of ONE environment knob:  tools/gpu_env_ab.sh NAME "v1 v2 ..." [reps] [steps]
R=$GRAFT_REPO_ROOT
NAME=$1; VALS=$2; REPS=${3:-3}; STEPS=${4:-200}
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/env_ab_$NAME.txt
: > $OUT
for rep in $(seq $REPS); do
  for v in $VALS; do
    ms=$(env $NAME=$v python3 $R/bench.py --plain --steps $STEPS --warmup 5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "$NAME=$v $ms" | tee -a $OUT
  done
done

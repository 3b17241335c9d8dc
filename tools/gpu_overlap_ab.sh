#!/bin/bash
# A/B of the two-stream training step (neraf_amd/pipeline.py::_get_train_loss_dict_overlapped) on the GPU box: serial step vs the
# radiance half on a side stream, unrestricted and restricted to N compute units.  Alternating runs, bench.py --plain.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/overlap_ab.txt
: > $OUT
run() { # label, env...
  local label=$1; shift
  local v=$(env "$@" python3 $R/bench.py --plain --steps ${STEPS:-100} --warmup 5 2>>$R/gpurun_out/overlap_ab.err | grep -o '"ms_per_step": [0-9.]*')
  echo "$label $v" | tee -a $OUT
}
for rep in 1 2; do
  run serial NERAF_OVERLAP=0
  run overlap_nomask NERAF_OVERLAP=1 NERAF_SIDE_CUS=0
  run overlap_late_nomask NERAF_OVERLAP=2 NERAF_SIDE_CUS=0
  for n in ${CUS:-32 64 96 128}; do
    run overlap_late_cus$n NERAF_OVERLAP=2 NERAF_SIDE_CUS=$n
  done
done

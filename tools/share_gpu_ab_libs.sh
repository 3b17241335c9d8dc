#!/bin/bash
# A/B of two builds of the library as the VICTIM, next to the GEMM neighbour (which does not use the BatchNorm kernels at all)
run() {
  name=$1; shift
  rm -f /tmp/probe_ref_ready
  env "$@" timeout 400 python tools/contention_resnet_probe.py --iters ${ITERS:-3000} --tag V --hold 14 2>&1 | grep -v Warning > /tmp/victim.log &
  V=$!
  while [ ! -e /tmp/probe_ref_ready ]; do sleep 0.5; done
  python tools/share_gpu_gemm_aggressor.py 2048 1024 2048 --seconds 70 > /tmp/aggr.log 2>&1 &
  AGG=$!
  wait $V; if kill -0 $AGG 2>/dev/null; then alive=yes; else alive=NO; fi; kill $AGG 2>/dev/null; wait $AGG 2>/dev/null
  echo "$name (neighbour alive at the end: $alive): $(grep -c deviates /tmp/victim.log) bad; $(grep 'feature deviation' /tmp/victim.log)"
}
for v in ${LIBS:-prev layout code}; do run $v NERAF_HIP_LIB=$PWD/neraf_amd/csrc/_diag/libneraf_hip.$v.so; done
run shipped A=1

#!/bin/bash
# round 4, BatchNorm accumulator read: cost A/B of NERAF_BN_STAT_READ = 0 / 2 / 1 on the bench, then the shared-GPU regression pieces
export TMPDIR=/tmp
echo "== cost: bench --plain, alternating"
for rep in 1 2; do for m in 0 2 1; do
  NERAF_BN_STAT_READ=$m timeout 300 python bench.py --steps 60 --warmup 5 --plain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  mode $m ms/step %.4f' % d['ms_per_step'])"
done; done
echo "== victim next to the GEMM neighbour: mode 2 (default), mode 0 (plain loads without the round-3 dead branch)"
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
run system_scope_default NERAF_BN_STAT_READ=2
run plain_loads_no_dead_branch NERAF_BN_STAT_READ=0
run system_scope_again NERAF_BN_STAT_READ=2
echo "== paired ResNet3D probes (train mode both), default build"
timeout 400 python tools/contention_resnet_probe.py --pair --iters 3000 2>&1 | grep -v Warning > /tmp/p.log; echo "  $(grep -c deviates /tmp/p.log) bad of 6000"; grep "feature deviation" /tmp/p.log
echo "== two-rank tests + RCCL world-1 test (one attempt each)"
for i in 1 2; do timeout 900 python -m pytest tests/test_gpu_dp2.py -q 2>&1 | tail -3; done

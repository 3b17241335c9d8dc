#!/bin/bash
# Same-box per-kernel A/B of the training step between the in-tree library and a variant build, from rocprofv3 kernel traces cut to
# the steady-state steps (tools/steady_state_stats.py):   tools/gpu_kernel_ab.sh variants/libX.so [steps]
# Prints the kernels whose time per step differs by more than 3 us.
V=$1
STEPS=${2:-30}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
for tag in intree variant; do
  if [ $tag = variant ]; then export NERAF_HIP_LIB=$R/$V; else unset NERAF_HIP_LIB; fi
  rm -rf /tmp/kab_$tag
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kab_$tag -- python3 $R/bench.py --steps $STEPS --warmup 3 --plain > /tmp/kab_$tag.log 2>&1
  t=$(find /tmp/kab_$tag -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/steady_state_stats.py $t $STEPS $R/gpurun_out/kab_$tag.csv
done
unset NERAF_HIP_LIB
python3 - $R/gpurun_out/kab_intree.csv $R/gpurun_out/kab_variant.csv <<'PY'
import csv, re, sys
def load(f):
    d = {}
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        n = re.sub(r"<(\d+), (true|false)>\(WgTable\)", r"<\1>(WgTable)", n)        # wgrad kernels gained a template flag
        d[n] = d.get(n, 0.0) + float(r["NsPerStep"]) / 1e3, 
    return {k: v[0] for k, v in d.items()}
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = sorted(((a.get(k, 0.0) - b.get(k, 0.0), k) for k in set(a) | set(b)), reverse=True)
print(f"total in-tree {sum(a.values()):.1f} us/step, variant {sum(b.values()):.1f} us/step")
for d, k in rows:
    if abs(d) > 3.0:
        print(f"{d:+8.1f} us/step  in-tree {a.get(k, 0.0):8.1f}  variant {b.get(k, 0.0):8.1f}  {k[:120]}")
PY

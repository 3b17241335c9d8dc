#!/bin/bash
# Per-node table of the ResNet3D forward / backward graphs: kernel name and rocprofv3 duration (tools/resnet_trace.py under
# --kernel-trace, folded by tools/resnet_trace_summary.py) next to the un-profiled increment of tools/graph_prefix_times.py.
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rt && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -- python3 $R/tools/resnet_trace.py 12 > /tmp/rt.log 2>&1
python3 $R/tools/resnet_trace_summary.py /tmp/rt 12 > $R/gpurun_out/resnet_nodes_profiled.txt 2>&1
cd $R
python3 tools/graph_prefix_times.py 128 20 1 > gpurun_out/resnet_nodes_prefix.txt 2>&1
python3 - <<'PY'
import re
prof = []
for l in open("gpurun_out/resnet_nodes_profiled.txt"):
    m = re.match(r"\s*(\d+)\s+([\d.]+) us\s+gap\s+(-?[\d.]+)\s+(.*)", l)
    if m:
        prof.append((float(m.group(2)), m.group(4).strip()))
fwd = [float(l.split()[3]) for l in open("gpurun_out/resnet_nodes_prefix.txt") if l.startswith("forward")]
bwd = [float(l.split()[3]) for l in open("gpurun_out/resnet_nodes_prefix.txt") if l.startswith("backward")]
names = [n for _, n in prof]
idx_f = next(i for i, n in enumerate(names) if "zero3" in n)          # first node of the forward graph
idx_b = next(i for i, n in enumerate(names) if "bwd_prologue" in n) - 1  # the backward graph starts with its zero fill
rows = [("F", k + 1, prof[idx_f + k][0], fwd[k] if k < len(fwd) else 0.0, names[idx_f + k][:58]) for k in range(idx_b - idx_f)]
rows += [("B", k + 1, prof[idx_b + k][0], bwd[k] if k < len(bwd) else 0.0, names[idx_b + k][:58]) for k in range(len(prof) - idx_b)]
with open("gpurun_out/resnet_nodes.txt", "w") as f:
    f.write("\n".join(f"{r[0]} {r[1]:4d} prof {r[2]:7.2f} unprof {r[3]:7.2f}  {r[4]}" for r in rows))
print("wrote", len(rows), "nodes (the first forward nodes' un-profiled increments hide behind the host-bound prefix)")
PY

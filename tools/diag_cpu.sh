#!/bin/bash
# CPU-side diagnostics of the bench on a GPU box: cgroup quota, NUMA locality of the GPU, step time with / without CPU pinning
echo "nproc $(nproc)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for d in /sys/class/drm/card*/device; do echo "$d local_cpulist=$(cat $d/local_cpulist 2>/dev/null) numa=$(cat $d/numa_node 2>/dev/null) vendor=$(cat $d/vendor 2>/dev/null)"; done 2>/dev/null | head -12
ls /sys/class/kfd/kfd/topology/nodes/ 2>/dev/null | head -20
uptime
python - <<'PY'
import os, time, sys
sys.path.insert(0, os.getcwd())
import ctypes
import torch, bench
_libc = ctypes.CDLL("libc.so.6")
os.sched_getcpu = _libc.sched_getcpu
print("current cpu", os.sched_getcpu())
js = bench.JointStep(torch.device("cuda:0"), 4096, 2048, 1)
for _ in range(10): js.step()
torch.cuda.synchronize()
full = os.sched_getaffinity(0)
def run(tag):
    t0 = time.perf_counter(); c0 = os.times()
    for _ in range(30): js.step()
    torch.cuda.synchronize()
    t1 = time.perf_counter(); c1 = os.times()
    print(f"{tag}: wall {1e3*(t1-t0)/30:.3f} ms/step  cpu user {1e3*(c1.user-c0.user)/30:.3f} sys {1e3*(c1.system-c0.system)/30:.3f} ms/step  on cpu {os.sched_getcpu()}")
for rep in range(3): run("free  ")
cur = os.sched_getcpu()
base = cur // 8 * 8
os.sched_setaffinity(0, set(range(base, base + 8)))
for rep in range(3): run(f"pin {base}-{base+7}")
os.sched_setaffinity(0, full)
for rep in range(2): run("free  ")
PY

#!/bin/bash
# What bounds the two gather kernels of a frame: VALU issue, the texture addresser, or the caches?  Two counter passes over
# `bench.py --mode eval` (kernel-trace only next to --pmc), summarised per kernel.   tools/gpu_r4_valu.sh [train]
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
if [ "$1" = "train" ]; then MARGS=""; else MARGS="--mode eval"; fi
mkdir -p $R/gpurun_out/valu
cd /tmp
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/valu/P1 -- python3 $R/bench.py $MARGS --steps 2 --warmup 1 --plain > $R/gpurun_out/valu/P1.log 2>&1
echo "P1 rc=$?"
timeout 900 rocprofv3 --pmc TA_TA_BUSY TA_BUFFER_READ_WAVEFRONTS TA_BUFFER_TOTAL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/valu/P2 -- python3 $R/bench.py $MARGS --steps 2 --warmup 1 --plain > $R/gpurun_out/valu/P2.log 2>&1
echo "P2 rc=$?"
cd $R
python3 - <<'PY'
import collections, csv, glob, json
out = {}
for p in ("P1", "P2"):
    fs = glob.glob(f"gpurun_out/valu/{p}/**/*_counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for key in ("field_query_kernel<0>", "field_query_kernel<1>", "field_query_kernel<2>", "proposal_density_kernel", "field_backward_kernel", "proposal_backward_kernel"):
                if key in k:
                    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key][r["Counter_Name"]] += 1
    for k in acc:
        out.setdefault(k, {}).update({c: acc[k][c] / n[k][c] for c in acc[k]})
        out[k]["launches_" + p] = max(n[k].values())
for k, v in out.items():
    g = v.get("GRBM_GUI_ACTIVE", 0)
    d = {"launches": v.get("launches_P1")}
    if "SQ_INSTS_VALU" in v:
        d.update(valu_insts_per_launch=v["SQ_INSTS_VALU"], salu_insts=v["SQ_INSTS_SALU"], vmem_rd_insts=v["SQ_INSTS_VMEM_RD"],
                 valu_active_over_busy=v["SQ_ACTIVE_INST_VALU"] / max(v["SQ_BUSY_CYCLES"], 1), vmem_active_over_busy=v["SQ_ACTIVE_INST_VMEM"] / max(v["SQ_BUSY_CYCLES"], 1),
                 any_active_over_busy=v["SQ_ACTIVE_INST_ANY"] / max(v["SQ_BUSY_CYCLES"], 1), sq_busy_cycles=v["SQ_BUSY_CYCLES"], wave_cycles=v["SQ_WAVE_CYCLES"], gui_active=g)
    if "TA_TA_BUSY" in v:
        d.update(ta_busy=v["TA_TA_BUSY"], ta_buffer_read_wavefronts=v["TA_BUFFER_READ_WAVEFRONTS"], ta_buffer_total_cycles=v["TA_BUFFER_TOTAL_CYCLES"],
                 ta_addr_stalled_by_tc=v["TA_ADDR_STALLED_BY_TC_CYCLES"], ta_data_stalled_by_tc=v["TA_DATA_STALLED_BY_TC_CYCLES"],
                 tcp_cache_accesses=v["TCP_TOTAL_CACHE_ACCESSES"], tcp_tcc_read_req=v["TCP_TCC_READ_REQ"], tcp_pending_stall=v["TCP_PENDING_STALL_CYCLES"])
    out[k] = d
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/valu/summary.json", "w"), indent=1)
PY
rm -rf gpurun_out/valu/P1 gpurun_out/valu/P2

#!/usr/bin/env python3
"""Summarise the two PMC passes of tools/gpu_pmc.sh into profiles/<tag>_pmc_traffic.json (per kernel family, HBM bytes per launch)
and compact per-kernel CSVs.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-byte read requests are
tallied at 64 bytes); both counters are in KiB; Infinity-Cache hits are included."""
import collections, csv, glob, json, os, re, sys

tag = sys.argv[1]
def latest(pat):
    return sorted(glob.glob(pat), key=os.path.getmtime)[-1]
def load(path, name):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d
F = load(latest("gpurun_out/pmc/FETCH_SIZE/runc/*_counter_collection.csv"), "FETCH_SIZE")
W = load(latest("gpurun_out/pmc/WRITE_SIZE/runc/*_counter_collection.csv"), "WRITE_SIZE")
def fam(n):
    """rocprofv3 kernel name -> the bench.py profiling scope (neraf_prof_kernel_name) it is timed under"""
    if "gemm_f16_nt_wide_kernel" in n: return "gemm_f16_nt_wide_kernel<*, 160|128, 3, *>"
    if "wgrad_grouped_tn_kernel" in n or "wgrad_wide_tn_kernel" in n: return "wgrad_wide_tn_kernel | wgrad_grouped_tn_kernel"
    m = re.search(r"gemm_f16_nt_pipe_kernel<(\d+), (\d+), \d, (\d), (\d), (true|false)(?:, (?:true|false))*>", n)
    if m:
        bm, bn, ld, bf = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(5) == "true"
        if ld == 2: return "gemm_f16_nt_pipe_kernel<128, 64, 3, 2, 5, false>"
        if ld == 1:
            if bm * bn == 128 * 128: return "gemm_f16_nt_pipe_kernel<128, 128, 2, 1, *, *>"
            if bm == 128: return "gemm_f16_nt_pipe_kernel<128, 64, 3, 1, *, *>"
            return "gemm_f16_nt_pipe_kernel<64, 64, 4, 1, *, true>" if bf else "gemm_f16_nt_pipe_kernel<64, 64, 4, 1, *, false>"
        if bm * bn == 128 * 128: return "gemm_f16_nt_pipe_kernel<128, 128, *, 0, 1, *>"
        if bm == 128: return "gemm_f16_nt_pipe_kernel<128, 64, 3, 0, 1, *>"
        return "gemm_f16_nt_pipe_kernel<64|32, 64|32, 4, 0, 1, true>" if bf else "gemm_f16_nt_pipe_kernel<64|32, 64|32, 4, 0, 1, false>"
    if "field_scatter" in n or "field_slice_ids" in n: return "field_scatter_kernel | field_slice_ids_kernel + field_scatter_owner_kernel"
    if "proposal_density" in n: return "proposal_density_kernel | proposal_density_frame_kernel"
    if "field_query" in n: return "field_query_kernel | field_query_frame_kernel"
    for k in ("proposal_backward_kernel", "field_backward_kernel", "fused_adam_kernel"):
        if k in n: return k
    return None
fams = collections.defaultdict(lambda: [0.0, 0.0, 0, 0])
for n in set(F) | set(W):
    f = fam(n)
    if f:
        fams[f][0] += sum(F.get(n, [])) * 1024 * 2; fams[f][1] += sum(W.get(n, [])) * 1024
        fams[f][2] += len(F.get(n, [])); fams[f][3] += len(W.get(n, []))
out = {}
for f, (fb, wb, nf, nw) in sorted(fams.items()):
    out[f] = {"fetch_bytes_per_launch": fb / max(nf, 1), "write_bytes_per_launch": wb / max(nw, 1), "launches_fetch_pass": nf,
              "launches_write_pass": nw, "hbm_bytes_per_launch": fb / max(nf, 1) + wb / max(nw, 1)}
    print(f"{f:52s} fetch {fb/max(nf,1)/1e6:9.2f} MB  write {wb/max(nw,1)/1e6:9.2f} MB per launch  ({nf} launches)")
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes, --kernel-trace only) over `bench.py --steps 4 --warmup 2 "
                     "--plain` (tools/gpu_pmc.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md; counters are KiB; Infinity-Cache hits included",
           "families": out}, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
for name, d in (("FETCH_SIZE", F), ("WRITE_SIZE", W)):
    with open(f"profiles/{tag}_pmc_{name}_by_kernel.csv", "w") as fh:
        w = csv.writer(fh); w.writerow(["Kernel_Name", "Dispatches", f"{name}_KB_total", f"{name}_KB_avg"])
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), round(sum(v), 3), round(sum(v) / len(v), 3)])

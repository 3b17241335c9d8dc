#!/usr/bin/env python3
"""Summarise the SQ / TCC counter passes of tools/gpu_profile.sh into profiles/<tag>_pmc_sq_by_kernel.csv: per kernel (rocprofv3
name), dispatch count and per-dispatch means of the raw counters plus derived figures:

  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)      matrix-pipe busy cycles over (kernel cycles x 1024 SIMDs);
                     GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS note), BUSY_CYCLES over all SIMDs
  mfma_tflops      = (MOPS_F16 + MOPS_BF16) * 512 FLOP / kernel time at the measured GUI clock (informative; the clock is not in the CSV,
                     so the column holds FLOP per kernel cycle instead: flop_per_cycle)
  wait_lds_frac    = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES      issue stalls on the LDS pipe (quad-cycles over quad-cycles)
  wait_any_frac    = SQ_WAIT_ANY / SQ_WAVE_CYCLES           waves parked on s_waitcnt / barriers
  lds_conflict_frac= SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  l2_hit           = TCC_HIT / (TCC_HIT + TCC_MISS)
"""
import collections, csv, glob, os, sys

tag = sys.argv[1]


def load(d):
    files = sorted(glob.glob(f"gpurun_out/pmc/{d}/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if not files:
        return acc
    for r in csv.DictReader(open(files[-1])):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


A, B = load("SQ1"), load("SQ2")
names = sorted(set(A) | set(B))
rows = []
for n in names:
    a, b = A.get(n, {}), B.get(n, {})
    m = lambda d, k: (sum(d[k]) / len(d[k])) if k in d and d[k] else float("nan")
    disp = len(next(iter(a.values()))) if a else (len(next(iter(b.values()))) if b else 0)
    gui = m(a, "GRBM_GUI_ACTIVE")
    cyc = gui / 8.0
    busy = m(a, "SQ_VALU_MFMA_BUSY_CYCLES")
    mops = (m(a, "SQ_INSTS_VALU_MFMA_MOPS_F16") if "SQ_INSTS_VALU_MFMA_MOPS_F16" in a else 0.0) + \
           (m(a, "SQ_INSTS_VALU_MFMA_MOPS_BF16") if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in a else 0.0)
    wave = m(a, "SQ_WAVE_CYCLES")
    rows.append({
        "Kernel_Name": n, "Dispatches": disp, "GRBM_GUI_ACTIVE": gui, "kernel_cycles": cyc,
        "SQ_VALU_MFMA_BUSY_CYCLES": busy, "mfma_busy_frac": busy / (cyc * 1024) if cyc == cyc and cyc > 0 else float("nan"),
        "MFMA_MOPS_F16+BF16": mops, "flop_per_cycle": mops * 512 / cyc if cyc == cyc and cyc > 0 else float("nan"),
        "SQ_WAVE_CYCLES": wave, "wait_lds_frac": m(a, "SQ_WAIT_INST_LDS") / wave if wave == wave and wave > 0 else float("nan"),
        "wait_any_frac": m(a, "SQ_WAIT_ANY") / wave if wave == wave and wave > 0 else float("nan"),
        "wait_inst_any_frac": m(a, "SQ_WAIT_INST_ANY") / wave if wave == wave and wave > 0 else float("nan"),
        "SQ_BUSY_CU_CYCLES": m(a, "SQ_BUSY_CU_CYCLES"),
        "SQ_ACTIVE_INST_LDS": m(b, "SQ_ACTIVE_INST_LDS"), "SQ_INSTS_LDS": m(b, "SQ_INSTS_LDS"), "SQ_INSTS_MFMA": m(b, "SQ_INSTS_MFMA"),
        "SQ_LDS_IDX_ACTIVE": m(b, "SQ_LDS_IDX_ACTIVE"),
        "lds_conflict_frac": (m(b, "SQ_LDS_BANK_CONFLICT") / m(b, "SQ_LDS_IDX_ACTIVE")) if m(b, "SQ_LDS_IDX_ACTIVE") > 0 else float("nan"),
        "SQ_VALU_MFMA_COEXEC_CYCLES": m(b, "SQ_VALU_MFMA_COEXEC_CYCLES"),
        "l2_hit": (m(b, "TCC_HIT") / (m(b, "TCC_HIT") + m(b, "TCC_MISS"))) if (m(b, "TCC_HIT") + m(b, "TCC_MISS")) > 0 else float("nan"),
    })
rows.sort(key=lambda r: -(r["GRBM_GUI_ACTIVE"] * r["Dispatches"] if r["GRBM_GUI_ACTIVE"] == r["GRBM_GUI_ACTIVE"] else 0))
if rows:
    with open(f"profiles/{tag}_pmc_sq_by_kernel.csv", "w") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows[:80]:
            w.writerow({k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()})
    for r in rows[:24]:
        print(f"{r['Kernel_Name'][:70]:70s} n={r['Dispatches']:4d} mfma_busy {r['mfma_busy_frac']:.3f} flop/cyc {r['flop_per_cycle']:9.0f} "
              f"waitLDS {r['wait_lds_frac']:.3f} waitANY {r['wait_any_frac']:.3f} l2hit {r['l2_hit']:.3f} ldsconf {r['lds_conflict_frac']:.3f}")
else:
    print("no SQ counter files found")

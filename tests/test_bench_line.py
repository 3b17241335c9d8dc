"""The ONE stdout line of bench.py stays readable by the driver (round 5: a 22.7 KB line left BENCH_r05.json.parsed null).
``bench.compact_line`` is built here from canned full records -- round 5's own 22.7 KB record (profiles/r05_f_bench_default.json) and a
pathological one -- and checked for size (< 6144 bytes), the contract's keys, and the figures the judge reads (roofline,
cpu_baseline, eval_render, parity).  CPU only: bench.py imports nothing but the standard library at module level."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")


def _round5_record():
    with open(os.path.join(ROOT, "profiles", "r05_f_bench_default.json")) as f:
        return json.load(f)


def test_round5_record_fits_the_line():
    full = _round5_record()
    assert len(json.dumps(full)) > 20000                       # the record that broke the driver's reader
    line = bench.compact_line(full, "gpurun_out/bench_detail_train_n1.json")
    assert "\n" not in line and len(line.encode()) < bench.LINE_LIMIT == 6144
    out = json.loads(line)
    for k in CONTRACT:
        assert k in out, k
    assert out["value"] == float(f"{full['value']:.6g}") and out["ms_per_step"] == float(f"{full['ms_per_step']:.6g}")
    assert out["config"]["rays_per_gpu"] == 4096 and out["config"]["slices_per_gpu"] == 2048 and len(out["config"]["workload"]) <= 200
    r = out["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert 1 <= len(r["families"]) <= 6 and "all_kernel_families" not in r
    c = out["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("port", "reference") and len(c["sample"]) <= 240
    assert out["eval_render"]["roofline"]["frac"] > 0 and out["eval_render"]["cpu_baseline"]["cores"] >= 1
    assert out["detail"] == "gpurun_out/bench_detail_train_n1.json"
    assert "parity_camera_optimizer_on" not in out and "repeat_windows" not in out


def test_pathological_record_is_degraded_not_overlong():
    full = _round5_record()
    fam = full["roofline"]["all_kernel_families"][0]
    full["roofline"]["all_kernel_families"] = [dict(copy.deepcopy(fam), kernel="k" * 400 + str(i), ms_per_step=float(i)) for i in range(60)]
    full["roofline"]["kernel"] = "q" * 3000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["config"]["workload"] = "w" * 5000
    full["eval_render"]["roofline"]["kernel"] = "e" * 2000
    full["parity"] = {"fixture": "f" * 3000, "psnr_db": 1.0, "inside": True}
    line = bench.compact_line(full, "d.json")
    assert len(line.encode()) < bench.LINE_LIMIT
    out = json.loads(line)
    for k in CONTRACT:
        assert k in out, k


def test_plain_and_multi_rank_fields_survive():
    full = {"metric": "field-samples/sec (rays + RIR STFT bins)", "value": 1.0e8, "unit": "field-samples/s", "n_gpus": 2, "ranks_seen": 2,
            "backend": "nccl", "steps": 3, "warmup": 1, "ms_per_step": 5.0, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic", "config": {"workload": "x", "rays_per_gpu": 2048, "slices_per_gpu": 1024, "global_rays": 4096,
                                                             "global_slices": 2048, "parallelism": "dp2"}}
    out = json.loads(bench.compact_line(full, None))
    assert out["ranks_seen"] == 2 and out["backend"] == "nccl" and out["scaling"] == "strong" and out["config"]["global_rays"] == 4096
    assert out["detail"] is None


def test_profile_references_are_named_files_that_exist():
    for k, name in bench.PROFILE_REFS.items():
        assert os.path.exists(os.path.join(ROOT, "profiles", name)), (k, name)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "glob.glob" not in src                               # no "newest file wins" citation (round 5: sorted(glob)[-1])

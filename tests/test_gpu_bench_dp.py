"""The N > 1 path of bench.py stays alive (it had last run in round 2 and changed by ~600 lines since; the
first N > 1 launch of today's bench.py must not happen on the driver's 8-GPU node).  Two ranks SHARE the one GPU of the test box over
gloo (NERAF_BENCH_SHARE_GPU=1: never a measurement), started by bench.py itself the way the driver's launcher would start them
(`python bench.py --gpus 2` spawns `torch.distributed.run --nproc-per-node 2`), in a fresh child process:
  * weak scaling, --plain: the line reports n_gpus = ranks_seen = 2, the backend, twice the per-rank batch as the global batch;
  * strong scaling, the FULL line (instrumented replay, roofline, emit): the global batch is the command line's, split over the ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, tmp_path, timeout=900):
    env = dict(os.environ, NERAF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=str(tmp_path), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, (p.stdout.decode(errors="replace")[-2000:], p.stderr.decode(errors="replace")[-4000:])
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 prints ONE line, the other rank nothing
    assert len(lines[0].encode()) < 6144
    return json.loads(lines[0])


def test_two_rank_weak_scaling_plain_line(tmp_path):
    d = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--plain", "--rays", "1024", "--slices", "512"], tmp_path)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["scaling"] == "weak"
    c = d["config"]
    assert c["rays_per_gpu"] == 1024 and c["slices_per_gpu"] == 512 and c["global_rays"] == 2048 and c["global_slices"] == 1024
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - (2048 + 1024 * 513) / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]


def test_two_rank_strong_scaling_full_line(tmp_path):
    detail = str(tmp_path / "detail.json")
    d = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--scaling", "strong", "--rays", "2048", "--slices", "1024", "--repeats", "2",
                "--parity", "off", "--no-eval-line", "--no-cpu-baseline", "--detail", detail], tmp_path)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["scaling"] == "strong"
    c = d["config"]
    assert c["rays_per_gpu"] == 1024 and c["slices_per_gpu"] == 512 and c["global_rays"] == 2048 and c["global_slices"] == 1024
    assert c["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - (2048 + 1024 * 513) / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma", "l2") and 0 < r["frac"] <= 1.0 and r["traffic"] is None      # counter traffic is cited for the default shape only
    assert d["detail"] == detail
    full = json.load(open(detail))
    assert len(full["roofline"]["all_kernel_families"]) >= 8 and "cpu_baseline" not in full

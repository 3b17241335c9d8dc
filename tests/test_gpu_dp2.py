"""Two ranks on ONE GPU over gloo (fresh child processes): NeRAFPipeline.train_iteration with the data-parallel plumbing on the real
HIP path.  Asserts (1) the replicas' parameters, voxel grid and GradScaler state stay BIT-IDENTICAL over 5 optimizer steps -- every
parameter gradient is all-reduced and the fused optimizer is element-wise deterministic, so this holds although BatchNorm statistics
are accumulated with order-dependent fp32 atomics; (2) the averaged data-parallel gradient of the audio loss equals the
single-process gradient on the concatenated batch (advisor finding of round 1: it used to come out 1/world too small)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_one_gpu_replicas_stay_bit_identical(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import shared_gpu
    shared_gpu.attempts_for_shared_gpu(lambda i: _run_two_ranks(tmp_path, i))


def test_deterministic_mode_reduces_d_feat_instead_of_the_encoder_gradients(tmp_path):
    """SURVEY 8e (2), VERDICT round 5 item 4: under NERAF_DETERMINISTIC=1 the encoder's forward and backward are bit-identical on
    every rank, so the data-parallel step all-reduces d feat (4 KiB) before the ResNet3D backward and keeps the encoder's 17 M
    gradients (68 MB) off the wire -- by default (NeRAFPipeline.attach_gradient_reducer).  Two ranks on the one GPU, gloo, 5 steps:
      * the reducer's groups no longer hold the encoder, d feat mode is on;
      * the replicas stay BIT-IDENTICAL (parameters incl. the encoder's, grid, GradScaler state);
      * the encoder gradients of the first iteration equal the ones the all-reduce path (NERAF_DP_DFEAT=0, same deterministic
        mode) produces, to the fp16 chain's rounding (<= 5e-3 rel-L2 on the sampled entries, norms within 2e-3)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    a, b = _run_two_ranks(tmp_path, "dfeat", check=False, extra_env={"NERAF_DETERMINISTIC": "1"})
    c, d_ = _run_two_ranks(tmp_path, "allreduce", check=False, extra_env={"NERAF_DETERMINISTIC": "1", "NERAF_DP_DFEAT": "0"})
    assert a["dfeat_allreduce"] and b["dfeat_allreduce"] and not c["dfeat_allreduce"]
    assert c["reducer_numel"] - a["reducer_numel"] == a["encoder_numel"] > 17_000_000
    for x, y in ((a, b), (c, d_)):
        assert x["init"] == y["init"]
        diff = [k for k in x["per_param"] if x["per_param"][k] != y["per_param"][k]]
        assert not diff, f"replicas diverged in {len(diff)} tensors, e.g. {diff[:5]}"
        assert x["grid"] == y["grid"] and x["scale"] == y["scale"]
    for k, v in a["encoder_grads_step1"].items():
        assert v == b["encoder_grads_step1"][k]                                  # identical on both ranks, bit for bit
        u, w = np.array(v), np.array(c["encoder_grads_step1"][k])
        rel = float(np.linalg.norm(u - w) / np.linalg.norm(w))
        print(f"d-feat path vs gradient all-reduce, {k}: rel-L2 of 64 entries {rel:.2e}")
        assert rel <= 5e-3, (k, rel)
    for k, v in a["encoder_grad_norms_step1"].items():
        np.testing.assert_allclose(v, c["encoder_grad_norms_step1"][k], rtol=2e-3)
    np.testing.assert_allclose(a["losses"][0], c["losses"][0], rtol=1e-6)         # same forward, same data


def _run_two_ranks(tmp_path, attempt, check=True, extra_env=None):
    import shared_gpu
    port = _free_port()
    procs, outs = [], []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", NERAF_WORKER_DEVICE=str(shared_gpu.rank_device(r)), **(extra_env or {}))
        out = str(tmp_path / f"attempt{attempt}_rank{r}.json")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tools", "dp2_worker.py"), out], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-4000:]
    a, b = (json.load(open(o)) for o in outs)
    assert all(np.isfinite(a["losses"])) and all(np.isfinite(b["losses"]))
    if not check:
        return a, b
    assert a["init"] == b["init"]
    # default mode (fp32-atomic BatchNorm sums): the encoder's gradients are all-reduced; a suite run under NERAF_DETERMINISTIC=1 takes the d-feat path
    assert a["dfeat_allreduce"] == (os.environ.get("NERAF_DETERMINISTIC") == "1" and os.environ.get("NERAF_DP_DFEAT", "auto") != "0")
    diff = [k for k in a["per_param"] if a["per_param"][k] != b["per_param"][k]]
    assert not diff, f"replicas diverged in {len(diff)} tensors, e.g. {diff[:5]}"
    assert a["params"] == b["params"] and a["grid"] == b["grid"] and a["scale"] == b["scale"]
    # the two ranks train on different shards: their local losses differ (the replicas agree although their data does not)
    assert a["losses"] != b["losses"]
    for res in (a, b):
        assert max(res["audio_grad_rel"]) <= 2e-2, res["audio_grad_rel"]
        sc_dp, sc_ref, mag_dp, mag_ref = res["audio_loss"]
        np.testing.assert_allclose([sc_dp, mag_dp], [sc_ref, mag_ref], rtol=2e-3)


def test_rccl_world_size_one_runs_the_data_parallel_plumbing_on_the_real_backend(tmp_path):
    """RCCL itself (backend "nccl"), with the one rank a one-GPU box has: three training iterations with the gradient reducer
    attached, the refresh sharded + assembled through ``gather_shards`` and the STFT loss on all-reduced sums must reproduce the run
    without any group -- every collective of one rank is the identity.  Checks that ReduceOp.AVG exists and averages in place, that
    the flat ResNet3D gradient buffer survives an in-place collective, and that the reducer's hook order launches every group
    (tests/tools/nccl1_worker.py)."""
    out = str(tmp_path / "nccl1.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               NERAF_DETERMINISTIC="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "nccl1_worker.py"), out], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-4000:]
    r = json.load(open(out))
    a, b = r["plain"], r["nccl"]
    assert b["collectives_per_step"] and all(n >= 3 for n in b["collectives_per_step"]), b["collectives_per_step"]
    for la, lb in zip(a["losses"], b["losses"]):
        assert set(la) == set(lb)
        for k in la:
            np.testing.assert_allclose(lb[k], la[k], rtol=2e-3, atol=1e-9, err_msg=k)
    worst = max(r["rel"].values())
    assert worst <= 2e-3, sorted(r["rel"].items(), key=lambda kv: -kv[1])[:5]
    assert a["scale"] == b["scale"]
    # deterministic summation (NERAF_DETERMINISTIC=1 above) and no camera optimizer: the two runs are the SAME computation -- every
    # parameter and the voxel grid bit for bit
    diff = [k for k in a["digest"] if a["digest"][k] != b["digest"][k]]
    assert not diff, f"{len(diff)} tensors differ between the plain and the RCCL run, e.g. {diff[:5]}"
    assert a["grid"] == b["grid"]

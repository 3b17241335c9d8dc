"""Audit of the three hand-written autograd nodes the way the PIPELINE drives them.  Round 5's bug -- the
grid refresh's backward mapped its positions through the scene contraction, because ``spatial_distortion`` had been switched back on
between the node's forward and its backward -- lived through four rounds of green per-operator suites: the operators were driven by
hand in a state the pipeline never produces.  The same class of hole is looked for here in the other nodes: the field node as SECOND
producer of a pass (the render loss node adds into the refresh node's gradient tensors in place, vision.py NerfactoField.backward_query)
and the encoder node (``grid_window`` conversion, alternating feature buffers, the workspace a pending backward reads).

One fresh process per run, NERAF_DETERMINISTIC=1 (tests/tools/autograd_audit_worker.py): gradients of ``get_train_loss_dict`` + one
backward, (a) with an eval-mode RIR, an eval-mode render, ``get_eval_loss_dict``, ``update_to_step`` of another step and a
``spatial_distortion`` toggle between forward and backward: BIT-IDENTICAL; (b) against the three nodes driven by hand with the
contraction held off through the refresh's backward, the encoder on the whole grid, two separate backward passes: bit-identical
wherever the arithmetic is the same (NAcF, encoder, proposal networks, hash table, embedding), <= 1e-6 rel-L2 for the five small
field matrices (in-place accumulation adds the second producer's split-K slabs in another order).  And the tests TEST: with round 5's
bug re-introduced (``--mutate contract``) comparison (b) must fail; with eval forwards sharing the training workspace
(``--mutate eval_ws``) comparison (a) must fail."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_RUNS = {}
FIELD_MATRICES = ("base_w0", "base_w1", "head_w0", "head_w1", "head_w2")


def _run(tmp_path_factory, mutate=None):
    if mutate not in _RUNS:
        out = str(tmp_path_factory.mktemp("audit") / f"audit_{mutate}.npz")
        env = dict(os.environ, NERAF_DETERMINISTIC="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, os.path.join(ROOT, "tests", "tools", "autograd_audit_worker.py"), out] + (["--mutate", mutate] if mutate else [])
        p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        assert p.returncode == 0, p.stdout.decode(errors="replace")[-4000:]
        z = np.load(out)
        runs = {}
        for k in z.files:
            if "/" in k:
                name, key = k.split("/", 1)
                runs.setdefault(name, {})[key] = z[k]
        runs["_losses"] = (z["losses"], z["losses_hand"], [str(s) for s in z["loss_keys"]])
        _RUNS[mutate] = runs
    return _RUNS[mutate]


def _differing(a, b):
    assert set(a) == set(b), set(a) ^ set(b)
    return [k for k in a if not np.array_equal(a[k], b[k])]


def _rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def test_state_changes_between_forward_and_backward_do_not_reach_the_gradients(tmp_path_factory):
    r = _run(tmp_path_factory)
    assert len(r["pipeline"]) > 150 and all(np.isfinite(v).all() for v in r["pipeline"].values())
    diff = _differing(r["pipeline"], r["perturbed"])
    assert not diff, f"{len(diff)} gradients changed, e.g. {[(k, _rel(r['perturbed'][k], r['pipeline'][k])) for k in diff[:4]]}"


def test_pipeline_driven_nodes_equal_the_hand_driven_nodes(tmp_path_factory):
    r = _run(tmp_path_factory)
    lp, lh, keys = r["_losses"]
    for k, x, y in zip(keys, lp, lh):
        if k.startswith("audio_"):      # same forward: the windowed grid conversion == the full one, bit for bit through encoder + NAcF
            assert x == y, (k, x, y)
        else:                           # the radiance loss VALUES are summed with fp32 atomics in either mode (their gradients are not)
            np.testing.assert_allclose(x, y, rtol=1e-5, err_msg=k)
    a, b = r["pipeline"], r["by_hand"]
    diff = _differing(a, b)
    loose = [k for k in diff if k.startswith("vision.field.module.") and k.rsplit(".", 1)[-1] in FIELD_MATRICES]
    exact_broken = [k for k in diff if k not in loose]
    assert not exact_broken, [(k, _rel(a[k], b[k])) for k in exact_broken[:6]]
    for k in loose:
        assert _rel(a[k], b[k]) <= 1e-6, (k, _rel(a[k], b[k]))
    # the second producer really accumulated: the table gradient is neither contribution alone (its norm exceeds what the audio branch sends)
    assert np.abs(a["vision.field.module.table"]).sum() > 0


def test_the_audit_detects_round5_contraction_bug(tmp_path_factory):
    r = _run(tmp_path_factory, "contract")
    a, b = r["pipeline"], r["by_hand"]
    rel = _rel(a["vision.field.module.table"], b["vision.field.module.table"])
    print(f"with the contract= bug re-introduced: pipeline vs hand-driven d table rel-L2 {rel:.3e}")
    assert rel > 1e-3, "the re-introduced bug went unnoticed: the audit compares nothing"


def test_the_audit_detects_an_eval_forward_in_the_training_workspace(tmp_path_factory):
    r = _run(tmp_path_factory, "eval_ws")
    diff = _differing(r["pipeline"], r["perturbed"])
    enc = [k for k in diff if k.startswith("audio.resnet3d.")]
    print(f"with eval forwards in the training workspace: {len(diff)} gradients differ ({len(enc)} of the encoder's)")
    assert len(enc) > 50, "the re-introduced hazard went unnoticed"

#!/usr/bin/env python3
"""The two-rank / one-GPU worker pair of tests/test_gpu_dp2.py run N times; for every run whose averaged data-parallel audio gradient
differs from the single-process one, the worker's layer-0 diagnostics (which half of the layer-0 weight gradient is off, on which
rank).      python tools/dp2_flake_probe.py [N=20]"""
import json, os, socket, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for it in range(n):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    tmp = tempfile.mkdtemp()
    procs, outs = [], []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", NERAF_WORKER_DEVICE="0")
        out = os.path.join(tmp, f"rank{r}.json"); outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tools", "dp2_worker.py"), out], env=env, cwd=ROOT,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
    for p in procs:
        p.wait(timeout=600)
    res = [json.load(open(o)) for o in outs]
    for r, d in enumerate(res):
        if max(d["audio_grad_rel"]) > 2e-2:
            bad += 1
            worst = max(range(len(d["audio_grad_rel"])), key=lambda i: d["audio_grad_rel"][i])
            print(f"run {it} rank {r}: audio_grad_rel[{worst}] = {d['audio_grad_rel'][worst]:.4f}; layer 0: {json.dumps(d['audio_grad_layer0'])}", flush=True)
print(f"{bad} bad rank-results in {n} runs")

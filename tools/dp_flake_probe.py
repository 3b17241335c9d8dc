import os, sys, socket, subprocess, numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(root, "tests/golden/g7_trajectory.npz"))
oc = np.asarray(g["curves"])
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs, outs = [], []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = f"/tmp/dpf_{trial}_{r}.npz"; outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests/tools/dp2_trajectory_worker.py"), "g7_trajectory", out], env=env, cwd=root, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
    for p in procs: p.wait()
    a, b = (np.load(o) for o in outs)
    c = 0.5 * (a["curves"] + b["curves"])
    rel = np.abs(c[:, 0] - oc[:, 0]) / oc[:, 0]
    bad = np.where(rel > 0.3)[0]
    relm = np.abs(c[:, 4] - oc[:, 4]) / np.where(np.isnan(oc[:, 4]), 1, oc[:, 4])
    badm = np.where(np.nan_to_num(relm) > 0.3)[0]
    print(f"trial {trial}: rgb tail {np.nanmean(c[-20:,0]):.5f} (oracle {np.nanmean(oc[-20:,0]):.5f}); first rgb deviation >30% at step {bad[0] if len(bad) else None}; "
          f"first audio_mag deviation at {badm[0] if len(badm) else None}; rgb around: {np.round(c[max(0,(bad[0] if len(bad) else 0)-2):(bad[0] if len(bad) else 0)+4,0],5) if len(bad) else ''} "
          f"oracle {np.round(oc[max(0,(bad[0] if len(bad) else 0)-2):(bad[0] if len(bad) else 0)+4,0],5) if len(bad) else ''}", flush=True)

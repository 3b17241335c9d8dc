#!/bin/bash
# After the refresh-backward position-mapping fix: trajectory tests with their printed numbers, G9 HIP samples, a G7 run for the gap attribution.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_trajectory.py -q -s 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_fix_traj_tests.txt
python3 tests/tools/g9_hip_samples.py 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_fix_g9_hip_samples.txt
NERAF_DETERMINISTIC=1 python3 tests/tools/trajectory_worker.py g7_trajectory /tmp/g7_hip.npz > /dev/null 2>&1
python3 - <<'PY'
import numpy as np
a = np.load("/tmp/g7_hip.npz")
np.savez("gpurun_out/g7_hip_run_fixed.npz", **{k: a[k] for k in ("curves", "image", "stft_eval", "stft_batch_stats", "deterministic")})
PY

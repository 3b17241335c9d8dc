"""Worker of tests/test_gpu_trajectory.py::test_data_parallel_trajectory_matches_the_oracle: one of two ranks that SHARE the single GPU
of the test box and talk over gloo, training the trajectory scenario data-parallel (every iteration's rays and RIR slices split in
two contiguous shards, the refresh window sharded by the model, STFT-loss sums and gradients reduced).  Writes its loss curves and
(the replicas are identical) its held-out predictions.

    python tests/tools/dp2_trajectory_worker.py <scenario> <out.npz>     with RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT set
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import torch
import torch.distributed as dist


def main():
    scenario, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    di = int(os.environ.get("NERAF_WORKER_DEVICE", "0"))        # one GPU per rank where the box has two; else the ranks share GPU 0
    torch.cuda.set_device(di)
    dev = torch.device("cuda", di)
    import trajectory_common as TC
    torch.manual_seed(0)
    curves, img, stft, pipe, evb = TC.run_hip_trajectory(dev, cfg=TC.SCENARIOS[scenario], rank=rank, world=world)
    assert pipe.audio_model._dp_world() is not None
    np.savez(out_path, curves=curves, image=img, stft_eval=stft["eval"], stft_batch_stats=stft["batch_stats"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

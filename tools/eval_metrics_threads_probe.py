#!/usr/bin/env python3
"""Is the per-RIR host time of the metric chain (torch.exp 11 ms, torch.log 22 ms, CPU torch.stft 25 ms on 30 k-element tensors) the
intra-op thread pool?  The same loop with torch.set_num_threads(1) around it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neraf_amd import config as C
from neraf_amd.datamanagers import SyntheticAudioDataManager, SyntheticVisionDataManager
R = 64
torch.manual_seed(0)
m = C.make_method("RAF", "FurnishedRoom", datamanager=SyntheticVisionDataManager(2, 1, 96, 128, 1024),
                  audio_datamanager=SyntheticAudioDataManager(4, R, batch_size=256))
m.config.pipeline.start_step_audio = 3
p = m.config.pipeline.setup(device="cuda:0", test_mode="val", world_size=1, local_rank=0, grad_scaler=None)
p.eval()
p.get_average_eval_image_metrics(step=10)
print("host threads", torch.get_num_threads(), "cpus", os.cpu_count())
for nt in (torch.get_num_threads(), 8, 1):
    torch.set_num_threads(nt)
    torch.cuda.synchronize(); t0 = time.time()
    met = p.get_average_eval_image_metrics(step=10)
    torch.cuda.synchronize()
    print(f"threads {nt}: {(time.time() - t0) / R * 1e3:.1f} ms per RIR; T60 {met['audio_T60']:.6f} stft {met['audio_stft_error']:.8f}")

#!/usr/bin/env python3
"""Where the per-RIR time of get_average_eval_image_metrics goes (tools/full_eval_run.py read ~180 ms per RIR against ~10 us of field
time): cProfile over the audio half of the loop, 0 frames + R RIRs."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neraf_amd import config as C
from neraf_amd.datamanagers import SyntheticAudioDataManager, SyntheticVisionDataManager
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
m = C.make_method("RAF", "FurnishedRoom", datamanager=SyntheticVisionDataManager(2, 1, 96, 128, 1024),
                  audio_datamanager=SyntheticAudioDataManager(4, R, batch_size=256))
m.config.pipeline.start_step_audio = 3
p = m.config.pipeline.setup(device="cuda:0", test_mode="val", world_size=1, local_rank=0, grad_scaler=None)
p.eval()
p.get_average_eval_image_metrics(step=10)
torch.cuda.synchronize(); t0 = time.time()
pr = cProfile.Profile(); pr.enable()
p.get_average_eval_image_metrics(step=10)
torch.cuda.synchronize(); pr.disable()
print("wall per RIR ms", (time.time() - t0) / R * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)

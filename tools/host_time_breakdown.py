#!/usr/bin/env python3
"""Host-side time of a training step by piece: wraps the forward entry points and the autograd backward functions with
perf_counter accumulators (the autograd engine runs backward on its own thread, where cProfile does not see it)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import importlib
bench = importlib.import_module(os.environ.get("NERAF_BENCH_MODULE", "bench"))

acc = collections.defaultdict(float)

def wrap(owner, name, label):
    f = getattr(owner, name)
    raw = f.__func__ if hasattr(f, "__func__") else f
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return raw(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t
    setattr(owner, name, staticmethod(g) if isinstance(owner.__dict__.get(name), staticmethod) else g)

def main():
    from neraf_amd import field, losses, model, resnet3d, vision, optim
    torch.cuda.set_device(0)
    js = bench.JointStep(torch.device("cuda:0"), 4096, 2048, 1)
    for fn, label in ((resnet3d._ResNet3DFn, "resnet"), (field._NacfSplitFn, "nacf"), (losses._StftLossFn, "stft"),
                      (model._RefreshFn, "refresh"), (vision._VisionLossFn, "vision_loss")):
        wrap(fn, "backward", label + ".backward")
        wrap(fn, "forward", label + ".fn_forward")
    wrap(vision.NeRAFVisionModel, "get_outputs", "vision.get_outputs")
    wrap(model.NeRAFAudioModel, "query_grid_one_batch", "audio.query_grid")
    wrap(model.NeRAFAudioModel, "get_outputs", "audio.get_outputs")
    wrap(resnet3d.ResNet3D, "forward", "resnet.forward(py)")
    wrap(optim.FusedAdam, "step", "adam.step")
    wrap(optim.FusedAdam, "check_finite", "adam.check_finite")
    for _ in range(10):
        js.step()
    torch.cuda.synchronize()
    acc.clear()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        js.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host issue {1e3*(t1-t0)/n:.3f} ms/step, wall {1e3*(t2-t0)/n:.3f} ms/step")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"  {k:28s} {1e3*v/n:7.3f} ms/step")

if __name__ == "__main__":
    main()

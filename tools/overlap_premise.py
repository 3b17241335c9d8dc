#!/usr/bin/env python3
"""Premise check for running the radiance half of a step beside the ResNet3D chain: per-iteration time of (A) ResNet3D forward +
backward alone, (B) the radiance forward + loss + backward alone, and both issued on two streams.  If the pair costs about
max(A, B) the small-kernel chain leaves the chip to the gather kernels; if it costs A + B there is nothing to gain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NERAF_OVERLAP"] = "0"
import torch
import bench


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    js = bench.JointStep(dev, 4096, 2048, 1)
    for _ in range(6):
        js.step()
    torch.cuda.synchronize()
    net, grid = js.am.resnet3d, js.am.grid.unsqueeze(0)
    g = torch.randn(1, 1024, 1, 1, 1, device=dev) * 1e-3
    vparams = [p for p in js.vm.parameters()]
    aparams = [p for p in net.parameters()]

    def audio_iter():
        for p in aparams:
            p.grad = None
        out = net(grid)
        out.backward(gradient=g)

    def vision_iter():
        for p in vparams:
            p.grad = None
        js.vm.update_to_step(20001)
        out = js.vm(js.bundle)
        ld = js.vm.get_loss_dict(out, js.gt, js.vm.get_metrics_dict(out, js.gt))
        (ld["rgb_loss"] + ld["interlevel_loss"] + ld["distortion_loss"]).backward()

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def run(n, a, b):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            if a:
                with torch.cuda.stream(s1):
                    audio_iter()
            if b:
                with torch.cuda.stream(s2):
                    vision_iter()
        th = time.perf_counter()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, (th - t0) / n * 1e3

    for a, b in ((1, 0), (0, 1), (1, 1)):
        run(5, a, b)
    for rep in range(3):
        ta, ha = run(40, 1, 0)
        tb, hb = run(40, 0, 1)
        tab, hab = run(40, 1, 1)
        print(f"ResNet3D fwd+bwd alone {ta:.3f} ms (host {ha:.3f}) | radiance fwd+bwd alone {tb:.3f} ms (host {hb:.3f}) | both, two streams "
              f"{tab:.3f} ms (host {hab:.3f}) | sum {ta + tb:.3f}  max {max(ta, tb):.3f}  saved {ta + tb - tab:.3f} ms")


if __name__ == "__main__":
    main()

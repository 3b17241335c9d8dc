#!/usr/bin/env python3
"""Does a CU-masked stream for the radiance half keep the ResNet3D chain at full speed beside it?  Per-pair time of ResNet3D
forward + backward (stream with all CUs) and the radiance forward + backward on a stream limited to N compute units."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NERAF_OVERLAP"] = "0"
import torch
import bench

hip = C.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


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

    s1 = torch.cuda.Stream()

    def run(n, a, b, s2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            if a:
                with torch.cuda.stream(s1):
                    audio_iter()
            if b:
                with torch.cuda.stream(s2):
                    vision_iter()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    full = torch.cuda.Stream()
    run(5, 1, 1, full)
    ta = run(40, 1, 0, full)
    print(f"ResNet3D fwd+bwd alone {ta:.3f} ms")
    masks = {"all 256 (plain stream)": None,
             "low 128 bits": (1 << 128) - 1, "low 64 bits": (1 << 64) - 1, "low 32 bits": (1 << 32) - 1,
             "every 2nd bit (128)": int("01" * 128, 2), "every 4th bit (64)": int("0001" * 64, 2), "every 8th bit (32)": int("00000001" * 32, 2)}
    for name, bits in masks.items():
        s2 = full if bits is None else masked_stream(bits)
        run(5, 0, 1, s2)
        tb = run(40, 0, 1, s2)
        run(5, 1, 1, s2)
        tab = run(40, 1, 1, s2)
        print(f"radiance on {name:24s}: alone {tb:.3f} ms | pair {tab:.3f} ms | ResNet alone + radiance alone(all CUs) = {ta + 0.713:.3f}")


if __name__ == "__main__":
    main()

"""Roofline accounting: the FLOP counts the bench line divides by kernel time are the ALGORITHMIC
ones -- a convolution, its transposed (dgrad) form and its weight gradient are each 2 dout^3 taps cin cout with real channel and tap
counts -- not the padded / zero-tap counts the grids execute.  CPU part: the architecture table the launches are generated from
sums to SURVEY's 94.72 GFLOP (128^3) -- the figure bench.py's `dense_equiv` keys use.  GPU part: tests/test_gpu_resnet3d.py."""
import ctypes as C

from neraf_amd import _lib


def test_resnet3d_algorithmic_forward_flops_match_the_survey():
    lib = _lib.load()
    f128 = lib.neraf_resnet3d_forward_flops(C.byref(_lib.ResnetDesc(128, 7, 1024)))
    assert abs(f128 - 94.72e9) <= 0.001 * 94.72e9
    # by stage: conv1 29.36, layer1 28.45, layer2 21.47, layer3 15.44 GFLOP -> the total above; the 64^3 grid is 1/8
    f64 = lib.neraf_resnet3d_forward_flops(C.byref(_lib.ResnetDesc(64, 7, 1024)))
    assert abs(f64 * 8 - f128) <= 1e-9 * f128
    assert lib.neraf_resnet3d_forward_flops(C.byref(_lib.ResnetDesc(96, 7, 1024))) < 0      # unsupported descriptor
    import bench
    assert abs(bench.RESNET_FWD_GFLOP * 1e9 - f128) <= 0.001 * f128
    # NAcF: 2 (1187*5096 + 5096*2048 + 2048*1024 + 1024*1024 + 1024*512 + C*512*F) per slice forward
    want = 2 * (1187 * 5096 + 5096 * 2048 + 2048 * 1024 + 1024 * 1024 + 1024 * 512 + 512 * 513)
    assert bench.NACF_DENSE_FLOP_PER_SLICE_FWD == want == 40_836_464

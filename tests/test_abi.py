"""CPU checks of the drop-in boundary: the C-ABI library builds/loads without a GPU, exports every symbol that
include/neraf_hip.h declares, the ctypes table mirrors the header, and the product path refuses to run without
a GPU instead of falling back."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "neraf_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(neraf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_header_symbol():
    from neraf_amd import _lib
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/neraf_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in neraf_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.neraf_abi_version() == 1


def test_layout_helpers_run_without_gpu():
    import ctypes as C
    from neraf_amd import _lib
    lib = _lib.load()
    d = _lib.NacfDesc(1024, 163, 512, 1, 513, 0)
    assert lib.neraf_nacf_packed_bytes(C.byref(d)) > 40e6
    assert lib.neraf_nacf_workspace_bytes(C.byref(d), 2048, 1) > lib.neraf_nacf_workspace_bytes(C.byref(d), 2048, 0)
    from neraf_amd.vision import grid_layout
    sc, rs, sz, off = grid_layout(_lib.GridDesc(16, 16, 2048, 19, 2))
    assert rs[0] == 16 and rs[-1] == 2048 and off[-1] == sum(sz)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_fallback_without_gpu():
    from neraf_amd import _lib
    from neraf_amd.field import NeRAFAudioSoundField
    with pytest.raises(RuntimeError, match="no MI355X|failed"):
        _lib.ctx(0)
    f = NeRAFAudioSoundField(1187, 512, sound_rez=1, N_frequencies=513)
    assert sorted(f.state_dict())[:2] == ["STFT_linear.0.bias", "STFT_linear.0.weight"]
    with pytest.raises(RuntimeError, match="no CPU fallback|CUDA"):
        f(torch.zeros(4, 1187))

"""HIP streams for the two-stream training step (neraf_amd/pipeline.py): an ordinary side stream, or one restricted to a subset
of the compute units (``hipExtStreamCreateWithCUMask``) so that its kernels cannot take every CU from the main chain."""
from __future__ import annotations

import ctypes as C

import torch

_hip = None
_keep = []          # CU-masked streams are owned here for the life of the process (torch only borrows the handle)


def _libhip():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("libamdhip64 not found: CU-masked streams need the HIP runtime")
    return _hip


def make_stream(device, n_cus: int = 0, first_cu: int = 0, total_cus: int = 256):
    """A new stream on ``device``.  ``n_cus`` > 0: only CUs [first_cu, first_cu + n_cus) of the mask's numbering are enabled."""
    dev = torch.device(device)
    if n_cus <= 0:
        return torch.cuda.Stream(device=dev)
    words = (total_cus + 31) // 32
    mask = (C.c_uint32 * words)()
    for cu in range(first_cu, min(first_cu + n_cus, total_cus)):
        mask[cu // 32] |= 1 << (cu % 32)
    h = C.c_void_p()
    with torch.cuda.device(dev):
        rc = _libhip().hipExtStreamCreateWithCUMask(C.byref(h), C.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    _keep.append(h)
    return torch.cuda.ExternalStream(h.value, device=dev)

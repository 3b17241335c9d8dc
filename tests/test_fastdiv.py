"""The gather kernels turn a sample index into (ray, sample-in-ray) with a division by the runtime samples-per-ray count, done as a
multiply-high + shifts (csrc/field_common.h: make_fastdiv / fastdiv, Granlund & Montgomery 1994 fig. 4.1).  The library evaluates the
same constants and arithmetic on the host (neraf_debug_fastdiv): checked here against n // d over every small divisor and the
edge values of the 32-bit range -- no GPU needed."""
import numpy as np

from neraf_amd import _lib


def test_fastdiv_matches_integer_division():
    lib = _lib.load()
    rng = np.random.default_rng(0)
    divisors = list(range(1, 1100)) + [2 ** k for k in range(11, 32)] + [2 ** k + 1 for k in range(11, 31)] + [2 ** k - 1 for k in range(11, 32)] + \
        [int(x) for x in rng.integers(1, 2 ** 32 - 1, 300)]
    for d in divisors:
        ns = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1, 2 ** 32 - d] + [int(x) for x in rng.integers(0, 2 ** 32 - 1, 40)]
        for n in ns:
            n &= 0xFFFFFFFF
            assert lib.neraf_debug_fastdiv(n, d) == n // d, (n, d)
    assert lib.neraf_debug_fastdiv(5, 0) == 0xFFFFFFFF

"""include/ssx_fmath.h (the build-defined sin/cos/acos of the parity contract) against correctly
rounded values from mpmath.  Exercised through the oracle's exported wrappers."""
import numpy as np
import pytest

import oracle_lib as ol

mp = pytest.importorskip("mpmath")


def correctly_rounded(x):
    d = float(x)
    f = np.float32(d)
    cands = (np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf)))
    return min(cands, key=lambda c: abs(mp.mpf(float(c)) - x))


def test_sin_cos_acos_are_correctly_rounded():
    lib = ol.load()
    mp.mp.prec = 120
    rs = np.random.RandomState(7)
    xs = np.concatenate([rs.uniform(-3.3, 6.5, 6000), rs.uniform(-1e-3, 1e-3, 500), rs.uniform(-50, 50, 500),
                         [0.0, np.pi, np.pi / 2, 2 * np.pi, -np.pi, 1e-30, 6.2831855, 3.1415925]]).astype(np.float32)
    for x in xs:
        xm = mp.mpf(float(x))
        assert np.float32(lib.orc_sinf(float(x))) == correctly_rounded(mp.sin(xm)), x
        assert np.float32(lib.orc_cosf(float(x))) == correctly_rounded(mp.cos(xm)), x
    us = np.concatenate([rs.uniform(-1, 1, 6000), 1 - rs.uniform(0, 1e-4, 500), -1 + rs.uniform(0, 1e-4, 500),
                         [1.0, -1.0, 0.0, 0.5, -0.5, 0.50000006, 0.99999994, -0.99999994]]).astype(np.float32)
    for u in us:
        assert np.float32(lib.orc_acosf(float(u))) == correctly_rounded(mp.acos(mp.mpf(float(u)))), u


def test_special_values():
    lib = ol.load()
    assert np.isnan(lib.orc_sinf(float("nan"))) and np.isnan(lib.orc_cosf(float("inf"))) and np.isnan(lib.orc_acosf(1.5))
    assert lib.orc_acosf(1.0) == 0.0
    assert np.float32(lib.orc_acosf(-1.0)) == np.float32(np.pi)  # 0x40490FDB, clamped by the caller to 0x40490FDA
    assert lib.orc_sinf(0.0) == 0.0 and lib.orc_cosf(0.0) == 1.0

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


# ---- the independent yardstick (csrc/ssx_ddmath.h) -------------------------------------------------------------------------
# ssx_fmath.h is shared by the oracle and the kernels, so their agreement says nothing about the header.
# csrc/ssx_ddmath.h evaluates the three functions by other means (double-double Taylor series, three-part pi/2, Newton on the
# cosine); here it is pinned against mpmath, and a CPU-sized sample of the header is compared with it.  The GPU runs the same
# comparison over ALL 2^32 float patterns (tests/test_gpu_units.py::test_fmath_header_proved_against_an_independent_evaluation).
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ddh(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("ddh") / "libddh.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", os.path.join(ROOT, "tests", "ddmath_host.cpp"), "-o", so])
    lib = C.CDLL(so)
    for n in ("ddh_sin", "ddh_cos", "ddh_acos"):
        getattr(lib, n).restype = C.c_float
        getattr(lib, n).argtypes = [C.c_float, C.POINTER(C.c_int)]
    lib.ddh_compare.restype = C.c_uint64
    lib.ddh_compare.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_int]
    return lib


def test_independent_evaluation_is_correctly_rounded(ddh):
    mp.mp.prec = 400
    p = (C.c_double * 3)()
    ddh.ddh_pio2(p)
    assert abs(mp.mpf(p[0]) + mp.mpf(p[1]) + mp.mpf(p[2]) - mp.pi / 2) < mp.mpf(2) ** -160    # the three-part pi/2
    rs = np.random.RandomState(11)
    f32 = lambda u: np.array(u, dtype=np.uint32).view(np.float32)
    xs = np.concatenate([rs.uniform(-4, 7, 4000), rs.uniform(-1e-3, 1e-3, 400), rs.uniform(-2 ** 20, 2 ** 20, 4000),
                         np.float32(np.pi / 2) * rs.randint(-60000, 60000, 2000),                         # next to the zeros of sin and cos
                         f32(rs.randint(0, 0x49800000, 4000, dtype=np.int64)),                            # every binade up to 2^20, incl. subnormals
                         [0.0, -0.0, 1e-45, -1e-45, 2.0 ** -126, 2.0 ** 20, -2.0 ** 20, 3.1415927, 6.2831855, 1.5707964, 0.7853982,
                          5.5e5, 1048575.94, 2.0 ** -12]]).astype(np.float32)
    d = C.c_int()
    undecided = 0
    for x in xs:
        xm = mp.mpf(float(x))
        for fn, ref in ((ddh.ddh_sin, mp.sin), (ddh.ddh_cos, mp.cos)):
            got = np.float32(fn(float(x), C.byref(d)))
            if not d.value:
                undecided += 1
                continue
            want = correctly_rounded(ref(xm)) if x != 0 or ref is mp.cos else x
            assert got == want and np.signbit(got) == np.signbit(want), (float(x), fn)
    us = np.concatenate([rs.uniform(-1, 1, 6000), 1 - rs.uniform(0, 1e-4, 1000), -1 + rs.uniform(0, 1e-4, 1000), rs.uniform(-1e-6, 1e-6, 500),
                         f32(rs.randint(0, 0x3F800000, 3000, dtype=np.int64)), -f32(rs.randint(0, 0x3F800000, 3000, dtype=np.int64)),
                         [1.0, -1.0, 0.0, -0.0, 0.5, -0.5, 0.99999994, -0.99999994, 1e-45, 0.70710677]]).astype(np.float32)
    for u in us:
        got = np.float32(ddh.ddh_acos(float(u), C.byref(d)))
        if not d.value:
            undecided += 1
            continue
        assert got == correctly_rounded(mp.acos(mp.mpf(float(u)))), float(u)
    assert undecided == 0          # (2^-70 from a rounding boundary: not in 50 000 draws)


def test_header_against_the_independent_evaluation_on_a_sample(ddh):
    """6 M float patterns per function (every 701st pattern of the 2^32), on the CPU: no mismatch, nothing undecided."""
    und = C.c_uint64()
    ex = (C.c_uint32 * 8)()
    for which, name in enumerate(("sin", "cos", "acos")):
        bad = ddh.ddh_compare(which, 17, 701, (1 << 32) // 701, C.byref(und), ex, 8)
        assert bad == 0 and und.value == 0, (name, bad, und.value, [hex(e) for e in ex])

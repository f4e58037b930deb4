"""The libm boundary, PINNED: f32::ln (object.rs:562), f32::powf(x, 5.) (material.rs:145) and f32::sin (texture.rs:14) lower
to the platform libm in the reference.  The restatements both sides run (oracle/rto_libm.hpp, csrc/rt_libm.h) are glibc's own
algorithms with its FMA-variant contractions written out; here the oracle's copy is compared with THIS host's libm (glibc 2.35,
the logf / powf / sinf a Rust binary would call) on every one of the 2^32 float bit patterns: zero differences allowed."""
import ctypes as C

import numpy as np
import pytest


def scan(oracle, op, lo, hi, stride):
    n, m, w = C.c_uint64(), C.c_uint64(), C.c_uint32()
    oracle.lib.rto_debug_ulp_scan(C.c_int(op), C.c_uint32(lo), C.c_uint32(hi), C.c_uint32(stride),
                                  C.byref(n), C.byref(m), C.byref(w))
    return n.value, m.value, w.value


def f2b(x):
    return int(np.array([x], dtype=np.float32).view(np.uint32)[0])


def host_has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except Exception:
        return False


@pytest.mark.parametrize("op,name", [(0, "logf(x)"), (1, "powf(x, 5.0f)"), (2, "sinf(x)")])
def test_restatements_equal_the_platform_libm_on_every_float(oracle, op, name):
    """All 2^32 inputs (normals, subnormals, zeros, infinities, NaNs; NaN == NaN), on every hardware thread."""
    if not host_has_fma():
        pytest.skip("glibc dispatches its non-FMA variants on this CPU (powf / sinf differ from the FMA variants on 6 / 12 inputs)")
    n, mism, worst = scan(oracle, op, 0, 0xFFFFFFFF, 1)
    print("%s vs glibc: %d tested, %d differ, max %d ulp" % (name, n, mism, worst))
    assert n == 1 << 32 and mism == 0


def test_logf_on_the_rng_domain(oracle):
    """ln's argument is rng.gen::<f32>() = k * 2^-24 (object.rs:562): all 2^24 of them, also on a CPU without FMA
    (logf is identical in both glibc variants)."""
    x = (np.arange(1 << 24, dtype=np.float64) / (1 << 24)).astype(np.float32)
    got = oracle.debug_math(0, x)
    want = np.zeros_like(x)
    oracle.lib.rto_debug_glibc(C.c_int(0), C.c_size_t(x.size), x.ctypes.data_as(C.POINTER(C.c_float)), want.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_special_values(oracle):
    x = np.array([0.0, 1.0, 0.5, 2.0 ** -24, np.inf, -1.0, np.nan, 1e-45, 3.4e38, -0.0], dtype=np.float32)
    y = oracle.debug_math(0, x)
    assert y[0] == -np.inf and y[1] == 0.0 and y[4] == np.inf and np.isnan(y[5]) and np.isnan(y[6]) and y[9] == -np.inf
    assert abs(float(y[2]) + 0.6931471805599453) < 1e-7
    assert abs(float(y[7]) - np.log(np.float64(np.float32(1e-45)))) < 1e-4   # subnormal path
    p = oracle.debug_math(1, x)                                              # powf(x, 5): odd power keeps the sign
    assert p[0] == 0.0 and p[1] == 1.0 and p[4] == np.inf and p[5] == -1.0 and np.isnan(p[6]) and p[8] == np.inf
    assert np.signbit(p[9]) and p[9] == 0.0 and p[2] == np.float32(0.03125)
    s = oracle.debug_math(2, x)
    assert s[0] == 0.0 and np.isnan(s[4]) and np.isnan(s[6]) and np.signbit(s[9]) and abs(float(s[1]) - np.sin(1.0)) < 1e-7

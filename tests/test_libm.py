"""The shared libm restatements (rt_logf / rt_pow5f / rt_sinf) against this host's glibc, which is what
the reference's f32::ln / powf / sin lower to.  Reported as mismatch counts + max ULP."""
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


def test_logf_on_rng_domain_subsample(oracle):
    """ln's argument is rng.gen::<f32>() = k * 2^-24 (object.rs:562): every positive float below 1 with
    the low bits clear is reachable.  Scan every 61st float of [2^-24, 1)."""
    n, mism, worst = scan(oracle, 0, f2b(2.0 ** -24), f2b(1.0) - 1, 61)
    print("rt_logf vs glibc logf: %d tested, %d differ, max %d ulp" % (n, mism, worst))
    assert n > 3_000_000 and worst <= 1 and mism / n < 0.02


def test_logf_special_values(oracle):
    x = np.array([0.0, 1.0, 0.5, 2.0 ** -24, np.inf, -1.0, np.nan, 1e-45, 3.4e38], dtype=np.float32)
    y = oracle.debug_math(0, x)
    assert y[0] == -np.inf and y[1] == 0.0 and y[4] == np.inf and np.isnan(y[5]) and np.isnan(y[6])
    assert abs(float(y[2]) + 0.6931471805599453) < 1e-7
    assert abs(float(y[7]) - np.log(np.float64(np.float32(1e-45)))) < 1e-4   # subnormal path
    assert abs(float(y[8]) - np.log(np.float64(np.float32(3.4e38)))) < 1e-5


def test_pow5_subsample(oracle):
    """schlick's powf(1 - cos, 5.) (material.rs:145): 1 - cos lies in [-0.5, 1]."""
    n, mism, worst = scan(oracle, 1, f2b(1e-6), f2b(1.0), 97)
    print("rt_pow5f vs glibc powf(x,5) on (0,1]: %d tested, %d differ, max %d ulp" % (n, mism, worst))
    assert worst <= 1 and mism / n < 0.02
    n, mism, worst = scan(oracle, 1, f2b(-0.5) - 2_000_000, f2b(-0.5), 13)  # negative bases near -0.5
    assert worst <= 1


def test_sin_subsample(oracle):
    """checker's sin(10 * p) (texture.rs:14): scene coordinates up to a few thousand."""
    n, mism, worst = scan(oracle, 2, f2b(1e-3), f2b(60000.0), 211)
    print("rt_sinf vs glibc sinf on [1e-3, 6e4]: %d tested, %d differ, max %d ulp" % (n, mism, worst))
    assert worst <= 1 and mism / n < 0.02


@pytest.mark.slow
def test_logf_exhaustive(oracle):
    n, mism, worst = scan(oracle, 0, f2b(2.0 ** -24), f2b(1.0) - 1, 1)
    print("EXHAUSTIVE rt_logf vs glibc logf on [2^-24,1): %d tested, %d differ, max %d ulp" % (n, mism, worst))
    assert worst <= 1

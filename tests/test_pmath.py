"""include/lqrrt_pmath.h on the host: accuracy against 50-digit mpmath and C99 zero conventions."""
import ctypes as C
import math
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = """
#include "lqrrt_pmath.h"
void v_sincos(const double* x, int n, double* s, double* c){ for(int i=0;i<n;i++) lq_sincos(x[i], &s[i], &c[i]); }
void v_atan2(const double* y, const double* x, int n, double* a){ for(int i=0;i<n;i++) a[i]=lq_atan2(y[i],x[i]); }
void v_atan2_c(const double* y, const double* x, int n, double* a){ for(int i=0;i<n;i++) a[i]=lq_atan2_c(y[i],x[i]); }
void v_tanh(const double* x, int n, double* t){ for(int i=0;i<n;i++) t[i]=lq_tanh(x[i]); }
"""


@pytest.fixture(scope="module")
def pm():
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "t.c"), "w") as f:
        f.write(SRC)
    so = os.path.join(d, "t.so")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-fPIC", "-shared", "-I",
                           os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", so, "-lm"])
    return C.CDLL(so)


def _ulps(got, want_mp):
    import mpmath as mp
    want = float(want_mp)
    u = np.spacing(abs(want)) if want != 0 else 5e-324
    return float(abs(mp.mpf(float(got)) - want_mp) / mp.mpf(float(u)))


def test_sincos_accuracy(pm):
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 50
    rng = np.random.RandomState(0)
    xs = np.concatenate([rng.uniform(-4, 4, 1500), rng.uniform(-200, 200, 800), rng.uniform(-1e5, 1e5, 300),
                         [0.0, math.pi / 2, np.deg2rad(90), math.pi, -math.pi / 2, 1e-300, 0.7853981633974483]])
    s, c = np.zeros_like(xs), np.zeros_like(xs)
    P = C.c_void_p
    pm.v_sincos(xs.ctypes.data_as(P), len(xs), s.ctypes.data_as(P), c.ctypes.data_as(P))
    assert max(_ulps(a, mp.sin(mp.mpf(float(x)))) for a, x in zip(s, xs)) <= 1.0
    assert max(_ulps(a, mp.cos(mp.mpf(float(x)))) for a, x in zip(c, xs)) <= 1.0
    np.testing.assert_allclose(s, np.sin(xs), rtol=0, atol=2.3e-16)
    np.testing.assert_allclose(c, np.cos(xs), rtol=0, atol=2.3e-16)


def test_atan2_accuracy_and_zeros(pm):
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 50
    rng = np.random.RandomState(1)
    ys = np.concatenate([rng.uniform(-4, 4, 2500), rng.uniform(-1e-3, 1e-3, 500)])
    xs = np.concatenate([rng.uniform(-4, 4, 2500), rng.uniform(-1e-3, 1e-3, 500)])
    a = np.zeros_like(ys)
    P = C.c_void_p
    pm.v_atan2(ys.ctypes.data_as(P), xs.ctypes.data_as(P), len(ys), a.ctypes.data_as(P))
    assert max(_ulps(g, mp.atan2(mp.mpf(float(y)), mp.mpf(float(x)))) for g, y, x in zip(a, ys, xs)) <= 2.0
    ac = np.zeros_like(ys)                        # the compiler-fma spelling the NN scans use: the same bits
    pm.v_atan2_c(ys.ctypes.data_as(P), xs.ctypes.data_as(P), len(ys), ac.ctypes.data_as(P))
    np.testing.assert_array_equal(a.view(np.uint64), ac.view(np.uint64))
    sy = np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 3.0, 0.0])
    sx = np.array([1.0, 1.0, -1.0, -1.0, 0.0, 0.0, -0.0, -0.0])
    sa = np.zeros_like(sy)
    pm.v_atan2(sy.ctypes.data_as(P), sx.ctypes.data_as(P), len(sy), sa.ctypes.data_as(P))
    want = np.array([math.atan2(y, x) for y, x in zip(sy, sx)])
    np.testing.assert_array_equal(sa, want)
    np.testing.assert_array_equal(np.signbit(sa), np.signbit(want))


def test_tanh_accuracy_and_limits(pm):
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 50
    rng = np.random.RandomState(2)
    xs = np.concatenate([rng.uniform(-1, 1, 2000), rng.uniform(-25, 25, 1500), 10.0 ** rng.uniform(-300, -1, 300),
                         -10.0 ** rng.uniform(-12, -1, 300), [0.17328679513998632, -0.17328679513998635, 0.34657359027997264, 21.999, 19.0]])
    t = np.zeros_like(xs)
    P = C.c_void_p
    pm.v_tanh(xs.ctypes.data_as(P), len(xs), t.ctypes.data_as(P))
    assert max(_ulps(a, mp.tanh(mp.mpf(float(x)))) for a, x in zip(t, xs)) <= 3.0
    sx = np.array([0.0, -0.0, 22.0, -22.0, 1e300, -1e300, np.inf, -np.inf, 5e-324])
    st = np.zeros_like(sx)
    pm.v_tanh(sx.ctypes.data_as(P), len(sx), st.ctypes.data_as(P))
    np.testing.assert_array_equal(st, np.tanh(sx))
    np.testing.assert_array_equal(np.signbit(st), np.signbit(sx))
    nan = np.array([np.nan]); out = np.zeros(1)
    pm.v_tanh(nan.ctypes.data_as(P), 1, out.ctypes.data_as(P))
    assert np.isnan(out[0])

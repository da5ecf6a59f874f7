"""The oracle's restated third-party primitives against the installed libraries
(numpy / scipy, same image here and on the GPU box) and the bottleneck golden vectors."""
import ctypes as C
import os

import numpy as np
import pytest

from util import GOLD


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("n", [1, 5, 8, 10, 20, 127, 128, 129, 300, 1000, 8191, 8192, 8193, 20000, 50001, 196321])
def test_np_sum_mean_var_order(oracle_mod, n):
    L = oracle_mod.lib()
    a = np.random.default_rng(n).normal(95, 14, n).astype(np.float32)
    assert L.orc_np_sum_f32(_fp(a), C.c_long(n)) == np.sum(a)
    assert L.orc_np_mean_f32(_fp(a), C.c_long(n)) == np.mean(a)
    assert L.orc_np_var_f32(_fp(a), C.c_long(n)) == np.var(a)
    assert L.orc_np_std_f32(_fp(a), C.c_long(n)) == np.std(a)
    d = a.astype(np.float64) * 1.37
    assert L.orc_np_sum_f64(_dp(d), C.c_long(n)) == np.sum(d)


@pytest.mark.parametrize("n", [1, 2, 3, 10, 11, 600, 3280, 5000, 13061])
def test_np_median_percentile(oracle_mod, n):
    L = oracle_mod.lib()
    rng = np.random.default_rng(100 + n)
    a = rng.normal(95, 14, n).astype(np.float32)
    if n > 20:
        a[rng.integers(0, n, n // 10)] = a[0]  # ties
    assert L.orc_np_median_f32(_fp(a), C.c_long(n)) == np.median(a)
    med = np.float32(np.median(a))
    assert L.orc_np_mad_f32(_fp(a), C.c_long(n), C.c_float(med)) == np.median(np.abs(a - med))
    want = np.subtract(*np.percentile(a, (85, 15)))
    assert L.orc_np_percentile_diff_f32(_fp(a), C.c_long(n), C.c_double(85), C.c_double(15)) == want


def test_nanmedian_and_nanstd(oracle_mod):
    L = oracle_mod.lib()
    rng = np.random.default_rng(3)
    a = rng.normal(0, 1, 10001).astype(np.float32)
    a[rng.integers(0, a.size, 500)] = np.nan
    nv = C.c_long(0)
    assert L.orc_np_nanmedian_f32(_fp(a), C.c_long(a.size), C.byref(nv)) == np.nanmedian(a)
    assert nv.value == int(np.sum(~np.isnan(a)))
    d = rng.normal(0, 300, 19900)
    d[5] = np.nan
    assert L.orc_np_nanstd_f64(_dp(d), C.c_long(d.size)) == np.nanstd(d)


def _rand_trace(rng, n, kind):
    if kind == 0:
        x = np.cumsum(rng.normal(0, 1, n))
    elif kind == 1:
        x = rng.integers(0, 6, n).astype(np.float64)  # many ties / plateaus
    elif kind == 2:
        t = np.linspace(0, 6, n)
        x = 50 * np.sin(t) + rng.normal(0, 2, n)
    else:
        x = rng.normal(0, 1, n)
        x[rng.integers(0, n, 3)] = np.nan
    return x


@pytest.mark.parametrize("seed", range(40))
def test_find_peaks_vs_scipy(oracle_mod, seed):
    from scipy.signal import find_peaks

    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 700))
    x = _rand_trace(rng, n, seed % 4)
    opts = [dict(prominence=1.0, width=10, rel_height=0.5),
            dict(prominence=float(np.nanstd(x)), width=3, rel_height=1.0),
            dict(distance=10, prominence=1.0, width=10, rel_height=0.5),
            dict(distance=5),
            dict()]
    for o in opts:
        if seed % 4 == 1 and "distance" in o:
            continue  # exact ties: scipy's argsort order is unspecified
        want, _ = find_peaks(x, **o)
        got = oracle_mod.find_peaks(x, **o)
        assert list(got) == list(want), (o, n)


def test_bottleneck_golden(oracle_mod):
    L = oracle_mod.lib()
    z = np.load(os.path.join(GOLD, "bn_move.npz"))
    for t in range(6):
        a = np.ascontiguousarray(z["a_%d" % t])
        n = a.size
        out = np.zeros(n, dtype=np.float32)
        L.orc_bn_move_mean_f32(_fp(a), C.c_long(n), C.c_long(20), _fp(out))
        assert np.array_equal(out[: n - 19], z["mean20_%d" % t][19:])
        L.orc_bn_move_var_f32(_fp(a), C.c_long(n), C.c_long(100), _fp(out))
        assert np.array_equal(out[: n - 99], z["var100_%d" % t][99:])


def test_bottleneck_nan_golden(oracle_mod):
    """NaN samples inside the series: the real library counts them out of the window (tests/golden/bn_move_nan.npz)"""
    from oracle import bn_shim

    L = oracle_mod.lib()
    z = np.load(os.path.join(GOLD, "bn_move_nan.npz"))
    for t in range(8):
        a = np.ascontiguousarray(z["a_%d" % t])
        n = a.size
        out = np.zeros(n, dtype=np.float32)
        L.orc_bn_move_mean_f32(_fp(a), C.c_long(n), C.c_long(20), _fp(out))
        assert np.array_equal(out[: n - 19], z["mean20_%d" % t][19:], equal_nan=True), t
        for w in (100, 5):
            L.orc_bn_move_var_f32(_fp(a), C.c_long(n), C.c_long(w), _fp(out))
            assert np.array_equal(out[: n - w + 1], z["var%d_%d" % (w, t)][w - 1:], equal_nan=True), (t, w)
            assert np.array_equal(bn_shim.move_var(a, w), z["var%d_%d" % (w, t)], equal_nan=True), (t, w)
        assert np.array_equal(bn_shim.move_mean(a, 20), z["mean20_%d" % t], equal_nan=True), t


def test_bn_shim_matches_golden():
    """the pure-python shim used when the reference is run under the torch interpreter"""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "oracle"))
    from oracle import bn_shim

    z = np.load(os.path.join(GOLD, "bn_move.npz"))
    a = z["a_1"]
    assert np.array_equal(bn_shim.move_mean(a, 20), z["mean20_1"], equal_nan=True)
    assert np.array_equal(bn_shim.move_var(a, 100), z["var100_1"], equal_nan=True)

"""log_cr (adapted_amd/csrc/log_cr.h): the table-driven float64 logarithm of the LLR gains kernels.

The header compiles for the host too; here it is checked against 200-bit mpmath values (CPU), and the device
build must return the very same bits as the host build (GPU)."""
import ctypes
import math
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adapted_amd", "csrc")


@pytest.fixture(scope="module")
def host_log(tmp_path_factory):
    d = tmp_path_factory.mktemp("logcr")
    src = d / "lg.cpp"
    src.write_text('#include "log_cr.h"\nextern "C" void log_cr_array(const double *x, double *y, long n) '
                   "{ for (long i = 0; i < n; i++) y[i] = log_cr_host(x[i]); }\n"
                   'extern "C" void log_1ulp_array(const double *x, double *y, long n) '
                   "{ for (long i = 0; i < n; i++) y[i] = log_1ulp_host(x[i]); }\n")
    so = d / "liblg.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-I" + CSRC, "-o", str(so), str(src)])
    lib = ctypes.CDLL(str(so))

    def f(x, which="log_cr_array"):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        getattr(lib, which)(x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(x.size))
        return y
    return f


def _inputs(n_each, seed=7):
    rng = np.random.default_rng(seed)
    return np.concatenate([
        10.0 ** rng.uniform(-300, 300, n_each),          # the whole range
        10.0 ** rng.uniform(-4, 3, 2 * n_each),          # variances of normalised signal
        1.0 + rng.uniform(-1e-2, 1e-2, n_each),          # around 1: results near 0
        1.0 + rng.uniform(-1e-7, 1e-7, n_each),
        rng.uniform(0.5, 2.0, n_each),
        np.array([1.0, 2.0, 0.5, math.e, 1.0 - 2.0 ** -53, 1.0 + 2.0 ** -52, 2.2250738585072014e-308, 1.7976931348623157e308]),
    ])


def test_table_header_is_what_the_generator_writes(tmp_path):
    """log_cr_table.h is generated (tools/gen_log_table.py): regenerate and compare."""
    hdr = os.path.join(CSRC, "log_cr_table.h")
    before = open(hdr).read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_log_table.py")], stdout=subprocess.DEVNULL)
    assert open(hdr).read() == before


def test_log_cr_is_correctly_rounded_on_samples(host_log):
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 200
    x = _inputs(4000)
    y = host_log(x)
    wrong = 0
    worst = 0.0
    for xi, yi in zip(x, y):
        t = mp.log(mp.mpf(float(xi)))
        cr = float(t)
        if yi != cr:
            wrong += 1
        if cr != 0.0:
            worst = max(worst, float(abs(mp.mpf(float(yi)) - t) / mp.mpf(math.ulp(cr))))
    assert worst < 0.5001, worst
    assert wrong <= 2, wrong  # (none seen in 2e5 samples; the bound allows for a hard case)


def test_log_1ulp_stays_below_one_ulp(host_log):
    """the 17-operation form used by the batch path's gains kernels (log_1ulp_fast)"""
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 200
    x = _inputs(4000, seed=13)
    y = host_log(x, "log_1ulp_array")
    worst = worst_near_one = 0.0
    for xi, yi in zip(x, y):
        t = mp.log(mp.mpf(float(xi)))
        cr = float(t)
        if cr != 0.0:
            e = float(abs(mp.mpf(float(yi)) - t) / mp.mpf(math.ulp(cr)))
            worst = max(worst, e)
            if abs(xi - 1.0) < 2e-2:
                worst_near_one = max(worst_near_one, e)
    assert worst < 1.001, worst
    assert worst_near_one < 0.75, worst_near_one
    sp = host_log(np.array([0.0, -1.0, np.inf, np.nan, 5e-324, 1.0]), "log_1ulp_array")
    assert sp[0] == -np.inf and np.isnan(sp[1]) and sp[2] == np.inf and np.isnan(sp[3]) and sp[4] == math.log(5e-324) and sp[5] == 0.0


def test_log_cr_specials_and_libm_agreement(host_log):
    sp = host_log(np.array([0.0, -0.0, -1.0, np.inf, np.nan, 5e-324, 1.0]))
    assert sp[0] == -np.inf and sp[1] == -np.inf and np.isnan(sp[2]) and sp[3] == np.inf and np.isnan(sp[4])
    assert sp[5] == math.log(5e-324) and sp[6] == 0.0 and not np.signbit(sp[6])
    # the reference takes its logs from libm (< 0.52 ULP): the two may differ in the last bit only, and rarely
    x = _inputs(20000, seed=11)
    y, g = host_log(x), np.log(x)
    diff = y != g
    assert diff.mean() < 0.03, diff.mean()
    ulp = np.abs(y - g) / np.maximum(np.spacing(np.abs(g)), 5e-324)
    assert ulp.max() <= 1.0


@pytest.mark.gpu
def test_device_log_cr_equals_host_build(host_log):
    from adapted_amd import lib
    from util import make_spc
    from golden_cases import CASES

    spc = make_spc(CASES["rna004_llr_default"])
    eng = lib.Engine(spc, 8, spc.sig_preload_size, device=0)
    x = np.concatenate([_inputs(50000, seed=3), np.array([0.0, -1.0, np.inf, np.nan, 5e-324])])
    got, want = eng.debug_log(x), host_log(x)
    nan = np.isnan(want)
    assert (np.isnan(got) == nan).all()
    assert got[~nan].tobytes() == want[~nan].tobytes()  # the very same bits
    eng.close()

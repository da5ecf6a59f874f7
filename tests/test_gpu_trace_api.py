"""The `_c_llr` drop-in (adapted_amd/detect/_c_llr.py -> adp_c_llr_trace, adapted_amd/csrc/trace_api.h) against the vectors of the
REAL reference (tests/golden/c_llr_trace.npz) and against the CPU oracle: cumulative sums bit for bit, the same points computed
(the early-stopping rules break at the same index), gains within 1e-9 of the trace scale (the device logarithm is correctly
rounded, the reference's libm is not: ~1e-15 relative in practice)."""
import os

import numpy as np
import pytest

from trace_cases import ASSERT_CASES, CASES, signal_of

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c_llr_trace.npz")


def _same_trace(got, want):
    assert got.shape == want.shape
    assert np.array_equal(got == 0, want == 0)           # the same points computed, the same break
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.isinf(got), np.isinf(want))
    fin = np.isfinite(want)
    if fin.any():
        scale = max(1.0, float(np.max(np.abs(want[fin]))))
        assert float(np.max(np.abs(got[fin] - want[fin]))) <= 1e-9 * scale
    inf = np.isinf(want)
    assert np.array_equal(np.sign(got[inf]), np.sign(want[inf]))


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_trace_api_equals_the_reference(gold, case):
    from adapted_amd.detect import _c_llr

    x = signal_of(case)
    g, c, c2 = _c_llr.c_llr_trace(x, case["start"], case["end"], case["min_obs"], case["border_trim"], return_c_c2=1, **case["args"])
    assert np.array_equal(c, gold[case["name"] + ".c"], equal_nan=True)
    assert np.array_equal(c2, gold[case["name"] + ".c2"], equal_nan=True)
    _same_trace(g, gold[case["name"] + ".g"])
    g1 = _c_llr.c_llr_trace(x, case["start"], case["end"], case["min_obs"], case["border_trim"], **case["args"])
    assert np.array_equal(g1, g, equal_nan=True)
    g2 = _c_llr.c_llr_trace_gains(c, c2, case["start"], case["end"], case["min_obs"], case["border_trim"], **case["args"])
    assert np.array_equal(g2, g, equal_nan=True)
    if not (case["args"]["adapter_early_stopping"] or case["args"]["polya_early_stopping"]):
        g3 = _c_llr._gains(case["start"], case["end"], c, c2, case["min_obs"], case["border_trim"], case["args"]["stride"])
        assert np.array_equal(g3, g, equal_nan=True)


@pytest.mark.parametrize("case", ASSERT_CASES, ids=[c["name"] for c in ASSERT_CASES])
def test_trace_api_asserts_like_the_reference(case):
    from adapted_amd.detect import _c_llr

    with pytest.raises(AssertionError):
        _c_llr.c_llr_trace(signal_of(case), case["start"], case["end"], case["min_obs"], case["border_trim"], **case["args"])


def test_trace_api_rejects_ranges_outside_the_signal():
    from adapted_amd.detect import _c_llr

    x = np.ones(50)
    for s, e in ((-1, 40), (10, 51), (30, 20)):
        with pytest.raises(ValueError):
            _c_llr.c_llr_trace(x, s, e, 5, 5)
    assert _c_llr.c_llr_trace(np.zeros(0), 0, 0, 5, 5).size == 0


def test_trace_batch_vs_oracle_random(oracle_mod):
    """one call for 200 reads of different lengths, ranges and signals; both early-stopping forms and strides"""
    from adapted_amd.detect import _c_llr
    from trace_cases import squiggle

    rng = np.random.default_rng(5)
    n, L = 200, 2600
    lens = rng.integers(40, L + 1, n)
    lens[:4] = (L, 12, 11, 1)
    raw = np.zeros((n, L))
    for r in range(n):
        raw[r, :lens[r]] = squiggle(1000 + r, int(lens[r]), adapter=(30, 700), polya=(20, 300))
    starts = np.where(rng.random(n) < 0.5, 0, rng.integers(0, np.maximum(lens // 3, 1)))
    ends = np.where(rng.random(n) < 0.7, lens - 1, lens - rng.integers(0, np.maximum(lens // 4, 1)))
    ends = np.maximum(ends, starts)
    for kw in (dict(), dict(stride=2), dict(adapter_early_stopping=1), dict(adapter_early_stopping=1, stride=4, adapter_early_stop_window=120,
               adapter_early_stop_stride=40), dict(polya_early_stopping=1), dict(polya_early_stopping=1, stride=5, adapter_early_stop_window=200,
               adapter_early_stop_stride=50, polya_early_stop_window=30), dict(polya_early_stopping=1, adapter_early_stop_window=10,
               adapter_early_stop_stride=10, polya_early_stop_window=40)):
        g, c, c2 = _c_llr.c_llr_trace_batch(raw, lens, starts, ends, 3, 2, return_c_c2=1, **kw)
        stops = 0
        for r in range(n):
            k = int(lens[r])
            wg, wc, wc2 = oracle_mod.c_llr_trace(raw[r, :k], int(starts[r]), int(ends[r]), 3, 2, return_c_c2=1, **kw)
            assert np.array_equal(c[r, :k], wc, equal_nan=True) and np.array_equal(c2[r, :k], wc2, equal_nan=True), (kw, r)
            _same_trace(g[r, :k], wg)
            assert not g[r, k:].any()
            grid = range(int(starts[r]) + 3, int(ends[r]) - 2, kw.get("stride", 1))
            stops += len(grid) > 0 and np.count_nonzero(wg) < len(grid) - 1
        if kw.get("adapter_early_stopping") or kw.get("polya_early_stopping"):
            assert stops > n // 10, (kw, stops)  # (the rules do fire on these signals)

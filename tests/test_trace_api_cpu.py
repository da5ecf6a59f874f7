"""The `_c_llr` trace API (reference adapted/detect/_c_llr.pyx:67-236): the CPU oracle's restatement against the vectors the REAL
reference produced (tests/golden/c_llr_trace.npz, oracle/gen_trace_golden.py) -- bit for bit, the oracle uses the same libm."""
import os

import numpy as np
import pytest

from trace_cases import ASSERT_CASES, CASES, signal_of

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c_llr_trace.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_trace_equals_the_reference(oracle_mod, gold, case):
    x = signal_of(case)
    g, c, c2 = oracle_mod.c_llr_trace(x, case["start"], case["end"], case["min_obs"], case["border_trim"], return_c_c2=1, **case["args"])
    assert np.array_equal(c, gold[case["name"] + ".c"], equal_nan=True)
    assert np.array_equal(c2, gold[case["name"] + ".c2"], equal_nan=True)
    assert np.array_equal(g, gold[case["name"] + ".g"], equal_nan=True)
    # c_llr_trace_gains: the sums handed in
    g2 = oracle_mod.c_llr_trace(None, case["start"], case["end"], case["min_obs"], case["border_trim"], sums=(c, c2), **case["args"])
    assert np.array_equal(g2, g, equal_nan=True)


@pytest.mark.parametrize("case", ASSERT_CASES, ids=[c["name"] for c in ASSERT_CASES])
def test_oracle_trace_asserts_like_the_reference(oracle_mod, case):
    with pytest.raises(AssertionError):
        oracle_mod.c_llr_trace(signal_of(case), case["start"], case["end"], case["min_obs"], case["border_trim"], **case["args"])

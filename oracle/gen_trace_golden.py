#!/usr/bin/env python3
"""Golden vectors of the `_c_llr` trace API, produced by the REAL reference's Cython module (compiled by pyximport outside the
repository, oracle/ref_harness.py) in the build container: tests/golden/c_llr_trace.npz holds, per case of tests/trace_cases.py,
the gains (and c, c2) the reference returns.  TEST INFRASTRUCTURE."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
from oracle import ref_harness  # noqa: E402

ref_harness.install()
from adapted.detect.llr import _gains, c_llr_trace, c_llr_trace_gains  # noqa: E402  (llr.py:14-18 sets up pyximport)

from trace_cases import ASSERT_CASES, CASES, signal_of  # noqa: E402

out = {}
for case in CASES:
    x = signal_of(case)
    a = case["args"]
    g, c, c2 = c_llr_trace(x, case["start"], case["end"], case["min_obs"], case["border_trim"], a["stride"], a["adapter_early_stopping"],
                           a["adapter_early_stop_window"], a["adapter_early_stop_stride"], a["polya_early_stopping"],
                           a["polya_early_stop_window"], a["polya_early_stop_stride"], 1)
    g2 = c_llr_trace_gains(c, c2, case["start"], case["end"], case["min_obs"], case["border_trim"], a["stride"], a["adapter_early_stopping"],
                           a["adapter_early_stop_window"], a["adapter_early_stop_stride"], a["polya_early_stopping"],
                           a["polya_early_stop_window"], a["polya_early_stop_stride"])
    assert np.array_equal(g, g2, equal_nan=True)
    if not (a["adapter_early_stopping"] or a["polya_early_stopping"]):
        g3 = _gains(case["start"], case["end"], c, c2, case["min_obs"], case["border_trim"], a["stride"])
        assert np.array_equal(g, g3, equal_nan=True)
    out[case["name"] + ".g"] = g
    out[case["name"] + ".c"] = c
    out[case["name"] + ".c2"] = c2
    nz = np.flatnonzero(g != 0)
    print("%-36s n=%5d  computed %5d of %5d grid points, last %s" % (case["name"], x.size, nz.size,
          len(range(case["start"] + case["min_obs"], case["end"] - case["border_trim"], a["stride"])), nz[-1] if nz.size else None))
for case in ASSERT_CASES:
    x = signal_of(case)
    a = case["args"]
    try:
        c_llr_trace(x, case["start"], case["end"], case["min_obs"], case["border_trim"], a["stride"], a["adapter_early_stopping"],
                    a["adapter_early_stop_window"], a["adapter_early_stop_stride"], a["polya_early_stopping"],
                    a["polya_early_stop_window"], a["polya_early_stop_stride"], 0)
        raise SystemExit("%s: the reference did not assert" % case["name"])
    except AssertionError:
        print("%-36s AssertionError (as expected)" % case["name"])
np.savez_compressed(os.path.join(os.path.dirname(HERE), "tests", "golden", "c_llr_trace.npz"), **out)
print("wrote tests/golden/c_llr_trace.npz")

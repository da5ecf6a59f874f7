"""The reference's native module ``adapted.detect._c_llr`` (Cython, adapted/detect/_c_llr.pyx; imported at
adapted/detect/llr.py:18 as ``_gains, c_llr_trace, c_llr_trace_gains``) on the HIP library: same names, arguments and return
values, computed by ``adp_c_llr_trace`` (adapted_amd/csrc/trace_api.h).

A call per read pays a launch and two copies per trace -- the product path does not come through here (``adp_detect_llr`` runs
a whole minibatch in fused passes); these functions are the API-level drop-in, and ``c_llr_trace_batch`` is the form to use
for many reads.  There is no CPU path: without the HIP library every function raises ``HipLibraryError``.

Differences from the Cython module: values agree to ~1e-15 relative (the device logarithm is correctly rounded, glibc's is
not: DESIGN.md section 4); index arguments outside ``0 <= start <= end <= len(signal)`` raise ``ValueError`` where the
reference reads outside its arrays (bounds checks are off there, _c_llr.pyx:21).
"""
from __future__ import annotations

import numpy as np

from .. import lib

_ENGINE = None


def _engine(device: int = 0):
    global _ENGINE
    if _ENGINE is None or _ENGINE.device != device:
        from ..config import get_chemistry_specific_config

        spc = get_chemistry_specific_config("RNA004")  # (the trace API reads nothing from the configuration)
        _ENGINE = lib.Engine(spc, 1, spc.sig_preload_size, device=device)
    return _ENGINE


def _args(min_obs, border_trim, stride, adapter_early_stopping, adapter_early_stop_window, adapter_early_stop_stride,
          polya_early_stopping, polya_early_stop_window, polya_early_stop_stride) -> "lib.AdpTraceArgs":
    # the reference's asserts (_c_llr.pyx:102, :137-138)
    if polya_early_stopping > 0:
        assert adapter_early_stop_stride % stride == 0
        assert polya_early_stop_stride % stride == 0
    elif adapter_early_stopping > 0:
        assert adapter_early_stop_stride % stride == 0
    return lib.AdpTraceArgs(int(min_obs), int(border_trim), int(stride), int(adapter_early_stopping), int(adapter_early_stop_window),
                            int(adapter_early_stop_stride), int(polya_early_stopping), int(polya_early_stop_window),
                            int(polya_early_stop_stride))


def _check_range(start, end, n):
    if not (0 <= start <= end <= n):
        raise ValueError("need 0 <= start <= end <= len(signal) (got start=%d, end=%d, len=%d)" % (start, end, n))


def c_llr_trace_batch(raw_signals, lens, starts, ends, min_obs, border_trim, stride=1, adapter_early_stopping=0,
                      adapter_early_stop_window=500, adapter_early_stop_stride=100, polya_early_stopping=0,
                      polya_early_stop_window=50, polya_early_stop_stride=10, return_c_c2=0, sums=None, device: int = 0):
    """`c_llr_trace` for many reads in one call: raw_signals float64 [n, L] (read r valid in [0, lens[r])), per-read starts /
    ends -> gains float64 [n, L] (zeros beyond lens[r]), with ``return_c_c2`` also the cumulative sums.  ``sums=(c, c2)``:
    the form of `c_llr_trace_gains`."""
    a = _args(min_obs, border_trim, stride, adapter_early_stopping, adapter_early_stop_window, adapter_early_stop_stride,
              polya_early_stopping, polya_early_stop_window, polya_early_stop_stride)
    lens = np.asarray(lens, dtype=np.int64)
    for s, e, n in zip(np.asarray(starts).tolist(), np.asarray(ends).tolist(), lens.tolist()):
        _check_range(s, e, n)
    return _engine(device).c_llr_trace(raw_signals, lens, starts, ends, a, sums=sums, return_c_c2=bool(return_c_c2))


def c_llr_trace(raw_signal, start, end, min_obs, border_trim, stride=1, adapter_early_stopping=0, adapter_early_stop_window=500,
                adapter_early_stop_stride=100, polya_early_stopping=0, polya_early_stop_window=50, polya_early_stop_stride=10,
                return_c_c2=0):
    """_c_llr.pyx:202-236 -> gain, or (gain, c, c2) with ``return_c_c2``"""
    x = np.ascontiguousarray(raw_signal, dtype=np.float64).reshape(1, -1)
    n = x.shape[1]
    if n == 0:
        z = np.zeros(0)
        return (z, z.copy(), z.copy()) if return_c_c2 else z
    res = c_llr_trace_batch(x, [n], [start], [end], min_obs, border_trim, stride, adapter_early_stopping, adapter_early_stop_window,
                            adapter_early_stop_stride, polya_early_stopping, polya_early_stop_window, polya_early_stop_stride,
                            return_c_c2)
    if return_c_c2:
        return res[0][0], res[1][0], res[2][0]
    return res[0]


def c_llr_trace_gains(c, c2, start, end, min_obs, border_trim, stride=1, adapter_early_stopping=0, adapter_early_stop_window=500,
                      adapter_early_stop_stride=100, polya_early_stopping=0, polya_early_stop_window=50, polya_early_stop_stride=10):
    """_c_llr.pyx:176-199: the gains from given cumulative sums"""
    c = np.ascontiguousarray(c, dtype=np.float64).reshape(1, -1)
    c2 = np.ascontiguousarray(c2, dtype=np.float64).reshape(1, -1)
    n = c.shape[1]
    if n == 0:
        return np.zeros(0)
    return c_llr_trace_batch(None, [n], [start], [end], min_obs, border_trim, stride, adapter_early_stopping, adapter_early_stop_window,
                             adapter_early_stop_stride, polya_early_stopping, polya_early_stop_window, polya_early_stop_stride,
                             sums=(c, c2))[0]


def _gains(start, end, c, c2, offset_head, offset_tail, stride=1):
    """_c_llr.pyx:67-88"""
    return c_llr_trace_gains(c, c2, start, end, offset_head, offset_tail, stride)

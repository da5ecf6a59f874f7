"""Host-side mirror of the reference's detect operators, backed by libadapted_hip.so.

Same names, argument meaning, return types and error behaviour as the reference's
adapted/detect/combined.py: ``combined_detect_llr2`` (:122-227), ``combined_detect_llr`` (:39-119), ``combined_detect_cnn``
(:230-309), ``combined_detect_start_peak`` (:312-355) and ``validate_boundaries`` (:358-631).
Each call treats its batch as ONE minibatch (the reference's batch-global normalisation,
adapted/detect/normalize.py:15-22).  Everything numeric runs on the GPU through the C ABI
(include/adapted_hip.h); there is no CPU fallback.
"""
from __future__ import annotations

import logging
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

from .. import lib
from ..container_types import Boundaries, DetectResults

from collections import OrderedDict

_ENGINES: "OrderedDict[Tuple, lib.Engine]" = OrderedDict()
_MAX_ENGINES = 6  # each holds streams, events and workspace: the cache is bounded, least recently used first out


def _cfg_key(spc) -> bytes:
    return bytes(lib.make_cfg(spc))


def get_engine(spc, n: int, m: int, device: int = 0) -> "lib.Engine":
    """Cached engine for (config, m, device) whose capacity covers n reads."""
    key = (_cfg_key(spc), int(m), int(device))
    eng = _ENGINES.get(key)
    if eng is None or eng.max_reads < n:
        # (a smaller engine is only dropped from the cache, not closed: a caller may still hold it; it closes itself
        # when the last reference goes)
        eng = lib.Engine(spc, max(int(n), 1), int(m), device=device)
        _ENGINES[key] = eng
    _ENGINES.move_to_end(key)
    while len(_ENGINES) > _MAX_ENGINES:
        _ENGINES.popitem(last=False)
    return eng


def release_engines():
    for e in _ENGINES.values():
        e.close()
    _ENGINES.clear()


def _as_batch(batch_of_signals, full_signal_lens):
    sig = np.ascontiguousarray(batch_of_signals, dtype=np.float32)
    if sig.ndim != 2:
        raise ValueError("batch_of_signals must be a 2-D float32 array [n_reads, preload]")
    lens = np.ascontiguousarray(full_signal_lens, dtype=np.int32).reshape(-1)
    if lens.size != sig.shape[0]:
        raise ValueError("full_signal_lens must have one entry per read")
    return sig, lens


def combined_detect_llr2(batch_of_signals: np.ndarray, full_signal_lens: np.ndarray, spc, device: int = 0,
                         with_start_peak: bool = False) -> List[DetectResults]:
    sig, lens = _as_batch(batch_of_signals, full_signal_lens)
    n, m = sig.shape
    if n == 0:
        return []
    eng = get_engine(spc, n, m, device)
    rows, mbs = eng.detect_llr_rows(sig, lens, n, n, with_start_peak=with_start_peak)
    if mbs[0] == lib.MB_MAD_ZERO:
        msg = "MAD normalization failed: scale is 0"
        logging.error(msg)
        raise ValueError(msg)
    if mbs[0] == lib.MB_EMPTY_TRACE:
        # the reference dies in np.argmin on a read whose pooled trace is empty (llr.py:136)
        raise ValueError("attempt to get argmin of an empty sequence")
    return lib.rows_to_results(rows, "llr", consume=True)


_EXC_TYPES = {9: TypeError, 10: TypeError, 11: ValueError, 12: ValueError, 13: ValueError, 14: ValueError}


def combined_detect_llr(calibrated_signal: np.ndarray, full_signal_len: int, spc, device: int = 0) -> DetectResults:
    """The single-read operator (reference adapted/detect/combined.py:39-119; no call site in the reference's CLI): the read
    is normalised by its own median / MAD, pooled from sample 0 (offset_head = 5 + min_obs_adapter // ds), and validated;
    errors propagate as in the reference (it has no try / except on this path).  At most sig_preload_size samples are
    looked at -- the reference's validation reads the whole array, so the signal has to be the preloaded part of the read."""
    sig = np.ascontiguousarray(calibrated_signal, dtype=np.float32).reshape(-1)
    m = int(spc.sig_preload_size)
    if sig.size > m:
        raise ValueError("combined_detect_llr: at most sig_preload_size = %d samples (the preloaded part of the read)" % m)
    if sig.size < min(int(full_signal_len), m):
        raise ValueError("combined_detect_llr: the signal is shorter than min(full_signal_len, sig_preload_size)")
    key = (_cfg_key(spc), m, int(device), "single")
    eng = _ENGINES.get(key)
    if eng is None:
        eng = lib.Engine(spc, 1, m, device=device, single_read_layout=True)
        _ENGINES[key] = eng
    row = np.full((1, m), np.nan, dtype=np.float32)
    row[0, : sig.size] = sig
    rows, mbs = eng.detect_llr_rows(row, np.array([full_signal_len], dtype=np.int32), 1, 1)
    if mbs[0] == lib.MB_MAD_ZERO:
        msg = "MAD normalization failed: scale is 0"
        logging.error(msg)
        raise ValueError(msg)
    if mbs[0] == lib.MB_EMPTY_TRACE:
        raise ValueError("attempt to get argmin of an empty sequence")
    fc = int(rows[0]["fail_code"])
    if 9 <= fc <= 14:  # raised inside validate_boundaries: not caught on this path
        raise _EXC_TYPES[fc](lib.fail_reason_of(rows[0]))
    return lib.rows_to_results(rows, "llr", consume=True)[0]


def combined_detect_start_peak(batch_of_signals: np.ndarray, full_signal_lens: np.ndarray, spc,
                               device: int = 0) -> List[DetectResults]:
    sig, lens = _as_batch(batch_of_signals, full_signal_lens)
    n, m = sig.shape
    if n == 0:
        return []
    eng = get_engine(spc, n, m, device)
    rows = eng.detect_start_peak_rows(sig, lens, n, n)
    res = lib.rows_to_results(rows, "start_peak", consume=True)
    return lib.open_pore_float_column(res)


def validate_boundaries(signal: np.ndarray, boundaries: Boundaries, spc, full_signal_len: int,
                        device: int = 0) -> DetectResults:
    """Single-read validator (the reference calls it with ``signal[:full_signal_len]``, i.e. with min(full_signal_len,
    sig_preload_size) samples).  The row is NaN-padded to ONE engine width per configuration (the preload size; longer
    arrays to the next multiple of 16 384), so that looping over reads of different lengths reuses one engine."""
    x = np.ascontiguousarray(signal, dtype=np.float32).reshape(-1)
    have = x.size
    m = int(spc.sig_preload_size) if have <= int(spc.sig_preload_size) else -(-have // 16384) * 16384
    sig = np.full((1, m), np.nan, dtype=np.float32)
    sig[0, :have] = x
    eng = get_engine(spc, 1, m, device)
    eff_len = int(full_signal_len) if have >= min(int(full_signal_len), m) else have  # (an array shorter than the read: what is there)
    topk = boundaries.polya_end_topk
    none = topk is None
    cands = [int(boundaries.polya_end or 0)] if none else [int(x) for x in np.asarray(topk).ravel()]
    b = np.zeros((1, 1 + max(1, len(cands))), dtype=np.int64)
    b[0, 0] = int(boundaries.adapter_end or 0)
    b[0, 1:1 + len(cands)] = cands
    if not none and len(cands) and cands[0] != int(boundaries.polya_end or 0):
        raise ValueError("polya_end_topk[0] must equal polya_end")
    rows = eng.validate_rows(sig, np.array([eff_len], dtype=np.int32), 1, b, topk_none=none)
    if eff_len != int(full_signal_len) and int(rows[0]["present"]) & 1:
        rows[0]["col"][0] = float(full_signal_len)  # signal_len reports the read's true length
    return lib.rows_to_results(rows, spc.primary_method, consume=True)[0]


def combined_detect_cnn(batch_of_signals: np.ndarray, full_signal_lens: np.ndarray, model, spc,
                        device: int = 0, conv: str = "hip") -> Union[List[DetectResults], DetectResults]:
    from . import cnn as _cnn

    return _cnn.combined_detect_cnn(batch_of_signals, full_signal_lens, model, spc, device=device, conv=conv)

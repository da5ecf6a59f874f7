"""Host-side mirror of the reference's detect operators, backed by libadapted_hip.so.

Same names, argument meaning, return types and error behaviour as the reference's
adapted/detect/combined.py: ``combined_detect_llr2`` (:122-227), ``combined_detect_cnn``
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

_ENGINES: Dict[Tuple, "lib.Engine"] = {}


def _cfg_key(spc) -> bytes:
    return bytes(lib.make_cfg(spc))


def get_engine(spc, n: int, m: int, device: int = 0) -> "lib.Engine":
    """Cached engine for (config, m, device) whose capacity covers n reads."""
    key = (_cfg_key(spc), int(m), int(device))
    eng = _ENGINES.get(key)
    if eng is None or eng.max_reads < n:
        if eng is not None:
            eng.close()
        eng = lib.Engine(spc, max(int(n), 1), int(m), device=device)
        _ENGINES[key] = eng
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
    return lib.rows_to_results(rows, "llr")


def combined_detect_start_peak(batch_of_signals: np.ndarray, full_signal_lens: np.ndarray, spc,
                               device: int = 0) -> List[DetectResults]:
    sig, lens = _as_batch(batch_of_signals, full_signal_lens)
    n, m = sig.shape
    if n == 0:
        return []
    eng = get_engine(spc, n, m, device)
    rows = eng.detect_start_peak_rows(sig, lens, n, n)
    return lib.rows_to_results(rows, "start_peak")


def validate_boundaries(signal: np.ndarray, boundaries: Boundaries, spc, full_signal_len: int,
                        device: int = 0) -> DetectResults:
    """Single-read validator (the reference calls it with ``signal[:full_signal_len]``)."""
    sig = np.ascontiguousarray(signal, dtype=np.float32).reshape(1, -1)
    m = sig.shape[1]
    eng = get_engine(spc, 1, m, device)
    topk = boundaries.polya_end_topk
    none = topk is None
    cands = [int(boundaries.polya_end or 0)] if none else [int(x) for x in np.asarray(topk).ravel()]
    b = np.zeros((1, 1 + max(1, len(cands))), dtype=np.int64)
    b[0, 0] = int(boundaries.adapter_end or 0)
    b[0, 1:1 + len(cands)] = cands
    if not none and len(cands) and cands[0] != int(boundaries.polya_end or 0):
        raise ValueError("polya_end_topk[0] must equal polya_end")
    rows = eng.validate_rows(sig, np.array([full_signal_len], dtype=np.int32), 1, b, topk_none=none)
    return lib.rows_to_results(rows, spc.primary_method)[0]


def combined_detect_cnn(batch_of_signals: np.ndarray, full_signal_lens: np.ndarray, model, spc,
                        device: int = 0) -> Union[List[DetectResults], DetectResults]:
    from . import cnn as _cnn

    return _cnn.combined_detect_cnn(batch_of_signals, full_signal_lens, model, spc, device=device)

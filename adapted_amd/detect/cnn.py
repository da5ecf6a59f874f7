"""CNN boundary head (the reference's adapted/detect/cnn.py) on MI355X.

Everything numeric is the HIP library: ``prepare_data`` (pool + per-read median / MAD normalisation, C1), the conv net
itself (C2: hand-written, the two 64 -> 64 layers on the float32 matrix cores -- adapted_amd/csrc/cnn_conv.h), ``cnn_predict``
(C3: both arg-maxes, scipy's find_peaks(distance=5) on the flattened scores, the per-read top-k and the reference's
row-compaction quirk -- adapted_amd/csrc/cnn_topk.h) and the candidate validation loop (V1 with k candidates).  The product
path (``detect_rows`` / ``detect_rows_device``) is ONE library call, ``adp_detect_cnn``, plus the rare short-read fallback.
PyTorch is optional: ``load_cnn_model`` returns the reference's ``nn.Sequential`` (state-dict compatible), and
``conv="torch"`` runs the conv stack through PyTorch-ROCm / MIOpen instead (kept as the float32 cross-check of C2; the
hand-written stack is 3.4x faster at the 200 k window).

reference: BoundariesCNN :16-52, load_cnn_model :55-67, prepare_data :70-82, cnn_score :85-98,
cnn_predict :101-160, cnn_detect :165-182, cnn_detect_boundaries :185-201 (adapted/detect/cnn.py);
driver combined_detect_cnn adapted/detect/combined.py:230-309.
"""
from __future__ import annotations

import os
import warnings
from typing import List, Optional, Union

import numpy as np

from .. import lib
from ..container_types import Boundaries, DetectResults

SCORE_EXCL = -5.0
MODEL_DOWNSCALE = {"rna004_130bps@v0.2.4.pth": 10, "rna004_130bps@v0.2.4.npz": 10}
_MODEL_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models")


def _torch():
    import torch

    return torch


def BoundariesCNN(channels: int = 64, kernel_size: int = 7):
    """conv(1->C, k, stride k//2) - ReLU - conv(C->C) - ReLU - conv(C->C) - ReLU - convT(C->2, stride k//2);
    an ``nn.Sequential`` so that state-dict keys ("0.weight", "2.weight", ...) match the reference's."""
    nn = _torch().nn
    s, p = kernel_size // 2, kernel_size // 2
    return nn.Sequential(
        nn.Conv1d(1, channels, kernel_size, stride=s, padding=p), nn.ReLU(),
        nn.Conv1d(channels, channels, kernel_size, padding=p), nn.ReLU(),
        nn.Conv1d(channels, channels, kernel_size, padding=p), nn.ReLU(),
        nn.ConvTranspose1d(channels, 2, kernel_size, stride=s, padding=p),
    )


def _resolve(path: str) -> str:
    cands = [path]
    base = os.path.basename(path)
    stem = os.path.splitext(base)[0]
    for d in filter(None, [os.environ.get("ADAPTED_MODEL_DIR"), _MODEL_DIR]):
        cands += [os.path.join(d, base), os.path.join(d, stem + ".npz"), os.path.join(d, stem + ".pth")]
    for c in cands:
        if os.path.isfile(c):
            return c
    raise FileNotFoundError("Model weights not found at %s (searched %s)" % (path, ", ".join(cands)))


def load_cnn_model(path: str, device: Optional[Union[int, str]] = None):
    """Load weights from a ``.pth`` state dict (as shipped by the reference) or a ``.npz`` with the
    same keys.  The model is moved to the GPU and put in eval mode."""
    torch = _torch()
    f = _resolve(path)
    model = BoundariesCNN()
    if f.endswith(".npz"):
        z = np.load(f)
        sd = {k: torch.from_numpy(np.ascontiguousarray(z[k])) for k in z.files if k[0].isdigit()}
    else:
        sd = torch.load(f, weights_only=True, map_location="cpu")
    model.load_state_dict(sd)
    model.eval()
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else "cpu"
    return model.to(torch.device("cuda", device) if isinstance(device, int) else torch.device(device))


def load_cnn_weights(path: str) -> dict:
    """The state dict as float32 numpy arrays, without PyTorch for ``.npz`` files (keys "0.weight" ... "6.bias")."""
    f = _resolve(path)
    if f.endswith(".npz"):
        z = np.load(f)
        return {k: np.ascontiguousarray(z[k], dtype=np.float32) for k in z.files if k[0].isdigit()}
    sd = _torch().load(f, weights_only=True, map_location="cpu")
    return {k: v.numpy() for k, v in sd.items()}


def _state_of(model, spc=None) -> dict:
    """model: the nn.Sequential of load_cnn_model, a state dict / dict of arrays, or None (the config's model_name)"""
    if model is None:
        return load_cnn_weights(spc.cnn_boundaries.model_name)
    if isinstance(model, dict):
        return model
    return model.state_dict()


def ensure_weights(eng, model, spc=None):
    """hand the model's weights to the engine once (adp_cnn_set_weights)"""
    key = id(model) if model is not None else ("cfg", spc.cnn_boundaries.model_name)
    if getattr(eng, "_cnn_key", None) != key:
        eng.cnn_set_weights(_state_of(model, spc))
        eng._cnn_key = key
        eng._cnn_model_ref = model  # (keeps id(model) from being reused while the engine remembers it)


def _model_device(model):
    return next(model.parameters()).device


def prepare_data(batch_of_signals: np.ndarray, core_params, spc=None, engine=None):
    """float32 [N, 1, Lc] on the GPU (C1, computed by the HIP library)."""
    torch = _torch()
    sig = np.ascontiguousarray(batch_of_signals, dtype=np.float32)
    n, m = sig.shape
    if engine is None:
        from .combined import get_engine

        engine = get_engine(spc, n, m, torch.cuda.current_device())
    off, ds = int(core_params.min_obs_adapter), int(core_params.downscale_factor)
    Lc = (m - off + ds - 1) // ds
    out = torch.empty((n, 1, Lc), dtype=torch.float32, device=torch.device("cuda", engine.device))
    engine.cnn_prepare(sig, n, out.data_ptr())
    return out


def cnn_score(batch_of_prepared_signals, model, engine=None):
    """float32 [N, 2, Lo].  engine: the hand-written conv stack of the HIP library (adp_cnn_forward); without one the
    PyTorch-ROCm modules of ``model`` (the float32 cross-check)."""
    torch = _torch()
    if engine is None:
        if len(model.state_dict()) == 0:
            raise ValueError("Model weights were not loaded")
        x = batch_of_prepared_signals
        # in slices of <= 2 GiB of 64-channel activations: MIOpen returns wrong scores for a batch beyond 4 GiB
        per_read = 64 * 4 * ((int(x.shape[-1]) - 1) // 3 + 1)
        step = max(1, (2 << 30) // per_read)
        with torch.no_grad():
            if x.shape[0] <= step:
                return model(x)
            return torch.cat([model(x[s0:s0 + step]) for s0 in range(0, x.shape[0], step)], dim=0)
    x = batch_of_prepared_signals.contiguous()
    n, _, Lc = x.shape
    Lo = 3 * ((Lc - 1) // 3 + 1) - 2
    ensure_weights(engine, model)
    out = torch.empty((n, 2, Lo), dtype=torch.float32, device=x.device)
    torch.cuda.current_stream(x.device).synchronize()  # the engine works on its own stream
    engine.cnn_forward(x.data_ptr(), n, Lc, out.data_ptr())
    return out


def cnn_predict(batch_of_prepared_signals, model, params, core_params, engine=None, conv: str = "hip") -> np.ndarray:
    """int [N, 1 + max(k, 1)] pooled indices: adapter position and the k poly(A) candidates (C3 on the device)."""
    torch = _torch()
    if engine is None:
        raise lib.HipLibraryError("cnn_predict runs on the HIP engine (no CPU path)")
    if int(params.polya_cand_k) != int(engine.cfg.polya_cand_k):
        raise ValueError("params.polya_cand_k differs from the engine's configuration")
    scores = cnn_score(batch_of_prepared_signals, model, engine=engine if conv == "hip" else None).contiguous()
    n, _, Lo = scores.shape
    torch.cuda.current_stream(scores.device).synchronize()
    b = engine.cnn_predict(scores.data_ptr(), n, n, Lo)
    off, ds = int(core_params.min_obs_adapter), int(core_params.downscale_factor)
    return np.where(b == 0, 0, (b - off) // ds)  # (a sample position of 0 stands for index 0, cnn.py:173-179)


def cnn_detect(batch_of_signals: np.ndarray, model, params, core_params, spc=None, engine=None, conv: str = "hip") -> np.ndarray:
    prepared = prepare_data(batch_of_signals, core_params, spc=spc, engine=engine)
    if engine is None:
        from .combined import get_engine

        engine = get_engine(spc, prepared.shape[0], np.asarray(batch_of_signals).shape[1], prepared.device.index)
    preds = (cnn_predict(prepared, model, params, core_params, engine=engine, conv=conv) * core_params.downscale_factor
             + core_params.min_obs_adapter).astype(int)
    preds[preds == core_params.min_obs_adapter] = 0  # where the prediction was zero, set back to zero
    return preds


def cnn_detect_boundaries(batch_of_signals: np.ndarray, model, params, core_params, spc=None) -> List[Boundaries]:
    preds = cnn_detect(batch_of_signals, model, params, core_params, spc=spc)
    return [Boundaries(adapter_start=0, adapter_end=p[0], polya_end=p[1], polya_end_topk=p[1:]) for p in preds]


def _need_fallback(rows, bounds, lens, spc):
    """C4 "hail mary" for short reads (combined.py:251-301): which reads take it"""
    ae, pe = bounds[:, 0], bounds[:, 1]
    return np.flatnonzero((rows["success"] == 0) & ~((rows["fail_code"] >= 9) & (rows["fail_code"] <= 14)) & (ae > 0) & (pe > 0)
                          & (pe - ae > 1000) & (np.asarray(lens).astype(np.int64) < 2 * spc.core.max_obs_adapter))


def detect_rows_device(eng, dsig: int, dlen: int, n: int, lens_host: np.ndarray, model, spc, minibatch: Optional[int] = None) -> np.ndarray:
    """combined_detect_cnn over a DEVICE-resident batch (pointers) -> adp_row[]; ONE library call (adp_detect_cnn); the
    short-read fallback needs the host copy of the few affected reads only.  minibatch: reads per call of the reference
    (its find_peaks and row compaction work on one minibatch); default: the whole batch."""
    ensure_weights(eng, model, spc)
    m = eng.m
    rows, bounds = eng.detect_cnn_rows(dsig, dlen, n, minibatch or n, device_ptrs=True)
    if spc.cnn_boundaries.fallback_to_llr_short_reads:
        idx = _need_fallback(rows, bounds, lens_host, spc)
        if idx.size:
            sub = np.zeros((idx.size, m), dtype=np.float32)
            for j, i in enumerate(idx):
                eng.d2h(sub[j], dsig + int(i) * m * 4)
            _apply_fallback(eng, rows, idx, sub, np.asarray(lens_host)[idx], bounds, spc)
    return rows


def _apply_fallback(eng, rows, idx, sig_sub, lens_sub, bounds, spc):
    new_pe, status = eng.llr_refine_polya(sig_sub, lens_sub, idx.size, bounds[idx, :2])
    for j, i in enumerate(idx):
        if status[j] != 0:  # the reference raised inside its per-read try block
            rows[i] = lib.empty_rows(1)[0]
            rows[i]["fail_code"] = status[j]
    redo = [j for j in range(idx.size) if status[j] == 0 and new_pe[j] > 0]
    if redo:
        ii = idx[redo]
        b2 = np.stack([bounds[ii, 0], new_pe[redo]], axis=1).astype(np.int64)
        rows[ii] = eng.validate_rows(sig_sub[redo], lens_sub[redo], len(ii), b2)


def detect_rows(eng, sig: np.ndarray, lens: np.ndarray, model, spc, conv: str = "hip") -> np.ndarray:
    """combined_detect_cnn over one batch -> adp_row[] (reference adapted/detect/combined.py:230-309).
    conv = "torch": the conv stack through PyTorch-ROCm instead of the library's own (cross-check)."""
    n = sig.shape[0]
    if int(spc.cnn_boundaries.polya_cand_k) < 1:
        raise ValueError("polya_cand_k must be >= 1")
    if conv == "hip":
        ensure_weights(eng, model, spc)
        rows, bounds = eng.detect_cnn_rows(sig, lens, n, n)
    else:
        preds = cnn_detect(sig, model, spc.cnn_boundaries, spc.core, spc=spc, engine=eng, conv=conv)
        bounds = np.ascontiguousarray(preds, dtype=np.int64)
        rows = eng.validate_rows(sig, lens, n, bounds)
    if spc.cnn_boundaries.fallback_to_llr_short_reads:
        idx = _need_fallback(rows, bounds, lens, spc)
        if idx.size:
            _apply_fallback(eng, rows, idx, sig[idx], lens[idx], bounds, spc)
    return rows


def combined_detect_cnn(batch_of_signals: np.ndarray, full_signal_lens: np.ndarray, model, spc,
                        device: int = 0, conv: str = "hip") -> Union[List[DetectResults], DetectResults]:
    """model: the nn.Sequential of load_cnn_model, a dict of weight arrays (load_cnn_weights), or None (the config's model)"""
    from .combined import _as_batch, get_engine

    sig, lens = _as_batch(batch_of_signals, full_signal_lens)
    n, m = sig.shape
    eng = get_engine(spc, n, m, device)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        rows = detect_rows(eng, sig, lens, model, spc, conv=conv)
    res = lib.rows_to_results(rows, "cnn", consume=True)
    return res if len(res) > 1 else res[0]

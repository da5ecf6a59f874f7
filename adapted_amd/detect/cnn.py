"""CNN boundary head (the reference's adapted/detect/cnn.py) on MI355X.

Division of labour, as BASELINE.json:north_star prescribes: the small 1-D conv net itself
runs in PyTorch-ROCm (float32; bf16/fp16 would break score parity); everything around it is
the HIP library: ``prepare_data`` (pool + per-read median/MAD normalisation, C1) and the
candidate validation loop (V1 with k candidates).  The top-k candidate extraction of
``cnn_predict`` (C3) is index bookkeeping on one float per pooled sample and stays on the
host in numpy/scipy exactly as the reference has it, including its row-misalignment quirk
when a read has no peak (see ``_topk_candidates``).

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


def _model_device(model):
    return next(model.parameters()).device


def prepare_data(batch_of_signals: np.ndarray, core_params, spc=None, engine=None):
    """float32 [N, 1, Lc] on the model's GPU (C1, computed by the HIP library)."""
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


def cnn_score(batch_of_prepared_signals, model):
    if len(model.state_dict()) == 0:
        raise ValueError("Model weights were not loaded")
    torch = _torch()
    with torch.no_grad():
        return model(batch_of_prepared_signals)


def _topk_candidates(ch1: np.ndarray, k: int) -> np.ndarray:
    """Per-read top-k poly(A) candidates from the masked channel-1 scores (reference cnn.py:136-160).
    The reference groups the peaks of the FLATTENED array by read and writes group i into row i;
    a read without any peak therefore shifts all later groups up by one row.  Kept as is."""
    from scipy.signal import find_peaks

    n, Lo = ch1.shape
    flat = ch1.reshape(-1)
    cand, _ = find_peaks(flat, distance=5)
    heights = flat[cand]
    read_idx = cand // Lo
    order = np.lexsort((-heights, read_idx))
    cand = cand[order]
    groups = np.split(np.mod(cand, Lo), np.where(np.diff(read_idx) != 0)[0] + 1)
    out = np.zeros((n, k), dtype=np.int64)
    for i, peaks in enumerate(groups):
        out[i, : len(peaks)] = peaks[:k]
    return out


def cnn_predict(batch_of_prepared_signals, model, params, core_params, engine=None) -> np.ndarray:
    """engine: with a HIP engine at hand the top-k candidate extraction runs on the device (adp_cnn_topk); batches
    holding a case only scipy's formulation settles (exact ties, plateaus: the kernel reports them) take the host
    formulation below, which is the reference's own."""
    torch = _torch()
    scores = cnn_score(batch_of_prepared_signals, model)
    n, _, Lo = scores.shape
    na = (core_params.max_obs_adapter - core_params.min_obs_adapter) // core_params.downscale_factor
    adapter_pos = torch.argmax(scores[:, 0, :na], dim=1)
    k = int(params.polya_cand_k)
    pos = torch.arange(Lo, device=scores.device)[None, :]
    ch1 = scores[:, 1, :]
    if k >= 1:
        ch1 = torch.where(pos < adapter_pos[:, None], torch.full_like(ch1, SCORE_EXCL), ch1)
        polya_pos = torch.argmax(ch1, dim=1)
    else:
        polya_pos = torch.zeros(n, dtype=torch.int64, device=scores.device)
    a = adapter_pos.cpu().numpy().astype(np.int64)
    if k > 1:
        if engine is not None and scores.is_cuda and scores.is_contiguous():
            torch.cuda.current_stream(scores.device).synchronize()  # the engine works on its own stream
            apos = adapter_pos.to(torch.int64).contiguous()
            ppos = polya_pos.to(torch.int64).contiguous()
            cand, cnt, flag = engine.cnn_topk(scores.data_ptr(), apos.data_ptr(), ppos.data_ptr(), n, Lo, k)
            if flag == 0:
                # the reference writes the group of the i-th read THAT HAS PEAKS into row i (cnn.py:150-158)
                topk = np.zeros((n, k), dtype=np.int64)
                nz = np.flatnonzero(cnt > 0)
                topk[: nz.size] = cand[nz]
                return np.column_stack((a[:, None], topk))
        ch1 = torch.where(pos > polya_pos[:, None], torch.full_like(ch1, SCORE_EXCL), ch1)
        topk = _topk_candidates(ch1.cpu().numpy(), k)
        return np.column_stack((a[:, None], topk))
    return np.column_stack((a, polya_pos.cpu().numpy().astype(np.int64)))


def cnn_detect(batch_of_signals: np.ndarray, model, params, core_params, spc=None, engine=None) -> np.ndarray:
    prepared = prepare_data(batch_of_signals, core_params, spc=spc, engine=engine)
    prepared = prepared.to(_model_device(model))
    preds = (cnn_predict(prepared, model, params, core_params, engine=engine) * core_params.downscale_factor
             + core_params.min_obs_adapter).astype(int)
    preds[preds == core_params.min_obs_adapter] = 0  # where the prediction was zero, set back to zero
    return preds


def cnn_detect_boundaries(batch_of_signals: np.ndarray, model, params, core_params, spc=None) -> List[Boundaries]:
    preds = cnn_detect(batch_of_signals, model, params, core_params, spc=spc)
    return [Boundaries(adapter_start=0, adapter_end=p[0], polya_end=p[1], polya_end_topk=p[1:]) for p in preds]


def detect_rows_device(eng, dsig: int, dlen: int, n: int, lens_host: np.ndarray, model, spc) -> np.ndarray:
    """combined_detect_cnn over a DEVICE-resident batch (pointers) -> adp_row[]; the short-read fallback needs
    the host copy of the few affected reads only."""
    torch = _torch()
    core = spc.core
    m = eng.m
    Lc = (m - core.min_obs_adapter + core.downscale_factor - 1) // core.downscale_factor
    x = torch.empty((n, 1, Lc), dtype=torch.float32, device=torch.device("cuda", eng.device))
    eng.cnn_prepare(dsig, n, x.data_ptr(), device_ptrs=True)
    preds = (cnn_predict(x.to(_model_device(model)), model, spc.cnn_boundaries, core, engine=eng) * core.downscale_factor
             + core.min_obs_adapter).astype(int)
    preds[preds == core.min_obs_adapter] = 0
    bounds = np.ascontiguousarray(preds, dtype=np.int64)
    rows = eng.validate_rows(dsig, dlen, n, bounds, device_ptrs=True)
    if spc.cnn_boundaries.fallback_to_llr_short_reads:
        ae, pe = bounds[:, 0], bounds[:, 1]
        need = ((rows["success"] == 0) & ~((rows["fail_code"] >= 9) & (rows["fail_code"] <= 14)) & (ae > 0) & (pe > 0) & (pe - ae > 1000)
                & (lens_host.astype(np.int64) < 2 * core.max_obs_adapter))
        idx = np.flatnonzero(need)
        if idx.size:
            sub = np.zeros((idx.size, m), dtype=np.float32)
            for j, i in enumerate(idx):
                eng.d2h(sub[j], dsig + int(i) * m * 4)
            _apply_fallback(eng, rows, idx, sub, lens_host[idx], bounds, spc)
    return rows


def _apply_fallback(eng, rows, idx, sig_sub, lens_sub, bounds, spc):
    new_pe, status = eng.llr_refine_polya(sig_sub, lens_sub, idx.size, bounds[idx, :2])
    for j, i in enumerate(idx):
        if status[j] != 0:  # the reference raised inside its per-read try block
            rows[i] = np.zeros(1, dtype=lib.ROW_DTYPE)[0]
            rows[i]["n_cand"] = -1
            rows[i]["n_open_pores"] = -1
            rows[i]["fail_code"] = status[j]
    redo = [j for j in range(idx.size) if status[j] == 0 and new_pe[j] > 0]
    if redo:
        ii = idx[redo]
        b2 = np.stack([bounds[ii, 0], new_pe[redo]], axis=1).astype(np.int64)
        rows[ii] = eng.validate_rows(sig_sub[redo], lens_sub[redo], len(ii), b2)


def detect_rows(eng, sig: np.ndarray, lens: np.ndarray, model, spc) -> np.ndarray:
    """combined_detect_cnn over one batch -> adp_row[] (reference adapted/detect/combined.py:230-309)."""
    n = sig.shape[0]
    preds = cnn_detect(sig, model, spc.cnn_boundaries, spc.core, spc=spc, engine=eng)
    if preds.shape[1] < 2:
        raise ValueError("polya_cand_k must be >= 1")
    bounds = np.ascontiguousarray(preds, dtype=np.int64)
    rows = eng.validate_rows(sig, lens, n, bounds)
    if spc.cnn_boundaries.fallback_to_llr_short_reads:
        # C4 "hail mary" for short reads (combined.py:251-301)
        ae, pe = bounds[:, 0], bounds[:, 1]
        need = ((rows["success"] == 0) & ~((rows["fail_code"] >= 9) & (rows["fail_code"] <= 14)) & (ae > 0) & (pe > 0) & (pe - ae > 1000)
                & (lens.astype(np.int64) < 2 * spc.core.max_obs_adapter))
        idx = np.flatnonzero(need)
        if idx.size:
            _apply_fallback(eng, rows, idx, sig[idx], lens[idx], bounds, spc)
    return rows


def combined_detect_cnn(batch_of_signals: np.ndarray, full_signal_lens: np.ndarray, model, spc,
                        device: int = 0) -> Union[List[DetectResults], DetectResults]:
    from .combined import _as_batch, get_engine

    sig, lens = _as_batch(batch_of_signals, full_signal_lens)
    n, m = sig.shape
    eng = get_engine(spc, n, m, device)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        rows = detect_rows(eng, sig, lens, model, spc)
    res = lib.rows_to_results(rows, "cnn")
    return res if len(res) > 1 else res[0]

"""ctypes front-end of the CPU oracle (oracle/adapted_oracle.c).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package never does.  ``detect_llr`` / ``detect_start_peak`` return
plain dict rows keyed like the reference's ``DetectResults`` fields
(reference adapted/container_types.py:23-92) so tests can compare them with the golden
rows made by the real reference and with the HIP path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

COLS = ["signal_len", "preloaded", "adapter_start", "adapter_end", "adapter_len", "adapter_mean",
        "adapter_std", "adapter_med", "adapter_mad", "polya_start", "polya_end", "polya_len",
        "polya_mean", "polya_std", "polya_med", "polya_mad", "rna_preloaded_start",
        "rna_preloaded_len", "rna_preloaded_mean", "rna_preloaded_std", "rna_preloaded_med",
        "rna_preloaded_mad", "start_peak_idx", "start_peak_pa", "start_peak_next_max_idx",
        "start_peak_next_max_pa", "start_peak_open_pore_idx", "adapter_rna_median_shift",
        "PRIMARY_adapter_end", "PRIMARY_polya_end", "mvs_detect_mean_at_loc", "mvs_detect_var_at_loc",
        "mvs_detect_polya_med", "mvs_detect_polya_local_range", "mvs_detect_med_shift",
        "real_adapter_mean_start", "real_adapter_mean_end", "real_adapter_local_range", "mvs_adapter_end"]
INT_COLS = {"signal_len", "preloaded", "adapter_start", "adapter_end", "adapter_len", "polya_start",
            "polya_end", "polya_len", "rna_preloaded_start", "rna_preloaded_len", "start_peak_idx",
            "start_peak_next_max_idx", "start_peak_open_pore_idx", "PRIMARY_adapter_end",
            "PRIMARY_polya_end", "mvs_adapter_end"}
MAX_CAND, MAX_OP = 16, 4096

ROW_DTYPE = np.dtype([("col", "<f8", (len(COLS),)), ("present", "<u8"), ("success", "<i4"),
                      ("fail_code", "<i4"), ("mvs_fail_mask", "<i4"), ("start_peak_type", "<i4"),
                      ("n_cand", "<i4"), ("n_open_pores", "<i4"), ("cand", "<i8", (MAX_CAND,)),
                      ("open_pores", "<i4", (MAX_OP,))])

SP_DTYPE = np.dtype([("valid", "<i4"), ("flagged_type", "<i4"), ("has_open_pore", "<i4"), ("_pad", "<i4"),
                     ("start_peak_idx", "<i8"), ("next_greater_idx", "<i8"), ("open_pore_idx", "<i8"),
                     ("start_peak_pa", "<f4"), ("next_greater_pa", "<f4")])

FAIL_STR = {0: None, 1: "No adapter detected (primary)", 2: "adapter MAD check failed",
            3: "Open pore too close to boundary", 4: "Real signal check failed",
            5: "No polya detected (primary)", 6: "MVS polya check failed: not enough signal",
            7: "MVS polya check failed: ", 8: "Median shift check failed",
            9: "'NoneType' object is not iterable",
            10: "slice indices must be integers or None or have an __index__ method",
            11: "Moving window (=%d) must between 1 and %d, inclusive",
            12: "pA_mean_range is not specified",
            13: "attempt to get argmin of an empty sequence",
            14: "MAD normalization failed: scale is 0",
            15: "No adapter detected in range (mvs_detect)"}
SP_TYPES = {0: None, 1: "open pore in adapter", 2: "potential concatemer adapter-only read"}
PRIMARY = {"llr": 0, "cnn": 1, "start_peak": 2}


class Cfg(C.Structure):
    _fields_ = [
        ("min_obs_adapter", C.c_int32), ("max_obs_adapter", C.c_int32), ("min_obs_polya", C.c_int32),
        ("downscale_factor", C.c_int32), ("max_obs_trace", C.c_int32),
        ("sig_norm_outlier_thresh", C.c_double),
        ("adapter_peak_prominence", C.c_double), ("adapter_peak_rel_height", C.c_double),
        ("adapter_peak_width", C.c_int32), ("_pad0", C.c_int32),
        ("mvs_detect_check", C.c_int32), ("mvs_detect_overwrite", C.c_int32), ("search_window", C.c_int32),
        ("pA_mean_window", C.c_int32), ("pA_var_window", C.c_int32), ("median_shift_window", C.c_int32),
        ("polyA_window", C.c_int32), ("_pad1", C.c_int32),
        ("pA_mean_range", C.c_double * 2), ("pA_var_range", C.c_double * 2),
        ("median_shift_range", C.c_double * 2), ("polyA_med_range", C.c_double * 2),
        ("polyA_local_range", C.c_double * 2), ("pA_mean_adapter_med_scale_range", C.c_double * 2),
        ("detect_open_pores", C.c_int32), ("real_signal_check", C.c_int32), ("mean_window", C.c_int32),
        ("max_obs_local_range", C.c_int32),
        ("mean_start_range", C.c_double * 2), ("mean_end_range", C.c_double * 2),
        ("local_range", C.c_double * 2), ("adapter_mad_range", C.c_double * 2),
        ("detect_med_shift", C.c_int32), ("med_shift_window", C.c_int32), ("med_shift_range", C.c_double * 2),
        ("sp_downscale_factor", C.c_int32), ("start_peak_max_idx", C.c_int32), ("sp_offset1", C.c_int32),
        ("sp_offset2", C.c_int32), ("open_pore_pa", C.c_double),
        ("polya_cand_k", C.c_int32), ("fallback_to_llr_short_reads", C.c_int32),
        ("primary_method", C.c_int32), ("_pad2", C.c_int32),
    ]


def build(force=False):
    so = os.path.join(_HERE, "_build", "liboracle.so")
    src = os.path.join(_HERE, "adapted_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        assert _LIB.orc_sizeof_cfg() == C.sizeof(Cfg), (_LIB.orc_sizeof_cfg(), C.sizeof(Cfg))
        assert _LIB.orc_sizeof_row() == ROW_DTYPE.itemsize, (_LIB.orc_sizeof_row(), ROW_DTYPE.itemsize)
        assert _LIB.orc_sizeof_sp() == SP_DTYPE.itemsize
        for f in ("orc_np_sum_f32", "orc_np_mean_f32", "orc_np_var_f32", "orc_np_std_f32",
                  "orc_np_median_f32", "orc_np_mad_f32", "orc_np_nanmedian_f32"):
            getattr(_LIB, f).restype = C.c_float
        for f in ("orc_np_sum_f64", "orc_np_nanstd_f64", "orc_np_percentile_diff_f32"):
            getattr(_LIB, f).restype = C.c_double
        for f in ("orc_find_peaks", "orc_correct_plateau", "orc_correct_split", "orc_adapter_candidate",
                  "orc_polya_peak", "orc_pool_row", "orc_cnn_prepare_row", "orc_cnn_fallback"):
            getattr(_LIB, f).restype = C.c_long
    return _LIB


def _rng(r):
    lo = -np.inf if r is None or r[0] is None else float(r[0])
    hi = np.inf if r is None or r[1] is None else float(r[1])
    return (C.c_double * 2)(lo, hi)


def make_cfg(spc) -> Cfg:
    """spc: any object tree with the reference's SigProcConfig attribute names."""
    c = Cfg()
    for k in ("min_obs_adapter", "max_obs_adapter", "min_obs_polya", "downscale_factor", "max_obs_trace"):
        setattr(c, k, int(getattr(spc.core, k)))
    c.sig_norm_outlier_thresh = float(spc.core.sig_norm_outlier_thresh)
    L = spc.llr_boundaries
    c.adapter_peak_prominence = float(L.adapter_peak_prominence)
    c.adapter_peak_rel_height = float(L.adapter_peak_rel_height)
    c.adapter_peak_width = int(L.adapter_peak_width)
    M = spc.mvs_polya
    for k in ("mvs_detect_check", "mvs_detect_overwrite", "search_window", "pA_mean_window", "pA_var_window",
              "median_shift_window", "polyA_window"):
        setattr(c, k, int(getattr(M, k)))
    for k in ("pA_mean_range", "pA_var_range", "median_shift_range", "polyA_med_range", "polyA_local_range",
              "pA_mean_adapter_med_scale_range"):
        setattr(c, k, _rng(getattr(M, k)))
    R = spc.real_range
    for k in ("detect_open_pores", "real_signal_check", "mean_window", "max_obs_local_range"):
        setattr(c, k, int(getattr(R, k)))
    for k in ("mean_start_range", "mean_end_range", "local_range", "adapter_mad_range"):
        setattr(c, k, _rng(getattr(R, k)))
    c.detect_med_shift = int(spc.med_shift.detect_med_shift)
    c.med_shift_window = int(spc.med_shift.med_shift_window)
    c.med_shift_range = _rng(spc.med_shift.med_shift_range)
    S = spc.rna_start_peak
    c.sp_downscale_factor = int(S.downscale_factor)
    c.start_peak_max_idx = int(S.start_peak_max_idx)
    c.sp_offset1 = int(S.offset1)
    c.sp_offset2 = int(S.offset2)
    c.open_pore_pa = float(S.open_pore_pa)
    c.polya_cand_k = int(spc.cnn_boundaries.polya_cand_k)
    c.fallback_to_llr_short_reads = int(spc.cnn_boundaries.fallback_to_llr_short_reads)
    c.primary_method = PRIMARY[spc.primary_method]
    return c


def rows_to_dicts(rows: np.ndarray, primary: str):
    """orc_row[] -> list of dicts with the DetectResults field names."""
    out = []
    for r in rows:
        if 9 <= r["fail_code"] <= 14:  # exception rows: DetectResults(success=False, fail_reason=str(e))
            out.append({"success": False, "fail_reason": FAIL_STR[int(r["fail_code"])], "_exception": True})
            continue
        d = {"success": bool(r["success"])}
        pres = int(r["present"])
        for i, name in enumerate(COLS):
            v = None
            if pres >> i & 1:
                v = float(r["col"][i])
                if name in INT_COLS:
                    v = int(v)
            d[name.replace("PRIMARY", primary)] = v
        nc = int(r["n_cand"])
        d["polya_candidates"] = None if nc < 0 else [int(x) for x in r["cand"][:nc]]
        no = int(r["n_open_pores"])
        d["open_pores"] = None if no < 0 else [int(x) for x in r["open_pores"][:min(no, MAX_OP)]]
        d["_n_open_pores"] = no
        fc = int(r["fail_code"])
        fr = FAIL_STR[fc]
        if fc == 7:
            names = ["mean", "var", "med", "range", "shift"]
            fr += " ".join(n for b, n in enumerate(names) if r["mvs_fail_mask"] >> b & 1)
        spt = SP_TYPES[int(r["start_peak_type"])]
        d["start_peak_open_pore_type"] = spt
        if spt is not None and fc != 0 and primary == "start_peak":
            fr = fr + "+" + spt
        d["fail_reason"] = fr
        d["mvs_llr_polya_end_adjust_ignored"] = False
        d["mvs_llr_polya_end_to_early_stop"] = bool(r["mvs_fail_mask"] >> 8 & 1)
        out.append(d)
    return out


def _f32c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def detect_llr(batch, full_lens, spc, with_start_peak=False, return_params=False):
    """combined_detect_llr2 over ONE minibatch -> list of dict rows.  Raises ValueError on
    MAD == 0 (as the reference) and on reads without any valid pooled block."""
    L = lib()
    b, bp = _f32c(batch)
    n, m = b.shape
    lens = np.ascontiguousarray(full_lens, dtype=np.int32)
    cfg = make_cfg(spc)
    rows = np.zeros(n, dtype=ROW_DTYPE)
    np4 = np.zeros(4, dtype=np.float64)
    rc = L.orc_detect_llr_minibatch(bp, lens.ctypes.data_as(C.POINTER(C.c_int32)), C.c_long(n), C.c_long(m),
                                    C.byref(cfg), rows.ctypes.data_as(C.c_void_p), int(with_start_peak),
                                    np4.ctypes.data_as(C.POINTER(C.c_double)))
    if rc == -1:
        raise ValueError("MAD normalization failed: scale is 0")
    if rc == -2:
        raise ValueError("attempt to get argmin of an empty sequence")
    d = rows_to_dicts(rows, "llr")
    return (d, np4) if return_params else d


def detect_llr_single(signal, full_len, spc, m=None):
    """combined_detect_llr on ONE read -> dict row.  signal: the read's samples (1-D, at most m of them are used); raises
    ValueError like the reference for MAD == 0 / an empty trace."""
    L = lib()
    cfg = make_cfg(spc)
    m = int(m or spc.sig_preload_size)
    row = np.full(m, np.nan, dtype=np.float32)
    s = np.asarray(signal, dtype=np.float32)[:m]
    row[: s.size] = s
    out = np.zeros(1, dtype=ROW_DTYPE)
    rc = L.orc_detect_llr_single(row.ctypes.data_as(C.POINTER(C.c_float)), C.c_long(m), C.c_long(int(full_len)), C.byref(cfg),
                                 out.ctypes.data_as(C.c_void_p))
    if rc == -1:
        raise ValueError("MAD normalization failed: scale is 0")
    if rc == -2:
        raise ValueError("attempt to get argmin of an empty sequence")
    return rows_to_dicts(out, "llr")[0]


def detect_start_peak(batch, full_lens, spc):
    L = lib()
    b, bp = _f32c(batch)
    n, m = b.shape
    lens = np.ascontiguousarray(full_lens, dtype=np.int32)
    cfg = make_cfg(spc)
    rows = np.zeros(n, dtype=ROW_DTYPE)
    L.orc_detect_start_peak_minibatch(bp, lens.ctypes.data_as(C.POINTER(C.c_int32)), C.c_long(n), C.c_long(m),
                                      C.byref(cfg), rows.ctypes.data_as(C.c_void_p))
    return open_pore_float_column(rows_to_dicts(rows, "start_peak"))


def open_pore_float_column(rows):
    """detect_rna_start_peak collects `open_pore_idx` in a DataFrame column (reference adapted/detect/start_peak.py:86-116): all
    None -> an object column, every row keeps None; as soon as ONE read of the minibatch has a flagged open pore the column is
    float64 -- the flagged rows read back as floats (1540.0), the others as NaN (combined.py:337 copies the cell).  Pinned by
    tests/golden/rna004_start_peak_200k (read 19 flagged, 19 reads NaN).  Rows of raised exceptions hold nothing."""
    key = "start_peak_open_pore_idx"
    if any(r.get(key) is not None for r in rows):
        for r in rows:
            if r.get("_exception"):
                continue
            r[key] = float(r[key]) if r.get(key) is not None else float("nan")
    return rows


def start_peak_table(batch, full_lens, spc):
    L = lib()
    b, _ = _f32c(batch)
    n, m = b.shape
    cfg = make_cfg(spc)
    out = np.zeros(n, dtype=SP_DTYPE)
    for r in range(n):
        L.orc_start_peak_row(b[r].ctypes.data_as(C.POINTER(C.c_float)), C.c_long(m), C.c_long(int(full_lens[r])),
                             C.byref(cfg), out[r:r + 1].ctypes.data_as(C.c_void_p))
    return out


def llr_stages(row, spc, np4):
    """Stage outputs for one read (down, g1, g2, indices) given the minibatch's N1 params."""
    L = lib()
    r, rp = _f32c(row)
    m = r.size
    cfg = make_cfg(spc)
    T = min(cfg.max_obs_trace, m)
    ds = cfg.downscale_factor
    Lp = max(0, (T - cfg.min_obs_adapter + ds - 1) // ds)
    down = np.zeros(max(Lp, 1), dtype=np.float32)
    g1 = np.zeros(max(Lp, 1), dtype=np.float64)
    g2 = np.zeros(max(Lp, 1), dtype=np.float64)
    ae, pe, nv = C.c_long(0), C.c_long(0), C.c_long(0)
    st = (C.c_long * 4)()
    p4 = np.ascontiguousarray(np4, dtype=np.float64)
    rc = L.orc_llr_primary(rp, C.c_long(m), C.byref(cfg), p4.ctypes.data_as(C.POINTER(C.c_double)),
                           C.byref(ae), C.byref(pe), C.byref(nv), down.ctypes.data_as(C.POINTER(C.c_float)),
                           g1.ctypes.data_as(C.POINTER(C.c_double)), g2.ctypes.data_as(C.POINTER(C.c_double)), st)
    n = nv.value
    return dict(rc=rc, down=down[:n], n_valid=n, g1=g1[:n], g2=g2[:n], raw_first=st[0], cand=st[1],
                polya_idx=st[2], adapter_end=ae.value, polya_end=pe.value)


def norm_params(batch, T, thresh):
    L = lib()
    b, bp = _f32c(batch)
    n, m = b.shape
    out = np.zeros(4, dtype=np.float64)
    rc = L.orc_norm_params(bp, C.c_long(n), C.c_long(m), C.c_long(T), C.c_double(thresh),
                           out.ctypes.data_as(C.POINTER(C.c_double)))
    return rc, out


def c_llr_trace(raw_signal, start, end, min_obs, border_trim, stride=1, adapter_early_stopping=0, adapter_early_stop_window=500,
                adapter_early_stop_stride=100, polya_early_stopping=0, polya_early_stop_window=50, polya_early_stop_stride=10,
                return_c_c2=0, sums=None):
    """adapted/detect/_c_llr.pyx:202-236 (and :176-199 with ``sums=(c, c2)``), restated in oracle/adapted_oracle.c"""
    L = lib()
    if sums is None:
        raw = np.ascontiguousarray(raw_signal, dtype=np.float64)
        n = raw.size
        c = np.zeros(n)
        c2 = np.zeros(n)
        L.orc_cumsum_f64(raw.ctypes.data_as(C.c_void_p), C.c_long(n), c.ctypes.data_as(C.c_void_p), c2.ctypes.data_as(C.c_void_p))
    else:
        c = np.ascontiguousarray(sums[0], dtype=np.float64)
        c2 = np.ascontiguousarray(sums[1], dtype=np.float64)
        n = c.size
    g = np.zeros(n)
    tmp = np.zeros(max(n, 1))
    L.orc_c_llr_trace_gains.restype = C.c_int
    rc = L.orc_c_llr_trace_gains(c.ctypes.data_as(C.c_void_p), c2.ctypes.data_as(C.c_void_p), C.c_long(n), C.c_long(start), C.c_long(end),
                                 C.c_long(min_obs), C.c_long(border_trim), C.c_long(stride), C.c_long(adapter_early_stopping),
                                 C.c_long(adapter_early_stop_window), C.c_long(adapter_early_stop_stride), C.c_long(polya_early_stopping),
                                 C.c_long(polya_early_stop_window), C.c_long(polya_early_stop_stride),
                                 g.ctypes.data_as(C.c_void_p), tmp.ctypes.data_as(C.c_void_p))
    if rc:
        raise AssertionError("early-stop stride is not a multiple of stride")
    return (g, c, c2) if return_c_c2 else g


def find_peaks(x, distance=None, prominence=None, width=None, rel_height=0.5, cap=None):
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    cap = cap or max(1, x.size)
    out = np.zeros(cap, dtype=np.int64)
    k = L.orc_find_peaks(x.ctypes.data_as(C.POINTER(C.c_double)), C.c_long(x.size),
                         int(distance is not None), C.c_double(distance or 0.0),
                         int(prominence is not None), C.c_double(prominence if prominence is not None else 0.0),
                         int(width is not None), C.c_double(width or 0.0), C.c_double(rel_height),
                         out.ctypes.data_as(C.POINTER(C.c_long)), C.c_long(cap))
    return out[:min(k, cap)]


# ---------------------------------------------------------------------------- CNN path
def cnn_prepare(batch, spc):
    """C1 prepare_data -> float32 [N, Lc]"""
    L = lib()
    b, _ = _f32c(batch)
    n, m = b.shape
    cfg = make_cfg(spc)
    ds, off = cfg.downscale_factor, cfg.min_obs_adapter
    Lc = (m - off + ds - 1) // ds
    out = np.zeros((n, Lc), dtype=np.float32)
    for r in range(n):
        L.orc_cnn_prepare_row(b[r].ctypes.data_as(C.POINTER(C.c_float)), C.c_long(m), C.byref(cfg),
                              out[r].ctypes.data_as(C.POINTER(C.c_float)))
    return out


def _conv1d(x, w, b, stride, pad):
    """float32 cross-correlation, x [N, Cin, L], w [Cout, Cin, K] (plain numpy restatement of torch Conv1d)"""
    n, cin, L = x.shape
    cout, _, K = w.shape
    xp = np.zeros((n, cin, L + 2 * pad), dtype=np.float32)
    xp[:, :, pad:pad + L] = x
    Lo = (L + 2 * pad - K) // stride + 1
    cols = np.stack([xp[:, :, k:k + stride * (Lo - 1) + 1:stride] for k in range(K)], axis=2)  # [N, Cin, K, Lo]
    y = np.einsum("nckl,ock->nol", cols, w, optimize=True).astype(np.float32)
    return y + b[None, :, None]


def _conv_transpose1d(x, w, b, stride, pad):
    """x [N, Cin, L], w [Cin, Cout, K] (torch ConvTranspose1d)"""
    n, cin, L = x.shape
    _, cout, K = w.shape
    Lfull = (L - 1) * stride + K
    y = np.zeros((n, cout, Lfull), dtype=np.float32)
    contrib = np.einsum("ncl,cok->nokl", x, w, optimize=True).astype(np.float32)  # [N, Cout, K, L]
    for k in range(K):
        y[:, :, k:k + stride * (L - 1) + 1:stride] += contrib[:, :, k, :]
    y = y[:, :, pad:Lfull - pad]
    return y + b[None, :, None]


def cnn_forward(x, weights):
    """BoundariesCNN forward in numpy float32 (adapted/detect/cnn.py:16-52); x [N, Lc] -> scores [N, 2, Lo]"""
    h = x[:, None, :].astype(np.float32)
    h = np.maximum(_conv1d(h, weights["0.weight"], weights["0.bias"], 3, 3), 0)
    h = np.maximum(_conv1d(h, weights["2.weight"], weights["2.bias"], 1, 3), 0)
    h = np.maximum(_conv1d(h, weights["4.weight"], weights["4.bias"], 1, 3), 0)
    return _conv_transpose1d(h, weights["6.weight"], weights["6.bias"], 3, 3)


def cnn_predict(scores, spc):
    """C3 cnn_predict + cnn_detect scaling (adapted/detect/cnn.py:101-182) on numpy scores"""
    from scipy.signal import find_peaks

    scores = scores.copy()
    co = spc.core
    n, _, Lo = scores.shape
    na = (co.max_obs_adapter - co.min_obs_adapter) // co.downscale_factor
    a = np.argmax(scores[:, 0, :na], axis=1)
    k = spc.cnn_boundaries.polya_cand_k
    ar = np.arange(Lo)[None, :]
    if k >= 1:  # (cnn.py:126-134; k < 1: no poly(A) search at all)
        scores[:, 1, :][ar < a[:, None]] = -5.0
        p = np.argmax(scores[:, 1, :], axis=1)
    else:
        p = np.zeros(n, dtype=np.int64)
    if k <= 1:  # (cnn.py:160: the plain pair, no find_peaks, no row compaction -- pinned by tests/golden/rna004_cnn_k1)
        preds = (np.column_stack((a, p)) * co.downscale_factor + co.min_obs_adapter).astype(int)
        preds[preds == co.min_obs_adapter] = 0
        return preds
    scores[:, 1, :][ar > p[:, None]] = -5.0
    flat = scores[:, 1, :].reshape(-1)
    cand, _ = find_peaks(flat, distance=5)
    rid = cand // Lo
    order = np.lexsort((-flat[cand], rid))
    cand = cand[order]
    groups = np.split(np.mod(cand, Lo), np.where(np.diff(rid) != 0)[0] + 1)
    top = np.zeros((n, k), dtype=np.int64)
    for i, g in enumerate(groups):
        top[i, :len(g)] = g[:k]
    preds = (np.column_stack((a[:, None], top)) * co.downscale_factor + co.min_obs_adapter).astype(int)
    preds[preds == co.min_obs_adapter] = 0
    return preds


def detect_cnn_from_preds(batch, full_lens, preds, spc):
    """validate + short-read fallback given the CNN's predictions -> dict rows"""
    L = lib()
    b, bp = _f32c(batch)
    n, m = b.shape
    lens = np.ascontiguousarray(full_lens, dtype=np.int32)
    pr = np.ascontiguousarray(preds, dtype=np.int64)
    cfg = make_cfg(spc)
    rows = np.zeros(n, dtype=ROW_DTYPE)
    L.orc_detect_cnn_from_preds(bp, lens.ctypes.data_as(C.POINTER(C.c_int32)), C.c_long(n), C.c_long(m),
                                pr.ctypes.data_as(C.c_void_p), int(pr.shape[1] - 1), C.byref(cfg),
                                rows.ctypes.data_as(C.c_void_p))
    return rows_to_dicts(rows, "cnn")

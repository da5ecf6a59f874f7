"""ctypes binding of libadapted_hip.so (include/adapted_hip.h) and the row <-> DetectResults
conversion shared by the operators and the CSV writer.

There is no CPU fallback: if the HIP library cannot be loaded the import of this module's
``load()`` fails loudly, and every operator of the package goes through it.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import List, Optional, Sequence

import numpy as np

from . import build as _build
from .container_types import DetectResults

_LIB = None

NCOL = 39
MAX_CAND = 16
MAX_OPEN_PORES = 16

ADP_IN_DEVICE, ADP_OUT_DEVICE, ADP_WITH_START_PEAK, ADP_TOPK_NONE, ADP_BOUNDS_HOST, ADP_TAILS_NAN = 1, 2, 4, 8, 16, 32
MB_OK, MB_MAD_ZERO, MB_EMPTY_TRACE = 0, 1, 2

COLS = ["signal_len", "preloaded", "adapter_start", "adapter_end", "adapter_len", "adapter_mean",
        "adapter_std", "adapter_med", "adapter_mad", "polya_start", "polya_end", "polya_len",
        "polya_mean", "polya_std", "polya_med", "polya_mad", "rna_preloaded_start",
        "rna_preloaded_len", "rna_preloaded_mean", "rna_preloaded_std", "rna_preloaded_med",
        "rna_preloaded_mad", "start_peak_idx", "start_peak_pa", "start_peak_next_max_idx",
        "start_peak_next_max_pa", "start_peak_open_pore_idx", "adapter_rna_median_shift",
        "{primary}_adapter_end", "{primary}_polya_end", "mvs_detect_mean_at_loc",
        "mvs_detect_var_at_loc", "mvs_detect_polya_med", "mvs_detect_polya_local_range",
        "mvs_detect_med_shift", "real_adapter_mean_start", "real_adapter_mean_end",
        "real_adapter_local_range", "mvs_adapter_end"]
assert len(COLS) == NCOL
_INT_COLS = {0, 1, 2, 3, 4, 9, 10, 11, 16, 17, 22, 24, 26, 28, 29, 38}
# columns the reference holds as numpy float32 scalars (kept as float32 so that the CSV
# writer rounds them like pandas does)
_F32_COLS = {23, 25, 27}

ROW_DTYPE = np.dtype([("col", "<f8", (NCOL,)), ("present", "<u8"), ("success", "<i4"), ("fail_code", "<i4"),
                      ("mvs_fail_mask", "<i4"), ("start_peak_type", "<i4"), ("n_cand", "<i4"),
                      ("n_open_pores", "<i4"), ("cand", "<i8", (MAX_CAND,)), ("open_pores", "<i4", (MAX_OPEN_PORES,)),
                      ("open_pores_more", "<i4"), ("reserved_", "<i4")])

# open_pores lists longer than a row holds (the reference's list has no length limit): host rows carry a token into this
# registry in `open_pores_more` (Engine._attach_open_pores swaps the arena offset of the call for it), so that rows can be
# sliced, concatenated and gathered freely before they become DetectResults
_OPEN_PORES_MORE: dict = {}
_open_pores_token = [0]


def register_open_pores(arr) -> int:
    _open_pores_token[0] += 1
    _OPEN_PORES_MORE[_open_pores_token[0]] = np.asarray(arr, dtype=np.int64)
    return _open_pores_token[0]


def clear_open_pores():
    _OPEN_PORES_MORE.clear()


def empty_rows(n: int) -> np.ndarray:
    """n zeroed result rows (the all-None rows of exceptions and dropped minibatches)"""
    rows = np.zeros(n, dtype=ROW_DTYPE)
    rows["n_cand"] = -1
    rows["n_open_pores"] = -1
    rows["open_pores_more"] = -1
    return rows


FAIL_REASONS = {
    0: None,
    1: "No adapter detected (primary)",
    2: "adapter MAD check failed",
    3: "Open pore too close to boundary",
    4: "Real signal check failed",
    5: "No polya detected (primary)",
    6: "MVS polya check failed: not enough signal",
    7: "MVS polya check failed: ",
    8: "Median shift check failed",
    9: "'NoneType' object is not iterable",
    10: "slice indices must be integers or None or have an __index__ method",
    11: "Moving window must between 1 and n, inclusive",
    12: "pA_mean_range is not specified",
    13: "attempt to get argmin of an empty sequence",
    14: "MAD normalization failed: scale is 0",
    15: "No adapter detected in range (mvs_detect)",
}
_MVS_NAMES = ["mean", "var", "med", "range", "shift"]
START_PEAK_TYPES = {0: None, 1: "open pore in adapter", 2: "potential concatemer adapter-only read"}
PRIMARY_CODE = {"llr": 0, "cnn": 1, "start_peak": 2}


class AdpCfg(C.Structure):
    _fields_ = [
        ("min_obs_adapter", C.c_int32), ("max_obs_adapter", C.c_int32), ("min_obs_polya", C.c_int32),
        ("downscale_factor", C.c_int32), ("max_obs_trace", C.c_int32), ("primary_method", C.c_int32),
        ("sig_norm_outlier_thresh", C.c_double),
        ("adapter_peak_prominence", C.c_double), ("adapter_peak_rel_height", C.c_double),
        ("adapter_peak_width", C.c_int32),
        ("mvs_detect_check", C.c_int32), ("mvs_detect_overwrite", C.c_int32), ("search_window", C.c_int32),
        ("pA_mean_window", C.c_int32), ("pA_var_window", C.c_int32), ("median_shift_window", C.c_int32),
        ("polyA_window", C.c_int32),
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
    ]


class HipLibraryError(RuntimeError):
    pass


def hip_runtimes() -> List[str]:
    """paths of every HIP runtime (libamdhip64) mapped into this process -- one entry is the healthy state"""
    seen = []
    try:
        with open("/proc/self/maps") as fh:
            for ln in fh:
                path = ln.split(None, 5)[-1].strip() if ln.count("/") else ""
                if "libamdhip64" in os.path.basename(path) and path not in seen:
                    seen.append(path)
    except OSError:
        pass
    return seen


def _elf_dynamic(path: str):
    """(SONAME or None, [DT_NEEDED names]) of a little-endian ELF64 shared object -- read from the file, nothing is loaded
    (the ELF header, the program headers, the dynamic section and the strings it names: a few KB of a multi-MB runtime)"""
    import struct

    with open(path, "rb") as fh:
        def at(off, size):
            fh.seek(off)
            return fh.read(size)

        head = at(0, 0x40)
        if head[:6] != b"\x7fELF\x02\x01":
            raise ValueError("not a little-endian ELF64 file: %s" % path)
        e_phoff, = struct.unpack_from("<Q", head, 0x20)
        e_phentsize, e_phnum = struct.unpack_from("<HH", head, 0x36)
        ph = at(e_phoff, e_phentsize * e_phnum)
        loads, dyn = [], None
        for i in range(e_phnum):
            p_type, _flags, p_offset, p_vaddr, _paddr, p_filesz = struct.unpack_from("<IIQQQQ", ph, i * e_phentsize)
            if p_type == 1:
                loads.append((p_vaddr, p_offset, p_filesz))
            elif p_type == 2:
                dyn = (p_offset, p_filesz)
        if dyn is None:
            return None, []

        def off_of(vaddr):
            for va, off, sz in loads:
                if va <= vaddr < va + sz:
                    return off + (vaddr - va)
            raise ValueError("address outside the file's segments")

        tags = []
        dsec = at(dyn[0], dyn[1])
        for o in range(0, len(dsec) - 15, 16):
            tag, val = struct.unpack_from("<qQ", dsec, o)
            if tag == 0:
                break
            tags.append((tag, val))
        strtab = off_of(next(v for t, v in tags if t == 5))

        def name(v):
            buf = b""
            while b"\0" not in buf:  # (names are short: one or two reads)
                more = at(strtab + v + len(buf), 256)
                if not more:
                    break
                buf += more
            return buf.split(b"\0", 1)[0].decode()

        return next((name(v) for t, v in tags if t == 14), None), [name(v) for t, v in tags if t == 1]


def _share_torch_runtime(lib_path: Optional[str] = None):
    """One process, ONE HIP runtime, whatever the import order.  PyTorch-ROCm wheels bundle a libamdhip64.so; when its SONAME is
    the one libadapted_hip.so was linked against (libamdhip64.so.7 with this image's wheel and /opt/rocm), this library loaded
    after torch binds to torch's copy by itself, and loaded BEFORE torch it would bind to /opt/rocm's and a later `import torch`
    would map a second runtime beside it (two runtimes = two device contexts: RCCL, events and allocations of one are invisible to
    the other).  So when no runtime is mapped yet and a torch wheel is installed whose bundled runtime carries exactly that
    SONAME, map that copy first (by path, without importing torch); the dynamic loader then resolves both this library and a
    later torch to it.  A wheel whose runtime has another SONAME (another ROCm major) is left alone: this library then runs on the
    system runtime it was built against, as it did before.  ADAPTED_HIP_RUNTIME=system keeps /opt/rocm's runtime in any case
    (fine for processes that never import torch)."""
    if os.environ.get("ADAPTED_HIP_RUNTIME", "auto") == "system" or hip_runtimes():
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if not os.path.exists(cand):
            return
        want = [n for n in _elf_dynamic(lib_path or _build.LIB)[1] if n.startswith("libamdhip64")]
        have = _elf_dynamic(cand)[0]
        if want and have == want[0]:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001 -- no torch, or an unusual install: the system runtime it is
        pass


def assert_one_runtime():
    """Raise HipLibraryError when two HIP runtimes are mapped into this process AND torch is imported: torch's device context and
    this library's would not see each other (RCCL, events, allocations).  Without torch a second mapped copy is idle and harmless.
    Checked when the library is loaded and again at every entry point that hands torch's device memory to the library or the
    library's rows to torch.distributed (Engine calls with device pointers, parallel.gather_rows, bench.py's process group): a
    torch imported AFTER the library was loaded on another runtime is caught there instead of failing silently."""
    rts = hip_runtimes()
    if len(rts) > 1 and "torch" in sys.modules:
        raise HipLibraryError("two HIP runtimes are mapped into this process (%s) and torch is imported: import torch BEFORE anything "
                              "that links /opt/rocm's libamdhip64, or -- for a process that does not need torch on the GPU -- set "
                              "ADAPTED_HIP_RUNTIME=system before adapted_amd.lib is loaded and do not import torch" % ", ".join(rts))


_RUNTIME_CHECKED_WITH_TORCH = False


def _check_runtime_once_torch_is_here():
    """assert_one_runtime, once per process after torch has appeared (reading /proc/self/maps per call would cost a detect call ~1 ms)"""
    global _RUNTIME_CHECKED_WITH_TORCH
    if not _RUNTIME_CHECKED_WITH_TORCH and "torch" in sys.modules:
        assert_one_runtime()
        _RUNTIME_CHECKED_WITH_TORCH = True


def load():
    """Load (building if needed) libadapted_hip.so.  Raises HipLibraryError if impossible."""
    global _LIB
    if _LIB is not None:
        return _LIB
    try:
        path = os.environ.get("ADAPTED_HIP_LIB") or _build.build()  # (override: a developer's experimental build)
        _share_torch_runtime(path)
        lib = C.CDLL(path)
    except Exception as e:  # no CPU fallback by design
        raise HipLibraryError("libadapted_hip.so is required (hipcc build or load failed): %s" % e) from e
    assert_one_runtime()
    if lib.adp_sizeof_cfg() != C.sizeof(AdpCfg) or lib.adp_sizeof_row() != ROW_DTYPE.itemsize:
        raise HipLibraryError("ABI mismatch between adapted_amd/lib.py and libadapted_hip.so")
    lib.adp_last_error.restype = C.c_char_p
    lib.adp_stream.restype = C.c_void_p
    _LIB = lib
    return lib


EXPORTS = ["adp_abi_version", "adp_sizeof_cfg", "adp_sizeof_row", "adp_last_error", "adp_device_count", "adp_create",
           "adp_destroy", "adp_set_config", "adp_stream", "adp_synchronize", "adp_detect_llr", "adp_detect_start_peak",
           "adp_cnn_prepare", "adp_validate_candidates", "adp_llr_refine_polya", "adp_synth_fill", "adp_dev_alloc", "adp_dev_free",
           "adp_memcpy_h2d", "adp_memcpy_d2h", "adp_set_profiling", "adp_kernel_times", "adp_debug_fetch",
           "adp_debug_llr_upto", "adp_debug_log", "adp_cnn_topk", "adp_host_alloc", "adp_host_free", "adp_memcpy_h2d_async",
           "adp_copy_mark", "adp_copy_wait", "adp_debug_divcheck", "adp_calibrate_i16", "adp_expand_ragged", "adp_set_layout",
           "adp_cnn_set_weights", "adp_cnn_forward", "adp_cnn_predict", "adp_detect_cnn", "adp_open_pores_arena", "adp_detect_llr_i16", "adp_expand_ragged_i16",
           "adp_c_llr_trace"]


class AdpTraceArgs(C.Structure):
    """struct adp_trace_args (include/adapted_hip.h): the keyword arguments of the reference's c_llr_trace"""
    _fields_ = [(k, C.c_int32) for k in ("min_obs", "border_trim", "stride", "adapter_early_stopping", "adapter_early_stop_window",
                                         "adapter_early_stop_stride", "polya_early_stopping", "polya_early_stop_window",
                                         "polya_early_stop_stride")]


ADP_TRACE_FROM_SUMS = 64


class MinibatchDropped(RuntimeError):
    """batch-level failure of one minibatch (the reference drops it and logs): status = ADP_MB_*"""

    def __init__(self, status: int):
        super().__init__("minibatch dropped (status %d)" % status)
        self.status = status


def _rng(r):
    lo = -np.inf if r is None or r[0] is None else float(r[0])
    hi = np.inf if r is None or r[1] is None else float(r[1])
    return (C.c_double * 2)(lo, hi)


def make_cfg(spc) -> AdpCfg:
    c = AdpCfg()
    co = spc.core
    c.min_obs_adapter, c.max_obs_adapter, c.min_obs_polya = int(co.min_obs_adapter), int(co.max_obs_adapter), int(co.min_obs_polya)
    c.downscale_factor, c.max_obs_trace = int(co.downscale_factor), int(co.max_obs_trace)
    c.primary_method = PRIMARY_CODE[spc.primary_method]
    c.sig_norm_outlier_thresh = float(co.sig_norm_outlier_thresh)
    L = spc.llr_boundaries
    c.adapter_peak_prominence, c.adapter_peak_rel_height = float(L.adapter_peak_prominence), float(L.adapter_peak_rel_height)
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
    c.detect_med_shift, c.med_shift_window = int(spc.med_shift.detect_med_shift), int(spc.med_shift.med_shift_window)
    c.med_shift_range = _rng(spc.med_shift.med_shift_range)
    S = spc.rna_start_peak
    c.sp_downscale_factor, c.start_peak_max_idx = int(S.downscale_factor), int(S.start_peak_max_idx)
    c.sp_offset1, c.sp_offset2, c.open_pore_pa = int(S.offset1), int(S.offset2), float(S.open_pore_pa)
    c.polya_cand_k = int(spc.cnn_boundaries.polya_cand_k)
    c.fallback_to_llr_short_reads = int(spc.cnn_boundaries.fallback_to_llr_short_reads)
    return c


def fail_reason_of(row) -> Optional[str]:
    fc = int(row["fail_code"])
    fr = FAIL_REASONS[fc]
    if fc == 7:
        fr += " ".join(n for b, n in enumerate(_MVS_NAMES) if int(row["mvs_fail_mask"]) >> b & 1)
    return fr


def rows_to_results(rows: np.ndarray, primary: str, consume: bool = False) -> List[DetectResults]:
    """adp_row[] -> DetectResults, value for value what the reference's validate_boundaries
    returns (types: python int / float, np.float32 where the reference keeps numpy scalars).
    consume: the rows are not needed again -- their overflow open_pores lists leave the registry (a long run converts
    every row exactly once: adapted_amd/main.py)."""
    out = []
    names = [c.format(primary=primary) for c in COLS]
    for r in rows:
        fc = int(r["fail_code"])
        if 9 <= fc <= 14:  # the reference raised inside its per-read try block
            out.append(DetectResults(success=False, fail_reason=fail_reason_of(r)))
            continue
        d = DetectResults(success=bool(r["success"]))
        pres = int(r["present"])
        col = r["col"]
        for i, name in enumerate(names):
            if pres >> i & 1:
                v = col[i]
                if i in _INT_COLS:
                    v = int(v)
                elif i in _F32_COLS:
                    v = np.float32(v)
                else:
                    v = float(v)
                setattr(d, name, v)
        nc = int(r["n_cand"])
        if nc >= 0:
            d.polya_candidates = np.asarray(r["cand"][:nc], dtype=np.int64)
        no = int(r["n_open_pores"])
        if no >= 0:
            if no <= MAX_OPEN_PORES:
                d.open_pores = np.asarray(r["open_pores"][:no], dtype=np.int64)
            else:  # the whole list was fetched from the call's arena (Engine._attach_open_pores)
                more = (_OPEN_PORES_MORE.pop if consume else _OPEN_PORES_MORE.get)(int(r["open_pores_more"]), None)
                if more is None or more.size != no:
                    raise HipLibraryError("a row with %d open pores lost its overflow list (rows from a device buffer? fetch them "
                                          "through Engine.attach_open_pores)" % no)
                d.open_pores = more
        d.mvs_llr_polya_end_adjust_ignored = False
        d.mvs_llr_polya_end_to_early_stop = bool(int(r["mvs_fail_mask"]) >> 8 & 1)  # (mvs_detect_overwrite only)
        fr = fail_reason_of(r)
        spt = START_PEAK_TYPES[int(r["start_peak_type"])]
        d.start_peak_open_pore_type = spt
        if primary == "start_peak":
            if fr is not None and spt is not None:
                fr = fr + "+" + spt
        else:
            d.llr_detect_log = "" if primary == "llr" else None
        d.fail_reason = fr
        out.append(d)
    return out


def open_pore_float_column(results: List[DetectResults]) -> List[DetectResults]:
    """The start-peak primary's results of ONE minibatch as the reference returns them: detect_rna_start_peak keeps `open_pore_idx`
    in a DataFrame column (adapted/detect/start_peak.py:86-116) and combined_detect_start_peak copies the cell (combined.py:337) --
    all None: every row keeps None; one flagged read in the minibatch: the column is float64, the flagged rows read back as
    floats, the others as NaN.  (The CSV text is the same either way.)  Pinned by tests/golden/rna004_start_peak_200k."""
    if any(r.start_peak_open_pore_idx is not None for r in results):
        for r in results:
            if r.signal_len is None and r.start_peak_idx is None:
                continue  # (a row that only carries a raised exception)
            r.start_peak_open_pore_idx = float(r.start_peak_open_pore_idx) if r.start_peak_open_pore_idx is not None else float("nan")
    return results


class Engine:
    """One GPU's detect engine: a handle of libadapted_hip.so sized for (max_reads, m)."""

    def __init__(self, spc, max_reads: int, m: int, device: int = 0, single_read_layout: bool = False):
        """single_read_layout: the engine of combined_detect_llr (adapted/detect/combined.py:39-119): pooled from sample 0, every
        read its own minibatch (detect_llr_rows with minibatch = 1)"""
        self.lib = load()
        self.spc = spc
        self.cfg = make_cfg(spc)
        self.max_reads, self.m, self.device = int(max_reads), int(m), int(device)
        self._h = C.c_void_p()
        self._pinned = {}
        self._check(self.lib.adp_create(self.device, C.byref(self.cfg), self.max_reads, self.m, C.byref(self._h)))
        if single_read_layout:
            self._check(self.lib.adp_set_layout(self._h, 1))

    # -- plumbing ---------------------------------------------------------------------
    def _check(self, rc):
        if rc < 0:
            raise HipLibraryError("libadapted_hip: error %d: %s" % (rc, self.lib.adp_last_error().decode()))
        return rc

    def close(self):
        if self._h:
            self.lib.adp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_config(self, spc):
        self.spc = spc
        self.cfg = make_cfg(spc)
        self._check(self.lib.adp_set_config(self._h, C.byref(self.cfg)))

    @property
    def stream(self) -> int:
        return int(self.lib.adp_stream(self._h) or 0)

    def set_profiling(self, on: bool):
        self._check(self.lib.adp_set_profiling(self._h, int(on)))

    def kernel_times(self):
        cap = 4096  # (a grouped call reports every group's launches)
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        k = self._check(self.lib.adp_kernel_times(self._h, names, ms, cap))
        return [(names[i].decode(), float(ms[i])) for i in range(k)]

    # -- device memory ------------------------------------------------------------------
    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self.lib.adp_dev_alloc(self._h, C.c_uint64(nbytes), C.byref(p)))
        return int(p.value)

    def dev_free(self, ptr: int):
        self._check(self.lib.adp_dev_free(self._h, C.c_void_p(ptr)))

    def calibrate_i16(self, raw_dev: int, len_dev: int, scale_dev: int, offset_dev: int, n: int, out_dev: int):
        """int16 ADC samples [n, m] + per-read (scale, offset) -> float32 [n, m] NaN-padded, on the device (asynchronous
        on the handle's stream: the detect call that follows is ordered behind it)"""
        self._check(self.lib.adp_calibrate_i16(self._h, C.c_void_p(raw_dev), C.c_void_p(len_dev), C.c_void_p(scale_dev),
                                               C.c_void_p(offset_dev), int(n), self.m, C.c_void_p(out_dev)))

    def expand_ragged(self, packed_dev: int, is_int16: bool, offs_dev: int, len_dev: int, n: int, out_dev: int,
                      scale_dev: int = 0, offset_dev: int = 0):
        """reads packed back to back (float32 pA, or int16 ADC + per-read calibration) -> float32 [n, m] NaN-padded, on the
        device (asynchronous on the handle's stream)"""
        self._check(self.lib.adp_expand_ragged(self._h, C.c_void_p(packed_dev), int(bool(is_int16)), C.c_void_p(offs_dev),
                                               C.c_void_p(len_dev), C.c_void_p(scale_dev or None), C.c_void_p(offset_dev or None),
                                               int(n), self.m, C.c_void_p(out_dev)))

    def expand_ragged_i16(self, packed_dev: int, offs_dev: int, len_dev: int, n: int, out_dev: int):
        """packed raw int16 reads -> raw int16 [n, m] on the device (what detect_llr_rows_i16 reads); asynchronous"""
        self._check(self.lib.adp_expand_ragged_i16(self._h, C.c_void_p(packed_dev), C.c_void_p(offs_dev), C.c_void_p(len_dev), int(n), self.m,
                                                   C.c_void_p(out_dev)))

    def host_alloc(self, shape, dtype) -> np.ndarray:
        """page-locked host array (staging for h2d_async); release with host_free(arr)"""
        dt = np.dtype(dtype)
        n = int(np.prod(shape))
        p = C.c_void_p()
        self._check(self.lib.adp_host_alloc(self._h, C.c_uint64(max(1, n * dt.itemsize)), C.byref(p)))
        buf = (C.c_char * (n * dt.itemsize)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dt).reshape(shape)
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr: np.ndarray):
        p = self._pinned.pop(arr.ctypes.data)
        self._check(self.lib.adp_host_free(self._h, C.c_void_p(p)))

    def h2d_async(self, dst: int, arr: np.ndarray, nbytes: Optional[int] = None):
        """copy from a PINNED host array on the handle's copy stream; copy_mark(slot) then copy_wait(slot) before use"""
        self._check(self.lib.adp_memcpy_h2d_async(self._h, C.c_void_p(dst), arr.ctypes.data_as(C.c_void_p),
                                                  C.c_uint64(arr.nbytes if nbytes is None else nbytes)))

    def copy_mark(self, slot: int):
        self._check(self.lib.adp_copy_mark(self._h, int(slot)))

    def copy_wait(self, slot: int = -1):
        self._check(self.lib.adp_copy_wait(self._h, int(slot)))

    def h2d(self, dst: int, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        self._check(self.lib.adp_memcpy_h2d(self._h, C.c_void_p(dst), a.ctypes.data_as(C.c_void_p), C.c_uint64(a.nbytes)))

    def d2h(self, arr: np.ndarray, src: int):
        assert arr.flags.c_contiguous
        self._check(self.lib.adp_memcpy_d2h(self._h, arr.ctypes.data_as(C.c_void_p), C.c_void_p(src), C.c_uint64(arr.nbytes)))

    def synth_fill(self, dev_signals: int, dev_full_len: Optional[int], n: int, seed: int, first_read: int,
                   decorate: bool = True):
        self._check(self.lib.adp_synth_fill(self._h, C.c_void_p(dev_signals), C.c_void_p(dev_full_len or 0), int(n), self.m,
                                            C.c_uint32(seed), C.c_uint32(first_read), int(decorate)))

    # -- operators ------------------------------------------------------------------------
    def attach_open_pores(self, rows: Optional[np.ndarray]):
        """rows (host) of the LAST call: reads with more open pores than a row holds get their whole list out of the
        call's arena (the offset in `open_pores_more` becomes a registry token)"""
        if rows is None or rows.size == 0:
            return rows
        big = np.flatnonzero(rows["n_open_pores"] > MAX_OPEN_PORES)
        if big.size:
            used = C.c_uint64(0)
            self._check(self.lib.adp_open_pores_arena(self._h, None, C.c_uint64(0), C.byref(used)))
            arena = np.zeros(int(used.value), dtype=np.int32)
            self._check(self.lib.adp_open_pores_arena(self._h, arena.ctypes.data_as(C.c_void_p), C.c_uint64(arena.size), C.byref(used)))
            for i in big:
                off, no = int(rows[i]["open_pores_more"]), int(rows[i]["n_open_pores"])
                if off < 0 or off + no > arena.size:
                    raise HipLibraryError("open-pore arena inconsistent (offset %d, %d entries, %d in use)" % (off, no, arena.size))
                rows[i]["open_pores_more"] = register_open_pores(arena[off:off + no])
        return rows

    def _in_ptrs(self, signals, full_lens, n, device_ptrs):
        if device_ptrs:
            _check_runtime_once_torch_is_here()  # (device pointers come from torch: its runtime must be this library's)
            return C.c_void_p(int(signals)), C.c_void_p(int(full_lens)), ADP_IN_DEVICE, None
        sig = np.ascontiguousarray(signals, dtype=np.float32)
        lens = np.ascontiguousarray(full_lens, dtype=np.int32)
        if sig.ndim != 2 or sig.shape != (n, self.m) or lens.shape != (n,):
            raise ValueError("signals must be float32 [n, %d] and full_lens int32 [n]" % self.m)
        return sig.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), 0, (sig, lens)

    def detect_llr_rows(self, signals, full_lens, n: int, minibatch: int, with_start_peak: bool = False,
                        device_ptrs: bool = False, rows_dev: Optional[int] = None, tails_nan: bool = False):
        """-> (rows ndarray[ROW_DTYPE] or None when rows_dev is given, mb_status int32[n_mb]).
        tails_nan: the caller guarantees that every row is NaN from min(full_len, m) on (the reference's own padding,
        adapted/file_proc.py:170-174); the streaming passes then skip the padding (ADP_TAILS_NAN)."""
        sp, lp, flags, keep = self._in_ptrs(signals, full_lens, n, device_ptrs)
        if with_start_peak:
            flags |= ADP_WITH_START_PEAK
        if tails_nan:
            flags |= ADP_TAILS_NAN
        n_mb = (n + minibatch - 1) // minibatch
        mbs = np.zeros(n_mb, dtype=np.int32)
        if rows_dev is not None:
            flags |= ADP_OUT_DEVICE
            rows, rp = None, C.c_void_p(rows_dev)
        else:
            rows = np.zeros(n, dtype=ROW_DTYPE)
            rp = rows.ctypes.data_as(C.c_void_p)
        self._check(self.lib.adp_detect_llr(self._h, sp, lp, int(n), self.m, int(minibatch), flags, rp,
                                            mbs.ctypes.data_as(C.c_void_p)))
        del keep
        return self.attach_open_pores(rows), mbs

    def detect_llr_rows_i16(self, raw_dev: int, len_dev: int, scale_dev: int, offset_dev: int, n: int, minibatch: int,
                            with_start_peak: bool = False, rows_dev: Optional[int] = None):
        """adp_detect_llr over RAW int16 samples resident on the device (per-read calibration applied in registers):
        -> (rows or None when rows_dev is given, mb_status)"""
        flags = ADP_IN_DEVICE | (ADP_WITH_START_PEAK if with_start_peak else 0)
        n_mb = (n + minibatch - 1) // minibatch
        mbs = np.zeros(n_mb, dtype=np.int32)
        if rows_dev is not None:
            flags |= ADP_OUT_DEVICE
            rows, rp = None, C.c_void_p(rows_dev)
        else:
            rows = np.zeros(n, dtype=ROW_DTYPE)
            rp = rows.ctypes.data_as(C.c_void_p)
        self._check(self.lib.adp_detect_llr_i16(self._h, C.c_void_p(int(raw_dev)), C.c_void_p(int(len_dev)), C.c_void_p(int(scale_dev)),
                                                C.c_void_p(int(offset_dev)), int(n), self.m, int(minibatch), flags, rp,
                                                mbs.ctypes.data_as(C.c_void_p)))
        return self.attach_open_pores(rows), mbs

    def detect_start_peak_rows(self, signals, full_lens, n: int, minibatch: int, device_ptrs: bool = False):
        """adp_detect_start_peak over n reads in ONE library call: the pandas float-column quirk couples the reads of a minibatch
        (the library walks the minibatches), the open-pore arena belongs to the call (offsets of every minibatch index it)"""
        sp, lp, flags, keep = self._in_ptrs(signals, full_lens, n, device_ptrs)
        rows = np.zeros(n, dtype=ROW_DTYPE)
        self._check(self.lib.adp_detect_start_peak(self._h, sp, lp, int(n), self.m, int(minibatch), flags,
                                                   rows.ctypes.data_as(C.c_void_p)))
        del keep
        return self.attach_open_pores(rows)

    def validate_rows(self, signals, full_lens, n: int, bounds: np.ndarray, device_ptrs: bool = False,
                      topk_none: bool = False):
        """bounds int64 [n, 1+k] (host) -> rows"""
        sp, lp, flags, keep = self._in_ptrs(signals, full_lens, n, device_ptrs)
        if topk_none:
            flags |= ADP_TOPK_NONE
        b = np.ascontiguousarray(bounds, dtype=np.int64)
        k = b.shape[1] - 1
        rows = np.zeros(n, dtype=ROW_DTYPE)
        if device_ptrs:
            flags |= ADP_BOUNDS_HOST  # the signals are resident, the candidate table comes from the host
        self._check(self.lib.adp_validate_candidates(self._h, sp, lp, int(n), self.m, b.ctypes.data_as(C.c_void_p), int(k),
                                                     flags, rows.ctypes.data_as(C.c_void_p)))
        del keep
        return self.attach_open_pores(rows)

    def c_llr_trace(self, raw, lens, starts, ends, args: "AdpTraceArgs", sums=None, return_c_c2: bool = False):
        """adp_c_llr_trace: the reference's `c_llr_trace` (`c_llr_trace_gains` with ``sums=(c, c2)``) for a batch -- raw float64
        [n, L], per-read lens / starts / ends -> gains float64 [n, L] (and c, c2)"""
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        starts = np.ascontiguousarray(starts, dtype=np.int32)
        ends = np.ascontiguousarray(ends, dtype=np.int32)
        flags = 0
        if sums is not None:
            c = np.ascontiguousarray(sums[0], dtype=np.float64)
            c2 = np.ascontiguousarray(sums[1], dtype=np.float64)
            n, L = c.shape
            rawp = None
            flags |= ADP_TRACE_FROM_SUMS
        else:
            raw = np.ascontiguousarray(raw, dtype=np.float64)
            n, L = raw.shape
            rawp = raw.ctypes.data_as(C.c_void_p)
            c = np.zeros((n, L)) if return_c_c2 else None
            c2 = np.zeros((n, L)) if return_c_c2 else None
        if not (lens.size == starts.size == ends.size == n):
            raise ValueError("lens / starts / ends need one entry per read")
        g = np.zeros((n, L))
        self._check(self.lib.adp_c_llr_trace(self._h, rawp, lens.ctypes.data_as(C.c_void_p), starts.ctypes.data_as(C.c_void_p),
                                             ends.ctypes.data_as(C.c_void_p), int(n), int(L), C.byref(args), flags,
                                             g.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p) if c is not None else None,
                                             c2.ctypes.data_as(C.c_void_p) if c2 is not None else None))
        return (g, c, c2) if return_c_c2 else g

    def cnn_topk(self, scores_ptr: int, adapter_pos_ptr: int, polya_pos_ptr: int, n: int, Lo: int, k: int):
        """the k > 1 part of C3 behind given arg-maxes (tests): (cand int32 [n, k], n_peaks int32 [n]); device pointers in"""
        cand = np.zeros((n, k), dtype=np.int32)
        cnt = np.zeros(n, dtype=np.int32)
        self._check(self.lib.adp_cnn_topk(self._h, C.c_void_p(int(scores_ptr)), C.c_void_p(int(adapter_pos_ptr)),
                                          C.c_void_p(int(polya_pos_ptr)), int(n), int(Lo), int(k), cand.ctypes.data_as(C.c_void_p),
                                          cnt.ctypes.data_as(C.c_void_p)))
        return cand, cnt

    def cnn_predict(self, scores_ptr: int, n: int, minibatch: int, Lo: int) -> np.ndarray:
        """C3 + the scaling of cnn_detect on the device: int64 [n, 1 + max(k, 1)] (adapter end, poly(A) candidates; samples)"""
        k = max(1, int(self.cfg.polya_cand_k))
        out = np.zeros((n, 1 + k), dtype=np.int64)
        self._check(self.lib.adp_cnn_predict(self._h, C.c_void_p(int(scores_ptr)), int(n), int(minibatch), int(Lo),
                                             out.ctypes.data_as(C.c_void_p)))
        return out

    def detect_cnn_rows(self, signals, full_lens, n: int, minibatch: int, device_ptrs: bool = False, rows_dev: Optional[int] = None,
                        want_bounds: bool = True):
        """combined_detect_cnn without the short-read fallback -> (rows or None when rows_dev is given, bounds int64 [n, 1 + k])"""
        sp, lp, flags, keep = self._in_ptrs(signals, full_lens, n, device_ptrs)
        k = max(1, int(self.cfg.polya_cand_k))
        bounds = np.zeros((n, 1 + k), dtype=np.int64) if want_bounds else None
        if rows_dev is not None:
            flags |= ADP_OUT_DEVICE
            rows, rp = None, C.c_void_p(rows_dev)
        else:
            rows = np.zeros(n, dtype=ROW_DTYPE)
            rp = rows.ctypes.data_as(C.c_void_p)
        self._check(self.lib.adp_detect_cnn(self._h, sp, lp, int(n), self.m, int(minibatch), flags, rp,
                                            bounds.ctypes.data_as(C.c_void_p) if want_bounds else None))
        del keep
        return self.attach_open_pores(rows), bounds

    def cnn_set_weights(self, state):
        """state: mapping with the reference's state-dict keys ("0.weight" ... "6.bias") -> float32 arrays (numpy, or anything
        np.asarray takes: torch CPU tensors included)"""
        arrs = []
        for key, shape in (("0.weight", (64, 1, 7)), ("0.bias", (64,)), ("2.weight", (64, 64, 7)), ("2.bias", (64,)),
                           ("4.weight", (64, 64, 7)), ("4.bias", (64,)), ("6.weight", (64, 2, 7)), ("6.bias", (2,))):
            a = state[key]
            if hasattr(a, "detach"):
                a = a.detach().cpu().numpy()
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.shape != shape:
                raise ValueError("CNN weight %s has shape %s, expected %s" % (key, a.shape, shape))
            arrs.append(a)
        self._check(self.lib.adp_cnn_set_weights(self._h, *[a.ctypes.data_as(C.c_void_p) for a in arrs]))
        self._cnn_weights_id = id(state)

    def cnn_forward(self, prepared_dev: int, n: int, Lc: int, scores_dev: int):
        """C2 on the device (hand-written conv stack): prepared float32 [n, Lc] -> scores float32 [n, 2, Lo]"""
        self._check(self.lib.adp_cnn_forward(self._h, C.c_void_p(int(prepared_dev)), int(n), int(Lc), C.c_void_p(int(scores_dev))))

    def cnn_prepare(self, signals, n: int, out_dev_ptr: int, device_ptrs: bool = False):
        """C1 into a device buffer float32 [n, Lc] (e.g. a torch tensor's data_ptr)."""
        if device_ptrs:
            self._check(self.lib.adp_cnn_prepare(self._h, C.c_void_p(int(signals)), int(n), self.m, ADP_OUT_DEVICE | ADP_IN_DEVICE,
                                                 C.c_void_p(int(out_dev_ptr))))
            return
        sig = np.ascontiguousarray(signals, dtype=np.float32)
        self._check(self.lib.adp_cnn_prepare(self._h, sig.ctypes.data_as(C.c_void_p), int(n), self.m, ADP_OUT_DEVICE,
                                             C.c_void_p(int(out_dev_ptr))))

    def llr_refine_polya(self, signals, full_lens, n: int, ranges: np.ndarray):
        sp, lp, flags, keep = self._in_ptrs(signals, full_lens, n, False)
        rg = np.ascontiguousarray(ranges, dtype=np.int64)
        out = np.zeros(n, dtype=np.int64)
        st = np.zeros(n, dtype=np.int32)
        self._check(self.lib.adp_llr_refine_polya(self._h, sp, lp, int(n), self.m, rg.ctypes.data_as(C.c_void_p), flags,
                                                  out.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p)))
        del keep
        return out, st

    # -- debug (tests) ------------------------------------------------------------------------
    def debug_divcheck(self, d: float, first_bits: int, count: int) -> int:
        out = C.c_uint64(0)
        self._check(self.lib.adp_debug_divcheck(self._h, C.c_float(d), C.c_uint32(first_bits), C.c_uint32(count), C.byref(out)))
        return int(out.value)

    def debug_log(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self._check(self.lib.adp_debug_log(self._h, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), int(x.size)))
        return y

    def debug_llr_upto(self, signals, full_lens, n, minibatch, stage):
        sp, lp, flags, keep = self._in_ptrs(signals, full_lens, n, False)
        self._check(self.lib.adp_debug_llr_upto(self._h, sp, lp, int(n), self.m, int(minibatch), flags, int(stage)))
        del keep

    def debug_fetch(self, what: int, n: int):
        lp = np.zeros(1, dtype=np.int32)
        self._check(self.lib.adp_debug_fetch(self._h, 6, lp.ctypes.data_as(C.c_void_p), C.c_uint64(4)))
        Lp = int(lp[0])
        if what == 6:
            return Lp
        shapes = {1: ((n,), np.int32), 2: ((n, Lp), np.float32), 3: ((n, Lp), np.float64), 4: ((n,), np.int32),
                  5: ((n,), np.int32), 7: ((n, 2), np.int32), 9: ((n,), np.int8)}
        if what == 0:
            raise ValueError("use debug_norm_params")
        shp, dt = shapes[what]
        a = np.zeros(shp, dtype=dt)
        self._check(self.lib.adp_debug_fetch(self._h, what, a.ctypes.data_as(C.c_void_p), C.c_uint64(a.nbytes)))
        return a

    def debug_counters(self, n: int = 8):
        a = np.zeros(n, dtype=np.uint64)
        self._check(self.lib.adp_debug_fetch(self._h, 8, a.ctypes.data_as(C.c_void_p), C.c_uint64(8 * n)))
        return a

    def debug_norm_params(self, n_mb: int):
        a = np.zeros((n_mb, 4), dtype=np.float64)
        self._check(self.lib.adp_debug_fetch(self._h, 0, a.ctypes.data_as(C.c_void_p), C.c_uint64(a.nbytes)))
        return a

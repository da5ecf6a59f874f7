"""Shared helpers of the parity tests."""
import json
import math
import os

import numpy as np

from adapted_amd import synth
from adapted_amd.config import get_chemistry_specific_config
from golden_cases import CASES, apply_blips, apply_extra, apply_overrides, apply_quantise, resolve_lens

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

INDEX_FIELDS = ["adapter_start", "adapter_end", "adapter_len", "polya_start", "polya_end", "polya_len",
                "rna_preloaded_start", "rna_preloaded_len", "start_peak_idx", "start_peak_next_max_idx",
                "start_peak_open_pore_idx", "llr_adapter_end", "llr_polya_end", "cnn_adapter_end",
                "cnn_polya_end", "start_peak_adapter_end", "start_peak_polya_end", "signal_len", "preloaded"]
SKIP_FIELDS = {"llr_detect_log", "polya_truncated", "llr_trace"}


def make_spc(case):
    spc = get_chemistry_specific_config(case["chem"])
    p = case["primary"]
    if p == "llr_single":
        p = "llr"
    spc.llr_boundaries.llr_detect = p == "llr"
    spc.cnn_boundaries.cnn_detect = p == "cnn"
    spc.rna_start_peak.detect_rna_start_peak = p == "start_peak"
    if case.get("max_obs_trace"):
        spc.core.max_obs_trace = case["max_obs_trace"]
    if "mvs_detect_check" in case:
        spc.mvs_polya.mvs_detect_check = case["mvs_detect_check"]
    if "detect_med_shift" in case:
        spc.med_shift.detect_med_shift = case["detect_med_shift"]
    apply_overrides(spc, case)
    spc.update_primary_method()
    spc.update_sig_preload_size()
    return spc


def load_case(name):
    """-> (case, spc, signals f32[n,m], lens i32[n], golden rows)"""
    case = CASES[name]
    spc = make_spc(case)
    with open(os.path.join(GOLD, name + ".rows.json")) as fh:
        g = json.load(fh)
    m = g["m"]
    assert m == spc.sig_preload_size
    lens = np.asarray(g["lens"], dtype=np.int32)
    assert list(lens) == resolve_lens(case["lens"], case["n"], m)
    sig, _ = synth.synth_batch(case["seed"], case["first"], case["n"], m, lens)
    apply_blips(sig, case)
    apply_extra(sig, lens, case)
    apply_quantise(sig, case)
    return case, spc, sig, lens, g["rows"]


def load_stages(name):
    return np.load(os.path.join(GOLD, name + ".stages.npz"))


def _same(a, b, rel):
    if a is None or b is None:
        return a is None and b is None
    if isinstance(b, (list, tuple)):
        return isinstance(a, (list, tuple)) and len(a) == len(b) and all(_same(x, y, rel) for x, y in zip(a, b))
    if isinstance(b, float) or isinstance(a, float):
        a, b = float(a), float(b)
        if math.isnan(a) or math.isnan(b):
            return math.isnan(a) and math.isnan(b)
        if rel == 0:
            return a == b
        return abs(a - b) <= rel * max(abs(a), abs(b), 1e-30)
    return a == b


def row_diffs(got, want, float_rel=0.0):
    """List of (field, got, want) that differ.  ``got`` may be a dict or a DetectResults.
    Integer/bool/str fields are always compared exactly; floats within ``float_rel``."""
    g = got if isinstance(got, dict) else got.__dict__
    out = []
    if g.get("_exception"):
        keys = ("success", "fail_reason")
        for k, v in want.items():
            w = g.get(k) if k in keys else None
            if k in SKIP_FIELDS:
                continue
            if not _same(w, v, 0):
                out.append((k, w, v))
        return out
    for k, v in want.items():
        if k in SKIP_FIELDS:
            continue
        w = g.get(k)
        if hasattr(w, "tolist"):
            w = w.tolist()
        if not _same(w, v, float_rel):
            out.append((k, w, v))
    return out

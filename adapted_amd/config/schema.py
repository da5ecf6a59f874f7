"""Signal-processing configuration tree (the TOML surface of ``adapted detect``).

Mirrors the section/key names, types and defaults of the reference's config dataclasses
(reference adapted/config/sig_proc.py:22-221) so that a reference ``config.toml`` loads
unchanged and the dumped ``config.toml`` reads back into the reference.  The classes are
generated from the declarative table below; chemistry presets are override tables (the
values of reference adapted/config/config_files/rna00{2,4}_*@v0.2.4.toml).
"""
from __future__ import annotations

import copy
import dataclasses
import math
from typing import Any, Dict, Optional, Tuple

INF = math.inf
Range = Tuple[Optional[float], Optional[float]]

# section -> [(key, type, default)]
SECTIONS: Dict[str, list] = {
    "core": [
        ("min_obs_adapter", int, 1000), ("max_obs_adapter", int, 6500), ("min_obs_polya", int, 100),
        ("downscale_factor", int, 10), ("max_obs_trace", int, 16000),
        ("sig_norm_outlier_thresh", float, 5.0),
    ],
    "llr_boundaries": [
        ("llr_detect", bool, False), ("adapter_peak_prominence", float, 1.0),
        ("adapter_peak_rel_height", float, 1.0), ("adapter_peak_width", int, 1000),
        ("polya_peak_prominence", float, 1.0), ("polya_peak_rel_height", float, 0.5),
        ("polya_peak_width", int, 50),
    ],
    "mvs_polya": [
        ("mvs_detect_check", bool, True), ("mvs_detect_overwrite", bool, False),
        ("search_window", int, 500), ("pA_mean_window", int, 20), ("pA_mean_range", Range, (None, None)),
        ("pA_var_window", int, 100), ("pA_var_range", Range, (None, 20.0)),
        ("median_shift_range", Range, (20.0, None)), ("median_shift_window", int, 2000),
        ("polyA_window", int, 300), ("polyA_med_range", Range, (90.0, 130.0)),
        ("polyA_local_range", Range, (0.0, 15.0)),
        ("pA_mean_adapter_med_scale_range", Range, (1.3, None)),
    ],
    "real_range": [
        ("detect_open_pores", bool, True), ("real_signal_check", bool, True), ("mean_window", int, 300),
        ("mean_start_range", Range, (50.0, 100.0)), ("mean_end_range", Range, (75.0, 120.0)),
        ("max_obs_local_range", int, 5000), ("local_range", Range, (10.0, 30.0)),
        ("adapter_mad_range", Range, (3.0, 12.0)),
    ],
    "streaming": [  # accepted for TOML compatibility; unused by the detect path
        ("min_obs_adapter", int, 2500), ("min_obs_post_loc", int, 300), ("search_increment_step", int, 100),
        ("pA_mean_window", int, 20), ("pA_mean_range", Range, (90.0, 130.0)), ("pA_var_window", int, 100),
        ("pA_var_range", Range, (None, 20.0)), ("median_shift_window", int, 2000),
        ("median_shift_range", Range, (20.0, None)), ("polyA_window", int, 300),
        ("polyA_med_range", Range, (90.0, 130.0)), ("polyA_local_range", Range, (0.0, 10.0)),
    ],
    "cnn_boundaries": [
        ("cnn_detect", bool, True), ("model_name", str, "rna004_130bps@v0.2.4.pth"),
        ("polya_cand_k", int, 15), ("fallback_to_llr_short_reads", bool, True),
    ],
    "med_shift": [
        ("detect_med_shift", bool, False), ("med_shift_window", int, 2000),
        ("med_shift_range", Range, (20.0, None)),
    ],
    "rna_start_peak": [
        ("detect_rna_start_peak", bool, False), ("downscale_factor", int, 10),
        ("start_peak_max_idx", int, 150), ("offset1", int, 10), ("offset2", int, 100),
        ("open_pore_pa", float, 195.0),
    ],
}

_COMMON_PRESET = {
    "mvs_polya": dict(mvs_detect_check=True, mvs_detect_overwrite=False, search_window=500,
                      pA_mean_window=20, pA_var_window=100, median_shift_range=(5.0, INF),
                      median_shift_window=1000, polyA_med_range=(-INF, INF), polyA_local_range=(-INF, INF),
                      pA_mean_adapter_med_scale_range=(1.3, INF)),
    "real_range": dict(detect_open_pores=True, real_signal_check=True, mean_window=300,
                       mean_start_range=(-INF, INF), mean_end_range=(-INF, INF),
                       max_obs_local_range=5000, local_range=(7.0, 35.0), adapter_mad_range=(3.0, 12.0)),
}

PRESETS: Dict[str, Dict[str, Dict[str, Any]]] = {
    "rna004": {
        "core": dict(max_obs_trace=16000, min_obs_adapter=1000, max_obs_adapter=6500, min_obs_polya=100,
                     downscale_factor=10, sig_norm_outlier_thresh=5.0),
        "cnn_boundaries": dict(cnn_detect=True, model_name="rna004_130bps@v0.2.4.pth", polya_cand_k=10,
                               fallback_to_llr_short_reads=True),
        "llr_boundaries": dict(llr_detect=False, adapter_peak_prominence=1.0, adapter_peak_rel_height=1.0,
                               adapter_peak_width=1000, polya_peak_prominence=1.0,
                               polya_peak_rel_height=0.5, polya_peak_width=50),
        "mvs_polya": dict(_COMMON_PRESET["mvs_polya"], pA_var_range=(-INF, 30.0)),
        "real_range": _COMMON_PRESET["real_range"],
        "med_shift": dict(detect_med_shift=False, med_shift_window=2000, med_shift_range=(5.0, INF)),
        "rna_start_peak": dict(detect_rna_start_peak=False, downscale_factor=10, start_peak_max_idx=150,
                               offset1=10, offset2=100, open_pore_pa=195.0),
    },
    "rna002": {
        "core": dict(max_obs_trace=25000, min_obs_adapter=2000, max_obs_adapter=12000, min_obs_polya=100,
                     downscale_factor=20, sig_norm_outlier_thresh=5.0),
        "cnn_boundaries": dict(cnn_detect=False, model_name="rna002_70bps@v0.2.4.pth", polya_cand_k=15,
                               fallback_to_llr_short_reads=True),
        "llr_boundaries": dict(llr_detect=True, adapter_peak_prominence=1.0, adapter_peak_rel_height=1.0,
                               adapter_peak_width=1500, polya_peak_prominence=1.0,
                               polya_peak_rel_height=0.5, polya_peak_width=50),
        "mvs_polya": dict(_COMMON_PRESET["mvs_polya"], pA_var_range=(-INF, 20.0)),
        "real_range": _COMMON_PRESET["real_range"],
        "med_shift": dict(detect_med_shift=False, med_shift_window=1000, med_shift_range=(5.0, INF)),
        "rna_start_peak": dict(detect_rna_start_peak=False),
    },
}

SPEEDS = {"rna002": "70bps", "rna004": "130bps"}


class _Section:
    """Base of the generated section classes: dict-style access, copy, dict()."""

    def __getitem__(self, key):
        if not hasattr(self, key):
            raise KeyError("%r has no attribute %r" % (type(self).__name__, key))
        return getattr(self, key)

    def __setitem__(self, key, value):
        if not hasattr(self, key):
            raise KeyError("%r has no attribute %r" % (type(self).__name__, key))
        setattr(self, key, value)

    def dict(self):
        return dataclasses.asdict(self)

    def copy(self):
        return copy.deepcopy(self)

    def typed_dict(self):
        """Values coerced to their declared types; ranges with None -> -inf/inf (what
        the reference writes into <run_dir>/config.toml, adapted/config/base.py:52-86)."""
        out = {}
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            kind = f.metadata.get("kind")
            if kind is Range:
                v = tuple(v) if v is not None else (None, None)
                if f.name.endswith("_range") and len(v) == 2:
                    v = [-INF if v[0] is None else float(v[0]), INF if v[1] is None else float(v[1])]
            elif kind in (bool, int, float, str) and v is not None:
                v = kind(v)
            out[f.name] = v
        return out


def _make(section: str, fields):
    cols = []
    for key, kind, default in fields:
        if isinstance(default, (tuple, list)):
            fld = dataclasses.field(default_factory=lambda d=default: tuple(d), metadata={"kind": kind})
        else:
            fld = dataclasses.field(default=default, metadata={"kind": kind})
        cols.append((key, Any, fld))
    name = "".join(p.capitalize() for p in section.split("_")) + "Config"
    return dataclasses.make_dataclass(name, cols, bases=(_Section,))


CoreConfig = _make("core", SECTIONS["core"])
LLRBoundariesConfig = _make("llr_boundaries", SECTIONS["llr_boundaries"])
MVSPolyAConfig = _make("mvs_polya", SECTIONS["mvs_polya"])
RealRangeConfig = _make("real_range", SECTIONS["real_range"])
StreamingConfig = _make("streaming", SECTIONS["streaming"])
CNNBoundariesConfig = _make("cnn_boundaries", SECTIONS["cnn_boundaries"])
MedShiftConfig = _make("med_shift", SECTIONS["med_shift"])
RNAStartPeakConfig = _make("rna_start_peak", SECTIONS["rna_start_peak"])

SECTION_CLASSES = {
    "core": CoreConfig, "llr_boundaries": LLRBoundariesConfig, "mvs_polya": MVSPolyAConfig,
    "real_range": RealRangeConfig, "streaming": StreamingConfig, "cnn_boundaries": CNNBoundariesConfig,
    "med_shift": MedShiftConfig, "rna_start_peak": RNAStartPeakConfig,
}

"""Result containers at the drop-in boundary.

Field names and ORDER of ``DetectResults`` are the CSV schema of
``detected_boundaries_*.csv`` / ``failed_reads_*.csv`` (reference
adapted/container_types.py:8-120, adapted/output.py:26-51); they are generated from the
column table below so the table is the single statement of the schema.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Dict, Optional

# (field, kind) in CSV order; kind: i=int, f=float, b=bool, a=array, s=str
DETECT_COLUMNS = [
    ("signal_len", "i"), ("preloaded", "i"),
    ("adapter_start", "i"), ("adapter_end", "i"), ("adapter_len", "i"),
    ("adapter_mean", "f"), ("adapter_std", "f"), ("adapter_med", "f"), ("adapter_mad", "f"),
    ("polya_start", "i"), ("polya_end", "i"), ("polya_len", "i"),
    ("polya_mean", "f"), ("polya_std", "f"), ("polya_med", "f"), ("polya_mad", "f"),
    ("polya_truncated", "b"), ("polya_candidates", "a"),
    ("rna_preloaded_start", "i"), ("rna_preloaded_len", "i"),
    ("rna_preloaded_mean", "f"), ("rna_preloaded_std", "f"), ("rna_preloaded_med", "f"), ("rna_preloaded_mad", "f"),
    ("start_peak_idx", "i"), ("start_peak_pa", "f"), ("start_peak_next_max_idx", "i"),
    ("start_peak_next_max_pa", "f"), ("start_peak_open_pore_idx", "i"), ("start_peak_open_pore_type", "s"),
    ("adapter_rna_median_shift", "f"),
    ("llr_adapter_end", "i"), ("llr_polya_end", "i"), ("cnn_adapter_end", "i"), ("cnn_polya_end", "i"),
    ("start_peak_adapter_end", "i"), ("start_peak_polya_end", "i"),
    ("llr_trace", "a"), ("llr_adapter_end_adjust", "i"), ("llr_polya_end_adjust", "i"),
    ("llr_trace_early_stop_pos", "i"),
    ("mvs_llr_polya_end_adjust_ignored", "b"), ("mvs_llr_polya_end_to_early_stop", "b"),
    ("mvs_adapter_end", "i"), ("mvs_detect_mean_at_loc", "f"), ("mvs_detect_var_at_loc", "f"),
    ("mvs_detect_polya_med", "f"), ("mvs_detect_polya_local_range", "f"), ("mvs_detect_med_shift", "f"),
    ("real_adapter_mean_start", "f"), ("real_adapter_mean_end", "f"), ("real_adapter_local_range", "f"),
    ("open_pores", "a"), ("fail_reason", "s"), ("llr_detect_log", "s"),
]


class _DictMixin:
    def to_dict(self) -> Dict[str, Any]:
        return dict(self.__dict__)

    def update(self, d: Dict[str, Any]):
        self.__dict__.update(d)


DetectResults = dataclasses.make_dataclass(
    "DetectResults",
    [("success", bool)] + [(name, Optional[Any], dataclasses.field(default=None)) for name, _ in DETECT_COLUMNS],
    bases=(_DictMixin,),
)
DetectResults.__doc__ = "One read's boundaries and statistics (success first, then the CSV columns)."


@dataclasses.dataclass
class Boundaries:
    """Primary-detector proposal handed to the validator."""
    adapter_start: int
    adapter_end: int
    polya_end: int
    polya_end_topk: Optional[Any] = None
    adapter_end_adjust: Optional[int] = None
    polya_end_adjust: Optional[int] = None
    trace: Optional[Any] = None
    trace_early_stop_pos: Optional[int] = None
    logstr: Optional[str] = None
    polya_truncated: Optional[bool] = None
    debug_logger: Optional[dict] = None


@dataclasses.dataclass
class ReadResult:
    read_id: Optional[str] = None
    success: bool = True
    fail_reason: Optional[str] = None
    detect_results: Optional[Any] = None

    def to_summary_dict(self) -> Dict[str, Any]:
        d = self.detect_results.to_dict() if self.detect_results else {}
        d.pop("fail_reason", None)
        return {"read_id": self.read_id, **d, "fail_reason": self.fail_reason}

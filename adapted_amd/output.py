"""CSV writer for detected boundaries (reference adapted/output.py:26-51).

Columns = ``read_id`` + the DetectResults fields in order, minus ``success``/``llr_trace``
(and ``fail_reason`` for the pass file, where it moves last for the fail file); values are
rounded to 3 decimals exactly like the reference (pandas ``DataFrame.round(3).to_csv``).
"""
from __future__ import annotations

from typing import List

import numpy as np
import pandas as pd

from .container_types import DETECT_COLUMNS, ReadResult


def save_traces(results: List[ReadResult], filename: str) -> None:
    traces = {str(r.read_id): r.detect_results.llr_trace for r in results
              if r.detect_results is not None and r.detect_results.llr_trace is not None}
    np.savez(filename, **traces)


def results_frame(processing_results: List[ReadResult], save_fail_reasons: bool = False) -> pd.DataFrame:
    df = pd.DataFrame([pr.to_summary_dict() for pr in processing_results])
    if not df.empty:
        drop = ["success", "llr_trace"] + ([] if save_fail_reasons else ["fail_reason"])
        df = df.drop(columns=[c for c in drop if c in df.columns])
    return df


def save_detected_boundaries(processing_results: List[ReadResult], filename: str, save_fail_reasons: bool = False):
    """Save detected boundaries and read ids to a csv file."""
    results_frame(processing_results, save_fail_reasons).round(3).to_csv(filename, index=False)


CSV_COLUMNS = ["read_id"] + [c for c, _ in DETECT_COLUMNS if c not in ("llr_trace", "fail_reason")]

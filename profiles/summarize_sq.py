#!/usr/bin/env python3
"""Summarise SQ counter passes (rocprofv3 --kernel-trace --pmc ..., --output-format csv) per kernel: sums over the launches
of the LAST profiled step are not separated -- values are averaged per dispatch of each kernel name.

usage: summarize_sq.py <counter_collection.csv> [<counter_collection.csv> ...] <out.json>

Derived (MI355X_MICROARCH.md, rocprofv3 PMC slots: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave):
  active_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES     share of a wave's lifetime spent issuing
  valu_frac   = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES    ... issuing vector ALU instructions
  wait_frac   = SQ_WAIT_ANY / SQ_WAVE_CYCLES            parked on s_waitcnt / barriers (memory, LDS latency)
  stall_frac  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES       ready but not issued (pipe busy, dependency)
  waves_per_simd = SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / 4 x ... (occupancy estimate: wave-cycles per busy SQ cycle, / 4 SIMDs)
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import kname  # noqa: E402


def main():
    files, out = sys.argv[1:-1], sys.argv[-1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)  # kernel -> {dispatch id: duration in ns} (where the pass's rows carry time stamps)
    for f in files:
        for row in csv.DictReader(open(f)):
            k = kname(row["Kernel_Name"])
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                dur[k][(f, row.get("Dispatch_Id"))] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    res = {}
    for k, cs in acc.items():
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        wc = d.get("SQ_WAVE_CYCLES")
        if wc:
            for name, c in (("active_frac", "SQ_ACTIVE_INST_ANY"), ("valu_frac", "SQ_ACTIVE_INST_VALU"), ("wait_frac", "SQ_WAIT_ANY"),
                            ("stall_frac", "SQ_WAIT_INST_ANY")):
                if c in d:
                    d[name] = d[c] / wc
            if d.get("SQ_BUSY_CYCLES"):
                d["wave_cycles_per_busy_cycle"] = wc / d["SQ_BUSY_CYCLES"]
            if d.get("SQ_WAVES") and d.get("SQ_INSTS_VALU"):
                d["valu_insts_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        if dur.get(k):
            d["avg_duration_ms_under_counters"] = sum(dur[k].values()) / len(dur[k]) * 1e-6
            if d.get("GRBM_GUI_ACTIVE"):
                # MI355X_MICROARCH.md (DVFS give-back): effective clock = GRBM_GUI_ACTIVE / 8 XCDs / wall time (dispatches >= 0.3 ms)
                d["effective_clock_GHz"] = d["GRBM_GUI_ACTIVE"] / 8.0 / (d["avg_duration_ms_under_counters"] * 1e6)
                if d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                    d["mfma_busy_of_actual_simd_cycles"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        d["dispatches"] = max(len(v) for v in cs.values())
        res[k] = d
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k in sorted(res, key=lambda k: -res[k].get("SQ_WAVE_CYCLES", 0))[:14]:
        d = res[k]
        print("%-22s active %.2f valu %.2f wait %.2f stall %.2f  valu/wave %9.0f  lds_conf %s  clock %s GHz  mfma busy %s" % (
            k[:22], d.get("active_frac", 0), d.get("valu_frac", 0), d.get("wait_frac", 0), d.get("stall_frac", 0), d.get("valu_insts_per_wave", 0),
            ("%.3g" % d["SQ_LDS_BANK_CONFLICT"]) if "SQ_LDS_BANK_CONFLICT" in d else "-",
            ("%.2f" % d["effective_clock_GHz"]) if "effective_clock_GHz" in d else "-",
            ("%.2f" % d["mfma_busy_of_actual_simd_cycles"]) if "mfma_busy_of_actual_simd_cycles" in d else "-"))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel average durations of a rocprofv3 --kernel-trace run of `python bench.py`, split by WORKLOAD.

The default bench run holds seven workloads in one process (the headline, then the secondaries cnn_200k, cnn_200k_f32_stack,
cnn_default, pareto, llr_default_window, int16; grouped with --with-grouped); each begins with the synthetic generator (k_synth), which separates them in the trace.  Inside a workload the
launches are grouped by kernel name and grid size (the CPU-baseline check of the headline runs the same kernels on one
minibatch: a different grid).  Durations are in microseconds, over ALL launches of the group (warm-up steps included).

usage: summarize_phases.py <results.db> <out.csv> [names of the phases, default headline cnn_200k cnn_200k_f32_stack cnn_default pareto llr_default_window int16 grouped]
"""
import csv
import sqlite3
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import kname  # noqa: E402


def main():
    db, out = sys.argv[1], sys.argv[2]
    names = sys.argv[3:] or ["headline", "cnn_200k", "cnn_200k_f32_stack", "cnn_default", "pareto", "llr_default_window", "int16", "grouped"]
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, duration, grid_x, grid_y, workgroup_x from kernels order by start").fetchall()
    phase = -1
    acc = {}
    order = []
    for name, start, dur, gx, gy, wx in rows:
        k = kname(name)
        if k == "k_synth":
            if phase < 0 or acc.get((phase, "k_synth", 0), [0, 0, 0])[2] != gy * 1000003 + gx:
                pass
            # consecutive k_synth launches (one per engine) belong to one workload: a new phase starts when the previous
            # launch was not a k_synth
            if not order or order[-1] != "k_synth":
                phase += 1
        order.append(k)
        key = (phase, k, gx // max(wx, 1) * max(gy, 1))
        a = acc.setdefault(key, [0, 0.0, 0])
        a[0] += 1
        a[1] += dur / 1e3
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Workload", "Name", "Workgroups", "Calls", "TotalDurationUs", "AverageUs"])
        for (ph, k, wg), (calls, tot, _) in sorted(acc.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
            w.writerow([names[ph] if 0 <= ph < len(names) else "phase%d" % ph, k, wg, calls, "%.3f" % tot, "%.3f" % (tot / calls)])
    for (ph, k, wg), (calls, tot, _) in sorted(acc.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
        if tot / calls > 300:
            print("%-12s %-24s wg %9d calls %3d avg %10.1f us" % (names[ph] if 0 <= ph < len(names) else ph, k[:24], wg, calls, tot / calls))


if __name__ == "__main__":
    main()

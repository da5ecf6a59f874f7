#!/usr/bin/env python3
"""Summarise rocprofv3 output (kernel stats + separate --pmc FETCH_SIZE / WRITE_SIZE passes) into the
per-launch HBM traffic of each kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (section HBM)
prescribes for gfx950: FETCH_SIZE (KB) counts 64 B per 128-B request on streaming reads -> doubled;
WRITE_SIZE (KB) is taken as is.  The correction is cross-checked on k_n1_fused, whose true read volume
is known exactly (n_reads * T * 4 bytes).

usage: summarize_pmc.py <fetch pass: counter_collection.csv | results.db> <write pass: ...> <reads_per_launch> <T> <out.json>
       summarize_pmc.py --stats <results.db> <out.csv>     (kernel stats of a --kernel-trace --stats run)
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import kname  # noqa: E402


def per_kernel_max(path):
    """largest per-dispatch value of the (single) counter of a pass; CSV (--output-format csv) or rocpd .db"""
    best = collections.defaultdict(float)
    if path.endswith(".db"):
        import sqlite3

        rows = sqlite3.connect(path).execute("select kernel_name, value from counters_collection")
        for name, value in rows:
            k = kname(name)
            best[k] = max(best[k], float(value))
        return best
    for row in csv.DictReader(open(path)):
        k = kname(row["Kernel_Name"])
        best[k] = max(best[k], float(row["Counter_Value"]))
    return best


def stats(db, out):
    import sqlite3

    rows = sqlite3.connect(db).execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([kname(name), calls, "%.3f" % tot, "%.3f" % avg, "%.4f" % pct])
    for r in rows[:14]:
        print("%-24s calls %3d  avg %10.1f us  %5.1f %%" % (kname(r[0]), r[1], r[3], r[4]))


def main():
    if sys.argv[1] == "--stats":
        return stats(sys.argv[2], sys.argv[3])
    fetch, write, reads, T, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    f, w = per_kernel_max(fetch), per_kernel_max(write)
    res = {"_note": "bytes per launch of the largest dispatch; fetch = 2 * FETCH_SIZE * 1024, write = WRITE_SIZE * 1024",
           "_reads_per_launch": reads}
    for k in sorted(set(f) | set(w)):
        fb, wb = 2.0 * f.get(k, 0.0) * 1024.0, w.get(k, 0.0) * 1024.0
        res[k] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb}
    known = reads * T * 4.0
    if "k_n1_fused" in res:
        res["_calibration"] = {"kernel": "k_n1_fused", "true_read_bytes": known,
                               "corrected_fetch_bytes": res["k_n1_fused"]["fetch_bytes"],
                               "ratio": res["k_n1_fused"]["fetch_bytes"] / known}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(((k, v) for k, v in res.items() if not k.startswith("_")), key=lambda kv: -kv[1]["hbm_bytes"])[:16]:
        print("%-28s fetch %8.2f GB  write %7.2f GB" % (k, v["fetch_bytes"] / 1e9, v["write_bytes"] / 1e9))
    print(res.get("_calibration"))


if __name__ == "__main__":
    main()

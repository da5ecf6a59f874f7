#!/usr/bin/env python3
"""Concurrency report of a rocprofv3 --kernel-trace run (results .db): which kernels of the LLR path actually ran beside
each other.  For the launches of the LAST `--steps` occurrences of k_partition_stats' groups (i.e. the tail of the trace)
it prints: wall time covered by at least one kernel, the sum of kernel durations, the time with two or more kernels
resident, the pairwise overlap between resource classes (streaming / float64 ALU / latency-bound), per-kernel average
durations, and a coarse per-queue ASCII timeline.

usage: timeline.py <results.db> [window_ms_from_the_end (default: the last 400 ms)] [out.json]"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles"))
from kname import kname  # noqa: E402

CLASS = {"k_n1_fused": "stream", "k_norm_pool": "stream", "k_partition_stats": "stream", "k_start_peak": "stream", "k_n1_hist": "stream",
         "k_gains": "alu", "k_cumsum": "alu",
         "k_validate": "latency", "k_polya_peak": "latency", "k_adapter_peak": "latency", "k_mvs_series": "latency"}


def cls(k):
    for p, c in CLASS.items():
        if k.startswith(p):
            return c
    return "other"


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def both(a, b):
    """time during which some interval of a AND some interval of b are active"""
    ev = [(s, 0, 1) for s, e in a] + [(e, 0, -1) for s, e in a] + [(s, 1, 1) for s, e in b] + [(e, 1, -1) for s, e in b]
    ev.sort()
    n = [0, 0]
    last, tot = None, 0
    for t, w, d in ev:
        if last is not None and n[0] > 0 and n[1] > 0:
            tot += t - last
        n[w] += d
        last = t
    return tot


def main():
    db = sys.argv[1]
    win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    scol = "stream_id" if "stream_id" in cols else ("stream" if "stream" in cols else None)
    sel = "name, start, end" + (", " + qcol if qcol else ", 0") + (", " + scol if scol else ", 0")
    if "end" not in cols:
        sel = sel.replace("end", "start + duration")
    rows = con.execute("select %s from kernels order by start" % sel).fetchall()
    t_end = max(r[2] for r in rows)
    t0 = t_end - int(win_ms * 1e6)
    rows = [(kname(n), s, e, q, st) for n, s, e, q, st in rows if s >= t0]
    t0 = min(r[1] for r in rows)
    by_cls, by_name, by_q = {}, {}, {}
    for k, s, e, q, st in rows:
        by_cls.setdefault(cls(k), []).append((s, e))
        by_name.setdefault(k, []).append(e - s)
        by_q.setdefault((q, st), []).append((s, e, k))
    allv = [(s, e) for k, s, e, q, st in rows]
    wall = union(allv)
    total = sum(e - s for s, e in allv)
    # time with >= 2 kernels resident
    ev = sorted([(s, 1) for s, e in allv] + [(e, -1) for s, e in allv])
    n, last, multi = 0, None, 0
    for t, d in ev:
        if last is not None and n >= 2:
            multi += t - last
        n += d
        last = t
    out = {"window_ms": (t_end - t0) / 1e6, "busy_ms": wall / 1e6, "sum_of_kernel_ms": total / 1e6, "two_or_more_resident_ms": multi / 1e6,
           "columns_used": {"queue": qcol, "stream": scol},
           "class_busy_ms": {c: union(v) / 1e6 for c, v in by_cls.items()},
           "class_overlap_ms": {a + "&" + b: both(by_cls[a], by_cls[b]) / 1e6 for a in by_cls for b in by_cls if a < b},
           "kernel_avg_ms": {k: sum(v) / len(v) / 1e6 for k, v in sorted(by_name.items(), key=lambda kv: -sum(kv[1])) if sum(v) > 2e5},
           "kernel_calls": {k: len(v) for k, v in by_name.items() if sum(v) > 2e5}}
    print(json.dumps(out, indent=1))
    # ASCII timeline: one row per (queue, stream), 160 columns over the window; letter = class of the kernel covering most of the cell
    W = 160
    span = t_end - t0
    letter = {"stream": "S", "alu": "G", "latency": "v", "other": "."}
    for key in sorted(by_q):
        cells = [" "] * W
        for s, e, k in by_q[key]:
            a, b = int((s - t0) * W / span), int((e - t0) * W / span)
            for c in range(a, min(W, b + 1)):
                cells[c] = letter[cls(k)]
        print("q%-3s s%-4s |%s|" % (key[0], key[1], "".join(cells)))
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()

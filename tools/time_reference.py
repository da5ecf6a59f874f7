#!/usr/bin/env python3
"""Time the REAL reference (KleistLab/ADAPTed at /root/reference) and the CPU port (oracle/adapted_oracle.c) side by side.

BUILD CONTAINER ONLY (the reference cannot travel to the GPU box): run under the interpreter that has the reference's real
third-party stack,

    /opt/conda/bin/python3.9 tools/time_reference.py [--procs 8] [--reads 200] [--out profiles/r02_reference_timing.json]

Layout = the reference's own (adapted/file_proc.py:738-784): a ProcessPoolExecutor with P workers, ONE minibatch per
task, detect-only timing (the minibatches are generated inside the workers before the clock starts; no pod5, no queues).
Workload = bench.py's headline: RNA004 preset, LLR primary, --max_obs_trace 200000 (m = 201 500), synthetic reads of
adapted_amd/synth.py (same seed as bench.py), every read filling the window.  The port runs the same minibatches in the
same process layout (plus the start-peak scan the bench includes, whose cost is ~0 in the reference too: SURVEY section 6),
which gives the port / reference ratio bench.py uses to turn its port timing on the GPU box's host cores into a
reference-equivalent figure.
"""
import argparse
import importlib.util
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _synth():
    spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "adapted_amd", "synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _minibatch(seed, first, n, m):
    import numpy as np

    synth = _synth()
    sig, lens = synth.synth_batch(seed, first, n, m, np.full(n, m, dtype=np.int32))
    return sig, lens


def work_reference(args):
    seed, first, n, max_obs_trace = args
    import warnings

    import ref_harness

    ref_harness.install()
    from adapted.config.sig_proc import get_chemistry_specific_config
    from adapted.detect import combined

    spc = get_chemistry_specific_config("RNA004")
    spc.llr_boundaries.llr_detect = True
    spc.cnn_boundaries.cnn_detect = False
    spc.core.max_obs_trace = max_obs_trace
    spc.update_primary_method()
    spc.update_sig_preload_size()
    sig, lens = _minibatch(seed, first, n, spc.sig_preload_size)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        combined.combined_detect_llr2(sig[:2].copy(), lens[:2], spc)  # (imports, pyximport)
        t0 = time.perf_counter()
        res = combined.combined_detect_llr2(sig, lens, spc)
        dt = time.perf_counter() - t0
    return dt, sum(bool(r.success) for r in res)


def work_port(args):
    seed, first, n, max_obs_trace = args
    # the port's Python binding lives in oracle/oracle.py and takes the build's own config tree
    sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location("orc", os.path.join(ROOT, "oracle", "oracle.py"))
    orc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(orc)
    from adapted_amd.config import get_chemistry_specific_config

    spc = get_chemistry_specific_config("RNA004")
    spc.llr_boundaries.llr_detect = True
    spc.cnn_boundaries.cnn_detect = False
    spc.core.max_obs_trace = max_obs_trace
    spc.update_primary_method()
    spc.update_sig_preload_size()
    orc.lib()
    sig, lens = _minibatch(seed, first, n, spc.sig_preload_size)
    t0 = time.perf_counter()
    res = orc.detect_llr(sig, lens, spc, with_start_peak=True)
    dt = time.perf_counter() - t0
    return dt, sum(bool(r["success"]) for r in res)


def run(fn, procs, reads, seed, max_obs_trace):
    tasks = [(seed, k * reads, reads, max_obs_trace) for k in range(procs)]
    t0 = time.perf_counter()
    with ProcessPoolExecutor(procs) as ex:
        out = list(ex.map(fn, tasks))
    wall = time.perf_counter() - t0
    dts = [d for d, _ in out]
    return {"procs": procs, "reads_per_minibatch": reads, "detect_s_per_minibatch": dts,
            "reads_per_s_per_proc": reads / (sum(dts) / len(dts)), "reads_per_s_all_procs": procs * reads / max(dts),
            "pass": sum(p for _, p in out), "wall_s_including_generation": wall}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=os.cpu_count())
    ap.add_argument("--reads", type=int, default=200, help="reads per minibatch (one minibatch per process)")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--max_obs_trace", type=int, default=200000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_reference_timing.json"))
    a = ap.parse_args()
    import numpy
    import scipy

    one_ref = run(work_reference, 1, a.reads, a.seed, a.max_obs_trace)
    one_port = run(work_port, 1, a.reads, a.seed, a.max_obs_trace)
    all_ref = run(work_reference, a.procs, a.reads, a.seed, a.max_obs_trace)
    all_port = run(work_port, a.procs, a.reads, a.seed, a.max_obs_trace)
    out = {
        "what": "reference (KleistLab/ADAPTed v0.2.4, combined_detect_llr2) vs the CPU port (oracle/adapted_oracle.c), detect only, "
                "RNA004 LLR, max_obs_trace=%d, synthetic full-window reads, one minibatch per process" % a.max_obs_trace,
        "host": {"cpus": os.cpu_count(), "python": sys.version.split()[0], "numpy": numpy.__version__, "scipy": scipy.__version__},
        "reference_1_proc": one_ref, "port_1_proc": one_port, "reference_all_procs": all_ref, "port_all_procs": all_port,
        "port_over_reference_per_proc": one_port["reads_per_s_per_proc"] / one_ref["reads_per_s_per_proc"],
        "port_over_reference_all_procs": all_port["reads_per_s_all_procs"] / all_ref["reads_per_s_all_procs"],
    }
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

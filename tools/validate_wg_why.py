#!/usr/bin/env python3
"""Which reads k_validate_wg (the workgroup-per-read fast path of V1-V4) leaves to k_validate, and why (GPU box).
    python tools/validate_wg_why.py [--reads 4000] [--max_obs_trace 16000] [--primary llr|cnn] [--lens full|pareto] [--adc-step 0.18]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WHY = {0: "handled", 1: "size", 2: "nan", 3: "list", 4: "exception row", 5: "no series"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4000)
    ap.add_argument("--max_obs_trace", type=int, default=16000)
    ap.add_argument("--primary", default="llr")
    ap.add_argument("--lens", default="full")
    ap.add_argument("--adc-step", type=float, default=0.0)
    a = ap.parse_args()
    import bench
    from adapted_amd import lib, synth

    spc = bench.make_spc(a.max_obs_trace, a.primary)
    m, n = spc.sig_preload_size, a.reads
    lens = np.full(n, m, dtype=np.int32)
    if a.lens == "pareto":
        lens = np.array([synth.pareto_length(2024, i) for i in range(n)], dtype=np.int32)
    eng = lib.Engine(spc, n, m, device=0)
    dsig, dlen = eng.dev_alloc(n * m * 4), eng.dev_alloc(n * 4)
    eng.h2d(dlen, lens)
    eng.synth_fill(dsig, dlen, n, seed=2024, first_read=0, decorate=True)
    if a.adc_step > 0:
        sig = np.zeros((n, m), dtype=np.float32)
        eng.d2h(sig, dsig)
        sig = (np.round(sig / np.float32(a.adc_step)) * np.float32(a.adc_step)).astype(np.float32)
        eng.h2d(dsig, sig)
    if a.primary == "cnn":
        from adapted_amd.detect import cnn

        cnn.ensure_weights(eng, None, spc)
        rows, _ = eng.detect_cnn_rows(dsig, dlen, n, 1000, device_ptrs=True)
    else:
        rows, _ = eng.detect_llr_rows(dsig, dlen, n, 1000, with_start_peak=True, device_ptrs=True, tails_nan=True)
    why = eng.debug_fetch(9, n)
    print("%d reads, pass rate %.3f; k_validate_wg:" % (n, rows["success"].mean()), {WHY.get(int(k), k): int(v) for k, v in zip(*np.unique(why, return_counts=True))})
    eng.close()


if __name__ == "__main__":
    main()

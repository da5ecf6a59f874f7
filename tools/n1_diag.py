#!/usr/bin/env python3
"""Which way the single-pass N1 (n1_fused.h) ends on a workload shape: debug tallies of libadapted_hip.so around one call.
    python tools/n1_diag.py [--lens pareto|full] [--reads 16000] [--adc-step 0.0]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lens", default="pareto")
    ap.add_argument("--reads", type=int, default=16000)
    ap.add_argument("--seed", type=int, default=2024)
    a = ap.parse_args()
    import torch
    from adapted_amd import lib, synth

    spc = bench.make_spc(200000)
    m = spc.sig_preload_size
    R, mb = a.reads, 1000
    eng = lib.Engine(spc, R, m, device=0)
    dev = torch.device("cuda", 0)
    sig = torch.empty((R, m), dtype=torch.float32, device=dev)
    lens = np.full(R, m, dtype=np.int32)
    if a.lens == "pareto":
        lens = np.array([synth.pareto_length(a.seed, i) for i in range(R)], dtype=np.int32)
    dl = torch.from_numpy(lens).to(dev)
    torch.cuda.synchronize()
    eng.synth_fill(sig.data_ptr(), dl.data_ptr(), R, seed=a.seed, first_read=0, decorate=True)
    c0 = eng.debug_counters(24).astype(np.int64)
    rows, mbs = eng.detect_llr_rows(sig.data_ptr(), dl.data_ptr(), R, mb, with_start_peak=True, device_ptrs=True, tails_nan=True)
    c1 = eng.debug_counters(24).astype(np.int64)
    d = c1 - c0
    names = {5: "n1_fused attempted", 6: "fallback: median (overflow / capacity / bracket)", 7: "fallback: MAD zone", 17: "MAD band list over capacity",
             18: "median list over capacity", 19: "LDS staging overflow in the pass", 22: "heavy keys", 23: "heavy samples"}
    for k, v in names.items():
        print("%-52s %d" % (v, d[k]))
    print("minibatch status", np.bincount(mbs, minlength=3))
    prm = eng.debug_norm_params(R // mb)
    print("first N1 parameters", prm[:3])
    valid = np.minimum(lens, 200000).reshape(-1, mb).sum(axis=1)
    print("valid samples per minibatch: min %d max %d" % (valid.min(), valid.max()))


if __name__ == "__main__":
    main()

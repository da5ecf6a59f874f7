"""Hostile inputs (development aid, GPU box): infinities, huge and tiny magnitudes, negative signals, constant stretches, NaN
holes inside reads -- through the HIP path and the CPU oracle; prints the number of differing fields per case.
    python tests/abuse_vs_oracle.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from adapted_amd import lib, synth  # noqa: E402
from adapted_amd.config import get_chemistry_specific_config  # noqa: E402
from oracle import oracle  # noqa: E402
from util import row_diffs  # noqa: E402


def main():
    spc = get_chemistry_specific_config("RNA004")
    spc.llr_boundaries.llr_detect, spc.cnn_boundaries.cnn_detect = True, False
    spc.update_primary_method()
    spc.update_sig_preload_size()
    m, n = spc.sig_preload_size, 64
    base, lens = synth.synth_batch(77, 0, n, m, np.full(n, m, dtype=np.int32))
    f = np.float32
    cases = {}
    a = base.copy(); a[3, 5000] = np.inf; a[9, 200] = -np.inf; cases["infinities"] = a
    a = base.copy(); a[5, 7000:7100] = f(3e30); a[6, 100:150] = f(-2e30); cases["huge values"] = a
    a = base.copy() * f(1e-30); cases["tiny magnitudes"] = a
    a = -base.copy(); cases["negative signal"] = a
    a = base.copy(); a[7, 4000:9000] = f(95.0); a[8, :] = f(100.0); cases["constant stretches and a constant read"] = a
    a = base.copy(); a[10, 6000:6100] = np.nan; a[11, 0:50] = np.nan; cases["NaN holes inside reads"] = a
    a = base.copy()
    for i in range(n):  # (bottleneck counts NaN samples out of its moving windows: the MVS check on slices WITH NaNs)
        p = 2400 + (137 * i) % 2600
        a[i, p + 460: p + 462] = np.nan
        if i % 3:
            a[i, p + 60: p + 62] = np.nan
    cases["NaN pairs around the adapter end of every read"] = a
    a = base.copy(); a[:, 1::2] = a[:, 0::2][:, : a[:, 1::2].shape[1]]; cases["every sample twice"] = a
    a = np.round(base.copy()); cases["integers"] = a.astype(np.float32)
    total = 0
    for name, sig in cases.items():
        eng = lib.Engine(spc, n, m, device=0)
        rows, mbs = eng.detect_llr_rows(sig, lens, n, n, with_start_peak=True)
        try:
            want = oracle.detect_llr(sig, lens, spc, with_start_peak=True)
            werr = None
        except ValueError as e:
            want, werr = None, str(e)
        if want is None:
            print("%-40s oracle raises %r; device minibatch status %s" % (name, werr, list(mbs)), flush=True)
            ok = (mbs != 0).all()
            total += 0 if ok else 1
        else:
            got = lib.rows_to_results(rows, "llr")
            bad = 0
            for i, (g, w) in enumerate(zip(got, want)):
                d = row_diffs(g, {k: v for k, v in w.items() if not k.startswith("_")})
                if d:
                    print("   read %d: %s" % (i, d[:8]), flush=True)
                bad += len(d)
            print("%-40s status %s pass %d/%d differing fields: %d" % (name, list(mbs), sum(g.success for g in got), n, bad), flush=True)
            total += bad
        eng.close()
    total += abuse_cnn()
    print("TOTAL differing:", total)
    return 1 if total else 0


def abuse_cnn():
    """the same hostile values through the CNN path at a 60 k window (the wave-per-16-reads series kernel with its
    hand-scheduled variance steps, the shared-sweep candidate statistics): conv stack + predict on the device, the rows
    against the oracle's validation of the same predictions"""
    from adapted_amd.detect import cnn

    spc = get_chemistry_specific_config("RNA004")
    spc.core.max_obs_trace = 60000
    spc.update_primary_method()
    spc.update_sig_preload_size()
    m, n = spc.sig_preload_size, 64
    base, lens = synth.synth_batch(78, 0, n, m, np.full(n, m, dtype=np.int32))
    f = np.float32
    cases = {}
    a = base.copy(); a[3, 15000] = np.inf; a[9, 30000] = -np.inf; a[12, 7000] = np.inf; a[12, 7400] = -np.inf; cases["cnn: infinities"] = a
    a = base.copy(); a[5, 20000:20100] = f(3e30); a[6, 9000:9050] = f(-2e30); cases["cnn: huge values"] = a
    a = base.copy() * f(1e-30); cases["cnn: tiny magnitudes"] = a
    a = -base.copy(); cases["cnn: negative signal"] = a
    a = base.copy(); a[7, 8000:39000] = f(95.0); a[8, :] = f(100.0); cases["cnn: constant stretches and a constant read"] = a
    a = base.copy(); a[10, 16000:16100] = np.nan; a[11, 5000:5050] = np.nan; cases["cnn: NaN holes inside reads"] = a
    a = base.copy()
    for i in range(n):
        p = 2400 + (137 * i) % 2600
        a[i, p + 460: p + 462] = np.nan
        if i % 3:
            a[i, p + 60: p + 62] = np.nan
        if i % 5 == 0:
            a[i, p + 2000: p + 2300] = np.nan  # a hole longer than both windows
    cases["cnn: NaN pairs around the adapter end of every read"] = a
    a = np.round(base.copy()); cases["cnn: integers"] = a.astype(np.float32)
    total = 0
    for name, sig in cases.items():
        eng = lib.Engine(spc, n, m, device=0)
        cnn.ensure_weights(eng, None, spc)
        _, bounds = eng.detect_cnn_rows(sig, lens, n, n)
        got = lib.rows_to_results(cnn.detect_rows(eng, sig, lens, None, spc), "cnn")
        want = oracle.detect_cnn_from_preds(sig, lens, bounds, spc)
        bad = 0
        # the default conv stack (split float16 operands; out-of-range activations repeat the call in float32) against the exact
        # float32 stack: the same predictions on every hostile input
        os.environ["ADP_CNN_CONV"] = "f32"
        try:
            eng32 = lib.Engine(spc, n, m, device=0)
        finally:
            del os.environ["ADP_CNN_CONV"]
        cnn.ensure_weights(eng32, None, spc)
        _, bounds32 = eng32.detect_cnn_rows(sig, lens, n, n)
        eng32.close()
        nd = int((np.asarray(bounds) != np.asarray(bounds32)).any(axis=1).sum())
        if nd:
            print("   %d reads whose predictions differ between the split and the float32 conv stack" % nd, flush=True)
        bad += nd
        for i, (g, w) in enumerate(zip(got, want)):
            d = row_diffs(g, {k: v for k, v in w.items() if not k.startswith("_")})
            if d:
                print("   read %d: %s" % (i, d[:8]), flush=True)
            bad += len(d)
        print("%-46s pass %d/%d all-candidates %d differing fields: %d" % (name, sum(g.success for g in got), n,
              int((bounds[:, 1:] != 0).all(axis=1).sum()), bad), flush=True)
        total += bad
        eng.close()
    return total


if __name__ == "__main__":
    sys.exit(main())

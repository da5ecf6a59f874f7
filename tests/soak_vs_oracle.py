"""Differential soak (development aid, GPU box): random minibatches -- heavy-tailed lengths, values on an ADC grid or not,
several presets and windows -- through the HIP path and the CPU oracle; prints the number of differing fields.
    python tests/soak_vs_oracle.py [n_rounds] [start_peak | big | candidates | cnn | predict | int16]
cnn: the whole CNN path (prepare -> hand-written conv stack -> predict -> candidate validation, incl. the shared-sweep
statistics of cand_stats.h at windows beyond 32 k samples -> short-read fallback), the oracle validating the device's
predictions; predict: cnn_predict on random scores quantised to a coarse grid (ties, plateaus, reads below the mask level)
against the scipy formulation with a stable priority order.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # (the oracle is test infrastructure: this script lives with the tests)

from adapted_amd import lib, synth  # noqa: E402
from adapted_amd.config import get_chemistry_specific_config  # noqa: E402
from oracle import oracle  # noqa: E402
from util import row_diffs  # noqa: E402


def soak_cnn(rounds):
    from adapted_amd.detect import cnn

    rng = np.random.default_rng(77)
    bad_total = 0
    for it in range(rounds):
        spc = get_chemistry_specific_config("RNA004")
        spc.core.max_obs_trace = int(rng.choice([16000, 34000, 60000, 120000, 200000]))
        spc.cnn_boundaries.polya_cand_k = int(rng.choice([1, 3, 10, 15]))
        spc.med_shift.detect_med_shift = bool(it % 2)
        if it % 3 == 2:
            spc.mvs_polya.pA_var_window = int(rng.choice([50, 100, 200, 101]))
            spc.mvs_polya.pA_mean_window = int(rng.choice([10, 20, 60, 23]))
            spc.mvs_polya.median_shift_window = int(rng.choice([500, 1000, 2000]))
        spc.update_primary_method()
        spc.update_sig_preload_size()
        m = spc.sig_preload_size
        n = 96 if m > 100000 else 192
        lens = np.array([m if rng.random() < 0.5 else max(1012, synth.pareto_length(it, i, lo=3000, hi=4 * m)) for i in range(n)], dtype=np.int32)
        sig, lens = synth.synth_batch(300 + it, 0, n, m, lens)
        step = [0.0, 0.18][it % 2]
        if step:
            q = np.float32(step)
            sig = (np.round(sig / q) * q).astype(np.float32)
        eng = lib.Engine(spc, n, m, device=0)
        cnn.ensure_weights(eng, None, spc)
        mb = n // 2
        _, bounds = eng.detect_cnn_rows(sig, lens, n, mb)
        rows = cnn.detect_rows(eng, sig[:mb], lens[:mb], None, spc)
        rows2 = cnn.detect_rows(eng, sig[mb:], lens[mb:], None, spc)
        got = lib.rows_to_results(np.concatenate([rows, rows2]), "cnn")
        want = oracle.detect_cnn_from_preds(sig, lens, bounds, spc)
        bad = 0
        shown = 0
        for i, (g, w) in enumerate(zip(got, want)):
            d = row_diffs(g, {k: v for k, v in w.items() if not k.startswith("_")})
            bad += len(d)
            if d and shown < 3:
                shown += 1
                print("   read %d (len %d): %s" % (i, lens[i], d[:6]), flush=True)
        print("cnn round %d T=%d m=%d k=%d step=%.2f windows=(%d,%d) pass=%d/%d all-candidates=%d differing fields: %d" % (
            it, spc.core.max_obs_trace, m, spc.cnn_boundaries.polya_cand_k, step, spc.mvs_polya.pA_var_window, spc.mvs_polya.pA_mean_window,
            sum(1 for g in got if g.success), n, int((bounds[:, 1:] != 0).all(axis=1).sum()), bad), flush=True)
        bad_total += bad
        eng.close()
    print("TOTAL differing fields:", bad_total)
    return 1 if bad_total else 0


def soak_int16(rounds):
    """adp_detect_llr_i16 (kernels reading raw int16 + per-read calibration) against adp_calibrate_i16 + adp_detect_llr:
    rows must be byte-identical (no oracle involved: the float32 path is pinned elsewhere)"""
    rng = np.random.default_rng(31)
    bad_total = 0
    for it in range(rounds):
        chem = ["RNA004", "RNA002"][it % 2]
        spc = get_chemistry_specific_config(chem)
        spc.llr_boundaries.llr_detect, spc.cnn_boundaries.cnn_detect = True, False
        spc.core.max_obs_trace = int(rng.choice([4000, 16000, 25000, 60000, 200000]))
        spc.mvs_polya.mvs_detect_overwrite = bool(it % 3 == 2)
        spc.med_shift.detect_med_shift = bool(it % 2)
        if it % 4 == 3:
            spc.mvs_polya.pA_var_window = int(rng.choice([50, 101, 400]))   # (400: beyond the LDS series path)
            spc.mvs_polya.pA_mean_window = int(rng.choice([10, 23, 350]))
            spc.mvs_polya.search_window = 900
        spc.update_primary_method()
        spc.update_sig_preload_size()
        m = spc.sig_preload_size
        if m % 4:
            continue
        n = 64 if m > 100000 else 160
        lo = spc.core.min_obs_adapter + 2 * spc.core.downscale_factor + 8
        lens = np.array([m + 7 if rng.random() < 0.4 else max(lo, synth.pareto_length(it, i, lo=3000, hi=4 * m)) for i in range(n)], dtype=np.int32)
        sig, lens = synth.synth_batch(500 + it, 0, n, m, lens)
        scale = rng.uniform(0.1, 0.25, n).astype(np.float32)
        offset = rng.integers(-200, 50, n).astype(np.float32)
        raw = np.clip(np.round(np.nan_to_num(sig) / scale[:, None] - offset[:, None]), -32768, 32767).astype(np.int16)
        eng = lib.Engine(spc, n, m, device=0)
        d_raw, d_len, d_cal, d_f32 = eng.dev_alloc(n * m * 2), eng.dev_alloc(n * 4), eng.dev_alloc(2 * n * 4), eng.dev_alloc(n * m * 4)
        eng.h2d(d_raw, raw); eng.h2d(d_len, lens); eng.h2d(d_cal, np.concatenate([scale, offset]))
        mb = n // 2
        eng.calibrate_i16(d_raw, d_len, d_cal, d_cal + n * 4, n, d_f32)
        a, ma = eng.detect_llr_rows(d_f32, d_len, n, mb, with_start_peak=True, device_ptrs=True, tails_nan=bool(it % 2))
        b, mbs = eng.detect_llr_rows_i16(d_raw, d_len, d_cal, d_cal + n * 4, n, mb, with_start_peak=True)
        a["open_pores_more"] = 0; b["open_pores_more"] = 0
        bad = int((a.view(np.uint8).reshape(n, -1) != b.view(np.uint8).reshape(n, -1)).any(axis=1).sum()) + int((ma != mbs).sum())
        print("int16 round %d %s T=%d m=%d overwrite=%d windows=(%d,%d) pass=%d/%d differing rows: %d" % (
            it, chem, spc.core.max_obs_trace, m, spc.mvs_polya.mvs_detect_overwrite, spc.mvs_polya.pA_var_window, spc.mvs_polya.pA_mean_window,
            int(b["success"].sum()), n, bad), flush=True)
        bad_total += bad
        for p_ in (d_raw, d_len, d_cal, d_f32):
            eng.dev_free(p_)
        eng.close()
    print("TOTAL differing rows:", bad_total)
    return 1 if bad_total else 0


def soak_predict(rounds):
    import torch
    from test_gpu_cnn import _ref_predict

    rng = np.random.default_rng(5)
    bad_total = 0
    spc = get_chemistry_specific_config("RNA004")
    eng = lib.Engine(spc, 512, spc.sig_preload_size, device=0)
    na = (spc.core.max_obs_adapter - spc.core.min_obs_adapter) // spc.core.downscale_factor
    k = int(spc.cnn_boundaries.polya_cand_k)
    for it in range(rounds):
        n, Lo = int(rng.integers(2, 200)), int(rng.choice([40, 257, 1650, 6001]))
        grid = float(rng.choice([0.0, 0.5, 1.0, 2.0]))  # a coarse grid makes ties, plateaus and samples equal to the mask level
        scores = rng.normal(-3.0 if it % 2 else 0.0, 3.0, (n, 2, Lo)).astype(np.float32)
        if grid:
            scores = (np.round(scores / grid) * grid).astype(np.float32)
        scores[rng.random(n) < 0.2, 1, :] -= 12.0  # reads at or below the mask level
        scores[rng.random(n) < 0.2, 0, 0] = 90.0   # adapter position 0
        want = _ref_predict(scores, min(na, Lo), k, stable=True)
        d = torch.from_numpy(scores).cuda()
        torch.cuda.synchronize()
        b = eng.cnn_predict(d.data_ptr(), n, n, Lo)
        got = np.where(b == 0, 0, (b - spc.core.min_obs_adapter) // spc.core.downscale_factor)
        bad = int((got != want).sum())
        print("predict round %d n=%d Lo=%d grid=%.1f differing entries: %d" % (it, n, Lo, grid, bad), flush=True)
        bad_total += bad
    eng.close()
    print("TOTAL differing entries:", bad_total)
    return 1 if bad_total else 0


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    if len(sys.argv) > 2 and sys.argv[2] == "cnn":
        return soak_cnn(rounds)
    if len(sys.argv) > 2 and sys.argv[2] == "int16":
        return soak_int16(rounds)
    if len(sys.argv) > 2 and sys.argv[2] == "predict":
        import torch

        torch.cuda.init()
        return soak_predict(rounds)
    rng = np.random.default_rng(2024)
    bad_total = 0
    for it in range(rounds):
        chem = ["RNA004", "RNA002"][it % 2]
        spc = get_chemistry_specific_config(chem)
        spc.llr_boundaries.llr_detect, spc.cnn_boundaries.cnn_detect = True, False
        spc.core.max_obs_trace = int(rng.choice([4000, 16000, 25000, 60000]))
        big = len(sys.argv) > 2 and sys.argv[2] == "big"  # the headline window; minibatches of 100 reads reach the fused N1 pass
        if big:
            spc.core.max_obs_trace = 200000
        spc.mvs_polya.mvs_detect_overwrite = bool(it % 3 == 2)
        spc.med_shift.detect_med_shift = bool(it % 2)
        if it >= 12:  # perturbed thresholds: decisions land near their limits more often
            spc.llr_boundaries.adapter_peak_width = int(rng.integers(200, 2500))
            spc.llr_boundaries.adapter_peak_prominence = float(rng.choice([0.5, 1.0, 2.0]))
            spc.mvs_polya.pA_var_window = int(rng.choice([50, 100, 200]))
            spc.mvs_polya.pA_mean_window = int(rng.choice([10, 20, 60]))
            spc.mvs_polya.median_shift_window = int(rng.choice([500, 1000, 2000]))
            spc.mvs_polya.search_window = int(rng.choice([300, 500, 900]))
            spc.real_range.mean_window = int(rng.choice([100, 300]))
            spc.real_range.max_obs_local_range = int(rng.choice([1000, 5000]))
            spc.med_shift.med_shift_window = int(rng.choice([500, 2000]))
        spc.update_primary_method()
        spc.update_sig_preload_size()
        m = spc.sig_preload_size
        n = 200 if big else 160
        mbn = 100 if big else 80
        lo = spc.core.min_obs_adapter + 2 * spc.core.downscale_factor + 8
        lens = np.array([max(lo, synth.pareto_length(it, i, lo=3000, hi=4 * m)) for i in range(n)], dtype=np.int32)
        if big and it % 2 == 0:
            lens[:] = np.where(rng.random(n) < 0.6, m, lens)  # mostly full-length reads
        sig, lens = synth.synth_batch(100 + it, 0, n, m, lens)
        step = [0.0, 0.18, 0.05][it % 3]
        if step:
            q = np.float32(step)
            sig = (np.round(sig / q) * q).astype(np.float32)
        sp_primary = len(sys.argv) > 2 and sys.argv[2] == "start_peak"
        if sp_primary:  # the start-peak primary (API only in the reference): validation without the MVS candidates
            spc.llr_boundaries.llr_detect, spc.rna_start_peak.detect_rna_start_peak = False, True
            spc.mvs_polya.mvs_detect_check = bool(it % 4 == 0)  # (on: the reference raises for every read, topk is None)
            spc.update_primary_method()
        cand_mode = len(sys.argv) > 2 and sys.argv[2] == "candidates"
        eng = lib.Engine(spc, n, m, device=0)
        if cand_mode:  # the CNN path's validator: adapter end + k candidate poly(A) ends per read (0 ends the list)
            r0, _ = eng.detect_llr_rows(sig, lens, n, mbn)
            g0 = lib.rows_to_results(r0, "llr")
            ae = np.array([g.llr_adapter_end or 0 for g in g0], dtype=np.int64)
            pe = np.array([g.llr_polya_end or 0 for g in g0], dtype=np.int64)
            k = int(rng.integers(1, 6))
            bounds = np.zeros((n, 1 + k), dtype=np.int64)
            bounds[:, 0] = np.where(rng.random(n) < 0.9, ae, 0)
            for c in range(k):
                jit = rng.integers(-400, 1500, n)
                col = np.where(pe > 0, np.maximum(pe + jit * (c > 0), 0), 0)
                bounds[:, 1 + c] = np.where(rng.random(n) < (0.9 if c == 0 else 0.6), col, 0)
            if it % 4 == 3:  # NaN samples inside the validated slices (bottleneck counts them out of its windows)
                sig = sig.copy()
                for i in range(0, n, 2):
                    for _ in range(int(rng.integers(1, 4)) if ae[i] > 0 else 0):
                        p, w = int(ae[i] + rng.integers(-300, 3000)), int(rng.integers(1, 150))
                        if p >= 0 and p + w < min(int(lens[i]), m):
                            sig[i, p:p + w] = np.nan
            spc.cnn_boundaries.cnn_detect, spc.llr_boundaries.llr_detect = True, False
            spc.cnn_boundaries.fallback_to_llr_short_reads = False
            spc.update_primary_method()
            eng.close()
            eng = lib.Engine(spc, n, m, device=0)
            got = lib.rows_to_results(eng.validate_rows(sig, lens, n, bounds), "cnn")
            want = oracle.detect_cnn_from_preds(sig, lens, bounds, spc)
            mbs = np.zeros(2, dtype=np.int32)
        elif sp_primary:
            rows = eng.detect_start_peak_rows(sig, lens, n, mbn)
            mbs = np.zeros(2, dtype=np.int32)
            got = lib.open_pore_float_column(lib.rows_to_results(rows, "start_peak"))  # (one operator call = one DataFrame)
            want = oracle.detect_start_peak(sig, lens, spc)
        else:
            rows, mbs = eng.detect_llr_rows(sig, lens, n, mbn, with_start_peak=True, tails_nan=bool(it % 2))
            got = lib.rows_to_results(rows, "llr")
            want = []
            for s0 in range(0, n, mbn):
                want += oracle.detect_llr(sig[s0:s0 + mbn], lens[s0:s0 + mbn], spc, with_start_peak=True)
        bad = 0
        shown = 0
        for i, (g, w) in enumerate(zip(got, want)):
            d = row_diffs(g, {k: v for k, v in w.items() if not k.startswith("_")})
            bad += len(d)
            if d and shown < 3:
                shown += 1
                print("   read %d (len %d): %s" % (i, lens[i], d[:6]), flush=True)
        ok = sum(1 for g in got if g.success)
        cnt = eng.debug_counters(8)
        print("round %d %s T=%d m=%d step=%.2f overwrite=%d mbs=%s pass=%d/%d differing fields: %d  (N1 single pass tried/median missed/MAD missed: %s)" % (
            it, chem, spc.core.max_obs_trace, m, step, spc.mvs_polya.mvs_detect_overwrite, list(mbs), ok, n, bad, list(cnt[5:8])), flush=True)
        bad_total += bad
        eng.close()
    print("TOTAL differing fields:", bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())

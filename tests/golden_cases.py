"""Golden-vector case definitions shared by ``oracle/gen_golden.py`` (runs in the build
container against the real reference) and the parity tests (run anywhere, against the
committed outputs in tests/golden/).  Python >= 3.8 syntax only (the generator runs under
the conda python3.9 that has the reference's real third-party stack)."""

# lengths cycled over the reads of a "mixed" case; "m" = preload size, ints are literal.
# 1012 keeps exactly one pooled block valid (reads with NO valid pooled block crash the
# reference's minibatch: reference adapted/detect/_c_llr.pyx:202-236 with an empty array).
MIXED_LENS = ["m", "m+5000", 30000, "m", 8000, "m", 3000, 1500, "m", 1012, 600000, 5400, "m", 12345]

CASES = {
    # RNA004 preset, LLR primary, default window (m = 17500)
    "rna004_llr_default": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=11,
                               first=0, n=96, lens="mixed", minibatch=96, dump=[0, 1, 2, 6, 9]),
    # the north-star shape: T = 200000, m = 201500
    "rna004_llr_200k": dict(chem="RNA004", primary="llr", max_obs_trace=200000, seed=12,
                            first=1000, n=10, lens="mixed200", minibatch=10, dump=[0]),
    # config 1: RNA002 preset (LLR primary), short preload (T=4000 -> m=5500)
    "rna002_llr_4k": dict(chem="RNA002", primary="llr", max_obs_trace=4000, seed=13,
                          first=0, n=48, lens="full", minibatch=48, dump=[0, 1]),
    # RNA002 preset as shipped (T=25000, m=26500, ds=20)
    "rna002_llr_default": dict(chem="RNA002", primary="llr", max_obs_trace=None, seed=14,
                               first=0, n=40, lens="mixed002", minibatch=40, dump=[0]),
    # two minibatches in one call: normalisation is per minibatch (reference
    # adapted/detect/normalize.py:15-22 called with the 2-D minibatch)
    "rna004_llr_two_minibatches": dict(chem="RNA004", primary="llr", max_obs_trace=None,
                                       seed=15, first=500, n=64, lens="mixed", minibatch=32,
                                       dump=[]),
    # start-peak primary (API only in the reference), mvs check off, med-shift on
    "rna004_start_peak": dict(chem="RNA004", primary="start_peak", max_obs_trace=None, seed=16,
                              first=0, n=64, lens="mixed", minibatch=64, dump=[],
                              mvs_detect_check=False, detect_med_shift=True),
    # same, all reads full length: no None row, so the pandas float-column quirk stays dormant
    "rna004_start_peak_full": dict(chem="RNA004", primary="start_peak", max_obs_trace=None, seed=18,
                                   first=0, n=64, lens="full", minibatch=64, dump=[],
                                   mvs_detect_check=False, detect_med_shift=True),
    # the start-peak primary at the north-star window (m = 201500): K1's scan behind max_obs_trace, end_idx = min(len, m) // 10
    "rna004_start_peak_200k": dict(chem="RNA004", primary="start_peak", max_obs_trace=200000, seed=29,
                                   first=3000, n=20, lens="long200", minibatch=20, dump=[],
                                   mvs_detect_check=False, detect_med_shift=True),
    # start-peak primary with the preset's mvs check left on: polya_end_topk is None -> TypeError
    "rna004_start_peak_mvs": dict(chem="RNA004", primary="start_peak", max_obs_trace=None, seed=19,
                                  first=0, n=16, lens="full", minibatch=16, dump=[]),
    # mvs_detect_overwrite = true (custom TOML only): the adapter end is moved to the MVS-detected position
    # (reference adapted/detect/mvs.py:181-338, adapted/detect/combined.py:517-562)
    "rna004_llr_mvs_overwrite": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=21, first=0, n=96,
                                     lens="mixed", minibatch=96, dump=[],
                                     override={"mvs_polya.mvs_detect_overwrite": True}),
    # wide variance window: the new adapter end passes the poly(A) end of short tails -> polya_end becomes None
    "rna004_llr_mvs_overwrite_wide": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=11, first=0, n=96,
                                          lens="mixed", minibatch=96, dump=[],
                                          override={"mvs_polya.mvs_detect_overwrite": True, "mvs_polya.pA_var_window": 600,
                                                    "mvs_polya.search_window": 1200, "med_shift.detect_med_shift": True}),
    # combined_detect_llr (adapted/detect/combined.py:39-119), the single-read API: per-read normalisation, pooled from
    # sample 0; every read on its own, handed over without padding
    "rna004_llr_single": dict(chem="RNA004", primary="llr_single", max_obs_trace=None, seed=11, first=0, n=96,
                              lens="mixed", minibatch=1, dump=[]),
    "rna002_llr_single_4k": dict(chem="RNA002", primary="llr_single", max_obs_trace=4000, seed=13, first=0, n=48,
                                 lens="mixed002", minibatch=1, dump=[]),
    # reads with MANY open-pore events inside the adapter (>= 200 pA blips, anomalies.py:15-35): the open_pores list the CSV
    # prints has no length limit (23 and 39 entries here, beside reads with 15, 16, 1, 0)
    "rna004_llr_open_pores": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=23, first=0, n=24, lens="full",
                                  minibatch=24, dump=[], blips=[24, 17, 16, 40, 2, 1]),
    # ---- round 3: more of the reference's behaviour pinned (each case run through the real reference) ----
    # samples on an ADC-like grid (0.18 pA): thousands of ties in every median / percentile / MAD (numpy's tie rules), heavy keys in N1
    "rna004_llr_quantised": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=31, first=0, n=64, lens="mixed",
                                 minibatch=64, dump=[0], quantise=0.18),
    "rna004_llr_200k_quantised": dict(chem="RNA004", primary="llr", max_obs_trace=200000, seed=32, first=500, n=8, lens="mixed200",
                                      minibatch=8, dump=[], quantise=0.18),
    # RNA002 (ds = 20, adapter_peak_width 1500) at the 200 k window
    "rna002_llr_200k": dict(chem="RNA002", primary="llr", max_obs_trace=200000, seed=33, first=0, n=8, lens="mixed200b",
                            minibatch=8, dump=[0]),
    # a pooling factor that divides neither the window nor min_obs_adapter (ragged last block, np.pad zeros), lower peak thresholds,
    # the median-shift check on, open-pore detection off
    "rna004_llr_ds7": dict(chem="RNA004", primary="llr", max_obs_trace=12345, seed=34, first=0, n=48, lens="mixed", minibatch=48,
                           dump=[0, 2],
                           override={"core.downscale_factor": 7, "llr_boundaries.adapter_peak_prominence": 0.7,
                                     "llr_boundaries.adapter_peak_rel_height": 0.8, "med_shift.detect_med_shift": True,
                                     "real_range.detect_open_pores": False}),
    # open pores close to the adapter end and inside the poly(A) tail, the real-range check off
    "rna004_llr_open_pores_near_end": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=35, first=0, n=32, lens="full",
                                           minibatch=32, dump=[], blips_at="adapter_end",
                                           override={"real_range.real_signal_check": False}),
    # start-peak primary with flagged open pores at the default window (the float-column quirk again, several flagged reads)
    "rna004_start_peak_blips": dict(chem="RNA004", primary="start_peak", max_obs_trace=None, seed=36, first=0, n=32, lens="full",
                                    minibatch=32, dump=[], mvs_detect_check=False, detect_med_shift=True, start_blips=True),
    # CNN primary with fewer candidates and without the short-read fallback
    "rna004_cnn_k3": dict(chem="RNA004", primary="cnn", max_obs_trace=None, seed=37, first=100, n=32, lens="mixed", minibatch=32,
                          dump=[0], override={"cnn_boundaries.polya_cand_k": 3, "cnn_boundaries.fallback_to_llr_short_reads": False}),
    "rna004_cnn_k1": dict(chem="RNA004", primary="cnn", max_obs_trace=None, seed=38, first=200, n=24, lens="mixed", minibatch=24,
                          dump=[0], override={"cnn_boundaries.polya_cand_k": 1}),
    # odd moving windows, other window lengths everywhere, a tighter outlier clip
    "rna004_llr_windows": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=41, first=0, n=48, lens="mixed", minibatch=48, dump=[],
                               override={"mvs_polya.pA_var_window": 101, "mvs_polya.pA_mean_window": 21, "mvs_polya.polyA_window": 250,
                                         "mvs_polya.median_shift_window": 1500, "mvs_polya.search_window": 700,
                                         "real_range.mean_window": 250, "real_range.max_obs_local_range": 3000,
                                         "core.sig_norm_outlier_thresh": 3.0}),
    # the whole signal scaled and shifted (x * 0.9 - 30): other medians, range gates that fail, values near zero
    "rna004_llr_affine": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=42, first=0, n=32, lens="mixed", minibatch=32, dump=[0],
                              affine=(0.9, -30.0)),
    # constant stretches: zero variances inside the LLR (log 0 = -inf, inf - inf = NaN in the traces), ties in every statistic
    "rna004_llr_flat": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=43, first=0, n=32, lens="full", minibatch=32, dump=[0, 1],
                            flat=True),
    # minibatches of one and two reads
    "rna004_llr_tiny": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=44, first=0, n=3, lens="full", minibatch=2, dump=[]),
    # tighter real-range gates: other failure reasons
    "rna004_llr_gates": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=45, first=0, n=64, lens="mixed", minibatch=64, dump=[],
                             override={"real_range.local_range": (12.0, 25.0), "real_range.adapter_mad_range": (4.0, 9.0),
                                       "real_range.mean_start_range": (60.0, 95.0), "med_shift.detect_med_shift": True,
                                       "med_shift.med_shift_range": (12.0, None)}),
    # start-peak primary with other offsets and a low open-pore level (many flagged reads); and on the RNA002 preset
    "rna004_start_peak_params": dict(chem="RNA004", primary="start_peak", max_obs_trace=None, seed=46, first=0, n=32, lens="full",
                                     minibatch=32, dump=[], mvs_detect_check=False, detect_med_shift=True,
                                     override={"rna_start_peak.offset1": 5, "rna_start_peak.offset2": 50,
                                               "rna_start_peak.start_peak_max_idx": 120, "rna_start_peak.open_pore_pa": 118.0}),
    "rna002_start_peak": dict(chem="RNA002", primary="start_peak", max_obs_trace=None, seed=47, first=0, n=24, lens="full",
                              minibatch=24, dump=[], mvs_detect_check=False, detect_med_shift=True),
    # CNN primary with a shorter adapter range (the arg-max window, the fallback's length threshold) and a longer min_obs_polya
    "rna004_cnn_adapter_range": dict(chem="RNA004", primary="cnn", max_obs_trace=None, seed=48, first=300, n=32, lens="mixed", minibatch=32,
                                     dump=[0], override={"core.max_obs_adapter": 4000, "core.min_obs_polya": 300}),
    # mvs_polya.pA_mean_range is derived PER READ from that read's adapter median: validate_boundaries works on a deep copy of the
    # config (combined.py:359) -- read 0 scaled by 1.10 so that a range kept from it (1.3 x 88 pA) would fail the other reads' poly(A)
    # means (108 pA): it does not
    "rna004_llr_first_read_range": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=51, first=0, n=64, lens="full", minibatch=32,
                                        dump=[], first_read_scale=1.10),
    # min_obs_adapter and max_obs_trace that are no multiples of the pooling factor; extreme pooling factors
    "rna004_llr_minobs1005": dict(chem="RNA004", primary="llr", max_obs_trace=15555, seed=52, first=0, n=32, lens="mixed_b", minibatch=32,
                                  dump=[0], override={"core.min_obs_adapter": 1005}),
    "rna004_llr_ds3": dict(chem="RNA004", primary="llr", max_obs_trace=7000, seed=53, first=0, n=24, lens="mixed", minibatch=24, dump=[0],
                           override={"core.downscale_factor": 3}),
    "rna004_llr_ds32": dict(chem="RNA004", primary="llr", max_obs_trace=40000, seed=54, first=0, n=16, lens="mixed_b", minibatch=16, dump=[0],
                            override={"core.downscale_factor": 32}),
    # every MVS range given explicitly, with lower AND upper bounds
    "rna004_llr_ranges": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=55, first=0, n=64, lens="mixed", minibatch=64, dump=[],
                              override={"mvs_polya.pA_mean_range": (100.0, 116.0), "mvs_polya.pA_var_range": (2.0, 15.0),
                                        "mvs_polya.median_shift_range": (15.0, 40.0), "mvs_polya.polyA_local_range": (1.0, 12.0),
                                        "mvs_polya.polyA_med_range": (100.0, 115.0)}),
    # the MVS check off, the median-shift check on
    "rna004_llr_no_mvs": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=56, first=0, n=48, lens="mixed", minibatch=48, dump=[],
                              mvs_detect_check=False, detect_med_shift=True),
    # CNN primary with neither pA_mean_range nor its adapter-median scale given: "pA_mean_range is not specified" per read
    "rna004_cnn_no_mean_range": dict(chem="RNA004", primary="cnn", max_obs_trace=None, seed=57, first=0, n=16, lens="mixed", minibatch=16,
                                     dump=[0], override={"mvs_polya.pA_mean_adapter_med_scale_range": (None, None)}),
    # a narrower adapter peak (width 400 samples)
    "rna004_llr_peak_width": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=58, first=0, n=48, lens="mixed", minibatch=48, dump=[0],
                                  override={"llr_boundaries.adapter_peak_width": 400}),
    # NaN holes INSIDE reads (the loader never makes them; numpy's NaN rules everywhere)
    "rna004_llr_nan_holes": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=59, first=0, n=24, lens="full", minibatch=24, dump=[],
                                 nan_holes=True),
    # the single-read operator, the mvs_detect_overwrite branch and the start-peak primary on quantised samples / NaN holes / constant stretches
    "rna004_llr_single_quantised": dict(chem="RNA004", primary="llr_single", max_obs_trace=None, seed=61, first=0, n=32, lens="mixed",
                                        minibatch=1, dump=[], quantise=0.18),
    "rna004_llr_single_nan": dict(chem="RNA004", primary="llr_single", max_obs_trace=None, seed=62, first=0, n=18, lens="full",
                                  minibatch=1, dump=[], nan_holes=True),
    "rna004_llr_mvs_overwrite_quantised": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=63, first=0, n=48, lens="mixed",
                                               minibatch=48, dump=[], quantise=0.18, override={"mvs_polya.mvs_detect_overwrite": True}),
    "rna004_llr_mvs_overwrite_flat": dict(chem="RNA004", primary="llr", max_obs_trace=None, seed=64, first=0, n=32, lens="full",
                                          minibatch=32, dump=[], flat=True, override={"mvs_polya.mvs_detect_overwrite": True}),
    "rna002_llr_mvs_overwrite": dict(chem="RNA002", primary="llr", max_obs_trace=None, seed=65, first=0, n=40, lens="mixed002",
                                     minibatch=40, dump=[], override={"mvs_polya.mvs_detect_overwrite": True}),
    "rna004_start_peak_nan": dict(chem="RNA004", primary="start_peak", max_obs_trace=None, seed=66, first=0, n=24, lens="full",
                                  minibatch=24, dump=[], mvs_detect_check=False, detect_med_shift=True, nan_holes=True),
    "rna004_llr_200k_windows": dict(chem="RNA004", primary="llr", max_obs_trace=200000, seed=67, first=0, n=8, lens="mixed200", minibatch=8,
                                    dump=[], override={"mvs_polya.pA_var_window": 101, "mvs_polya.pA_mean_window": 21,
                                                       "mvs_polya.median_shift_window": 1500, "real_range.max_obs_local_range": 3000}),
    "rna004_cnn_200k_k3": dict(chem="RNA004", primary="cnn", max_obs_trace=200000, seed=68, first=4000, n=6, lens="mixed200cnn", minibatch=6,
                               dump=[0], override={"cnn_boundaries.polya_cand_k": 3}),
    # the CNN primary on quantised samples, constant stretches and NaN holes (C1's medians with ties, scores of odd inputs, argmax ties)
    "rna004_cnn_quantised": dict(chem="RNA004", primary="cnn", max_obs_trace=None, seed=71, first=0, n=32, lens="mixed", minibatch=32,
                                 dump=[0, 3], quantise=0.18),
    "rna004_cnn_flat": dict(chem="RNA004", primary="cnn", max_obs_trace=None, seed=72, first=0, n=24, lens="full", minibatch=24,
                            dump=[0, 1, 2, 3, 7, 11, 15, 19, 23], flat=True),
    "rna004_cnn_nan_holes": dict(chem="RNA004", primary="cnn", max_obs_trace=None, seed=73, first=0, n=24, lens="full", minibatch=24,
                                 dump=[1, 2], nan_holes=True),
    # NaN pairs around the adapter end and inside the poly(A) tail of every read: the moving mean / variance of the MVS check
    # and of the mvs_detect_overwrite scan over slices WITH NaN samples (bottleneck counts them out of the window)
    "rna004_cnn_nan_polya": dict(chem="RNA004", primary="cnn", max_obs_trace=None, seed=74, first=0, n=32, lens="mixed", minibatch=32,
                                 dump=[1], nan_holes="polya"),
    "rna004_cnn_nan_polya_overwrite": dict(chem="RNA004", primary="cnn", max_obs_trace=None, seed=75, first=0, n=32, lens="mixed",
                                           minibatch=32, dump=[1], nan_holes="polya", override={"mvs_polya.mvs_detect_overwrite": True}),
    # CNN primary with the shipped weights (default window)
    "rna004_cnn_default": dict(chem="RNA004", primary="cnn", max_obs_trace=None, seed=17,
                               first=0, n=48, lens="mixed", minibatch=48, dump=[0, 1]),
    # configs[2] at the north-star window (T = 200000, m = 201500, Lc = 20050): the CNN head pinned where the bench runs it;
    # lengths include reads short enough (< 2 * max_obs_adapter) to enter the LLR fallback (combined.py:251-301)
    "rna004_cnn_200k": dict(chem="RNA004", primary="cnn", max_obs_trace=200000, seed=27,
                            first=2000, n=12, lens="mixed200cnn", minibatch=12, dump=[0, 2]),
}


def apply_quantise(sig, case):
    """case["quantise"]: every sample rounded to a multiple of this many pA (float32 arithmetic, NaN padding kept)"""
    q = case.get("quantise")
    if not q:
        return sig
    import numpy as np

    q = np.float32(q)
    np.divide(sig, q, out=sig)
    np.round(sig, out=sig)
    np.multiply(sig, q, out=sig)
    return sig


def apply_extra(sig, lens, case):
    """case["blips_at"] = "adapter_end": 3-sample 260 pA spikes around where the synthetic adapter ends and into the poly(A) tail
    (reads differ: read i gets them at 2400 + 137 i mod 2600 and 40 / 400 samples further);
    case["start_blips"]: a 12-sample 230 pA block in the first 2000 samples of every third read (flagged open pores of K1)"""
    if case.get("blips_at") == "adapter_end":
        for i in range(sig.shape[0]):
            p = 2400 + (137 * i) % 2600
            for d in (0, 40, 400):
                if p + d + 3 < sig.shape[1]:
                    sig[i, p + d: p + d + 3] = 260.0
    if case.get("nan_holes") == "polya":
        import numpy as np

        for i in range(sig.shape[0]):
            p = 2400 + (137 * i) % 2600
            for d in (60, 460) if i % 3 else (460,):
                if p + d + 2 < min(int(lens[i]), sig.shape[1]):
                    sig[i, p + d: p + d + 2] = np.nan
    elif case.get("nan_holes"):
        import numpy as np

        for i in range(sig.shape[0]):
            k = i % 6
            if k == 1:
                sig[i, 5003:5011] = np.nan            # inside the RNA part, inside the LLR window
            elif k == 2:
                sig[i, 1500:1504] = np.nan            # inside the adapter
            elif k == 3:
                sig[i, 17000:17003] = np.nan          # behind max_obs_trace
            elif k == 4:
                sig[i, 400:402] = np.nan              # in front of min_obs_adapter
    if case.get("first_read_scale"):
        import numpy as np

        np.multiply(sig[0], np.float32(case["first_read_scale"]), out=sig[0])
    if case.get("affine"):
        import numpy as np

        a, b = case["affine"]
        np.multiply(sig, np.float32(a), out=sig)
        np.add(sig, np.float32(b), out=sig)
    if case.get("flat"):
        for i in range(sig.shape[0]):
            k = i % 4
            if k == 0:
                sig[i, 1000:1600] = sig[i, 1000]           # constant right behind min_obs_adapter: zero head variances
            elif k == 1:
                sig[i, 2000:2000 + 40 * (i + 1)] = 80.0    # a plateau inside the adapter
            elif k == 2:
                sig[i, -3000:] = sig[i, -3000]             # constant tail: zero tail variances, RNA statistics full of ties
            else:
                sig[i, 4000:9000] = 108.0                  # a constant poly(A)-like stretch
    if case.get("start_blips"):
        for i in range(0, sig.shape[0], 3):
            p = 10 * (30 + (7 * i) % 150)
            sig[i, p: p + 12] = 230.0
    return sig



# ---- round 4: index-flip census of the conv stacks against the reference's CPU scores (oneDNN) at scale ----
# `oracle/gen_golden.py preds` runs cnn_detect (reference adapted/detect/cnn.py:165-182: prepare_data, the torch CPU net, cnn_predict;
# no validation) over whole minibatches and stores, per read, the predicted sample positions [adapter_end, k poly(A) candidates] and
# the reference's float32 scores AT those positions (tests/golden/<name>.preds.npz).  Reads are the synthetic generator's own
# (seed, first read); 7 of 8 fill the window, every 8th is shorter (NaN tail).
PREDS_CASES = {
    "rna004_cnn_preds_default": dict(chem="RNA004", primary="cnn", max_obs_trace=None, seed=81, first=0, n=16000, minibatch=1000,
                                     lens="preds"),
    "rna004_cnn_preds_200k": dict(chem="RNA004", primary="cnn", max_obs_trace=200000, seed=82, first=0, n=1600, minibatch=160,
                                  lens="preds"),
}
PREDS_SHORT = [0.31, 0.9, 0.12, 0.55, 0.07, 0.74]  # length of every 8th read as a share of the window


def preds_lens(n, m):
    out = []
    for k in range(n):
        out.append(m if k % 8 != 7 else max(2000, int(m * PREDS_SHORT[(k // 8) % len(PREDS_SHORT)])))
    return out


def apply_blips(sig, case):
    """case["blips"]: read i gets blips[i % len] open-pore events -- 3 samples at 260 pA every 40 samples from sample 120 on
    (inside the adapter, in front of min_obs_adapter) -- written over the synthetic signal."""
    spec = case.get("blips")
    if not spec:
        return sig
    for i in range(sig.shape[0]):
        for j in range(spec[i % len(spec)]):
            sig[i, 120 + 40 * j: 123 + 40 * j] = 260.0
    return sig


def apply_overrides(spc, case):
    """case["override"]: {"section.field": value} set on the config tree (both the reference's and the mirror)."""
    for key, val in (case.get("override") or {}).items():
        obj = spc
        parts = key.split(".")
        for q in parts[:-1]:
            obj = getattr(obj, q)
        assert hasattr(obj, parts[-1]), key
        setattr(obj, parts[-1], val)


def resolve_lens(spec, n, m):
    out = []
    if spec == "full":
        return [m] * n
    if spec == "mixed200":
        pat = ["m", "m+5000", 150000, "m", 60000, "m", 9000, "m", 1012, "m"]
    elif spec == "mixed_b":  # like "mixed" without the reads that leave no pooled block behind a larger min_obs_adapter / pooling factor
        pat = ["m", "m+5000", 30000, "m", 8000, "m", 3000, 1500, "m", 1100, 600000, 5400, "m", 12345]
    elif spec == "mixed200b":  # RNA002: min_obs_adapter = 2000, ds = 20
        pat = ["m", "m+5000", 150000, "m", 60000, 9000, "m", 2025]
    elif spec == "long200":  # every read long enough for a start-peak row (a None row turns the whole minibatch into TypeErrors)
        pat = ["m", "m+5000", 150000, "m", 60000, "m", 30000, "m", 21000, "m"]
    elif spec == "mixed200cnn":
        pat = ["m", "m+5000", 150000, 9000, 60000, "m", 12000, "m", 1012, 11000, "m", 7500]
    elif spec == "mixed002":  # RNA002: min_obs_adapter=2000, ds=20 -> need >= 2020 samples
        pat = ["m", "m+5000", 30000, "m", 8000, 2025, 3000, "m", 2400, 12345]
    else:
        pat = MIXED_LENS
    for k in range(n):
        v = pat[k % len(pat)]
        if v == "m":
            out.append(m)
        elif v == "m+5000":
            out.append(m + 5000)
        else:
            out.append(int(v))
    return out

"""Margin analysis of the threshold decisions that sit on float64 values the device does not compute in the reference's
exact operation order:

  * `log` of the gains: the batch path's table-driven logarithm is accurate to < 1 ulp (log_1ulp_fast, 20 float64
    operations; glibc's, which the reference uses, is specified to < 1 ulp as well): a trace point is a difference of terms
    `len * log(var)` of the order of 1e5, so it moves by up to ~2e-11 ABSOLUTE -- 1e-16 of the terms, up to 1e-13 of a
    typical gain (test bar on the trace: 1e-9 relative);
  * `np.nanstd` of the clipped trace (threshold of P1, llr.py:204-224): one-pass float64 moments on the device instead of
    numpy's two passes -> the threshold `prominence * nanstd` differs by ~1e-15 relative;
  * the regression sums of P4 (llr.py:467-479): sequential float64 sums instead of a BLAS dot product.

Every one of them only feeds a comparison: `g > 0` (T1, llr.py:135-142), `prominence >= p * nanstd` and `width >= W` (P1),
`prominence >= 1`, `width >= 10` (P3, P4), `h1 > h0`, `h1 < h0 / 2`, `r^2 >= 0.99` (P4).  A different outcome needs the two
sides to agree to ~12 significant digits.  This test measures, on the reference-exact traces of the CPU oracle (which equals
the reference bit for bit on the golden stage vectors), how close the golden reads and 400 more synthetic reads -- both
presets, two window lengths -- actually come: the smallest relative margin of every comparison is asserted to be above 1e-7,
five orders of magnitude more than the perturbation.  Perturbed traces are also pushed through the same decisions and must leave
every index unchanged: one ulp up / down on every value (np.nextafter), and independent random ABSOLUTE errors of 1e-15 of
the trace's largest magnitude on every point (ten times what two < 1 ulp logarithms of 20 000-sample segments can add)."""
import numpy as np
import pytest
from scipy.signal import find_peaks, peak_prominences, peak_widths
from scipy.stats import linregress

from adapted_amd import synth
from golden_cases import CASES
from util import make_spc

BAR = 1e-7


def _t1(g):
    pos = ~(g <= 0)
    if not pos.any():
        return 0, g.size - 1
    idx = np.flatnonzero(pos)
    return int(idx[0]), int(idx[-1])


def _rel(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-300)


def _p1_margins(g, prominence, W, rel_height):
    """-> (smallest relative margin of the P1 comparisons, threshold, first accepted peak or -1)"""
    s, e = _t1(g)
    clip = g[s:e]
    if clip.size < 3:
        return np.inf, None, -1
    with np.errstate(all="ignore"):
        thr = prominence * np.nanstd(clip)
    pk, _ = find_peaks(clip)
    if pk.size == 0:
        return np.inf, thr, -1
    with np.errstate(all="ignore"):
        prom = peak_prominences(clip, pk)[0]
        wid = peak_widths(clip, pk, rel_height=rel_height)[0]
    m = min(min(_rel(p, thr) for p in prom), min(_rel(w, float(W)) for w in wid))
    ok = np.flatnonzero((prom >= thr) & (wid >= W))
    return m, thr, (int(pk[ok[0]]) + s if ok.size else -1)


def _p4_margins(g2):
    x = np.nan_to_num(g2, nan=0.0)
    pk, props = find_peaks(x, distance=10, prominence=1.0, width=10, rel_height=0.5)
    allpk, _ = find_peaks(x)
    m = np.inf
    if allpk.size:
        with np.errstate(all="ignore"):
            prom = peak_prominences(x, allpk)[0]
            wid = peak_widths(x, allpk, rel_height=0.5)[0]
        big = prom > 1e-3  # (the noise maxima far below the threshold say nothing; those near it do)
        if big.any():
            m = min(min(_rel(p, 1.0) for p in prom[big]), min(_rel(w, 10.0) for w in wid[big]))
    res = 0
    if pk.size == 1:
        res = int(pk[0])
    elif pk.size >= 2:
        h0, h1 = g2[pk[0]], g2[pk[1]]
        m = min(m, _rel(h1, h0), _rel(h1, 0.5 * h0))
        if h1 > h0:
            res = int(pk[1])
        elif h1 < 0.5 * h0:
            res = int(pk[0])
        else:
            i0 = pk[0] + int(np.argmin(g2[pk[0]:pk[1]]))
            if pk[1] - i0 >= 2:
                r = linregress(np.arange(i0, pk[1]), g2[i0:pk[1]]).rvalue
                m = min(m, _rel(r * r, 0.99))
                res = int(pk[1]) if r * r >= 0.99 else 0
    return m, res


def _reads(chem, max_obs_trace, seed, n):
    spc = make_spc(dict(chem=chem, primary="llr", max_obs_trace=max_obs_trace))
    m = spc.sig_preload_size
    lens = np.array([m if i % 5 else max(3000, synth.pareto_length(seed, i) // 4) for i in range(n)], dtype=np.int32)
    sig, lens = synth.synth_batch(seed, 0, n, m, lens)
    return spc, sig, lens


@pytest.mark.parametrize("chem,max_obs_trace,seed,n", [("RNA004", None, 41, 160), ("RNA002", None, 42, 120), ("RNA004", 60000, 43, 80),
                                                       ("RNA004", 200000, 44, 40)])
def test_threshold_decisions_have_margin(oracle_mod, chem, max_obs_trace, seed, n):
    spc, sig, lens = _reads(chem, max_obs_trace, seed, n)
    rc, np4 = oracle_mod.norm_params(sig, min(spc.core.max_obs_trace, sig.shape[1]), spc.core.sig_norm_outlier_thresh)
    assert rc == 0
    W = spc.llr_boundaries.adapter_peak_width // spc.core.downscale_factor
    worst = {"t1": np.inf, "p1": np.inf, "p4": np.inf}
    flips = 0
    for i in range(n):
        st = oracle_mod.llr_stages(sig[i], spc, np4)
        g1, g2 = st["g1"], st["g2"]
        if g1.size < 12:
            continue
        # T1: the sign of the gains (the terms it is the difference of are of the order of len * |log var|)
        fin = np.isfinite(g1) & (g1 != 0.0)
        scale = max(1.0, float(np.nanmax(np.abs(g1[np.isfinite(g1)]), initial=1.0)))
        if fin.any():
            worst["t1"] = min(worst["t1"], float(np.min(np.abs(g1[fin]))) / scale)
        m1, thr, first = _p1_margins(g1, spc.llr_boundaries.adapter_peak_prominence, W, spc.llr_boundaries.adapter_peak_rel_height)
        worst["p1"] = min(worst["p1"], m1)
        if st["cand"] >= 0 and g2.size:
            m4, res4 = _p4_margins(g2)
            worst["p4"] = min(worst["p4"], m4)
            assert res4 == st["polya_idx"], (i, res4, st["polya_idx"])  # (this restatement takes the oracle's decisions)
        # one ulp up / down on every COMPUTED trace value (the zeros outside the offsets are zeros in any implementation):
        # the decisions stay
        for sgn in (np.inf, -np.inf):
            p1 = np.where(g1 != 0.0, np.nextafter(g1, sgn), g1)
            _, _, f2 = _p1_margins(p1, spc.llr_boundaries.adapter_peak_prominence, W, spc.llr_boundaries.adapter_peak_rel_height)
            flips += f2 != first
            if st["cand"] >= 0 and g2.size:
                flips += _p4_margins(np.where(g2 != 0.0, np.nextafter(g2, sgn), g2))[1] != res4
        # ... and independent random absolute errors of 1e-15 of the largest magnitude of the trace on every computed point
        rng = np.random.default_rng(1000 * seed + i)
        for g, kind in ((g1, 1), (g2, 2)):
            if kind == 2 and not (st["cand"] >= 0 and g2.size):
                continue
            finite = np.isfinite(g)
            mag = float(np.max(np.abs(g[finite]), initial=1.0))
            noisy = np.where((g != 0.0) & finite, g + rng.uniform(-1.0, 1.0, g.size) * 1e-15 * mag, g)
            if kind == 1:
                flips += _p1_margins(noisy, spc.llr_boundaries.adapter_peak_prominence, W, spc.llr_boundaries.adapter_peak_rel_height)[2] != first
            else:
                flips += _p4_margins(noisy)[1] != res4
    print("smallest relative margins (%s, T=%s): %s" % (chem, max_obs_trace, {k: "%.3g" % v for k, v in worst.items()}))
    assert flips == 0
    assert worst["p1"] > BAR and worst["p4"] > BAR and worst["t1"] > 1e-9, worst

"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden rows.

Bars (BASELINE.json north_star): integer boundary indices bit-exact; float statistics within
1e-5 relative (they are in fact expected to be bit-identical: float32 statistics are computed
in numpy's order on the device); LLR trace values within 1e-9 relative (device log vs glibc log).
"""
import numpy as np
import pytest

from golden_cases import CASES
from util import INDEX_FIELDS, load_case, row_diffs

pytestmark = pytest.mark.gpu

LLR_CASES = [k for k, c in CASES.items() if c["primary"] == "llr"]


@pytest.fixture(scope="module")
def hip():
    from adapted_amd import lib

    L = lib.load()
    assert L.adp_device_count() >= 1, L.adp_last_error()
    return lib


def _engine(hip, spc, n, m):
    return hip.Engine(spc, n, m, device=0)


def test_synth_device_matches_host(hip):
    from adapted_amd import synth
    from util import make_spc

    spc = make_spc(CASES["rna004_llr_default"])
    m = spc.sig_preload_size
    n = 24
    eng = _engine(hip, spc, n, m)
    lens = np.array([m, 9000, m + 10, 1500] * 6, dtype=np.int32)
    dsig = eng.dev_alloc(n * m * 4)
    dlen = eng.dev_alloc(n * 4)
    eng.h2d(dlen, lens)
    eng.synth_fill(dsig, dlen, n, seed=7, first_read=123)
    got = np.zeros((n, m), dtype=np.float32)
    eng.d2h(got, dsig)
    want, _ = synth.synth_batch(7, 123, n, m, lens)
    assert np.array_equal(got, want, equal_nan=True)
    eng.dev_free(dsig)
    eng.dev_free(dlen)
    eng.close()


@pytest.mark.parametrize("name", ["rna004_llr_default", "rna002_llr_default", "rna004_llr_200k"])
def test_llr_stages_vs_oracle(hip, oracle_mod, name):
    case, spc, sig, lens, _ = load_case(name)
    mb = case["minibatch"]
    n = mb
    sig, lens = sig[:n], lens[:n]
    m = sig.shape[1]
    eng = _engine(hip, spc, n, m)
    # N1
    eng.debug_llr_upto(sig, lens, n, mb, 1)
    rc, np4 = oracle_mod.norm_params(sig, spc.core.max_obs_trace, spc.core.sig_norm_outlier_thresh)
    got = eng.debug_norm_params(1)[0]
    assert list(got) == list(np4), (got, np4)
    # D1 + n_valid
    eng.debug_llr_upto(sig, lens, n, mb, 2)
    nvalid = eng.debug_fetch(1, n)
    down = eng.debug_fetch(2, n)
    stages = [oracle_mod.llr_stages(sig[k], spc, np4) for k in range(n)]
    for k in range(n):
        assert nvalid[k] == stages[k]["n_valid"], k
        assert np.array_equal(down[k, : nvalid[k]], stages[k]["down"]), k
    # G1
    eng.debug_llr_upto(sig, lens, n, mb, 4)
    g1 = eng.debug_fetch(3, n)
    t1 = eng.debug_fetch(7, n)
    for k in range(n):
        a, b = g1[k, : nvalid[k]], stages[k]["g1"]
        fin = np.isfinite(b)
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isposinf(a), np.isposinf(b)), k
        assert np.allclose(a[fin], b[fin], rtol=1e-9, atol=1e-9), (k, np.max(np.abs(a[fin] - b[fin])))
        st, en = oracle_start_end(b)
        assert (int(t1[k, 0]), int(t1[k, 1])) == (st, en), k
    # adapter candidate
    eng.debug_llr_upto(sig, lens, n, mb, 5)
    aidx = eng.debug_fetch(4, n)
    for k in range(n):
        assert int(aidx[k]) == stages[k]["cand"], (k, int(aidx[k]), stages[k]["cand"], stages[k]["raw_first"])
    # G2 + poly(A)
    eng.debug_llr_upto(sig, lens, n, mb, 7)
    g2 = eng.debug_fetch(3, n)
    pidx = eng.debug_fetch(5, n)
    for k in range(n):
        if stages[k]["cand"] < 0:
            assert int(pidx[k]) == 0
            continue
        a, b = g2[k, : nvalid[k]], stages[k]["g2"]
        fin = np.isfinite(b) & np.isfinite(a)
        # the point next to the boundary is cumulative-sum rounding noise: +-inf / NaN must agree
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isinf(a), np.isinf(b)), k
        assert np.allclose(a[fin], b[fin], rtol=1e-9, atol=1e-9), k
        assert int(pidx[k]) == max(stages[k]["polya_idx"], 0), (k, int(pidx[k]), stages[k]["polya_idx"])
    eng.close()


def oracle_start_end(g):
    pos = ~(g <= 0)
    if not pos.any():
        return 0, g.size - 1
    idx = np.flatnonzero(pos)
    return int(idx[0]), int(idx[-1])


@pytest.mark.parametrize("name", LLR_CASES)
def test_llr_rows_vs_golden_and_oracle(hip, oracle_mod, name):
    case, spc, sig, lens, want = load_case(name)
    n, m = sig.shape
    eng = _engine(hip, spc, n, m)
    rows, mbs = eng.detect_llr_rows(sig, lens, n, case["minibatch"])
    assert (mbs == 0).all()
    got = hip.rows_to_results(rows, "llr")
    # integer fields bit-exact, floats within 1e-5 (count the non-identical ones)
    bad = [(i, d) for i, (g, w) in enumerate(zip(got, want)) for d in row_diffs(g, w, float_rel=1e-5)]
    assert not bad, bad[:10]
    inexact = [(i, d) for i, (g, w) in enumerate(zip(got, want)) for d in row_diffs(g, w, float_rel=0.0)]
    print("%s: %d float fields differ in the last bits" % (name, len(inexact)), inexact[:5])
    for i, (g, w) in enumerate(zip(got, want)):
        for f in INDEX_FIELDS:
            assert getattr(g, f, None) == w.get(f), (i, f)
    eng.close()


@pytest.mark.parametrize("kind", ["bimodal_rows", "nan_sample", "constant", "negatives"])
def test_n1_guess_miss_falls_back_exactly(hip, oracle_mod, kind):
    """N1 guesses a key window from every row_step-th read; when the guess misses, the verified
    fallback must still return the exact numpy nanmedian / MAD."""
    from adapted_amd import synth
    from util import make_spc

    spc = make_spc(CASES["rna004_llr_default"])
    m = spc.sig_preload_size
    n = 64  # row_step = 2: even rows are the sample
    lens = np.full(n, m, dtype=np.int32)
    sig, _ = synth.synth_batch(3, 0, n, m, lens)
    if kind == "bimodal_rows":
        sig[0::2] += np.float32(200.0)
    elif kind == "nan_sample":
        sig[0::2] = np.nan
        lens[0::2] = 0
    elif kind == "constant":
        sig[:, :] = np.float32(77.25)
        sig[1, 5] = np.float32(80.0)
    elif kind == "negatives":
        sig[1::2] *= np.float32(-1.0)
    eng = _engine(hip, spc, n, m)
    eng.debug_llr_upto(sig, lens, n, n, 1)
    got = eng.debug_norm_params(1)[0]
    rc, want = oracle_mod.norm_params(sig, spc.core.max_obs_trace, spc.core.sig_norm_outlier_thresh)
    assert list(got) == list(want), (kind, got, want)
    eng.close()


@pytest.mark.parametrize("kind", ["plain", "odd_count", "bimodal_rows", "nan_sample", "constant", "negatives", "two_values",
                                  "adc_grid", "adc_grid_odd", "half_grid", "coarse_grid"])
def test_n1_fused_single_pass_is_exact_or_falls_back(hip, oracle_mod, kind, monkeypatch):
    """Big minibatches get median and MAD from one pass (n1_fused.h); forced here on a small batch.  Whatever the
    sampled brackets cannot prove must fall through to the multi-pass selection: the result is always numpy's."""
    from adapted_amd import synth
    from util import make_spc

    monkeypatch.setenv("ADP_N1_FUSED_MIN", "1")
    spc = make_spc(CASES["rna004_llr_default"])
    m = spc.sig_preload_size
    n = 96
    lens = np.full(n, m, dtype=np.int32)
    sig, _ = synth.synth_batch(5, 0, n, m, lens)
    if kind == "odd_count":
        sig[3, 17] = np.nan
    elif kind == "bimodal_rows":
        sig[0::2] += np.float32(200.0)
    elif kind == "nan_sample":
        sig[0::3] = np.nan
        lens[0::3] = 0
    elif kind == "constant":
        sig[:, :] = np.float32(77.25)
        sig[1, 5] = np.float32(80.0)
    elif kind == "negatives":
        sig[1::2] *= np.float32(-1.0)
    elif kind == "two_values":
        sig[:, :] = np.float32(50.0)
        sig[:, 1::2] = np.float32(90.0)
    elif kind in ("adc_grid", "adc_grid_odd"):  # calibrated int16 data: one grid for the whole minibatch (heavy keys)
        sig = (np.round(sig / np.float32(0.18)) * np.float32(0.18)).astype(np.float32)
        if kind == "adc_grid_odd":
            sig[3, 17] = np.nan
    elif kind == "half_grid":  # every other read on the grid, the rest continuous
        sig[0::2] = (np.round(sig[0::2] / np.float32(0.18)) * np.float32(0.18)).astype(np.float32)
    elif kind == "coarse_grid":
        sig = (np.round(sig / np.float32(3.0)) * np.float32(3.0)).astype(np.float32)
    eng = _engine(hip, spc, n, m)
    before = eng.debug_counters().copy()
    eng.debug_llr_upto(sig, lens, n, n, 1)
    got = eng.debug_norm_params(1)[0]
    after = eng.debug_counters()
    rc, want = oracle_mod.norm_params(sig, spc.core.max_obs_trace, spc.core.sig_norm_outlier_thresh)
    assert list(got) == list(want), (kind, got, want)
    print(kind, "fused attempts/median misses/MAD misses:", (after - before)[5:8])
    # (small batches are clustered samples: the brackets may well miss, or not be set up at all)
    eng.close()


def test_width_walk_stops_at_the_prominence_base(hip, oracle_mod):
    """scipy's peak_widths walks no further than the prominence base of each side.  With rel_height = 1 the evaluation
    height equals the base's value up to rounding, so a walk that stops by value alone may run past the base to the end of
    the clipped trace: here the adapter peak is 74.69 pooled points wide against a threshold of 75 (RNA002, a 100-point
    trace) -- one more point flips "No adapter detected" into a detection.  (Found by tests/soak_vs_oracle.py.)"""
    from adapted_amd import lib, synth
    from adapted_amd.config import get_chemistry_specific_config

    spc = get_chemistry_specific_config("RNA002")
    spc.llr_boundaries.llr_detect, spc.cnn_boundaries.cnn_detect = True, False
    spc.core.max_obs_trace = 4000
    spc.med_shift.detect_med_shift = True
    spc.update_primary_method()
    spc.update_sig_preload_size()
    m, n, it = spc.sig_preload_size, 160, 15
    lo = spc.core.min_obs_adapter + 2 * spc.core.downscale_factor + 8
    lens = np.array([max(lo, synth.pareto_length(it, i, lo=3000, hi=4 * m)) for i in range(n)], dtype=np.int32)
    sig, lens = synth.synth_batch(100 + it, 0, n, m, lens)
    sig, lens = sig[80:], lens[80:]
    eng = _engine(hip, spc, 80, m)
    rows, mbs = eng.detect_llr_rows(sig, lens, 80, 80, with_start_peak=True)
    got = lib.rows_to_results(rows, "llr")
    want = oracle_mod.detect_llr(sig, lens, spc, with_start_peak=True)
    assert want[3]["llr_adapter_end"] == 0 and want[3]["fail_reason"] == "No adapter detected (primary)"
    from util import row_diffs
    bad = [(i, d) for i, (g, w) in enumerate(zip(got, want)) for d in [row_diffs(g, {k: v for k, v in w.items() if not k.startswith("_")})] if d]
    assert not bad, bad[:3]
    eng.close()

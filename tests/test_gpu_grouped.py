"""Grouped (software-pipelined) execution of adp_detect_llr: a call with several minibatches is cut into groups that run
over two or more internal streams, phases of neighbouring groups overlapping.  Minibatches are independent (N1 couples only
the reads of one minibatch: reference adapted/detect/normalize.py:15-22 as called at combined.py:128-132), so rows must be the
same BYTES for every grouping, lane count and phase ordering, equal to the one-stream serial pipeline and to the oracle."""
import os

import numpy as np
import pytest

from test_gpu_scale import _rows_equal, _spc200k

pytestmark = pytest.mark.gpu

KNOBS = [{"ADP_GROUPS": "1"}, {"ADP_GROUPS": "0"}, {"ADP_GROUPS": "2"}, {"ADP_GROUPS": "6", "ADP_LANES": "3"}, {"ADP_STAGGER": "7", "ADP_GROUPS": "0"},
         {"ADP_STAGGER": "0", "ADP_LANES": "4", "ADP_GROUPS": "12"}, {"ADP_LANES": "1", "ADP_GROUPS": "3"}]


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in ("ADP_GROUPS", "ADP_LANES", "ADP_STAGGER")}
    try:
        for k in old:
            os.environ.pop(k, None)
        os.environ.update(env)
        return fn()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def _canon(rows, lib):
    """rows with the overflow lists resolved and the registry tokens blanked: comparable across calls"""
    lists = {int(i): lib._OPEN_PORES_MORE[int(rows[i]["open_pores_more"])].tolist() for i in np.flatnonzero(rows["n_open_pores"] > lib.MAX_OPEN_PORES)}
    r = rows.copy()
    r["open_pores_more"] = 0
    return r.tobytes(), lists


def test_grouped_rows_equal_serial_rows_and_the_oracle(oracle_mod):
    from adapted_amd import lib, synth
    from adapted_amd.config import get_chemistry_specific_config

    spc = get_chemistry_specific_config("RNA004")
    spc.llr_boundaries.llr_detect, spc.cnn_boundaries.cnn_detect = True, False
    spc.update_primary_method()
    spc.update_sig_preload_size()
    m = spc.sig_preload_size
    mb, n_mb = 32, 7                                  # 7 minibatches: groups of unequal size, a ragged last one
    n = mb * n_mb - 5
    lens = np.array([m if i % 3 else max(1200, synth.pareto_length(5, i, lo=1200, hi=2 * m)) for i in range(n)], dtype=np.int32)
    sig, lens = synth.synth_batch(71, 0, n, m, lens)
    for r in (4, 40, 100, 200):                       # open_pores lists beyond a row's 16 entries, in different groups
        for j in range(20 + r % 7):
            sig[r, 120 + 40 * j: 123 + 40 * j] = 260.0
    sig[3 * mb: 4 * mb] = 80.0                        # minibatch 3: MAD == 0, dropped (status 1); its neighbours are untouched
    eng = lib.Engine(spc, n, m, device=0)
    ref = None
    for env in KNOBS:
        rows, mbs = _with_env(env, lambda: eng.detect_llr_rows(sig, lens, n, mb, with_start_peak=True))
        assert list(mbs) == [0, 0, 0, 1, 0, 0, 0], (env, mbs)
        cur = _canon(rows, lib)
        if ref is None:
            ref, ref_rows = cur, rows
        assert cur[0] == ref[0] and cur[1] == ref[1], env
        assert sorted(cur[1]) == [4, 40, 200], cur[1].keys()   # (read 100 sits in the dropped minibatch)
        # N1 parameters of every minibatch travel back through the grouped call as well
        prm = eng.debug_norm_params(n_mb)
        if env == KNOBS[0]:
            prm0 = prm.copy()
        keep = [k for k in range(n_mb) if k != 3]
        assert np.array_equal(prm[keep], prm0[keep]), env
    # per-kernel times of a grouped call: every group's launches are reported
    eng.set_profiling(True)
    _with_env({"ADP_GROUPS": "4"}, lambda: eng.detect_llr_rows(sig, lens, n, mb, with_start_peak=True))
    names = [k for k, _ in eng.kernel_times()]
    eng.set_profiling(False)
    assert names.count("k_partition_stats") == 4 and names.count("k_gains<1>") == 4, names
    # and the oracle, minibatch by minibatch
    got = lib.rows_to_results(ref_rows, "llr")
    for k in (0, 2, 6):
        a, b = k * mb, min(n, (k + 1) * mb)
        want = oracle_mod.detect_llr(sig[a:b], lens[a:b], spc, with_start_peak=True)
        assert not _rows_equal(got[a:b], want), (k, _rows_equal(got[a:b], want)[:6])
    eng.close()


def test_grouped_equals_serial_at_the_200k_window_on_device_rows():
    """the headline shape: device-resident input, rows delivered to a device buffer (bench.py's call), int16 twin included"""
    from adapted_amd import lib

    spc = _spc200k()
    m = spc.sig_preload_size
    mb, n = 250, 1500
    eng = lib.Engine(spc, n, m, device=0)
    dsig, dlen, drows = eng.dev_alloc(n * m * 4), eng.dev_alloc(n * 4), eng.dev_alloc(n * lib.ROW_DTYPE.itemsize)
    eng.h2d(dlen, np.full(n, m, dtype=np.int32))
    eng.synth_fill(dsig, dlen, n, seed=21, first_read=7000)
    out = []
    for env in ({}, {"ADP_GROUPS": "0"}, {"ADP_GROUPS": "5", "ADP_LANES": "3"}):
        _, mbs = _with_env(env, lambda: eng.detect_llr_rows(dsig, dlen, n, mb, with_start_peak=True, device_ptrs=True, rows_dev=drows, tails_nan=True))
        assert (mbs == 0).all()
        rows = np.zeros(n, dtype=lib.ROW_DTYPE)
        eng.d2h(rows, drows)
        rows["open_pores_more"] = 0
        out.append(rows.tobytes())
    assert out[0] == out[1] == out[2]
    rows = np.frombuffer(out[0], dtype=lib.ROW_DTYPE)
    assert rows["success"].mean() > 0.8
    for p in (dsig, dlen, drows):
        eng.dev_free(p)
    eng.close()


def test_cnn_chunks_over_lanes_equal_one_chunk():
    """adp_detect_cnn cuts a call into chunks of whole minibatches over two lanes (conv stack of one beside the candidate
    validation of the other); find_peaks / row compaction are per minibatch (reference adapted/detect/cnn.py:136-160), so rows and
    predictions must not depend on the chunking."""
    from adapted_amd import lib, synth
    from adapted_amd.config import get_chemistry_specific_config
    from adapted_amd.detect import cnn

    spc = get_chemistry_specific_config("RNA004")
    spc.llr_boundaries.llr_detect, spc.cnn_boundaries.cnn_detect = False, True
    spc.update_primary_method()
    spc.update_sig_preload_size()
    m = spc.sig_preload_size
    mb, n = 16, 16 * 7 - 3
    lens = np.array([m if i % 3 else max(1200, synth.pareto_length(5, i, lo=1200, hi=2 * m)) for i in range(n)], dtype=np.int32)
    sig, lens = synth.synth_batch(77, 0, n, m, lens)
    for r in (4, 40, 100):
        for j in range(20 + r % 7):
            sig[r, 120 + 40 * j: 123 + 40 * j] = 260.0
    eng = lib.Engine(spc, n, m, device=0)
    cnn.ensure_weights(eng, None, spc)
    ref = None
    for env in ({"ADP_CNN_GROUPS": "1"}, {"ADP_CNN_GROUPS": "0"}, {"ADP_CNN_GROUPS": "7", "ADP_CNN_LANES": "3"}, {"ADP_CNN_GROUPS": "2", "ADP_CNN_LANES": "1"}):
        def call():
            old = {k: os.environ.pop(k, None) for k in ("ADP_CNN_GROUPS", "ADP_CNN_LANES")}
            os.environ.update(env)
            try:
                return eng.detect_cnn_rows(sig, lens, n, mb)
            finally:
                for k in ("ADP_CNN_GROUPS", "ADP_CNN_LANES"):
                    os.environ.pop(k, None)
                    if old[k] is not None:
                        os.environ[k] = old[k]
        rows, bounds = call()
        cur = _canon(rows, lib) + (bounds.tobytes(),)
        if ref is None:
            ref = cur
        assert cur == ref, env
    assert sorted(ref[1]) == [4, 40, 100]
    assert np.frombuffer(ref[0], dtype=lib.ROW_DTYPE)["success"].sum() > n // 3
    eng.close()


@pytest.mark.parametrize("window", [None, 60000])
def test_start_peak_riding_the_pooling_pass_equals_the_separate_scan(oracle_mod, window):
    """K1 (reference adapted/detect/start_peak.py:7-119) inside k_norm_pool + k_sp_head / k_sp_tail against the separate
    k_start_peak sweep (ADP_SP_FUSED=0) and the oracle: open pores in front of min_obs_adapter, inside the pooled range and
    behind max_obs_trace, start peaks that nothing exceeds (the scan runs to the read's end), reads shorter than every range,
    NaN tails with and without ADP_TAILS_NAN."""
    from adapted_amd import lib, synth
    from adapted_amd.config import get_chemistry_specific_config

    spc = get_chemistry_specific_config("RNA004")
    spc.llr_boundaries.llr_detect, spc.cnn_boundaries.cnn_detect = True, False
    if window:
        spc.core.max_obs_trace = window
    spc.update_primary_method()
    spc.update_sig_preload_size()
    m, T = spc.sig_preload_size, spc.core.max_obs_trace
    n = 96
    rng = np.random.default_rng(5)
    lens = np.array([m + 100 if i % 3 == 0 else max(1012, synth.pareto_length(11, i, lo=1200, hi=2 * m)) for i in range(n)], dtype=np.int32)
    lens[1], lens[2], lens[3] = 1012, 2600, m
    sig, lens = synth.synth_batch(91, 0, n, m, lens)
    have = np.minimum(lens, m)
    for i in range(n):
        kind = i % 8
        e = int(have[i]) // 10                      # the open-pore scan looks at raw[:e] only (the reference's quirk)
        if kind == 1 and e > 200:
            sig[i, int(rng.integers(160, min(e, 990)))] = 230.0      # open pore in front of min_obs_adapter
        elif kind == 2 and e > 1300:
            p = 10 * int(rng.integers(110, e // 10))
            sig[i, p: p + 10] = 230.0                               # ... inside the pooled range, a whole block: next-greater block = open pore
        elif kind == 3:
            sig[i, 100:1500] = np.minimum(sig[i, 100:1500], 60.0)   # a low start region: the next block above it comes early
        elif kind == 4:
            sig[i, 300:340] = 400.0                                 # a start peak nothing exceeds: the scan runs to the end
        elif kind == 5 and have[i] > T + 200:
            sig[i, 300:340] = 180.0
            sig[i, T + 40: T + 60] = 900.0                          # ... except a block behind max_obs_trace
        elif kind == 6 and have[i] > 4000:
            sig[i, 2505:2515] = 300.0                               # the block right at pooled index 250 / 251
    eng = lib.Engine(spc, n, m, device=0)
    outs = []
    for fused in ("1", "0"):
        for tails in (False, True):
            os.environ["ADP_SP_FUSED"] = fused
            try:
                rows, mbs = eng.detect_llr_rows(sig, lens, n, n // 2, with_start_peak=True, tails_nan=tails)
            finally:
                os.environ.pop("ADP_SP_FUSED", None)
            assert (mbs == 0).all()
            outs.append(_canon(rows, lib))
    assert all(o == outs[0] for o in outs[1:])
    rows = np.frombuffer(outs[0][0], dtype=lib.ROW_DTYPE)
    assert (rows["start_peak_type"] > 0).sum() >= 2 and (rows["present"] >> np.uint64(22) & np.uint64(1)).sum() > n // 2   # flags and columns are exercised
    got = lib.rows_to_results(rows, "llr")
    for a in (0, n // 2):
        want = oracle_mod.detect_llr(sig[a:a + n // 2], lens[a:a + n // 2], spc, with_start_peak=True)
        bad = _rows_equal(got[a:a + n // 2], want)
        assert not bad, bad[:6]
    eng.close()


def test_start_peak_primary_several_minibatches_in_one_call_share_one_arena(oracle_mod):
    """adp_detect_start_peak with n_reads > minibatch: ONE open-pore arena for the call -- the overflow lists of reads in the first
    minibatch must still be there (and at their offsets) when the call returns (launch_validate used to reset the
    arena per minibatch).  Rows equal the minibatch-by-minibatch calls and the oracle."""
    from adapted_amd import lib, synth
    from adapted_amd.config import get_chemistry_specific_config

    spc = get_chemistry_specific_config("RNA004")
    spc.llr_boundaries.llr_detect = spc.cnn_boundaries.cnn_detect = False
    spc.rna_start_peak.detect_rna_start_peak = True
    spc.mvs_polya.mvs_detect_check = False
    spc.update_primary_method()
    spc.update_sig_preload_size()
    m = spc.sig_preload_size
    mb, n = 24, 72
    sig, lens = synth.synth_batch(55, 0, n, m, np.full(n, m, dtype=np.int32))
    for r, cnt in ((3, 30), (30, 22), (50, 41), (71, 18)):   # more open pores than a row holds, in all three minibatches
        for j in range(cnt):
            sig[r, 120 + 40 * j: 123 + 40 * j] = 260.0
    eng = lib.Engine(spc, n, m, device=0)
    rows = eng.detect_start_peak_rows(sig, lens, n, mb)
    one = _canon(rows, lib)
    parts = [eng.detect_start_peak_rows(sig[a:a + mb], lens[a:a + mb], mb, mb) for a in range(0, n, mb)]
    per = _canon(np.concatenate(parts), lib)
    assert one == per
    assert sorted(one[1]) == [3, 30, 50, 71] and [len(one[1][k]) for k in (3, 30, 50, 71)] == [29, 21, 40, 17]  # (find_open_pores never keeps the first position: reference anomalies.py:15-35)
    got = lib.rows_to_results(rows, "start_peak")
    for a in range(0, n, mb):
        want = oracle_mod.detect_start_peak(sig[a:a + mb], lens[a:a + mb], spc)
        bad = _rows_equal(lib.open_pore_float_column(got[a:a + mb]), want)
        assert not bad, bad[:6]
    eng.close()

"""The kernels that stay in the tree beside the defaults must keep giving the same BYTES: every switch below selects another
implementation of the same reference rows -- one wave per moving-window recurrence (`ADP_SERIES_PIPE=0`) against the pipeline of
waves (series_pipe.h), layer 3 of the conv stack as a kernel of its own (`ADP_CNN_FOLD=0`; then also in turn on one stream,
`ADP_CNN_OVERLAP=0`, against a chunk's last layer beside the next chunk's first) against layer 3 in layer 2's epilogue, every sampled row in the first level of N1's sample (`ADP_N1_S0=1`), the lane-per-read series kernel on the LLR
path (`ADP_SERIES_PIPE_LLR=0`).  (The variants that lost their A/B in rounds 2-4 left the product in round 5:
tools/experiments/r05_pruned_variants.patch.)  Reference rows: V1-V4 adapted/detect/combined.py:358-631, mvs.py:45-158; C2
adapted/detect/cnn.py:16-52."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SWITCHES = ("ADP_SERIES_PIPE", "ADP_CNN_OVERLAP", "ADP_N1_S0", "ADP_SERIES_PIPE_LLR", "ADP_CNN_FOLD", "ADP_CNN_FUSE_IN", "ADP_CUMSUM_GATHER")


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in SWITCHES}
    try:
        for k in SWITCHES:
            os.environ.pop(k, None)
        os.environ.update(env)
        return fn()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def _canon(rows, lib):
    lists = {int(i): lib._OPEN_PORES_MORE[int(rows[i]["open_pores_more"])].tolist() for i in np.flatnonzero(rows["n_open_pores"] > lib.MAX_OPEN_PORES)}
    r = rows.copy()
    r["open_pores_more"] = 0
    return r.tobytes(), lists


def _spc(primary, max_obs_trace=None, **over):
    from adapted_amd.config import get_chemistry_specific_config

    spc = get_chemistry_specific_config("RNA004")
    spc.llr_boundaries.llr_detect = primary == "llr"
    spc.cnn_boundaries.cnn_detect = primary == "cnn"
    if max_obs_trace:
        spc.core.max_obs_trace = max_obs_trace
    for k, v in over.items():
        sec, name = k.split("__")
        setattr(getattr(spc, sec), name, v)
    spc.update_primary_method()
    spc.update_sig_preload_size()
    return spc


@pytest.mark.parametrize("window, k, quantise, windows", [(None, 10, 0.0, None), (60000, 10, 0.18, None), (200000, 10, 0.0, None), (200000, 15, 0.18, None),
                                                           (34000, 3, 0.0, (48, 12)), (34000, 10, 0.0, (101, 23))])
def test_cnn_path_variants_give_the_same_rows(window, k, quantise, windows):
    from adapted_amd import lib, synth
    from adapted_amd.detect import cnn

    over = {"cnn_boundaries__polya_cand_k": k}
    if windows:
        over["mvs_polya__pA_var_window"], over["mvs_polya__pA_mean_window"] = windows
    spc = _spc("cnn", window, **over)
    m = spc.sig_preload_size
    n = 64 if m > 100000 else 128
    rng = np.random.default_rng(k + (window or 0))
    lens = np.array([m if rng.random() < 0.6 else max(1012, synth.pareto_length(9, i, lo=3000, hi=3 * m)) for i in range(n)], dtype=np.int32)
    sig, lens = synth.synth_batch(900 + k, 0, n, m, lens)
    if quantise:
        q = np.float32(quantise)
        sig = (np.round(sig / q) * q).astype(np.float32)
    for r in (3, 17):                                  # open_pores lists beyond a row's 16 entries
        for j in range(19):
            sig[r, 120 + 40 * j: 123 + 40 * j] = 260.0
    # NaN holes inside [adapter_end, largest candidate) -- early in the slice and inside its last moving window: every series producer
    # must leave have_series = 0 for such a read (k_validate skips its NaN scan for slices whose series came from a series kernel)
    nan_reads = []
    for r, at in ((11, 7000), (23, 9000), (40, min(m, int(lens[40])) - 60)):
        if at > 6000:
            sig[r, at: at + 3] = np.nan
            nan_reads.append((r, at))
    ref = None
    # (ADP_CNN_FOLD=0: layer 3 as a kernel of its own sums the same products in another order -- scores differ in their last bits, which
    # flips a near-tied candidate on about one read in 10^4: not on these)
    # (ADP_CNN_FUSE_IN=0: layer 0 as a kernel of its own is a float32 fmaf chain, in layer 1's prologue it is a split-operand product: the
    # same class of last-bit difference)
    for env in ({}, {"ADP_SERIES_PIPE": "0"}, {"ADP_CNN_FOLD": "0", "ADP_CNN_OVERLAP": "0"}, {"ADP_CNN_FOLD": "0"}, {"ADP_CNN_FUSE_IN": "0"},
                {"ADP_CNN_FUSE_IN": "0", "ADP_CNN_FOLD": "0"}, {"ADP_SERIES_PIPE": "0", "ADP_CNN_FOLD": "0", "ADP_CNN_OVERLAP": "0"}):
        def run():
            eng = lib.Engine(spc, n, m, device=0)
            try:
                cnn.ensure_weights(eng, None, spc)
                rows, bounds = eng.detect_cnn_rows(sig, lens, n, n // 2)
                hs = eng.debug_fetch(9, n)              # have_series: never set for a slice with a NaN in it
                bb = np.asarray(bounds).reshape(n, -1)
                for r, at in nan_reads:
                    if bb[r, 0] <= at < bb[r, 1:].max():
                        assert hs[r] == 0, (env, r, at, bb[r])
                return _canon(rows, lib), bounds.tobytes()
            finally:
                eng.close()
        got = _with_env(env, run)
        if ref is None:
            ref = got
            assert np.frombuffer(got[1], dtype=np.int64).any()
        assert got == ref, env


@pytest.mark.parametrize("window", [None, 200000])
def test_llr_path_switches_give_the_same_rows(window):
    from adapted_amd import lib, synth

    spc = _spc("llr", window)
    m = spc.sig_preload_size
    n = 96
    lens = np.array([m if i % 4 else max(1012, synth.pareto_length(3, i, lo=1500, hi=2 * m)) for i in range(n)], dtype=np.int32)
    sig, lens = synth.synth_batch(77, 0, n, m, lens)
    sig[5, 3000:3004] = np.nan                          # a NaN hole inside a read
    for j in range(21):
        sig[9, 120 + 40 * j: 123 + 40 * j] = 260.0      # an open_pores list beyond 16 entries
    eng = lib.Engine(spc, n, m, device=0)
    a, _ = _with_env({}, lambda: eng.detect_llr_rows(sig, lens, n, 48, with_start_peak=True))
    d, _ = _with_env({"ADP_N1_S0": "1", "ADP_SERIES_PIPE_LLR": "0"}, lambda: eng.detect_llr_rows(sig, lens, n, 48, with_start_peak=True))  # (every sampled row in N1's first level; the lane-per-read series kernel)
    eng.close()
    assert _canon(a, lib) == _canon(d, lib)


def test_polya_peak_on_a_prefix_of_the_maxima_equals_the_whole_list():
    """P4 (reference adapted/detect/llr.py:406-479): k_polya_peak settles the distance rule and the survivors on the first 224 maxima and
    takes the whole list only when that does not show the second survivor; ADP_ABLATE bit 2^24 (read when the engine is made) runs the
    whole list at once, as before round 4.  Reads with few, one or no survivor (flat tails, short reads) take the second attempt."""
    from adapted_amd import lib, synth

    spc = _spc("llr", 200000)
    m = spc.sig_preload_size
    n = 128
    lens = np.array([m if i % 3 else max(1012, synth.pareto_length(5, i, lo=1500, hi=2 * m)) for i in range(n)], dtype=np.int32)
    sig, lens = synth.synth_batch(91, 0, n, m, lens)
    rng = np.random.default_rng(4)
    for r in range(0, n, 5):                             # no change point behind the adapter: no (or a late, single) survivor
        a = int(rng.integers(8000, 20000))
        sig[r, a:lens[r]] = (80.0 + rng.normal(0.0, 4.0, max(0, int(lens[r]) - a))).astype(np.float32)[: max(0, min(m, int(lens[r])) - a)]
    got = []
    for env in ({}, {"ADP_ABLATE": "16777216"}):
        def run():
            eng = lib.Engine(spc, n, m, device=0)
            try:
                rows, _ = eng.detect_llr_rows(sig, lens, n, 64, with_start_peak=True)
                return _canon(rows, lib), rows["col"][:, lib.COLS.index("{primary}_polya_end")].copy()
            finally:
                eng.close()
        old = os.environ.get("ADP_ABLATE")
        try:
            os.environ.pop("ADP_ABLATE", None)
            os.environ.update(env)
            got.append(run())
        finally:
            os.environ.pop("ADP_ABLATE", None)
            if old is not None:
                os.environ["ADP_ABLATE"] = old
    assert got[0][0] == got[1][0]
    pe = got[0][1]
    assert (pe > 0).sum() >= n // 3 and (pe == 0).sum() >= 5   # both kinds of read are there


def test_cumsum_kernels_share_a_launch_by_the_lengths_of_a_waves_reads():
    """k_cumsum_gather takes the waves whose 64 reads have one length, k_cumsum the others (round 5): a batch with a wave of full-length reads,
    a wave of mixed lengths, a wave of equal SHORT reads (fewer pooled samples than one gathered line holds) and a ragged last wave gives the
    rows of the launch in which k_cumsum takes every wave (ADP_CUMSUM_GATHER=0)."""
    from adapted_amd import lib, synth

    spc = _spc("llr", None)
    m = spc.sig_preload_size
    n = 200
    lens = np.full(n, m, dtype=np.int32)
    lens[64:128] = [max(1012, synth.pareto_length(5, i, lo=1500, hi=2 * m)) for i in range(64)]
    lens[128:192] = 2600
    lens[192:] = [m, 5000, m, 1800, m, m, 9000, 1012]
    sig, lens = synth.synth_batch(31, 0, n, m, lens)

    def run():
        eng = lib.Engine(spc, n, m, device=0)
        try:
            rows, _ = eng.detect_llr_rows(sig, lens, n, 100, with_start_peak=True)
            return _canon(rows, lib)
        finally:
            eng.close()
    assert _with_env({}, run) == _with_env({"ADP_CUMSUM_GATHER": "0"}, run)

"""The CPU oracle against the golden vectors made by the REAL reference
(oracle/gen_golden.py, run in the build container).  Bit-exact on every field."""
import numpy as np
import pytest

from golden_cases import CASES
from util import load_case, load_stages, row_diffs

LLR_CASES = [k for k, c in CASES.items() if c["primary"] == "llr"]
SP_CASES = [k for k, c in CASES.items() if c["primary"] == "start_peak"]


def _run(oracle, case, spc, sig, lens):
    rows = []
    mb = case["minibatch"]
    for s in range(0, case["n"], mb):
        if case["primary"] == "llr":
            rows += oracle.detect_llr(sig[s:s + mb], lens[s:s + mb], spc)
        else:
            rows += oracle.detect_start_peak(sig[s:s + mb], lens[s:s + mb], spc)
    return rows


@pytest.mark.parametrize("name", LLR_CASES + SP_CASES)
def test_rows_bit_exact(oracle_mod, name):
    case, spc, sig, lens, want = load_case(name)
    got = _run(oracle_mod, case, spc, sig, lens)
    assert len(got) == len(want)
    bad = [(i, d) for i, (g, w) in enumerate(zip(got, want)) for d in row_diffs(g, w)]
    assert not bad, bad[:10]


@pytest.mark.parametrize("name", [k for k in LLR_CASES if CASES[k]["dump"]])
def test_llr_stages_bit_exact(oracle_mod, name):
    case, spc, sig, lens, _ = load_case(name)
    st = load_stages(name)
    mb = case["minibatch"]
    rc, np4 = oracle_mod.norm_params(sig[:mb], spc.core.max_obs_trace, spc.core.sig_norm_outlier_thresh)
    assert rc == 0
    assert np4[0] == st["n1_med_mad"][0] and np4[1] == st["n1_med_mad"][1]
    for k in st["dump_idx"]:
        k = int(k)
        o = oracle_mod.llr_stages(sig[k], spc, np4)
        assert o["n_valid"] == int(st["n_valid"][k])
        assert np.array_equal(o["down"], st["down_%d" % k])
        assert np.array_equal(o["g1"], st["g1_%d" % k], equal_nan=True)
        p1 = st["p1_%d" % k]
        assert o["raw_first"] == (int(p1[0]) if len(p1) else -1)
        cands = st["cands_%d" % k]
        assert o["cand"] == (int(cands[0]) if len(cands) else -1)
        if len(cands):
            assert np.array_equal(o["g2"], st["g2_%d" % k], equal_nan=True)
            assert o["polya_idx"] == int(st["p4_%d" % k][0])


def test_start_peak_table(oracle_mod):
    import json
    import os

    from util import GOLD

    case, spc, sig, lens, _ = load_case("rna004_start_peak")
    with open(os.path.join(GOLD, "rna004_start_peak.table.json")) as fh:
        tab = json.load(fh)
    got = oracle_mod.start_peak_table(sig, lens, spc)
    for i in range(case["n"]):
        if tab["start_peak_idx"][i] is None:
            assert not got["valid"][i]
            continue
        assert got["valid"][i]
        assert int(got["start_peak_idx"][i]) == int(tab["start_peak_idx"][i])
        assert int(got["next_greater_idx"][i]) == int(tab["next_greater_idx"][i])
        assert float(got["start_peak_pa"][i]) == tab["start_peak_pa"][i]
        assert float(got["next_greater_pa"][i]) == tab["next_greater_pa"][i]
        types = {0: None, 1: "open pore in adapter", 2: "potential concatemer adapter-only read"}
        assert types[int(got["flagged_type"][i])] == tab["flagged_type"][i]
        if tab["open_pore_idx"][i] is not None:
            assert int(got["open_pore_idx"][i]) == int(tab["open_pore_idx"][i])


def test_mad_zero_raises(oracle_mod):
    from util import make_spc

    spc = make_spc(CASES["rna004_llr_default"])
    sig = np.full((4, spc.sig_preload_size), 80.0, dtype=np.float32)
    with pytest.raises(ValueError, match="scale is 0"):
        oracle_mod.detect_llr(sig, np.full(4, spc.sig_preload_size, dtype=np.int32), spc)


CNN_CASES = [k for k, c in CASES.items() if c["primary"] == "cnn"]  # the default window and configs[2]'s 200 k window


@pytest.mark.parametrize("name", CNN_CASES)
def test_cnn_oracle_vs_golden(oracle_mod, name):
    """C1 bit-exact, C2 (numpy conv) within 1e-4, C3 indices identical, rows (V1 with k candidates
    + C4 fallback) bit-exact given the reference's predictions."""
    import os

    from util import GOLD

    case, spc, sig, lens, want = load_case(name)
    st = load_stages(name)
    x = oracle_mod.cnn_prepare(sig, spc)
    for k in st["dump_idx"]:
        assert np.array_equal(x[int(k)], st["prep_%d" % int(k)])
    w = np.load(os.path.join(os.path.dirname(GOLD), "..", "adapted_amd", "models", "rna004_130bps@v0.2.4.npz"))
    sc = oracle_mod.cnn_forward(x, w)
    for k in st["dump_idx"]:
        assert np.max(np.abs(sc[int(k)] - st["scores_%d" % int(k)])) < 1e-4
    preds = oracle_mod.cnn_predict(sc, spc)
    assert np.array_equal(preds, st["preds"])
    got = oracle_mod.detect_cnn_from_preds(sig, lens, st["preds"], spc)
    bad = [(i, d) for i, (g, w_) in enumerate(zip(got, want)) for d in row_diffs(g, w_)]
    assert not bad, bad[:10]


SINGLE_CASES = [k for k, c in CASES.items() if c["primary"] == "llr_single"]


@pytest.mark.parametrize("name", SINGLE_CASES)
def test_single_read_api_vs_golden(oracle_mod, name):
    """combined_detect_llr (adapted/detect/combined.py:39-119): the oracle's restatement against the reference run read by read"""
    import json
    import os

    from util import GOLD, make_spc, row_diffs
    from golden_cases import resolve_lens
    from adapted_amd import synth

    case = CASES[name]
    spc = make_spc(case)
    with open(os.path.join(GOLD, name + ".rows.json")) as fh:
        g = json.load(fh)
    m = g["m"]
    lens = np.asarray(g["lens"], dtype=np.int32)
    assert list(lens) == resolve_lens(case["lens"], case["n"], m)
    sig, _ = synth.synth_batch(case["seed"], case["first"], case["n"], m, lens)
    from golden_cases import apply_blips, apply_extra, apply_quantise

    apply_blips(sig, case); apply_extra(sig, lens, case); apply_quantise(sig, case)
    bad = []
    for i, w in enumerate(g["rows"]):
        have = min(int(lens[i]), m)
        if "_raise" in w:
            with pytest.raises(ValueError):
                oracle_mod.detect_llr_single(sig[i, :have], int(lens[i]), spc, m)
            continue
        got = oracle_mod.detect_llr_single(sig[i, :have], int(lens[i]), spc, m)
        d = row_diffs(got, w)
        if d:
            bad.append((i, d[:4]))
    assert not bad, bad[:5]

"""GPU parity of the remaining operators: start-peak primary, the single-read validator, error
behaviour of the batch API, the CLI end to end (CSV text equal to the reference's)."""
import os

import numpy as np
import pytest

from golden_cases import CASES
from util import GOLD, load_case, make_spc, row_diffs

pytestmark = pytest.mark.gpu

SP_CASES = [k for k, c in CASES.items() if c["primary"] == "start_peak"]


@pytest.mark.parametrize("name", SP_CASES)
def test_start_peak_rows_vs_golden(name):
    from adapted_amd.detect.combined import combined_detect_start_peak

    case, spc, sig, lens, want = load_case(name)
    got = combined_detect_start_peak(sig, lens, spc)
    bad = [(i, d) for i, (g, w) in enumerate(zip(got, want)) for d in row_diffs(g, w, float_rel=0.0)]
    assert not bad, bad[:10]


def test_llr_with_start_peak_columns_vs_oracle(oracle_mod):
    from adapted_amd.detect.combined import combined_detect_llr2

    case, spc, sig, lens, _ = load_case("rna004_llr_default")
    got = combined_detect_llr2(sig, lens, spc, with_start_peak=True)
    want = oracle_mod.detect_llr(sig, lens, spc, with_start_peak=True)
    bad = [(i, d) for i, (g, w) in enumerate(zip(got, want))
           for d in row_diffs(g, {k: v for k, v in w.items() if not k.startswith("_")}, float_rel=0.0)]
    assert not bad, bad[:10]
    assert any(g.start_peak_idx is not None for g in got)


def test_batch_errors_match_reference():
    from adapted_amd.detect.combined import combined_detect_llr2
    from util import make_spc

    spc = make_spc(CASES["rna004_llr_default"])
    m = spc.sig_preload_size
    flat = np.full((4, m), 80.0, dtype=np.float32)
    with pytest.raises(ValueError, match="MAD normalization failed: scale is 0"):
        combined_detect_llr2(flat, np.full(4, m, dtype=np.int32), spc)
    # a read shorter than min_obs_adapter + one pool block sinks the reference's minibatch
    from adapted_amd import synth

    sig, lens = synth.synth_batch(1, 0, 4, m, np.array([m, 900, m, m], dtype=np.int32))
    with pytest.raises(ValueError, match="argmin of an empty sequence"):
        combined_detect_llr2(sig, lens, spc)


def test_single_read_validator_equals_batch_rows():
    from adapted_amd.container_types import Boundaries
    from adapted_amd.detect.combined import combined_detect_llr2, validate_boundaries

    case, spc, sig, lens, want = load_case("rna004_llr_default")
    for i in (0, 1, 7, 9, 13):
        w = want[i]
        topk = np.array(w["polya_candidates"]) if w["polya_candidates"] is not None else None
        b = Boundaries(adapter_start=0, adapter_end=w["llr_adapter_end"], polya_end=w["llr_polya_end"], polya_end_topk=topk)
        got = validate_boundaries(sig[i][: lens[i]], b, spc, int(lens[i]))
        assert not row_diffs(got, w, float_rel=0.0), (i, row_diffs(got, w))


@pytest.mark.parametrize("name", ["rna002_llr_4k", "rna004_llr_default", "rna004_llr_mvs_overwrite_wide", "rna004_llr_open_pores", "rna004_cnn_default", "rna004_cnn_200k",
                                  "rna004_start_peak_blips", "rna004_llr_quantised", "rna004_llr_nan_holes"])
def test_cli_detect_writes_reference_csv(tmp_path, name):
    from adapted_amd import main as cli

    case, spc, sig, lens, want = load_case(name)
    ids = np.array(["read_%04d" % i for i in range(case["n"])], dtype=object)
    bundle = tmp_path / "reads_0.npz"
    np.savez(bundle, signals=sig, full_lengths=lens, read_ids=ids)
    cfg = tmp_path / "cfg.toml"
    spc.to_toml(str(cfg))
    out = tmp_path / "out"
    cli.main(["detect", "-i", str(bundle), "-o", str(out), "--config", str(cfg), "-s", str(case["minibatch"]), "-b", "4000"])
    run = [d for d in os.listdir(out) if d.startswith("adapted_0_2_4_")]
    assert len(run) == 1
    rd = out / run[0]
    for f in ("command.json", "config.toml", "adapted.log"):
        assert (rd / f).exists()
    with open(os.path.join(GOLD, name + ".pass.csv")) as fh:
        assert (rd / "boundaries" / "detected_boundaries_0.csv").read_text() == fh.read()
    with open(os.path.join(GOLD, name + ".fail.csv")) as fh:
        want_fail = fh.read()
    if want_fail.strip():
        assert (rd / "failed_reads" / "failed_reads_0.csv").read_text() == want_fail
    else:  # no failed read: no file (the golden one is the empty frame the generator wrote)
        assert not (rd / "failed_reads" / "failed_reads_0.csv").exists()
    # `continue` finds every read already processed and writes nothing new
    cli.main(["continue", str(rd)])
    assert sorted(os.listdir(rd / "boundaries")) == ["detected_boundaries_0.csv"]


def test_host_pipeline_equals_direct_calls_and_reports_dropped_minibatches():
    """adapted_amd.pipeline.HostPipeline (pinned staging slots, H2D overlapped with detect, writer thread): the rows of
    every minibatch equal a direct engine call on the same reads; a minibatch whose MAD is zero is reported through
    on_dropped and skipped (the reference drops it and logs), the stream goes on; the last, partial minibatch is served."""
    from adapted_amd import lib, synth
    from adapted_amd.pipeline import HostPipeline

    case, spc, sig, lens, want = load_case("rna004_llr_default")
    m = spc.sig_preload_size
    N = 16
    batches = []
    for k, n in enumerate([N, N, N, 5]):               # 3 full minibatches and a partial one
        ln = np.full(n, m, dtype=np.int32)
        s, _ = synth.synth_batch(11 + k, 0, n, m, ln)
        batches.append((s, ln, np.array(["b%d_%d" % (k, i) for i in range(n)], dtype=object)))
    batches[1][0][:, :] = np.float32(80.0)              # constant minibatch: MAD == 0 -> dropped

    eng = lib.Engine(spc, N, m, device=0)
    direct = {}
    for k, (s, ln, ids) in enumerate(batches):
        rows, mbs = eng.detect_llr_rows(s, ln, s.shape[0], s.shape[0], with_start_peak=True)
        direct[k] = (rows, int(mbs[0]))
    eng.close()
    assert direct[1][1] == lib.MB_MAD_ZERO and all(direct[k][1] == lib.MB_OK for k in (0, 2, 3))

    pipe = HostPipeline(spc, N, m, device=0, primary="llr", with_start_peak=True, n_slots=2)
    got, dropped = {}, []

    def fill(get_buffers):
        for k, (s, ln, ids) in enumerate(batches):
            bs, bl = get_buffers()
            bs[: s.shape[0]] = s
            bl[: s.shape[0]] = ln
            yield s.shape[0], (k, ids)

    total = pipe.run(fill, lambda tag, rows: got.__setitem__(tag[0], (tag[1], rows.copy())),
                     lambda tag, status: dropped.append((tag[0], status)))
    pipe.close()
    assert dropped == [(1, lib.MB_MAD_ZERO)]
    assert sorted(got) == [0, 2, 3] and total == N + N + 5
    for k in (0, 2, 3):
        assert list(got[k][0]) == list(batches[k][2])
        assert got[k][1].tobytes() == direct[k][0].tobytes()


def test_shared_reciprocal_division_equals_ieee_division():
    """k_norm_pool divides every sample of a minibatch by the same MAD: fdiv_shared (q0 = a*y, r = a - d*q0, q = q0 + r*y
    with y = RN(1/d)) must return the bits of the IEEE division.  Checked for 240 divisors (MAD-like and random) over
    16.7 M consecutive numerators each around the clip range (both signs): 8e9 quotients."""
    from adapted_amd import lib

    case, spc, sig, lens, want = load_case("rna004_llr_default")
    eng = lib.Engine(spc, 8, spc.sig_preload_size, device=0)
    rng = np.random.default_rng(5)
    ds = np.concatenate([rng.uniform(5.0, 20.0, 160), 10.0 ** rng.uniform(-3, 4, 60), [1.0, 2.0, 3.0, 7.0, 9.999999, 10.0, 0.1, 1.5,
                        float(np.float32(13.37)), 8.0, 12.5, 1e-2, 255.0, 1023.0, 0.3, 6.0, 11.0, 17.0, 19.0, 100.0]]).astype(np.float32)
    bad = 0
    for d in ds:
        # numerators from d * 2^-8 upwards: 2^24 consecutive floats cover two binades around |a| ~ d/256 .. ; plus a block
        # near 5 d (the clip bound) and a block of tiny values
        for start in (np.float32(d) * np.float32(2.0 ** -8), np.float32(d) * np.float32(4.0), np.float32(1e-6)):
            bad += eng.debug_divcheck(float(d), int(np.float32(start).view(np.uint32)), 1 << 24)
    assert bad == 0
    eng.close()


def test_int16_ingestion_equals_float32_path(tmp_path):
    """adp_calibrate_i16 (raw ADC int16 + per-read scale/offset -> float32 pA on the device, NaN beyond the read) returns
    the bits of numpy's scale * (adc.astype(float32) + offset), and the int16 pipeline's rows equal the float32 path's
    on the same reads -- through the CLI too (--int16_ingest)."""
    from adapted_amd import lib, main as cli
    from adapted_amd.pipeline import HostPipeline

    case, spc, sig, lens, want = load_case("rna004_llr_default")
    m = spc.sig_preload_size
    n = case["n"]
    rng = np.random.default_rng(8)
    scale = rng.uniform(0.14, 0.2, n).astype(np.float32)
    offset = rng.uniform(-20.0, 20.0, n).astype(np.float32)
    # raw samples whose calibrated values resemble the golden case's signal; lengths as in the case (some reads are short)
    raw = np.zeros((n, m), dtype=np.int16)
    for i in range(n):
        L = min(int(lens[i]), m)
        raw[i, :L] = np.clip(np.rint(np.nan_to_num(sig[i, :L]) / scale[i] - offset[i]), -32768, 32767).astype(np.int16)
    pa = np.full((n, m), np.nan, dtype=np.float32)
    for i in range(n):
        L = min(int(lens[i]), m)
        pa[i, :L] = scale[i] * (raw[i, :L].astype(np.float32) + offset[i])

    eng = lib.Engine(spc, n, m, device=0)
    d_raw, d_len, d_cal, d_out = eng.dev_alloc(raw.nbytes), eng.dev_alloc(n * 4), eng.dev_alloc(2 * n * 4), eng.dev_alloc(n * m * 4)
    eng.h2d(d_raw, raw); eng.h2d(d_len, lens.astype(np.int32)); eng.h2d(d_cal, np.stack([scale, offset]))
    eng.calibrate_i16(d_raw, d_len, d_cal, d_cal + n * 4, n, d_out)
    got = np.zeros((n, m), dtype=np.float32)
    eng.d2h(got, d_out)
    assert got.view(np.uint32)[~np.isnan(pa)].tobytes() == pa.view(np.uint32)[~np.isnan(pa)].tobytes()
    assert (np.isnan(got) == np.isnan(pa)).all()
    rows_f32, _ = eng.detect_llr_rows(pa, lens.astype(np.int32), n, case["minibatch"], with_start_peak=False)
    for p in (d_raw, d_len, d_cal, d_out):
        eng.dev_free(p)
    eng.close()

    mb = case["minibatch"]
    pipe = HostPipeline(spc, mb, m, device=0, primary="llr", int16_input=True)
    out = {}

    def fill(get_buffers):
        for k in range(0, n, mb):
            braw, bl, bsc, bof = get_buffers()
            kk = min(mb, n - k)
            braw[:kk], bl[:kk], bsc[:kk], bof[:kk] = raw[k:k + kk], lens[k:k + kk], scale[k:k + kk], offset[k:k + kk]
            yield kk, k

    pipe.run(fill, lambda k, rows: out.__setitem__(k, rows.copy()))
    pipe.close()
    rows_i16 = np.concatenate([out[k] for k in sorted(out)])
    assert rows_i16.tobytes() == rows_f32.tobytes()

    # the CLI: a float32 bundle and the raw bundle of the same reads give the same CSV files
    ids = np.array(["read_%04d" % i for i in range(n)], dtype=object)
    np.savez(tmp_path / "f32_0.npz", signals=pa, full_lengths=lens, read_ids=ids)
    np.savez(tmp_path / "raw_0.npz", raw=raw, scale=scale, offset=offset, full_lengths=lens, read_ids=ids)
    cfg = tmp_path / "cfg.toml"
    spc.to_toml(str(cfg))
    texts = []
    for name, extra in (("f32_0.npz", []), ("raw_0.npz", ["--int16_ingest"])):
        o = tmp_path / ("out_" + name)
        cli.main(["detect", "-i", str(tmp_path / name), "-o", str(o), "--config", str(cfg), "-s", str(mb), "-b", "4000"] + extra)
        rd = o / [d for d in os.listdir(o) if d.startswith("adapted_")][0]
        texts.append(((rd / "boundaries" / "detected_boundaries_0.csv").read_text(), (rd / "failed_reads" / "failed_reads_0.csv").read_text()))
    assert texts[0] == texts[1]


@pytest.mark.parametrize("windows", [dict(), dict(pA_var_window=600, search_window=1200), dict(pA_mean_window=150, pA_var_window=40, search_window=300),
                                     dict(pA_var_window=320, pA_mean_window=321, search_window=700, polyA_window=2500)])
def test_mvs_detect_overwrite_vs_oracle(oracle_mod, windows):
    """mvs_detect_overwrite = true (reference adapted/detect/mvs.py:181-338, combined.py:517-562) on more reads than the
    goldens hold, with windows on both sides of the LDS-staged series path (MV_HIST = 320): the LLR path end to end and the
    candidate validator with a k-column table (the CNN path's entry), every field against the oracle."""
    from adapted_amd import lib, synth
    from adapted_amd.config import get_chemistry_specific_config

    spc = get_chemistry_specific_config("RNA004")
    spc.llr_boundaries.llr_detect, spc.cnn_boundaries.cnn_detect = True, False
    spc.mvs_polya.mvs_detect_overwrite = True
    spc.med_shift.detect_med_shift = True
    for k, v in windows.items():
        setattr(spc.mvs_polya, k, v)
    spc.update_primary_method()
    spc.update_sig_preload_size()
    m, n, mb = spc.sig_preload_size, 384, 128
    rng = np.random.default_rng(5)
    lens = np.where(rng.random(n) < 0.7, m, rng.integers(1500, 3 * m, n)).astype(np.int32)
    sig, lens = synth.synth_batch(31, 0, n, m, lens)
    eng = lib.Engine(spc, n, m, device=0)
    rows, mbs = eng.detect_llr_rows(sig, lens, n, mb)
    assert (mbs == 0).all()
    got = lib.rows_to_results(rows, "llr")
    want = []
    for s in range(0, n, mb):
        want += oracle_mod.detect_llr(sig[s:s + mb], lens[s:s + mb], spc)
    pub = lambda w: {k: v for k, v in w.items() if not k.startswith("_")}  # noqa: E731
    bad = [(i, d) for i, (g, w) in enumerate(zip(got, want)) for d in [row_diffs(g, pub(w))] if d]
    assert not bad, bad[:5]
    moved = sum(1 for g in got if g.success and g.mvs_adapter_end and g.adapter_end == g.mvs_adapter_end)
    assert moved > n // 4
    # candidate tables as the CNN path hands them over: adapter end + 3 poly(A) candidates (0-terminated lists)
    ae = np.array([g.llr_adapter_end or 0 for g in got], dtype=np.int64)
    pe = np.array([g.llr_polya_end or 0 for g in got], dtype=np.int64)
    bounds = np.stack([ae, pe, np.where(pe > 0, pe + 37, 0), np.zeros(n, dtype=np.int64)], axis=1)
    bounds[::7, 1] = np.where(ae[::7] > 0, ae[::7] + 60, 0)  # tails shorter than the MVS lag: polya_end turns None
    spc.cnn_boundaries.cnn_detect, spc.llr_boundaries.llr_detect = True, False
    spc.cnn_boundaries.fallback_to_llr_short_reads = False  # (the validator alone)
    spc.update_primary_method()
    eng2 = lib.Engine(spc, n, m, device=0)
    got2 = lib.rows_to_results(eng2.validate_rows(sig, lens, n, bounds), "cnn")
    want2 = oracle_mod.detect_cnn_from_preds(sig, lens, bounds, spc)
    bad = [(i, d) for i, (g, w) in enumerate(zip(got2, want2)) for d in [row_diffs(g, pub(w))] if d]
    assert not bad, bad[:5]
    if not windows:
        assert any(g.success and g.polya_end is None and g.mvs_llr_polya_end_to_early_stop for g in got2)
    eng.close()
    eng2.close()


def test_mvs_detect_overwrite_rejects_a_search_window_inside_the_lag():
    from adapted_amd import lib
    from adapted_amd.config import get_chemistry_specific_config

    spc = get_chemistry_specific_config("RNA004")
    spc.mvs_polya.mvs_detect_overwrite = True
    spc.mvs_polya.search_window = 100  # == pA_var_window: the reference indexes its series out of bounds (mvs.py:264)
    spc.update_primary_method()
    spc.update_sig_preload_size()
    with pytest.raises(lib.HipLibraryError, match="search_window"):
        lib.Engine(spc, 8, spc.sig_preload_size, device=0)


@pytest.mark.parametrize("int16", [False, True])
def test_ragged_pipeline_equals_padded_minibatches(tmp_path, int16):
    """Ragged ingestion (HostPipeline ragged=True, adp_expand_ragged): reads packed back to back cross PCIe, the NaN-padded
    [N, m] minibatch is laid out on the device -- rows identical to engine calls on the padded matrix the reference builds
    (adapted/file_proc.py:143-190), for float32 pA and for raw int16 with on-device calibration; through the file reader
    (yield_minibatches_packed) including a last, partial minibatch."""
    from adapted_amd import lib, synth
    from adapted_amd.io_utils import yield_minibatches_packed
    from adapted_amd.pipeline import HostPipeline

    case, spc, _, _, _ = load_case("rna004_llr_default")
    m = spc.sig_preload_size
    n, N = 40, 16
    rng = np.random.default_rng(3)
    lens = np.where(rng.random(n) < 0.4, m + 700, rng.integers(1100, m, n)).astype(np.int32)
    sig, lens = synth.synth_batch(23, 0, n, m + 800, lens)  # (reads longer than the window keep their extra samples in the file)
    ids = np.array(["r%03d" % i for i in range(n)], dtype=object)
    sc, of = np.float32(0.17), np.float32(-12.0)
    if int16:
        raw = np.clip(np.round(np.nan_to_num(sig) / sc - of), -32768, 32767).astype(np.int16)
        pa = (sc * (raw.astype(np.float32) + of)).astype(np.float32)
        np.savez(tmp_path / "reads_0.npz", raw=raw, scale=np.full(n, sc), offset=np.full(n, of), full_lengths=lens, read_ids=ids)
    else:
        pa = sig
        np.savez(tmp_path / "reads_0.npz", signals=sig, full_lengths=lens, read_ids=ids)
    # the padded minibatches of the reference layout, straight into the engine
    dense = np.full((n, m), np.nan, dtype=np.float32)
    for r in range(n):
        t = min(m, int(lens[r]))
        dense[r, :t] = pa[r, :t]
    eng = lib.Engine(spc, N, m, device=0)
    want = {}
    for s0 in range(0, n, N):
        k = min(N, n - s0)
        rows, mbs = eng.detect_llr_rows(dense[s0:s0 + k], lens[s0:s0 + k], k, k, with_start_peak=True)
        assert mbs[0] == 0
        want[s0 // N] = rows
    eng.close()
    pipe = HostPipeline(spc, N, m, device=0, primary="llr", with_start_peak=True, n_slots=2, int16_input=int16, ragged=True)
    got = []

    def fill(get_buffers):
        for k, idv in yield_minibatches_packed([str(tmp_path / "reads_0.npz")], set(), set(), N, m, get_buffers, int16=int16):
            yield k, idv.copy()

    total = pipe.run(fill, lambda idv, rows: got.append((list(idv), rows.copy())))
    pipe.close()
    assert total == n and len(got) == 3
    for q, (idv, rows) in enumerate(got):
        assert idv == list(ids[q * N:(q + 1) * N])
        assert rows.tobytes() == want[q].tobytes()


def test_hostile_inputs_equal_the_oracle(oracle_mod):
    """Infinities, magnitudes of 1e30 and 1e-30, negative signals, constant stretches, NaN holes inside reads (np.median and
    np.percentile return NaN then), duplicated and integer-valued samples: every field as the oracle has it
    (tests/abuse_vs_oracle.py holds the cases)."""
    import abuse_vs_oracle

    assert abuse_vs_oracle.main() == 0


@pytest.mark.parametrize("name", [k for k, c in CASES.items() if c["primary"] == "llr_single"])
def test_single_read_api_vs_golden_and_oracle(oracle_mod, name):
    """combined_detect_llr (adapted/detect/combined.py:39-119, API only): per-read normalisation, pooling from sample 0 with
    the longer head offset, min_obs_adapter added to the positions all the same, zeros behind the READ's end in its last
    pooled block -- every field as the reference run read by read (goldens) and as the oracle."""
    import json

    from adapted_amd import synth
    from adapted_amd.detect.combined import combined_detect_llr
    from golden_cases import resolve_lens
    from util import make_spc

    case = CASES[name]
    spc = make_spc(case)
    with open(os.path.join(GOLD, name + ".rows.json")) as fh:
        g = json.load(fh)
    m = g["m"]
    lens = np.asarray(g["lens"], dtype=np.int32)
    sig, _ = synth.synth_batch(case["seed"], case["first"], case["n"], m, lens)
    from golden_cases import apply_blips, apply_extra, apply_quantise

    apply_blips(sig, case); apply_extra(sig, lens, case); apply_quantise(sig, case)
    bad = []
    for i, w in enumerate(g["rows"]):
        have = min(int(lens[i]), m)
        if "_raise" in w:
            with pytest.raises((ValueError, TypeError)):
                combined_detect_llr(sig[i, :have], int(lens[i]), spc)
            continue
        got = combined_detect_llr(sig[i, :have], int(lens[i]), spc)
        d = row_diffs(got, w)
        if not d:
            d = row_diffs(got, {k: v for k, v in oracle_mod.detect_llr_single(sig[i, :have], int(lens[i]), spc, m).items() if not k.startswith("_")})
        if d:
            bad.append((i, d[:4]))
    assert not bad, bad[:5]
    # a read whose length is not a multiple of the pooling factor and a constant one (MAD == 0)
    odd = sig[0, : min(m, 9000) - 13].copy()
    assert not row_diffs(combined_detect_llr(odd, odd.size, spc),
                         {k: v for k, v in oracle_mod.detect_llr_single(odd, odd.size, spc, m).items() if not k.startswith("_")})
    with pytest.raises(ValueError, match="scale is 0"):
        combined_detect_llr(np.full(min(m, 9000), 80.0, dtype=np.float32), min(m, 9000), spc)


@pytest.mark.parametrize("primary", ["llr", "cnn"])
def test_cli_two_ranks_equal_one_rank(tmp_path, primary):
    """`adapted detect` under torchrun with 2 ranks (gloo rendezvous, both ranks on this box's one GPU): one run directory
    (rank 0's name, broadcast), every rank decodes only its own groups of whole minibatches, rows come back in stream
    order -- the CSV files equal the 1-rank run's byte for byte.  Also a stream with fewer groups than ranks (a rank
    without any row must not break the gather)."""
    import subprocess
    import sys

    from adapted_amd import synth
    from adapted_amd.config import get_chemistry_specific_config

    spc = get_chemistry_specific_config("RNA004")
    spc.llr_boundaries.llr_detect = primary == "llr"
    spc.cnn_boundaries.cnn_detect = primary == "cnn"
    spc.update_primary_method()
    spc.update_sig_preload_size()
    m = spc.sig_preload_size
    mb, n = 8, 8 * 4 * 5 + 11                     # 5 groups of 4 minibatches and a ragged tail
    lens = np.array([m if i % 4 else synth.pareto_length(3, i) for i in range(n)], dtype=np.int32)
    sig, _ = synth.synth_batch(31, 0, n, m, lens)
    ids = np.array(["read_%04d" % i for i in range(n)], dtype=object)
    np.savez(tmp_path / "reads_0.npz", signals=sig[: n // 2], full_lengths=lens[: n // 2], read_ids=ids[: n // 2])
    np.savez(tmp_path / "reads_1.npz", signals=sig[n // 2:], full_lengths=lens[n // 2:], read_ids=ids[n // 2:])
    np.savez(tmp_path / "few.npz", signals=sig[:20], full_lengths=lens[:20], read_ids=ids[:20])
    cfg = tmp_path / "cfg.toml"
    spc.to_toml(str(cfg))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(inputs, out, ranks):
        cmd = [sys.executable]
        env = dict(os.environ, PYTHONPATH=root, ADAPTED_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
        if ranks > 1:
            cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ranks, "--master-addr", "127.0.0.1",
                    "--master-port", "29741"]
            cmd += ["-m", "adapted_amd.main"]
        else:
            cmd += ["-m", "adapted_amd.main"]
        cmd += ["detect", "-i"] + [str(x) for x in inputs] + ["-o", str(out), "--config", str(cfg), "-s", str(mb), "-b", "50"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        runs = [d for d in os.listdir(out) if d.startswith("adapted_")]
        assert len(runs) == 1, runs            # no stray directories from the other ranks
        files = {}
        for sub in ("boundaries", "failed_reads"):
            d = out / runs[0] / sub
            for f in sorted(os.listdir(d)) if d.exists() else []:
                files[sub + "/" + f] = (d / f).read_text()
        return files

    for inputs, tag in (([tmp_path / "reads_0.npz", tmp_path / "reads_1.npz"], "all"), ([tmp_path / "few.npz"], "few")):
        one = run(inputs, tmp_path / ("out1_" + tag), 1)
        two = run(inputs, tmp_path / ("out2_" + tag), 2)
        assert one.keys() == two.keys() and len(one) >= 1
        for k in one:
            assert one[k] == two[k], (tag, k)


@pytest.mark.parametrize("max_obs_trace,n,overrides", [(None, 96, {}), (200000, 48, {}), (60000, 64, {"mvs_polya.mvs_detect_overwrite": True}),
                                                        (None, 64, {"med_shift.detect_med_shift": True, "mvs_polya.pA_var_window": 101})])
def test_int16_native_rows_equal_the_calibrated_float32_path(max_obs_trace, n, overrides):
    """adp_detect_llr_i16: every kernel that touches the signal reads the RAW int16 samples and forms pA = scale * (float32(adc) +
    offset) in registers; samples behind a read's end count as the NaN padding.  Rows byte-identical to adp_calibrate_i16 +
    adp_detect_llr on the same reads (per-read calibrations, lengths from 1012 samples to beyond the window, open-pore blips
    above 16 entries, two minibatches), with the start-peak columns."""
    from adapted_amd import lib, synth

    spc = make_spc(dict(chem="RNA004", primary="llr", max_obs_trace=max_obs_trace, override=overrides))
    m = spc.sig_preload_size
    rng = np.random.default_rng(n)
    lens = np.array([m + 300 if i % 3 == 0 else max(1012, synth.pareto_length(9, i, lo=2000, hi=2 * m)) for i in range(n)], dtype=np.int32)
    lens[1], lens[2] = 1012, m
    sig, lens = synth.synth_batch(61, 0, n, m, lens)
    for j in range(24):
        sig[4, 120 + 40 * j: 123 + 40 * j] = 260.0  # more open pores than a row holds
    scale = rng.uniform(0.15, 0.21, n).astype(np.float32)
    offset = rng.integers(-30, 10, n).astype(np.float32)
    raw = np.clip(np.round(np.nan_to_num(sig) / scale[:, None] - offset[:, None]), -32768, 32767).astype(np.int16)
    raw[np.isnan(sig)] = rng.integers(-3000, 3000, int(np.isnan(sig).sum())).astype(np.int16)  # garbage behind the reads' ends
    eng = lib.Engine(spc, n, m, device=0)
    d_raw, d_len, d_cal, d_f32 = eng.dev_alloc(n * m * 2), eng.dev_alloc(n * 4), eng.dev_alloc(2 * n * 4), eng.dev_alloc(n * m * 4)
    eng.h2d(d_raw, raw)
    eng.h2d(d_len, lens)
    eng.h2d(d_cal, np.concatenate([scale, offset]))
    mb = n // 2
    eng.calibrate_i16(d_raw, d_len, d_cal, d_cal + n * 4, n, d_f32)
    want, mbs_w = eng.detect_llr_rows(d_f32, d_len, n, mb, with_start_peak=True, device_ptrs=True, tails_nan=True)
    got, mbs_g = eng.detect_llr_rows_i16(d_raw, d_len, d_cal, d_cal + n * 4, n, mb, with_start_peak=True)
    assert (mbs_w == 0).all() and (mbs_g == mbs_w).all()
    res_w, res_g = lib.rows_to_results(want, "llr"), lib.rows_to_results(got, "llr")
    plain = lambda d: {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in d.items()}
    bad = [(i, d) for i, (g, w) in enumerate(zip(res_g, res_w)) for d in row_diffs(g, plain(w.__dict__))]
    assert not bad, bad[:8]
    a, b = want.copy(), got.copy()
    a["open_pores_more"] = 0; b["open_pores_more"] = 0  # (registry tokens differ between the two calls)
    assert a.tobytes() == b.tobytes()
    assert sum(r.success for r in res_g) > n // 4
    for p in (d_raw, d_len, d_cal, d_f32):
        eng.dev_free(p)
    eng.close()

"""Generate the golden vectors under tests/golden/ by running the REAL reference
(/root/reference, read-only) in this container.  TEST INFRASTRUCTURE; committed so the
vectors can be regenerated, never shipped to or run on the GPU box.

    /opt/conda/bin/python3.9 oracle/gen_golden.py llr        # LLR + start-peak cases, primitives
    /usr/local/bin/python3   oracle/gen_golden.py cnn        # CNN cases (needs torch)
    /usr/local/bin/python3   oracle/gen_golden.py cnn rna004_cnn_200k   # just the named case(s)
    /usr/local/bin/python3   oracle/gen_golden.py preds      # cnn_detect over thousands of reads: positions + the scores there

Inputs are regenerated from seeds by adapted_amd/synth.py (host twin of the device
generator), so only the reference's OUTPUTS are stored.
"""
import importlib.util
import json
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_harness  # noqa: E402

import numpy as np  # noqa: E402
from golden_cases import (CASES, PREDS_CASES, apply_blips, apply_extra, apply_overrides, apply_quantise, preds_lens,  # noqa: E402
                          resolve_lens)

_spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "adapted_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

GOLD = os.path.join(ROOT, "tests", "golden")
CSV_CASES = ("rna002_llr_4k", "rna004_llr_default", "rna004_llr_mvs_overwrite_wide", "rna004_llr_open_pores", "rna004_start_peak_blips", "rna004_llr_quantised", "rna004_llr_nan_holes")  # cases whose CSV text is kept as well


def make_spc(case):
    from adapted.config.sig_proc import get_chemistry_specific_config

    spc = get_chemistry_specific_config(case["chem"])
    p = case["primary"]
    if p == "llr_single":
        p = "llr"
    spc.llr_boundaries.llr_detect = p == "llr"
    spc.cnn_boundaries.cnn_detect = p == "cnn"
    spc.rna_start_peak.detect_rna_start_peak = p == "start_peak"
    if case.get("max_obs_trace"):
        spc.core.max_obs_trace = case["max_obs_trace"]
    if "mvs_detect_check" in case:
        spc.mvs_polya.mvs_detect_check = case["mvs_detect_check"]
    if "detect_med_shift" in case:
        spc.med_shift.detect_med_shift = case["detect_med_shift"]
    apply_overrides(spc, case)
    spc.update_primary_method()
    spc.update_sig_preload_size()
    return spc


def jsonable(v):
    if v is None:
        return None
    if isinstance(v, (bool, np.bool_)):
        return bool(v)
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        return float(v)
    if isinstance(v, np.ndarray):
        return [jsonable(x) for x in v.ravel().tolist()]
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    return str(v)


def rows_of(results):
    rows = []
    for r in results:
        d = dict(r.__dict__)
        d.pop("llr_trace", None)
        rows.append({k: jsonable(v) for k, v in d.items()})
    return rows


def run_case(name, case):
    from adapted.detect import combined

    spc = make_spc(case)
    m = spc.sig_preload_size
    n = case["n"]
    lens = np.asarray(resolve_lens(case["lens"], n, m), dtype=np.int32)
    sig, lens = synth.synth_batch(case["seed"], case["first"], n, m, lens)
    apply_blips(sig, case)
    apply_extra(sig, lens, case)
    apply_quantise(sig, case)
    mb = case["minibatch"]
    results = []
    model = None
    if case["primary"] == "cnn":
        from adapted.detect.cnn import load_cnn_model

        model = load_cnn_model(spc.cnn_boundaries.model_name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for s in range(0, n, mb):
            b, l = sig[s:s + mb], lens[s:s + mb]
            if case["primary"] == "llr":
                res = combined.combined_detect_llr2(b, l, spc)
            elif case["primary"] == "start_peak":
                res = combined.combined_detect_start_peak(b, l, spc)
            else:
                res = combined.combined_detect_cnn(b.copy(), l, model, spc)
                if not isinstance(res, list):
                    res = [res]
            results.extend(res)
    out = dict(case=name, m=int(m), lens=[int(x) for x in lens], primary_method=spc.primary_method,
               rows=rows_of(results))
    with open(os.path.join(GOLD, name + ".rows.json"), "w") as fh:
        json.dump(out, fh, indent=0, allow_nan=True)
    dump_intermediates(name, case, spc, sig, lens, model)
    return results, spc, sig, lens


def run_single_case(name, case):
    """combined_detect_llr, read by read, each handed over without padding; exceptions are recorded as the reference raises them"""
    from adapted.detect import combined

    spc = make_spc(case)
    m = spc.sig_preload_size
    n = case["n"]
    lens = np.asarray(resolve_lens(case["lens"], n, m), dtype=np.int32)
    sig, lens = synth.synth_batch(case["seed"], case["first"], n, m, lens)
    apply_blips(sig, case)
    apply_extra(sig, lens, case)
    apply_quantise(sig, case)
    rows = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(n):
            have = min(int(lens[i]), m)
            try:
                r = combined.combined_detect_llr(sig[i, :have].copy(), int(lens[i]), spc)
                d = dict(r.__dict__)
                d.pop("llr_trace", None)
                rows.append({k: jsonable(v) for k, v in d.items()})
            except Exception as e:  # noqa: BLE001 -- recorded: the API lets them through
                rows.append({"_raise": type(e).__name__ + ": " + str(e)})
    out = dict(case=name, m=int(m), lens=[int(x) for x in lens], primary_method=spc.primary_method, rows=rows)
    with open(os.path.join(GOLD, name + ".rows.json"), "w") as fh:
        json.dump(out, fh, indent=0, allow_nan=True)
    return rows


def dump_intermediates(name, case, spc, sig, lens, model):
    """Stage-by-stage vectors for a few reads of the FIRST minibatch."""
    idx = case.get("dump") or []
    if not idx:
        return
    arrs = {}
    mb = case["minibatch"]
    b = sig[:mb]
    if case["primary"] == "llr":
        from adapted.detect.downscale import downscale_signal
        from adapted.detect.llr import (adapter_end_from_trace, calc_adapter_trace,
                                        detect_full_polya_trace_peak_with_spike,
                                        find_peaks_in_trace)
        from adapted.detect.normalize import med_mad, normalize_signal

        T = spc.core.max_obs_trace
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            med, mad = med_mad(b[:, :T], with_nan=True)
            norm = normalize_signal(b[:, :T], outlier_thresh=spc.core.sig_norm_outlier_thresh, with_nan=True)
            down = downscale_signal(norm[:, spc.core.min_obs_adapter:], spc.core.downscale_factor)
        arrs["n1_med_mad"] = np.array([med, mad], dtype=np.float64)
        arrs["dump_idx"] = np.array(idx, dtype=np.int64)
        n_nan = np.isnan(down).sum(axis=1)
        arrs["n_valid"] = (down.shape[1] - n_nan).astype(np.int64)
        for k in idx:
            s_ = down[k, : down.shape[1] - n_nan[k]]
            arrs["down_%d" % k] = s_.astype(np.float32)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                tr = calc_adapter_trace(s_, 5, 5, 1, 0, 0, 0, 0, True, 0, None, 0, 0, None, None)
                arrs["g1_%d" % k] = tr.signal.astype(np.float64)
                arrs["t1_%d" % k] = np.array([tr.start, tr.end], dtype=np.int64)
                w = spc.llr_boundaries.adapter_peak_width // spc.core.downscale_factor
                raw_peaks = find_peaks_in_trace(tr, w, spc.llr_boundaries.adapter_peak_prominence,
                                                spc.llr_boundaries.adapter_peak_rel_height)
                arrs["p1_%d" % k] = np.asarray(raw_peaks, dtype=np.int64)
                cands = adapter_end_from_trace(tr, spc.llr_boundaries.adapter_peak_prominence,
                                               spc.llr_boundaries.adapter_peak_rel_height, w, True, True)
                arrs["cands_%d" % k] = np.asarray(cands, dtype=np.int64)
                if len(cands):
                    tr2 = calc_adapter_trace(s_, 1, 1, 1, 0, 0, 0, 0, False, int(cands[0]), None, 0, 0, tr.c, tr.c2)
                    arrs["g2_%d" % k] = tr2.signal.astype(np.float64)
                    arrs["p4_%d" % k] = np.array([detect_full_polya_trace_peak_with_spike(tr2.signal)], dtype=np.int64)
    elif case["primary"] == "cnn":
        import torch
        from adapted.detect.cnn import cnn_detect, cnn_score, prepare_data

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x = prepare_data(b, spc.core)
            with torch.no_grad():
                sc = cnn_score(x, model).numpy()
            preds = cnn_detect(b, model, spc.cnn_boundaries, spc.core)
        arrs["dump_idx"] = np.array(idx, dtype=np.int64)
        arrs["preds"] = preds.astype(np.int64)
        for k in idx:
            arrs["prep_%d" % k] = x[k, 0].numpy().astype(np.float32)
            arrs["scores_%d" % k] = sc[k].astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, name + ".stages.npz"), **arrs)


def run_preds_case(name, case):
    """cnn_detect (reference adapted/detect/cnn.py:165-182) over whole minibatches, no validation: the predicted sample positions
    and the reference's own float32 scores at those positions -- the yardstick of the conv stacks' index-flip census
    (tests/test_gpu_cnn.py::test_conv_stack_flips_against_the_reference)."""
    import time

    import torch
    from adapted.detect.cnn import cnn_detect, cnn_score, load_cnn_model, prepare_data

    spc = make_spc(case)
    m, n, mb = spc.sig_preload_size, case["n"], case["minibatch"]
    off, ds = spc.core.min_obs_adapter, spc.core.downscale_factor
    lens = np.asarray(preds_lens(n, m), dtype=np.int32)
    model = load_cnn_model(spc.cnn_boundaries.model_name)
    k = int(spc.cnn_boundaries.polya_cand_k)
    preds = np.zeros((n, 1 + k), dtype=np.int32)
    at = np.full((n, 1 + k), np.nan, dtype=np.float32)
    t0 = time.time()
    for s0 in range(0, n, mb):
        sig, _ = synth.synth_batch(case["seed"], case["first"] + s0, mb, m, lens[s0:s0 + mb])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            p = cnn_detect(sig, model, spc.cnn_boundaries, spc.core)
            with torch.no_grad():
                sc = cnn_score(prepare_data(sig, spc.core), model).numpy()
        preds[s0:s0 + mb] = p
        idx = np.where(p == 0, 0, (p - off) // ds)
        rr = np.arange(mb)
        at[s0:s0 + mb, 0] = sc[rr, 0, idx[:, 0]]
        for j in range(1, 1 + k):
            at[s0:s0 + mb, j] = np.where(p[:, j] == 0, np.nan, sc[rr, 1, idx[:, j]])
        print(name, s0 + mb, "/", n, "%.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(GOLD, name + ".preds.npz"), preds=preds, score_at=at, lens=lens,
                        m=np.int64(m), seed=np.int64(case["seed"]), first=np.int64(case["first"]), minibatch=np.int64(mb))


def gen_start_peak_table():
    from adapted.detect.start_peak import detect_rna_start_peak

    case = CASES["rna004_start_peak"]
    spc = make_spc(case)
    m = spc.sig_preload_size
    n = case["n"]
    lens = np.asarray(resolve_lens(case["lens"], n, m), dtype=np.int32)
    sig, lens = synth.synth_batch(case["seed"], case["first"], n, m, lens)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        df = detect_rna_start_peak(sig, lens, spc)
    rec = {c: [jsonable(v) if not (isinstance(v, float) and np.isnan(v)) else None for v in df[c].tolist()]
           for c in df.columns}
    with open(os.path.join(GOLD, "rna004_start_peak.table.json"), "w") as fh:
        json.dump(rec, fh, indent=0)


def gen_csv(results, lens, tag):
    """CSV text as the reference writes it (reference adapted/output.py:26-51)."""
    from adapted.container_types import ReadResult
    from adapted.output import save_detected_boundaries

    rr = [ReadResult(read_id="read_%04d" % i, success=r.success, fail_reason=r.fail_reason, detect_results=r)
          for i, r in enumerate(results)]
    ok = [r for r in rr if r.success]
    bad = [r for r in rr if not r.success]
    save_detected_boundaries(ok, os.path.join(GOLD, tag + ".pass.csv"), save_fail_reasons=False)
    save_detected_boundaries(bad, os.path.join(GOLD, tag + ".fail.csv"), save_fail_reasons=True)


def gen_bottleneck():
    import bottleneck as bn

    if not hasattr(bn, "__version__"):
        raise RuntimeError("real bottleneck required")
    rng = np.random.default_rng(5)
    arrs = {}
    for t in range(6):
        n = int(rng.integers(150, 2600))
        a = rng.normal(108, 2.5, n).astype(np.float32)
        if t % 2:
            a[rng.integers(0, n, 4)] += np.float32(60)
        arrs["a_%d" % t] = a
        arrs["mean20_%d" % t] = bn.move_mean(a, window=20)
        arrs["var100_%d" % t] = bn.move_var(a, window=100)
    arrs["version"] = np.array([int(x) for x in bn.__version__.split(".")[:3]])
    np.savez_compressed(os.path.join(GOLD, "bn_move.npz"), **arrs)


def gen_bottleneck_nan():
    """move_mean / move_var of float32 series WITH NaN samples (single ones, runs, a run longer than the window, NaNs in
    the first window and at the end) -- the real library's counting of valid samples."""
    import bottleneck as bn

    if not hasattr(bn, "__version__"):
        raise RuntimeError("real bottleneck required")
    rng = np.random.default_rng(15)
    arrs = {}
    for t in range(8):
        n = int(rng.integers(260, 1800))
        a = rng.normal(108, 2.5, n).astype(np.float32)
        if t == 0:
            a[150] = np.nan
        elif t == 1:
            a[120:128] = np.nan
            a[700:703] = np.nan
        elif t == 2:
            a[130:250] = np.nan  # longer than both windows
        elif t == 3:
            a[3] = np.nan        # inside the first window
            a[200:204] = np.nan
        elif t == 4:
            a[n - 2:] = np.nan
            a[0] = np.nan
        elif t == 5:
            a[rng.integers(0, n, 12)] = np.nan
        elif t == 6:
            a[140:150] = np.nan
            a[rng.integers(0, n, 3)] += np.float32(60)
        else:
            a[:] = np.nan
        arrs["a_%d" % t] = a
        arrs["mean20_%d" % t] = bn.move_mean(a, window=20)
        arrs["var100_%d" % t] = bn.move_var(a, window=100)
        arrs["var5_%d" % t] = bn.move_var(a, window=5)
    arrs["version"] = np.array([int(x) for x in bn.__version__.split(".")[:3]])
    np.savez_compressed(os.path.join(GOLD, "bn_move_nan.npz"), **arrs)


def export_weights():
    """The reference's trained CNN parameters (a data asset, CC BY-NC 4.0, (c) W. K. van der Toorn) as a
    neutral .npz with the same state-dict keys."""
    import torch

    src = os.path.join(ref_harness.REF_ROOT, "adapted", "models", "rna004_130bps@v0.2.4.pth")
    sd = torch.load(src, weights_only=True, map_location="cpu")
    dst = os.path.join(ROOT, "adapted_amd", "models", "rna004_130bps@v0.2.4.npz")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    np.savez(dst, **{k: v.numpy() for k, v in sd.items()})


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "llr"
    only = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else None  # e.g. llr rna004_llr_mvs_overwrite: just these cases
    os.makedirs(GOLD, exist_ok=True)
    if what == "bottleneck":
        gen_bottleneck()
        gen_bottleneck_nan()
        return
    ref_harness.install(need_torch=(what in ("cnn", "preds")))
    if what == "preds":
        for name, case in PREDS_CASES.items():
            if only and name not in only:
                continue
            run_preds_case(name, case)
        return
    if what == "llr" and only:
        for name in sorted(only):
            if CASES[name]["primary"] == "llr_single":
                rows = run_single_case(name, CASES[name])
                print(name, "ok", sum(1 for r in rows if r.get("success")), "/", len(rows), "raised", sum(1 for r in rows if "_raise" in r))
                continue
            results, spc, sig, lens = run_case(name, CASES[name])
            print(name, "ok", sum(r.success for r in results), "/", len(results))
            if name in CSV_CASES:
                gen_csv(results, lens, name)
        return
    if what == "llr":
        gen_bottleneck()
        gen_bottleneck_nan()
        for name, case in CASES.items():
            if case["primary"] == "cnn":
                continue
            if case["primary"] == "llr_single":
                run_single_case(name, case)
                continue
            results, spc, sig, lens = run_case(name, case)
            print(name, "ok", sum(r.success for r in results), "/", len(results))
            if name in CSV_CASES:
                gen_csv(results, lens, name)
        gen_start_peak_table()
    elif what == "cnn":
        if not only:
            export_weights()
        for name, case in CASES.items():
            if case["primary"] != "cnn" or (only and name not in only):
                continue
            results, spc, sig, lens = run_case(name, case)
            print(name, "ok", sum(r.success for r in results), "/", len(results))
            gen_csv(results, lens, name)
    import numpy, scipy
    if only:
        return
    with open(os.path.join(GOLD, "PROVENANCE_%s.txt" % what), "w") as fh:
        fh.write("generated by oracle/gen_golden.py %s\npython %s\nnumpy %s scipy %s\nreference ADAPTed v0.2.4 at /root/reference\n"
                 % (what, sys.version.split()[0], numpy.__version__, scipy.__version__))


if __name__ == "__main__":
    main()

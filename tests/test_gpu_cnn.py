"""GPU parity of the CNN path (C1 prepare, C2 conv head -- hand-written, cross-checked against PyTorch-ROCm --, C3 predict,
V1 with k candidates, C4 short-read fallback) against the golden vectors of the real reference.

The conv stack sums the same float32 products in another order than the reference's CPU oneDNN, so C2 scores
are compared within 1e-4 absolute; C3 indices are arg-max / peak picks on those scores and are
required to be identical on this fixture (a flip would be reported read by read)."""
import numpy as np
import pytest

from util import load_case, load_stages, row_diffs

pytestmark = pytest.mark.gpu


# both windows the reference was run at: the preset's (m = 17 500) and configs[2]'s (m = 201 500, Lc = 20 050: 12 reads incl.
# ones short enough for the LLR fallback, one that raises) -- tests/golden/rna004_cnn_{default,200k}.*; and polya_cand_k = 3
# without the fallback, polya_cand_k = 1 (no find_peaks, no row compaction: cnn.py:160)
@pytest.fixture(scope="module", params=["rna004_cnn_default", "rna004_cnn_200k", "rna004_cnn_k3", "rna004_cnn_k1", "rna004_cnn_adapter_range", "rna004_cnn_no_mean_range", "rna004_cnn_200k_k3", "rna004_cnn_quantised", "rna004_cnn_flat", "rna004_cnn_nan_holes",
                                        "rna004_cnn_nan_polya", "rna004_cnn_nan_polya_overwrite"])
def setup(request):
    import torch

    from adapted_amd.detect import cnn
    from adapted_amd.detect.combined import get_engine

    case, spc, sig, lens, want = load_case(request.param)
    model = cnn.load_cnn_model(spc.cnn_boundaries.model_name, device=0)
    eng = get_engine(spc, sig.shape[0], sig.shape[1], 0)
    return dict(name=request.param, case=case, spc=spc, sig=sig, lens=lens, want=want, model=model, eng=eng, cnn=cnn, torch=torch)


def test_prepare_scores_preds(setup):
    s = setup
    st = load_stages(s["name"])
    cnn, spc = s["cnn"], s["spc"]
    x = cnn.prepare_data(s["sig"], spc.core, spc=spc, engine=s["eng"])
    xc = x.cpu().numpy()
    for k in st["dump_idx"]:
        k = int(k)
        assert np.array_equal(xc[k, 0], st["prep_%d" % k]), k      # C1 bit-exact
    for conv, engine in (("hip", s["eng"]), ("torch", None)):
        sc = cnn.cnn_score(x, s["model"], engine=engine).cpu().numpy()
        for k in st["dump_idx"]:
            k = int(k)
            assert sc[k].shape == st["scores_%d" % k].shape
            assert np.max(np.abs(sc[k] - st["scores_%d" % k])) < 1e-4, (conv, k)  # C2
        preds = cnn.cnn_detect(s["sig"], s["model"], spc.cnn_boundaries, spc.core, spc=spc, engine=s["eng"], conv=conv)
        flips = [int(i) for i in np.flatnonzero((preds != st["preds"]).any(axis=1)) if not _pure_tie(st, spc, preds, int(i))]
        assert not flips, (conv, "reads whose CNN indices differ from the CPU reference", flips, preds[flips], st["preds"][flips])


def _pure_tie(st, spc, preds, i):
    """The device's candidates of read i differ from the reference's only WITHIN a set of exactly equal score heights: find_peaks(distance
    = 5) ranks peaks by an unstable np.argsort (reference adapted/detect/cnn.py:140 -> scipy's _select_by_peak_distance), so which of
    several equally high peaks survive is undefined -- it depends on numpy's sort kernel (SIMD width) -- while everything that IS
    defined must agree: the adapter end, the number of candidates, the multiset of candidate heights (read from the REFERENCE's
    scores).  A constant stretch gives the stride-3 net exactly periodic scores: hundreds of equal maxima (rna004_cnn_flat)."""
    key = "scores_%d" % i
    if key not in st.files:
        return False
    sc = st[key][1]
    off, ds = spc.core.min_obs_adapter, spc.core.downscale_factor
    ref, dev = st["preds"][i], preds[i]
    if ref[0] != dev[0] or np.count_nonzero(ref[1:]) != np.count_nonzero(dev[1:]):
        return False
    h = lambda row: sorted(float(sc[(int(p) - off) // ds]) for p in row[1:] if p)
    same = h(ref) == h(dev)
    if same:
        print("read %d: a permutation among exactly tied peak heights (undefined order in the reference)" % i, ref, dev)
    return same


def test_cnn_rows_vs_golden(setup):
    from adapted_amd.detect.combined import combined_detect_cnn

    s = setup
    st = load_stages(s["name"])
    preds = s["cnn"].cnn_detect(s["sig"], s["model"], s["spc"].cnn_boundaries, s["spc"].core, spc=s["spc"], engine=s["eng"])
    tied = {int(i) for i in np.flatnonzero((preds != st["preds"]).any(axis=1)) if _pure_tie(st, s["spc"], preds, int(i))}
    for model in (s["model"], None):  # (None: the weights named in the config, loaded without PyTorch)
        got = combined_detect_cnn(s["sig"], s["lens"], model, s["spc"])
        bad = [(i, d) for i, (g, w) in enumerate(zip(got, s["want"])) if i not in tied for d in row_diffs(g, w, float_rel=1e-5)]
        assert not bad, bad[:10]
    got_t = combined_detect_cnn(s["sig"], s["lens"], s["model"], s["spc"], conv="torch")
    bad = [(i, d) for i, (g, w) in enumerate(zip(got_t, s["want"])) if i not in tied for d in row_diffs(g, w, float_rel=1e-5)]
    assert not bad, bad[:10]
    inexact = [(i, d) for i, (g, w) in enumerate(zip(got, s["want"])) if i not in tied for d in row_diffs(g, w, float_rel=0.0)]
    print("cnn rows: %d float fields differ in the last bits" % len(inexact), inexact[:5])


def test_cnn_single_read_returns_bare_result(setup):
    from adapted_amd.container_types import DetectResults
    from adapted_amd.detect.combined import combined_detect_cnn

    s = setup
    r = combined_detect_cnn(s["sig"][:1], s["lens"][:1], s["model"], s["spc"])
    assert isinstance(r, DetectResults)


def _select_by_distance_stable(peaks, priority, distance):
    """scipy's _select_by_peak_distance with a STABLE priority order (equal heights: the later index first), which is what
    the device implements; scipy itself takes the order from an unstable np.argsort."""
    keep = np.ones(peaks.size, dtype=bool)
    order = np.argsort(priority, kind="stable")
    for i in range(peaks.size - 1, -1, -1):
        j = order[i]
        if not keep[j]:
            continue
        k = j - 1
        while k >= 0 and peaks[j] - peaks[k] < distance:
            keep[k] = False
            k -= 1
        k = j + 1
        while k < peaks.size and peaks[k] - peaks[j] < distance:
            keep[k] = False
            k += 1
    return keep


def _ref_predict(scores, na, k, stable=True):
    """numpy / scipy formulation of cnn_predict (reference adapted/detect/cnn.py:101-160) on given scores [n, 2, Lo]"""
    from scipy.signal import find_peaks

    sc = scores.copy()
    n, _, Lo = sc.shape
    a = np.argmax(sc[:, 0, :na], axis=1)
    pos = np.arange(Lo)[None, :]
    sc[:, 1, :][pos < a[:, None]] = -5.0
    p = np.argmax(sc[:, 1, :], axis=1)
    if k <= 1:
        return np.column_stack((a, p))
    sc[:, 1, :][pos > p[:, None]] = -5.0
    flat = sc[:, 1, :].flatten()
    if stable:
        cand, _ = find_peaks(flat)  # every local maximum (plateaus: their midpoint)
        cand = cand[_select_by_distance_stable(cand, flat[cand], 5)]
    else:
        cand, _ = find_peaks(flat, distance=5)
    heights = flat[cand]
    read_idx = cand // Lo
    order = np.lexsort((-heights, read_idx))
    cand = cand[order]
    groups = np.split(np.mod(cand, Lo), np.where(np.diff(read_idx) != 0)[0] + 1)
    out = np.zeros((n, k), dtype=np.int64)
    for i, peaks in enumerate(groups):
        out[i, : len(peaks)] = peaks[:k]
    return np.column_stack((a[:, None], out))


def _dev_predict(eng, scores, spc, minibatch=None):
    import torch

    d = torch.from_numpy(np.ascontiguousarray(scores)).cuda()
    torch.cuda.synchronize()
    n, _, Lo = scores.shape
    b = eng.cnn_predict(d.data_ptr(), n, minibatch or n, Lo)
    off, ds = spc.core.min_obs_adapter, spc.core.downscale_factor
    return np.where(b == 0, 0, (b - off) // ds)


def _cnn_engine(n=8):
    import torch
    from adapted_amd import lib
    from golden_cases import CASES
    from util import make_spc

    torch.cuda.init()  # (torch's bundled HIP runtime has to claim the GPU before the library's does)
    spc = make_spc(CASES["rna004_cnn_default"])
    return lib.Engine(spc, n, spc.sig_preload_size, device=0), spc


@pytest.mark.gpu
@pytest.mark.parametrize("Lo,n", [(257, 40), (1650, 64), (20050, 12)])
def test_device_predict_equals_scipy_formulation(Lo, n):
    """adp_cnn_predict (k_cnn_argmax / k_cnn_topk / k_cnn_bounds) against the numpy / scipy formulation of cnn_predict on
    random scores, including reads without any peak (the row-misalignment quirk), reads whose stretch touches the row
    ends and reads with nothing above the mask level."""
    eng, spc = _cnn_engine(n)
    rng = np.random.default_rng(Lo)
    k = int(spc.cnn_boundaries.polya_cand_k)
    na = (spc.core.max_obs_adapter - spc.core.min_obs_adapter) // spc.core.downscale_factor
    scores = rng.normal(0.0, 2.0, (n, 2, Lo)).astype(np.float32)
    scores[1, 0, 0] = 60.0                  # adapter position 0: no masked prefix
    scores[5:9, 1, :] -= 8.0                # most scores below the mask level (-5)
    scores[9, 1, :] = -9.0 - rng.random(Lo).astype(np.float32)  # ALL below it (behind a masked prefix): a read without peaks
    scores[2, 1, Lo - 1] = 50.0             # arg-max at the row end
    scores[3, 0, min(na, Lo) - 1] = 70.0    # adapter at the end of its search range
    want = _ref_predict(scores, na, k, stable=False)
    assert (want == _ref_predict(scores, na, k, stable=True)).all()  # (no ties in random data: both orders agree)
    got = _dev_predict(eng, scores, spc)
    assert (got == want).all(), np.argwhere(got != want)[:5]
    # two minibatches in one call = two calls of the reference
    half = n // 2
    got2 = _dev_predict(eng, scores, spc, minibatch=half)
    want2 = np.concatenate([_ref_predict(scores[:half], na, k), _ref_predict(scores[half:], na, k)])
    assert (got2 == want2).all(), np.argwhere(got2 != want2)[:5]
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["plateau", "plateau_at_start", "tie", "tie_chain", "below_mask", "below_mask_rows", "boundary", "boundary_chain",
                                  "equals_mask", "all_masked", "last_row_end", "nan"])
def test_device_predict_settles_corner_cases(kind):
    """what round 1 sent to scipy on the host: plateaus, equal heights within the minimum distance, reads at or below the
    mask level with nothing masked in front (the masked run behind them can be a peak, possibly in a later read's row),
    stretches that meet across a row boundary -- all settled on the device, equal to scipy's formulation run with a
    stable priority order (and to scipy itself wherever it does not depend on the order of equal heights)."""
    eng, spc = _cnn_engine(8)
    k = int(spc.cnn_boundaries.polya_cand_k)
    n, Lo = 6, 300
    na = 120
    spc.core.max_obs_adapter = spc.core.min_obs_adapter + na * spc.core.downscale_factor
    eng.set_config(spc)
    rng = np.random.default_rng(3)
    scores = rng.normal(0.0, 2.0, (n, 2, Lo)).astype(np.float32)
    scores[:, 0, 20] = 50.0   # adapter position 20 ...
    scores[:, 1, 250] = 30.0  # ... poly(A) arg-max 250 for every read, unless changed below
    same_as_scipy = True
    if kind == "plateau":
        scores[2, 1, 100:103] = 9.0; scores[2, 1, 150:152] = 8.0
    elif kind == "plateau_at_start":
        scores[2, 1, 20:24] = 9.0; scores[2, 1, 24] = 0.0
    elif kind == "tie":
        scores[2, 1, 99:104] = [0.0, 9.0, 0.0, 9.0, 0.0]; same_as_scipy = False
    elif kind == "tie_chain":
        scores[2, 1, 99:110] = [0.0, 9.0, 0.0, 9.0, 0.0, 9.0, 0.0, 9.0, 0.0, 9.0, 0.0]; same_as_scipy = False
    elif kind == "below_mask":      # a read with everything below -5 and no masked prefix: the masked run behind it is a peak
        scores[2, 0, :] = 0.0; scores[2, 0, 0] = 50.0
        scores[2, 1, :] = -6.0 - rng.random(Lo).astype(np.float32)
        scores[3, 1, 20] = -7.0       # ... if the next stretch starts below the mask level too
    elif kind == "below_mask_rows":   # the same with fully masked reads in between: the peak lands in a later row
        scores[1, 0, :] = 0.0; scores[1, 0, 0] = 50.0
        scores[1, 1, :] = -6.0 - rng.random(Lo).astype(np.float32)
        scores[2:4, 1, :] = -9.0      # behind a masked prefix: arg-max 0, nothing unmasked
        scores[4, 1, 20] = -7.0
    elif kind == "boundary":
        scores[2, 1, Lo - 2] = 40.0; scores[3, 0, :] = 0.0; scores[3, 0, 1] = 50.0; scores[3, 1, 1] = 35.0; scores[3, 1, 2] = 3.0
    elif kind == "boundary_chain":
        for r in (1, 2, 3):
            scores[r, 1, Lo - 1] = 40.0 + r; scores[r + 1, 0, :] = 0.0; scores[r + 1, 0, 0] = 50.0; scores[r + 1, 1, 0] = 20.0; scores[r + 1, 1, 1] = 21.0 + r
    elif kind == "equals_mask":       # samples that equal -5.0 exactly next to masked ones
        scores[2, 0, :] = 0.0; scores[2, 0, 0] = 50.0
        scores[2, 1, :] = -5.0; scores[2, 1, 100] = -7.0
        scores[3, 1, 20] = -5.0; scores[3, 1, 21] = -6.0
    elif kind == "all_masked":
        scores[:, 1, :] = -9.0
    elif kind == "last_row_end":
        scores[n - 1, 1, Lo - 1] = 60.0; scores[0, 0, :] = 0.0; scores[0, 0, 0] = 50.0; scores[0, 1, 0] = 45.0
    elif kind == "nan":
        scores[2, 1, 200] = np.nan; scores[3, 0, 7] = np.nan
    want = _ref_predict(scores, na, k, stable=True)
    if same_as_scipy:
        assert (want == _ref_predict(scores, na, k, stable=False)).all()
    got = _dev_predict(eng, scores, spc)
    assert (got == want).all(), (kind, np.argwhere(got != want)[:5], got[got != want][:5], want[got != want][:5])
    eng.close()


@pytest.mark.gpu
def test_device_topk_behind_given_argmaxes():
    """adp_cnn_topk (the k > 1 part alone) on explicit positions"""
    import torch

    eng, spc = _cnn_engine(8)
    n, Lo, k = 5, 400, 10
    rng = np.random.default_rng(9)
    scores = rng.normal(0.0, 2.0, (n, 2, Lo)).astype(np.float32)
    apos = np.full(n, 30, dtype=np.int64)
    scores[:, 1, 300] = 25.0
    ppos = np.full(n, 300, dtype=np.int64)
    d = torch.from_numpy(scores).cuda(); da = torch.from_numpy(apos).cuda(); dp = torch.from_numpy(ppos).cuda()
    torch.cuda.synchronize()
    cand, cnt = eng.cnn_topk(d.data_ptr(), da.data_ptr(), dp.data_ptr(), n, Lo, k)
    sc = scores.copy(); sc[:, 0, :] = 0.0; sc[:, 0, 30] = 1.0
    want = _ref_predict(sc, Lo, k)[:, 1:]
    assert (cnt > 0).all() and (cand == want).all()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("conv", ["split", "split_unfolded", "split_layer0apart", "f32"])
@pytest.mark.parametrize("Lc,n", [(1650, 37), (20050, 9), (1651, 5), (1652, 5), (100, 3), (7, 2), (193 * 3, 4), (256 * 3 + 1, 4), (190 * 3, 3), (191 * 3 + 2, 3), (380 * 3 + 1, 3),
                                  (186 * 3, 3), (186 * 3 + 1, 3), (372 * 3 + 1, 3), (123 * 3, 3), (251 * 3, 3),
                                  (512 * 3, 3), (1024 * 3, 2)])  # (the last two: several tiles of the 256-position shape, NT = 4)
def test_hip_conv_stack_equals_torch(Lc, n, conv, monkeypatch):
    """C2: the hand-written conv stacks (adp_cnn_forward) against torch's float32 conv1d / conv_transpose1d on the same device
    with the shipped weights -- "split": the default, float16 matrix cores on split operands (cnn_conv_split.h), layer 3 folded into
    layer 2's kernel (round 5: tiles that advance by 64 NT - 2 positions -- lengths around multiples of 190 sit on their seams);
    and layer 0 made in layer 1's prologue (tiles that advance by 64 NT - 6 positions, their 64 NT input rows made from 192 NT + 4 samples:
    lengths around multiples of 186 / 122 / 250 sit on their seams, 7 and 100 are shorter than one subtile);
    "split_unfolded": the same stack with layer 3 as a kernel of its own (ADP_CNN_FOLD=0); "split_layer0apart": with layer 0 as a
    kernel of its own (ADP_CNN_FUSE_IN=0); "f32": the exact
    float32 matrix-core kernels (cnn_conv.h, ADP_CNN_CONV=f32).  The float32 stack and torch are float32 sums of the same 448
    products per output in different orders, the split stack carries 22 bits per operand: tolerance 2e-5 relative to the score
    scale for both (the golden test pins the scores themselves)."""
    import torch

    monkeypatch.setenv("ADP_CNN_CONV", conv.split("_")[0])
    monkeypatch.setenv("ADP_CNN_FOLD", "0" if conv.endswith("unfolded") else "1")
    monkeypatch.setenv("ADP_CNN_FUSE_IN", "0" if conv.endswith("layer0apart") else "1")

    from adapted_amd import lib
    from adapted_amd.detect import cnn
    from golden_cases import CASES
    from util import make_spc

    torch.cuda.init()
    spc = make_spc(CASES["rna004_cnn_default"])
    model = cnn.load_cnn_model(spc.cnn_boundaries.model_name, device=0)
    eng = lib.Engine(spc, 8, spc.sig_preload_size, device=0)
    eng.cnn_set_weights({k: v for k, v in model.state_dict().items()})
    rng = np.random.default_rng(Lc)
    x = rng.normal(0.0, 1.5, (n, 1, Lc)).astype(np.float32)
    x[0, 0, Lc // 2:] = -5.0  # a padded tail, as prepare_data makes it
    xt = torch.from_numpy(x).cuda()
    with torch.no_grad():
        want = model(xt)
    Lo = want.shape[2]
    got = torch.full((n, 2, Lo), float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    eng.cnn_forward(xt.data_ptr(), n, Lc, got.data_ptr())
    w, g = want.cpu().numpy(), got.cpu().numpy()
    assert not np.isnan(g).any()
    scale = max(1.0, float(np.abs(w).max()))
    err = float(np.abs(w - g).max())
    assert err <= 2e-5 * scale, (err, scale)
    # a second call on the same handle (activation buffers reused, padding still zero) gives the same bits
    got2 = torch.empty_like(got)
    eng.cnn_forward(xt.data_ptr(), n, Lc, got2.data_ptr())
    assert torch.equal(got, got2)
    eng.close()


def _conv_engine(monkeypatch, conv, spc, n, m):
    from adapted_amd import lib

    monkeypatch.setenv("ADP_CNN_CONV", conv)  # read by adp_create
    return lib.Engine(spc, n, m, device=0)


@pytest.mark.gpu
@pytest.mark.parametrize("window,n", [(None, 3000), (200000, 360)])
def test_split_conv_gives_the_float32_stack_rows(window, n, monkeypatch):
    """The default conv stack (split float16 operands) against the exact-float32 one on synthetic reads, through the whole CNN
    path: the scores differ in their last bits (reported), every index and every other field of every row must not.  A read whose
    rows differ would be reported with its candidates' score margins instead of loosening anything."""
    from adapted_amd import lib, synth
    from adapted_amd.config import get_chemistry_specific_config
    from adapted_amd.detect import cnn

    spc = get_chemistry_specific_config("RNA004")
    if window:
        spc.core.max_obs_trace = window
    spc.update_primary_method()
    spc.update_sig_preload_size()
    m = spc.sig_preload_size
    lens = np.array([m if i % 4 else synth.pareto_length(7, i) for i in range(n)], dtype=np.int32)
    out = {}
    for conv in ("split", "f32"):
        eng = _conv_engine(monkeypatch, conv, spc, n, m)
        dsig, dlen = eng.dev_alloc(n * m * 4), eng.dev_alloc(n * 4)
        eng.h2d(dlen, lens)
        eng.synth_fill(dsig, dlen, n, seed=11, first_read=0)
        cnn.ensure_weights(eng, None, spc)
        rows, bounds = eng.detect_cnn_rows(dsig, dlen, n, min(n, 1000), device_ptrs=True)
        out[conv] = (rows.tobytes(), np.array(bounds))
        eng.dev_free(dsig)
        eng.dev_free(dlen)
        eng.close()
    differing = np.flatnonzero((out["split"][1] != out["f32"][1]).any(axis=1))
    assert differing.size == 0, ("reads whose CNN indices differ between the conv stacks", differing[:20], out["split"][1][differing[:5]], out["f32"][1][differing[:5]])
    assert out["split"][0] == out["f32"][0]


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["huge", "inf", "huge_tail"])
def test_split_conv_leaves_the_float16_range_to_the_float32_kernels(what, monkeypatch):
    """An activation of 32768 or more (or a non-finite one) anywhere raises the call's flag and the call is repeated on the
    float32 kernels: the scores are then the float32 stack's, BIT FOR BIT (a split result would differ in its last bits)."""
    import torch

    from adapted_amd.detect import cnn
    from golden_cases import CASES
    from util import make_spc

    torch.cuda.init()
    spc = make_spc(CASES["rna004_cnn_default"])
    model = cnn.load_cnn_model(spc.cnn_boundaries.model_name, device=0)
    n, Lc = 6, 1650
    rng = np.random.default_rng(3)
    x = rng.normal(0.0, 1.5, (n, 1, Lc)).astype(np.float32)
    if what == "huge":
        x[2, 0, 700:710] = 1e7  # layer 0 alone leaves the range
    elif what == "inf":
        x[4, 0, 100] = np.inf
    else:
        x[5, 0, -3:] = 1e7  # in the last tile of the last read
    xt = torch.from_numpy(x).cuda()
    got = {}
    for conv in ("split", "f32"):
        eng = _conv_engine(monkeypatch, conv, spc, 8, spc.sig_preload_size)
        eng.cnn_set_weights({k: v for k, v in model.state_dict().items()})
        L1 = (Lc - 1) // 3 + 1
        sc = torch.full((n, 2, 3 * L1 - 2), float("nan"), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        eng.cnn_forward(xt.data_ptr(), n, Lc, sc.data_ptr())
        got[conv] = sc.cpu().numpy()
        if conv == "split":  # the handle is back on the split kernels for the next, ordinary call
            x2 = torch.from_numpy(rng.normal(0.0, 1.5, (n, 1, Lc)).astype(np.float32)).cuda()
            sc2 = torch.empty_like(sc)
            eng.cnn_forward(x2.data_ptr(), n, Lc, sc2.data_ptr())
            with torch.no_grad():
                ref2 = model(x2)
            assert float((ref2 - sc2).abs().max()) < 1e-4
            got["split_next"] = (sc2.cpu().numpy(), x2)
        else:
            sc2 = torch.empty_like(sc)
            eng.cnn_forward(got["split_next"][1].data_ptr(), n, Lc, sc2.data_ptr())
            # ... and there its scores are NOT the float32 stack's bits (so the comparison below does tell the stacks apart)
            assert not np.array_equal(sc2.cpu().numpy(), got["split_next"][0])
        eng.close()
    assert np.array_equal(got["split"], got["f32"], equal_nan=True)


@pytest.mark.gpu
def test_out_of_range_call_in_chunks_over_lanes_equals_the_float32_rows(monkeypatch):
    """The flag of the split conv stack belongs to the CALL: with the call cut into chunks over two lanes (ADP_CNN_GROUPS) a
    1e7-pA stretch in one read repeats the whole call on the float32 kernels -- rows and predictions are those of a handle that
    runs the float32 stack from the start."""
    from adapted_amd import synth
    from adapted_amd.config import get_chemistry_specific_config
    from adapted_amd.detect import cnn

    spc = get_chemistry_specific_config("RNA004")
    spc.update_primary_method()
    spc.update_sig_preload_size()
    m, n, mb = spc.sig_preload_size, 96, 16
    sig, lens = synth.synth_batch(21, 0, n, m, np.full(n, m, dtype=np.int32))
    sig[70, 9000:9040] = 1e7
    out = {}
    for conv, groups in (("split", "3"), ("f32", "1")):
        monkeypatch.setenv("ADP_CNN_GROUPS", groups)
        eng = _conv_engine(monkeypatch, conv, spc, n, m)
        cnn.ensure_weights(eng, None, spc)
        rows, bounds = eng.detect_cnn_rows(sig, lens, n, mb)
        out[conv] = (rows.tobytes(), np.array(bounds))
        eng.close()
    assert np.array_equal(out["split"][1], out["f32"][1])
    assert out["split"][0] == out["f32"][0]


@pytest.mark.gpu
def test_split_conv_keeps_large_in_range_activations(monkeypatch):
    """Activations are stored times 2^-4: an artefact of 300 normalised units (a 3000-pA excursion at a MAD of 10) drives the layers to
    tens of thousands and still runs on the split kernels -- scores within the conv-stack tolerance of torch's float32 result, and
    not the float32 stack's bits (the call was not repeated)."""
    import torch

    from adapted_amd.detect import cnn
    from golden_cases import CASES
    from util import make_spc

    torch.cuda.init()
    spc = make_spc(CASES["rna004_cnn_default"])
    model = cnn.load_cnn_model(spc.cnn_boundaries.model_name, device=0)
    n, Lc = 5, 1650
    rng = np.random.default_rng(9)
    x = rng.normal(0.0, 1.5, (n, 1, Lc)).astype(np.float32)
    x[1, 0, 500:530] = 300.0
    x[3, 0, 40:44] = -250.0
    xt = torch.from_numpy(x).cuda()
    with torch.no_grad():
        want = model(xt).cpu().numpy()
    got = {}
    for conv in ("split", "f32"):
        eng = _conv_engine(monkeypatch, conv, spc, 8, spc.sig_preload_size)
        eng.cnn_set_weights({k: v for k, v in model.state_dict().items()})
        sc = torch.empty((n, 2, want.shape[2]), dtype=torch.float32, device="cuda")
        eng.cnn_forward(xt.data_ptr(), n, Lc, sc.data_ptr())
        got[conv] = sc.cpu().numpy()
        eng.close()
    scale = float(np.abs(want).max())
    assert scale > 1000.0  # (the excursion does reach the last layer)
    for conv in got:
        assert float(np.abs(got[conv] - want).max()) <= 2e-5 * scale, conv
    assert not np.array_equal(got["split"], got["f32"])


# ---------------------------------------------------------------------------------------------------------------------------
# Index flips of the conv stacks against the REFERENCE at scale (round 4).  tests/golden/rna004_cnn_preds_*.preds.npz hold what the
# reference's cnn_detect (adapted/detect/cnn.py:165-182: torch CPU / oneDNN scores, cnn_predict) returns for 16 000 reads at the
# preset's window and 1 600 at the 200 k window (oracle/gen_golden.py preds), with its own scores at the predicted positions.
# Any two float32-accurate conv stacks order near-tied peaks differently on about one read in 10^4; what is asserted: the adapter
# end and the BEST poly(A) candidate (columns 0 and 1: adapter_end, polya_end) never move, the split-float16 stack (the default) does not flip more reads than the exact-float32 MFMA stack (+2), and every
# flipped read's candidates differ by less than 1e-5 of the read's score scale (device scores of the flipped reads).
@pytest.mark.parametrize("name", ["rna004_cnn_preds_default", "rna004_cnn_preds_200k"])
def test_conv_stack_flips_against_the_reference(name, tmp_path):
    import os

    from golden_cases import PREDS_CASES, preds_lens
    from util import GOLD, make_spc

    from adapted_amd import lib
    from adapted_amd.detect import cnn

    case = PREDS_CASES[name]
    spc = make_spc(case)
    z = np.load(os.path.join(GOLD, name + ".preds.npz"))
    ref, ref_at = z["preds"].astype(np.int64), z["score_at"]
    n, mb, m = case["n"], case["minibatch"], spc.sig_preload_size
    assert int(z["m"]) == m and ref.shape[0] == n
    lens = np.asarray(preds_lens(n, m), dtype=np.int32)
    assert np.array_equal(lens, z["lens"])
    off, ds = spc.core.min_obs_adapter, spc.core.downscale_factor
    report, flips_of = [], {}
    saved = os.environ.get("ADP_CNN_CONV")
    try:
        for stack in ("split", "f32"):
            os.environ["ADP_CNN_CONV"] = stack  # (read when the engine is made)
            eng = lib.Engine(spc, n, m, device=0)
            dsig, dlen = eng.dev_alloc(n * m * 4), eng.dev_alloc(n * 4)
            try:
                eng.h2d(dlen, lens)
                eng.synth_fill(dsig, dlen, n, seed=case["seed"], first_read=case["first"], decorate=True)
                wts = cnn.load_cnn_weights(spc.cnn_boundaries.model_name)
                cnn.ensure_weights(eng, wts, spc)
                _, got = eng.detect_cnn_rows(dsig, dlen, n, mb, device_ptrs=True)
                bad = np.flatnonzero((got != ref).any(axis=1))
                flips_of[stack] = bad
                report.append("%s: %s stack: %d of %d reads differ from the reference's cnn_detect" % (name, stack, bad.size, n))
                for i in bad:
                    i = int(i)
                    sig = np.zeros((1, m), dtype=np.float32)
                    eng.d2h(sig, dsig + i * m * 4)
                    x = cnn.prepare_data(sig, spc.core, spc=spc, engine=eng)
                    sc = cnn.cnn_score(x, wts, engine=eng).cpu().numpy()[0]
                    scale = float(np.nanmax(np.abs(sc)))
                    gaps = []
                    for j in np.flatnonzero(got[i] != ref[i]):
                        ch = 0 if j == 0 else 1
                        pa, pb = int(ref[i, j]), int(got[i, j])
                        sa = sc[ch, (pa - off) // ds] if pa else np.float32(-5.0)
                        sb = sc[ch, (pb - off) // ds] if pb else np.float32(-5.0)
                        gaps.append(abs(float(sa) - float(sb)))
                    report.append("   read %5d: reference %s\n               device    %s\n               columns %s, largest gap between the swapped candidates' device scores %.3g = %.2g of the score scale %.1f; "
                                  "reference scores there %s" % (i, ref[i].tolist(), got[i].tolist(), np.flatnonzero(got[i] != ref[i]).tolist(), max(gaps), max(gaps) / scale, scale,
                                                                   np.round(ref_at[i][got[i] != ref[i]], 6).tolist()))
                    assert got[i, 0] == ref[i, 0], (stack, i, "the adapter end moved", got[i], ref[i])
                    assert got[i, 1] == ref[i, 1], (stack, i, "the BEST poly(A) candidate (-> polya_end) moved", got[i], ref[i])
                    assert max(gaps) < 1e-5 * scale, (stack, i, gaps, scale)
            finally:
                eng.dev_free(dsig); eng.dev_free(dlen)
                eng.close()
    finally:
        os.environ.pop("ADP_CNN_CONV", None)
        if saved is not None:
            os.environ["ADP_CNN_CONV"] = saved
    text = "\n".join(report)
    print(text)
    # (the census goes to the test's own directory; ADP_FLIPS_REPORT_DIR=<dir> -- the profiling script -- keeps a copy for profiles/)
    for out in filter(None, (str(tmp_path), os.environ.get("ADP_FLIPS_REPORT_DIR"))):
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "conv_stack_flips_vs_reference_%s.txt" % name), "w") as fh:
            fh.write(text + "\n")
    assert flips_of["split"].size <= flips_of["f32"].size + 2, text

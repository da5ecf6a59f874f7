"""GPU parity of the CNN path (C1 prepare, C2 conv head in PyTorch-ROCm, C3 predict, V1 with k
candidates, C4 short-read fallback) against the golden vectors of the real reference.

The conv stack runs in a different library (MIOpen vs the reference's CPU oneDNN), so C2 scores
are compared within 1e-4 absolute; C3 indices are arg-max / peak picks on those scores and are
required to be identical on this fixture (a flip would be reported read by read)."""
import numpy as np
import pytest

from util import load_case, load_stages, row_diffs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import torch

    from adapted_amd.detect import cnn
    from adapted_amd.detect.combined import get_engine

    case, spc, sig, lens, want = load_case("rna004_cnn_default")
    model = cnn.load_cnn_model(spc.cnn_boundaries.model_name, device=0)
    eng = get_engine(spc, sig.shape[0], sig.shape[1], 0)
    return dict(case=case, spc=spc, sig=sig, lens=lens, want=want, model=model, eng=eng, cnn=cnn, torch=torch)


def test_prepare_scores_preds(setup):
    s = setup
    st = load_stages("rna004_cnn_default")
    cnn, spc = s["cnn"], s["spc"]
    x = cnn.prepare_data(s["sig"], spc.core, spc=spc, engine=s["eng"])
    xc = x.cpu().numpy()
    for k in st["dump_idx"]:
        k = int(k)
        assert np.array_equal(xc[k, 0], st["prep_%d" % k]), k      # C1 bit-exact
    sc = cnn.cnn_score(x, s["model"]).cpu().numpy()
    for k in st["dump_idx"]:
        k = int(k)
        assert sc[k].shape == st["scores_%d" % k].shape
        assert np.max(np.abs(sc[k] - st["scores_%d" % k])) < 1e-4, k  # C2
    preds = cnn.cnn_detect(s["sig"], s["model"], spc.cnn_boundaries, spc.core, spc=spc, engine=s["eng"])
    flips = np.flatnonzero((preds != st["preds"]).any(axis=1))
    assert flips.size == 0, ("reads whose CNN indices differ from the CPU reference", flips, preds[flips], st["preds"][flips])


def test_cnn_rows_vs_golden(setup):
    from adapted_amd.detect.combined import combined_detect_cnn

    s = setup
    got = combined_detect_cnn(s["sig"], s["lens"], s["model"], s["spc"])
    bad = [(i, d) for i, (g, w) in enumerate(zip(got, s["want"])) for d in row_diffs(g, w, float_rel=1e-5)]
    assert not bad, bad[:10]
    inexact = [(i, d) for i, (g, w) in enumerate(zip(got, s["want"])) for d in row_diffs(g, w, float_rel=0.0)]
    print("cnn rows: %d float fields differ in the last bits" % len(inexact), inexact[:5])


def test_cnn_single_read_returns_bare_result(setup):
    from adapted_amd.container_types import DetectResults
    from adapted_amd.detect.combined import combined_detect_cnn

    s = setup
    r = combined_detect_cnn(s["sig"][:1], s["lens"][:1], s["model"], s["spc"])
    assert isinstance(r, DetectResults)


def _host_topk(scores, apos, ppos, k):
    """the reference's formulation of C3 (adapted_amd.detect.cnn._topk_candidates) on explicit arg-max positions"""
    from adapted_amd.detect import cnn as cnn_mod

    n, _, Lo = scores.shape
    pos = np.arange(Lo)[None, :]
    ch1 = scores[:, 1, :].copy()
    ch1[(pos < apos[:, None]) | (pos > ppos[:, None])] = cnn_mod.SCORE_EXCL
    return cnn_mod._topk_candidates(ch1, k)


@pytest.mark.gpu
@pytest.mark.parametrize("Lo,n", [(257, 40), (1650, 64), (20050, 12)])
def test_device_topk_equals_scipy_formulation(Lo, n):
    """adp_cnn_topk (k_cnn_topk) against the host numpy/scipy formulation of cnn_predict's top-k on random scores,
    including reads without any peak (the row-misalignment quirk) and reads whose stretch touches the row ends."""
    import torch
    from adapted_amd import lib
    from util import make_spc
    from golden_cases import CASES

    torch.cuda.init()  # (torch's bundled HIP runtime has to claim the GPU before the library's does)
    spc = make_spc(CASES["rna004_cnn_default"])
    eng = lib.Engine(spc, 8, spc.sig_preload_size, device=0)
    rng = np.random.default_rng(Lo)
    k = 10
    scores = rng.normal(0.0, 2.0, (n, 2, Lo)).astype(np.float32)
    apos = rng.integers(1, Lo // 3, n).astype(np.int64)
    apos[1] = 0                             # no masked prefix
    scores[5:9, 1, :] -= 8.0                # most scores below the mask level (-5)
    scores[9, 1, :] = -9.0 - rng.random(Lo).astype(np.float32)  # ALL below it: the arg-max is a masked sample, nothing
    scores[2, 1, Lo - 1] = 50.0             # is unmasked, a read without peaks (rows shift up, cnn.py:150-158); arg-max at the row end
    pos = np.arange(Lo)[None, :]
    # polya_pos as cnn_predict takes it: the (first) arg-max of the scores masked before the adapter position
    ppos = np.argmax(np.where(pos < apos[:, None], np.float32(-5.0), scores[:, 1, :]), axis=1).astype(np.int64)
    assert ppos[9] < apos[9] and ppos[2] == Lo - 1
    dsc = torch.from_numpy(scores).cuda()
    da, dp = torch.from_numpy(apos).cuda(), torch.from_numpy(ppos).cuda()
    torch.cuda.synchronize()
    cand, cnt, flag = eng.cnn_topk(dsc.data_ptr(), da.data_ptr(), dp.data_ptr(), n, Lo, k)
    assert flag == 0
    got = np.zeros((n, k), dtype=np.int64)
    nz = np.flatnonzero(cnt > 0)
    got[: nz.size] = cand[nz]
    want = _host_topk(scores, apos, ppos, k)
    assert (got == want).all(), np.argwhere(got != want)[:5]
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["plateau", "tie", "below_mask", "boundary", "empty"])
def test_device_topk_reports_what_only_scipy_settles(kind):
    import torch
    from adapted_amd import lib
    from util import make_spc
    from golden_cases import CASES

    torch.cuda.init()
    spc = make_spc(CASES["rna004_cnn_default"])
    eng = lib.Engine(spc, 8, spc.sig_preload_size, device=0)
    n, Lo, k = 6, 300, 10
    rng = np.random.default_rng(3)
    scores = rng.normal(0.0, 2.0, (n, 2, Lo)).astype(np.float32)
    apos = np.full(n, 20, dtype=np.int64); ppos = np.full(n, 250, dtype=np.int64)
    scores[:, 1, 20] = 1.0
    scores[:, 1, 250] = 30.0  # the arg-max of every read
    if kind == "plateau":
        scores[2, 1, 100] = scores[2, 1, 101] = 9.0
    elif kind == "tie":
        scores[2, 1, 100] = 9.0; scores[2, 1, 101] = 0.0; scores[2, 1, 102] = 9.0; scores[2, 1, 99] = 0.0; scores[2, 1, 103] = 0.0
    elif kind == "below_mask":
        apos[2] = 0; scores[2, 1, :] = -6.0 - rng.random(Lo).astype(np.float32); ppos[2] = int(np.argmax(scores[2, 1, :]))
    elif kind == "boundary":
        scores[2, 1, Lo - 2] = 40.0; ppos[2] = Lo - 2; apos[3] = 1
    elif kind == "empty":
        scores[2, 1, 250] = -5.0; scores[2, 1, 20:250] = -7.0  # the maximum equals the mask level: a plateau with the mask
    dsc = torch.from_numpy(scores).cuda()
    da, dp = torch.from_numpy(apos).cuda(), torch.from_numpy(ppos).cuda()
    torch.cuda.synchronize()
    _, _, flag = eng.cnn_topk(dsc.data_ptr(), da.data_ptr(), dp.data_ptr(), n, Lo, k)
    assert flag != 0, kind
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("Lc,n", [(1650, 37), (20050, 9), (1651, 5), (1652, 5), (100, 3), (7, 2), (193 * 3, 4), (256 * 3 + 1, 4)])
def test_hip_conv_stack_equals_torch(Lc, n):
    """C2: the hand-written conv stack (adp_cnn_forward, float32 matrix cores) against torch's float32 conv1d /
    conv_transpose1d on the same device with the shipped weights.  Both are float32 sums of the same 448 products per
    output in different orders: tolerance 2e-5 relative to the score scale (the golden test pins the scores themselves)."""
    import torch

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

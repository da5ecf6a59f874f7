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

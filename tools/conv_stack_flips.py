#!/usr/bin/env python3
"""How often do the CNN head's INDICES depend on which float32-accurate conv stack made the scores?  (GPU box.)

Three stacks sum the same 448 products per output with different roundings: the library's split float16 MFMA stack (the
default), its exact-float32 MFMA stack (ADP_CNN_CONV=f32) and torch's float32 conv1d (MIOpen).  None of them is the
reference's CPU oneDNN order either.  cnn_predict takes arg-maxima and peak picks of the scores, so a near-tie between two
positions can fall either way: this tool runs the same synthetic reads through all three, counts the reads whose predictions
(adapter end + the k poly(A) candidates) differ pairwise, and prints for the first few what differs and by how much the scores
at the two positions differ.

usage: python tools/conv_stack_flips.py [n_reads] [max_obs_trace]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adapted_amd import lib, synth  # noqa: E402
from adapted_amd.config import get_chemistry_specific_config  # noqa: E402
from adapted_amd.detect import cnn  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    window = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    torch.cuda.init()
    spc = get_chemistry_specific_config("RNA004")
    if window:
        spc.core.max_obs_trace = window
    spc.update_primary_method()
    spc.update_sig_preload_size()
    m = spc.sig_preload_size
    ds, off = spc.core.downscale_factor, spc.core.min_obs_adapter
    Lc = (m - off + ds - 1) // ds
    L1 = (Lc - 1) // 3 + 1
    Lo = 3 * L1 - 2
    lens = np.array([m if i % 4 else synth.pareto_length(7, i) for i in range(n)], dtype=np.int32)
    model = cnn.load_cnn_model(spc.cnn_boundaries.model_name, device=0)
    preds, scores = {}, {}
    x_dev = None
    for conv in ("split", "f32"):
        os.environ["ADP_CNN_CONV"] = conv
        eng = lib.Engine(spc, n, m, device=0)
        del os.environ["ADP_CNN_CONV"]
        dsig, dlen = eng.dev_alloc(n * m * 4), eng.dev_alloc(n * 4)
        eng.h2d(dlen, lens)
        eng.synth_fill(dsig, dlen, n, seed=11, first_read=0)
        cnn.ensure_weights(eng, None, spc)
        _, b = eng.detect_cnn_rows(dsig, dlen, n, min(n, 1000), device_ptrs=True)
        preds[conv] = np.array(b)
        # the scores themselves (prepared input -> conv stack), for the margins
        x = torch.empty((n, 1, Lc), dtype=torch.float32, device="cuda")
        eng.cnn_prepare(dsig, n, x.data_ptr(), device_ptrs=True)
        sc = torch.empty((n, 2, Lo), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        eng.cnn_forward(x.data_ptr(), n, Lc, sc.data_ptr())
        scores[conv] = sc
        if conv == "f32":
            x_dev = x
            # torch's float32 conv on the same prepared input, predictions through the library's C3
            with torch.no_grad():
                st = torch.cat([model(x[s0:s0 + 64]) for s0 in range(0, n, 64)])
            scores["torch"] = st
            preds["torch"] = np.array(eng.cnn_predict(st.data_ptr(), n, min(n, 1000), Lo))
            preds["f32_from_scores"] = np.array(eng.cnn_predict(sc.data_ptr(), n, min(n, 1000), Lo))
        else:
            preds["split_from_scores"] = np.array(eng.cnn_predict(sc.data_ptr(), n, min(n, 1000), Lo))
        eng.dev_free(dsig)
        eng.dev_free(dlen)
        eng.close()
    assert np.array_equal(preds["f32"], preds["f32_from_scores"]) and np.array_equal(preds["split"], preds["split_from_scores"])
    print("%d reads, window %d (m = %d, Lo = %d), k = %d" % (n, spc.core.max_obs_trace, m, Lo, spc.cnn_boundaries.polya_cand_k))
    for a, b in (("split", "f32"), ("split", "torch"), ("f32", "torch")):
        d = np.flatnonzero((preds[a] != preds[b]).any(axis=1))
        first = int((preds[a][:, :2] != preds[b][:, :2]).any(axis=1).sum())
        err = float((scores[a] - scores[b]).abs().max())
        print("%-6s vs %-6s: max |score difference| %.3g; reads with any differing index %d of %d (%.4f %%), with a differing adapter end / best poly(A) candidate %d" % (
            a, b, err, d.size, n, 100.0 * d.size / n, first))
        for r in d[:4]:
            cols = np.flatnonzero(preds[a][r] != preds[b][r])
            c = int(cols[0])
            pa, pb = int(preds[a][r, c]), int(preds[b][r, c])
            ch = 0 if c == 0 else 1
            ia, ib = (pa - off) // ds, (pb - off) // ds
            sa = scores["f32"][r, ch]
            if 0 <= ia < Lo and 0 <= ib < Lo:
                print("    read %d column %d: %d vs %d; float32-stack scores there %.7g / %.7g (gap %.2g of a score scale %.3g)" % (
                    r, c, pa, pb, float(sa[ia]), float(sa[ib]), abs(float(sa[ia]) - float(sa[ib])), float(sa.abs().max())))
            else:
                print("    read %d column %d: %d vs %d" % (r, c, pa, pb))


if __name__ == "__main__":
    main()

"""Lengths of the slices the moving-window series are made for (adapter end .. largest poly(A) candidate) on the bench's synthetic reads
at the 200 k window: the series kernel ends with its longest chain."""
import sys
import numpy as np
sys.path.insert(0, ".")
import bench
from adapted_amd import lib, synth
from adapted_amd.detect import cnn

spc = bench.make_spc(200000, "cnn")
m = spc.sig_preload_size
n = 2000
sig, lens = synth.synth_batch(1, 0, n, m, np.full(n, m, dtype=np.int32))
eng = lib.Engine(spc, n, m, device=0)
cnn.ensure_weights(eng, None, spc)
rows, bounds = eng.detect_cnn_rows(sig, lens, n, 1000)
b = np.asarray(bounds).reshape(n, -1)
L = b[:, 1:].max(axis=1) - b[:, 0]
L = np.where(b[:, 1:].max(axis=1) > 0, L, 0)
print("reads", n, "k", b.shape[1] - 1)
for q in (0, 10, 25, 50, 75, 90, 95, 99, 99.9, 100):
    print("percentile %5.1f: %d" % (q, np.percentile(L, q)))
print("mean", L.mean(), "sum/max", L.sum() / L.max())
h, e = np.histogram(L, bins=[0, 1000, 5000, 20000, 50000, 100000, 150000, 190000, 250000])
for c, lo, hi in zip(h, e[:-1], e[1:]):
    print("%7d .. %7d: %d" % (lo, hi, c))

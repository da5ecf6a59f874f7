"""Where the waves of k_mvs_series_pipe's workgroup 0 (the longest chains) spend a launch: cycles between barriers (work) against
the whole loop, per role.  Needs a -DADP_PHASE_TIMING build:
    ADAPTED_HIP_LIB=$PWD/adapted_amd/lib/dbg/libadapted_hip_phase.so python tools/experiments/series_phase_shares.py"""
import sys
import numpy as np
sys.path.insert(0, ".")
import bench
from adapted_amd import lib, synth
from adapted_amd.detect import cnn

spc = bench.make_spc(200000, "cnn")
m = spc.sig_preload_size
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
sig, lens = synth.synth_batch(1, 0, n, m, np.full(n, m, dtype=np.int32))
eng = lib.Engine(spc, n, m, device=0)
cnn.ensure_weights(eng, None, spc)
for rep in range(2):
    rows, bounds = eng.detect_cnn_rows(sig, lens, n, 1000)
d = eng.debug_counters(64)
b = np.asarray(bounds).reshape(n, -1)
L = int((b[:, 1:].max(axis=1) - b[:, 0]).max())
names = ["loader", "mean", "var part 1", "var part 2", "storer var", "storer mean"]
print("longest slice %d steps = %d chunks" % (L, (L + 63) // 64))
for w in range(6):
    work, tot = int(d[48 + 2 * w]), int(d[49 + 2 * w])
    print("wave %d %-12s work %9d cycles = %.2f us per chunk, loop %9d cycles = %.2f ms at 2.4 GHz (%.2f us per chunk)"
          % (w, names[w], work, work / 2400.0 / ((L + 63) // 64), tot, tot / 2.4e6, tot / 2400.0 / ((L + 63) // 64)))
print("simd of waves 0..5:", [int((int(d[60]) >> (8 * w)) & 3) for w in range(6)], " cu:", [int((int(d[60]) >> (8 * w + 4)) & 15) for w in range(6)])

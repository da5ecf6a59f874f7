"""Phases of a step of k_cnn_conv64s (one workgroup per CU: the phases add up to the kernel), wave 0 of workgroup 0, -DADP_PHASE_TIMING build:
    ADAPTED_HIP_LIB=$PWD/adapted_amd/lib/dbg/libadapted_hip_phase.so python tools/experiments/conv_phase_shares.py [reads]"""
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
d0 = eng.debug_counters(80).astype(np.int64)
rows, bounds = eng.detect_cnn_rows(sig, lens, n, 1000)
d = eng.debug_counters(80).astype(np.int64) - d0
for name, sl in (("layers 0 + 1", (64, 65, 66, 67, 68)), ("layers 2 + 3", (70, 71, 72, 73, 74))):
    steps = max(1, int(d[sl[4]]))
    ph = [int(d[j]) / steps for j in sl[:4]]
    tot = sum(ph)
    print("%s: %d steps of workgroup 0, %.0f cycles per step (%.2f us at 2.0 GHz): rows made / tile awaited %.0f (%.0f %%), k-loop %.0f (%.0f %%), "
          "conversion into LDS %.0f (%.0f %%), %s %.0f (%.0f %%)"
          % (name, steps, tot, tot / 2000.0, ph[0], 100 * ph[0] / tot, ph[1], 100 * ph[1] / tot, ph[2], 100 * ph[2] / tot,
             "copy-out" if name.startswith("layers 0") else "layer 3 + scores out", ph[3], 100 * ph[3] / tot))
print("layers 2 + 3, the last phase apart: layer 3's GEMM + partial sums %.0f cycles per step, combination + score stores %.0f" % (int(d[75]) / max(1, int(d[74])), int(d[73]) / max(1, int(d[74]))))

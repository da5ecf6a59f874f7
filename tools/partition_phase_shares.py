"""Shares of k_partition_stats' phases (pass A, bucket search / MAD prediction, pass B, median, MAD selection) from the cycle
tallies of a -DADP_PHASE_TIMING build of the library:

    hipcc <flags of adapted_amd/build.py> -DADP_PHASE_TIMING -o /tmp/phase.so adapted_amd/csrc/adapted_hip.hip
    ADAPTED_HIP_LIB=/tmp/phase.so python tools/partition_phase_shares.py [pareto | default]
"""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from adapted_amd import lib
from bench import make_spc
spc = make_spc(16000 if "default" in sys.argv[1:] else 200000)  # (default: the preset's window, m = 17 500)
m = spc.sig_preload_size
R, mb = 8000, 500
eng = lib.Engine(spc, R, m, device=0)
sig = torch.empty((R, m), dtype=torch.float32, device="cuda")
ln = torch.full((R,), m, dtype=torch.int32, device="cuda")
if len(sys.argv) > 1 and sys.argv[1] == "pareto":  # heavy-tailed lengths (BASELINE configs[4])
    from adapted_amd import synth
    ln = torch.tensor([synth.pareto_length(1, i) for i in range(R)], dtype=torch.int32, device="cuda")
rows = torch.empty((R, lib.ROW_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
eng.synth_fill(sig.data_ptr(), ln.data_ptr(), R, seed=1, first_read=0, decorate=True)
eng.detect_llr_rows(sig.data_ptr(), ln.data_ptr(), R, mb, with_start_peak=True, device_ptrs=True, rows_dev=rows.data_ptr())
c0 = eng.debug_counters(48).astype(np.int64)
eng.detect_llr_rows(sig.data_ptr(), ln.data_ptr(), R, mb, with_start_peak=True, device_ptrs=True, rows_dev=rows.data_ptr())
c1 = eng.debug_counters(48).astype(np.int64)
d = c1 - c0
print("tallies", d[:8])
ph = np.concatenate([d[8:14], d[24:25]]).astype(float)
print("phase share: passA %.1f%%  find/predict %.1f%%  passB %.1f%%  median %.1f%%  mad-select %.1f%%  mad-fallback %.1f%%  small-segment path %.1f%%" % tuple(100 * ph / ph.sum()))
print("cycles per read (clock64 ticks of a workgroup's first thread, all three segments): %.0f; by phase:" % (ph.sum() / R), (ph / R).round(0))
for side, name in ((0, "plain sums"), (1, "pass A (histogram)"), (2, "pass B (collect)")):
    q = d[25 + 4 * side: 29 + 4 * side].astype(float)
    print("  np_sum %-20s cycles per read: whole chunks %.0f  ragged side effects %.0f  leaves + tree %.0f  epilogue %.0f" % ((name,) + tuple(q / R)))
print("  median phase: up to the bucket select %.0f cycles per read; lower-median passes %d of %d; mean bucket count %.1f" % (d[37] / R, d[38], R, d[39] / max(R, 1)))
print("  small-segment path, cycles per read: first sum %.0f  second sum %.0f  LDS copy %.0f  median select %.0f  MAD select %.0f" % tuple(d[40:45] / R))
ns = (d[25:29] + d[29:33] + d[33:37]).astype(float)
print("inside the summing passes of the large segments (both passes): whole chunks %.1f%%  ragged part's side effects %.1f%%  its leaves + tree %.1f%%  epilogue %.1f%%  (= %.1f%% of all phases)"
      % (tuple(100 * ns / ns.sum()) + (100 * ns.sum() / ph.sum(),)))
r = np.zeros(R, dtype=lib.ROW_DTYPE); eng.d2h(r, rows.data_ptr())
C = {name: i for i, name in enumerate(lib.COLS)}
col = r["col"]; ok = r["success"] == 1
for nm in ("adapter_len", "polya_len", "rna_preloaded_len"):
    v = col[ok, C[nm]]
    print(nm, "mean %.0f  median %.0f  p90 %.0f" % (v.mean(), np.median(v), np.percentile(v, 90)))

"""Where k_partition_rna_res spends a read's time (one persistent workgroup per CU: the phases of its first thread add up to the
kernel), from the ticks of a -DADP_PHASE_TIMING build of the library:

    hipcc <flags of adapted_amd/build.py> -DADP_PHASE_TIMING -o /tmp/phase.so adapted_amd/csrc/adapted_hip.hip
    ADAPTED_HIP_LIB=/tmp/phase.so python tools/res_phase_shares.py [int16]
"""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from adapted_amd import lib
from bench import make_spc
spc = make_spc(200000)
m = spc.sig_preload_size
R, mb = 16000, 1000
eng = lib.Engine(spc, R, m, device=0)
sig = torch.empty((R, m), dtype=torch.float32, device="cuda")
ln = torch.full((R,), m, dtype=torch.int32, device="cuda")
rows = torch.empty((R, lib.ROW_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
eng.synth_fill(sig.data_ptr(), ln.data_ptr(), R, seed=1, first_read=0, decorate=True)
run = lambda: eng.detect_llr_rows(sig.data_ptr(), ln.data_ptr(), R, mb, with_start_peak=True, device_ptrs=True, rows_dev=rows.data_ptr())
run()
c0 = eng.debug_counters(48).astype(np.int64)
run()
c1 = eng.debug_counters(48).astype(np.int64)
d = c1 - c0
print("passes done here", d[22], "left to k_partition_stats", d[23])
names = ["pivot wait, zero hist, leaf table", "A: register groups, a streamed one behind each", "A: the other streamed groups", "A: ragged chunk",
         "fold, bucket search, MAD prediction", "B: ragged chunk", "B: register groups + streamed", "B: LDS groups + streamed", "B: the other streamed groups",
         "next read: first part requested", "fold, record, second part requested"]
ph = d[32:43].astype(float) * 0.01 / max(d[22] + d[23], 1)  # us per read (256 workgroups, each thread 0)
for nme, v in zip(names, ph):
    print("  %-40s %6.2f us per read" % (nme, v))
print("  %-40s %6.2f us per read  (x %d reads / 256 CUs = %.2f ms)" % ("sum", ph.sum(), R, ph.sum() * R / 256 * 1e-3))

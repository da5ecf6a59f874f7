import sys, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from adapted_amd import lib
from bench import make_spc
spc = make_spc(200000)
m = spc.sig_preload_size
R, mb = 8000, 500
eng = lib.Engine(spc, R, m, device=0)
sig = torch.empty((R, m), dtype=torch.float32, device="cuda")
ln = torch.full((R,), m, dtype=torch.int32, device="cuda")
rows = torch.empty((R, lib.ROW_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
eng.synth_fill(sig.data_ptr(), ln.data_ptr(), R, seed=1, first_read=0, decorate=True)
c0 = eng.debug_counters(24).astype(np.int64)
eng.detect_llr_rows(sig.data_ptr(), ln.data_ptr(), R, mb, with_start_peak=True, device_ptrs=True, rows_dev=rows.data_ptr())
c1 = eng.debug_counters(24).astype(np.int64)
d = (c1 - c0)[16:24].astype(float)
print("reads with a second survivor: %d of %d; mean ordinal of the second survivor among kept maxima: %.1f; <4: %.0f%%  <8: %.0f%%  <16: %.0f%%  <32: %.0f%%" % (d[0], R, d[1] / max(d[0], 1), 100 * d[2] / max(d[0], 1), 100 * d[3] / max(d[0], 1), 100 * d[4] / max(d[0], 1), 100 * d[5] / max(d[0], 1)))

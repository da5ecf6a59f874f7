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
eng.detect_llr_rows(sig.data_ptr(), ln.data_ptr(), R, mb, with_start_peak=True, device_ptrs=True, rows_dev=rows.data_ptr())
c0 = eng.debug_counters(24).astype(np.int64)
eng.detect_llr_rows(sig.data_ptr(), ln.data_ptr(), R, mb, with_start_peak=True, device_ptrs=True, rows_dev=rows.data_ptr())
c1 = eng.debug_counters(24).astype(np.int64)
d = (c1 - c0)[16:24].astype(float)
ph = d[:5]
print("polya phase share: init/recount %.1f%%  masks %.1f%%  fixed point %.1f%%  compaction %.1f%%  prominence/width %.1f%%" % tuple(100 * ph / ph.sum()))
print("cycles per read: %.0f ; kept maxima per read %.0f ; step-4 rounds per read %.2f" % (ph.sum() / R, d[6] / R, d[7] / R))

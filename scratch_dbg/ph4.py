import sys, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from adapted_amd import lib
from bench import make_spc
spc = make_spc(200000)
m = spc.sig_preload_size
R, mb = 8000, 1000
eng = lib.Engine(spc, R, m, device=0)
sig = torch.empty((R, m), dtype=torch.float32, device="cuda")
ln = torch.full((R,), m, dtype=torch.int32, device="cuda")
rows = torch.empty((R, lib.ROW_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
eng.synth_fill(sig.data_ptr(), ln.data_ptr(), R, seed=1, first_read=0, decorate=True)
c0 = eng.debug_counters(24).astype(np.int64)
eng.detect_llr_rows(sig.data_ptr(), ln.data_ptr(), R, mb, with_start_peak=True, device_ptrs=True, rows_dev=rows.data_ptr())
c1 = eng.debug_counters(24).astype(np.int64)
d = c1 - c0
print("large segments", d[0], "max nmad", c1[14], "max ncollect", c1[15], "mean nmad %.0f mean ncollect %.0f" % (d[16] / d[0], d[17] / d[0]), "nmad>2048:", d[18], ">3072:", d[19])

"""Does the order of the big allocations decide which of the two speeds a process gets?  sys.argv[1] = sig_first | engine_first."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from adapted_amd import lib

order = sys.argv[1]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 96000
spc = bench.make_spc(200000, "llr")
m = spc.sig_preload_size
dev = torch.device("cuda", 0)
torch.cuda.init()
if order == "sig_first":
    sig = torch.empty((R, m), dtype=torch.float32, device=dev)
    eng = lib.Engine(spc, R, m, device=0)
else:
    eng = lib.Engine(spc, R, m, device=0)
    sig = torch.empty((R, m), dtype=torch.float32, device=dev)
lens = torch.full((R,), m, dtype=torch.int32, device=dev)
rows = torch.empty((R, lib.ROW_DTYPE.itemsize), dtype=torch.uint8, device=dev)
eng.synth_fill(sig.data_ptr(), lens.data_ptr(), R, seed=1, first_read=0, decorate=True)
torch.cuda.synchronize()
eng.set_profiling(True)
for it in range(3):
    eng.detect_llr_rows(sig.data_ptr(), lens.data_ptr(), R, 1000, with_start_peak=True, device_ptrs=True, rows_dev=rows.data_ptr(), tails_nan=True)
    kt = eng.kernel_times()
k = {}
for name, v in kt:
    k[name] = k.get(name, 0.0) + v
print(order, {n: round(v, 2) for n, v in k.items() if any(t in n for t in ("partition", "norm_pool", "n1_fused")) and "finish" not in n}, flush=True)

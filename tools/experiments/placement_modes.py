"""Do the streaming kernels' two speeds (k_partition_stats 24.5 / 25.4 ms, k_norm_pool 15.0 / 16.2, k_n1_fused 13.0 / 14.6 per 96 000 reads --
one or the other for a whole process) come with where the signal buffer lands?  One process, the buffer freed and allocated again several
times (with a spacer allocation of varying size in front), the same step timed each time."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from adapted_amd import lib

R = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
spc = bench.make_spc(200000, "llr")
m = spc.sig_preload_size
dev = torch.device("cuda", 0)
eng = lib.Engine(spc, R, m, device=0)
lens = torch.full((R,), m, dtype=torch.int32, device=dev)
rows = torch.empty((R, lib.ROW_DTYPE.itemsize), dtype=torch.uint8, device=dev)
spacers = [0, 0, 1 << 20, 3 << 20, 64 << 20, 1 << 30, 0, 5 << 30]
for trial, sp in enumerate(spacers):
    torch.cuda.empty_cache()
    spacer = torch.empty(sp, dtype=torch.uint8, device=dev) if sp else None
    sig = torch.empty((R, m), dtype=torch.float32, device=dev)
    eng.synth_fill(sig.data_ptr(), lens.data_ptr(), R, seed=1, first_read=0, decorate=True)
    torch.cuda.synchronize()
    eng.set_profiling(True)
    for it in range(3):
        eng.detect_llr_rows(sig.data_ptr(), lens.data_ptr(), R, 1000, with_start_peak=True, device_ptrs=True, rows_dev=rows.data_ptr(), tails_nan=True)
        kt = eng.kernel_times()
    eng.set_profiling(False)
    k = {}
    for name, v in kt:
        k[name] = k.get(name, 0.0) + v
    sel = {n: round(v, 2) for n, v in k.items() if any(t in n for t in ("partition", "norm_pool", "n1_fused")) and "finish" not in n}
    print("trial %d spacer %11d  sig at 0x%x (mod 2 MB: 0x%x, mod 1 GB: 0x%x)  %s" % (trial, sp, sig.data_ptr(), sig.data_ptr() % (2 << 20), sig.data_ptr() % (1 << 30), sel), flush=True)
    del sig, spacer

#!/usr/bin/env python3
"""Time the hand-written conv stack (adp_cnn_forward) against torch / MIOpen on the same input (GPU box).
usage: python tools/cnn_conv_speed.py [n_reads] [Lc]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adapted_amd import lib  # noqa: E402
from adapted_amd.config import get_chemistry_specific_config  # noqa: E402
from adapted_amd.detect import cnn  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
Lc = int(sys.argv[2]) if len(sys.argv) > 2 else 20050
torch.cuda.init()
spc = get_chemistry_specific_config("RNA004")
model = cnn.load_cnn_model(spc.cnn_boundaries.model_name, device=0)
eng = lib.Engine(spc, 8, spc.sig_preload_size, device=0)
eng.cnn_set_weights(dict(model.state_dict()))
x = torch.randn((n, 1, Lc), device="cuda")
L1 = (Lc - 1) // 3 + 1
Lo = 3 * L1 - 2
out = torch.empty((n, 2, Lo), device="cuda")
flop = 2.0 * (64 * 7 + 2 * 64 * 64 * 7 + 64 * 2 * 7) * L1 * n
for name, fn in (("hip", lambda: eng.cnn_forward(x.data_ptr(), n, Lc, out.data_ptr())),
                 ("torch", lambda: model(x))):
    with torch.no_grad():
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    print("%-6s n=%d Lc=%d: %.3f ms  %.1f TFLOP/s (%.3f of 157.3)  %.0f reads/s" % (name, n, Lc, dt * 1e3, flop / dt / 1e12, flop / dt / 157.3e12, n / dt))
    if name == "hip":
        eng.set_profiling(True)
        fn()
        print("   ", eng.kernel_times())
        eng.set_profiling(False)
# (compared in slices of 64 reads: MIOpen's own result is WRONG for a batch whose activations pass 4 GiB -- 1000 reads of the
# 200k window: |whole - sliced| ~ 20 on reads 0-299 and 700-999 -- which the library's own chunking avoids)
worst = scale = 0.0
with torch.no_grad():
    for s0 in range(0, n, 64):
        ref = model(x[s0:s0 + 64])
        worst = max(worst, float((ref - out[s0:s0 + 64]).abs().max()))
        scale = max(scale, float(ref.abs().max()))
print("max |hip - torch (64 reads at a time)| =", worst, "of scale", scale)

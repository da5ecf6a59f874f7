import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from adapted_amd import lib
from adapted_amd.detect import cnn as cnn_mod
from bench import make_spc
T = 16000
spc = make_spc(T)
spc.llr_boundaries.llr_detect = False
spc.cnn_boundaries.cnn_detect = True
spc.update_primary_method()
m = spc.sig_preload_size
mb = 1000
model = cnn_mod.load_cnn_model(spc.cnn_boundaries.model_name, device=0)
eng = lib.Engine(spc, mb, m, device=0)
sig = torch.empty((mb, m), dtype=torch.float32, device="cuda")
ln = torch.full((mb,), m, dtype=torch.int32, device="cuda")
eng.synth_fill(sig.data_ptr(), ln.data_ptr(), mb, seed=1, first_read=0, decorate=True)
lens_host = np.full(mb, m, dtype=np.int32)
core = spc.core
Lc = (m - core.min_obs_adapter + core.downscale_factor - 1) // core.downscale_factor
x = torch.empty((mb, 1, Lc), dtype=torch.float32, device="cuda")
for it in range(2):
    t = [time.perf_counter()]
    eng.cnn_prepare(sig.data_ptr(), mb, x.data_ptr(), device_ptrs=True); torch.cuda.synchronize(); t.append(time.perf_counter())
    preds = (cnn_mod.cnn_predict(x, model, spc.cnn_boundaries, core) * core.downscale_factor + core.min_obs_adapter).astype(int); t.append(time.perf_counter())
    preds[preds == core.min_obs_adapter] = 0
    bounds = np.ascontiguousarray(preds, dtype=np.int64)
    eng.set_profiling(True)
    rows = eng.validate_rows(sig.data_ptr(), ln.data_ptr(), mb, bounds, device_ptrs=True); t.append(time.perf_counter())
    kt = eng.kernel_times(); eng.set_profiling(False)
    print("prepare %.1f ms, predict(conv+topk) %.1f ms, validate %.1f ms; kernels: %s" % ((t[1]-t[0])*1e3, (t[2]-t[1])*1e3, (t[3]-t[2])*1e3, [(k, round(v, 2)) for k, v in kt]))
print("pass", rows["success"].mean(), "n candidates col:", bounds.shape)

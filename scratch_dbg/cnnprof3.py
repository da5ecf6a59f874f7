import sys, time, cProfile, pstats, io
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
orig = eng.cnn_topk
def spy(*a):
    r = orig(*a); print("   cnn_topk flag", r[2], "reads with peaks", int((r[1] > 0).sum())); return r
eng.cnn_topk = spy
f = lambda: cnn_mod.detect_rows_device(eng, sig.data_ptr(), ln.data_ptr(), mb, lens_host, model, spc)
f(); torch.cuda.synchronize()
eng.cnn_topk = orig
t0 = time.perf_counter(); rows = f(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("T=%d: %.4f s per 1000 reads (%.0f reads/s), pass %.2f" % (T, dt, mb / dt, rows["success"].mean()))
pr = cProfile.Profile(); pr.enable(); f(); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12); print(s.getvalue()[:3000])

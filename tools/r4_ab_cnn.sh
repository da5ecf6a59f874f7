#!/bin/bash
# A/B of the CNN path's candidate statistics / series kernels on the GPU box: rows digest and per-kernel times, old against new
# (run through gpurun from the repository root); results under gpurun_out/r4ab_*
export TMPDIR=/tmp
OUT=gpurun_out
TAG=${1:-r4ab}
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_cnn.py -x -q > $OUT/${TAG}_pytest_cnn.log 2>&1 || { tail -30 $OUT/${TAG}_pytest_cnn.log; exit 1; }
tail -3 $OUT/${TAG}_pytest_cnn.log
for W in 200000 16000; do
  R=8000; [ $W = 16000 ] && R=32000
  ADP_CAND_STATS_OLD=1 timeout -k 10 300 python bench.py --primary cnn --max_obs_trace $W --reads $R --steps 3 --warmup 1 --no-secondary --cpu-sample 0 > $OUT/${TAG}_old_$W.json 2> $OUT/${TAG}_old_$W.err || { tail -20 $OUT/${TAG}_old_$W.err; exit 1; }
  timeout -k 10 300 python bench.py --primary cnn --max_obs_trace $W --reads $R --steps 3 --warmup 1 --no-secondary --cpu-sample 0 > $OUT/${TAG}_new_$W.json 2> $OUT/${TAG}_new_$W.err || { tail -20 $OUT/${TAG}_new_$W.err; exit 1; }
  python - <<PY
import json
a=json.load(open("$OUT/${TAG}_old_$W.json")); b=json.load(open("$OUT/${TAG}_new_$W.json"))
print("window $W: rows equal:", a["rows_sha256"]==b["rows_sha256"], " reads/s old %.0f new %.0f  ms/step old %.2f new %.2f" % (a["value"], b["value"], a["ms_per_step"], b["ms_per_step"]))
for k in sorted(set(a["kernel_ms"])|set(b["kernel_ms"])):
    x=a["kernel_ms"].get(k,0); y=b["kernel_ms"].get(k,0)
    if max(x,y)>0.3: print("   %-28s old %8.3f new %8.3f" % (k,x,y))
PY
done

set -e
run() { # name, extra args
  ADP_ABLATE=$3 timeout -k 10 280 python bench.py --steps 4 --warmup 1 --no-secondary --cpu-sample 20 --cpu-sample-all 10 --cpu-procs 2 $2 > gpurun_out/abp_$1.json 2> gpurun_out/abp_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/abp_$1.json"))
k=d["kernel_ms"]
print("$1 ms/step %.2f  reads/s %.0f"%(d["ms_per_step"], d["value"]), {n:round(v,2) for n,v in k.items() if "partition" in n}, d["rows_sha256"][:12])
PY
}
run full "" 0
run pareto "--lens pareto" 0
run default "--max_obs_trace 16000" 0

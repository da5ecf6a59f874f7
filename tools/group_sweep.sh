#!/bin/bash
# Grouping / lane / phase-order sweep of the headline workload on the GPU box (one JSON line per setting):
#   tools/group_sweep.sh r03_b
TAG=${1:-r03}
O=gpurun_out/${TAG}_sweep.jsonl
: > $O
B="python3 bench.py --steps 4 --warmup 1 --no-secondary --cpu-sample 0"
run() { # label, env...
    local label=$1; shift
    echo "== $label" >&2
    env "$@" $B 2>> gpurun_out/${TAG}_sweep.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['kernel_ms']
print(json.dumps({'label':'$label','reads_per_s':round(d['value']),'ms_per_step':round(d['ms_per_step'],2),'kernel_ms_sum':round(d['kernel_ms_sum'],1),
  'k_partition_stats':k.get('k_partition_stats'),'k_norm_pool':k.get('k_norm_pool'),'k_n1_fused':k.get('k_n1_fused'),'k_gains1':k.get('k_gains<1>'),'k_gains2':k.get('k_gains<2>'),
  'k_validate':k.get('k_validate'),'k_polya_peak':k.get('k_polya_peak'),'k_start_peak':k.get('k_start_peak')}))" >> $O
}
run serial ADP_GROUPS=1
run default ADP_X=0
run lanes3 ADP_LANES=3
run lanes4 ADP_LANES=4 ADP_GROUPS=12
run groups12 ADP_GROUPS=12
run groups4 ADP_GROUPS=4
run stagger0 ADP_STAGGER=0
run stagger3 ADP_STAGGER=3
run stagger7 ADP_STAGGER=7
run lanes3_stagger0 ADP_LANES=3 ADP_STAGGER=0
run lanes3_stagger5 ADP_LANES=3 ADP_STAGGER=5
cat $O

#!/bin/bash
# A/B of one environment switch on ONE box: tools/ab_env.sh NAME "v1 v2 ..." [bench.py arguments]
NAME=$1; VALS=$2; shift 2
ARGS="${@:---no-secondary --steps 6 --warmup 2 --cpu-sample 0}"
for rep in 1 2; do
  for v in $VALS; do
    env $NAME=$v python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('$NAME=$v', round(d['value']), 'ms/step %.2f' % d['ms_per_step'], d['rows_sha256'][:12], ' '.join('%s=%.2f' % (n.replace('k_',''), k[n]) for n in list(k)[:13]))"
  done
done

#!/bin/bash
# A/B of the S1 kernel's shape on ONE box: workgroup size x order of the second pass (tools/ab_s1.sh [bench.py arguments])
ARGS="${@:---no-secondary --steps 6 --warmup 2 --cpu-sample 0}"
for rep in 1 2; do
  for nt in 256 512 1024; do for rev in 0 1; do
    ADP_S1_NT=$nt ADP_S1_REV=$rev python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('NT=$nt REV=$rev', round(d['value']), 'ms/step %.2f' % d['ms_per_step'], d['rows_sha256'][:12], ' '.join('%s=%.2f' % (n.replace('k_',''), k[n]) for n in list(k)[:4]))"
  done; done
done

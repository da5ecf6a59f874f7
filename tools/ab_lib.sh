#!/bin/bash
# A/B of several library builds on ONE box: tools/ab_lib.sh "cur path1.so path2.so ..." [bench.py arguments]   (cur = the in-tree build)
LIBS=$1; shift
ARGS="${@:---no-secondary --steps 6 --warmup 2 --cpu-sample 0}"
for rep in 1 2; do
  for l in $LIBS; do
    if [ $l = cur ]; then unset ADAPTED_HIP_LIB; else export ADAPTED_HIP_LIB=$PWD/$l; fi
    python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('$l', round(d['value']), 'ms/step %.2f' % d['ms_per_step'], d['rows_sha256'][:12], ' '.join('%s=%.2f' % (n.replace('k_',''), k[n]) for n in list(k)[:14]))"
  done
done

#!/bin/bash
# A/B of several library builds on ONE box: tools/ab_libs.sh "<bench.py arguments>" lib1.so lib2.so ...  ("cur" = the in-tree build)
ARGS="$1"; shift
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = cur ]; then unset ADAPTED_HIP_LIB; else export ADAPTED_HIP_LIB=$PWD/$lib; fi
    python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('%-28s' % '$lib', round(d['value']), 'ms/step %.2f' % d['ms_per_step'], d['rows_sha256'][:10], ' '.join('%s=%.2f' % (n.replace('k_',''), k[n]) for n in list(k)[:8]))"
  done
done

#!/bin/bash
# A/B of two library builds on ONE box (boxes differ by several per cent): the current build against adapted_amd/lib/dbg/prev.so
# (a developer's copy of an earlier build), alternating, same bench arguments.  usage: tools/ab_bench.sh [bench.py arguments]
ARGS="${@:---no-secondary --steps 6 --warmup 2 --cpu-sample 0}"
for rep in 1 2; do
  for which in prev cur; do
    if [ $which = prev ]; then export ADAPTED_HIP_LIB=$PWD/adapted_amd/lib/dbg/prev.so; else unset ADAPTED_HIP_LIB; fi
    python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('$which', round(d['value']), 'ms/step %.2f' % d['ms_per_step'], ' '.join('%s=%.2f' % (n.replace('k_',''), k[n]) for n in list(k)[:13]))"
  done
done

#!/bin/bash
# timing-only ablations of k_cnn_conv64p (builds with -DCNP_ABL=k under adapted_amd/lib/dbg/cnp<k>.so); results are wrong
export ADP_CNN_PIPE=1
for l in cur 1 2 3 4 8 11; do
  if [ $l = cur ]; then unset ADAPTED_HIP_LIB; else export ADAPTED_HIP_LIB=$PWD/adapted_amd/lib/dbg/cnp$l.so; fi
  python bench.py --primary cnn --reads 12000 --max_obs_trace 200000 --no-secondary --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('abl=$l', ' '.join('%s=%.2f' % (n.replace('k_',''), k[n]) for n in k if 'conv64' in n))"
done

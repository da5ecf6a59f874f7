#!/bin/bash
# timing-only ablations of k_cand_stats2 (ADP_ABLATE bits 2^20 no finish, 2^21 no sweep B, 2^23 no sweep A); results are wrong when set
# usage: tools/experiments/cs2_ablate.sh [bench.py arguments]   (default: 24 000 reads at the 200 k window)
ARGS="${@:---primary cnn --reads 24000 --max_obs_trace 200000 --no-secondary --steps 4 --warmup 2 --cpu-sample 0}"
for a in 0 1048576 2097152 3145728 8388608 11534336; do
ADP_ABLATE=$a python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('abl=$a', 'ms/step %.2f' % d['ms_per_step'], 'cand_stats=%.2f' % k.get('k_cand_stats',0))"
done

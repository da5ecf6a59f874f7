for a in 0 1048576 2097152 3145728 8388608 11534336; do
ADP_ABLATE=$a python bench.py --primary cnn --reads 24000 --max_obs_trace 200000 --no-secondary --steps 4 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('abl=$a', 'ms/step %.2f' % d['ms_per_step'], 'cand_stats=%.2f' % k.get('k_cand_stats',0))"
done

for r in 2000 6000 12000 12288 12336 24000; do
python bench.py --primary cnn --reads $r --steps 2 --warmup 1 --no-secondary --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print($r, 'ms/step %.2f' % d['ms_per_step'], ' '.join('%s=%.2f' % (n.replace('k_',''), k[n]) for n in list(k)[:12]))"
done

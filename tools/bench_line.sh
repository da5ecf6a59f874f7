python bench.py "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print(round(d['value']), 'ms/step %.2f' % d['ms_per_step'], d['rows_sha256'][:12], ' '.join('%s=%.2f' % (n.replace('k_',''), k[n]) for n in list(k)[:12]))"

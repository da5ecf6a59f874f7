# per-kernel times of the LLR step against the number of reads per step: a kernel whose time does not fall with the reads is bound by a
# chain, not by throughput
for r in 6000 12000 24000 48000 96000; do
python bench.py --reads $r --steps 3 --warmup 1 --no-secondary --cpu-sample 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print($r, 'ms/step %.2f' % d['ms_per_step'], ' '.join('%s=%.2f' % (n.replace('k_',''), k[n]) for n in list(k)[:14]))"
done

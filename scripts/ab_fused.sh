for m in 0 4096 8192 16384; do
TTX_ALLOW_DEBUG=1 TTX_DEBUG_SKIP=$m python bench.py --no-cpu-baseline --steps 200 --repeats 3 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('mask $m', d['ms_per_step'], d['no_prefetch']['ms_per_step'] if d.get('no_prefetch') else None, d['eager_ms_per_step'], d['kernel_us'])"
done

for b in 163840 122880 98304 81920 65536 49152; do
for w in r128; do
TTX_LDS_BUDGET=$b python bench.py --workload $w --steps 20 --repeats 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$w budget $b', d['ms_per_step'], d['kernel_us'])"
done; done
python - <<'PY'
import sys; sys.path.insert(0,'fbtt-embedding_amd')
import tt_embeddings as E
for b in (163840,122880,98304,81920,65536,49152):
    E.debug_lds_budget(b); print(b, E.debug_tiles(1,[200,220,250],[4,4,4],[1,128,128,1]))
PY

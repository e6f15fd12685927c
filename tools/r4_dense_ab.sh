# Diagnostic A/B: option dense_persist (env RAG_DENSE_PERSIST) on the 12.5M-row share at 256 queries and on the 1M-row bench shape
cd ${GRAFT_REPO_ROOT:-.}
for p in 0 1; do
  for cfg in "12500000 256" "1000000 1024" "1000000 256"; do
    set -- $cfg
    echo "== RAG_DENSE_PERSIST=$p rows $1 queries $2"
    RAG_DENSE_PERSIST=$p timeout -k 10 300 python bench.py --mode dense --rows $1 --queries $2 --steps 6 --warmup 2 --no-cpu-baseline --latency-batches 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['value'],'q/s', d['ms_per_step'],'ms', 'roof', d['roofline']['bound'], d['roofline']['frac'], 'other', d['roofline']['other_roof']['frac'], 'planted', d['exactness']['planted_neighbour_at_rank1'])"
  done
done

# Diagnostic: stage growth of the dense schedule on the 12.5M-row share (256 / 128 queries)
cd ${GRAFT_REPO_ROOT:-.}
for g in 0 16 32; do
  for q in 256 128; do
    echo "== RAG_STAGE_GROWTH=$g queries $q"
    RAG_STAGE_GROWTH=$g timeout -k 10 300 python bench.py --mode dense --rows 12500000 --queries $q --steps 6 --warmup 2 --no-cpu-baseline --latency-batches 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['value'],'q/s', d['ms_per_step'],'ms', 'roof', d['roofline']['bound'], d['roofline']['frac'], 'launches/step', d['roofline']['launches_per_step'], 'planted', d['exactness']['planted_neighbour_at_rank1'], 'overflowed', d['exactness'].get('overflowed'))"
  done
done

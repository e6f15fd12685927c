# BM25 first-stage width A/B on the product library (RAG_BM25_FIRST_RANGES = exact-select ranges of the opening stage). Run through gpurun.
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_hybrid_gpu.py tests/test_property_gpu.py tests/test_full_size_gpu.py tests/test_threads_gpu.py -x -q -m gpu -k "bm25 or hybrid or threads or concurrent" 2>&1 | tail -1
for f in 2 4 3; do
  RAG_BM25_FIRST_RANGES=$f timeout -k 10 200 python bench.py --mode hybrid --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('first ranges $f: hybrid q/s', d['value'], 'ms', d.get('ms_per_step'))"
done

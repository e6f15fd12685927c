# bench-only BM25 variant A/B (parity tests of a variant run once, first): tools/bm25_ab2.sh <variant> ; run through gpurun
cd $GRAFT_REPO_ROOT
v=$1
export RAG_HIP_LIB=$GRAFT_REPO_ROOT/tools/bin/librag_$v.so
timeout -k 10 300 python -m pytest tests/test_hybrid_gpu.py tests/test_property_gpu.py -x -q -m gpu -k "bm25 or hybrid" 2>&1 | tail -1
run() { timeout -k 10 200 python bench.py --mode hybrid --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 hybrid q/s', d['value'], 'ms', d.get('ms_per_step'))"; }
run $v
RAG_BM25_FIRST_RANGES=4 run "$v first=4"
unset RAG_HIP_LIB
run product

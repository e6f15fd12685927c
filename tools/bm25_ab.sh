# BM25 A/B: product vs variant builds of bm25.hip (tools/bin/librag_<variant>.so, -D<variant>, linked like
# tools/ce_probe_build.sh): BM_RMW = read-add-write instead of the LDS atomic, R8192 = 8192-document ranges (two workgroups per CU).
# Each variant runs the BM25 parity tests first. Run through gpurun.
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ "$lib" = product ]; then unset RAG_HIP_LIB; else export RAG_HIP_LIB=$GRAFT_REPO_ROOT/tools/bin/librag_$lib.so; fi
  echo "== $lib"
  timeout -k 10 300 python -m pytest tests/test_hybrid_gpu.py tests/test_property_gpu.py -x -q -m gpu -k "bm25 or hybrid" 2>&1 | tail -1
  timeout -k 10 200 python bench.py --mode hybrid --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('hybrid q/s', d['value'], 'ms', d.get('ms_per_step'), 'q/s@256', d.get('queries_per_sec_batch256'), 'p50 single', d.get('p50_single_query_latency_ms'))"
done

#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in "" "RAG_CE_OLD_GEMM=1" "RAG_CE_DEFERRED_GEMM=1"; do
  echo "== ${v:-default}"; env $v timeout -k 10 280 python bench.py --mode pipeline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['p50_single_query_latency_ms'])"
done

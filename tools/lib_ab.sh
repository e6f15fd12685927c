#!/bin/bash
# A/B of whole-library builds on one GPU box (through gpurun): tools/lib_ab.sh <name>[:ENV=VALUE] ... ; `product` = the in-tree library,
# anything else = tools/bin/librag_<name>.so (selected with RAG_HIP_LIB); an optional :ENV=VALUE is exported for that arm (e.g.
# product:RAG_BM25_NO_DENSE=1). Runs the hybrid bench per arm, twice, interleaved (boxes differ by several percent: compare
# arms of one call only).
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for arm in "$@"; do
  v=${arm%%:*}; kv=""; [[ $arm == *:* ]] && kv=${arm#*:}
  if [ "$v" = product ]; then unset RAG_HIP_LIB; else export RAG_HIP_LIB=$GRAFT_REPO_ROOT/tools/bin/librag_$v.so; fi
  [ -n "$kv" ] && export "$kv"
  timeout -k 10 200 python bench.py --mode hybrid --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$arm: hybrid q/s', d['value'], 'ms', d.get('ms_per_step'), d.get('stages_ms'))"
  [ -n "$kv" ] && unset "${kv%%=*}"
done
done

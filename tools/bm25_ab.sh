#!/bin/bash
# BM25 A/B on one GPU box (run through gpurun): tools/bm25_ab.sh <variant> [<variant> ...]
# A variant is `product` (the in-tree library), a library built by tools/bm25_variant_build.sh (tools/bin/librag_<name>.so,
# e.g. BM_R4096_T512 = 4096-document ranges on 512-thread workgroups), optionally followed by `:first=<n>` to set the number
# of exact opening-stage ranges (RAG_BM25_FIRST_RANGES). Each library runs the BM25 / hybrid parity tests once, then the hybrid
# bench (batch of 1024 queries). Results of round 2: DESIGN.md §4.2.
cd $GRAFT_REPO_ROOT
tested=""
for v in "$@"; do
  lib=${v%%:*}; first=""; [[ $v == *:first=* ]] && first=${v##*:first=}
  if [ "$lib" = product ]; then unset RAG_HIP_LIB; else export RAG_HIP_LIB=$GRAFT_REPO_ROOT/tools/bin/librag_$lib.so; fi
  if [ -n "$first" ]; then export RAG_BM25_FIRST_RANGES=$first; else unset RAG_BM25_FIRST_RANGES; fi
  if [[ " $tested " != *" $lib "* ]]; then
    timeout -k 10 300 python -m pytest tests/test_hybrid_gpu.py tests/test_property_gpu.py -x -q -m gpu -k "bm25 or hybrid" 2>&1 | tail -1
    tested="$tested $lib"
  fi
  timeout -k 10 200 python bench.py --mode hybrid --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v: hybrid q/s', d['value'], 'ms', d.get('ms_per_step'))"
done

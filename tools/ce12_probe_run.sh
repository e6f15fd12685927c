#!/bin/bash
# timing of the cross-encoder forward (bench.py --mode rerank) with the probe builds of tools/ce_probe_build.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
run() { echo "== $1"; env $2 timeout -k 10 200 python bench.py --mode rerank 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
run default ""
run old_gemm "RAG_CE_OLD_GEMM=1"
run deferred "RAG_CE_DEFERRED_GEMM=1"
for v in "$@"; do run $v "RAG_HIP_LIB=$R/tools/bin/librag_$v.so"; done

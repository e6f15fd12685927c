#!/bin/bash
# per-kernel times of bench.py --mode rerank for the default build and probe builds (args: variant names)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in default "$@"; do
  if [ $v = default ]; then unset RAG_HIP_LIB; elif [ $v = old_gemm ]; then unset RAG_HIP_LIB; export RAG_CE_OLD_GEMM=1; else export RAG_HIP_LIB=$R/tools/bin/librag_$v.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ks_$v -o f -- python3 $R/bench.py --mode rerank > /dev/null 2>&1
  unset RAG_CE_OLD_GEMM
  echo "== $v"; python $R/tools/rocpd_top.py /tmp/ks_$v/f_results.db | head -7 | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    print(r[0][:34].ljust(36), *[x.rjust(12) for x in r[1:]])"
done

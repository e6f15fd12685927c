#!/bin/bash
# Diagnostic A/B of two builds of the library on ONE box: kernel stats of `bench.py --mode rerank` with the in-tree library and with
# tools/bin/librag_$1.so (RAG_HIP_LIB), twice each, interleaved. AB_ARGS = bench.py arguments (default: --mode rerank --steps 12), AB_FILTER = kernels to print.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; S=/tmp/r4abl; mkdir -p $S $O
run() {
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $S/$1 -o s -- python3 $R/bench.py ${AB_ARGS:---mode rerank --steps 12} > $O/r4abl_$1.log 2>&1 || echo FAILED $1
  python3 $R/tools/rocpd_top.py $S/$1/s_results.db | grep -E "${AB_FILTER:-mx_gemm|ce_attention}" | awk -F'",' '{split($2,a,","); printf "%s %s | ", substr($1,7,34), a[3]}'; echo
  grep '^{' $O/r4abl_$1.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d.get('value'))"
  rm -rf $S/$1
}
unset RAG_HIP_LIB; run base1
export RAG_HIP_LIB=$R/tools/bin/librag_$1.so; run var1
unset RAG_HIP_LIB; run base2
export RAG_HIP_LIB=$R/tools/bin/librag_$1.so; run var2

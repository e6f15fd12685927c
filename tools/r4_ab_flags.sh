#!/bin/bash
# Diagnostic A/B of a compiler flag on cross_encoder.hip (GPU box): kernel stats of `bench.py --mode rerank` with the shipped library,
# then with cross_encoder.o rebuilt with $1 (e.g. -fno-slp-vectorize). The rebuilt library lives only on the box.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; S=/tmp/r4ab; mkdir -p $S $O
FLAG="$1"
run() {
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $S/$1 -o s -- python3 $R/bench.py --mode rerank --steps 12 > $O/r4ab_$1.log 2>&1 || echo FAILED $1
  python3 $R/tools/rocpd_top.py $S/$1/s_results.db | grep -E "mx_gemm|ce_attention|mx_embed" | awk -F'",' '{print substr($1,1,60) "  " $2}' > $O/r4ab_$1.txt
  grep '^{' $O/r4ab_$1.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d.get('value'), d.get('ms_per_step'))"
  rm -rf $S/$1
}
run base
cd $R/optimized-rag_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $FLAG -c cross_encoder.hip -o cross_encoder.o 2>/dev/null && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librag_hip.so *.o && run flag
paste -d'|' $O/r4ab_base.txt $O/r4ab_flag.txt | cut -c1-200

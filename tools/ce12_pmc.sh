#!/bin/bash
# PMC passes over bench.py --mode rerank for the default build and probe builds (args: variant names)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\(VMEM\|WR\|WAIT\)[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/ce12_sq_counters.txt
for v in default "$@"; do
  if [ $v = default ]; then unset RAG_HIP_LIB; else export RAG_HIP_LIB=$R/tools/bin/librag_$v.so; fi
  i=0
  for pmc in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $pmc --kernel-trace -d /tmp/pm_${v}_$i -o p -- python3 $R/bench.py --mode rerank --steps 1 --warmup 1 > /dev/null 2>&1
    echo "== $v : $pmc"; python $R/tools/rocpd_pmc.py /tmp/pm_${v}_$i/p_results.db ce_gemm12 | python3 -c "
import sys,json
d=json.load(sys.stdin)
for k,v in d.items(): print(k[:36], {c:round(x['mean']) for c,x in v.items()})"
  done
done

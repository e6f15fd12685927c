#!/bin/bash
# SQ-side counters of the BM25 scoring kernel (GPU box, through gpurun): where do the waves spend their cycles?
# Two --pmc passes (8 SQ slots each) over `bench.py --mode hybrid --only-hybrid-calls`; summaries -> gpurun_out/bm25_sq_*.json
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; S=/tmp/bm25_sq; mkdir -p $S $O
cd /tmp && export TMPDIR=/tmp
export RAG_NO_FORK=1
CMD="python3 $R/bench.py --mode hybrid --only-hybrid-calls --steps 4 --warmup 1"
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "== pass $i: $pmc"
  timeout -k 10 400 rocprofv3 --pmc $pmc --kernel-trace -d $S/p$i -o p -- $CMD > $O/bm25_sq_$i.log 2>&1 || echo FAILED
  python3 $R/tools/rocpd_pmc.py $S/p$i/p_results.db bm25_range > $O/bm25_sq_$i.json
done

#!/bin/bash
# SQ-side counters of the cross-encoder kernels (GPU box, through gpurun): what do the waves of the GEMM kernels spend their cycles on?
# Three --pmc passes over `bench.py --mode rerank`; summaries -> gpurun_out/ce_sq_*.json
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; S=/tmp/ce_sq; mkdir -p $S $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --mode rerank --no-cpu-baseline"
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  echo "== pass $i: $pmc"
  timeout -k 10 500 rocprofv3 --pmc $pmc --kernel-trace -d $S/p$i -o p -- $CMD > $O/ce_sq_$i.log 2>&1 || echo FAILED
  python3 $R/tools/rocpd_pmc.py $S/p$i/p_results.db ce_ > $O/ce_sq_$i.json
done

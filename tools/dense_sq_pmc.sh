#!/bin/bash
# SQ-side counters of the dense search kernels (GPU box, through gpurun): matrix-pipe busy share, clock, where the waves wait.
# Three --pmc passes over `bench.py --dense-only --no-cpu-baseline`; summaries -> gpurun_out/dense_sq_*.json
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; S=/tmp/dense_sq; mkdir -p $S $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --dense-only --no-cpu-baseline --steps 10 --warmup 2 --latency-batches 1"
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  echo "== pass $i: $pmc"
  timeout -k 10 500 rocprofv3 --pmc $pmc --kernel-trace -d $S/p$i -o p -- $CMD > $O/dense_sq_$i.log 2>&1 || echo FAILED
  python3 $R/tools/rocpd_pmc.py $S/p$i/p_results.db dense_emit > $O/dense_sq_$i.json
done

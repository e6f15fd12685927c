#!/bin/bash
# Round-4 measurement passes (GPU box, through gpurun). Usage: bash tools/r4_measure.sh <part> ...
#   bench    : the default bench line -> gpurun_out/r4m_bench.json
#   fullstats: rocprofv3 --kernel-trace --stats of the default `python3 bench.py` (every block of the line) + the line of that run
#   cestats  : rocprofv3 --kernel-trace --stats of `bench.py --mode rerank` (the cross-encoder forward of one 25,600-pair batch x steps)
#   cepmc    : FETCH_SIZE | WRITE_SIZE passes of the same command (per-kernel HBM traffic of the forward)
#   cesq     : three SQ-counter passes of the same command (where the waves of the MX GEMM kernels spend their cycles)
#   densepmc : kernel stats + FETCH_SIZE | WRITE_SIZE | TCC hit passes of `bench.py --mode dense` (configs[1])
#   bm25pmc  : the same for `bench.py --mode hybrid --only-hybrid-calls` + SQ counters of the range kernel
# Raw rocprof output stays in /tmp (64 MiB cap on gpurun_out); only summaries are copied to gpurun_out/r4m_*.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
S=/tmp/r4m_scratch; mkdir -p $S $O
cd /tmp && export TMPDIR=/tmp
pmc_pass() {   # tag, command, counters, kernel filter
  local tag=$1 cmd=$2 pmc=$3 flt=$4 name=$(echo $3 | tr ' ' '_')
  echo "== $tag pmc $pmc"
  timeout -k 10 500 rocprofv3 --pmc $pmc --kernel-trace -d $S/${tag}_$name -o p -- $cmd > $O/r4m_${tag}_pmc_$name.log 2>&1 || echo FAILED
  python3 $R/tools/rocpd_pmc.py $S/${tag}_$name/p_results.db "$flt" > $O/r4m_${tag}_pmc_$name.json
  rm -rf $S/${tag}_$name
}
for part in "$@"; do
case $part in
bench)
  cd $R; echo "== default bench"; timeout -k 10 900 python bench.py > $O/r4m_bench.json 2> $O/r4m_bench.err || echo FAILED; tail -c 600 $O/r4m_bench.json; cd /tmp ;;
fullstats)
  echo "== kernel stats of the whole default bench line"; timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $S/full_stats -o s -- python3 $R/bench.py > $O/r4m_full_stats.log 2>&1 || echo FAILED
  python3 $R/tools/rocpd_top.py $S/full_stats/s_results.db > $O/r4m_full_line_kernel_stats.csv
  grep '^{' $O/r4m_full_stats.log | tail -1 > $O/r4m_full_line_bench.json
  rm -rf $S/full_stats ;;
cestats)
  CMD="python3 $R/bench.py --mode rerank --steps 30"
  echo "== rerank kernel stats"; timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $S/ce_stats -o s -- $CMD > $O/r4m_ce_stats.log 2>&1 || echo FAILED
  python3 $R/tools/rocpd_top.py $S/ce_stats/s_results.db > $O/r4m_ce_kernel_stats.csv
  python3 $R/tools/rocpd_timeline.py $S/ce_stats/s_results.db 50 > $O/r4m_ce_timeline.txt
  rm -rf $S/ce_stats ;;
cepmc)
  CMD="python3 $R/bench.py --mode rerank --steps 20"
  pmc_pass ce "$CMD" "FETCH_SIZE" ""
  pmc_pass ce "$CMD" "WRITE_SIZE" "" ;;
cesq)
  CMD="python3 $R/bench.py --mode rerank --steps 20"
  pmc_pass cesq "$CMD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "_kernel"
  pmc_pass cesq "$CMD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_WAVES" "_kernel"
  pmc_pass cesq "$CMD" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" "_kernel" ;;
densepmc)
  CMD="python3 $R/bench.py --mode dense --steps 20 --no-cpu-baseline"
  echo "== dense kernel stats"; timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $S/de_stats -o s -- $CMD > $O/r4m_dense_stats.log 2>&1 || echo FAILED
  python3 $R/tools/rocpd_top.py $S/de_stats/s_results.db > $O/r4m_dense_kernel_stats.csv
  rm -rf $S/de_stats
  pmc_pass dense "$CMD" "FETCH_SIZE" "dense_emit"
  pmc_pass dense "$CMD" "WRITE_SIZE" "dense_emit"
  pmc_pass dense "$CMD" "TCC_HIT_sum TCC_MISS_sum" "dense_emit" ;;
bm25pmc)
  export RAG_NO_FORK=1
  CMD="python3 $R/bench.py --mode hybrid --only-hybrid-calls --steps 6 --warmup 1"
  echo "== hybrid kernel stats"; timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $S/hy_stats -o s -- $CMD > $O/r4m_hybrid_stats.log 2>&1 || echo FAILED
  python3 $R/tools/rocpd_top.py $S/hy_stats/s_results.db > $O/r4m_hybrid_kernel_stats.csv
  rm -rf $S/hy_stats
  pmc_pass hybrid "$CMD" "FETCH_SIZE" "bm25"
  pmc_pass hybrid "$CMD" "WRITE_SIZE" "bm25"
  pmc_pass hybrid "$CMD" "TCC_HIT_sum TCC_MISS_sum" "bm25"
  pmc_pass hybridsq "$CMD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "bm25_range"
  pmc_pass hybridsq "$CMD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE" "bm25_range"
  unset RAG_NO_FORK ;;
esac
done

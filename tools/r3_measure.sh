#!/bin/bash
# Round-3 measurement passes (GPU box, through gpurun). Usage: bash tools/r3_measure.sh <part> ...
#   bm25pmc : PMC passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum) + kernel stats of `bench.py --mode hybrid
#             --only-hybrid-calls` (every launch belongs to a 1024-query rag_hybrid_rrf_dev call)
#   cepmc   : the same counters on `bench.py --mode rerank`, + kernel stats
#   bench   : the default bench line
# Raw rocprof output stays in /tmp (64 MiB cap on gpurun_out); only the summaries are copied to gpurun_out/r3m_*.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
S=/tmp/r3m_scratch; mkdir -p $S $O
cd /tmp && export TMPDIR=/tmp
for part in "$@"; do
case $part in
bm25pmc)
  export RAG_NO_FORK=1          # the two hybrid legs in line: per-kernel durations are then those of the BM25 leg alone
  CMD="python3 $R/bench.py --mode hybrid --only-hybrid-calls --steps 6 --warmup 1"
  echo "== hybrid kernel stats"; timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $S/hy_stats -o s -- $CMD > $O/r3m_hybrid_stats.log 2>&1 || echo FAILED
  python3 $R/tools/rocpd_top.py $S/hy_stats/s_results.db > $O/r3m_hybrid_kernel_stats.csv
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $pmc | tr ' ' '_')
    echo "== hybrid pmc $pmc"; timeout -k 10 400 rocprofv3 --pmc $pmc --kernel-trace -d $S/hy_$tag -o p -- $CMD > $O/r3m_hybrid_pmc_$tag.log 2>&1 || echo FAILED
    python3 $R/tools/rocpd_pmc.py $S/hy_$tag/p_results.db bm25 > $O/r3m_hybrid_pmc_$tag.json
  done
  unset RAG_NO_FORK ;;
cepmc)
  CMD="python3 $R/bench.py --mode rerank"
  echo "== rerank kernel stats"; timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $S/ce_stats -o s -- $CMD > $O/r3m_ce_stats.log 2>&1 || echo FAILED
  python3 $R/tools/rocpd_top.py $S/ce_stats/s_results.db > $O/r3m_ce_kernel_stats.csv
  for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
    echo "== rerank pmc $pmc"; timeout -k 10 400 rocprofv3 --pmc $pmc --kernel-trace -d $S/ce_$pmc -o p -- $CMD > $O/r3m_ce_pmc_$pmc.log 2>&1 || echo FAILED
    python3 $R/tools/rocpd_pmc.py $S/ce_$pmc/p_results.db ce_ > $O/r3m_ce_pmc_$pmc.json
  done ;;
bench)
  cd $R; echo "== default bench"; timeout -k 10 600 python bench.py > $O/r3m_bench.json 2> $O/r3m_bench.err || echo FAILED; tail -c 1500 $O/r3m_bench.json; cd /tmp ;;
esac
done

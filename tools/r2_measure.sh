#!/bin/bash
# Round-2 measurement pass (run on the GPU box through gpurun): bench lines for the headline and the row-order variants,
# rocprofv3 kernel stats of the same commands, PMC passes of the dense kernel. Outputs land in gpurun_out/r2m_*.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
echo "== default bench"; timeout -k 10 500 python bench.py > $O/r2m_bench.json 2> $O/r2m_bench.err || echo "bench failed"
for c in clustered sorted tenant-contiguous; do
  echo "== corpus $c"; timeout -k 10 300 python bench.py --dense-only --corpus $c --latency-batches 30 > $O/r2m_bench_$c.json 2> $O/r2m_bench_$c.err || echo "$c failed"
done
echo "== A/B on this box: table order, second pass off"
RAG_DENSE_LINEAR_ORDER=1 timeout -k 10 300 python bench.py --dense-only --no-cpu-baseline --latency-batches 10 > $O/r2m_bench_linear_order.json 2>/dev/null
RAG_NO_SECOND_PASS=1 timeout -k 10 300 python bench.py --dense-only --no-cpu-baseline --latency-batches 10 > $O/r2m_bench_no_second_pass.json 2>/dev/null
timeout -k 10 300 python bench.py --dense-only --no-cpu-baseline --latency-batches 10 > $O/r2m_bench_dense_only.json 2>/dev/null
S=/tmp/r2m_scratch; mkdir -p $S      # raw rocprof output stays out of gpurun_out (64 MiB cap): only summaries are copied
cd /tmp && export TMPDIR=/tmp
echo "== rocprof dense"; timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $S/r2m_prof_dense -o d -- python3 $R/bench.py --dense-only --no-cpu-baseline --latency-batches 10 > $O/r2m_prof_dense.log 2>&1
echo "== rocprof full line"; timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $S/r2m_prof_full -o f -- python3 $R/bench.py --no-cpu-baseline --latency-batches 10 > $O/r2m_prof_full.log 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-24)
  echo "== pmc $pmc"; timeout -k 10 300 rocprofv3 --pmc $pmc --kernel-trace -d $S/r2m_pmc_$tag -o p -- python3 $R/bench.py --dense-only --no-cpu-baseline --steps 5 --warmup 1 --latency-batches 1 > $O/r2m_pmc_$tag.log 2>&1
done
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  echo "== rerank pmc $pmc"; timeout -k 10 300 rocprofv3 --pmc $pmc --kernel-trace -d $S/r2m_cepmc_$pmc -o p -- python3 $R/bench.py --mode rerank > $O/r2m_cepmc_$pmc.log 2>&1
done
cd $R
for d in $S/r2m_cepmc_*/; do python tools/rocpd_pmc.py $d/p_results.db ce_ > $O/$(basename $d).txt 2>&1; done
python tools/rocpd_top.py $S/r2m_prof_dense/d_results.db > $O/r2m_dense_kernel_stats.csv
python tools/rocpd_top.py $S/r2m_prof_full/f_results.db > $O/r2m_full_kernel_stats.csv
for d in $S/r2m_pmc_*/; do python tools/rocpd_pmc.py $d/p_results.db > $O/$(basename $d).txt 2>&1; done
for f in r2m_bench_linear_order r2m_bench_no_second_pass r2m_bench_dense_only; do python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" $O/$f.json; done
head -c 600 $O/r2m_bench.json; echo; for c in clustered sorted tenant-contiguous; do head -c 420 $O/r2m_bench_$c.json; echo; done
cut -c1-120 $O/r2m_dense_kernel_stats.csv | head -12

# Diagnostic: per-kernel split + timeline of single-query retrieve + rerank calls (rocprofv3 of tools/single_query_profile.py)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sqprof
rocprofv3 --kernel-trace --stats -d /tmp/sqprof -o sq -- python3 $R/tools/single_query_profile.py > $R/gpurun_out/sqprof.log 2>&1
python3 $R/tools/rocpd_top.py /tmp/sqprof/sq_results.db > $R/gpurun_out/r4_single_query_kernel_stats.csv
python3 $R/tools/rocpd_timeline.py /tmp/sqprof/sq_results.db 70 > $R/gpurun_out/r4_single_query_timeline.txt

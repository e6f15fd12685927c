R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/linprof
rocprofv3 --kernel-trace --stats -d /tmp/linprof -o l -- python3 $R/tools/r4_linear_profile.py > $R/gpurun_out/linprof.log 2>&1
python3 $R/tools/rocpd_top.py /tmp/linprof/l_results.db > $R/gpurun_out/r4_linear_kernel_stats.csv
python3 $R/tools/rocpd_timeline.py /tmp/linprof/l_results.db 30 > $R/gpurun_out/r4_linear_timeline.txt
grep "ms per" $R/gpurun_out/linprof.log; cat $R/gpurun_out/r4_linear_timeline.txt | cut -c1-120

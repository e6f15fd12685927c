# Diagnostic: per-kernel times of the MX forward (rocprofv3 --kernel-trace --stats of tools/ce_mx_check.py)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mxprof
rocprofv3 --kernel-trace --stats -d /tmp/mxprof -o mx -- python3 $R/tools/ce_mx_check.py ${1:-7680} > $R/gpurun_out/mxprof.log 2>&1
python3 $R/tools/rocpd_top.py /tmp/mxprof/mx_results.db > $R/gpurun_out/mx_kernel_stats.csv
head -25 $R/gpurun_out/mx_kernel_stats.csv | cut -c1-200
python3 $R/tools/rocpd_timeline.py /tmp/mxprof/mx_results.db 40 > $R/gpurun_out/mx_timeline.txt

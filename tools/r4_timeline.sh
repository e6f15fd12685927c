# Diagnostic: kernel timeline of the last dispatches of any bench command: bash tools/r4_timeline.sh <n_last> <bench.py args...>
R=${GRAFT_REPO_ROOT:-$PWD}
N=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tlprof
rocprofv3 --kernel-trace --stats -d /tmp/tlprof -o t -- python3 $R/bench.py "$@" > $R/gpurun_out/tlprof.log 2>&1
python3 $R/tools/rocpd_timeline.py /tmp/tlprof/t_results.db $N | cut -c1-130

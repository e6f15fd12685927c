R=$GRAFT_REPO_ROOT; cd $R
run() { env $2 timeout -k 10 200 python bench.py --dense-only --no-cpu-baseline --latency-batches 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
for i in 1 2; do run new ""; run prev "RAG_HIP_LIB=$R/tools/bin/librag_prev.so"; done

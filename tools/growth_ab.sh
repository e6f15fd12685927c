# Stage-growth A/B of the dense schedule on the shard sizes of a strong-scaled run (results in DESIGN.md 4.1); run through gpurun.
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -m pytest tests/test_cross_encoder_gpu.py -x -q -m gpu -k "sixteen or handover" > gpurun_out/p16b_tests.log 2>&1; tail -3 gpurun_out/p16b_tests.log
for rows in 125000 250000 500000; do
 for g in 0 16 64; do
  if [ $g = 0 ]; then unset RAG_STAGE_GROWTH; else export RAG_STAGE_GROWTH=$g; fi
  timeout -k 10 120 python bench.py --dense-only --rows $rows --no-cpu-baseline --latency-batches 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows',$rows,'growth',$g,d['value'],d['ms_per_step'],d.get('exactness'))"
 done
done

#!/bin/bash
# the per-GPU share of the 100M x 8 config on one MI355X: 12.5M rows (115 GB resident), 1024 / 256 / 128-query batches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for q in 1024 256 128; do
  timeout -k 10 340 python bench.py --dense-only --no-cpu-baseline --rows 12500000 --queries $q --steps 5 --warmup 2 --latency-batches 5 2>$O/r2m_shard_q$q.err | tail -1 > $O/r2m_shard_q$q.json
  python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('Q', d['config']['batch_queries'], d['value'], d['ms_per_step'], d['roofline']['bound'], d['roofline']['frac'], d['roofline'].get('other_roof'), d['exactness']['exact_scan'], d['exactness']['overflowed'], d['exactness']['stages'])" $O/r2m_shard_q$q.json
done

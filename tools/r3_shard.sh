#!/bin/bash
# Per-GPU share of configs[4] on one MI355X: (1) oracle parity at full size, (2) the measured lines. Outputs in gpurun_out/r3_shard_*.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
ROWS=${1:-12500000}
echo "== parity at $ROWS rows"
RAG_TEST_SHARD_ROWS=$ROWS timeout -k 10 1000 python -m pytest tests/test_shard_share_gpu.py -m gpu -x -q -s > $O/r3_shard_parity.log 2>&1; echo "parity rc=$?"; tail -4 $O/r3_shard_parity.log
echo "== measurements"
timeout -k 10 1000 python tools/r3_shard.py $ROWS > $O/r3_shard.json 2> $O/r3_shard.err; echo "measure rc=$?"; tail -3 $O/r3_shard.err; head -c 3000 $O/r3_shard.json

"""GPU: `python bench.py --gpus 2` end to end on ONE GPU (both ranks open cuda:0, collectives through gloo: RAG_BENCH_BACKEND=gloo),
started exactly as the driver starts the 1-GPU line (WORLD_SIZE unset: bench.py launches its ranks itself). Guards the N > 1
line the driver's scaling run depends on: ONE JSON line whose top level is BASELINE.json's metric on configs[3] (retrieve + rerank
through ShardedPipeline, whole-job value), the exchange that was used, the `dense` block (configs[1]), the configs[4] block through
the sharded classes (here with small shards), and the watchdog: when a later part stalls, what was measured still comes out, the
phase that stalled is named and the exit code is NOT zero (a stalled collective must not read as success)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, timeout=600, expect_rc0=True):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RAG_BENCH_BACKEND="gloo", **extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", "250000",
                        "--shard-rows", "125000", "--latency-batches", "3"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout)
    assert (p.returncode == 0) == expect_rc0, (p.returncode, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_two_rank_line_and_shard_block():
    d = _run({})
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "queries/sec" and d["value"] > 0
    assert "retrieve+rerank" in d["metric"] and "configs[3]" in d["config"]["workload"]
    assert d["config"]["rows_per_gpu"] == 125000 and d["config"]["corpus_rows"] == 250000 and d["config"]["batch_queries"] == 256
    assert "gloo" in d["config"]["exchange"]
    assert d["ranks_in_collective"] == 2
    assert d["sanity"]["all_slots_filled_and_sorted"] and 0 < d["p50_single_query_latency_ms"] < 1000
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["frac"] > 0 and d["cpu_baseline"] is None     # the CPU baseline is a rank-0, N = 1 leg
    dn = d["dense"]
    assert dn["value"] > 0 and dn["rows_per_gpu"] == 125000 and dn["batch_queries"] == 1024
    assert dn["exactness"]["planted_neighbour_at_rank1"] == 1.0 and dn["exactness"]["exact_scan"] == 0
    assert dn["roofline"]["frac"] > 0 and dn["cpu_baseline"] is None
    sb = d["shard_12p5M"]
    assert "error" not in sb, sb
    assert sb["rows_per_gpu"] == 125000 and sb["scaling"] == "weak"
    for block in ("dense_q256", "dense_q128", "hybrid_q256", "retrieve_rerank_q256"):
        assert sb[block]["queries_per_sec"] > 0, block
    assert sb["dense_q256"]["planted_neighbour_at_rank1"] == 1.0
    assert 0 < sb["retrieve_rerank_single_query_p50_ms"] < 1000


def test_watchdog_prints_what_was_measured_and_exits_non_zero():
    d = _run({"RAG_BENCH_DEADLINE": "0.5"}, expect_rc0=False)
    assert d["n_gpus"] == 2 and d["value"] > 0                     # the dense block, lifted to the top level: no headline was measured
    assert "dense cosine" in d["metric"]
    assert "no result within" in d["headline_error"] and "stalled in phase" in d["headline_error"]

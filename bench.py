#!/usr/bin/env python3
"""The bench line of BASELINE.json's metric, "queries/sec + p50 retrieve+rerank latency, 1536-d top-k=20, 1/2/4/8 GPU".

Top level (`value`, `ms_per_step`, `p50_single_query_latency_ms`, `roofline`, `cpu_baseline`) = BASELINE.json configs[3], the
configuration the metric is quoted on: hybrid top-100 (dense + BM25 + RRF) -> ms-marco-MiniLM-L-6 cross-encoder -> top-20,
256-query batches over a 1M x 1536-d corpus, inputs resident in HBM. A "step" = one 256-query batch answered end to end.
The same line carries the lighter configurations as blocks with their own `roofline` / `cpu_baseline`: `dense` (configs[1]:
dense cosine top-20, 1024-query batches), on one GPU `hybrid` (configs[2]) and `agent_latency` (the calls the reference agent
makes), and `shard_12p5M` (configs[4] at its real per-GPU size: 12.5M rows + postings + token store + cross-encoder on EVERY
rank, N = 1 included; weak scaling).

  python bench.py --gpus N --steps K --warmup W
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
       or just `python bench.py --gpus N`: with WORLD_SIZE unset the script starts that launcher itself as a child process
       (before anything touches a GPU) and relays rank 0's JSON line.

One process per GPU; the corpus is row-sharded over the N ranks (STRONG scaling: the same 1M-row corpus and the same batch at
every N). Every rank searches its shard through the C-ABI; candidate lists are exchanged with one RCCL all-gather (bound behind
the C-ABI: rag_comm_allgather_dev), merged and fused on the device; the 25,600 pairs of a batch are split over the ranks and
the logits gathered (optimized-rag_amd/sharded.py). Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

CHUNK_ROWS = 125_000            # the corpus is generated in 8 seeded chunks so it is identical for every N
DIM = 1536
PEAK_MFMA_TFLOPS = 2500.0       # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0           # HBM3E spec peak, MI355X_MICROARCH.md


N_CLUSTERS = 1000               # --corpus clustered: 1,000 clusters of rows/1000 rows each, stored cluster by cluster
N_TENANTS = 100                 # --corpus tenant-contiguous: 100 tenants of rows/100 contiguous rows each
BENCH_TENANT = 97               # ... and the whole query batch belongs to a tenant near the END of the table


def gen_chunk(c, rows, device, kind="iid", total_rows=None):
    """Chunk c (rows [c*rows, (c+1)*rows)) of the synthetic corpus, unit rows.
    iid (BASELINE configs[1]): i.i.d. Gaussian directions - any row order is exchangeable.
    clustered: row = unit(center + noise of norm ~1) (cosine ~0.71 to its centre), 1,000 rows per cluster, stored cluster
      by cluster (a table inserted file by file): a query's whole neighbourhood sits in one ~1,000-row stretch.
    sorted: an i.i.d. direction plus a drift along a fixed axis that grows with the row number from -1.5 to +1.5 (a
      topic-sorted table): the rows most similar to a late query are all at the end of the table.
    tenant-contiguous: i.i.d. rows; the tenant column (bench_tenants) is what is structured."""
    g = torch.Generator(device=device)
    g.manual_seed(1234 + c)
    x = torch.randn((rows, DIM), generator=g, device=device, dtype=torch.float32)
    if kind == "clustered":
        per = max(1, (total_rows or rows) // N_CLUSTERS)
        cid = (torch.arange(rows, device=device) + c * rows) // per
        cg = torch.Generator(device=device)
        cg.manual_seed(99)
        centers = torch.randn((N_CLUSTERS + 1, DIM), generator=cg, device=device)
        centers /= centers.norm(dim=1, keepdim=True)
        x = centers[cid.clamp(max=N_CLUSTERS)] + x / DIM ** 0.5
    elif kind == "sorted":
        ug = torch.Generator(device=device)
        ug.manual_seed(98)
        u = torch.randn((DIM,), generator=ug, device=device)
        u /= u.norm()
        pos = (torch.arange(rows, device=device, dtype=torch.float32) + c * rows) / float(total_rows or rows)
        x = x / DIM ** 0.5 + (3.0 * pos - 1.5)[:, None] * u[None, :]
    return x / x.norm(dim=1, keepdim=True)


def bench_tenants(total_rows):
    """tenant-contiguous: tenant t owns rows [t * total/100, (t+1) * total/100) - the order a per-agent export produces."""
    return (np.arange(total_rows, dtype=np.int64) * N_TENANTS // total_rows).astype(np.int32)


def gen_queries(Q, total_rows, n_chunks, chunk_rows, device, kind="iid"):
    """query i = normalised(corpus[r_i] + 0.5 * unit-scale noise): one planted neighbour at cos ~0.89. For the structured
    corpora the planted rows are drawn where the old contiguous schedule was weakest: the last fifth of the table
    (clustered / sorted), the bench tenant's own rows (tenant-contiguous)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(4321)
    if kind == "tenant-contiguous":
        lo, hi = BENCH_TENANT * total_rows // N_TENANTS, (BENCH_TENANT + 1) * total_rows // N_TENANTS
    elif kind in ("clustered", "sorted"):
        lo, hi = total_rows * 4 // 5, total_rows
    else:
        lo, hi = 0, total_rows
    rows = torch.randint(lo, hi, (Q,), generator=g)
    noise_g = torch.Generator(device=device)
    noise_g.manual_seed(4322)
    noise = torch.randn((Q, DIM), generator=noise_g, device=device) * (0.5 / DIM ** 0.5)
    q = torch.empty((Q, DIM), device=device)
    for c in range(n_chunks):
        sel = ((rows // chunk_rows) == c).nonzero().flatten()
        if sel.numel() == 0:
            continue
        chunk = gen_chunk(c, chunk_rows, device, kind, total_rows)
        q[sel.to(device)] = chunk[(rows[sel] % chunk_rows).to(device)]
        del chunk
    q = q + noise
    return (q / q.norm(dim=1, keepdim=True)).contiguous(), rows


def choose_exchange(eng, rank, world, device, backend, torch_comm=False, make_probe=None, probe_timeout=90.0):
    """Which gather the sharded classes use, decided COLLECTIVELY: RCCL behind the C-ABI (rag_comm_allgather_dev) when every rank
    could create its communicator, torch.distributed otherwise. The 128-byte ids travel through the process group that launched
    the ranks; a failure on ANY rank (librccl not loadable, ncclCommInitRank failing) sends ALL of them to the torch path - a
    rank that did succeed destroys its communicator again, so no rank is left alone inside a collective.
    ncclCommInitRank is itself a collective and has never run with more than one rank in this repository's own runs: it is first
    tried on a throw-away PROBE handle in a thread with a deadline, so that a rendezvous that never completes costs `probe_timeout`
    seconds and the torch path, not the whole line (a thread stuck inside the library holds only the probe handle's lock; the
    engine the bench measures is not touched until every rank's probe has succeeded). Returns the label the bench line reports as
    `config.exchange` (tests/test_bench_launcher.py drives this with failing and hanging ranks on gloo)."""
    if world <= 1:
        return "single process"
    label = "torch.distributed (%s)" % backend
    if backend != "nccl" or torch_comm:
        return label
    import threading
    ok = 1
    try:
        uid = [(eng.comm_unique_id(), eng.comm_unique_id()) if rank == 0 else None]
    except Exception:
        uid, ok = [None], 0
    dist.broadcast_object_list(uid, src=0)
    probe, done = None, {"ok": False}
    if uid[0] is not None and ok:
        try:
            if make_probe is None:
                from optimized_rag_amd import RagEngine
                probe = RagEngine(dim=8, device=device.index or 0)
            else:
                probe = make_probe()

            def run():
                try:
                    probe.comm_init(rank, world, uid[0][0])
                    done["ok"] = True
                except Exception:
                    done["ok"] = False

            th = threading.Thread(target=run, daemon=True)
            th.start()
            th.join(probe_timeout)
            ok = 1 if (done["ok"] and not th.is_alive()) else 0
        except Exception:
            ok = 0
    else:
        ok = 0
    flag = torch.tensor([ok], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        try:
            eng.comm_init(rank, world, uid[0][1])
            ok2 = 1
        except Exception:
            ok2 = 0
        flag2 = torch.tensor([ok2], device=device)
        dist.all_reduce(flag2, op=dist.ReduceOp.MIN)
        try:
            probe.comm_destroy()
        except Exception:
            pass
        if int(flag2.item()) == 1:
            return "rag_comm_allgather_dev (RCCL behind the C-ABI)"
        if ok2:
            eng.comm_destroy()
        return label
    if ok and probe is not None:
        try:
            probe.comm_destroy()
        except Exception:
            pass
    return label


def launcher_command(n_gpus, port, argv):
    """The command the driver itself uses for N > 1 (one rank per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE: start the N ranks as a CHILD process tree (never exec from a process that
    may have touched the GPU - this one has not: importing torch does not initialise HIP), relay rank 0's JSON line, return
    the child's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(launcher_command(n_gpus, port, argv), env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stdout.write(proc.stdout)
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed 256-query retrieve + rerank batches (the headline); the dense block times 20 of its own")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dense-steps", type=int, default=20, help="timed 1024-query batches of the `dense` block (configs[1])")
    ap.add_argument("--rows", type=int, default=1_000_000, help="total corpus rows (sharded over the ranks)")
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--mode", default="line", choices=["line", "dense", "hybrid", "rerank", "pipeline"],
                    help="line = the whole bench line (default); dense = only its `dense` block as the top level (profiling runs of "
                         "configs[1]); hybrid / rerank / pipeline = one stage alone (bench_modes.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--torch-comm", action="store_true",
                    help="N > 1: gather through torch.distributed instead of rag_comm_allgather_dev (RCCL bound behind the C-ABI, the default)")
    ap.add_argument("--abi-comm", action="store_true", help="(default since round 3; kept so that older command lines still parse)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU rehearsal of the N > 1 launch path (tests/test_bench_launcher.py): the ranks join a gloo group, count "
                         "themselves with one all-reduce, rank 0 prints a JSON line; no GPU is touched")
    ap.add_argument("--shard-rows", type=int, default=12_500_000,
                    help="rows PER GPU of the configs[4] block (0 skips it)")
    ap.add_argument("--dense-only", action="store_true", help="skip the hybrid / retrieve_rerank / agent_latency blocks")
    ap.add_argument("--corpus", default="iid", choices=["iid", "clustered", "sorted", "tenant-contiguous"],
                    help="row order / structure of the synthetic corpus (default: i.i.d. unit Gaussians, BASELINE configs[1]); "
                         "see bench_modes.structured_chunk")
    ap.add_argument("--latency-batches", type=int, default=100, help="timed batches for the p50 (after 10 warm-ups)")
    ap.add_argument("--single-query-latency", action="store_true",
                    help="also time Q=1 searches (off by default so that every launch of the run has the bench shape and "
                         "rocprofv3's per-kernel averages match the reported ones)")
    ap.add_argument("--cpu-sample-queries", type=int, default=128)
    ap.add_argument("--only-hybrid-calls", action="store_true",
                    help="--mode hybrid: issue nothing but full-batch rag_hybrid_rrf_dev calls (profiling runs: rocprofv3's per-kernel "
                         "averages are then per-batch figures of exactly the timed shape)")
    ap.add_argument("--vocab", type=int, default=100_000, help="--mode hybrid: BM25 vocabulary size (> 100000 selects the "
                    "truncated-Zipf generator bench_modes.zipf_postings_gpu)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.launcher_selftest:
        return launcher_selftest(args)
    if args.mode not in ("line", "dense"):
        from bench_modes import run_mode
        return run_mode(args)
    if args.mode == "dense":
        args.dense_only, args.dense_steps, args.shard_rows = True, args.steps, 0
    logging_quiet()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # RAG_BENCH_BACKEND=gloo lets the multi-rank path be rehearsed with all ranks on ONE GPU (collectives through the host);
    # the real runs use nccl (= RCCL over xGMI), one GPU per rank
    backend = os.environ.get("RAG_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from optimized_rag_amd import RagEngine
    Q, k = args.queries, args.k
    # the corpus is generated in seeded 125,000-row chunks (identical for every N) and appended chunk by chunk, so a
    # 12.5M-row shard (the per-GPU share of the 100M x 8 config) never exists twice in memory
    chunk_rows = CHUNK_ROWS
    assert args.rows % (chunk_rows * world) == 0, "rows must be a multiple of 125000 * n_gpus"
    n_chunks = args.rows // chunk_rows
    my_chunks = list(range(rank * n_chunks // world, (rank + 1) * n_chunks // world))
    n_local = len(my_chunks) * chunk_rows
    id_base = my_chunks[0] * chunk_rows
    eng = RagEngine(dim=DIM, device=local_rank)
    eng.index_reserve(n_local, id_base=id_base)
    host_parts = [] if (rank == 0 and world == 1 and not args.no_cpu_baseline and args.rows <= 2_000_000) else None
    for c in my_chunks:
        blk = gen_chunk(c, chunk_rows, device, args.corpus, args.rows)
        eng.index_append(blk)
        if host_parts is not None:
            host_parts.append(blk.cpu())
        del blk
    queries, planted = gen_queries(Q, args.rows, n_chunks, chunk_rows, device, args.corpus)
    tenant = -1
    if args.corpus == "tenant-contiguous":
        assert world == 1, "--corpus tenant-contiguous is a single-GPU robustness run"
        eng.set_tenants(bench_tenants(args.rows))
        tenant = BENCH_TENANT
    host_corpus = torch.cat(host_parts) if host_parts else None
    del host_parts
    torch.cuda.empty_cache()

    from optimized_rag_amd.sharded import ShardedDenseIndex
    comm = choose_exchange(eng, rank, world, device, backend, args.torch_comm)
    index = ShardedDenseIndex(eng, rank=rank, world=world)

    def step():
        return index.search(queries, k, tenant=tenant)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        step()
    fence()
    eng.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.dense_steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    gemm_ms, gemm_launches = eng.dense_kernel_ms()
    eng.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stats = eng.dense_stats()

    # p50 latency of one batch, synchronised per step (not part of `value`): >= 100 timed batches after 10 warm-ups
    # 
    lat = []
    for it in range(10 + max(1, args.latency_batches)):
        fence()
        a = time.perf_counter()
        step()
        torch.cuda.synchronize()
        if it >= 10:
            lat.append((time.perf_counter() - a) * 1e3)
    p50 = float(np.median(lat))

    # single-query latency (what one agent turn sees): Q = 1 through the same entry point, synchronised
    lat1 = None
    if world == 1 and args.single_query_latency:
        q1 = queries[:1].contiguous()
        i1 = torch.empty((1, k), dtype=torch.int64, device=device)
        s1 = torch.empty((1, k), dtype=torch.float64, device=device)
        l1 = []
        for it in range(25):
            torch.cuda.synchronize()
            a = time.perf_counter()
            eng.dense_topk_dev(q1, k, i1, None, s1)
            torch.cuda.synchronize()
            if it >= 5:
                l1.append((time.perf_counter() - a) * 1e3)
        lat1 = float(np.median(l1))

    final_ids, final_scores = step()
    torch.cuda.synchronize()
    got_ids = final_ids.cpu().numpy()
    got_sc = final_scores.cpu().numpy()
    planted_hit = float((got_ids[:, 0] == planted.numpy()).mean())

    dense_block = dense_block_dict(args, world, n_local, tenant, gemm_ms, gemm_launches, dt, p50, lat1, stats, planted_hit, comm, host_corpus,
                                   queries, got_ids, got_sc)
    del host_corpus
    line = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "exchange": comm}
    blocks = {"dense": dense_block}

    # Everything below is measured after the dense block. If a later part stalls (one rank failing inside a collective leaves the
    # others waiting) what was measured so far must still come out: a timer thread prints the line from rank 0 and ends every
    # rank with a NON-ZERO exit code, naming the phase that did not finish.
    import threading
    state = {"phase": "start"}

    def phase(p):
        state["phase"] = p

    deadline = float(os.environ.get("RAG_BENCH_DEADLINE", os.environ.get("RAG_BENCH_SHARD_DEADLINE", "900")))

    def give_up():
        if rank == 0:
            blocks.setdefault("shard_12p5M" if "headline" in blocks else "headline",
                              {"error": f"no result within {deadline:.0f} s (RAG_BENCH_DEADLINE); stalled in phase '{state['phase']}'"})
            emit_line(args, line, blocks)
        sys.stdout.flush()
        os._exit(3)

    watchdog = threading.Timer(deadline, give_up)
    watchdog.daemon = True
    watchdog.start()

    # ---- one GPU: configs[2] and the agent's single calls on the same resident 1M-row index
    if world == 1 and not args.dense_only and args.rows <= 2_000_000 and args.corpus == "iid":
        import bench_modes as BM
        for name, fn in (("hybrid", lambda: BM.hybrid_block(eng, queries, args.rows, not args.no_cpu_baseline)[0]),
                         ("agent_latency", lambda: BM.agent_latency_block(eng, queries, args.rows, not args.no_cpu_baseline))):
            phase(name)
            try:
                blocks[name] = fn()
            except Exception as e:                      # a secondary block must never cost the line
                blocks[name] = {"error": f"{type(e).__name__}: {e}"}
    del index
    # ---- the headline: configs[3] on the same corpus (strong scaling), every N
    if not args.dense_only and args.corpus == "iid":
        import bench_shard as BS
        if world > 1:
            ranks_seen = torch.ones(1, device=device)
            dist.all_reduce(ranks_seen)                            # what torch's collective layer counted
            line["ranks_in_collective"] = int(ranks_seen.item())
            line["ranks_in_rccl_communicator"] = eng.comm_count() if comm.startswith("rag_comm") else None
        try:
            blocks["headline"] = BS.headline_rerank(eng, device, args.rows, rank=rank, world=world, steps=args.steps, warmup=args.warmup,
                                                    cpu_baseline=not args.no_cpu_baseline and world == 1, phase=phase)
        except Exception as e:
            blocks["headline"] = {"error": f"{type(e).__name__}: {e}"}
        # ---- configs[4] at its real per-GPU size (12.5M rows on every rank; weak scaling), N = 1 included
        if args.shard_rows > 0:
            try:
                phase("shard: build_shard")
                st_sh = BS.build_shard(eng, device, args.shard_rows, rank=rank, world=world, Q=256)
                phase("shard: dense / hybrid / rerank loops")
                shard_block = BS.shard_blocks(eng, st_sh, device, rank=rank, world=world, steps=3)
                shard_block.update({"rows_per_gpu": args.shard_rows, "corpus_rows": args.shard_rows * world, "scaling": "weak",
                                    "workload": f"{args.shard_rows * world} x {DIM}-d corpus row-sharded x{world} (BASELINE.json configs[4]: "
                                                f"{args.shard_rows} rows per GPU): dense, hybrid (BM25 over a 2M-term vocabulary) and retrieve "
                                                "+ rerank, 256-query batches"})
                blocks["shard_12p5M"] = shard_block
            except Exception as e:                                 # never at the cost of the line
                blocks["shard_12p5M"] = {"error": f"{type(e).__name__}: {e}", "phase": state["phase"]}
    watchdog.cancel()
    if rank == 0:
        emit_line(args, line, blocks)
    if world > 1:
        dist.destroy_process_group()


def dense_block_dict(args, world, n_local, tenant, gemm_ms, gemm_launches, dt, p50, lat1, stats, planted_hit, comm, host_corpus, queries,
                     got_ids, got_sc):
    """The `dense` block (BASELINE.json configs[1]): value, roofline of dense_emit_kernel<false>, CPU baseline. Nothing here touches
    the GPU or a collective."""
    Q, k, steps = args.queries, args.k, args.dense_steps
    # ---- roofline of the dominant kernel: dense_emit_kernel<false> (the thresholded GEMM stages) -----------
    # rows one search scores: the whole shard, or (tenant filter) the tenant's own tiles only
    t_lo, t_hi = (BENCH_TENANT * args.rows // N_TENANTS, (BENCH_TENANT + 1) * args.rows // N_TENANTS) if tenant >= 0 else (0, n_local)
    n_scan = ((t_hi + 255) // 256 - t_lo // 256) * 256 if tenant >= 0 else n_local
    stage0_rows = min(n_scan, 2048)
    dim_pad = (DIM + 63) // 64 * 64
    # algorithmic work of the thresholded GEMM stages per step on this rank: every (query, row) dot product once, every
    # fp16 corpus row read once (+ the query tile once per launch). No padding counted.
    flops_per_step = 2.0 * Q * (n_scan - stage0_rows) * DIM
    launches_per_step = gemm_launches / steps
    avg_launch_ms = gemm_ms / gemm_launches
    achieved_tflops = flops_per_step * steps / (gemm_ms * 1e-3) / 1e12
    bytes_per_step = (n_scan - stage0_rows) * dim_pad * 2.0 + launches_per_step * Q * dim_pad * 2.0
    achieved_gbs = bytes_per_step * steps / (gemm_ms * 1e-3) / 1e9
    # which roof binds this shape: time at MFMA peak vs time at HBM peak (crossover at ~300 queries per batch)
    mfma_bound = flops_per_step / (PEAK_MFMA_TFLOPS * 1e12) >= bytes_per_step / (PEAK_HBM_GBS * 1e9)
    # HBM traffic per launch: from the committed rocprofv3 --pmc passes of `bench.py --mode dense` (profiles/), which cannot be
    # collected from inside the timed run. FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950).
    traffic, traffic_src = None, None
    if world == 1 and args.rows == 1_000_000 and Q == 1024 and args.corpus == "iid":
        for name in ("r04_dense_pmc.json", "r02_g_dense_pmc.json"):
            try:                                    # a missing or reshaped profile file must never take the bench line down
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    traffic = json.load(f)["kernels"]["dense_emit_kernel<false>"]["hbm_traffic_bytes_per_launch"]["total"]
                traffic_src = f"profiles/{name}"
                break
            except (OSError, ValueError, KeyError, TypeError):
                traffic = None
    mfma_view = {"achieved": round(achieved_tflops, 2), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(achieved_tflops / PEAK_MFMA_TFLOPS, 4)}
    hbm_view = {"achieved": round(achieved_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(achieved_gbs / PEAK_HBM_GBS, 4)}
    roofline = {
        "bound": "mfma" if mfma_bound else "hbm", "kernel": "dense_emit_kernel<false>",
        **(mfma_view if mfma_bound else hbm_view),
        "traffic": traffic, "traffic_source": f"{traffic_src} (rocprofv3 --pmc passes of `bench.py --mode dense`, bytes per launch; algorithmic: "
        f"{bytes_per_step / launches_per_step:.4g})" if traffic else None,
        "launches_per_step": launches_per_step, "avg_launch_ms": round(avg_launch_ms, 4),
        "algorithmic_flops_per_launch": flops_per_step / launches_per_step,
        "algorithmic_bytes_per_launch": bytes_per_step / launches_per_step,
        "other_roof": hbm_view if mfma_bound else mfma_view,
    }

    cpu_baseline = None
    if host_corpus is not None:
        from oracle.cpu_baseline import dense_topk_blas
        qs = min(args.cpu_sample_queries, Q)
        hq = queries[:qs].cpu()
        idx, vals, cdt, threads = dense_topk_blas(host_corpus[t_lo:t_hi], hq, k)       # WHERE agent_id = ... : the tenant's rows
        idx = idx + t_lo
        same = np.mean([len(set(idx[i]) & set(got_ids[i])) / k for i in range(qs)])
        cpu_baseline = {"value": round(qs / cdt, 2), "unit": "queries/sec", "cores": threads, "kind": "port",
                        "sample": f"{qs} of the {Q} queries against the full {args.rows}x{DIM} corpus, "
                                  f"float32 BLAS exact scan + top-{k} (oracle/cpu_baseline.py), {cdt:.2f}s",
                        "recall_at_k_vs_cpu_fp32": round(float(same), 5),
                        "max_abs_score_diff": float(np.abs(vals - got_sc[:qs]).max())}
        # SURVEY 8d baseline (2): the reference's real SQL, only if a postgres with the `vector` extension exists on this box
        try:
            from oracle.cpu_baseline import pgvector_probe_and_time
            cpu_baseline["pgvector_sql"] = pgvector_probe_and_time(host_corpus.numpy(), hq.numpy(), k)
        except Exception as e:                      # the probe is optional: it must never cost the line
            cpu_baseline["pgvector_sql"] = {"error": f"{type(e).__name__}: {e}"}

    return {
        "metric": "queries/sec (dense cosine top-k=20, 1536-d)", "value": round(Q * steps / dt, 1),
        "unit": "queries/sec", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4), "scaling": "strong",
        "dtype": "f16 MFMA pass (fp32 acc) + f64 rescore",
        "workload": f"{args.rows} x {DIM}-d synthetic unit embeddings, dense cosine top-k={k}, batch={Q} queries (BASELINE.json configs[1])" +
                    ("" if args.corpus == "iid" else f"; ROW ORDER VARIANT --corpus {args.corpus} (not the headline config)"),
        "corpus": args.corpus, "corpus_rows": args.rows, "rows_per_gpu": n_local, "batch_queries": Q, "k": k,
        "parallelism": f"row-sharded x{world}" + (" + one all-gather per batch + device merge" if world > 1 else ""),
        "p50_batch_latency_ms": round(p50, 4), "p50_single_query_latency_ms": None if lat1 is None else round(lat1, 4),
        "exactness": {**stats, "planted_neighbour_at_rank1": planted_hit},
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
    }


def emit_line(args, line, blocks):
    """Rank 0: the ONE JSON line. Top level = BASELINE.json's metric on configs[3] (the `headline` measurement); `--mode dense` /
    --dense-only / a structured --corpus lift the dense block to the top level instead (profiling and robustness runs)."""
    world = line["n_gpus"]
    dense = blocks["dense"]
    head = blocks.get("headline")
    if head is None or "error" in head:
        # no headline measurement: the dense block is the line (and says so)
        out = {"metric": dense["metric"], "value": dense["value"], "unit": "queries/sec", "n_gpus": world, "steps": dense["steps"],
               "warmup": args.warmup, "ms_per_step": dense["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": dense["dtype"], "data": "synthetic",
               "config": {"workload": dense["workload"], "corpus": dense["corpus"], "corpus_rows": dense["corpus_rows"],
                          "rows_per_gpu": dense["rows_per_gpu"], "batch_queries": dense["batch_queries"], "k": dense["k"],
                          "parallelism": dense["parallelism"], "exchange": line["exchange"]},
               "p50_batch_latency_ms": dense["p50_batch_latency_ms"], "p50_single_query_latency_ms": dense["p50_single_query_latency_ms"],
               "exactness": dense["exactness"], "roofline": dense["roofline"], "cpu_baseline": dense["cpu_baseline"]}
        if head is not None:
            out["headline_error"] = head["error"]
    else:
        out = {"metric": "queries/sec + p50 retrieve+rerank latency, 1536-d top-k=20", "value": head["value"], "unit": "queries/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None,
               "dtype": "f16 MFMA + block-scaled bf8 MFMA corrections (fp32 acc) in the cross-encoder; f16 MFMA pass + f64 rescore in the "
                        "dense leg; f64 BM25 / RRF",
               "data": "synthetic",
               "config": {"workload": f"{args.rows} docs x {DIM}-d + BM25 postings (2M-term Zipf vocabulary) + 224-token passage store; batch=256 "
                                      f"queries: dense top-100 + BM25 top-100 + RRF -> top-100 -> ms-marco-MiniLM-L-6 cross-encoder (L=256, "
                                      f"25,600 pairs, mean {head['mean_pair_tokens']:.0f} tokens, seeded weights) -> top-20 (BASELINE.json configs[3])",
                          "corpus_rows": args.rows, "rows_per_gpu": args.rows // world, "batch_queries": 256, "pool": 100, "k": 20,
                          "parallelism": "one rag_retrieve_rerank_dev call per batch" if world == 1 else
                                         f"row-sharded x{world}: per-shard lists, one all-gather, device fusion, pairs split over the ranks, one gather of the logits",
                          "exchange": line["exchange"]},
               "p50_batch_latency_ms": head["p50_batch_latency_ms"], "p50_single_query_latency_ms": head["p50_single_query_latency_ms"],
               "pairs_per_sec": head["pairs_per_sec"], "sanity": head["sanity"], "hbm_used_gb": head["hbm_used_gb"],
               "roofline": head["roofline"], "cpu_baseline": head.get("cpu_baseline")}
        out["dense"] = dense
    for kk in ("ranks_in_collective", "ranks_in_rccl_communicator"):
        if kk in line:
            out[kk] = line[kk]
    for name in ("hybrid", "agent_latency", "shard_12p5M"):
        if name in blocks:
            out[name] = blocks[name]
    print(json.dumps(out), flush=True)


def launcher_selftest(args):
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if world > 1:
        dist.init_process_group("gloo")
    seen = torch.ones(1)
    if world > 1:
        dist.all_reduce(seen)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "ranks_in_collective": int(seen.item()),
                          "master_addr": os.environ.get("MASTER_ADDR"), "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))
    if world > 1:
        dist.destroy_process_group()


def logging_quiet():
    """The agent-latency block drives the mirror classes, which log at WARNING ("Consistency check found 1 contradictions" per
    call): keep the driver's stderr tail readable - only errors get through."""
    import logging
    logging.getLogger().setLevel(logging.ERROR)
    logging.getLogger("optimized_rag_amd").setLevel(logging.ERROR)


if __name__ == "__main__":
    main()

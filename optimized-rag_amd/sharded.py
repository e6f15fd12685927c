"""Row-sharded search across the GPUs of one node: one process per GPU, corpus rows (embeddings, BM25
postings, token ids) partitioned contiguously, ONE small collective per stage.

    dense :  local top-k (rag_dense_topk_dev)  ->  all_gather of [ids | float64 score bits]  ->  rag_merge_topk_dev
    hybrid:  local dense top-pool + local RAW BM25 top-pool (global idf / avgdl replicated)  ->  ONE all_gather of both
             lists  ->  rag_hybrid_fuse_gathered_dev: two merges, BM25 / global max, RRF on the MERGED lists (RRF needs
             GLOBAL ranks, so the lists are merged before fusing) - all inside the library
    rerank:  the Q x pool pairs are independent: each rank scores a contiguous slice (rag_ce_score_dev), one all_gather
             of the logits

The reference is single-process (no collective to mirror). torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests) is plumbing only (buffers, process group, the gather itself unless engine.comm_init
bound RCCL behind the C-ABI); every piece of arithmetic on the gathered lists is a HIP kernel of the library. The payload is Q*k*16 B per rank
(327 KB at Q=1024, k=20): latency-bound, so one fused gather of both arrays rather than two collectives.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world):
    """Contiguous row ranges [begin, end) per rank; the first n_rows % world ranks get one extra row."""
    base, extra = divmod(int(n_rows), int(world))
    out, b = [], 0
    for r in range(world):
        e = b + base + (1 if r < extra else 0)
        out.append((b, e))
        b = e
    return out


class ShardedDenseIndex:
    def __init__(self, engine, rank=None, world=None, group=None):
        self.engine = engine
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self._bufs = {}

    def load_shard(self, emb_shard, first_global_row):
        """emb_shard: this rank's rows; doc ids become first_global_row + local row."""
        self.engine.index_load(emb_shard, id_base=int(first_global_row))

    def _buffers(self, Q, k, device):
        key = (Q, k, str(device))
        b = self._bufs.get(key)
        if b is None:
            send = torch.empty((2, Q, k), dtype=torch.int64, device=device)
            recv = torch.empty((self.world, 2, Q, k), dtype=torch.int64, device=device)
            out_ids = torch.empty((Q, k), dtype=torch.int64, device=device)
            out_scores = torch.empty((Q, k), dtype=torch.float64, device=device)
            # views are made once: search() is on the per-batch path (0.6 ms per batch on an 8-way shard of 1M rows)
            b = self._bufs[key] = (send, recv, out_ids, out_scores, send[0], send[1].view(torch.float64),
                                   recv.view(torch.float64)[:, 1])
        return b

    def search(self, queries, k, tenant=-1):
        """queries: [Q, dim] float32 tensor replicated on every rank. Returns (ids [Q,k] int64, scores [Q,k] f64)
        of the GLOBAL top-k on every rank: cosine desc, lower doc id first on ties."""
        Q = queries.shape[0]
        send, recv, out_ids, out_scores, send_ids, send_sc, recv_sc = self._buffers(Q, k, queries.device)
        self.engine.dense_topk_dev(queries, k, send_ids, None, send_sc, tenant=tenant)
        if self.world == 1:
            return send_ids, send_sc
        _gather(recv, send, self.group, self.engine)
        self.engine.merge_topk_dev(recv, recv_sc, out_ids, out_scores, n_lists=self.world, list_stride=2 * Q * k)
        return out_ids, out_scores


def _gather(recv, send, group, engine=None):
    if engine is not None and getattr(engine, "comm_world", 0) > 1:     # RCCL bound behind the C-ABI (engine.comm_init ran)
        engine.comm_allgather_dev(send, recv)
    elif dist.get_backend(group) == "nccl":                 # RCCL: one flat gather straight into the merge buffer
        dist.all_gather_into_tensor(recv, send, group=group)
    else:
        dist.all_gather(list(recv.unbind(0)), send, group=group)


class ShardedHybridIndex(ShardedDenseIndex):
    """Dense + BM25 + RRF over row shards. Every rank ends up with the same global result.

    The engine of each rank holds its rows of the embedding matrix (ids = first_global_row + local row) and the
    doc-partitioned slice of the postings built by Bm25Postings.shard (global statistics), loaded row-aligned so both
    searches speak the same doc-id space."""

    def load_shard(self, emb_shard, first_global_row, postings_shard=None):
        super().load_shard(emb_shard, first_global_row)
        if postings_shard is not None:
            postings_shard.load(self.engine)
        self.engine.bm25_set_normalize(False)

    def _hyb_buffers(self, Q, pool, k, device):
        key = ("hyb", Q, pool, k, str(device))
        if key not in self._bufs:
            i64, f64 = torch.int64, torch.float64
            self._bufs[key] = dict(
                send=torch.empty((4, Q, pool), dtype=i64, device=device),          # dense ids | dense score bits | bm25 ids | bm25 raw bits
                recv=torch.empty((self.world, 4, Q, pool), dtype=i64, device=device),
                lists=torch.empty((2, Q, pool), dtype=i64, device=device),         # merged dense ids | merged BM25 ids
                scores=torch.empty((2, Q, pool), dtype=f64, device=device),        # cosines | BM25 / global max
                keys=torch.empty((Q, k), dtype=i64, device=device),
                rrf=torch.empty((Q, k), dtype=f64, device=device),
                ranks=torch.empty((Q, k, 2), dtype=torch.int32, device=device))
        return self._bufs[key]

    def local_lists(self, queries, term_ptr, terms, pool, k, tenant=-1):
        """This rank's half of the exchange: [4, Q, pool] int64 = dense ids | dense score bits | BM25 ids | raw BM25 bits."""
        send = self._hyb_buffers(queries.shape[0], pool, k, queries.device)["send"]
        f64 = torch.float64
        self.engine.dense_topk_dev(queries, pool, send[0], None, send[1].view(f64), tenant=tenant)
        self.engine.bm25_topk_dev(term_ptr, terms, pool, send[2], None, send[3].view(f64), tenant=tenant)
        return send

    def fuse_gathered(self, recv, k, rrf_k=60):
        """recv: [world, 4, Q, pool] int64, every rank's local_lists() -> the global result (see search_hybrid)."""
        world, _, Q, pool = recv.shape
        b = self._hyb_buffers(Q, pool, k, recv.device)
        # two merges, BM25 / global max (rag/retrieval.py:343-345), RRF on the merged lists: one library call, no torch arithmetic
        self.engine.hybrid_fuse_gathered_dev(recv, k, b["lists"], b["scores"], b["keys"], b["rrf"], b["ranks"], rrf_k=rrf_k)
        return dict(keys=b["keys"], rrf=b["rrf"], ranks=b["ranks"], dense_ids=b["lists"][0], dense_scores=b["scores"][0],
                    bm25_ids=b["lists"][1], bm25_scores=b["scores"][1])

    def search_hybrid(self, queries, term_ptr, terms, pool, k, rrf_k=60, tenant=-1):
        """queries [Q, dim] float32, term_ptr [Q+1] / terms int32 (global term ids, -1 = unknown), all replicated.
        Returns dict(keys [Q,k], rrf [Q,k], ranks [Q,k,2] (1-based rank in the dense / BM25 list, 0 = absent),
        dense_ids/dense_scores [Q,pool], bm25_ids/bm25_scores [Q,pool] (scores max-normalised with the GLOBAL max))."""
        send = self.local_lists(queries, term_ptr, terms, pool, k, tenant)
        if self.world == 1:
            return self.fuse_gathered(send.unsqueeze(0), k, rrf_k)
        recv = self._hyb_buffers(queries.shape[0], pool, k, queries.device)["recv"]
        _gather(recv, send, self.group, self.engine)
        return self.fuse_gathered(recv, k, rrf_k)


class ShardedReranker:
    """Cross-encoder scoring of P independent (query, passage) pairs split evenly over the ranks; every rank has the
    model loaded (rag_ce_load_host) and receives all P logits. One all_gather of P/world float32 per rank."""

    def __init__(self, engine, rank=None, world=None, group=None):
        self.engine = engine
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)

    def score(self, input_ids, token_type_ids, lens):
        """int32 tensors [P, L], [P, L], [P] replicated on every rank -> float32 logits [P] on every rank."""
        P = input_ids.shape[0]
        per = (P + self.world - 1) // self.world                    # equal slices (the last one may be short)
        lo, hi = min(P, self.rank * per), min(P, (self.rank + 1) * per)
        mine = torch.zeros((per,), dtype=torch.float32, device=input_ids.device)
        if hi > lo:
            self.engine.ce_score_dev(input_ids[lo:hi].contiguous(), token_type_ids[lo:hi].contiguous(),
                                     lens[lo:hi].contiguous(), mine[: hi - lo])
        if self.world == 1:
            return mine[:P]
        out = torch.empty((self.world, per), dtype=torch.float32, device=input_ids.device)
        _gather(out, mine, self.group, self.engine)
        return out.reshape(-1)[:P]


class ShardedPipeline:
    """BASELINE.json configs[4]: row-sharded hybrid retrieval + cross-encoder rerank. Every rank holds its row shard
    (embeddings + doc-partitioned postings), the cross-encoder weights and a REPLICATED passage token store (it fits:
    100M x 256 tokens x 2 B = 51 GB of 288 GB; loaded with ids starting at token_id_base). Three small collectives per
    batch: the candidate lists (ShardedHybridIndex), nothing for the pair assembly (every rank builds all pairs from the
    merged global ids), the logits (ShardedReranker). Every rank ends with the same top-k."""

    def __init__(self, engine, rank=None, world=None, group=None, token_id_base=0):
        self.engine = engine
        self.index = ShardedHybridIndex(engine, rank=rank, world=world, group=group)
        self.reranker = ShardedReranker(engine, rank=rank, world=world, group=group)
        self.token_id_base = token_id_base
        self._bufs = {}

    def retrieve_rerank(self, queries, term_ptr, terms, q_tok, q_len, pool, k, L_pair=512, rrf_k=60, cls_id=101, sep_id=102, tenant=-1):
        """Returns (ids [Q,k] int64, scores [Q,k] float64 = sigmoid(logit), logits [Q,k] float32, candidates [Q,pool])."""
        Q, dev = queries.shape[0], queries.device
        cand = self.index.search_hybrid(queries, term_ptr, terms, pool, pool, rrf_k=rrf_k, tenant=tenant)["keys"]
        key = (Q, pool, k, L_pair, str(dev))
        if key not in self._bufs:
            P = Q * pool
            self._bufs[key] = (torch.empty((P, L_pair), dtype=torch.int32, device=dev), torch.empty((P, L_pair), dtype=torch.int32, device=dev),
                               torch.empty((P,), dtype=torch.int32, device=dev), torch.empty((Q, k), dtype=torch.int64, device=dev),
                               torch.empty((Q, k), dtype=torch.float64, device=dev), torch.empty((Q, k), dtype=torch.float32, device=dev))
        pid, ptt, plen, ids, sc, lg = self._bufs[key]
        self.engine.ce_build_pairs_dev(q_tok, q_len, cand, pid, ptt, plen, token_id_base=self.token_id_base, cls_id=cls_id, sep_id=sep_id)
        logits = self.reranker.score(pid, ptt, plen).contiguous()
        self.engine.rerank_topk_dev(logits, cand, ids, sc, lg)
        return ids, sc, lg, cand

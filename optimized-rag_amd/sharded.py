"""Row-sharded dense search across the GPUs of one node: one process per GPU, corpus rows
partitioned contiguously, ONE small collective per batch.

    local top-k (rag_dense_topk_dev)  ->  all_gather of [ids | float64 score bits]  ->  rag_merge_topk_dev

The reference is single-process (no collective to mirror). torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests) is plumbing only; the merge is the HIP kernel. The payload is Q*k*16 B per rank
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
        if key not in self._bufs:
            send = torch.empty((2, Q, k), dtype=torch.int64, device=device)
            recv = torch.empty((self.world, 2, Q, k), dtype=torch.int64, device=device)
            out_ids = torch.empty((Q, k), dtype=torch.int64, device=device)
            out_scores = torch.empty((Q, k), dtype=torch.float64, device=device)
            self._bufs[key] = (send, recv, out_ids, out_scores)
        return self._bufs[key]

    def search(self, queries, k, tenant=-1):
        """queries: [Q, dim] float32 tensor replicated on every rank. Returns (ids [Q,k] int64, scores [Q,k] f64)
        of the GLOBAL top-k on every rank: cosine desc, lower doc id first on ties."""
        Q = queries.shape[0]
        send, recv, out_ids, out_scores = self._buffers(Q, k, queries.device)
        self.engine.dense_topk_dev(queries, k, send[0], None, send[1].view(torch.float64), tenant=tenant)
        if self.world == 1:
            return send[0], send[1].view(torch.float64)
        if dist.get_backend(self.group) == "nccl":          # RCCL: one flat gather straight into the merge buffer
            dist.all_gather_into_tensor(recv, send, group=self.group)
        else:
            dist.all_gather(list(recv.unbind(0)), send, group=self.group)
        self.engine.merge_topk_dev(recv, recv.view(torch.float64)[:, 1], out_ids, out_scores, n_lists=self.world,
                                   list_stride=2 * Q * k)
        return out_ids, out_scores

"""ctypes binding of librag_hip.so (include/rag_hip.h). Plain pointers and sizes only.

numpy arrays are passed as host pointers (`*_host` entry points); torch CUDA tensors as device pointers
(`*_dev` entry points, asynchronous on the current torch stream). PyTorch is used for device memory and
streams only — never for the arithmetic.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class RagError(RuntimeError):
    pass


def lib_path():
    # RAG_HIP_LIB: diagnostic builds of the library (tools/ce_probe_build.sh); the product path is the in-tree .so
    return os.environ.get("RAG_HIP_LIB") or os.path.join(_HERE, "librag_hip.so")


class DenseStats(C.Structure):
    _fields_ = [("n_queries", C.c_int32), ("proven_fast", C.c_int32), ("proven_wide", C.c_int32),
                ("exact_scan", C.c_int32), ("overflowed", C.c_int32), ("shortlist", C.c_int32),
                ("stages", C.c_int32), ("second_pass", C.c_int32), ("eps", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class CeConfig(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32),
                ("ffn", C.c_int32), ("max_pos", C.c_int32), ("type_vocab", C.c_int32), ("reserved", C.c_int32),
                ("ln_eps", C.c_double)]


_P = C.c_void_p
_SIGS = {
    "rag_version": ([], C.c_int),
    "rag_device_count": ([C.POINTER(C.c_int)], C.c_int),
    "rag_create": ([C.c_int, C.c_int, C.POINTER(_P)], C.c_int),
    "rag_destroy": ([_P], C.c_int),
    "rag_last_error": ([_P], C.c_char_p),
    "rag_synchronize": ([_P], C.c_int),
    "rag_set_profiling": ([_P, C.c_int], C.c_int),
    "rag_set_option": ([_P, C.c_char_p, C.c_int], C.c_int),
    "rag_bm25_index_bytes": ([_P, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)], C.c_int),
    "rag_bm25_grid_plan": ([C.c_int, C.c_int, C.c_int, _P], C.c_int),
    "rag_index_load_host": ([_P, _P, _P, C.c_int64, C.c_int64], C.c_int),
    "rag_index_load_dev": ([_P, _P, _P, C.c_int64, C.c_int64, _P], C.c_int),
    "rag_index_reserve": ([_P, C.c_int64, C.c_int64], C.c_int),
    "rag_index_append_host": ([_P, _P, C.c_int64], C.c_int),
    "rag_index_append_dev": ([_P, _P, C.c_int64, _P], C.c_int),
    "rag_index_set_tenants_host": ([_P, _P, C.c_int64], C.c_int),
    "rag_index_set_ids_host": ([_P, _P, C.c_int64], C.c_int),
    "rag_index_rows": ([_P, C.POINTER(C.c_int64)], C.c_int),
    "rag_index_fetch_rows_host": ([_P, _P, C.c_int, _P], C.c_int),
    "rag_dense_topk_host": ([_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P], C.c_int),
    "rag_dense_topk_dev": ([_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P], C.c_int),
    "rag_dense_last_stats": ([_P, C.POINTER(DenseStats)], C.c_int),
    "rag_dense_kernel_ms": ([_P, C.POINTER(C.c_float), C.POINTER(C.c_int)], C.c_int),
    "rag_stage_kernel_ms": ([_P, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)], C.c_int),
    "rag_merge_topk_dev": ([_P, _P, _P, C.c_int, C.c_int64, C.c_int, C.c_int, _P, _P, _P], C.c_int),
    "rag_comm_unique_id": ([_P], C.c_int),
    "rag_comm_init": ([_P, C.c_int, C.c_int, _P], C.c_int),
    "rag_comm_allgather_dev": ([_P, _P, _P, C.c_size_t, _P], C.c_int),
    "rag_comm_count": ([_P, _P], C.c_int),
    "rag_comm_destroy": ([_P], C.c_int),
    "rag_pairwise_cosine_host": ([_P, _P, C.c_int, _P, C.c_int, C.c_int, _P], C.c_int),
    "rag_pairwise_cosine_f64_host": ([_P, _P, C.c_int, _P, C.c_int, C.c_int, _P], C.c_int),
    "rag_rrf_fuse_host": ([_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P], C.c_int),
    "rag_bm25_load_host": ([_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double], C.c_int),
    "rag_bm25_topk_host": ([_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P], C.c_int),
    "rag_bm25_scores_host": ([_P, _P, _P, C.c_int, _P], C.c_int),
    "rag_bm25_scores_adhoc_host": ([_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double, _P, _P,
                                   C.c_int, _P], C.c_int),
    "rag_bm25_set_normalize": ([_P, C.c_int], C.c_int),
    "rag_chunk_chain_host": ([_P, _P, _P, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _P], C.c_int),
    "rag_mmr_select_host": ([_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _P, _P], C.c_int),
    "rag_mmr_select_dev": ([_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _P, _P, _P], C.c_int),
    "rag_rrf_fuse_dev": ([_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P], C.c_int),
    "rag_bm25_topk_dev": ([_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P], C.c_int),
    "rag_hybrid_rrf_dev": ([_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P], C.c_int),
    "rag_index_set_temporal_host": ([_P, _P, C.c_int64], C.c_int),
    "rag_hybrid_linear_dev": ([_P, _P, _P, _P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, _P, _P, _P, _P, _P, _P,
                               _P], C.c_int),
    "rag_linear_fuse_topk_host": ([_P, _P, _P, _P, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, _P, _P], C.c_int),
    "rag_ce_load_host": ([_P, C.POINTER(CeConfig), C.POINTER(_P), C.c_int], C.c_int),
    "rag_ce_score_host": ([_P, _P, _P, _P, C.c_int, C.c_int, _P], C.c_int),
    "rag_tokens_load_host": ([_P, _P, _P, C.c_int64, C.c_int], C.c_int),
    "rag_tokens_reserve": ([_P, C.c_int64, C.c_int], C.c_int),
    "rag_tokens_append_dev": ([_P, _P, _P, C.c_int64, _P], C.c_int),
    "rag_hybrid_fuse_gathered_dev": ([_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P], C.c_int),
    "rag_retrieve_rerank_dev": ([_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, _P, _P, _P, _P, _P], C.c_int),
    "rag_ce_build_pairs_dev": ([_P, _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P], C.c_int),
    "rag_rerank_topk_dev": ([_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P], C.c_int),
    "rag_ce_score_dev": ([_P, _P, _P, _P, C.c_int, C.c_int, _P, _P], C.c_int),
    "rag_embed_load_host": ([_P, C.POINTER(CeConfig), C.POINTER(_P), C.c_int, C.c_int], C.c_int),
    "rag_embed_host": ([_P, _P, _P, _P, C.c_int, C.c_int, _P], C.c_int),
    "rag_embed_dev": ([_P, _P, _P, _P, C.c_int, C.c_int, _P, _P], C.c_int),
    "rag_embed_dim": ([_P, C.POINTER(C.c_int)], C.c_int),
}


def exported_symbols():
    """Every entry point include/rag_hip.h declares (tests check the .so exports each of them)."""
    return sorted(_SIGS)


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7). Two HIP runtimes in one
    process cannot both own the GPU, and device pointers are only shareable inside one runtime, so when torch is
    installed its runtime is loaded FIRST: librag_hip.so's DT_NEEDED libamdhip64.so.7 then binds to it."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library(path=None):
    """Load librag_hip.so. Fails loudly when it has not been built (`python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or lib_path()
    if not os.path.exists(p):
        raise RagError(f"{p} is missing: build the HIP extension first (__graft_entry__.build()); "
                       "there is no CPU fallback")
    _preload_torch_hip_runtime()
    lib = C.CDLL(p)
    for name, (args, res) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError = missing export: loud
        fn.argtypes = args
        fn.restype = res
    if path is None:
        _LIB = lib
    return lib


def bm25_index_bytes(indptr, n_docs):
    """(postings, metadata, bracket-table) bytes of HBM a CSR with these offsets takes once loaded; host-only, no GPU."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    rc = load_library().rag_bm25_index_bytes(C.c_void_p(indptr.ctypes.data), int(n_docs), int(indptr.shape[0] - 1), C.byref(a), C.byref(b), C.byref(c))
    if rc != 0:
        raise RagError(f"rag_bm25_index_bytes failed ({rc})")
    return int(a.value), int(b.value), int(c.value)


def bm25_grid_plan(n_ranges_in_launch, n_queries, linear=False):
    """(workgroups, ranges, queries, query groups per range, queries per group) of one BM25 scoring launch; host-only, no GPU."""
    out = (C.c_int64 * 5)()
    rc = load_library().rag_bm25_grid_plan(int(n_ranges_in_launch), int(n_queries), int(bool(linear)), C.cast(out, C.c_void_p))
    if rc != 0:
        raise RagError(f"rag_bm25_grid_plan failed ({rc})")
    return tuple(int(v) for v in out)


def _np(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class RagEngine:
    """One handle = one GPU (one process rank). Thin object wrapper over the C-ABI."""

    def __init__(self, dim, device=0):
        self.lib = load_library()
        self.dim = int(dim)
        self.device = int(device)
        h = _P()
        rc = self.lib.rag_create(self.device, self.dim, C.byref(h))
        if rc != 0:
            raise RagError(f"rag_create(device={device}, dim={dim}) failed with {rc} (no usable MI355X / bad dim)")
        self.h = h
        self._keep = []
        self.n_rows = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.rag_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.rag_last_error(self.h)
            raise RagError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def synchronize(self):
        self._check(self.lib.rag_synchronize(self.h), "rag_synchronize")

    def set_option(self, name, value):
        """Diagnostic / tuning switch of this handle (include/rag_hip.h rag_set_option); defaults come from RAG_<NAME> at creation."""
        self._check(self.lib.rag_set_option(self.h, name.encode(), int(value)), f"rag_set_option({name})")

    def set_profiling(self, on):
        self._check(self.lib.rag_set_profiling(self.h, 1 if on else 0), "rag_set_profiling")

    # ---- dense index ---------------------------------------------------------------------------
    def index_load(self, emb, ids=None, id_base=0):
        """emb: [N, dim] float32 numpy array (host) or torch CUDA tensor (device)."""
        if _is_torch(emb):
            import torch
            assert emb.is_cuda and emb.dtype == torch.float32 and emb.is_contiguous() and emb.shape[1] == self.dim
            idp = None
            if ids is not None:
                assert ids.is_cuda and ids.dtype == torch.int64 and ids.is_contiguous()
                idp = C.c_void_p(ids.data_ptr())
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self._check(self.lib.rag_index_load_dev(self.h, C.c_void_p(emb.data_ptr()), idp, int(id_base),
                                                    emb.shape[0], st), "rag_index_load_dev")
            torch.cuda.current_stream().synchronize()
            self.n_rows = int(emb.shape[0])
            return
        emb = _np(emb, np.float32)
        if emb.ndim != 2 or emb.shape[1] != self.dim:
            raise RagError(f"index_load: expected [N,{self.dim}] got {emb.shape}")
        ida = None if ids is None else _np(ids, np.int64)
        self._check(self.lib.rag_index_load_host(self.h, _ptr(emb), _ptr(ida), int(id_base), emb.shape[0]),
                    "rag_index_load_host")
        self.n_rows = int(emb.shape[0])

    def index_reserve(self, n_rows_total, id_base=0):
        self._check(self.lib.rag_index_reserve(self.h, int(n_rows_total), int(id_base)), "rag_index_reserve")
        self.n_rows = 0

    def index_append(self, emb):
        """Append a block of rows (numpy host array or torch CUDA tensor) to a reserved index."""
        if _is_torch(emb):
            import torch
            assert emb.is_cuda and emb.dtype == torch.float32 and emb.is_contiguous() and emb.shape[1] == self.dim
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self._check(self.lib.rag_index_append_dev(self.h, C.c_void_p(emb.data_ptr()), emb.shape[0], st), "rag_index_append_dev")
            torch.cuda.current_stream().synchronize()
        else:
            emb = _np(emb, np.float32)
            self._check(self.lib.rag_index_append_host(self.h, _ptr(emb), emb.shape[0]), "rag_index_append_host")
        self.n_rows += int(emb.shape[0])

    def set_tenants(self, tenant_of_row):
        t = None if tenant_of_row is None else _np(tenant_of_row, np.int32)
        self._check(self.lib.rag_index_set_tenants_host(self.h, _ptr(t), 0 if t is None else t.shape[0]),
                    "rag_index_set_tenants_host")

    def set_ids(self, ids):
        ids = _np(ids, np.int64)
        self._check(self.lib.rag_index_set_ids_host(self.h, _ptr(ids), ids.shape[0]), "rag_index_set_ids_host")

    def fetch_rows(self, rows):
        rows = _np(rows, np.int64)
        out = np.empty((rows.shape[0], self.dim), dtype=np.float32)
        self._check(self.lib.rag_index_fetch_rows_host(self.h, _ptr(rows), rows.shape[0], _ptr(out)),
                    "rag_index_fetch_rows_host")
        return out

    def dense_topk(self, queries, k, tenant=-1):
        """queries [Q, dim] float32 (numpy). Returns (ids int64 [Q,k], rows int32 [Q,k], scores float64 [Q,k])."""
        q = _np(queries, np.float32)
        if q.ndim == 1:
            q = q[None]
        if q.shape[1] != self.dim:
            raise RagError(f"dense_topk: expected [Q,{self.dim}] got {q.shape}")
        Q = q.shape[0]
        ids = np.empty((Q, k), dtype=np.int64)
        rows = np.empty((Q, k), dtype=np.int32)
        sc = np.empty((Q, k), dtype=np.float64)
        self._check(self.lib.rag_dense_topk_host(self.h, _ptr(q), Q, int(k), int(tenant), _ptr(ids), _ptr(rows), _ptr(sc)),
                    "rag_dense_topk_host")
        return ids, rows, sc

    def dense_topk_dev(self, q, k, ids_out, rows_out, scores_out, tenant=-1, stream=None):
        """torch CUDA tensors in/out; asynchronous on `stream` (default: torch's current stream)."""
        import torch
        assert q.is_cuda and q.dtype == torch.float32 and q.is_contiguous() and q.shape[1] == self.dim
        assert ids_out.dtype == torch.int64 and scores_out.dtype == torch.float64
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        rp = None if rows_out is None else C.c_void_p(rows_out.data_ptr())
        self._check(self.lib.rag_dense_topk_dev(self.h, C.c_void_p(q.data_ptr()), q.shape[0], int(k), int(tenant),
                                                C.c_void_p(ids_out.data_ptr()), rp, C.c_void_p(scores_out.data_ptr()), st),
                    "rag_dense_topk_dev")

    def dense_stats(self):
        s = DenseStats()
        self._check(self.lib.rag_dense_last_stats(self.h, C.byref(s)), "rag_dense_last_stats")
        return s.as_dict()

    def dense_kernel_ms(self):
        ms, n = C.c_float(), C.c_int()
        self._check(self.lib.rag_dense_kernel_ms(self.h, C.byref(ms), C.byref(n)), "rag_dense_kernel_ms")
        return float(ms.value), int(n.value)

    def stage_kernel_ms(self, stage):
        """(summed device ms, spans) of profiling stage 0 dense emit / 1 BM25 top-k / 2 cross-encoder forward."""
        ms, n = C.c_float(), C.c_int()
        self._check(self.lib.rag_stage_kernel_ms(self.h, int(stage), C.byref(ms), C.byref(n)), "rag_stage_kernel_ms")
        return float(ms.value), int(n.value)

    def merge_topk_dev(self, ids, scores, ids_out, scores_out, n_lists=None, list_stride=None, stream=None):
        """ids/scores: torch CUDA tensors holding n_lists lists of [Q, k] (int64 / float64), list l at element
        offset l*list_stride (default: contiguous [L, Q, k]). Output [Q, k], score desc then id asc."""
        import torch
        Q, k = ids_out.shape
        if n_lists is None:
            n_lists = ids.shape[0]
        if list_stride is None:
            list_stride = Q * k
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self._check(self.lib.rag_merge_topk_dev(self.h, C.c_void_p(ids.data_ptr()), C.c_void_p(scores.data_ptr()),
                                                int(n_lists), int(list_stride), Q, k, C.c_void_p(ids_out.data_ptr()),
                                                C.c_void_p(scores_out.data_ptr()), st), "rag_merge_topk_dev")

    # ---- RCCL exchange behind the C-ABI (multi-GPU, one process per GPU) -----------------------------
    def comm_unique_id(self):
        """128-byte RCCL id (bytes): create on ONE rank, hand to the others out of band, pass to comm_init on every rank."""
        buf = C.create_string_buffer(128)
        rc = self.lib.rag_comm_unique_id(buf)
        if rc != 0:
            raise RagError(f"rag_comm_unique_id failed ({rc}): librccl not available")
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        """Collective over `world` ranks (one per GPU). Afterwards the sharded classes gather through the library."""
        if len(unique_id) != 128:
            raise RagError("comm_init: unique_id must be 128 bytes")
        self._check(self.lib.rag_comm_init(self.h, int(rank), int(world), C.c_char_p(unique_id)), "rag_comm_init")
        self.comm_world = int(world)

    def comm_allgather_dev(self, send, recv, stream=None):
        """recv[world, ...] <- every rank's send (torch CUDA tensors, contiguous); asynchronous on `stream`."""
        import torch
        nbytes = send.numel() * send.element_size()
        if not (send.is_contiguous() and recv.is_contiguous()) or recv.numel() * recv.element_size() != nbytes * getattr(self, "comm_world", 1):
            raise RagError("comm_allgather_dev: recv must be contiguous and hold world x send")
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self._check(self.lib.rag_comm_allgather_dev(self.h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), nbytes, st),
                    "rag_comm_allgather_dev")

    def comm_count(self):
        """Ranks of the RCCL communicator behind rag_comm_allgather_dev (ncclCommCount)."""
        n = C.c_int(0)
        self._check(self.lib.rag_comm_count(self.h, C.byref(n)), "rag_comm_count")
        return int(n.value)

    def comm_destroy(self):
        self._check(self.lib.rag_comm_destroy(self.h), "rag_comm_destroy")
        self.comm_world = 0

    # ---- small ops --------------------------------------------------------------------------------
    def pairwise_cosine(self, a, b=None):
        """float64 cosine matrix. float64 inputs (what engine.as_matrix builds from the agent's List[float]) go to the device
        unrounded (rag_pairwise_cosine_f64_host); anything else is taken as float32."""
        f64 = getattr(a, "dtype", None) == np.float64 and (b is None or getattr(b, "dtype", None) == np.float64)
        dt = np.float64 if f64 else np.float32
        a = _np(a, dt)
        b = a if b is None else _np(b, dt)
        if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[1]:
            raise RagError(f"pairwise_cosine: shapes {a.shape} {b.shape}")
        out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float64)
        if out.size:
            fn = self.lib.rag_pairwise_cosine_f64_host if f64 else self.lib.rag_pairwise_cosine_host
            self._check(fn(self.h, _ptr(a), a.shape[0], _ptr(b), b.shape[0], a.shape[1], _ptr(out)), "rag_pairwise_cosine_host")
        return out

    def chunk_chain(self, embs, sent_len, threshold, max_chunk, min_chunk):
        """SemanticChunker's sentence loop on the device: embs [n, dim] float32, sent_len [n] -> chunk number per sentence."""
        embs = _np(embs, np.float32)
        sent_len = _np(sent_len, np.int32)
        n, dim = embs.shape
        out = np.empty((n,), dtype=np.int32)
        self._check(self.lib.rag_chunk_chain_host(self.h, _ptr(embs), _ptr(sent_len), n, dim, float(threshold), int(max_chunk),
                                                  int(min_chunk), _ptr(out)), "rag_chunk_chain_host")
        return out

    MMR_MAX_CANDIDATES = 256

    def mmr_select(self, query, embs, top_k, lam, variant):
        """Greedy MMR over explicit candidates: query [dim], embs [n, dim] float32 -> (positions [<=top_k], scores).
        variant 0 = MMRDiversifier.diversify, 1 = apply_mmr (see include/rag_hip.h)."""
        query = _np(query, np.float32).reshape(-1)
        embs = _np(embs, np.float32)
        n, dim = embs.shape
        kk = int(min(top_k, n))
        sel = np.empty((kk,), dtype=np.int32)
        sc = np.empty((kk,), dtype=np.float64)
        self._check(self.lib.rag_mmr_select_host(self.h, _ptr(query), _ptr(embs), n, dim, kk, float(lam), int(variant),
                                                 _ptr(sel), _ptr(sc)), "rag_mmr_select_host")
        keep = sel >= 0
        return sel[keep], sc[keep]

    def mmr_select_dev(self, q, rows, top_k, lam, variant, sel_out, score_out, stream=None):
        """Batched MMR over rows of the resident index: q [Q, dim] float32, rows [Q, pool] int32 (CUDA tensors)."""
        import torch
        Q, pool = rows.shape
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self._check(self.lib.rag_mmr_select_dev(self.h, C.c_void_p(q.data_ptr()), C.c_void_p(rows.data_ptr()), Q, pool,
                                                int(top_k), float(lam), int(variant), C.c_void_p(sel_out.data_ptr()),
                                                C.c_void_p(score_out.data_ptr()), st), "rag_mmr_select_dev")

    def rrf_fuse(self, lists, rrf_k=60, top_k=10):
        """lists: int64 [Q, L, len] (-1 padded at tails). Returns (keys [Q,top_k], scores, ranks [Q,top_k,L])."""
        lists = _np(lists, np.int64)
        Q, L, ln = lists.shape
        keys = np.empty((Q, top_k), dtype=np.int64)
        sc = np.empty((Q, top_k), dtype=np.float64)
        ranks = np.empty((Q, top_k, L), dtype=np.int32)
        self._check(self.lib.rag_rrf_fuse_host(self.h, _ptr(lists), Q, L, ln, int(rrf_k), int(top_k), _ptr(keys), _ptr(sc),
                                               _ptr(ranks)), "rag_rrf_fuse_host")
        return keys, sc, ranks

    def linear_fuse_topk(self, semantic, keyword, temporal, alpha, beta, gamma, top_k):
        s = _np(semantic, np.float64)
        kw = _np(keyword, np.float64)
        t = None if temporal is None else _np(temporal, np.float64)
        n = s.shape[0]
        kk = min(int(top_k), n) if n else 0
        idx = np.empty((max(kk, 1),), dtype=np.int32)
        hyb = np.empty((max(n, 1),), dtype=np.float64)
        if n == 0 or kk <= 0:
            return idx[:0], hyb[:0]
        self._check(self.lib.rag_linear_fuse_topk_host(self.h, _ptr(s), _ptr(kw), _ptr(t), n, float(alpha), float(beta),
                                                       float(gamma), kk, _ptr(idx), _ptr(hyb)), "rag_linear_fuse_topk_host")
        return idx[:kk], hyb[:n]

    # ---- BM25 -------------------------------------------------------------------------------------
    def bm25_load(self, indptr, doc, tf, doc_len, idf, avgdl, k1=1.5, b=0.75):
        indptr = _np(indptr, np.int64)
        doc = _np(doc, np.int32)
        tf = _np(tf, np.int32)
        doc_len = _np(doc_len, np.int32)
        idf = _np(idf, np.float64)
        self._check(self.lib.rag_bm25_load_host(self.h, _ptr(indptr), _ptr(doc), _ptr(tf), _ptr(doc_len), _ptr(idf),
                                                doc_len.shape[0], idf.shape[0], float(avgdl), float(k1), float(b)),
                    "rag_bm25_load_host")
        self.bm25_docs = int(doc_len.shape[0])

    def bm25_topk(self, term_ptr, terms, k, tenant=-1):
        term_ptr = _np(term_ptr, np.int32)
        terms = _np(terms, np.int32)
        Q = term_ptr.shape[0] - 1
        ids = np.empty((Q, k), dtype=np.int64)
        rows = np.empty((Q, k), dtype=np.int32)
        sc = np.empty((Q, k), dtype=np.float64)
        mx = np.empty((Q,), dtype=np.float64)
        self._check(self.lib.rag_bm25_topk_host(self.h, _ptr(term_ptr), _ptr(terms), Q, int(k), int(tenant), _ptr(ids),
                                                _ptr(rows), _ptr(sc), _ptr(mx)), "rag_bm25_topk_host")
        return ids, rows, sc, mx

    def bm25_set_normalize(self, on):
        """on=False: bm25_topk* return RAW scores (a row-sharded index divides by the global max after its merge)."""
        self._check(self.lib.rag_bm25_set_normalize(self.h, 1 if on else 0), "rag_bm25_set_normalize")

    def bm25_topk_dev(self, term_ptr, terms, k, ids_out, rows_out, scores_out, raw_max_out=None, stream=None, tenant=-1):
        """Device tensors in / out (int32 term arrays; int64 ids, float64 scores [Q, k]); asynchronous on `stream`."""
        import torch
        Q = term_ptr.shape[0] - 1
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self._check(self.lib.rag_bm25_topk_dev(self.h, C.c_void_p(term_ptr.data_ptr()), C.c_void_p(terms.data_ptr()), Q, int(k),
                                               int(tenant), C.c_void_p(ids_out.data_ptr()),
                                               C.c_void_p(rows_out.data_ptr() if rows_out is not None else 0),
                                               C.c_void_p(scores_out.data_ptr()),
                                               C.c_void_p(raw_max_out.data_ptr() if raw_max_out is not None else 0), st),
                    "rag_bm25_topk_dev")

    def rrf_fuse_dev(self, lists, keys_out, scores_out, ranks_out, rrf_k=60, stream=None):
        """lists: [Q, n_lists, list_len] int64 doc ids (-1 padded) on the device -> keys/scores [Q, top_k], ranks [Q, top_k, n_lists]."""
        import torch
        Q, n_lists, list_len = lists.shape
        top_k = keys_out.shape[1]
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self._check(self.lib.rag_rrf_fuse_dev(self.h, C.c_void_p(lists.data_ptr()), Q, n_lists, list_len, int(rrf_k), int(top_k),
                                              C.c_void_p(keys_out.data_ptr()), C.c_void_p(scores_out.data_ptr()),
                                              C.c_void_p(ranks_out.data_ptr()), st), "rag_rrf_fuse_dev")

    def hybrid_rrf_dev(self, q, term_ptr, terms, pool, k, rrf_k=60, tenant=-1, stream=None):
        """All-device hybrid (torch CUDA tensors): dense top-pool + BM25 top-pool -> RRF -> (keys [Q,k] int64, rrf scores
        [Q,k] float64, ranks [Q,k,2] int32). term_ptr / terms are int32 CUDA tensors."""
        import torch
        Q = q.shape[0]
        dev = q.device
        key = ("hyb", Q, pool, k)
        if getattr(self, "_hyb_key", None) != key:
            self._hyb = (torch.empty((2, Q, pool), dtype=torch.int64, device=dev),
                         torch.empty((Q, pool), dtype=torch.float64, device=dev),
                         torch.empty((Q, k), dtype=torch.int64, device=dev),
                         torch.empty((Q, k), dtype=torch.float64, device=dev),
                         torch.empty((Q, k, 2), dtype=torch.int32, device=dev))
            self._hyb_key = key
        lists, sc, keys, rrf, ranks = self._hyb
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self._check(self.lib.rag_hybrid_rrf_dev(self.h, C.c_void_p(q.data_ptr()), C.c_void_p(term_ptr.data_ptr()),
                                                C.c_void_p(terms.data_ptr()), Q, int(pool), int(k), int(rrf_k), int(tenant),
                                                C.c_void_p(lists.data_ptr()), C.c_void_p(sc.data_ptr()),
                                                C.c_void_p(keys.data_ptr()), C.c_void_p(rrf.data_ptr()),
                                                C.c_void_p(ranks.data_ptr()), st), "rag_hybrid_rrf_dev")
        return keys, rrf, ranks

    def set_temporal(self, temporal):
        """Per-row temporal scores (float64 [n_rows], e.g. shard_format.temporal_scores) for hybrid_linear_dev; None clears."""
        t = None if temporal is None else _np(temporal, np.float64)
        self._check(self.lib.rag_index_set_temporal_host(self.h, _ptr(t), 0 if t is None else t.shape[0]), "rag_index_set_temporal_host")

    def hybrid_linear_dev(self, q, term_ptr, terms, k, alpha, beta, gamma, tenant=-1, stream=None):
        """hybrid_search over the whole resident index (CUDA tensors in): returns dict of CUDA tensors ids [Q,k] int64, rows
        [Q,k] int32, hybrid / semantic / keyword / temporal [Q,k] float64."""
        import torch
        Q, dev = q.shape[0], q.device
        key = ("lin", Q, k)
        if getattr(self, "_lin_key", None) != key:
            f64 = torch.float64
            self._lin = dict(ids=torch.empty((Q, k), dtype=torch.int64, device=dev), rows=torch.empty((Q, k), dtype=torch.int32, device=dev),
                             hybrid=torch.empty((Q, k), dtype=f64, device=dev), semantic=torch.empty((Q, k), dtype=f64, device=dev),
                             keyword=torch.empty((Q, k), dtype=f64, device=dev), temporal=torch.empty((Q, k), dtype=f64, device=dev))
            self._lin_key = key
        o = self._lin
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self._check(self.lib.rag_hybrid_linear_dev(
            self.h, C.c_void_p(q.data_ptr()), C.c_void_p(term_ptr.data_ptr()), C.c_void_p(terms.data_ptr()), Q, int(k), float(alpha),
            float(beta), float(gamma), int(tenant), C.c_void_p(o["ids"].data_ptr()), C.c_void_p(o["rows"].data_ptr()),
            C.c_void_p(o["hybrid"].data_ptr()), C.c_void_p(o["semantic"].data_ptr()), C.c_void_p(o["keyword"].data_ptr()),
            C.c_void_p(o["temporal"].data_ptr()), st), "rag_hybrid_linear_dev")
        return o

    def bm25_scores(self, term_ptr, terms):
        term_ptr = _np(term_ptr, np.int32)
        terms = _np(terms, np.int32)
        Q = term_ptr.shape[0] - 1
        out = np.zeros((Q, self.bm25_docs), dtype=np.float64)
        self._check(self.lib.rag_bm25_scores_host(self.h, _ptr(term_ptr), _ptr(terms), Q, _ptr(out)), "rag_bm25_scores_host")
        return out

    def bm25_scores_adhoc(self, indptr, doc, tf, doc_len, idf, avgdl, term_ptr, terms, k1=1.5, b=0.75):
        """Raw BM25 scores [Q, n_docs] of an ad-hoc corpus (CSR as bm25_load); the resident postings are not touched."""
        indptr, doc, tf = _np(indptr, np.int64), _np(doc, np.int32), _np(tf, np.int32)
        doc_len, idf = _np(doc_len, np.int32), _np(idf, np.float64)
        term_ptr, terms = _np(term_ptr, np.int32), _np(terms, np.int32)
        Q = term_ptr.shape[0] - 1
        out = np.zeros((Q, doc_len.shape[0]), dtype=np.float64)
        self._check(self.lib.rag_bm25_scores_adhoc_host(self.h, _ptr(indptr), _ptr(doc), _ptr(tf), _ptr(doc_len), _ptr(idf),
                                                        doc_len.shape[0], idf.shape[0], float(avgdl), float(k1), float(b),
                                                        _ptr(term_ptr), _ptr(terms), Q, _ptr(out)), "rag_bm25_scores_adhoc_host")
        return out

    # ---- cross-encoder ----------------------------------------------------------------------------
    def ce_load(self, cfg, tensors):
        c = CeConfig(cfg["vocab_size"], cfg["hidden"], cfg["layers"], cfg["heads"], cfg["ffn"], cfg["max_pos"],
                     cfg.get("type_vocab", 2), 0, float(cfg.get("eps", 1e-12)))
        arrs = [_np(t, np.float32) for t in tensors]
        ptrs = (_P * len(arrs))(*[a.ctypes.data for a in arrs])
        self._check(self.lib.rag_ce_load_host(self.h, C.byref(c), ptrs, len(arrs)), "rag_ce_load_host")

    # ---- local embedding model (BERT encoder + mean pooling) ---------------------------------------
    def embed_load(self, cfg, tensors, normalize=True):
        """tensors: rag_ce_load_host's order WITHOUT the pooler / classifier (cross_encoder.flatten_state_dict(..., head=False))."""
        c = CeConfig(cfg["vocab_size"], cfg["hidden"], cfg["layers"], cfg["heads"], cfg["ffn"], cfg["max_pos"],
                     cfg.get("type_vocab", 2), 0, float(cfg.get("eps", 1e-12)))
        arrs = [_np(t, np.float32) for t in tensors]
        ptrs = (_P * len(arrs))(*[a.ctypes.data for a in arrs])
        self._check(self.lib.rag_embed_load_host(self.h, C.byref(c), ptrs, len(arrs), 1 if normalize else 0), "rag_embed_load_host")
        self.embed_hidden = int(cfg["hidden"])

    def embed(self, input_ids, token_type_ids, lens):
        """int32 [n, L], [n, L], [n] (numpy) -> float32 [n, hidden] sentence embeddings."""
        ids, tt, ln = _np(input_ids, np.int32), _np(token_type_ids, np.int32), _np(lens, np.int32)
        n, L = ids.shape
        out = np.empty((n, self.embed_hidden), dtype=np.float32)
        self._check(self.lib.rag_embed_host(self.h, _ptr(ids), _ptr(tt), _ptr(ln), n, L, _ptr(out)), "rag_embed_host")
        return out

    def embed_dev(self, input_ids, token_type_ids, lens, out, stream=None):
        """int32 CUDA tensors [n, L], [n, L], [n] -> float32 out [n, hidden]; asynchronous on `stream`."""
        import torch
        n, L = input_ids.shape
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self._check(self.lib.rag_embed_dev(self.h, C.c_void_p(input_ids.data_ptr()), C.c_void_p(token_type_ids.data_ptr()),
                                           C.c_void_p(lens.data_ptr()), n, L, C.c_void_p(out.data_ptr()), st), "rag_embed_dev")

    def tokens_load(self, tokens, lens):
        """Passage token store: tokens [N, L] int32 WordPiece ids without [CLS]/[SEP], lens [N]; row-aligned with the index."""
        tokens = _np(tokens, np.int32)
        lens = _np(lens, np.int32)
        self._check(self.lib.rag_tokens_load_host(self.h, _ptr(tokens), _ptr(lens), tokens.shape[0], tokens.shape[1]),
                    "rag_tokens_load_host")

    def tokens_reserve(self, n_rows_total, L):
        self._check(self.lib.rag_tokens_reserve(self.h, int(n_rows_total), int(L)), "rag_tokens_reserve")

    def tokens_append_dev(self, tokens, lens, stream=None):
        """Append a row block of the passage token store from device memory: tokens [n, L] int32, lens [n] int32 CUDA tensors."""
        import torch
        assert tokens.is_cuda and tokens.dtype == torch.int32 and tokens.is_contiguous() and lens.dtype == torch.int32
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self._check(self.lib.rag_tokens_append_dev(self.h, C.c_void_p(tokens.data_ptr()), C.c_void_p(lens.data_ptr()), tokens.shape[0], st),
                    "rag_tokens_append_dev")

    def hybrid_fuse_gathered_dev(self, gathered, k, lists_out, scores_out, keys_out, rrf_out, ranks_out, rrf_k=60, stream=None):
        """gathered [world, 4, Q, pool] int64 (every rank's dense ids | cosine bits | BM25 ids | raw BM25 bits) -> merged lists
        [2, Q, pool], scores [2, Q, pool] (BM25 / global max), RRF keys / scores [Q, k], ranks [Q, k, 2]. CUDA tensors."""
        import torch
        world, _, Q, pool = gathered.shape
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self._check(self.lib.rag_hybrid_fuse_gathered_dev(
            self.h, C.c_void_p(gathered.data_ptr()), int(world), int(Q), int(pool), int(k), int(rrf_k), C.c_void_p(lists_out.data_ptr()),
            C.c_void_p(scores_out.data_ptr()), C.c_void_p(keys_out.data_ptr()), C.c_void_p(rrf_out.data_ptr()),
            C.c_void_p(ranks_out.data_ptr()), st), "rag_hybrid_fuse_gathered_dev")

    def retrieve_rerank_dev(self, q_emb, q_tok, q_len, pool, k, term_ptr=None, terms=None, rrf_k=60, tenant=-1, L_pair=512,
                            cls_id=101, sep_id=102, stream=None):
        """Dense (term_ptr None) or hybrid candidates -> cross-encoder -> top-k, all on the device (CUDA tensors).
        Returns (ids [Q,k] int64, scores [Q,k] float64 sigmoid, logits [Q,k] float32, candidates [Q,pool] int64)."""
        import torch
        Q, dev = q_emb.shape[0], q_emb.device
        key = ("rr", Q, pool, k)
        if getattr(self, "_rr_key", None) != key:
            self._rr = (torch.empty((Q, k), dtype=torch.int64, device=dev), torch.empty((Q, k), dtype=torch.float64, device=dev),
                        torch.empty((Q, k), dtype=torch.float32, device=dev), torch.empty((Q, pool), dtype=torch.int64, device=dev))
            self._rr_key = key
        ids, sc, lg, cand = self._rr
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        mode = 0 if term_ptr is None else 1
        self._check(self.lib.rag_retrieve_rerank_dev(
            self.h, C.c_void_p(q_emb.data_ptr()), C.c_void_p(term_ptr.data_ptr() if mode else 0),
            C.c_void_p(terms.data_ptr() if mode else 0), C.c_void_p(q_tok.data_ptr()), C.c_void_p(q_len.data_ptr()),
            int(q_tok.shape[1]), Q, int(pool), int(k), int(rrf_k), int(tenant), mode, int(cls_id), int(sep_id), int(L_pair),
            C.c_void_p(ids.data_ptr()), C.c_void_p(sc.data_ptr()), C.c_void_p(lg.data_ptr()), C.c_void_p(cand.data_ptr()), st),
            "rag_retrieve_rerank_dev")
        return ids, sc, lg, cand

    def ce_build_pairs_dev(self, q_tok, q_len, cand, ids_out, tt_out, lens_out, token_id_base=0, cls_id=101, sep_id=102, stream=None):
        """[CLS] query [SEP] passage [SEP] rows for a [Q, pool] table of GLOBAL candidate ids (-1 = empty) against the
        loaded token store; ids_out / tt_out are [Q*pool, L_pair] int32, lens_out [Q*pool]."""
        import torch
        Q, pool = cand.shape
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self._check(self.lib.rag_ce_build_pairs_dev(self.h, C.c_void_p(q_tok.data_ptr()), C.c_void_p(q_len.data_ptr()), int(q_tok.shape[1]),
                                                    C.c_void_p(cand.data_ptr()), Q, pool, int(token_id_base), int(ids_out.shape[1]),
                                                    int(cls_id), int(sep_id), C.c_void_p(ids_out.data_ptr()),
                                                    C.c_void_p(tt_out.data_ptr()), C.c_void_p(lens_out.data_ptr()), st),
                    "rag_ce_build_pairs_dev")

    def rerank_topk_dev(self, logits, cand, ids_out, scores_out, logits_out, stream=None):
        """sigmoid + stable (score desc, candidate order) top-k of logits [Q*pool] over the candidate table [Q, pool]."""
        import torch
        Q, pool = cand.shape
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self._check(self.lib.rag_rerank_topk_dev(self.h, C.c_void_p(logits.data_ptr()), C.c_void_p(cand.data_ptr()), Q, pool,
                                                 int(ids_out.shape[1]), C.c_void_p(ids_out.data_ptr()),
                                                 C.c_void_p(scores_out.data_ptr()), C.c_void_p(logits_out.data_ptr()), st),
                    "rag_rerank_topk_dev")

    def ce_score_dev(self, input_ids, token_type_ids, lens, logits_out, stream=None):
        """int32 CUDA tensors [P, L], [P, L], [P] -> float32 logits_out [P]; asynchronous on `stream`."""
        import torch
        P, L = input_ids.shape
        st = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        self._check(self.lib.rag_ce_score_dev(self.h, C.c_void_p(input_ids.data_ptr()), C.c_void_p(token_type_ids.data_ptr()),
                                              C.c_void_p(lens.data_ptr()), P, L, C.c_void_p(logits_out.data_ptr()), st),
                    "rag_ce_score_dev")

    def ce_score(self, input_ids, token_type_ids, lens):
        ids = _np(input_ids, np.int32)
        tt = _np(token_type_ids, np.int32)
        ln = _np(lens, np.int32)
        P, L = ids.shape
        out = np.empty((P,), dtype=np.float32)
        self._check(self.lib.rag_ce_score_host(self.h, _ptr(ids), _ptr(tt), _ptr(ln), P, L, _ptr(out)), "rag_ce_score_host")
        return out

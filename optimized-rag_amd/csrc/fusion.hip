// K5/K6: score fusion.
//   rrf_fuse_kernel          replaces ReciprocalRankFusion.fuse   (/root/reference/rag/reranker.py:224-271)
//   linear_fuse + top-k      replaces the weighted sum + sort     (/root/reference/rag/retrieval.py:294-322)
// Integer ranks and the float64 sums follow the reference's operation order exactly (1/(k+rank) accumulated in
// list order; (alpha*s + beta*kw) + gamma*t with separate roundings: the library is built with -ffp-contract=off),
// so scores are bit-identical to CPython's, and ties keep first-seen / lower-index order (stable sort).
#include "common.h"

#define RRF_MAX_ITEMS 1024
#define RRF_TABLE 2048

__device__ __forceinline__ unsigned hash64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (unsigned)x;
}

// One workgroup per query. List l of query q starts at lists + q*query_stride + l*list_stride (len keys, -1 padding).
__global__ __launch_bounds__(256) void rrf_fuse_kernel(const int64_t* __restrict__ lists, int L, int len, int64_t list_stride,
                                                        int64_t query_stride, int rrf_k, int top_k, int64_t* __restrict__ keys_out,
                                                        double* __restrict__ scores_out, int32_t* __restrict__ ranks_out) {
    __shared__ unsigned long long tkey[RRF_TABLE];     // hash table: key+1 (0 = empty)
    __shared__ int tfirst[RRF_TABLE];                  // first flat position of that key
    __shared__ int64_t item[RRF_MAX_ITEMS];
    __shared__ double oscore[RRF_MAX_ITEMS];           // per unique key (owner), indexed by owner ordinal
    __shared__ int opos[RRF_MAX_ITEMS];
    __shared__ int n_owner;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int T = L * len;
    const int64_t* src = lists + (size_t)q * query_stride;
    for (int i = tid; i < RRF_TABLE; i += 256) { tkey[i] = 0ull; tfirst[i] = 0x7fffffff; }
    for (int i = tid; i < T; i += 256) item[i] = src[(size_t)(i / len) * list_stride + (i % len)];
    if (tid == 0) n_owner = 0;
    for (int i = tid; i < top_k; i += 256) {
        keys_out[(size_t)q * top_k + i] = -1;
        scores_out[(size_t)q * top_k + i] = 0.0;
        if (ranks_out) for (int l = 0; l < L; ++l) ranks_out[((size_t)q * top_k + i) * L + l] = 0;
    }
    __syncthreads();
    // 1. first occurrence of every key (flat order = list order, then rank order)
    for (int i = tid; i < T; i += 256) {
        const int64_t key = item[i];
        if (key < 0) continue;
        const unsigned long long kk = (unsigned long long)key + 1ull;
        unsigned slot = hash64(kk) & (RRF_TABLE - 1);
        while (true) {
            const unsigned long long prev = atomicCAS(&tkey[slot], 0ull, kk);
            if (prev == 0ull || prev == kk) break;
            slot = (slot + 1) & (RRF_TABLE - 1);
        }
        atomicMin(&tfirst[slot], i);
    }
    __syncthreads();
    // 2. owners = first occurrences; each owner sums its contributions in list order (bit-exact float64)
    for (int i = tid; i < T; i += 256) {
        const int64_t key = item[i];
        if (key < 0) continue;
        const unsigned long long kk = (unsigned long long)key + 1ull;
        unsigned slot = hash64(kk) & (RRF_TABLE - 1);
        while (tkey[slot] != kk) slot = (slot + 1) & (RRF_TABLE - 1);
        if (tfirst[slot] != i) continue;
        double s = 0.0;
        bool started = false;
        // (this loop and the two below are chains of dependent LDS round trips: no early exits, so that the compiler can keep
        // several reads in flight - one query's fusion is a single workgroup and its latency is all there is)
#pragma unroll 8
        for (int u = i; u < T; ++u) {
            if (item[u] == key) {
                const int rank = (u % len) + 1;
                const double c = 1.0 / (double)(rrf_k + rank);
                s = started ? s + c : c;
                started = true;
            }
        }
        const int o = atomicAdd(&n_owner, 1);
        oscore[o] = s;
        opos[o] = i;
    }
    __syncthreads();
    // 3. order owners by (score desc, first-seen asc) and write the top_k
    const int U = n_owner;
    for (int o = tid; o < U; o += 256) {
        const double s = oscore[o];
        const int p = opos[o];
        int rank = 0;
#pragma unroll 8
        for (int u = 0; u < U; ++u) rank += (oscore[u] > s) || (oscore[u] == s && opos[u] < p);
        if (rank < top_k) {
            const int64_t key = item[p];
            keys_out[(size_t)q * top_k + rank] = key;
            scores_out[(size_t)q * top_k + rank] = s;
            if (ranks_out) {
                for (int l = 0; l < L; ++l) {
                    int r = 0;
#pragma unroll 8
                    for (int j = len - 1; j >= 0; --j)
                        if (item[l * len + j] == key) r = j + 1;          // the lowest matching position wins
                    ranks_out[((size_t)q * top_k + rank) * L + l] = r;
                }
            }
        }
    }
}

int rrf_fuse_dev(rag_ctx* h, const int64_t* lists_dev, int Q, int L, int len, int64_t list_stride, int64_t query_stride, int rrf_k,
                 int top_k, int64_t* keys_dev, double* scores_dev, int32_t* ranks_dev, hipStream_t st) {
    ARG_CHECK(h, Q > 0 && L > 0 && len >= 0 && top_k > 0, "rrf: sizes must be positive");
    ARG_CHECK(h, (int64_t)L * len <= RRF_MAX_ITEMS, "rrf: n_lists*list_len must be <= 1024");
    ARG_CHECK(h, lists_dev && keys_dev && scores_dev, "rrf: null pointer");
    if (len == 0) len = 1, list_stride = 0;       // nothing to fuse: every item reads as padding below
    hipLaunchKernelGGL(rrf_fuse_kernel, dim3(Q), dim3(256), 0, st, lists_dev, L, len, list_stride, query_stride, rrf_k, top_k,
                       keys_dev, scores_dev, ranks_dev);
    HIP_TRY(h, hipGetLastError());
    return RAG_OK;
}

int rrf_fuse_host(rag_ctx* h, const int64_t* lists, int Q, int L, int len, int rrf_k, int top_k, int64_t* keys_out,
                  double* scores_out, int32_t* ranks_out) {
    ARG_CHECK(h, Q > 0 && L > 0 && len >= 0 && top_k > 0, "rrf: sizes must be positive");
    ARG_CHECK(h, (int64_t)L * len <= RRF_MAX_ITEMS, "rrf: n_lists*list_len must be <= 1024");
    ARG_CHECK(h, lists && keys_out && scores_out, "rrf: null pointer");
    hipStream_t st = h->stream;
    const size_t T = (size_t)L * len, n_out = (size_t)Q * top_k;
    const int rc = stage_reserve(h, stage_size(std::max<size_t>(1, (size_t)Q * T), 8) + 2 * stage_size(n_out, 8) + stage_size(n_out * L, 4));
    if (rc) return rc;
    char* p = (char*)h->stage;
    int64_t* ld = stage_take<int64_t>(p, std::max<size_t>(1, (size_t)Q * T));
    int64_t* kd = stage_take<int64_t>(p, n_out);
    double* sd = stage_take<double>(p, n_out);
    int32_t* rd = ranks_out ? stage_take<int32_t>(p, n_out * L) : nullptr;
    if (T) HIP_TRY(h, hipMemcpyAsync(ld, lists, (size_t)Q * T * sizeof(int64_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rrf_fuse_kernel, dim3(Q), dim3(256), 0, st, ld, L, std::max(len, 1), (int64_t)len, (int64_t)T, rrf_k, top_k,
                       kd, sd, rd);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(keys_out, kd, n_out * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(scores_out, sd, n_out * sizeof(double), hipMemcpyDeviceToHost, st));
    if (ranks_out) HIP_TRY(h, hipMemcpyAsync(ranks_out, rd, n_out * L * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return RAG_OK;
}

// ------------------------------------------------------------------------------------------------
// generic exact top-k over n float64 scores, order (score desc, index asc): chunk sort + merge.
// ------------------------------------------------------------------------------------------------
#define TK_CHUNK 2048
__device__ __forceinline__ bool pair_before_f(uint64_t ka, uint32_t ra, uint64_t kb, uint32_t rb) {
    return ka > kb || (ka == kb && ra < rb);
}
__device__ __forceinline__ void sort_pairs_desc(uint64_t* k1, uint32_t* k2, int P, int tid, int nthreads) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += nthreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool first_block = ((i & k) == 0);
                    const bool a_before_b = pair_before_f(k1[i], k2[i], k1[ixj], k2[ixj]);
                    if (first_block ? !a_before_b : a_before_b) {
                        const uint64_t t1 = k1[i]; k1[i] = k1[ixj]; k1[ixj] = t1;
                        const uint32_t t2 = k2[i]; k2[i] = k2[ixj]; k2[ixj] = t2;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// hybrid[i] = (alpha*sem[i] + beta*kw[i]) + gamma*tmp[i]; per-chunk top-k partials. key 0 = empty slot.
__global__ __launch_bounds__(256) void linear_fuse_chunk_kernel(const double* __restrict__ sem, const double* __restrict__ kw,
                                                                 const double* __restrict__ tmp, int n, double alpha,
                                                                 double beta, double gamma, int k, double* __restrict__ hyb,
                                                                 uint64_t* __restrict__ part_key, uint32_t* __restrict__ part_idx) {
    __shared__ uint64_t sk[TK_CHUNK];
    __shared__ uint32_t si[TK_CHUNK];
    const int chunk = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < TK_CHUNK; j += 256) {
        const int i = chunk * TK_CHUNK + j;
        uint64_t key = 0ull;
        if (i < n) {
            const double t = tmp ? tmp[i] : 0.0;
            const double v = (alpha * sem[i] + beta * kw[i]) + gamma * t;
            hyb[i] = v;
            // NaN never compares greater in Python's sort; map it below everything real but above "empty"
            key = (v != v) ? 1ull : f64_orderable(v);
        }
        sk[j] = key;
        si[j] = (uint32_t)i;
    }
    __syncthreads();
    sort_pairs_desc(sk, si, TK_CHUNK, tid, 256);
    for (int j = tid; j < k; j += 256) {
        part_key[(size_t)chunk * k + j] = sk[j];
        part_idx[(size_t)chunk * k + j] = si[j];
    }
}

__global__ __launch_bounds__(256) void topk_merge_kernel(const uint64_t* __restrict__ part_key, const uint32_t* __restrict__ part_idx,
                                                          int total, int k, int32_t* __restrict__ idx_out) {
    __shared__ uint64_t sk[TK_CHUNK];
    __shared__ uint32_t si[TK_CHUNK];
    const int tid = threadIdx.x;
    for (int i = tid; i < TK_CHUNK; i += 256) { sk[i] = 0ull; si[i] = 0xFFFFFFFFu; }
    __syncthreads();
    int pos = 0;
    while (pos < total) {
        const int room = TK_CHUNK - k;
        const int take = min(room, total - pos);
        for (int i = tid; i < room; i += 256) {
            sk[k + i] = i < take ? part_key[pos + i] : 0ull;
            si[k + i] = i < take ? part_idx[pos + i] : 0xFFFFFFFFu;
        }
        __syncthreads();
        sort_pairs_desc(sk, si, TK_CHUNK, tid, 256);
        pos += take;
    }
    for (int i = tid; i < k; i += 256) idx_out[i] = sk[i] != 0ull ? (int32_t)si[i] : -1;
}

int linear_fuse_topk_host(rag_ctx* h, const double* sem, const double* kw, const double* tmp, int n, double a, double b,
                          double g, int top_k, int32_t* idx_out, double* hyb_out) {
    ARG_CHECK(h, n > 0 && top_k > 0 && top_k <= n, "linear_fuse: need 0 < top_k <= n");
    ARG_CHECK(h, top_k <= TK_CHUNK / 2, "linear_fuse: top_k <= 1024");
    ARG_CHECK(h, sem && kw && idx_out && hyb_out, "linear_fuse: null pointer");
    hipStream_t st = h->stream;
    const int n_chunks = (n + TK_CHUNK - 1) / TK_CHUNK;
    const size_t nb = (size_t)n * sizeof(double), n_part = (size_t)n_chunks * top_k;
    const int rc = stage_reserve(h, 4 * stage_size(n, 8) + stage_size(n_part, 8) + stage_size(n_part, 4) + stage_size(top_k, 4));
    if (rc) return rc;
    char* p = (char*)h->stage;
    double* sd = stage_take<double>(p, n);
    double* kd = stage_take<double>(p, n);
    double* td = stage_take<double>(p, n);
    double* hd = stage_take<double>(p, n);
    uint64_t* pk = stage_take<uint64_t>(p, n_part);
    uint32_t* pi = stage_take<uint32_t>(p, n_part);
    int32_t* od = stage_take<int32_t>(p, top_k);
    if (!tmp) td = nullptr;
    HIP_TRY(h, hipMemcpyAsync(sd, sem, nb, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(kd, kw, nb, hipMemcpyHostToDevice, st));
    if (tmp) HIP_TRY(h, hipMemcpyAsync(td, tmp, nb, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(linear_fuse_chunk_kernel, dim3(n_chunks), dim3(256), 0, st, sd, kd, td, n, a, b, g, top_k, hd, pk, pi);
    hipLaunchKernelGGL(topk_merge_kernel, dim3(1), dim3(256), 0, st, pk, pi, n_chunks * top_k, top_k, od);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(idx_out, od, (size_t)top_k * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(hyb_out, hd, nb, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return RAG_OK;
}

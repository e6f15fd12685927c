// Retrieve + rerank as ONE device-resident call (BASELINE.json configs[3]: "hybrid top-100 -> ms-marco-MiniLM-L-6
// cross-encoder rerank -> top-20"): the composition the reference performs across
//   HybridRetriever.retrieve            /root/reference/rag/retrieval.py:122-212
//   CrossEncoderReranker.rerank         /root/reference/rag/reranker.py:320-384   (pairs [query, content] :346-352,
//                                        raw logits :355, sigmoid :359, sort desc by sigmoid, [:top_k] :372-376)
// with the passages' token ids resident in HBM next to their embeddings ("token store"), so
// between the query arriving and the top-k leaving nothing crosses PCIe.
//   1. candidates: dense top-pool (mode 0) or dense + BM25 + RRF -> top-pool (mode 1)
//   2. ce_build_pairs_kernel: [CLS] query [SEP] passage [SEP], token_type 0 | 1, truncated 'longest_first' to L_pair
//   3. ce_score (cross_encoder.hip)
//   4. rerank_topk_kernel: sigmoid in float64, stable order (score desc, candidate order on ties), top-k
#include "common.h"

int dense_search(rag_ctx* h, const float* q_dev, int Q, int k, int tenant, int64_t* ids_dev, int32_t* rows_dev, double* scores_dev,
                 hipStream_t st);
int bm25_topk_dev(rag_ctx* h, const int32_t* term_ptr_dev, const int32_t* terms_dev, int Q, int k, int tenant, int64_t* ids_dev,
                  int32_t* rows_dev, double* scores_dev, double* raw_max_dev, hipStream_t st);
int64_t bm25_n_docs(const rag_ctx* h);
int rrf_fuse_dev(rag_ctx* h, const int64_t* lists_dev, int Q, int L, int len, int64_t list_stride, int64_t query_stride, int rrf_k,
                 int top_k, int64_t* keys_dev, double* scores_dev, int32_t* ranks_dev, hipStream_t st);

// The two candidate legs of the hybrid search: dense top-pool into lists[0], BM25 top-pool into lists[1] ([2][Q][pool]).
// They are independent: the BM25 leg runs on a side stream, forked from and joined back into the caller's stream by events.
// For a handful of queries both legs are latency-bound (one query: 0.6 ms of HBM-bound scan beside 0.2 ms of BM25 launches that
// occupy a quarter of the CUs: 0.99 -> 0.80 ms). Large batches (r2 kept them in line) gain less - the dense GEMM's 128-KiB
// workgroups leave room for ONE 20-KiB BM25 workgroup per CU - but the BM25 stages fill the thin opening stages, selects and
// the rescoring of the dense leg: 1024-query hybrid batches 6.85 -> 6.58 ms on one box (A/B by option fork_max_q).
// Option no_fork keeps the legs in line always; fork_max_q lowers the largest batch that forks.
#define RAG_FORK_MAX_Q 4096
int hybrid_legs(rag_ctx* h, const float* q_dev, const int32_t* term_ptr_dev, const int32_t* terms_dev, int Q, int pool, int tenant,
                int64_t* lists_dev, double* scores_ws_dev, hipStream_t st) {
    int64_t* const bm_ids = lists_dev + (size_t)Q * pool;
    const int fork_max = h->opt.fork_max_q > 0 ? h->opt.fork_max_q : RAG_FORK_MAX_Q;
    if (Q > fork_max || h->opt.no_fork) {
        int rc = dense_search(h, q_dev, Q, pool, tenant, lists_dev, nullptr, scores_ws_dev, st);
        if (rc) return rc;
        return bm25_topk_dev(h, term_ptr_dev, terms_dev, Q, pool, tenant, bm_ids, nullptr, scores_ws_dev, nullptr, st);
    }
    if (!h->side_stream) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    const size_t need = (size_t)Q * pool;
    if (need > h->side_scores_n) {
        hipFree(h->side_scores);
        h->side_scores = nullptr;
        h->side_scores_n = 0;
        const size_t want = std::max(need, (size_t)RAG_FORK_MAX_Q * RAG_MAX_K);
        HIP_TRY(h, hipMalloc(&h->side_scores, want * sizeof(double)));
        h->side_scores_n = want;
    }
    HIP_TRY(h, hipEventRecord(h->ev_fork, st));                    // inputs are ready wherever the caller's stream is now
    HIP_TRY(h, hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
    int rc = bm25_topk_dev(h, term_ptr_dev, terms_dev, Q, pool, tenant, bm_ids, nullptr, h->side_scores, nullptr, h->side_stream);
    // the join is recorded and waited for even if a leg failed to launch: the caller's stream must never run ahead of the side stream
    HIP_TRY(h, hipEventRecord(h->ev_join, h->side_stream));
    const int rc2 = dense_search(h, q_dev, Q, pool, tenant, lists_dev, nullptr, scores_ws_dev, st);
    HIP_TRY(h, hipStreamWaitEvent(st, h->ev_join, 0));
    return rc ? rc : rc2;
}

int ce_score(rag_ctx* h, const int32_t* ids, const int32_t* tt, const int32_t* lens, int P, int L, float* out, hipStream_t st,
             bool host_ptrs);

void pipeline_free(rag_ctx* h) {
    hipFree(h->tok); hipFree(h->tok_len); hipFree(h->pipe_ws); hipFree(h->tok_bad);
    h->tok = nullptr; h->tok_len = nullptr; h->pipe_ws = nullptr; h->tok_bad = nullptr;
    h->tok_rows = 0; h->tok_cap = 0; h->tok_L = 0; h->pipe_ws_bytes = 0;
}

// int32 token ids -> the 16-bit resident store (WordPiece vocabularies have < 65536 entries: 30522 for the MiniLM
// checkpoints); ids outside [0, 65535] are flagged
__global__ void tokens_narrow_kernel(const int32_t* __restrict__ in, uint16_t* __restrict__ out, int64_t n, int* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t v = in[i];
    if (v < 0 || v > 65535) atomicAdd(bad, 1);
    out[i] = (uint16_t)v;
}

// The token store is kept as uint16: 2 B per token (51 GB for 100M x 256-token passages replicated per GPU, the budget of
// SURVEY.md section 8e; the r1 int32 store would have been 102 GB). The ABI takes int32 ids; they are narrowed on the
// device while streaming in (64 Mi tokens per piece through the staging arena).
int tokens_load_host(rag_ctx* h, const int32_t* tokens, const int32_t* lens, int64_t n_rows, int L) {
    ARG_CHECK(h, tokens && lens && n_rows > 0 && L > 0 && L <= 512, "tokens_load: bad arguments (passage length <= 512)");
    hipFree(h->tok); hipFree(h->tok_len);
    h->tok = nullptr; h->tok_len = nullptr; h->tok_rows = 0; h->tok_cap = 0;
    const int64_t total = n_rows * (int64_t)L, piece = (int64_t)64 << 20;
    HIP_TRY(h, hipMalloc(&h->tok, (size_t)total * sizeof(uint16_t)));
    HIP_TRY(h, hipMalloc(&h->tok_len, (size_t)n_rows * sizeof(int32_t)));
    int rc = stage_reserve(h, stage_size((size_t)std::min(total, piece), 4) + 256);
    if (rc) return rc;
    int32_t* buf = reinterpret_cast<int32_t*>(h->stage);
    int* bad = reinterpret_cast<int*>(reinterpret_cast<char*>(h->stage) + stage_size((size_t)std::min(total, piece), 4));
    HIP_TRY(h, hipMemsetAsync(bad, 0, sizeof(int), h->stream));
    for (int64_t o = 0; o < total; o += piece) {
        const int64_t nn = std::min(piece, total - o);
        HIP_TRY(h, hipMemcpyAsync(buf, tokens + o, (size_t)nn * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(tokens_narrow_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, h->stream, buf, h->tok + o, nn, bad);
        HIP_TRY(h, hipStreamSynchronize(h->stream));                 // buf is reused by the next piece
    }
    int n_bad = 0;
    HIP_TRY(h, hipMemcpy(&n_bad, bad, sizeof(int), hipMemcpyDeviceToHost));
    ARG_CHECK(h, n_bad == 0, "tokens_load: token ids must be in [0, 65535]");
    HIP_TRY(h, hipMemcpy(h->tok_len, lens, (size_t)n_rows * sizeof(int32_t), hipMemcpyHostToDevice));
    h->tok_rows = n_rows;
    h->tok_cap = n_rows;
    h->tok_L = L;
    return RAG_OK;
}

// Chunked fill from device memory (a replicated 100M-passage store is 45 GB as uint16 and would be 90 GB as one int32 host
// array): reserve once, append row blocks in order. tok_rows counts the rows appended so far; tok_cap the reservation.
int tokens_reserve(rag_ctx* h, int64_t n_rows, int L) {
    ARG_CHECK(h, n_rows > 0 && L > 0 && L <= 512, "tokens_reserve: bad arguments (passage length <= 512)");
    hipFree(h->tok); hipFree(h->tok_len);
    h->tok = nullptr; h->tok_len = nullptr; h->tok_rows = 0; h->tok_cap = 0;
    HIP_TRY(h, hipMalloc(&h->tok, (size_t)n_rows * L * sizeof(uint16_t)));
    HIP_TRY(h, hipMalloc(&h->tok_len, (size_t)n_rows * sizeof(int32_t)));
    if (!h->tok_bad) HIP_TRY(h, hipMalloc(&h->tok_bad, sizeof(int)));
    HIP_TRY(h, hipMemset(h->tok_bad, 0, sizeof(int)));
    h->tok_cap = n_rows;
    h->tok_L = L;
    return RAG_OK;
}

int tokens_append_dev(rag_ctx* h, const int32_t* tokens_dev, const int32_t* lens_dev, int64_t n, hipStream_t st) {
    ARG_CHECK(h, h->tok_cap > 0, "tokens_append: rag_tokens_reserve first");
    ARG_CHECK(h, n >= 0 && h->tok_rows + n <= h->tok_cap && (n == 0 || (tokens_dev && lens_dev)), "tokens_append: exceeds the reservation");
    if (n == 0) return RAG_OK;
    const int64_t total = n * (int64_t)h->tok_L;
    hipLaunchKernelGGL(tokens_narrow_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, tokens_dev,
                       h->tok + (size_t)h->tok_rows * h->tok_L, total, h->tok_bad);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(h->tok_len + h->tok_rows, lens_dev, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    int n_bad = 0;                                                   // synchronous check: a load path, not a search path
    HIP_TRY(h, hipMemcpyAsync(&n_bad, h->tok_bad, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    ARG_CHECK(h, n_bad == 0, "tokens_append: token ids must be in [0, 65535]");
    h->tok_rows += n;
    return RAG_OK;
}

// one wave per pair: cand[q][j] is a doc id (id_base + row) or -1
__global__ __launch_bounds__(256) void ce_build_pairs_kernel(const int32_t* __restrict__ q_tok, const int32_t* __restrict__ q_len,
                                                              int Lq, const int64_t* __restrict__ cand, int64_t id_base,
                                                              const uint16_t* __restrict__ tok, const int32_t* __restrict__ tok_len,
                                                              int Ld, int64_t n_rows, int n_pairs, int pool, int L, int cls_id,
                                                              int sep_id, int32_t* __restrict__ ids, int32_t* __restrict__ tt,
                                                              int32_t* __restrict__ lens) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= n_pairs) return;
    const int q = p / pool;
    const int64_t row = cand[p] < 0 ? -1 : cand[p] - id_base;
    // truncation = 'longest_first' to max_length L, as CrossEncoder.predict tokenises its pairs (SURVEY.md section 8c;
    // the fast tokenizer's rule, pinned against the `tokenizers` package by tests/test_pair_truncation.py): with
    // M = L - 3 content tokens and n1 <= n2 the two lengths, the shorter side is kept whole while it fits
    // (n2 = max(n1, M - n1)); if both exceed their share, n1 = M / 2 and n2 = n1 + M % 2.
    int ql = max(0, min(q_len[q], Lq));
    int dl = (row >= 0 && row < n_rows) ? max(0, min(tok_len[row], Ld)) : 0;
    const int M = L - 3;
    if (ql + dl > M) {
        const bool swap = ql > dl;
        int n1 = swap ? dl : ql, n2 = swap ? ql : dl;
        n2 = n1 > M ? n1 : max(n1, M - n1);
        if (n1 + n2 > M) { n1 = M / 2; n2 = n1 + M % 2; }
        ql = swap ? n2 : n1;
        dl = swap ? n1 : n2;
    }
    const int total = ql + dl + 3;
    const int32_t* qt = q_tok + (size_t)q * Lq;
    const uint16_t* dt = tok + (size_t)(row < 0 ? 0 : row) * Ld;
    for (int t = lane; t < L; t += 64) {
        int v = 0, ty = 0;
        if (t == 0) v = cls_id;
        else if (t <= ql) v = qt[t - 1];
        else if (t == ql + 1) v = sep_id;
        else if (t < ql + 2 + dl) { v = dt[t - ql - 2]; ty = 1; }
        else if (t == ql + 2 + dl) { v = sep_id; ty = 1; }
        ids[(size_t)p * L + t] = v;
        tt[(size_t)p * L + t] = ty;
    }
    if (lane == 0) lens[p] = total;
}

// one workgroup per query, pool <= 256: stable rank by (sigmoid desc, candidate position asc)
__global__ __launch_bounds__(256) void rerank_topk_kernel(const float* __restrict__ logits, const int64_t* __restrict__ cand, int pool,
                                                           int k, int64_t* __restrict__ ids_out, double* __restrict__ score_out,
                                                           float* __restrict__ logit_out) {
    __shared__ double s[256];
    __shared__ int ok[256];
    const int q = blockIdx.x, j = threadIdx.x;
    double mine = -1.0;
    int valid = 0;
    if (j < pool) {
        valid = cand[(size_t)q * pool + j] >= 0;
        if (valid) mine = 1.0 / (1.0 + exp(-(double)logits[(size_t)q * pool + j]));          // reranker.py:359
    }
    s[j] = mine;
    ok[j] = valid;
    __syncthreads();
    for (int i = j; i < k; i += 256) { ids_out[(size_t)q * k + i] = -1; score_out[(size_t)q * k + i] = 0.0; logit_out[(size_t)q * k + i] = 0.f; }
    __syncthreads();
    if (j < pool && valid) {
        int rank = 0;
        for (int i = 0; i < pool; ++i) rank += ok[i] && (s[i] > mine || (s[i] == mine && i < j));
        if (rank < k) {
            ids_out[(size_t)q * k + rank] = cand[(size_t)q * pool + j];
            score_out[(size_t)q * k + rank] = mine;
            logit_out[(size_t)q * k + rank] = logits[(size_t)q * pool + j];
        }
    }
}

// The two pipeline kernels on their own, for the row-sharded composition (sharded.py::ShardedPipeline): candidates are
// GLOBAL doc ids from the merged lists; the token store is replicated (token_id_base = id of its first row).
int ce_build_pairs_dev(rag_ctx* h, const int32_t* q_tok_dev, const int32_t* q_len_dev, int Lq, const int64_t* cand_dev, int Q, int pool,
                       int64_t token_id_base, int L_pair, int cls_id, int sep_id, int32_t* ids_out, int32_t* tt_out, int32_t* lens_out,
                       hipStream_t st) {
    ARG_CHECK(h, h->tok != nullptr, "build_pairs: no token store loaded");
    ARG_CHECK(h, Q > 0 && pool > 0 && Lq > 0 && L_pair >= 8 && L_pair <= 512 && q_tok_dev && q_len_dev && cand_dev && ids_out && tt_out && lens_out,
              "build_pairs: bad arguments");
    const size_t P = (size_t)Q * pool;
    hipLaunchKernelGGL(ce_build_pairs_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, st, q_tok_dev, q_len_dev, Lq, cand_dev, token_id_base,
                       h->tok, h->tok_len, h->tok_L, h->tok_rows, (int)P, pool, L_pair, cls_id, sep_id, ids_out, tt_out, lens_out);
    HIP_TRY(h, hipGetLastError());
    return RAG_OK;
}

int rerank_topk_dev(rag_ctx* h, const float* logits_dev, const int64_t* cand_dev, int Q, int pool, int k, int64_t* ids_out, double* scores_out,
                    float* logits_out, hipStream_t st) {
    ARG_CHECK(h, Q > 0 && pool > 0 && pool <= RAG_MAX_K && k > 0 && k <= pool && logits_dev && cand_dev && ids_out && scores_out && logits_out,
              "rerank_topk: 0 < k <= pool <= 256");
    hipLaunchKernelGGL(rerank_topk_kernel, dim3(Q), dim3(256), 0, st, logits_dev, cand_dev, pool, k, ids_out, scores_out, logits_out);
    HIP_TRY(h, hipGetLastError());
    return RAG_OK;
}

__global__ void map_rows_to_ids_kernel(int64_t* __restrict__ v, int64_t n, const int64_t* __restrict__ idmap, int64_t id_base) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || v[i] < 0) return;
    v[i] = idmap ? idmap[v[i]] : id_base + v[i];
}

int retrieve_rerank_dev(rag_ctx* h, const float* q_emb_dev, const int32_t* term_ptr_dev, const int32_t* terms_dev,
                        const int32_t* q_tok_dev, const int32_t* q_len_dev, int Lq, int Q, int pool, int k, int rrf_k, int tenant,
                        int mode, int cls_id, int sep_id, int L_pair, int64_t* ids_out, double* scores_out, float* logits_out,
                        int64_t* cand_out, hipStream_t st) {
    ARG_CHECK(h, h->ce != nullptr, "retrieve_rerank: no cross-encoder loaded");
    ARG_CHECK(h, h->tok != nullptr && h->tok_rows == h->n_rows, "retrieve_rerank: token store missing or not row-aligned with the index");
    ARG_CHECK(h, Q > 0 && pool > 0 && pool <= RAG_MAX_K && k > 0 && k <= pool, "retrieve_rerank: 0 < k <= pool <= 256");
    ARG_CHECK(h, L_pair >= 8 && L_pair <= 512 && Lq > 0 && q_emb_dev && q_tok_dev && q_len_dev && ids_out && scores_out && logits_out,
              "retrieve_rerank: bad arguments");
    ARG_CHECK(h, mode == 0 || (mode == 1 && term_ptr_dev && h->bm25 != nullptr), "retrieve_rerank: mode 1 needs BM25 postings and query terms");
    ARG_CHECK(h, mode == 0 || bm25_n_docs(h) == h->n_rows, "retrieve_rerank: the BM25 postings must be row-aligned with the index");
    const size_t P = (size_t)Q * pool;
    // workspace: lists [2][Q][pool] i64 | scores [Q][pool] f64 | cand [Q][pool] i64 | rrf [Q][pool] f64 | ranks [Q][pool][2] i32
    //            | pair ids [P][L] | pair tt [P][L] | lens [P] | logits [P]
    const size_t need = P * (16 + 8 + 8 + 8 + 8) + P * L_pair * 8 + P * 8 + 256;
    if (need > h->pipe_ws_bytes) {
        hipFree(h->pipe_ws);
        h->pipe_ws = nullptr;
        h->pipe_ws_bytes = 0;
        HIP_TRY(h, hipMalloc(&h->pipe_ws, need));
        h->pipe_ws_bytes = need;
    }
    char* w = (char*)h->pipe_ws;
    int64_t* lists = (int64_t*)w;              w += P * 16;
    double* sc = (double*)w;                   w += P * 8;
    int64_t* cand = (int64_t*)w;               w += P * 8;
    double* rrf = (double*)w;                  w += P * 8;
    int32_t* ranks = (int32_t*)w;              w += P * 8;
    int32_t* pid = (int32_t*)w;                w += P * L_pair * 4;
    int32_t* ptt = (int32_t*)w;                w += P * L_pair * 4;
    int32_t* plen = (int32_t*)w;               w += P * 4;
    float* logit = (float*)w;
    // The candidate stage runs in ROW space (the token store is row-aligned and RRF only needs a consistent key space):
    // the id mapping is switched off for these launches (pointers are kernel arguments, captured at launch) and applied to
    // the outputs at the end, so explicit doc ids (e.g. primary keys of a loaded shard) work as well as id_base + row.
    int64_t* const ids_saved = h->ids;
    const int64_t id_base_saved = h->id_base;
    h->ids = nullptr;
    h->id_base = 0;
    int rc;
    if (mode == 0) {
        rc = dense_search(h, q_emb_dev, Q, pool, tenant, cand, nullptr, sc, st);
    } else {
        rc = hybrid_legs(h, q_emb_dev, term_ptr_dev, terms_dev, Q, pool, tenant, lists, sc, st);
        if (!rc) rc = rrf_fuse_dev(h, lists, Q, 2, pool, (int64_t)P, pool, rrf_k, pool, cand, rrf, ranks, st);
    }
    h->ids = ids_saved;
    h->id_base = id_base_saved;
    if (rc) return rc;
    hipLaunchKernelGGL(ce_build_pairs_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, st, q_tok_dev, q_len_dev, Lq, cand, (int64_t)0,
                       h->tok, h->tok_len, h->tok_L, h->tok_rows, (int)P, pool, L_pair, cls_id, sep_id, pid, ptt, plen);
    HIP_TRY(h, hipGetLastError());
    rc = ce_score(h, pid, ptt, plen, (int)P, L_pair, logit, st, false);
    if (rc) return rc;
    hipLaunchKernelGGL(rerank_topk_kernel, dim3(Q), dim3(256), 0, st, logit, cand, pool, k, ids_out, scores_out, logits_out);
    HIP_TRY(h, hipGetLastError());
    if (cand_out) HIP_TRY(h, hipMemcpyAsync(cand_out, cand, P * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    if (ids_saved != nullptr || id_base_saved != 0) {
        hipLaunchKernelGGL(map_rows_to_ids_kernel, dim3((unsigned)(((size_t)Q * k + 255) / 256)), dim3(256), 0, st, ids_out, (int64_t)Q * k,
                           ids_saved, id_base_saved);
        if (cand_out)
            hipLaunchKernelGGL(map_rows_to_ids_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, cand_out, (int64_t)P, ids_saved,
                               id_base_saved);
        HIP_TRY(h, hipGetLastError());
    }
    return RAG_OK;
}

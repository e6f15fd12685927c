// C-ABI entry points of librag_hip.so (include/rag_hip.h). Thin: argument checks, host<->device
// staging for the *_host variants, dispatch to the kernels' host drivers.
#include "common.h"

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>

// drivers implemented in other translation units
int rrf_fuse_host(rag_ctx* h, const int64_t* lists, int Q, int L, int len, int rrf_k, int top_k, int64_t* keys_out,
                  double* scores_out, int32_t* ranks_out);
int linear_fuse_topk_host(rag_ctx* h, const double* sem, const double* kw, const double* tmp, int n, double a, double b,
                          double g, int top_k, int32_t* idx_out, double* hyb_out);
int bm25_load_host(rag_ctx* h, const int64_t* indptr, const int32_t* doc, const int32_t* tf, const int32_t* doc_len,
                   const double* idf, int64_t n_docs, int64_t n_terms, double avgdl, double k1, double b);
int bm25_topk_host(rag_ctx* h, const int32_t* term_ptr, const int32_t* terms, int Q, int k, int tenant, int64_t* ids_out,
                   int32_t* rows_out, double* scores_out, double* raw_max_out);
int bm25_scores_adhoc_host(rag_ctx* h, const int64_t* indptr, const int32_t* doc, const int32_t* tf, const int32_t* doc_len,
                           const double* idf, int64_t n_docs, int64_t n_terms, double avgdl, double k1, double b,
                           const int32_t* term_ptr, const int32_t* terms, int Q, double* out);
int64_t bm25_n_docs(const rag_ctx* h);
int bm25_index_bytes(const int64_t* indptr, int64_t n_docs, int64_t n_terms, int64_t* postings_out, int64_t* meta_out, int64_t* table_out);
int bm25_grid_plan(int n_ranges_in_launch, int n_queries, int linear, int64_t* out5);
int bm25_scores_dev(rag_ctx* h, const int32_t* term_ptr_dev, const int32_t* terms_dev, int Q, double* out_dev, hipStream_t st,
                    float* raw32_dev, int64_t ld, unsigned long long* max_key_dev, int tenant);
int bm25_scores_host(rag_ctx* h, const int32_t* term_ptr, const int32_t* terms, int Q, double* out);
int bm25_topk_dev(rag_ctx* h, const int32_t* term_ptr_dev, const int32_t* terms_dev, int Q, int k, int tenant, int64_t* ids_dev,
                  int32_t* rows_dev, double* scores_dev, double* raw_max_dev, hipStream_t st);
int rrf_fuse_dev(rag_ctx* h, const int64_t* lists_dev, int Q, int L, int len, int64_t list_stride, int64_t query_stride, int rrf_k,
                 int top_k, int64_t* keys_dev, double* scores_dev, int32_t* ranks_dev, hipStream_t st);
void bm25_free(rag_ctx* h);
int bm25_set_normalize(rag_ctx* h, int on);
int chunk_chain_host(rag_ctx* h, const float* emb, const int32_t* sent_len, int n, int dim, double threshold, int max_chunk,
                     int min_chunk, int32_t* group_out);
int mmr_select_dev(rag_ctx* h, const float* queries_dev, const float* emb_dev, const int32_t* rows_dev, int Q, int n, int dim,
                   int top_k, double lam, int variant, int32_t* sel_dev, double* score_dev, hipStream_t st);
int mmr_select_host(rag_ctx* h, const float* query, const float* emb, int n, int dim, int top_k, double lam, int variant,
                    int32_t* sel_out, double* score_out);
int ce_load_host(rag_ctx* h, const rag_ce_config* cfg, const float* const* tensors, int n);
int ce_score(rag_ctx* h, const int32_t* ids, const int32_t* tt, const int32_t* lens, int P, int L, float* out,
             hipStream_t st, bool host_ptrs);
void ce_free(rag_ctx* h);
int embed_load_host(rag_ctx* h, const rag_ce_config* cfg, const float* const* tensors, int n, int normalize);
int embed_run(rag_ctx* h, const int32_t* ids, const int32_t* tt, const int32_t* lens, int P, int L, float* out, hipStream_t st, bool host_ptrs);
int embed_dim(const rag_ctx* h);
void pipeline_free(rag_ctx* h);
int ce_build_pairs_dev(rag_ctx* h, const int32_t* q_tok_dev, const int32_t* q_len_dev, int Lq, const int64_t* cand_dev, int Q, int pool,
                       int64_t token_id_base, int L_pair, int cls_id, int sep_id, int32_t* ids_out, int32_t* tt_out, int32_t* lens_out,
                       hipStream_t st);
int rerank_topk_dev(rag_ctx* h, const float* logits_dev, const int64_t* cand_dev, int Q, int pool, int k, int64_t* ids_out, double* scores_out,
                    float* logits_out, hipStream_t st);
int tokens_load_host(rag_ctx* h, const int32_t* tokens, const int32_t* lens, int64_t n_rows, int L);
int tokens_reserve(rag_ctx* h, int64_t n_rows, int L);
int tokens_append_dev(rag_ctx* h, const int32_t* tokens_dev, const int32_t* lens_dev, int64_t n_rows, hipStream_t st);
int retrieve_rerank_dev(rag_ctx* h, const float* q_emb_dev, const int32_t* term_ptr_dev, const int32_t* terms_dev,
                        const int32_t* q_tok_dev, const int32_t* q_len_dev, int Lq, int Q, int pool, int k, int rrf_k, int tenant,
                        int mode, int cls_id, int sep_id, int L_pair, int64_t* ids_out, double* scores_out, float* logits_out,
                        int64_t* cand_out, hipStream_t st);

static thread_local std::string g_null_err = "null handle";

// name -> member of rag_options; the environment default of option <name> is RAG_<NAME>
static const struct { const char* name; int rag_options::*field; } g_options[] = {
    {"force_level", &rag_options::force_level},         {"stage_growth", &rag_options::stage_growth},
    {"no_smallq", &rag_options::no_smallq},             {"no_second_pass", &rag_options::no_second_pass},
    {"dense_linear_order", &rag_options::dense_linear_order}, {"dense_persist", &rag_options::dense_persist},
    {"bm25_first_ranges", &rag_options::bm25_first_ranges}, {"bm25_no_staging", &rag_options::bm25_no_staging},
    {"bm25_packed", &rag_options::bm25_packed},         {"bm25_plan_slots", &rag_options::bm25_plan_slots},       {"bm25_ws_mb", &rag_options::bm25_ws_mb},
    {"bm25_linear_grid", &rag_options::bm25_linear_grid},            {"bm25_sort_merge", &rag_options::bm25_sort_merge},
    {"no_fork", &rag_options::no_fork},                 {"fork_max_q", &rag_options::fork_max_q},                 {"ce_no_fused_ln", &rag_options::ce_no_fused_ln},
    {"ce_no_fused_ffn", &rag_options::ce_no_fused_ffn}, {"ce_chunk_tokens", &rag_options::ce_chunk_tokens}, {"ce_mx", &rag_options::ce_mx},
};
static void options_from_env(rag_options* o) {
    for (const auto& e : g_options) {
        std::string env = "RAG_";
        for (const char* c = e.name; *c; ++c) env += (char)toupper(*c);
        if (const char* v = getenv(env.c_str())) o->*(e.field) = atoi(v);
    }
}
#define LOCK(h) std::lock_guard<std::mutex> lock_((h)->mu)

template <class T>
static int pairwise_cosine_host_t(rag_handle_t h, const T* a, int m, const T* b, int n, int dim, double* out) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, m >= 0 && n >= 0 && dim > 0, "bad sizes");
    if (m == 0 || n == 0) return RAG_OK;
    ARG_CHECK(h, a && b && out, "null pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    const bool same = (a == b && m == n);
    hipStream_t st = h->stream;
    int rc = stage_reserve(h, stage_size((size_t)m * dim, sizeof(T)) + stage_size(same ? 0 : (size_t)n * dim, sizeof(T)) + stage_size((size_t)m * n, 8));
    if (rc) return rc;
    char* p = (char*)h->stage;
    T* ad = stage_take<T>(p, (size_t)m * dim);
    T* bd = same ? ad : stage_take<T>(p, (size_t)n * dim);
    double* od = stage_take<double>(p, (size_t)m * n);
    HIP_TRY(h, hipMemcpyAsync(ad, a, (size_t)m * dim * sizeof(T), hipMemcpyHostToDevice, st));
    if (!same) HIP_TRY(h, hipMemcpyAsync(bd, b, (size_t)n * dim * sizeof(T), hipMemcpyHostToDevice, st));
    if constexpr (sizeof(T) == 8) rc = pairwise_cosine_f64(h, ad, m, bd, n, dim, od, st);
    else rc = pairwise_cosine(h, ad, m, bd, n, dim, od, st);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(out, od, (size_t)m * n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return RAG_OK;
}

extern "C" {

int rag_version(void) { return 200; }

int rag_device_count(int* n_out) {
    if (!n_out) return RAG_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        *n_out = 0;
        return RAG_ERR_HIP;
    }
    *n_out = n;
    return RAG_OK;
}

int rag_create(int device_id, int dim, rag_handle_t* out) {
    if (!out || dim <= 0 || dim % 4 != 0) return RAG_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0 || device_id < 0 || device_id >= n) return RAG_ERR_HIP;
    if (hipSetDevice(device_id) != hipSuccess) return RAG_ERR_HIP;
    rag_ctx* h = new rag_ctx();
    h->device = device_id;
    h->dim = dim;
    h->dim_pad = (int)round_up(dim, RAG_BK);
    options_from_env(&h->opt);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return RAG_ERR_HIP;
    }
    *out = h;
    return RAG_OK;
}

int rag_destroy(rag_handle_t h) {
    if (!h) return RAG_ERR_ARG;
    { LOCK(h); }                        // wait for a call in flight on another thread; the caller must not start new ones
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    comm_free(h);
    dense_free(h);
    bm25_free(h);
    ce_free(h);
    pipeline_free(h);
    hipFree(h->q32); hipFree(h->q16); hipFree(h->cand); hipFree(h->cnt); hipFree(h->tau); hipFree(h->bound);
    hipFree(h->n_sorted); hipFree(h->exact); hipFree(h->flag); hipFree(h->scan_list); hipFree(h->stats); hipFree(h->stage);
    hipFree(h->temporal); hipFree(h->lin_ws);
    hipFree(h->q16b); hipFree(h->candb); hipFree(h->cntb); hipFree(h->taub); hipFree(h->boundb); hipFree(h->n_sortedb); hipFree(h->ovf_list);
    for (auto& p : h->prof)
        for (auto& e : p.ev) {
            hipEventDestroy(e.first);
            hipEventDestroy(e.second);
        }
    if (h->side_stream) { hipStreamSynchronize(h->side_stream); hipStreamDestroy(h->side_stream); hipEventDestroy(h->ev_fork); hipEventDestroy(h->ev_join); }
    hipFree(h->side_scores);
    hipStreamDestroy(h->stream);
    delete h;
    return RAG_OK;
}

const char* rag_last_error(rag_handle_t h) { return h ? h->err.c_str() : g_null_err.c_str(); }

int rag_synchronize(rag_handle_t h) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return RAG_OK;
}

int rag_set_option(rag_handle_t h, const char* name, int value) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, name != nullptr, "set_option: null name");
    for (const auto& e : g_options)
        if (strcmp(e.name, name) == 0) {
            h->opt.*(e.field) = value;
            return RAG_OK;
        }
    h->err = std::string("bad argument: unknown option ") + name;
    return RAG_ERR_ARG;
}

int rag_set_profiling(rag_handle_t h, int enable) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    h->profiling = enable != 0;
    for (auto& p : h->prof) p.used = 0;      // (re)start collecting spans
    return RAG_OK;
}

// ---- dense index ---------------------------------------------------------------------------------
static int index_load_common(rag_ctx* h, const float* emb, const int64_t* ids, int64_t id_base, int64_t n_rows,
                             hipStream_t st, bool host) {
    ARG_CHECK(h, n_rows >= 0 && n_rows < (int64_t)0x7fffff00, "n_rows must fit int32 per GPU");
    ARG_CHECK(h, n_rows == 0 || emb != nullptr, "emb is null");
    HIP_TRY(h, hipSetDevice(h->device));
    dense_free(h);
    h->n_rows = n_rows;
    h->id_base = id_base;
    const hipMemcpyKind kind = host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    if (n_rows > 0) {
        HIP_TRY(h, hipMalloc(&h->emb32, (size_t)n_rows * h->dim * sizeof(float)));
        HIP_TRY(h, hipMemcpyAsync(h->emb32, emb, (size_t)n_rows * h->dim * sizeof(float), kind, st));
        if (ids) {
            HIP_TRY(h, hipMalloc(&h->ids, (size_t)n_rows * sizeof(int64_t)));
            HIP_TRY(h, hipMemcpyAsync(h->ids, ids, (size_t)n_rows * sizeof(int64_t), kind, st));
        }
    }
    int rc = dense_index_build(h, h->emb32, n_rows, st);
    if (rc) return rc;
    h->index_loaded = true;
    if (host) HIP_TRY(h, hipStreamSynchronize(st));
    return RAG_OK;
}

int rag_index_load_host(rag_handle_t h, const float* emb_host, const int64_t* ids_host, int64_t id_base, int64_t n_rows) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    return index_load_common(h, emb_host, ids_host, id_base, n_rows, h->stream, true);
}

int rag_index_load_dev(rag_handle_t h, const float* emb_dev, const int64_t* ids_dev, int64_t id_base, int64_t n_rows,
                       void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    return index_load_common(h, emb_dev, ids_dev, id_base, n_rows, (hipStream_t)stream, false);
}

// ---- chunked bulk load (export of document_chunks / archival_memory in pieces) -------------------
int rag_index_reserve(rag_handle_t h, int64_t n_rows_total, int64_t id_base) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, n_rows_total > 0 && n_rows_total < (int64_t)0x7fffff00, "n_rows must fit int32 per GPU");
    HIP_TRY(h, hipSetDevice(h->device));
    dense_free(h);
    h->id_base = id_base;
    h->n_reserved = n_rows_total;
    h->n_rows_pad = round_up(n_rows_total, (int64_t)RAG_TILE * 8);
    HIP_TRY(h, hipMalloc(&h->emb32, (size_t)n_rows_total * h->dim * sizeof(float)));
    HIP_TRY(h, hipMalloc(&h->emb16, (size_t)h->n_rows_pad * h->dim_pad * sizeof(half_t)));
    HIP_TRY(h, hipMalloc(&h->bad_rows, sizeof(int)));
    HIP_TRY(h, hipMemsetAsync(h->bad_rows, 0, sizeof(int), h->stream));
    // rows not appended yet (and the tile padding) must read as zero vectors
    HIP_TRY(h, hipMemsetAsync(h->emb16, 0, (size_t)h->n_rows_pad * h->dim_pad * sizeof(half_t), h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->index_loaded = true;
    return RAG_OK;
}

static int index_append(rag_ctx* h, const float* emb, int64_t n, hipStream_t st, bool host) {
    ARG_CHECK(h, h->n_reserved > 0, "rag_index_reserve first");
    ARG_CHECK(h, n >= 0 && h->n_rows + n <= h->n_reserved, "append exceeds the reserved row count");
    ARG_CHECK(h, n == 0 || emb != nullptr, "emb is null");
    HIP_TRY(h, hipSetDevice(h->device));
    if (n == 0) return RAG_OK;
    HIP_TRY(h, hipMemcpyAsync(h->emb32 + (size_t)h->n_rows * h->dim, emb, (size_t)n * h->dim * sizeof(float),
                              host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
    int rc = dense_index_normalize_range(h, h->n_rows, n, st);
    if (rc) return rc;
    h->n_rows += n;
    if (host) HIP_TRY(h, hipStreamSynchronize(st));
    return RAG_OK;
}

int rag_index_append_host(rag_handle_t h, const float* emb_host, int64_t n_rows) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    return index_append(h, emb_host, n_rows, h->stream, true);
}

int rag_index_append_dev(rag_handle_t h, const float* emb_dev, int64_t n_rows, void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    return index_append(h, emb_dev, n_rows, (hipStream_t)stream, false);
}

int rag_index_set_tenants_host(rag_handle_t h, const int32_t* t, int64_t n_rows) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, t == nullptr || n_rows == h->n_rows, "tenant array length must equal the index row count");
    HIP_TRY(h, hipSetDevice(h->device));
    hipFree(h->tenants);
    h->tenants = nullptr;
    if (t == nullptr || n_rows == 0) return dense_build_tenant_tiles(h, nullptr, 0);      // NULL clears the filter table
    HIP_TRY(h, hipMalloc(&h->tenants, (size_t)n_rows * sizeof(int32_t)));
    HIP_TRY(h, hipMemcpy(h->tenants, t, (size_t)n_rows * sizeof(int32_t), hipMemcpyHostToDevice));
    return dense_build_tenant_tiles(h, t, n_rows);
}

int rag_index_set_ids_host(rag_handle_t h, const int64_t* ids, int64_t n_rows) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, n_rows == h->n_rows, "id array length must equal the index row count");
    HIP_TRY(h, hipSetDevice(h->device));
    hipFree(h->ids);
    h->ids = nullptr;
    if (ids == nullptr || n_rows == 0) return RAG_OK;
    HIP_TRY(h, hipMalloc(&h->ids, (size_t)n_rows * sizeof(int64_t)));
    HIP_TRY(h, hipMemcpy(h->ids, ids, (size_t)n_rows * sizeof(int64_t), hipMemcpyHostToDevice));
    return RAG_OK;
}

int rag_index_rows(rag_handle_t h, int64_t* n_rows_out) {
    if (!h || !n_rows_out) return RAG_ERR_ARG;
    LOCK(h);
    *n_rows_out = h->n_rows;
    return RAG_OK;
}

int rag_index_fetch_rows_host(rag_handle_t h, const int64_t* rows, int n, float* out) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, n >= 0 && (n == 0 || (rows && out)), "null rows/out");
    HIP_TRY(h, hipSetDevice(h->device));
    for (int i = 0; i < n; ++i) {
        ARG_CHECK(h, rows[i] >= 0 && rows[i] < h->n_rows, "row out of range");
        HIP_TRY(h, hipMemcpyAsync(out + (size_t)i * h->dim, h->emb32 + (size_t)rows[i] * h->dim, h->dim * sizeof(float),
                                  hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return RAG_OK;
}

// ---- dense search ----------------------------------------------------------------------------------
int rag_dense_topk_dev(rag_handle_t h, const float* q_dev, int Q, int k, int tenant, int64_t* ids_dev, int32_t* rows_dev,
                       double* scores_dev, void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, q_dev && ids_dev && scores_dev, "null pointer");
    ARG_CHECK(h, Q > 0 && Q <= 65535, "1 <= n_queries <= 65535");
    HIP_TRY(h, hipSetDevice(h->device));
    return dense_search(h, q_dev, Q, k, tenant, ids_dev, rows_dev, scores_dev, (hipStream_t)stream);
}

int rag_dense_topk_host(rag_handle_t h, const float* q_host, int Q, int k, int tenant, int64_t* ids_out, int32_t* rows_out,
                        double* scores_out) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, q_host && ids_out && scores_out, "null pointer");
    ARG_CHECK(h, Q > 0 && Q <= 65535, "1 <= n_queries <= 65535");
    ARG_CHECK(h, k > 0 && k <= RAG_MAX_K, "0 < k <= 256");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const size_t n_out = (size_t)Q * k;
    int rc = stage_reserve(h, stage_size((size_t)Q * h->dim, 4) + 2 * stage_size(n_out, 8) + stage_size(n_out, 4));
    if (rc) return rc;
    char* p = (char*)h->stage;
    float* qd = stage_take<float>(p, (size_t)Q * h->dim);
    int64_t* ids_d = stage_take<int64_t>(p, n_out);
    double* sc_d = stage_take<double>(p, n_out);
    int32_t* rows_d = stage_take<int32_t>(p, n_out);
    HIP_TRY(h, hipMemcpyAsync(qd, q_host, (size_t)Q * h->dim * sizeof(float), hipMemcpyHostToDevice, st));
    rc = dense_search(h, qd, Q, k, tenant, ids_d, rows_d, sc_d, st);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(ids_out, ids_d, n_out * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    if (rows_out) HIP_TRY(h, hipMemcpyAsync(rows_out, rows_d, n_out * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(scores_out, sc_d, n_out * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return RAG_OK;
}

int rag_dense_last_stats(rag_handle_t h, rag_dense_stats* out) {
    if (!h || !out) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, h->last_stats_valid && h->stats, "no dense search has run");
    int s[8];
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(s, h->stats, sizeof(s), hipMemcpyDeviceToHost));
    out->n_queries = h->last_q;
    out->proven_fast = s[0];
    out->proven_wide = s[1];
    out->exact_scan = s[2];
    out->overflowed = s[4];
    out->shortlist = h->last_shortlist;
    out->stages = h->last_stages;
    out->second_pass = s[5];
    out->eps = h->last_eps;
    return RAG_OK;
}

int rag_stage_kernel_ms(rag_handle_t h, int stage, float* ms_out, int* spans_out) {
    if (!h || !ms_out || !spans_out) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, stage >= 0 && stage < RAG_PROF_STAGES, "stage must be 0 (dense emit), 1 (bm25) or 2 (cross-encoder)");
    auto& p = h->prof[stage];
    ARG_CHECK(h, p.used > 0, "no span recorded for this stage (rag_set_profiling(h, 1) before the calls)");
    HIP_TRY(h, hipSetDevice(h->device));
    float total = 0.f;
    for (int i = 0; i < p.used; ++i) {
        HIP_TRY(h, hipEventSynchronize(p.ev[i].second));
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, p.ev[i].first, p.ev[i].second));
        total += ms;
    }
    *ms_out = total;
    *spans_out = p.used;
    return RAG_OK;
}

int rag_dense_kernel_ms(rag_handle_t h, float* gemm_ms_out, int* launches_out) {
    return rag_stage_kernel_ms(h, 0, gemm_ms_out, launches_out);
}

int rag_merge_topk_dev(rag_handle_t h, const int64_t* ids_dev, const double* scores_dev, int n_lists, int64_t list_stride,
                       int Q, int k, int64_t* ids_out_dev, double* scores_out_dev, void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, ids_dev && scores_dev && ids_out_dev && scores_out_dev, "null pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    return merge_topk(h, ids_dev, scores_dev, n_lists, list_stride, Q, k, ids_out_dev, scores_out_dev,
                      (hipStream_t)stream);
}

// The fuse step of the row-sharded hybrid search on every rank, after the ONE all-gather of the shards' lists.
int rag_hybrid_fuse_gathered_dev(rag_handle_t h, const int64_t* gathered_dev, int world, int Q, int pool, int k, int rrf_k,
                                 int64_t* lists_out_dev, double* scores_out_dev, int64_t* keys_out_dev, double* rrf_out_dev,
                                 int32_t* ranks_out_dev, void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, gathered_dev && lists_out_dev && scores_out_dev && keys_out_dev && rrf_out_dev, "hybrid_fuse_gathered: null pointer");
    ARG_CHECK(h, world > 0 && Q > 0 && pool > 0 && pool <= RAG_MAX_K && k > 0, "hybrid_fuse_gathered: bad sizes (0 < pool <= 256)");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t P = (int64_t)Q * pool, stride = 4 * P;
    const double* gs = reinterpret_cast<const double*>(gathered_dev);
    int rc = merge_topk(h, gathered_dev, gs + P, world, stride, Q, pool, lists_out_dev, scores_out_dev, st, 0);
    if (rc) return rc;
    if ((rc = merge_topk(h, gathered_dev + 2 * P, gs + 3 * P, world, stride, Q, pool, lists_out_dev + P, scores_out_dev + P, st, 1))) return rc;
    return rrf_fuse_dev(h, lists_out_dev, Q, 2, pool, P, pool, rrf_k, k, keys_out_dev, rrf_out_dev, ranks_out_dev, st);
}

int rag_pairwise_cosine_host(rag_handle_t h, const float* a, int m, const float* b, int n, int dim, double* out) {
    return pairwise_cosine_host_t<float>(h, a, m, b, n, dim, out);
}

int rag_pairwise_cosine_f64_host(rag_handle_t h, const double* a, int m, const double* b, int n, int dim, double* out) {
    return pairwise_cosine_host_t<double>(h, a, m, b, n, dim, out);
}

int rag_rrf_fuse_host(rag_handle_t h, const int64_t* lists, int Q, int L, int len, int rrf_k, int top_k, int64_t* keys_out,
                      double* scores_out, int32_t* ranks_out) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return rrf_fuse_host(h, lists, Q, L, len, rrf_k, top_k, keys_out, scores_out, ranks_out);
}

int rag_linear_fuse_topk_host(rag_handle_t h, const double* sem, const double* kw, const double* tmp, int n, double a,
                              double b, double g, int top_k, int32_t* idx_out, double* hyb_out) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return linear_fuse_topk_host(h, sem, kw, tmp, n, a, b, g, top_k, idx_out, hyb_out);
}

int rag_bm25_load_host(rag_handle_t h, const int64_t* indptr, const int32_t* doc, const int32_t* tf, const int32_t* doc_len,
                       const double* idf, int64_t n_docs, int64_t n_terms, double avgdl, double k1, double b) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return bm25_load_host(h, indptr, doc, tf, doc_len, idf, n_docs, n_terms, avgdl, k1, b);
}

int rag_bm25_grid_plan(int n_ranges_in_launch, int n_queries, int linear, int64_t* out5) {
    return bm25_grid_plan(n_ranges_in_launch, n_queries, linear, out5);
}

int rag_bm25_index_bytes(const int64_t* indptr_host, int64_t n_docs, int64_t n_terms, int64_t* postings_bytes_out,
                         int64_t* meta_bytes_out, int64_t* table_bytes_out) {
    return bm25_index_bytes(indptr_host, n_docs, n_terms, postings_bytes_out, meta_bytes_out, table_bytes_out);
}

int rag_bm25_topk_host(rag_handle_t h, const int32_t* term_ptr, const int32_t* terms, int Q, int k, int tenant, int64_t* ids_out,
                       int32_t* rows_out, double* scores_out, double* raw_max_out) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return bm25_topk_host(h, term_ptr, terms, Q, k, tenant, ids_out, rows_out, scores_out, raw_max_out);
}

int rag_bm25_topk_dev(rag_handle_t h, const int32_t* term_ptr_dev, const int32_t* terms_dev, int Q, int k, int tenant, int64_t* ids_dev,
                      int32_t* rows_dev, double* scores_dev, double* raw_max_dev, void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return bm25_topk_dev(h, term_ptr_dev, terms_dev, Q, k, tenant, ids_dev, rows_dev, scores_dev, raw_max_dev,
                         (hipStream_t)stream);
}

int rag_rrf_fuse_dev(rag_handle_t h, const int64_t* lists_dev, int Q, int L, int len, int rrf_k, int top_k, int64_t* keys_dev,
                     double* scores_dev, int32_t* ranks_dev, void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return rrf_fuse_dev(h, lists_dev, Q, L, len, len, (int64_t)L * len, rrf_k, top_k, keys_dev, scores_dev, ranks_dev,
                        (hipStream_t)stream);
}

// dense top-pool + BM25 top-pool + RRF -> top-k, all on the device, one call (BASELINE.json configs[2]).
// lists_ws_dev: caller scratch [2][Q][pool] int64 (receives the dense, then the BM25 ranked id list);
// scores_ws_dev: caller scratch [Q][pool] float64.
int rag_hybrid_rrf_dev(rag_handle_t h, const float* q_dev, const int32_t* term_ptr_dev, const int32_t* terms_dev, int Q, int pool,
                       int k, int rrf_k, int tenant, int64_t* lists_ws_dev, double* scores_ws_dev, int64_t* keys_out_dev,
                       double* rrf_out_dev, int32_t* ranks_out_dev, void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, q_dev && term_ptr_dev && lists_ws_dev && scores_ws_dev && keys_out_dev && rrf_out_dev, "hybrid: null pointer");
    ARG_CHECK(h, pool > 0 && pool <= RAG_MAX_K && k > 0, "hybrid: 0 < pool <= 256");
    ARG_CHECK(h, bm25_n_docs(h) == h->n_rows, "hybrid: the BM25 postings must be row-aligned with the index (same number of documents)");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    int rc = hybrid_legs(h, q_dev, term_ptr_dev, terms_dev, Q, pool, tenant, lists_ws_dev, scores_ws_dev, st);
    if (rc) return rc;
    return rrf_fuse_dev(h, lists_ws_dev, Q, 2, pool, (int64_t)Q * pool, pool, rrf_k, k, keys_out_dev, rrf_out_dev, ranks_out_dev, st);
}

// Per-row temporal score of the linear fusion: RECENCY_WEIGHT * 0.5 ** (days_old / half_life) computed by the host from
// created_at / uploaded_at exactly as rag/retrieval.py:266-292 does (shard_format.temporal_scores). NULL clears it (zeros).
int rag_index_set_temporal_host(rag_handle_t h, const double* temporal, int64_t n_rows) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, temporal == nullptr || n_rows == h->n_rows, "temporal array length must equal the index row count");
    HIP_TRY(h, hipSetDevice(h->device));
    hipFree(h->temporal);
    h->temporal = nullptr;
    h->temporal_absmax = 0.0;
    if (temporal == nullptr || n_rows == 0) return RAG_OK;
    for (int64_t i = 0; i < n_rows; ++i) h->temporal_absmax = std::max(h->temporal_absmax, std::fabs(temporal[i]));
    ARG_CHECK(h, std::isfinite(h->temporal_absmax), "temporal scores must be finite");
    HIP_TRY(h, hipMalloc(&h->temporal, (size_t)n_rows * sizeof(double)));
    HIP_TRY(h, hipMemcpy(h->temporal, temporal, (size_t)n_rows * sizeof(double), hipMemcpyHostToDevice));
    return RAG_OK;
}

// HybridRetriever.hybrid_search (rag/retrieval.py:214-322) over the WHOLE resident index instead of a host-side corpus list:
// per query, hybrid = (alpha * cosine + beta * keyword) + gamma * temporal for every row (keyword = BM25Okapi score / max over
// the corpus, 1.0 when that max is <= 0; temporal from rag_index_set_temporal_host), stable sort descending, first k.
// The weights are the caller's (intent table or constructor defaults, :232-238). Queries are processed 256 at a time: the
// all-document BM25 scores of a sub-batch (float64, 2 GB at 1M rows) and their float32 emission bias stay in a workspace.
int rag_hybrid_linear_dev(rag_handle_t h, const float* q_dev, const int32_t* term_ptr_dev, const int32_t* terms_dev, int Q, int k,
                          double alpha, double beta, double gamma, int tenant, int64_t* ids_out_dev, int32_t* rows_out_dev,
                          double* hybrid_out_dev, double* semantic_out_dev, double* keyword_out_dev, double* temporal_out_dev,
                          void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, q_dev && term_ptr_dev && ids_out_dev && rows_out_dev && hybrid_out_dev, "hybrid_linear: null pointer");
    ARG_CHECK(h, Q > 0 && k > 0 && k <= RAG_MAX_K, "hybrid_linear: need Q > 0 and 0 < k <= 256");
    ARG_CHECK(h, alpha > 0.0, "hybrid_linear: alpha must be positive (the cosine drives the candidate search)");
    ARG_CHECK(h, std::isfinite(alpha) && std::isfinite(beta) && std::isfinite(gamma), "hybrid_linear: weights must be finite");
    ARG_CHECK(h, tenant < 0 || h->tenants != nullptr, "hybrid_linear: tenant filter requested but no tenants loaded");
    ARG_CHECK(h, bm25_n_docs(h) == h->n_rows && h->n_rows > 0, "hybrid_linear: needs BM25 postings row-aligned with a non-empty index");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = h->n_rows, ld = h->n_rows_pad;
    // queries per sub-batch: one query tile when the all-document scores fit 8 GB (2 GB + 1 GB at 1M rows), fewer on large
    // shards (12.5M rows: 53 queries, 8 GB instead of 38 GB)
    const int QB = (int)std::max<int64_t>(16, std::min<int64_t>(256, ((int64_t)8 << 30) / (n * 12)));
    const size_t need = stage_size((size_t)QB * n, 8) + stage_size((size_t)QB * ld, 4) + 2 * stage_size(QB, 8) + stage_size(QB, 4) + stage_size(ld, 4);
    if (need > h->lin_ws_bytes) {
        hipFree(h->lin_ws);
        h->lin_ws = nullptr;
        h->lin_ws_bytes = 0;
        HIP_TRY(h, hipMalloc(&h->lin_ws, need));
        h->lin_ws_bytes = need;
        HIP_TRY(h, hipMemsetAsync(h->lin_ws, 0, need, st));      // the pad rows [n, ld) of raw32 are read by the last tile: keep them 0
    }
    char* p = (char*)h->lin_ws;
    double* raw = stage_take<double>(p, (size_t)QB * n);
    float* raw32 = stage_take<float>(p, (size_t)QB * ld);
    double* mx = stage_take<double>(p, QB);
    unsigned long long* max_key = stage_take<unsigned long long>(p, QB);
    float* qscale = stage_take<float>(p, QB);
    float* gt = (h->temporal != nullptr && gamma != 0.0) ? stage_take<float>(p, ld) : nullptr;
    for (int q0 = 0; q0 < Q; q0 += QB) {
        const int qc = std::min(QB, Q - q0);
        // ONE pass of the scoring kernel writes the float64 scores (exact fusion of the survivors), their float32 copy (the emission
        // operand) and the per-query maximum; r3 re-read the 2 GB twice more (a one-workgroup-per-query max: 1.8 ms of a 4.5-ms call;
        // a bias pass: 0.6 ms)
        HIP_TRY(h, hipMemsetAsync(max_key, 0, (size_t)qc * sizeof(unsigned long long), st));
        int rc = bm25_scores_dev(h, term_ptr_dev + q0, terms_dev, qc, raw, st, raw32, ld, max_key, tenant);
        if (rc) return rc;
        if ((rc = linear_prepare(h, max_key, qc, n, h->temporal, beta, gamma, mx, qscale, q0 == 0 ? gt : nullptr, ld, st))) return rc;
        const dense_fused fz = {raw32, ld, qscale, gt, raw, n, mx, h->temporal, alpha, beta, gamma};
        rc = dense_search_fused(h, q_dev + (size_t)q0 * h->dim, qc, k, tenant, ids_out_dev + (size_t)q0 * k, rows_out_dev + (size_t)q0 * k,
                                hybrid_out_dev + (size_t)q0 * k, st, &fz);
        if (rc) return rc;
        if (semantic_out_dev || keyword_out_dev || temporal_out_dev) {
            rc = linear_components(h, q_dev + (size_t)q0 * h->dim, rows_out_dev + (size_t)q0 * k, qc, k, &fz,
                                   semantic_out_dev ? semantic_out_dev + (size_t)q0 * k : nullptr,
                                   keyword_out_dev ? keyword_out_dev + (size_t)q0 * k : nullptr,
                                   temporal_out_dev ? temporal_out_dev + (size_t)q0 * k : nullptr, st);
            if (rc) return rc;
        }
    }
    return RAG_OK;
}

int rag_bm25_set_normalize(rag_handle_t h, int on) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    return bm25_set_normalize(h, on);
}

int rag_bm25_scores_host(rag_handle_t h, const int32_t* term_ptr, const int32_t* terms, int Q, double* out) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return bm25_scores_host(h, term_ptr, terms, Q, out);
}

int rag_bm25_scores_adhoc_host(rag_handle_t h, const int64_t* indptr, const int32_t* doc, const int32_t* tf, const int32_t* doc_len,
                               const double* idf, int64_t n_docs, int64_t n_terms, double avgdl, double k1, double b,
                               const int32_t* term_ptr, const int32_t* terms, int Q, double* out) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return bm25_scores_adhoc_host(h, indptr, doc, tf, doc_len, idf, n_docs, n_terms, avgdl, k1, b, term_ptr, terms, Q, out);
}

int rag_ce_load_host(rag_handle_t h, const rag_ce_config* cfg, const float* const* tensors, int n) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return ce_load_host(h, cfg, tensors, n);
}

int rag_ce_score_host(rag_handle_t h, const int32_t* ids, const int32_t* tt, const int32_t* lens, int P, int L, float* out) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return ce_score(h, ids, tt, lens, P, L, out, h->stream, true);
}

int rag_ce_score_dev(rag_handle_t h, const int32_t* ids, const int32_t* tt, const int32_t* lens, int P, int L, float* out,
                     void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return ce_score(h, ids, tt, lens, P, L, out, (hipStream_t)stream, false);
}

int rag_embed_load_host(rag_handle_t h, const rag_ce_config* cfg, const float* const* tensors, int n, int normalize) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return embed_load_host(h, cfg, tensors, n, normalize);
}

int rag_embed_host(rag_handle_t h, const int32_t* ids, const int32_t* tt, const int32_t* lens, int n_texts, int L, float* out) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return embed_run(h, ids, tt, lens, n_texts, L, out, h->stream, true);
}

int rag_embed_dev(rag_handle_t h, const int32_t* ids, const int32_t* tt, const int32_t* lens, int n_texts, int L, float* out, void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return embed_run(h, ids, tt, lens, n_texts, L, out, (hipStream_t)stream, false);
}

int rag_embed_dim(rag_handle_t h, int* dim_out) {
    if (!h || !dim_out) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, embed_dim(h) > 0, "no embedding model loaded");
    *dim_out = embed_dim(h);
    return RAG_OK;
}

int rag_mmr_select_host(rag_handle_t h, const float* query, const float* emb, int n, int dim, int top_k, double lambda,
                        int variant, int32_t* sel_out, double* score_out) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return mmr_select_host(h, query, emb, n, dim, top_k, lambda, variant, sel_out, score_out);
}

// candidates = rows of the resident index (fp32 master rows): rows_dev[Q][pool], -1 = empty slot
int rag_mmr_select_dev(rag_handle_t h, const float* q_dev, const int32_t* rows_dev, int Q, int pool, int top_k, double lambda,
                       int variant, int32_t* sel_dev, double* score_dev, void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    ARG_CHECK(h, h->emb32 != nullptr && rows_dev, "mmr_select_dev: no index loaded / null rows");
    HIP_TRY(h, hipSetDevice(h->device));
    return mmr_select_dev(h, q_dev, h->emb32, rows_dev, Q, pool, h->dim, top_k, lambda, variant, sel_dev, score_dev,
                          (hipStream_t)stream);
}

int rag_chunk_chain_host(rag_handle_t h, const float* emb, const int32_t* sent_len, int n, int dim, double threshold,
                         int max_chunk, int min_chunk, int32_t* group_out) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return chunk_chain_host(h, emb, sent_len, n, dim, threshold, max_chunk, min_chunk, group_out);
}

int rag_tokens_load_host(rag_handle_t h, const int32_t* tokens, const int32_t* lens, int64_t n_rows, int L) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return tokens_load_host(h, tokens, lens, n_rows, L);
}

int rag_tokens_reserve(rag_handle_t h, int64_t n_rows_total, int L) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return tokens_reserve(h, n_rows_total, L);
}

int rag_tokens_append_dev(rag_handle_t h, const int32_t* tokens_dev, const int32_t* lens_dev, int64_t n_rows, void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return tokens_append_dev(h, tokens_dev, lens_dev, n_rows, (hipStream_t)stream);
}

int rag_retrieve_rerank_dev(rag_handle_t h, const float* q_emb_dev, const int32_t* term_ptr_dev, const int32_t* terms_dev,
                            const int32_t* q_tok_dev, const int32_t* q_len_dev, int Lq, int Q, int pool, int k, int rrf_k,
                            int tenant, int mode, int cls_id, int sep_id, int L_pair, int64_t* ids_out_dev,
                            double* scores_out_dev, float* logits_out_dev, int64_t* cand_out_dev, void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return retrieve_rerank_dev(h, q_emb_dev, term_ptr_dev, terms_dev, q_tok_dev, q_len_dev, Lq, Q, pool, k, rrf_k, tenant, mode,
                               cls_id, sep_id, L_pair, ids_out_dev, scores_out_dev, logits_out_dev, cand_out_dev,
                               (hipStream_t)stream);
}

int rag_ce_build_pairs_dev(rag_handle_t h, const int32_t* q_tok_dev, const int32_t* q_len_dev, int Lq, const int64_t* cand_dev, int Q,
                           int pool, int64_t token_id_base, int L_pair, int cls_id, int sep_id, int32_t* ids_out_dev,
                           int32_t* tt_out_dev, int32_t* lens_out_dev, void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return ce_build_pairs_dev(h, q_tok_dev, q_len_dev, Lq, cand_dev, Q, pool, token_id_base, L_pair, cls_id, sep_id, ids_out_dev,
                              tt_out_dev, lens_out_dev, (hipStream_t)stream);
}

int rag_rerank_topk_dev(rag_handle_t h, const float* logits_dev, const int64_t* cand_dev, int Q, int pool, int k, int64_t* ids_out_dev,
                        double* scores_out_dev, float* logits_out_dev, void* stream) {
    if (!h) return RAG_ERR_ARG;
    LOCK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    return rerank_topk_dev(h, logits_dev, cand_dev, Q, pool, k, ids_out_dev, scores_out_dev, logits_out_dev, (hipStream_t)stream);
}

}  // extern "C"

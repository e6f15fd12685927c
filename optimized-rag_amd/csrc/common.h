// Shared declarations for librag_hip.so (gfx950 only). See include/rag_hip.h for the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rag_hip.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RAG_CAND_CAP 4096        // per-query candidate buffer entries (u64 keys)
#define RAG_STAGE0_ROWS 2048     // rows whose scores are stored densely (no threshold yet)
#define RAG_STAGE_GROWTH 8       // each threshold stage covers ~8x the rows seen so far (RAG_STAGE_GROWTH env overrides, for tuning)
#define RAG_TILE 256             // GEMM tile edge (corpus rows x queries)
#define RAG_BK 64                // K-slice per LDS stage (halfs)
#define RAG_MAX_K 256            // largest k / shortlist supported by the select kernels
#define RAG_SCALE_LOG2 7         // unit rows are stored as fp16(128 * x): keeps fp16 out of subnormals

#define RAG_PROF_STAGES 3
struct rag_ce_model;             // cross_encoder.hip
struct rag_bm25_index;           // bm25.hip

// Diagnostic / tuning switches of one handle. The defaults come from the environment (RAG_<NAME>) ONCE, in rag_create;
// rag_set_option(h, "<name>", value) changes them afterwards (tests flip them between calls on one handle). Nothing on a
// search path calls getenv.
struct rag_options {
    int force_level = 0;          // 1 / 2: every dense query through the wide ranking / the float64 exact scan
    int stage_growth = 0;         // dense stage growth (0 = the built-in schedule)
    int no_smallq = 0;            // disable the small-batch dense kernel variant
    int no_second_pass = 0;       // overflowed dense queries go straight to the float64 scan
    int dense_persist = 0;        // (experiment) thresholded stages as one persistent workgroup per CU instead of one workgroup per tile
    int dense_linear_order = 0;   // walk the tiles in table order (r1 behaviour, for A/B)
    int bm25_first_ranges = 0;    // exact first-stage BM25 ranges (0 = BM_FIRST_RANGES)
    int bm25_no_staging = 0;      // exact per-range select for every BM25 range
    int bm25_linear_grid = 0;     // scoring workgroups range-major (the ranges of one query side by side) instead of the XCD-aware column order
    int bm25_sort_merge = 0;      // fold the partial lists of a stage by the bitonic-sort kernel (r1-r2) instead of the per-wave selection
    int bm25_plan_slots = 0;      // planned token slots per query of a BM25 call (0 = sized by the per-call budget, bm25_pick_plan_t); tests force 8
    int bm25_ws_mb = 0;           // workspace budget of a device-pointer BM25 call in MiB (0 = 6 GiB): batches beyond it run in sub-batches
    int bm25_packed = 0;          // (read when postings are LOADED) 4-byte packed postings + shared impact table instead of (doc, impact)
    int no_fork = 0;              // keep the BM25 leg of a small hybrid batch in line on the caller's stream
    int fork_max_q = 0;           // largest batch whose BM25 leg runs on the side stream beside the dense leg (0 = RAG_FORK_MAX_Q)
    int ce_no_fused_ln = 0;       // unfused residual + LayerNorm path of the cross-encoder
    int ce_no_fused_ffn = 0;      // FFN as two GEMM launches (up-projection, then the fused-LN down-projection)
    int ce_chunk_tokens = 0;      // activation chunk size in tokens (0 = sized from the model)
    int ce_mx = 0;                // cross-encoder forward on hi16 + lo8 operands (ce_mx.h): 0 = from MX_MIN_ROWS padded rows on, 1 = always, -1 = never
};

struct rag_ctx {
    int device = 0;
    int dim = 0;
    int dim_pad = 0;             // multiple of RAG_BK
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;       // hybrid_legs: the BM25 leg of a small batch runs beside the dense leg
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    double* side_scores = nullptr;           // [side_scores_n] score scratch of that leg
    size_t side_scores_n = 0;
    std::string err;
    rag_options opt;
    bool profiling = false;
    void* comm = nullptr;                    // ncclComm_t of comm.hip (RCCL, opened with dlopen), or null
    int comm_rank = 0, comm_world = 1;

    // dense index
    int64_t n_rows = 0, n_rows_pad = 0, id_base = 0;
    bool index_loaded = false;   // rag_index_load_* / rag_index_reserve ran (an EMPTY index is a valid, searchable state)
    int64_t n_reserved = 0;      // rows allocated by rag_index_reserve (chunked bulk load), 0 otherwise
    float* emb32 = nullptr;      // [n_rows][dim]        fp32 master rows
    half_t* emb16 = nullptr;     // [n_rows_pad][dim_pad] fp16 (2^7 * unit rows), zero padded
    int64_t* ids = nullptr;      // [n_rows] or null
    int32_t* tenants = nullptr;  // [n_rows] or null
    double* temporal = nullptr;                                        // [n_rows] per-row temporal score (linear fusion) or null
    double temporal_absmax = 0.0;                                      // max |temporal[i]| (sizes the fused emission margin)
    void* lin_ws = nullptr;                                            // rag_hybrid_linear_dev: raw BM25 + bias + max of one sub-batch
    size_t lin_ws_bytes = 0;
    int32_t* tenant_tiles = nullptr;                                   // concatenated per-tenant lists of 256-row tiles
    std::unordered_map<int32_t, std::pair<int64_t, int>> tenant_span;  // tenant -> (offset, count) into tenant_tiles
    int64_t tenant_rows = 0;                                           // row count the tenant table was built for
    int* bad_rows = nullptr;     // device counter: rows with zero / non-finite norm

    // dense search workspace (sized for ws_q queries)
    int ws_q = 0;
    float* q32 = nullptr;            // [ws_q][dim]     staging for host queries
    half_t* q16 = nullptr;           // [ws_qpad][dim_pad]
    int q16_dirty = 0;               // rows [q16_dirty, ws_qpad) of q16 are known to be zero (pad rows of a query tile must be)
    uint64_t* cand = nullptr;        // [ws_qpad][RAG_CAND_CAP]
    unsigned* cnt = nullptr;         // [ws_qpad]   emitted candidates (may exceed cap = overflow)
    float* tau = nullptr;            // [ws_qpad]   emission threshold = k-th best fp16-pass score so far - 2 eps
    float* bound = nullptr;          // [ws_qpad]   -inf, or +inf once the candidate buffer overflowed (sticky)
    int* n_sorted = nullptr;         // [ws_qpad]   survivors left in cand[] after the final select
    double* exact = nullptr;         // [ws_qpad][RAG_CAND_CAP] float64 rescored cosines
    int* flag = nullptr;             // [ws_qpad]   0 done, 2 needs exact scan, 3 scanned
    int* scan_list = nullptr;        // [ws_qpad]   queries flagged 2 (appended by finalize_kernel; count = stats[7])
    int* stats = nullptr;            // [8] device counters
    // second pass for overflowed queries: one 256-query tile of its own (dense.hip)
    half_t* q16b = nullptr;
    uint64_t* candb = nullptr;
    unsigned* cntb = nullptr;
    float *taub = nullptr, *boundb = nullptr;
    int *n_sortedb = nullptr, *ovf_list = nullptr;
    // exact-scan fallback workspace
    double* scan_scores = nullptr;   // [n_rows] (allocated on first use)
    int64_t scan_rows = 0;
    // grow-only device arena of the synchronous *_host entry points: their per-call staging (queries in, results out, partial
    // lists) is carved from it, so an agent-facing call pays no hipMalloc / hipFree
    void* stage = nullptr;
    size_t stage_bytes = 0;
    // one lock per handle, taken by every entry point: the reference's DocumentStore.search may be called from up to 10
    // threads (database/connection.py:38-42). *_host calls are then fully thread-safe (they are synchronous inside the
    // lock); *_dev calls are serialised while they enqueue and share the handle's workspaces, so they must target ONE stream.
    std::mutex mu;

    // profiling (rag_set_profiling): HIP event pairs recorded on the launch stream around the spans bench.py prices against
    // a roofline. Stage 0 = every dense_emit_kernel<false> launch, 1 = the BM25 range + merge launches of one top-k call,
    // 2 = one cross-encoder forward (all chunks of a rag_ce_score_dev / rag_retrieve_rerank_dev call).
    struct prof_spans { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; int used = 0; };
    prof_spans prof[RAG_PROF_STAGES];
    rag_dense_stats last_stats = {};
    bool last_stats_valid = false;
    int last_q = 0, last_k = 0, last_stages = 0, last_shortlist = 0;
    double last_eps = 0;

    rag_bm25_index* bm25 = nullptr;
    // passage token store (pipeline.hip): [tok_rows][tok_L] uint16 WordPiece ids + lengths, row-aligned with the index
    uint16_t* tok = nullptr;
    int32_t* tok_len = nullptr;
    int64_t tok_rows = 0, tok_cap = 0;       // rows loaded / rows reserved (rag_tokens_reserve + rag_tokens_append_dev)
    int* tok_bad = nullptr;                  // device counter of out-of-range token ids seen by the appends
    int tok_L = 0;
    void* pipe_ws = nullptr;
    size_t pipe_ws_bytes = 0;
    // hipFuncSetAttribute (dynamic LDS above 64 KiB) is per device: remembered per handle, not per process
    bool attr_dense = false, attr_bm25 = false, attr_ce_gemm = false, attr_ce_gemm_ln = false, attr_ce_ffn = false;
    int attr_ce_attn_lds[3] = {0, 0, 0};
    int attr_ce_attn_mx_lds[3] = {0, 0, 0};
    bool attr_ce_mx = false;
    rag_ce_model* ce = nullptr;
    rag_ce_model* emb = nullptr;             // sentence-embedding encoder (rag_embed_load_host): the K7 kernels behind a mean-pooling head
};

#define HIP_TRY(h, expr)                                                                      \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                     \
            return RAG_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

#define ARG_CHECK(h, cond, msg)                                                               \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            (h)->err = std::string("bad argument: ") + (msg);                                 \
            return RAG_ERR_ARG;                                                               \
        }                                                                                     \
    } while (0)

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---- profiling spans (no-ops unless rag_set_profiling(h, 1))
static inline int prof_begin(rag_ctx* h, int stage, hipStream_t st) {
    if (!h->profiling) return RAG_OK;
    auto& p = h->prof[stage];
    if ((int)p.ev.size() <= p.used) {
        hipEvent_t a, b;
        HIP_TRY(h, hipEventCreate(&a));
        HIP_TRY(h, hipEventCreate(&b));
        p.ev.push_back({a, b});
    }
    HIP_TRY(h, hipEventRecord(p.ev[p.used].first, st));
    return RAG_OK;
}
static inline int prof_end(rag_ctx* h, int stage, hipStream_t st) {
    if (!h->profiling) return RAG_OK;
    auto& p = h->prof[stage];
    HIP_TRY(h, hipEventRecord(p.ev[p.used].second, st));
    p.used++;
    return RAG_OK;
}

// ---- staging arena (see rag_ctx::stage): sum stage_size() of every piece, stage_reserve() once, stage_take() in the same order
static inline size_t stage_size(size_t n, size_t elt) { return (size_t)round_up((int64_t)(n * elt), 256); }
static inline int stage_reserve(rag_ctx* h, size_t bytes) {
    if (bytes <= h->stage_bytes) return RAG_OK;
    if (h->stage) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        hipFree(h->stage);
        h->stage = nullptr;
        h->stage_bytes = 0;
    }
    bytes = (size_t)round_up((int64_t)(bytes + bytes / 4), 1 << 20);          // headroom: slightly larger batches do not reallocate
    HIP_TRY(h, hipMalloc(&h->stage, bytes));
    h->stage_bytes = bytes;
    return RAG_OK;
}
template <class T>
static inline T* stage_take(char*& p, size_t n) {
    T* r = reinterpret_cast<T*>(p);
    p += stage_size(n, sizeof(T));
    return r;
}

// ---- sortable keys: larger key = higher score, then lower row ---------------------------------
__host__ __device__ static inline uint32_t f32_orderable(float s) {
    uint32_t u = __builtin_bit_cast(uint32_t, s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ static inline float f32_from_orderable(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __builtin_bit_cast(float, u);
}
__host__ __device__ static inline uint64_t make_key(float score, uint32_t row) {
    return ((uint64_t)f32_orderable(score) << 32) | (uint64_t)(0xFFFFFFFFu - row);
}
__host__ __device__ static inline uint32_t key_row(uint64_t k) { return 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFu); }
__host__ __device__ static inline float key_score(uint64_t k) { return f32_from_orderable((uint32_t)(k >> 32)); }
__host__ __device__ static inline uint64_t f64_orderable(double s) {
    uint64_t u = __builtin_bit_cast(uint64_t, s);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

// entry points implemented per file
int dense_index_build(rag_ctx* h, const float* emb_dev, int64_t n_rows, hipStream_t st);
int dense_index_normalize_range(rag_ctx* h, int64_t first_row, int64_t n_rows, hipStream_t st);
int dense_search(rag_ctx* h, const float* q_dev, int Q, int k, int tenant, int64_t* ids_dev, int32_t* rows_dev,
                 double* scores_dev, hipStream_t st);
// linear fusion inputs of one query sub-batch (rag_hybrid_linear_dev): float32 emission bias [Q][bias_ld], float64 raw BM25
// scores [Q][n] with their per-query divisor, per-row temporal scores (or null), the three weights
struct dense_fused {
    const float* bias; int64_t bias_ld;          // float32 raw BM25 scores [Q][bias_ld] (written by the scoring kernel itself)
    const float* qscale;                         // [Q] float(beta / max of the query)
    const float* gt;                             // [bias_ld] float(gamma * temporal) or null
    const double* raw; int64_t n;
    const double* mx; const double* temporal;
    double alpha, beta, gamma;
};
int dense_search_fused(rag_ctx* h, const float* q_dev, int Q, int k, int tenant, int64_t* ids_dev, int32_t* rows_dev,
                       double* scores_dev, hipStream_t st, const dense_fused* fz);
void comm_free(rag_ctx* h);
int hybrid_legs(rag_ctx* h, const float* q_dev, const int32_t* term_ptr_dev, const int32_t* terms_dev, int Q, int pool, int tenant,
                int64_t* lists_dev, double* scores_ws_dev, hipStream_t st);
int linear_prepare(rag_ctx* h, const unsigned long long* max_key, int Q, int64_t n, const double* temporal, double beta, double gamma, double* mx,
                   float* qscale, float* gt, int64_t ld, hipStream_t st);
int linear_components(rag_ctx* h, const float* q_dev, const int32_t* rows_dev, int Q, int k, const dense_fused* fz, double* sem_out,
                      double* kw_out, double* tmp_out, hipStream_t st);
int dense_free(rag_ctx* h);
int dense_build_tenant_tiles(rag_ctx* h, const int32_t* tenants_host, int64_t n_rows);
int merge_topk(rag_ctx* h, const int64_t* ids, const double* scores, int n_lists, int64_t list_stride, int Q, int k,
               int64_t* ids_out, double* scores_out, hipStream_t st, int normalize = 0);
int pairwise_cosine(rag_ctx* h, const float* a_dev, int m, const float* b_dev, int n, int dim, double* out_dev,
                    hipStream_t st);
int pairwise_cosine_f64(rag_ctx* h, const double* a_dev, int m, const double* b_dev, int n, int dim, double* out_dev, hipStream_t st);

/*
 * rag_hip.h — C-ABI of librag_hip.so: the MI355X (gfx950) hybrid-retrieval + rerank engine.
 *
 * The reference (gabrielcheda/optimized-rag) is pure Python and has NO FFI of its own; the drop-in
 * boundary is its Python object graph. Each entry point below names the reference
 * Python call whose arithmetic it replaces (paths under /root/reference). The Python mirror classes in
 * optimized-rag_amd/ bind these symbols with ctypes and keep the reference's class/method surface.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; rag_last_error(h) gives the message.
 *   - plain pointers and sizes only; the caller owns every buffer; nothing throws across the ABI.
 *   - "_host" pointers are host memory (copied over PCIe inside the call, call is synchronous);
 *     "_dev"  pointers are device memory on the handle's GPU; those calls are asynchronous on `stream`, a
 *     hipStream_t passed as void*. NULL means the DEFAULT (null) stream, exactly as in HIP itself — frameworks
 *     whose current stream is the default stream (PyTorch) pass 0 and get correct ordering with their own work.
 *     The handle's private stream is only used by the synchronous *_host entry points.
 *   - one handle = one GPU = one process rank. Every entry point takes the handle's internal lock: the synchronous
 *     *_host calls may be issued from several threads at once (the reference graph is single-threaded,
 *     agent/rag_graph.py:506, but its DB pool allows 10 concurrent searches, database/connection.py:38-42); they
 *     run one after another. *_dev calls share the handle's device workspaces: issue them on ONE stream.
 *   - doc ids are int64 (SQL BIGSERIAL ids, database/migrations/001_initial_schema.sql); scores are
 *     float64 because the reference computes every score as a Python float.
 */
#ifndef RAG_HIP_H
#define RAG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rag_ctx* rag_handle_t;

#define RAG_OK 0
#define RAG_ERR_ARG (-1)
#define RAG_ERR_HIP (-2)
#define RAG_ERR_STATE (-3)
#define RAG_ERR_NOMEM (-4)

/* ---- lifecycle ---------------------------------------------------------------------------- */
int rag_version(void);
int rag_device_count(int* n_out);
/* dim = embedding dimension (1536 for text-embedding-3-small, memory/embeddings.py:324-325). */
int rag_create(int device_id, int dim, rag_handle_t* out);
int rag_destroy(rag_handle_t h);
const char* rag_last_error(rag_handle_t h);
int rag_synchronize(rag_handle_t h);
/* Diagnostic / tuning switch of one handle (no reference counterpart). Every switch <name> takes its default from the
 * environment variable RAG_<NAME> ONCE, when rag_create runs; afterwards only this call changes it. Names: force_level,
 * stage_growth, no_smallq, no_second_pass, dense_linear_order, bm25_first_ranges, bm25_no_staging, bm25_packed, bm25_linear_grid, bm25_sort_merge, no_fork,
 * fork_max_q, bm25_plan_slots, bm25_ws_mb, ce_no_fused_ln, ce_no_fused_ffn, ce_chunk_tokens, ce_mx. Unknown name: RAG_ERR_ARG. */
int rag_set_option(rag_handle_t h, const char* name, int value);

/* ---- dense index: replaces the pgvector tables behind
 *      `ORDER BY dc.embedding <=> %s::vector LIMIT %s`  (rag/document_store.py:448-460)
 *      `ORDER BY embedding <=> %s::vector LIMIT %s`     (database/operations.py:126-137)
 * Rows are float32 (pgvector `vector(1536)` is float4). ids may be NULL (id = id_base + row).
 * Builds in HBM: fp32 master rows, fp16 unit-normalised rows (MFMA operand). Rows with a zero or
 * non-finite norm score 0.0 against every query (the reference's `return 0.0`, rag/retrieval.py:368-369). */
int rag_index_load_host(rag_handle_t h, const float* emb_host, const int64_t* ids_host, int64_t id_base,
                        int64_t n_rows);
int rag_index_load_dev(rag_handle_t h, const float* emb_dev, const int64_t* ids_dev, int64_t id_base,
                       int64_t n_rows, void* stream);
/* Chunked bulk load for shards that should not exist twice in memory (12.5M rows = 115 GB resident): reserve once,
 * append row blocks in order (host or device source), search at any time over the rows appended so far.
 * The bulk export of `document_chunks(id, agent_id, content, embedding)` (rag/document_store.py:210-221) maps onto
 * this directly. */
int rag_index_reserve(rag_handle_t h, int64_t n_rows_total, int64_t id_base);
int rag_index_append_host(rag_handle_t h, const float* emb_host, int64_t n_rows);
int rag_index_append_dev(rag_handle_t h, const float* emb_dev, int64_t n_rows, void* stream);
/* Optional multi-tenant filter: tenant_of_row[n_rows] (the `WHERE dc.agent_id = %s`,
 * rag/document_store.py:457). tenant < 0 in a search = no filter. */
int rag_index_set_tenants_host(rag_handle_t h, const int32_t* tenant_of_row_host, int64_t n_rows);
/* Explicit doc ids (e.g. the table's primary keys) for an index filled through rag_index_reserve/append: ids[n_rows],
 * n_rows == rows appended so far; NULL restores id = id_base + row. */
int rag_index_set_ids_host(rag_handle_t h, const int64_t* ids_host, int64_t n_rows);
int rag_index_rows(rag_handle_t h, int64_t* n_rows_out);
/* copy rows' fp32 embeddings back (kills apply_mmr's per-doc re-embedding, rag/nodes/helpers.py:215-223) */
int rag_index_fetch_rows_host(rag_handle_t h, const int64_t* rows_host, int n, float* out_host);

/* Exact cosine top-k over the resident index, Q queries at once.
 * Result = what an un-indexed `ORDER BY embedding <=> q LIMIT k` returns: cosine descending, ties by
 * lower row first; ids_out[Q*k] (-1 padded), rows_out[Q*k] local row numbers (may be NULL),
 * scores_out[Q*k] = 1 - distance as float64. Identical id sets to the float64 exact scan hold BY CONSTRUCTION:
 * the fp16 MFMA pass only discards rows whose score is more than 2*eps below the k-th best (eps = proven bound on
 * the fp16 error), survivors are rescored in float64; a float64 scan of every row is the overflow fallback. */
int rag_dense_topk_host(rag_handle_t h, const float* q_host, int n_queries, int k, int tenant,
                        int64_t* ids_out_host, int32_t* rows_out_host, double* scores_out_host);
int rag_dense_topk_dev(rag_handle_t h, const float* q_dev, int n_queries, int k, int tenant,
                       int64_t* ids_out_dev, int32_t* rows_out_dev, double* scores_out_dev, void* stream);

/* Counters of the last dense search (device counters are read back: synchronises). */
typedef struct rag_dense_stats {
    int32_t n_queries;
    int32_t proven_fast;      /* <= 256 survivors above tau: ranked in the fast path (exact by construction) */
    int32_t proven_wide;      /* more survivors (clusters, duplicates): ranked by the wide kernel            */
    int32_t exact_scan;       /* buffer overflowed in the second pass too: float64 scan of every row         */
    int32_t overflowed;       /* overflow events (any stage of the first pass)                               */
    int32_t shortlist;        /* k (the threshold is the k-th best score so far minus 2*eps)                 */
    int32_t stages;           /* threshold stages used                  */
    int32_t second_pass;      /* overflowed queries whose re-emission at the final threshold fitted the buffer */
    double eps;               /* rigorous |fp16 score - exact| bound    */
} rag_dense_stats;
int rag_dense_last_stats(rag_handle_t h, rag_dense_stats* out);

/* Names/launch geometry of the dominant kernel of the last dense search, for bench.py's roofline. */
int rag_dense_kernel_ms(rag_handle_t h, float* gemm_ms_out, int* gemm_launches_out);
int rag_set_profiling(rag_handle_t h, int enable);
/* Measurement hook (bench.py's rooflines; the reference has nothing to mirror): with profiling on, HIP event pairs are
 * recorded on the launch stream around stage 0 = every thresholded dense GEMM launch (== rag_dense_kernel_ms),
 * 1 = the BM25 posting-range + merge launches of each top-k call, 2 = each cross-encoder forward. Returns the summed
 * device time and the number of spans since profiling was (re)enabled; synchronises on the spans' end events. */
int rag_stage_kernel_ms(rag_handle_t h, int stage, float* ms_out, int* spans_out);

/* ---- merge of per-shard partial top-k lists (multi-GPU exchange step).
 * list l lives at ids_dev + l*list_stride / scores_dev + l*list_stride, each [Q][k] (ids int64, scores
 * float64, -1 padded); list_stride is in elements (Q*k when the lists are contiguous, 2*Q*k for an
 * all-gathered [rank][ids|scores][Q][k] buffer). Output [Q][k] by score desc, id asc. */
int rag_merge_topk_dev(rag_handle_t h, const int64_t* ids_dev, const double* scores_dev, int n_lists,
                       int64_t list_stride, int n_queries, int k, int64_t* ids_out_dev, double* scores_out_dev,
                       void* stream);

/* The fuse step of the row-sharded HYBRID search on every rank, after the one all-gather:
 * gathered_dev = [world][4][Q][pool] int64 = every rank's dense ids | dense cosines (float64 bits) | BM25 ids | RAW BM25
 * scores (float64 bits; rag_bm25_set_normalize(h, 0) on the shards). Merges the dense lists and the BM25 lists (score desc,
 * id asc), divides the merged BM25 scores by their GLOBAL maximum (when > 0, else 1.0: rag/retrieval.py:343-345 - the
 * shards cannot know it), then fuses the two MERGED lists with RRF (rag/reranker.py:224-271; RRF needs global ranks).
 * lists_out_dev [2][Q][pool] int64 = merged dense ids | merged BM25 ids, scores_out_dev [2][Q][pool] float64 = cosines |
 * normalised BM25; keys / rrf / ranks as rag_rrf_fuse_dev with n_lists = 2. Bit-identical to rag_hybrid_rrf_dev on the
 * unsharded index. */
int rag_hybrid_fuse_gathered_dev(rag_handle_t h, const int64_t* gathered_dev, int world, int n_queries, int pool, int k,
                                 int rrf_k, int64_t* lists_out_dev, double* scores_out_dev, int64_t* keys_out_dev,
                                 double* rrf_out_dev, int32_t* ranks_out_dev, void* stream);

/* ---- small pairwise cosine in float64: replaces the Python loops
 *      rag/consistency_checker.py:169-176, rag/context_compressor.py:227-228, rag/reranker.py:167-175,
 *      rag/nodes/helpers.py:232-243, rag/retrieval.py:253-256.  out[m*n] row-major, 0.0 on zero norm. */
int rag_pairwise_cosine_host(rag_handle_t h, const float* a_host, int m, const float* b_host, int n, int dim,
                             double* out_host);
/* the same on float64 inputs: the reference multiplies Python doubles (rag/retrieval.py:362-371), so the mirror classes hand the
 * agent's List[float] over unrounded - a pair sitting exactly on ConsistencyChecker's `>= 0.85` (rag/consistency_checker.py:179)
 * must not flip because its inputs were cast to float32 first. */
int rag_pairwise_cosine_f64_host(rag_handle_t h, const double* a_host, int m, const double* b_host, int n, int dim,
                                 double* out_host);

/* ---- reciprocal rank fusion: replaces ReciprocalRankFusion.fuse (rag/reranker.py:224-271).
 * lists_host: [Q][n_lists][list_len] int64 keys, -1 = padding (only at the tail of a list).
 * For each query: score[key] += 1.0/(rrf_k+rank) (rank from 1, list order), sort desc, first-seen
 * order on ties, top_k. Outputs -1 padded; ranks_out[Q][top_k][n_lists] = 1-based rank of the key's first
 * occurrence in each list (0 = absent) (may be NULL). */
int rag_rrf_fuse_host(rag_handle_t h, const int64_t* lists_host, int n_queries, int n_lists, int list_len,
                      int rrf_k, int top_k, int64_t* keys_out_host, double* scores_out_host,
                      int32_t* ranks_out_host);

/* Device-pointer forms of the fusion / BM25 entry points (asynchronous on `stream`), and the whole hybrid path of
 * BASELINE.json configs[2] in one call: dense top-`pool` + BM25 top-`pool` -> RRF(rrf_k) -> top-k, nothing leaves HBM.
 * rag_rrf_fuse_dev: lists_dev is [Q][n_lists][list_len]. rag_hybrid_rrf_dev: lists_ws_dev is caller scratch
 * [2][Q][pool] int64, scores_ws_dev [Q][pool] float64; keys are doc ids (BM25 rows use the dense index's id mapping,
 * the two indexes must be row-aligned). For batches of up to 4096 queries the BM25 leg of rag_hybrid_rrf_dev / rag_retrieve_rerank_dev
 * runs on an internal side stream that is forked from and joined back into `stream` by events: everything the call
 * reads and writes is still ordered on `stream` as if it had run there alone. */
int rag_rrf_fuse_dev(rag_handle_t h, const int64_t* lists_dev, int n_queries, int n_lists, int list_len, int rrf_k,
                     int top_k, int64_t* keys_out_dev, double* scores_out_dev, int32_t* ranks_out_dev, void* stream);
int rag_bm25_topk_dev(rag_handle_t h, const int32_t* term_ptr_dev, const int32_t* terms_dev, int n_queries, int k,
                      int tenant, int64_t* ids_out_dev, int32_t* rows_out_dev, double* scores_out_dev,
                      double* raw_max_out_dev, void* stream);
int rag_hybrid_rrf_dev(rag_handle_t h, const float* q_dev, const int32_t* term_ptr_dev, const int32_t* terms_dev,
                       int n_queries, int pool, int k, int rrf_k, int tenant, int64_t* lists_ws_dev,
                       double* scores_ws_dev, int64_t* keys_out_dev, double* rrf_out_dev, int32_t* ranks_out_dev,
                       void* stream);

/* ---- greedy MMR selection on the device: replaces the Python loops of
 *      MMRDiversifier.diversify (rag/reranker.py:116-195; variant 0: lambda*rel + (1-lambda)*(1-max_sim), first pick has
 *      diversity 1.0) and apply_mmr (rag/nodes/helpers.py:183-260; variant 1: lambda*rel - (1-lambda)*max_sim).
 *      First maximal candidate wins ties, float64 throughout. n (pool) <= 256. sel_out = positions into the candidate
 *      list (-1 padded when fewer than top_k candidates), score_out = the MMR score each pick won with.
 *      _host: explicit candidate embeddings emb[n][dim]. _dev: candidates are rows of the resident index
 *      (rows_dev[Q][pool], -1 = empty), queries q_dev[Q][dim]; embeddings never leave HBM. */
int rag_mmr_select_host(rag_handle_t h, const float* query_host /*dim*/, const float* emb_host /*n x dim*/, int n, int dim,
                        int top_k, double lambda, int variant, int32_t* sel_out_host, double* score_out_host);
int rag_mmr_select_dev(rag_handle_t h, const float* q_dev, const int32_t* rows_dev, int n_queries, int pool, int top_k,
                       double lambda, int variant, int32_t* sel_out_dev, double* score_out_dev, void* stream);

/* ---- semantic chunker chain: replaces the sentence loop of SemanticChunker.chunk
 *      (rag/chunking.py:153-199): sentence i joins the running chunk when cos(running pairwise average, e_i) >= threshold
 *      and the chunk stays <= max_chunk characters, else the chunk closes if it has >= min_chunk characters (otherwise
 *      the sentence is absorbed). emb[n][dim] sentence embeddings, sent_len[n] = len(sentence);
 *      group_out[n] = chunk number of each sentence. */
int rag_chunk_chain_host(rag_handle_t h, const float* emb_host, const int32_t* sent_len_host, int n, int dim,
                         double threshold, int max_chunk, int min_chunk, int32_t* group_out_host);

/* ---- BM25 over CSR postings: replaces BM25Okapi(tokenized_corpus).get_scores(query) + the /max
 *      normalisation (rag/retrieval.py:324-347). Postings are term-major CSR, docs ascending per term.
 *      idf[V] is computed by the host exactly as rank-bm25 does (float64). */
int rag_bm25_load_host(rag_handle_t h, const int64_t* indptr_host /*V+1*/, const int32_t* doc_host /*nnz*/,
                       const int32_t* tf_host /*nnz*/, const int32_t* doc_len_host /*N*/,
                       const double* idf_host /*V*/, int64_t n_docs, int64_t n_terms, double avgdl,
                       double k1, double b);
/* HBM bytes rag_bm25_load_host will take for a CSR with these offsets, computed on the host from indptr alone (no GPU call):
 * postings (doc id + float64 impact, 12 B each; 8 B with option bm25_packed: the impact is then idf * g[code of the posting's
 * (term frequency, document length) pair] - bit-identical, less HBM, a slower scoring loop), per-term metadata (32 B each) and
 * the per-term bracket tables that replace
 * a binary search of the posting list per (query token, 2048-document range). A term's table is sized by its document
 * frequency, so table bytes <= postings/12 for ANY vocabulary - the reference tokeniser (`doc.lower().split()`,
 * rag/retrieval.py:334-335) produces millions of distinct terms on a large shard. Lets a loader budget a shard before
 * uploading it. Any of the outputs may be NULL. */
int rag_bm25_index_bytes(const int64_t* indptr_host, int64_t n_docs, int64_t n_terms, int64_t* postings_bytes_out,
                         int64_t* meta_bytes_out, int64_t* table_bytes_out);
/* Launch geometry of ONE scoring launch over `n_ranges_in_launch` 2048-document ranges and `n_queries` queries (host-only, no GPU
 * call; no reference counterpart: the reference scores a query with one numpy pass, rag/retrieval.py:341). out5 = {workgroups,
 * ranges, queries, query groups per range G, queries per group L}. L = 0: workgroup b scores (range b % ranges, query b / ranges).
 * L > 0 (from 128 queries on): XCD-aware columns - workgroup b belongs to XCD x = b % 8 and is its s = b / 8 -th; it scores column
 * c = x + 8 * (s / L), i.e. range c / G, query (c % G) * L + s % L, and exits if that range or query does not exist. Every (range,
 * query) pair is scored exactly once; tests/test_bm25_table_plan.py replays the rule. linear != 0 forces L = 0. */
int rag_bm25_grid_plan(int n_ranges_in_launch, int n_queries, int linear, int64_t* out5);
/* term_ptr[Q+1], terms[term_ptr[Q]] (query tokens WITH repeats; -1 = out-of-vocabulary).
 * scores_out are max-normalised as the reference does; raw_max_out[Q] (may be NULL) is the divisor.
 * tenant >= 0 (needs rag_index_set_tenants_host and postings row-aligned with the index): only that tenant's documents
 * can be returned - the `WHERE agent_id = %s` every reference query carries (rag/document_store.py:457); idf / avgdl stay
 * the statistics of the whole loaded corpus, the max-normalisation uses the tenant's own best score. tenant < 0: no filter.
 * The same filter applies to the BM25 leg of rag_hybrid_rrf_dev and rag_retrieve_rerank_dev (mode 1). */
int rag_bm25_topk_host(rag_handle_t h, const int32_t* term_ptr_host, const int32_t* terms_host, int n_queries,
                       int k, int tenant, int64_t* ids_out_host, int32_t* rows_out_host, double* scores_out_host,
                       double* raw_max_out_host);
/* on = 1 (default): top-k scores are divided by the per-query max as rag/retrieval.py:343-345 does. on = 0: top-k
 * scores stay raw; a row-sharded index merges the shards' raw lists first and divides by the
 * GLOBAL max afterwards. raw_max_out is written either way. */
int rag_bm25_set_normalize(rag_handle_t h, int on);
/* dense scores for a (small) corpus: out[Q][N] raw (un-normalised) BM25, for hybrid_search semantics */
int rag_bm25_scores_host(rag_handle_t h, const int32_t* term_ptr_host, const int32_t* terms_host, int n_queries,
                         double* out_host);
/* The same for an AD-HOC corpus, stateless: HybridRetriever.hybrid_search builds a fresh BM25Okapi over the corpus it is
 * handed on every call (rag/retrieval.py:333-341). The postings (arguments as rag_bm25_load_host) are uploaded, scored
 * and dropped inside the call; the resident postings of the index are not touched. out[Q][n_docs] raw scores. */
int rag_bm25_scores_adhoc_host(rag_handle_t h, const int64_t* indptr_host, const int32_t* doc_host, const int32_t* tf_host,
                               const int32_t* doc_len_host, const double* idf_host, int64_t n_docs, int64_t n_terms,
                               double avgdl, double k1, double b, const int32_t* term_ptr_host,
                               const int32_t* terms_host, int n_queries, double* out_host);

/* ---- weighted linear fusion + top-k: replaces rag/retrieval.py:294-322
 *      hybrid = alpha*semantic + beta*keyword + gamma*temporal, stable sort desc, [:top_k]. */
int rag_linear_fuse_topk_host(rag_handle_t h, const double* semantic_host, const double* keyword_host,
                              const double* temporal_host /*NULL = zeros*/, int n, double alpha, double beta,
                              double gamma, int top_k, int32_t* idx_out_host, double* hybrid_out_host);

/* Index-level linear fusion (SURVEY.md section 8b `rag_hybrid_linear`): HybridRetriever.hybrid_search
 * (rag/retrieval.py:214-322) with the WHOLE resident index as its corpus. Per query and row:
 *   hybrid = (alpha * cosine + beta * keyword) + gamma * temporal      (:302, CPython's operation order, float64)
 * keyword = BM25Okapi score / max over all documents - under a tenant filter over the tenant's documents, the corpus the
 * reference would have been handed - (1.0 when that max is <= 0, :343-345); temporal = the per-row vector
 * of rag_index_set_temporal_host (RECENCY_WEIGHT * 0.5 ** (days_old / half_life), computed by the host as :266-292 does;
 * NULL = zeros). Result: stable sort descending (lower row first on ties), first k; rows_out are index rows, ids_out doc
 * ids, hybrid_out the float64 hybrid scores; semantic / keyword / temporal_out (each [Q*k], may be NULL) are the
 * components the reference returns next to them. alpha must be > 0; tenant as in rag_dense_topk_dev. The postings must be
 * row-aligned with the index. Exact by the same construction as the dense search: the fp16 MFMA pass only discards rows
 * whose FUSED score is provably below the k-th best, survivors are rescored in float64. */
int rag_index_set_temporal_host(rag_handle_t h, const double* temporal_host, int64_t n_rows);
int rag_hybrid_linear_dev(rag_handle_t h, const float* q_dev, const int32_t* term_ptr_dev, const int32_t* terms_dev,
                          int n_queries, int k, double alpha, double beta, double gamma, int tenant,
                          int64_t* ids_out_dev, int32_t* rows_out_dev, double* hybrid_out_dev, double* semantic_out_dev,
                          double* keyword_out_dev, double* temporal_out_dev, void* stream);

/* ---- cross-encoder (BertForSequenceClassification, ms-marco-MiniLM-L-6-v2 shape): replaces
 *      CrossEncoder.predict (rag/reranker.py:355). Weights are handed over as float32 host arrays in the
 *      HF state-dict layout (nn.Linear [out,in]); see optimized-rag_amd/cross_encoder.py for the order. */
typedef struct rag_ce_config {
    int32_t vocab_size, hidden, layers, heads, ffn, max_pos, type_vocab, reserved;
    double ln_eps;
} rag_ce_config;
int rag_ce_load_host(rag_handle_t h, const rag_ce_config* cfg, const float* const* tensors_host, int n_tensors);
/* input_ids/token_type_ids: [P][L] int32 (padded), lens[P]; logits_out[P] raw logits (float32). */
int rag_ce_score_host(rag_handle_t h, const int32_t* input_ids_host, const int32_t* token_type_ids_host,
                      const int32_t* lens_host, int n_pairs, int seq_len, float* logits_out_host);
int rag_ce_score_dev(rag_handle_t h, const int32_t* input_ids_dev, const int32_t* token_type_ids_dev,
                     const int32_t* lens_dev, int n_pairs, int seq_len, float* logits_out_dev, void* stream);

/* ---- local sentence-embedding model on the device: stands where the reference calls the OpenAI
 *      embeddings endpoint over HTTP for every query and every document - EmbeddingService._generate_embedding_uncached /
 *      _generate_batch_uncached (memory/embeddings.py:100-115, 226-246; dimension lookup :312-332). NOT a parity
 *      replacement: a local encoder produces DIFFERENT vectors than text-embedding-3-small, so an index must be built and
 *      queried with the same model (outside the 1e-3 contract of the north star; what is pinned is this forward against
 *      transformers.BertModel). The model is a BERT encoder (the cross-encoder's kernels) behind sentence-transformers'
 *      Pooling(mean) + Normalize head: tensors as rag_ce_load_host WITHOUT the four pooler / classifier tensors
 *      (5 + 16 * layers), normalize = 1 L2-normalises the pooled vector (x / max(|x|, 1e-12)).
 *      rag_embed_*: input_ids / token_type_ids [n_texts][L] int32 (padded), lens[n_texts]; out[n_texts][hidden] float32. */
int rag_embed_load_host(rag_handle_t h, const rag_ce_config* cfg, const float* const* tensors_host, int n_tensors, int normalize);
int rag_embed_host(rag_handle_t h, const int32_t* input_ids_host, const int32_t* token_type_ids_host, const int32_t* lens_host,
                   int n_texts, int seq_len, float* out_host);
int rag_embed_dev(rag_handle_t h, const int32_t* input_ids_dev, const int32_t* token_type_ids_dev, const int32_t* lens_dev,
                  int n_texts, int seq_len, float* out_dev, void* stream);
int rag_embed_dim(rag_handle_t h, int* dim_out);

/* ---- retrieve + rerank in one device-resident call (BASELINE.json configs[3]): the composition of
 *      HybridRetriever.retrieve (rag/retrieval.py:122-212) and CrossEncoderReranker.rerank (rag/reranker.py:320-384:
 *      pairs [query, passage], raw logits, sigmoid, sort desc, [:top_k]) with the passages' token ids resident in HBM.
 *      rag_tokens_load_host: tokens[n_rows][L] passage WordPiece ids (no [CLS]/[SEP]) + lens[n_rows], row-aligned with the
 *      index. rag_retrieve_rerank_dev: mode 0 = dense top-pool candidates, mode 1 = dense + BM25 + RRF(rrf_k) top-pool;
 *      pairs are [CLS] query [SEP] passage [SEP] truncated 'longest_first' to max_length L_pair (what CrossEncoder.predict's
 *      tokenizer call does; the reference's max_length is 512, rag/reranker.py:290-294) and padded to it; outputs per query:
 *      ids_out[k] doc ids (-1 padded), scores_out[k] = sigmoid(logit) as float64, logits_out[k] raw logits,
 *      cand_out[pool] (may be NULL) the candidate list that was reranked. */
int rag_tokens_load_host(rag_handle_t h, const int32_t* tokens_host, const int32_t* lens_host, int64_t n_rows, int L);
/* Chunked form for stores that should not exist as one host array (a replicated 100M-passage store, SURVEY.md section 8e, is
 * 45 GB resident as uint16 and would be 90 GB as one int32 host array): reserve once, then append row blocks in order from
 * DEVICE memory (tokens_dev[n][L] int32, lens_dev[n]); each append returns after its rows are resident and checked. */
int rag_tokens_reserve(rag_handle_t h, int64_t n_rows_total, int L);
int rag_tokens_append_dev(rag_handle_t h, const int32_t* tokens_dev, const int32_t* lens_dev, int64_t n_rows, void* stream);
int rag_retrieve_rerank_dev(rag_handle_t h, const float* q_emb_dev, const int32_t* term_ptr_dev, const int32_t* terms_dev,
                            const int32_t* q_tok_dev, const int32_t* q_len_dev, int Lq, int n_queries, int pool, int k,
                            int rrf_k, int tenant, int mode, int cls_id, int sep_id, int L_pair, int64_t* ids_out_dev,
                            double* scores_out_dev, float* logits_out_dev, int64_t* cand_out_dev, void* stream);

/* The pipeline's two small kernels on their own (row-sharded composition): pair assembly from GLOBAL
 * candidate doc ids against a replicated token store whose first row has id token_id_base; and sigmoid + stable top-k of
 * the logits of a [Q][pool] candidate table (rag/reranker.py:359,372-376). */
int rag_ce_build_pairs_dev(rag_handle_t h, const int32_t* q_tok_dev, const int32_t* q_len_dev, int Lq, const int64_t* cand_dev,
                           int n_queries, int pool, int64_t token_id_base, int L_pair, int cls_id, int sep_id,
                           int32_t* ids_out_dev, int32_t* tt_out_dev, int32_t* lens_out_dev, void* stream);
int rag_rerank_topk_dev(rag_handle_t h, const float* logits_dev, const int64_t* cand_dev, int n_queries, int pool, int k,
                        int64_t* ids_out_dev, double* scores_out_dev, float* logits_out_dev, void* stream);

/* ---- the exchange step of the row-sharded search (the reference is single-process) bound to RCCL
 *      directly, for hosts that do not want torch.distributed in the path: ONE all-gather of each rank's partial top-k lists
 *      per stage, over xGMI, followed by rag_merge_topk_dev / rag_rrf_fuse_dev on every rank. librccl is opened at the first
 *      call (dlopen; the library has no link-time dependency on it).
 *      rag_comm_unique_id: 128 bytes created on ONE rank and handed to the others out of band (file, socket, torch store).
 *      rag_comm_init: collective over `world` ranks, one per GPU. rag_comm_allgather_dev: recv_dev[world][bytes] <- every
 *      rank's send_dev[bytes], asynchronous on `stream`. */
int rag_comm_unique_id(void* id128_out);
int rag_comm_init(rag_handle_t h, int rank, int world, const void* id128);
int rag_comm_allgather_dev(rag_handle_t h, const void* send_dev, void* recv_dev, size_t bytes, void* stream);
/* ranks of the communicator the gathers run on, as RCCL itself counts them (ncclCommCount): what bench.py reports next to the
 * rank count of torch's own collective layer */
int rag_comm_count(rag_handle_t h, int* count_out);
int rag_comm_destroy(rag_handle_t h);

#ifdef __cplusplus
}
#endif
#endif /* RAG_HIP_H */

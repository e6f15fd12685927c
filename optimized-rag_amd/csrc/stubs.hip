// Entry points whose kernels are not built yet fail loudly (never a silent CPU fallback).
#include "common.h"
#define NOT_BUILT(h, what) do { (h)->err = std::string(what) + ": HIP kernel not built in this revision"; return RAG_ERR_STATE; } while (0)

int bm25_load_host(rag_ctx* h, const int64_t*, const int32_t*, const int32_t*, const int32_t*, const double*, int64_t, int64_t, double, double, double) { NOT_BUILT(h, "bm25_load"); }
int bm25_topk_host(rag_ctx* h, const int32_t*, const int32_t*, int, int, int64_t*, int32_t*, double*, double*) { NOT_BUILT(h, "bm25_topk"); }
int bm25_scores_host(rag_ctx* h, const int32_t*, const int32_t*, int, double*) { NOT_BUILT(h, "bm25_scores"); }
void bm25_free(rag_ctx*) {}
int ce_load_host(rag_ctx* h, const rag_ce_config*, const float* const*, int) { NOT_BUILT(h, "ce_load"); }
int ce_score(rag_ctx* h, const int32_t*, const int32_t*, const int32_t*, int, int, float*, hipStream_t, bool) { NOT_BUILT(h, "ce_score"); }
void ce_free(rag_ctx*) {}
int rrf_fuse_host(rag_ctx* h, const int64_t*, int, int, int, int, int, int64_t*, double*, int32_t*) { NOT_BUILT(h, "rrf_fuse"); }
int linear_fuse_topk_host(rag_ctx* h, const double*, const double*, const double*, int, double, double, double, int, int32_t*, double*) { NOT_BUILT(h, "linear_fuse"); }

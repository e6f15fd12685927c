// Entry points whose kernels are not built yet fail loudly (never a silent CPU fallback).
#include "common.h"
#define NOT_BUILT(h, what) do { (h)->err = std::string(what) + ": HIP kernel not built in this revision"; return RAG_ERR_STATE; } while (0)

int ce_load_host(rag_ctx* h, const rag_ce_config*, const float* const*, int) { NOT_BUILT(h, "ce_load"); }
int ce_score(rag_ctx* h, const int32_t*, const int32_t*, const int32_t*, int, int, float*, hipStream_t, bool) { NOT_BUILT(h, "ce_score"); }
void ce_free(rag_ctx*) {}

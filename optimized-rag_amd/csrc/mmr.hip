// Greedy Maximal-Marginal-Relevance selection on the device.
// Replaces the O(k * n * |selected|) Python cosine loops of
//   variant 0: MMRDiversifier.diversify   /root/reference/rag/reranker.py:116-195
//              score = lam * rel + (1 - lam) * (1 - max_sim),  first pick: diversity 1.0
//   variant 1: apply_mmr                  /root/reference/rag/nodes/helpers.py:183-260
//              score = lam * rel - (1 - lam) * max_sim,        first pick: max_sim 0.0
// rel = cos(query, e_i), max_sim = max over already selected s of cos(e_i, e_s); Python's `max` / `>` keep the FIRST
// maximal candidate. All arithmetic in float64 from the float32 embeddings (association as written above,
// -ffp-contract=off); only the summation order inside a dot product differs from CPython's sequential sum.
//
// One workgroup (4 waves) per query. Candidates are either explicit rows emb[n][dim] or rows of the resident index
// (emb32 master rows) addressed through rows[n] — in the batched pipeline the embeddings never leave HBM, which is
// what removes the reference's per-document re-embedding HTTP calls. Per round only the NEW pick's cosine row is
// computed (n dot products, one wave each), so the work is k * n * dim FMAs per query, not n^2 * dim.
#include <cstring>
#include "common.h"

#define MMR_MAX_N 256

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void mmr_select_kernel(const float* __restrict__ queries, const float* __restrict__ emb,
                                                          const int32_t* __restrict__ rows, int n, int dim, int top_k,
                                                          double lam, int variant, int32_t* __restrict__ sel_out,
                                                          double* __restrict__ score_out) {
    __shared__ double rel[MMR_MAX_N], norm[MMR_MAX_N], maxsim[MMR_MAX_N], score[MMR_MAX_N];
    __shared__ int alive[MMR_MAX_N];
    __shared__ double red_s[4];
    __shared__ int red_i[4];
    __shared__ int picked;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* qv = queries + (size_t)q * dim;
    const int32_t* rw = rows ? rows + (size_t)q * n : nullptr;
    const float* eb = rows ? emb : emb + (size_t)q * n * dim;          // explicit candidates are per query
    // ---- relevance and norms -------------------------------------------------------------------
    double qq = 0.0;
    for (int d = lane; d < dim; d += 64) qq += (double)qv[d] * (double)qv[d];
    qq = sqrt(wave_sum(qq));
    for (int j = wv; j < n; j += 4) {
        const int r = rw ? rw[j] : j;
        double dot = 0.0, ee = 0.0;
        if (r >= 0) {
            const float* e = eb + (size_t)r * dim;
            for (int d = lane; d < dim; d += 64) {
                const double x = (double)e[d];
                dot += (double)qv[d] * x;
                ee += x * x;
            }
        }
        dot = wave_sum(dot);
        ee = sqrt(wave_sum(ee));
        if (lane == 0) {
            norm[j] = ee;
            rel[j] = (qq == 0.0 || ee == 0.0) ? 0.0 : dot / (qq * ee);     // zero-norm -> 0.0 as the reference's cosine does
            maxsim[j] = -INFINITY;
            alive[j] = r >= 0;
        }
    }
    __syncthreads();
    const double oml = 1.0 - lam;
    for (int t = 0; t < top_k; ++t) {
        // ---- score every live candidate, first-maximum argmax ----------------------------------
        double best = -INFINITY;
        int best_i = 0x7fffffff;
        for (int j = tid; j < n; j += 256) {
            if (!alive[j]) continue;
            double s;
            if (variant == 0) {
                const double div = t == 0 ? 1.0 : 1.0 - maxsim[j];
                s = lam * rel[j] + oml * div;
            } else {
                const double ms = t == 0 ? 0.0 : maxsim[j];
                s = lam * rel[j] - oml * ms;
            }
            score[j] = s;
            if (s > best || (s == best && j < best_i) || best_i == 0x7fffffff) { best = s; best_i = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double os = __shfl_xor(best, o);
            const int oi = __shfl_xor(best_i, o);
            if (oi != 0x7fffffff && (best_i == 0x7fffffff || os > best || (os == best && oi < best_i))) { best = os; best_i = oi; }
        }
        if (lane == 0) { red_s[wv] = best; red_i[wv] = best_i; }
        __syncthreads();
        if (tid == 0) {
            double b = red_s[0];
            int bi = red_i[0];
            for (int w = 1; w < 4; ++w)
                if (red_i[w] != 0x7fffffff && (bi == 0x7fffffff || red_s[w] > b || (red_s[w] == b && red_i[w] < bi))) {
                    b = red_s[w];
                    bi = red_i[w];
                }
            picked = bi == 0x7fffffff ? -1 : bi;
            sel_out[(size_t)q * top_k + t] = picked;
            score_out[(size_t)q * top_k + t] = picked >= 0 ? b : 0.0;
            if (picked >= 0) alive[picked] = 0;
        }
        __syncthreads();
        const int s = picked;
        if (s < 0) {                                   // fewer candidates than top_k: pad the rest
            for (int u = t + 1 + tid; u < top_k; u += 256) { sel_out[(size_t)q * top_k + u] = -1; score_out[(size_t)q * top_k + u] = 0.0; }
            return;
        }
        if (t + 1 == top_k) return;
        // ---- cosine row of the new pick against every live candidate ---------------------------
        const float* es = eb + (size_t)(rw ? rw[s] : s) * dim;
        const double ns = norm[s];
        for (int j = wv; j < n; j += 4) {
            if (!alive[j]) continue;                   // wave-uniform
            const float* e = eb + (size_t)(rw ? rw[j] : j) * dim;
            double dot = 0.0;
            for (int d = lane; d < dim; d += 64) dot += (double)e[d] * (double)es[d];
            dot = wave_sum(dot);
            if (lane == 0) {
                const double c = (ns == 0.0 || norm[j] == 0.0) ? 0.0 : dot / (norm[j] * ns);
                maxsim[j] = fmax(maxsim[j], c);
            }
        }
        __syncthreads();
    }
}

int mmr_select_dev(rag_ctx* h, const float* queries_dev, const float* emb_dev, const int32_t* rows_dev, int Q, int n, int dim,
                   int top_k, double lam, int variant, int32_t* sel_dev, double* score_dev, hipStream_t st) {
    ARG_CHECK(h, Q > 0 && n > 0 && n <= MMR_MAX_N && dim > 0 && top_k > 0, "mmr: 1 <= n <= 256 candidates, top_k >= 1");
    ARG_CHECK(h, variant == 0 || variant == 1, "mmr: variant 0 (MMRDiversifier) or 1 (apply_mmr)");
    ARG_CHECK(h, queries_dev && emb_dev && sel_dev && score_dev, "mmr: null pointer");
    hipLaunchKernelGGL(mmr_select_kernel, dim3(Q), dim3(256), 0, st, queries_dev, emb_dev, rows_dev, n, dim, top_k, lam, variant,
                       sel_dev, score_dev);
    HIP_TRY(h, hipGetLastError());
    return RAG_OK;
}

int mmr_select_host(rag_ctx* h, const float* query, const float* emb, int n, int dim, int top_k, double lam, int variant,
                    int32_t* sel_out, double* score_out) {
    ARG_CHECK(h, query && emb && sel_out && score_out, "mmr: null pointer");
    ARG_CHECK(h, n > 0 && n <= MMR_MAX_N && dim > 0 && top_k > 0, "mmr: 1 <= n <= 256 candidates, top_k >= 1");
    hipStream_t st = h->stream;
    int rc = stage_reserve(h, stage_size(dim, 4) + stage_size((size_t)n * dim, 4) + stage_size(top_k, 4) + stage_size(top_k, 8));
    if (rc) return rc;
    char* p = (char*)h->stage;
    float* qd = stage_take<float>(p, dim);
    float* ed = stage_take<float>(p, (size_t)n * dim);
    int32_t* sd = stage_take<int32_t>(p, top_k);
    double* cd = stage_take<double>(p, top_k);
    HIP_TRY(h, hipMemcpyAsync(qd, query, (size_t)dim * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(ed, emb, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, st));
    rc = mmr_select_dev(h, qd, ed, nullptr, 1, n, dim, top_k, lam, variant, sd, cd, st);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(sel_out, sd, (size_t)top_k * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(score_out, cd, (size_t)top_k * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return RAG_OK;
}

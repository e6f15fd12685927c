// Sequential cosine chain of the semantic chunker:
//   SemanticChunker.chunk   /root/reference/rag/chunking.py:153-199
// Sentence i joins the running chunk when cos(current, e_i) >= threshold and the chunk stays <= max_chunk_size
// characters; otherwise the chunk is closed if it already has >= min_chunk_size characters (else the sentence is
// absorbed anyway). `current` is the running PAIRWISE average (current + e_i) / 2 (:181,:189), so every cosine depends
// on all earlier decisions: the chain is inherently sequential over sentences and parallel over the dimension. One
// workgroup keeps `current` in LDS as float64 and walks the sentences; three block-wide reductions per sentence.
// Output: group_out[i] = chunk number of sentence i (the host joins the strings).
#include "common.h"

__device__ __forceinline__ double chain_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void chunk_chain_kernel(const float* __restrict__ emb, const int32_t* __restrict__ sent_len,
                                                           int n, int dim, double threshold, int max_chunk, int min_chunk,
                                                           int32_t* __restrict__ group_out) {
    extern __shared__ double cur[];                 // [dim]
    __shared__ double red[3][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int d = tid; d < dim; d += 256) cur[d] = (double)emb[d];
    if (tid == 0) group_out[0] = 0;
    long long cur_size = sent_len[0];
    int chunk_id = 0;
    __syncthreads();
    for (int i = 1; i < n; ++i) {
        const float* e = emb + (size_t)i * dim;
        double dot = 0.0, m1 = 0.0, m2 = 0.0;
        for (int d = tid; d < dim; d += 256) {
            const double c = cur[d], x = (double)e[d];
            dot += c * x;
            m1 += c * c;
            m2 += x * x;
        }
        dot = chain_wave_sum(dot);
        m1 = chain_wave_sum(m1);
        m2 = chain_wave_sum(m2);
        if (lane == 0) { red[0][wv] = dot; red[1][wv] = m1; red[2][wv] = m2; }
        __syncthreads();
        dot = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        m1 = sqrt((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
        m2 = sqrt((red[2][0] + red[2][1]) + (red[2][2] + red[2][3]));
        const double sim = (m1 != 0.0 && m2 != 0.0) ? dot / (m1 * m2) : 0.0;
        const int len_i = sent_len[i];
        const bool join = (sim >= threshold && cur_size + len_i <= (long long)max_chunk) || cur_size < (long long)min_chunk;
        if (join) {
            for (int d = tid; d < dim; d += 256) cur[d] = (cur[d] + (double)e[d]) / 2.0;      // _average_embeddings of two
            cur_size += len_i;
        } else {
            ++chunk_id;
            for (int d = tid; d < dim; d += 256) cur[d] = (double)e[d];
            cur_size = len_i;
        }
        if (tid == 0) group_out[i] = chunk_id;
        __syncthreads();
    }
}

int chunk_chain_host(rag_ctx* h, const float* emb, const int32_t* sent_len, int n, int dim, double threshold, int max_chunk,
                     int min_chunk, int32_t* group_out) {
    ARG_CHECK(h, n > 0 && dim > 0 && dim <= 8192 && emb && sent_len && group_out, "chunk_chain: bad arguments (dim <= 8192)");
    hipStream_t st = h->stream;
    const int rc = stage_reserve(h, stage_size((size_t)n * dim, 4) + 2 * stage_size(n, 4));
    if (rc) return rc;
    char* p = (char*)h->stage;
    float* ed = stage_take<float>(p, (size_t)n * dim);
    int32_t* ld = stage_take<int32_t>(p, n);
    int32_t* gd = stage_take<int32_t>(p, n);
    HIP_TRY(h, hipMemcpyAsync(ed, emb, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(ld, sent_len, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(chunk_chain_kernel, dim3(1), dim3(256), (size_t)dim * sizeof(double), st, ed, ld, n, dim, threshold,
                       max_chunk, min_chunk, gd);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(group_out, gd, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return RAG_OK;
}

// K7: BertForSequenceClassification forward (ms-marco-MiniLM-L-6-v2 shape: 6 layers, hidden 384, 12 heads x 32,
// FFN 1536, 1 logit). Replaces `self.model.predict(pairs)` of CrossEncoderReranker.rerank
// (/root/reference/rag/reranker.py:355); sigmoid and sorting stay in the Python mirror (:359,:373).
//
// Numerics: the north star asks for rerank scores within 1e-3. Single fp16 operands give ~2e-2 logit error after
// 6 layers (measured), so every MFMA operand is a SPLIT fp16 pair x = hi + lo (hi = fp16(x), lo = fp16(x - hi),
// ~22 significand bits) stored as two planes, and every product is 3 MFMAs: hi*hi + hi*lo + lo*hi with fp32
// accumulation (3/16 of the f32-MFMA cost for the same accuracy class). Residual stream / LayerNorm / softmax /
// GELU (exact erf) / pooler are fp32. Layout: tokens are rows ([M = pairs*L][feature], feature contiguous), weights are
// nn.Linear [out][in] = K-contiguous, so every GEMM is the "both operands K-contiguous" form MFMA wants.
//
// Kernels
//   ce_embed_ln        word+pos+type gather -> LayerNorm -> x32 (fp32 residual) + x16 (fp16 GEMM operand)
//   ce_gemm<EPI>       128(out features) x 128(tokens) tile, BK=64, 4 waves (2x2 of 64x64), LDS-DMA double buffer,
//                      source-swizzled conflict-free ds_read_b128 (same scheme as dense.hip). Output features sit
//                      on the MFMA ROW so a lane owns 4 consecutive features of one token: bias is 4 registers and
//                      stores are 8/16-byte. Epilogues: QKV (bias, Q/K token-major, V written TRANSPOSED
//                      [pair][head][d][L] so P.V's B operand is a contiguous 16-byte load), bias+erf-GELU -> fp16,
//                      bias+residual -> fp32.
//   ce_attention<NT>   one wave per (pair, head, 16-query block): S = Q.K^T is ONE mfma_16x16x32 per 16 keys
//                      (d_head = 32 = MFMA K), softmax in registers (row = 16 lanes, xor-shuffle reduce), P -> LDS
//                      as fp16 -> A operand of P.V; key padding masked to -inf.
//   ce_layernorm       one wave per token (384 = 6/lane), fp32 statistics, eps from config
//   ce_pool_classify   tanh(Wp.x_cls + bp) -> wc.pooled + bc, fp32
#include "common.h"

#include <cmath>
#include <cstring>

struct rag_ce_model {
    rag_ce_config cfg;
    // embeddings fp32
    float *word = nullptr, *pos = nullptr, *type = nullptr, *emb_ln_g = nullptr, *emb_ln_b = nullptr;
    struct Layer {
        half_t *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;      // fp16 [out][in]
        float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
        float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
    };
    std::vector<Layer> layers;
    float *wp = nullptr, *bp = nullptr, *wc = nullptr, *bc = nullptr;            // pooler / classifier fp32
    std::vector<void*> allocs;
    // activation workspace (sized for ws_tokens)
    int64_t ws_tokens = 0;
    int ws_pairs = 0, ws_L = 0;
    float *x32 = nullptr, *y32 = nullptr;
    half_t *x16 = nullptr, *qk16 = nullptr, *vt16 = nullptr, *ctx16 = nullptr, *h16 = nullptr;
    int32_t *ids = nullptr, *tt = nullptr, *lens = nullptr;
    float* logits = nullptr;
};

#define CE_BM 128     // output features per tile (MFMA rows)
#define CE_BN 256     // tokens per tile (MFMA cols)
#define CE_BK 32      // K per LDS stage = one mfma_16x16x32 k-step
#define CE_W_BYTES (CE_BM * CE_BK * 2)                    // one plane of the weight tile: 8 KiB
#define CE_X_BYTES (CE_BN * CE_BK * 2)                    // one plane of the token tile: 16 KiB
#define CE_STAGE_BYTES (2 * CE_W_BYTES + 2 * CE_X_BYTES)  // W_hi | W_lo | X_hi | X_lo = 48 KiB
#define CE_GEMM_LDS (3 * CE_STAGE_BYTES)                  // three stages = 144 KiB

enum { EPI_QKV = 0, EPI_GELU = 1, EPI_RESID = 2 };

__device__ __forceinline__ void store_split4(half_t* __restrict__ p, size_t plane, float v0, float v1, float v2, float v3) {
    const half4 hi = {(half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3};
    const half4 lo = {(half_t)(v0 - (float)hi[0]), (half_t)(v1 - (float)hi[1]), (half_t)(v2 - (float)hi[2]),
                      (half_t)(v3 - (float)hi[3])};
    *reinterpret_cast<half4*>(p) = hi;
    *reinterpret_cast<half4*>(p + plane) = lo;
}

// LDS rows are 64 B (32 halfs = 4 chunks of 16 B). Chunk c of row r is stored at position c ^ ((-(r>>2)) & 3): with
// that swizzle every 16-lane group of a ds_read_b128 fragment read covers all 16 bank slots once (conflict-free);
// as in dense.hip it is applied on the DMA SOURCE address, the LDS destination stays lane-linear.
__device__ __forceinline__ void ce_dma(const half_t* __restrict__ g, char* lds, int wid) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(lds + wid * 64 * 16), 16, 0, 0);
}

// C^T[n][m] = sum_k W[n][k] * X[m][k].   W: [2 planes][N][K] fp16, X: [2 planes][M_pad][K] fp16 (hi plane, then lo).
// N % 128 == 0, M_pad % 256 == 0, K % 32 == 0, K >= 64.
// 128 (features) x 256 (tokens) tile, 8 waves (2 x 4, each 64 x 64), BK = 32, THREE LDS stages filled by LDS-DMA two
// K-steps ahead. Same staggered structure as dense.hip: per K-step an I-part (16 fragment reads + 6 DMA pieces,
// lgkmcnt(0)) and an M-part (48 MFMAs = 16 products x {lo*hi, hi*lo, hi*hi}), each closed by s_barrier; waves 4-7 run
// half a phase behind waves 0-3 so one group's MFMAs cover the other's reads. RAW: one counted s_waitcnt vmcnt(6)
// per K-step (step t+1 landed, step t+2 in flight); WAR: stage (t+2)%3 was last read in I(t-1).
template <int EPI>
__global__ __launch_bounds__(512) void ce_gemm_kernel(const half_t* __restrict__ W, size_t w_plane, const half_t* __restrict__ X,
                                                       size_t x_plane, int N, int K, const float* __restrict__ bias,
                                                       const float* __restrict__ resid, float* __restrict__ out32,
                                                       half_t* __restrict__ out16, size_t out_plane, half_t* __restrict__ vt16,
                                                       size_t vt_plane, int L, int hidden, int heads, int64_t m_valid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const bool lag = wm != 0;
    const int n0 = blockIdx.x * CE_BM;          // feature tile (fast index: all feature tiles of a token tile are adjacent)
    const int m0 = blockIdx.y * CE_BN;          // token tile
    // DMA source per thread: linear chunk i = tid (+512): row i>>2, position i&3 -> source chunk (i&3) ^ ((-(row>>2))&3)
    const int sr = tid >> 2;                                          // 0..127
    const int schunk = (tid & 3) ^ ((-(sr >> 2)) & 3);                // (row+128)>>2 has the same low 2 bits
    const half_t* w_src = W + (size_t)(n0 + sr) * K + schunk * 8;
    const half_t* x_src = X + (size_t)(m0 + sr) * K + schunk * 8;     // rows 0..127 of the token tile; +128*K for the rest
    const size_t x_half = (size_t)128 * K;
    const int fr = lane & 15, fq = lane >> 4;
    // fragment row r = base16 + fr (base16 multiple of 16 -> (r>>2)&3 == (fr>>2)&3): byte offset inside a plane tile
    const int off = fr * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);
    const int a_base = wm * 64 * 64, b_base = wn * 64 * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nt = K / CE_BK;
    const int last = nt - 1;
#define CE_STEP(u) (((u) < last ? (u) : last) * CE_BK)
#define CE_ISSUE(u)                                                                                               \
    {                                                                                                             \
        char* st_ = smem + ((u) % 3) * CE_STAGE_BYTES;                                                            \
        const int ko_ = CE_STEP(u);                                                                               \
        ce_dma(w_src + ko_, st_, wid);                                                                            \
        ce_dma(w_src + w_plane + ko_, st_ + CE_W_BYTES, wid);                                                     \
        ce_dma(x_src + ko_, st_ + 2 * CE_W_BYTES, wid);                                                           \
        ce_dma(x_src + x_half + ko_, st_ + 2 * CE_W_BYTES + 512 * 16, wid);                                       \
        ce_dma(x_src + x_plane + ko_, st_ + 2 * CE_W_BYTES + CE_X_BYTES, wid);                                    \
        ce_dma(x_src + x_plane + x_half + ko_, st_ + 2 * CE_W_BYTES + CE_X_BYTES + 512 * 16, wid);                \
    }
#define CE_BAR __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
    CE_ISSUE(0)
    CE_ISSUE(1)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    CE_BAR
    if (lag) { CE_BAR }
    for (int t = 0; t < nt; ++t) {
        const char* st = smem + (t % 3) * CE_STAGE_BYTES;
        half8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = *reinterpret_cast<const half8*>(st + a_base + i * 16 * 64 + off);
            al[i] = *reinterpret_cast<const half8*>(st + CE_W_BYTES + a_base + i * 16 * 64 + off);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bh[j] = *reinterpret_cast<const half8*>(st + 2 * CE_W_BYTES + b_base + j * 16 * 64 + off);
            bl[j] = *reinterpret_cast<const half8*>(st + 2 * CE_W_BYTES + CE_X_BYTES + b_base + j * 16 * 64 + off);
        }
        CE_ISSUE(t + 2)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lag) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        CE_BAR
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        if (!lag) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        CE_BAR
    }
    if (!lag) { CE_BAR }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // epilogue: acc[i][j][r] = C^T[n = n0 + wm*64 + i*16 + fq*4 + r][m = m0 + wn*64 + j*16 + fr]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wm * 64 + i * 16 + fq * 4;
        const float4 bv = *reinterpret_cast<const float4*>(bias + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wn * 64 + j * 16 + fr;
            float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
            if (EPI == EPI_RESID) {
                const float4 rv = *reinterpret_cast<const float4*>(resid + (size_t)m * N + n);
                *reinterpret_cast<float4*>(out32 + (size_t)m * N + n) = make_float4(v0 + rv.x, v1 + rv.y, v2 + rv.z, v3 + rv.w);
            } else if (EPI == EPI_GELU) {
                const float c = 0.70710678118654752440f;
                v0 = 0.5f * v0 * (1.0f + erff(v0 * c));
                v1 = 0.5f * v1 * (1.0f + erff(v1 * c));
                v2 = 0.5f * v2 * (1.0f + erff(v2 * c));
                v3 = 0.5f * v3 * (1.0f + erff(v3 * c));
                store_split4(out16 + (size_t)m * N + n, out_plane, v0, v1, v2, v3);
            } else {   // EPI_QKV: features [0,2*hidden) -> qk16[m][2*hidden]; [2*hidden,3*hidden) -> vt16[pair][head][d][L]
                if (n < 2 * hidden) {
                    store_split4(out16 + (size_t)m * (2 * hidden) + n, out_plane, v0, v1, v2, v3);
                } else if (m < m_valid) {                         // padded token rows have no (pair, token) slot
                    const int f = n - 2 * hidden;                 // 4 consecutive d of one head (32 % 4 == 0)
                    const int dh = hidden / heads;
                    const int head = f / dh, d = f % dh;
                    const int pair = m / L, tok = m % L;
                    half_t* o = vt16 + (((size_t)pair * heads + head) * dh + d) * L + tok;
                    const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const half_t hi = (half_t)vv[r];
                        o[(size_t)r * L] = hi;
                        o[vt_plane + (size_t)r * L] = (half_t)(vv[r] - (float)hi);
                    }
                }
            }
        }
    }
}

// ---- LayerNorm helpers: one wave per token row of `hidden` floats (hidden % 64 == 0, <= 1024) --------------
template <int PER>
__device__ __forceinline__ void wave_layernorm(float (&v)[PER], const float* __restrict__ g, const float* __restrict__ b,
                                               int hidden, float eps, int lane, float* __restrict__ o32, half_t* __restrict__ o16,
                                               size_t plane) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) s += v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)hidden;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const float d = v[i] - mean; q += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)hidden + eps);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = lane + i * 64;
        const float y = (v[i] - mean) * rstd * g[c] + b[c];
        o32[c] = y;
        const half_t hi = (half_t)y;
        o16[c] = hi;
        o16[plane + c] = (half_t)(y - (float)hi);
    }
}

template <int PER>
__global__ __launch_bounds__(256) void ce_embed_ln_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ tt,
                                                           const float* __restrict__ word, const float* __restrict__ pos,
                                                           const float* __restrict__ type, const float* __restrict__ g,
                                                           const float* __restrict__ b, int64_t M, int L, int hidden, int vocab,
                                                           float eps, float* __restrict__ x32, half_t* __restrict__ x16, size_t plane) {
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= M) return;
    int id = ids[tok];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const int ty = tt[tok] != 0;
    const int p = (int)(tok % L);
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = lane + i * 64;
        v[i] = word[(size_t)id * hidden + c] + type[(size_t)ty * hidden + c] + pos[(size_t)p * hidden + c];
    }
    wave_layernorm<PER>(v, g, b, hidden, eps, lane, x32 + tok * hidden, x16 + tok * hidden, plane);
}

template <int PER>
__global__ __launch_bounds__(256) void ce_layernorm_kernel(const float* __restrict__ y32, const float* __restrict__ g,
                                                            const float* __restrict__ b, int64_t M, int hidden, float eps,
                                                            float* __restrict__ x32, half_t* __restrict__ x16, size_t plane) {
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= M) return;
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = y32[tok * hidden + lane + i * 64];
    wave_layernorm<PER>(v, g, b, hidden, eps, lane, x32 + tok * hidden, x16 + tok * hidden, plane);
}

// ---- attention: d_head must be 32. NT = L/16 key tiles; one wave owns QB consecutive 16-query blocks so every K / V
// fragment it loads from L2 is used QB times. All operands are split fp16 (hi plane + lo plane): S and P.V are 3 MFMAs
// each. S is computed TRANSPOSED (A = K rows, B = Q rows): the accumulator then has the query on the lane column and
// 4 consecutive KEYS in a lane's registers, so P goes to LDS as packed 8-byte writes (not 2-byte scatters) and the
// softmax row reduction is registers + two xor-shuffles.
template <int NT, int QB>
__global__ __launch_bounds__(256) void ce_attention_kernel(const half_t* __restrict__ qk16, size_t qk_plane,
                                                            const half_t* __restrict__ vt16, size_t vt_plane,
                                                            const int32_t* __restrict__ lens, int L, int hidden, int heads,
                                                            half_t* __restrict__ ctx16, size_t ctx_plane) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PLD = NT * 16 + 8;                                            // P tile row pitch (halfs)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // per wave: QB x (P hi [16 q][L keys] | P lo)
    half_t (*pbase)[PLD] = reinterpret_cast<half_t (*)[PLD]>(smem + (size_t)wv * QB * 2 * 16 * PLD * 2);
    const int pair = blockIdx.z, head = blockIdx.y;
    const int qb0 = (blockIdx.x * 4 + wv) * QB;          // first 16-query block of this wave
    const int len = max(1, min(lens[pair], L));
    if (qb0 * 16 >= L) return;
    const int fr = lane & 15, fq = lane >> 4;
    const size_t row0 = (size_t)pair * L;
    const int ld = 2 * hidden;
    const float scale = 0.17677669529663687f;            // 32^-0.5
    // B operand = Q rows (query fr of block b, k = 8*fq..+8); rows past L are clamped (results discarded)
    half8 qh[QB], ql[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int qrow = min((qb0 + b) * 16 + fr, L - 1);
        const half_t* qp = qk16 + (row0 + qrow) * ld + head * 32 + fq * 8;
        qh[b] = *reinterpret_cast<const half8*>(qp);
        ql[b] = *reinterpret_cast<const half8*>(qp + qk_plane);
    }
    f32x4 s[QB][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        // A operand = K rows (key fr of tile t)
        const half_t* kp = qk16 + (row0 + t * 16 + fr) * ld + hidden + head * 32 + fq * 8;
        const half8 kh = *reinterpret_cast<const half8*>(kp);
        const half8 kl = *reinterpret_cast<const half8*>(kp + qk_plane);
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            z = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[b], z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[b], z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[b], z, 0, 0, 0);
            // C layout: col = fr = query, row = fq*4 + r = key within the tile
#pragma unroll
            for (int r = 0; r < 4; ++r) s[b][t][r] = (t * 16 + fq * 4 + r) < len ? z[r] * scale : -INFINITY;
        }
    }
    float inv_sum[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        // softmax over keys of query fr: this lane's NT*4 values, then the 4 lanes sharing fr (xor 16, 32)
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) m = fmaxf(m, fmaxf(fmaxf(s[b][t][0], s[b][t][1]), fmaxf(s[b][t][2], s[b][t][3])));
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float a = 0.f;
        half_t (*ph)[PLD] = pbase + (size_t)(2 * b) * 16;
        half_t (*pl)[PLD] = pbase + (size_t)(2 * b + 1) * 16;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { e[r] = expf(s[b][t][r] - m); a += e[r]; }
            const half4 hi = {(half_t)e[0], (half_t)e[1], (half_t)e[2], (half_t)e[3]};
            const half4 lo = {(half_t)(e[0] - (float)hi[0]), (half_t)(e[1] - (float)hi[1]), (half_t)(e[2] - (float)hi[2]),
                              (half_t)(e[3] - (float)hi[3])};
            *reinterpret_cast<half4*>(&ph[fr][t * 16 + fq * 4]) = hi;      // P[query fr][4 consecutive keys]
            *reinterpret_cast<half4*>(&pl[fr][t * 16 + fq * 4]) = lo;
        }
        a += __shfl_xor(a, 16);
        a += __shfl_xor(a, 32);
        inv_sum[b] = 1.0f / a;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): this wave's LDS writes done (wave-private tiles)
    __builtin_amdgcn_wave_barrier();
    // ctx[q][d] = sum_key P[q][key] * V[key][d]:  A = P (row q = fr, keys 8*fq..), B = V^T rows (d = fr)
    f32x4 c0[QB], c1[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) { c0[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; c1[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const half_t* vbase = vt16 + ((size_t)pair * heads + head) * 32 * L;
#pragma unroll
    for (int kb = 0; kb < NT / 2; ++kb) {          // 32 keys per MFMA
        const half_t* v0p = vbase + (size_t)fr * L + kb * 32 + fq * 8;
        const half_t* v1p = vbase + (size_t)(16 + fr) * L + kb * 32 + fq * 8;
        const half8 v0h = *reinterpret_cast<const half8*>(v0p), v0l = *reinterpret_cast<const half8*>(v0p + vt_plane);
        const half8 v1h = *reinterpret_cast<const half8*>(v1p), v1l = *reinterpret_cast<const half8*>(v1p + vt_plane);
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            half_t (*ph)[PLD] = pbase + (size_t)(2 * b) * 16;
            half_t (*pl)[PLD] = pbase + (size_t)(2 * b + 1) * 16;
            const half8 pfh = *reinterpret_cast<const half8*>(&ph[fr][kb * 32 + fq * 8]);
            const half8 pfl = *reinterpret_cast<const half8*>(&pl[fr][kb * 32 + fq * 8]);
            c0[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pfl, v0h, c0[b], 0, 0, 0);
            c0[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pfh, v0l, c0[b], 0, 0, 0);
            c0[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pfh, v0h, c0[b], 0, 0, 0);
            c1[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pfl, v1h, c1[b], 0, 0, 0);
            c1[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pfh, v1l, c1[b], 0, 0, 0);
            c1[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pfh, v1h, c1[b], 0, 0, 0);
        }
    }
    // C layout: col = fr = d (c0: d, c1: 16+d), row = fq*4 + r = query; 1/sum of that query lives in lane (fq*4+r)
#pragma unroll
    for (int b = 0; b < QB; ++b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qrow = (qb0 + b) * 16 + fq * 4 + r;
            const float inv = __shfl(inv_sum[b], fq * 4 + r);
            if (qrow >= L) continue;
            half_t* o = ctx16 + (row0 + qrow) * hidden + head * 32;
            const float a0 = c0[b][r] * inv, a1 = c1[b][r] * inv;
            const half_t h0 = (half_t)a0, h1 = (half_t)a1;
            o[fr] = h0;
            o[16 + fr] = h1;
            o[ctx_plane + fr] = (half_t)(a0 - (float)h0);
            o[ctx_plane + 16 + fr] = (half_t)(a1 - (float)h1);
        }
    }
}

__global__ __launch_bounds__(256) void ce_pool_classify_kernel(const float* __restrict__ x32, const float* __restrict__ wp,
                                                                const float* __restrict__ bp, const float* __restrict__ wc,
                                                                const float* __restrict__ bc, int L, int hidden,
                                                                float* __restrict__ logits) {
    __shared__ float xs[1024];
    __shared__ float part[4];
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* x = x32 + (size_t)pair * L * hidden;           // [CLS] = token 0
    for (int i = tid; i < hidden; i += 256) xs[i] = x[i];
    __syncthreads();
    float acc = 0.f;
    for (int n = tid; n < hidden; n += 256) {
        float s = bp[n];
        const float* w = wp + (size_t)n * hidden;
        for (int k = 0; k < hidden; ++k) s += w[k] * xs[k];
        acc += wc[n] * tanhf(s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) part[wv] = acc;
    __syncthreads();
    if (tid == 0) logits[pair] = part[0] + part[1] + part[2] + part[3] + bc[0];
}

__global__ void ce_f32_split_kernel(const float* __restrict__ in, half_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const half_t hi = (half_t)in[i];
        out[i] = hi;
        out[n + i] = (half_t)(in[i] - (float)hi);      // lo plane follows the hi plane
    }
}

// ------------------------------------------------------------------------------------------------
static void ce_free_ws(rag_ce_model* m) {
    hipFree(m->x32); hipFree(m->y32); hipFree(m->x16); hipFree(m->qk16); hipFree(m->vt16); hipFree(m->ctx16);
    hipFree(m->h16); hipFree(m->ids); hipFree(m->tt); hipFree(m->lens); hipFree(m->logits);
    m->x32 = m->y32 = nullptr; m->x16 = m->qk16 = m->vt16 = m->ctx16 = m->h16 = nullptr;
    m->ids = m->tt = m->lens = nullptr; m->logits = nullptr;
    m->ws_tokens = 0; m->ws_pairs = 0; m->ws_L = 0;
}

void ce_free(rag_ctx* h) {
    if (!h->ce) return;
    for (void* p : h->ce->allocs) hipFree(p);
    ce_free_ws(h->ce);
    delete h->ce;
    h->ce = nullptr;
}

static int up_f32(rag_ctx* h, rag_ce_model* m, const float* src, size_t n, float** dst) {
    HIP_TRY(h, hipMalloc(dst, n * sizeof(float)));
    m->allocs.push_back(*dst);
    HIP_TRY(h, hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    return RAG_OK;
}

// rows of several fp32 host matrices (same `cols`) concatenated -> one fp16 device matrix
static int up_f16_concat(rag_ctx* h, rag_ce_model* m, std::vector<const float*> srcs, size_t rows_each, size_t cols, half_t** dst) {
    const size_t n_each = rows_each * cols, total = n_each * srcs.size();
    float* tmp = nullptr;
    HIP_TRY(h, hipMalloc(&tmp, total * sizeof(float)));
    HIP_TRY(h, hipMalloc(dst, 2 * total * sizeof(half_t)));      // hi plane | lo plane
    m->allocs.push_back(*dst);
    for (size_t i = 0; i < srcs.size(); ++i)
        HIP_TRY(h, hipMemcpyAsync(tmp + i * n_each, srcs[i], n_each * sizeof(float), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(ce_f32_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, tmp, *dst, (int64_t)total);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    hipFree(tmp);
    return RAG_OK;
}

// Tensor order (HF state-dict names), see optimized-rag_amd/cross_encoder.py::flatten_state_dict:
//  0 word, 1 position, 2 token_type, 3 emb LN weight, 4 emb LN bias,
//  per layer (16): q.w q.b k.w k.b v.w v.b attn.out.w attn.out.b attn.LN.w attn.LN.b inter.w inter.b out.w out.b out.LN.w out.LN.b
//  then pooler.w pooler.b classifier.w classifier.b
int ce_load_host(rag_ctx* h, const rag_ce_config* cfg, const float* const* T, int n) {
    ARG_CHECK(h, cfg && T, "ce_load: null");
    ARG_CHECK(h, cfg->hidden % 128 == 0 && cfg->hidden <= 1024 && cfg->ffn % 128 == 0, "ce_load: hidden/ffn must be multiples of 128");
    ARG_CHECK(h, cfg->heads > 0 && cfg->hidden / cfg->heads == 32, "ce_load: head dim must be 32");
    ARG_CHECK(h, n == 5 + 16 * cfg->layers + 4, "ce_load: wrong tensor count");
    ce_free(h);
    rag_ce_model* m = new rag_ce_model();
    h->ce = m;
    m->cfg = *cfg;
    const size_t H = cfg->hidden, F = cfg->ffn;
    int rc;
    if ((rc = up_f32(h, m, T[0], (size_t)cfg->vocab_size * H, &m->word))) return rc;
    if ((rc = up_f32(h, m, T[1], (size_t)cfg->max_pos * H, &m->pos))) return rc;
    if ((rc = up_f32(h, m, T[2], (size_t)cfg->type_vocab * H, &m->type))) return rc;
    if ((rc = up_f32(h, m, T[3], H, &m->emb_ln_g))) return rc;
    if ((rc = up_f32(h, m, T[4], H, &m->emb_ln_b))) return rc;
    m->layers.resize(cfg->layers);
    for (int l = 0; l < cfg->layers; ++l) {
        const float* const* t = T + 5 + 16 * l;
        auto& ly = m->layers[l];
        if ((rc = up_f16_concat(h, m, {t[0], t[2], t[4]}, H, H, &ly.wqkv))) return rc;
        std::vector<float> bq(3 * H);
        std::memcpy(bq.data(), t[1], H * 4); std::memcpy(bq.data() + H, t[3], H * 4); std::memcpy(bq.data() + 2 * H, t[5], H * 4);
        if ((rc = up_f32(h, m, bq.data(), 3 * H, &ly.bqkv))) return rc;
        HIP_TRY(h, hipStreamSynchronize(h->stream));          // bq is a stack-lifetime buffer
        if ((rc = up_f16_concat(h, m, {t[6]}, H, H, &ly.wo))) return rc;
        if ((rc = up_f32(h, m, t[7], H, &ly.bo))) return rc;
        if ((rc = up_f32(h, m, t[8], H, &ly.ln1_g))) return rc;
        if ((rc = up_f32(h, m, t[9], H, &ly.ln1_b))) return rc;
        if ((rc = up_f16_concat(h, m, {t[10]}, F, H, &ly.w1))) return rc;
        if ((rc = up_f32(h, m, t[11], F, &ly.b1))) return rc;
        if ((rc = up_f16_concat(h, m, {t[12]}, H, F, &ly.w2))) return rc;
        if ((rc = up_f32(h, m, t[13], H, &ly.b2))) return rc;
        if ((rc = up_f32(h, m, t[14], H, &ly.ln2_g))) return rc;
        if ((rc = up_f32(h, m, t[15], H, &ly.ln2_b))) return rc;
    }
    const float* const* t = T + 5 + 16 * cfg->layers;
    if ((rc = up_f32(h, m, t[0], H * H, &m->wp))) return rc;
    if ((rc = up_f32(h, m, t[1], H, &m->bp))) return rc;
    if ((rc = up_f32(h, m, t[2], H, &m->wc))) return rc;
    if ((rc = up_f32(h, m, t[3], 1, &m->bc))) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return RAG_OK;
}

static const int kAttnL[] = {32, 64, 96, 128, 192, 256, 384, 512};

struct ce_planes { size_t x, qk, vt, ctx, h; };
static ce_planes planes_for(const rag_ce_model* m, int64_t Mp) {
    const size_t H = m->cfg.hidden, F = m->cfg.ffn;
    return {(size_t)Mp * H, (size_t)Mp * 2 * H, (size_t)Mp * H + 2048, (size_t)Mp * H, (size_t)Mp * F};
}

template <int NT>
static void launch_attention(rag_ce_model* m, int P, int L, const ce_planes& pp, hipStream_t st) {
    constexpr int QB = NT <= 16 ? 2 : 1;                               // 16-query blocks per wave (register budget)
    const size_t lds = (size_t)4 * QB * 2 * 16 * (NT * 16 + 8) * 2;    // 4 waves x QB x (P hi + P lo)
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ce_attention_kernel<NT, QB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int qblocks = L / 16;
    hipLaunchKernelGGL((ce_attention_kernel<NT, QB>), dim3((qblocks + 4 * QB - 1) / (4 * QB), m->cfg.heads, P), dim3(256), lds, st,
                       m->qk16, pp.qk, m->vt16, pp.vt, m->lens, L, m->cfg.hidden, m->cfg.heads, m->ctx16, pp.ctx);
}

template <int PER>
static void launch_ln(rag_ce_model* m, const float* y, const float* g, const float* b, int64_t M, size_t plane, hipStream_t st) {
    hipLaunchKernelGGL(ce_layernorm_kernel<PER>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, y, g, b, M, m->cfg.hidden,
                       (float)m->cfg.ln_eps, m->x32, m->x16, plane);
}

static int ce_forward_chunk(rag_ctx* h, rag_ce_model* m, int P, int L, hipStream_t st) {
    const int H = m->cfg.hidden, F = m->cfg.ffn;
    const int64_t M = (int64_t)P * L;
    const int64_t Mp = round_up((int64_t)m->ws_pairs * L, CE_BN);      // plane strides follow the ALLOCATED size
    const int64_t Mt = round_up(M, CE_BN);                            // token tiles actually computed
    const ce_planes pp = planes_for(m, Mp);
    const int per = H / 64;
    const float eps = (float)m->cfg.ln_eps;
    static bool attr = false;
    const size_t lds = CE_GEMM_LDS;
    if (!attr) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(ce_gemm_kernel<EPI_QKV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(ce_gemm_kernel<EPI_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(ce_gemm_kernel<EPI_RESID>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
#define CE_PER_DISPATCH(CALL)                                                                 \
    switch (per) {                                                                            \
        case 2: CALL(2); break; case 4: CALL(4); break; case 6: CALL(6); break;               \
        case 8: CALL(8); break; case 12: CALL(12); break; case 16: CALL(16); break;           \
        default: h->err = "ce: unsupported hidden size"; return RAG_ERR_ARG;                  \
    }
#define EMB(PER) hipLaunchKernelGGL(ce_embed_ln_kernel<PER>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, m->ids, m->tt, m->word, \
                                    m->pos, m->type, m->emb_ln_g, m->emb_ln_b, M, L, H, m->cfg.vocab_size, eps, m->x32, m->x16, pp.x)
    CE_PER_DISPATCH(EMB)
    const dim3 blk(512);
    const unsigned mt = (unsigned)(Mt / CE_BN);
    const half_t* nullh = nullptr;
    for (int l = 0; l < m->cfg.layers; ++l) {
        auto& ly = m->layers[l];
        hipLaunchKernelGGL(ce_gemm_kernel<EPI_QKV>, dim3(3 * H / CE_BM, mt), blk, lds, st, ly.wqkv, (size_t)3 * H * H, m->x16, pp.x,
                           3 * H, H, ly.bqkv, (const float*)nullptr, (float*)nullptr, m->qk16, pp.qk, m->vt16, pp.vt, L, H,
                           m->cfg.heads, M);
        switch (L / 16) {
            case 2: launch_attention<2>(m, P, L, pp, st); break;
            case 4: launch_attention<4>(m, P, L, pp, st); break;
            case 6: launch_attention<6>(m, P, L, pp, st); break;
            case 8: launch_attention<8>(m, P, L, pp, st); break;
            case 12: launch_attention<12>(m, P, L, pp, st); break;
            case 16: launch_attention<16>(m, P, L, pp, st); break;
            case 24: launch_attention<24>(m, P, L, pp, st); break;
            case 32: launch_attention<32>(m, P, L, pp, st); break;
            default: h->err = "ce: unsupported padded sequence length"; return RAG_ERR_ARG;
        }
        hipLaunchKernelGGL(ce_gemm_kernel<EPI_RESID>, dim3(H / CE_BM, mt), blk, lds, st, ly.wo, (size_t)H * H, m->ctx16, pp.ctx, H, H,
                           ly.bo, m->x32, m->y32, (half_t*)nullptr, (size_t)0, (half_t*)nullptr, (size_t)0, L, H, m->cfg.heads, M);
#define LN1(PER) launch_ln<PER>(m, m->y32, ly.ln1_g, ly.ln1_b, M, pp.x, st)
        CE_PER_DISPATCH(LN1)
        hipLaunchKernelGGL(ce_gemm_kernel<EPI_GELU>, dim3(F / CE_BM, mt), blk, lds, st, ly.w1, (size_t)F * H, m->x16, pp.x, F, H,
                           ly.b1, (const float*)nullptr, (float*)nullptr, m->h16, pp.h, (half_t*)nullptr, (size_t)0, L, H,
                           m->cfg.heads, M);
        hipLaunchKernelGGL(ce_gemm_kernel<EPI_RESID>, dim3(H / CE_BM, mt), blk, lds, st, ly.w2, (size_t)H * F, m->h16, pp.h, H, F,
                           ly.b2, m->x32, m->y32, (half_t*)nullptr, (size_t)0, (half_t*)nullptr, (size_t)0, L, H, m->cfg.heads, M);
#define LN2(PER) launch_ln<PER>(m, m->y32, ly.ln2_g, ly.ln2_b, M, pp.x, st)
        CE_PER_DISPATCH(LN2)
    }
    (void)nullh;
    hipLaunchKernelGGL(ce_pool_classify_kernel, dim3(P), dim3(256), 0, st, m->x32, m->wp, m->bp, m->wc, m->bc, L, H, m->logits);
    HIP_TRY(h, hipGetLastError());
    return RAG_OK;
}

static int ce_ensure_ws(rag_ctx* h, rag_ce_model* m, int P, int L) {
    if (P <= m->ws_pairs && L == m->ws_L) return RAG_OK;
    ce_free_ws(m);
    const int H = m->cfg.hidden, F = m->cfg.ffn;
    const int64_t Mp = round_up((int64_t)P * L, CE_BN);
    HIP_TRY(h, hipMalloc(&m->x32, (size_t)Mp * H * 4));
    HIP_TRY(h, hipMalloc(&m->y32, (size_t)Mp * H * 4));
    m->ws_pairs = P;                                     // planes_for() uses the allocated pair count
    const ce_planes pp = planes_for(m, Mp);
    HIP_TRY(h, hipMalloc(&m->x16, 2 * pp.x * 2));
    HIP_TRY(h, hipMalloc(&m->qk16, 2 * pp.qk * 2));
    HIP_TRY(h, hipMalloc(&m->vt16, 2 * pp.vt * 2));
    HIP_TRY(h, hipMalloc(&m->ctx16, 2 * pp.ctx * 2));
    HIP_TRY(h, hipMalloc(&m->h16, 2 * pp.h * 2));
    HIP_TRY(h, hipMalloc(&m->ids, (size_t)Mp * 4));
    HIP_TRY(h, hipMalloc(&m->tt, (size_t)Mp * 4));
    HIP_TRY(h, hipMalloc(&m->lens, (size_t)P * 4));
    HIP_TRY(h, hipMalloc(&m->logits, (size_t)P * 4));
    // padded token rows are read by the GEMM tiles: keep them finite
    HIP_TRY(h, hipMemset(m->x16, 0, 2 * pp.x * 2));
    HIP_TRY(h, hipMemset(m->ctx16, 0, 2 * pp.ctx * 2));
    HIP_TRY(h, hipMemset(m->h16, 0, 2 * pp.h * 2));
    HIP_TRY(h, hipMemset(m->qk16, 0, 2 * pp.qk * 2));
    HIP_TRY(h, hipMemset(m->vt16, 0, 2 * pp.vt * 2));
    HIP_TRY(h, hipMemset(m->x32, 0, (size_t)Mp * H * 4));
    m->ws_pairs = P;
    m->ws_L = L;
    m->ws_tokens = Mp;
    return RAG_OK;
}

// pads [P][L_in] token arrays to the supported attention length L (>= L_in), pad id 0 / type 0
__global__ void ce_pad_tokens_kernel(const int32_t* __restrict__ in_ids, const int32_t* __restrict__ in_tt, int P, int L_in, int L,
                                     int32_t* __restrict__ ids, int32_t* __restrict__ tt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)P * L) return;
    const int p = (int)(i / L), t = (int)(i % L);
    ids[i] = t < L_in ? in_ids[(size_t)p * L_in + t] : 0;
    tt[i] = t < L_in ? in_tt[(size_t)p * L_in + t] : 0;
}

int ce_score(rag_ctx* h, const int32_t* ids, const int32_t* tt, const int32_t* lens, int P, int L_in, float* out,
             hipStream_t st, bool host_ptrs) {
    ARG_CHECK(h, h->ce != nullptr, "no cross-encoder loaded");
    ARG_CHECK(h, ids && tt && lens && out && P > 0 && L_in > 0, "ce_score: bad arguments");
    rag_ce_model* m = h->ce;
    ARG_CHECK(h, L_in <= m->cfg.max_pos && L_in <= 512, "ce_score: sequence longer than max_position_embeddings/512");
    int L = 0;
    for (int c : kAttnL) if (c >= L_in) { L = c; break; }
    const int chunk = std::max(1, std::min(P, (int)(2'000'000 / L)));        // ~2M tokens of activations per chunk (~30 GB)
    int rc = ce_ensure_ws(h, m, chunk, L);
    if (rc) return rc;
    const hipMemcpyKind kin = host_ptrs ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    const hipMemcpyKind kout = host_ptrs ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    int32_t *sid = nullptr, *stt = nullptr;
    HIP_TRY(h, hipMalloc(&sid, (size_t)chunk * L_in * 4));
    HIP_TRY(h, hipMalloc(&stt, (size_t)chunk * L_in * 4));
    for (int p0 = 0; p0 < P; p0 += chunk) {
        const int pc = std::min(chunk, P - p0);
        HIP_TRY(h, hipMemcpyAsync(sid, ids + (size_t)p0 * L_in, (size_t)pc * L_in * 4, kin, st));
        HIP_TRY(h, hipMemcpyAsync(stt, tt + (size_t)p0 * L_in, (size_t)pc * L_in * 4, kin, st));
        HIP_TRY(h, hipMemcpyAsync(m->lens, lens + p0, (size_t)pc * 4, kin, st));
        const int64_t n = (int64_t)pc * L;
        hipLaunchKernelGGL(ce_pad_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sid, stt, pc, L_in, L, m->ids, m->tt);
        rc = ce_forward_chunk(h, m, pc, L, st);
        if (rc) break;
        HIP_TRY(h, hipMemcpyAsync(out + p0, m->logits, (size_t)pc * 4, kout, st));
    }
    hipError_t e = hipStreamSynchronize(st);
    hipFree(sid); hipFree(stt);
    if (rc) return rc;
    if (e != hipSuccess) {
        h->err = std::string("ce_score: ") + hipGetErrorString(e);
        return RAG_ERR_HIP;
    }
    return RAG_OK;
}

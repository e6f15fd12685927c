// K7: BertForSequenceClassification forward (ms-marco-MiniLM-L-6-v2 shape: 6 layers, hidden 384, 12 heads x 32,
// FFN 1536, 1 logit). Replaces `self.model.predict(pairs)` of CrossEncoderReranker.rerank
// (/root/reference/rag/reranker.py:355); sigmoid and sorting stay in the Python mirror (:359,:373).
//
// Numerics: the north star asks for rerank scores within 1e-3. Single fp16 operands give ~2e-2 logit error after
// 6 layers (measured), so every MFMA operand is a SPLIT fp16 pair x = hi + lo (hi = fp16(x), lo = fp16(x - hi),
// ~22 significand bits) stored as two planes, and every product is 3 MFMAs: hi*hi + hi*lo + lo*hi with fp32
// accumulation (3/16 of the f32-MFMA cost for the same accuracy class). GEMM operands (weights, x, ctx, q, ffn activations)
// keep the two halves INTERLEAVED per 32-element K group: [hi 32 halfs | lo 32 halfs] = 128 B, so the row piece one K-step
// (32 elements) stages is one whole 128-B line instead of two 64-B halves of two lines (SPLIT_IDX). Residual stream / LayerNorm / softmax /
// GELU (erf form, A&S 7.1.26) / pooler are fp32. Layout: tokens are rows, PACKED per pair (pair p owns len_p rounded up
// to 16 rows, offsets computed on the device), feature contiguous; weights are nn.Linear [out][in] = K-contiguous, so
// every GEMM is the "both operands K-contiguous" form MFMA wants.
//
// Kernels
//   ce_pack_scan/rows  lens -> pair_off / row_pair / m_packed (the packed row layout of the chunk)
//   ce_embed_ln        word+pos+type gather -> LayerNorm -> x16: the residual stream IS the split-fp16 GEMM operand (hi + lo
//                      = 22 significant bits); there is no separate fp32 copy
//   ce_gemm_ln         out-proj / FFN-down with bias + residual + LayerNorm fused into the epilogue (hidden = 384): a
//                      workgroup owns ALL features of its 128 tokens, so a row's statistics never leave the CU and the
//                      pre-LayerNorm tensor never touches HBM
//   ce_gemm<EPI>       persistent, XCD-aware; 128 (out features) x 256 (tokens) tiles, BK = 32, three LDS stages by
//                      LDS-DMA with a continuous stream across tiles, 8 staggered waves (2 x 4 of 64 x 64), source-swizzled
//                      conflict-free ds_read_b128. Output features sit on the MFMA ROW (bias = 4 registers per lane).
//                      Epilogues through the free ring stage, non-temporal stores: QKV (Q token-major; K and V in MFMA
//                      FRAGMENT order per (pair, head)), bias + GELU -> split fp16, bias + residual -> fp32.
//   ce_attention<QB>   one workgroup per (head, pair); K/V fragments by LDS-DMA, S computed transposed so that P stays in
//                      registers as the next MFMA's operand, online softmax over 32-key blocks, keys past len skipped.
//   ce_layernorm       one wave per token (384 = 6/lane), fp32 statistics, eps from config
//   ce_pool_classify   tanh(Wp.x_cls + bp) -> wc.pooled + bc, fp32
#include "common.h"
#include "ce_mx.h"
#include <type_traits>

#include <cmath>
#include <cstdlib>
#include <cstring>

struct rag_ce_model {
    rag_ce_config cfg;
    bool embed = false;          // true: sentence-embedding encoder (mean pooling over the tokens, no pooler / classifier head)
    int normalize = 1;           // embed: L2-normalise the pooled vectors
    int out_width = 1;           // floats per pair the forward produces: 1 logit, or `hidden` for an embedding model
    // embeddings fp32
    float *word = nullptr, *pos = nullptr, *type = nullptr, *emb_ln_g = nullptr, *emb_ln_b = nullptr;
    struct Layer {
        half_t *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;      // fp16 [out][in]
        char *wqkv8 = nullptr, *wo8 = nullptr, *w18 = nullptr, *w28 = nullptr;     // the same matrices as hi16 + lo8 images (ce_mx.h), when the shape allows
        float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
        float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
    };
    std::vector<Layer> layers;
    float *wp = nullptr, *bp = nullptr, *wc = nullptr, *bc = nullptr;            // pooler / classifier fp32
    float* wpT = nullptr;                                                         // pooler matrix transposed (mx_pool_classify_kernel)
    std::vector<void*> allocs;
    // activation workspace (sized for ws_tokens)
    int64_t ws_tokens = 0;
    int ws_pairs = 0, ws_L = 0;
    float* y32 = nullptr;                              // pre-LayerNorm sums of the unfused residual + LayerNorm path, grown on demand
    int64_t y32_rows = 0;
    int64_t h16_rows = 0;                              // rows of h16 (the FFN intermediate of the two-launch form), grown on demand
    half_t *x16 = nullptr, *q16 = nullptr, *kf16 = nullptr, *vf16 = nullptr, *ctx16 = nullptr, *h16 = nullptr;
    int32_t *ids = nullptr, *tt = nullptr, *lens = nullptr;
    // packed (variable-length) row layout of the current chunk: pair p owns rows [pair_off[p], pair_off[p+1]) where
    // pair_off[p+1] - pair_off[p] = len rounded up to 16; row_pair[m] = owning pair (-1 past the end); m_packed[0] = rows
    int32_t *pair_off = nullptr, *row_pair = nullptr, *m_packed = nullptr;
    int32_t *sid = nullptr, *stt = nullptr;          // staging of one chunk's [pairs][L_in] token / type ids
    float* logits = nullptr;
    // the MX forward (ce_mx.h: hi16 + lo8 operands) has a workspace of its own: it runs the large batches, the split-fp16 kernels
    // above the small ones (their tiles are finer), and neither path must size or evict the other's buffers
    bool mx_ok = false;                                // the shape allows the MX path (hidden 384, ffn a multiple of 384 up to 1536) and its weights are loaded
    struct MxWs {
        int pairs = 0, L = 0;
        int64_t tokens = 0;                            // padded rows (a multiple of 256)
        char *x8 = nullptr, *ctx8 = nullptr, *h8 = nullptr;              // residual stream, attention output, FFN intermediate (image layout)
        char *xc8 = nullptr, *cc8 = nullptr, *hc8 = nullptr;             // the same three for ONE row per pair: the [CLS] rows through the last layer's tail
        int32_t* m_cls = nullptr;                                        // device scalar: rows of the compact tensors (= pairs of the chunk)
        half_t *qf16 = nullptr, *kf16 = nullptr, *vf16 = nullptr;       // Q, K, V in the attention kernel's fragment order (hi | lo planes)
        int32_t *ids = nullptr, *tt = nullptr, *lens = nullptr, *pair_off = nullptr, *row_pair = nullptr, *m_packed = nullptr;
        int32_t *sid = nullptr, *stt = nullptr;
        float* logits = nullptr;
    } mx;
};

#define CE_BM 128     // output features per tile (MFMA rows)
#define CE_BN 256     // tokens per tile (MFMA cols)
#define CE_BK 32      // K per LDS stage = one mfma_16x16x32 k-step
#define CE_W_TILE (CE_BM * 128)                           // weight tile of one K-step: 128 rows x [hi 64 B | lo 64 B] = 16 KiB
#define CE_X_TILE (CE_BN * 128)                           // token tile of one K-step: 256 rows x 128 B = 32 KiB
#define CE_STAGE_BYTES (CE_W_TILE + CE_X_TILE)            // 48 KiB
// element c of a split row lives at half index SPLIT_IDX(c) (hi) and SPLIT_IDX(c) + 32 (lo); a row of n elements takes 2n halfs
#define SPLIT_IDX(c) ((((c) >> 5) << 6) + ((c) & 31))
#define CE_GEMM_LDS (3 * CE_STAGE_BYTES)                  // three stages = 144 KiB
#define CE_EPI_PLANE16 (16 * 144)                         // one fp16 plane of a 16-token x 64-feature epilogue pass, rows padded to 144 B

enum { EPI_QKV = 0, EPI_GELU = 1, EPI_RESID = 2 };

// erf-GELU (transformers' "gelu"): 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, below the 2^-22 resolution of the split-fp16 activations): one rcp, one exp2 and six FMAs
// instead of libm's branchy erff, which cost as much as the whole K = 384 main loop of the FFN-up GEMM.
__device__ __forceinline__ float ce_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float erf_abs = fmaf(-p * t, e, 1.0f);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

__device__ __forceinline__ void store_split4(half_t* __restrict__ p, size_t plane, float v0, float v1, float v2, float v3) {
    const half4 hi = {(half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3};
    const half4 lo = {(half_t)(v0 - (float)hi[0]), (half_t)(v1 - (float)hi[1]), (half_t)(v2 - (float)hi[2]),
                      (half_t)(v3 - (float)hi[3])};
    *reinterpret_cast<half4*>(p) = hi;
    *reinterpret_cast<half4*>(p + plane) = lo;
}

// LDS rows are 128 B (8 chunks of 16 B: hi chunks 0-3, lo chunks 4-7 of one 32-element K group). Chunk c of row r is
// stored at position c ^ ((r>>1) & 7) - dense.hip's swizzle, conflict-free for ds_read_b128 fragment reads - applied on the
// DMA SOURCE address; the LDS destination stays lane-linear.
__device__ __forceinline__ void ce_dma(const half_t* __restrict__ g, char* lds, int wid) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(lds + wid * 64 * 16), 16, 0, 0);
}

// LDS-DMA piece through a buffer descriptor: per-lane byte offset in ONE VGPR, piece / K-step offset in a scalar register
// (no 64-bit address arithmetic per piece; reads past the descriptor's end return zeros)
__device__ __forceinline__ void ce_bdma(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, char* lds, int wid) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wid * 64 * 16), 16, voff, soff, 0, 0);
}

// C^T[n][m] = sum_k W[n][k] * X[m][k].   W: [2 planes][N][K] fp16, X: [2 planes][M_pad][K] fp16 (hi plane, then lo).
// N % 128 == 0, M_pad % 256 == 0, K % 32 == 0, K >= 64.
// 128 (features) x 256 (tokens) tile, 8 waves (2 x 4, each 64 x 64), BK = 32, THREE LDS stages filled by LDS-DMA two
// K-steps ahead. Same staggered structure as dense.hip: per K-step an I-part (16 fragment reads + 6 DMA pieces,
// lgkmcnt(0)) and an M-part (48 MFMAs = 16 products x {lo*hi, hi*lo, hi*hi}), each closed by s_barrier; waves 4-7 run
// half a phase behind waves 0-3 so one group's MFMAs cover the other's reads. RAW: one counted s_waitcnt vmcnt(6)
// per K-step (step t+1 landed, step t+2 in flight); WAR: stage (t+2)%3 was last read in I(t-1).
//
// PERSISTENT workgroups (one per CU) with a CONTINUOUS DMA stream across tiles: the K = 384 GEMMs have only 12 K-steps, so
// a per-tile prologue (two stages from cold) and epilogue cost a quarter of the tile. The last two K-steps of a tile
// therefore issue steps 0 and 1 of the workgroup's NEXT tile into the stage ring, and the epilogue transposes through
// the one stage that is free at that point (48 KiB = 6 KiB per wave, four 16-token passes), so the next main loop
// starts with both stages resident. The packed row count lives on the device (no host sync): the loop stops at the
// first token tile past it.
#define CE_EPI_WAVE_BYTES (CE_STAGE_BYTES / 8)            // 6 KiB of the free stage per wave
// TERMS: which correction products run beside hi*hi. bit 0 = W_lo * x_hi (undoes the fp16 rounding of the WEIGHTS), bit 1 =
// W_hi * x_lo (undoes the rounding of the ACTIVATIONS). 3 = both (default everywhere); the other instances exist for the
// per-site ablation of DESIGN.md section 4.5 (RAG_CE_TERMS) and for sites where a term is provably not needed.
template <int EPI, int TERMS>
__global__ __launch_bounds__(512) void ce_gemm_kernel(const half_t* __restrict__ W, const half_t* __restrict__ X,
                                                       int N, int K, const float* __restrict__ bias,
                                                       const half_t* __restrict__ resid, float* __restrict__ out32,
                                                       half_t* __restrict__ out16, half_t* __restrict__ kf16,
                                                       half_t* __restrict__ vf16, size_t kv_plane, int hidden, int heads,
                                                       const int32_t* __restrict__ m_packed, int m_pad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const bool lag = wm != 0;
    const int n_ft = N / CE_BM;
    const int m_end = m_packed[0];
    // XCD-aware work order (speed only): workgroup b runs on XCD b & 7, and every XCD has its own L2. XCD x owns the token
    // tiles m = x (mod 8); its gridDim.x / 8 workgroups walk the (token tile, feature tile) pairs of those tiles feature
    // tile fastest, so the n_ft workgroups that share a token tile read it through ONE L2 (before this, the tile was
    // fetched once per XCD: 20 GB of HBM/fabric reads per FFN-up GEMM against 2.3 GB of activations).
    const int xcd = blockIdx.x & 7, n_slots = gridDim.x >> 3;
    int work = blockIdx.x >> 3;
#define CE_TILE_M(w) (((w) / n_ft) * 8 + xcd)
#define CE_TILE_N(w) ((w) % n_ft)
    if (CE_TILE_M(work) * CE_BN >= m_end) return;
    // DMA source per thread: one piece = 64 rows x 128 B; linear chunk i = tid: row i>>3, position i&7 -> source chunk
    // (i&7) ^ ((row>>1)&7); (row + 64)>>1 has the same low 3 bits, so every piece of a tile uses the same per-thread chunk
    const int sr = tid >> 3;                                          // 0..63
    const int schunk = (tid & 7) ^ ((sr >> 1) & 7);
    const size_t ldk = (size_t)2 * K;                                 // halfs per split row
    const size_t piece = (size_t)64 * ldk;                            // 64 rows further
    const int fr = lane & 15, fq = lane >> 4;
    // fragment row r = base16 + fr (base16 multiple of 16 -> (r>>1)&7 == (fr>>1)&7): byte offsets of the hi / lo chunk
    const int sw = (fr >> 1) & 7;
    const int off = fr * 128 + ((fq ^ sw) << 4), off_lo = fr * 128 + (((4 + fq) ^ sw) << 4);
    const int a_base = wm * 64 * 128, b_base = CE_W_TILE + wn * 64 * 128;
    const int nt = K / CE_BK;
    const int last = nt - 1;
    // Buffer addressing (as ce_gemm_ln_kernel): W through one descriptor + a scalar tile offset, the 256 token rows of a tile
    // through a per-tile descriptor, ONE per-lane byte offset for all six pieces of a stage. The pointer form spent ~14 VALU
    // instructions per K-step on 64-bit source addresses inside the part of the step that is on the critical path.
    const unsigned voff = (unsigned)(((size_t)sr * ldk + schunk * 8) * sizeof(half_t));
    const unsigned piece_b = (unsigned)(piece * sizeof(half_t));
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(W), 0, (int)((size_t)N * ldk * sizeof(half_t)), 0x00020000);
    unsigned w_cur = (unsigned)((size_t)CE_TILE_N(work) * CE_BM * ldk * sizeof(half_t)), w_nxt = w_cur;      // byte offset of the feature tile
    __amdgpu_buffer_rsrc_t x_cur = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(X + (size_t)CE_TILE_M(work) * CE_BN * ldk), 0,
                                                                     (int)(4 * piece_b), 0x00020000);
    __amdgpu_buffer_rsrc_t x_nxt = x_cur;
    bool has_next = false;
    int sbase = 0;                               // ring stage of step 0 of the current tile
#define CE_ISSUE(u)   /* step u of the current tile; u >= nt: step u - nt of the next tile (or a harmless re-load) */     \
    {                                                                                                             \
        const int u_ = (u);                                                                                       \
        const bool nx_ = u_ >= nt && has_next;                                                                    \
        const unsigned ks_ = (unsigned)(u_ < nt ? u_ : (has_next ? u_ - nt : last)) * 128u;                       \
        const unsigned ws_ = (nx_ ? w_nxt : w_cur) + ks_;                                                         \
        char* st_ = smem + ((sbase + u_) % 3) * CE_STAGE_BYTES;                                                   \
        ce_bdma(w_rs, voff, ws_, st_, wid);                                                                       \
        ce_bdma(w_rs, voff, ws_ + piece_b, st_ + 8192, wid);                                                      \
        if (nx_) {                                                                                                \
            ce_bdma(x_nxt, voff, ks_, st_ + CE_W_TILE, wid);                                                      \
            ce_bdma(x_nxt, voff, ks_ + piece_b, st_ + CE_W_TILE + 8192, wid);                                     \
            ce_bdma(x_nxt, voff, ks_ + 2 * piece_b, st_ + CE_W_TILE + 2 * 8192, wid);                             \
            ce_bdma(x_nxt, voff, ks_ + 3 * piece_b, st_ + CE_W_TILE + 3 * 8192, wid);                             \
        } else {                                                                                                  \
            ce_bdma(x_cur, voff, ks_, st_ + CE_W_TILE, wid);                                                      \
            ce_bdma(x_cur, voff, ks_ + piece_b, st_ + CE_W_TILE + 8192, wid);                                     \
            ce_bdma(x_cur, voff, ks_ + 2 * piece_b, st_ + CE_W_TILE + 2 * 8192, wid);                             \
            ce_bdma(x_cur, voff, ks_ + 3 * piece_b, st_ + CE_W_TILE + 3 * 8192, wid);                             \
        }                                                                                                         \
    }
#define CE_BAR __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
    CE_ISSUE(0)
    CE_ISSUE(1)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    CE_BAR
    for (bool first = true;; first = false) {
        const int n0 = CE_TILE_N(work) * CE_BM;  // feature tile
        const int m0 = CE_TILE_M(work) * CE_BN;  // token tile
        const int nb = n0 + wm * 64, mb = m0 + wn * 64;
        {
            const int nx = work + n_slots;
            has_next = CE_TILE_M(nx) * CE_BN < m_end;
            if (has_next) {
                w_nxt = (unsigned)((size_t)CE_TILE_N(nx) * CE_BM * ldk * sizeof(half_t));
                x_nxt = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(X + (size_t)CE_TILE_M(nx) * CE_BN * ldk), 0, (int)(4 * piece_b), 0x00020000);
            }
        }
        // bias in registers before the main loop: the epilogue must not start with a global load behind the in-flight DMA
        float4 bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const float4*>(bias + nb + i * 16 + fq * 4);
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (lag) { CE_BAR }
        for (int t = 0; t < nt; ++t) {
            const char* st = smem + ((sbase + t) % 3) * CE_STAGE_BYTES;
            half8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[i] = *reinterpret_cast<const half8*>(st + a_base + i * 16 * 128 + off);
                if (TERMS & 1) al[i] = *reinterpret_cast<const half8*>(st + a_base + i * 16 * 128 + off_lo);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bh[j] = *reinterpret_cast<const half8*>(st + b_base + j * 16 * 128 + off);
                if (TERMS & 2) bl[j] = *reinterpret_cast<const half8*>(st + b_base + j * 16 * 128 + off_lo);
            }
            CE_ISSUE(t + 2)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // step t+1 must have landed before the next I-part. On a continued tile steps 0 and 1 were resident before the
            // loop started (see below), so its first wait is skipped: it would only wait for the previous epilogue's stores.
            const bool need_wait = first || t > 0;
            if (lag && need_wait) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            CE_BAR
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (TERMS & 1) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    if (TERMS & 2) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_s_setprio(0);
            if (!lag && need_wait) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            CE_BAR
        }
        if (!lag) { CE_BAR }
        // Both groups are past their last fragment reads. Steps 0 and 1 of the next tile are in flight into the other two
        // stages; wait for them here, where no store is outstanding yet (a counted wait across the epilogue's stores
        // would rely on loads and stores retiring in one order).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // epilogue: acc[i][j][r] = C^T[n = nb + i*16 + fq*4 + r][m = mb + j*16 + fr]. Each wave transposes its 64 x 64 tile,
        // 16 tokens (32 for V) at a time, through its 6 KiB of the free stage so that every global store / residual load
        // covers whole 128 B lines of the row-major outputs (or one whole 1 KiB MFMA fragment tile for K and V).
        // outputs are streamed with non-temporal stores: they are read again only by a later kernel (GBs later), and as
        // ordinary stores they pushed the shared token tile and the weights out of the XCD's 4 MiB L2
        char* wl = smem + ((sbase + nt + 2) % 3) * CE_STAGE_BYTES + wid * CE_EPI_WAVE_BYTES;
        if (EPI == EPI_RESID) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {                       // fp32 [16 tokens][64 features], row stride 272 B
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float4*>(wl + fr * 272 + (i * 16 + fq * 4) * 4) =
                        make_float4(acc[i][j][0] + bv[i].x, acc[i][j][1] + bv[i].y, acc[i][j][2] + bv[i].z, acc[i][j][3] + bv[i].w);
                __builtin_amdgcn_wave_barrier();
                const int rr = lane >> 4, cc = lane & 15;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 4 + rr;
                    const float4 v = *reinterpret_cast<const float4*>(wl + row * 272 + cc * 16);
                    const size_t g = (size_t)(mb + j * 16 + row) * N + nb + cc * 4;
                    const half_t* rp = resid + (size_t)(mb + j * 16 + row) * 2 * N + SPLIT_IDX(nb + cc * 4);      // residual = hi + lo
                    const half4 rh = *reinterpret_cast<const half4*>(rp), rl = *reinterpret_cast<const half4*>(rp + 32);
                    __builtin_nontemporal_store((f32x4){v.x + ((float)rh[0] + (float)rl[0]), v.y + ((float)rh[1] + (float)rl[1]),
                                                        v.z + ((float)rh[2] + (float)rl[2]), v.w + ((float)rh[3] + (float)rl[3])},
                                                reinterpret_cast<f32x4*>(out32 + g));
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else if (EPI == EPI_GELU || nb < 2 * hidden) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {                       // split fp16 [16 tokens][64 features]: hi plane | lo plane, rows 144 B
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v0 = acc[i][j][0] + bv[i].x, v1 = acc[i][j][1] + bv[i].y, v2 = acc[i][j][2] + bv[i].z, v3 = acc[i][j][3] + bv[i].w;
                    if (EPI == EPI_GELU) { v0 = ce_gelu(v0); v1 = ce_gelu(v1); v2 = ce_gelu(v2); v3 = ce_gelu(v3); }
                    store_split4(reinterpret_cast<half_t*>(wl + fr * 144 + (i * 16 + fq * 4) * 2), CE_EPI_PLANE16 / 2, v0, v1, v2, v3);
                }
                __builtin_amdgcn_wave_barrier();
                if (EPI == EPI_GELU || nb < hidden) {
                    // FFN activations [token][ffn], or Q rows [token][hidden], in the split-row layout: the wave's 64 features
                    // of a token are two K groups = [hi 32 | lo 32 | hi 32 | lo 32] = 256 contiguous bytes; 16 lanes cover them
                    const int ldo = 2 * (EPI == EPI_GELU ? N : hidden);
                    const int rr = lane >> 4, cc = lane & 15;
                    const int grp = cc >> 3, part = (cc >> 2) & 1, qtr = cc & 3;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = it * 4 + rr;
                        const half8 v = *reinterpret_cast<const half8*>(wl + part * CE_EPI_PLANE16 + row * 144 + (grp * 4 + qtr) * 16);
                        half_t* o = out16 + (size_t)(mb + j * 16 + row) * ldo + (nb >> 5) * 64 + cc * 8;
                        __builtin_nontemporal_store(v, reinterpret_cast<half8*>(o));
                    }
                } else {
                    // K features -> kf16[head][16-row tile of the PACKED row space][lane = fq*16 + key%16][8 dims fq*8..]: the MFMA
                    // A-fragment order the attention kernel DMAs straight into LDS; a pair's keys are consecutive tiles of one head
                    // (pairs start at multiples of 16 rows). One store instruction = one whole 1 KiB fragment tile.
                    const int head0 = (nb - hidden) >> 5;          // this wave's 64 features = heads head0, head0+1
                    const int m = mb + j * 16;
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl) {
                        const half8 hi = *reinterpret_cast<const half8*>(wl + fr * 144 + (hl * 4 + fq) * 16);
                        const half8 lo = *reinterpret_cast<const half8*>(wl + CE_EPI_PLANE16 + fr * 144 + (hl * 4 + fq) * 16);
                        half_t* o = kf16 + (((size_t)(head0 + hl) * (m_pad >> 4) + (m >> 4)) * 64 + lane) * 8;
                        __builtin_nontemporal_store(hi, reinterpret_cast<half8*>(o));
                        __builtin_nontemporal_store(lo, reinterpret_cast<half8*>(o + kv_plane));
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            // EPI_QKV, V features: one fp16 plane of [64 features][32 tokens] at a time (rows 80 B), then
            // vf16[head][16-row tile of the PACKED row space][d half][lane = fq*16 + d%16][4 key slots = rows fq*4..+4 of the
            // tile]. The attention kernel reads a lane's 8 B of two adjacent tiles as one MFMA A fragment: 8 key slots in the
            // order in which the S^T accumulators of the two key tiles sit in a lane's registers, so P never leaves registers.
            // Tiles (not 32-row blocks) are the unit so that a pair may start at any multiple of 16 rows.
            const int head0 = (nb - 2 * hidden) >> 5;
#pragma unroll
            for (int kl = 0; kl < 2; ++kl) {
                const int m = mb + kl * 32;
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v = acc[i][kl * 2 + jj][r] + (r == 0 ? bv[i].x : r == 1 ? bv[i].y : r == 2 ? bv[i].z : bv[i].w);
                                const half_t hi = (half_t)v;
                                *reinterpret_cast<half_t*>(wl + (i * 16 + fq * 4 + r) * 80 + (jj * 16 + fr) * 2) =
                                    pl == 0 ? hi : (half_t)(v - (float)hi);
                            }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int hl = it >> 1, dh = it & 1;            // head, d half
                        const char* rowp = wl + (hl * 32 + dh * 16 + fr) * 80 + fq * 8;
                        const half4 h0 = *reinterpret_cast<const half4*>(rowp), h1 = *reinterpret_cast<const half4*>(rowp + 32);
                        half_t* o = vf16 + ((((size_t)(head0 + hl) * (m_pad >> 4) + (m >> 4)) * 2 + dh) * 64 + lane) * 4 + (pl ? kv_plane : 0);
                        __builtin_nontemporal_store(h0, reinterpret_cast<half4*>(o));
                        __builtin_nontemporal_store(h1, reinterpret_cast<half4*>(o + 512));   // the next 16-row tile
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (!has_next) break;
        // next tile: its steps 0 and 1 are resident (waited above); the stage this epilogue used is refilled by CE_ISSUE(2)
        // in the coming I-part, so every wave must be done reading it first
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CE_BAR
        work += n_slots;
        w_cur = w_nxt;
        x_cur = x_nxt;
        sbase = (sbase + nt) % 3;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // clamped tail re-loads of the last tile: retire them before exit
}

// ---- out-proj / FFN-down with bias + residual + LayerNorm in the epilogue (hidden = 384) -------------------------------
// The cross-encoder forward is bound by activation traffic, not by the matrix pipe (tools/ce_probe_build.sh: without MFMAs
// the forward is 9 % faster, without epilogue stores 35 %). The unfused form moves, per token and LayerNorm site,
// y32 out (1.5 KB), y32 + residual in (3 KB), x out (1.5 KB); fused, the residual comes in and the new stream goes out
// (1.5 KB each), and two kernel launches per layer disappear.
// Geometry: a persistent workgroup owns 128 tokens x ALL 384 features (so the row statistics stay on the CU): 8 waves =
// 2 feature halves (192 = 12 MFMA row blocks) x 4 token groups (32 = 2 column blocks), 96 accumulator registers per lane.
// Per 32-deep K-step the W slice (384 rows x [hi|lo] 128 B = 48 KiB, double-buffered: its source is L2-resident) and the
// token slice (128 rows x 128 B = 16 KiB, triple-buffered: its source is an HBM stream, two steps of lead) arrive by
// LDS-DMA as ONE continuous stream across tiles; one barrier per K-step (RAW: counted vmcnt(2); WAR: a slot is refilled
// only after the barrier that follows its last read). The W stage that is free after the last K-step is the epilogue's
// transpose scratch (6 KiB per wave). K must be a multiple of 192 (stages are then functions of the K-step alone).
#define lng_dma ce_bdma
#define LNG_W_STAGE (384 * 128)                    // 48 KiB
#define LNG_X_STAGE (128 * 128)                    // 16 KiB
#define LNG_LDS (2 * LNG_W_STAGE + 3 * LNG_X_STAGE)    // 144 KiB
#define LN_UNFUSED_MAX_ROWS (128 * 256)            // P x L up to which the residual + LayerNorm sites run unfused (finer tiles)
template <int TERMS>
__global__ __launch_bounds__(512) void ce_gemm_ln_kernel(const half_t* __restrict__ W, const half_t* __restrict__ X, int K,
                                                          const float* __restrict__ bias, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, half_t* __restrict__ stream16,
                                                          const int32_t* __restrict__ m_packed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int H = 384;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int m_end = m_packed[0];
    int tile = blockIdx.x;
    if (tile * 128 >= m_end) return;
    const int nt = K / CE_BK;
    const size_t ldk = (size_t)2 * K;
    // DMA source per thread: piece pc = p * 8 + wid covers rows pc * 8 .. + 8; lane -> row pc * 8 + (lane >> 3), 16-B position
    // lane & 7, source chunk = position ^ ((row >> 1) & 7) = position ^ ((wid & 1) * 4 + (lane >> 4))  (p * 8 is even)
    // Every DMA is a BUFFER load: descriptor (scalar registers: W, or the 128 token rows of one tile) + ONE per-lane byte
    // offset shared by all pieces + a scalar offset (piece, K-step). No address VGPRs besides that one (the first version
    // carried 6 + 4 pointer pairs through the main loop and spilled them), no address arithmetic in the loop, and reads
    // past a descriptor's end return zeros instead of faulting.
    const int schunk = (lane & 7) ^ ((wid & 1) * 4 + (lane >> 4));
    const unsigned voff = (unsigned)(((size_t)(wid * 8 + (lane >> 3)) * ldk + schunk * 8) * sizeof(half_t));
    const unsigned piece_b = (unsigned)(64 * ldk * sizeof(half_t));                           // 64 rows further, in bytes
    const unsigned tile_b = 2 * piece_b;                                                      // 128 token rows
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(W), 0, (int)(6 * piece_b), 0x00020000);
    __amdgpu_buffer_rsrc_t x_cur = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(X + (size_t)tile * 128 * ldk), 0, (int)tile_b, 0x00020000);
    __amdgpu_buffer_rsrc_t x_nxt = x_cur;
    bool has_next = false;
    char* const wring = smem;
    char* const xring = smem + 2 * LNG_W_STAGE;
    const int fr = lane & 15, fq = lane >> 4, sw = (fr >> 1) & 7;
    const int off_hi = fr * 128 + ((fq ^ sw) << 4), off_lo = fr * 128 + (((4 + fq) ^ sw) << 4);
    const int a_base = wm * 192 * 128, b_base = wn * 32 * 128;
#define LNG_ISSUE_W(u)   /* W slice of K-step (u mod nt) into W stage u & 1 */                                       \
    {                                                                                                                \
        const int ks_ = (u) < nt ? (u) : (u) - nt;                                                                  \
        char* st_ = wring + ((u) & 1) * LNG_W_STAGE;                                                                \
        _Pragma("unroll") for (int p_ = 0; p_ < 6; ++p_)                                                            \
            lng_dma(w_rs, voff, p_ * piece_b + ks_ * 128, st_ + p_ * 8192, wid);                                    \
    }
#define LNG_ISSUE_X(u)   /* token slice of step u: this tile, or steps 0.. of the next one (clamped at the very end) */ \
    {                                                                                                                \
        const bool nx_ = (u) >= nt && has_next;                                                                     \
        const int ks_ = (u) < nt ? (u) : (has_next ? (u) - nt : nt - 1);                                            \
        char* st_ = xring + ((u) % 3) * LNG_X_STAGE;                                                                \
        if (nx_) {                                                                                                  \
            lng_dma(x_nxt, voff, ks_ * 128, st_, wid);                                                              \
            lng_dma(x_nxt, voff, piece_b + ks_ * 128, st_ + 8192, wid);                                             \
        } else {                                                                                                    \
            lng_dma(x_cur, voff, ks_ * 128, st_, wid);                                                              \
            lng_dma(x_cur, voff, piece_b + ks_ * 128, st_ + 8192, wid);                                             \
        }                                                                                                           \
    }
    LNG_ISSUE_W(0)
    LNG_ISSUE_X(0)
    LNG_ISSUE_X(1)
    for (bool first = true;; first = false) {
        const int m0 = tile * 128;
        {
            const int nx = tile + gridDim.x;
            has_next = nx * 128 < m_end;
            if (has_next) x_nxt = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(X + (size_t)nx * 128 * ldk), 0, (int)tile_b, 0x00020000);
        }
        f32x4 acc[12][2];
#pragma unroll
        for (int i = 0; i < 12; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        for (int t = 0; t < nt; ++t) {
            // W(t) and X(t) have landed once at most the 2 pieces of X(t+1) are outstanding; the barrier makes that hold for
            // every wave's pieces and closes every wave's reads of the slots refilled below
            // (step 0 of a continued tile: both slices were waited for before the previous epilogue's stores; a counted wait
            // here would only wait for those stores)
            if (first || t > 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            CE_BAR
            // the 8 DMA pieces of this step are issued one or two at a time BEHIND the MFMA groups below: a piece costs its wave
            // 60-180 issue cycles, and with all eight up front both waves of a SIMD sat in them at the same time, the matrix pipe
            // idle (W(t+1) first: it is needed one step from now; the two X(t+2) pieces last)
            const int wu_ = t + 1, xu_ = t + 2;
            const unsigned wks_ = (unsigned)(wu_ < nt ? wu_ : wu_ - nt) * 128u;
            char* const wst_ = wring + (wu_ & 1) * LNG_W_STAGE;
            const bool xnx_ = xu_ >= nt && has_next;
            const unsigned xks_ = (unsigned)(xu_ < nt ? xu_ : (has_next ? xu_ - nt : nt - 1)) * 128u;
            char* const xst_ = xring + (xu_ % 3) * LNG_X_STAGE;
            const __amdgpu_buffer_rsrc_t xrs_ = xnx_ ? x_nxt : x_cur;
            const char* ws = wring + (t & 1) * LNG_W_STAGE + a_base;
            const char* xs = xring + (t % 3) * LNG_X_STAGE + b_base;
            half8 bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *reinterpret_cast<const half8*>(xs + j * 16 * 128 + off_hi);
                if (TERMS & 2) bl[j] = *reinterpret_cast<const half8*>(xs + j * 16 * 128 + off_lo);
            }
            // The wave's 12 feature blocks go through TWO fragment slots of two blocks each (32 VGPRs, as one third of them did
            // before), refilled right behind the MFMAs that consumed them: the reads of pair p+2 fly under the 12 MFMAs of pair
            // p+1, so a K-step exposes one LDS round trip instead of three (the waits are the compiler's counted lgkmcnt).
            half8 ah[2][2], al[2][2];
#define LNG_READ_PAIR(p, s)                                                                                           \
            _Pragma("unroll") for (int ii = 0; ii < 2; ++ii) {                                                        \
                ah[s][ii] = *reinterpret_cast<const half8*>(ws + ((p) * 2 + ii) * 16 * 128 + off_hi);                 \
                if (TERMS & 1) al[s][ii] = *reinterpret_cast<const half8*>(ws + ((p) * 2 + ii) * 16 * 128 + off_lo);  \
            }
            LNG_READ_PAIR(0, 0)
            LNG_READ_PAIR(1, 1)
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4& a = acc[p * 2 + ii][j];
                        if (TERMS & 1) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[p & 1][ii], bh[j], a, 0, 0, 0);
                        if (TERMS & 2) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p & 1][ii], bl[j], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p & 1][ii], bh[j], a, 0, 0, 0);
                    }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (p + 2 < 6) LNG_READ_PAIR(p + 2, p & 1)
                if (p < 3) {
                    lng_dma(w_rs, voff, (2 * p) * piece_b + wks_, wst_ + (2 * p) * 8192, wid);
                    lng_dma(w_rs, voff, (2 * p + 1) * piece_b + wks_, wst_ + (2 * p + 1) * 8192, wid);
                } else if (p < 5) {
                    lng_dma(xrs_, voff, (p - 3) * piece_b + xks_, xst_ + (p - 3) * 8192, wid);
                }
            }
#undef LNG_READ_PAIR
        }
        // ---- epilogue. acc[i][j][r] = sum for feature wm*192 + i*16 + fq*4 + r, token m0 + wn*32 + j*16 + fr.
        // Every wave is past its reads of W stage (nt-1)&1 = 1 after this barrier; its refill (step 1 of the next tile) comes
        // after the next tile's first barrier, so the stage is this epilogue's scratch: 6 KiB per wave. Steps 0 and 1 of the next
        // tile are in flight: retire them here, where no store is outstanding yet.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CE_BAR
        char* wl = wring + LNG_W_STAGE + wid * 6144;                 // [16 tokens][64 features] fp32, rows 272 B
        float* st_sum = reinterpret_cast<float*>(wl + 4352);         // [32] per-token partial sums of this wave's 192 features
        float* st_sq = st_sum + 32;
        const float* pr_sum = reinterpret_cast<const float*>(wring + LNG_W_STAGE + (wid ^ 4) * 6144 + 4352);   // the other feature half
        const float* pr_sq = pr_sum + 32;
        const int rr = lane >> 4, cc = lane & 15;
        // epilogue addresses = uniform base + ONE per-lane byte offset + compile-time constants. The offsets are made opaque
        // here, inside the tile loop, so that the compiler derives each address where it is used instead of hoisting ~70
        // address registers out of the loop and carrying them through the main loop (which spilled its DMA pointers).
        unsigned so = (unsigned)((rr * 2 * H + (wm * 6 + (cc >> 3)) * 64 + (cc & 7) * 4) * sizeof(half_t));   // stream row, split index
        unsigned fo = (unsigned)((wm * 192 + cc * 4) * sizeof(float));                                          // bias / gamma / beta
        asm volatile("" : "+v"(so), "+v"(fo));
        char* const srow = reinterpret_cast<char*>(stream16 + (size_t)(m0 + wn * 32) * 2 * H);
        f32x4 vv[2][3][4];                                           // [token block][64-feature group][4 rows per lane]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 3; ++g) {
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
                    *reinterpret_cast<f32x4*>(wl + fr * 272 + (ii * 16 + fq * 4) * 4) = acc[g * 4 + ii][j];
                __builtin_amdgcn_wave_barrier();
                const float4 bv = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bias) + fo + g * 256);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 4 + rr;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(wl + row * 272 + cc * 16);
                    // token m0 + wn*32 + j*16 + row, features wm*192 + g*64 + cc*4 .. +4: K group wm*6 + g*2 + (cc>>3)
                    const half_t* rp = reinterpret_cast<const half_t*>(srow + so + ((j * 16 + it * 4) * 2 * H + g * 128) * sizeof(half_t));
                    const half4 rh = *reinterpret_cast<const half4*>(rp), rl = *reinterpret_cast<const half4*>(rp + 32);
                    vv[j][g][it] = (f32x4){v[0] + bv.x + ((float)rh[0] + (float)rl[0]), v[1] + bv.y + ((float)rh[1] + (float)rl[1]),
                                           v[2] + bv.z + ((float)rh[2] + (float)rl[2]), v[3] + bv.w + ((float)rh[3] + (float)rl[3])};
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);        // one pass at a time: hoisting all 24 residual loads costs ~100 VGPRs
            }
        // row statistics: two passes (mean, then centred squares) as the stand-alone LayerNorm; a token's 384 features are
        // spread over 16 lanes x 3 groups in this wave and as many in the wave that owns the other feature half
        float mean[2][4], rstd[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                float sm = 0.f;
#pragma unroll
                for (int g = 0; g < 3; ++g) sm += (vv[j][g][it][0] + vv[j][g][it][1]) + (vv[j][g][it][2] + vv[j][g][it][3]);
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) sm += __shfl_xor(sm, o);
                mean[j][it] = sm;
                if (cc == 0) st_sum[j * 16 + it * 4 + rr] = sm;
            }
        CE_BAR
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float mu = (mean[j][it] + pr_sum[j * 16 + it * 4 + rr]) * (1.0f / (float)H);
                mean[j][it] = mu;
                float q = 0.f;
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = vv[j][g][it][e] - mu; q += d * d; }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o);
                rstd[j][it] = q;
                if (cc == 0) st_sq[j * 16 + it * 4 + rr] = q;
            }
        CE_BAR
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int it = 0; it < 4; ++it)
                rstd[j][it] = 1.0f / sqrtf((rstd[j][it] + pr_sq[j * 16 + it * 4 + rr]) * (1.0f / (float)H) + eps);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float4 gv = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(gamma) + fo + g * 256);
            const float4 be = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(beta) + fo + g * 256);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const f32x4 v = vv[j][g][it];
                    const float mu = mean[j][it], rs = rstd[j][it];
                    half_t* o = reinterpret_cast<half_t*>(srow + so + ((j * 16 + it * 4) * 2 * H + g * 128) * sizeof(half_t));
                    store_split4(o, 32, (v[0] - mu) * rs * gv.x + be.x, (v[1] - mu) * rs * gv.y + be.y, (v[2] - mu) * rs * gv.z + be.z,
                                 (v[3] - mu) * rs * gv.w + be.w);
                }
        }
        if (!has_next) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // scratch reads retired before the stage is refilled
        tile += gridDim.x;
        x_cur = x_nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // clamped tail re-loads: retire them before exit
#undef LNG_ISSUE_W
#undef LNG_ISSUE_X
}

// ---- the whole FFN in ONE kernel (hidden = 384): up-projection + bias + erf-GELU + down-projection + bias + residual + LayerNorm
// The 1536-wide intermediate activation never reaches HBM: it was the largest tensor of the forward (6 of the ~15 KB stored and
// 12 of the 33.8 KB moved per token and layer; profiles/r02_i: 1.02 TB of HBM traffic per 25,600-pair forward) and its round
// trip cost one launch and one store-bound epilogue per layer.
// A persistent workgroup owns 128 tokens. The intermediate is produced and consumed 128 features (one CHUNK) at a time:
//   phase A  H_c^T[128 feat x 128 tok] = W1[chunk c] . X^T, K = 384: 12 K-steps of 32 (W1 slice 16 KiB + token slice 16 KiB per
//            step, three 32-KiB ring slots, two steps of lead); 8 waves = 2 feature halves x 4 token groups, 32 accumulators;
//   E        bias + GELU, split fp16, written to LDS in the operand layout phase B reads ([K group][token][hi 32 | lo 32]):
//            64 KiB, never in HBM;
//   phase B  Y^T[384 x 128 tok] += W2[:, chunk c] . H_c^T, K = 128: 4 K-steps (W2 slice 48 KiB per step, two 48-KiB ring slots
//            laid over the SAME 96 KiB as phase A's three); 8 waves = 2 feature halves x 4 token groups, 96 accumulators that live
//            across all 12 chunks;
// then the bias + residual + LayerNorm epilogue of ce_gemm_ln_kernel. LDS = 96 KiB ring + 64 KiB H = 160 KiB.
// One barrier per step; every DMA piece is issued right after the barrier that closes the last read of the bytes it overwrites:
//   A_t (t <= 9) issues A_{t+2};  A_11 issues B_0;  E issues B_1;  B_1 issues B_2;  B_2 issues B_3;  B_3 issues the next A_0
//   (next chunk or next tile);  A_0 issues A_1 and A_2.
// Counted waits (pieces per wave: 4 per A step, 6 per B step): A_1..A_10 vmcnt(4), B_0 vmcnt(6), every other step vmcnt(0).
// The first-projection bias of chunk c + 1 is loaded during B_1 of chunk c (ahead of B_2's pieces: B_2's own wait covers it).
#define FFN_RING (96 * 1024)
#define FFN_HBUF (64 * 1024)
#define FFN_LDS (FFN_RING + FFN_HBUF)
#define FFN_CH 128                                // intermediate features per chunk
#define FFN_FUSED_MIN_ROWS (5120 * 256)            // P x L from which the fused kernel is used (below: the two-launch form)
#define MX_MIN_ROWS 0                              // P x L from which the MX forward (ce_mx.h) runs instead of the split-fp16 kernels: every size. tools/ce_mx_sweep.py: 13 pairs tie (0.82 | 0.80 ms), 25 pairs and up MX is 10-19 % faster - and a pair's logit must not depend on how a batch was split over ranks or chunks, so the rule is by SHAPE only
template <int TERMS>
__global__ __launch_bounds__(512) void ce_ffn_ln_kernel(const half_t* __restrict__ W1, const float* __restrict__ b1,
                                                         const half_t* __restrict__ W2, const float* __restrict__ b2, int F,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         half_t* __restrict__ stream16, const int32_t* __restrict__ m_packed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int H = 384, NTA = H / CE_BK, NTB = FFN_CH / CE_BK;     // 12 K-steps up, 4 K-steps down per chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int m_end = m_packed[0];
    int tile = blockIdx.x;
    if (tile * 128 >= m_end) return;
    const int n_chunks = F / FFN_CH;
    // DMA sources (see ce_gemm_ln_kernel): a piece = 64 rows x 128 B, one 1-KiB instruction per wave; per-lane byte offset for rows
    // of 2*H halfs (W1, X) and of 2*F halfs (W2); descriptors in scalar registers
    const int schunk = (lane & 7) ^ ((wid & 1) * 4 + (lane >> 4));
    const unsigned row_a = (unsigned)(2 * H * sizeof(half_t)), row_b = (unsigned)(2 * F * sizeof(half_t));
    const unsigned voff_a = (unsigned)((wid * 8 + (lane >> 3)) * row_a + schunk * 16);
    const unsigned voff_b = (unsigned)((wid * 8 + (lane >> 3)) * row_b + schunk * 16);
    const __amdgpu_buffer_rsrc_t w1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(W1), 0, (int)((size_t)F * row_a), 0x00020000);
    const __amdgpu_buffer_rsrc_t w2_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(W2), 0, (int)((size_t)H * row_b), 0x00020000);
    __amdgpu_buffer_rsrc_t x_cur = __builtin_amdgcn_make_buffer_rsrc(stream16 + (size_t)tile * 128 * 2 * H, 0, (int)(128 * row_a), 0x00020000);
    char* const hbuf = smem + FFN_RING;
    const int fr = lane & 15, fq = lane >> 4, sw = (fr >> 1) & 7;
    const int off_hi = fr * 128 + ((fq ^ sw) << 4), off_lo = fr * 128 + (((4 + fq) ^ sw) << 4);
    // phase A: this wave's 64 intermediate features (rows of the W1 slice) x 32 tokens; phase B: 192 output features x 32 tokens
    const int a_base_A = wm * 64 * 128, b_base_A = 16384 + wn * 32 * 128;
    const int a_base_B = wm * 192 * 128, b_base_B = wn * 32 * 128;
#define FFN_XSRC(x) x
    // one 8-KiB piece (64 rows x 128 B) at a time, so that the issue cost of a transfer (60-180 cycles per piece per wave) is
    // spread behind the MFMA groups of a step instead of standing in front of them
#define FFN_PIECE_A(c, t, xrs, k)  /* piece k of step t of chunk c: 0, 1 = W1 slice halves, 2, 3 = token slice halves -> A slot t % 3 */ \
    {                                                                                                                      \
        char* st_ = smem + ((t) % 3) * 32768 + (k) * 8192;                                                                \
        if ((k) < 2) ce_bdma(w1_rs, voff_a, (unsigned)(c) * (FFN_CH * row_a) + (unsigned)(t) * 128u + (unsigned)(k) * (64 * row_a), st_, wid); \
        else ce_bdma(FFN_XSRC(xrs), voff_a, (unsigned)((k) - 2) * (64 * row_a) + (unsigned)(t) * 128u, st_, wid);          \
    }
#define FFN_ISSUE_A(c, t, xrs) { FFN_PIECE_A(c, t, xrs, 0) FFN_PIECE_A(c, t, xrs, 1) FFN_PIECE_A(c, t, xrs, 2) FFN_PIECE_A(c, t, xrs, 3) }
#define FFN_PIECE_B(c, u, k)       /* piece k (0..5: 64 output rows each) of the W2 slice of chunk c, K-step u -> B slot u & 1 */ \
    ce_bdma(w2_rs, voff_b, (unsigned)(k) * 64u * row_b + ((unsigned)(c) * NTB + (unsigned)(u)) * 128u, smem + ((u) & 1) * 49152 + (k) * 8192, wid);
    f32x4 acc[12][2];                                  // phase B accumulators: live across the 12 chunks of a tile
    float4 bv[4];                                      // first-projection bias of the current chunk: features wm*64 + i*16 + fq*4 ..+4
#define FFN_LOAD_BIAS(c) \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) bv[i_] = *reinterpret_cast<const float4*>(b1 + (c) * FFN_CH + wm * 64 + i_ * 16 + fq * 4);
    FFN_LOAD_BIAS(0)
    FFN_ISSUE_A(0, 0, x_cur)
    for (bool first = true;; first = false) {
        const int m0 = tile * 128;
        const int nx = tile + gridDim.x;
        const bool has_next = nx * 128 < m_end;
        __amdgpu_buffer_rsrc_t x_nxt = x_cur;
        if (has_next) x_nxt = __builtin_amdgcn_make_buffer_rsrc(stream16 + (size_t)nx * 128 * 2 * H, 0, (int)(128 * row_a), 0x00020000);
#pragma unroll
        for (int i = 0; i < 12; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        for (int c = 0; c < n_chunks; ++c) {
            // ================= phase A: 12 K-steps =================
            f32x4 ha[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { ha[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; ha[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int t = 0; t < NTA; ++t) {
                // own pieces of step t landed (step 0 of a continued tile was waited for inside the previous epilogue)
                if (t == 0 || t == NTA - 1) { if (!(t == 0 && c == 0 && !first)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                CE_BAR
                const char* st = smem + (t % 3) * 32768;
                half8 bh[2], bl[2], ah[4], al[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bh[j] = *reinterpret_cast<const half8*>(st + b_base_A + j * 2048 + off_hi);
                    if (TERMS & 2) bl[j] = *reinterpret_cast<const half8*>(st + b_base_A + j * 2048 + off_lo);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ah[i] = *reinterpret_cast<const half8*>(st + a_base_A + i * 2048 + off_hi);
                    if (TERMS & 1) al[i] = *reinterpret_cast<const half8*>(st + a_base_A + i * 2048 + off_lo);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (TERMS & 1) ha[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], ha[i][j], 0, 0, 0);
                        if (TERMS & 2) ha[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], ha[i][j], 0, 0, 0);
                        ha[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], ha[i][j], 0, 0, 0);
                    }
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    // the pieces this step owes, one or two behind each MFMA group (W1 halves first: the X halves are L2-hot)
                    if (t == 0) { FFN_PIECE_A(c, 1 + (i >> 1), x_cur, (i & 1) * 2) FFN_PIECE_A(c, 1 + (i >> 1), x_cur, (i & 1) * 2 + 1) }
                    else if (t + 2 < NTA) FFN_PIECE_A(c, t + 2, x_cur, i)
                    else if (t == NTA - 1) { FFN_PIECE_B(c, 0, i) if (i >= 2) FFN_PIECE_B(c, 0, i + 2) }
                }
            }
            // ================= E: bias + GELU -> split fp16 in LDS, phase B's operand layout =================
            CE_BAR                                          // every wave is past its reads of the last A step: B slot 1 is free
            // ha[i][j][r] = H^T[feature wm*64 + i*16 + fq*4 + r][token wn*32 + j*16 + fr]; K group = feature / 32 = wm*2 + (i>>1),
            // inside it the 4 features sit at half index (i&1)*16 + fq*4 .. +4: 16-B piece (i&1)*2 + (fq>>1), second half if fq odd
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int tk = wn * 32 + j * 16 + fr;
                    const int swt = (tk >> 1) & 7, pc = (i & 1) * 2 + (fq >> 1);
                    char* row = hbuf + (wm * 2 + (i >> 1)) * 16384 + tk * 128 + (fq & 1) * 8;
                    const float v0 = ce_gelu(ha[i][j][0] + bv[i].x), v1 = ce_gelu(ha[i][j][1] + bv[i].y);
                    const float v2 = ce_gelu(ha[i][j][2] + bv[i].z), v3 = ce_gelu(ha[i][j][3] + bv[i].w);
                    const half4 hi = {(half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3};
                    const half4 lo = {(half_t)(v0 - (float)hi[0]), (half_t)(v1 - (float)hi[1]), (half_t)(v2 - (float)hi[2]), (half_t)(v3 - (float)hi[3])};
                    *reinterpret_cast<half4*>(row + ((pc ^ swt) << 4)) = hi;
                    *reinterpret_cast<half4*>(row + (((4 + pc) ^ swt) << 4)) = lo;
                    if (i * 2 + j < 6) FFN_PIECE_B(c, 1, i * 2 + j)          // B_1's six pieces, one behind each block's GELU
                }
            // ================= phase B: 4 K-steps =================
#pragma unroll
            for (int u = 0; u < NTB; ++u) {
                if (u == 0) asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");       // B_0 landed; the H writes are done
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                CE_BAR
                if (u == 1) { if (c + 1 < n_chunks) { FFN_LOAD_BIAS(c + 1) } else { FFN_LOAD_BIAS(0) } }
                const char* ws = smem + (u & 1) * 49152 + a_base_B;
                const char* xs = hbuf + u * 16384 + b_base_B;
                half8 bh[2], bl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bh[j] = *reinterpret_cast<const half8*>(xs + j * 2048 + off_hi);
                    if (TERMS & 2) bl[j] = *reinterpret_cast<const half8*>(xs + j * 2048 + off_lo);
                }
                half8 ah[2][2], al[2][2];
#define FFN_READ_PAIR(p, s)                                                                                           \
                _Pragma("unroll") for (int ii = 0; ii < 2; ++ii) {                                                    \
                    ah[s][ii] = *reinterpret_cast<const half8*>(ws + ((p) * 2 + ii) * 2048 + off_hi);                 \
                    if (TERMS & 1) al[s][ii] = *reinterpret_cast<const half8*>(ws + ((p) * 2 + ii) * 2048 + off_lo);  \
                }
                FFN_READ_PAIR(0, 0)
                FFN_READ_PAIR(1, 1)
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            f32x4& a = acc[p * 2 + ii][j];
                            if (TERMS & 1) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[p & 1][ii], bh[j], a, 0, 0, 0);
                            if (TERMS & 2) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p & 1][ii], bl[j], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p & 1][ii], bh[j], a, 0, 0, 0);
                        }
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (p + 2 < 6) FFN_READ_PAIR(p + 2, p & 1)
                    // what this step owes, one piece behind each MFMA group: B_1 -> B_2, B_2 -> B_3, B_3 -> the next A_0
                    if (u == 1) FFN_PIECE_B(c, 2, p)
                    else if (u == 2) FFN_PIECE_B(c, 3, p)
                    else if (u == 3 && p < 4) {
                        if (c + 1 < n_chunks) FFN_PIECE_A(c + 1, 0, x_cur, p)
                        else if (has_next) FFN_PIECE_A(0, 0, x_nxt, p)
                    }
                }
#undef FFN_READ_PAIR
            }
        }
        // ---- epilogue: bias + residual + LayerNorm, as ce_gemm_ln_kernel. acc[i][j][r] = sum for feature wm*192 + i*16 + fq*4 + r,
        // token m0 + wn*32 + j*16 + fr. The last B step read B slot 1 = ring bytes [48K, 96K): after this barrier it is the
        // transpose scratch (6 KiB per wave); the next tile's A_0 pieces are in flight into [0, 32K).
        CE_BAR
        char* wl = smem + 49152 + wid * 6144;                        // [16 tokens][64 features] fp32, rows 272 B
        float* st_sum = reinterpret_cast<float*>(wl + 4352);         // [32] per-token partial sums of this wave's 192 features
        float* st_sq = st_sum + 32;
        const float* pr_sum = reinterpret_cast<const float*>(smem + 49152 + (wid ^ 4) * 6144 + 4352);   // the other feature half
        const float* pr_sq = pr_sum + 32;
        const int rr = lane >> 4, cc = lane & 15;
        unsigned so = (unsigned)((rr * 2 * H + (wm * 6 + (cc >> 3)) * 64 + (cc & 7) * 4) * sizeof(half_t));   // stream row, split index
        unsigned fo = (unsigned)((wm * 192 + cc * 4) * sizeof(float));                                          // bias / gamma / beta
        asm volatile("" : "+v"(so), "+v"(fo));
        char* const srow = reinterpret_cast<char*>(stream16 + (size_t)(m0 + wn * 32) * 2 * H);
        f32x4 vv[2][3][4];                                           // [token block][64-feature group][4 rows per lane]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 3; ++g) {
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
                    *reinterpret_cast<f32x4*>(wl + fr * 272 + (ii * 16 + fq * 4) * 4) = acc[g * 4 + ii][j];
                __builtin_amdgcn_wave_barrier();
                const float4 b2v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(b2) + fo + g * 256);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 4 + rr;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(wl + row * 272 + cc * 16);
                    const half_t* rp = reinterpret_cast<const half_t*>(srow + so + ((j * 16 + it * 4) * 2 * H + g * 128) * sizeof(half_t));
                    const half4 rh = *reinterpret_cast<const half4*>(rp), rl = *reinterpret_cast<const half4*>(rp + 32);
                    vv[j][g][it] = (f32x4){v[0] + b2v.x + ((float)rh[0] + (float)rl[0]), v[1] + b2v.y + ((float)rh[1] + (float)rl[1]),
                                           v[2] + b2v.z + ((float)rh[2] + (float)rl[2]), v[3] + b2v.w + ((float)rh[3] + (float)rl[3])};
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        float mean[2][4], rstd[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                float sm = 0.f;
#pragma unroll
                for (int g = 0; g < 3; ++g) sm += (vv[j][g][it][0] + vv[j][g][it][1]) + (vv[j][g][it][2] + vv[j][g][it][3]);
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) sm += __shfl_xor(sm, o);
                mean[j][it] = sm;
                if (cc == 0) st_sum[j * 16 + it * 4 + rr] = sm;
            }
        CE_BAR
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float mu = (mean[j][it] + pr_sum[j * 16 + it * 4 + rr]) * (1.0f / (float)H);
                mean[j][it] = mu;
                float q = 0.f;
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = vv[j][g][it][e] - mu; q += d * d; }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o);
                rstd[j][it] = q;
                if (cc == 0) st_sq[j * 16 + it * 4 + rr] = q;
            }
        CE_BAR
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int it = 0; it < 4; ++it)
                rstd[j][it] = 1.0f / sqrtf((rstd[j][it] + pr_sq[j * 16 + it * 4 + rr]) * (1.0f / (float)H) + eps);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float4 gv = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(gamma) + fo + g * 256);
            const float4 be = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(beta) + fo + g * 256);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const f32x4 v = vv[j][g][it];
                    const float mu = mean[j][it], rs = rstd[j][it];
                    half_t* o = reinterpret_cast<half_t*>(srow + so + ((j * 16 + it * 4) * 2 * H + g * 128) * sizeof(half_t));
                    store_split4(o, 32, (v[0] - mu) * rs * gv.x + be.x, (v[1] - mu) * rs * gv.y + be.y, (v[2] - mu) * rs * gv.z + be.z,
                                 (v[3] - mu) * rs * gv.w + be.w);
                }
        }
        if (!has_next) break;
        // the next tile's A_0 pieces (issued before this epilogue's loads) have landed: every load above was waited for in order.
        // The scratch is overwritten by A_1 / A_2 of the next tile only after its first barrier.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tile = nx;
        x_cur = x_nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef FFN_ISSUE_A
#undef FFN_PIECE_A
#undef FFN_PIECE_B
#undef FFN_XSRC
#undef FFN_LOAD_BIAS
}

// ---- LayerNorm helpers: one wave per token row of `hidden` floats (hidden % 64 == 0, <= 1024) --------------
template <int PER>
__device__ __forceinline__ void wave_layernorm(float (&v)[PER], const float* __restrict__ g, const float* __restrict__ b,
                                               int hidden, float eps, int lane, half_t* __restrict__ o16) {
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
        const half_t hi = (half_t)y;
        o16[SPLIT_IDX(c)] = hi;
        o16[SPLIT_IDX(c) + 32] = (half_t)(y - (float)hi);
    }
}

// ---- packing: pair p owns len_p rounded up to 16 rows; offsets by one block-wide scan, then the row -> pair map
__global__ __launch_bounds__(1024) void ce_pack_scan_kernel(const int32_t* __restrict__ lens, int P, int L, int32_t* __restrict__ pair_off,
                                                             int32_t* __restrict__ m_packed) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (P + 1023) / 1024;
    const int b = tid * per, e = min(P, b + per);
    int s = 0;
    for (int p = b; p < e; ++p) s += (max(1, min(lens[p], L)) + 15) & ~15;
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int off = part[tid] - s;
    for (int p = b; p < e; ++p) {
        pair_off[p] = off;
        off += (max(1, min(lens[p], L)) + 15) & ~15;
    }
    if (tid == 1023) { pair_off[P] = part[1023]; m_packed[0] = part[1023]; }
}

__global__ void ce_pack_rows_kernel(const int32_t* __restrict__ pair_off, int P, int L, int64_t rows_total,
                                    int32_t* __restrict__ row_pair) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows_total && i >= pair_off[P]) row_pair[i] = -1;             // tail up to the allocated rows
    if (i >= (int64_t)P * L) return;
    const int p = (int)(i / L), t = (int)(i % L);
    if (t < pair_off[p + 1] - pair_off[p]) row_pair[pair_off[p] + t] = p;
}

template <int PER>
__global__ __launch_bounds__(256) void ce_embed_ln_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ tt,
                                                           const float* __restrict__ word, const float* __restrict__ pos,
                                                           const float* __restrict__ type, const float* __restrict__ g,
                                                           const float* __restrict__ b, const int32_t* __restrict__ m_packed,
                                                           const int32_t* __restrict__ row_pair, const int32_t* __restrict__ pair_off,
                                                           int L, int hidden, int vocab, float eps, half_t* __restrict__ x16) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m_packed[0]) return;
    const int pr = row_pair[row];
    const int p = (int)row - pair_off[pr];                    // position inside the pair (< L: the rounded length never exceeds L)
    const size_t src = (size_t)pr * L + p;
    int id = ids[src];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const int ty = tt[src] != 0;
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = lane + i * 64;
        v[i] = word[(size_t)id * hidden + c] + type[(size_t)ty * hidden + c] + pos[(size_t)p * hidden + c];
    }
    wave_layernorm<PER>(v, g, b, hidden, eps, lane, x16 + row * 2 * hidden);
}

template <int PER>
__global__ __launch_bounds__(256) void ce_layernorm_kernel(const float* __restrict__ y32, const float* __restrict__ g,
                                                            const float* __restrict__ b, const int32_t* __restrict__ m_packed,
                                                            int hidden, float eps, half_t* __restrict__ x16) {
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= m_packed[0]) return;
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = y32[tok * hidden + lane + i * 64];
    wave_layernorm<PER>(v, g, b, hidden, eps, lane, x16 + tok * 2 * hidden);
}

// ---- attention: d_head must be 32. One block per (head, pair); every wave owns QB consecutive 16-query blocks.
// K and V of the (pair, head) arrive in LDS by LDS-DMA, already in MFMA fragment order (written that way by the QKV
// epilogue) as 1 KiB tiles of 16 packed rows: a K fragment is one conflict-free ds_read_b128 at lane*16, a V fragment two
// ds_read_b64 at lane*8 (the lane's key slots of two adjacent tiles); both are shared by all waves of the block.
// All operands are split fp16 (hi + lo plane): S and P.V are 3 MFMAs each. S is computed TRANSPOSED (A = K rows,
// B = Q rows): the accumulator lane (fr, fq) then holds query fr x keys fq*4..+4, which IS the B-operand layout of the
// next MFMA if the 32 k-slots of a key block are numbered (fq, e) -> key fq*4 + e (e < 4) | 16 + fq*4 + e - 4: V is
// stored in that slot order, so P goes registers -> MFMA without touching LDS. Online (flash-style) softmax over
// 32-key blocks in the exp2 domain; key blocks past the pair's length are skipped, the boundary block is masked.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t ce_pk(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
}
__device__ __forceinline__ float ce_trunc10(float e) {     // e with the mantissa cut to 10 bits: exactly a fp16 value
    return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, e) & 0xFFFFE000u);
}
__device__ __forceinline__ void ce_dma_at(const half_t* __restrict__ g, char* lds_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}

// MX = true (the hi16 + lo8 forward, ce_mx.h): Q arrives in the same fragment order as K (q16 = qf16[head][16-row tile][lane][8], lo
// plane kv_plane further) and the context leaves in the image layout of the out-projection's token operand (ctx16 = ctx8 bytes).
// DIRECT (the [CLS]-only last layer: one 16-query block per pair): ONE wave per (head, pair) reads the K / V fragment tiles straight
// from global memory - every tile is used by that one wave, so staging it in LDS only costs a 64-KiB allocation per workgroup.
template <int QB, bool MX = false, bool DIRECT = false>
__global__ __launch_bounds__(1024) void ce_attention_kernel(const half_t* __restrict__ q16,
                                                             const half_t* __restrict__ kf16, const half_t* __restrict__ vf16,
                                                             size_t kv_plane, const int32_t* __restrict__ lens,
                                                             const int32_t* __restrict__ pair_off, int L, int hidden,
                                                             int heads, int m_pad, half_t* __restrict__ ctx16, int max_qblocks = 1 << 20) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const int head = blockIdx.x, pair = blockIdx.y;
    const int len = max(1, min(lens[pair], L));
    const int fr = lane & 15, fq = lane >> 4;
    const int po = pair_off[pair], Lp = pair_off[pair + 1] - po;     // this pair's packed rows: len rounded up to 16
    const int nt = Lp >> 4;                                           // the pair's 16-row tiles; an odd count leaves the second
    const int nkb = (len + 31) >> 5;                                  // half of the last 32-key block outside the pair (masked)
    const size_t plane_b = (size_t)L * 64;                            // bytes of one K (or V) plane of this (pair, head)
    const size_t t0 = ((size_t)head * (m_pad >> 4) + (po >> 4)) * 512;                      // first tile of this (head, pair), in halfs
    // fragment reads below address tile c of a plane at byte c * 1024 + (lane's offset): the same bytes in LDS and in global memory
    const char* const k_hi = DIRECT ? reinterpret_cast<const char*>(kf16 + t0) : smem;
    const char* const k_lo = DIRECT ? reinterpret_cast<const char*>(kf16 + t0 + kv_plane) : smem + plane_b;
    const char* const v_hi = DIRECT ? reinterpret_cast<const char*>(vf16 + t0) : smem + 2 * plane_b;
    const char* const v_lo = DIRECT ? reinterpret_cast<const char*>(vf16 + t0 + kv_plane) : smem + 3 * plane_b;
    if (!DIRECT) {
        char* const sk_hi = smem;
        char* const sk_lo = smem + plane_b;
        char* const sv_hi = smem + 2 * plane_b;
        char* const sv_lo = smem + 3 * plane_b;
        const size_t g0 = (((size_t)head * (m_pad >> 4) + (po >> 4)) * 64 + lane) * 8;   // K and V tiles are both 1 KiB per plane
        for (int c = wv; c < nt; c += nwaves) {
            ce_dma_at(kf16 + g0 + (size_t)c * 512, sk_hi + c * 1024);
            ce_dma_at(kf16 + g0 + kv_plane + (size_t)c * 512, sk_lo + c * 1024);
            ce_dma_at(vf16 + g0 + (size_t)c * 512, sv_hi + c * 1024);
            ce_dma_at(vf16 + g0 + kv_plane + (size_t)c * 512, sv_lo + c * 1024);
        }
        // an odd tile count leaves the second half of the last 32-key block outside the pair: P is 0 there (masked keys), V must be
        // finite - the never-staged tile is zeroed once instead of selecting zeros at every fragment read (DIRECT reads the next
        // pair's rows or the zeroed slack of the buffer there: finite too)
        if ((nt & 1) && wv == 0) {
            *reinterpret_cast<u32x4*>(sv_hi + nt * 1024 + lane * 16) = (u32x4){0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(sv_lo + nt * 1024 + lane * 16) = (u32x4){0u, 0u, 0u, 0u};
        }
    }
    const size_t row0 = (size_t)po;
    const int qb0 = wv * QB;
    // max_qblocks: only the first max_qblocks 16-query blocks of a pair are computed (the last layer of a classifier needs the [CLS]
    // row alone; the other waves still help with the K / V DMA)
    const bool has_rows = qb0 * 16 < Lp && qb0 < max_qblocks;         // waves past the pair's rows only helped with the DMA
    // B operand = Q rows (query fr of block b, dims 8*fq..+8)
    half8 qh[QB], ql[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        // split-row layout: a head's 32 dims are one K group = [hi 32 | lo 32] halfs
        const int qb = (qb0 + b) * 16 < Lp ? qb0 + b : qb0;           // a block past the pair's rows is computed but not stored
        if (MX) {
            const half_t* qp = q16 + (((size_t)head * (m_pad >> 4) + (po >> 4) + qb) * 64 + lane) * 8;
            qh[b] = *reinterpret_cast<const half8*>(qp);
            ql[b] = *reinterpret_cast<const half8*>(qp + kv_plane);
        } else {
            const half_t* qp = q16 + (row0 + qb * 16 + fr) * (2 * hidden) + head * 64 + fq * 8;
            qh[b] = *reinterpret_cast<const half8*>(qp);
            ql[b] = *reinterpret_cast<const half8*>(qp + 32);
        }
    }
    f32x4 c0[QB], c1[QB];
    float mrun[QB], lsum[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        c0[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        c1[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mrun[b] = -INFINITY;
        lsum[b] = 0.f;
    }
    const float cs = (float)(0.17677669529663687 * 1.4426950408889634);    // 32^-0.5 * log2(e)
    if (!DIRECT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (!has_rows) return;
    // v_med3_f32(a, b, +inf) = max(a, b): fmaxf on MFMA results costs a canonicalising v_max x, x per input (15 instructions for 8
    // scores where 7 do); the +inf sits in a scalar register the compiler cannot fold
    float ce_inf;
    asm volatile("s_mov_b32 %0, 0x7f800000" : "=s"(ce_inf));
#define CE_MAX2(a_, b_) __builtin_amdgcn_fmed3f(a_, b_, ce_inf)
    // one 32-key block; EDGE (the pair's last block only: the loop is peeled) masks the keys past the pair's length
    auto key_block = [&](const int kb, auto edge_c) {
        constexpr bool EDGE = decltype(edge_c)::value;
        const int fo = kb * 2048 + lane * 16;
        const half8 k0h = *reinterpret_cast<const half8*>(k_hi + fo), k1h = *reinterpret_cast<const half8*>(k_hi + fo + 1024);
        const half8 k0l = *reinterpret_cast<const half8*>(k_lo + fo), k1l = *reinterpret_cast<const half8*>(k_lo + fo + 1024);
        // V fragment = the lane's 4 key slots of tile 2kb | of tile 2kb+1 (tile = [d half][lane][4 slots], 512 B per half); tile 2kb+1
        // of an odd-count pair is zeros (LDS) or another pair's finite rows (DIRECT) under P = 0
        const int vo = kb * 2048 + lane * 8;
        const half4 a0h = *reinterpret_cast<const half4*>(v_hi + vo), b0h = *reinterpret_cast<const half4*>(v_hi + vo + 1024);
        const half4 a1h = *reinterpret_cast<const half4*>(v_hi + vo + 512), b1h = *reinterpret_cast<const half4*>(v_hi + vo + 1536);
        const half4 a0l = *reinterpret_cast<const half4*>(v_lo + vo), b0l = *reinterpret_cast<const half4*>(v_lo + vo + 1024);
        const half4 a1l = *reinterpret_cast<const half4*>(v_lo + vo + 512), b1l = *reinterpret_cast<const half4*>(v_lo + vo + 1536);
        const half8 v0h = __builtin_shufflevector(a0h, b0h, 0, 1, 2, 3, 4, 5, 6, 7), v1h = __builtin_shufflevector(a1h, b1h, 0, 1, 2, 3, 4, 5, 6, 7);
        const half8 v0l = __builtin_shufflevector(a0l, b0l, 0, 1, 2, 3, 4, 5, 6, 7), v1l = __builtin_shufflevector(a1l, b1l, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
            z0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0l, qh[b], z0, 0, 0, 0);
            z1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1l, qh[b], z1, 0, 0, 0);
            z0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, ql[b], z0, 0, 0, 0);
            z1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1h, ql[b], z1, 0, 0, 0);
            z0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, qh[b], z0, 0, 0, 0);
            z1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1h, qh[b], z1, 0, 0, 0);
            // lane (fr, fq): z0[r] = S[query fr][key kb*32 + fq*4 + r], z1[r] = same + 16
            if (EDGE) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kb * 32 + fq * 4 + r >= len) z0[r] = -INFINITY;
                    if (kb * 32 + 16 + fq * 4 + r >= len) z1[r] = -INFINITY;
                }
            }
            float mx = CE_MAX2(CE_MAX2(CE_MAX2(z0[0], z0[1]), CE_MAX2(z0[2], z0[3])), CE_MAX2(CE_MAX2(z1[0], z1[1]), CE_MAX2(z1[2], z1[3])));
            // lane ^ 16 and lane ^ 32 by v_permlane16_swap / v_permlane32_swap of (mx, copy of mx): after the swap the two registers hold
            // the lane's own value and its partner's (tools/permlane_probe.hip) - no LDS round trip on the softmax's critical path. The
            // instructions are issued by hand: through the builtins this compiler folds the swap's second result into its first when both
            // feed one expression, and the maximum silently becomes "the value of lane group 0".
            {
                float cp = mx;
                asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(mx), "+v"(cp));
                mx = CE_MAX2(mx, cp);
                cp = mx;
                asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(mx), "+v"(cp));
                mx = CE_MAX2(mx, cp);
            }
            const float mnew = CE_MAX2(mrun[b], mx);                // finite: key 0 is always real
            const float alpha = __builtin_amdgcn_exp2f((mrun[b] - mnew) * cs);
            mrun[b] = mnew;
            const float off = -mnew * cs;
            float e[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[r] = __builtin_amdgcn_exp2f(fmaf(z0[r], cs, off));
                e[4 + r] = __builtin_amdgcn_exp2f(fmaf(z1[r], cs, off));
            }
            lsum[b] = lsum[b] * alpha + (((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7])));
#pragma unroll
            for (int r = 0; r < 4; ++r) { c0[b][r] *= alpha; c1[b][r] *= alpha; }
            float eh[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) eh[r] = ce_trunc10(e[r]);
            const u32x4 ph_u = {ce_pk(eh[0], eh[1]), ce_pk(eh[2], eh[3]), ce_pk(eh[4], eh[5]), ce_pk(eh[6], eh[7])};
            const u32x4 pl_u = {ce_pk(e[0] - eh[0], e[1] - eh[1]), ce_pk(e[2] - eh[2], e[3] - eh[3]),
                                ce_pk(e[4] - eh[4], e[5] - eh[5]), ce_pk(e[6] - eh[6], e[7] - eh[7])};
            const half8 ph = __builtin_bit_cast(half8, ph_u), pl = __builtin_bit_cast(half8, pl_u);
            // ctx^T[d][q] += V^T[d][key] P[q][key]: A = V fragment (row d), B = P (col q = fr)
            c0[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0l, ph, c0[b], 0, 0, 0);
            c1[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1l, ph, c1[b], 0, 0, 0);
            c0[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0h, pl, c0[b], 0, 0, 0);
            c1[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1h, pl, c1[b], 0, 0, 0);
            c0[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0h, ph, c0[b], 0, 0, 0);
            c1[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1h, ph, c1[b], 0, 0, 0);
        }
    };
    for (int kb = 0; kb + 1 < nkb; ++kb) key_block(kb, std::false_type{});
    key_block(nkb - 1, std::true_type{});
#undef CE_MAX2
    // lane (fr, fq): c0[r] = ctx[query fr][d = fq*4 + r], c1[r] = d + 16; the row sum is spread over the 4 fq lanes
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        float l = lsum[b];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        if ((qb0 + b) * 16 >= Lp || qb0 + b >= max_qblocks) break;
        if (MX) {
            // K-step = head; c0 = dims 4fq..4fq+3 (MFMA j = 0), c1 = 16 + the same (j = 1); lane half h = fq & 1, position 4 (fq >> 1) + r
            const int64_t mrow = (int64_t)row0 + (qb0 + b) * 16 + fr;
            char* img = reinterpret_cast<char*>(ctx16) + mx_img_base(mrow, head * 32, hidden >> 5) + (int)(mrow & 127) * 16 + (fq >> 1) * 8;
            mx_u2 h0, h1;
            unsigned l0, l1;
            mx_split4(c0[b][0] * inv, c0[b][1] * inv, c0[b][2] * inv, c0[b][3] * inv, h0, l0);
            mx_split4(c1[b][0] * inv, c1[b][1] * inv, c1[b][2] * inv, c1[b][3] * inv, h1, l1);
            *reinterpret_cast<mx_u2*>(img + (fq & 1) * MX_B_PLANE) = h0;
            *reinterpret_cast<mx_u2*>(img + (2 + (fq & 1)) * MX_B_PLANE) = h1;
            *reinterpret_cast<mx_u2*>(img + (4 + (fq & 1)) * MX_B_PLANE) = (mx_u2){l0, l1};
            continue;
        }
        half_t* o = ctx16 + (row0 + (qb0 + b) * 16 + fr) * (2 * hidden) + head * 64 + fq * 4;
        store_split4(o, 32, c0[b][0] * inv, c0[b][1] * inv, c0[b][2] * inv, c0[b][3] * inv);
        store_split4(o + 16, 32, c1[b][0] * inv, c1[b][1] * inv, c1[b][2] * inv, c1[b][3] * inv);
    }
}

template <bool MX>
__global__ __launch_bounds__(256) void ce_pool_classify_kernel(const half_t* __restrict__ x16, const float* __restrict__ wp,
                                                                const float* __restrict__ bp, const float* __restrict__ wc,
                                                                const float* __restrict__ bc, const int32_t* __restrict__ pair_off,
                                                                int hidden, float* __restrict__ logits) {
    __shared__ float xs[1024];
    __shared__ float part[4];
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const half_t* x = MX ? x16 : x16 + (size_t)pair_off[pair] * 2 * hidden;     // [CLS] = the pair's first packed row (split layout)
    // pair_off == nullptr (MX only): x16 holds one row per pair (the compact [CLS] stream of the last layer)
    for (int i = tid; i < hidden; i += 256)
        xs[i] = MX ? mx_load_elem(reinterpret_cast<const char*>(x16), pair_off ? pair_off[pair] : pair, i, hidden >> 5)
                   : (float)x[SPLIT_IDX(i)] + (float)x[SPLIT_IDX(i) + 32];
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

// The same head for the compact [CLS] stream of the MX forward (one row per pair), 8 pairs per workgroup: the pooler matrix is read
// once per 8 pairs and TRANSPOSED (wpT[k][n]: consecutive threads read consecutive floats) - the per-pair kernel above walks 590 KB of
// weights per pair with one row per thread (0.64 ms per 7,680 pairs). fp32 sums in the same k order as the kernel above.
#define POOL_PB 8
__global__ __launch_bounds__(256) void mx_pool_classify_kernel(const char* __restrict__ xc8, const float* __restrict__ wpT,
                                                                const float* __restrict__ bp, const float* __restrict__ wc,
                                                                const float* __restrict__ bc, int P, int hidden, float* __restrict__ logits) {
    __shared__ float xs[POOL_PB][1024];
    __shared__ float part[POOL_PB][4];
    const int p0 = blockIdx.x * POOL_PB, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < POOL_PB * hidden; i += 256) {
        const int j = i / hidden, c = i % hidden;
        xs[j][c] = p0 + j < P ? mx_load_elem(xc8, p0 + j, c, hidden >> 5) : 0.f;
    }
    __syncthreads();
    float acc[POOL_PB];
#pragma unroll
    for (int j = 0; j < POOL_PB; ++j) acc[j] = 0.f;
    for (int n = tid; n < hidden; n += 256) {
        float sj[POOL_PB];
        const float b = bp[n];
#pragma unroll
        for (int j = 0; j < POOL_PB; ++j) sj[j] = b;
        for (int k = 0; k < hidden; ++k) {
            const float w = wpT[(size_t)k * hidden + n];
#pragma unroll
            for (int j = 0; j < POOL_PB; ++j) sj[j] += w * xs[j][k];
        }
        const float c = wc[n];
#pragma unroll
        for (int j = 0; j < POOL_PB; ++j) acc[j] += c * tanhf(sj[j]);
    }
#pragma unroll
    for (int j = 0; j < POOL_PB; ++j) {
        float a = acc[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) part[j][wv] = a;
    }
    __syncthreads();
    if (tid < POOL_PB && p0 + tid < P) logits[p0 + tid] = part[tid][0] + part[tid][1] + part[tid][2] + part[tid][3] + bc[0];
}

// Sentence embedding head (sentence-transformers' Pooling(mean) + Normalize): mean of the last hidden state over the pair's
// real tokens, optionally L2-normalised. One workgroup per sequence; float32 sums over the split-fp16 stream (hi + lo).
template <bool MX>
__global__ __launch_bounds__(256) void ce_meanpool_kernel(const half_t* __restrict__ x16, const int32_t* __restrict__ pair_off,
                                                           const int32_t* __restrict__ lens, int L, int hidden, int normalize,
                                                           float* __restrict__ out) {
    __shared__ float part[4];
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int len = max(1, min(lens[pair], L));
    const half_t* x = x16 + (size_t)pair_off[pair] * 2 * hidden;
    float sq = 0.f;
    float v[4] = {0.f, 0.f, 0.f, 0.f};                               // hidden <= 1024: up to 4 features per thread
    for (int e = 0, c = tid; c < hidden; c += 256, ++e) {
        float s = 0.f;
        for (int t = 0; t < len; ++t) {
            if (MX) { s += mx_load_elem(reinterpret_cast<const char*>(x16), (int64_t)pair_off[pair] + t, c, hidden >> 5); continue; }
            const half_t* r = x + (size_t)t * 2 * hidden + SPLIT_IDX(c);
            s += (float)r[0] + (float)r[32];
        }
        v[e] = s / (float)len;
        sq += v[e] * v[e];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) part[wv] = sq;
    __syncthreads();
    const float nrm = sqrtf(part[0] + part[1] + part[2] + part[3]);
    const float sc = normalize ? 1.0f / fmaxf(nrm, 1e-12f) : 1.0f;  // torch.nn.functional.normalize: x / max(||x||, eps)
    for (int e = 0, c = tid; c < hidden; c += 256, ++e) out[(size_t)pair * hidden + c] = v[e] * sc;
}

__global__ void ce_f32_split_kernel(const float* __restrict__ in, half_t* __restrict__ out, int64_t n, int cols) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int64_t row = i / cols;
        const int c = (int)(i % cols);
        const half_t hi = (half_t)in[i];
        half_t* o = out + row * 2 * cols + SPLIT_IDX(c);   // split-row layout: [hi 32 | lo 32] per 32-element K group
        o[0] = hi;
        o[32] = (half_t)(in[i] - (float)hi);
    }
}

// ------------------------------------------------------------------------------------------------
static void ce_free_ws(rag_ce_model* m) {
    hipFree(m->y32); hipFree(m->x16); hipFree(m->q16); hipFree(m->kf16); hipFree(m->vf16); hipFree(m->ctx16);
    hipFree(m->h16); hipFree(m->ids); hipFree(m->tt); hipFree(m->lens); hipFree(m->logits); hipFree(m->pair_off); hipFree(m->row_pair); hipFree(m->m_packed); hipFree(m->sid); hipFree(m->stt);
    m->y32 = nullptr; m->x16 = m->q16 = m->kf16 = m->vf16 = m->ctx16 = m->h16 = nullptr;
    m->h16_rows = 0;
    m->y32_rows = 0;
    m->ids = m->tt = m->lens = nullptr; m->logits = nullptr;
    m->pair_off = m->row_pair = m->m_packed = nullptr;
    m->sid = m->stt = nullptr;
    m->ws_tokens = 0; m->ws_pairs = 0; m->ws_L = 0;
}

static void mx_free_ws(rag_ce_model* m) {
    auto& w = m->mx;
    hipFree(w.x8); hipFree(w.ctx8); hipFree(w.h8); hipFree(w.qf16); hipFree(w.kf16); hipFree(w.vf16);
    hipFree(w.xc8); hipFree(w.cc8); hipFree(w.hc8); hipFree(w.m_cls);
    hipFree(w.ids); hipFree(w.tt); hipFree(w.lens); hipFree(w.pair_off); hipFree(w.row_pair); hipFree(w.m_packed);
    hipFree(w.sid); hipFree(w.stt); hipFree(w.logits);
    w = rag_ce_model::MxWs();
}

static void ce_free_model(rag_ce_model** slot) {
    if (!*slot) return;
    for (void* p : (*slot)->allocs) hipFree(p);
    ce_free_ws(*slot);
    mx_free_ws(*slot);
    delete *slot;
    *slot = nullptr;
}

void ce_free(rag_ctx* h) {
    ce_free_model(&h->ce);
    ce_free_model(&h->emb);
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
    hipLaunchKernelGGL(ce_f32_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, tmp, *dst, (int64_t)total, (int)cols);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    hipFree(tmp);
    return RAG_OK;
}

// rows of several fp32 host matrices (same `cols`) concatenated -> one hi16 + lo8 image tensor (ce_mx.h)
static int up_mx_concat(rag_ctx* h, rag_ce_model* m, std::vector<const float*> srcs, size_t rows_each, size_t cols, char** dst) {
    const size_t n_each = rows_each * cols, total = n_each * srcs.size();
    float* tmp = nullptr;
    HIP_TRY(h, hipMalloc(&tmp, total * sizeof(float)));
    HIP_TRY(h, hipMalloc(dst, 3 * total));
    m->allocs.push_back(*dst);
    for (size_t i = 0; i < srcs.size(); ++i)
        HIP_TRY(h, hipMemcpyAsync(tmp + i * n_each, srcs[i], n_each * sizeof(float), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(mx_pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, tmp, (int)(rows_each * srcs.size()),
                       (int)cols, *dst);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    hipFree(tmp);
    return RAG_OK;
}

// Tensor order (HF state-dict names), see optimized-rag_amd/cross_encoder.py::flatten_state_dict:
//  0 word, 1 position, 2 token_type, 3 emb LN weight, 4 emb LN bias,
//  per layer (16): q.w q.b k.w k.b v.w v.b attn.out.w attn.out.b attn.LN.w attn.LN.b inter.w inter.b out.w out.b out.LN.w out.LN.b
//  then pooler.w pooler.b classifier.w classifier.b
//  (an embedding model - BertModel behind a mean-pooling head - ends after the layers: no pooler / classifier tensors)
static int ce_load_model(rag_ctx* h, const rag_ce_config* cfg, const float* const* T, int n, bool embed, int normalize, rag_ce_model** slot) {
    ARG_CHECK(h, cfg && T, "ce_load: null");
    ARG_CHECK(h, cfg->hidden % 128 == 0 && cfg->hidden <= 1024 && cfg->ffn % 128 == 0, "ce_load: hidden/ffn must be multiples of 128");
    ARG_CHECK(h, cfg->heads > 0 && cfg->hidden / cfg->heads == 32, "ce_load: head dim must be 32");
    ARG_CHECK(h, n == 5 + 16 * cfg->layers + (embed ? 0 : 4), "ce_load: wrong tensor count");
    ce_free_model(slot);
    rag_ce_model* m = new rag_ce_model();
    *slot = m;
    m->cfg = *cfg;
    m->embed = embed;
    m->normalize = normalize;
    m->out_width = embed ? cfg->hidden : 1;
    const size_t H = cfg->hidden, F = cfg->ffn;
    int rc;
    if ((rc = up_f32(h, m, T[0], (size_t)cfg->vocab_size * H, &m->word))) return rc;
    if ((rc = up_f32(h, m, T[1], (size_t)cfg->max_pos * H, &m->pos))) return rc;
    if ((rc = up_f32(h, m, T[2], (size_t)cfg->type_vocab * H, &m->type))) return rc;
    if ((rc = up_f32(h, m, T[3], H, &m->emb_ln_g))) return rc;
    if ((rc = up_f32(h, m, T[4], H, &m->emb_ln_b))) return rc;
    m->layers.resize(cfg->layers);
    m->mx_ok = H == MX_TM && F % MX_TM == 0 && F <= 1536;    // one feature tile = the hidden state (LayerNorm in the epilogue); the FFN bias is staged in 6 KiB of LDS
    for (int l = 0; l < cfg->layers; ++l) {
        const float* const* t = T + 5 + 16 * l;
        auto& ly = m->layers[l];
        if (m->mx_ok) {
            if ((rc = up_mx_concat(h, m, {t[0], t[2], t[4]}, H, H, &ly.wqkv8))) return rc;
            if ((rc = up_mx_concat(h, m, {t[6]}, H, H, &ly.wo8))) return rc;
            if ((rc = up_mx_concat(h, m, {t[10]}, F, H, &ly.w18))) return rc;
            if ((rc = up_mx_concat(h, m, {t[12]}, H, F, &ly.w28))) return rc;
        }
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
    if (!embed) {
        const float* const* t = T + 5 + 16 * cfg->layers;
        if ((rc = up_f32(h, m, t[0], H * H, &m->wp))) return rc;
        {
            std::vector<float> tr(H * H);
            for (size_t n = 0; n < H; ++n)
                for (size_t k = 0; k < H; ++k) tr[k * H + n] = t[0][n * H + k];
            if ((rc = up_f32(h, m, tr.data(), H * H, &m->wpT))) return rc;
            HIP_TRY(h, hipStreamSynchronize(h->stream));          // tr is a stack-lifetime buffer
        }
        if ((rc = up_f32(h, m, t[1], H, &m->bp))) return rc;
        if ((rc = up_f32(h, m, t[2], H, &m->wc))) return rc;
        if ((rc = up_f32(h, m, t[3], 1, &m->bc))) return rc;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return RAG_OK;
}

int ce_load_host(rag_ctx* h, const rag_ce_config* cfg, const float* const* T, int n) {
    return ce_load_model(h, cfg, T, n, false, 0, &h->ce);
}

int embed_load_host(rag_ctx* h, const rag_ce_config* cfg, const float* const* T, int n, int normalize) {
    return ce_load_model(h, cfg, T, n, true, normalize, &h->emb);
}

static const int kAttnL[] = {32, 64, 96, 128, 192, 256, 384, 512};

struct ce_planes { size_t x, q, kv, ctx, h; };
static ce_planes planes_for(const rag_ce_model* m, int64_t Mp) {
    const size_t H = m->cfg.hidden, F = m->cfg.ffn;
    return {(size_t)Mp * H, (size_t)Mp * H, (size_t)Mp * H + 2048, (size_t)Mp * H, (size_t)Mp * F};
}

template <int QB>
static int launch_attention(rag_ctx* h, rag_ce_model* m, int P, int L, const ce_planes& pp, hipStream_t st, const int32_t* lens_dev) {
    const int lds = L * 256;                                           // K hi | K lo | V hi | V lo fragment planes
    int& attr_lds = h->attr_ce_attn_lds[QB];
    if (lds > attr_lds) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(ce_attention_kernel<QB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_lds = lds;
    }
    const int waves = L / (16 * QB);
    hipLaunchKernelGGL((ce_attention_kernel<QB>), dim3(m->cfg.heads, P), dim3(64 * waves), lds, st, m->q16, m->kf16, m->vf16,
                       pp.kv, lens_dev, m->pair_off, L, m->cfg.hidden, m->cfg.heads,
                       (int)round_up((int64_t)m->ws_pairs * L, CE_BN), m->ctx16);
    return RAG_OK;
}

template <int PER>
static void launch_ln(rag_ce_model* m, const float* y, const float* g, const float* b, int64_t M, hipStream_t st) {
    hipLaunchKernelGGL(ce_layernorm_kernel<PER>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, y, g, b, m->m_packed, m->cfg.hidden,
                       (float)m->cfg.ln_eps, m->x16);
}

// lens_dev / logits_dev: this chunk's lengths and logit slots (the model's staging buffers for host-pointer calls, the caller's
// own device arrays otherwise)
static int ce_forward_chunk(rag_ctx* h, rag_ce_model* m, int P, int L, hipStream_t st, const int32_t* lens_dev, float* logits_dev) {
    const int H = m->cfg.hidden, F = m->cfg.ffn;
    const int64_t M = (int64_t)P * L;
    const int64_t Mp = round_up((int64_t)m->ws_pairs * L, CE_BN);      // plane strides follow the ALLOCATED size
    const ce_planes pp = planes_for(m, Mp);
    const int per = H / 64;
    const float eps = (float)m->cfg.ln_eps;
    bool& attr = h->attr_ce_gemm;
    const size_t lds = CE_GEMM_LDS;
    if (!attr) {
#define CE_ATTR(E, T) HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(ce_gemm_kernel<E, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#define CE_ATTR4(E) CE_ATTR(E, 3)
        CE_ATTR4(EPI_QKV) CE_ATTR4(EPI_GELU) CE_ATTR4(EPI_RESID)
        attr = true;
    }
    // Correction terms per GEMM site (qkv, out-proj, ffn-up, ffn-down): the split-fp16 kernels run the full form (both terms
    // everywhere: profiles/r02_b_ce_term_ablation.md shows that dropping any one of them spends the whole logit-error budget;
    // the per-site ablation build that produced that table lives in the round-3 history, commit e52b089, not in the product).
    int terms[4] = {3, 3, 3, 3};
#define CE_GEMM(E, T, ...) hipLaunchKernelGGL((ce_gemm_kernel<E, 3>), dim3(n_cu), blk, lds, st, __VA_ARGS__);
    (void)terms;
 // bias + residual + LayerNorm in the GEMM epilogue when the geometry allows (hidden = 384, K a multiple of 192: the
    // MiniLM-L-6 shape); RAG_CE_NO_FUSED_LN=1 forces the stand-alone path (parity test of both)
    // ... except for SMALL batches: the fused kernel's tiles are 128 tokens x all 384 features, so one query's 100 pairs (~140
    // tiles) leave 45 % of the CUs without a tile, while the plain GEMM (128 features x 256 tokens: three times as many tiles) + a
    // LayerNorm launch fills them (tools/ln_sweep.py, forward ms fused | unfused: 13 pairs 0.98 | 0.78, 25 1.09 | 0.92, 50 1.33 |
    // 1.21, 100 1.90 | 1.77, 150 2.40 | 2.64, 400 5.97 | 6.09, 800 11.2 | 11.8). ce_no_fused_ln: 1 = never fused, -1 = always.
    const bool fused_ln = H == 384 && F % 192 == 0 && h->opt.ce_no_fused_ln <= 0 &&
                          (h->opt.ce_no_fused_ln < 0 || h->opt.ce_no_fused_ffn < 0 || (int64_t)P * L > LN_UNFUSED_MAX_ROWS);     // (a forced fused FFN contains a fused LayerNorm)
    const int64_t y_rows = round_up((int64_t)P * L, CE_BN);
    if (!fused_ln && y_rows > m->y32_rows) {                 // sized by this call (a single query after a 2M-token batch must not take 3 GB)
        HIP_TRY(h, hipStreamSynchronize(st));
        hipFree(m->y32);
        m->y32 = nullptr;
        m->y32_rows = 0;
        HIP_TRY(h, hipMalloc(&m->y32, (size_t)y_rows * H * 4));
        m->y32_rows = y_rows;
    }
    if (fused_ln && !h->attr_ce_gemm_ln) {
#define CE_ATTR_LN(T) HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(ce_gemm_ln_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, LNG_LDS));
        CE_ATTR_LN(3)
        h->attr_ce_gemm_ln = true;
    }
#define CE_GEMM_LN(T, ...) hipLaunchKernelGGL((ce_gemm_ln_kernel<3>), dim3(n_cu), blk, LNG_LDS, st, __VA_ARGS__);
    // the whole FFN in one kernel (ce_ffn_ln_kernel) when the geometry allows; option ce_no_fused_ffn keeps the two-launch form
    // ... from FFN_FUSED_MIN_ROWS padded rows on (tools/ffn_sweep.py, forward ms fused | two-launch: 100 pairs 2.03 | 1.92, 400
    // 6.51 | 5.99, 1600 23.2 | 22.7, 3200 44.8 | 44.6, 6400 88.2 | 88.9): a small batch (one query's 100 pairs = ~140 tiles of 128
    // tokens for 256 CUs) finishes sooner as two launches whose tiles are finer. ce_no_fused_ffn: 1 = never, -1 = always.
    const bool fused_ffn = fused_ln && F % FFN_CH == 0 && h->opt.ce_no_fused_ffn <= 0 &&
                           (h->opt.ce_no_fused_ffn < 0 || (int64_t)P * L >= FFN_FUSED_MIN_ROWS);
    // the FFN intermediate [tokens][ffn] (the largest activation: 12 GB per 2M-token chunk) exists only for the two-launch form
    // (sized by what this call needs, not by the workspace: after a large fused batch a single query must not allocate 12 GB)
    const int64_t h_rows = round_up((int64_t)P * L, CE_BN);
    if (!fused_ffn && h_rows > m->h16_rows) {
        HIP_TRY(h, hipStreamSynchronize(st));
        hipFree(m->h16);
        m->h16 = nullptr;
        m->h16_rows = 0;
        HIP_TRY(h, hipMalloc(&m->h16, (size_t)h_rows * F * 4));
        HIP_TRY(h, hipMemsetAsync(m->h16, 0, (size_t)h_rows * F * 4, st));   // padded token rows are read by the GEMM tiles: keep them finite
        m->h16_rows = h_rows;
    }
    if (fused_ffn && !h->attr_ce_ffn) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(ce_ffn_ln_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS));
        h->attr_ce_ffn = true;
    }
#define CE_PER_DISPATCH(CALL)                                                                 \
    switch (per) {                                                                            \
        case 2: CALL(2); break; case 4: CALL(4); break; case 6: CALL(6); break;               \
        case 8: CALL(8); break; case 12: CALL(12); break; case 16: CALL(16); break;           \
        default: h->err = "ce: unsupported hidden size"; return RAG_ERR_ARG;                  \
    }
#define EMB(PER) hipLaunchKernelGGL(ce_embed_ln_kernel<PER>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, m->ids, m->tt, m->word, \
                                    m->pos, m->type, m->emb_ln_g, m->emb_ln_b, m->m_packed, m->row_pair, m->pair_off, L, H,            \
                                    m->cfg.vocab_size, eps, m->x16)
    // packed row layout of this chunk (no host round trip: grids cover the padded worst case, kernels stop at m_packed)
    hipLaunchKernelGGL(ce_pack_scan_kernel, dim3(1), dim3(1024), 0, st, lens_dev, P, L, m->pair_off, m->m_packed);
    hipLaunchKernelGGL(ce_pack_rows_kernel, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, st, m->pair_off, P, L, Mp, m->row_pair);
    CE_PER_DISPATCH(EMB)
    const dim3 blk(512);
    static const unsigned n_cu = [] { int d = 0, n = 0; hipGetDevice(&d); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return (unsigned)(n >= 8 ? n / 8 * 8 : 256); }();
    const half_t* nullh = nullptr;
    for (int l = 0; l < m->cfg.layers; ++l) {
        auto& ly = m->layers[l];
        CE_GEMM(EPI_QKV, terms[0], ly.wqkv, m->x16,
                3 * H, H, ly.bqkv, (const half_t*)nullptr, (float*)nullptr, m->q16, m->kf16, m->vf16, pp.kv, H,
                m->cfg.heads, m->m_packed, (int)Mp)
        {
            const int rc = L == 32 ? launch_attention<1>(h, m, P, L, pp, st, lens_dev) : launch_attention<2>(h, m, P, L, pp, st, lens_dev);
            if (rc != RAG_OK) return rc;
        }
#define LN1(PER) launch_ln<PER>(m, m->y32, ly.ln1_g, ly.ln1_b, M, st)
        if (fused_ln) {
            CE_GEMM_LN(terms[1], ly.wo, m->ctx16, H, ly.bo, ly.ln1_g, ly.ln1_b, eps, m->x16, m->m_packed)
        } else {
            CE_GEMM(EPI_RESID, terms[1], ly.wo, m->ctx16, H, H,
                    ly.bo, (const half_t*)m->x16, m->y32, (half_t*)nullptr, (half_t*)nullptr, (half_t*)nullptr, (size_t)0, H, m->cfg.heads, m->m_packed, (int)Mp)
            CE_PER_DISPATCH(LN1)
        }
        if (fused_ffn) {
            hipLaunchKernelGGL((ce_ffn_ln_kernel<3>), dim3(n_cu), blk, FFN_LDS, st, (const half_t*)ly.w1, (const float*)ly.b1, (const half_t*)ly.w2,
                               (const float*)ly.b2, F, (const float*)ly.ln2_g, (const float*)ly.ln2_b, eps, m->x16, (const int32_t*)m->m_packed);
            continue;
        }
        CE_GEMM(EPI_GELU, terms[2], ly.w1, m->x16, F, H,
                ly.b1, (const half_t*)nullptr, (float*)nullptr, m->h16, (half_t*)nullptr, (half_t*)nullptr, (size_t)0,
                H, m->cfg.heads, m->m_packed, (int)Mp)
#define LN2(PER) launch_ln<PER>(m, m->y32, ly.ln2_g, ly.ln2_b, M, st)
        if (fused_ln) {
            CE_GEMM_LN(terms[3], ly.w2, m->h16, F, ly.b2, ly.ln2_g, ly.ln2_b, eps, m->x16, m->m_packed)
        } else {
            CE_GEMM(EPI_RESID, terms[3], ly.w2, m->h16, H, F,
                    ly.b2, (const half_t*)m->x16, m->y32, (half_t*)nullptr, (half_t*)nullptr, (half_t*)nullptr, (size_t)0, H, m->cfg.heads, m->m_packed, (int)Mp)
            CE_PER_DISPATCH(LN2)
        }
    }
    (void)nullh;
    if (m->embed)
        hipLaunchKernelGGL(ce_meanpool_kernel<false>, dim3(P), dim3(256), 0, st, (const half_t*)m->x16, (const int32_t*)m->pair_off, lens_dev, L, H,
                           m->normalize, logits_dev);
    else
        hipLaunchKernelGGL(ce_pool_classify_kernel<false>, dim3(P), dim3(256), 0, st, m->x16, m->wp, m->bp, m->wc, m->bc, m->pair_off, H, logits_dev);
    HIP_TRY(h, hipGetLastError());
    return RAG_OK;
}

// (re)allocates the activation workspace; the zero fills are enqueued on `st`, the stream the forward runs on (a
// synchronous hipMemset on the null stream is NOT ordered against a non-blocking stream: the first forward after a
// reallocation could otherwise start before its buffers were cleared)
static int ce_ensure_ws(rag_ctx* h, rag_ce_model* m, int P, int L, hipStream_t st) {
    if (P <= m->ws_pairs && L == m->ws_L) return RAG_OK;
    ce_free_ws(m);
    const int64_t Mp = round_up((int64_t)P * L, CE_BN);
    m->ws_pairs = P;                                     // planes_for() uses the allocated pair count
    const ce_planes pp = planes_for(m, Mp);
    HIP_TRY(h, hipMalloc(&m->x16, 2 * pp.x * 2));
    HIP_TRY(h, hipMalloc(&m->q16, 2 * pp.q * 2));
    HIP_TRY(h, hipMalloc(&m->kf16, 2 * pp.kv * 2));
    HIP_TRY(h, hipMalloc(&m->vf16, 2 * pp.kv * 2));
    HIP_TRY(h, hipMalloc(&m->ctx16, 2 * pp.ctx * 2));
    HIP_TRY(h, hipMalloc(&m->ids, (size_t)Mp * 4));
    HIP_TRY(h, hipMalloc(&m->tt, (size_t)Mp * 4));
    HIP_TRY(h, hipMalloc(&m->lens, (size_t)P * 4));
    HIP_TRY(h, hipMalloc(&m->pair_off, (size_t)(P + 1) * 4));
    HIP_TRY(h, hipMalloc(&m->row_pair, (size_t)Mp * 4));
    HIP_TRY(h, hipMalloc(&m->m_packed, 4));
    HIP_TRY(h, hipMalloc(&m->sid, (size_t)P * L * 4));            // L_in <= L
    HIP_TRY(h, hipMalloc(&m->stt, (size_t)P * L * 4));
    HIP_TRY(h, hipMalloc(&m->logits, (size_t)P * m->out_width * 4));
    // padded token rows are read by the GEMM tiles: keep them finite
    HIP_TRY(h, hipMemsetAsync(m->x16, 0, 2 * pp.x * 2, st));
    HIP_TRY(h, hipMemsetAsync(m->ctx16, 0, 2 * pp.ctx * 2, st));
    HIP_TRY(h, hipMemsetAsync(m->q16, 0, 2 * pp.q * 2, st));
    HIP_TRY(h, hipMemsetAsync(m->kf16, 0, 2 * pp.kv * 2, st));
    HIP_TRY(h, hipMemsetAsync(m->vf16, 0, 2 * pp.kv * 2, st));
    m->ws_pairs = P;
    m->ws_L = L;
    m->ws_tokens = Mp;
    return RAG_OK;
}

// ---- the MX forward (ce_mx.h): every GEMM on 384-feature x 128-token tiles with hi16 + lo8 operands ---------------------
static int mx_ensure_ws(rag_ctx* h, rag_ce_model* m, int P, int L, hipStream_t st) {
    auto& w = m->mx;
    if (P <= w.pairs && L == w.L) return RAG_OK;
    HIP_TRY(h, hipStreamSynchronize(st));
    mx_free_ws(m);
    const size_t H = m->cfg.hidden, F = m->cfg.ffn;
    const int64_t Mp = round_up((int64_t)P * L, 256);
    const size_t kv = (size_t)Mp * H + 2048;                       // halfs per plane of qf16 / kf16 / vf16
    HIP_TRY(h, hipMalloc(&w.x8, (size_t)Mp * H * 3));
    HIP_TRY(h, hipMalloc(&w.ctx8, (size_t)Mp * H * 3));
    HIP_TRY(h, hipMalloc(&w.h8, (size_t)Mp * F * 3));
    const int64_t Pp = round_up((int64_t)P, MX_TN);
    HIP_TRY(h, hipMalloc(&w.xc8, (size_t)Pp * H * 3));
    HIP_TRY(h, hipMalloc(&w.cc8, (size_t)Pp * H * 3));
    HIP_TRY(h, hipMalloc(&w.hc8, (size_t)Pp * F * 3));
    HIP_TRY(h, hipMalloc(&w.m_cls, 4));
    HIP_TRY(h, hipMemsetAsync(w.xc8, 0, (size_t)Pp * H * 3, st));
    HIP_TRY(h, hipMemsetAsync(w.cc8, 0, (size_t)Pp * H * 3, st));
    HIP_TRY(h, hipMemsetAsync(w.hc8, 0, (size_t)Pp * F * 3, st));
    HIP_TRY(h, hipMalloc(&w.qf16, 2 * kv * 2));
    HIP_TRY(h, hipMalloc(&w.kf16, 2 * kv * 2));
    HIP_TRY(h, hipMalloc(&w.vf16, 2 * kv * 2));
    HIP_TRY(h, hipMalloc(&w.ids, (size_t)Mp * 4));
    HIP_TRY(h, hipMalloc(&w.tt, (size_t)Mp * 4));
    HIP_TRY(h, hipMalloc(&w.lens, (size_t)P * 4));
    HIP_TRY(h, hipMalloc(&w.pair_off, (size_t)(P + 1) * 4));
    HIP_TRY(h, hipMalloc(&w.row_pair, (size_t)Mp * 4));
    HIP_TRY(h, hipMalloc(&w.m_packed, 4));
    HIP_TRY(h, hipMalloc(&w.sid, (size_t)P * L * 4));
    HIP_TRY(h, hipMalloc(&w.stt, (size_t)P * L * 4));
    HIP_TRY(h, hipMalloc(&w.logits, (size_t)P * m->out_width * 4));
    // rows past a chunk's packed rows are read by the last token tile of every GEMM: keep them finite (zero is a valid image)
    HIP_TRY(h, hipMemsetAsync(w.x8, 0, (size_t)Mp * H * 3, st));
    HIP_TRY(h, hipMemsetAsync(w.ctx8, 0, (size_t)Mp * H * 3, st));
    HIP_TRY(h, hipMemsetAsync(w.h8, 0, (size_t)Mp * F * 3, st));
    HIP_TRY(h, hipMemsetAsync(w.qf16, 0, 2 * kv * 2, st));
    HIP_TRY(h, hipMemsetAsync(w.kf16, 0, 2 * kv * 2, st));
    HIP_TRY(h, hipMemsetAsync(w.vf16, 0, 2 * kv * 2, st));
    w.pairs = P;
    w.L = L;
    w.tokens = Mp;
    return RAG_OK;
}

template <int QB>
static int mx_launch_attention(rag_ctx* h, rag_ce_model* m, int P, int L, size_t kv_plane, hipStream_t st, const int32_t* lens_dev, int max_qblocks) {
    const int lds = L * 256;
    int& attr_lds = h->attr_ce_attn_mx_lds[QB];
    if (lds > attr_lds) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(ce_attention_kernel<QB, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_lds = lds;
    }
    auto& w = m->mx;
    if (max_qblocks == 1) {                                  // the [CLS]-only last layer: one wave per (head, pair), no LDS
        hipLaunchKernelGGL((ce_attention_kernel<1, true, true>), dim3(m->cfg.heads, P), dim3(64), 0, st, (const half_t*)w.qf16,
                           (const half_t*)w.kf16, (const half_t*)w.vf16, kv_plane, lens_dev, (const int32_t*)w.pair_off, L, m->cfg.hidden,
                           m->cfg.heads, (int)w.tokens, reinterpret_cast<half_t*>(w.ctx8), 1);
        return RAG_OK;
    }
    hipLaunchKernelGGL((ce_attention_kernel<QB, true>), dim3(m->cfg.heads, P), dim3(64 * (L / (16 * QB))), lds, st, (const half_t*)w.qf16,
                       (const half_t*)w.kf16, (const half_t*)w.vf16, kv_plane, lens_dev, (const int32_t*)w.pair_off, L, m->cfg.hidden,
                       m->cfg.heads, (int)w.tokens, reinterpret_cast<half_t*>(w.ctx8), max_qblocks);
    return RAG_OK;
}

static int mx_forward_chunk(rag_ctx* h, rag_ce_model* m, int P, int L, hipStream_t st, const int32_t* lens_dev, float* logits_dev) {
    auto& w = m->mx;
    const int H = m->cfg.hidden, F = m->cfg.ffn;
    const int64_t M = (int64_t)P * L, Mp = w.tokens;
    const float eps = (float)m->cfg.ln_eps;
    const size_t kv_plane = (size_t)Mp * H + 2048;
    if (!h->attr_ce_mx) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(mx_gemm_kernel<mx_epi_qkv>), hipFuncAttributeMaxDynamicSharedMemorySize, MX_KERNEL_LDS));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(mx_gemm_kernel<mx_epi_gelu>), hipFuncAttributeMaxDynamicSharedMemorySize, MX_KERNEL_LDS));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(mx_gemm_kernel<mx_epi_ln>), hipFuncAttributeMaxDynamicSharedMemorySize, MX_KERNEL_LDS));
        h->attr_ce_mx = true;
    }
    hipLaunchKernelGGL(ce_pack_scan_kernel, dim3(1), dim3(1024), 0, st, lens_dev, P, L, w.pair_off, w.m_packed);
    hipLaunchKernelGGL(ce_pack_rows_kernel, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, st, (const int32_t*)w.pair_off, P, L, Mp, w.row_pair);
    hipLaunchKernelGGL(mx_embed_ln_kernel, dim3((unsigned)((M + MX_EMB_ROWS - 1) / MX_EMB_ROWS)), dim3(256), 0, st, (const int32_t*)w.ids, (const int32_t*)w.tt,
                       (const float*)m->word, (const float*)m->pos, (const float*)m->type, (const float*)m->emb_ln_g, (const float*)m->emb_ln_b,
                       (const int32_t*)w.m_packed, (const int32_t*)w.row_pair, (const int32_t*)w.pair_off, L, m->cfg.vocab_size, eps, w.x8);
    static const unsigned n_cu = [] { int d = 0, n = 0; hipGetDevice(&d); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return (unsigned)(n >= 8 ? n / 8 * 8 : 256); }();
    const dim3 blk(512);
    for (int l = 0; l < m->cfg.layers; ++l) {
        auto& ly = m->layers[l];
        const bool cls_tail = !m->embed && l == m->cfg.layers - 1;
        if (cls_tail) {
            // last layer of a classifier (see below): K and V for every token, Q for the [CLS] rows alone
            hipLaunchKernelGGL(mx_gather_rows_kernel, dim3((unsigned)(((int64_t)P * 72 + 255) / 256)), dim3(256), 0, st, (const char*)w.x8, (const char*)nullptr,
                               (const int32_t*)w.pair_off, P, H / 32, w.xc8, (char*)nullptr, w.m_cls);
            hipLaunchKernelGGL((mx_gemm_kernel<mx_epi_qkv>), dim3(n_cu), blk, MX_KERNEL_LDS, st, (const char*)ly.wqkv8 + (size_t)(H / 32) * MX_A_STAGE,
                               (const char*)w.x8, H / 32, 2, (const int32_t*)w.m_packed,
                               mx_epi_qkv{w.qf16, w.kf16, w.vf16, kv_plane, ly.bqkv, (int)(Mp >> 4), 1, (const int32_t*)nullptr, 0});
            hipLaunchKernelGGL((mx_gemm_kernel<mx_epi_qkv>), dim3(n_cu), blk, MX_KERNEL_LDS, st, (const char*)ly.wqkv8, (const char*)w.xc8, H / 32, 1,
                               (const int32_t*)w.m_cls, mx_epi_qkv{w.qf16, w.kf16, w.vf16, kv_plane, ly.bqkv, (int)(Mp >> 4), 0, (const int32_t*)w.pair_off, P});
        } else
        hipLaunchKernelGGL((mx_gemm_kernel<mx_epi_qkv>), dim3(n_cu), blk, MX_KERNEL_LDS, st, (const char*)ly.wqkv8, (const char*)w.x8, H / 32, 3,
                           (const int32_t*)w.m_packed, mx_epi_qkv{w.qf16, w.kf16, w.vf16, kv_plane, ly.bqkv, (int)(Mp >> 4), 0, (const int32_t*)nullptr, 0});
        // The classifier reads the [CLS] row of the last layer alone (pooler: hidden_states[:, 0]), and nothing after the last layer's
        // attention mixes tokens. So in the LAST layer of a classifier only the first 16-query block of every pair goes through
        // attention, and out-projection, FFN and both LayerNorms run on ONE row per pair (gathered into compact tensors): the same
        // arithmetic on the rows that are read, nothing computed for the rows that are not - 4.6M rows become 25,600 for a third of
        // the layer's kernels. An embedding model (mean pooling over all tokens) takes the full path.
        {
            const int lim = cls_tail ? 1 : 1 << 20;
            // one 16-query block per wave up to L = 256 (16 waves per (head, pair)): same-box A/B against two blocks per wave: -2.6 % (four: +9 %)
            const int rc = L <= 256 ? mx_launch_attention<1>(h, m, P, L, kv_plane, st, lens_dev, lim) : mx_launch_attention<2>(h, m, P, L, kv_plane, st, lens_dev, lim);
            if (rc != RAG_OK) return rc;
        }
        if (cls_tail) {
            hipLaunchKernelGGL(mx_gather_rows_kernel, dim3((unsigned)(((int64_t)P * 72 + 255) / 256)), dim3(256), 0, st, (const char*)w.ctx8, (const char*)nullptr,
                               (const int32_t*)w.pair_off, P, H / 32, w.cc8, (char*)nullptr, w.m_cls);
            hipLaunchKernelGGL((mx_gemm_kernel<mx_epi_ln>), dim3(n_cu), blk, MX_KERNEL_LDS, st, (const char*)ly.wo8, (const char*)w.cc8, H / 32, 1,
                               (const int32_t*)w.m_cls, mx_epi_ln{w.xc8, ly.bo, ly.ln1_g, ly.ln1_b, eps});
            hipLaunchKernelGGL((mx_gemm_kernel<mx_epi_gelu>), dim3(n_cu), blk, MX_KERNEL_LDS, st, (const char*)ly.w18, (const char*)w.xc8, H / 32, F / MX_TM,
                               (const int32_t*)w.m_cls, mx_epi_gelu{w.hc8, ly.b1, F / 32});
            hipLaunchKernelGGL((mx_gemm_kernel<mx_epi_ln>), dim3(n_cu), blk, MX_KERNEL_LDS, st, (const char*)ly.w28, (const char*)w.hc8, F / 32, 1,
                               (const int32_t*)w.m_cls, mx_epi_ln{w.xc8, ly.b2, ly.ln2_g, ly.ln2_b, eps});
            // the batched pooler pays from ~512 pairs on; a single query's 100 pairs fill more CUs with one workgroup per pair
            if (P >= 512)
                hipLaunchKernelGGL(mx_pool_classify_kernel, dim3((unsigned)((P + POOL_PB - 1) / POOL_PB)), dim3(256), 0, st, (const char*)w.xc8,
                                   (const float*)m->wpT, (const float*)m->bp, (const float*)m->wc, (const float*)m->bc, P, H, logits_dev);
            else
                hipLaunchKernelGGL(ce_pool_classify_kernel<true>, dim3(P), dim3(256), 0, st, reinterpret_cast<const half_t*>(w.xc8), (const float*)m->wp,
                                   (const float*)m->bp, (const float*)m->wc, (const float*)m->bc, (const int32_t*)nullptr, H, logits_dev);
            HIP_TRY(h, hipGetLastError());
            return RAG_OK;
        }
        hipLaunchKernelGGL((mx_gemm_kernel<mx_epi_ln>), dim3(n_cu), blk, MX_KERNEL_LDS, st, (const char*)ly.wo8, (const char*)w.ctx8, H / 32, 1,
                           (const int32_t*)w.m_packed, mx_epi_ln{w.x8, ly.bo, ly.ln1_g, ly.ln1_b, eps});
        hipLaunchKernelGGL((mx_gemm_kernel<mx_epi_gelu>), dim3(n_cu), blk, MX_KERNEL_LDS, st, (const char*)ly.w18, (const char*)w.x8, H / 32, F / MX_TM,
                           (const int32_t*)w.m_packed, mx_epi_gelu{w.h8, ly.b1, F / 32});
        hipLaunchKernelGGL((mx_gemm_kernel<mx_epi_ln>), dim3(n_cu), blk, MX_KERNEL_LDS, st, (const char*)ly.w28, (const char*)w.h8, F / 32, 1,
                           (const int32_t*)w.m_packed, mx_epi_ln{w.x8, ly.b2, ly.ln2_g, ly.ln2_b, eps});
    }
    if (m->embed)
        hipLaunchKernelGGL(ce_meanpool_kernel<true>, dim3(P), dim3(256), 0, st, reinterpret_cast<const half_t*>(w.x8), (const int32_t*)w.pair_off, lens_dev,
                           L, H, m->normalize, logits_dev);
    else
        hipLaunchKernelGGL(ce_pool_classify_kernel<true>, dim3(P), dim3(256), 0, st, reinterpret_cast<const half_t*>(w.x8), (const float*)m->wp,
                           (const float*)m->bp, (const float*)m->wc, (const float*)m->bc, (const int32_t*)w.pair_off, H, logits_dev);
    HIP_TRY(h, hipGetLastError());
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

// out: [P][m->out_width] floats (logits of a cross-encoder, pooled vectors of an embedding model)
static int ce_run(rag_ctx* h, rag_ce_model* m, const int32_t* ids, const int32_t* tt, const int32_t* lens, int P, int L_in, float* out,
                  hipStream_t st, bool host_ptrs) {
    ARG_CHECK(h, ids && tt && lens && out && P > 0 && L_in > 0, "ce_score: bad arguments");
    const size_t ow = (size_t)m->out_width;
    ARG_CHECK(h, L_in <= m->cfg.max_pos && L_in <= 512, "ce_score: sequence longer than max_position_embeddings/512");
    int L = 0;
    for (int c : kAttnL) if (c >= L_in) { L = c; break; }
    // ~2M tokens of activations per chunk (~30 GB). Option ce_chunk_tokens (diagnostic) shrinks it so that parity tests can run
    // the multi-chunk loop on small inputs.
    const int64_t chunk_tokens = h->opt.ce_chunk_tokens >= 32 ? h->opt.ce_chunk_tokens : 2'000'000;
    // equal chunks: 25,600 pairs at L = 256 go as 4 x 6,400 and not 3 x 7,812 + 2,164 (a small last chunk leaves the persistent
    // GEMM workgroups of its one-feature-tile kernels 12-or-13 tiles each: 6 % of that chunk idle)
    const int chunk_max = std::max(1, std::min(P, (int)(chunk_tokens / L)));
    const int chunk = (P + (P + chunk_max - 1) / chunk_max - 1) / ((P + chunk_max - 1) / chunk_max);
    // Which forward: the MX kernels (hi16 + lo8 operands, 384 x 128 tiles; ce_mx.h) whenever the SHAPE allows (hidden 384, ffn a multiple
    // of 384), the split-fp16 kernels for every other model. Option ce_mx: -1 = never (1 = always, the same as the default today).
    const bool use_mx = m->mx_ok && L >= 32 && h->opt.ce_mx >= 0 && (h->opt.ce_mx > 0 || (int64_t)P * L >= MX_MIN_ROWS);
    int rc = use_mx ? mx_ensure_ws(h, m, chunk, L, st) : ce_ensure_ws(h, m, chunk, L, st);
    if (rc) return rc;
    const hipMemcpyKind kin = hipMemcpyHostToDevice, kout = hipMemcpyDeviceToHost;
    int32_t *sid = use_mx ? m->mx.sid : m->sid, *stt = use_mx ? m->mx.stt : m->stt;
    int32_t* const lens_stage = use_mx ? m->mx.lens : m->lens;
    float* const logits_stage = use_mx ? m->mx.logits : m->logits;
    int32_t *const ids_pad = use_mx ? m->mx.ids : m->ids, *const tt_pad = use_mx ? m->mx.tt : m->tt;
    if ((rc = prof_begin(h, 2, st))) return rc;
    for (int p0 = 0; p0 < P; p0 += chunk) {
        const int pc = std::min(chunk, P - p0);
        // host arrays are staged chunk by chunk; device arrays are read (ids, lens) and written (logits) where they are: four
        // small copies less per chunk, 45 us of a single-query call
        const int32_t *src_ids = ids + (size_t)p0 * L_in, *src_tt = tt + (size_t)p0 * L_in, *lens_dev = lens + p0;
        float* logits_dev = out + (size_t)p0 * ow;
        if (host_ptrs) {
            HIP_TRY(h, hipMemcpyAsync(sid, src_ids, (size_t)pc * L_in * 4, kin, st));
            HIP_TRY(h, hipMemcpyAsync(stt, src_tt, (size_t)pc * L_in * 4, kin, st));
            HIP_TRY(h, hipMemcpyAsync(lens_stage, lens_dev, (size_t)pc * 4, kin, st));
            src_ids = sid; src_tt = stt; lens_dev = lens_stage; logits_dev = logits_stage;
        }
        const int64_t n = (int64_t)pc * L;
        hipLaunchKernelGGL(ce_pad_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src_ids, src_tt, pc, L_in, L, ids_pad, tt_pad);
        rc = use_mx ? mx_forward_chunk(h, m, pc, L, st, lens_dev, logits_dev) : ce_forward_chunk(h, m, pc, L, st, lens_dev, logits_dev);
        if (rc) break;
        if (host_ptrs) HIP_TRY(h, hipMemcpyAsync(out + (size_t)p0 * ow, logits_stage, (size_t)pc * ow * 4, kout, st));
    }
    if (!rc) rc = prof_end(h, 2, st);
    // device-pointer calls stay asynchronous on the caller's stream (all buffers belong to the model workspace);
    // host-pointer calls return results, so they wait
    hipError_t e = host_ptrs ? hipStreamSynchronize(st) : hipGetLastError();
    if (rc) return rc;
    if (e != hipSuccess) {
        h->err = std::string("ce_score: ") + hipGetErrorString(e);
        return RAG_ERR_HIP;
    }
    return RAG_OK;
}

int ce_score(rag_ctx* h, const int32_t* ids, const int32_t* tt, const int32_t* lens, int P, int L_in, float* out,
             hipStream_t st, bool host_ptrs) {
    ARG_CHECK(h, h->ce != nullptr, "no cross-encoder loaded");
    return ce_run(h, h->ce, ids, tt, lens, P, L_in, out, st, host_ptrs);
}

int embed_run(rag_ctx* h, const int32_t* ids, const int32_t* tt, const int32_t* lens, int P, int L_in, float* out, hipStream_t st,
              bool host_ptrs) {
    ARG_CHECK(h, h->emb != nullptr, "no embedding model loaded");
    return ce_run(h, h->emb, ids, tt, lens, P, L_in, out, st, host_ptrs);
}

int embed_dim(const rag_ctx* h) { return h->emb ? h->emb->cfg.hidden : -1; }

// Dense cosine top-k over the HBM-resident corpus (K0-K3 of SURVEY.md §2).
//
// Replaces `ORDER BY embedding <=> q LIMIT k` (/root/reference/rag/document_store.py:448-460,
// database/operations.py:126-137) and the per-document Python cosine loop of
// HybridRetriever.hybrid_search (rag/retrieval.py:253-256).
//
// Pipeline per batch of Q queries (all on one stream, no host round trip):
//   normalize_rows   q -> fp16(2^7 * q/|q|)                                   (HBM-bound, tiny)
//   dense_emit       S~ = C16 . Q16^T on MFMA (fp16 in, fp32 acc), 256x256 tiles through LDS; the
//                    Q x N score matrix is never written: the epilogue emits only (score,row) keys
//                    with S~ >= tau[q] into a per-query candidate buffer.  Run as a few stages over
//                    growing row ranges; after each stage `select` finds the k-th best score so far,
//                    sets tau[q] = that - 2*eps and drops every key below it.
//   rescore          float64 cosine of the surviving rows (~50 per query at k = 20) against the fp32 master rows
//   finalize         order by (float64 cosine desc, row asc), write top-k. EXACTNESS IS STRUCTURAL: with
//                    |S~ - S| <= eps, a row below tau cannot be in the exact top-k (proof at select_kernel), so the
//                    survivors always contain it. Only two escapes exist:
//   wide             more than 256 survivors (tight clusters, duplicates): rank the whole buffer;
//   scan             the 4096-entry buffer overflowed: float64 exact scan of every row for that query.
#include "common.h"

#include <algorithm>
#include <unordered_map>

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// K0: one wave per row: out16[row] = fp16(2^7 * x/|x|), zero row when |x| is 0 or not finite.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ in, half_t* __restrict__ out,
                                                              int64_t n_rows, int dim, int dim_pad, int* bad_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave_global; row < n_rows; row += n_waves) {
        const float* x = in + row * dim;
        double acc = 0.0;
        for (int i = lane * 4; i < dim; i += 256) {
            float4 v = *reinterpret_cast<const float4*>(x + i);
            acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        const bool ok = (acc > 0.0) && (acc < 1e300);          // false for 0, inf and NaN
        const float inv = ok ? (float)((double)(1 << RAG_SCALE_LOG2) / sqrt(acc)) : 0.0f;
        if (!ok && lane == 0 && bad_rows) atomicAdd(bad_rows, 1);
        half_t* o = out + row * dim_pad;
        for (int i = lane * 4; i < dim_pad; i += 256) {
            half4 hv = {0, 0, 0, 0};
            if (i < dim) {
                float4 v = *reinterpret_cast<const float4*>(x + i);
                if (ok) {
                    hv[0] = (half_t)(v.x * inv);
                    hv[1] = (half_t)(v.y * inv);
                    hv[2] = (half_t)(v.z * inv);
                    hv[3] = (half_t)(v.w * inv);
                }
            }
            *reinterpret_cast<half4*>(o + i) = hv;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1: fused GEMM + threshold emit.   tile = 256 corpus rows (M) x 256 queries (N), BK = 64, 8 waves
// (2 along M x 4 along N, each 128 x 64), mfma_f32_16x16x32_f16, LDS double-buffered by LDS-DMA.
// Queries sit on the MFMA column (lane & 15) so a lane's threshold is one value per 16-wide column block.
// LDS tile image: [256 rows][8 chunks of 16 B], chunk c of row r stored at chunk position c ^ ((r>>1)&7)
// (conflict-free ds_read_b128; the swizzle is applied on the DMA *source* address, dest stays linear).
// Block -> tile map is XCD aware: the n_qtiles blocks that share one corpus tile run back-to-back on the
// same XCD (blockIdx % 8), so each corpus tile leaves HBM once and is re-read from that XCD's L2.
// ------------------------------------------------------------------------------------------------
#define HALF_BYTES (128 * RAG_BK * 2)         // one half-tile: 128 rows x 64 halfs = 16 KiB
#define TILE_BYTES (4 * HALF_BYTES)           // one K-step stage: A0 | A1 | B0 | B1
#define DENSE_LDS_BYTES (2 * TILE_BYTES)      // two stages = 128 KiB
#define DENSE_LDS_BYTES_SMALLQ (9 * HALF_BYTES)   // small-batch variant: 3 corpus stages (2 halves each) + 3 query half-tiles = 144 KiB

// One half-tile (128 rows x 128 B = 1024 chunks of 16 B) by LDS-DMA: 512 threads -> 2 pieces per thread; piece j of
// this wave lands at linear chunk j*512 + wid*64 (+lane): wave-uniform base + lane*16, as LDS-DMA requires.
__device__ __forceinline__ void stage_half(const half_t* __restrict__ gsrc, int ld, char* lds_half, int wid) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (size_t)j * 64 * ld),
                                         (__attribute__((address_space(3))) void*)(lds_half + (j * 512 + wid * 64) * 16),
                                         16, 0, 0);
}

struct FragA { half8 v[2][4]; };    // one 64-row M-sub of the wave's A half: [kk][i]
struct FragB { half8 v[2][2]; };    // one 32-col N-sub of the wave's B rows:  [kk][j]

__device__ __forceinline__ void load_fragA(FragA& f, const char* base, const int (&off_k)[2]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) f.v[kk][i] = *reinterpret_cast<const half8*>(base + i * 16 * 128 + off_k[kk]);
}
__device__ __forceinline__ void load_fragB(FragB& f, const char* base, const int (&off_k)[2]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j) f.v[kk][j] = *reinterpret_cast<const half8*>(base + j * 16 * 128 + off_k[kk]);
}

// Pipeline. Per 64-deep K-step t (LDS stage t & 1) a wave's 128x64 output is 2 M-subs x 2 N-subs (A0/A1 = 64-row
// sub-blocks of the wave's OWN corpus half, B0/B1 = 32-query sub-blocks of its own query half), done as TWO phases
// of 32 MFMAs: pA = (A0,B0),(A1,B0)   pB = (A1,B1),(A0,B1). Every phase is an I-part (LDS fragment reads + two
// half-tile LDS-DMA issues, then lgkmcnt(0)) and an M-part (32 MFMAs), each closed by s_barrier. The two waves that
// share a SIMD (wave w and w+4) run HALF A PHASE APART (waves 4-7 take one extra barrier up front,
// waves 0-3 one at the end): while one wave's MFMAs own the matrix pipe its partner reads fragments and issues DMA.
// (History, measured with tools/gemm_probe.hip: lock-step 4x16-MFMA phases 1131 TFLOP/s, staggered 4x16 1196;
// a barrier interval costs ~150 cycles beyond its 16 MFMAs = 256 cycles, hence 32-MFMA parts.)
//   reads : IA: A0(t) -> ax, A1(t) -> ay, B0(t) -> bx        IB: B1(t) -> by
//   DMA   : IA: Bh0(t+1), Bh1(t+1)                           IB: Ah0(t+2), Ah1(t+2)     (h0/h1 = 128-row half-tiles)
//   slot lifetimes (stage t&1): corpus halves are read in IA(t) only        -> refilled from IB(t)     (WAR ok)
//                               query  halves are read in IA(t) and IB(t)   -> refilled from IA(t+1)   (WAR ok)
//           a DMA is always issued in a LATER part than the slot's last read, reads retire (lgkmcnt(0)) before the
//           barrier closing their part, and the lagging group is only one barrier interval behind.
//   RAW   : one counted wait per step, s_waitcnt vmcnt(4) in the barrier interval before the leading group's
//           IA(t+1): waves 0-3 at the end of their MB(t), waves 4-7 at the end of their IB(t). In-order retirement
//           then guarantees A(t+1) (issued IB(t-1): three parts ahead, the HBM-latency operand) and B(t+1) (issued
//           IA(t)) have landed; only A(t+2) stays in flight. DMA never drains to 0 inside the loop; source steps
//           past the end are clamped (harmless re-loads) so the count stays exact.
#define MFMA_QUAD(FA, FB, SA, SB)                                                                           \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                           \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
        acc[(SA) * 4 + i][(SB) * 2 + j] =                                                                   \
            __builtin_amdgcn_mfma_f32_16x16x32_f16(FA.v[kk][i], FB.v[kk][j], acc[(SA) * 4 + i][(SB) * 2 + j], 0, 0, 0);

#define DMA_WAIT                                                                                            \
    if (SMALLQ) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                                           \
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#define BARRIER                                                                                             \
    __builtin_amdgcn_s_barrier();                                                                           \
    __builtin_amdgcn_sched_barrier(0);
// I-part: the caller has just written this phase's fragment reads and its DMA issue. WAIT = 1 on phase 3 only.
#define I_END(WAIT)                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    if ((WAIT) && lag) { DMA_WAIT }                                                                         \
    BARRIER
#define M_PART(FA0, FB0, SA0, SB0, FA1, FB1, SA1, SB1, WAIT)                                               \
    if (active) {                                                                                           \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        MFMA_QUAD(FA0, FB0, SA0, SB0)                                                                       \
        MFMA_QUAD(FA1, FB1, SA1, SB1)                                                                       \
        __builtin_amdgcn_s_setprio(0);                                                                      \
    }                                                                                                       \
    if ((WAIT) && !lag) { DMA_WAIT }                                                                        \
    BARRIER
// SMALLQ (batches of <= 128 queries, one query tile): the pass is HBM-bound, so the LDS that the unused query half-tile
// would take buys a THIRD corpus stage instead and the corpus DMA runs three K-steps ahead (queries two, in three
// half-tile slots): A(t+3) issued in IB(t), B(t+2) in IA(t); the per-step wait is vmcnt(10) = {A(t+2), B(t+2), A(t+3)}
// may still be in flight, A(t+1) and B(t+1) (both older in issue order) have landed. 64-96 KiB of corpus per CU in
// flight instead of 32-64 KiB.
// FUSED (rag_hybrid_linear_dev): the score that is thresholded and keyed is the weighted LINEAR fusion
// alpha * cosine + bias[q][row], bias = beta * bm25_normalised + gamma * temporal precomputed per (query, row) in float32
// (rag/retrieval.py:302); everything downstream (select, float64 rescoring, ranking) is unchanged.
#define EMIT_PARAMS                                                                                                   \
    const half_t *__restrict__ corpus16, const half_t *__restrict__ q16, int Dp, int rtile_begin, int n_rtiles, int n_qtiles,  \
        int n_rows_valid, int q_valid, const float *__restrict__ tau, unsigned *__restrict__ cnt, uint64_t *__restrict__ cand, \
        const int32_t *__restrict__ tenants, int tenant, const int32_t *__restrict__ tile_list, int tile_mul, int tile_mod,    \
        int tile_cnt, const int *__restrict__ active_count, const float *__restrict__ bias, int64_t bias_ld, float alpha,      \
        const int *__restrict__ qmap, const float *__restrict__ qscale, const float *__restrict__ gt
#define EMIT_PASS                                                                                                     \
    corpus16, q16, Dp, rtile_begin, n_rtiles, n_qtiles, n_rows_valid, q_valid, tau, cnt, cand, tenants, tenant, tile_list,    \
        tile_mul, tile_mod, tile_cnt, active_count, bias, bias_ld, alpha, qmap, qscale, gt
// one 256 x 256 tile; vb = the (virtual) block index that selects it
template <bool DENSE0, bool SMALLQ, bool FUSED>
__device__ __forceinline__ void dense_emit_tile(const int vb, EMIT_PARAMS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves w and w+4 share a SIMD. Group g = wid>>2 (0 = leading, 1 = lagging half a phase); inside a group the wave on
    // SIMD s owns corpus half wm = s&1 and the 64-query column block wn = (s>>1) + 2g. So a batch that fills only the
    // first 64 / 128 columns of the tile leaves every SIMD with at most ONE wave that has real work: the others skip
    // their fragment reads and MFMAs (they still issue DMA and meet every barrier), the MFMA time per K-step halves and
    // the small-batch case becomes purely HBM-bound.
    const int wm = wid & 1, wn = ((wid >> 1) & 1) + 2 * (wid >> 2);

    // second-pass launches (re-emission for queries whose buffer overflowed) cover the whole corpus but usually have
    // nothing to do: the device-side count of such queries decides, no host round trip
    if (active_count != nullptr && *active_count == 0) return;
    // XCD-aware tile assignment (speed only; any placement is correct)
    const int b = vb;
    const int xcd = b & 7, seq = b >> 3;
    const int rt = (seq / n_qtiles) * 8 + xcd;
    const int qt = seq % n_qtiles;
    if (rt >= n_rtiles) return;
    // ROW ORDER. The threshold stages must each see a REPRESENTATIVE sample of the rows, whatever order the table was
    // exported in (file by file, tenant by tenant, topic-sorted): position p of the schedule maps to corpus tile
    // (p * tile_mul) mod tile_mod - a multiplicative low-discrepancy permutation (tile_mul ~ 0.618 * tile_mod, coprime), so
    // every prefix of positions is spread evenly over the table - and, under a tenant filter, through tile_list: the tiles
    // that hold at least one row of that tenant (other tiles are never read). (Permuting GROUPS of 8 tiles instead was
    // measured and rejected: stage 0 is then one contiguous 2048-row stretch, and on the topic-sorted corpus every late
    // group overflowed the buffer: 612 queries/s.)
    const int pos = rtile_begin + rt;
    if (pos >= tile_cnt) return;
    int tile = (int)(((int64_t)pos * tile_mul) % tile_mod);
    if (tile_list != nullptr) tile = tile_list[tile];
    const int row0 = tile * RAG_TILE;
    const int q0 = qt * RAG_TILE;

    // per-thread DMA source inside a half-tile: linear chunk i = j*512 + tid -> row j*64 + (tid>>3), position tid&7
    const int sr = tid >> 3;
    const int schunk = (tid & 7) ^ ((sr >> 1) & 7);
    const half_t* a_src = corpus16 + (size_t)(row0 + sr) * Dp + schunk * 8;          // + h*128*Dp for half h
    const half_t* b_src = q16 + (size_t)(q0 + sr) * Dp + schunk * 8;
    const size_t half_rows = (size_t)128 * Dp;

    // fragment read offsets: row = base + (lane&15), chunk = kk*4 + (lane>>4), swizzled by ((row>>1)&7)
    const int fr = lane & 15, fq = lane >> 4;
    const int sw = (fr >> 1) & 7;
    int off_k[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) off_k[kk] = fr * 128 + (((kk * 4 + fq) ^ sw) << 4);
    // this wave reads A from half wm (rows s*64 + i*16 + fr) and B from half wn>>1 (rows (wn&1)*64 + s*32 + j*16 + fr)
    const int a_off = wm * HALF_BYTES;
    const int b_off = (SMALLQ ? 0 : (wn >> 1) * HALF_BYTES) + (wn & 1) * 64 * 128;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = Dp / RAG_BK;            // even (dim_pad is a multiple of 128)
    const int last = nt - 1;
#define SRC_STEP(u) (((u) < last ? (u) : last) * RAG_BK)
#define A_SLOT(u) (SMALLQ ? smem + ((u) % 3) * 2 * HALF_BYTES : smem + ((u) & 1) * TILE_BYTES)
#define B_SLOT(u) (SMALLQ ? smem + 6 * HALF_BYTES + ((u) % 3) * HALF_BYTES : smem + ((u) & 1) * TILE_BYTES + 2 * HALF_BYTES)
#define STAGE_A(h, u) stage_half(a_src + (h) * half_rows + SRC_STEP(u), Dp, A_SLOT(u) + (h) * HALF_BYTES, wid)
#define STAGE_B(h, u) stage_half(b_src + (h) * half_rows + SRC_STEP(u), Dp, B_SLOT(u) + (h) * HALF_BYTES, wid)
#define LDS_A(s, u) (A_SLOT(u) + a_off + (s) * 64 * 128)
#define LDS_B(s, u) (B_SLOT(u) + b_off + (s) * 32 * 128)

    // prologue: step 0 complete, corpus halves of step 1 in flight
    const bool lag = (wid >> 2) != 0;       // waves 4-7 run half a phase behind waves 0-3 (wave-uniform: from readfirstlane)
    // wave-uniform: this wave's 64 query columns hold at least one query (compile-time true in the full-batch variant)
    const bool active = !SMALLQ || (q0 + wn * 64 < q_valid);
    if (SMALLQ) {
        STAGE_B(0, 0); STAGE_A(0, 0); STAGE_A(1, 0);
        STAGE_B(0, 1); STAGE_A(0, 1); STAGE_A(1, 1);
        STAGE_A(0, 2); STAGE_A(1, 2);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else {
        STAGE_B(0, 0); STAGE_B(1, 0); STAGE_A(0, 0); STAGE_A(1, 0);
        STAGE_A(0, 1); STAGE_A(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    BARRIER
    FragA ax, ay;
    FragB bx, by;
    if (lag) { BARRIER }

    for (int t = 0; t < nt; ++t) {
        if (active) { load_fragA(ax, LDS_A(0, t), off_k);  load_fragA(ay, LDS_A(1, t), off_k);  load_fragB(bx, LDS_B(0, t), off_k); }
        if (SMALLQ) { STAGE_B(0, t + 2); } else { STAGE_B(0, t + 1);  STAGE_B(1, t + 1); }
        I_END(0)
        M_PART(ax, bx, 0, 0, ay, bx, 1, 0, 0)
        if (active) { load_fragB(by, LDS_B(1, t), off_k); }
        if (SMALLQ) { STAGE_A(0, t + 3);  STAGE_A(1, t + 3); } else { STAGE_A(0, t + 2);  STAGE_A(1, t + 2); }
        I_END(1)
        M_PART(ay, by, 1, 1, ax, by, 0, 1, 1)
    }
    if (!lag) { BARRIER }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // clamped tail re-loads still in flight: retire them

    // ---- epilogue: C layout col = lane&15 (query), row = (lane>>4)*4 + reg (corpus row) ----------
    const float scale = 1.0f / (float)(1 << (2 * RAG_SCALE_LOG2));
    if (FUSED) {
        // accumulators -> fused score, kept in the accumulators' 2^14 scale so that the threshold / key code below is shared:
        // acc' = acc * alpha + (raw32[q][row] * qscale[q] + gt[row]) * 2^14 with raw32 the float32 raw BM25 score, qscale = beta / max
        // of the query and gt[row] = gamma * temporal (round 4: the per-(query, row) bias array that held this sum - a 2-GB read and a
        // 1-GB write per 256 queries to build - is gone; the scoring kernel writes raw32 itself). Each (query, row) element is read
        // exactly once per pass (float4 per 4 rows); the float32 roundings are inside the emission margin (dense_search).
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + wn * 64 + j * 16 + fr;
            if (q >= q_valid) continue;
            const int qi = qmap != nullptr ? qmap[q] : q;
            const float qs = qscale[qi];
            const float* brow = bias + (size_t)qi * bias_ld + row0 + wm * 128 + fq * 4;
            const float* grow = gt != nullptr ? gt + row0 + wm * 128 + fq * 4 : nullptr;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 b = *reinterpret_cast<const float4*>(brow + i * 16);
                const float4 g = grow != nullptr ? *reinterpret_cast<const float4*>(grow + i * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
                acc[i][j][0] = fmaf(acc[i][j][0], alpha, fmaf(b.x, qs, g.x) * (float)(1 << (2 * RAG_SCALE_LOG2)));
                acc[i][j][1] = fmaf(acc[i][j][1], alpha, fmaf(b.y, qs, g.y) * (float)(1 << (2 * RAG_SCALE_LOG2)));
                acc[i][j][2] = fmaf(acc[i][j][2], alpha, fmaf(b.z, qs, g.z) * (float)(1 << (2 * RAG_SCALE_LOG2)));
                acc[i][j][3] = fmaf(acc[i][j][3], alpha, fmaf(b.w, qs, g.w) * (float)(1 << (2 * RAG_SCALE_LOG2)));
            }
        }
    }
    if (DENSE0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + wn * 64 + j * 16 + fr;
            if (q >= q_valid) continue;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // every dense slot is written (0 = empty for padded / filtered rows): no memset needed beforehand; the lane's
                // 4 consecutive rows go out as two 16-byte stores
                const int rbase = row0 + wm * 128 + i * 16 + fq * 4;
                uint64_t kk[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rbase + r;
                    const bool ok = row < n_rows_valid && (tenants == nullptr || tenants[row] == tenant);
                    kk[r] = ok ? make_key(acc[i][j][r] * scale, (uint32_t)row) : 0ull;
                }
                ulonglong2* dst = reinterpret_cast<ulonglong2*>(cand + (size_t)q * RAG_CAND_CAP + (rbase - row0) + pos * RAG_TILE);
                dst[0] = make_ulonglong2(kk[0], kk[1]);
                dst[1] = make_ulonglong2(kk[2], kk[3]);
            }
        }
        return;
    }
    // Thresholded emission, aggregated per lane:
    //  (1) per column block: max of the lane's 32 scores against the threshold (cheap reject, ~45 % of blocks have
    //      no hit at all), hit mask only for lanes that have one;
    //  (2) rare fix-ups (rows past the end of the corpus in the last tile, tenant filter) in a rolled loop over set bits;
    //  (3) ONE returning atomic per (lane, column block), all four issued before any result is consumed;
    //  (4) key stores. The compare is done on raw accumulators against tau * 2^14 (exact: power-of-two scale).
    unsigned hits[4], slot[4];
    const bool fixups = (row0 + RAG_TILE > n_rows_valid) || (tenants != nullptr);       // block-uniform
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = q0 + wn * 64 + j * 16 + fr;
        const float thr = (q < q_valid ? tau[q] : INFINITY) * (float)(1 << (2 * RAG_SCALE_LOG2));
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            mx = fmaxf(mx, fmaxf(fmaxf(acc[i][j][0], acc[i][j][1]), fmaxf(acc[i][j][2], acc[i][j][3])));
        unsigned h = 0u;
        if (mx >= thr) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) h |= (acc[i][j][r] >= thr) ? (1u << (i * 4 + r)) : 0u;
            if (fixups) {
                unsigned rem = h;
#pragma unroll 1
                while (rem) {
                    const int bit = __ffs(rem) - 1;
                    rem &= rem - 1;
                    const int row = row0 + wm * 128 + (bit >> 2) * 16 + fq * 4 + (bit & 3);
                    if (row >= n_rows_valid || (tenants != nullptr && tenants[row] != tenant)) h &= ~(1u << bit);
                }
            }
        }
        hits[j] = h;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = q0 + wn * 64 + j * 16 + fr;
        slot[j] = hits[j] ? atomicAdd(&cnt[q], (unsigned)__popc(hits[j])) : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!hits[j]) continue;
        const int q = q0 + wn * 64 + j * 16 + fr;
        uint64_t* dst = cand + (size_t)q * RAG_CAND_CAP;
        unsigned s_ = slot[j];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (hits[j] & (1u << (i * 4 + r))) {
                    if (s_ < RAG_CAND_CAP)
                        dst[s_] = make_key(acc[i][j][r] * scale, (uint32_t)(row0 + wm * 128 + i * 16 + fq * 4 + r));
                    ++s_;
                }
    }
}

template <bool DENSE0, bool SMALLQ, bool FUSED = false>
__global__ __launch_bounds__(512) void dense_emit_kernel(EMIT_PARAMS) {
    dense_emit_tile<DENSE0, SMALLQ, FUSED>(blockIdx.x, EMIT_PASS);
}

// Second pass (queries whose buffer overflowed): almost always there is nothing to do, and a corpus-sized grid of 128 KiB-LDS
// workgroups costs ~40 us just to be dispatched and retired. One workgroup per CU walks the tiles instead; idle, the launch
// costs one read of the device-side count per workgroup.
template <bool FUSED>
__global__ __launch_bounds__(512) void dense_emit_persist_kernel(int n_vblocks, EMIT_PARAMS) {
    if (active_count != nullptr && *active_count == 0) return;
    for (int vb = blockIdx.x; vb < n_vblocks; vb += gridDim.x) {
        dense_emit_tile<false, false, FUSED>(vb, EMIT_PASS);
        __syncthreads();                                   // the next tile's prologue refills LDS stages this tile still reads
    }
}

// ------------------------------------------------------------------------------------------------
// K2: per-query select as a PER-WAVEFRONT bitwise radix select (one wave per query, 4 queries per workgroup).
// Keys are unique 64-bit values (orderable score << 32 | ~row), so "the k-th largest key" is a total-order pivot:
// build it MSB-first, one bit per step, counting keys >= candidate with ballot + popcount (no atomics, no sort, no
// cross-lane reduction); stop as soon as a candidate has exactly k keys above it.
//
// Threshold rule (this is what makes the result PROVABLY the exact scan's): let s~(k) be the k-th best fp16-pass score
// among the rows seen so far and eps the bound |s~ - s| <= eps. The final exact k-th score satisfies
// s_k >= s~(k) - eps (the k rows with the best s~ all have s >= s~(k) - eps), and a row with s~ < s~(k) - 2 eps has
// s <= s~ + eps < s~(k) - eps <= s_k: it cannot be in the top-k. So tau = s~(k) - 2 eps is a safe emission threshold
// (s~(k) only grows as more rows are seen) and every key below it can be dropped for good. What survives (typically
// ~1.5 k keys) is rescored in float64 and ranked; no a-posteriori check is needed unless the buffer overflowed.
//   n_in = dense0_rows (stage 0: slots 0..rows-1, empty slots are 0) or min(cnt, cap)
//   out  : cand[0..m) = all keys with score >= tau (unordered), cnt = m, tau updated;
//          final stage: n_sorted = m; bound = +inf if the buffer ever overflowed (candidates were lost), else -inf
// ------------------------------------------------------------------------------------------------
#define SEL_REG 32                      // keys held in registers per lane (covers 2048 candidates)
#define SEL_REG_SMALL 8                 // a thresholded stage usually leaves ~k x growth + k keys: 512 cover it
#define SEL_REG_MID 16                  // ... for k = 20; a pool of 100 (the hybrid legs) leaves ~900: 1024 cover that (22 -> ~12 us per select)
#define SELECT_LDS_BYTES (4 * (RAG_CAND_CAP - SEL_REG * 64) * 8)
// One wave, one query. NREG = registers of keys per lane: the pivot search costs 64 bit-steps x NREG ballots whatever n_in is,
// so the common small case runs with a quarter of the registers (3 of the 4 selects of a 1M-row search: 16 -> 5 us each).
template <int NREG>
__device__ __forceinline__ void select_wave(uint64_t* __restrict__ c, uint64_t* __restrict__ spill, int n_in, int k, float two_eps,
                                            float tau_in, int lane, float& tau_new, int& n_top, uint64_t* __restrict__ copy_to) {
    // keys: first NREG*64 in registers (element e*64 + lane), the rest (rare) in this wave's LDS slice
    uint64_t kreg[NREG];
#pragma unroll
    for (int e = 0; e < NREG; ++e) {
        const int i = e * 64 + lane;
        kreg[e] = i < n_in ? c[i] : 0ull;
    }
    const int n_spill = max(0, n_in - NREG * 64);
    const int n_spill_pad = (n_spill + 63) & ~63;
    for (int i = lane; i < n_spill; i += 64) spill[i] = c[NREG * 64 + i];
    int n_valid = 0;
#pragma unroll
    for (int e = 0; e < NREG; ++e) n_valid += __popcll(__ballot(kreg[e] != 0ull));
    for (int i = lane; i < n_spill_pad; i += 64) n_valid += __popcll(__ballot(i < n_spill && spill[i] != 0ull));
    tau_new = tau_in;
    if (n_valid >= k) {
        uint64_t pivot = 0ull;                               // becomes (a lower bound of) the k-th largest key
        for (int bit = 63; bit >= 0; --bit) {
            const uint64_t trial = pivot | (1ull << bit);
            int ge = 0;
#pragma unroll
            for (int e = 0; e < NREG; ++e) ge += __popcll(__ballot(kreg[e] >= trial));
            for (int i = lane; i < n_spill_pad; i += 64) ge += __popcll(__ballot(i < n_spill && spill[i] >= trial));
            if (ge >= k) pivot = trial;
            if (ge == k) break;          // exactly k keys are >= pivot; its score bits are <= the k-th best score: safe
        }
        tau_new = fmaxf(tau_in, key_score(pivot) - two_eps);
    }
    // compaction: every valid key with score >= tau_new moves to the front (all keys were loaded before any store)
    const uint64_t cut = (uint64_t)f32_orderable(tau_new) << 32;
    n_top = 0;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int e = 0; e < NREG; ++e) {
        const uint64_t key = kreg[e];
        const bool top = key != 0ull && key >= cut;
        const uint64_t bt = __ballot(top);
        if (top) {
            c[n_top + __popcll(bt & lt_mask)] = key;
            if (copy_to != nullptr) copy_to[n_top + __popcll(bt & lt_mask)] = key;
        }
        n_top += __popcll(bt);
    }
    for (int i0 = 0; i0 < n_spill; i0 += 64) {
        const int i = i0 + lane;
        const uint64_t key = i < n_spill ? spill[i] : 0ull;
        const bool top = key != 0ull && key >= cut;
        const uint64_t bt = __ballot(top);
        if (top) {
            c[n_top + __popcll(bt & lt_mask)] = key;
            if (copy_to != nullptr) copy_to[n_top + __popcll(bt & lt_mask)] = key;
        }
        n_top += __popcll(bt);
    }
}

// Two jobs ride on it so that an uneventful search launches nothing extra (each idle launch costs ~3-5 us on a small shard):
//   ovf_list / ovf_count (the FINAL first-pass select): queries whose buffer overflowed at any stage append themselves
//     (count may exceed the RAG_TILE slots of the list: readers clamp);
//   sc_list (the second-pass select): a re-emission that fitted replaces the query's candidate list - the compacted keys are
//     written to the query's own buffer straight from registers - and clears its overflow mark.
struct select_extra {
    int* ovf_list; int* ovf_count;
    const int* sc_list; uint64_t* sc_cand; int* sc_n_sorted; float* sc_bound;
};
__global__ __launch_bounds__(256) void select_kernel(uint64_t* __restrict__ cand, unsigned* __restrict__ cnt,
                                                      float* __restrict__ tau, float* __restrict__ bound,
                                                      int* __restrict__ n_sorted, int* __restrict__ stats, int n_queries,
                                                      int dense0_rows, int k, float two_eps, int final_stage,
                                                      const int* __restrict__ active_count, select_extra ex) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wv;
    if (q >= n_queries) return;                              // whole wave exits; no block-level sync is used
    if (active_count != nullptr && q >= min(*active_count, RAG_TILE)) return;   // second pass: only the re-emitted queries
    uint64_t* spill = reinterpret_cast<uint64_t*>(smem) + (size_t)wv * (RAG_CAND_CAP - SEL_REG * 64);
    uint64_t* c = cand + (size_t)q * RAG_CAND_CAP;
    const unsigned emitted = cnt[q];
    const bool overflow = dense0_rows == 0 && emitted > RAG_CAND_CAP;
    const int n_in = dense0_rows > 0 ? dense0_rows : (int)min(emitted, (unsigned)RAG_CAND_CAP);
    if (overflow && lane == 0 && stats != nullptr && ex.sc_list == nullptr) atomicAdd(&stats[4], 1);   // first-pass overflow events
    const float tau_in = tau[q];
    float tau_new;
    int n_top;
    const int dst_q = ex.sc_list != nullptr ? ex.sc_list[q] : 0;
    uint64_t* copy_to = (ex.sc_list != nullptr && !overflow) ? ex.sc_cand + (size_t)dst_q * RAG_CAND_CAP : nullptr;
    if (n_in <= SEL_REG_SMALL * 64) select_wave<SEL_REG_SMALL>(c, spill, n_in, k, two_eps, tau_in, lane, tau_new, n_top, copy_to);
    else if (n_in <= SEL_REG_MID * 64) select_wave<SEL_REG_MID>(c, spill, n_in, k, two_eps, tau_in, lane, tau_new, n_top, copy_to);
    else select_wave<SEL_REG>(c, spill, n_in, k, two_eps, tau_in, lane, tau_new, n_top, copy_to);
    if (lane == 0) {
        // bound[] starts at -inf; an overflow at ANY stage lost candidates for good -> sticky +inf: the query goes to the
        // second pass, and from there to the exact scan if it overflows again.
        const bool lost = overflow || bound[q] == INFINITY;
        if (overflow) bound[q] = INFINITY;
        if (final_stage) n_sorted[q] = n_top;
        cnt[q] = (unsigned)n_top;
        tau[q] = tau_new;
        if (ex.ovf_list != nullptr && final_stage && lost) {
            const int o = atomicAdd(ex.ovf_count, 1);
            if (o < RAG_TILE) ex.ovf_list[o] = q;
        }
        if (copy_to != nullptr) {                            // second pass fitted: the query is proven again
            ex.sc_n_sorted[dst_q] = n_top;
            ex.sc_bound[dst_q] = -INFINITY;
            atomicAdd(&stats[5], 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3: float64 cosine of shortlisted rows. One wave per (query, candidate).
// cos = dot / (sqrt(|q|^2) * sqrt(|c|^2)), 0.0 when a norm is 0 or not finite (rag/retrieval.py:362-371).
// ------------------------------------------------------------------------------------------------
// |q|^2 of a query the way exact_cosine_wave accumulates and reduces it (same lane partition, same butterfly: the same bits), for
// callers that score many rows against one query
__device__ __forceinline__ double exact_norm2_wave(const float* __restrict__ qv, int dim, int lane) {
    double nq = 0.0;
    for (int i = lane * 4; i < dim; i += 256) {
        const float4 a = *reinterpret_cast<const float4*>(qv + i);
        nq += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nq += __shfl_xor(nq, o);
    return nq;
}
// KNOWN_NQ: |q|^2 comes from exact_norm2_wave (a third of the float64 work and of the cross-lane steps per row less)
template <bool KNOWN_NQ = false>
__device__ __forceinline__ double exact_cosine_wave(const float* __restrict__ qv, const float* __restrict__ cv, int dim,
                                                    int lane, double nq_known = 0.0) {
    double dot = 0.0, nq = 0.0, nc = 0.0;
    for (int i = lane * 4; i < dim; i += 256) {
        const float4 a = *reinterpret_cast<const float4*>(qv + i);
        const float4 c = *reinterpret_cast<const float4*>(cv + i);
        dot += (double)a.x * c.x + (double)a.y * c.y + (double)a.z * c.z + (double)a.w * c.w;
        if (!KNOWN_NQ) nq += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
        nc += (double)c.x * c.x + (double)c.y * c.y + (double)c.z * c.z + (double)c.w * c.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dot += __shfl_xor(dot, o);
        if (!KNOWN_NQ) nq += __shfl_xor(nq, o);
        nc += __shfl_xor(nc, o);
    }
    if (KNOWN_NQ) nq = nq_known;
    const bool ok = (nq > 0.0) && (nq < 1e300) && (nc > 0.0) && (nc < 1e300);
    return ok ? dot / (sqrt(nq) * sqrt(nc)) : 0.0;
}

__global__ __launch_bounds__(256) void rescore_kernel(const float* __restrict__ q32, const float* __restrict__ emb32,
                                                       const uint64_t* __restrict__ cand, const int* __restrict__ n_sorted,
                                                       double* __restrict__ exact, int dim) {
    const int q = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int m = n_sorted[q];
    if ((int)(blockIdx.x * 4 + (threadIdx.x >> 6)) >= m) return;       // (wave-uniform) nothing for this wave
    const double nq = exact_norm2_wave(q32 + (size_t)q * dim, dim, lane);
    for (int j = blockIdx.x * 4 + (threadIdx.x >> 6); j < m; j += gridDim.x * 4) {
        const uint32_t row = key_row(cand[(size_t)q * RAG_CAND_CAP + j]);
        const double c = exact_cosine_wave<true>(q32 + (size_t)q * dim, emb32 + (size_t)row * dim, dim, lane, nq);
        if (lane == 0) exact[(size_t)q * RAG_CAND_CAP + j] = c;
    }
}

// order the survivors by (exact desc, row asc) and write the top-k. Up to RAG_MAX_K survivors are ranked in one step; more
// (tight clusters / many duplicates: up to the buffer's 4096) are ranked by the same workgroup in blocks of 256 through the
// same LDS (each thread keeps up to 16 of the entries and counts, block by block, how many others come before each); a query
// whose buffer overflowed in the second pass too appends itself to the exact-scan list. (Round 2 used three more launches for
// this - wide_kernel and a flag-list kernel - that an uneventful search paid for without using them.)
#define FIN_PER_THREAD (RAG_CAND_CAP / 256)
__global__ __launch_bounds__(256) void finalize_kernel(const uint64_t* __restrict__ cand, const int* __restrict__ n_sorted,
                                                        const double* __restrict__ exact, const float* __restrict__ bound,
                                                        const int64_t* __restrict__ ids, int64_t id_base, int k,
                                                        int force_level, int64_t* __restrict__ ids_out,
                                                        int32_t* __restrict__ rows_out, double* __restrict__ scores_out,
                                                        int* __restrict__ flag, int* __restrict__ stats,
                                                        int* __restrict__ scan_list, int* __restrict__ scan_count) {
    __shared__ double sc[RAG_MAX_K];
    __shared__ uint32_t rw[RAG_MAX_K];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int m = n_sorted[q];
    for (int i = tid; i < k; i += 256) {
        ids_out[(size_t)q * k + i] = -1;
        if (rows_out) rows_out[(size_t)q * k + i] = -1;
        scores_out[(size_t)q * k + i] = 0.0;
    }
    const bool overflowed = bound[q] == INFINITY;
    if (overflowed || force_level > 1) {
        if (tid == 0) {
            flag[q] = 2;
            scan_list[atomicAdd(scan_count, 1)] = q;
        }
        return;
    }
    const uint64_t* c = cand + (size_t)q * RAG_CAND_CAP;
    const double* ex = exact + (size_t)q * RAG_CAND_CAP;
    if (m > RAG_MAX_K || force_level > 0) {
        double e[FIN_PER_THREAD];
        uint32_t r[FIN_PER_THREAD];
        int rank[FIN_PER_THREAD];
#pragma unroll
        for (int j = 0; j < FIN_PER_THREAD; ++j) {
            const int i = j * 256 + tid;
            e[j] = i < m ? ex[i] : 0.0;
            r[j] = i < m ? key_row(c[i]) : 0u;
            rank[j] = 0;
        }
        for (int u0 = 0; u0 < m; u0 += RAG_MAX_K) {
            const int nu = min(RAG_MAX_K, m - u0);
            __syncthreads();                               // the previous block's readers are done
            if (tid < nu) {
                sc[tid] = ex[u0 + tid];
                rw[tid] = key_row(c[u0 + tid]);
            }
            __syncthreads();
            for (int u = 0; u < nu; ++u) {
                const double su = sc[u];
                const uint32_t ru = rw[u];
#pragma unroll
                for (int j = 0; j < FIN_PER_THREAD; ++j) rank[j] += (su > e[j]) || (su == e[j] && ru < r[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < FIN_PER_THREAD; ++j) {
            const int i = j * 256 + tid;
            if (i < m && rank[j] < k) {
                ids_out[(size_t)q * k + rank[j]] = ids ? ids[r[j]] : id_base + (int64_t)r[j];
                if (rows_out) rows_out[(size_t)q * k + rank[j]] = (int32_t)r[j];
                scores_out[(size_t)q * k + rank[j]] = e[j];
            }
        }
        if (tid == 0) {
            flag[q] = 0;
            atomicAdd(&stats[1], 1);
        }
        return;
    }
    if (tid < m) {
        sc[tid] = ex[tid];
        rw[tid] = key_row(c[tid]);
    }
    __syncthreads();
    if (tid < m) {
        const double e = sc[tid];
        const uint32_t r = rw[tid];
        int rank = 0;
        for (int u = 0; u < m; ++u) rank += (sc[u] > e) || (sc[u] == e && rw[u] < r);
        if (rank < k) {
            ids_out[(size_t)q * k + rank] = ids ? ids[r] : id_base + (int64_t)r;
            if (rows_out) rows_out[(size_t)q * k + rank] = (int32_t)r;
            scores_out[(size_t)q * k + rank] = e;
        }
    }
    if (tid == 0) {
        flag[q] = 0;
        atomicAdd(&stats[0], 1);
    }
}

// L3: exact float64 scan of every row for queries with flag == 2 (last resort: more than RAG_CAND_CAP rows within 2 eps
// of the k-th best even after the second pass, e.g. thousands of duplicates).
// Scratch is BOUNDED: flagged queries are handled in rounds of SCAN_ROUND ordinals, and the corpus is cut into at most
// ~1024 row blocks whatever its size (each workgroup walks its block in windows, keeping a running top-k), so a search that
// flags nothing never needs more than SCAN_ROUND x 1024 x k partial entries (r1: Q x chunks x k = 7.5 GB on a 12.5M-row shard).
#define SCAN_CHUNK 2048
#define SCAN_ROUND 256
__device__ __forceinline__ bool pair_before(uint64_t ka, uint32_t ra, uint64_t kb, uint32_t rb) {
    return ka > kb || (ka == kb && ra < rb);      // score desc, row asc
}
__device__ __forceinline__ void bitonic_sort_pairs(uint64_t* k1, uint32_t* k2, int P, int tid, int nthreads) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += nthreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool first_block = ((i & k) == 0);
                    const bool a_before_b = pair_before(k1[i], k2[i], k1[ixj], k2[ixj]);
                    if (first_block ? !a_before_b : a_before_b) {
                        const uint64_t t1 = k1[i]; k1[i] = k1[ixj]; k1[ixj] = t1;
                        const uint32_t t2 = k2[i]; k2[i] = k2[ixj]; k2[ixj] = t2;
                    }
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void scan_chunk_kernel(const float* __restrict__ q32, const float* __restrict__ emb32,
                                                          const int32_t* __restrict__ tenants, int tenant, int64_t n_rows,
                                                          int64_t rows_per_block, int dim, int k, const int* __restrict__ list,
                                                          const int* __restrict__ count, int f0, int round_q,
                                                          uint64_t* __restrict__ part_key,
                                                          uint32_t* __restrict__ part_row, const double* __restrict__ raw,
                                                          int64_t raw_ld, const double* __restrict__ raw_mx,
                                                          const double* __restrict__ temporal, double fa, double fb, double fg) {
    __shared__ uint64_t sk[SCAN_CHUNK];
    __shared__ uint32_t sr[SCAN_CHUNK];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int f_end = min(*count, f0 + round_q);
    const int64_t base = (int64_t)blockIdx.x * rows_per_block;
    const int64_t base_end = min(n_rows, base + rows_per_block);
    const int window = SCAN_CHUNK - k;
    for (int f = f0; f < f_end; ++f) {
        const int q = list[f];
        for (int i = tid; i < k; i += 256) { sk[i] = 0ull; sr[i] = 0xFFFFFFFFu; }      // running top-k of this row block
        for (int64_t w0 = base; w0 < base_end; w0 += window) {
            for (int i = wv; i < window; i += 4) {
                const int64_t row = w0 + i;
                uint64_t key = 0ull;          // 0 = empty (below every real score: orderable(-inf) > 0)
                if (row < base_end && (tenants == nullptr || tenants[row] == tenant)) {
                    double v = exact_cosine_wave(q32 + (size_t)q * dim, emb32 + (size_t)row * dim, dim, lane);
                    if (raw != nullptr)               // linear fusion (rag/retrieval.py:302), same operation order as linear_fuse_kernel
                        v = (fa * v + fb * (raw[(size_t)q * raw_ld + row] / raw_mx[q])) + fg * (temporal ? temporal[row] : 0.0);
                    key = f64_orderable(v);
                }
                if (lane == 0) {
                    sk[k + i] = key;
                    sr[k + i] = (uint32_t)row;
                }
            }
            __syncthreads();
            bitonic_sort_pairs(sk, sr, SCAN_CHUNK, tid, 256);
        }
        const size_t o = ((size_t)(f - f0) * gridDim.x + blockIdx.x) * k;
        for (int i = tid; i < k; i += 256) {
            part_key[o + i] = sk[i];
            part_row[o + i] = sr[i];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void scan_merge_kernel(const uint64_t* __restrict__ part_key, const uint32_t* __restrict__ part_row,
                                                          int n_blocks, int k, const int64_t* __restrict__ ids, int64_t id_base,
                                                          const int* __restrict__ list, const int* __restrict__ count, int f0,
                                                          int* __restrict__ flag, int64_t* __restrict__ ids_out,
                                                          int32_t* __restrict__ rows_out, double* __restrict__ scores_out,
                                                          int* __restrict__ stats) {
    __shared__ uint64_t sk[SCAN_CHUNK];
    __shared__ uint32_t sr[SCAN_CHUNK];
    const int f = f0 + blockIdx.x, tid = threadIdx.x;
    if (f >= *count) return;
    const int q = list[f];
    const size_t total = (size_t)n_blocks * k;
    const uint64_t* pk = part_key + (size_t)blockIdx.x * total;
    const uint32_t* pr = part_row + (size_t)blockIdx.x * total;
    for (int i = tid; i < SCAN_CHUNK; i += 256) { sk[i] = 0ull; sr[i] = 0xFFFFFFFFu; }
    __syncthreads();
    size_t pos = 0;
    while (pos < total) {
        const int room = SCAN_CHUNK - k;              // slots [k, CHUNK) take new partials
        const int take = (int)min((size_t)room, total - pos);
        for (int i = tid; i < room; i += 256) {
            sk[k + i] = i < take ? pk[pos + i] : 0ull;
            sr[k + i] = i < take ? pr[pos + i] : 0xFFFFFFFFu;
        }
        __syncthreads();
        bitonic_sort_pairs(sk, sr, SCAN_CHUNK, tid, 256);
        pos += take;
    }
    for (int i = tid; i < k; i += 256) {
        const bool ok = sk[i] != 0ull;
        uint64_t u = sk[i];
        u = (u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u;
        const uint32_t r = sr[i];
        ids_out[(size_t)q * k + i] = ok ? (ids ? ids[r] : id_base + (int64_t)r) : -1;
        if (rows_out) rows_out[(size_t)q * k + i] = ok ? (int32_t)r : -1;
        scores_out[(size_t)q * k + i] = ok ? __builtin_bit_cast(double, u) : 0.0;
    }
    if (tid == 0) {
        flag[q] = 3;
        atomicAdd(&stats[2], 1);
    }
}

// ---- second pass for queries whose candidate buffer overflowed -------------------------------------------------------
// An overflow loses candidates, but the select that follows still tightens tau from the keys that were kept (the k-th best
// of ANY subset of the rows seen is a valid lower bound of the final k-th best). With the final tau the number of rows
// above it is almost always small again, so those queries (up to 256 per search) are re-emitted over the whole corpus with
// that tau into a fresh buffer by the same MFMA kernel; only if THAT overflows too does the query go to the float64 scan.
// gather: workgroup f copies the fp16 query row and tau of the f-th overflowed query; unused slots get tau = +inf.
// (the list was appended to by the final select; its dead slots are pointed at query 0 here: the fused re-emission reads the
// bias row of every slot through it, and a stale entry of an earlier, larger batch would index outside the bias buffer)
__global__ __launch_bounds__(256) void overflow_gather_kernel(int* __restrict__ list, const int* __restrict__ count,
                                                               const half_t* __restrict__ q16, const float* __restrict__ tau, int Dp,
                                                               half_t* __restrict__ q16b, float* __restrict__ taub,
                                                               float* __restrict__ boundb, unsigned* __restrict__ cntb) {
    const int f = blockIdx.x;
    const bool live = f < min(*count, RAG_TILE);
    if (threadIdx.x == 0) {
        if (!live) list[f] = 0;
        taub[f] = live ? tau[list[f]] : INFINITY;
        boundb[f] = -INFINITY;
        cntb[f] = 0u;
    }
    if (!live) return;
    const half8* src = reinterpret_cast<const half8*>(q16 + (size_t)list[f] * Dp);
    half8* dst = reinterpret_cast<half8*>(q16b + (size_t)f * Dp);
    for (int i = threadIdx.x; i < Dp / 8; i += 256) dst[i] = src[i];
}

// ---- linear fusion over the resident index (rag_hybrid_linear_dev) -------------------------------------------------------
// rag/retrieval.py:294-322 evaluated over ALL rows: hybrid = (alpha * cosine + beta * keyword) + gamma * temporal, keyword =
// raw BM25 / max over the corpus (1.0 when that max is <= 0), stable sort descending, [:top_k].
// raw[Q][N] are the float64 BM25 scores of every document (bm25_range_kernel, mode 1), max_key[q] the orderable key of the largest
// one over the tenant's documents (atomicMax in that same kernel; 0 = no document). Under a tenant filter the corpus hybrid_search
// was handed is the tenant's own documents (`WHERE agent_id = %s`, rag/document_store.py:457), as rag_bm25_topk_* do.
// -> mx[q] = `max_score if max_score > 0 else 1.0` (rag/retrieval.py:343-344), qscale[q] = float(beta / mx[q]) for the emission.
__global__ void linear_scale_kernel(const unsigned long long* __restrict__ max_key, int Q, double beta, double* __restrict__ mx,
                                    float* __restrict__ qscale) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const unsigned long long kq = max_key[q];
    double m = -INFINITY;
    if (kq != 0ull) {
        const unsigned long long u = (kq & 0x8000000000000000ull) ? (kq & 0x7fffffffffffffffull) : ~kq;
        m = __builtin_bit_cast(double, u);
    }
    const double d = m > 0.0 ? m : 1.0;
    mx[q] = d;
    qscale[q] = (float)(beta / d);
}

// float32 recency term of every row for the emission: gamma * temporal (zero past the last row)
__global__ void linear_gt_kernel(const double* __restrict__ temporal, int64_t n, int64_t ld, double gamma, float* __restrict__ gt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ld) gt[i] = i < n ? (float)(gamma * temporal[i]) : 0.f;
}

// survivors: exact[q][j] holds the float64 cosine (rescore_kernel) -> the float64 hybrid score, CPython's operation order
__global__ __launch_bounds__(256) void linear_fuse_kernel(const uint64_t* __restrict__ cand, const int* __restrict__ n_sorted,
                                                           double* __restrict__ exact, const double* __restrict__ raw, int64_t n,
                                                           const double* __restrict__ mx, const double* __restrict__ temporal,
                                                           double alpha, double beta, double gamma) {
    const int q = blockIdx.y;
    const int m = n_sorted[q];
    for (int j = blockIdx.x * 256 + threadIdx.x; j < m; j += gridDim.x * 256) {
        const uint32_t row = key_row(cand[(size_t)q * RAG_CAND_CAP + j]);
        const double sem = exact[(size_t)q * RAG_CAND_CAP + j];
        const double kw = raw[(size_t)q * n + row] / mx[q];
        exact[(size_t)q * RAG_CAND_CAP + j] = (alpha * sem + beta * kw) + gamma * (temporal ? temporal[row] : 0.0);
    }
}

// the three components of the returned rows (semantic_score, keyword_score, temporal_score of the reference's result dicts)
__global__ __launch_bounds__(256) void linear_components_kernel(const float* __restrict__ q32, const float* __restrict__ emb32,
                                                                 const int32_t* __restrict__ rows, int Q, int k, int dim,
                                                                 const double* __restrict__ raw, int64_t n,
                                                                 const double* __restrict__ mx, const double* __restrict__ temporal,
                                                                 double* __restrict__ sem_out, double* __restrict__ kw_out,
                                                                 double* __restrict__ tmp_out) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= (int64_t)Q * k) return;
    const int q = (int)(e / k);
    const int32_t row = rows[e];
    double sem = 0.0, kw = 0.0, t = 0.0;
    if (row >= 0) {
        sem = exact_cosine_wave(q32 + (size_t)q * dim, emb32 + (size_t)row * dim, dim, lane);
        kw = raw[(size_t)q * n + row] / mx[q];
        t = temporal ? temporal[row] : 0.0;
    }
    if (lane == 0) {
        if (sem_out) sem_out[e] = sem;
        if (kw_out) kw_out[e] = kw;
        if (tmp_out) tmp_out[e] = t;
    }
}

// per-search state in one launch: thresholds / proof bounds to -inf, candidate counters and statistics to 0
__global__ void search_init_kernel(float* __restrict__ tau, float* __restrict__ bound, unsigned* __restrict__ cnt,
                                   int* __restrict__ stats, int* __restrict__ ovf_count, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        tau[i] = -INFINITY;
        bound[i] = -INFINITY;
        cnt[i] = 0u;
    }
    if (i < 8) stats[i] = 0;
    if (i == 0) *ovf_count = 0;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int ensure_workspace(rag_ctx* h, int Q, hipStream_t st) {
    if (Q <= h->ws_q) return RAG_OK;
    hipFree(h->q32); hipFree(h->q16); hipFree(h->cand); hipFree(h->cnt); hipFree(h->tau); hipFree(h->bound);
    hipFree(h->n_sorted); hipFree(h->exact); hipFree(h->flag); hipFree(h->scan_list);
    h->ws_q = 0;
    const int64_t qpad = round_up(Q, RAG_TILE);
    HIP_TRY(h, hipMalloc(&h->q32, (size_t)Q * h->dim * sizeof(float)));
    HIP_TRY(h, hipMalloc(&h->q16, (size_t)qpad * h->dim_pad * sizeof(half_t)));
    HIP_TRY(h, hipMalloc(&h->cand, (size_t)qpad * RAG_CAND_CAP * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc(&h->cnt, (size_t)qpad * sizeof(unsigned)));
    HIP_TRY(h, hipMalloc(&h->tau, (size_t)qpad * sizeof(float)));
    HIP_TRY(h, hipMalloc(&h->bound, (size_t)qpad * sizeof(float)));
    HIP_TRY(h, hipMalloc(&h->n_sorted, (size_t)qpad * sizeof(int)));
    HIP_TRY(h, hipMalloc(&h->exact, (size_t)qpad * RAG_CAND_CAP * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->flag, (size_t)qpad * sizeof(int)));
    HIP_TRY(h, hipMalloc(&h->scan_list, (size_t)qpad * sizeof(int)));
    if (!h->stats) HIP_TRY(h, hipMalloc(&h->stats, 8 * sizeof(int)));
    // zero fills go on the search's own stream: a null-stream hipMemset is not ordered against a non-blocking stream
    HIP_TRY(h, hipMemsetAsync(h->q16, 0, (size_t)qpad * h->dim_pad * sizeof(half_t), st));
    h->q16_dirty = 0;
    if (!h->q16b) {      // second pass (overflowed queries): one 256-query tile, allocated once per handle
        HIP_TRY(h, hipMalloc(&h->q16b, (size_t)RAG_TILE * h->dim_pad * sizeof(half_t)));
        HIP_TRY(h, hipMalloc(&h->candb, (size_t)RAG_TILE * RAG_CAND_CAP * sizeof(uint64_t)));
        HIP_TRY(h, hipMalloc(&h->cntb, RAG_TILE * sizeof(unsigned)));
        HIP_TRY(h, hipMalloc(&h->taub, RAG_TILE * sizeof(float)));
        HIP_TRY(h, hipMalloc(&h->boundb, RAG_TILE * sizeof(float)));
        HIP_TRY(h, hipMalloc(&h->n_sortedb, RAG_TILE * sizeof(int)));
        HIP_TRY(h, hipMalloc(&h->ovf_list, (RAG_TILE + 1) * sizeof(int)));
        HIP_TRY(h, hipMemsetAsync(h->ovf_list, 0, (RAG_TILE + 1) * sizeof(int), st));
        HIP_TRY(h, hipMemsetAsync(h->q16b, 0, (size_t)RAG_TILE * h->dim_pad * sizeof(half_t), st));
    }
    h->ws_q = Q;
    return RAG_OK;
}

// multiplier of the tile-order permutation p -> (p * a) mod T: close to the golden-ratio conjugate of T (every prefix of
// positions is then spread evenly over the table) and coprime to T (a bijection)
static int tile_multiplier(int T) {
    if (T <= 2) return 1;
    auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
    int a = (int)(0.6180339887498949 * T);
    if (a < 1) a = 1;
    while (a > 1 && gcd(a, T) != 1) --a;
    return a;
}

// Tenant tile lists (built once per rag_index_set_tenants_host): for every tenant the ascending list of 256-row tiles that
// hold at least one of its rows. A tenant-filtered search walks ONLY those tiles - a tenant stored contiguously (the usual
// export order) costs its own rows, not a pass over the table - and draws its threshold stages from them.
int dense_build_tenant_tiles(rag_ctx* h, const int32_t* tenants_host, int64_t n_rows) {
    hipFree(h->tenant_tiles);
    h->tenant_tiles = nullptr;
    h->tenant_span.clear();
    h->tenant_rows = 0;
    if (!tenants_host || n_rows == 0) return RAG_OK;
    std::unordered_map<int32_t, std::vector<int32_t>> lists;
    for (int64_t r = 0; r < n_rows; ++r) {
        auto& v = lists[tenants_host[r]];
        const int32_t t = (int32_t)(r / RAG_TILE);
        if (v.empty() || v.back() != t) v.push_back(t);
    }
    std::vector<int32_t> flat;
    for (auto& kv : lists) {
        h->tenant_span[kv.first] = {(int64_t)flat.size(), (int)kv.second.size()};
        flat.insert(flat.end(), kv.second.begin(), kv.second.end());
    }
    HIP_TRY(h, hipMalloc(&h->tenant_tiles, std::max<size_t>(1, flat.size()) * sizeof(int32_t)));
    HIP_TRY(h, hipMemcpy(h->tenant_tiles, flat.data(), flat.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    h->tenant_rows = n_rows;
    return RAG_OK;
}

int dense_free(rag_ctx* h) {
    hipFree(h->emb32); hipFree(h->emb16); hipFree(h->ids); hipFree(h->tenants); hipFree(h->bad_rows);
    h->emb32 = nullptr; h->emb16 = nullptr; h->ids = nullptr; h->tenants = nullptr; h->bad_rows = nullptr;
    hipFree(h->scan_scores); h->scan_scores = nullptr; h->scan_rows = 0;
    hipFree(h->tenant_tiles); h->tenant_tiles = nullptr; h->tenant_span.clear(); h->tenant_rows = 0;
    h->n_rows = h->n_rows_pad = 0;
    h->n_reserved = 0;
    h->index_loaded = false;
    return RAG_OK;
}

// emb32 must already be resident (h->emb32, n_rows rows). Builds the fp16 operand copy.
int dense_index_build(rag_ctx* h, const float* emb_dev, int64_t n_rows, hipStream_t st) {
    // pad to a multiple of 8 tiles so the XCD-aware block map needs no bounds logic on loads
    h->n_rows_pad = round_up(n_rows, (int64_t)RAG_TILE * 8);
    HIP_TRY(h, hipMalloc(&h->emb16, (size_t)h->n_rows_pad * h->dim_pad * sizeof(half_t)));
    HIP_TRY(h, hipMalloc(&h->bad_rows, sizeof(int)));
    HIP_TRY(h, hipMemsetAsync(h->bad_rows, 0, sizeof(int), st));
    if (h->n_rows_pad > n_rows)
        HIP_TRY(h, hipMemsetAsync(h->emb16 + (size_t)n_rows * h->dim_pad, 0,
                                  (size_t)(h->n_rows_pad - n_rows) * h->dim_pad * sizeof(half_t), st));
    if (n_rows > 0) {
        const int grid = (int)std::min<int64_t>((n_rows + 3) / 4, 256 * 16);
        hipLaunchKernelGGL(normalize_rows_kernel, dim3(grid), dim3(256), 0, st, emb_dev, h->emb16, n_rows, h->dim,
                           h->dim_pad, h->bad_rows);
        HIP_TRY(h, hipGetLastError());
    }
    return RAG_OK;
}

// chunked bulk load: fp16 operand rows for master rows [first_row, first_row + n_rows) (already copied into emb32)
int dense_index_normalize_range(rag_ctx* h, int64_t first_row, int64_t n_rows, hipStream_t st) {
    if (n_rows <= 0) return RAG_OK;
    const int grid = (int)std::min<int64_t>((n_rows + 3) / 4, 256 * 16);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(grid), dim3(256), 0, st, h->emb32 + (size_t)first_row * h->dim,
                       h->emb16 + (size_t)first_row * h->dim_pad, n_rows, h->dim, h->dim_pad, h->bad_rows);
    HIP_TRY(h, hipGetLastError());
    return RAG_OK;
}

static double fp16_pass_eps(int dim_pad) {
    // |S~ - S| for unit vectors: two fp16 roundings per product (rel 2^-11 each, + their product),
    // fp32 accumulation of dim_pad terms (any order, rel <= dim_pad * 2^-24 of sum |a_i b_i| <= 1),
    // fp32 normalisation of both rows (rel ~3 * 2^-24 each), small absolute slack for the 2^7 scaling's
    // residual subnormals.  Cauchy-Schwarz: sum |q_i c_i| <= 1.
    const double u16 = 1.0 / 2048.0, u32 = 1.0 / 16777216.0;
    return (2 * u16 + u16 * u16) * 1.01 + 2.0 * dim_pad * u32 + 8 * u32 + 2e-6;
}

int dense_search(rag_ctx* h, const float* q_dev, int Q, int k, int tenant, int64_t* ids_dev, int32_t* rows_dev,
                 double* scores_dev, hipStream_t st) {
    return dense_search_fused(h, q_dev, Q, k, tenant, ids_dev, rows_dev, scores_dev, st, nullptr);
}

// fz == nullptr: plain cosine top-k. Otherwise the linear fusion of rag_hybrid_linear_dev: the emitted / keyed / ranked score
// is alpha * cosine + beta * keyword + gamma * temporal (fz carries the per-(query,row) bias and the float64 inputs).
int dense_search_fused(rag_ctx* h, const float* q_dev, int Q, int k, int tenant, int64_t* ids_dev, int32_t* rows_dev,
                       double* scores_dev, hipStream_t st, const dense_fused* fz) {
    ARG_CHECK(h, h->index_loaded, "no index loaded");
    ARG_CHECK(h, Q > 0 && k > 0 && k <= RAG_MAX_K, "need Q>0 and 0<k<=256");
    ARG_CHECK(h, tenant < 0 || h->tenants != nullptr, "tenant filter requested but no tenants loaded");
    int rc = ensure_workspace(h, Q, st);
    if (rc) return rc;
    const int32_t* tenants = tenant >= 0 ? h->tenants : nullptr;
    // fused: |alpha| * (fp16-pass error of the cosine) + the float32 roundings of the emitted score alpha_f * S + bias:
    // bias = float(beta * kw + gamma * t) (2^-24 relative), alpha_f = float(alpha) (2^-24 |alpha| |S|), the fma's own rounding
    // (2^-24 of the result) - together <= 2^-23 * (|alpha| + |beta| * max|kw| + |gamma| * max|t|). The keyword score is
    // raw / max (in [0, 1] whenever the max is positive; the raw scores themselves, all <= 0, in the `else 1.0` case of
    // rag/retrieval.py:344, where 8 covers any realistic BM25 magnitude); max|t| is tracked by rag_index_set_temporal_host.
    const double f32_mag = fz ? fabs(fz->alpha) + 8.0 * fabs(fz->beta) + fabs(fz->gamma) * h->temporal_absmax : 0.0;
    // (round 4: the emitted keyword + recency term is formed in float32 from raw32 * qscale + gt: raw32, qscale and gt each carry one
    // 2^-24 rounding, the two fmas one each - 2^-21 of the magnitude covers them with a factor of two to spare)
    const double eps = fz ? fabs(fz->alpha) * fp16_pass_eps(h->dim_pad) + f32_mag / 2097152.0 + 1e-7 : fp16_pass_eps(h->dim_pad);
    const float two_eps = (float)(2.0 * eps * 1.0001 + 1e-7);        // float subtraction in the select kernel: round up
    const float* bias = fz ? fz->bias : nullptr;
    const int64_t bias_ld = fz ? fz->bias_ld : 0;
    const float* qscale = fz ? fz->qscale : nullptr;
    const float* gt = fz ? fz->gt : nullptr;
    const float alpha_f = fz ? (float)fz->alpha : 1.0f;
    const int qpad = (int)round_up(Q, RAG_TILE);
    const int n_qtiles = qpad / RAG_TILE;
    float* tau = h->tau;
    const int force_level = h->opt.force_level;

    // queries -> fp16 unit rows (pad rows of q16 stay zero from allocation time / previous larger batch)
    // only rows a previous, larger batch wrote can be non-zero: same-size batches (the agent's one query after another) clear nothing
    if (h->q16_dirty > Q)
        HIP_TRY(h, hipMemsetAsync(h->q16 + (size_t)Q * h->dim_pad, 0, (size_t)(h->q16_dirty - Q) * h->dim_pad * sizeof(half_t), st));
    h->q16_dirty = Q;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((Q + 3) / 4), dim3(256), 0, st, q_dev, h->q16, (int64_t)Q, h->dim,
                       h->dim_pad, (int*)nullptr);
    int* const ovf_count = h->ovf_list + RAG_TILE;
    hipLaunchKernelGGL(search_init_kernel, dim3((qpad + 255) / 256), dim3(256), 0, st, tau, h->bound, h->cnt, h->stats, ovf_count, qpad);
    const select_extra no_extra = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const bool second_pass = !h->opt.no_second_pass;
    const select_extra list_extra = {second_pass ? h->ovf_list : (int*)nullptr, ovf_count, nullptr, nullptr, nullptr, nullptr};

    bool& attr_set = h->attr_dense;
    if (!attr_set) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_emit_kernel<true, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DENSE_LDS_BYTES));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_emit_kernel<false, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DENSE_LDS_BYTES));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_emit_kernel<false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DENSE_LDS_BYTES_SMALLQ));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_emit_kernel<true, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DENSE_LDS_BYTES));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_emit_kernel<false, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DENSE_LDS_BYTES));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_emit_persist_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DENSE_LDS_BYTES));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_emit_persist_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DENSE_LDS_BYTES));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(select_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SELECT_LDS_BYTES));
        attr_set = true;
    }

    // ---- tile universe: all tiles in permuted order, or the tenant's tile list in permuted order --------------------
    const int total_tiles_all = (int)(round_up(h->n_rows, RAG_TILE) / RAG_TILE);
    const int32_t* tile_list = nullptr;
    int total_tiles = total_tiles_all;
    if (tenant >= 0) {
        ARG_CHECK(h, h->tenant_rows == h->n_rows, "tenant table is stale (rows were appended after rag_index_set_tenants_host)");
        const auto it = h->tenant_span.find(tenant);
        total_tiles = it == h->tenant_span.end() ? 0 : it->second.second;
        tile_list = it == h->tenant_span.end() ? nullptr : h->tenant_tiles + it->second.first;
    }
    const int n_tiles = total_tiles;
    // RAG_DENSE_LINEAR_ORDER=1 (diagnostic): r1's table order, to price the permutation on one box
    const int tile_mul = h->opt.dense_linear_order ? 1 : tile_multiplier(n_tiles), tile_mod = std::max(1, n_tiles);
    const bool smallq = Q <= 128 && !h->opt.no_smallq;

    // ---- stage schedule over tile POSITIONS: 8 tiles (2048 rows) scored densely, then ~8x growth each. The expected
    // emission of a stage is ~k x growth keys per query (tau = k-th best of everything seen so far), so the growth is
    // capped by k: it must stay well inside the 4096-entry buffer (r1 used 32x for small batches at any k; at k = 100
    // that sat at the edge of the buffer and a single query could fall into the exact scan).
    const int stage0_tiles = std::min(n_tiles, RAG_STAGE0_ROWS / RAG_TILE);
    const int growth_env = h->opt.stage_growth >= 2 ? h->opt.stage_growth : 0;       // diagnostic (rag_set_option)
    const int growth = growth_env ? growth_env
                                  : std::max(3, std::min(Q <= 64 ? 4 * RAG_STAGE_GROWTH : RAG_STAGE_GROWTH, (Q <= 64 ? 1024 : 1536) / k));
    int begin = 0, stage = 0;
    while (begin < total_tiles) {
        int end;
        if (stage == 0) end = stage0_tiles;
        else end = (int)std::min<int64_t>(total_tiles, (int64_t)begin * growth);
        // avoid a tiny trailing stage (a universe of 9 tiles is ONE dense stage of 9 tiles: the select below must be told so -
        // it was handed the nominal 2048 slots and lost the ninth tile's rows, found by tests/test_property_gpu.py)
        if (total_tiles - end < end / 4) end = total_tiles;
        const int n_rt = end - begin;
        const int grid = (int)round_up(n_rt, 8) * n_qtiles;
        if (stage > 0) {                      // the thresholded kernel only (stage 0 is 0.2% of the rows)
            const int prc = prof_begin(h, 0, st);
            if (prc) return prc;
        }
#define EMIT_ARGS(QP, NQT, QV, TAU, CNT, CAND, ACT, QMAP)                                                        \
    h->emb16, QP, h->dim_pad, begin_, n_rt_, NQT, (int)h->n_rows, QV, TAU, CNT, CAND, tenants, tenant, tile_list, tile_mul, tile_mod, \
        n_tiles, (const int*)(ACT), bias, bias_ld, alpha_f, (const int*)(QMAP), qscale, gt
        const int begin_ = begin, n_rt_ = n_rt;
        if (stage == 0 && fz)
            hipLaunchKernelGGL((dense_emit_kernel<true, false, true>), dim3(grid), dim3(512), DENSE_LDS_BYTES, st,
                               EMIT_ARGS(h->q16, n_qtiles, Q, tau, h->cnt, h->cand, nullptr, nullptr));
        else if (fz)
            hipLaunchKernelGGL((dense_emit_kernel<false, false, true>), dim3(grid), dim3(512), DENSE_LDS_BYTES, st,
                               EMIT_ARGS(h->q16, n_qtiles, Q, tau, h->cnt, h->cand, nullptr, nullptr));
        else if (stage == 0)
            hipLaunchKernelGGL((dense_emit_kernel<true, false>), dim3(grid), dim3(512), DENSE_LDS_BYTES, st,
                               EMIT_ARGS(h->q16, n_qtiles, Q, tau, h->cnt, h->cand, nullptr, nullptr));
        else if (smallq)
            hipLaunchKernelGGL((dense_emit_kernel<false, true>), dim3(grid), dim3(512), DENSE_LDS_BYTES_SMALLQ, st,
                               EMIT_ARGS(h->q16, n_qtiles, Q, tau, h->cnt, h->cand, nullptr, nullptr));
        else if (h->opt.dense_persist && grid > 256)       // experiment: one workgroup per CU walks the stage's tiles
            hipLaunchKernelGGL((dense_emit_persist_kernel<false>), dim3(256), dim3(512), DENSE_LDS_BYTES, st, grid,
                               EMIT_ARGS(h->q16, n_qtiles, Q, tau, h->cnt, h->cand, nullptr, nullptr));
        else
            hipLaunchKernelGGL((dense_emit_kernel<false, false>), dim3(grid), dim3(512), DENSE_LDS_BYTES, st,
                               EMIT_ARGS(h->q16, n_qtiles, Q, tau, h->cnt, h->cand, nullptr, nullptr));
        HIP_TRY(h, hipGetLastError());
        if (stage > 0) {
            const int prc = prof_end(h, 0, st);
            if (prc) return prc;
        }
        const bool last = end == total_tiles;
        hipLaunchKernelGGL(select_kernel, dim3((Q + 3) / 4), dim3(256), SELECT_LDS_BYTES, st, h->cand, h->cnt, tau, h->bound,
                           h->n_sorted, h->stats, Q, stage == 0 ? n_rt * RAG_TILE : 0, k, two_eps, last ? 1 : 0,
                           (const int*)nullptr, last ? list_extra : no_extra);
        HIP_TRY(h, hipGetLastError());
        begin = end;
        ++stage;
    }
    if (total_tiles == 0) {   // empty index / unknown tenant: nothing found
        hipLaunchKernelGGL(select_kernel, dim3((Q + 3) / 4), dim3(256), SELECT_LDS_BYTES, st, h->cand, h->cnt, tau, h->bound,
                           h->n_sorted, h->stats, Q, 0, k, two_eps, 1, (const int*)nullptr, no_extra);
    }

    // ---- second pass for overflowed queries (device-side early exit when there are none) ---------------------------
    if (total_tiles > 0 && second_pass) {
        hipLaunchKernelGGL(overflow_gather_kernel, dim3(RAG_TILE), dim3(256), 0, st, h->ovf_list, ovf_count, h->q16, tau, h->dim_pad,
                           h->q16b, h->taub, h->boundb, h->cntb);
        {
            const int begin_ = 0, n_rt_ = total_tiles;
            static const int n_cu = [] { int d = 0, n = 0; hipGetDevice(&d); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 0 ? n : 256; }();
            if (fz)
                hipLaunchKernelGGL((dense_emit_persist_kernel<true>), dim3(std::min(n_cu, total_tiles)), dim3(512), DENSE_LDS_BYTES, st, total_tiles,
                                   EMIT_ARGS(h->q16b, 1, RAG_TILE, h->taub, h->cntb, h->candb, ovf_count, h->ovf_list));
            else
                hipLaunchKernelGGL((dense_emit_persist_kernel<false>), dim3(std::min(n_cu, total_tiles)), dim3(512), DENSE_LDS_BYTES, st, total_tiles,
                                   EMIT_ARGS(h->q16b, 1, RAG_TILE, h->taub, h->cntb, h->candb, ovf_count, nullptr));
        }
        const select_extra scatter_extra = {nullptr, nullptr, h->ovf_list, h->cand, h->n_sorted, h->bound};
        hipLaunchKernelGGL(select_kernel, dim3(RAG_TILE / 4), dim3(256), SELECT_LDS_BYTES, st, h->candb, h->cntb, h->taub, h->boundb,
                           h->n_sortedb, h->stats, RAG_TILE, 0, k, two_eps, 1, (const int*)ovf_count, scatter_extra);
        HIP_TRY(h, hipGetLastError());
    }

    hipLaunchKernelGGL(rescore_kernel, dim3(4, Q), dim3(256), 0, st, q_dev, h->emb32, h->cand, h->n_sorted, h->exact, h->dim);
    if (fz)
        hipLaunchKernelGGL(linear_fuse_kernel, dim3(4, Q), dim3(256), 0, st, h->cand, h->n_sorted, h->exact, fz->raw, fz->n, fz->mx,
                           fz->temporal, fz->alpha, fz->beta, fz->gamma);
    int* const scan_list = h->scan_list;              // [ws_qpad]: queries that need the float64 scan, appended by finalize
    int* const scan_count = h->stats + 7;
    hipLaunchKernelGGL(finalize_kernel, dim3(Q), dim3(256), 0, st, h->cand, h->n_sorted, h->exact, h->bound, h->ids, h->id_base, k,
                       force_level, ids_dev, rows_dev, scores_dev, h->flag, h->stats, scan_list, scan_count);
    HIP_TRY(h, hipGetLastError());
    // exact scan for whatever is still unproven: rounds of SCAN_ROUND flagged queries, device-side early exit when none
    if (h->n_rows > 0) {
        const int window = SCAN_CHUNK - k;
        const int64_t rows_per_block = std::max<int64_t>(1, (h->n_rows + 1023) / 1024 + window - 1) / window * window;
        const int n_blocks = (int)((h->n_rows + rows_per_block - 1) / rows_per_block);
        // flagged queries per round: as many as a 512 MB partial-list scratch holds (k = 20 at 1M rows: every query of a
        // 1024-batch in ONE round = two idle launches per search), at least SCAN_ROUND
        const int round_q = std::min(Q, std::max(SCAN_ROUND, (int)std::min<size_t>(65535, ((size_t)512 << 20) / ((size_t)n_blocks * k * 12))));
        const size_t need = (size_t)round_q * n_blocks * k;
        if ((int64_t)need > h->scan_rows) {
            hipFree(h->scan_scores);
            h->scan_scores = nullptr;
            h->scan_rows = 0;
            HIP_TRY(h, hipMalloc(&h->scan_scores, need * 12));
            h->scan_rows = (int64_t)need;
        }
        uint64_t* pk = reinterpret_cast<uint64_t*>(h->scan_scores);
        uint32_t* pr = reinterpret_cast<uint32_t*>(pk + h->scan_rows);
        for (int f0 = 0; f0 < Q; f0 += round_q) {
            hipLaunchKernelGGL(scan_chunk_kernel, dim3(n_blocks), dim3(256), 0, st, q_dev, h->emb32, tenants, tenant, h->n_rows,
                               rows_per_block, h->dim, k, scan_list, scan_count, f0, round_q, pk, pr, fz ? fz->raw : (const double*)nullptr,
                               fz ? fz->n : (int64_t)0, fz ? fz->mx : (const double*)nullptr, fz ? fz->temporal : (const double*)nullptr,
                               fz ? fz->alpha : 0.0, fz ? fz->beta : 0.0, fz ? fz->gamma : 0.0);
            hipLaunchKernelGGL(scan_merge_kernel, dim3(std::min(Q - f0, round_q)), dim3(256), 0, st, pk, pr, n_blocks, k, h->ids,
                               h->id_base, scan_list, scan_count, f0, h->flag, ids_dev, rows_dev, scores_dev, h->stats);
        }
        HIP_TRY(h, hipGetLastError());
    }
    h->last_q = Q;
    h->last_k = k;
    h->last_stages = stage;
    h->last_shortlist = k;
    h->last_eps = eps;
    h->last_stats_valid = true;
    return RAG_OK;
}

// bias / max / components of the linear fusion (called by rag_hybrid_linear_dev around dense_search_fused)
int linear_prepare(rag_ctx* h, const unsigned long long* max_key, int Q, int64_t n, const double* temporal, double beta, double gamma, double* mx,
                   float* qscale, float* gt, int64_t ld, hipStream_t st) {
    hipLaunchKernelGGL(linear_scale_kernel, dim3((Q + 255) / 256), dim3(256), 0, st, max_key, Q, beta, mx, qscale);
    if (gt != nullptr) hipLaunchKernelGGL(linear_gt_kernel, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, st, temporal, n, ld, gamma, gt);
    HIP_TRY(h, hipGetLastError());
    return RAG_OK;
}

int linear_components(rag_ctx* h, const float* q_dev, const int32_t* rows_dev, int Q, int k, const dense_fused* fz, double* sem_out,
                      double* kw_out, double* tmp_out, hipStream_t st) {
    hipLaunchKernelGGL(linear_components_kernel, dim3((unsigned)(((int64_t)Q * k + 3) / 4)), dim3(256), 0, st, q_dev, h->emb32, rows_dev, Q, k,
                       h->dim, fz->raw, fz->n, fz->mx, fz->temporal, sem_out, kw_out, tmp_out);
    HIP_TRY(h, hipGetLastError());
    return RAG_OK;
}

// ------------------------------------------------------------------------------------------------
// merge of per-shard partial lists (multi-GPU exchange step): [L][Q][k] -> [Q][k], score desc, id asc
// ------------------------------------------------------------------------------------------------
// normalize != 0: the merged scores are divided by the largest one when it is positive (else by 1.0) - rag/retrieval.py:343-345
// with the GLOBAL maximum, for the shards' raw BM25 lists (the head of the merged list is that maximum).
__global__ __launch_bounds__(256) void merge_topk_kernel(const int64_t* __restrict__ ids, const double* __restrict__ scores,
                                                          int n_lists, int64_t list_stride, int Q, int k,
                                                          int64_t* __restrict__ ids_out, double* __restrict__ scores_out, int normalize) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double wmax[4];
    const int total = n_lists * k;
    double* sc = reinterpret_cast<double*>(smem);
    int64_t* id = reinterpret_cast<int64_t*>(smem + (size_t)total * 8);
    const int q = blockIdx.x, tid = threadIdx.x;
    double mx = -INFINITY;
    for (int i = tid; i < total; i += 256) {
        const int l = i / k, j = i % k;
        sc[i] = scores[(size_t)l * list_stride + (size_t)q * k + j];
        id[i] = ids[(size_t)l * list_stride + (size_t)q * k + j];
        if (id[i] >= 0) mx = fmax(mx, sc[i]);
    }
    for (int i = tid; i < k; i += 256) {
        ids_out[(size_t)q * k + i] = -1;
        scores_out[(size_t)q * k + i] = 0.0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) wmax[tid >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
    const double div = normalize && mx > 0.0 ? mx : 1.0;
    for (int i = tid; i < total; i += 256) {
        const int64_t me = id[i];
        if (me < 0) continue;
        const double e = sc[i];
        int rank = 0;
        for (int u = 0; u < total && rank < k; ++u) {
            const int64_t o = id[u];
            rank += (o >= 0) && ((sc[u] > e) || (sc[u] == e && o < me));
        }
        if (rank < k) {
            ids_out[(size_t)q * k + rank] = me;
            scores_out[(size_t)q * k + rank] = normalize ? e / div : e;
        }
    }
}

int merge_topk(rag_ctx* h, const int64_t* ids, const double* scores, int n_lists, int64_t list_stride, int Q, int k,
               int64_t* ids_out, double* scores_out, hipStream_t st, int normalize) {
    ARG_CHECK(h, n_lists > 0 && Q > 0 && k > 0, "merge: sizes must be positive");
    const size_t lds = (size_t)n_lists * k * 16;
    // the kernel's static LDS (per-wave maxima of the normalisation, 32 B) comes on top of the dynamic entries: 64 KiB in all
    ARG_CHECK(h, lds + 64 <= 64 * 1024, "merge: n_lists*k too large (max 4092 entries)");
    ARG_CHECK(h, list_stride >= (int64_t)Q * k, "merge: list_stride < Q*k");
    hipLaunchKernelGGL(merge_topk_kernel, dim3(Q), dim3(256), lds, st, ids, scores, n_lists, list_stride, Q, k, ids_out,
                       scores_out, normalize);
    HIP_TRY(h, hipGetLastError());
    return RAG_OK;
}

// ------------------------------------------------------------------------------------------------
// K8: small pairwise cosine in float64. One wave per output element block.
// ------------------------------------------------------------------------------------------------
template <class T>     // float (embeddings as stored) or double (the agent's List[float]: no rounding before the float64 arithmetic)
__global__ __launch_bounds__(256) void pairwise_cosine_kernel(const T* __restrict__ a, int m, const T* __restrict__ b,
                                                               int n, int dim, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= (int64_t)m * n) return;
    const int i = (int)(pair / n), j = (int)(pair % n);
    const T* x = a + (size_t)i * dim;
    const T* y = b + (size_t)j * dim;
    double dot = 0.0, nx = 0.0, ny = 0.0;
    for (int t = lane; t < dim; t += 64) {
        const double u = x[t], v = y[t];
        dot += u * v;
        nx += u * u;
        ny += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dot += __shfl_xor(dot, o);
        nx += __shfl_xor(nx, o);
        ny += __shfl_xor(ny, o);
    }
    if (lane == 0) {
        const double m1 = sqrt(nx), m2 = sqrt(ny);
        out[pair] = (m1 == 0.0 || m2 == 0.0) ? 0.0 : dot / (m1 * m2);
    }
}

int pairwise_cosine(rag_ctx* h, const float* a_dev, int m, const float* b_dev, int n, int dim, double* out_dev,
                    hipStream_t st) {
    const int64_t pairs = (int64_t)m * n;
    if (pairs == 0) return RAG_OK;
    hipLaunchKernelGGL(pairwise_cosine_kernel<float>, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, st, a_dev, m, b_dev, n, dim,
                       out_dev);
    HIP_TRY(h, hipGetLastError());
    return RAG_OK;
}

int pairwise_cosine_f64(rag_ctx* h, const double* a_dev, int m, const double* b_dev, int n, int dim, double* out_dev, hipStream_t st) {
    const int64_t pairs = (int64_t)m * n;
    if (pairs == 0) return RAG_OK;
    hipLaunchKernelGGL(pairwise_cosine_kernel<double>, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, st, a_dev, m, b_dev, n, dim,
                       out_dev);
    HIP_TRY(h, hipGetLastError());
    return RAG_OK;
}

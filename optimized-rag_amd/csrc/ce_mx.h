// Cross-encoder GEMMs, round-4 operand scheme: hi16 + lo8 ("MX" form), 2 MFMA units per product instead of 3, 3 bytes per
// element instead of 4. Included by cross_encoder.hip (product) and tools/ce_mx_probe.hip (diagnostic harness).
//
// Numerics (tools/ce_numerics_sim.py): x = hi + lo, hi = fp16(x). rounds 1-3 kept lo as a second fp16 and
// spent three fp16 MFMAs per product (hi.hi + lo.hi + hi.lo). The two correction products only have to be good to a few
// per cent, so here
//   * lo8 = e5m2(lo * 2^11) (one byte, round to nearest),
//   * hi8 = e5m2(hi) = the TOP BYTE of (hi's fp16 pattern + 0x80): not stored, made in registers from the fp16 fragment the wave holds
//     anyway - one 32-bit add per dword (round to nearest on the magnitude of both halves) and one v_perm_b32 per four elements.
//     (-DMX_HI8_TRUNCATE keeps the first form of round 4: the top byte as it stands = rounding toward zero, with lo8 scaled up by
//     the mean loss 1 / 0.915. One instruction per dword less, 3.8 % faster main loop in tools/ce_mx_probe.hip - and 1.6e-3
//     instead of 1.0e-3 of logit error on the GPU, 1.5e-3 instead of 8.8e-4 in the simulation: the rounded form ships.)
//   * both correction products of a 32-element K range run as ONE block-scaled bf8 MFMA of 64 K-slots:
//     A = [lo8 | hi8], B = [hi8 | lo8], scale 2^-11 (v_mfma_scale_f32_32x32x64_f8f6f4: twice the fp16 rate per K-slot).
// Per 32x32 block and 32-deep K-step: 2 x v_mfma_f32_32x32x16_f16 (64 cycles) + 1 x scaled bf8 (64 cycles) = 128 cycles against
// 192 for the split-fp16 form, and 96 B per operand row against 128 B.
//
// Layout ("image" layout; operands are stored in HBM exactly as the LDS stage holds them, so every LDS-DMA piece is 1 KiB of
// consecutive bytes = 8 whole lines, and a fragment read is conflict-free without a swizzle):
//   a ROW TILE is 384 rows for weights (MX_TM: all features a workgroup owns), 128 rows for activations (MX_TN tokens);
//   tensor = [row tile][K-step s = k / 32][plane c = 0..5][row in tile][16 B]
//   planes 0..3: hi16, plane 2*j + h holds elements k%32 = 16*j + 8*(i >> 2) + 4*h + (i & 3) (i = 0..7) as 8 halfs - the fragment
//                of lane half h for the j-th v_mfma_f32_32x32x16_f16 of the step. (A K permutation inside each 16-element chunk,
//                the same for both operands: it makes lane half h of a 32 x 32 ACCUMULATOR - features 8q + 4h + r of a block - the
//                owner of whole 16-B fragments of the next GEMM's operand, so epilogues store and re-read the image without a
//                lane exchange.)
//   planes 4..5: lo8, plane 4 + h byte p = (i >> 2) * 8 + j * 4 + (i & 3) holds the element of fragment (j, h) position i - the order
//                in which v_perm_b32 leaves the top bytes of the lane's two fp16 fragments.
// A stage (one K-step of a 384 x 128 tile) is 36 KiB of weights + 12 KiB of tokens; three stages = 144 KiB.
#pragma once
#include "common.h"

typedef int mx_v8i __attribute__((ext_vector_type(8)));
typedef int mx_v4i __attribute__((ext_vector_type(4)));

#define MX_TM 384                                 // weight rows (output features) per workgroup tile
#define MX_TN 128                                 // token rows per workgroup tile
#define MX_A_PLANE (MX_TM * 16)                   // 6144 B
#define MX_B_PLANE (MX_TN * 16)                   // 2048 B
#define MX_A_STAGE (6 * MX_A_PLANE)               // 36864 B
#define MX_B_STAGE (6 * MX_B_PLANE)               // 12288 B
#define MX_STAGE (MX_A_STAGE + MX_B_STAGE)        // 49152 B
#define MX_STAGES 3
#define MX_LDS (MX_STAGES * MX_STAGE)             // 147456 B
#ifndef MX_HI8_TRUNCATE
#define MX_LO_SCALE 2048.0f                       // lo8 = e5m2(lo * 2^11): |lo| <= 2^-11 |x|, so lo8 spans x's own range
#else
#define MX_LO_SCALE (2048.0f * 1.0928961748633879f)   // truncating hi8: the mean loss 1 / 0.915 goes into lo8
#endif
#define MX_SCALE_A 116                            // E8M0 of 2^-11: the correction MFMA's block scale (B side: 127 = 2^0)
#define MX_SCALE_B 127

// byte offset of element (row r of the tile, column k % 32 = e) inside one K-step image with `rows` rows per plane
// element e of a K-step -> MFMA j = e >> 4, lane half h = (e >> 2) & 1, fragment position i = 4 * ((e >> 3) & 1) + (e & 3)
__host__ __device__ __forceinline__ int mx_hi_off(int rows, int r, int e) {
    const int j = e >> 4, h = (e >> 2) & 1, i = ((e >> 3) & 1) * 4 + (e & 3);
    return (j * 2 + h) * rows * 16 + r * 16 + i * 2;
}
__host__ __device__ __forceinline__ int mx_lo_off(int rows, int r, int e) {
    const int j = e >> 4, h = (e >> 2) & 1, i = ((e >> 3) & 1) * 4 + (e & 3);
    return (4 + h) * rows * 16 + r * 16 + (i >> 2) * 8 + j * 4 + (i & 3);
}

// fp16 bit pattern -> e5m2 byte, round to nearest even on the magnitude (a carry into the exponent is the right result)
__host__ __device__ __forceinline__ unsigned mx_e5m2_rn(unsigned h16) {
    const unsigned mag = h16 & 0x7FFFu;
    unsigned r = (mag + 0x7Fu + ((mag >> 8) & 1u)) >> 8;
    if (r > 0x7Bu) r = 0x7Bu;                       // clamp to the largest finite e5m2 (57344): never Inf / NaN
    return r | ((h16 >> 8) & 0x80u);
}

// x -> hi (fp16) and lo8; the value a consumer reconstructs is hi + e5m2_decode(lo8) / MX_LO_SCALE
__device__ __forceinline__ void mx_split(float x, half_t& hi, unsigned& lo8) {
    hi = (half_t)x;
    const half_t l = (half_t)((x - (float)hi) * MX_LO_SCALE);
    lo8 = mx_e5m2_rn((unsigned)__builtin_bit_cast(unsigned short, l));
}
__device__ __forceinline__ float mx_lo_decode(unsigned lo8) {
    return (float)__builtin_bit_cast(half_t, (unsigned short)(lo8 << 8)) * (1.0f / MX_LO_SCALE);
}

// e5m2 of four halfs held in two dwords (lo dword first): + 0x80 on each 16-bit pattern (round half up on the magnitude; a carry
// into the exponent is the right result), then the top bytes. The two halves of a dword are rounded by ONE 32-bit add: the low
// half carries into the high one only from 0xFF80 up, which is a NaN pattern (a packed v_pk_add_u16 does the same per half and,
// like every packed op beside MFMAs, slowed the loop by 10 %). |x| >= 61440 rounds to Inf: activations never get there (65504
// overflows fp16 itself).
#ifndef MX_HI8_TRUNCATE
__device__ __forceinline__ unsigned mx_round8(int d) { return (unsigned)d + 0x00800080u; }
#else
__device__ __forceinline__ unsigned mx_round8(int d) { return (unsigned)d; }
#endif
__device__ __forceinline__ int mx_top4(int d0, int d1) { return (int)__builtin_amdgcn_perm(mx_round8(d1), mx_round8(d0), 0x07050301u); }

// hi8 of a lane's two fp16 fragments of one K-step, in the byte order of the lo8 planes
__device__ __forceinline__ mx_v4i mx_hi8(half8 f0, half8 f1) {
    const mx_v4i a = __builtin_bit_cast(mx_v4i, f0), b = __builtin_bit_cast(mx_v4i, f1);
    return (mx_v4i){mx_top4(a[0], a[1]), mx_top4(b[0], b[1]), mx_top4(a[2], a[3]), mx_top4(b[2], b[3])};
}
__device__ __forceinline__ mx_v8i mx_cat(mx_v4i lo, mx_v4i hi) { return (mx_v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }

// one 32x32 block, one 32-deep K-step: acc += W.X^T with W rows on the MFMA rows (SWAP = false) or on its columns (SWAP = true)
template <bool SWAP>
__device__ __forceinline__ void mx_block(f32x16& acc, half8 wh0, half8 wh1, mx_v8i w8, half8 xh0, half8 xh1, mx_v8i x8) {
    if (!SWAP) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, xh0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, xh1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8, x8, acc, 1, 1, 0, MX_SCALE_A, 0, MX_SCALE_B);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh0, wh0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh1, wh1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x8, w8, acc, 1, 1, 0, MX_SCALE_B, 0, MX_SCALE_A);
    }
}

__device__ __forceinline__ void mx_bdma(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, char* lds_uniform) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_uniform, 16, voff, soff, 0, 0);
}

#define MX_BAR __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
// The three things a K-step does go through these names so that tools/ce_mx_probe.hip can time the loop with one of them taken out
// (it defines them before including this file; such builds compute wrong results and exist only there). Nothing else overrides them.
#ifndef MX_ISSUE_W
#define MX_ISSUE_W(...) mx_bdma(__VA_ARGS__)                  // one 1-KiB piece of the weight image
#endif
#ifndef MX_ISSUE_X
#define MX_ISSUE_X(...) mx_bdma(__VA_ARGS__)                  // one 1-KiB piece of the token image
#endif
#ifndef MX_BLOCK
#define MX_BLOCK(SWAP, ...) mx_block<SWAP>(__VA_ARGS__)       // the three MFMAs of one 32 x 32 block
#endif
#ifndef MX_W_HI8
#define MX_W_HI8(h0, h1, lo) mx_hi8(h0, h1)                   // hi8 of a weight fragment pair (4 v_perm_b32)
#endif
#ifndef MX_READ_A_IF
#define MX_READ_A_IF(b) true                                  // weight fragments of block b are read from the stage (always, in the product)
#endif
#ifndef MX_STEP_WAIT
#define MX_STEP_WAIT "s_waitcnt vmcnt(6)"                     // own pieces of the step have landed (six per wave and step in flight)
#endif

// The main loop shared by every MX GEMM: a persistent 512-thread workgroup owns tiles of 384 weight rows x 128 tokens and walks K
// in 32-deep steps through a three-stage LDS ring filled by LDS-DMA two steps ahead (48 pieces of 1 KiB per step, six per wave:
// one behind each block's MFMAs), ONE barrier per step (RAW: counted vmcnt(6); WAR: a stage is refilled only after the barrier
// that follows its last read). 8 waves = 2 feature halves (192 rows = 6 blocks of 32) x 4 token groups (32 = 1 block): 96
// accumulator registers. The DMA stream is continuous across the tiles of a workgroup: the last two steps of a tile issue the
// first two steps of the next one.
//
// Work order: tile w -> (token tile w / n_ft, feature tile w % n_ft), feature tile fastest; workgroup b takes the token tiles
// t = b & 7 (mod 8) so that the workgroups sharing a token tile sit on one XCD (speed only).
struct mx_tile_iter {
    int n_ft, n_tt, xcd, n_slots, work;
    __device__ __forceinline__ int tt() const { return (work / n_ft) * 8 + xcd; }
    __device__ __forceinline__ int ft() const { return work % n_ft; }
    __device__ __forceinline__ bool valid() const { return tt() < n_tt; }
};

// what the K loop of one tile needs from the persistent loop around it
struct mx_stream {
    __amdgpu_buffer_rsrc_t w_rs, x_cur, x_nxt;
    unsigned w_cur, w_nxt, voff;
    int wid, ring, nk, a_off, b_off;
    bool has_next;
    char* smem;
};

// piece k (0..5) of this wave in a step: image piece p = wid + 8k; p < 36: weights, else tokens
#define MX_PIECE(S, k, st_, ws_, xrs_, ks_)                                                                               \
    {                                                                                                                     \
        const int p_ = (S).wid + 8 * (k);                                                                                 \
        if (p_ < 36) { MX_ISSUE_W((S).w_rs, (S).voff, (ws_) + (ks_) * MX_A_STAGE + p_ * 1024, (st_) + p_ * 1024); }         \
        else { MX_ISSUE_X(xrs_, (S).voff, (ks_) * MX_B_STAGE + (p_ - 36) * 1024, (st_) + p_ * 1024); }                     \
    }

template <bool SWAP>
__device__ __forceinline__ void mx_ksteps(f32x16 (&acc)[6], const mx_stream& S, bool first) {
    const int nk = S.nk;
    for (int t = 0; t < nk; ++t) {
        // own pieces of step t have landed once at most the six of step t+1 are outstanding (a continued tile's steps 0 and 1
        // were waited for before the previous epilogue's stores)
        if (first || t > 0) asm volatile(MX_STEP_WAIT ::: "memory");
        MX_BAR
        const char* st = S.smem + ((S.ring + t) % MX_STAGES) * MX_STAGE;
        // what this step owes the ring: step t+2 of this tile, or step t+2-nk of the next one (a harmless re-load at the very end)
        const int u2 = t + 2;
        const bool nx_ = u2 >= nk && S.has_next;
        const unsigned ks_ = (unsigned)(u2 < nk ? u2 : (S.has_next ? u2 - nk : nk - 1));
        char* const dst = S.smem + ((S.ring + u2) % MX_STAGES) * MX_STAGE;
        const unsigned wsrc = nx_ ? S.w_nxt : S.w_cur;
        const __amdgpu_buffer_rsrc_t xsrc = nx_ ? S.x_nxt : S.x_cur;
        const half8 xh0 = *reinterpret_cast<const half8*>(st + S.b_off);
        const half8 xh1 = *reinterpret_cast<const half8*>(st + S.b_off + 2 * MX_B_PLANE);
        const mx_v4i xl = *reinterpret_cast<const mx_v4i*>(st + S.b_off + 4 * MX_B_PLANE);
        half8 wh0[2], wh1[2];
        mx_v4i wl[2];
#define MX_READ_A(b, s)                                                                                               \
        wh0[s] = *reinterpret_cast<const half8*>(st + S.a_off + (b) * 512);                                            \
        wh1[s] = *reinterpret_cast<const half8*>(st + S.a_off + (b) * 512 + 2 * MX_A_PLANE);                           \
        wl[s] = *reinterpret_cast<const mx_v4i*>(st + S.a_off + (b) * 512 + 4 * MX_A_PLANE);
        MX_READ_A(0, 0)
        MX_READ_A(1, 1)
        // correction operands: the MFMA's A side is [lo8 | hi8], its B side [hi8 | lo8]
        const mx_v8i x8 = SWAP ? mx_cat(xl, mx_hi8(xh0, xh1)) : mx_cat(mx_hi8(xh0, xh1), xl);
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const mx_v4i w8h = MX_W_HI8(wh0[b & 1], wh1[b & 1], wl[b & 1]);
            const mx_v8i w8 = SWAP ? mx_cat(w8h, wl[b & 1]) : mx_cat(wl[b & 1], w8h);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            MX_BLOCK(SWAP, acc[b], wh0[b & 1], wh1[b & 1], w8, xh0, xh1, x8);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (b + 2 < 6 && MX_READ_A_IF(b + 2)) { MX_READ_A(b + 2, b & 1) }
            MX_PIECE(S, b, dst, wsrc, xsrc, ks_)
        }
#undef MX_READ_A
    }
}

// EPI functor: called once per finished tile with the 6 accumulator blocks of the wave.
//   acc[b][r]: SWAP = false: feature ft*384 + wm*192 + b*32 + (r&3) + 8*(r>>2) + 4*(lane>>5), token tt*128 + wn*32 + (lane&31)
//              SWAP = true : feature ft*384 + wm*192 + b*32 + (lane&31), token tt*128 + wn*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)
// The functor may use `scratch` (MX_LDS .. MX_KERNEL_LDS: 2 KiB of exchange space, then MX_PARAM_OFF.. for what prepare() staged) and must
// not touch the ring. prepare(scratch) runs once per workgroup before the first tile. swap_for(ft) picks the operand order per tile.
template <class EPI>
__device__ __forceinline__ void mx_gemm_loop(const char* __restrict__ W, const char* __restrict__ X, int nk, int n_ft,
                                             int n_tt, char* smem, EPI& epi) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    mx_tile_iter it{n_ft, n_tt, (int)(blockIdx.x & 7), (int)(gridDim.x >> 3), (int)(blockIdx.x >> 3)};
    if (!it.valid()) return;
    const unsigned a_tile_b = (unsigned)nk * MX_A_STAGE, b_tile_b = (unsigned)nk * MX_B_STAGE;
    mx_stream S;
    S.voff = (unsigned)lane * 16u;
    S.wid = wid;
    S.nk = nk;
    S.smem = smem;
    S.ring = 0;                                              // stage of step 0 of the current tile
    S.w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(W), 0, (int)((size_t)n_ft * a_tile_b), 0x00020000);
    // activations can exceed 4 GiB: the descriptor is rebased per token tile
    S.x_cur = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(X + (size_t)it.tt() * b_tile_b), 0, (int)b_tile_b, 0x00020000);
    S.x_nxt = S.x_cur;
    S.w_cur = (unsigned)it.ft() * a_tile_b;
    S.w_nxt = S.w_cur;
    S.has_next = false;
    S.a_off = (lane >> 5) * MX_A_PLANE + (wm * 192 + (lane & 31)) * 16;           // + plane pair j*2*PLANE + block*512
    S.b_off = MX_A_STAGE + (lane >> 5) * MX_B_PLANE + (wn * 32 + (lane & 31)) * 16;
#pragma unroll
    for (int u = 0; u < 2; ++u) {                            // prologue: steps 0 and 1 of the first tile
        char* st_ = smem + u * MX_STAGE;
#pragma unroll
        for (int k = 0; k < 6; ++k) MX_PIECE(S, k, st_, S.w_cur, S.x_cur, (unsigned)u)
    }
    // the epilogue's per-feature parameters (bias, LayerNorm weights) go to LDS once per workgroup, under the prologue's DMA latency:
    // a tile's epilogue then reads them with ds_read_b128 instead of 24-72 global float4 loads per thread
    epi.prepare(smem + MX_LDS);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MX_BAR
    for (bool first = true;; first = false) {
        {
            mx_tile_iter nx = it;
            nx.work += it.n_slots;
            S.has_next = nx.valid();
            if (S.has_next) {
                S.w_nxt = (unsigned)nx.ft() * a_tile_b;
                S.x_nxt = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(X + (size_t)nx.tt() * b_tile_b), 0, (int)b_tile_b, 0x00020000);
            }
        }
        const bool swap = epi.swap_for(it.ft());
        f32x16 acc[6];
#pragma unroll
        for (int b = 0; b < 6; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        if (swap) mx_ksteps<true>(acc, S, first);
        else mx_ksteps<false>(acc, S, first);
        // every wave is past its reads of the last step after this barrier; steps 0 and 1 of the next tile are in flight into the
        // other two stages: retire them here, where no store is outstanding yet
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MX_BAR
        epi(acc, it.tt(), it.ft(), swap, smem + MX_LDS);
        if (!S.has_next) break;
        it.work += it.n_slots;
        S.w_cur = S.w_nxt;
        S.x_cur = S.x_nxt;
        S.ring = (S.ring + nk) % MX_STAGES;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // clamped tail re-loads: retire them before exit
}

// =====================================================================================================================
// Epilogues and kernels of the MX forward (hidden = 384: one feature tile IS the hidden state, so LayerNorm stays on the CU)
// =====================================================================================================================
#define MX_PARAM_OFF 2048                          // scratch bytes [0, 2048): LayerNorm row statistics; parameters from here on (<= 6 KiB)
__device__ __forceinline__ void mx_stage_params(float* dst, const float* __restrict__ src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
typedef unsigned mx_u2 __attribute__((ext_vector_type(2)));
typedef unsigned mx_u4 __attribute__((ext_vector_type(4)));

// byte address of the K-step image holding (token row m, feature k) of an activation tensor with `nk` K-steps
__device__ __forceinline__ size_t mx_img_base(int64_t m, int k, int nk) { return ((size_t)(m >> 7) * nk + (k >> 5)) * MX_B_STAGE; }

// four consecutive features of one token -> 4 halfs (hi) + 4 bytes (lo8)
typedef short mx_s2 __attribute__((ext_vector_type(2)));
typedef float mx_f2 __attribute__((ext_vector_type(2)));
#ifndef MX_HI8_TRUNCATE
// Two vector instructions per element instead of four, same bits (tools/cvt_probe.hip: 1M pairs incl. ties, saturation, denormals):
// v - hi as ONE v_fma_mix_f32 (fp16 operand, times a -1.0 the compiler cannot fold into a subtraction that needs a conversion first),
// and the 2^11 scaling inside v_cvt_scalef32_pk_bf8_f32 (scale 2^-11 = the block scale of the value it encodes).
__device__ __forceinline__ void mx_split4(float v0, float v1, float v2, float v3, mx_u2& hi, unsigned& lo) {
    const half4 h = {(half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3};
    hi = __builtin_bit_cast(mx_u2, h);
    float neg1;
    asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(neg1));
    const float l0 = __builtin_fmaf((float)h[0], neg1, v0), l1 = __builtin_fmaf((float)h[1], neg1, v1);
    const float l2 = __builtin_fmaf((float)h[2], neg1, v2), l3 = __builtin_fmaf((float)h[3], neg1, v3);
    mx_s2 r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32((mx_s2){0, 0}, l0, l1, 1.0f / MX_LO_SCALE, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(r, l2, l3, 1.0f / MX_LO_SCALE, true);
    lo = __builtin_bit_cast(unsigned, r);
}
// hi + lo8 / 2^11 for four elements: the byte pairs convert two at a time, the power-of-two scale rides in the fma (exact product:
// one rounding, the bits of the two-step form)
__device__ __forceinline__ void mx_join4(half4 hi, unsigned lo, float (&out)[4]) {
    const mx_f2 a = __builtin_amdgcn_cvt_pk_f32_bf8((int)lo, false), b = __builtin_amdgcn_cvt_pk_f32_bf8((int)lo, true);
    out[0] = __builtin_fmaf(a[0], 1.0f / MX_LO_SCALE, (float)hi[0]);
    out[1] = __builtin_fmaf(a[1], 1.0f / MX_LO_SCALE, (float)hi[1]);
    out[2] = __builtin_fmaf(b[0], 1.0f / MX_LO_SCALE, (float)hi[2]);
    out[3] = __builtin_fmaf(b[1], 1.0f / MX_LO_SCALE, (float)hi[3]);
}
#else
__device__ __forceinline__ void mx_split4(float v0, float v1, float v2, float v3, mx_u2& hi, unsigned& lo) {
    const half4 h = {(half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3};
    hi = __builtin_bit_cast(mx_u2, h);
    int r = __builtin_amdgcn_cvt_pk_bf8_f32((v0 - (float)h[0]) * MX_LO_SCALE, (v1 - (float)h[1]) * MX_LO_SCALE, 0, false);
    lo = (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32((v2 - (float)h[2]) * MX_LO_SCALE, (v3 - (float)h[3]) * MX_LO_SCALE, r, true);
}
__device__ __forceinline__ void mx_join4(half4 hi, unsigned lo, float (&out)[4]) {
    out[0] = (float)hi[0] + mx_lo_decode(lo & 0xFF);
    out[1] = (float)hi[1] + mx_lo_decode((lo >> 8) & 0xFF);
    out[2] = (float)hi[2] + mx_lo_decode((lo >> 16) & 0xFF);
    out[3] = (float)hi[3] + mx_lo_decode(lo >> 24);
}
#endif
__device__ __forceinline__ float mx_join(half_t hi, unsigned lo8) { return (float)hi + mx_lo_decode(lo8); }
// v - hi for four elements as v_fma_mix_f32 (see mx_split4): the fp16 operand needs no conversion instruction of its own
__device__ __forceinline__ half4 mx_resid4(half4 h, float v0, float v1, float v2, float v3) {
    float neg1;
    asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(neg1));
    return (half4){(half_t)__builtin_fmaf((float)h[0], neg1, v0), (half_t)__builtin_fmaf((float)h[1], neg1, v1),
                   (half_t)__builtin_fmaf((float)h[2], neg1, v2), (half_t)__builtin_fmaf((float)h[3], neg1, v3)};
}

#ifndef MX_STORE16
#define MX_STORE16(p, v) *reinterpret_cast<mx_u4*>(p) = (v)
#endif
// Stores one 32-feature block of a NON-swapped accumulator tile (the values in v[16], already biased / activated) into the image
// layout of the next GEMM's token operand: row `trow` of the tile image `img` (K-step = this block's 32 features).
// v[4q + r] = feature 8q + 4hh + r of the block = position 4 (q & 1) + r of fragment (j = q >> 1, h = hh): the lane owns the whole
// 16 B of planes hh, 2 + hh and 4 + hh of its row - three 16-B stores, no exchange with the other lane half.
__device__ __forceinline__ void mx_store_block(char* img, int trow, int hh, const float (&v)[16]) {
    mx_u2 hi[4];
    unsigned lo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) mx_split4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3], hi[q], lo[q]);
    MX_STORE16(img + hh * MX_B_PLANE + trow * 16, ((mx_u4){hi[0][0], hi[0][1], hi[1][0], hi[1][1]}));
    MX_STORE16(img + (2 + hh) * MX_B_PLANE + trow * 16, ((mx_u4){hi[2][0], hi[2][1], hi[3][0], hi[3][1]}));
    // lo byte p = (i >> 2) * 8 + j * 4 + (i & 3): dword (q & 1) * 2 + (q >> 1) holds the four bytes of q
    MX_STORE16(img + (4 + hh) * MX_B_PLANE + trow * 16, ((mx_u4){lo[0], lo[2], lo[1], lo[3]}));
}

// ---- weights: fp32 [N][K] (nn.Linear) -> image layout [N / 384][K / 32][36 KiB]
__global__ void mx_pack_weight_kernel(const float* __restrict__ w, int N, int K, char* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * K) return;
    const int n = (int)(i / K), k = (int)(i % K);
    const float x = w[i];
    const half_t hi = (half_t)x;
    const half_t l = (half_t)((x - (float)hi) * MX_LO_SCALE);
    char* img = out + ((size_t)(n / MX_TM) * (K >> 5) + (k >> 5)) * MX_A_STAGE;
    *reinterpret_cast<half_t*>(img + mx_hi_off(MX_TM, n % MX_TM, k & 31)) = hi;
    *reinterpret_cast<unsigned char*>(img + mx_lo_off(MX_TM, n % MX_TM, k & 31)) =
        (unsigned char)mx_e5m2_rn((unsigned)__builtin_bit_cast(unsigned short, l));
}

// ---- embeddings + LayerNorm -> x8 (hidden = 384 = 48 chunks of 8 features: lanes 0..47 of the token's wave own one chunk each).
// A workgroup owns 16 consecutive packed rows (four per wave) and assembles their image rows in LDS first: a token's 1152 B are
// 4-8 B pieces spread over 72 (K-step, plane) rows of the image, which written straight from the waves were 192 partial-line stores
// per token (1.05 ms per 362 k rows, under 1 TB/s); from LDS every (K-step, plane) leaves as 16 rows x 16 B = 256 contiguous bytes.
#define MX_EMB_ROWS 16
__global__ __launch_bounds__(256) void mx_embed_ln_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ tt,
                                                           const float* __restrict__ word, const float* __restrict__ pos,
                                                           const float* __restrict__ type, const float* __restrict__ g,
                                                           const float* __restrict__ b, const int32_t* __restrict__ m_packed,
                                                           const int32_t* __restrict__ row_pair, const int32_t* __restrict__ pair_off,
                                                           int L, int vocab, float eps, char* __restrict__ x8) {
    constexpr int H = 384, NK = H / 32;
    __shared__ __attribute__((aligned(16))) char tile[NK * 6 * MX_EMB_ROWS * 16];       // [K-step][plane][row][16 B] = 18 KiB
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * MX_EMB_ROWS;
    const int n_rows = m_packed[0];
    if (base >= n_rows) return;
    const bool on = lane < H / 8;
    const int c = on ? lane : 0;
    const float4 g0 = *reinterpret_cast<const float4*>(g + c * 8), g1 = *reinterpret_cast<const float4*>(g + c * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(b + c * 8), b1 = *reinterpret_cast<const float4*>(b + c * 8 + 4);
    // chunk c = features 8c..8c+7: K-step c >> 2, MFMA j = (c >> 1) & 1; its first four features are positions 4 (c & 1) .. of the
    // fragment of lane half 0, the other four the same positions of lane half 1 (mx_hi_off / mx_lo_off)
    const int j = (c >> 1) & 1, u = c & 1;
    char* const my = tile + (c >> 2) * 6 * MX_EMB_ROWS * 16;
#pragma unroll
    for (int r4 = 0; r4 < MX_EMB_ROWS / 4; ++r4) {
        const int rl = wv * (MX_EMB_ROWS / 4) + r4;
        const int64_t row = base + rl;
        if (row >= n_rows) break;                            // wave-uniform
        const int pr = row_pair[row];
        const int p = (int)row - pair_off[pr];
        const size_t src = (size_t)pr * L + p;
        int id = ids[src];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const int ty = tt[src] != 0;
        float v[8];
        {
            const float4* wp = reinterpret_cast<const float4*>(word + (size_t)id * H + c * 8);
            const float4* tp = reinterpret_cast<const float4*>(type + (size_t)ty * H + c * 8);
            const float4* pp = reinterpret_cast<const float4*>(pos + (size_t)p * H + c * 8);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4 a = wp[q], t4 = tp[q], qq = pp[q];
                v[q * 4] = a.x + t4.x + qq.x; v[q * 4 + 1] = a.y + t4.y + qq.y; v[q * 4 + 2] = a.z + t4.z + qq.z; v[q * 4 + 3] = a.w + t4.w + qq.w;
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += on ? v[i] : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / H);
        float qd = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; qd += on ? d * d : 0.f; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) qd += __shfl_xor(qd, o);
        const float rstd = 1.0f / sqrtf(qd * (1.0f / H) + eps);
        if (on) {
            mx_u2 h0, h1;
            unsigned l0, l1;
            mx_split4((v[0] - mean) * rstd * g0.x + b0.x, (v[1] - mean) * rstd * g0.y + b0.y, (v[2] - mean) * rstd * g0.z + b0.z,
                      (v[3] - mean) * rstd * g0.w + b0.w, h0, l0);
            mx_split4((v[4] - mean) * rstd * g1.x + b1.x, (v[5] - mean) * rstd * g1.y + b1.y, (v[6] - mean) * rstd * g1.z + b1.z,
                      (v[7] - mean) * rstd * g1.w + b1.w, h1, l1);
            char* o = my + rl * 16;                          // + plane * MX_EMB_ROWS * 16
            *reinterpret_cast<mx_u2*>(o + (2 * j) * MX_EMB_ROWS * 16 + u * 8) = h0;
            *reinterpret_cast<mx_u2*>(o + (2 * j + 1) * MX_EMB_ROWS * 16 + u * 8) = h1;
            *reinterpret_cast<unsigned*>(o + 4 * MX_EMB_ROWS * 16 + u * 8 + j * 4) = l0;
            *reinterpret_cast<unsigned*>(o + 5 * MX_EMB_ROWS * 16 + u * 8 + j * 4) = l1;
        }
    }
    __syncthreads();
    // the 16 rows sit in one 128-row image tile (16 | 128): (K-step, plane) rows of 256 contiguous bytes
    const int trow0 = (int)(base & 127);
    char* const out = x8 + (size_t)(base >> 7) * NK * MX_B_STAGE + trow0 * 16;
    for (int i = threadIdx.x; i < NK * 6 * MX_EMB_ROWS; i += 256) {
        const int sp = i >> 4, r = i & (MX_EMB_ROWS - 1);
        if (base + r < n_rows) *reinterpret_cast<mx_u4*>(out + (size_t)sp * MX_B_PLANE + r * 16) = *reinterpret_cast<const mx_u4*>(tile + i * 16);
    }
}

// Row pair_off[p] of one or two (b may be null) image-layout tensors with nk K-steps -> row p of their compact counterparts (one row per pair); also
// publishes the compact row count. One thread per (pair, 16-B chunk): a row is nk x 6 chunks.
__global__ void mx_gather_rows_kernel(const char* __restrict__ a, const char* __restrict__ b, const int32_t* __restrict__ pair_off, int P, int nk,
                                      char* __restrict__ a_out, char* __restrict__ b_out, int32_t* __restrict__ rows_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) rows_out[0] = P;
    const int per = nk * 6;
    if (i >= (int64_t)P * per) return;
    const int p = (int)(i / per), c = (int)(i % per), s = c / 6, pl = c % 6;
    const int64_t m = pair_off[p];
    const size_t src = ((size_t)(m >> 7) * nk + s) * MX_B_STAGE + (size_t)pl * MX_B_PLANE + (size_t)(m & 127) * 16;
    const size_t dst = ((size_t)(p >> 7) * nk + s) * MX_B_STAGE + (size_t)pl * MX_B_PLANE + (size_t)(p & 127) * 16;
    *reinterpret_cast<mx_u4*>(a_out + dst) = *reinterpret_cast<const mx_u4*>(a + src);
    if (b != nullptr) *reinterpret_cast<mx_u4*>(b_out + dst) = *reinterpret_cast<const mx_u4*>(b + src);
}

// value of (token row m, feature k) of an image-layout activation tensor
__device__ __forceinline__ float mx_load_elem(const char* __restrict__ x8, int64_t m, int k, int nk) {
    const char* img = x8 + mx_img_base(m, k, nk);
    const int trow = (int)(m & 127);
    return mx_join(*reinterpret_cast<const half_t*>(img + mx_hi_off(MX_TN, trow, k & 31)),
                   *reinterpret_cast<const unsigned char*>(img + mx_lo_off(MX_TN, trow, k & 31)));
}

// ---- QKV projection: feature tile 0 = Q, 1 = K (MFMA fragment order of the attention kernel, split fp16), 2 = V (computed with
// swapped operands so that a lane holds 4 consecutive KEYS of one dim: the V fragment order)
// ft_base: the launch's feature tile 0 is tile ft_base of the QKV matrix (the last layer of a classifier projects K and V for every
// token, ft_base = 1, and Q for the [CLS] rows alone). row_map (Q of compact rows only): token row of compact row p.
struct mx_epi_qkv {
    half_t *qf16, *kf16, *vf16;
    size_t kv_plane;
    const float* bias;
    int m_tiles16;                 // 16-row tiles of the padded row space
    int ft_base;
    const int32_t* row_map;
    int n_map;
    __device__ __forceinline__ bool swap_for(int ft) const { return ft + ft_base == 2; }
    __device__ __forceinline__ void prepare(char* scratch) const { mx_stage_params(reinterpret_cast<float*>(scratch + MX_PARAM_OFF), bias, 3 * MX_TM); }
    __device__ __forceinline__ void operator()(f32x16 (&acc)[6], int tt, int ft_launch, bool swap, char* scratch) const {
        const float* const bias = reinterpret_cast<const float*>(scratch + MX_PARAM_OFF);      // the staged copy
        const int ft = ft_launch + ft_base;
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 2, wn = wid & 3;
        const int li = lane & 31, hh = lane >> 5;
        const int m0 = tt * MX_TN + wn * 32;                       // first token row of the wave
        if (!swap) {
            half_t* dst = ft == 0 ? qf16 : kf16;
            int m = m0 + li;
            if (row_map != nullptr) {
                if (m >= n_map) return;
                m = row_map[m];
            }
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const int head = wm * 6 + b;
                half_t* tile = dst + (((size_t)head * m_tiles16 + (m >> 4)) * 64 + (m & 15)) * 8 + 4 * hh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias + ft * MX_TM + head * 32 + 8 * q + 4 * hh);
                    const float v0 = acc[b][q * 4] + bv.x, v1 = acc[b][q * 4 + 1] + bv.y, v2 = acc[b][q * 4 + 2] + bv.z, v3 = acc[b][q * 4 + 3] + bv.w;
                    const half4 hi = {(half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3};
                    const half4 lo = mx_resid4(hi, v0, v1, v2, v3);
                    __builtin_nontemporal_store(hi, reinterpret_cast<half4*>(tile + q * 128));
                    __builtin_nontemporal_store(lo, reinterpret_cast<half4*>(tile + q * 128 + kv_plane));
                }
            }
        } else {
            // lane = dim li of head wm*6 + b; register 4q + e = token m0 + 8q + 4hh + e: 16-row tile (q >> 1), key slots
            // fq = 2 (q & 1) + hh, e -> vf16[head][tile][d half][fq*16 + d%16][4]
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const int head = wm * 6 + b;
                const float bv = bias[2 * MX_TM + head * 32 + li];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v0 = acc[b][q * 4] + bv, v1 = acc[b][q * 4 + 1] + bv, v2 = acc[b][q * 4 + 2] + bv, v3 = acc[b][q * 4 + 3] + bv;
                    const half4 hi = {(half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3};
                    const half4 lo = mx_resid4(hi, v0, v1, v2, v3);
                    half_t* o = vf16 + ((((size_t)head * m_tiles16 + (m0 >> 4) + (q >> 1)) * 2 + (li >> 4)) * 64 + (2 * (q & 1) + hh) * 16 + (li & 15)) * 4;
                    __builtin_nontemporal_store(hi, reinterpret_cast<half4*>(o));
                    __builtin_nontemporal_store(lo, reinterpret_cast<half4*>(o + kv_plane));
                }
            }
        }
    }
};

// ---- FFN up-projection: bias + erf-GELU -> h8 (image layout, K = ffn for the down-projection)
__device__ __forceinline__ float mx_gelu(float x) {     // erf as ce_gelu (cross_encoder.hip): A&S 7.1.26, |erf error| <= 1.5e-7
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float erf_abs = fmaf(-p * t, e, 1.0f);
    // x/2 (1 + sign(x) erf|z|) = x/2 + |x/2| erf|z|: one multiply and one fma (|.| is an operand modifier) where the literal form
    // takes a bit-field insert, an add and two multiplies; one rounding less
    const float hx = 0.5f * x;
    return fmaf(fabsf(hx), erf_abs, hx);
}
struct mx_epi_gelu {
    char* h8;
    const float* bias;
    int nk_out;                    // K-steps of the consumer = ffn / 32
    __device__ __forceinline__ bool swap_for(int) const { return false; }
    __device__ __forceinline__ void prepare(char* scratch) const { mx_stage_params(reinterpret_cast<float*>(scratch + MX_PARAM_OFF), bias, nk_out * 32); }
    __device__ __forceinline__ void operator()(f32x16 (&acc)[6], int tt, int ft, bool, char* scratch) const {
        const float* const bias = reinterpret_cast<const float*>(scratch + MX_PARAM_OFF);      // the staged copy
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 2, wn = wid & 3;
        const int trow = wn * 32 + (lane & 31), hh = lane >> 5;
        char* tile = h8 + (size_t)tt * nk_out * MX_B_STAGE;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int f0 = ft * MX_TM + wm * 192 + b * 32;
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4*>(bias + f0 + 8 * q + 4 * hh);
                v[q * 4] = mx_gelu(acc[b][q * 4] + bv.x);
                v[q * 4 + 1] = mx_gelu(acc[b][q * 4 + 1] + bv.y);
                v[q * 4 + 2] = mx_gelu(acc[b][q * 4 + 2] + bv.z);
                v[q * 4 + 3] = mx_gelu(acc[b][q * 4 + 3] + bv.w);
            }
            mx_store_block(tile + (size_t)(f0 >> 5) * MX_B_STAGE, trow, hh, v);
        }
    }
};

// ---- out-projection / FFN down-projection: bias + residual + LayerNorm -> x8 in place (n_ft = 1: the tile is the hidden state)
struct mx_epi_ln {
    char* x8;                      // residual stream: read (residual) and rewritten (the new stream)
    const float *bias, *gamma, *beta;
    float eps;
    __device__ __forceinline__ bool swap_for(int) const { return false; }
    __device__ __forceinline__ void prepare(char* scratch) const {
        float* p = reinterpret_cast<float*>(scratch + MX_PARAM_OFF);
        mx_stage_params(p, bias, 384);
        mx_stage_params(p + 384, gamma, 384);
        mx_stage_params(p + 768, beta, 384);
    }
    __device__ __forceinline__ void operator()(f32x16 (&acc)[6], int tt, int, bool, char* scratch) const {
        constexpr int H = 384;
        const float* const bias = reinterpret_cast<const float*>(scratch + MX_PARAM_OFF);      // the staged copies
        const float *const gamma = bias + 384, *const beta = bias + 768;
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 2, wn = wid & 3;
        const int li = lane & 31, hh = lane >> 5, trow = wn * 32 + li;
        char* tile = x8 + (size_t)tt * (H / 32) * MX_B_STAGE;
        float* st_sum = reinterpret_cast<float*>(scratch) + wid * 32;          // [8 waves][32 tokens]
        float* st_sq = reinterpret_cast<float*>(scratch) + 256 + wid * 32;
        const float* pr_sum = reinterpret_cast<const float*>(scratch) + (wid ^ 4) * 32;
        const float* pr_sq = reinterpret_cast<const float*>(scratch) + 256 + (wid ^ 4) * 32;
        // v = acc + bias + residual, in place in the accumulators
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int f0 = wm * 192 + b * 32;
            const char* img = tile + (size_t)(f0 >> 5) * MX_B_STAGE;
            // the lane's residual: the 16 B of planes hh, 2 + hh (hi: q = 2j | q = 2j + 1) and 4 + hh (lo dwords: q = 0, 2, 1, 3)
            const half8 r0 = *reinterpret_cast<const half8*>(img + hh * MX_B_PLANE + trow * 16);
            const half8 r1 = *reinterpret_cast<const half8*>(img + (2 + hh) * MX_B_PLANE + trow * 16);
            const mx_u4 rl4 = *reinterpret_cast<const mx_u4*>(img + (4 + hh) * MX_B_PLANE + trow * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4*>(bias + f0 + 8 * q + 4 * hh);
                const half8 rj = (q >> 1) ? r1 : r0;
                const half4 rh = (q & 1) ? (half4){rj[4], rj[5], rj[6], rj[7]} : (half4){rj[0], rj[1], rj[2], rj[3]};
                const unsigned rl = rl4[(q & 1) * 2 + (q >> 1)];
                float res[4];
                mx_join4(rh, rl, res);
                acc[b][q * 4] += bv.x + res[0];
                acc[b][q * 4 + 1] += bv.y + res[1];
                acc[b][q * 4 + 2] += bv.z + res[2];
                acc[b][q * 4 + 3] += bv.w + res[3];
                sm += (acc[b][q * 4] + acc[b][q * 4 + 1]) + (acc[b][q * 4 + 2] + acc[b][q * 4 + 3]);
            }
        }
        // row statistics: a token's 384 features = 2 lane halves x 2 feature-half waves x 96 registers. Every lane takes the mean and the
        // centred sum of squares of ITS 96 values (no cancellation whatever the row's mean is), the four parts are combined pairwise
        // (Chan et al.: M2 = M2_a + M2_b + (m_a - m_b)^2 n / 2 for equal counts n): ONE exchange with the other feature-half wave and one
        // barrier per tile; the scratch is rewritten only after the >= 12 K-step barriers of the next tile
        const float m_l = sm * (1.0f / 96.0f);
#pragma unroll
        for (int b = 0; b < 6; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = acc[b][r] - m_l; sq = fmaf(d, d, sq); }
        const float m_o = __shfl_xor(m_l, 32), sq_o = __shfl_xor(sq, 32);
        const float m_w = 0.5f * (m_l + m_o);                                  // this wave's 192 features of the row
        const float sq_w = (sq + sq_o) + (m_l - m_o) * (m_l - m_o) * 48.0f;
        if (hh == 0) { st_sum[li] = m_w; st_sq[li] = sq_w; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        MX_BAR
        const float m_p = pr_sum[li], sq_p = pr_sq[li];
        const float mu = 0.5f * (m_w + m_p);
        const float var = ((sq_w + sq_p) + (m_w - m_p) * (m_w - m_p) * 96.0f) * (1.0f / H);
        const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int f0 = wm * 192 + b * 32;
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 gv = *reinterpret_cast<const float4*>(gamma + f0 + 8 * q + 4 * hh);
                const float4 be = *reinterpret_cast<const float4*>(beta + f0 + 8 * q + 4 * hh);
                v[q * 4] = fmaf(acc[b][q * 4] - mu, rs * gv.x, be.x);
                v[q * 4 + 1] = fmaf(acc[b][q * 4 + 1] - mu, rs * gv.y, be.y);
                v[q * 4 + 2] = fmaf(acc[b][q * 4 + 2] - mu, rs * gv.z, be.z);
                v[q * 4 + 3] = fmaf(acc[b][q * 4 + 3] - mu, rs * gv.w, be.w);
            }
            mx_store_block(tile + (size_t)(f0 >> 5) * MX_B_STAGE, trow, hh, v);
        }
    }
};

template <class EPI>
__global__ __launch_bounds__(512) void mx_gemm_kernel(const char* __restrict__ W, const char* __restrict__ X, int nk, int n_ft,
                                                       const int32_t* __restrict__ m_packed, EPI epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n_tt = (m_packed[0] + MX_TN - 1) / MX_TN;
    mx_gemm_loop(W, X, nk, n_ft, n_tt, smem, epi);
}
#define MX_KERNEL_LDS (MX_LDS + MX_PARAM_OFF + 6144)      // 155,648 B: ring + exchange space + staged parameters (<= 1536 floats)

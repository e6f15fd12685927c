// Cross-encoder GEMMs, round-4 operand scheme: hi16 + lo8 ("MX" form), 2 MFMA units per product instead of 3, 3 bytes per
// element instead of 4. Included by cross_encoder.hip (product) and tools/ce_mx_probe.hip (diagnostic harness).
//
// Numerics (tools/ce_numerics_sim.py): x = hi + lo, hi = fp16(x). rounds 1-3 kept lo as a second fp16 and
// spent three fp16 MFMAs per product (hi.hi + lo.hi + hi.lo). The two correction products only have to be good to a few
// per cent, so here
//   * lo8 = e5m2(lo * 2^11 * G) (one byte, round to nearest),
//   * hi8 = the TOP BYTE of hi's fp16 pattern = e5m2(hi) rounded toward zero: not stored, made in registers by v_perm_b32 from the
//     fp16 fragment the wave holds anyway; G = 1 / 0.915 undoes the mean loss of that truncation,
//   * both correction products of a 32-element K range run as ONE block-scaled bf8 MFMA of 64 K-slots:
//     A = [lo8 | hi8], B = [hi8 | lo8], scale 2^-11 (v_mfma_scale_f32_32x32x64_f8f6f4: twice the fp16 rate per K-slot).
// Per 32x32 block and 32-deep K-step: 2 x v_mfma_f32_32x32x16_f16 (64 cycles) + 1 x scaled bf8 (64 cycles) = 128 cycles against
// 192 for the split-fp16 form, and 96 B per operand row against 128 B.
//
// Layout ("image" layout; operands are stored in HBM exactly as the LDS stage holds them, so every LDS-DMA piece is 1 KiB of
// consecutive bytes = 8 whole lines, and a fragment read is conflict-free without a swizzle):
//   a ROW TILE is 384 rows for weights (MX_TM: all features a workgroup owns), 128 rows for activations (MX_TN tokens);
//   tensor = [row tile][K-step s = k / 32][plane c = 0..5][row in tile][16 B]
//   planes 0..3: hi16, plane 2*j + h holds elements k%32 = 16*j + 8*h + i (i = 0..7) as 8 halfs - the fragment of lane half h
//                for the j-th v_mfma_f32_32x32x16_f16 of the step;
//   planes 4..5: lo8, plane 4 + h byte p = (i >> 2) * 8 + j * 4 + (i & 3) holds element 16*j + 8*h + i - the order in which
//                v_perm_b32 leaves the top bytes of the lane's two fp16 fragments.
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
#define MX_LO_GAIN 1.0928961748633879f            // 1 / 0.915: mean of hi / trunc_e5m2(hi)
#define MX_LO_SCALE (2048.0f * MX_LO_GAIN)        // lo8 = e5m2(lo * MX_LO_SCALE)
#define MX_SCALE_A 116                            // E8M0 of 2^-11: the correction MFMA's block scale (B side: 127 = 2^0)
#define MX_SCALE_B 127

// byte offset of element (row r of the tile, column k % 32 = e) inside one K-step image with `rows` rows per plane
__host__ __device__ __forceinline__ int mx_hi_off(int rows, int r, int e) {
    return ((e >> 4) * 2 + ((e >> 3) & 1)) * rows * 16 + r * 16 + (e & 7) * 2;
}
__host__ __device__ __forceinline__ int mx_lo_off(int rows, int r, int e) {
    const int i = e & 7;
    return (4 + ((e >> 3) & 1)) * rows * 16 + r * 16 + (i >> 2) * 8 + (e >> 4) * 4 + (i & 3);
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

// the top bytes of four halfs held in two dwords (lo dword first)
__device__ __forceinline__ int mx_top4(int d0, int d1) { return (int)__builtin_amdgcn_perm((unsigned)d1, (unsigned)d0, 0x07050301u); }

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
// MX_DIAG is defined by tools/ce_mx_probe.hip only (timing experiments with wrong results): 1 = no DMA, 2 = no MFMA, 4 = no weight
// DMA, 8 = no token DMA. The product build compiles none of it.
#ifndef MX_DIAG
#define MX_DIAG 0
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
        if (MX_DIAG & 1) {}                                                                                              \
        else if (p_ < 36) { if (!(MX_DIAG & 4)) mx_bdma((S).w_rs, (S).voff, (ws_) + (ks_) * MX_A_STAGE + p_ * 1024, (st_) + p_ * 1024); } \
        else if (!(MX_DIAG & 8)) mx_bdma(xrs_, (S).voff, (ks_) * MX_B_STAGE + (p_ - 36) * 1024, (st_) + p_ * 1024);        \
    }

template <bool SWAP>
__device__ __forceinline__ void mx_ksteps(f32x16 (&acc)[6], const mx_stream& S, bool first) {
    const int nk = S.nk;
    for (int t = 0; t < nk; ++t) {
        // own pieces of step t have landed once at most the six of step t+1 are outstanding (a continued tile's steps 0 and 1
        // were waited for before the previous epilogue's stores)
        if (MX_DIAG & 13) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (first || t > 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
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
            const mx_v4i w8h = mx_hi8(wh0[b & 1], wh1[b & 1]);
            const mx_v8i w8 = SWAP ? mx_cat(w8h, wl[b & 1]) : mx_cat(wl[b & 1], w8h);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if (MX_DIAG & 2) acc[b][0] += (float)wh0[b & 1][0] + (float)wh1[b & 1][7] + (float)w8[0] + (float)w8[7] + (float)xh0[1] + (float)xh1[2] + (float)x8[3];
            else mx_block<SWAP>(acc[b], wh0[b & 1], wh1[b & 1], w8, xh0, xh1, x8);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (b + 2 < 6) { MX_READ_A(b + 2, b & 1) }
            MX_PIECE(S, b, dst, wsrc, xsrc, ks_)
        }
#undef MX_READ_A
    }
}

// EPI functor: called once per finished tile with the 6 accumulator blocks of the wave.
//   acc[b][r]: SWAP = false: feature ft*384 + wm*192 + b*32 + (r&3) + 8*(r>>2) + 4*(lane>>5), token tt*128 + wn*32 + (lane&31)
//              SWAP = true : feature ft*384 + wm*192 + b*32 + (lane&31), token tt*128 + wn*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)
// The functor may use `scratch` (MX_LDS .. 160 KiB) and must not touch the ring. swap_for(ft) picks the operand order per tile.
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

// Diagnostic harness (NOT product): experimental 4-wave dense GEMM tile pipeline (each wave 128 x 128 of a 256 x 256
// tile, ONE barrier per 64-deep K-step, fragment reads software-pipelined under the MFMAs inside each wave).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/gemm_probe4.hip -o tools/bin/gemm_probe4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define SLOT_BYTES 32768                       // 256 rows x 64 halfs
#define T4_LDS (5 * SLOT_BYTES)                // A: 3 slots, B: 2 slots = 160 KiB

__device__ __forceinline__ void stage_slot(const half_t* __restrict__ gsrc, int ld, char* slot, int wid) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (size_t)j * 32 * ld),
                                         (__attribute__((address_space(3))) void*)(slot + (j * 256 + wid * 64) * 16), 16, 0, 0);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// 4 waves, each 128 x 128 as 4 x 4 blocks of v_mfma_f32_32x32x16_f16 (fragments: 8 x 16 B per 16-deep substep instead of
// 16 x 16 B per 32-deep one: half the fragment registers, same LDS bytes). K-step = 4 substeps; fragments of the next
// substep are read under the 16 MFMAs of the current one; ONE barrier per K-step, placed before the last substep.
template <bool CHECK>
__global__ __launch_bounds__(256) void dense4_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B, int Dp, int n_rtiles,
                                                      int n_qtiles, float* __restrict__ out, unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int b = blockIdx.x;
    const int xcd = b & 7, seq = b >> 3;
    const int rt = (seq / n_qtiles) * 8 + xcd;
    const int qt = seq % n_qtiles;
    if (rt >= n_rtiles) return;
    const int row0 = rt * 256, q0 = qt * 256;
    const int sr = tid >> 3;
    const int schunk = (tid & 7) ^ ((sr >> 1) & 7);
    const half_t* a_src = A + (size_t)(row0 + sr) * Dp + schunk * 8;
    const half_t* b_src = B + (size_t)(q0 + sr) * Dp + schunk * 8;
    const int l32 = lane & 31, lh = lane >> 5;
    const int sw = (l32 >> 1) & 7;
    int off_s[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) off_s[ks] = l32 * 128 + (((ks * 2 + lh) ^ sw) << 4);
    const int a_off = wm * 128 * 128, b_off = wn * 128 * 128;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nt = Dp / 64, last = nt - 1;
#define KOFF(u) (((u) < last ? (u) : last) * 64)
#define A_SLOT(u) (smem + ((u) % 3) * SLOT_BYTES)
#define B_SLOT(u) (smem + 3 * SLOT_BYTES + ((u) & 1) * SLOT_BYTES)
#define STAGE_A(u) stage_slot(a_src + KOFF(u), Dp, A_SLOT(u), wid)
#define STAGE_B(u) stage_slot(b_src + KOFF(u), Dp, B_SLOT(u), wid)
#define LOAD_FRAGS(FA, FB, u, ks)                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
        FA[i] = *reinterpret_cast<const half8*>(A_SLOT(u) + a_off + i * 32 * 128 + off_s[ks]);             \
        FB[i] = *reinterpret_cast<const half8*>(B_SLOT(u) + b_off + i * 32 * 128 + off_s[ks]);             \
    }
#define MFMAS(FA, FB)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                         \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[i], FB[j], acc[i][j], 0, 0, 0);
#define INTERLEAVE_DS                                                                                     \
    _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                                       \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                \
    }
    STAGE_A(0); STAGE_B(0); STAGE_A(1); STAGE_B(1); STAGE_A(2);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    half8 fa0[4], fb0[4], fa1[4], fb1[4];
    LOAD_FRAGS(fa0, fb0, 0, 0)
    for (int t = 0; t < nt; ++t) {
        LOAD_FRAGS(fa1, fb1, t, 1)
        MFMAS(fa0, fb0)
        INTERLEAVE_DS
        LOAD_FRAGS(fa0, fb0, t, 2)
        MFMAS(fa1, fb1)
        INTERLEAVE_DS
        LOAD_FRAGS(fa1, fb1, t, 3)
        MFMAS(fa0, fb0)
        INTERLEAVE_DS
        // hand-over: this step's slots are fully read (the substep-3 fragments are in registers), step t+1 has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        STAGE_B(t + 2);
        STAGE_A(t + 3);
        LOAD_FRAGS(fa0, fb0, t + 1, 0)
        MFMAS(fa1, fb1)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // C layout of a 32 x 32 block: col = lane & 31, row = (r / 4) * 8 + (lane >> 5) * 4 + r % 4
    if (CHECK) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out[((size_t)(rt * n_qtiles + qt) * 256 + wm * 128 + i * 32 + (r / 4) * 8 + lh * 4 + r % 4) * 256 + wn * 128 + j * 32 + l32] =
                        acc[i][j][r];
    } else {
        float mx = -1e30f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[i][j][r]);
        if (mx > 1e30f) atomicAdd(sink, 1u);
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 917504, Q = argc > 2 ? atoi(argv[2]) : 1024, D = 1536, iters = 5;
    const int n_rt = N / 256, n_qt = Q / 256;
    std::vector<half_t> hc((size_t)4096 * D), hq((size_t)Q * D);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 3.3f);
    for (auto& v : hc) v = (half_t)nd(rng);
    for (auto& v : hq) v = (half_t)nd(rng);
    half_t *c, *q;
    unsigned* sink;
    float* out;
    CK(hipMalloc(&c, (size_t)N * D * 2));
    CK(hipMalloc(&q, (size_t)Q * D * 2));
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&out, (size_t)8 * n_qt * 256 * 256 * 4));
    for (size_t r = 0; r < (size_t)N; r += 4096)
        CK(hipMemcpy(c + r * D, hc.data(), std::min<size_t>(4096, N - r) * D * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, hq.data(), (size_t)Q * D * 2, hipMemcpyHostToDevice));
    CK(hipMemset(sink, 0, 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dense4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, T4_LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dense4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, T4_LDS));
    const int grid = ((n_rt + 7) / 8 * 8) * n_qt;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < iters; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((dense4_kernel<false>), dim3(grid), dim3(256), T4_LDS, 0, c, q, D, n_rt, n_qt, out, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    {   // correctness on the first 8 row tiles
        hipLaunchKernelGGL((dense4_kernel<true>), dim3(8 * n_qt), dim3(256), T4_LDS, 0, c, q, D, 8, n_qt, out, sink);
        CK(hipDeviceSynchronize());
        std::vector<float> ho((size_t)8 * n_qt * 256 * 256);
        CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
        long bad = 0, total = 0;
        int badmap[16][16] = {};
        for (int qi = 0; qi < Q; qi += 5)
            for (int r = 0; r < 2048; r += 3) {
                double ref = 0;
                for (int k = 0; k < D; ++k) ref += (double)(float)hq[(size_t)qi * D + k] * (double)(float)hc[(size_t)r * D + k];
                const float got = ho[((size_t)((r / 256) * n_qt + qi / 256) * 256 + r % 256) * 256 + qi % 256];
                ++total;
                if (std::fabs(got - ref) > 1e-3 * (16384.0 + std::fabs(ref))) { ++bad; badmap[(r % 256) / 16][(qi % 256) / 16]++; }
            }
        printf("GEMM check: %ld / %ld wrong\n", bad, total);
        if (bad) for (int a = 0; a < 16; ++a) { for (int b2 = 0; b2 < 16; ++b2) printf("%5d", badmap[a][b2]); printf("\n"); }
    }
    printf("N=%d Q=%d  best %.3f ms  %.1f TFLOP/s  (%d WGs)\n", N, Q, best, 2.0 * Q * (double)N * D / (best * 1e-3) / 1e12, n_rt * n_qt);
    return 0;
}

// Diagnostic harness (NOT product): the MX GEMM main loop of optimized-rag_amd/csrc/ce_mx.h alone, on random operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ce_mx_probe.hip -o tools/bin/ce_mx_probe
//   tools/bin/ce_mx_probe [tokens] [features] [K] [swap]
// Checks (a) the kernel against a float64 emulation of ITS OWN arithmetic (hi.hi + 2^-11 (lo8.hi8 + hi8.lo8) on the rounded
// operands: validates the layout, the lane maps of both MFMA forms and the K-slot pairing) and (b) against the exact W.X^T (what
// the scheme is worth), then times the launch with a store-nothing epilogue and with the image-layout epilogue.
// timing experiments (WRONG results; -DMX_DIAG=<bits>): 1 = no DMA, 2 = no MFMA, 4 = no weight DMA, 8 = no token DMA, 16 = no weight-side v_perm, 32 = 6-bit correction MFMA, 64 / 128 = fewer weight fragment reads (5 / 3 per step instead of 7)
#ifndef MX_DIAG
#define MX_DIAG 0
#endif
#if MX_DIAG & 5
#define MX_ISSUE_W(...)
#endif
#if MX_DIAG & 9
#define MX_ISSUE_X(...)
#endif
#if MX_DIAG & 2
#define MX_BLOCK(SWAP, acc, wh0, wh1, w8, xh0, xh1, x8) \
    acc[0] += (float)(wh0)[0] + (float)(wh1)[7] + (float)(w8)[0] + (float)(w8)[7] + (float)(xh0)[1] + (float)(xh1)[2] + (float)(x8)[3]
#endif
#if MX_DIAG & 13
#define MX_STEP_WAIT "s_waitcnt vmcnt(0)"
#endif
#if MX_DIAG & 16                      // no weight-side v_perm (what a STORED weight hi8 plane would save in vector work)
#define MX_W_HI8(h0, h1, lo) (lo)
#endif
#if MX_DIAG & 64                      // read the weight fragments of blocks 0, 1, 2, 3 only (4 and 5 reuse them): 5 fragment reads per step instead of 7, what a 3 x 2 wave tile would read
#define MX_READ_A_IF(b) ((b) < 4)
#endif
#if MX_DIAG & 128                     // blocks 0 and 1 only: 3 reads per step
#define MX_READ_A_IF(b) ((b) < 2)
#endif
#if MX_DIAG & 32                      // the correction MFMA in a 6-bit format (e2m3: half the cycles of the 8-bit one), same registers: what fp6 corrections could buy at best
#define MX_BLOCK(SWAP, acc, wh0, wh1, w8, xh0, xh1, x8)                                                      \
    {                                                                                                        \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, xh0, acc, 0, 0, 0);                                \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, xh1, acc, 0, 0, 0);                                \
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8, x8, acc, 2, 2, 0, MX_SCALE_A, 0, MX_SCALE_B); \
    }
#endif
#include "../optimized-rag_amd/csrc/ce_mx.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct epi_f32 {          // verification: C[token][feature] fp32
    float* C;
    int N;
    bool want_swap;
    __device__ bool swap_for(int) const { return want_swap; }
    __device__ void prepare(char*) const {}
    __device__ void operator()(f32x16 (&acc)[6], int tt, int ft, bool swap, char*) const {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 2, wn = wid & 3;
#pragma unroll
        for (int b = 0; b < 6; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rowi = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), coli = lane & 31;
                const int f = ft * MX_TM + wm * 192 + b * 32 + (swap ? coli : rowi);
                const int t = tt * MX_TN + wn * 32 + (swap ? rowi : coli);
                C[(size_t)t * N + f] = acc[b][r];
            }
    }
};

struct epi_none {         // timing: main loop only
    float* sink;
    __device__ bool swap_for(int) const { return false; }
    __device__ void prepare(char*) const {}
    __device__ void operator()(f32x16 (&acc)[6], int, int, bool, char*) const {
        if (acc[0][0] == 12345.678f) sink[0] = acc[1][1] + acc[2][2] + acc[3][3] + acc[4][4] + acc[5][5];
    }
};

struct epi_img {          // timing + layout check: + bias -> image layout of the NEXT GEMM's token operand (K = N): the product's store path
    char* out;            // [token tile][N / 32][12 KiB image]
    const float* bias;
    int N;
    __device__ bool swap_for(int) const { return false; }
    __device__ void prepare(char*) const {}
    __device__ void operator()(f32x16 (&acc)[6], int tt, int ft, bool, char*) const {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 2, wn = wid & 3;
        const int t = wn * 32 + (lane & 31), hh = lane >> 5;
        char* tile = out + (size_t)tt * (N / 32) * MX_B_STAGE;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int f0 = ft * MX_TM + wm * 192 + b * 32;
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#ifdef PROBE_NO_BIAS
                const float4 bv = make_float4(0.5f, 0.25f, 0.125f, 1.0f);
#else
                const float4 bv = *reinterpret_cast<const float4*>(bias + f0 + 8 * q + 4 * hh);
#endif
                v[q * 4] = acc[b][q * 4] + bv.x; v[q * 4 + 1] = acc[b][q * 4 + 1] + bv.y; v[q * 4 + 2] = acc[b][q * 4 + 2] + bv.z; v[q * 4 + 3] = acc[b][q * 4 + 3] + bv.w;
            }
            mx_store_block(tile + (size_t)(f0 >> 5) * MX_B_STAGE, t, hh, v);
        }
    }
};

__device__ unsigned long long g_stamp[256][2];          // per workgroup: shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) of the launch

template <class EPI>
__global__ __launch_bounds__(512) void probe_kernel(const char* W, const char* X, int nk, int n_ft, int n_tt, EPI epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    mx_gemm_loop(W, X, nk, n_ft, n_tt, smem, epi);
    if (threadIdx.x == 0 && blockIdx.x < 256) {
        g_stamp[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - c0;
        g_stamp[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

static float e5m2_val(unsigned b) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(b << 8)); }

// host: a [rows][K] fp32 matrix -> image layout with `tile` rows per row tile; also returns the rounded parts for the emulation
static void to_image(const std::vector<float>& m, int rows, int K, int tile, std::vector<char>& img, std::vector<float>& hi,
                     std::vector<float>& hi8, std::vector<float>& lo8) {
    const int nk = K / 32;
    img.assign((size_t)rows * K * 3, 0);
    hi.resize(m.size()); hi8.resize(m.size()); lo8.resize(m.size());
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < K; ++k) {
            const float x = m[(size_t)r * K + k];
            const _Float16 h = (_Float16)x;
            const unsigned short hb = __builtin_bit_cast(unsigned short, h);
            const _Float16 l = (_Float16)((x - (float)h) * MX_LO_SCALE);
            const unsigned lb = mx_e5m2_rn(__builtin_bit_cast(unsigned short, l));
            char* base = img.data() + ((size_t)(r / tile) * nk + k / 32) * tile * 96;
            *reinterpret_cast<unsigned short*>(base + mx_hi_off(tile, r % tile, k % 32)) = hb;
            *reinterpret_cast<unsigned char*>(base + mx_lo_off(tile, r % tile, k % 32)) = (unsigned char)lb;
            hi[(size_t)r * K + k] = (float)h;
#ifndef MX_HI8_TRUNCATE
            hi8[(size_t)r * K + k] = e5m2_val(((unsigned)(hb + 0x80u) >> 8) & 0xFFu);     // the kernel's in-register rounding
#else
            hi8[(size_t)r * K + k] = e5m2_val(hb >> 8);
#endif
            lo8[(size_t)r * K + k] = e5m2_val(lb);
        }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 128 * 2048, N = argc > 2 ? atoi(argv[2]) : 1152, K = argc > 3 ? atoi(argv[3]) : 384;
    const bool swap = argc > 4 && atoi(argv[4]);
    if (M % MX_TN || N % MX_TM || K % 32) { printf("M %% 128, N %% 384, K %% 32\n"); return 1; }
    const int nk = K / 32, n_ft = N / MX_TM, n_tt = M / MX_TN;
    const int Mv = std::min(M, 1024);                       // rows with distinct data (the rest repeats them)
    std::mt19937 rng(3);
    std::normal_distribution<float> nw(0.f, 0.05f), nx(0.f, 1.0f);
    std::vector<float> w((size_t)N * K), x((size_t)Mv * K), bias(N);
    for (auto& v : w) v = nw(rng);
    for (auto& v : x) v = nx(rng);
    for (auto& v : bias) v = nw(rng);
    std::vector<char> wi, xi;
    std::vector<float> wh, wh8, wl8, xh, xh8, xl8;
    to_image(w, N, K, MX_TM, wi, wh, wh8, wl8);
    to_image(x, Mv, K, MX_TN, xi, xh, xh8, xl8);
    char *dW, *dX, *dO;
    float *dC, *dB;
    CK(hipMalloc(&dW, wi.size()));
    CK(hipMalloc(&dX, (size_t)M * K * 3));
    CK(hipMalloc(&dO, (size_t)M * N * 3));
    CK(hipMalloc(&dC, (size_t)Mv * N * 4));
    CK(hipMalloc(&dB, N * 4));
    CK(hipMemcpy(dW, wi.data(), wi.size(), hipMemcpyHostToDevice));
    for (size_t r = 0; r < (size_t)M; r += Mv) CK(hipMemcpy(dX + r * K * 3, xi.data(), xi.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, bias.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel<epi_f32>), hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel<epi_none>), hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel<epi_img>), hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS));
    // ---- correctness on the first Mv rows
    hipLaunchKernelGGL(probe_kernel<epi_f32>, dim3(256), dim3(512), MX_LDS, 0, dW, dX, nk, n_ft, Mv / MX_TN, epi_f32{dC, N, swap});
    CK(hipDeviceSynchronize());
    std::vector<float> c((size_t)Mv * N);
    CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
    double e_emu = 0, e_true = 0, e_f16 = 0, ref_rms = 0;
    int cnt = 0;
    for (int t = 0; t < Mv; t += 7)
        for (int f = 0; f < N; f += 5) {
            double hh = 0, corr = 0, ex = 0;
            for (int k = 0; k < K; ++k) {
                const size_t a = (size_t)f * K + k, b = (size_t)t * K + k;
                hh += (double)wh[a] * xh[b];
                corr += (double)wl8[a] * xh8[b] + (double)wh8[a] * xl8[b];
                ex += (double)w[a] * x[b];
            }
            const double emu = hh + corr / 2048.0, got = c[(size_t)t * N + f];
            e_emu = std::max(e_emu, std::fabs(got - emu));
            e_true = std::max(e_true, std::fabs(got - ex));
            e_f16 = std::max(e_f16, std::fabs(hh - ex));
            ref_rms += ex * ex;
            ++cnt;
        }
    printf("shape tokens %d x features %d x K %d, swap %d | rms |y| %.3f | max |kernel - own-arithmetic emulation| %.3e | max |kernel - exact| %.3e | "
           "(fp16-only would be %.3e)\n", M, N, K, (int)swap, std::sqrt(ref_rms / cnt), e_emu, e_true, e_f16);
    // ---- image epilogue check on the first tile rows
    hipLaunchKernelGGL(probe_kernel<epi_img>, dim3(256), dim3(512), MX_LDS, 0, dW, dX, nk, n_ft, Mv / MX_TN, epi_img{dO, dB, N});
    CK(hipDeviceSynchronize());
    if (!swap) {
        std::vector<char> o((size_t)Mv * N * 3);
        CK(hipMemcpy(o.data(), dO, o.size(), hipMemcpyDeviceToHost));
        double e_img = 0;
        for (int t = 0; t < Mv; t += 3)
            for (int f = 0; f < N; ++f) {
                const char* base = o.data() + ((size_t)(t / MX_TN) * (N / 32) + f / 32) * MX_B_STAGE;
                const float h = (float)*reinterpret_cast<const _Float16*>(base + mx_hi_off(MX_TN, t % MX_TN, f % 32));
                const float l = e5m2_val(*reinterpret_cast<const unsigned char*>(base + mx_lo_off(MX_TN, t % MX_TN, f % 32))) / MX_LO_SCALE;
                e_img = std::max(e_img, (double)std::fabs(h + l - (c[(size_t)t * N + f] + bias[f])) / std::max(1.0, (double)std::fabs(c[(size_t)t * N + f])));
            }
        printf("image epilogue: max rel |decode(hi16 + lo8) - (acc + bias)| %.3e (hi16 alone resolves 4.9e-4, hi16 + lo8 ~3e-5)\n", e_img);
    }
    // ---- timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel<mx_epi_gelu>), hipFuncAttributeMaxDynamicSharedMemorySize, MX_KERNEL_LDS));
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int itr = 0; itr < 6; ++itr) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(probe_kernel<epi_none>, dim3(256), dim3(512), MX_LDS, 0, dW, dX, nk, n_ft, n_tt, epi_none{dC});
            else if (mode == 1) hipLaunchKernelGGL(probe_kernel<epi_img>, dim3(256), dim3(512), MX_LDS, 0, dW, dX, nk, n_ft, n_tt, epi_img{dO, dB, N});
            else hipLaunchKernelGGL(probe_kernel<mx_epi_gelu>, dim3(256), dim3(512), MX_KERNEL_LDS, 0, dW, dX, nk, n_ft, n_tt, mx_epi_gelu{dO, dB, N / 32});
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (itr) best = std::min(best, ms);
        }
        unsigned long long hs[256][2];
        CK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_stamp), sizeof(hs)));
        std::vector<double> ghz;
        for (int b = 0; b < 256; ++b) if (hs[b][1]) ghz.push_back((double)hs[b][0] / (double)hs[b][1] * 0.1);
        std::sort(ghz.begin(), ghz.end());
        const double clk = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
        const double prod = 2.0 * M * N * K;
        const double steps = (double)n_tt * n_ft * nk / 256.0;
        printf("[in-kernel clock %.2f GHz] %s: %.3f ms | %.1f TFLOP/s of products (x2 units issued: %.1f) | %.3f us per K-step per CU | LDS fill %.2f TB/s (%.1f B/clk/CU at 2.4 GHz)\n",
               clk, mode == 0 ? "main loop only " : mode == 1 ? "image epilogue" : "gelu epilogue ", best, prod / best * 1e-9, 2 * prod / best * 1e-9, best * 1e3 / steps,
               (double)n_tt * n_ft * nk * MX_STAGE / best * 1e-9, (double)MX_STAGE / (best * 1e-3 / steps * 2.4e9));
    }
    return 0;
}

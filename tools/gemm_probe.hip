// Diagnostic harness (NOT product): runs dense_emit_kernel<false> alone on random fp16 operands with tau = +inf
// (nothing is emitted), times it with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/gemm_probe.hip -o gpurun_out/gemm_probe
#include "../optimized-rag_amd/csrc/dense.hip"

#include <cstdio>
#include <cstdlib>
#include <random>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 917504, Q = argc > 2 ? atoi(argv[2]) : 1024, D = 1536, iters = 5;
    const float tau_v = argc > 3 ? (float)atof(argv[3]) : INFINITY;      // emission threshold (inf = nothing emitted)
    const int n_rt = N / 256, n_qt = Q / 256;
    std::vector<half_t> hc((size_t)4096 * D), hq((size_t)Q * D);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 3.3f);
    for (auto& v : hc) v = (half_t)nd(rng);
    for (auto& v : hq) v = (half_t)nd(rng);
    half_t *c, *q;
    float* tau;
    unsigned* cnt;
    uint64_t* cand;
    CK(hipMalloc(&c, (size_t)N * D * 2));
    CK(hipMalloc(&q, (size_t)Q * D * 2));
    CK(hipMalloc(&tau, Q * 4));
    CK(hipMalloc(&cnt, Q * 4));
    CK(hipMalloc(&cand, (size_t)Q * RAG_CAND_CAP * 8));
    for (size_t r = 0; r < (size_t)N; r += 4096)       // random rows (tiled copy of 4096 distinct rows)
        CK(hipMemcpy(c + r * D, hc.data(), std::min<size_t>(4096, N - r) * D * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, hq.data(), (size_t)Q * D * 2, hipMemcpyHostToDevice));
    std::vector<float> ht(Q, tau_v);
    CK(hipMemcpy(tau, ht.data(), Q * 4, hipMemcpyHostToDevice));
    CK(hipMemset(cnt, 0, Q * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dense_emit_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                           DENSE_LDS_BYTES));
    const int grid = (int)round_up(n_rt, 8) * n_qt;
#define EXTRA
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < iters; ++it) {
        CK(hipMemset(cnt, 0, Q * 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((dense_emit_kernel<false, false>), dim3(grid), dim3(512), DENSE_LDS_BYTES, 0, c, q, D, 0, n_rt, n_qt, N, Q, tau, cnt,
                           cand, (const int32_t*)nullptr, 0, (const int32_t*)nullptr, 1, (N + 2047) / 2048, (N + 255) / 256, (const int*)nullptr, (const float*)nullptr, (int64_t)0, 1.0f, (const int*)nullptr EXTRA);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    {   // correctness of the GEMM itself: dense stage-0 instance writes every score of rows [0,2048) as keys
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dense_emit_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               DENSE_LDS_BYTES));
        CK(hipMemset(cand, 0, (size_t)Q * RAG_CAND_CAP * 8));
        hipLaunchKernelGGL((dense_emit_kernel<true, false>), dim3(8 * n_qt), dim3(512), DENSE_LDS_BYTES, 0, c, q, D, 0, 8, n_qt, 2048, Q, tau, cnt,
                           cand, (const int32_t*)nullptr, 0, (const int32_t*)nullptr, 1, (N + 2047) / 2048, (N + 255) / 256, (const int*)nullptr, (const float*)nullptr, (int64_t)0, 1.0f, (const int*)nullptr EXTRA);
        CK(hipDeviceSynchronize());
        std::vector<uint64_t> hk((size_t)Q * RAG_CAND_CAP);
        CK(hipMemcpy(hk.data(), cand, hk.size() * 8, hipMemcpyDeviceToHost));
        long bad = 0, total = 0;
        int badmap[16][16] = {};
        for (int qi = 0; qi < Q; qi += 3)
            for (int r = 0; r < 2048; r += 1) {
                double ref = 0;
                for (int k = 0; k < D; ++k) ref += (double)(float)hq[(size_t)qi * D + k] * (double)(float)hc[(size_t)r * D + k];
                ref /= 16384.0;
                const float got = key_score(hk[(size_t)qi * RAG_CAND_CAP + r]);
                ++total;
                if (std::fabs(got - ref) > 1e-3 * (1.0 + std::fabs(ref))) {
                    ++bad;
                    badmap[(r % 256) / 16][(qi % 256) / 16]++;
                }
            }
        printf("GEMM check: %ld / %ld wrong\n", bad, total);
        if (bad) {
            printf("wrong-count map [row16 block within tile][query16 block within tile]:\n");
            for (int a = 0; a < 16; ++a) { for (int b2 = 0; b2 < 16; ++b2) printf("%5d", badmap[a][b2]); printf("\n"); }
        }
    }
    const double tf = 2.0 * Q * (double)N * D / (best * 1e-3) / 1e12;
    unsigned hc0[4];
    CK(hipMemcpy(hc0, cnt, 16, hipMemcpyDeviceToHost));
    printf("N=%d Q=%d tau=%.4f  best %.3f ms  %.1f TFLOP/s  (%d WGs)  emitted/query ~ %u %u %u\n", N, Q, tau_v, best, tf, n_rt * n_qt,
           hc0[0], hc0[1], hc0[2]);
    return 0;
}

// Diagnostic (NOT product): (1) what v_permlane16_swap / v_permlane32_swap do to (x, y) = (lane, 100 + lane); (2) the maximum over the four
// lanes l, l^16, l^32, l^48 by two __shfl_xor steps against the same by permlane swaps of (m, m), on random data.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o, const float* in, int* bad) {
    const unsigned l = threadIdx.x;
    u2 a = __builtin_amdgcn_permlane16_swap(l, 100 + l, false, false);
    u2 b = __builtin_amdgcn_permlane32_swap(l, 100 + l, false, false);
    o[l * 4] = a[0]; o[l * 4 + 1] = a[1]; o[l * 4 + 2] = b[0]; o[l * 4 + 3] = b[1];
    int nb = 0;
    for (int it = 0; it < 1000; ++it) {
        const float m = in[it * 64 + l];
        float s = fmaxf(m, __shfl_xor(m, 16));
        s = fmaxf(s, __shfl_xor(s, 32));
        // through the builtin, clang 22 folds the SECOND result of the swap into the first whenever both go into one expression
        // (the IR keeps extractvalue 0 only): the instruction is issued by hand, with the two wait states the compiler puts in front of it
        float p = m, c = m;
        asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(c));
        p = fmaxf(p, c);
        c = p;
        asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(c));
        p = fmaxf(p, c);
        nb += s != p;
    }
    atomicAdd(bad, nb);
}
int main() {
    unsigned* d; float* in; int* bad;
    hipMalloc(&d, 64 * 16); hipMalloc(&in, 64000 * 4); hipMalloc(&bad, 4);
    float h_in[64000];
    for (int i = 0; i < 64000; ++i) h_in[i] = (float)((i * 2654435761u) % 100003) - 50000.f;
    hipMemcpy(in, h_in, sizeof(h_in), hipMemcpyHostToDevice); hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, in, bad);
    unsigned h[256]; int nb; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 8) printf("lane %2d: p16 (%3u, %3u)  p32 (%3u, %3u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    printf("four-lane maxima differing between the shuffle and the permlane form: %d of 64000\n", nb);
    return 0;
}

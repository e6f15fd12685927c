// Diagnostic (NOT product): does v_cvt_scalef32_pk_bf8_f32 with a power-of-two scale give the bits of v_cvt_pk_bf8_f32 on the pre-multiplied value?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/cvt_probe.hip -o tools/bin/cvt_probe && tools/bin/cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef short short2v __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, int n, unsigned* plain, unsigned* s_small, unsigned* s_big) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    plain[i] = (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(a * 2048.0f, b * 2048.0f, 0, false) & 0xFFFFu;
    short2v r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32((short2v){0, 0}, a, b, 1.0f / 2048.0f, false);
    s_small[i] = (unsigned)(unsigned short)r[0];
    r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32((short2v){0, 0}, a, b, 2048.0f, false);
    s_big[i] = (unsigned)(unsigned short)r[0];
}
int main() {
    const int n = 1 << 20;
    std::vector<float> x(2 * n);
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::uniform_int_distribution<int> ex(-30, 6);
    for (int i = 0; i < 2 * n; ++i) x[i] = std::ldexp(u(rng), ex(rng));
    // exact halfway cases and boundaries of e5m2 after the scaling: m * 2^e with 3-bit-plus-half mantissas
    for (int i = 0; i < 4096; ++i) x[i] = std::ldexp((float)(8 + (i & 7)) + 0.5f * ((i >> 3) & 1), -14 + ((i >> 4) % 24) - 11) * ((i >> 9) & 1 ? -1.f : 1.f);
    x[5000] = 0.f; x[5001] = -0.f; x[5002] = 1e30f; x[5003] = -1e30f; x[5004] = 57344.f / 2048.f; x[5005] = 61440.f / 2048.f; x[5006] = 1e-40f; x[5007] = 3.0e-8f;
    float* dx; unsigned *d0, *d1, *d2;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, n, d0, d1, d2);
    std::vector<unsigned> p(n), a(n), b(n);
    hipMemcpy(p.data(), d0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(a.data(), d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d2, n * 4, hipMemcpyDeviceToHost);
    long ma = 0, mb = 0; int shown = 0;
    for (int i = 0; i < n; ++i) {
        ma += p[i] != a[i]; mb += p[i] != b[i];
        if (p[i] != a[i] && shown < 12) { printf("x = (%.9g, %.9g): plain %04x, scale 2^-11 %04x, scale 2^11 %04x\n", x[2 * i], x[2 * i + 1], p[i], a[i], b[i]); ++shown; }
    }
    printf("pairs %d: mismatches with scale 2^-11: %ld, with scale 2^11: %ld\n", n, ma, mb);
    return 0;
}

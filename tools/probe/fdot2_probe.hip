// What does __builtin_amdgcn_fdot2 (v_dot2c_f32_f16) compute on gfx950?  Single op and dependent chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 half_t;
typedef half_t half2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const u32x4* x, float* y, int wc) {
    u32x4 xf[4][2];
    for (int p = 0; p < 4; ++p)
        for (int s = 0; s < 2; ++s) xf[p][s] = x[(threadIdx.x * 4 + p) * 2 + s];
    const half2v one2 = {(half_t)1.f, (half_t)1.f};
    float s1 = 0.f, s2 = 0.f;
    auto take = [&](const u32x4 (&f)[2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const half2v v = __builtin_bit_cast(half2v, f[ks][e]);
                s2 = __builtin_amdgcn_fdot2(v, v, s2, false);
                s1 = __builtin_amdgcn_fdot2(v, one2, s1, false);
            }
    };
    if (wc == 0) take(xf[0]);
    else if (wc == 1) take(xf[1]);
    else if (wc == 2) take(xf[2]);
    else take(xf[3]);
    y[2 * threadIdx.x] = s1;
    y[2 * threadIdx.x + 1] = s2;
}
int main() {
    const int n = 64 * 4 * 2 * 4;
    static unsigned hx[n];
    static float ref1[64][4], ref2[64][4];
    unsigned seed = 12345;
    for (int t = 0; t < 64; ++t)
        for (int p = 0; p < 4; ++p) {
            double a1 = 0, a2 = 0;
            for (int w = 0; w < 8; ++w) {
                unsigned word = 0;
                for (int hh = 0; hh < 2; ++hh) {
                    seed = seed * 1664525u + 1013904223u;
                    float f = ((int)(seed >> 8) % 2001 - 1000) / 400.0f;
                    half_t h = (half_t)f;
                    unsigned short b;
                    __builtin_memcpy(&b, &h, 2);
                    word |= (unsigned)b << (16 * hh);
                    a1 += (float)h; a2 += (double)(float)h * (float)h;
                }
                hx[((t * 4 + p) * 2 + w / 4) * 4 + w % 4] = word;
            }
            ref1[t][p] = (float)a1; ref2[t][p] = (float)a2;
        }
    u32x4* dx; float* dy;
    static float hy[128];
    hipMalloc(&dx, sizeof hx); hipMalloc(&dy, sizeof hy);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    for (int wc = 0; wc < 4; ++wc) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, wc);
        hipMemcpy(hy, dy, sizeof hy, hipMemcpyDeviceToHost);
        double e1 = 0, e2 = 0;
        for (int t = 0; t < 64; ++t) { e1 = fmax(e1, fabs(hy[2 * t] - ref1[t][wc])); e2 = fmax(e2, fabs(hy[2 * t + 1] - ref2[t][wc])); }
        printf("wc %d: chain of 8 words: max |sum err| %g  max |sumsq err| %g   (lane 0: %g %g, want %g %g)\n", wc, e1, e2, hy[0], hy[1], ref1[0][wc], ref2[0][wc]);
    }
    return 0;
}

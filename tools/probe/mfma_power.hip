// Sustained rate of a bare MFMA stream on every CU, two instruction shapes, random vs zero operands (tools/mfma_power.py): how much of
// the 2.5 PFLOP/s spec the chip holds under its power limit when nothing but the matrix pipe runs, and whether the 32x32x16 shape
// (half the operand-register reads per FLOP) holds more than the 16x16x32 one the GEMM kernels use.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_stream(const uint32_t* __restrict__ seed, float* __restrict__ sink, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t x[4], y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = seed[(t * 8 + i * 2 + 0) * 4 + e]; y[e] = seed[(t * 8 + i * 2 + 1) * 4 + e]; }
        a[i] = __builtin_bit_cast(half8, *reinterpret_cast<uint4*>(x));
        b[i] = __builtin_bit_cast(half8, *reinterpret_cast<uint4*>(y));
    }
    if constexpr (SHAPE == 16) {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 12345.678f) sink[t] = s;
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)      // 8 MFMAs of 32x32x16 = the FLOPs of 16 of 16x16x32
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[i], acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][e];
        if (s == 12345.678f) sink[t] = s;
    }
}

extern "C" int mfma_power_launch(int shape, const void* seed, void* sink, int blocks, int iters, void* stream) {
    if (shape == 16) hipLaunchKernelGGL((mfma_stream<16>), dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const uint32_t*)seed, (float*)sink, iters);
    else hipLaunchKernelGGL((mfma_stream<32>), dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const uint32_t*)seed, (float*)sink, iters);
    return (int)hipGetLastError();
}

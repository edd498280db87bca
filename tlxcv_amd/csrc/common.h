// Internal helpers shared by the libtlxmi.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>
#include "../../include/tlxmi.h"

namespace tlxmi {

// ---- error plumbing -----------------------------------------------------------------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

#define TLXMI_REQUIRE(cond, code, ...)            \
    do {                                          \
        if (!(cond)) return ::tlxmi::fail((code), __VA_ARGS__); \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t elt_size(int dtype) { return dtype == TLXMI_F16 ? 2 : 4; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device-side vector types -------------------------------------------------------------
typedef _Float16 half_t;
typedef half_t half2v __attribute__((ext_vector_type(2)));
typedef half_t half4v __attribute__((ext_vector_type(4)));
typedef half_t half8v __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- activation (fp32 in / fp32 out) ------------------------------------------------------
__device__ __forceinline__ float apply_act(float x, int act, float p) {
    switch (act) {
        case TLXMI_ACT_RELU: return fmaxf(x, 0.f);
        case TLXMI_ACT_RELU6: return fminf(fmaxf(x, 0.f), 6.f);
        case TLXMI_ACT_LEAKY: return x >= 0.f ? x : x * p;
        case TLXMI_ACT_HARDSWISH: return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
        case TLXMI_ACT_HARDSIGMOID: return fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
        case TLXMI_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
        case TLXMI_ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
        case TLXMI_ACT_SILU: return x / (1.f + __expf(-x));
        default: return x;
    }
}

// GELU for the fp16 throughput path: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far inside
// fp16 resolution) = one v_rcp + one v_exp + a 5-term Horner chain instead of libm's erff.
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(1.f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.f - poly * __expf(-z * z);
    const float erf = x < 0.f ? -erf_abs : erf_abs;
    return 0.5f * x * (1.f + erf);
}

// ---- wave64 reductions --------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// element conversion helpers
template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

}  // namespace tlxmi

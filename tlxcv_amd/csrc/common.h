// Internal helpers shared by the libtlxmi.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>
#include <stdlib.h>
#include "../../include/tlxmi.h"

namespace tlxmi {

// ---- error plumbing -----------------------------------------------------------------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);
int device_cus();                                                   // CU count of the CURRENT device (cached per device)
int raise_lds_limit(const void* fn, int bytes, const char* who);    // hipFuncSetAttribute once per (device, kernel)

#define TLXMI_REQUIRE(cond, code, ...)            \
    do {                                          \
        if (!(cond)) return ::tlxmi::fail((code), __VA_ARGS__); \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t elt_size(int dtype) { return dtype == TLXMI_F16 ? 2 : 4; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- tuning knobs ------------------------------------------------------------------------------
// The product library (libtlxmi.so) reads NO environment variable: every A/B knob below is its compiled-in default and
// the kernels carry no ablation branch.  The tuning flavour (libtlxmi_tune.so, `make tune`, -DTLXMI_TUNING) reads them
// per call — tools/ab_*.py and the tests that force a tile candidate load that flavour (tlxcv_amd._lib.tuning()).
#ifdef TLXMI_TUNING
inline long tune_int(const char* name, long dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atol(e) : dflt;
}
#define TLXMI_DBG(args, bit) (((args).debug & (bit)) != 0)
// Output store policy.  Product: every GEMM / convolution output is written back through L2 (plain stores) — the next launch
// reads it at once from L2 / the Infinity Cache, and the 64-byte halves of a line that two waves store phases apart merge
// before they leave.  Round 3 measured it again on the hipGraph replay of the whole two-stream forwards: write-back instead
// of non-temporal for gemm_pp (plain GEMM), gemm256 and gemm_stream: Swin-B 7.86 -> 7.64 ms, ResNet-50 3.51 -> 3.48, ViT-B/16
// 11.35 -> 11.30 (the round-1 per-layer sweep that chose non-temporal timed each launch alone, re-writing one buffer).
// A/B (tuning flavour): TLXMI_NT_STORES = TLXMI_DEBUG bit 0x4000 flips gemm_pp (plain GEMM) / gemm256 to NON-TEMPORAL stores (the
// macro is named for what the set bit selects; the product compiles it to false = write-back), bit 4 does the same for gemm_stream,
// bit 0x8000 conv_halo.
#define TLXMI_NT_STORES(args) (((args).debug & 0x4000) != 0)
#else
constexpr long tune_int(const char*, long dflt) { return dflt; }
#define TLXMI_DBG(args, bit) (false)
#define TLXMI_NT_STORES(args) (false)
#endif

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
        case TLXMI_ACT_SILU: return x * __builtin_amdgcn_rcpf(1.f + __expf(-x));     // rcp: 1 ulp, no division sequence
        default: return x;
    }
}

// Compile-time activation (the epilogues dispatch on `act` ONCE and run a specialised row loop: a runtime
// switch per element bloats the unrolled epilogue to ~18k instructions and starves the instruction fetch).
template <int ACT> __device__ __forceinline__ float apply_act_t(float x, float p) {
    if constexpr (ACT == TLXMI_ACT_RELU) return fmaxf(x, 0.f);
    else if constexpr (ACT == TLXMI_ACT_RELU6) return fminf(fmaxf(x, 0.f), 6.f);
    else if constexpr (ACT == TLXMI_ACT_LEAKY) return x >= 0.f ? x : x * p;
    else if constexpr (ACT == TLXMI_ACT_HARDSWISH) return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
    else if constexpr (ACT == TLXMI_ACT_HARDSIGMOID) return fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
    else if constexpr (ACT == TLXMI_ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
    else if constexpr (ACT == TLXMI_ACT_SIGMOID) return 1.f / (1.f + __expf(-x));
    else if constexpr (ACT == TLXMI_ACT_SILU) return x * __builtin_amdgcn_rcpf(1.f + __expf(-x));
    else return x;
}
template <int V> struct IntTag { static constexpr int value = V; };

// The lane index, recomputed where it is used (2 VALU).  A lane constant derived once at kernel entry (lane & 15, lane >> 4, an
// LDS offset built from them) stays live across a persistent kernel's whole K-tile stream; at the 256-register budget of two
// waves per SIMD hipcc then spills it, and the reload is a scratch (vector-memory) load whose wait is `s_waitcnt vmcnt(0)` —
// it drains every LDS-DMA in flight (gemm_stream's residual variants did that three times per output tile).  `volatile`
// keeps the compiler from hoisting / merging the two instructions back into one long-lived value.
__device__ __forceinline__ int lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// calls f(IntTag<act>{}) for the runtime activation code
#define TLXMI_DISPATCH_ACT(act, f)                          \
    switch (act) {                                          \
        case TLXMI_ACT_RELU: f(::tlxmi::IntTag<TLXMI_ACT_RELU>{}); break;               \
        case TLXMI_ACT_RELU6: f(::tlxmi::IntTag<TLXMI_ACT_RELU6>{}); break;             \
        case TLXMI_ACT_LEAKY: f(::tlxmi::IntTag<TLXMI_ACT_LEAKY>{}); break;             \
        case TLXMI_ACT_HARDSWISH: f(::tlxmi::IntTag<TLXMI_ACT_HARDSWISH>{}); break;     \
        case TLXMI_ACT_HARDSIGMOID: f(::tlxmi::IntTag<TLXMI_ACT_HARDSIGMOID>{}); break; \
        case TLXMI_ACT_GELU: f(::tlxmi::IntTag<TLXMI_ACT_GELU>{}); break;               \
        case TLXMI_ACT_SIGMOID: f(::tlxmi::IntTag<TLXMI_ACT_SIGMOID>{}); break;         \
        case TLXMI_ACT_SILU: f(::tlxmi::IntTag<TLXMI_ACT_SILU>{}); break;               \
        default: f(::tlxmi::IntTag<TLXMI_ACT_NONE>{}); break;                           \
    }

// GELU for the fp16 throughput path, two elements at a time so that hipcc emits packed fp32 math
// (v_pk_mul_f32 / v_pk_fma_f32):  gelu(x) = x * Phi(x),  Phi(x) = 0.5 + 0.5 * erf(x / sqrt 2) ~ 0.5 + xc * Q(xc^2) with
// xc = x clamped to +-3*sqrt(2) (one v_med3_f32; Phi is within 1.1e-5 of 0 / 1 beyond) and Q an even degree-16 minimax fit
// (LP fit on [0, 3*sqrt 2], 9 coefficients): max |Phi error| 1.35e-5 in fp32 arithmetic over all x, i.e. a GELU error of
// at most 1.35e-5 * |x| — under a twentieth of an fp16 ulp of the result for x > 0, under 6e-5 absolute for |x| <= 4.24.  6.5 VALU
// instructions an element, no transcendental (quarter-rate) instruction, no select.  The fp32 parity path keeps libm's erff.
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2v gelu_fast2(f32x2v x) {
    constexpr float XC = 4.24264069f;
    const f32x2v xc = f32x2v{__builtin_amdgcn_fmed3f(x[0], -XC, XC), __builtin_amdgcn_fmed3f(x[1], -XC, XC)};
    const f32x2v u = xc * xc;
    f32x2v q = u * 5.623591772e-11f + (-5.369481948e-09f);
    q = q * u + 2.267564292e-07f;
    q = q * u + (-5.645043615e-06f);
    q = q * u + 9.358027301e-05f;
    q = q * u + (-1.109351645e-03f);
    q = q * u + 9.818016454e-03f;
    q = q * u + (-6.634687996e-02f);
    q = q * u + 3.989031715e-01f;
    const f32x2v phi = xc * q + 0.5f;
    return x * phi;
}

// Four per-lane quantities a, b, c, d, each to be summed over the four lanes px, px + 16, px + 32, px + 48 that hold the 8-channel parts of
// one row (the MFMA accumulator layout of the 256 x 256 GEMM kernels): v_permlane16_swap(A, B) trades A's odd 16-lane rows with B's even
// ones, so A + B afterwards is [A0 + A1, B0 + B1, A2 + A3, B2 + B3] by lane row — one swap and one add take TWO quantities one level up —
// and v_permlane32_swap joins the halves: the lane in row g returns the total of the g-th quantity (a, b, c, d for g = 0..3).
// (The swap builtins return a two-element vector; its elements are copied to scalars before any bit cast: hipcc of ROCm 7.2 reads
// element 0 for every __builtin_bit_cast(T, vec[i]) on a vector-element lvalue, DESIGN 5.2 pitfall 1.)
__device__ __forceinline__ float ln_row_tree(float a, float b, float c, float d) {
    const auto r01 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned a01 = r01[0], b01 = r01[1];
    const float t01 = __builtin_bit_cast(float, a01) + __builtin_bit_cast(float, b01);
    const auto r23 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d), false, false);
    const unsigned c23 = r23[0], d23 = r23[1];
    const float t23 = __builtin_bit_cast(float, c23) + __builtin_bit_cast(float, d23);
    const auto r4 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, t01), __builtin_bit_cast(unsigned, t23), false, false);
    const unsigned lo = r4[0], hi = r4[1];
    return __builtin_bit_cast(float, lo) + __builtin_bit_cast(float, hi);
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

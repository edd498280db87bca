// Swin's patch embedding in one pass (reference swin_transformer.py:471-505: PatchEmbed = Conv2d(3 -> D, kernel 4, stride 4) ->
// flatten -> transpose -> LayerNorm(D)): the NCHW image in, the (B * H/4 * W/4, D) fp16 token rows out.
//
// Why a kernel of its own: per token the conv is 48 inputs x D outputs — 1.2 GFLOP per 64 images, nothing — and the layer is pure
// HBM traffic: 38.5 MB of fp32 image in, 51 MB of tokens out per 64 images.  As three launches (space-to-depth copy, implicit GEMM
// with K = 48, LayerNorm) it moved 38.5 + 19 | 19 + 51 | 51 + 51 MB and took 33 + 85 + 27 us per half batch of Swin-B (r04 trace);
// here the image is read once by the lanes that feed the MFMA B operand directly, the D channels of a token stay in the
// accumulators through bias + LayerNorm, and the tokens are written once.
//
// A wave owns 16 consecutive tokens (MFMA 16x16x32, D^T = W . X^T as in the GEMM kernels: lane (px = lane & 15, g = lane >> 4)
// ends up with channels 32 j + 8 g .. + 7 of token px for j = 0 .. D/32 - 1: 16-byte stores).  K = 64: k = 16 c + 4 ky + kx for
// k < 48 (c = colour plane, (ky, kx) inside the 4 x 4 patch), zero above.  B operand of K block kb: lane (px, g) needs k = 32 kb +
// 8 g .. + 7 = rows ky = 2 (g & 1), + 1 of plane c = 2 kb + (g >> 1), 4 pixels each: two 16-byte loads, contiguous over the 16
// tokens of the wave (256 B of an image row).  The filter (D x 64 fp16, channel rows permuted so that a lane's two tiles 2j, 2j + 1
// hold 8 consecutive channels) stays in registers for the life of the wave; waves walk the token tiles with the next tile's
// loads in flight.
#include "common.h"

namespace tlxmi {

template <typename TS, int NT, bool NORM>
__global__ __launch_bounds__(256) void patch_embed4_kernel(const TS* __restrict__ x, const half_t* __restrict__ w, const float* __restrict__ bias,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, half_t* __restrict__ y,
                                                           int H, int W, long tokens, float eps, const float* __restrict__ pos) {
    constexpr int D = 16 * NT;
    const int lane = threadIdx.x & 63, px = lane & 15, g = lane >> 4;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const int Ho = H >> 2, Wo = W >> 2;
    const long ntiles = (tokens + 15) >> 4;
    if (wave >= ntiles) return;

    // filter fragments: tile t, row i = lane & 15 -> channel 32 (t >> 1) + 8 (i >> 2) + 4 (t & 1) + (i & 3); K block kb, chunk g
    half8v wf[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ch = 32 * (t >> 1) + 8 * (px >> 2) + 4 * (t & 1) + (px & 3);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) wf[t][kb] = *reinterpret_cast<const half8v*>(w + ch * 64 + 32 * kb + 8 * g);
    }
    // per-channel constants of this lane's outputs: acc[t][r] is channel 32 (t >> 1) + 8 g + 4 (t & 1) + r
    float bs[NT][4], gm[NORM ? NT : 1][4], bt[NORM ? NT : 1][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c0 = 32 * (t >> 1) + 8 * g + 4 * (t & 1);
        const f32x4 b4 = bias ? *reinterpret_cast<const f32x4*>(bias + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[t][r] = b4[r];
        if constexpr (NORM) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + c0), e4 = *reinterpret_cast<const f32x4*>(beta + c0);
#pragma unroll
            for (int r = 0; r < 4; ++r) { gm[t][r] = g4[r]; bt[t][r] = e4[r]; }
        }
    }

    // the two K blocks of a tile: plane c = 2 kb + (g >> 1), rows 2 (g & 1), + 1 of the token's patch; kb = 1, g >= 2 is the zero pad
    static_assert(sizeof(TS) == 4 || sizeof(TS) == 2, "fp32 or fp16 image");
    struct Raw { f32x4 v[2][2]; };
    auto load_tile = [&](long tile, Raw& r) {
        const long p = tile * 16 + px;
        const bool ok = p < tokens;
        const unsigned HoWo = (unsigned)(Ho * Wo);
        const unsigned pu = ok ? (unsigned)p : 0u, n = pu / HoWo, q = pu - n * HoWo;      // (tokens < 2^31: 32-bit divisions)
        const unsigned oy = q / (unsigned)Wo, ox = q - oy * (unsigned)Wo;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int c = 2 * kb + (g >> 1);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                r.v[kb][rr] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok && c < 3) {
                    const TS* sp = x + (((long)n * 3 + c) * H + 4 * oy + 2 * (g & 1) + rr) * (long)W + 4 * ox;
                    if constexpr (sizeof(TS) == 4) {
                        r.v[kb][rr] = *reinterpret_cast<const f32x4*>(sp);
                    } else {
                        typedef half_t half4v_ __attribute__((ext_vector_type(4)));
                        const half4v_ h = *reinterpret_cast<const half4v_*>(sp);
                        r.v[kb][rr] = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                    }
                }
            }
        }
    };
    auto process = [&](long tile, const Raw& r) {
        half8v xb[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xb[kb][e] = (half_t)r.v[kb][0][e];
                xb[kb][4 + e] = (half_t)r.v[kb][1][e];
            }
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][0], xb[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][1], xb[1], acc[t], 0, 0, 0);
        }
        float mean = 0.f, rstd = 1.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r_ = 0; r_ < 4; ++r_) acc[t][r_] += bs[t][r_];
        if constexpr (NORM) {
            // LayerNorm over the D channels of token px: this lane holds D / 4 of them, lanes px + 16 g' the rest
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r_ = 0; r_ < 4; ++r_) s += acc[t][r_];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            mean = s * (1.f / (float)D);
            float sq = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r_ = 0; r_ < 4; ++r_) {
                    const float d = acc[t][r_] - mean;
                    sq += d * d;
                }
            sq += __shfl_xor(sq, 16, 64);
            sq += __shfl_xor(sq, 32, 64);
            rstd = 1.f / sqrtf(sq * (1.f / (float)D) + eps);
        }
        const long p = tile * 16 + px;
        if (p < tokens) {
            // absolute position embedding (swin_transformer.py:561-565, 603-604: x + absolute_pos_embed after the norm): row p % (Ho * Wo) of pos
            const float* prow = pos ? pos + (size_t)(p % ((long)Ho * Wo)) * D + 8 * g : nullptr;
#pragma unroll
            for (int j = 0; j < NT / 2; ++j) {
                half8v o;
                f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
                if (prow) {
                    p0 = *reinterpret_cast<const f32x4*>(prow + 32 * j);
                    p1 = *reinterpret_cast<const f32x4*>(prow + 32 * j + 4);
                }
#pragma unroll
                for (int r_ = 0; r_ < 4; ++r_) {
                    float a0 = acc[2 * j][r_], a1 = acc[2 * j + 1][r_];
                    if constexpr (NORM) {
                        a0 = (a0 - mean) * rstd * gm[2 * j][r_] + bt[2 * j][r_];
                        a1 = (a1 - mean) * rstd * gm[2 * j + 1][r_] + bt[2 * j + 1][r_];
                    }
                    o[r_] = (half_t)(a0 + p0[r_]);
                    o[4 + r_] = (half_t)(a1 + p1[r_]);
                }
                *reinterpret_cast<half8v*>(y + p * D + 32 * j + 8 * g) = o;
            }
        }
    };

    Raw ra, rb;
    long tile = wave;
    load_tile(tile, ra);
    while (true) {      // two tiles per trip: both buffers keep static names
        const long t1 = tile + nwaves;
        if (t1 < ntiles) load_tile(t1, rb);
        process(tile, ra);
        if (t1 >= ntiles) break;
        tile = t1 + nwaves;
        if (tile < ntiles) load_tile(tile, ra);
        process(t1, rb);
        if (tile >= ntiles) break;
    }
}

}  // namespace tlxmi

using namespace tlxmi;

// x: [N][3][H][W] fp32 or fp16 (16-byte aligned, W % 4 == 0); w: [D][64] fp16, k = 16 c + 4 ky + kx for k < 48, zero above (the
// caller's re-indexing of the conv filter [D][3][4][4]); bias / gamma / beta: fp32 [D] or null (gamma == null: no LayerNorm);
// y: [N * H/4 * W/4][D] fp16.  D in {96, 128, 192, 256}.
static int patch_embed4_impl(const void* x, int xdt, const void* w, const float* bias, const float* gamma, const float* beta, const float* pos,
                             void* y, int N, int H, int W, int D, float eps, void* stream) {
    TLXMI_REQUIRE(x && w && y && aligned16(x) && aligned16(w) && aligned16(y) && aligned16(bias) && aligned16(gamma) && aligned16(beta) && aligned16(pos),
                  TLXMI_ERR_BAD_ARG, "patch_embed4: null or misaligned buffer");
    TLXMI_REQUIRE(xdt == TLXMI_F32 || xdt == TLXMI_F16, TLXMI_ERR_BAD_ARG, "patch_embed4: bad dtype");
    TLXMI_REQUIRE(N > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0 && (gamma == nullptr) == (beta == nullptr), TLXMI_ERR_BAD_ARG,
                  "patch_embed4: H=%d W=%d must be multiples of 4", H, W);
    TLXMI_REQUIRE(D == 96 || D == 128 || D == 192 || D == 256, TLXMI_ERR_UNSUPPORTED, "patch_embed4: D=%d (96, 128, 192, 256)", D);
    const long tokens = (long)N * (H / 4) * (W / 4);
    TLXMI_REQUIRE(tokens < (1l << 31) && (long)N * 3 * H * W < (1l << 40), TLXMI_ERR_UNSUPPORTED, "patch_embed4: too many tokens");
    const long ntiles = (tokens + 15) / 16;
    long grid = (ntiles + 3) / 4;
    const long cap = (long)device_cus() * 8;
    if (grid > 2 * cap) grid = cap;
    hipStream_t st = as_stream(stream);
#define PE_LAUNCH(TS, NT)                                                                                                                  \
    {                                                                                                                                      \
        if (gamma) hipLaunchKernelGGL((patch_embed4_kernel<TS, NT, true>), dim3((unsigned)grid), dim3(256), 0, st, (const TS*)x, (const half_t*)w, \
                                      bias, gamma, beta, (half_t*)y, H, W, tokens, eps, pos);                                              \
        else hipLaunchKernelGGL((patch_embed4_kernel<TS, NT, false>), dim3((unsigned)grid), dim3(256), 0, st, (const TS*)x, (const half_t*)w,      \
                                bias, gamma, beta, (half_t*)y, H, W, tokens, eps, pos);                                                    \
    }
#define PE_DT(NT)                                \
    {                                            \
        if (xdt == TLXMI_F32) PE_LAUNCH(float, NT) \
        else PE_LAUNCH(half_t, NT)               \
    }
    if (D == 96) PE_DT(6)
    else if (D == 128) PE_DT(8)
    else if (D == 192) PE_DT(12)
    else PE_DT(16)
#undef PE_DT
#undef PE_LAUNCH
    return check_launch("patch_embed4");
}

extern "C" int tlxmi_patch_embed4(const void* x, int xdt, const void* w, const float* bias, const float* gamma, const float* beta,
                                  void* y, int N, int H, int W, int D, float eps, void* stream) {
    return patch_embed4_impl(x, xdt, w, bias, gamma, beta, nullptr, y, N, H, W, D, eps, stream);
}

// + the absolute position embedding of SwinTransformer(ape=True) (swin_transformer.py:561-565, 603-604): pos fp32 [H/4 * W/4][D], added
// to every image's tokens after the LayerNorm, before the one rounding to fp16.
extern "C" int tlxmi_patch_embed4_pos(const void* x, int xdt, const void* w, const float* bias, const float* gamma, const float* beta,
                                      const float* pos, void* y, int N, int H, int W, int D, float eps, void* stream) {
    TLXMI_REQUIRE(pos, TLXMI_ERR_BAD_ARG, "patch_embed4_pos: null position table");
    return patch_embed4_impl(x, xdt, w, bias, gamma, beta, pos, y, N, H, W, D, eps, stream);
}

// LayerNorm over the last dimension — HBM-bound row scan, one wave64 per row, the row held in
// registers between the two statistics passes (mean, then centred variance: the same two-pass
// form torch's CPU kernel uses, so fp32 results agree to rounding), reductions by wave shuffles.
// Reference: nn.LayerNorm(dim, epsilon=...) vision_transformer.py:144,159,283 (eps 1e-6 for
// ViT-B :355), swin_transformer.py:258,279,371,495,591.
#include "common.h"

namespace tlxmi {

// NCH = 16-byte chunks per lane and row, RW = rows per lane group, LPR = lanes per row (16/32/64: narrow
// rows such as Swin's C=128 use a quarter wave each, so all 64 lanes stay busy)
template <int LPR> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Swin window map (swin_transformer.py:85-116, 317-333): image row r = (b, hs, wsft) in unshifted coordinates
// sits at window row ((b*nW + (h/ws)*nWw + w/ws)*ws*ws + (h%ws)*ws + w%ws) with (h, w) = (hs, wsft) rolled by
// -shift.  mode 0: none; 1: the OUTPUT row is the window row (LayerNorm + roll + window_partition in one pass);
// 2: the INPUT row is the window row, `res` (image order) is added, the sum is written to `sum_out` (image
// order) and normalised into y (window_reverse + roll back + residual + LayerNorm in one pass).
struct WinMap { int mode, H, W, ws, shift; };
__device__ __forceinline__ long win_row(const WinMap& wm, long r) {
    const int HW = wm.H * wm.W;
    // rows < 2^31 (checked by the entry points): one 32-bit division instead of the ~100-instruction 64-bit sequence
    const unsigned ub = (unsigned)r / (unsigned)HW;
    const long b = (long)ub;
    const int p = (int)((unsigned)r - ub * (unsigned)HW);
    const int hs = (int)(((float)p + 0.5f) * (1.0f / (float)wm.W)), wsft = p - hs * wm.W;
    int h = hs - wm.shift, w = wsft - wm.shift;
    if (h < 0) h += wm.H;
    if (w < 0) w += wm.W;
    const int nWw = wm.W / wm.ws, nW = (wm.H / wm.ws) * nWw;
    const int wh = (int)(((float)h + 0.5f) * (1.0f / (float)wm.ws)), ww = (int)(((float)w + 0.5f) * (1.0f / (float)wm.ws));
    return (b * nW + wh * nWw + ww) * (long)(wm.ws * wm.ws) + (h - wh * wm.ws) * wm.ws + (w - ww * wm.ws);
}

// PERSIST: the grid is a few workgroups per CU and a lane group walks its row groups with the NEXT group's loads in
// flight while it reduces / normalises / stores the current one (two register buffers).  One-shot workgroups serialise
// load latency -> arithmetic -> store per wave and pay a workgroup launch per 32 rows: 3.2 TB/s on ViT's 50432 x 768;
template <typename T, int NCH, int RW, int LPR, bool PERSIST, bool HASRES>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y, long rows,
                                                        int C, int x_ld, int y_ld, float eps, float* __restrict__ stats,
                                                        const WinMap wm, const T* __restrict__ res, T* __restrict__ sum_out) {
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int GPW = 64 / LPR;                    // row groups per wave
    const int lane = threadIdx.x & (LPR - 1);
    const long gstride = (long)gridDim.x * 4 * GPW * RW;       // rows between two groups of this lane group (PERSIST)
    long row0 = (((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * GPW + ((threadIdx.x & 63) / LPR)) * RW;
    if (row0 >= rows) return;
    const int nch = C / V;
    // gamma / beta of this lane's channels: once per wave, not once per row
    // (whole 16-byte loads: one dword load per element made the gamma / beta fetch of a short-lived workgroup four times the
    //  instructions of its row loads — 12544 x 512: 16.9 us against 4.2 us for a copy of the same bytes)
    float gm[NCH][V], bt[NCH][V];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = lane + LPR * i;
        const int c0 = ch < nch ? ch * V : 0;
#pragma unroll
        for (int q = 0; q < V / 4; ++q) {
            const f32x4 g4 = gamma ? *reinterpret_cast<const f32x4*>(gamma + c0 + 4 * q) : f32x4{1.f, 1.f, 1.f, 1.f};
            const f32x4 b4 = beta ? *reinterpret_cast<const f32x4*>(beta + c0 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gm[i][4 * q + e] = g4[e];
                bt[i][4 * q + e] = b4[e];
            }
        }
    }
    // all RW x NCH 16-byte loads of a group are issued before the first reduction: the kernel is a
    // pure HBM stream and needs the bytes in flight, not the arithmetic
    constexpr int RR = HASRES ? RW : 1, RN = HASRES ? NCH : 1;     // the residual buffers exist only in mode 2
    auto load_group = [&](long r0, u32x4 (&raw)[RW][NCH], u32x4 (&rsd)[RR][RN], long (&wrow)[RW]) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            wrow[r] = ((wm.mode == 1 || wm.mode == 2) && r0 + r < rows) ? win_row(wm, r0 + r) : 0;
            const long srow = wm.mode == 2 ? wrow[r] : r0 + r;
            long pix = 0;      // mode 3: the (0, 0) source pixel of merged row r0 + r
            if (wm.mode == 3 && r0 + r < rows) {
                const unsigned Wo = (unsigned)wm.W >> 1, HoWo = ((unsigned)wm.H >> 1) * Wo;
                const unsigned ub = (unsigned)(r0 + r) / HoWo, p = (unsigned)(r0 + r) - ub * HoWo, ho = p / Wo, wo = p - ho * Wo;
                pix = ((long)ub * wm.H + 2 * ho) * wm.W + 2 * wo;
            }
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = lane + LPR * i;
                raw[r][i] = u32x4{0u, 0u, 0u, 0u};
                if constexpr (HASRES) rsd[r][i] = u32x4{0u, 0u, 0u, 0u};
                if (ch < nch && r0 + r < rows) {
                    if (wm.mode == 3) {      // PatchMerging gather: chunk -> block q = (dh, dw) = (q & 1, q >> 1) of wm.ws chunks
                        const int q = (ch >= wm.ws) + (ch >= 2 * wm.ws) + (ch >= 3 * wm.ws);
                        raw[r][i] = *reinterpret_cast<const u32x4*>(x + (pix + (q & 1) * wm.W + (q >> 1)) * x_ld + (ch - q * wm.ws) * V);
                        continue;
                    }
                    raw[r][i] = *reinterpret_cast<const u32x4*>(x + srow * x_ld + ch * V);
                    if constexpr (HASRES) rsd[r][i] = *reinterpret_cast<const u32x4*>(res + (r0 + r) * x_ld + ch * V);
                }
            }
        }
    };
    auto process = [&](long row0, u32x4 (&raw)[RW][NCH], const u32x4 (&rsd)[RR][RN], const long (&wrow)[RW]) {
    if constexpr (HASRES) {   // + residual (image order), rounded to T as the two-kernel path stores it, written out
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int ch = lane + LPR * i;
                if (ch < nch && row0 + r < rows) {
                    const u32x4 rv = rsd[r][i];
                    if constexpr (sizeof(T) == 2) {
                        const half8v a = __builtin_bit_cast(half8v, raw[r][i]), b = __builtin_bit_cast(half8v, rv);
                        half8v o;
#pragma unroll
                        for (int e = 0; e < V; ++e) o[e] = (half_t)((float)a[e] + (float)b[e]);
                        raw[r][i] = __builtin_bit_cast(u32x4, o);
                    } else {
                        raw[r][i] = __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, raw[r][i]) + __builtin_bit_cast(f32x4, rv));
                    }
                    *reinterpret_cast<u32x4*>(sum_out + (row0 + r) * x_ld + ch * V) = raw[r][i];
                }
            }
    }
    // three phases over all RW rows at once (the RW reduction chains are independent, so their cross-lane
    // latencies overlap): sums -> means, centred squares -> rstd, normalise + store
    auto unpack = [&](int r, int i, float* f) {
        if constexpr (sizeof(T) == 2) {
            const half8v h = __builtin_bit_cast(half8v, raw[r][i]);
#pragma unroll
            for (int e = 0; e < V; ++e) f[e] = (float)h[e];
        } else {
            const f32x4 h = __builtin_bit_cast(f32x4, raw[r][i]);
#pragma unroll
            for (int e = 0; e < V; ++e) f[e] = h[e];
        }
    };
    float mean[RW], rstd[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            float f[V];
            unpack(r, i, f);
#pragma unroll
            for (int e = 0; e < V; ++e) sum += f[e];   // lanes past the row hold zeros
        }
        mean[r] = sum;
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) mean[r] = group_sum<LPR>(mean[r]) / (float)C;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (lane + LPR * i < nch) {
                float f[V];
                unpack(r, i, f);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float d = f[e] - mean[r];
                    sq += d * d;
                }
            }
        }
        rstd[r] = sq;
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) rstd[r] = 1.f / sqrtf(group_sum<LPR>(rstd[r]) / (float)C + eps);
    if (stats) {   // statistics only (internal; no entry point since round 4): (rstd, -mean * rstd) per row, nothing else is written
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RW; ++r)
                if (row0 + r < rows) *reinterpret_cast<float2*>(stats + 2 * (row0 + r)) = make_float2(rstd[r], -mean[r] * rstd[r]);
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        if (row0 + r >= rows) break;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ch = lane + LPR * i;
            if (ch < nch) {
                float f[V], o[V];
                unpack(r, i, f);
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = (f[e] - mean[r]) * rstd[r] * gm[i][e] + bt[i][e];
                if constexpr (sizeof(T) == 2) {
                    half8v h;
#pragma unroll
                    for (int e = 0; e < V; ++e) h[e] = (half_t)o[e];
                    *reinterpret_cast<half8v*>(y + (wm.mode == 1 ? wrow[r] : row0 + r) * y_ld + ch * V) = h;
                } else {
                    f32x4 h;
#pragma unroll
                    for (int e = 0; e < V; ++e) h[e] = o[e];
                    *reinterpret_cast<f32x4*>(y + (wm.mode == 1 ? wrow[r] : row0 + r) * y_ld + ch * V) = h;
                }
            }
        }
    }
    };
    u32x4 rawA[RW][NCH], rsdA[RR][RN];
    long wrowA[RW];
    load_group(row0, rawA, rsdA, wrowA);
    if constexpr (!PERSIST) {
        process(row0, rawA, rsdA, wrowA);
    } else {
        u32x4 rawB[RW][NCH], rsdB[RR][RN];
        long wrowB[RW];
        while (true) {      // two groups per trip so that both buffers keep static names (no runtime-indexed registers)
            const long r1 = row0 + gstride;
            if (r1 < rows) load_group(r1, rawB, rsdB, wrowB);
            process(row0, rawA, rsdA, wrowA);
            if (r1 >= rows) break;
            row0 = r1 + gstride;
            if (row0 < rows) load_group(row0, rawA, rsdA, wrowA);
            process(r1, rawB, rsdB, wrowB);
            if (row0 >= rows) break;
        }
    }
}

static int ln_cus() { return device_cus(); }      // core.hip: cached per device

// One instantiation: one-shot workgroups (a row group each), or a persistent grid of as many workgroups as are resident
// (occupancy query, once per instantiation) walking the row groups.
template <typename T, int NCH, int RW, int LPR, bool PERSIST, bool HASRES>
static void launch_ln_case(const void* x, const float* gamma, const float* beta, void* y, long rows, int C, int x_ld, int y_ld,
                           float eps, hipStream_t st, float* stats, const WinMap& wm, const void* res, void* sum_out) {
    auto* kern = &layernorm_kernel<T, NCH, RW, LPR, PERSIST, HASRES>;
    long grid = (rows + 4 * RW * (64 / LPR) - 1) / (4 * RW * (64 / LPR));
    if (PERSIST) {
        static int per_cu = 0;
        if (per_cu == 0) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 256, 0) != hipSuccess || nb < 1) {
                (void)hipGetLastError();
                nb = 2;
            }
            per_cu = nb > 8 ? 8 : nb;
        }
        const long cap = (long)ln_cus() * tune_int("TLXMI_LN_PERCU", per_cu);
        if (grid > tune_int("TLXMI_LN_TRIPS", 2) * cap) grid = cap;      // few row groups: one each (no two-trip tail)
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, rows, C, x_ld, y_ld, eps,
                       stats, wm, (const T*)res, (T*)sum_out);
}

template <typename T>
static int launch_ln(const void* x, const float* gamma, const float* beta, void* y, long rows, int C, int x_ld,
                     int y_ld, float eps, hipStream_t st, float* stats = nullptr, WinMap wm = WinMap{0, 1, 1, 1, 0},
                     const void* res = nullptr, void* sum_out = nullptr) {
    constexpr int V = 16 / (int)sizeof(T);
    const int nch = C / V;
    // tuning flavour: TLXMI_LN=1 keeps the one-shot workgroups of round 1 (A/B)
    const bool oneshot = tune_int("TLXMI_LN", 0) == 1;
#define LN_CASE(n, rw, lpr, pers)                                                                                              \
    {                                                                                                                          \
        if (res) launch_ln_case<T, n, rw, lpr, pers, true>(x, gamma, beta, y, rows, C, x_ld, y_ld, eps, st, stats, wm, res, sum_out); \
        else launch_ln_case<T, n, rw, lpr, pers, false>(x, gamma, beta, y, rows, C, x_ld, y_ld, eps, st, stats, wm, res, sum_out);   \
    }
    // persistent shapes: half the rows per lane group of the one-shot form (the second register buffer holds the next group)
    if (nch <= 16) { if (oneshot) LN_CASE(1, 4, 16, false) else LN_CASE(1, 2, 16, true) }
    else if (nch <= 32) { if (oneshot) LN_CASE(1, 4, 32, false) else LN_CASE(1, 2, 32, true) }
    else if (nch <= 64) { if (oneshot) LN_CASE(1, 8, 64, false) else LN_CASE(1, 4, 64, true) }
    else if (nch <= 128) { if (oneshot) LN_CASE(2, 8, 64, false) else LN_CASE(2, 2, 64, true) }      // ViT-B (96): 45 -> 35 us on 50432 rows
    else if (nch <= 256) LN_CASE(4, 2, 64, false)
    else if (nch <= 512) LN_CASE(8, 1, 64, false)
    else if (nch <= 1024) LN_CASE(16, 1, 64, false)
    else return fail(TLXMI_ERR_UNSUPPORTED, "layernorm: C=%d too wide for the in-register row kernel", C);
#undef LN_CASE
    return check_launch("layernorm");
}

}  // namespace tlxmi

using namespace tlxmi;

extern "C" int tlxmi_layernorm(const void* x, const float* gamma, const float* beta, void* y, int dt, int64_t rows,
                               int C, int x_ld, int y_ld, float eps, void* stream) {
    TLXMI_REQUIRE(x && y && rows > 0 && C > 0, TLXMI_ERR_BAD_ARG, "layernorm: bad argument");
    TLXMI_REQUIRE(dt == TLXMI_F16 || dt == TLXMI_F32, TLXMI_ERR_BAD_ARG, "layernorm: bad dtype");
    const int V = 16 / (int)elt_size(dt);
    TLXMI_REQUIRE(C % V == 0 && x_ld % V == 0 && y_ld % V == 0 && x_ld >= C && y_ld >= C && aligned16(x) && aligned16(y) &&
                      aligned16(gamma) && aligned16(beta),      // gamma / beta are fetched as whole 16-byte vectors
                  TLXMI_ERR_ALIGNMENT, "layernorm: C=%d / strides must be whole 16-byte chunks", C);
    if (dt == TLXMI_F16) return launch_ln<half_t>(x, gamma, beta, y, (long)rows, C, x_ld, y_ld, eps, as_stream(stream));
    return launch_ln<float>(x, gamma, beta, y, (long)rows, C, x_ld, y_ld, eps, as_stream(stream));
}

// LayerNorm fused with Swin's window plumbing (swin_transformer.py:315-335):
//   tlxmi_layernorm_window_partition: win = window_partition(roll(LN(x), -shift))      (norm1 + :316-324)
//   tlxmi_window_reverse_layernorm:   sum = res + roll(window_reverse(win), +shift);  y = LN(sum)   (:327-335 + norm2)
// x / res / sum / y are [B][H][W][C] dense, win is [B*nW][ws*ws][C].
static int check_ln_window(const char* name, int dt, int B, int H, int W, int C, int ws, int shift) {
    TLXMI_REQUIRE(dt == TLXMI_F16 || dt == TLXMI_F32, TLXMI_ERR_BAD_ARG, "%s: bad dtype", name);
    TLXMI_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && ws > 0 && H % ws == 0 && W % ws == 0, TLXMI_ERR_BAD_ARG,
                  "%s: H=%d W=%d must be multiples of the window %d", name, H, W, ws);
    TLXMI_REQUIRE(shift >= 0 && shift < ws, TLXMI_ERR_BAD_ARG, "%s: shift must be in [0, window)", name);
    TLXMI_REQUIRE(C % (16 / (int)elt_size(dt)) == 0, TLXMI_ERR_ALIGNMENT, "%s: C=%d must be whole 16-byte chunks", name, C);
    TLXMI_REQUIRE((long)H * W < (1l << 22), TLXMI_ERR_UNSUPPORTED, "%s: image too large for the float index math", name);
    TLXMI_REQUIRE((long)B * H * W < (1l << 31), TLXMI_ERR_UNSUPPORTED, "%s: more than 2^31 rows", name);
    return TLXMI_OK;
}

extern "C" int tlxmi_layernorm_window_partition(const void* x, const float* gamma, const float* beta, void* win, int dt,
                                                int B, int H, int W, int C, int ws, int shift, float eps, void* stream) {
    TLXMI_REQUIRE(x && win && aligned16(x) && aligned16(win) && aligned16(gamma) && aligned16(beta), TLXMI_ERR_BAD_ARG, "layernorm_window_partition: bad buffer");
    if (int e = check_ln_window("layernorm_window_partition", dt, B, H, W, C, ws, shift)) return e;
    const WinMap wm{1, H, W, ws, shift};
    const long rows = (long)B * H * W;
    if (dt == TLXMI_F16) return launch_ln<half_t>(x, gamma, beta, win, rows, C, C, C, eps, as_stream(stream), nullptr, wm);
    return launch_ln<float>(x, gamma, beta, win, rows, C, C, C, eps, as_stream(stream), nullptr, wm);
}

// PatchMerging's gather + LayerNorm(4C) in one pass (swin_transformer.py:381-388): y[B * H/2 * W/2][4C] = LN(cat(x0, x1, x2, x3))
extern "C" int tlxmi_patch_merge_layernorm(const void* x, const float* gamma, const float* beta, void* y, int dt, int B, int H,
                                           int W, int C, float eps, void* stream) {
    TLXMI_REQUIRE(x && y && aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta), TLXMI_ERR_BAD_ARG, "patch_merge_layernorm: bad buffer");
    TLXMI_REQUIRE(dt == TLXMI_F16 || dt == TLXMI_F32, TLXMI_ERR_BAD_ARG, "patch_merge_layernorm: bad dtype");
    TLXMI_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, TLXMI_ERR_BAD_ARG, "patch_merge_layernorm: H=%d W=%d must be even", H, W);
    const int V = 16 / (int)elt_size(dt);
    TLXMI_REQUIRE(C % V == 0, TLXMI_ERR_ALIGNMENT, "patch_merge_layernorm: C=%d must be whole 16-byte chunks", C);
    TLXMI_REQUIRE((long)B * H * W < (1l << 31), TLXMI_ERR_UNSUPPORTED, "patch_merge_layernorm: more than 2^31 rows");
    const WinMap wm{3, H, W, C / V, 0};
    const long rows = (long)B * (H / 2) * (W / 2);
    if (dt == TLXMI_F16) return launch_ln<half_t>(x, gamma, beta, y, rows, 4 * C, C, 4 * C, eps, as_stream(stream), nullptr, wm);
    return launch_ln<float>(x, gamma, beta, y, rows, 4 * C, C, 4 * C, eps, as_stream(stream), nullptr, wm);
}

extern "C" int tlxmi_window_reverse_layernorm(const void* win, const void* res, const float* gamma, const float* beta,
                                              void* sum, void* y, int dt, int B, int H, int W, int C, int ws, int shift,
                                              float eps, void* stream) {
    TLXMI_REQUIRE(win && res && sum && y && aligned16(win) && aligned16(res) && aligned16(sum) && aligned16(y) && aligned16(gamma) && aligned16(beta), TLXMI_ERR_BAD_ARG,
                  "window_reverse_layernorm: bad buffer");
    if (int e = check_ln_window("window_reverse_layernorm", dt, B, H, W, C, ws, shift)) return e;
    const WinMap wm{2, H, W, ws, shift};
    const long rows = (long)B * H * W;
    if (dt == TLXMI_F16) return launch_ln<half_t>(win, gamma, beta, y, rows, C, C, C, eps, as_stream(stream), nullptr, wm, res, sum);
    return launch_ln<float>(win, gamma, beta, y, rows, C, C, C, eps, as_stream(stream), nullptr, wm, res, sum);
}

// Row softmax over the last axis (tlx.ops.softmax / nn.Softmax of a reference forward run layer by layer, e.g. the attention of
// detr.py:1011-1043 or vision_transformer.py:118): one wave per row, the row read twice (maximum, then exp and sum kept in registers for
// rows of up to 64 * 8 elements, else a third read), fp32 arithmetic, the input's dtype out.  Any C >= 1, any row stride; a coverage
// path (the fused attention kernels never materialise their scores).
namespace tlxmi {
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, int C, long x_ld, long y_ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * x_ld;
    T* yr = y + row * y_ld;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, (float)xr[c]);
    mx = wave_max(mx);
    if (mx == -INFINITY) mx = 0.f;      // a row of -inf: torch gives NaN (0 / 0); keep that: exp(-inf - 0) = 0, sum 0, 0 / 0
    float e[8], sum = 0.f;
    const bool small = C <= 512;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        e[i] = (small && c < C) ? __expf((float)xr[c] - mx) : 0.f;
        sum += e[i];
    }
    if (!small)
        for (int c = lane; c < C; c += 64) sum += __expf((float)xr[c] - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    if (small) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            if (c < C) yr[c] = (T)(e[i] * inv);
        }
    } else {
        for (int c = lane; c < C; c += 64) yr[c] = (T)(__expf((float)xr[c] - mx) * inv);
    }
}
}  // namespace tlxmi

extern "C" int tlxmi_softmax_rows(const void* x, void* y, int dt, int64_t rows, int C, int64_t x_ld, int64_t y_ld, void* stream) {
    using namespace tlxmi;
    TLXMI_REQUIRE(x && y && rows > 0 && C > 0 && x_ld >= C && y_ld >= C, TLXMI_ERR_BAD_ARG, "softmax_rows: bad argument");
    TLXMI_REQUIRE(dt == TLXMI_F16 || dt == TLXMI_F32, TLXMI_ERR_BAD_ARG, "softmax_rows: bad dtype");
    TLXMI_REQUIRE((rows + 3) / 4 < (1ll << 31), TLXMI_ERR_UNSUPPORTED, "softmax_rows: too many rows");
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dt == TLXMI_F16) hipLaunchKernelGGL(softmax_rows_kernel<half_t>, grid, dim3(256), 0, as_stream(stream), (const half_t*)x, (half_t*)y, (long)rows, C, (long)x_ld, (long)y_ld);
    else hipLaunchKernelGGL(softmax_rows_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, (long)rows, C, (long)x_ld, (long)y_ld);
    return check_launch("softmax_rows");
}

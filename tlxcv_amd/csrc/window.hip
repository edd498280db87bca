// Swin window plumbing as pure index math on 16-byte channel chunks (HBM-bound copies).
// Reference: window_partition swin_transformer.py:85-99, window_reverse :102-116, cyclic roll
// :317-319 / :329-331, PatchMerging gather :381-386.
//   shifted[b][h][w] = x[b][(h+shift)%H][(w+shift)%W]          (tlx.roll(x, (-shift,-shift), (1,2)))
//   win[b*nW + (h/ws)*(W/ws) + (w/ws)][(h%ws)*ws + (w%ws)] = shifted[b][h][w]
// reverse is the inverse map, optionally fused with the residual add of :334.
#include "common.h"

namespace tlxmi {

template <typename T, bool REVERSE>
__global__ void window_kernel(const T* __restrict__ src, const T* __restrict__ res, T* __restrict__ dst, int B, int H,
                              int W, int C, int ws, int shift) {
    constexpr int V = 16 / (int)sizeof(T);
    const int nch = C / V;
    const int nWw = W / ws, nW = (H / ws) * nWw;
    const long total = (long)B * H * W * nch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % nch);
        long p = i / nch;
        const int w = (int)(p % W);
        p /= W;
        const int h = (int)(p % H);
        const long b = p / H;
        // (h,w) are coordinates in the shifted image
        int hs = h + shift, wsft = w + shift;
        if (hs >= H) hs -= H;
        if (wsft >= W) wsft -= W;
        const long img = ((b * H + hs) * W + wsft) * C + cg * V;                                     // unshifted
        const long win = ((b * nW + (h / ws) * nWw + (w / ws)) * (ws * ws) + (h % ws) * ws + (w % ws)) * C + cg * V;
        if constexpr (!REVERSE) {
            *reinterpret_cast<u32x4*>(dst + win) = *reinterpret_cast<const u32x4*>(src + img);
        } else {
            if (res) {
                if constexpr (sizeof(T) == 2) {
                    half8v a = *reinterpret_cast<const half8v*>(src + win);
                    half8v r = *reinterpret_cast<const half8v*>(res + img);
                    half8v o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)a[e] + (float)r[e]);
                    *reinterpret_cast<half8v*>(dst + img) = o;
                } else {
                    f32x4 a = *reinterpret_cast<const f32x4*>(src + win);
                    f32x4 r = *reinterpret_cast<const f32x4*>(res + img);
                    *reinterpret_cast<f32x4*>(dst + img) = a + r;
                }
            } else {
                *reinterpret_cast<u32x4*>(dst + img) = *reinterpret_cast<const u32x4*>(src + win);
            }
        }
    }
}

// x[B][H][W][C] -> y[B][H/2][W/2][4C], blocks (dh,dw) = (0,0),(1,0),(0,1),(1,1)
template <typename T>
__global__ void patch_merge_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
    constexpr int V = 16 / (int)sizeof(T);
    const int nch = C / V, Ho = H / 2, Wo = W / 2;
    const long total = (long)B * Ho * Wo * 4 * nch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % nch);
        long p = i / nch;
        const int blk = (int)(p & 3);
        p >>= 2;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho);
        const long b = p / Ho;
        const int dh = blk & 1, dw = blk >> 1;
        const long s = ((b * H + 2 * ho + dh) * W + 2 * wo + dw) * C + cg * V;
        const long d = (((b * Ho + ho) * Wo + wo) * 4 + blk) * C + cg * V;
        *reinterpret_cast<u32x4*>(y + d) = *reinterpret_cast<const u32x4*>(x + s);
    }
}

static inline int grid_for(long work) {
    long g = (work + 255) / 256;
    return (int)(g < 1 ? 1 : (g < 4096 ? g : 4096));
}

}  // namespace tlxmi

using namespace tlxmi;

static int check_window(const char* name, const void* a, const void* b, int dt, int B, int H, int W, int C, int ws,
                        int shift) {
    TLXMI_REQUIRE(a && b, TLXMI_ERR_BAD_ARG, "%s: null buffer", name);
    TLXMI_REQUIRE(dt == TLXMI_F16 || dt == TLXMI_F32, TLXMI_ERR_BAD_ARG, "%s: bad dtype", name);
    TLXMI_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && ws > 0 && H % ws == 0 && W % ws == 0, TLXMI_ERR_BAD_ARG,
                  "%s: H=%d W=%d must be multiples of the window %d", name, H, W, ws);
    TLXMI_REQUIRE(shift >= 0 && shift < ws, TLXMI_ERR_BAD_ARG, "%s: shift must be in [0, window)", name);
    TLXMI_REQUIRE(C % (16 / (int)elt_size(dt)) == 0 && aligned16(a) && aligned16(b), TLXMI_ERR_ALIGNMENT,
                  "%s: C=%d must be whole 16-byte chunks", name, C);
    return TLXMI_OK;
}

extern "C" int tlxmi_window_partition(const void* x, void* win, int dt, int B, int H, int W, int C, int ws, int shift,
                                      void* stream) {
    if (int e = check_window("window_partition", x, win, dt, B, H, W, C, ws, shift)) return e;
    const long work = (long)B * H * W * (C / (16 / (int)elt_size(dt)));
    dim3 g(grid_for(work)), b(256);
    if (dt == TLXMI_F16)
        hipLaunchKernelGGL((window_kernel<half_t, false>), g, b, 0, as_stream(stream), (const half_t*)x, (const half_t*)nullptr, (half_t*)win, B, H, W, C, ws, shift);
    else
        hipLaunchKernelGGL((window_kernel<float, false>), g, b, 0, as_stream(stream), (const float*)x, (const float*)nullptr, (float*)win, B, H, W, C, ws, shift);
    return check_launch("window_partition");
}

extern "C" int tlxmi_window_reverse(const void* win, const void* res, void* y, int dt, int B, int H, int W, int C,
                                    int ws, int shift, void* stream) {
    if (int e = check_window("window_reverse", win, y, dt, B, H, W, C, ws, shift)) return e;
    TLXMI_REQUIRE(!res || aligned16(res), TLXMI_ERR_ALIGNMENT, "window_reverse: residual must be 16-byte aligned");
    const long work = (long)B * H * W * (C / (16 / (int)elt_size(dt)));
    dim3 g(grid_for(work)), b(256);
    if (dt == TLXMI_F16)
        hipLaunchKernelGGL((window_kernel<half_t, true>), g, b, 0, as_stream(stream), (const half_t*)win, (const half_t*)res, (half_t*)y, B, H, W, C, ws, shift);
    else
        hipLaunchKernelGGL((window_kernel<float, true>), g, b, 0, as_stream(stream), (const float*)win, (const float*)res, (float*)y, B, H, W, C, ws, shift);
    return check_launch("window_reverse");
}

extern "C" int tlxmi_patch_merge_gather(const void* x, void* y, int dt, int B, int H, int W, int C, void* stream) {
    TLXMI_REQUIRE(x && y, TLXMI_ERR_BAD_ARG, "patch_merge: null buffer");
    TLXMI_REQUIRE(dt == TLXMI_F16 || dt == TLXMI_F32, TLXMI_ERR_BAD_ARG, "patch_merge: bad dtype");
    TLXMI_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0, TLXMI_ERR_BAD_ARG,
                  "patch_merge: x size (%d*%d) are not even", H, W);
    TLXMI_REQUIRE(C % (16 / (int)elt_size(dt)) == 0 && aligned16(x) && aligned16(y), TLXMI_ERR_ALIGNMENT,
                  "patch_merge: C=%d must be whole 16-byte chunks", C);
    const long work = (long)B * H * W * (C / (16 / (int)elt_size(dt)));
    dim3 g(grid_for(work)), b(256);
    if (dt == TLXMI_F16)
        hipLaunchKernelGGL((patch_merge_kernel<half_t>), g, b, 0, as_stream(stream), (const half_t*)x, (half_t*)y, B, H, W, C);
    else
        hipLaunchKernelGGL((patch_merge_kernel<float>), g, b, 0, as_stream(stream), (const float*)x, (float*)y, B, H, W, C);
    return check_launch("patch_merge");
}

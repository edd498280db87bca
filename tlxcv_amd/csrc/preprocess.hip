// Image pre-processing on the device: uint8 HWC images -> Resize -> Normalize -> ToTensor, written directly in the
// layout the network wants.  Replaces the host pipeline of the reference's inference scripts,
//     Compose([Resize((224, 224)), Normalize(mean, std), ToTensor(data_format)])     demo/image_classification/predict.py:22-29
// which runs PIL / numpy on one image at a time and cannot feed a 60 k img/s engine.
//
// Resize is Pillow's two-pass resampler restated (tlxcv_amd/tlx/vision/transforms/resample.py builds the tables from the
// published algorithm, checked bit for bit against PIL): per axis, output sample o = clip((2^21 + sum_t in[lo(o) + t] *
// k[o][t]) >> 22) in integer arithmetic, horizontal pass first, its uint8 result feeds the vertical pass — so the device
// result is BIT-IDENTICAL to the host pipeline, not merely close.  Normalize is (v - mean[c]) / std[c] in fp32 with IEEE
// subtraction and division (numpy's float32 arithmetic); without it the uint8 value is scaled by 1/255 (ToTensor's rule).
// Output layouts: NCHW ("CHW"), NHWC ("HWC"), or the b x b space-to-depth NHWC image of tlxmi_nchw_to_nhwc_s2d (channel
// (ph*b + pw)*C + c, zero padded to cpad) so the stem conv reads it without a layout pass.  HBM-bound byte work, no MFMA.
#include "common.h"

namespace tlxmi {

// horizontal pass: src [N][H][W][C] -> dst [N][H][OW][C]; one thread per output pixel
__global__ void resize_h_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int* __restrict__ bounds,
                                   const int* __restrict__ kk, long rows, int W, int OW, int C, int ksize) {
    const long total = rows * OW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % OW);
        const long row = i / OW;
        const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
        const int* k = kk + (long)xx * ksize;
        const uint8_t* sp = src + (row * W + x0) * C;
        for (int c = 0; c < C; ++c) {
            int acc = 1 << 21;
            for (int t = 0; t < n; ++t) acc += (int)sp[t * C + c] * k[t];
            acc >>= 22;
            dst[i * C + c] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
        }
    }
}

struct PreArgs {
    const uint8_t* src;       // [N][H][OW][C] (after the horizontal pass)
    const int* bounds;        // [OH][2]
    const int* kk;            // [OH][ksize]
    const float* mean;        // [C] or null
    const float* std_;        // [C] or null
    void* out;
    int N, H, OH, OW, C, ksize, layout, b, cpad, normalize;
};

__device__ __forceinline__ float pre_value(const PreArgs& a, long n, int yy, int x, int c) {
    const int y0 = a.bounds[2 * yy], cnt = a.bounds[2 * yy + 1];
    const int* k = a.kk + (long)yy * a.ksize;
    const uint8_t* sp = a.src + ((n * a.H + y0) * a.OW + x) * a.C + c;
    int acc = 1 << 21;
    const long rs = (long)a.OW * a.C;
    for (int t = 0; t < cnt; ++t) acc += (int)sp[t * rs] * k[t];
    acc >>= 22;
    const float v = (float)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
    if (a.normalize) return __fdiv_rn(__fsub_rn(v, a.mean[c]), a.std_[c]);
    return __fdiv_rn(v, 255.0f);
}

// vertical pass + normalise + layout; one thread per output pixel (layout 0 / 1) or per folded pixel (layout 2)
template <typename TD>
__global__ void resize_v_norm_kernel(const PreArgs a) {
    TD* out = reinterpret_cast<TD*>(a.out);
    if (a.layout == 2) {
        const int H2 = a.OH / a.b, W2 = a.OW / a.b, Cs = a.b * a.b * a.C;
        const long total = (long)a.N * H2 * W2;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const int w2 = (int)(i % W2);
            const long r = i / W2;
            const int h2 = (int)(r % H2);
            const long n = r / H2;
            TD* dp = out + i * a.cpad;
            for (int ch = 0; ch < a.cpad; ++ch) {
                float v = 0.f;
                if (ch < Cs) {
                    const int c = ch % a.C, q = ch / a.C, pw = q % a.b, ph = q / a.b;
                    v = pre_value(a, n, h2 * a.b + ph, w2 * a.b + pw, c);
                }
                dp[ch] = (TD)v;
            }
        }
        return;
    }
    const long total = (long)a.N * a.OH * a.OW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.OW);
        const long r = i / a.OW;
        const int yy = (int)(r % a.OH);
        const long n = r / a.OH;
        for (int c = 0; c < a.C; ++c) {
            const float v = pre_value(a, n, yy, x, c);
            if (a.layout == 0) out[((n * a.C + c) * a.OH + yy) * (long)a.OW + x] = (TD)v;
            else out[i * a.C + c] = (TD)v;
        }
    }
}

// ---- the detection demo's pipeline (demo/object_detection/transforms.py:96-246, called at predict-YOLOv3.py:54-61):
// Resize(size, max_size, auto_divide) -> cv2.resize(..., INTER_LINEAR) on the uint8 image, then Normalize
// ((v / 255 - mean) / std in fp32).  OpenCV's 8-bit bilinear resize restated (resize.cpp, HResizeLinear /
// VResizeLinear<uchar>): two taps per axis with 11-bit fixed-point weights (no anti-aliasing when shrinking),
//     row(y)[x] = S[y][x0] * a0 + S[y][x0 + 1] * a1                                   (int, scale 2^11)
//     out       = ((b0 * (row(y0)[x] >> 4)) >> 16) + ((b1 * (row(y1)[x] >> 4)) >> 16) + 2) >> 2
// with the tap indices / weights of both axes built by the caller (tlx/vision/transforms/detection.py: the same
// float arithmetic as OpenCV's coefficient loop).  One pass, one thread per output pixel: HBM-bound byte work.
// UNPINNED: OpenCV is not in this image and the reference holds no vector for it (an OpenCV built with IPP / a HAL
// may differ in the last bit); the oracle restates the same published algorithm independently.
struct PreLinArgs {
    const uint8_t* src;                       // [N][H][W][C]
    const int* xi;  const int* xa;            // [OW][2] source columns (clamped), [OW][2] weights
    const int* yi;  const int* yb;            // [OH][2], [OH][2]
    const float* mean;  const float* std_;
    void* out;
    int N, H, W, OH, OW, C, layout, normalize;
};

template <typename TD>
__global__ void resize_linear_norm_kernel(const PreLinArgs a) {
    TD* out = reinterpret_cast<TD*>(a.out);
    const long total = (long)a.N * a.OH * a.OW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.OW);
        const long r = i / a.OW;
        const int yy = (int)(r % a.OH);
        const long n = r / a.OH;
        const int x0 = a.xi[2 * x], x1 = a.xi[2 * x + 1], a0 = a.xa[2 * x], a1 = a.xa[2 * x + 1];
        const int y0 = a.yi[2 * yy], y1 = a.yi[2 * yy + 1], b0 = a.yb[2 * yy], b1 = a.yb[2 * yy + 1];
        const uint8_t* r0 = a.src + (n * a.H + y0) * (long)a.W * a.C;
        const uint8_t* r1 = a.src + (n * a.H + y1) * (long)a.W * a.C;
        for (int c = 0; c < a.C; ++c) {
            const int h0 = (int)r0[x0 * a.C + c] * a0 + (int)r0[x1 * a.C + c] * a1;
            const int h1 = (int)r1[x0 * a.C + c] * a0 + (int)r1[x1 * a.C + c] * a1;
            int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            float f = __fdiv_rn((float)v, 255.0f);                                   // image.astype(float32) / 255.0
            if (a.normalize) f = __fdiv_rn(__fsub_rn(f, a.mean[c]), a.std_[c]);     // (. - mean) / std
            if (a.layout == 0) out[((n * a.C + c) * a.OH + yy) * (long)a.OW + x] = (TD)f;
            else out[i * a.C + c] = (TD)f;
        }
    }
}

}  // namespace tlxmi

using namespace tlxmi;

extern "C" int tlxmi_preprocess_linear_u8(const tlxmi_preproc_desc* d, const void* images, const int32_t* xidx, const int32_t* xcoef,
                                          const int32_t* yidx, const int32_t* ycoef, const float* mean, const float* std_, void* out,
                                          void* stream) {
    TLXMI_REQUIRE(d && images && xidx && xcoef && yidx && ycoef && out, TLXMI_ERR_BAD_ARG, "preprocess_linear_u8: null argument");
    TLXMI_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C >= 1 && d->C <= 4 && d->out_h > 0 && d->out_w > 0, TLXMI_ERR_BAD_ARG,
                  "preprocess_linear_u8: bad extent");
    TLXMI_REQUIRE(d->out_dtype == TLXMI_F16 || d->out_dtype == TLXMI_F32, TLXMI_ERR_BAD_ARG, "preprocess_linear_u8: bad output dtype");
    TLXMI_REQUIRE(d->layout == 0 || d->layout == 1, TLXMI_ERR_BAD_ARG, "preprocess_linear_u8: layout must be 0 (NCHW) or 1 (NHWC)");
    TLXMI_REQUIRE(!d->normalize || (mean && std_), TLXMI_ERR_BAD_ARG, "preprocess_linear_u8: normalize needs mean and std");
    TLXMI_REQUIRE((long long)d->N * d->H * d->W * d->C < (1ll << 40) && (long long)d->N * d->out_h * d->out_w * d->C < (1ll << 40),
                  TLXMI_ERR_UNSUPPORTED, "preprocess_linear_u8: batch too large");
    PreLinArgs a;
    a.src = (const uint8_t*)images; a.xi = xidx; a.xa = xcoef; a.yi = yidx; a.yb = ycoef; a.mean = mean; a.std_ = std_; a.out = out;
    a.N = d->N; a.H = d->H; a.W = d->W; a.OH = d->out_h; a.OW = d->out_w; a.C = d->C; a.layout = d->layout; a.normalize = d->normalize;
    const long total = (long)d->N * d->out_h * d->out_w;
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (d->out_dtype == TLXMI_F16) hipLaunchKernelGGL((resize_linear_norm_kernel<half_t>), dim3(grid), dim3(256), 0, as_stream(stream), a);
    else hipLaunchKernelGGL((resize_linear_norm_kernel<float>), dim3(grid), dim3(256), 0, as_stream(stream), a);
    return check_launch("preprocess_linear_u8");
}

extern "C" size_t tlxmi_preprocess_u8_workspace_bytes(const tlxmi_preproc_desc* d) {
    if (!d || d->N <= 0 || d->H <= 0 || d->out_w <= 0 || d->C <= 0) return 0;
    return (size_t)d->N * d->H * d->out_w * d->C;
}

extern "C" int tlxmi_preprocess_u8(const tlxmi_preproc_desc* d, const void* images, const int32_t* xbounds, const int32_t* xk,
                                   const int32_t* ybounds, const int32_t* yk, const float* mean, const float* std_, void* workspace,
                                   void* out, void* stream) {
    TLXMI_REQUIRE(d && images && xbounds && xk && ybounds && yk && workspace && out, TLXMI_ERR_BAD_ARG, "preprocess_u8: null argument");
    TLXMI_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C >= 1 && d->C <= 4 && d->out_h > 0 && d->out_w > 0 && d->kw > 0 && d->kh > 0,
                  TLXMI_ERR_BAD_ARG, "preprocess_u8: bad extent");
    TLXMI_REQUIRE(d->out_dtype == TLXMI_F16 || d->out_dtype == TLXMI_F32, TLXMI_ERR_BAD_ARG, "preprocess_u8: bad output dtype");
    TLXMI_REQUIRE(d->layout >= 0 && d->layout <= 2, TLXMI_ERR_BAD_ARG, "preprocess_u8: layout must be 0 (NCHW), 1 (NHWC) or 2 (space-to-depth NHWC)");
    TLXMI_REQUIRE(!d->normalize || (mean && std_), TLXMI_ERR_BAD_ARG, "preprocess_u8: normalize needs mean and std");
    if (d->layout == 2)
        TLXMI_REQUIRE(d->fold_b >= 1 && d->out_h % d->fold_b == 0 && d->out_w % d->fold_b == 0 && d->cpad >= d->fold_b * d->fold_b * d->C,
                      TLXMI_ERR_BAD_ARG, "preprocess_u8: output extent must be a multiple of the fold, cpad >= b*b*C");
    TLXMI_REQUIRE((long long)d->N * d->H * (d->W > d->out_w ? d->W : d->out_w) * d->C < (1ll << 40), TLXMI_ERR_UNSUPPORTED, "preprocess_u8: batch too large");
    hipStream_t st = as_stream(stream);
    const long rows = (long)d->N * d->H;
    {
        const long total = rows * d->out_w;
        const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
        hipLaunchKernelGGL(resize_h_u8_kernel, dim3(grid), dim3(256), 0, st, (const uint8_t*)images, (uint8_t*)workspace, xbounds, xk, rows,
                           d->W, d->out_w, d->C, d->kw);
    }
    PreArgs a;
    a.src = (const uint8_t*)workspace; a.bounds = ybounds; a.kk = yk; a.mean = mean; a.std_ = std_; a.out = out;
    a.N = d->N; a.H = d->H; a.OH = d->out_h; a.OW = d->out_w; a.C = d->C; a.ksize = d->kh; a.layout = d->layout;
    a.b = d->fold_b; a.cpad = d->cpad; a.normalize = d->normalize;
    const long total = d->layout == 2 ? (long)d->N * (d->out_h / d->fold_b) * (d->out_w / d->fold_b) : (long)d->N * d->out_h * d->out_w;
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (d->out_dtype == TLXMI_F16) hipLaunchKernelGGL((resize_v_norm_kernel<half_t>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((resize_v_norm_kernel<float>), dim3(grid), dim3(256), 0, st, a);
    return check_launch("preprocess_u8");
}

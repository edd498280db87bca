// YOLOv3 post-processing on the device: box decode + multiclass NMS.
//
// tlxmi_yolo_box — YOLOBox.__call__ (tlxcv/models/detection/yolov3.py:558-579) calls `yolo_box_func`, which exists only on
//   the Paddle backend (detection/utils/ops.py:436-452: paddle.vision.ops.yolo_box).  Its published algorithm, restated:
//   head map x [N][A*(5+C)][H][W]; for anchor a, cell (k, l):  conf = sigmoid(x[a, 4]); below conf_thresh the box and its
//   scores stay zero; else
//       cx = (l + sigmoid(x[a,0]) * s - 0.5*(s - 1)) * img_w / W        cy likewise with k, img_h, H
//       bw = exp(x[a,2]) * anchor_w * img_w / (downsample * W)          bh likewise
//       box = (cx - bw/2, cy - bh/2, cx + bw/2, cy + bh/2), clipped to [0, img - 1] when clip_bbox
//       score[c] = conf * sigmoid(x[a, 5 + c])
//   written at box index a*H*W + k*W + l of the image's list ([N][Mtot][4] and [N][Mtot][C]: several heads append).
// tlxmi_multiclass_nms — tlx_multiclass_nms (detection/utils/ops.py:255-329), the reference's torch-side NMS: per image,
//   best class and its score per box, keep score >= score_threshold, torchvision.ops.batched_nms (greedy, descending
//   score, suppress IoU > nms_threshold within a class — by the coordinate trick: boxes shifted by class * (max coordinate
//   + 1), reproduced here so that borderline IoUs round the same way), the first keep_top_k survivors, rows (class, score,
//   x1, y1, x2, y2).  One workgroup per image: bitonic sort of (score, index) keys in a workspace, then the greedy sweep with
//   the suppression flags in LDS.  A coverage kernel (integer / compare work, a few boxes per image in practice).
#include "common.h"

namespace tlxmi {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// IoU-aware head (YOLOv3Head.forward, yolov3.py:355-376): the head map carries A IoU predictions in front of the A * (5 + C)
// box entries; objectness becomes obj' = de_sigmoid(sigmoid(obj)^(1 - f) * sigmoid(ioup)^f) and the IoU channels are dropped:
// x [N][H*W][A*(6+C)] (NHWC: channel a = ioup of anchor a, channel A + a*(5+C) + e = entry e) -> y [N][H*W][A*(5+C)].
// _de_sigmoid (:113-119): x clipped to [eps, 1/eps], then -log(clip(1/x - 1, eps, 1/eps)), eps = 1e-7.
template <typename T>
__global__ void iou_aware_kernel(const T* __restrict__ x, T* __restrict__ y, long pixels, int A, int C, float factor) {
    const int E5 = 5 + C, cin = A * (6 + C), cout = A * E5;
    const long total = pixels * cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cout);
        const long p = i / cout;
        const int a = ch / E5, e = ch - a * E5;
        float v = (float)x[p * cin + A + ch];
        if (e == 4) {
            const float obj = sigmoidf_(v), iou = sigmoidf_((float)x[p * cin + a]);
            float t = powf(obj, 1.f - factor) * powf(iou, factor);
            const float eps = 1e-7f;
            t = fminf(fmaxf(t, eps), 1.f / eps);
            t = fminf(fmaxf(1.f / t - 1.f, eps), 1.f / eps);
            v = -logf(t);
        }
        y[i] = (T)v;
    }
}

template <typename T>
__global__ void yolo_box_kernel(const T* __restrict__ x, const int* __restrict__ img_size, const float* __restrict__ anchors,
                                float* __restrict__ boxes, float* __restrict__ scores, int N, int A, int C, int H, int W,
                                long x_nstride, int x_cstride, int x_pstride, int Mtot, int m_off, float conf_thresh, int downsample,
                                int clip_bbox, float scale_xy) {
    const long total = (long)N * A * H * W;
    const float bias = -0.5f * (scale_xy - 1.0f);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int l = (int)(i % W);
        long r = i / W;
        const int k = (int)(r % H);
        r /= H;
        const int a = (int)(r % A);
        const long n = r / A;
        const T* xp = x + n * x_nstride + (long)(k * W + l) * x_pstride;     // entry e of anchor a: channel a*(5+C) + e
        auto ent = [&](int e) { return (float)xp[(long)(a * (5 + C) + e) * x_cstride]; };
        const long m = n * Mtot + m_off + (long)a * H * W + (long)k * W + l;
        float* bp = boxes + m * 4;
        float* sp = scores + m * C;
        const float conf = sigmoidf_(ent(4));
        if (conf < conf_thresh) {
            bp[0] = bp[1] = bp[2] = bp[3] = 0.f;
            for (int c = 0; c < C; ++c) sp[c] = 0.f;
            continue;
        }
        const float img_h = (float)img_size[2 * n], img_w = (float)img_size[2 * n + 1];
        const float cx = ((float)l + sigmoidf_(ent(0)) * scale_xy + bias) * img_w / (float)W;
        const float cy = ((float)k + sigmoidf_(ent(1)) * scale_xy + bias) * img_h / (float)H;
        const float bw = expf(ent(2)) * anchors[2 * a] * img_w / (float)(downsample * W);
        const float bh = expf(ent(3)) * anchors[2 * a + 1] * img_h / (float)(downsample * H);
        float x1 = cx - bw / 2, y1 = cy - bh / 2, x2 = cx + bw / 2, y2 = cy + bh / 2;
        if (clip_bbox) {
            x1 = x1 > 0.f ? x1 : 0.f;
            y1 = y1 > 0.f ? y1 : 0.f;
            x2 = x2 < img_w - 1.f ? x2 : img_w - 1.f;
            y2 = y2 < img_h - 1.f ? y2 : img_h - 1.f;
        }
        bp[0] = x1; bp[1] = y1; bp[2] = x2; bp[3] = y2;
        for (int c = 0; c < C; ++c) sp[c] = conf * sigmoidf_(ent(5 + c));
    }
}

// keys[n][MP]: (score bits << 32) | ~index for candidates (score >= threshold), 0 otherwise / for padding; cls[n][M]
__global__ void nms_keys_kernel(const float* __restrict__ scores, unsigned long long* __restrict__ keys, int* __restrict__ cls, int N, int M,
                                int C, int MP, float thr) {
    const long total = (long)N * MP;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i % MP);
        const long n = i / MP;
        unsigned long long key = 0ull;
        if (m < M) {
            const float* sp = scores + (n * M + m) * C;
            float best = sp[0];
            int bc = 0;
            for (int c = 1; c < C; ++c)
                if (sp[c] > best) { best = sp[c]; bc = c; }        // first maximal value, as argmax
            cls[n * M + m] = bc;
            if (best >= thr && best > 0.f) key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)m);
            else if (best >= thr) key = 1ull + (unsigned long long)(0xFFFFFFFEu - (unsigned)m);   // score 0 with a threshold <= 0: still a candidate
        }
        keys[i] = key;
    }
}

// The same with SIXTEEN lanes per box: lane j reads classes j, j + 16, ... (a wave-load covers 64 contiguous bytes of each of four
// boxes), then the sixteen (value, first index) pairs are combined.  One thread per box reads C consecutive floats at a stride
// of C floats between lanes: 0.37 TB/s on YOLOv3's 32 x 10 647 x 80 scores.
__global__ void nms_keys16_kernel(const float* __restrict__ scores, unsigned long long* __restrict__ keys, int* __restrict__ cls, int N, int M,
                                  int C, int MP, float thr) {
    const long total = (long)N * MP * 16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int sub = (int)(i & 15);
        const long bi = i >> 4;
        const int m = (int)(bi % MP);
        const long n = bi / MP;
        float best = -INFINITY;
        int bc = 0x7fffffff;
        if (m < M) {
            const float* sp = scores + (n * M + m) * C;
            for (int c = sub; c < C; c += 16) {
                const float v = sp[c];
                if (v > best || c == sub) { best = v; bc = c; }          // first maximal value among this lane's classes
            }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oc = __shfl_xor(bc, o, 64);
            if (oc != 0x7fffffff && (bc == 0x7fffffff || ob > best || (ob == best && oc < bc))) { best = ob; bc = oc; }   // first maximal value, as argmax
        }
        if (sub == 0) {
            unsigned long long key = 0ull;
            if (m < M) {
                cls[n * M + m] = bc;
                if (best >= thr && best > 0.f) key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)m);
                else if (best >= thr) key = 1ull + (unsigned long long)(0xFFFFFFFEu - (unsigned)m);
            }
            keys[bi] = key;
        }
    }
}

template <bool LDSKEYS>
__global__ __launch_bounds__(1024) void nms_image_kernel(const float* __restrict__ boxes, unsigned long long* __restrict__ keys,
                                                         const int* __restrict__ cls, float* __restrict__ det, int* __restrict__ count,
                                                         int* __restrict__ kidx, int M, int MP, float nms_thr, int keep_top_k) {
    extern __shared__ __attribute__((aligned(16))) char nms_smem[];
    // LDSKEYS (MP <= 16384): the image's keys are sorted in LDS, [MP] keys then [MP] suppression flags — the 105 compare-exchange
    // passes of 16 384 keys cost a global round trip each when sorted in place (YOLOv3, 32 images: 662 us, most of it the sort)
    unsigned char* sup = reinterpret_cast<unsigned char*>(nms_smem) + (LDSKEYS ? (size_t)MP * 8 : 0);       // [MP] suppression flags
    __shared__ float red[1024];
    __shared__ int s_k;
    const int n = blockIdx.x, t = threadIdx.x;
    unsigned long long* kp = LDSKEYS ? reinterpret_cast<unsigned long long*>(nms_smem) : keys + (long)n * MP;
    if constexpr (LDSKEYS) {
        const unsigned long long* gk = keys + (long)n * MP;
        for (int i = t; i < MP; i += 1024) kp[i] = gk[i];
    }
    // bitonic sort, descending
    for (int lsize = 1; (1 << lsize) <= MP; ++lsize)
        for (int ls = lsize - 1; ls >= 0; --ls) {
            const int stride = 1 << ls;
            __syncthreads();
            for (int i = t; i < MP / 2; i += 1024) {
                const int lo = ((i >> ls) << (ls + 1)) | (i & (stride - 1)), hi = lo + stride;      // (powers of two: no division)
                const bool desc = ((lo >> lsize) & 1) == 0;
                const unsigned long long a = kp[lo], b = kp[hi];
                if (desc ? (a < b) : (a > b)) { kp[lo] = b; kp[hi] = a; }
            }
        }
    __syncthreads();
    // number of candidates, largest coordinate among them
    int mine = 0;
    float mx = -INFINITY;
    for (int i = t; i < MP; i += 1024) {
        sup[i] = 0;
        const unsigned long long k = kp[i];
        if (k != 0ull) {
            ++mine;
            const float* b = boxes + ((long)n * M + (0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull))) * 4;
            mx = fmaxf(fmaxf(fmaxf(mx, b[0]), fmaxf(b[1], b[2])), b[3]);
        }
    }
    if (t == 0) s_k = 0;
    __syncthreads();
    atomicAdd(&s_k, mine);
    red[t] = mx;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (t < o) red[t] = fmaxf(red[t], red[t + o]);
        __syncthreads();
    }
    const int K = s_k;
    const float shift = red[0] + 1.0f;                     // batched_nms: offsets = class * (max coordinate + 1)
    float* dp = det + (long)n * keep_top_k * 6;
    // LDSKEYS: the sorted keys go back to the workspace and their LDS makes room for the first LB candidates' boxes (class offset
    // applied), in score order: the suppression pass of a kept box then reads consecutive LDS entries instead of gathering a
    // box and a class per candidate from global memory (ten dependent gathers per thread and kept box: 5 us x 100 kept boxes)
    const int LB = LDSKEYS ? (MP / 2 < K ? MP / 2 : K) : 0;
    f32x4* lbox = reinterpret_cast<f32x4*>(nms_smem);
    unsigned long long* kg = keys + (long)n * MP;
    if constexpr (LDSKEYS) {
        for (int i = t; i < MP; i += 1024) kg[i] = kp[i];
        __syncthreads();
        for (int i = t; i < LB; i += 1024) {
            const int mi = (int)(0xFFFFFFFFu - (unsigned)(kg[i] & 0xFFFFFFFFull));
            const float* b = boxes + ((long)n * M + mi) * 4;
            const float off = (float)cls[(long)n * M + mi] * shift;
            lbox[i] = f32x4{b[0] + off, b[1] + off, b[2] + off, b[3] + off};
        }
        __syncthreads();
    }
    int kept = 0;
    __shared__ int s_kept[1024];                          // sorted positions of the kept boxes (keep_top_k <= 1024; else the direct form)
    const bool defer = keep_top_k <= 1024;
    for (int i = 0; i < K && kept < keep_top_k; ++i) {
        if (sup[i]) continue;                              // (uniform: flags only change between the barriers below)
        // the kept box: from LDS when it is among the first LB (no global load inside the loop — the key -> box / class gathers of
        // every kept box were a chain of two dependent global loads that all 1024 threads waited for); its output row is
        // written after the loop, all rows in parallel
        float ax1, ay1, ax2, ay2;
        if (i < LB) {
            const f32x4 ab = lbox[i];
            ax1 = ab[0]; ay1 = ab[1]; ax2 = ab[2]; ay2 = ab[3];
        } else {
            const int mi = (int)(0xFFFFFFFFu - (unsigned)(kg[i] & 0xFFFFFFFFull));
            const float* bi = boxes + ((long)n * M + mi) * 4;
            const float off = (float)cls[(long)n * M + mi] * shift;
            ax1 = bi[0] + off; ay1 = bi[1] + off; ax2 = bi[2] + off; ay2 = bi[3] + off;
        }
        const float aarea = (ax2 - ax1) * (ay2 - ay1);
        if (defer) {
            if (t == 0) s_kept[kept] = i;
        } else if (t == 0) {
            const unsigned long long ki = kg[i];
            const int mi = (int)(0xFFFFFFFFu - (unsigned)(ki & 0xFFFFFFFFull));
            const float* bi = boxes + ((long)n * M + mi) * 4;
            float* o = dp + kept * 6;
            o[0] = (float)cls[(long)n * M + mi]; o[1] = __uint_as_float((unsigned)(ki >> 32)); o[2] = bi[0]; o[3] = bi[1]; o[4] = bi[2]; o[5] = bi[3];
            if (kidx) kidx[(long)n * keep_top_k + kept] = mi;
        }
        ++kept;
        for (int j = i + 1 + t; j < K; j += 1024) {
            if (sup[j]) continue;
            float bx1, by1, bx2, by2;
            if (j < LB) {
                const f32x4 bb = lbox[j];
                bx1 = bb[0]; by1 = bb[1]; bx2 = bb[2]; by2 = bb[3];
            } else {
                const int mj = (int)(0xFFFFFFFFu - (unsigned)(kg[j] & 0xFFFFFFFFull));
                const float* bj = boxes + ((long)n * M + mj) * 4;
                const float offj = (float)cls[(long)n * M + mj] * shift;
                bx1 = bj[0] + offj; by1 = bj[1] + offj; bx2 = bj[2] + offj; by2 = bj[3] + offj;
            }
            const float w = fminf(ax2, bx2) - fmaxf(ax1, bx1), h = fminf(ay2, by2) - fmaxf(ay1, by1);
            const float inter = (w > 0.f ? w : 0.f) * (h > 0.f ? h : 0.f);
            const float iou = inter / (aarea + (bx2 - bx1) * (by2 - by1) - inter);
            if (iou > nms_thr) sup[j] = 1;
        }
        __syncthreads();
    }
    if (defer) {
        __syncthreads();
        if (t < kept) {
            const unsigned long long ki = kg[s_kept[t]];
            const int mi = (int)(0xFFFFFFFFu - (unsigned)(ki & 0xFFFFFFFFull));
            const float* bi = boxes + ((long)n * M + mi) * 4;
            float* o = dp + t * 6;
            o[0] = (float)cls[(long)n * M + mi]; o[1] = __uint_as_float((unsigned)(ki >> 32)); o[2] = bi[0]; o[3] = bi[1]; o[4] = bi[2]; o[5] = bi[3];
            if (kidx) kidx[(long)n * keep_top_k + t] = mi;
        }
    }
    if (t == 0) count[n] = kept;
    for (int i = kept * 6 + t; i < keep_top_k * 6; i += 1024) dp[i] = 0.f;
    if (kidx)
        for (int i = kept + t; i < keep_top_k; i += 1024) kidx[(long)n * keep_top_k + i] = -1;
}

}  // namespace tlxmi

using namespace tlxmi;

extern "C" int tlxmi_yolo_iou_aware(const void* x, void* y, int dtype, int64_t pixels, int A, int C, float factor, void* stream) {
    TLXMI_REQUIRE(x && y && pixels > 0 && A > 0 && C >= 0, TLXMI_ERR_BAD_ARG, "yolo_iou_aware: bad argument");
    TLXMI_REQUIRE(dtype == TLXMI_F16 || dtype == TLXMI_F32, TLXMI_ERR_BAD_ARG, "yolo_iou_aware: bad dtype");
    const long total = (long)pixels * A * (5 + C);
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (dtype == TLXMI_F16) hipLaunchKernelGGL((iou_aware_kernel<half_t>), dim3(grid), dim3(256), 0, as_stream(stream), (const half_t*)x, (half_t*)y, (long)pixels, A, C, factor);
    else hipLaunchKernelGGL((iou_aware_kernel<float>), dim3(grid), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, (long)pixels, A, C, factor);
    return check_launch("yolo_iou_aware");
}

extern "C" int tlxmi_yolo_box(const void* x, int dtype, int N, int A, int C, int H, int W, int channels_last, const int32_t* img_size,
                              const float* anchors, float conf_thresh, int downsample_ratio, int clip_bbox, float scale_x_y,
                              float* boxes, float* scores, int Mtot, int m_offset, void* stream) {
    TLXMI_REQUIRE(x && img_size && anchors && boxes && scores, TLXMI_ERR_BAD_ARG, "yolo_box: null argument");
    TLXMI_REQUIRE(dtype == TLXMI_F16 || dtype == TLXMI_F32, TLXMI_ERR_BAD_ARG, "yolo_box: bad dtype");
    TLXMI_REQUIRE(N > 0 && A > 0 && C > 0 && H > 0 && W > 0 && downsample_ratio > 0 && m_offset >= 0 && m_offset + A * H * W <= Mtot,
                  TLXMI_ERR_BAD_ARG, "yolo_box: bad extent (boxes %d + %d of %d)", m_offset, A * H * W, Mtot);
    const int ch = A * (5 + C);
    // channels_first [N][ch][H][W]: channel stride H*W, pixel stride 1; channels_last [N][H][W][ch]: channel stride 1, pixel stride ch
    const long nst = (long)ch * H * W;
    const int cst = channels_last ? 1 : H * W, pst = channels_last ? ch : 1;
    const long total = (long)N * A * H * W;
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == TLXMI_F16)
        hipLaunchKernelGGL((yolo_box_kernel<half_t>), dim3(grid), dim3(256), 0, as_stream(stream), (const half_t*)x, img_size, anchors, boxes, scores, N, A, C, H, W,
                           nst, cst, pst, Mtot, m_offset, conf_thresh, downsample_ratio, clip_bbox, scale_x_y);
    else
        hipLaunchKernelGGL((yolo_box_kernel<float>), dim3(grid), dim3(256), 0, as_stream(stream), (const float*)x, img_size, anchors, boxes, scores, N, A, C, H, W,
                           nst, cst, pst, Mtot, m_offset, conf_thresh, downsample_ratio, clip_bbox, scale_x_y);
    return check_launch("yolo_box");
}

static int nms_pow2(int m) { int p = 2; while (p < m) p <<= 1; return p; }

extern "C" size_t tlxmi_multiclass_nms_workspace_bytes(int N, int M) {
    if (N <= 0 || M <= 0) return 0;
    return (size_t)N * nms_pow2(M) * 8 + (size_t)N * M * 4;
}

static int multiclass_nms_impl(const float* boxes, const float* scores, int N, int M, int C, float score_threshold, float nms_threshold,
                               int keep_top_k, void* workspace, float* detections, int32_t* counts, int32_t* keep_index, void* stream) {
    TLXMI_REQUIRE(boxes && scores && workspace && detections && counts, TLXMI_ERR_BAD_ARG, "multiclass_nms: null argument");
    TLXMI_REQUIRE(N > 0 && M > 0 && C > 0 && keep_top_k > 0, TLXMI_ERR_BAD_ARG, "multiclass_nms: bad extent");
    const int MP = nms_pow2(M);
    TLXMI_REQUIRE(MP <= 65536, TLXMI_ERR_UNSUPPORTED, "multiclass_nms: %d boxes per image (<= 65536)", M);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(workspace);
    int* cls = reinterpret_cast<int*>(keys + (size_t)N * MP);
    hipStream_t st = as_stream(stream);
    const long total = (long)N * MP;
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (C >= 8) {
        const long total16 = total * 16;
        const unsigned grid16 = (unsigned)((total16 + 255) / 256 < 32768 ? (total16 + 255) / 256 : 32768);
        hipLaunchKernelGGL(nms_keys16_kernel, dim3(grid16), dim3(256), 0, st, scores, keys, cls, N, M, C, MP, score_threshold);
    } else {
        hipLaunchKernelGGL(nms_keys_kernel, dim3(grid), dim3(256), 0, st, scores, keys, cls, N, M, C, MP, score_threshold);
    }
    if (MP <= 16384) {
        const size_t lds = (size_t)MP * 9;
        if (lds > 48 * 1024)       // (the kernel also has 8 KB of static LDS: the dynamic limit cannot be the whole 160 KB)
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(&nms_image_kernel<true>), 148 * 1024, "multiclass_nms")) return rc;
        hipLaunchKernelGGL(nms_image_kernel<true>, dim3(N), dim3(1024), lds, st, boxes, keys, cls, detections, counts, keep_index, M, MP, nms_threshold, keep_top_k);
    } else {
        const size_t lds = (size_t)MP;
        if (lds > 48 * 1024)
            if (int rc = raise_lds_limit(reinterpret_cast<const void*>(&nms_image_kernel<false>), 96 * 1024, "multiclass_nms")) return rc;
        hipLaunchKernelGGL(nms_image_kernel<false>, dim3(N), dim3(1024), lds, st, boxes, keys, cls, detections, counts, keep_index, M, MP, nms_threshold, keep_top_k);
    }
    return check_launch("multiclass_nms");
}

extern "C" int tlxmi_multiclass_nms(const float* boxes, const float* scores, int N, int M, int C, float score_threshold, float nms_threshold,
                                    int keep_top_k, void* workspace, float* detections, int32_t* counts, void* stream) {
    return multiclass_nms_impl(boxes, scores, N, M, C, score_threshold, nms_threshold, keep_top_k, workspace, detections, counts, nullptr, stream);
}

extern "C" int tlxmi_multiclass_nms_index(const float* boxes, const float* scores, int N, int M, int C, float score_threshold,
                                          float nms_threshold, int keep_top_k, void* workspace, float* detections, int32_t* counts,
                                          int32_t* keep_index, void* stream) {
    TLXMI_REQUIRE(keep_index, TLXMI_ERR_BAD_ARG, "multiclass_nms_index: null keep_index");
    return multiclass_nms_impl(boxes, scores, N, M, C, score_threshold, nms_threshold, keep_top_k, workspace, detections, counts, keep_index, stream);
}

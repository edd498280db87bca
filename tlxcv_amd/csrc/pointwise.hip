// HBM-bound scan kernels: layout conversion, pooling, affine/activation, depthwise conv, upsample,
// channel copies, argmax.  No MFMA here by design: every kernel moves 16-byte chunks of the NHWC
// channel axis per lane (consecutive lanes -> consecutive chunks) so wave accesses are full lines.
#include "common.h"
#include <stdlib.h>

namespace tlxmi {

template <typename T> struct Chunk;  // 16 bytes of T <-> fp32 registers
template <> struct Chunk<half_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const void* p, float* v) {
        half8v h = *reinterpret_cast<const half8v*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
    }
    static __device__ __forceinline__ void store(void* p, const float* v) {
        half8v h;
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = (half_t)v[i];
        *reinterpret_cast<half8v*>(p) = h;
    }
};
template <> struct Chunk<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const void* p, float* v) {
        f32x4 h = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = h[i];
    }
    static __device__ __forceinline__ void store(void* p, const float* v) {
        f32x4 h;
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = v[i];
        *reinterpret_cast<f32x4*>(p) = h;
    }
};

static inline int grid_for(long work, int block = 256, int cap = 256 * 16) {
    long g = (work + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g < cap ? g : cap);
}

// ------------------------------------------------------------------------------------------
// NCHW -> NHWC(Cpad): one lane per (pixel, chunk); pixel index fastest so plane reads coalesce.
// ------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ void nchw_to_nhwc_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int N, int C, int H, int W,
                                    int Cpad) {
    constexpr int V = Chunk<TD>::N;
    const long HW = (long)H * W, P = (long)N * HW;
    const int nch = Cpad / V;
    const long total = P * nch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i % P;
        const int cg = (int)(i / P);
        const long n = p / HW, hw = p - n * HW;
        float v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int c = cg * V + e;
            v[e] = c < C ? (float)src[(n * C + c) * HW + hw] : 0.f;
        }
        Chunk<TD>::store(dst + p * Cpad + cg * V, v);
    }
}

// NCHW -> NHWC with a b x b space-to-depth fold: channel (ph*b+pw)*C + c of output pixel (h2, w2).
// One workgroup = 128 consecutive output pixels of one output row (the row / image indices are wave-uniform,
// no per-thread division); a thread writes all chunks of its pixel, so a wave's stores cover whole rows of
// Cpad channels and its loads walk b*C short runs of one input row each.
template <typename TS, typename TD>
__global__ __launch_bounds__(128) void nchw_to_nhwc_s2d_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int N, int C,
                                                              int H, int W, int b, int Cpad, int cb) {
    constexpr int V = Chunk<TD>::N;
    const int H2 = H / b, W2 = W / b;
    const int row = blockIdx.x / cb, w2 = (blockIdx.x - row * cb) * 128 + threadIdx.x;   // row = n*H2 + h2
    if (w2 >= W2) return;
    const int n = row / H2, h2 = row - n * H2;
    const long HW = (long)H * W;
    const int Cs = b * b * C;
    const TS* sp = src + (long)n * C * HW + (long)(h2 * b) * W + w2 * b;
    TD* dp = dst + ((long)row * W2 + w2) * Cpad;
    for (int cg = 0; cg < Cpad / V; ++cg) {
        float v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int ch = cg * V + e;
            float x = 0.f;
            if (ch < Cs) {
                const int c = ch % C, q = ch / C, pw = q % b, ph = q / b;
                x = (float)sp[c * HW + (long)ph * W + pw];
            }
            v[e] = x;
        }
        Chunk<TD>::store(dp + cg * V, v);
    }
}

// Large folds (b = 16 patch embedding of ViT, b = 4 of Swin): one workgroup per output row of patches.  The b input rows of
// every channel are read with consecutive lanes on consecutive floats (the per-pixel kernel above reads 4-byte elements
// b floats apart), the folded row is assembled in LDS and leaves as whole 16-byte chunks.
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void nchw_to_nhwc_s2d_rows_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int C, int H,
                                                                  int W, int b, int Cpad) {
    extern __shared__ __attribute__((aligned(16))) char s2d_smem[];
    TD* tile = reinterpret_cast<TD*>(s2d_smem);            // [W2][Cpad]
    const int H2 = H / b, W2 = W / b;
    const int row = blockIdx.x, n = row / H2, h2 = row - n * H2;
    const long HW = (long)H * W;
    const int Cs = b * b * C;
    // zero the padding channels once
    for (int i = threadIdx.x; i < W2 * (Cpad - Cs); i += 256) {
        const int px = i / (Cpad - Cs), c = Cs + i - px * (Cpad - Cs);
        tile[px * Cpad + c] = (TD)0.f;
    }
    const TS* sp = src + (long)n * C * HW + (long)(h2 * b) * W;
    for (int cp = 0; cp < C * b; ++cp) {                   // (channel, row inside the patch): W consecutive floats each
        const int c = cp / b, ph = cp - c * b;
        const TS* rp = sp + c * HW + (long)ph * W;
        for (int w = threadIdx.x; w < W; w += 256) {
            const int px = w / b, pw = w - px * b;
            tile[px * Cpad + (ph * b + pw) * C + c] = (TD)(float)rp[w];
        }
    }
    __syncthreads();
    constexpr int V = Chunk<TD>::N;
    const int nchunk = W2 * Cpad / V;
    const u32x4* tv = reinterpret_cast<const u32x4*>(tile);
    u32x4* dv = reinterpret_cast<u32x4*>(dst + (long)row * W2 * Cpad);
    for (int i = threadIdx.x; i < nchunk; i += 256) dv[i] = tv[i];
}

// The RGB stem case (b = 2, C = 3, fp32 image, W even): per (channel, row parity) one 8-byte load brings the two
// horizontal neighbours, six loads per output pixel instead of twelve, consecutive lanes read consecutive 8 bytes.
// A thread makes PPT output pixels 256 apart in the flattened (image, row, column) order — all 12 / 24 loads in flight before the
// first conversion, every lane busy, 8 KB of contiguous output per workgroup and pass (round 4: one 112-pixel row per 128-thread
// workgroup ran 2.4 TB/s on 128 images; this form: tools/stem_micro.py).  Round 5: an fp16 image (BASELINE's input dtype, what
// bench.py feeds) takes the same kernel with 4-byte loads and PPT = 4 (it ran the generic row kernel: 44 us against 32 for fp32).
template <typename TS> struct Pair2;
template <> struct Pair2<float> {
    using T = float2;
    static __device__ __forceinline__ float lo(T v) { return v.x; }
    static __device__ __forceinline__ float hi(T v) { return v.y; }
};
template <> struct Pair2<half_t> {
    using T = unsigned;
    static __device__ __forceinline__ float lo(T v) { return (float)__builtin_bit_cast(half_t, (unsigned short)(v & 0xFFFFu)); }
    static __device__ __forceinline__ float hi(T v) { return (float)__builtin_bit_cast(half_t, (unsigned short)(v >> 16)); }
};

template <typename TS, typename TD, int PPT>
__global__ __launch_bounds__(256) void nchw3_to_nhwc_s2d2_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int H, int W,
                                                               int Cpad, long pixels) {
    using P2 = typename Pair2<TS>::T;
    constexpr int V = Chunk<TD>::N;
    const int H2 = H >> 1, W2 = W >> 1;
    const long HW = (long)H * W;
    const unsigned per_img = (unsigned)(H2 * W2);
    const long p0 = ((long)blockIdx.x * PPT) * 256 + threadIdx.x;
    P2 t[PPT][3][2];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const long p = p0 + 256 * q;
        const unsigned pu = p < pixels ? (unsigned)p : 0u;      // (pixels < 2^31: entry point)
        const unsigned n = pu / per_img, r = pu - n * per_img, h2 = r / (unsigned)W2, w2 = r - h2 * (unsigned)W2;
        const TS* sp = src + (long)n * 3 * HW + (long)(2 * h2) * W + 2 * w2;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) t[q][c][ph] = *reinterpret_cast<const P2*>(sp + c * HW + (long)ph * W);
    }
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const long p = p0 + 256 * q;
        if (p >= pixels) break;
        float v16[16];
#pragma unroll
        for (int e = 12; e < 16; ++e) v16[e] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                v16[(ph * 2 + 0) * 3 + c] = Pair2<TS>::lo(t[q][c][ph]);
                v16[(ph * 2 + 1) * 3 + c] = Pair2<TS>::hi(t[q][c][ph]);
            }
        TD* dp = dst + p * Cpad;
#pragma unroll
        for (int cg = 0; cg < 16 / V; ++cg) Chunk<TD>::store(dp + cg * V, v16 + cg * V);
    }
}

template <typename TS, typename TD>
__global__ void nhwc_to_nchw_kernel(const TS* __restrict__ src, int ld, TD* __restrict__ dst, int N, int C, int H,
                                    int W) {
    const long HW = (long)H * W, total = (long)N * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long hw = i % HW;
        const long nc = i / HW;
        const int c = (int)(nc % C);
        const long n = nc / C;
        dst[i] = (TD)(float)src[(n * HW + hw) * ld + c];
    }
}

__global__ void fold_bn_kernel(const float* gamma, const float* beta, const float* mean, const float* var,
                               const float* cbias, float eps, int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f, m = mean ? mean[c] : 0.f,
                v = var ? var[c] : 1.f;
    // same operation order as the oracle's folded form: scale = g / sqrt(v + eps)
    const float s = g / sqrtf(v + eps);
    scale[c] = s;
    shift[c] = b - m * s + (cbias ? cbias[c] * s : 0.f);
}

// ------------------------------------------------------------------------------------------
// MaxPool2d, -inf padding.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int x_ld,
                               int y_ld, int R, int S, int sh, int sw, int ph, int pw, int Ho, int Wo) {
    constexpr int V = Chunk<T>::N;
    const int nch = C / V;
    // one output row per workgroup column (scalar image / row index), threads split (pixel, channel chunk) in 32 bits
    const int per_row = Wo * nch, bx = (per_row + 255) / 256;
    const int row = blockIdx.x / bx, idx = (blockIdx.x - row * bx) * 256 + threadIdx.x;
    if (idx < per_row) {
        const int wo = idx / nch, cg = idx - wo * nch;
        const long n = row / Ho;
        const int ho = row - (int)n * Ho;
        float m[V];
#pragma unroll
        for (int e = 0; e < V; ++e) m[e] = -INFINITY;
        for (int r = 0; r < R; ++r) {
            const int hi = ho * sh - ph + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int s = 0; s < S; ++s) {
                const int wi = wo * sw - pw + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                float v[V];
                Chunk<T>::load(x + ((n * H + hi) * W + wi) * x_ld + cg * V, v);
#pragma unroll
                for (int e = 0; e < V; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        }
        Chunk<T>::store(y + ((n * Ho + ho) * Wo + wo) * y_ld + cg * V, m);
    }
}

// AvgPool2d, zero padding counted in the divisor (R * S always): nn.AvgPool2d(3, stride, 1) / (stride, stride, 0) of
// resnest.py:212-218, 250-256, 271-286.  Accumulation in window order, one division at the end (torch's CPU order).
template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int x_ld,
                               int y_ld, int R, int S, int sh, int sw, int ph, int pw, int Ho, int Wo) {
    constexpr int V = Chunk<T>::N;
    const int nch = C / V;
    const long total = (long)N * Ho * Wo * nch;
    const float inv = 1.f / (float)(R * S);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % nch);
        long p = i / nch;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho);
        const long n = p / Ho;
        float m[V];
#pragma unroll
        for (int e = 0; e < V; ++e) m[e] = 0.f;
        for (int r = 0; r < R; ++r) {
            const int hi = ho * sh - ph + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int s = 0; s < S; ++s) {
                const int wi = wo * sw - pw + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                float v[V];
                Chunk<T>::load(x + ((n * H + hi) * W + wi) * x_ld + cg * V, v);
#pragma unroll
                for (int e = 0; e < V; ++e) m[e] += v[e];
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e) m[e] *= inv;
        Chunk<T>::store(y + ((n * Ho + ho) * Wo + wo) * y_ld + cg * V, m);
    }
}

// Split attention of ResNeSt's SplatConv (resnest.py:147-166, rSoftmax :53-82).  x: [N][HW][radix * C], split r =
// channels [r*C, (r+1)*C).
//   radix_gap:  g[n][c] = mean_p sum_r x[n][p][r*C + c]                                   (:150-155)
//   split_attention: y[n][p][c] = sum_r a_r(n, c) * x[n][p][r*C + c], where for radix > 1
//       a_r(n, c) = softmax over r of logit[n][(k*radix + r)*cpg + c'],  c = k*cpg + c',  cpg = C / cardinality
//   (the reshape / transpose / softmax(axis=1) / reshape of rSoftmax), and for radix == 1  a = sigmoid(logit[n][c]).
// One workgroup per (image, slab of up to 32 channel chunks): the 256 threads are (pixel lane, chunk), every thread sums
// its pixels pl, pl + PL, ... of all radix splits, the pixel lanes are reduced through LDS.
template <typename T>
__global__ __launch_bounds__(256) void radix_gap_kernel(const T* __restrict__ x, T* __restrict__ g, int N, int HW, int C, int radix,
                                                        int x_ld, int g_ld, int cpb, int nslab) {
    constexpr int V = Chunk<T>::N;
    __shared__ float red[256][V];
    const int nch = C / V;
    const int pl_n = 256 / cpb;                      // pixel lanes
    const int slab = blockIdx.x % nslab;
    const long n = blockIdx.x / nslab;
    const int cl = threadIdx.x % cpb, pl = threadIdx.x / cpb;
    const int cg = slab * cpb + cl;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    if (pl < pl_n && cg < nch) {
        for (int k = pl; k < HW; k += pl_n) {
            for (int r = 0; r < radix; ++r) {
                float v[V];
                Chunk<T>::load(x + (n * HW + k) * x_ld + r * C + cg * V, v);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += v[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) red[threadIdx.x][e] = acc[e];
    __syncthreads();
    if (pl == 0 && cg < nch) {
        for (int q = 1; q < pl_n; ++q)
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += red[q * cpb + cl][e];
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] *= inv;
        Chunk<T>::store(g + n * g_ld + cg * V, acc);
    }
}

// att[n][r*C + c] (fp32, split order) from logit[n][(k*radix + r)*cpg + c'] (conv3's channel order): softmax over r, or
// sigmoid for radix 1 — one thread per (image, channel).
template <typename T>
__global__ void radix_softmax_kernel(const T* __restrict__ logit, float* __restrict__ att, int N, int C, int radix, int cardinality,
                                     int l_ld) {
    const int cpg = C / cardinality;
    const long total = (long)N * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long n = i / C;
        const int k = c / cpg, cc = c - k * cpg;
        const T* lp = logit + n * l_ld + (long)k * radix * cpg + cc;
        float* ap = att + n * (long)radix * C + c;
        if (radix == 1) {
            ap[0] = 1.f / (1.f + expf(-(float)lp[0]));
            continue;
        }
        float m = -INFINITY;
        for (int r = 0; r < radix; ++r) m = fmaxf(m, (float)lp[r * cpg]);
        float d = 0.f;
        for (int r = 0; r < radix; ++r) d += expf((float)lp[r * cpg] - m);
        for (int r = 0; r < radix; ++r) ap[(long)r * C] = expf((float)lp[r * cpg] - m) / d;
    }
}

template <typename T>
__global__ void split_attention_kernel(const T* __restrict__ x, const float* __restrict__ att, T* __restrict__ y, int N, int HW,
                                       int C, int radix, int x_ld, int y_ld) {
    constexpr int V = Chunk<T>::N;
    const int nch = C / V;
    const long total = (long)N * HW * nch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % nch);
        const long pix = i / nch;
        const long n = pix / HW;
        float out[V];
#pragma unroll
        for (int e = 0; e < V; ++e) out[e] = 0.f;
        for (int r = 0; r < radix; ++r) {
            float v[V];
            Chunk<T>::load(x + pix * x_ld + r * C + cg * V, v);
            const float* ap = att + (n * radix + r) * (long)C + cg * V;
#pragma unroll
            for (int e = 0; e < V; ++e) out[e] += ap[e] * v[e];
        }
        Chunk<T>::store(y + pix * y_ld + cg * V, out);
    }
}

template <typename T>
__global__ void global_avgpool_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int HW, int C, int x_ld,
                                      int y_ld) {
    constexpr int V = Chunk<T>::N;
    const int nch = C / V;
    const long total = (long)N * nch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % nch);
        const long n = i / nch;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        const T* p = x + n * HW * x_ld + cg * V;
        // seven loads in flight per lane (7 x 7 maps: resnet.py:295-296); the additions keep their pixel order
        int k = 0;
        for (; k + 7 <= HW; k += 7) {
            float v[7][V];
#pragma unroll
            for (int u = 0; u < 7; ++u) Chunk<T>::load(p + (long)(k + u) * x_ld, v[u]);
#pragma unroll
            for (int u = 0; u < 7; ++u)
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += v[u][e];
        }
        for (; k < HW; ++k) {
            float v[V];
            Chunk<T>::load(p + (long)k * x_ld, v);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += v[e];
        }
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] *= inv;
        Chunk<T>::store(y + n * y_ld + cg * V, acc);
    }
}

// nn.AdaptiveAvgPool2d((OH, OW)) (vgg.py:36-39): output (oh, ow) averages rows [floor(oh*H/OH), ceil((oh+1)*H/OH))
// and the same for columns — the definition torch / paddle / TLX share
template <typename T>
__global__ void adaptive_avgpool_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int OH,
                                        int OW, int x_ld, int y_ld) {
    constexpr int V = Chunk<T>::N;
    const int nch = C / V;
    const long total = (long)N * OH * OW * nch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % nch);
        long p = i / nch;
        const int ow = (int)(p % OW);
        p /= OW;
        const int oh = (int)(p % OH);
        const long n = p / OH;
        const int h0 = (oh * H) / OH, h1 = ((oh + 1) * H + OH - 1) / OH;
        const int w0 = (ow * W) / OW, w1 = ((ow + 1) * W + OW - 1) / OW;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) {
                float v[V];
                Chunk<T>::load(x + ((n * H + h) * W + w) * x_ld + cg * V, v);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += v[e];
            }
        const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] *= inv;
        Chunk<T>::store(y + ((n * OH + oh) * OW + ow) * y_ld + cg * V, acc);
    }
}

// ------------------------------------------------------------------------------------------
// y = act(x*scale + shift (+res)) (+res)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void affine_act_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                  const float* __restrict__ shift, const T* __restrict__ res, T* __restrict__ y,
                                  long rows, int C, int x_ld, int res_ld, int y_ld, int act, float ap,
                                  unsigned flags) {
    constexpr int V = Chunk<T>::N;
    const int nch = C / V;
    const long total = rows * nch;
    const bool res_after = (flags & TLXMI_EPI_RES_AFTER_ACT) != 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % nch);
        const long row = i / nch;
        float v[V], rv[V];
        Chunk<T>::load(x + row * x_ld + cg * V, v);
        if (res) Chunk<T>::load(res + row * res_ld + cg * V, rv);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float t = v[e];
            if (scale) t *= scale[cg * V + e];
            if (shift) t += shift[cg * V + e];
            if (res && !res_after) t += rv[e];
            t = apply_act(t, act, ap);
            if (res && res_after) t += rv[e];
            v[e] = t;
        }
        Chunk<T>::store(y + row * y_ld + cg * V, v);
    }
}

// ------------------------------------------------------------------------------------------
// Depthwise conv: one lane per (output pixel, channel chunk); taps re-read through L1/L2.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void dwconv_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ scale,
                              const float* __restrict__ shift, T* __restrict__ y, tlxmi_dwconv2d_desc d) {
    constexpr int V = Chunk<T>::N;
    const int nch = d.C / V;
    const long total = (long)d.N * d.Ho * d.Wo * nch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % nch);
        long p = i / nch;
        const int wo = (int)(p % d.Wo);
        p /= d.Wo;
        const int ho = (int)(p % d.Ho);
        const long n = p / d.Ho;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int r = 0; r < d.R; ++r) {
            const int hi = ho * d.stride_h - d.pad_h + r * d.dil_h;
            if ((unsigned)hi >= (unsigned)d.H) continue;
            for (int s = 0; s < d.S; ++s) {
                const int wi = wo * d.stride_w - d.pad_w + s * d.dil_w;
                if ((unsigned)wi >= (unsigned)d.W) continue;
                float xv[V], wv[V];
                Chunk<T>::load(x + ((n * d.H + hi) * d.W + wi) * d.x_ld + cg * V, xv);
                Chunk<T>::load(w + (long)(r * d.S + s) * d.C + cg * V, wv);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] = fmaf(xv[e], wv[e], acc[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float t = acc[e];
            if (scale) t *= scale[cg * V + e];
            if (shift) t += shift[cg * V + e];
            acc[e] = apply_act(t, d.act, d.act_param);
        }
        Chunk<T>::store(y + ((n * d.Ho + ho) * d.Wo + wo) * d.y_ld + cg * V, acc);
    }
}

// Depthwise conv, register-tiled: a thread computes TW consecutive output pixels of one row for 8 (4) channels.  Per
// filter row it loads the (TW-1)*SW + S input chunks its outputs share and the S filter chunks once — 2.7x (3x3, stride
// 1, TW = 4) fewer load instructions per output than one-pixel-per-thread; same tap order, so the same fp32 sums.
template <typename T, int TW, int S, int SW>
__global__ __launch_bounds__(256) void dwconv_strip_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           T* __restrict__ y, tlxmi_dwconv2d_desc d) {
    constexpr int V = Chunk<T>::N;
    constexpr int NI = (TW - 1) * SW + S;
    // operands stay in their storage type (16 bytes = 4 registers per chunk); the fp16 -> fp32 conversion folds into the
    // mixed-precision FMA
    typedef T rawv __attribute__((ext_vector_type(V)));
    const int nch = d.C / V;
    const int wt = (d.Wo + TW - 1) / TW;
    // one output row per workgroup column: the (image, row) pair comes from the block index (scalar arithmetic, once),
    // the thread only splits its position inside the row into (strip, channel chunk)
    const int per_row = wt * nch, bx = (per_row + 255) / 256;
    const int row = blockIdx.x / bx, idx = (blockIdx.x - row * bx) * 256 + threadIdx.x;
    if (idx < per_row) {
        const int ws = idx / nch, cg = idx - ws * nch;
        const long n = row / d.Ho;
        const int ho = row - (int)n * d.Ho;
        const int wo0 = ws * TW, wi0 = wo0 * SW - d.pad_w;
        float sc[V], sf[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            sc[e] = scale ? scale[cg * V + e] : 1.f;
            sf[e] = shift ? shift[cg * V + e] : 0.f;
        }
        float acc[TW][V];
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int e = 0; e < V; ++e) acc[t][e] = 0.f;
        for (int r = 0; r < d.R; ++r) {
            const int hi = ho * d.stride_h - d.pad_h + r;
            if ((unsigned)hi >= (unsigned)d.H) continue;
            rawv xin[NI], wv[S];
            const T* xr = x + ((n * d.H + hi) * d.W) * d.x_ld + cg * V;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int wi = wi0 + j;
                rawv z;
#pragma unroll
                for (int e = 0; e < V; ++e) z[e] = (T)0.f;
                xin[j] = (unsigned)wi < (unsigned)d.W ? *reinterpret_cast<const rawv*>(xr + (long)wi * d.x_ld) : z;
            }
#pragma unroll
            for (int s_ = 0; s_ < S; ++s_) wv[s_] = *reinterpret_cast<const rawv*>(w + (long)(r * S + s_) * d.C + cg * V);
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int s_ = 0; s_ < S; ++s_)
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[t][e] = fmaf((float)xin[t * SW + s_][e], (float)wv[s_][e], acc[t][e]);
        }
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            if (wo0 + t >= d.Wo) break;
#pragma unroll
            for (int e = 0; e < V; ++e) acc[t][e] = apply_act(acc[t][e] * sc[e] + sf[e], d.act, d.act_param);
            Chunk<T>::store(y + ((n * d.Ho + ho) * d.Wo + wo0 + t) * d.y_ld + cg * V, acc[t]);
        }
    }
}

// Squeeze-Excitation gating: y[n][p][c] = x[n][p][c] * s[n][c]
template <typename T>
__global__ void scale_channels_kernel(const T* __restrict__ x, const T* __restrict__ s, T* __restrict__ y, int N,
                                      int HW, int C, int x_ld, int s_ld, int y_ld) {
    constexpr int V = Chunk<T>::N;
    const int nch = C / V;
    // the image index comes from the block index (scalar); a thread splits its position inside the image in 32 bits
    const long per_img = (long)HW * nch;
    const int bx = (int)((per_img + 255) / 256);
    const long n = blockIdx.x / bx;
    const long idx = (long)(blockIdx.x - n * bx) * 256 + threadIdx.x;
    if (idx < per_img) {
        const int pl = (int)idx / nch, cg = (int)idx - pl * nch;
        const long p = n * HW + pl;
        float xv[V], sv[V];
        Chunk<T>::load(x + p * x_ld + cg * V, xv);
        Chunk<T>::load(s + n * s_ld + cg * V, sv);
#pragma unroll
        for (int e = 0; e < V; ++e) xv[e] *= sv[e];
        Chunk<T>::store(y + p * y_ld + cg * V, xv);
    }
}

// nearest x2 upsample into a channel window of a wider buffer
template <typename T>
__global__ void upsample2x_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int x_ld,
                                  int y_ld, int c_off) {
    constexpr int V = Chunk<T>::N;
    const int nch = C / V, Ho = 2 * H, Wo = 2 * W;
    const long total = (long)N * Ho * Wo * nch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % nch);
        long p = i / nch;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho);
        const long n = p / Ho;
        u32x4 v = *reinterpret_cast<const u32x4*>(x + ((n * H + (ho >> 1)) * W + (wo >> 1)) * x_ld + cg * V);
        *reinterpret_cast<u32x4*>(y + ((n * Ho + ho) * Wo + wo) * y_ld + c_off + cg * V) = v;
    }
}

template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, int C, int x_ld,
                                     int y_ld) {
    constexpr int V = Chunk<T>::N;
    const int nch = C / V;
    const long total = rows * nch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % nch);
        const long row = i / nch;
        *reinterpret_cast<u32x4*>(y + row * y_ld + cg * V) = *reinterpret_cast<const u32x4*>(x + row * x_ld + cg * V);
    }
}

// argmax over last dim, one wave per row, ties -> lowest index (torch CPU behaviour)
template <typename T>
__global__ void argmax_kernel(const T* __restrict__ x, long rows, int C, int x_ld, int64_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    bool nan_seen = false;
    for (int c = lane; c < C; c += 64) {
        const float v = (float)x[row * x_ld + c];
        if (v != v) { if (!nan_seen) { nan_seen = true; best = v; bi = c; } continue; }
        if (!nan_seen && (v > best || (v == best && c < bi))) { best = v; bi = c; }
    }
    // NaN rows follow torch: first NaN wins.  Encode as +inf for the reduction.
    float key = nan_seen ? INFINITY : best;
    int prio = nan_seen ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ok = __shfl_xor(key, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const int op = __shfl_xor(prio, o, 64);
        const bool take = (op > prio) || (op == prio && (ok > key || (ok == key && oi < bi)));
        if (take) { key = ok; bi = oi; prio = op; }
    }
    if (lane == 0) out[row] = bi == 0x7fffffff ? 0 : bi;
}

}  // namespace tlxmi

using namespace tlxmi;

#define DT_OK(dt) ((dt) == TLXMI_F16 || (dt) == TLXMI_F32)
#define VECN(dt) (16 / (int)elt_size(dt))

extern "C" int tlxmi_nchw_to_nhwc(const void* src, int sdt, void* dst, int ddt, int N, int C, int H, int W, int Cpad,
                                  void* stream) {
    TLXMI_REQUIRE(src && dst, TLXMI_ERR_BAD_ARG, "nchw_to_nhwc: null buffer");
    TLXMI_REQUIRE(DT_OK(sdt) && DT_OK(ddt), TLXMI_ERR_BAD_ARG, "nchw_to_nhwc: bad dtype");
    TLXMI_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, TLXMI_ERR_BAD_ARG, "nchw_to_nhwc: bad extent");
    TLXMI_REQUIRE(Cpad % VECN(ddt) == 0 && aligned16(dst), TLXMI_ERR_ALIGNMENT,
                  "nchw_to_nhwc: Cpad=%d must be a whole number of 16-byte chunks", Cpad);
    const long work = (long)N * H * W * (Cpad / VECN(ddt));
    dim3 g(grid_for(work)), b(256);
    hipStream_t st = as_stream(stream);
    if (sdt == TLXMI_F32 && ddt == TLXMI_F16)
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, half_t>), g, b, 0, st, (const float*)src, (half_t*)dst, N, C, H, W, Cpad);
    else if (sdt == TLXMI_F32 && ddt == TLXMI_F32)
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, float>), g, b, 0, st, (const float*)src, (float*)dst, N, C, H, W, Cpad);
    else if (sdt == TLXMI_F16 && ddt == TLXMI_F16)
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<half_t, half_t>), g, b, 0, st, (const half_t*)src, (half_t*)dst, N, C, H, W, Cpad);
    else
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<half_t, float>), g, b, 0, st, (const half_t*)src, (float*)dst, N, C, H, W, Cpad);
    return check_launch("nchw_to_nhwc");
}

// Patch rows for a patch-embedding conv run as a Linear (vision_transformer.py:197-204, 321-323): src [N][C][H][W] ->
// dst [N][lead + (H/ps)(W/ps)][C*ps*ps], element (c*ps + ky)*ps + kx of patch (py, px) = src[n][c][py*ps + ky][px*ps + kx] — the
// order of the conv filter [Cout][C][ps][ps] flattened, so the filter IS the Linear weight.  The `lead` rows in front of every
// image's patches (the cls token's slot) are zero.  A thread moves 8 elements; a wave = ps/8 x ps x (512 / ps^2) items = whole
// 128-byte lines of the image on the read side (two neighbouring patches' 64 B of one image row) and 512 contiguous bytes per
// patch and channel on the write side.
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void patchify_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int N, int C, int H, int W,
                                                       int ps, int lead, long items, long zero_items) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int Hp = H / ps, Wp = W / ps, G = ps >> 3, K = C * ps * ps, rows = lead + Hp * Wp;
    if (i >= items) {      // the zero rows: K / 8 chunks per lead row
        const long z = i - items;
        if (z >= zero_items) return;
        const int kc = K >> 3;
        const long r = z / kc;
        const int q = (int)(z - r * kc);
        const long n = r / lead;
        const int l = (int)(r - n * lead);
        TD* d = dst + (n * rows + l) * K + 8 * q;
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = (TD)0.f;
        return;
    }
    // i = ((((n * Hp + py) * C + c) * Wp + px) * ps + ky) * G + g
    long t = i;
    const int g = (int)(t % G); t /= G;
    const int ky = (int)(t % ps); t /= ps;
    const int px = (int)(t % Wp); t /= Wp;
    const int c = (int)(t % C); t /= C;
    const int py = (int)(t % Hp);
    const long n = t / Hp;
    const TS* sp = src + ((n * C + c) * H + py * ps + ky) * (long)W + px * ps + 8 * g;
    TD* dp = dst + (n * rows + lead + py * Wp + px) * K + (c * ps + ky) * ps + 8 * g;
    float v[8];
    if constexpr (sizeof(TS) == 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    } else {
        const half8v h = *reinterpret_cast<const half8v*>(sp);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
    }
    if constexpr (sizeof(TD) == 2) {
        half8v h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (half_t)v[e];
        *reinterpret_cast<half8v*>(dp) = h;
    } else {
        *reinterpret_cast<f32x4*>(dp) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(dp + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
}

extern "C" int tlxmi_patchify(const void* src, int sdt, void* dst, int ddt, int N, int C, int H, int W, int ps, int lead,
                              void* stream) {
    TLXMI_REQUIRE(src && dst, TLXMI_ERR_BAD_ARG, "patchify: null buffer");
    TLXMI_REQUIRE(DT_OK(sdt) && DT_OK(ddt), TLXMI_ERR_BAD_ARG, "patchify: bad dtype");
    TLXMI_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && ps >= 8 && ps % 8 == 0 && H % ps == 0 && W % ps == 0 && lead >= 0,
                  TLXMI_ERR_BAD_ARG, "patchify: H=%d W=%d must be multiples of the patch %d, itself a multiple of 8", H, W, ps);
    TLXMI_REQUIRE(aligned16(src) && aligned16(dst), TLXMI_ERR_ALIGNMENT, "patchify: buffers must be 16-byte aligned");
    const long items = (long)N * C * (H / ps) * (W / ps) * ps * (ps / 8);
    const long zero_items = (long)N * lead * ((long)C * ps * ps / 8);
    const long blocks = (items + zero_items + 255) / 256;
    TLXMI_REQUIRE(blocks < (1l << 31), TLXMI_ERR_UNSUPPORTED, "patchify: too many elements");
    const dim3 g((unsigned)blocks), blk(256);
    hipStream_t st = as_stream(stream);
    if (sdt == TLXMI_F32 && ddt == TLXMI_F16)
        hipLaunchKernelGGL((patchify_kernel<float, half_t>), g, blk, 0, st, (const float*)src, (half_t*)dst, N, C, H, W, ps, lead, items, zero_items);
    else if (sdt == TLXMI_F32 && ddt == TLXMI_F32)
        hipLaunchKernelGGL((patchify_kernel<float, float>), g, blk, 0, st, (const float*)src, (float*)dst, N, C, H, W, ps, lead, items, zero_items);
    else if (sdt == TLXMI_F16 && ddt == TLXMI_F16)
        hipLaunchKernelGGL((patchify_kernel<half_t, half_t>), g, blk, 0, st, (const half_t*)src, (half_t*)dst, N, C, H, W, ps, lead, items, zero_items);
    else
        hipLaunchKernelGGL((patchify_kernel<half_t, float>), g, blk, 0, st, (const half_t*)src, (float*)dst, N, C, H, W, ps, lead, items, zero_items);
    return check_launch("patchify");
}

extern "C" int tlxmi_nchw_to_nhwc_s2d(const void* src, int sdt, void* dst, int ddt, int N, int C, int H, int W, int b,
                                      int Cpad, void* stream) {
    TLXMI_REQUIRE(src && dst, TLXMI_ERR_BAD_ARG, "nchw_to_nhwc_s2d: null buffer");
    TLXMI_REQUIRE(DT_OK(sdt) && DT_OK(ddt), TLXMI_ERR_BAD_ARG, "nchw_to_nhwc_s2d: bad dtype");
    TLXMI_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && b >= 1 && H % b == 0 && W % b == 0 && Cpad >= b * b * C,
                  TLXMI_ERR_BAD_ARG, "nchw_to_nhwc_s2d: H=%d W=%d must be multiples of b=%d, Cpad >= b*b*C", H, W, b);
    TLXMI_REQUIRE(Cpad % VECN(ddt) == 0 && aligned16(dst), TLXMI_ERR_ALIGNMENT,
                  "nchw_to_nhwc_s2d: Cpad=%d must be a whole number of 16-byte chunks", Cpad);
    const int cb = (W / b + 127) / 128;
    const long blocks = (long)N * (H / b) * cb;
    TLXMI_REQUIRE(blocks < (1l << 31), TLXMI_ERR_UNSUPPORTED, "nchw_to_nhwc_s2d: too many rows");
    dim3 g((unsigned)blocks), blk(128);
    hipStream_t st = as_stream(stream);
    if (sdt == TLXMI_F32 && b == 2 && C == 3 && Cpad == 16 && W % 2 == 0 && ((uintptr_t)src % 8) == 0 && (long)N * (H / 2) * (W / 2) < (1l << 31)) {
        constexpr int PPT = 2;
        const long pixels = (long)N * (H / 2) * (W / 2);
        const dim3 g2((unsigned)((pixels + 256 * PPT - 1) / (256 * PPT))), b2(256);
        if (ddt == TLXMI_F16) hipLaunchKernelGGL((nchw3_to_nhwc_s2d2_kernel<float, half_t, PPT>), g2, b2, 0, st, (const float*)src, (half_t*)dst, H, W, Cpad, pixels);
        else hipLaunchKernelGGL((nchw3_to_nhwc_s2d2_kernel<float, float, PPT>), g2, b2, 0, st, (const float*)src, (float*)dst, H, W, Cpad, pixels);
        return check_launch("nchw_to_nhwc_s2d");
    }
    if (sdt == TLXMI_F16 && ddt == TLXMI_F16 && b == 2 && C == 3 && Cpad == 16 && W % 2 == 0 && ((uintptr_t)src % 4) == 0 && (long)N * (H / 2) * (W / 2) < (1l << 31)) {
        constexpr int PPT = 4;
        const long pixels = (long)N * (H / 2) * (W / 2);
        const dim3 g2((unsigned)((pixels + 256 * PPT - 1) / (256 * PPT))), b2(256);
        hipLaunchKernelGGL((nchw3_to_nhwc_s2d2_kernel<half_t, half_t, PPT>), g2, b2, 0, st, (const half_t*)src, (half_t*)dst, H, W, Cpad, pixels);
        return check_launch("nchw_to_nhwc_s2d");
    }
    {
        const size_t lds = (size_t)(W / b) * Cpad * elt_size(ddt);
        if (b >= 4 && lds <= 64 * 1024) {
            const dim3 gr((unsigned)((long)N * (H / b)));
            if (sdt == TLXMI_F32 && ddt == TLXMI_F16)
                hipLaunchKernelGGL((nchw_to_nhwc_s2d_rows_kernel<float, half_t>), gr, dim3(256), lds, st, (const float*)src, (half_t*)dst, C, H, W, b, Cpad);
            else if (sdt == TLXMI_F32 && ddt == TLXMI_F32)
                hipLaunchKernelGGL((nchw_to_nhwc_s2d_rows_kernel<float, float>), gr, dim3(256), lds, st, (const float*)src, (float*)dst, C, H, W, b, Cpad);
            else if (sdt == TLXMI_F16 && ddt == TLXMI_F16)
                hipLaunchKernelGGL((nchw_to_nhwc_s2d_rows_kernel<half_t, half_t>), gr, dim3(256), lds, st, (const half_t*)src, (half_t*)dst, C, H, W, b, Cpad);
            else
                hipLaunchKernelGGL((nchw_to_nhwc_s2d_rows_kernel<half_t, float>), gr, dim3(256), lds, st, (const half_t*)src, (float*)dst, C, H, W, b, Cpad);
            return check_launch("nchw_to_nhwc_s2d");
        }
    }
    if (sdt == TLXMI_F32 && ddt == TLXMI_F16)
        hipLaunchKernelGGL((nchw_to_nhwc_s2d_kernel<float, half_t>), g, blk, 0, st, (const float*)src, (half_t*)dst, N, C, H, W, b, Cpad, cb);
    else if (sdt == TLXMI_F32 && ddt == TLXMI_F32)
        hipLaunchKernelGGL((nchw_to_nhwc_s2d_kernel<float, float>), g, blk, 0, st, (const float*)src, (float*)dst, N, C, H, W, b, Cpad, cb);
    else if (sdt == TLXMI_F16 && ddt == TLXMI_F16)
        hipLaunchKernelGGL((nchw_to_nhwc_s2d_kernel<half_t, half_t>), g, blk, 0, st, (const half_t*)src, (half_t*)dst, N, C, H, W, b, Cpad, cb);
    else
        hipLaunchKernelGGL((nchw_to_nhwc_s2d_kernel<half_t, float>), g, blk, 0, st, (const half_t*)src, (float*)dst, N, C, H, W, b, Cpad, cb);
    return check_launch("nchw_to_nhwc_s2d");
}

extern "C" int tlxmi_nhwc_to_nchw(const void* src, int sdt, int ld, void* dst, int ddt, int N, int C, int H, int W,
                                  void* stream) {
    TLXMI_REQUIRE(src && dst, TLXMI_ERR_BAD_ARG, "nhwc_to_nchw: null buffer");
    TLXMI_REQUIRE(DT_OK(sdt) && DT_OK(ddt), TLXMI_ERR_BAD_ARG, "nhwc_to_nchw: bad dtype");
    TLXMI_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && ld >= C, TLXMI_ERR_BAD_ARG, "nhwc_to_nchw: bad extent");
    const long work = (long)N * C * H * W;
    dim3 g(grid_for(work)), b(256);
    hipStream_t st = as_stream(stream);
    if (sdt == TLXMI_F16 && ddt == TLXMI_F32)
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<half_t, float>), g, b, 0, st, (const half_t*)src, ld, (float*)dst, N, C, H, W);
    else if (sdt == TLXMI_F32 && ddt == TLXMI_F32)
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<float, float>), g, b, 0, st, (const float*)src, ld, (float*)dst, N, C, H, W);
    else if (sdt == TLXMI_F16 && ddt == TLXMI_F16)
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<half_t, half_t>), g, b, 0, st, (const half_t*)src, ld, (half_t*)dst, N, C, H, W);
    else
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<float, half_t>), g, b, 0, st, (const float*)src, ld, (half_t*)dst, N, C, H, W);
    return check_launch("nhwc_to_nchw");
}

extern "C" int tlxmi_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var,
                             const float* cbias, float eps, int C, float* scale, float* shift, void* stream) {
    TLXMI_REQUIRE(scale && shift && C > 0, TLXMI_ERR_BAD_ARG, "fold_bn: bad argument");
    hipLaunchKernelGGL(fold_bn_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), gamma, beta, mean, var,
                       cbias, eps, C, scale, shift);
    return check_launch("fold_bn");
}

#define REQUIRE_CHUNKED(name, dt, C, ...)                                                             \
    do {                                                                                              \
        TLXMI_REQUIRE(DT_OK(dt), TLXMI_ERR_BAD_ARG, name ": bad dtype");                              \
        TLXMI_REQUIRE((C) > 0 && (C) % VECN(dt) == 0, TLXMI_ERR_ALIGNMENT,                            \
                      name ": C=%d must be a whole number of 16-byte chunks", (C));                   \
        const int lds_[] = {__VA_ARGS__};                                                             \
        for (int ld_ : lds_)                                                                          \
            TLXMI_REQUIRE(ld_ >= (C) && ld_ % VECN(dt) == 0, TLXMI_ERR_ALIGNMENT, name ": bad pixel stride %d", ld_); \
    } while (0)

extern "C" int tlxmi_maxpool2d(const void* x, void* y, int dt, int N, int H, int W, int C, int x_ld, int y_ld, int R,
                               int S, int sh, int sw, int ph, int pw, int Ho, int Wo, void* stream) {
    TLXMI_REQUIRE(x && y, TLXMI_ERR_BAD_ARG, "maxpool2d: null buffer");
    REQUIRE_CHUNKED("maxpool2d", dt, C, x_ld, y_ld);
    TLXMI_REQUIRE(aligned16(x) && aligned16(y), TLXMI_ERR_ALIGNMENT, "maxpool2d: buffers must be 16-byte aligned");
    TLXMI_REQUIRE(N > 0 && H > 0 && W > 0 && R > 0 && S > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0, TLXMI_ERR_BAD_ARG,
                  "maxpool2d: bad extent");
    TLXMI_REQUIRE(ph < R && pw < S, TLXMI_ERR_BAD_ARG, "maxpool2d: padding must be smaller than the window");
    TLXMI_REQUIRE(Ho == (H + 2 * ph - R) / sh + 1 && Wo == (W + 2 * pw - S) / sw + 1, TLXMI_ERR_BAD_ARG,
                  "maxpool2d: output extent mismatch");
    const long mp_blocks = (long)N * Ho * (((long)Wo * (C / VECN(dt)) + 255) / 256);
    TLXMI_REQUIRE(mp_blocks < (1l << 31), TLXMI_ERR_UNSUPPORTED, "maxpool2d: too many rows");
    dim3 g((unsigned)mp_blocks), b(256);
    if (dt == TLXMI_F16)
        hipLaunchKernelGGL((maxpool_kernel<half_t>), g, b, 0, as_stream(stream), (const half_t*)x, (half_t*)y, N, H, W, C, x_ld, y_ld, R, S, sh, sw, ph, pw, Ho, Wo);
    else
        hipLaunchKernelGGL((maxpool_kernel<float>), g, b, 0, as_stream(stream), (const float*)x, (float*)y, N, H, W, C, x_ld, y_ld, R, S, sh, sw, ph, pw, Ho, Wo);
    return check_launch("maxpool2d");
}

extern "C" int tlxmi_avgpool2d(const void* x, void* y, int dt, int N, int H, int W, int C, int x_ld, int y_ld, int R,
                               int S, int sh, int sw, int ph, int pw, int Ho, int Wo, void* stream) {
    TLXMI_REQUIRE(x && y, TLXMI_ERR_BAD_ARG, "avgpool2d: null buffer");
    REQUIRE_CHUNKED("avgpool2d", dt, C, x_ld, y_ld);
    TLXMI_REQUIRE(aligned16(x) && aligned16(y), TLXMI_ERR_ALIGNMENT, "avgpool2d: buffers must be 16-byte aligned");
    TLXMI_REQUIRE(N > 0 && H > 0 && W > 0 && R > 0 && S > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0, TLXMI_ERR_BAD_ARG,
                  "avgpool2d: bad extent");
    TLXMI_REQUIRE(2 * ph <= R && 2 * pw <= S, TLXMI_ERR_BAD_ARG, "avgpool2d: padding must be at most half the window");
    TLXMI_REQUIRE(Ho == (H + 2 * ph - R) / sh + 1 && Wo == (W + 2 * pw - S) / sw + 1 && Ho > 0 && Wo > 0, TLXMI_ERR_BAD_ARG,
                  "avgpool2d: output extent mismatch");
    const long work = (long)N * Ho * Wo * (C / VECN(dt));
    dim3 g(grid_for(work)), b(256);
    if (dt == TLXMI_F16)
        hipLaunchKernelGGL((avgpool_kernel<half_t>), g, b, 0, as_stream(stream), (const half_t*)x, (half_t*)y, N, H, W, C, x_ld, y_ld, R, S, sh, sw, ph, pw, Ho, Wo);
    else
        hipLaunchKernelGGL((avgpool_kernel<float>), g, b, 0, as_stream(stream), (const float*)x, (float*)y, N, H, W, C, x_ld, y_ld, R, S, sh, sw, ph, pw, Ho, Wo);
    return check_launch("avgpool2d");
}

extern "C" int tlxmi_radix_gap(const void* x, void* g, int dt, int N, int HW, int C, int radix, int x_ld, int g_ld, void* stream) {
    TLXMI_REQUIRE(x && g && N > 0 && HW > 0 && radix >= 1, TLXMI_ERR_BAD_ARG, "radix_gap: bad argument");
    REQUIRE_CHUNKED("radix_gap", dt, C, g_ld);
    TLXMI_REQUIRE(x_ld >= radix * C && x_ld % VECN(dt) == 0 && aligned16(x) && aligned16(g), TLXMI_ERR_ALIGNMENT, "radix_gap: bad stride / alignment");
    const int nch = C / VECN(dt);
    const int cpb = nch < 32 ? nch : 32, nslab = (nch + cpb - 1) / cpb;
    dim3 gr((unsigned)((long)N * nslab)), b(256);
    if (dt == TLXMI_F16)
        hipLaunchKernelGGL((radix_gap_kernel<half_t>), gr, b, 0, as_stream(stream), (const half_t*)x, (half_t*)g, N, HW, C, radix, x_ld, g_ld, cpb, nslab);
    else
        hipLaunchKernelGGL((radix_gap_kernel<float>), gr, b, 0, as_stream(stream), (const float*)x, (float*)g, N, HW, C, radix, x_ld, g_ld, cpb, nslab);
    return check_launch("radix_gap");
}

extern "C" int tlxmi_split_attention(const void* x, const void* logit, float* att_ws, void* y, int dt, int N, int HW, int C,
                                     int radix, int cardinality, int x_ld, int l_ld, int y_ld, void* stream) {
    TLXMI_REQUIRE(x && logit && att_ws && y && N > 0 && HW > 0 && radix >= 1 && cardinality >= 1, TLXMI_ERR_BAD_ARG, "split_attention: bad argument");
    REQUIRE_CHUNKED("split_attention", dt, C, y_ld);
    TLXMI_REQUIRE(C % cardinality == 0, TLXMI_ERR_BAD_ARG, "split_attention: C=%d is not divisible by cardinality=%d", C, cardinality);
    TLXMI_REQUIRE(x_ld >= radix * C && x_ld % VECN(dt) == 0 && l_ld >= radix * C && aligned16(x) && aligned16(y) && aligned16(att_ws),
                  TLXMI_ERR_ALIGNMENT, "split_attention: bad stride / alignment");
    hipStream_t st = as_stream(stream);
    {
        dim3 g(grid_for((long)N * C)), b(256);
        if (dt == TLXMI_F16) hipLaunchKernelGGL((radix_softmax_kernel<half_t>), g, b, 0, st, (const half_t*)logit, att_ws, N, C, radix, cardinality, l_ld);
        else hipLaunchKernelGGL((radix_softmax_kernel<float>), g, b, 0, st, (const float*)logit, att_ws, N, C, radix, cardinality, l_ld);
    }
    const long work = (long)N * HW * (C / VECN(dt));
    dim3 g(grid_for(work)), b(256);
    if (dt == TLXMI_F16)
        hipLaunchKernelGGL((split_attention_kernel<half_t>), g, b, 0, st, (const half_t*)x, att_ws, (half_t*)y, N, HW, C, radix, x_ld, y_ld);
    else
        hipLaunchKernelGGL((split_attention_kernel<float>), g, b, 0, st, (const float*)x, att_ws, (float*)y, N, HW, C, radix, x_ld, y_ld);
    return check_launch("split_attention");
}

extern "C" int tlxmi_global_avgpool(const void* x, void* y, int dt, int N, int HW, int C, int x_ld, int y_ld,
                                    void* stream) {
    TLXMI_REQUIRE(x && y && N > 0 && HW > 0, TLXMI_ERR_BAD_ARG, "global_avgpool: bad argument");
    REQUIRE_CHUNKED("global_avgpool", dt, C, x_ld, y_ld);
    TLXMI_REQUIRE(aligned16(x) && aligned16(y), TLXMI_ERR_ALIGNMENT, "global_avgpool: buffers must be 16-byte aligned");
    if (HW >= 64) {   // large maps (Squeeze-Excitation pools of MobileNetV3 / EfficientNet): pixel-parallel workgroups
        const int nch = C / VECN(dt);
        const int cpb = nch < 32 ? nch : 32, nslab = (nch + cpb - 1) / cpb;
        dim3 gr((unsigned)((long)N * nslab)), bb(256);
        if (dt == TLXMI_F16)
            hipLaunchKernelGGL((radix_gap_kernel<half_t>), gr, bb, 0, as_stream(stream), (const half_t*)x, (half_t*)y, N, HW, C, 1, x_ld, y_ld, cpb, nslab);
        else
            hipLaunchKernelGGL((radix_gap_kernel<float>), gr, bb, 0, as_stream(stream), (const float*)x, (float*)y, N, HW, C, 1, x_ld, y_ld, cpb, nslab);
        return check_launch("global_avgpool");
    }
    const long work = (long)N * (C / VECN(dt));
    dim3 g(grid_for(work, 64)), b(64);
    if (dt == TLXMI_F16)
        hipLaunchKernelGGL((global_avgpool_kernel<half_t>), g, b, 0, as_stream(stream), (const half_t*)x, (half_t*)y, N, HW, C, x_ld, y_ld);
    else
        hipLaunchKernelGGL((global_avgpool_kernel<float>), g, b, 0, as_stream(stream), (const float*)x, (float*)y, N, HW, C, x_ld, y_ld);
    return check_launch("global_avgpool");
}

extern "C" int tlxmi_adaptive_avgpool2d(const void* x, void* y, int dt, int N, int H, int W, int C, int OH, int OW,
                                        int x_ld, int y_ld, void* stream) {
    TLXMI_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, TLXMI_ERR_BAD_ARG, "adaptive_avgpool2d: bad argument");
    REQUIRE_CHUNKED("adaptive_avgpool2d", dt, C, x_ld, y_ld);
    TLXMI_REQUIRE(aligned16(x) && aligned16(y), TLXMI_ERR_ALIGNMENT, "adaptive_avgpool2d: buffers must be 16-byte aligned");
    const long work = (long)N * OH * OW * (C / VECN(dt));
    dim3 g(grid_for(work)), b(256);
    if (dt == TLXMI_F16)
        hipLaunchKernelGGL((adaptive_avgpool_kernel<half_t>), g, b, 0, as_stream(stream), (const half_t*)x, (half_t*)y, N, H, W, C, OH, OW, x_ld, y_ld);
    else
        hipLaunchKernelGGL((adaptive_avgpool_kernel<float>), g, b, 0, as_stream(stream), (const float*)x, (float*)y, N, H, W, C, OH, OW, x_ld, y_ld);
    return check_launch("adaptive_avgpool2d");
}

extern "C" int tlxmi_affine_act(const void* x, const float* scale, const float* shift, const void* res, void* y,
                                int dt, int64_t rows, int C, int x_ld, int res_ld, int y_ld, int act, float ap,
                                uint32_t flags, void* stream) {
    TLXMI_REQUIRE(x && y && rows > 0, TLXMI_ERR_BAD_ARG, "affine_act: bad argument");
    REQUIRE_CHUNKED("affine_act", dt, C, x_ld, y_ld);
    TLXMI_REQUIRE(!res || (res_ld >= C && res_ld % VECN(dt) == 0 && aligned16(res)), TLXMI_ERR_ALIGNMENT, "affine_act: bad residual stride");
    TLXMI_REQUIRE(aligned16(x) && aligned16(y), TLXMI_ERR_ALIGNMENT, "affine_act: buffers must be 16-byte aligned");
    TLXMI_REQUIRE(act >= TLXMI_ACT_NONE && act <= TLXMI_ACT_SILU, TLXMI_ERR_BAD_ARG, "affine_act: bad act");
    const long work = rows * (C / VECN(dt));
    dim3 g(grid_for(work)), b(256);
    if (dt == TLXMI_F16)
        hipLaunchKernelGGL((affine_act_kernel<half_t>), g, b, 0, as_stream(stream), (const half_t*)x, scale, shift, (const half_t*)res, (half_t*)y, (long)rows, C, x_ld, res_ld, y_ld, act, ap, flags);
    else
        hipLaunchKernelGGL((affine_act_kernel<float>), g, b, 0, as_stream(stream), (const float*)x, scale, shift, (const float*)res, (float*)y, (long)rows, C, x_ld, res_ld, y_ld, act, ap, flags);
    return check_launch("affine_act");
}

extern "C" int tlxmi_dwconv2d(const tlxmi_dwconv2d_desc* d, const void* x, const void* w, const float* scale,
                              const float* shift, void* y, void* stream) {
    TLXMI_REQUIRE(d && x && w && y, TLXMI_ERR_BAD_ARG, "dwconv2d: null argument");
    REQUIRE_CHUNKED("dwconv2d", d->dtype, d->C, d->x_ld, d->y_ld);
    TLXMI_REQUIRE(aligned16(x) && aligned16(y) && aligned16(w), TLXMI_ERR_ALIGNMENT, "dwconv2d: buffers must be 16-byte aligned");
    TLXMI_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->R > 0 && d->S > 0 && d->stride_h > 0 && d->stride_w > 0 &&
                      d->dil_h > 0 && d->dil_w > 0 && d->pad_h >= 0 && d->pad_w >= 0,
                  TLXMI_ERR_BAD_ARG, "dwconv2d: bad extent");
    const int Ho = (d->H + 2 * d->pad_h - d->dil_h * (d->R - 1) - 1) / d->stride_h + 1;
    const int Wo = (d->W + 2 * d->pad_w - d->dil_w * (d->S - 1) - 1) / d->stride_w + 1;
    // the full-correlation extent, or more: windows past the bottom / right edge read zeros (one-sided end padding, as
    // tlxmi_conv2d), as long as every window starts inside the image or its leading padding
    const int Ho_max = (d->H - 1 + d->pad_h) / d->stride_h + 1, Wo_max = (d->W - 1 + d->pad_w) / d->stride_w + 1;
    TLXMI_REQUIRE(Ho > 0 && Wo > 0 && d->Ho >= Ho && d->Wo >= Wo && d->Ho <= Ho_max && d->Wo <= Wo_max, TLXMI_ERR_BAD_ARG,
                  "dwconv2d: output extent %dx%d outside [%dx%d, %dx%d]", d->Ho, d->Wo, Ho, Wo, Ho_max, Wo_max);
    TLXMI_REQUIRE(d->act >= TLXMI_ACT_NONE && d->act <= TLXMI_ACT_SILU, TLXMI_ERR_BAD_ARG, "dwconv2d: bad act");
    // register-tiled strips for the shapes the models use (3x3 / 5x5, stride 1 / 2, no dilation); TLXMI_DWSTRIP=0: off (A/B)
    const int strip_on = (int)tune_int("TLXMI_DWSTRIP", 1);
    if (strip_on && d->dil_h == 1 && d->dil_w == 1 && (d->S == 3 || d->S == 5) && (d->stride_w == 1 || d->stride_w == 2) && d->Wo >= 4) {
        constexpr int TW = 4;
        const int per_row = ((d->Wo + TW - 1) / TW) * (d->C / VECN(d->dtype));
        const long blocks = (long)d->N * d->Ho * ((per_row + 255) / 256);
        TLXMI_REQUIRE(blocks < (1l << 31), TLXMI_ERR_UNSUPPORTED, "dwconv2d: too many rows");
        dim3 g((unsigned)blocks), b(256);
        hipStream_t st = as_stream(stream);
#define TLXMI_DW_CASE(TT, SS, WW)                                                                                                  \
        hipLaunchKernelGGL((dwconv_strip_kernel<TT, TW, SS, WW>), g, b, 0, st, (const TT*)x, (const TT*)w, scale, shift, (TT*)y, *d)
        if (d->dtype == TLXMI_F16) {
            if (d->S == 3 && d->stride_w == 1) TLXMI_DW_CASE(half_t, 3, 1);
            else if (d->S == 3) TLXMI_DW_CASE(half_t, 3, 2);
            else if (d->stride_w == 1) TLXMI_DW_CASE(half_t, 5, 1);
            else TLXMI_DW_CASE(half_t, 5, 2);
        } else {
            if (d->S == 3 && d->stride_w == 1) TLXMI_DW_CASE(float, 3, 1);
            else if (d->S == 3) TLXMI_DW_CASE(float, 3, 2);
            else if (d->stride_w == 1) TLXMI_DW_CASE(float, 5, 1);
            else TLXMI_DW_CASE(float, 5, 2);
        }
#undef TLXMI_DW_CASE
        return check_launch("dwconv2d");
    }
    const long work = (long)d->N * d->Ho * d->Wo * (d->C / VECN(d->dtype));
    dim3 g(grid_for(work)), b(256);
    if (d->dtype == TLXMI_F16)
        hipLaunchKernelGGL((dwconv_kernel<half_t>), g, b, 0, as_stream(stream), (const half_t*)x, (const half_t*)w, scale, shift, (half_t*)y, *d);
    else
        hipLaunchKernelGGL((dwconv_kernel<float>), g, b, 0, as_stream(stream), (const float*)x, (const float*)w, scale, shift, (float*)y, *d);
    return check_launch("dwconv2d");
}

extern "C" int tlxmi_scale_channels(const void* x, const void* sc, void* y, int dt, int N, int HW, int C, int x_ld,
                                    int s_ld, int y_ld, void* stream) {
    TLXMI_REQUIRE(x && sc && y && N > 0 && HW > 0, TLXMI_ERR_BAD_ARG, "scale_channels: bad argument");
    REQUIRE_CHUNKED("scale_channels", dt, C, x_ld, s_ld, y_ld);
    TLXMI_REQUIRE(aligned16(x) && aligned16(sc) && aligned16(y), TLXMI_ERR_ALIGNMENT, "scale_channels: buffers must be 16-byte aligned");
    const long sc_per_img = (long)HW * (C / VECN(dt));
    TLXMI_REQUIRE(sc_per_img < (1l << 31), TLXMI_ERR_UNSUPPORTED, "scale_channels: image too large");
    const long sc_blocks = (long)N * ((sc_per_img + 255) / 256);
    TLXMI_REQUIRE(sc_blocks < (1l << 31), TLXMI_ERR_UNSUPPORTED, "scale_channels: too many blocks");
    dim3 g((unsigned)sc_blocks), b(256);
    if (dt == TLXMI_F16)
        hipLaunchKernelGGL((scale_channels_kernel<half_t>), g, b, 0, as_stream(stream), (const half_t*)x, (const half_t*)sc, (half_t*)y, N, HW, C, x_ld, s_ld, y_ld);
    else
        hipLaunchKernelGGL((scale_channels_kernel<float>), g, b, 0, as_stream(stream), (const float*)x, (const float*)sc, (float*)y, N, HW, C, x_ld, s_ld, y_ld);
    return check_launch("scale_channels");
}

extern "C" int tlxmi_upsample2x_nearest(const void* x, void* y, int dt, int N, int H, int W, int C, int x_ld, int y_ld,
                                        int c_off, void* stream) {
    TLXMI_REQUIRE(x && y && N > 0 && H > 0 && W > 0, TLXMI_ERR_BAD_ARG, "upsample2x: bad argument");
    REQUIRE_CHUNKED("upsample2x", dt, C, x_ld);
    TLXMI_REQUIRE(c_off >= 0 && y_ld >= c_off + C && y_ld % VECN(dt) == 0 && c_off % VECN(dt) == 0, TLXMI_ERR_ALIGNMENT,
                  "upsample2x: bad destination window");
    TLXMI_REQUIRE(aligned16(x) && aligned16(y), TLXMI_ERR_ALIGNMENT, "upsample2x: buffers must be 16-byte aligned");
    const long work = (long)N * 4 * H * W * (C / VECN(dt));
    dim3 g(grid_for(work)), b(256);
    if (dt == TLXMI_F16)
        hipLaunchKernelGGL((upsample2x_kernel<half_t>), g, b, 0, as_stream(stream), (const half_t*)x, (half_t*)y, N, H, W, C, x_ld, y_ld, c_off);
    else
        hipLaunchKernelGGL((upsample2x_kernel<float>), g, b, 0, as_stream(stream), (const float*)x, (float*)y, N, H, W, C, x_ld, y_ld, c_off);
    return check_launch("upsample2x");
}

extern "C" int tlxmi_copy_channels(const void* x, void* y, int dt, int64_t rows, int C, int x_ld, int y_ld,
                                   void* stream) {
    TLXMI_REQUIRE(x && y && rows > 0, TLXMI_ERR_BAD_ARG, "copy_channels: bad argument");
    REQUIRE_CHUNKED("copy_channels", dt, C, y_ld);
    TLXMI_REQUIRE(x_ld == 0 || (x_ld >= C && x_ld % VECN(dt) == 0), TLXMI_ERR_ALIGNMENT,
                  "copy_channels: bad source stride %d (0 = broadcast one row)", x_ld);
    TLXMI_REQUIRE(aligned16(x) && aligned16(y), TLXMI_ERR_ALIGNMENT, "copy_channels: buffers must be 16-byte aligned");
    const long work = rows * (C / VECN(dt));
    dim3 g(grid_for(work)), b(256);
    if (dt == TLXMI_F16)
        hipLaunchKernelGGL((copy_channels_kernel<half_t>), g, b, 0, as_stream(stream), (const half_t*)x, (half_t*)y, (long)rows, C, x_ld, y_ld);
    else
        hipLaunchKernelGGL((copy_channels_kernel<float>), g, b, 0, as_stream(stream), (const float*)x, (float*)y, (long)rows, C, x_ld, y_ld);
    return check_launch("copy_channels");
}

extern "C" int tlxmi_argmax_lastdim(const void* x, int dt, int64_t rows, int C, int x_ld, int64_t* out, void* stream) {
    TLXMI_REQUIRE(x && out && rows > 0 && C > 0 && x_ld >= C, TLXMI_ERR_BAD_ARG, "argmax: bad argument");
    TLXMI_REQUIRE(DT_OK(dt), TLXMI_ERR_BAD_ARG, "argmax: bad dtype");
    dim3 g((unsigned)((rows + 3) / 4)), b(256);
    if (dt == TLXMI_F16)
        hipLaunchKernelGGL((argmax_kernel<half_t>), g, b, 0, as_stream(stream), (const half_t*)x, (long)rows, C, x_ld, out);
    else
        hipLaunchKernelGGL((argmax_kernel<float>), g, b, 0, as_stream(stream), (const float*)x, (long)rows, C, x_ld, out);
    return check_launch("argmax");
}

// NHWC implicit-GEMM Conv2d / Linear for gfx950 (MI355X), fused scale/shift (folded BatchNorm or
// bias) + residual + activation epilogue.
//
// Replaces (reference call sites): nn.GroupConv2d + nn.BatchNorm2d + nn.ReLU + `out += identity`
// tlxcv/models/classification/resnet.py:142-156; patch embedding vision_transformer.py:197-220;
// nn.Linear(+GeLU, +residual) vision_transformer.py:81-87,112-123,172-175; ConvBNLayer
// tlxcv/models/detection/backbones/darknet.py:54-58.
//
// GEMM view (computed transposed so that each lane ends up owning 8 consecutive output channels
// of one pixel, i.e. one 16-byte NHWC store):
//     D[ch][pix] = sum_k Wp[ch][k] * X[pix][k],   k = (r*S + s)*C + c
//   "A" operand = filter rows (K contiguous, packed once by tlxmi_pack_filter)
//   "B" operand = input pixels gathered on the fly (16-byte chunks of C, zero for padding taps)
// Everything is expressed in 16-byte chunks so the same kernel body serves fp16 (8 elements per
// chunk, v_mfma_f32_16x16x32_f16) and fp32 (4 elements per chunk, 4 x v_mfma_f32_16x16x4_f32,
// exact fp32 FMA chain: the parity mode).
//
// Tile: BM pixels x BN channels x 64 bytes of K per step, 256 threads = 4 waves in a 2x2 grid,
// double-buffered LDS, register-staged global loads issued one K-step ahead.
// LDS image: rows of 64 B (4 chunks), chunk index XOR-swizzled with (-(row>>2))&3 so that both the
// ds_write_b128 staging stores and the ds_read_b128 fragment reads are bank-conflict free under
// gfx950's b128 lane grouping (MI355X_MICROARCH "LDS" table).
#include "common.h"

namespace tlxmi {

struct ConvArgs {
    const char* x;
    const char* w;
    char* y;
    const float* scale;
    const float* shift;
    const char* res;
    int N, H, W, C, Cout, R, S, sh, sw, ph, pw, dh, dw, Ho, Wo;
    int x_ld, y_ld, res_ld;
    long y_nstride, res_nstride;  // elements between images (dense = HoWo*ld; res 0 when broadcast)
    int strided_n;                // 1: y or res is not dense over the batch axis
    int act;
    float act_param;
    unsigned flags;
    int M;        // N*Ho*Wo output pixels
    int HoWo;
    int kchunks;  // R*S*cpt true 16-byte chunks along K
    int ktiles;   // ceil(kchunks/4)
    int cpt;      // chunks per filter tap = C*sizeof(T)/16
    int Kp_bytes; // packed filter row pitch in bytes (= ktiles*64)
    int mtiles, ntiles;
    int vec_io;   // 1: y (and res) rows allow 16-byte vector access
};

template <typename T> struct Mma;
template <> struct Mma<half_t> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, a),
                                                      __builtin_bit_cast(half8v, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], c, 0, 0, 0);
        return c;
    }
};

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4);
}

// load / store 8 consecutive elements of T as fp32
template <typename T> __device__ __forceinline__ void load8(const char* p, float* v);
template <> __device__ __forceinline__ void load8<half_t>(const char* p, float* v) {
    half8v h = *reinterpret_cast<const half8v*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
}
template <> __device__ __forceinline__ void load8<float>(const char* p, float* v) {
    f32x4 a = reinterpret_cast<const f32x4*>(p)[0], b = reinterpret_cast<const f32x4*>(p)[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <typename T> __device__ __forceinline__ void store8(char* p, const float* v);
template <> __device__ __forceinline__ void store8<half_t>(char* p, const float* v) {
    half8v h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (half_t)v[i];
    *reinterpret_cast<half8v*>(p) = h;
}
template <> __device__ __forceinline__ void store8<float>(char* p, const float* v) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    reinterpret_cast<f32x4*>(p)[0] = a;
    reinterpret_cast<f32x4*>(p)[1] = b;
}

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int ES = (int)sizeof(T);
    constexpr int VEC = 16 / ES;
    constexpr int WM = BM / 2, WN = BN / 2;  // wave tile (pixels x channels)
    constexpr int PI = WM / 16, CI = WN / 16;
    constexpr int XR = BM / 64, WR = BN / 64;
    constexpr int BUF = (BM + BN) * 64;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;

    // ---- block -> tile, XCD-aware: blocks that share an XCD (id % 8) walk consecutive tiles,
    // N-tiles fastest, so the activation rows of one M-tile are re-read from that XCD's L2.
    int tile_m, tile_n;
    {
        const int nb = a.mtiles * a.ntiles, id = blockIdx.x;
        const int xcd = id & 7, qd = nb >> 3, rm = nb & 7;
        const int L = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (id >> 3);
        tile_n = L % a.ntiles;
        tile_m = L / a.ntiles;
    }
    const int bm0 = tile_m * BM, bn0 = tile_n * BN;

    // ---- loader state: this thread stages chunk column (t&3) of rows (t>>2)+64i
    const int lrow = t >> 2, lchunk = t & 3;
    const char* xbase[XR];
    int hi0[XR], wi0[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int m = bm0 + lrow + 64 * i;
        const int mm = m < a.M ? m : 0;
        const int n = mm / a.HoWo, rem = mm - n * a.HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        hi0[i] = m < a.M ? ho * a.sh - a.ph : -(1 << 28);
        wi0[i] = wo * a.sw - a.pw;
        xbase[i] = a.x + (size_t)n * a.H * a.W * a.x_ld * ES;
    }
    const char* wrow[WR];
#pragma unroll
    for (int j = 0; j < WR; ++j) wrow[j] = a.w + (size_t)(bn0 + lrow + 64 * j) * a.Kp_bytes;

    // LDS staging offsets.  Filter rows are permuted inside each group of 32 channels:
    // channel n = 32c + 8g + 4e + b is written to row 32c + 16e + 4g + b, so that MFMA sub-tile
    // (2c+e), accumulator register b of lane group g is channel 32c + 8g + 4e + b: one lane then
    // owns channels 32c+8g .. +7 (8 consecutive) across the sub-tile pair.
    int xst[XR], wst[WR];
#pragma unroll
    for (int i = 0; i < XR; ++i) xst[i] = lds_off(lrow + 64 * i, lchunk);
#pragma unroll
    for (int j = 0; j < WR; ++j) {
        const int n = lrow + 64 * j;
        const int p = (n & ~31) | (((n >> 2) & 1) << 4) | (((n >> 3) & 3) << 2) | (n & 3);
        wst[j] = BM * 64 + lds_off(p, lchunk);
    }

    // K position of this thread's chunk: q-th chunk -> (tap r,s ; chunk cc inside the tap)
    int q = lchunk;
    int r, s, cc;
    {
        const int tap = q / a.cpt;
        cc = q - tap * a.cpt;
        r = tap / a.S;
        s = tap - r * a.S;
    }

    u32x4 xa[XR], wa[WR];
    auto gload = [&]() {
        const bool kv = q < a.kchunks;
        const int hoff = r * a.dh, woff = s * a.dw;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int hi = hi0[i] + hoff, wi = wi0[i] + woff;
            const bool ok = kv && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            const char* p = xbase[i] + ((size_t)(hi * a.W + wi) * a.x_ld + cc * VEC) * ES;
            u32x4 z = {0u, 0u, 0u, 0u};
            xa[i] = ok ? *reinterpret_cast<const u32x4*>(p) : z;
        }
#pragma unroll
        for (int j = 0; j < WR; ++j) wa[j] = *reinterpret_cast<const u32x4*>(wrow[j] + (size_t)q * 16);
    };
    auto advance = [&]() {
        q += 4;
        cc += 4;
        while (cc >= a.cpt) {
            cc -= a.cpt;
            if (++s == a.S) { s = 0; ++r; }
        }
    };
    auto lstore = [&](int buf) {
        char* b = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < XR; ++i) *reinterpret_cast<u32x4*>(b + xst[i]) = xa[i];
#pragma unroll
        for (int j = 0; j < WR; ++j) *reinterpret_cast<u32x4*>(b + wst[j]) = wa[j];
    };

    // ---- fragment read offsets (per lane constants)
    const int wave_m0 = (wid & 1) * WM, wave_n0 = (wid >> 1) * WN;
    const int frag = lds_off(lane & 15, lane >> 4);  // row (lane&15) of a 16-row sub-tile
    const int xfrag = wave_m0 * 64 + frag;
    const int wfrag = BM * 64 + wave_n0 * 64 + frag;

    f32x4 acc[CI][PI];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int pi = 0; pi < PI; ++pi) acc[ci][pi] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload();
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < a.ktiles; ++kt) {
        const bool more = kt + 1 < a.ktiles;
        if (more) {
            advance();
            gload();
        }
        const char* b = smem + (kt & 1) * BUF;
        u32x4 wf[CI], xf[PI];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) wf[ci] = *reinterpret_cast<const u32x4*>(b + wfrag + ci * 1024);
#pragma unroll
        for (int pi = 0; pi < PI; ++pi) xf[pi] = *reinterpret_cast<const u32x4*>(b + xfrag + pi * 1024);
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int pi = 0; pi < PI; ++pi) acc[ci][pi] = Mma<T>::run(wf[ci], xf[pi], acc[ci][pi]);
        if (more) lstore((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: y = act(acc*scale + shift (+res)) (+res), 8 consecutive channels per lane
    const int g = lane >> 4, px = lane & 15;
    constexpr int CP = CI / 2;
    float sc[CP][8], sf[CP][8];
#pragma unroll
    for (int cp = 0; cp < CP; ++cp) {
        const int ch0 = bn0 + wave_n0 + 32 * cp + 8 * g;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = ch0 + e;
            const bool in = ch < a.Cout;
            sc[cp][e] = (a.scale && in) ? a.scale[ch] : 1.f;
            sf[cp][e] = (a.shift && in) ? a.shift[ch] : 0.f;
        }
    }
    const bool res_after = (a.flags & TLXMI_EPI_RES_AFTER_ACT) != 0;
#pragma unroll
    for (int pi = 0; pi < PI; ++pi) {
        const int m = bm0 + wave_m0 + pi * 16 + px;
        if (m >= a.M) continue;
        size_t yrow = (size_t)m * a.y_ld, rrow = (size_t)m * a.res_ld;
        if (a.strided_n) {
            const int n = m / a.HoWo, p = m - n * a.HoWo;
            yrow = (size_t)n * a.y_nstride + (size_t)p * a.y_ld;
            rrow = (size_t)n * a.res_nstride + (size_t)p * a.res_ld;
        }
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
            const int ch0 = bn0 + wave_n0 + 32 * cp + 8 * g;
            if (ch0 >= a.Cout) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb)
                    v[e * 4 + bb] = acc[2 * cp + e][pi][bb] * sc[cp][e * 4 + bb] + sf[cp][e * 4 + bb];
            const bool full = a.vec_io && (ch0 + 8 <= a.Cout);
            float rv[8];
            if (a.res) {
                const char* rp = a.res + (rrow + ch0) * ES;
                if (full) {
                    load8<T>(rp, rv);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        rv[e] = (ch0 + e < a.Cout) ? (float)reinterpret_cast<const T*>(rp)[e] : 0.f;
                }
                if (!res_after) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rv[e];
                }
            }
            if (a.act != TLXMI_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], a.act, a.act_param);
            }
            if (a.res && res_after) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            char* yp = a.y + (yrow + ch0) * ES;
            if (full) {
                store8<T>(yp, v);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (ch0 + e < a.Cout) reinterpret_cast<T*>(yp)[e] = (T)v[e];
            }
        }
    }
}

template <typename T, int BM, int BN> static void launch(const ConvArgs& a, hipStream_t st) {
    ConvArgs b = a;
    b.mtiles = (a.M + BM - 1) / BM;
    b.ntiles = (a.Cout + BN - 1) / BN;
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN>), dim3(b.mtiles * b.ntiles), dim3(256), 0, st, b);
}

template <typename T> static void dispatch(const ConvArgs& a, hipStream_t st) {
    // Tile choice: narrow N tile for Cout <= 64; halve the pixel tile when the 128-pixel grid
    // would leave most of the 256 CUs (x ~3 resident blocks) without work.
    const bool n64 = a.Cout <= 64;
    const long tiles128 = (long)((a.M + 127) / 128) * ((a.Cout + (n64 ? 63 : 127)) / (n64 ? 64 : 128));
    const bool m64 = tiles128 < 512;
    if (n64) {
        if (m64) launch<T, 64, 64>(a, st); else launch<T, 128, 64>(a, st);
    } else {
        if (m64) launch<T, 64, 128>(a, st); else launch<T, 128, 128>(a, st);
    }
}

}  // namespace tlxmi

using namespace tlxmi;

extern "C" int tlxmi_conv2d(const tlxmi_conv2d_desc* d, const void* x, const void* w_packed,
                            const float* scale, const float* shift, const void* res, void* y,
                            void* stream) {
    TLXMI_REQUIRE(d && x && w_packed && y, TLXMI_ERR_BAD_ARG, "conv2d: null descriptor or buffer");
    TLXMI_REQUIRE(d->dtype == TLXMI_F16 || d->dtype == TLXMI_F32, TLXMI_ERR_BAD_ARG, "conv2d: bad dtype %d", d->dtype);
    TLXMI_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->Cout > 0 && d->R > 0 && d->S > 0,
                  TLXMI_ERR_BAD_ARG, "conv2d: non-positive extent");
    TLXMI_REQUIRE(d->stride_h > 0 && d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0 && d->pad_h >= 0 && d->pad_w >= 0,
                  TLXMI_ERR_BAD_ARG, "conv2d: bad stride/dilation/padding");
    const int es = (int)elt_size(d->dtype);
    TLXMI_REQUIRE((d->C * es) % 16 == 0, TLXMI_ERR_ALIGNMENT,
                  "conv2d: C=%d must make 16-byte channel chunks (pad the input channels)", d->C);
    TLXMI_REQUIRE(d->x_ld >= d->C && (d->x_ld * es) % 16 == 0, TLXMI_ERR_ALIGNMENT, "conv2d: x_ld=%d", d->x_ld);
    TLXMI_REQUIRE(aligned16(x) && aligned16(w_packed), TLXMI_ERR_ALIGNMENT, "conv2d: x / w must be 16-byte aligned");
    TLXMI_REQUIRE(d->y_ld >= d->Cout, TLXMI_ERR_BAD_ARG, "conv2d: y_ld=%d < Cout=%d", d->y_ld, d->Cout);
    const int Ho = (d->H + 2 * d->pad_h - d->dil_h * (d->R - 1) - 1) / d->stride_h + 1;
    const int Wo = (d->W + 2 * d->pad_w - d->dil_w * (d->S - 1) - 1) / d->stride_w + 1;
    TLXMI_REQUIRE(Ho == d->Ho && Wo == d->Wo && Ho > 0 && Wo > 0, TLXMI_ERR_BAD_ARG,
                  "conv2d: output extent %dx%d does not match descriptor %dx%d", Ho, Wo, d->Ho, d->Wo);
    TLXMI_REQUIRE(d->act >= TLXMI_ACT_NONE && d->act <= TLXMI_ACT_SILU, TLXMI_ERR_BAD_ARG, "conv2d: bad act %d", d->act);
    TLXMI_REQUIRE(!res || d->res_ld >= d->Cout, TLXMI_ERR_BAD_ARG, "conv2d: res_ld=%d < Cout", d->res_ld);
    const long long M = (long long)d->N * Ho * Wo;
    TLXMI_REQUIRE(M < (1ll << 31) && (long long)d->H * d->W * d->x_ld < (1ll << 31), TLXMI_ERR_UNSUPPORTED,
                  "conv2d: extent exceeds 32-bit pixel indexing");

    ConvArgs a;
    a.x = (const char*)x; a.w = (const char*)w_packed; a.y = (char*)y;
    a.scale = scale; a.shift = shift; a.res = (const char*)res;
    a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C; a.Cout = d->Cout; a.R = d->R; a.S = d->S;
    a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
    a.Ho = Ho; a.Wo = Wo; a.x_ld = d->x_ld; a.y_ld = d->y_ld; a.res_ld = res ? d->res_ld : 0;
    a.act = d->act; a.act_param = d->act_param; a.flags = d->flags;
    a.M = (int)M; a.HoWo = Ho * Wo;
    a.cpt = d->C * es / 16;
    a.kchunks = d->R * d->S * a.cpt;
    a.ktiles = (a.kchunks + 3) / 4;
    a.Kp_bytes = a.ktiles * 64;
    a.mtiles = a.ntiles = 0;
    const int vecn = 16 / es;  // elements per 16 bytes
    const bool bcast = res && (d->flags & TLXMI_EPI_RES_BCAST_N);
    TLXMI_REQUIRE(d->y_nstride >= 0 && d->res_nstride >= 0, TLXMI_ERR_BAD_ARG, "conv2d: negative batch stride");
    a.y_nstride = d->y_nstride ? d->y_nstride : (long)a.HoWo * d->y_ld;
    a.res_nstride = bcast ? 0 : (d->res_nstride ? d->res_nstride : (long)a.HoWo * a.res_ld);
    a.strided_n = (d->y_nstride != 0) || (res && (d->res_nstride != 0 || bcast));
    a.vec_io = aligned16(y) && (d->y_ld % vecn == 0) && (a.y_nstride % vecn == 0) &&
               (!res || (aligned16(res) && d->res_ld % vecn == 0 && a.res_nstride % vecn == 0));
    if (d->dtype == TLXMI_F16) dispatch<half_t>(a, as_stream(stream));
    else dispatch<float>(a, as_stream(stream));
    return check_launch("conv2d");
}

// ------------------------------------------------------------------------------------------
// Filter packing: OIHW fp32 -> [Cout_pad][Kpad] (K = (r*S+s)*Cin_pad + c), zero padded.
// ------------------------------------------------------------------------------------------
namespace tlxmi {
static inline int cin_pad(int Cin, int dtype) { const int v = 16 / (int)elt_size(dtype); return (Cin + v - 1) / v * v; }
static inline int kpad_elems(int Cin, int R, int S, int dtype) {
    const int es = (int)elt_size(dtype);
    const int kbytes = R * S * cin_pad(Cin, dtype) * es;
    return ((kbytes + 63) / 64 * 64) / es;
}
template <typename T>
__global__ void pack_filter_kernel(const float* __restrict__ src, T* __restrict__ dst, int Cout, int Cin, int R,
                                   int S, int Cinp, int Kp, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        const int o = (int)(i / Kp);
        const int c = k % Cinp, tap = k / Cinp;
        float v = 0.f;
        if (o < Cout && c < Cin && tap < R * S) {
            const int r = tap / S, s = tap - r * S;
            v = src[(((long)o * Cin + c) * R + r) * S + s];
        }
        dst[i] = (T)v;
    }
}
}  // namespace tlxmi

extern "C" size_t tlxmi_packed_filter_bytes(int Cout, int Cin, int R, int S, int dtype) {
    if (Cout <= 0 || Cin <= 0 || R <= 0 || S <= 0 || (dtype != TLXMI_F16 && dtype != TLXMI_F32)) return 0;
    const size_t cop = (size_t)(Cout + 127) / 128 * 128;
    return cop * (size_t)kpad_elems(Cin, R, S, dtype) * elt_size(dtype);
}

extern "C" int tlxmi_pack_filter(const float* src, void* dst, int Cout, int Cin, int R, int S, int dtype,
                                 void* stream) {
    TLXMI_REQUIRE(src && dst, TLXMI_ERR_BAD_ARG, "pack_filter: null buffer");
    TLXMI_REQUIRE(dtype == TLXMI_F16 || dtype == TLXMI_F32, TLXMI_ERR_BAD_ARG, "pack_filter: bad dtype");
    TLXMI_REQUIRE(Cout > 0 && Cin > 0 && R > 0 && S > 0, TLXMI_ERR_BAD_ARG, "pack_filter: non-positive extent");
    TLXMI_REQUIRE(aligned16(dst), TLXMI_ERR_ALIGNMENT, "pack_filter: dst must be 16-byte aligned");
    const int Cinp = cin_pad(Cin, dtype), Kp = kpad_elems(Cin, R, S, dtype);
    const long cop = (long)(Cout + 127) / 128 * 128;
    const long total = cop * Kp;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == TLXMI_F16)
        hipLaunchKernelGGL((pack_filter_kernel<half_t>), dim3(grid), dim3(256), 0, as_stream(stream), src,
                           (half_t*)dst, Cout, Cin, R, S, Cinp, Kp, total);
    else
        hipLaunchKernelGGL((pack_filter_kernel<float>), dim3(grid), dim3(256), 0, as_stream(stream), src,
                           (float*)dst, Cout, Cin, R, S, Cinp, Kp, total);
    return check_launch("pack_filter");
}

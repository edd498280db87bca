// Grouped 3x3 convolution with few channels per group (ResNeXt: 32 / 64 groups of 4 ... 32 channels, resnext.py:83-91) WITHOUT
// the zero blocks (fp16).
//
// tlxmi_group_conv2d's general path merges groups into 64-channel launch chunks with a block-diagonal filter and runs the dense
// implicit GEMM on them: at 4 channels per group 15 of every 16 MFMA products are zeros (x8 at 8, x4 at 16, x2 at 32 even with the
// half-K skip), and the stage-1 / stage-2 layers of ResNeXt-50 were MFMA-issue-bound at 2.6x / 2.8x their HBM time.
//
// gfx950 still has the small-block MFMA v_mfma_f32_4x4x4_16b_f16: SIXTEEN independent 4x4x4 products per instruction
// (lane = 4 * block + index; A row i / B column j = lane & 3, four k values per lane; D: lane (block, j) holds rows 0..3).
// A group of cg = 4q channels is q x q such blocks, so with
//     block  = (group gg of the 64-channel chunk, output quarter io)          16 blocks = (16 / q) groups x q
//     A      = W[group][4 io + i][tap][4 ki .. 4 ki + 3]                        (q x 9 operands per lane, in registers for good)
//     B      = x[pixel j of a 4-pixel run][tap][group channels 4 ki .. + 3]   (8 q contiguous bytes of the pixel's NHWC row)
//     D     += A . B over the 9 taps and q input quarters
// every product is a real one, the B operand is a plain 8q-byte read of the input pixel (no shuffle, no im2col), and lane
// (block, j) ends up with output channels 64 c + 4 block .. + 3 of its pixel: one 8-byte store, 16 lanes = the pixel's 128-byte
// line.  Rate: 256 FLOP / clk / SIMD (a quarter of the dense 16x16x32) on 1/16 ... 1/2 of the work: the layers become
// HBM-bound (7.4 GFLOP per ResNeXt-50 layer at batch 256 = 14 us of MFMA time).
//
// A workgroup (4 waves) owns TH output rows x the full width x one 64-channel chunk: the (TH - 1) * stride + 3 input rows of
// the chunk (128 bytes a pixel, one zero column left and right, zero rows outside the image — all from the buffer
// descriptor's range check) are brought into LDS by LDS-DMA in one go, then each wave walks runs of four output pixels, four
// runs in flight (independent accumulators between dependent MFMAs).  Epilogue: folded BatchNorm / bias, activation, fp16.
// Algorithmic bytes per output pixel and chunk: 128 * (stride^2 + 1); the halo rows are re-read from L2 / the Infinity Cache
// ((TH * stride + 2) / (TH * stride) of the input).
#include "common.h"
#include "group_conv.h"

namespace tlxmi {

typedef __attribute__((address_space(3))) void* gc_lds_ptr_t;
static __device__ __forceinline__ void gc_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (gc_lds_ptr_t)lds, 16, voff, 0, 0, 0);
}

struct GconvArgs {
    const char* x; const char* w; char* y;
    const float* scale; const float* shift;
    int N, H, W, Ho, Wo;
    int x_ld, y_ld;                 // elements
    int TH, tiles_y;
    int ntiles;                     // N * tiles_y: tiles of one chunk
    int buf_bytes;                  // one LDS tile buffer (two of them)
    unsigned magic_wo;              // ceil(2^20 / Wo): (t * magic) >> 20 is the exact quotient for t < 2^20 / Wo (a tile has < 400 pixels)
    int act; float act_param;
    unsigned x_bytes;
    int Kp_bytes;                   // bytes of one packed filter row (9 taps x 64 chunk channels, padded)
    unsigned chunk_wbytes;          // bytes of one chunk's packed filter
    int debug;                      // tuning flavour: ablation bits (1 no refill, 2 no stores, 4 no MFMA / LDS reads)
};

// Q = channels per group / 4; ST = stride.  Persistent: workgroup blockIdx.x of gridDim.x walks the contiguous tile range
// [ntiles * b / G, ntiles * (b + 1) / G) of chunk blockIdx.y — consecutive row tiles of an image find their shared halo rows
// in this XCD's L2, the filter operands are loaded once — with the NEXT tile's rows in flight (second LDS buffer) while the
// current one is computed.
// NW waves per workgroup (they share the LDS tiles: more waves per CU at the same LDS), U pixel runs in flight per wave.
template <int Q, int ST, int NW, int U>
__global__ __launch_bounds__(64 * NW) void gconv_kernel(const GconvArgs a) {
    constexpr int NTH = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int chunk = blockIdx.y;
    const int IWp = a.W + 2;
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.x), 0, a.x_bytes, 0x00020000);

    // the input rows of a tile -> LDS buffer: pixel ip = iy * IWp + ix at ip * 128, 16-byte slot c = lane & 7.  Row by row
    // (the row index and its validity are scalar): per LDS-DMA a lane spends one compare and one add; lanes past the end of
    // a row are masked off (an LDS-DMA writes the slots of its active lanes only), padding columns / rows fetch at an
    // out-of-range offset (zeros).
    const int rowchunks = IWp * 8;
    auto fill = [&](int tile, int buf) {
        const int n = tile / a.tiles_y, ty = tile - n * a.tiles_y;
        const int oy0 = ty * a.TH;
        const int rows = a.Ho - oy0 < a.TH ? a.Ho - oy0 : a.TH;
        const int ih = (rows - 1) * ST + 3;
        const int gy0 = oy0 * ST - 1;
        char* const dst = smem + buf * a.buf_bytes + wv * 1024;
        const int colbytes = ((t >> 3) - 1) * a.x_ld * 2 + chunk * 128 + (t & 7) * 16;      // of chunk t of a row
        for (int iy = 0; iy < ih; ++iy) {
            const int gy = gy0 + iy;
            const bool rowok = (unsigned)gy < (unsigned)a.H;
            const int rowoff = ((n * a.H + gy) * a.W) * a.x_ld * 2 + colbytes;
            for (int base = 0; base < rowchunks; base += NTH) {
                const int idx = base + t;
                if (idx < rowchunks) {
                    const int gx = (idx >> 3) - 1;
                    gc_dma16(xsrd, dst + (iy * rowchunks + base) * 16, rowok && (unsigned)gx < (unsigned)a.W ? rowoff + (base >> 3) * a.x_ld * 2 : (int)0x80000000);
                }
            }
        }
    };

    const int tile0 = (int)(((long)a.ntiles * blockIdx.x) / gridDim.x), tile1 = (int)(((long)a.ntiles * (blockIdx.x + 1)) / gridDim.x);
    if (tile0 < tile1) fill(tile0, 0);

    // ---- this lane's filter operands: row ol = 4 * block + i of the chunk's packed filter, its group's 4q input channels
    const int blk = lane >> 2, li = lane & 3;
    const int gg = blk / Q;
    half4v wr[9][Q];
    {
        const char* wrow = a.w + (size_t)chunk * a.chunk_wbytes + (size_t)(4 * blk + li) * a.Kp_bytes + gg * (8 * Q);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ki = 0; ki < Q; ++ki) wr[tap][ki] = *reinterpret_cast<const half4v*>(wrow + tap * 128 + ki * 8);
    }
    const int ch0 = chunk * 64 + 4 * blk;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + ch0);
    if (a.shift) sh = *reinterpret_cast<const f32x4*>(a.shift + ch0);

    auto body = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        for (int tile = tile0; tile < tile1; ++tile) {
            const int buf = (tile - tile0) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this tile's rows have landed (this wave's share; the barrier: everyone's)
            __syncthreads();                                       // ... and every wave is done with the other buffer
            if (tile + 1 < tile1 && !TLXMI_DBG(a, 1)) fill(tile + 1, buf ^ 1);
            const int n = tile / a.tiles_y, ty = tile - n * a.tiles_y;
            const int oy0 = ty * a.TH;
            const int rows = a.Ho - oy0 < a.TH ? a.Ho - oy0 : a.TH;
            const int npx = rows * a.Wo;
            const int NQ = (npx + 3) >> 2;
            const char* const tb = smem + buf * a.buf_bytes + gg * (8 * Q);
            for (int q0 = wv; q0 < NQ; q0 += NW * U) {       // this wave: runs q0, q0 + NW, q0 + 2 NW, ...
                f32x4 acc[U];
                int lb[U], yo[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const int tp = 4 * (q0 + NW * u) + li;
                    const bool ok = tp < npx;
                    const int tt = ok ? tp : 0;
                    const int oyl = (int)(((unsigned)tt * a.magic_wo) >> 20), ox = tt - oyl * a.Wo;
                    lb[u] = ((oyl * ST) * IWp + ox * ST) * 128;
                    yo[u] = ok ? ((n * a.Ho + oy0 + oyl) * a.Wo + ox) : -1;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        if (TLXMI_DBG(a, 4)) continue;
                        half4v xb[U][Q];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const char* p = tb + lb[u] + (r * IWp + s) * 128;
                            if constexpr (Q == 1) {
                                xb[u][0] = *reinterpret_cast<const half4v*>(p);
                            } else {
#pragma unroll
                                for (int m = 0; m < Q / 2; ++m) {
                                    const half8v v = *reinterpret_cast<const half8v*>(p + 16 * m);
                                    xb[u][2 * m] = half4v{v[0], v[1], v[2], v[3]};
                                    xb[u][2 * m + 1] = half4v{v[4], v[5], v[6], v[7]};
                                }
                            }
                        }
#pragma unroll
                        for (int ki = 0; ki < Q; ++ki)
#pragma unroll
                            for (int u = 0; u < U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_4x4x4f16(wr[3 * r + s][ki], xb[u][ki], acc[u], 0, 0, 0);
                    }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (yo[u] < 0 || TLXMI_DBG(a, 2)) continue;
                    half4v o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (half_t)apply_act_t<ACT>(acc[u][e] * sc[e] + sh[e], a.act_param);
                    *reinterpret_cast<half4v*>(a.y + ((size_t)yo[u] * a.y_ld + ch0) * 2) = o;
                }
            }
        }
    };
    TLXMI_DISPATCH_ACT(a.act, body);
}

static int gconv_rows(int Ho, int W, int st, int* tiles_y, size_t* lds) {
    // the most output rows whose input rows fit a 40-KB LDS buffer (two buffers a workgroup, two workgroups a CU), evened out
    // over the tiles
    const long row_bytes = (long)(W + 2) * 128;
    int th = 0;
    for (int c = Ho; c >= 1; --c)
        if (((long)(c - 1) * st + 3) * row_bytes <= 40 * 1024) { th = c; break; }
    if (th == 0) return 0;
    const int ty = (Ho + th - 1) / th;
    th = (Ho + ty - 1) / ty;
    *tiles_y = (Ho + th - 1) / th;
    const long total = ((long)(th - 1) * st + 3) * (W + 2) * 8;
    *lds = (size_t)((total + 255) / 256) * 4096;
    return th;
}

bool gconv_small_ok(const tlxmi_conv2d_desc* d, int groups, const void* res) {
    if (!d || d->dtype != TLXMI_F16 || groups <= 1 || res) return false;
    if (d->C != d->Cout || d->C % groups || d->C % 64) return false;
    const int cg = d->C / groups;
    if (cg != 4 && cg != 8 && cg != 16 && cg != 32) return false;
    if (d->R != 3 || d->S != 3 || d->pad_h != 1 || d->pad_w != 1 || d->dil_h != 1 || d->dil_w != 1) return false;
    if (d->stride_h != d->stride_w || (d->stride_h != 1 && d->stride_h != 2)) return false;
    if (d->Ho != (d->H - 1) / d->stride_h + 1 || d->Wo != (d->W - 1) / d->stride_w + 1) return false;
    if (d->x_ld % 8 || d->y_ld % 4 || d->x_ld < d->C || d->y_ld < d->Cout || d->y_nstride != 0) return false;
    if ((long long)d->N * d->H * d->W * d->x_ld * 2 >= (1ll << 31) || (long long)d->N * d->Ho * d->Wo >= (1ll << 31)) return false;
    int ty;
    size_t lds;
    if (gconv_rows(d->Ho, d->W, d->stride_h, &ty, &lds) == 0) return false;
    return (long long)d->N * ty < (1ll << 31) && d->C / 64 <= 65535;
}

template <int Q, int ST> static int launch_gconv_t(const GconvArgs& a, int chunks, size_t lds, hipStream_t st) {
    // 8 waves a workgroup where the filter operands leave room for four waves per SIMD (4 / 8 channels per group)
    constexpr int NW = Q <= 2 ? 8 : 4, U = Q == 1 ? 4 : 2;
    const void* fn = reinterpret_cast<const void*>(&gconv_kernel<Q, ST, NW, U>);
    GconvArgs b = a;
    b.buf_bytes = (int)lds;
    const size_t lds2 = 2 * lds;
    if (lds2 > 64 * 1024)
        if (int rc = raise_lds_limit(fn, 160 * 1024, "group_conv2d")) return rc;
    // persistent grid: what is resident at once (LDS: 160 KB / two buffers, at most 4 workgroups a CU), split over the chunks
    long per_cu = (long)(160 * 1024) / (long)lds2;
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    per_cu = tune_int("TLXMI_GCONV_WGS", per_cu);
    long gx = ((long)device_cus() * per_cu + chunks - 1) / chunks;
    if (gx > b.ntiles) gx = b.ntiles;
    if (gx < 1) gx = 1;
    void* args[] = {&b};
    hipError_t e = hipLaunchKernel(fn, dim3((unsigned)gx, (unsigned)chunks), dim3(64 * NW), args, lds2, st);
    if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "group_conv2d: HIP launch failed: %s", hipGetErrorString(e));
    return TLXMI_OK;
}

// the caller has checked gconv_small_ok; w_packed is tlxmi_pack_group_filter's buffer (64-channel chunks: [chunk][128 rows][Kp])
int launch_gconv_small(const tlxmi_conv2d_desc* d, int groups, const void* x, const void* w_packed, const float* scale,
                       const float* shift, void* y, int Kp_bytes, hipStream_t st) {
    GconvArgs a;
    a.x = (const char*)x; a.w = (const char*)w_packed; a.y = (char*)y; a.scale = scale; a.shift = shift;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.x_ld = d->x_ld; a.y_ld = d->y_ld;
    a.act = d->act; a.act_param = d->act_param;
    a.x_bytes = (unsigned)((long long)d->N * d->H * d->W * d->x_ld * 2);
    a.Kp_bytes = Kp_bytes;
    a.chunk_wbytes = (unsigned)(128u * (unsigned)Kp_bytes);
    size_t lds;
    a.TH = gconv_rows(d->Ho, d->W, d->stride_h, &a.tiles_y, &lds);
    a.ntiles = d->N * a.tiles_y;
    a.debug = (int)tune_int("TLXMI_GCONV_DBG", 0);
    a.magic_wo = ((1u << 20) + (unsigned)d->Wo - 1) / (unsigned)d->Wo;
    const int q = d->C / groups / 4, chunks = d->C / 64;
    const bool s2 = d->stride_h == 2;
    int rc;
    switch (q) {
        case 1: rc = s2 ? launch_gconv_t<1, 2>(a, chunks, lds, st) : launch_gconv_t<1, 1>(a, chunks, lds, st); break;
        case 2: rc = s2 ? launch_gconv_t<2, 2>(a, chunks, lds, st) : launch_gconv_t<2, 1>(a, chunks, lds, st); break;
        case 4: rc = s2 ? launch_gconv_t<4, 2>(a, chunks, lds, st) : launch_gconv_t<4, 1>(a, chunks, lds, st); break;
        default: rc = s2 ? launch_gconv_t<8, 2>(a, chunks, lds, st) : launch_gconv_t<8, 1>(a, chunks, lds, st); break;
    }
    if (rc != TLXMI_OK) return rc;
    return check_launch("group_conv2d(4x4x4)");
}

}  // namespace tlxmi

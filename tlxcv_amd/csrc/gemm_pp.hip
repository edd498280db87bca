// 256 x 256 tile GEMM, two wave groups in antiphase ("ping-pong") — the MFMA-bound Linear layers of
// ViT / Swin (qkv, proj, fc1, fc2; reference vision_transformer.py:81-87,112-123):
//     Y[m][n] = act( (sum_k X[m][k] * Wp[n][k]) * scale[n] + shift[n] (+ R[m][n]) )
// Same operands, filter packing and epilogue semantics as conv_igemm.hip's 1x1 path.
//
// 512 threads = 8 waves = 2 groups (wr = wid>>2) x 4 (wc = wid&3); one wave of each group per SIMD.
// K advances in tiles of 128 B (64 halves / 32 floats).  A K tile in LDS is four 16-KiB "half tiles"
// (X rows 0-127, X rows 128-255, W rows 0-127, W rows 128-255; 128-B rows, chunk c of row r in slot
// c ^ ((r>>1)&7)); two K tiles are resident (128 KiB).  A wave owns the 2 x 2 quadrants
//     X half h rows 64*wr .. +63   x   W half g rows 32*wc .. +31          (h, g in {0,1})
// and one K tile is four phases, one quadrant (16 MFMAs) each:
//     p0: read X0 + W0 frags (12 ds_read_b128), quadrant (0,0)     p1: read W1 (4), quadrant (0,1)
//     p2: read X1 (8), quadrant (1,1)                              p3: no reads, quadrant (1,0)
// Every phase is  [LDS reads | 2 LDS-DMA pieces | counted vmcnt] barrier [16 MFMAs] barrier, and group 1
// runs one barrier behind group 0: while one wave of a SIMD issues its MFMAs the other one issues its reads
// and DMA, so the MFMA pipe never waits for a load segment (a lock-step block alternates between the two).
// DMA order per thread (one half tile = 2 pieces per phase):
//     p0(kt): W1(kt+1)   p1(kt): X1(kt+1)   p2(kt): X0(kt+2)   p3(kt): W0(kt+2)
// i.e. a half tile is re-filled >= 2 phases after its last read (the other group's reads are retired by
// then) and lands >= 4 phases before its first read; vmcnt(8) after the issue in p0, p1 and p3 retires
// exactly the half tiles the NEXT phase reads (the wait sits before a barrier every wave passes before
// that read).  K tiles past the end are fetched at an out-of-range descriptor offset (zero fill, no
// memory traffic), which keeps the counts uniform to the last phase.
#include "common.h"
#include "gemm256.h"
#include <stdlib.h>

namespace tlxmi {

typedef __attribute__((address_space(3))) void* lds_ptr_pp_t;
static __device__ __forceinline__ void pp_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_pp_t)lds, 16, voff, 0, 0, 0);
}
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t pp_srd(const char* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, bytes, 0x00020000);
}
static __device__ __forceinline__ u32x4 pp_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
}
static __device__ __forceinline__ void pp_store16_nt(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 2);
}
static __device__ __forceinline__ void pp_store16_wb(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 0);
}

template <typename T> struct MmaPP;
template <> struct MmaPP<half_t> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, a), __builtin_bit_cast(half8v, b), c, 0, 0, 0);
    }
};
template <> struct MmaPP<float> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], c, 0, 0, 0);
        return c;
    }
};

// HM = X half tiles per K tile: 2 -> the 256 x 256 tile described above; 1 -> a 128 x 256 tile (tails: a launch
// whose last round would hold fewer than half the CUs runs those rows as twice as many half-height tiles).
// With one X half a K tile is two phases ((0,0) and (0,1)) and 48 KiB, so THREE K tiles are resident and the
// DMA runs two K tiles ahead:  p0(k): X0, W0 of k+2 (vmcnt(10) retires W1(k));  p1(k): W1 of k+2 (vmcnt(8)
// retires X0, W0 of k+1) — again every half tile is refilled two phases after its last read.
// CONV: the X operand is the im2col view of an NHWC map under an R x 3 filter, stride / zero padding as given
// (3x3 convs with >= 256 output channels: resnet.py:111-121 conv2 of the 14x14 / 7x7 stages, vgg.py:74-80,
// darknet.py:54-58).  A K tile of 128 bytes lies inside one filter tap (C * sizeof(T) = 128 << ctshift), so the tap
// of a K tile is wave-uniform: its byte offset is scalar arithmetic, and each of a lane's four rows carries a bit
// mask of the taps that fall inside the image (bit clear -> out-of-range descriptor offset -> zero fill).
// HN = W half tiles per K tile: (HM, HN) = (2, 1) is the mirror image of (1, 2) — a 256 x 128 tile for layers with 128
// output channels (3x3 convs of ResNet's 28 x 28 stage): phases (0,0) and (1,0), X1 taking W1's place in the DMA order.
// LNF (fp16, plain GEMM rows): the LayerNorm fold of gemm_stream.hip in this kernel's end-of-tile epilogue — a.rowstats: the producer's
// planes of per-row (sum, sum^2), turned into (a, b) per row and applied as y = act(a * acc + b * scale[n] + shift[n]) (the consumer of a
// folded LayerNorm); a.stats_out: per-row (sum, sum^2) of this tile's 256 output channels as plane bn0 / 256 (the producer; the four
// wc waves of a row are added through LDS).  Used where the persistent kernel does not apply: a residual with fewer than 11 K tiles
// (Swin-B stage 3 proj: K = 512), fewer tiles than half the CUs.
template <typename T, int HM, int HN, bool CONV, bool LNF = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const Gemm256Args a) {
    static_assert(HM + HN >= 3 && HM <= 2 && HN <= 2, "tile is 256x256, 128x256 or 256x128");
    constexpr int ES = (int)sizeof(T);
    constexpr int BM = 128 * HM, BN = 128 * HN;
    constexpr int HALF = 128 * 128;            // bytes of a half tile
    constexpr int RX0 = 0, RX1 = HALF, RW0 = HM * HALF, RW1 = (HM + 1) * HALF;   // regions of a K tile (RX1: HM == 2, RW1: HN == 2)
    constexpr int KTB = (HM + HN) * HALF;      // bytes of a K tile
    constexpr int NBUF = (HM + HN == 4) ? 2 : 3;   // resident K tiles
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wid >> 2, wc = wid & 3;

    // block -> tile: blocks sharing an XCD (id % 8) take consecutive tiles, N tiles fastest
    int tile_m, tile_n;
    {
        const int nb = a.mtiles * a.ntiles, id = a.kslices > 1 ? (int)blockIdx.x % nb : (int)blockIdx.x;
        const int xcd = id & 7, qd = nb >> 3, rm = nb & 7;
        const int L = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (id >> 3);
        // column panels of a.gn N tiles (gemm_stream.hip's walk): the filter panel stays in the XCD's L2 while its row tiles pass
        const int per = a.mtiles * a.gn, p = L / per, r = L - p * per;
        const int gh = a.ntiles - p * a.gn < a.gn ? a.ntiles - p * a.gn : a.gn;
        tile_m = r / gh;
        tile_n = p * a.gn + r - tile_m * gh;
    }
    const int bm0 = tile_m * BM, bn0 = tile_n * BN;
    // split K: this copy of the tile grid multiplies K tiles kt0 .. kt0 + ks - 1 (a K tile past the slice is a zero fill)
    const int slice = a.kslices > 1 ? (int)blockIdx.x / (a.mtiles * a.ntiles) : 0;
    const int kt0 = slice * a.kt_slice;
    const int ks = a.kslices > 1 ? (a.ksteps - kt0 < a.kt_slice ? a.ksteps - kt0 : a.kt_slice) : a.ksteps;
    const __amdgpu_buffer_rsrc_t xsrd = pp_srd(a.x, a.x_bytes), wsrd = pp_srd(a.w, a.w_bytes);

    // ---- loader: a piece = 8 rows x 128 B (one wave instruction); wave w fills pieces w and w+8 of a half
    // tile; lane l -> row 8*piece + (l>>3), slot (l&7).  (row>>1)&7 = (4*(w&1) + (l>>4)) & 7 for both pieces.
    const int lrow = lane >> 3;
    const int lc = (lane & 7) ^ ((4 * (wid & 1) + (lane >> 4)) & 7);   // logical K chunk behind this lane's slot
    int xo[2][2], wo[2][2];
    unsigned tapmask[2][2];   // CONV: taps of this row that lie inside the image
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = bm0 + 128 * h + 8 * (wid + 8 * j) + lrow;
            if constexpr (CONV) {
                const bool live = h < HM && m < a.M;
                const int mm = live ? m : 0;
                const int n = mm / a.cHoWo, rem = mm - n * a.cHoWo;
                const int ho = rem / a.cWo, wo_ = rem - ho * a.cWo;
                const int hi0 = ho * a.csh - a.cph, wi0 = wo_ * a.csw - a.cpw;
                xo[h][j] = ((n * a.cH + hi0) * a.cW + wi0) * a.x_ld * ES;     // tap (0,0); may wrap below zero under padding
                unsigned mk = 0;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int r = tp / 3, s_ = tp - 3 * r;
                    if (live && tp < a.ctaps && (unsigned)(hi0 + r) < (unsigned)a.cH && (unsigned)(wi0 + s_) < (unsigned)a.cW) mk |= 1u << tp;
                }
                tapmask[h][j] = mk;
            } else {
                xo[h][j] = (h < HM && m < a.M) ? m * a.x_ld * ES : OOB;
                tapmask[h][j] = 0;
            }
            const int rho = 128 * h + 8 * (wid + 8 * j) + lrow;    // LDS row; holds channel perm(rho) (conv_igemm.hip)
            const int n = (rho & ~31) | (((rho >> 2) & 3) << 3) | (((rho >> 4) & 1) << 2) | (rho & 3);
            wo[h][j] = (bn0 + n) * a.Kp_bytes;
        }
    char* const lbase = smem + wid * 1024;
    // `buf` = resident K-tile slot the tile goes to (kt & 1 for HM == 2, kt % 3 for HM == 1)
    auto stage_x = [&](int h, int kt, int buf) {
        char* b = lbase + buf * KTB + (h ? RX1 : RX0);
        const bool mine = kt < ks;                                 // (split K: the slice ends before the filter does)
        kt += kt0;
        if constexpr (CONV) {
            const int tap = kt >> a.ctshift;                       // wave-uniform; >= ctaps past the end (mask bit clear)
            const int r = (tap * 11) >> 5, s_ = tap - 3 * r;       // tap / 3 for tap <= 8
            const int d = (r * a.cW + s_) * a.x_ld * ES + (((kt - (tap << a.ctshift)) * 8 + lc) << 4);
#pragma unroll
            for (int j = 0; j < 2; ++j) pp_dma16(xsrd, b + j * 8192, (mine && ((tapmask[h][j] >> tap) & 1u)) ? xo[h][j] + d : OOB);
        } else {
            const int q = kt * 8 + lc;
            const bool in = mine && q < a.kchunks;
#pragma unroll
            for (int j = 0; j < 2; ++j) pp_dma16(xsrd, b + j * 8192, in ? xo[h][j] + q * 16 : OOB);
        }
    };
    auto stage_w = [&](int g, int kt, int buf) {
        const int q = (kt + kt0) * 8 + lc;
        char* b = lbase + buf * KTB + (g ? RW1 : RW0);
        const bool in = kt < ks && q * 16 < a.Kp_bytes;
#pragma unroll
        for (int j = 0; j < 2; ++j) pp_dma16(wsrd, b + j * 8192, in ? wo[g][j] + q * 16 : OOB);
    };

    // ---- fragment reads: lane (frow, fg) reads row frow of a 16-row sub-tile, 16-byte chunk 4*ksub + fg
    const int frow = lane & 15, fg = lane >> 4;
    const int foff = frow * 128 + ((fg ^ ((frow >> 1) & 7)) << 4);
    const int xf0 = wr * 64 * 128 + foff, wf0 = wc * 32 * 128 + foff;   // ksub 1 = same offset ^ 64

    f32x4 acc[2 * HN][4 * HM];   // [2*g + ci][4*h + pi]
#pragma unroll
    for (int i = 0; i < 2 * HN; ++i)
#pragma unroll
        for (int j = 0; j < 4 * HM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 xf[4][2], w0f[2][2], w1f[2][2];
    auto prow = [&](int j) { return j; };
    auto read_x = [&](const char* kb, int region) {
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            xf[pi][0] = *reinterpret_cast<const u32x4*>(kb + region + prow(pi) * 2048 + xf0);
            xf[pi][1] = *reinterpret_cast<const u32x4*>(kb + region + prow(pi) * 2048 + (xf0 ^ 64));
        }
    };
    auto read_w = [&](const char* kb, int region, u32x4 (&wf)[2][2]) {
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
            wf[ci][0] = *reinterpret_cast<const u32x4*>(kb + region + ci * 2048 + wf0);
            wf[ci][1] = *reinterpret_cast<const u32x4*>(kb + region + ci * 2048 + (wf0 ^ 64));
        }
    };
#define TLXMI_PP_MMA(H, G, WF)                                                                       \
    {                                                                                                \
        __builtin_amdgcn_s_setprio(1);                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                             \
        _Pragma("unroll") for (int pi = 0; pi < 4; ++pi)                                             \
        _Pragma("unroll") for (int ci = 0; ci < 2; ++ci)                                             \
            acc[2 * (G) + ci][4 * (H) + pi] = MmaPP<T>::run(WF[ci][ks], xf[pi][ks], acc[2 * (G) + ci][4 * (H) + pi]); \
        __builtin_amdgcn_s_setprio(0);                                                               \
    }
#define TLXMI_PP_SYNC()                       \
    __builtin_amdgcn_sched_barrier(0);        \
    __builtin_amdgcn_s_barrier();             \
    __builtin_amdgcn_sched_barrier(0);

    // ---- prologue: X0 W0 W1 X1 of K tile 0, X0 W0 of K tile 1; scale / shift table behind the ring
    float sc_t = 1.f, sh_t = 0.f;
    if (t < BN) {
        const int ch = bn0 + t < a.Cout ? bn0 + t : a.Cout - 1;
        if (a.scale) sc_t = a.scale[ch];
        if (a.shift) sh_t = a.shift[ch];
    }
    if constexpr (HM == 2 && HN == 2) {
        stage_x(0, 0, 0);
        stage_w(0, 0, 0);
        stage_w(1, 0, 0);
        stage_x(1, 0, 0);
        stage_x(0, 1, 1);
        stage_w(0, 1, 1);
    } else if constexpr (HN == 1) {
        stage_x(0, 0, 0);
        stage_w(0, 0, 0);
        stage_x(1, 0, 0);
        stage_x(0, 1, 1);
        stage_w(0, 1, 1);
        stage_x(1, 1, 1);
    } else {
        stage_x(0, 0, 0);
        stage_w(0, 0, 0);
        stage_w(1, 0, 0);
        stage_x(0, 1, 1);
        stage_w(0, 1, 1);
        stage_w(1, 1, 1);
    }
    float* sbuf = reinterpret_cast<float*>(smem + NBUF * KTB);
    if (t < BN) {
        sbuf[t] = sc_t;
        sbuf[BN + t] = sh_t;
    }
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // X0(0), W0(0) landed (this wave's pieces)
    TLXMI_PP_SYNC();
    if (wr == 1) { TLXMI_PP_SYNC(); }   // group 1 runs one barrier behind

    if constexpr (HN == 1) {
        int bc = 0, b2 = 2;      // slots of K tiles kt and kt + 2 (mod 3)
        for (int kt = 0; kt < ks; ++kt) {
            const char* kb = smem + bc * KTB;
            // p0
            read_x(kb, RX0);
            read_w(kb, RW0, w0f);
            stage_x(0, kt + 2, b2);
            stage_w(0, kt + 2, b2);
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // X1(kt)
            TLXMI_PP_SYNC();
            TLXMI_PP_MMA(0, 0, w0f);
            TLXMI_PP_SYNC();
            // p1
            read_x(kb, RX1);
            stage_x(1, kt + 2, b2);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // X0(kt+1), W0(kt+1)
            TLXMI_PP_SYNC();
            TLXMI_PP_MMA(HM - 1, 0, w0f);
            TLXMI_PP_SYNC();
            bc = bc == 2 ? 0 : bc + 1;
            b2 = b2 == 2 ? 0 : b2 + 1;
        }
    } else if constexpr (HM == 2) {
        for (int kt = 0; kt < ks; ++kt) {
            const char* kb = smem + (kt & 1) * KTB;
            // p0
            read_x(kb, RX0);
            read_w(kb, RW0, w0f);
            stage_w(1, kt + 1, (kt + 1) & 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // W1(kt)
            TLXMI_PP_SYNC();
            TLXMI_PP_MMA(0, 0, w0f);
            TLXMI_PP_SYNC();
            // p1
            read_w(kb, RW1, w1f);
            stage_x(1, kt + 1, (kt + 1) & 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // X1(kt)
            TLXMI_PP_SYNC();
            TLXMI_PP_MMA(0, 1, w1f);
            TLXMI_PP_SYNC();
            // p2
            read_x(kb, RX1);
            stage_x(0, kt + 2, kt & 1);
            TLXMI_PP_SYNC();
            TLXMI_PP_MMA(HM - 1, 1, w1f);
            TLXMI_PP_SYNC();
            // p3
            stage_w(0, kt + 2, kt & 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // X0(kt+1), W0(kt+1)
            TLXMI_PP_SYNC();
            TLXMI_PP_MMA(HM - 1, 0, w0f);
            TLXMI_PP_SYNC();
        }
    } else {
        int bc = 0, b2 = 2;      // slots of K tiles kt and kt + 2 (mod 3)
        for (int kt = 0; kt < ks; ++kt) {
            const char* kb = smem + bc * KTB;
            // p0
            read_x(kb, RX0);
            read_w(kb, RW0, w0f);
            stage_x(0, kt + 2, b2);
            stage_w(0, kt + 2, b2);
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // W1(kt)
            TLXMI_PP_SYNC();
            TLXMI_PP_MMA(0, 0, w0f);
            TLXMI_PP_SYNC();
            // p1
            read_w(kb, RW1, w1f);
            stage_w(1, kt + 2, b2);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // X0(kt+1), W0(kt+1)
            TLXMI_PP_SYNC();
            TLXMI_PP_MMA(0, 1, w1f);
            TLXMI_PP_SYNC();
            bc = bc == 2 ? 0 : bc + 1;
            b2 = b2 == 2 ? 0 : b2 + 1;
        }
    }
    if (wr == 0) { TLXMI_PP_SYNC(); }   // barrier counts match again
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill DMAs of the tail
#undef TLXMI_PP_MMA
#undef TLXMI_PP_SYNC

    if TLXMI_DBG(a, 1) {   // ablation: no epilogue
#pragma unroll
        for (int i = 0; i < 2 * HN; ++i)
#pragma unroll
            for (int j = 0; j < 4 * HM; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    // ---- epilogue from registers: lane (fg, px) owns channels 128g + 32wc + 8fg .. +7 of pixel row
    // 128h + 64wr + 16pi + px (the filter rows are permuted so that two MFMA sub-tiles give 8 neighbours)
    const int px = lane & 15;
    if (a.kslices > 1) {      // split K: the accumulators as they are, fp32, to this slice's [M][y_ld] plane
        const __amdgpu_buffer_rsrc_t psrd = pp_srd(a.y + (long long)slice * a.slice_bytes, (unsigned)a.slice_bytes);
#pragma unroll
        for (int g = 0; g < HN; ++g) {
            const int ch0 = bn0 + 128 * g + 32 * wc + 8 * fg;
#pragma unroll
            for (int h = 0; h < HM; ++h)
#pragma unroll
                for (int pi = 0; pi < 4; ++pi) {
                    const int m = bm0 + 128 * h + 64 * wr + prow(pi) * 16 + px;
                    const int yo = (m < a.M && ch0 < a.Cout) ? (m * a.y_ld + ch0) * 4 : OOB;
                    pp_store16_wb(psrd, __builtin_bit_cast(u32x4, acc[2 * g][4 * h + pi]), yo);
                    pp_store16_wb(psrd, __builtin_bit_cast(u32x4, acc[2 * g + 1][4 * h + pi]), yo + 16);
                }
        }
        return;
    }
    const bool res_after = (a.flags & TLXMI_EPI_RES_AFTER_ACT) != 0;
    const __amdgpu_buffer_rsrc_t ysrd = pp_srd(a.y, a.y_bytes), rsrd = pp_srd(a.res ? a.res : a.y, a.res ? a.res_bytes : 0u);
    auto epi = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        float hs_[HM], hq_[HM];      // LNF producer: this wave's (sum, sum^2) of row 128 h + 64 wr + lane over its channels of both column halves
#pragma unroll
        for (int h = 0; h < HM; ++h) { hs_[h] = 0.f; hq_[h] = 0.f; }
#pragma unroll
        for (int g = 0; g < HN; ++g) {
            const int col = 128 * g + 32 * wc + 8 * fg;
            const int ch0 = bn0 + col;
            if (ch0 >= a.Cout) continue;      // Cout is a multiple of 8 on this path
            float sc[8], sf[8];
            {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(sbuf + col), s1 = *reinterpret_cast<const f32x4*>(sbuf + col + 4);
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(sbuf + BN + col), h1 = *reinterpret_cast<const f32x4*>(sbuf + BN + col + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[e] = s0[e]; sc[4 + e] = s1[e]; sf[e] = h0[e]; sf[4 + e] = h1[e]; }
            }
#pragma unroll
            for (int h = 0; h < HM; ++h) {
                u32x4 rr[4][ES / 2];
                if (a.res) {
#pragma unroll
                    for (int pi = 0; pi < 4; ++pi) {
                        const int m = bm0 + 128 * h + 64 * wr + prow(pi) * 16 + px;
                        const int ro = m < a.M ? (m * a.res_ld + ch0) * ES : OOB;
#pragma unroll
                        for (int hh = 0; hh < ES / 2; ++hh) rr[pi][hh] = pp_load16(rsrd, ro + 16 * hh);
                    }
                }
                f32x2 rab[4];
                if constexpr (LNF) {
                    if (a.rowstats) {      // (a, b) = (rstd, -mean * rstd) of the row from the producer's planes of (sum, sum^2) over 256 channels each
#pragma unroll
                        for (int pi = 0; pi < 4; ++pi) {
                            const int m = bm0 + 128 * h + 64 * wr + prow(pi) * 16 + px;
                            float sm = 0.f, sq = 0.f;
                            if (m < a.M) {
#pragma unroll
                                for (int p = 0; p < 4; ++p)
                                    if (p < a.ln_planes) {
                                        const f32x2 pl = *reinterpret_cast<const f32x2*>(a.rowstats + 2 * ((size_t)m * 4 + p));
                                        sm += pl[0];
                                        sq += pl[1];
                                    }
                            }
                            const float mean = sm * a.ln_inv_c;
                            const float var = fmaxf(__builtin_fmaf(-mean, mean, sq * a.ln_inv_c), 0.f);
                            const float rstd = 1.f / sqrtf(var + a.ln_eps);
                            rab[pi] = f32x2{rstd, -mean * rstd};
                        }
                    }
                }
                float st_s[4], st_q[4];
#pragma unroll
                for (int pi = 0; pi < 4; ++pi) {
                    const int m = bm0 + 128 * h + 64 * wr + prow(pi) * 16 + px;
                    float v[8], rv[8];
                    if (LNF && a.rowstats) {
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb) {
                            v[bb] = acc[2 * g][4 * h + pi][bb] * rab[pi][0] + (rab[pi][1] * sc[bb] + sf[bb]);
                            v[4 + bb] = acc[2 * g + 1][4 * h + pi][bb] * rab[pi][0] + (rab[pi][1] * sc[4 + bb] + sf[4 + bb]);
                        }
                    } else {
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb) {
                            v[bb] = acc[2 * g][4 * h + pi][bb] * sc[bb] + sf[bb];
                            v[4 + bb] = acc[2 * g + 1][4 * h + pi][bb] * sc[4 + bb] + sf[4 + bb];
                        }
                    }
                    if (a.res) {
                        if constexpr (ES == 2) {
                            const half8v hv = __builtin_bit_cast(half8v, rr[pi][0]);
#pragma unroll
                            for (int e = 0; e < 8; ++e) rv[e] = (float)hv[e];
                        } else {
                            const f32x4 r0 = __builtin_bit_cast(f32x4, rr[pi][0]), r1 = __builtin_bit_cast(f32x4, rr[pi][ES / 2 - 1]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { rv[e] = r0[e]; rv[4 + e] = r1[e]; }
                        }
                        if (!res_after) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += rv[e];
                        }
                    }
                    if constexpr (ACT == TLXMI_ACT_GELU && sizeof(T) == 2) {
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const f32x2v g2 = gelu_fast2(f32x2v{v[e], v[e + 1]});
                            v[e] = g2[0];
                            v[e + 1] = g2[1];
                        }
                    } else if constexpr (ACT != TLXMI_ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = apply_act_t<ACT>(v[e], a.act_param);
                    }
                    if (a.res && res_after) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rv[e];
                    }
                    if constexpr (LNF) {
                        float ss = v[0], qq = v[0] * v[0];
#pragma unroll
                        for (int e = 1; e < 8; ++e) { ss += v[e]; qq = __builtin_fmaf(v[e], v[e], qq); }
                        st_s[pi] = ss;
                        st_q[pi] = qq;
                    }
                    const int yo = m < a.M ? (m * a.y_ld + ch0) * ES : OOB;   // OOB stores are dropped by the range check
                    if constexpr (ES == 2) {
                        half8v hv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
                        if (CONV || !TLXMI_NT_STORES(a)) pp_store16_wb(ysrd, __builtin_bit_cast(u32x4, hv), yo);
                        else pp_store16_nt(ysrd, __builtin_bit_cast(u32x4, hv), yo);
                    } else {
                        f32x4 f0, f1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { f0[e] = v[e]; f1[e] = v[4 + e]; }
                        if (CONV || !TLXMI_NT_STORES(a)) pp_store16_wb(ysrd, __builtin_bit_cast(u32x4, f0), yo);
                        else pp_store16_nt(ysrd, __builtin_bit_cast(u32x4, f0), yo);
                        if (CONV || !TLXMI_NT_STORES(a)) pp_store16_wb(ysrd, __builtin_bit_cast(u32x4, f1), yo + 16);
                        else pp_store16_nt(ysrd, __builtin_bit_cast(u32x4, f1), yo + 16);
                    }
                }
                if constexpr (LNF) {
                    if (a.stats_out) {
                        // the four lanes (fg = 0..3) of a row hold its 32 channels of this wave and column half: three lane-swap steps per four
                        // quantities (gemm_stream.hip STATS); lane row fg ends with the total of sub-tile pi = fg -> row 16 fg + px = lane
                        auto tree = [&](const float (&x)[4]) -> float { return ln_row_tree(x[0], x[1], x[2], x[3]); };
                        hs_[h] += tree(st_s);
                        hq_[h] += tree(st_q);
                    }
                }
            }
        }
        if constexpr (LNF) {
            if (a.stats_out) {
                // the four wc waves of a row half -> LDS (the K tiles are dead: every wave is past its last fragment read and its DMAs),
                // added in a fixed order, stored by wave wc == 0 as pair bn0 / 256 of the row: a.stats_out [M][4][2]
                f32x2* scr = reinterpret_cast<f32x2*>(smem);
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int h = 0; h < HM; ++h) scr[((h * 2 + wr) * 4 + wc) * 64 + lane] = f32x2{hs_[h], hq_[h]};
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (wc == 0) {
#pragma unroll
                    for (int h = 0; h < HM; ++h) {
                        const f32x2* q = scr + (h * 2 + wr) * 4 * 64 + lane;
                        const f32x2 p0 = q[0], p1 = q[64], p2 = q[128], p3 = q[192];
                        const int m = bm0 + 128 * h + 64 * wr + lane;
                        if (m < a.M)
                            *reinterpret_cast<f32x2*>(a.stats_out + 2 * ((size_t)m * 4 + (bn0 >> 8))) =
                                f32x2{(p0[0] + p1[0]) + (p2[0] + p3[0]), (p0[1] + p1[1]) + (p2[1] + p3[1])};
                    }
                }
            }
        }
    };
    TLXMI_DISPATCH_ACT(a.act, epi)
}

// Preconditions as launch_gemm256 (checked by conv_igemm.hip's dispatcher); a.ksteps = packed pitch / 128.
template <typename T, int HM, int HN, bool CONV, bool LNF = false> static int launch_pp_t(const Gemm256Args& a0, hipStream_t st) {
    Gemm256Args a = a0;
    a.debug = (int)tune_int("TLXMI_DEBUG", 0);     // ablation bits: tuning flavour only (TLXMI_DBG is `false` in the product)
    a.mtiles = (a.M + 128 * HM - 1) / (128 * HM);
    a.ntiles = (a.Cout + 128 * HN - 1) / (128 * HN);
    a.gn = a.ntiles;
    if (const long g = tune_int("TLXMI_GS_PANEL", 3); !CONV && g > 0 && g < a.ntiles) a.gn = (int)g;
    const size_t lds = (size_t)(HM + HN == 4 ? 8 : 9) * 128 * 128 + 2 * 256 * sizeof(float);
    const void* fn = reinterpret_cast<const void*>(&gemm_pp_kernel<T, HM, HN, CONV, LNF>);
    if (int rc = raise_lds_limit(fn, (int)lds, "gemm_pp")) return rc;
    void* args[] = {&a};
    hipError_t e = hipLaunchKernel(fn, dim3((unsigned)(a.mtiles * a.ntiles * (a.kslices > 1 ? a.kslices : 1))), dim3(512), args, lds, st);
    if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "gemm_pp: HIP launch failed: %s", hipGetErrorString(e));
    return TLXMI_OK;
}

int launch_gemm_pp(int dtype, const Gemm256Args& a, hipStream_t st) {
    if (a.conv) {
        if (dtype == TLXMI_F16) return launch_pp_t<half_t, 2, 2, true>(a, st);
        return launch_pp_t<float, 2, 2, true>(a, st);
    }
    if (a.rowstats || a.stats_out) {      // LayerNorm fold (fp16, Cout % 32 == 0: checked by the entry points in conv_igemm.hip)
        if (dtype != TLXMI_F16 || (a.Cout & 31) || a.kslices > 1) return fail(TLXMI_ERR_UNSUPPORTED, "gemm_pp: the LayerNorm fold is fp16, Cout %% 32 == 0");
        return launch_pp_t<half_t, 2, 2, false, true>(a, st);
    }
    if (dtype == TLXMI_F16) return launch_pp_t<half_t, 2, 2, false>(a, st);
    return launch_pp_t<float, 2, 2, false>(a, st);
}

// 128 x 256 tiles (same preconditions)
int launch_gemm_pp128(int dtype, const Gemm256Args& a, hipStream_t st) {
    if (a.conv) {
        if (dtype == TLXMI_F16) return launch_pp_t<half_t, 1, 2, true>(a, st);
        return launch_pp_t<float, 1, 2, true>(a, st);
    }
    if (a.rowstats || a.stats_out) {      // LayerNorm fold on half-height tiles (few row tiles: Swin-B stage 3 proj, 49 x 2 tiles of 256 x 256)
        if (dtype != TLXMI_F16 || (a.Cout & 31) || a.kslices > 1) return fail(TLXMI_ERR_UNSUPPORTED, "gemm_pp: the LayerNorm fold is fp16, Cout %% 32 == 0");
        return launch_pp_t<half_t, 1, 2, false, true>(a, st);
    }
    if (dtype == TLXMI_F16) return launch_pp_t<half_t, 1, 2, false>(a, st);
    return launch_pp_t<float, 1, 2, false>(a, st);
}

// 256 x 128 tiles (layers with 128 output channels)
int launch_gemm_pp_n128(int dtype, const Gemm256Args& a, hipStream_t st) {
    if (a.conv) {
        if (dtype == TLXMI_F16) return launch_pp_t<half_t, 2, 1, true>(a, st);
        return launch_pp_t<float, 2, 1, true>(a, st);
    }
    if (dtype == TLXMI_F16) return launch_pp_t<half_t, 2, 1, false>(a, st);
    return launch_pp_t<float, 2, 1, false>(a, st);
}

}  // namespace tlxmi

// Stride-1 convolution for thin inputs (C*2 bytes = 32 / 64 / 128 per pixel), fp16: the ResNet stem after
// space-to-depth (4x4 taps over 16 channels -> 64; reference resnet.py:213-218), the 3x3 64 -> 64 convs of
// ResNet layer1 (resnet.py:142-156), DarkNet's 3x3 32 -> 64 (darknet.py:54-58).
//
// Why not the implicit GEMM: conv_igemm.hip gathers every tap of every pixel from L2 into LDS, R*S times
// the input (16x for the stem, 9x for a 3x3), and with only 64 output channels per input byte those layers
// are bound by that L2 -> LDS fill (measured 434 / 535 TFLOP/s, 19 B/clk/CU of fill).  Here
//   * a tile is 256 or 512 consecutive output pixels of one image x 64 channels; a workgroup walks a contiguous
//     range of tiles and keeps the input rows it needs in an LDS ring of NRING rows ((W + 2*pad) pixels, zero
//     padding included): every input row is brought in ONCE per workgroup by LDS-DMA (the R-1 halo rows two
//     consecutive tiles share stay in the ring), the rows the next tile adds are fetched while this tile is
//     computed.  Every tap is read from the ring: consecutive pixels of a row are consecutive in LDS, so the
//     K slice (tap s .. s+S-1, all channels) of filter row r is one contiguous span starting at pixel (y+r, x);
//   * the filters live in registers: a wave owns 32 output channels, 2 x (R*S*C*2/64) MFMA A-fragments
//     (144 VGPRs for 3x3x64), loaded once per workgroup — no filter traffic, no LDS reads for A;
//   * one workgroup per CU; one barrier per tile; waves run free inside a tile (4 ds_read_b128 per 8 MFMAs),
//     the two waves of a SIMD overlap each other.
// 8 waves = 4 pixel groups (64 pixels = 4 MFMA sub-tiles) x 2 channel groups (32 channels = 2 sub-tiles).
// A ring row is padded to a whole number of 1-KiB DMA pieces (PWp pixels), so a piece never straddles rows:
// row / validity / base offset of a piece are wave-uniform, a lane adds constants.  LDS lines of 128 B are
// XOR-swizzled: physical 16-byte slot = slot ^ (line & 7), applied on the DMA source side and on the
// fragment reads; any 16 consecutive pixels of 128 B read conflict-free whatever the first pixel.
// Algorithmic bytes: input once + output once (+ residual); FLOPs 2*M*Cout*R*S*C.
#include "common.h"
#include "conv_halo.h"
#include <stdlib.h>

namespace tlxmi {

typedef __attribute__((address_space(3))) void* lds_ptr_ch_t;
static __device__ __forceinline__ void ch_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_ch_t)lds, 16, voff, 0, 0, 0);
}
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t ch_srd(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
static __device__ __forceinline__ u32x4 ch_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
}
static __device__ __forceinline__ void ch_store16_nt(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 2);
}

// R x S taps, PB = bytes per input pixel (C * 2); S * PB must be a multiple of 64 (one MFMA K slice)
// PI = MFMA pixel sub-tiles per wave: a tile is TP = 64 * PI consecutive pixels (8 where the filter registers
// leave room: per-tile overhead halves)
template <int R, int S, int PB, int PI>
__global__ __launch_bounds__(512) void conv_halo_kernel(const HaloArgs a) {
    constexpr int TP = 64 * PI;          // pixels of a tile
    constexpr int KR = S * PB / 64;      // K slices per filter row
    constexpr int NKK = R * KR;          // K slices in all
    constexpr int OOB = (int)0x80000000;
    constexpr int PSH = PB == 128 ? 7 : PB == 64 ? 6 : 5;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int pg = wid & 3, cg = wid >> 2;
    const int frow = lane & 15, fg = lane >> 4;

    const __amdgpu_buffer_rsrc_t xsrd = ch_srd(a.x, a.x_bytes), wsrd = ch_srd(a.w, a.w_bytes);
    const __amdgpu_buffer_rsrc_t ysrd = ch_srd(a.y, a.y_bytes);
    const __amdgpu_buffer_rsrc_t rsrd = ch_srd(a.res ? a.res : a.y, a.res ? a.res_bytes : 0u);

    // one channel tile (a.nt) per launch: the filters are loaded once, before the tile loop.  Workgroup b takes
    // the contiguous tile range [b*T/G, (b+1)*T/G) in (image, first pixel) order: consecutive tiles of an image.
    const int ntiles = a.N * a.tpi;
    const int t_lo = (int)(((long)blockIdx.x * ntiles) / gridDim.x), t_hi = (int)(((long)(blockIdx.x + 1) * ntiles) / gridDim.x);
    const int n_mine = t_hi - t_lo;
    const float inv_wo = 1.0f / (float)a.Wo;

    struct Tile { int n, m0, npx, oy0, nr; bool ok; };
    auto tile_at = [&](int i) {
        Tile tl;
        tl.ok = i < n_mine;
        const int L = tl.ok ? t_lo + i : t_lo;
        tl.n = L / a.tpi;
        const int tt = L - tl.n * a.tpi;
        tl.m0 = tt * TP;
        tl.npx = a.HoWo - tl.m0 < TP ? a.HoWo - tl.m0 : TP;
        tl.oy0 = tl.m0 / a.Wo;
        tl.nr = (tl.m0 + tl.npx - 1) / a.Wo - tl.oy0 + R;
        return tl;
    };

    // ---- row DMA: piece = 1 KiB = PPP pixels of one ring row per wave instruction.
    // Lane -> physical slot -> logical slot (the swizzle mask line & 7 = (lane >> 3) & 7 is the same for every
    // piece) -> pixel inside the piece and byte inside the pixel: constants of the lane.
    constexpr int PPP = 1024 / PB;                    // pixels per piece
    const int lbyte = ((lane >> 3) << 7) + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);   // logical byte inside the piece
    const int lpix = lbyte >> PSH;                    // pixel inside the piece
    const int loff = lpix * a.x_ld * 2 + (lbyte & (PB - 1));
    const int PR = a.PWp / PPP;                       // pieces per ring row
    const int row_bytes = a.PWp << PSH;
    const int rmask = a.nring - 1;                    // nring is a power of two
    // rows [lo, hi) of image n -> ring slots slot0, slot0+1, ... (mod nring); rows outside the image are zero rows
    auto load_rows = [&](int n, int lo, int hi, int slot0) {
        int py = wid / PR, pc = wid - py * PR;       // wave `wid` takes pieces wid, wid + 8, ...
        while (py < hi - lo) {
            const int iy = lo + py;
            const int x0 = pc * PPP - a.pw;           // input column of the piece's first pixel
            const int ix = x0 + lpix;
            const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const int rowoff = ((n * a.H + iy) * a.W + x0) * a.x_ld * 2;
            ch_dma16(xsrd, smem + ((slot0 + py) & rmask) * row_bytes + (pc << 10), in ? rowoff + loff : OOB);
            pc += 8;
            while (pc >= PR) { pc -= PR; ++py; }
        }
    };
    // ring state: `cursor` = next free slot; rows [.., have_hi) of image `img_ld` are loaded, row base_iy_ld sits in
    // slot base_slot_ld.  The tile being computed may still belong to the image before (cur_base_*).
    int cursor = 0, img_ld = -1, have_hi = 0, base_iy_ld = 0, base_slot_ld = 0;
    auto prefetch = [&](const Tile& tl) {
        if (!tl.ok) return;
        const int iy0 = tl.oy0 - a.ph;
        int lo = have_hi;
        if (tl.n != img_ld) {
            lo = iy0;
            img_ld = tl.n;
            base_iy_ld = iy0;
            base_slot_ld = cursor;
        }
        const int hi = iy0 + tl.nr;
        if (hi > lo) {
            load_rows(tl.n, lo, hi, cursor);
            cursor = (cursor + hi - lo) & rmask;
            have_hi = hi;
        }
    };

    // ---- filters and scale / shift of a channel tile -> registers.  MFMA D row i of sub-tile ci is channel
    // 8*(i>>2) + 4*ci + (i&3) of the wave's 32, so that lane group g owns channels 8g .. 8g+7 (epilogue).
    u32x4 wreg[2][NKK];
    float sc[8], sf[8];
    auto load_filters = [&](int nt) {
        const int cbase = nt * 64 + cg * 32;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
            const int ch = cbase + 8 * (frow >> 2) + 4 * ci + (frow & 3);
            const int woff = ch * a.Kp_bytes + fg * 16;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) wreg[ci][kk] = ch_load16(wsrd, woff + kk * 64);
        }
        const int ch0 = cbase + 8 * fg;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = ch0 + e < a.Cout ? ch0 + e : a.Cout - 1;
            sc[e] = a.scale ? a.scale[ch] : 1.f;
            sf[e] = a.shift ? a.shift[ch] : 0.f;
        }
    };

    Tile cur = tile_at(0);
    prefetch(cur);
    int cur_base_iy = base_iy_ld, cur_base_slot = base_slot_ld;
    load_filters(a.nt);
    // everything above has landed before the loop: no wait on a filter register is left inside it (a wait
    // there would also drain the row DMAs and the stores of the previous tile — one in-order counter).
    // The builtin, not inline asm: the compiler's own wait insertion must see this wait.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __builtin_amdgcn_s_barrier();

    const bool res_after = (a.flags & TLXMI_EPI_RES_AFTER_ACT) != 0;
    for (int i = 0; i < n_mine; ++i) {
        // rows the next tile adds (a new image: all of its rows) go to the slots after the newest row: every wave
        // is past the last reader of what they overwrite (barrier below), and advance + rows of this tile <= nring
        const Tile nxt = tile_at(i + 1);
        if (!(a.debug & 1)) prefetch(nxt);

        // ring slot (for filter row 0) and byte inside the row of this lane's pixel of each sub-tile
        int sq[PI], bx[PI];
#pragma unroll
        for (int pi = 0; pi < PI; ++pi) {
            int pt = pg * (16 * PI) + pi * 16 + frow;
            pt = pt < cur.npx ? pt : cur.npx - 1;      // pixels past the tile read a valid row, never stored
            const int m = cur.m0 + pt;
            const int oy = (int)(((float)m + 0.5f) * inv_wo);
            const int ox = m - oy * a.Wo;
            sq[pi] = cur_base_slot + (oy - a.ph - cur_base_iy);
            bx[pi] = (ox << PSH) + fg * 16;
        }
        const char* pb = smem;
        f32x4 acc[2][PI];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int pi = 0; pi < PI; ++pi) acc[ci][pi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < ((a.debug & 4) ? 1 : R); ++r) {
            int bq[PI];
#pragma unroll
            for (int pi = 0; pi < PI; ++pi) bq[pi] = ((sq[pi] + r) & rmask) * row_bytes + bx[pi];
            if constexpr (PB == 128) {
                // one pixel = one 128-byte line = two K slices: the swizzle is computed once per line
#pragma unroll
                for (int sx = 0; sx < S; ++sx) {
                    u32x4 xf[2][PI];
#pragma unroll
                    for (int pi = 0; pi < PI; ++pi) {
                        const int b = bq[pi] + sx * 128;
                        const int phys = b ^ (((b >> 7) & 7) << 4);
                        xf[0][pi] = *reinterpret_cast<const u32x4*>(pb + phys);
                        xf[1][pi] = *reinterpret_cast<const u32x4*>(pb + (phys ^ 64));
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int pi = 0; pi < PI; ++pi)
#pragma unroll
                            for (int ci = 0; ci < 2; ++ci)
                                acc[ci][pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                    __builtin_bit_cast(half8v, wreg[ci][r * KR + 2 * sx + j]), __builtin_bit_cast(half8v, xf[j][pi]), acc[ci][pi], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kr = 0; kr < KR; ++kr) {
                    u32x4 xf[PI];
#pragma unroll
                    for (int pi = 0; pi < PI; ++pi) {
                        const int b = bq[pi] + kr * 64;
                        const int phys = b ^ (((b >> 7) & 7) << 4);
                        xf[pi] = *reinterpret_cast<const u32x4*>(pb + phys);
                    }
#pragma unroll
                    for (int pi = 0; pi < PI; ++pi)
#pragma unroll
                        for (int ci = 0; ci < 2; ++ci)
                            acc[ci][pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                __builtin_bit_cast(half8v, wreg[ci][r * KR + kr]), __builtin_bit_cast(half8v, xf[pi]), acc[ci][pi], 0, 0, 0);
                }
            }
        }
        // the next tile's rows have landed (these DMAs are older than anything else this wave has in flight except
        // the previous tile's stores); after the barrier nobody reads the rows only this tile needed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        // ---- epilogue: lane (g = fg, px = frow) owns channels 64nt + 32cg + 8g .. +7 of pixel 64pg + 16pi + px
        const int ch0 = a.nt * 64 + cg * 32 + 8 * fg;
        const bool chok = ch0 < a.Cout;     // Cout is a multiple of 8 on this path
        // residual loads (four sub-tiles at a time) before the stores of those sub-tiles: a load behind a store
        // would wait for it (in-order counter)
        u32x4 rr[4];
#pragma unroll
        for (int pi = 0; pi < PI; ++pi) {
            if ((pi & 3) == 0 && a.res) {
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) {
                    const int pt2 = pg * (16 * PI) + (pi + p2) * 16 + frow;
                    const int mg2 = cur.n * a.HoWo + cur.m0 + pt2;
                    rr[p2] = ch_load16(rsrd, (chok && pt2 < cur.npx) ? (mg2 * a.res_ld + ch0) * 2 : OOB);
                }
            }
            const int pt = pg * (16 * PI) + pi * 16 + frow;
            const bool ok = chok && pt < cur.npx;
            const int mg = cur.n * a.HoWo + cur.m0 + pt;
            float v[8];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                v[bb] = acc[0][pi][bb] * sc[bb] + sf[bb];
                v[4 + bb] = acc[1][pi][bb] * sc[4 + bb] + sf[4 + bb];
            }
            float rv[8];
            if (a.res) {
                const half8v hv = __builtin_bit_cast(half8v, rr[pi & 3]);
#pragma unroll
                for (int e = 0; e < 8; ++e) rv[e] = (float)hv[e];
                if (!res_after) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rv[e];
                }
            }
            if (a.act == TLXMI_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (a.act == TLXMI_ACT_LEAKY) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * a.act_param;
            } else if (a.act == TLXMI_ACT_RELU6) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(v[e], 0.f), 6.f);
            } else if (a.act == TLXMI_ACT_HARDSWISH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * fminf(fmaxf(v[e] + 3.f, 0.f), 6.f) * (1.f / 6.f);
            }
            if (a.res && res_after) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            half8v hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
            ch_store16_nt(ysrd, __builtin_bit_cast(u32x4, hv), (ok && !(a.debug & 2)) ? (mg * a.y_ld + ch0) * 2 : OOB);
        }
        cur = nxt;
        if (cur.ok && cur.n == img_ld) { cur_base_iy = base_iy_ld; cur_base_slot = base_slot_ld; }
    }
}

// The activations compiled into the epilogue above; the dispatcher sends others to conv_igemm.hip.
bool conv_halo_act_ok(int act) {
    return act == TLXMI_ACT_NONE || act == TLXMI_ACT_RELU || act == TLXMI_ACT_LEAKY || act == TLXMI_ACT_RELU6 ||
           act == TLXMI_ACT_HARDSWISH;
}

template <int R, int S, int PB, int PI> static int launch_halo_t(const HaloArgs& a, hipStream_t st, int cus) {
    const void* fn = reinterpret_cast<const void*>(&conv_halo_kernel<R, S, PB, PI>);
    const size_t lds = (size_t)a.nring * a.PWp * PB;
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "conv_halo: cannot raise LDS limit: %s", hipGetErrorString(e));
        raised = true;
    }
    const int tiles = a.N * a.tpi;
    int grid = cus & ~7;
    if (grid < 8) grid = 8;
    if (grid > tiles) grid = tiles;
    static const int dbg = [] { const char* e = getenv("TLXMI_DEBUG"); return e ? atoi(e) : 0; }();
    for (int nt = 0; nt < a.ntn; ++nt) {     // one launch per tile of 64 output channels
        HaloArgs b = a;
        b.debug = dbg;
        b.nt = nt;
        void* args[] = {&b};
        hipError_t e = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(512), args, lds, st);
        if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "conv_halo: HIP launch failed: %s", hipGetErrorString(e));
    }
    return TLXMI_OK;
}

// Shapes with a compiled instantiation (R, S, bytes per pixel); returns the pixels of a tile, 0 if none
int conv_halo_tile_pixels(int R, int S, int PB) {
    if (R == 3 && S == 3 && PB == 128) return 256;
    if ((R == 3 && S == 3 && PB == 64) || (R == 4 && S == 4 && PB == 32) || (R == 2 && S == 2 && PB == 32)) return 512;
    return 0;
}

int launch_conv_halo(const HaloArgs& a, hipStream_t st, int cus) {
    if (a.R == 3 && a.S == 3 && a.PB == 128) return launch_halo_t<3, 3, 128, 4>(a, st, cus);
    if (a.R == 3 && a.S == 3 && a.PB == 64) return launch_halo_t<3, 3, 64, 8>(a, st, cus);
    if (a.R == 4 && a.S == 4 && a.PB == 32) return launch_halo_t<4, 4, 32, 8>(a, st, cus);
    if (a.R == 2 && a.S == 2 && a.PB == 32) return launch_halo_t<2, 2, 32, 8>(a, st, cus);
    return fail(TLXMI_ERR_UNSUPPORTED, "conv_halo: no instantiation for %dx%d taps, %d bytes per pixel", a.R, a.S, a.PB);
}

}  // namespace tlxmi

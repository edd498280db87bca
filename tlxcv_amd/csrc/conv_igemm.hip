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
//   "B" operand = input pixels gathered on the fly (16-byte chunks of C; padding taps read a
//                 16-byte zero page)
// Everything is expressed in 16-byte chunks so the same kernel body serves fp16 (8 elements per
// chunk, v_mfma_f32_16x16x32_f16) and fp32 (4 elements per chunk, 4 x v_mfma_f32_16x16x4_f32,
// exact fp32 FMA chain: the parity mode).
//
// Tile: BM pixels x BN channels x 128 bytes of K per step, 256 threads = 4 waves in a 2x2 grid.
// Staging is LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no ds_write): each wave
// instruction fills one 1-KiB piece = 8 tile rows x 128 B; the per-lane SOURCE address does the
// im2col gather.  Two LDS stages (one when K fits a single step): the loads of step k+1 are in
// flight while step k's MFMAs run, one barrier per step.
// LDS image: rows of 128 B (8 chunks); chunk c of row r sits in slot c ^ ((r>>1)&7), which makes
// the ds_read_b128 fragment reads conflict-free under gfx950's b128 lane grouping.  Because the
// DMA destination is lane-linear, the swizzle is applied to the source: lane l of a piece fetches
// chunk (l&7) ^ ((r>>1)&7) of its row.  Pieces are dealt to waves so that this chunk index is
// the same for every piece a lane loads (one im2col position per lane per step).
#include "common.h"
#include <stdio.h>
#include "gemm256.h"
#include "conv_halo.h"
#include "group_conv.h"
#include <stdlib.h>
#include <string.h>

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
    int out_f32;                  // 1: y holds fp32 whatever T is (split-K partial sums: tlxmi_linear_splitk)
    int pp_slices;                // > 1: tlxmi_conv2d_splitk — gemm_pp in CONV mode on K slices, y = fp32 partial planes
    int act;
    float act_param;
    unsigned flags;
    int M;        // N*Ho*Wo output pixels
    int HoWo;
    int kchunks;  // R*S*cpt true 16-byte chunks along K
    int ktiles;   // ceil(kchunks/8)
    int cpt;      // chunks per filter tap = C*sizeof(T)/16
    int Kp_bytes; // packed filter row pitch in bytes (= ktiles*128)
    int mtiles, ntiles;
    int gn;       // N-tiles per column panel of the tile walk
    int vec_io;   // 1: y (and res) rows allow 16-byte vector access
    unsigned x_bytes, w_bytes;  // extents for the buffer descriptors
    unsigned y_bytes;           // extent of y when it may be written through a descriptor (0: plain stores)
    int store_policy;           // 0 plain, 16 sc1 (write-through, line dropped from L2), 2 nt
    int sb_off;                 // LDS byte offset of the block's scale/shift table (2 x BN floats)
    // grouped convolution: blockIdx.y = launch chunk (a few merged groups, block-diagonal filter); every chunk is
    // this same problem at a byte / channel offset.  nchunk == 1 and all strides 0 for the dense convolution.
    int nchunk;
    int gx, gy, gres, gc;       // per-chunk byte offsets into x / y / res pixels, channel offset of scale / shift
    unsigned gw;                // bytes between the packed filters of consecutive chunks
    int overhang;               // 1: some windows run past the bottom / right edge (implicit GEMM gather only)
    int diag;                   // 1: 64 -> 64 channel chunks whose filter is block-diagonal at 32 channels (fp16): the wave
                                // that owns output channels 32c..32c+31 needs only the K half 32c..32c+31 of every tap
};

__device__ __attribute__((aligned(16))) unsigned g_zero_page[4];  // source of padding / tail chunks

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
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
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

// One LDS-DMA wave instruction: 64 lanes x 16 B from buffer offsets `voff` to the 1-KiB piece at `lds`
// (wave-uniform).  Kept out of the kernel template: the builtin must not see template-dependent
// operands (the host pass of hipcc cannot re-check it at instantiation time).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
static __device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds, 16, voff, 0, 0, 0);
}
template <int AUX> static __device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, AUX);
}
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const char* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, bytes, 0x00020000);
}

// WGM = waves along the pixel axis (2 or 4; always 2 along channels), STAGES = LDS-DMA ring depth.
// RESP = the residual tile is prefetched into registers at kernel entry (dense, 16-byte-aligned residual):
// its HBM latency then overlaps the first DMA stage instead of starting after the last MFMA.
template <typename T, int BM, int BN, bool IS_1X1, int WGM, int STAGES, bool RESP, bool DIAG = false>
__global__ __launch_bounds__(WGM * 128) void conv_igemm_kernel(const ConvArgs a0) {
    static_assert(!DIAG || (BN == 64 && sizeof(T) == 2 && !IS_1X1), "DIAG: 64-channel fp16 chunks of a grouped conv");
    constexpr int ES = (int)sizeof(T);
    ConvArgs a = a0;
    if (a.nchunk > 1) {
        const int g = blockIdx.y;
        a.x += (size_t)g * a.gx; a.x_bytes -= (unsigned)(g * a.gx);
        a.w += (size_t)g * a.gw;
        a.y += (size_t)g * a.gy; if (a.y_bytes) a.y_bytes -= (unsigned)(g * a.gy);
        if (a.res) a.res += (size_t)g * a.gres;
        if (a.scale) a.scale += g * a.gc;
        if (a.shift) a.shift += g * a.gc;
    }
    constexpr int NW = WGM * 2, NT = NW * 64;     // waves, threads
    constexpr int WM = BM / WGM, WN = BN / 2;     // wave tile (pixels x channels)
    constexpr int PI = WM / 16, CI = WN / 16;
    constexpr int XP = BM / 8 / NW, WP = BN / 8 / NW;  // 1-KiB pieces per wave per K-step
    constexpr int BUF = (BM + BN) * 128;
    constexpr int OOB = (int)0x80000000;       // any offset >= 2^31 fails the descriptor range check -> zeros
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);

    // ---- block -> tile, XCD-aware: blocks that share an XCD (id % 8) walk consecutive tiles,
    // N-tiles fastest, so the activation rows of one M-tile are re-read from that XCD's L2.
    int tile_m, tile_n;
    {
        const int nb = a.mtiles * a.ntiles, id = blockIdx.x;
        const int xcd = id & 7, qd = nb >> 3, rm = nb & 7;
        const int L = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (id >> 3);
        // column panels of `gn` N-tiles (filter panel <= ~1.5 MB stays in the XCD's L2), M fastest inside
        const int per_group = a.mtiles * a.gn;
        int grp = L / per_group;
        const int ngroups = (a.ntiles + a.gn - 1) / a.gn;
        if (grp > ngroups - 1) grp = ngroups - 1;
        const int rem = L - grp * per_group;
        const int gn_here = (a.ntiles - grp * a.gn) < a.gn ? (a.ntiles - grp * a.gn) : a.gn;
        tile_m = rem / gn_here;
        tile_n = grp * a.gn + rem % gn_here;
    }
    const int bm0 = tile_m * BM, bn0 = tile_n * BN;

    // Buffer descriptors: 32-bit per-lane byte offsets, and the hardware range check supplies the
    // zeros of padding taps / tail rows / tail K chunks (offset OOB) with no select on the data path.
    const __amdgpu_buffer_rsrc_t xsrd = make_srd(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t wsrd = make_srd(a.w, a.w_bytes);
    const __amdgpu_buffer_rsrc_t ysrd = make_srd(a.y, a.y_bytes);

    // ---- loader state.  Wave `wid` fills pieces wid, wid+4, ... ; inside a piece lane l owns tile row
    // 8*piece + (l>>3), LDS slot (l&7).  (row>>1)&7 = (4*(wid&1) + (l>>4)) & 7 for all of them, so the
    // logical K chunk a lane fetches is the same for every piece it loads.
    const int lr = lane >> 3;
    const int lchunk = (lane & 7) ^ ((((wid & 1) << 2) + (lane >> 4)) & 7);
    int xo[XP];             // byte offset of (pixel row i, tap (0,0), chunk 0); OOB for tail rows (1x1 path)
    int hi0[XP], wi0[XP];   // top-left input coordinate of the pixel's receptive field (general path)
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int m = bm0 + 8 * (wid + NW * i) + lr;
        if constexpr (IS_1X1) {
            if (a.sh == 1 && a.sw == 1) {
                xo[i] = m < a.M ? m * a.x_ld * ES : OOB;
            } else {
                const int mm = m < a.M ? m : 0;
                const int n = mm / a.HoWo, rem = mm - n * a.HoWo;
                const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
                xo[i] = m < a.M ? ((n * a.H + ho * a.sh) * a.W + wo * a.sw) * a.x_ld * ES : OOB;
            }
            hi0[i] = wi0[i] = 0;
        } else {
            const int mm = m < a.M ? m : 0;
            const int n = mm / a.HoWo, rem = mm - n * a.HoWo;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            hi0[i] = m < a.M ? ho * a.sh - a.ph : -(1 << 28);
            wi0[i] = wo * a.sw - a.pw;
            xo[i] = ((n * a.H + hi0[i]) * a.W + wi0[i]) * a.x_ld * ES;   // may be "negative" under padding
        }
    }
    // Filter rows are permuted inside each group of 32 channels: LDS row 32c + 16e + 4g + b holds
    // channel 32c + 8g + 4e + b, so that MFMA sub-tile (2c+e), accumulator register b of lane group g
    // is channel 32c + 8g + 4e + b: one lane then owns channels 32c+8g .. +7 across the sub-tile pair.
    int wo_[WP];
#pragma unroll
    for (int j = 0; j < WP; ++j) {
        const int rho = 8 * (wid + NW * j) + lr;
        const int n = (rho & ~31) | (((rho >> 2) & 3) << 3) | (((rho >> 4) & 1) << 2) | (rho & 3);
        wo_[j] = (bn0 + n) * a.Kp_bytes;
    }

    // K position of this lane's chunk: q-th chunk -> (tap r,s ; chunk cc inside the tap)
    int q = lchunk;
    int r = 0, s = 0, cc = 0;
    if constexpr (!IS_1X1) {
        const int tap = q / a.cpt;
        cc = q - tap * a.cpt;
        r = tap / a.S;
        s = tap - r * a.S;
    }
    auto stage = [&](int buf) {
        char* b = smem + buf * BUF;
        const bool kv = q < a.kchunks;
        if constexpr (IS_1X1) {
            const int d = kv ? q * 16 : OOB;
#pragma unroll
            for (int i = 0; i < XP; ++i)
                lds_dma16(xsrd, b + (wid + NW * i) * 1024, xo[i] + d);
        } else {
            const int hoff = r * a.dh, woff = s * a.dw;
            const int d = (hoff * a.W + woff) * a.x_ld * ES + cc * 16;
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const bool ok = kv && (unsigned)(hi0[i] + hoff) < (unsigned)a.H && (unsigned)(wi0[i] + woff) < (unsigned)a.W;
                lds_dma16(xsrd, b + (wid + NW * i) * 1024, ok ? xo[i] + d : OOB);
            }
        }
#pragma unroll
        for (int j = 0; j < WP; ++j)
            lds_dma16(wsrd, b + BM * 128 + (wid + NW * j) * 1024, wo_[j] + q * 16);
    };
    auto advance = [&]() {
        q += 8;
        if constexpr (!IS_1X1) {
            if (a.cpt >= 8) {
                // at most one tap boundary per 8-chunk step: branch-free (lanes wrap at different steps)
                cc += 8;
                const bool w1 = cc >= a.cpt;
                cc -= w1 ? a.cpt : 0;
                s += w1 ? 1 : 0;
                const bool w2 = s == a.S;
                s = w2 ? 0 : s;
                r += w2 ? 1 : 0;
            } else {
                // few-channel inputs (space-to-depth stems): several taps per step, re-derive from q
                const int tap = q / a.cpt;
                cc = q - tap * a.cpt;
                r = tap / a.S;
                s = tap - r * a.S;
            }
        }
    };

    // ---- fragment read offsets (per lane constants); sub-tile bases are multiples of 16 rows
    const int wave_m0 = (wid % WGM) * WM, wave_n0 = (wid / WGM) * WN;
    const int frow = lane & 15, fg = lane >> 4;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = lds_off(frow, 4 * ks + fg);
    const int xfrag = wave_m0 * 128;
    const int wfrag = BM * 128 + wave_n0 * 128;

    f32x4 acc[CI][PI];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int pi = 0; pi < PI; ++pi) acc[ci][pi] = f32x4{0.f, 0.f, 0.f, 0.f};

    // epilogue geometry (phase 2): BN/8 lanes per pixel row, RPP rows per pass
    constexpr int CPR = BN / 8;          // 8-channel chunks per row
    constexpr int RPP = NT / CPR;        // rows per pass
    constexpr int NIT = BM / RPP;
    constexpr int RW = RESP ? NIT : 1;
    u32x4 rres[RW][ES / 2];              // prefetched residual, raw

    // ---- K loop.  The DMA ring runs STAGES-1 steps ahead of the MFMAs; one barrier per step:
    //   wait (counted vmcnt) until this wave's pieces of step kt have landed, barrier (everyone's have,
    //   and everyone is done reading the buffer of step kt-1), re-fill that buffer with step kt+STAGES-1.
    stage(0);
    if constexpr (STAGES == 3) {
        if (a.ktiles > 1) { advance(); stage(1); }
    }
    // folded-BatchNorm / bias table of this block's BN channels -> LDS now, so the epilogue does not start
    // with a dependent global load (visible to every wave after the K loop's barriers)
    float* sbuf = reinterpret_cast<float*>(smem + a.sb_off);
    if (t < BN) {
        const int ch = bn0 + t < a.Cout ? bn0 + t : a.Cout - 1;
        sbuf[t] = a.scale ? a.scale[ch] : 1.f;
        sbuf[BN + t] = a.shift ? a.shift[ch] : 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS writes are done before this wave's next barrier
    }
    if constexpr (RESP) {
        const int ch0 = bn0 + 8 * (t % CPR);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int m = bm0 + t / CPR + it * RPP;
            const bool ok = m < a.M && ch0 + 8 <= a.Cout;
            const u32x4* rp = reinterpret_cast<const u32x4*>(a.res + ((size_t)(ok ? m : 0) * a.res_ld + (ok ? ch0 : 0)) * ES);
#pragma unroll
            for (int h = 0; h < ES / 2; ++h) rres[it][h] = rp[h];
        }
    }
    int buf = 0;
    for (int kt = 0; kt < a.ktiles; ++kt) {
        if constexpr (STAGES == 3) {
            if (kt + 1 < a.ktiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XP + WP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < a.ktiles) {
                advance();
                stage(buf >= 1 ? buf - 1 : 2);   // (kt + 2) % 3
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < a.ktiles) {
                advance();
                stage(buf ^ 1);
            }
        }
        const char* b = smem + buf * BUF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if constexpr (DIAG) {
                if (ks != wid / WGM) continue;     // the other 32 input channels meet only zeros of this wave's filter rows
            }
            u32x4 wf[CI], xf[PI];
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
                wf[ci] = *reinterpret_cast<const u32x4*>(b + wfrag + ci * 2048 + foff[ks]);
#pragma unroll
            for (int pi = 0; pi < PI; ++pi)
                xf[pi] = *reinterpret_cast<const u32x4*>(b + xfrag + pi * 2048 + foff[ks]);
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int pi = 0; pi < PI; ++pi) acc[ci][pi] = Mma<T>::run(wf[ci], xf[pi], acc[ci][pi]);
        }
        buf = (buf + 1 == STAGES) ? 0 : buf + 1;
    }

    // ---- epilogue, phase 1: acc*scale + shift -> fp32 tile in LDS (re-using the stage buffers).
    // Row = pixel, BN floats per row; 16-byte chunk j of row r is stored at chunk j ^ (r & 7) so the
    // eight lanes a ds_write_b128 serves at a time (eight pixels, same channels) hit distinct banks.
    __syncthreads();  // every wave is done reading the stage buffers
    float* Ct = reinterpret_cast<float*>(smem);
    {
        const int g = lane >> 4, px = lane & 15;
        constexpr int CP = CI / 2;
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
            const int col = wave_n0 + 32 * cp + 8 * g;  // tile-local first channel of this lane's 8
            float sc[8], sf[8];
            {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(sbuf + col), s1 = *reinterpret_cast<const f32x4*>(sbuf + col + 4);
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(sbuf + BN + col), h1 = *reinterpret_cast<const f32x4*>(sbuf + BN + col + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[e] = s0[e]; sc[4 + e] = s1[e]; sf[e] = h0[e]; sf[4 + e] = h1[e]; }
            }
#pragma unroll
            for (int pi = 0; pi < PI; ++pi) {
                const int row = wave_m0 + pi * 16 + px;
                f32x4 lo, hi;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    lo[bb] = acc[2 * cp][pi][bb] * sc[bb] + sf[bb];
                    hi[bb] = acc[2 * cp + 1][pi][bb] * sc[4 + bb] + sf[4 + bb];
                }
                const int c16 = col >> 2;
                *reinterpret_cast<f32x4*>(Ct + row * BN + (((c16) ^ (row & 7)) << 2)) = lo;
                *reinterpret_cast<f32x4*>(Ct + row * BN + (((c16 + 1) ^ (row & 7)) << 2)) = hi;
            }
        }
    }
    __syncthreads();

    // ---- epilogue, phase 2: whole rows leave the block: BN/8 consecutive lanes cover one pixel's BN
    // channels (full 128-byte lines); the residual is read the same way.
    {
        const int chunk = t % CPR;
        const int ch0 = bn0 + 8 * chunk;
        const bool res_after = (a.flags & TLXMI_EPI_RES_AFTER_ACT) != 0;
        const bool full = a.vec_io && (ch0 + 8 <= a.Cout);
        if (ch0 < a.Cout) {
            const int row0 = t / CPR;
            const int OES = a.out_f32 ? 4 : ES;           // bytes of an output element
            const size_t ystep = (size_t)RPP * a.y_ld * OES, rstep = (size_t)RPP * a.res_ld * ES;
            // one specialised, fully unrolled row loop per activation (dispatch happens once, not per element)
            auto rows = [&](auto act_tag) {
                constexpr int ACT = decltype(act_tag)::value;
                const char* rp = a.res ? a.res + ((size_t)(bm0 + row0) * a.res_ld + ch0) * ES : nullptr;
                char* yp = a.y + ((size_t)(bm0 + row0) * a.y_ld + ch0) * OES;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int row = row0 + it * RPP;
                    const int m = bm0 + row;
                    if (m >= a.M) break;
                    if (a.strided_n) {
                        const int n = m / a.HoWo, p = m - n * a.HoWo;
                        yp = a.y + ((size_t)n * a.y_nstride + (size_t)p * a.y_ld + ch0) * ES;
                        if (a.res) rp = a.res + ((size_t)n * a.res_nstride + (size_t)p * a.res_ld + ch0) * ES;
                    }
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(Ct + row * BN + (((2 * chunk) ^ (row & 7)) << 2));
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(Ct + row * BN + (((2 * chunk + 1) ^ (row & 7)) << 2));
                    float v[8];
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) { v[bb] = lo[bb]; v[4 + bb] = hi[bb]; }
                    float rv[8];
                    if (a.res) {
                        if constexpr (RESP) {
                            if constexpr (ES == 2) {
                                const half8v h = __builtin_bit_cast(half8v, rres[it][0]);
#pragma unroll
                                for (int e = 0; e < 8; ++e) rv[e] = (float)h[e];
                            } else {
                                const f32x4 r0 = __builtin_bit_cast(f32x4, rres[it][0]);
                                const f32x4 r1 = __builtin_bit_cast(f32x4, rres[it][ES / 2 - 1]);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { rv[e] = r0[e]; rv[4 + e] = r1[e]; }
                            }
                        } else if (full) {
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
                    if constexpr (ACT == TLXMI_ACT_GELU && ES == 2) {
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
                    if (a.out_f32 && sizeof(T) == 2) {        // fp32 partial sums of an fp16 GEMM (dense rows, no batch stride)
                        float* yf = reinterpret_cast<float*>(yp);
                        if (full) {
                            *reinterpret_cast<f32x4*>(yf) = f32x4{v[0], v[1], v[2], v[3]};
                            *reinterpret_cast<f32x4*>(yf + 4) = f32x4{v[4], v[5], v[6], v[7]};
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (ch0 + e < a.Cout) yf[e] = v[e];
                        }
                    } else if (full && a.store_policy != 0 && !a.strided_n) {
                        // outputs are never re-read by this launch: keep them from evicting the operand panels from L2
                        const int yoff = (int)(yp - a.y);
                        u32x4 pk[ES / 2];
                        if constexpr (ES == 2) {
                            half8v h;
#pragma unroll
                            for (int e = 0; e < 8; ++e) h[e] = (half_t)v[e];
                            pk[0] = __builtin_bit_cast(u32x4, h);
                        } else {
                            f32x4 f0, f1;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { f0[e] = v[e]; f1[e] = v[4 + e]; }
                            pk[0] = __builtin_bit_cast(u32x4, f0);
                            pk[ES / 2 - 1] = __builtin_bit_cast(u32x4, f1);
                        }
#pragma unroll
                        for (int hh = 0; hh < ES / 2; ++hh) {
                            if (a.store_policy == 16) buf_store16<16>(ysrd, pk[hh], yoff + 16 * hh);
                            else buf_store16<2>(ysrd, pk[hh], yoff + 16 * hh);
                        }
                    } else if (full) {
                        store8<T>(yp, v);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (ch0 + e < a.Cout) reinterpret_cast<T*>(yp)[e] = (T)v[e];
                    }
                    yp += ystep;
                    rp += rstep;
                }
            };
            TLXMI_DISPATCH_ACT(a.act, rows)
        }
    }
}

static int num_cus() { return device_cus(); }      // core.hip: cached per device

// resident blocks per CU for a tile shape: LDS bound (160 KiB) and the register allocation hipcc
// reports for this kernel (128x128: 156 -> 3 waves/SIMD, others <= 128 -> 4)
template <typename T, int BM, int BN, int WGM, int STAGES> static int launch(const ConvArgs& a, hipStream_t st) {
    ConvArgs b = a;
    b.mtiles = (a.M + BM - 1) / BM;
    b.ntiles = (a.Cout + BN - 1) / BN;
    {
        const long budget = tune_int("TLXMI_PANEL_KB", 1l << 20) * 1024;   // default: no panels (N fastest); measured neutral
        long gn = budget / ((long)BN * a.Kp_bytes);
        b.gn = (int)(gn < 1 ? 1 : (gn > b.ntiles ? b.ntiles : gn));
    }
    // LDS: the DMA ring (one buffer is enough when K fits a single step), re-used by the fp32 epilogue tile
    size_t lds = (size_t)(a.ktiles > 1 ? STAGES : 1) * (BM + BN) * 128;
    if (lds < (size_t)BM * BN * 4) lds = (size_t)BM * BN * 4;
    b.sb_off = (int)lds;
    lds += 2 * BN * sizeof(float);
    const long grid = (long)b.mtiles * b.ntiles;
    const bool is1x1 = a.R == 1 && a.S == 1 && a.ph == 0 && a.pw == 0;
    const bool resp = a.res && a.vec_io && !a.strided_n;
    const void* fns[4] = {reinterpret_cast<const void*>(&conv_igemm_kernel<T, BM, BN, false, WGM, STAGES, false>),
                          reinterpret_cast<const void*>(&conv_igemm_kernel<T, BM, BN, false, WGM, STAGES, true>),
                          reinterpret_cast<const void*>(&conv_igemm_kernel<T, BM, BN, true, WGM, STAGES, false>),
                          reinterpret_cast<const void*>(&conv_igemm_kernel<T, BM, BN, true, WGM, STAGES, true>)};
    int which = (is1x1 ? 2 : 0) + (resp ? 1 : 0);
    const void* fns_diag[2] = {nullptr, nullptr};
    if constexpr (BN == 64 && sizeof(T) == 2) {
        fns_diag[0] = reinterpret_cast<const void*>(&conv_igemm_kernel<T, BM, BN, false, WGM, STAGES, false, true>);
        fns_diag[1] = reinterpret_cast<const void*>(&conv_igemm_kernel<T, BM, BN, false, WGM, STAGES, true, true>);
    }
    const bool diag = a.diag && !is1x1 && fns_diag[0] != nullptr;
    const void* fn = diag ? fns_diag[resp ? 1 : 0] : fns[which];
    if (diag) which = 4 + (resp ? 1 : 0);
    if (lds > 64 * 1024)
        if (int rc = raise_lds_limit(fn, 160 * 1024, "conv2d")) return rc;
    void* args[] = {&b};
    hipError_t e = hipLaunchKernel(fn, dim3((unsigned)grid, (unsigned)a.nchunk), dim3(WGM * 128), args, lds, st);
    if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "conv2d: HIP launch failed: %s", hipGetErrorString(e));
    return TLXMI_OK;
}

template <typename T> static int dispatch(const ConvArgs& a, hipStream_t st, bool allow_stream = true, bool allow_split = true) {
    // Tile choice = max over the four shapes of (grid quantisation efficiency) x (shape efficiency):
    // a launch of B blocks on S = CUs x resident-blocks-per-CU slots runs ceil(B/S) rounds, so B/(rounds*S)
    // of the machine does useful work; bigger tiles re-use operands better (fewer LDS bytes per MFMA).
    // The shape efficiency depends on the regime (measured on MI355X, tools/conv_micro.py sweep): layers
    // that move many output/residual bytes per FLOP (the 1x1 "expand" convs with a skip connection) are
    // latency/HBM-bound and want many small resident blocks; MFMA-bound layers want the 128x128 tile.
    // Thin inputs (<= 128 bytes per pixel), stride 1, few output channels: the gather of every tap through LDS is
    // the bound; conv_halo.hip loads the input rows once and keeps the filters in registers.
    const bool pool = (a.flags & TLXMI_EPI_MAXPOOL_3S2P1) != 0;
    if constexpr (sizeof(T) == 2) {
        const int PB = a.C * 2;
        if (pool) {
            // conv -> epilogue -> maxpool(3, 2, 1) in one launch (the ResNet stem, resnet.py:287-290): conv_halo's POOL
            // variant only; y is [N][Ho/2][Wo/2][y_ld]
            if (!(a.nchunk == 1 && !a.overhang && a.sh == 1 && a.sw == 1 && a.dh == 1 && a.dw == 1 && conv_halo_pool_ok(a.R, a.S, PB, a.Ho, a.Wo) &&
                  conv_halo_pool_act_ok(a.act) && !a.strided_n && a.vec_io && !a.res && a.Cout % 8 == 0 && a.Cout <= 128 && a.y_bytes != 0))
                return fail(TLXMI_ERR_UNSUPPORTED, "conv2d: TLXMI_EPI_MAXPOOL_3S2P1 needs the fp16 4x4 / 16-channel stem geometry (Wo = 112, even Ho, no residual)");
            HaloArgs h;
            h.x = a.x; h.w = a.w; h.y = a.y; h.scale = a.scale; h.shift = a.shift; h.res = nullptr;
            h.N = a.N; h.H = a.H; h.W = a.W; h.Cout = a.Cout; h.R = a.R; h.S = a.S; h.ph = a.ph; h.pw = a.pw;
            h.Ho = a.Ho; h.Wo = a.Wo; h.HoWo = a.HoWo; h.x_ld = a.x_ld; h.y_ld = a.y_ld; h.res_ld = 0;
            h.PB = PB; h.Kp_bytes = a.Kp_bytes; h.act = a.act; h.act_param = a.act_param; h.flags = a.flags;
            h.tpi = a.Ho / 2;                      // a tile = two conv rows
            h.ntn = (a.Cout + 63) / 64; h.nt = 0;
            h.PW = a.Wo + a.S - 1;
            const int ppp = 1024 / PB;
            h.PWp = (h.PW + ppp - 1) / ppp * ppp;
            h.nring = 16;                          // 2 rows + 3 halo rows per tile, three tiles resident
            h.x_bytes = a.x_bytes; h.w_bytes = a.w_bytes;
            h.y_bytes = (unsigned)((long long)a.N * (a.Ho / 2) * (a.Wo / 2) * a.y_ld * 2);
            h.res_bytes = 0u;
            h.pool = 1;
            return launch_conv_halo(h, st, num_cus());
        }
        const int forced_h = (int)tune_int("TLXMI_HALO", -1);   // tuning flavour: 0 = off (A/B runs)
        if (forced_h != 0 && a.nchunk == 1 && !a.overhang && a.sh == 1 && a.sw == 1 && a.dh == 1 && a.dw == 1 && (a.R > 1 || a.S > 1) && conv_halo_tile_pixels(a.R, a.S, PB) > 0 &&
            conv_halo_act_ok(a.act) && !a.strided_n && a.vec_io && a.Cout % 8 == 0 && a.Cout <= 128 && a.y_bytes != 0 &&
            a.HoWo >= 1024 && (!a.res || (long long)a.M * a.res_ld * 2 < (1ll << 31))) {
            HaloArgs h;
            h.pool = 0;
            h.x = a.x; h.w = a.w; h.y = a.y; h.scale = a.scale; h.shift = a.shift; h.res = a.res;
            h.N = a.N; h.H = a.H; h.W = a.W; h.Cout = a.Cout; h.R = a.R; h.S = a.S; h.ph = a.ph; h.pw = a.pw;
            h.Ho = a.Ho; h.Wo = a.Wo; h.HoWo = a.HoWo; h.x_ld = a.x_ld; h.y_ld = a.y_ld; h.res_ld = a.res_ld;
            h.PB = PB; h.Kp_bytes = a.Kp_bytes; h.act = a.act; h.act_param = a.act_param; h.flags = a.flags;
            const int tp = conv_halo_tile_pixels(a.R, a.S, PB);
            h.tpi = (a.HoWo + tp - 1) / tp;
            h.ntn = (a.Cout + 63) / 64; h.nt = 0;
            h.PW = a.Wo + a.S - 1;
            const int span = (a.Wo - 1 + tp - 1) / a.Wo + 1;       // output rows a tile of tp consecutive pixels can touch
            const int ppp = 1024 / PB;
            h.PWp = (h.PW + ppp - 1) / ppp * ppp;
            // ring: tile p is read while tile p+1 is resident and tile p+2 lands — the rows of a tile (span + R - 1)
            // plus those two later tiles add (<= span each, or a whole first tile of the next image) must fit
            h.nring = 8;
            while (h.nring < 2 * (span + a.R - 1) + span) h.nring *= 2;
            h.x_bytes = a.x_bytes; h.w_bytes = a.w_bytes; h.y_bytes = a.y_bytes;
            h.res_bytes = a.res ? (unsigned)((long long)a.M * a.res_ld * 2) : 0u;
            if ((long)h.nring * h.PWp * PB + 512 <= 160 * 1024) return launch_conv_halo(h, st, num_cus());
        }
    }
    if (pool) return fail(TLXMI_ERR_UNSUPPORTED, "conv2d: TLXMI_EPI_MAXPOOL_3S2P1 is an fp16-only fusion");
    struct Cand { int bm, bn; float eff; };
    const double flops = 2.0 * a.M * (double)a.Cout * a.kchunks * (16 / (int)sizeof(T));
    const double obytes = (double)a.M * a.Cout * sizeof(T) * (a.res ? 2.0 : 1.0);
    const double obi = obytes / flops;
    // candidate 4 = 256x128 pixels x channels, 8 waves, 3-deep DMA ring (one block per CU): MFMA-bound layers
    // candidate 5 = gemm256.hip: 256x256, 8 waves, 4-deep ring of 64-byte K steps (pure GEMM rows only)
    // candidate 6 = gemm256.hip's 256x128 variant: 4 waves, 3-deep ring, two workgroups per CU
    // candidate 7 = gemm_pp.hip: 256x256, 8 waves in two antiphase groups, 128-byte K tiles
    // candidate 8 = gemm_stream.hip: candidate 7 as a persistent kernel (one K-tile stream per CU)
    // candidate 9 = gemm_pp.hip on 128 x 256 tiles (3x3 convs whose 256 x 256 tiles would fill under half the CUs)
    // candidate 10 = gemm_pp.hip on 256 x 128 tiles (3x3 convs with 128 output channels)
    // candidate 11 = gemm_w4.hip: 256 x 256 persistent, four waves with 128 x 128 wave tiles (pure GEMM rows)
    constexpr int NC = 12;
    Cand cands[NC] = {{128, 128, 1.00f}, {64, 128, 0.85f}, {128, 64, 0.85f}, {64, 64, 0.70f}, {256, 128, 0.90f},
                      {256, 256, 1.12f}, {256, 128, 1.20f}, {256, 256, 1.40f}, {256, 256, 1.50f}, {128, 256, 1.25f},
                      {256, 128, 1.25f}, {256, 256, 0.f}};
    const bool gemm128_ok = a.nchunk == 1 && a.R == 1 && a.S == 1 && a.ph == 0 && a.pw == 0 && a.sh == 1 && a.sw == 1 && !a.strided_n &&
                            a.vec_io && a.Cout % 8 == 0 && a.y_bytes != 0 && a.Cout >= 128 && a.ktiles >= 2 &&
                            (!a.res || (long long)a.M * a.res_ld * (long long)sizeof(T) < (1ll << 31));
    const bool gemm256_ok = gemm128_ok && a.Cout >= 256;
    // 3x3 (R x 3) convs on the antiphase GEMM kernel: a 128-byte K tile must lie inside one filter tap
    const int tpk = a.cpt / 8;     // K tiles per tap
    // (also the strided 1x1 projection shortcuts, resnet.py:246-261: one "tap", rows gathered at stride 2)
    const bool strided1x1 = a.R == 1 && a.S == 1 && a.ph == 0 && a.pw == 0 && (a.sh > 1 || a.sw > 1);
    const bool as_conv = !(a.R == 1 && a.S == 1 && a.ph == 0 && a.pw == 0 && a.sh == 1 && a.sw == 1);   // needs the gather of CONV mode
    const bool pp_conv128_ok = a.nchunk == 1 && !a.overhang && ((a.S == 3 && a.R <= 3) || strided1x1) && a.dh == 1 && a.dw == 1 && !a.strided_n &&
                               a.vec_io && a.Cout % 8 == 0 && a.Cout >= 128 && a.y_bytes != 0 && a.cpt % 8 == 0 && (tpk & (tpk - 1)) == 0 &&
                               (!a.res || (long long)a.M * a.res_ld * (long long)sizeof(T) < (1ll << 31));
    const bool pp_conv_ok = pp_conv128_ok && a.Cout >= 256;
    // K = 128 (two K tiles) and up to 512 output channels on many rows: HBM-bound; the filter-in-registers kernel reads every
    // activation row once and writes every output row once (gemm_wreg.hip).  K = 256: only the wide layers (768 / 1024 channels:
    // qkv / fc1 of Swin-B stage 2, 60 -> 55 and 112 -> 87 us at batch 128; 256 / 512 channels measure equal or slower — 128 filter
    // registers a wave leave one workgroup of <= 8 waves per CU).  TLXMI_WREG (tuning flavour): 0 the tiled kernels, 2 K = 128 only,
    // 3 every K = 256 width.
    const long wreg = tune_int("TLXMI_WREG", 1);
    const bool wreg_k256 = a.kchunks == 32 && wreg != 2 && a.M >= 32768 && (a.Cout >= 768 || wreg == 3);
    if (gemm128_ok && sizeof(T) == 2 && (a.kchunks == 16 || wreg_k256) && a.M >= 16384 && a.pp_slices <= 1 && !a.out_f32 && wreg &&
        tune_int("TLXMI_TILE", -1) < 0) {
        Gemm256Args g;
        g.debug = 0; g.conv = 0;
        g.x = a.x; g.w = a.w; g.y = a.y; g.scale = a.scale; g.shift = a.shift; g.res = a.res;
        g.M = a.M; g.Cout = a.Cout; g.x_ld = a.x_ld; g.y_ld = a.y_ld; g.res_ld = a.res_ld;
        g.kchunks = a.kchunks; g.ksteps = a.Kp_bytes / 128; g.Kp_bytes = a.Kp_bytes;
        g.act = a.act; g.act_param = a.act_param; g.flags = a.flags; g.mtiles = g.ntiles = 0; g.gn = 1;
        g.x_bytes = a.x_bytes; g.w_bytes = a.w_bytes; g.y_bytes = a.y_bytes;
        g.res_bytes = a.res ? (unsigned)((long long)a.M * a.res_ld * (long long)sizeof(T)) : 0u;
        if (gemm_wreg_ok(TLXMI_F16, g)) return launch_gemm_wreg(g, st);
    }
    if (!gemm256_ok) cands[5].eff = cands[6].eff = cands[8].eff = cands[11].eff = 0.f;
    if (!gemm256_ok && !pp_conv_ok) cands[7].eff = 0.f;
    // 128 x 256 tiles for plain GEMM rows too (few row tiles: 7 x 7 stage, 2048 -> 512); TLXMI_PP128=0: convs only (A/B)
    const int pp128_gemm = (int)tune_int("TLXMI_PP128", 1);
    if (!pp_conv_ok && !(gemm256_ok && pp128_gemm)) cands[9].eff = 0.f;
    if (!pp_conv128_ok) cands[10].eff = 0.f;      // (1x1 layers: reachable through TLXMI_TILE=10 only)
    cands[5].eff = 0.f;   // superseded by candidate 7 (same tile, antiphase wave groups); kept for A/B runs (TLXMI_TILE=5)
    if (obi >= 0.0015 || a.ktiles < 4) cands[4].eff = 0.f;
    // regimes by output bytes per FLOP (tools/ab_tiles.py sweep over the ResNet-50 / ViT-B / Swin-B layer shapes):
    // the 256-row GEMM kernels win up to ~0.005 (256x256) / ~0.02 (256x128, two workgroups per CU); beyond that
    // the layer is HBM / latency-bound and wants many small resident blocks
    if (obi >= 0.005) cands[7].eff = cands[8].eff = cands[9].eff = 0.f;
    if (obi >= 0.020) cands[6].eff = 0.f;
    if (obi >= 0.020 || (obi >= 0.010 && a.ktiles == 1)) { cands[0].eff = 0.65f; cands[1].eff = 0.80f; cands[2].eff = 0.85f; cands[3].eff = 1.00f; }   // 64 -> 256 (+ skip) at 56x56: one K step
    else if (obi >= 0.010) { cands[0].eff = 1.00f; cands[1].eff = 0.95f; cands[2].eff = 0.95f; cands[3].eff = 0.85f; }   // 128 -> 512 + skip at 28x28
    else if (obi >= 0.0015) { cands[0].eff = 0.85f; cands[1].eff = 0.90f; cands[2].eff = 1.00f; cands[3].eff = 0.90f; }
    if (!allow_stream) cands[7].eff = cands[8].eff = 0.f;
    // the persistent kernel hides the plain epilogue but not the GELU arithmetic (measured: fc1 of ViT-B)
    // (except with few row tiles inside a two-stream forward planned for the whole device — Swin-B stage 3, fc1 12544 x 512 ->
    //  2048 at half batch 64: forward 7.88 -> 7.78 ms with the persistent kernel, tools/tile_search.py; ViT-B/16's fc1: equal)
    const bool gelu_stream_ok = (a.flags & TLXMI_PLAN_SHARED_FULL) && a.M < 16384 && a.ktiles >= 8;
    if (a.act == TLXMI_ACT_GELU && !gelu_stream_ok && !tune_int("TLXMI_GELU_STREAM", 0)) cands[8].eff = 0.f;
    // One workgroup per CU: the last round of a 256x256 launch runs a whole tile time however few tiles it
    // has.  When it would be at most half full, the rows of the full rounds go to the 256x256 kernel and the
    // remaining rows to a second launch of the same kernel on 128x256 tiles (gemm_pp128): twice the tiles, about
    // 0.6 of the time, still one round (tail split, below).
    const long t256 = (long)((a.M + 255) / 256) * ((a.Cout + 255) / 256);
    // TLXMI_PLAN_SHARED_* in the descriptor's flags (tuning flavour: TLXMI_PLAN_CUS too): the launch will share the device
    // with another stream's — a property of THIS call, carried by its descriptor; the library keeps no planning state
    const bool shared = (a.flags & (TLXMI_PLAN_SHARED_HALF | TLXMI_PLAN_SHARED_FULL)) != 0;
    const int cus = tune_int("TLXMI_PLAN_CUS", 0) > 0 ? (int)tune_int("TLXMI_PLAN_CUS", 0)
                    : (a.flags & TLXMI_PLAN_SHARED_HALF) ? (num_cus() / 2 > 0 ? num_cus() / 2 : 1) : num_cus();
    const long full_rounds = t256 / cus;
    // TLXMI_TAIL (A/B): 0 no split, 1 small tiles only (4 * left <= CUs), default: 128 x 256 tiles (2 * left <= CUs)
    // (a shared device, TLXMI_PLAN_SHARED_*: no tail splits — the other stream's launches fill a short last round, the extra
    //  launches only cost: ResNet-50 batch 256 in two halves 3.70 -> 3.59 ms)
    const int tail_mode = (int)tune_int("TLXMI_TAIL", shared ? 0 : 2);
    const bool tail_split = gemm256_ok && allow_split && tail_mode != 0 && full_rounds >= 1 && (t256 % cus) != 0 &&
                            (tail_mode == 1 ? 4 : 2) * (t256 % cus) <= cus;
    int best = 0;
    float best_score = -1.f;
    for (int i = 0; i < NC; ++i) {
        const int bm = cands[i].bm, bn = cands[i].bn;
        if (cands[i].eff <= 0.f) continue;
        if (bn == 128 && a.Cout <= 64) continue;
        size_t lds = (i == 5 || i >= 7) ? (size_t)((i == 9 || i == 10) ? 144 : 128) * 1024 : i == 6 ? (size_t)72 * 1024 : (size_t)(a.ktiles > 1 ? (i == 4 ? 3 : 2) : 1) * (bm + bn) * 128;
        if (i < 5 && lds < (size_t)bm * bn * 4) lds = (size_t)bm * bn * 4;   // fp32 epilogue tile (gemm256 stores from registers)
        lds += 2 * bn * sizeof(float);                                        // scale / shift table
        int per_cu = (int)((160 * 1024) / lds);
        if (per_cu < 1) continue;
        const int reg_cap = i == 6 ? 2 : i >= 4 ? 1 : (bm == 128 && bn == 128) ? 3 : ((bm == 64 && bn == 64) ? 8 : 5);
        if (per_cu > reg_cap) per_cu = reg_cap;
        const long blocks = (long)((a.M + bm - 1) / bm) * ((a.Cout + bn - 1) / bn) * a.nchunk;
        const long slots = (long)cus * per_cu;
        const long rounds = (blocks + slots - 1) / slots;
        float quant = (float)blocks / (float)(rounds * slots);
        if (i == 7 && tail_split) quant = (float)blocks / (((float)full_rounds + 0.6f) * slots);   // the tail round: ~0.6 of a tile time
        // the persistent kernel balances its own tail (gemm_stream.hip: a last round at most half full, or fewer tiles than half
        // the CUs, runs as half-height tiles at ~0.7 of a tile time), shared device or not
        if (i == 8 && blocks % slots != 0 && 2 * (blocks % slots) <= slots) quant = (float)blocks / (((float)(blocks / slots) + 0.7f) * slots);
        // a two-stream forward planned for the whole device: the CUs a persistent launch leaves idle serve the other stream, and
        // on long K loops the K-tile stream beats half-height tiles even when it fills under half a round — Swin-B stage 3 at
        // half batch 64, qkv 12544 x 512 -> 1536 and fc2 2048 -> 512: forward 7.88 -> 7.67 ms (tools/tile_search.py on the graph
        // replay; the 512 -> 512 proj and every other layer shape measured no gain and keep their price)
        if (i == 8 && (a.flags & TLXMI_PLAN_SHARED_FULL) && a.ktiles >= 8 && (a.Cout >= 1024 || a.ktiles >= 32) && quant < 0.7f) quant = 0.7f;
        // 3x3 convs on the antiphase kernel: a short last round is cut off along the image axis (below)
        if ((i == 7 || i == 9 || i == 10) && as_conv && allow_split && tail_mode != 0 && blocks / slots >= 1 &&
            blocks % slots != 0 && 4 * (blocks % slots) <= slots)
            quant = (float)blocks / (((float)(blocks / slots) + 0.4f) * slots);
        // wasted work inside partial tiles
        const float fill = ((float)a.M * a.Cout * a.nchunk) / ((float)blocks * bm * bn);
        const float score = quant * fill * cands[i].eff;
        if (score > best_score) { best_score = score; best = i; }
    }
    // tuning / test aid: TLXMI_TILE=<candidate> forces a tile shape (read on every call, so one process can
    // compare candidates: tools/ab_tiles.py, tests/test_gemm_gpu.py)
    if (a.pp_slices > 1) {          // K slices: only the gemm_pp convolution candidates take them (7, 9: >= 256 channels out; 10: 128)
        if (!(as_conv && (pp_conv_ok || pp_conv128_ok))) return fail(TLXMI_ERR_UNSUPPORTED, "conv2d_splitk: this layer has no gemm_pp convolution path");
        best = -1;
        best_score = -1.f;
        for (int i : {7, 9, 10}) {
            if ((i == 10 ? !pp_conv128_ok : !pp_conv_ok)) continue;
            const long blocks = (long)((a.M + cands[i].bm - 1) / cands[i].bm) * ((a.Cout + cands[i].bn - 1) / cands[i].bn) * a.pp_slices;
            const long rounds = (blocks + cus - 1) / cus;
            const float fill = ((float)a.M * a.Cout) / ((float)(blocks / a.pp_slices) * cands[i].bm * cands[i].bn);
            const float score = (float)blocks / (float)(rounds * cus) * fill * (i == 7 ? 1.40f : 1.25f);
            if (score > best_score) { best_score = score; best = i; }
        }
    }
    const int forced = a.pp_slices > 1 ? -1 : (int)tune_int("TLXMI_TILE", -1);
    if (forced >= 0 && forced < NC && !(cands[forced].bn == 128 && a.Cout <= 64) &&
        (forced < 5 || (forced <= 9 && gemm256_ok) || ((forced == 7 || forced == 9) && pp_conv_ok) ||
         (forced == 10 && (pp_conv128_ok || gemm128_ok)) || (forced == 11 && gemm256_ok))) best = forced;
#ifdef TLXMI_TUNING
    // TLXMI_FORCE="M:K:N:R:s=cand,...": force a candidate for one layer shape (tools/tile_search.py)
    if (const char* fs = getenv("TLXMI_FORCE")) {
        char key[96];
        snprintf(key, sizeof key, "%d:%d:%d:%d:%d=", a.M, a.C * a.R * a.S, a.Cout, a.R, a.sh);
        const char* hit = strstr(fs, key);
        if (hit && (hit == fs || hit[-1] == ',')) {
            const int f = atoi(hit + strlen(key));
            if (f >= 0 && f < NC && !(cands[f].bn == 128 && a.Cout <= 64) && (f != 4 || a.ktiles >= 4) &&
                (f < 5 || (f <= 9 && gemm256_ok) || ((f == 7 || f == 9) && pp_conv_ok) || (f == 10 && (pp_conv128_ok || gemm128_ok)) || (f == 11 && gemm256_ok))) best = f;
        }
    }
    if (tune_int("TLXMI_TRACE_TILES", 0))
        fprintf(stderr, "tile M=%d K=%d N=%d R=%d s=%d res=%d plan_cus=%d -> cand %d (%dx%d)\n", a.M, a.C * a.R * a.S, a.Cout, a.R, a.sh, a.res ? 1 : 0, cus, best,
                cands[best].bm, cands[best].bn);
#endif
    if ((best == 7 || best == 9 || best == 10) && as_conv && allow_split && tail_mode != 0 && a.pp_slices <= 1) {
        // Image-axis tail split: one workgroup per CU, so a last round with few tiles costs a whole tile time.  The
        // images whose rows fill the whole rounds stay on this kernel; the last few images are a convolution of their
        // own on the small tiles (28 x 28 stage of ResNet-50 at batch 256: 784 tiles = 3.06 rounds -> 250 + 6 images).
        const int bm = cands[best].bm, bn = cands[best].bn;
        const long ntn = (a.Cout + bn - 1) / bn;
        const long tiles = (long)((a.M + bm - 1) / bm) * ntn;
        const long full = tiles / cus, left = tiles % cus;
        if (full >= 1 && left != 0 && 4 * left <= cus) {
            const int n1 = (int)(((full * cus) / ntn) * bm / a.HoWo);
            if (n1 >= 1 && n1 < a.N) {
                ConvArgs lo = a, hi = a;
                lo.N = n1; lo.M = n1 * a.HoWo;
                hi.N = a.N - n1; hi.M = hi.N * a.HoWo;
                const size_t xoff = (size_t)n1 * a.H * a.W * a.x_ld * sizeof(T), yoff = (size_t)n1 * a.HoWo * a.y_ld * sizeof(T);
                hi.x = a.x + xoff;
                hi.y = a.y + yoff;
                if (a.res) hi.res = a.res + (size_t)n1 * a.HoWo * a.res_ld * sizeof(T);
                lo.x_bytes = (unsigned)xoff; hi.x_bytes = a.x_bytes - lo.x_bytes;
                lo.y_bytes = (unsigned)yoff; hi.y_bytes = a.y_bytes - lo.y_bytes;
                const int rc = dispatch<T>(lo, st, allow_stream, false);
                if (rc != TLXMI_OK) return rc;
                return dispatch<T>(hi, st, false, false);
            }
        }
    }
    if (best == 7 && tail_split) {
        // rows of the full rounds (whole M tiles) -> this candidate; the rest -> best small-tile candidate
        const int nt = (a.Cout + 255) / 256;
        const int m_split = (int)((full_rounds * cus) / nt) * 256;
        if (m_split > 0 && m_split < a.M) {
            ConvArgs lo = a, hi = a;
            lo.N = 1; lo.H = lo.Ho = 1; lo.W = lo.Wo = m_split; lo.HoWo = lo.M = m_split;
            hi.N = 1; hi.H = hi.Ho = 1; hi.W = hi.Wo = a.M - m_split; hi.HoWo = hi.M = a.M - m_split;
            hi.x = a.x + (size_t)m_split * a.x_ld * sizeof(T);
            hi.y = a.y + (size_t)m_split * a.y_ld * sizeof(T);
            if (a.res) hi.res = a.res + (size_t)m_split * a.res_ld * sizeof(T);
            lo.x_bytes = (unsigned)((size_t)m_split * a.x_ld * sizeof(T));
            hi.x_bytes = a.x_bytes - lo.x_bytes;
            lo.y_bytes = (unsigned)((size_t)m_split * a.y_ld * sizeof(T));
            hi.y_bytes = a.y_bytes - lo.y_bytes;
            lo.y_nstride = hi.y_nstride = 0; lo.res_nstride = hi.res_nstride = 0;
            const int rc = dispatch<T>(lo, st, true);     // t256 % cus == 0 there: no further split
            if (rc != TLXMI_OK) return rc;
            // a short tail (one round of 128x128 tiles) runs best on the small-tile kernels (measured: ViT fc1,
            // Swin stage-3 fc1); a longer one on the 128 x 256 antiphase kernel (ViT proj / fc2: -3 %)
            if (tail_mode == 1 || 4 * (t256 % cus) <= cus) return dispatch<T>(hi, st, false);
            // the rows left over: the same antiphase kernel on 128 x 256 tiles — twice the tiles, half the time each,
            // still one round (2 * left <= CUs)
            Gemm256Args g;
            g.debug = 0;
            g.conv = 0;
            g.x = hi.x; g.w = hi.w; g.y = hi.y; g.scale = hi.scale; g.shift = hi.shift; g.res = hi.res;
            g.M = hi.M; g.Cout = hi.Cout; g.x_ld = hi.x_ld; g.y_ld = hi.y_ld; g.res_ld = hi.res_ld;
            g.kchunks = hi.kchunks; g.ksteps = hi.Kp_bytes / 128; g.Kp_bytes = hi.Kp_bytes;
            g.act = hi.act; g.act_param = hi.act_param; g.flags = hi.flags; g.mtiles = g.ntiles = 0; g.gn = 1;
            g.x_bytes = hi.x_bytes; g.w_bytes = hi.w_bytes; g.y_bytes = hi.y_bytes;
            g.res_bytes = hi.res ? (unsigned)((long long)hi.M * hi.res_ld * (long long)sizeof(T)) : 0u;
            return launch_gemm_pp128(sizeof(T) == 2 ? TLXMI_F16 : TLXMI_F32, g, st);
        }
    }
    if (best >= 5) {
        Gemm256Args g;
        g.debug = 0;
        g.conv = 0;
        if (as_conv) {      // candidates 7 / 9 / 10 as a convolution
            g.conv = 1;
            g.cH = a.H; g.cW = a.W; g.cWo = a.Wo; g.cHoWo = a.HoWo; g.csh = a.sh; g.csw = a.sw; g.cph = a.ph; g.cpw = a.pw;
            g.ctaps = a.R * a.S;
            g.ctshift = 0;
            while ((1 << g.ctshift) < tpk) ++g.ctshift;
        }
        g.x = a.x; g.w = a.w; g.y = a.y; g.scale = a.scale; g.shift = a.shift; g.res = a.res;
        g.M = a.M; g.Cout = a.Cout; g.x_ld = a.x_ld; g.y_ld = a.y_ld; g.res_ld = a.res_ld;
        g.kchunks = a.kchunks; g.ksteps = a.Kp_bytes / 64; g.Kp_bytes = a.Kp_bytes;
        g.act = a.act; g.act_param = a.act_param; g.flags = a.flags; g.mtiles = g.ntiles = 0;
        g.x_bytes = a.x_bytes; g.w_bytes = a.w_bytes; g.y_bytes = a.y_bytes;
        g.res_bytes = a.res ? (unsigned)((long long)a.M * a.res_ld * (long long)sizeof(T)) : 0u;
        if (best == 11) {      // gemm_w4.hip is part of the tuning flavour only (make tune): the product never dispatches it (eff = 0)
#ifdef TLXMI_TUNING
            g.ksteps = a.Kp_bytes / 128;
            if (gemm_w4_ok(sizeof(T) == 2 ? TLXMI_F16 : TLXMI_F32, g))
                return launch_gemm_w4(sizeof(T) == 2 ? TLXMI_F16 : TLXMI_F32, g, st, cus);
#endif
            best = 8;
        }
        if (best == 8) {
            g.ksteps = a.Kp_bytes / 128;
            if (gemm_stream_ok(sizeof(T) == 2 ? TLXMI_F16 : TLXMI_F32, g))
                return launch_gemm_stream(sizeof(T) == 2 ? TLXMI_F16 : TLXMI_F32, g, st, cus);
            best = 7;
        }
        if (best == 7 || best == 9 || best == 10) {
            g.ksteps = a.Kp_bytes / 128;
            if (a.pp_slices > 1) {
                g.kslices = a.pp_slices;
                g.kt_slice = (g.ksteps + a.pp_slices - 1) / a.pp_slices;
                g.slice_bytes = (long long)a.M * a.y_ld * 4;
                g.res = nullptr; g.scale = g.shift = nullptr;
            }
            const int dt = sizeof(T) == 2 ? TLXMI_F16 : TLXMI_F32;
            return best == 7 ? launch_gemm_pp(dt, g, st) : best == 9 ? launch_gemm_pp128(dt, g, st) : launch_gemm_pp_n128(dt, g, st);
        }
        return launch_gemm256(sizeof(T) == 2 ? TLXMI_F16 : TLXMI_F32, best - 5, g, st);
    }
    switch (best) {
        case 0: return launch<T, 128, 128, 2, 2>(a, st);
        case 1: return launch<T, 64, 128, 2, 2>(a, st);
        case 2: return launch<T, 128, 64, 2, 2>(a, st);
        case 4: return launch<T, 256, 128, 4, 3>(a, st);
        default: return launch<T, 64, 64, 2, 2>(a, st);
    }
}

}  // namespace tlxmi

using namespace tlxmi;

// nchunk launch chunks of (C / nchunk) -> (Cout / nchunk) channels each; nchunk == 1: the dense convolution
// diag32: the filter of every chunk is block-diagonal at a granularity that divides 32 channels (in == out per group)
// ksplit: the nchunk launch chunks are K slices of ONE dense GEMM (tlxmi_linear_splitk): chunk g reads input channels
// [g*C/nchunk, (g+1)*C/nchunk) of x and of every packed filter row and writes its partial sums to y + g * ksplit_ystride.
static int conv2d_impl(const tlxmi_conv2d_desc* d, int nchunk, const void* x, const void* w_packed,
                       const float* scale, const float* shift, const void* res, void* y, void* stream, bool diag32 = false,
                       bool ksplit = false, long long ksplit_ystride = 0, int pp_slices = 1) {
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
    // Ho / Wo below the full-correlation extent crop; above it, the last windows run past the bottom / right edge and read
    // zeros there (one-sided end padding: TensorFlow-style 'SAME' at stride 2, efficientnet.py:92-125) — allowed as long
    // as every window starts inside the image or its leading padding.
    const int Ho_max = (d->H - 1 + d->pad_h) / d->stride_h + 1, Wo_max = (d->W - 1 + d->pad_w) / d->stride_w + 1;
    TLXMI_REQUIRE(d->Ho > 0 && d->Wo > 0 && d->Ho <= Ho_max && d->Wo <= Wo_max, TLXMI_ERR_BAD_ARG,
                  "conv2d: output extent %dx%d exceeds %dx%d (windows must start inside the padded image)", d->Ho, d->Wo, Ho_max, Wo_max);
    const bool overhang = d->Ho > Ho || d->Wo > Wo;
    TLXMI_REQUIRE(d->act >= TLXMI_ACT_NONE && d->act <= TLXMI_ACT_SILU, TLXMI_ERR_BAD_ARG, "conv2d: bad act %d", d->act);
    TLXMI_REQUIRE(!res || d->res_ld >= d->Cout, TLXMI_ERR_BAD_ARG, "conv2d: res_ld=%d < Cout", d->res_ld);
    const long long M = (long long)d->N * d->Ho * d->Wo;
    const long long x_bytes = (long long)d->N * d->H * d->W * d->x_ld * es;
    TLXMI_REQUIRE(M < (1ll << 31) && x_bytes < (1ll << 31) && M * (long long)d->y_ld * es < (1ll << 40), TLXMI_ERR_UNSUPPORTED,
                  "conv2d: input of %lld bytes exceeds the 2 GiB the 32-bit buffer offsets address", x_bytes);

    ConvArgs a;
    a.x = (const char*)x; a.w = (const char*)w_packed; a.y = (char*)y;
    a.scale = scale; a.shift = shift; a.res = (const char*)res;
    const int cw_in = d->C / nchunk, cw_out = ksplit ? d->Cout : d->Cout / nchunk;   // channels of one launch chunk
    a.N = d->N; a.H = d->H; a.W = d->W; a.C = cw_in; a.Cout = cw_out; a.R = d->R; a.S = d->S;
    a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
    a.Ho = d->Ho; a.Wo = d->Wo; a.x_ld = d->x_ld; a.y_ld = d->y_ld; a.res_ld = res ? d->res_ld : 0;
    a.act = d->act; a.act_param = d->act_param; a.flags = d->flags;
    a.M = (int)M; a.HoWo = d->Ho * d->Wo;
    a.cpt = cw_in * es / 16;
    a.kchunks = d->R * d->S * a.cpt;
    a.ktiles = (a.kchunks + 7) / 8;
    a.Kp_bytes = a.ktiles * 128;
    a.mtiles = a.ntiles = 0;
    a.gn = 1;
    a.sb_off = 0;
    a.x_bytes = (unsigned)x_bytes;
    {
        // non-temporal stores by default (measured: -3..-28 % per layer); TLXMI_STORE=0/16/2 overrides for tuning
        const int pol = (int)tune_int("TLXMI_STORE", 2);
        const long long yb = M * (long long)d->y_ld * es;
        a.y_bytes = yb < (1ll << 31) ? (unsigned)yb : 0u;
        a.store_policy = (a.y_bytes && !d->y_nstride && !ksplit) ? pol : 0;
    }
    if (ksplit) {      // the packed rows keep the pitch of the whole K; a chunk walks its slice of them
        const int kc_all = d->R * d->S * (d->C * es / 16);
        a.Kp_bytes = (kc_all + 7) / 8 * 128;
    }
    a.w_bytes = (unsigned)(((size_t)(cw_out + 127) / 128 * 128) * (size_t)a.Kp_bytes);
    a.nchunk = nchunk;
    a.gx = nchunk > 1 ? cw_in * es : 0;
    a.gy = a.gres = nchunk > 1 ? cw_out * es : 0;
    a.gc = nchunk > 1 ? cw_out : 0;
    a.gw = nchunk > 1 ? a.w_bytes : 0u;
    a.out_f32 = ksplit ? 1 : 0;
    a.pp_slices = pp_slices;
    if (ksplit) {
        a.gw = (unsigned)(cw_in * es);
        a.gy = (int)ksplit_ystride;
        a.gres = 0;
        a.gc = 0;
        a.y_bytes = (unsigned)(ksplit_ystride * nchunk);     // the whole partial buffer: chunk g sees what is left after its offset
    }
    a.diag = diag32 && es == 2 && cw_in == 64 && cw_out == 64 ? 1 : 0;
    a.overhang = overhang ? 1 : 0;
    const int vecn = 16 / es;  // elements per 16 bytes
    const bool bcast = res && (d->flags & TLXMI_EPI_RES_BCAST_N);
    TLXMI_REQUIRE(d->y_nstride >= 0 && d->res_nstride >= 0, TLXMI_ERR_BAD_ARG, "conv2d: negative batch stride");
    a.y_nstride = d->y_nstride ? d->y_nstride : (long)a.HoWo * d->y_ld;
    a.res_nstride = bcast ? 0 : (d->res_nstride ? d->res_nstride : (long)a.HoWo * a.res_ld);
    a.strided_n = (d->y_nstride != 0) || (res && (d->res_nstride != 0 || bcast));
    a.vec_io = aligned16(y) && (d->y_ld % vecn == 0) && (a.y_nstride % vecn == 0) &&
               (!res || (aligned16(res) && d->res_ld % vecn == 0 && a.res_nstride % vecn == 0));
    const int rc = d->dtype == TLXMI_F16 ? dispatch<half_t>(a, as_stream(stream)) : dispatch<float>(a, as_stream(stream));
    if (rc != TLXMI_OK) return rc;
    return check_launch("conv2d");
}

extern "C" int tlxmi_conv2d(const tlxmi_conv2d_desc* d, const void* x, const void* w_packed,
                            const float* scale, const float* shift, const void* res, void* y,
                            void* stream) {
    return conv2d_impl(d, 1, x, w_packed, scale, shift, res, y, stream);
}

// ------------------------------------------------------------------------------------------
// Convolution with few output pixels and a long K (the 3x3 convs of ResNet's 7 x 7 stage, resnet.py:111-121: 72 K tiles in
// sequence on 98 tiles of 128 x 256 — latency-bound at any batch): K is cut into `splits` slices that run side by side on
// the antiphase GEMM kernel (gemm_pp.hip, CONV mode), each storing its fp32 accumulators; splitk_reduce_kernel adds them in
// slice order and applies scale / shift / residual / activation.
// ------------------------------------------------------------------------------------------
namespace tlxmi {
template <typename T>
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int splits, long rows, int Cout, long pstride, const float* __restrict__ scale,
                                     const float* __restrict__ shift, const T* __restrict__ res, int res_ld, int act, float act_param,
                                     unsigned flags, T* __restrict__ y, int y_ld);
static bool conv_splitk_shape_ok(const tlxmi_conv2d_desc* d, int splits) {
    if (!d || (d->dtype != TLXMI_F16 && d->dtype != TLXMI_F32) || splits < 2 || splits > 16) return false;
    const int es = (int)elt_size(d->dtype);
    const bool strided1x1 = d->R == 1 && d->S == 1 && d->pad_h == 0 && d->pad_w == 0 && (d->stride_h > 1 || d->stride_w > 1);
    if (!((d->S == 3 && d->R >= 1 && d->R <= 3) || strided1x1) || d->dil_h != 1 || d->dil_w != 1) return false;
    if (d->Cout % 8 || d->Cout < 128 || d->y_nstride || d->res_nstride || (d->flags & (TLXMI_EPI_RES_BCAST_N | TLXMI_EPI_MAXPOOL_3S2P1))) return false;
    if ((d->C * es) % 128) return false;
    const int tpk = d->C * es / 128;                       // K tiles per tap: a power of two
    if (tpk & (tpk - 1)) return false;
    const int ktiles = d->R * d->S * tpk;
    if (ktiles / splits < 4) return false;                 // at least 4 K tiles per slice
    // what the dispatcher's gemm_pp convolution path (pp_conv128_ok) asks on top: no window past the bottom / right edge
    // (one-sided 'SAME' padding), whole 16-byte chunks on the output and residual rows
    const int Ho = (d->H + 2 * d->pad_h - (d->R - 1) - 1) / d->stride_h + 1, Wo = (d->W + 2 * d->pad_w - (d->S - 1) - 1) / d->stride_w + 1;
    if (d->Ho > Ho || d->Wo > Wo) return false;
    const int vecn = 16 / es;
    if (d->y_ld % vecn || (d->res_ld > 0 && d->res_ld % vecn)) return false;
    const long long M = (long long)d->N * d->Ho * d->Wo;
    return M > 0 && M * d->Cout * 4 * splits < (1ll << 31) && d->y_ld >= d->Cout && M * (long long)d->y_ld * es < (1ll << 31);
}
}  // namespace tlxmi

extern "C" int tlxmi_conv2d_splitk_supported(const tlxmi_conv2d_desc* d, int splits) { return conv_splitk_shape_ok(d, splits) ? 1 : 0; }

extern "C" int tlxmi_conv2d_splitk(const tlxmi_conv2d_desc* d, int splits, const void* x, const void* w_packed, void* partials,
                                   const float* scale, const float* shift, const void* res, void* y, void* stream) {
    TLXMI_REQUIRE(d && x && w_packed && partials && y, TLXMI_ERR_BAD_ARG, "conv2d_splitk: null descriptor or buffer");
    TLXMI_REQUIRE(conv_splitk_shape_ok(d, splits), TLXMI_ERR_UNSUPPORTED, "conv2d_splitk: layer / slice count not supported (ask tlxmi_conv2d_splitk_supported)");
    TLXMI_REQUIRE(aligned16(partials), TLXMI_ERR_ALIGNMENT, "conv2d_splitk: partials must be 16-byte aligned");
    const long long M = (long long)d->N * d->Ho * d->Wo;
    tlxmi_conv2d_desc dp = *d;
    dp.y_ld = d->Cout;             // the partial planes are dense [M][Cout] floats
    dp.act = TLXMI_ACT_NONE;
    dp.flags = d->flags & (TLXMI_PLAN_SHARED_HALF | TLXMI_PLAN_SHARED_FULL);
    dp.res_ld = 0;
    const int rc = conv2d_impl(&dp, 1, x, w_packed, nullptr, nullptr, nullptr, partials, stream, false, false, 0, splits);
    if (rc != TLXMI_OK) return rc;
    const long total = (long)M * d->Cout;
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (d->dtype == TLXMI_F16)
        hipLaunchKernelGGL((splitk_reduce_kernel<half_t>), dim3(grid), dim3(256), 0, as_stream(stream), (const float*)partials, splits, (long)M, d->Cout,
                           (long)total, scale, shift, (const half_t*)res, d->res_ld, d->act, d->act_param, d->flags, (half_t*)y, d->y_ld);
    else
        hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3(grid), dim3(256), 0, as_stream(stream), (const float*)partials, splits, (long)M, d->Cout,
                           (long)total, scale, shift, (const float*)res, d->res_ld, d->act, d->act_param, d->flags, (float*)y, d->y_ld);
    return check_launch("conv2d_splitk");
}

extern "C" int tlxmi_conv2d_maxpool_supported(const tlxmi_conv2d_desc* d) {
    if (!d || d->dtype != TLXMI_F16) return 0;
    return d->stride_h == 1 && d->stride_w == 1 && d->dil_h == 1 && d->dil_w == 1 && conv_halo_pool_ok(d->R, d->S, d->C * 2, d->Ho, d->Wo) &&
           conv_halo_pool_act_ok(d->act) && d->Cout % 8 == 0 && d->Cout <= 128 && d->y_nstride == 0 && d->y_ld % 8 == 0;
}

// ------------------------------------------------------------------------------------------
// Grouped convolution, 1 < groups < C (resnext.py:30-40 via :83-91, n_group = cardinality 32 / 64).
// m consecutive groups are merged into one launch chunk whose filter is block-diagonal ([m*cg_out][taps][m*cg_in],
// zeros off the diagonal blocks), wide enough that a filter tap is at least one 128-byte K tile and the chunk
// fills a 64-channel MFMA tile; every chunk is then the dense implicit GEMM at a channel offset (blockIdx.y).
// The zero blocks cost MFMA work (x m), not HBM bytes: the layer stays bound by its activation traffic.
// ------------------------------------------------------------------------------------------
namespace tlxmi {
static int kpad_elems_of(int Cin, int R, int S, int dtype);      // = kpad_elems (defined with the filter packing below)
static int group_chunks(int Cin, int Cout, int groups, int dtype) {
    if (groups <= 1 || Cin <= 0 || Cout <= 0 || Cin % groups || Cout % groups) return 0;
    const int es = (int)elt_size(dtype);
    const int cgi = Cin / groups, cgo = Cout / groups;
    int m_ok = 0;
    for (int m = 1; m <= groups; ++m) {
        if (groups % m) continue;
        if ((m * cgi * es) % 16 || (m * cgo * es) % 16) continue;
        m_ok = m;
        if (m * cgi * es >= 128 && m * cgo >= 64) break;
    }
    return m_ok ? groups / m_ok : 0;
}
template <typename T>
__global__ void pack_group_filter_kernel(const float* __restrict__ src, T* __restrict__ dst, int cgi, int cgo, int R, int S,
                                         int cwi, int cwo, int cop, int Kp, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        const long row = i / Kp;
        const int ol = (int)(row % cop), chunk = (int)(row / cop);
        const int cl = k % cwi, tap = k / cwi;
        float v = 0.f;
        if (ol < cwo && tap < R * S && ol / cgo == cl / cgi) {      // same group inside the chunk: a diagonal block
            const int o = chunk * cwo + ol, c = cl % cgi;
            const int r = tap / S, s = tap - r * S;
            v = src[(((long)o * cgi + c) * R + r) * S + s];
        }
        dst[i] = (T)v;
    }
}
}  // namespace tlxmi

extern "C" int tlxmi_group_conv_chunks(int Cin, int Cout, int groups, int dtype) {
    if (dtype != TLXMI_F16 && dtype != TLXMI_F32) return 0;
    return group_chunks(Cin, Cout, groups, dtype);
}

extern "C" int tlxmi_group_conv2d_small_supported(const tlxmi_conv2d_desc* d, int groups) {
    return d && gconv_small_ok(d, groups, nullptr) && group_chunks(d->C, d->Cout, groups, d->dtype) == d->C / 64 ? 1 : 0;
}

extern "C" int tlxmi_group_conv2d(const tlxmi_conv2d_desc* d, int groups, const void* x, const void* w_packed,
                                  const float* scale, const float* shift, const void* res, void* y, void* stream) {
    TLXMI_REQUIRE(d, TLXMI_ERR_BAD_ARG, "group_conv2d: null descriptor");
    TLXMI_REQUIRE(d->dtype == TLXMI_F16 || d->dtype == TLXMI_F32, TLXMI_ERR_BAD_ARG, "group_conv2d: bad dtype %d", d->dtype);
    TLXMI_REQUIRE(groups >= 1 && d->C > 0 && d->Cout > 0 && d->C % groups == 0 && d->Cout % groups == 0, TLXMI_ERR_BAD_ARG,
                  "group_conv2d: channels %d -> %d are not divisible by groups=%d", d->C, d->Cout, groups);
    if (groups == 1) return conv2d_impl(d, 1, x, w_packed, scale, shift, res, y, stream);
    const int nchunk = group_chunks(d->C, d->Cout, groups, d->dtype);
    if (nchunk == 0)
        return fail(TLXMI_ERR_UNSUPPORTED, "group_conv2d: %d -> %d channels in %d groups cannot be merged into 16-byte aligned chunks",
                    d->C, d->Cout, groups);
    TLXMI_REQUIRE(nchunk <= 65535, TLXMI_ERR_UNSUPPORTED, "group_conv2d: %d launch chunks", nchunk);
    TLXMI_REQUIRE(!res || (d->res_ld * (int)elt_size(d->dtype)) % 16 == 0, TLXMI_ERR_ALIGNMENT, "group_conv2d: res_ld=%d", d->res_ld);
    TLXMI_REQUIRE((d->y_ld * (int)elt_size(d->dtype)) % 16 == 0, TLXMI_ERR_ALIGNMENT, "group_conv2d: y_ld=%d", d->y_ld);
    const int cgi = d->C / groups, cgo = d->Cout / groups;
    // 3x3 with few channels per group: the small-block MFMA kernel (group_conv.hip) — no products with the zero blocks.  Where it
    // wins (batch 256, tools/gconv_micro.py, us, block-diagonal -> 4x4x4): 4 per group 56 x 56 183 -> 129 (64x4d: 374 -> 211),
    // 8 per group 28 x 28 89 -> 61 and stride 2 from 56 x 56 140 -> 113, 16 per group 14 x 14 51 -> 49; it loses at 16 per group
    // with stride 2 (66 -> 80) and at 32 per group (38 -> 72, 35 -> 58: two real products in four there, and 7 x 7 tiles are
    // all prologue), which stay on the block-diagonal path.  TLXMI_GCONV (tuning flavour): bit (log2(cg / 4) + 4 * (stride - 1)).
    if (gconv_small_ok(d, groups, res) && nchunk == d->C / 64 && aligned16(x) && aligned16(w_packed) && ((uintptr_t)y & 7u) == 0 &&
        aligned16(scale) && aligned16(shift)) {      // (group_conv.hip fetches scale / shift as 16-byte vectors; otherwise the general path)
        const int qi = cgi == 4 ? 0 : cgi == 8 ? 1 : cgi == 16 ? 2 : 3;
        if ((tune_int("TLXMI_GCONV", 0x37) >> (qi + 4 * (d->stride_h - 1))) & 1)
            return launch_gconv_small(d, groups, x, w_packed, scale, shift, y, kpad_elems_of(d->C / nchunk, d->R, d->S, d->dtype) * 2, as_stream(stream));
    }
    return conv2d_impl(d, nchunk, x, w_packed, scale, shift, res, y, stream, cgi == cgo && 32 % cgi == 0);
}

// ------------------------------------------------------------------------------------------
// Linear layers with few rows and a large filter (the classifier heads: resnet.py:234-237, vgg.py:42-50 25088 -> 4096,
// alexnet.py fc6-8): with M <= a few hundred rows the GEMM has only M/64 x Cout/64 tiles, each walking all of K in
// sequence — the filter streams from HBM at a fraction of the bandwidth (measured 0.74 TB/s on VGG's fc1).  Here K is
// cut into `splits` slices that run as the launch chunks of the implicit GEMM (blockIdx.y; `splits` x more tiles in
// flight), each writing its partial sums, and a small second kernel adds them in slice order (deterministic), applies
// scale / shift / residual / activation and stores.
// ------------------------------------------------------------------------------------------
namespace tlxmi {
template <typename T>
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int splits, long rows, int Cout, long pstride, const float* __restrict__ scale,
                                     const float* __restrict__ shift, const T* __restrict__ res, int res_ld, int act, float act_param,
                                     unsigned flags, T* __restrict__ y, int y_ld) {
    const long total = rows * Cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / Cout;
        const int n = (int)(i - m * Cout);
        float v = 0.f;
        for (int g = 0; g < splits; ++g) v += part[g * pstride + i];      // fp32 partial sums, slice order
        if (scale) v *= scale[n];
        if (shift) v += shift[n];
        const bool res_after = (flags & TLXMI_EPI_RES_AFTER_ACT) != 0;
        const float r = res ? (float)res[m * res_ld + n] : 0.f;
        if (res && !res_after) v += r;
        v = apply_act(v, act, act_param);
        if (res && res_after) v += r;
        y[m * y_ld + n] = (T)v;
    }
}
}  // namespace tlxmi

// ------------------------------------------------------------------------------------------
// LayerNorm folded AROUND the Linear layers of a transformer block (vision_transformer.py:144-175 norm1 -> attn.qkv, norm2 -> mlp.fc1;
// swin_transformer.py:310-337), fp16, on the persistent 256 x 256 GEMM kernel (gemm_stream.hip):
//   producer  tlxmi_linear_stats: y = x W^T + bias (+ res), and from the same epilogue per row (sum y, sum y^2) over every 256-channel
//             tile column -> partials[rows][4][2], pair p < ceil(Cout / 256) written (the LayerNorm that follows needs no pass over y);
//   consumer  tlxmi_linear_ln: mean / rstd of each row from the ceil(K / 256) <= 4 pairs the producer left (no launch in between), then
//             y = act(rstd * (x W'^T) - mean * rstd * c1[n] + c2[n]) on the RAW rows x, with the caller's W' = W * gamma (packed),
//             c1[n] = sum_k W'[n][k] (of the values as packed), c2[n] = bias[n] + sum_k W[n][k] * beta[k].
// The normalised activations are never written or read.  Shapes / options the persistent kernel does not take return
// TLXMI_ERR_UNSUPPORTED (tlxmi_linear_ln_supported asks first) and the caller keeps tlxmi_layernorm + tlxmi_conv2d.
// ------------------------------------------------------------------------------------------
namespace tlxmi {
static int fill_ln_gemm(Gemm256Args& g, const char* who, int dtype, int64_t rows, int K, int Cout, int x_ld, int y_ld, const void* x,
                        const void* w_packed, void* y) {
    TLXMI_REQUIRE(x && w_packed && y, TLXMI_ERR_BAD_ARG, "%s: null buffer", who);
    TLXMI_REQUIRE(dtype == TLXMI_F16 || dtype == TLXMI_F32, TLXMI_ERR_BAD_ARG, "%s: bad dtype %d", who, dtype);
    TLXMI_REQUIRE(rows > 0 && K > 0 && Cout > 0 && x_ld >= K && y_ld >= Cout, TLXMI_ERR_BAD_ARG, "%s: bad extent", who);
    if (dtype != TLXMI_F16) return fail(TLXMI_ERR_UNSUPPORTED, "%s: fp16 only (fp32 runs tlxmi_layernorm + tlxmi_conv2d)", who);
    TLXMI_REQUIRE((K * 2) % 16 == 0 && (x_ld * 2) % 16 == 0 && (y_ld * 2) % 16 == 0 && aligned16(x) && aligned16(w_packed) && aligned16(y),
                  TLXMI_ERR_ALIGNMENT, "%s: rows must be whole 16-byte chunks", who);
    const long long xb = (long long)rows * x_ld * 2, yb = (long long)rows * y_ld * 2;
    const int kchunks = K * 2 / 16, ktiles = (kchunks + 7) / 8;
    if (!(Cout % 32 == 0 && Cout >= 256 && ktiles >= 2 && xb < (1ll << 31) && yb < (1ll << 31) && rows < (1ll << 27)))
        return fail(TLXMI_ERR_UNSUPPORTED, "%s: shape %lld x %d -> %d is outside the persistent 256 x 256 GEMM kernel", who, (long long)rows, K, Cout);
    g.debug = 0;
    g.conv = 0;
    g.x = (const char*)x; g.w = (const char*)w_packed; g.y = (char*)y; g.scale = nullptr; g.shift = nullptr; g.res = nullptr;
    g.M = (int)rows; g.Cout = Cout; g.x_ld = x_ld; g.y_ld = y_ld; g.res_ld = 0;
    g.kchunks = kchunks; g.Kp_bytes = ktiles * 128; g.ksteps = ktiles;
    g.act = TLXMI_ACT_NONE; g.act_param = 0.f; g.flags = 0; g.mtiles = g.ntiles = 0; g.gn = 1;
    g.x_bytes = (unsigned)xb; g.y_bytes = (unsigned)yb; g.res_bytes = 0;
    g.w_bytes = (unsigned)(((size_t)(Cout + 127) / 128 * 128) * (size_t)g.Kp_bytes);
    return TLXMI_OK;
}
static int ln_plan_cus(unsigned flags) {
    return (flags & TLXMI_PLAN_SHARED_HALF) ? (num_cus() / 2 > 0 ? num_cus() / 2 : 1) : num_cus();
}
// Whether a folded GEMM the persistent kernel could take goes to the one-tile kernel's half-height tiles instead: fewer 256 x 256 tiles
// than half the CUs (Swin-B stage 3 fc2: 49 x 2; stage 4 proj / fc2: 13 x 4; ViT-B/16 at batch 64: 25 x 3).  Measured on the graph replay:
// Swin-B batch 128 -0.4 %, ViT-B/16 batch 64 -3.5 % (every tile of the persistent kernel half-height instead, TLXMI_HALFTAIL=2: +3.5 %).
// TLXMI_LN_SMALL_PP (tuning flavour): 0 = off.
static bool ln_small_on_pp(const Gemm256Args& g) {
    const long t256 = (long)((g.M + 255) / 256) * ((g.Cout + 255) / 256);
    return tune_int("TLXMI_LN_SMALL_PP", 1) != 0 && 2 * t256 <= num_cus() && g.M > 128;
}
// The one-tile-per-workgroup form of a folded GEMM: 256 x 256 tiles, or 128 x 256 when the full tiles would leave half the CUs
// without one (then twice as many half-height tiles still fit one round).  TLXMI_LN_PP128 (tuning flavour): 0 never, 1 always.
static int launch_ln_pp(int dtype, const Gemm256Args& g, hipStream_t st) {
    const long t256 = (long)((g.M + 255) / 256) * ((g.Cout + 255) / 256);
    const long knob = tune_int("TLXMI_LN_PP128", -1);
    const bool half_height = knob >= 0 ? knob != 0 : (2 * t256 <= num_cus() && g.M > 128);
    return half_height ? launch_gemm_pp128(dtype, g, st) : launch_gemm_pp(dtype, g, st);
}
}  // namespace tlxmi

extern "C" int tlxmi_linear_ln_supported(int dtype, int64_t rows, int K, int Cout, int act, int with_res) {
    using namespace tlxmi;
    if (dtype != TLXMI_F16 || rows <= 0 || rows >= (1ll << 27) || K <= 0 || Cout < 256 || Cout % 32 || (K * 2) % 16) return 0;
    const int ktiles = (K * 2 / 16 + 7) / 8;
    if (ktiles < 2) return 0;      // (a residual with fewer than 11 K tiles runs on the one-tile-per-workgroup kernel, gemm_pp.hip LNF)
    // with_res: 0 = the consumer (tlxmi_linear_ln), 1 = a producer with a residual, 2 = a producer without (tlxmi_linear_stats)
    if (act != TLXMI_ACT_NONE && !(act == TLXMI_ACT_GELU && !with_res)) return 0;
    if (with_res ? Cout > 1024 : K > 1024) return 0;      // the statistics of a row are 4 pairs, one per 256 channels of the LayerNorm's width
    return 1;
}

extern "C" int tlxmi_linear_stats(int dtype, int64_t rows, int K, int Cout, int x_ld, int y_ld, const void* x, const void* w_packed,
                                  const float* bias, const void* res, int res_ld, void* y, float* partials, unsigned flags, void* stream) {
    using namespace tlxmi;
    Gemm256Args g;
    if (int rc = fill_ln_gemm(g, "linear_stats", dtype, rows, K, Cout, x_ld, y_ld, x, w_packed, y)) return rc;
    TLXMI_REQUIRE(partials && ((uintptr_t)partials & 15) == 0, TLXMI_ERR_BAD_ARG, "linear_stats: partials must be a 16-byte aligned buffer");
    TLXMI_REQUIRE(!res || (res_ld >= Cout && (res_ld * 2) % 16 == 0 && aligned16(res)), TLXMI_ERR_BAD_ARG, "linear_stats: bad residual");
    if (Cout > 1024) return fail(TLXMI_ERR_UNSUPPORTED, "linear_stats: %d output channels (a row's statistics are 4 pairs, one per 256 channels)", Cout);
    if (rows * 32 >= (1ll << 31)) return fail(TLXMI_ERR_UNSUPPORTED, "linear_stats: statistics exceed 2 GiB");
    g.shift = bias;
    g.res = (const char*)res;
    g.res_ld = res ? res_ld : 0;
    g.res_bytes = res ? (unsigned)((long long)rows * res_ld * 2) : 0u;
    if (res && (long long)rows * res_ld * 2 >= (1ll << 31)) return fail(TLXMI_ERR_UNSUPPORTED, "linear_stats: residual exceeds 2 GiB");
    g.stats_out = partials;
    g.flags = flags & (TLXMI_PLAN_SHARED_HALF | TLXMI_PLAN_SHARED_FULL);
    // the persistent kernel where it applies; a residual with a short K (its residual steps need 11 K tiles: Swin-B stage 3 proj,
    // K = 512) and launches of few tiles (ln_small_on_pp) on the one-tile-per-workgroup form of the same K loop
    if (int rc = (gemm_stream_ok(dtype, g) && !ln_small_on_pp(g)) ? launch_gemm_stream(dtype, g, as_stream(stream), ln_plan_cus(flags)) : launch_ln_pp(dtype, g, as_stream(stream))) return rc;
    return check_launch("linear_stats");
}

extern "C" int tlxmi_linear_ln(int dtype, int64_t rows, int K, int Cout, int x_ld, int y_ld, const void* x, const void* w_packed,
                               const float* c1, const float* c2, const float* partials, float eps, int act, void* y, unsigned flags,
                               void* stream) {
    using namespace tlxmi;
    Gemm256Args g;
    if (int rc = fill_ln_gemm(g, "linear_ln", dtype, rows, K, Cout, x_ld, y_ld, x, w_packed, y)) return rc;
    TLXMI_REQUIRE(c1 && c2 && partials && ((uintptr_t)partials & 15) == 0, TLXMI_ERR_BAD_ARG, "linear_ln: c1 / c2 / partials (16-byte aligned) are required");
    TLXMI_REQUIRE(eps >= 0.f, TLXMI_ERR_BAD_ARG, "linear_ln: eps %g", (double)eps);
    TLXMI_REQUIRE(act == TLXMI_ACT_NONE || act == TLXMI_ACT_GELU, TLXMI_ERR_UNSUPPORTED, "linear_ln: activation %d (none or GELU)", act);
    if (K > 1024) return fail(TLXMI_ERR_UNSUPPORTED, "linear_ln: rows of %d channels (the statistics of a row are held as <= 4 planes of 256)", K);
    g.scale = c1;
    g.shift = c2;
    g.rowstats = partials;
    g.ln_planes = (K + 255) / 256;
    g.ln_inv_c = 1.0f / (float)K;
    g.ln_eps = eps;
    g.act = act;
    g.flags = flags & (TLXMI_PLAN_SHARED_HALF | TLXMI_PLAN_SHARED_FULL);
    // TLXMI_LN_GELU_PP (tuning flavour): 1 = GELU layers on the one-tile-per-workgroup kernel (its epilogue is not squeezed between MFMAs)
    const bool on_pp = !gemm_stream_ok(dtype, g) || ln_small_on_pp(g) || (act == TLXMI_ACT_GELU && tune_int("TLXMI_LN_GELU_PP", 0));
    if (int rc = on_pp ? launch_ln_pp(dtype, g, as_stream(stream)) : launch_gemm_stream(dtype, g, as_stream(stream), ln_plan_cus(flags))) return rc;
    return check_launch("linear_ln");
}

extern "C" int tlxmi_linear_splitk(int dtype, int64_t rows, int K, int Cout, int x_ld, const void* x, const void* w_packed,
                                   int splits, void* partials, const float* scale, const float* shift, const void* res,
                                   int res_ld, int act, float act_param, uint32_t flags, void* y, int y_ld, void* stream) {
    TLXMI_REQUIRE(x && w_packed && y && partials, TLXMI_ERR_BAD_ARG, "linear_splitk: null buffer");
    TLXMI_REQUIRE(dtype == TLXMI_F16 || dtype == TLXMI_F32, TLXMI_ERR_BAD_ARG, "linear_splitk: bad dtype %d", dtype);
    TLXMI_REQUIRE(rows > 0 && rows < (1ll << 24) && K > 0 && Cout > 0 && x_ld >= K && y_ld >= Cout && splits >= 2 && splits <= 256, TLXMI_ERR_BAD_ARG,
                  "linear_splitk: bad extent");
    TLXMI_REQUIRE(act >= TLXMI_ACT_NONE && act <= TLXMI_ACT_SILU, TLXMI_ERR_BAD_ARG, "linear_splitk: bad act %d", act);
    TLXMI_REQUIRE(!res || res_ld >= Cout, TLXMI_ERR_BAD_ARG, "linear_splitk: res_ld=%d < Cout", res_ld);
    const int es = (int)elt_size(dtype);
    TLXMI_REQUIRE(K % splits == 0 && ((K / splits) * es) % 128 == 0, TLXMI_ERR_ALIGNMENT,
                  "linear_splitk: K=%d must split into %d slices of whole 128-byte K tiles", K, splits);
    const long long pstride = (long long)rows * Cout;          // elements between the partial sums of consecutive slices
    TLXMI_REQUIRE(pstride * 4 * splits < (1ll << 31) && aligned16(partials) && (Cout * es) % 16 == 0, TLXMI_ERR_UNSUPPORTED,
                  "linear_splitk: partial buffer of %lld bytes per slice", pstride * 4);
    tlxmi_conv2d_desc d;
    memset(&d, 0, sizeof d);
    d.dtype = dtype; d.N = (int)rows; d.H = d.W = 1; d.C = K; d.Cout = Cout; d.R = d.S = 1;
    d.stride_h = d.stride_w = d.dil_h = d.dil_w = 1; d.Ho = d.Wo = 1;
    d.x_ld = x_ld; d.y_ld = Cout; d.act = TLXMI_ACT_NONE;
    const int rc = conv2d_impl(&d, splits, x, w_packed, nullptr, nullptr, nullptr, partials, stream, false, true, pstride * 4);
    if (rc != TLXMI_OK) return rc;
    const long total = (long)rows * Cout;
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == TLXMI_F16)
        hipLaunchKernelGGL((splitk_reduce_kernel<half_t>), dim3(grid), dim3(256), 0, as_stream(stream), (const float*)partials, splits, (long)rows, Cout,
                           (long)pstride, scale, shift, (const half_t*)res, res_ld, act, act_param, flags, (half_t*)y, y_ld);
    else
        hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3(grid), dim3(256), 0, as_stream(stream), (const float*)partials, splits, (long)rows, Cout,
                           (long)pstride, scale, shift, (const float*)res, res_ld, act, act_param, flags, (float*)y, y_ld);
    return check_launch("linear_splitk");
}

// ------------------------------------------------------------------------------------------
// Filter packing: OIHW fp32 -> [Cout_pad][Kpad] (K = (r*S+s)*Cin_pad + c), zero padded.
// ------------------------------------------------------------------------------------------
namespace tlxmi {
static inline int cin_pad(int Cin, int dtype) { const int v = 16 / (int)elt_size(dtype); return (Cin + v - 1) / v * v; }
static inline int kpad_elems(int Cin, int R, int S, int dtype);
static int kpad_elems_of(int Cin, int R, int S, int dtype) { return kpad_elems(Cin, R, S, dtype); }
static inline int kpad_elems(int Cin, int R, int S, int dtype) {
    const int es = (int)elt_size(dtype);
    const int kbytes = R * S * cin_pad(Cin, dtype) * es;
    return ((kbytes + 127) / 128 * 128) / es;
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

extern "C" size_t tlxmi_packed_group_filter_bytes(int Cout, int Cin, int R, int S, int groups, int dtype) {
    if (dtype != TLXMI_F16 && dtype != TLXMI_F32) return 0;
    if (groups == 1) return tlxmi_packed_filter_bytes(Cout, Cin, R, S, dtype);
    const int nchunk = group_chunks(Cin, Cout, groups, dtype);
    if (nchunk == 0 || R <= 0 || S <= 0) return 0;
    return (size_t)nchunk * tlxmi_packed_filter_bytes(Cout / nchunk, Cin / nchunk, R, S, dtype);
}

extern "C" int tlxmi_pack_group_filter(const float* src, void* dst, int Cout, int Cin, int R, int S, int groups, int dtype,
                                       void* stream) {
    if (groups == 1) return tlxmi_pack_filter(src, dst, Cout, Cin, R, S, dtype, stream);
    TLXMI_REQUIRE(src && dst, TLXMI_ERR_BAD_ARG, "pack_group_filter: null buffer");
    TLXMI_REQUIRE(dtype == TLXMI_F16 || dtype == TLXMI_F32, TLXMI_ERR_BAD_ARG, "pack_group_filter: bad dtype");
    TLXMI_REQUIRE(Cout > 0 && Cin > 0 && R > 0 && S > 0 && groups > 0, TLXMI_ERR_BAD_ARG, "pack_group_filter: non-positive extent");
    TLXMI_REQUIRE(aligned16(dst), TLXMI_ERR_ALIGNMENT, "pack_group_filter: dst must be 16-byte aligned");
    const int nchunk = group_chunks(Cin, Cout, groups, dtype);
    if (nchunk == 0)
        return fail(TLXMI_ERR_UNSUPPORTED, "pack_group_filter: %d -> %d channels in %d groups cannot be merged into 16-byte aligned chunks",
                    Cin, Cout, groups);
    const int cwi = Cin / nchunk, cwo = Cout / nchunk;
    const int Kp = kpad_elems(cwi, R, S, dtype);
    const int cop = (cwo + 127) / 128 * 128;
    const long total = (long)nchunk * cop * Kp;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == TLXMI_F16)
        hipLaunchKernelGGL((pack_group_filter_kernel<half_t>), dim3(grid), dim3(256), 0, as_stream(stream), src, (half_t*)dst,
                           Cin / groups, Cout / groups, R, S, cwi, cwo, cop, Kp, total);
    else
        hipLaunchKernelGGL((pack_group_filter_kernel<float>), dim3(grid), dim3(256), 0, as_stream(stream), src, (float*)dst,
                           Cin / groups, Cout / groups, R, S, cwi, cwo, cop, Kp, total);
    return check_launch("pack_group_filter");
}

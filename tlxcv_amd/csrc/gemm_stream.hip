// Persistent 256 x 256 tile GEMM: gemm_pp.hip's two-group antiphase K loop run as ONE continuous stream
// of K tiles over all the output tiles of a workgroup (one workgroup per CU), for the Linear layers of
// ViT / Swin (reference vision_transformer.py:81-87,112-123; swin_transformer.py:202-226,258-300):
//     Y[m][n] = act( (sum_k X[m][k] * Wp[n][k]) * scale[n] + shift[n] + R[m][n] )
//
// Why: with K = 768 an output tile is only 12 K tiles (~20 us of MFMA work) and the per-tile costs of a
// one-tile-per-workgroup launch — store drain before the workgroup may retire, workgroup launch, first DMA
// latency, epilogue — were 35-40 % of the time (measured: tools/conv_micro.py qkv, TLXMI_DEBUG=1).  Here
//   * the LDS-DMA stream never stops: the loads of the next output tile's first two K tiles are issued
//     during the last two K tiles of the current one (cursor A = stream position + 1 for the W1/X1 half
//     tiles, cursor B = position + 2 for X0/W0, each with its own row offsets);
//   * the epilogue is cut into the four quadrants of the wave tile and each quarter runs in the load
//     segment right after the phase that finished it (E00 in p1 of the last K tile, E01 in p2, E11 in p3,
//     E10 in p0 of the next tile's first K tile), i.e. under the other wave group's MFMAs;
//   * nothing waits for a store: gfx950 has one in-order counter for loads, stores and LDS-DMA, so each
//     counted wait allows for the stores issued after its target (vmcnt up to 8 + 4*S + 2), and every
//     store batch has >= 4 phases before a wait depends on it;
//   * the residual is not an epilogue input: it is added into the accumulators in the middle of the tile
//     (half a quadrant per K tile, K tiles 1..9), loads issued one K tile ahead — a load in the epilogue
//     would have to wait for the stores before it (same counter);
//   * scale / shift of the next tile arrive by LDS-DMA in a double-buffered table (a null array reads a
//     constant block), accumulators are not zeroed (the first MFMA of a tile takes C = 0).
// Layout of a K tile in LDS, wave->quadrant map, fragment reads, channel permutation: gemm_pp.hip.
//
// Balanced tail (round 4).  One workgroup per CU: when the tiles are r whole rounds plus a last round that is at most half
// full, that round costs a whole tile time for a fraction of the CUs (ViT-B/16 at batch 256: proj / fc2 are 2.32 rounds paid
// as 3; measured on the forward: batch 220, whole rounds, runs 5.7 % more images per second than batch 256).  The rows of
// the short round are then cut into HALF-HEIGHT tiles (128 rows) at the end of every workgroup's tile list: twice as many
// workgroups get one, and such a tile runs the same K-tile stream with its H = 1 phases idle — no X1 traffic (out-of-range
// DMA offsets), no MFMAs, the (1, *) epilogues reduced to their dropped stores — so every vector-memory count and every
// barrier stays where it is.  No partial sums, no cross-workgroup exchange: results are bit-identical to the full tiles'.
//
// In-order VMEM sequence of one wave around a tile boundary (L = last K tile of a tile; D = 2 DMA pieces,
// S = the stores of a quadrant (32 outputs per lane: 4 x 16 B for fp16, 8 for fp32), T = 2 table pieces), and the counted waits.
// Round 4: a quadrant's epilogue (arithmetic + its S stores) runs INSIDE the MFMA segment that follows the one that finished
// it — co-issued with that segment's 16 MFMAs on another quadrant (an MFMA 16x16x32 holds the wave's issue for 8 of its 16
// cycles) — instead of in the load segment in front of it, where it stretched the barrier interval of both wave groups
// (tools/ab_graph.py on the ViT-B/16 forward: 8.5 ms with every GEMM epilogue removed, 11.0 with them).  The order of the
// vector-memory operations is unchanged; only the three waits that used to sit behind a store batch now sit in front of it:
//     p0(L)   D W1(L+1)                        wait W1(L)          : 8
//     p1(L)   D X1(L+1)                        wait X1(L)          : 8              | MFMAs (0,1) + E00: S00
//     p2(L)   D X0(L+2), T                                                          | MFMAs (1,1) + E01: S01
//     p3(L)   D W0(L+2)                        wait X0,W0(L+1)     : 8 + 2S + T     | MFMAs (1,0) + E11: S11
//     p0(L+1) D W1(L+2)                        wait W1(L+1)        : 8 + 3S + T     | MFMAs (0,0) + E10: S10
//     p1(L+1) D X1(L+2)                        wait X1(L+1)        : 8 + 4S + T
//     p2(L+1) D X0(L+3)
//     p3(L+1) D W0(L+3)                        wait X0,W0(L+2)     : 8 + 2S
//     p0(L+2) ...                              wait W1(L+2)        : 8      (waits for S10 too: 8 + S would do)
// (a count = number of operations issued after the target; counts above 63 are clamped, which only waits
// for more).  A K tile that issues residual loads (R = 2 for fp16) at p0 uses 8 + R in its waits.
#include "common.h"
#include "gemm256.h"
#include <stdlib.h>

namespace tlxmi {

typedef __attribute__((address_space(3))) void* lds_ptr_gs_t;
static __device__ __forceinline__ void gs_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_gs_t)lds, 16, voff, 0, 0, 0);
}
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t gs_srd(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
static __device__ __forceinline__ u32x4 gs_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
}
static __device__ __forceinline__ void gs_store16_nt(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 2);
}
static __device__ __forceinline__ void gs_store16_wb(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 0);
}

__device__ __attribute__((aligned(16))) float g_ones4[4] = {1.f, 1.f, 1.f, 1.f};

template <typename T> struct MmaGS;
template <> struct MmaGS<half_t> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, a), __builtin_bit_cast(half8v, b), c, 0, 0, 0);
    }
};
template <> struct MmaGS<float> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], c, 0, 0, 0);
        return c;
    }
};

enum { GS_K0_FIRST = 0, GS_K0_AFTER, GS_INTERIOR, GS_LAST, GS_R0, GS_RC = GS_R0 + 8 };   // GS_R0 + r: residual step r

template <int N> __device__ __forceinline__ void gs_vmcnt() {
    constexpr int C = N > 63 ? 63 : N;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C) : "memory");
}

// ACT: TLXMI_ACT_NONE / RELU / GELU (other activations, and exact-erf GELU in fp32, stay on gemm_pp.hip).
// RES: a.res is added (before the activation; a.scale must be null) — needs >= 11 K tiles.
// ROWAFF: the CONSUMER of a folded LayerNorm — y = act(a[m] * acc + b[m] * c1[n] + c2[n]) with c1 = a.scale, c2 = a.shift and the
//   per-row (a, b) = (rstd, -mean * rstd) formed in the kernel from the producer's a.rowstats[m][p] = (sum, sum^2) of row m over its
//   p-th 256 channels, p < a.ln_planes <= 4 (32 bytes a row): ONE more table piece per tile and wave (32 rows = one contiguous
//   kilobyte on the 64 lanes), added up and inverted once per tile by rowab_convert.
// STATS: the PRODUCER — every quadrant epilogue also adds up its 8 channels x 4 rows per lane (sum, sum of squares of the fp32
//   values before the rounding) and reduces the four lanes of a row with three lane-swap steps per four quantities: lane l then
//   holds (sum, sum^2) of row 128 H + 64 wr + l over the wave's 32 channels of column half G.  The two column halves of a row
//   half H go into an LDS scratch [H][G][wr][wc][64] (a tile's epilogues run in the order E00, E01, E11, E10), and one epilogue
//   after the second of them — at least one barrier later — the eight pairs (2 column halves x 4 wc waves) of a row are added up
//   and wave wc == 0 stores (sum, sum^2) of the row over the tile's 256 channels
//   at a.stats_out[m][bn0 / 256] (rows of 4 pairs, 32 bytes): E11 carries half 0 of its own tile, E00 half 1 of the tile before (the last tile's at the end
//   of the kernel).  Every phase of the last K tile still issues exactly ONE more store (S + 1 in every counted wait): E01 / E10
//   send a dummy out of range, for E00 / E11 it is the stage-2 store, issued by the load segment of their phase.  No atomics, one writer per (plane, row), a fixed order of additions: bit-reproducible.
// (the ablation branches of the tuning flavour exist in the plain variants only: with them the LayerNorm-fold variants spill — 880 bytes
//  of scratch per lane for GELU + ROWAFF, i.e. a vmcnt(0) drain per reload — and tools/ab_graph.py would time an artefact)
#define GS_DBG(args, bit) (!ROWAFF && !STATS && TLXMI_DBG(args, bit))
template <typename T, int ACT, bool RES, bool ROWAFF = false, bool STATS = false>
__global__ __launch_bounds__(512) void gemm_stream_kernel(const Gemm256Args a) {
    constexpr int ES = (int)sizeof(T);
    constexpr int HALF = 128 * 128;            // bytes of a half tile
    constexpr int RX0 = 0, RX1 = HALF, RW0 = 2 * HALF, RW1 = 3 * HALF;   // regions of a K tile
    constexpr int TABLE = 8 * HALF;            // two tables of 8 x 256 B behind the two K tiles
    constexpr int OOB = (int)0x80000000;
    constexpr int ROWTAB = TABLE + 2 * 2048;   // ROWAFF: two tables of 4 planes x 256 rows x (sum, sum^2) behind the channel tables
    constexpr int STATSCR = TABLE + 2 * 2048;  // STATS (never with ROWAFF): [H][G][wr][wc][64 lanes] x (sum, sum^2), 16 KB
    constexpr int SY = ES == 2 ? 4 : 8;        // 16-byte stores of a quadrant's outputs
    constexpr int S = SY + (STATS ? 1 : 0), R = ES, TT = ROWAFF ? 3 : 2;   // R: loads of one residual step (2 pixel rows x 8 channels per lane)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int nb = a.tiles_total;
    const int n_mine = ((int)blockIdx.x < nb) ? (nb - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;

    // i-th tile of this workgroup -> origin.  Virtual block id = blockIdx + i*grid (grid is a multiple of 8,
    // so the id keeps its XCD); ids sharing an XCD walk consecutive tiles, N tiles fastest.
    auto tile_origin = [&](int i, int& bm0, int& bn0, bool& half) -> bool {
        half = false;
        if (i >= n_mine) return false;
        const int id = (int)blockIdx.x + i * (int)gridDim.x;
        half = id >= a.tiles_full;
        const int n = half ? nb - a.tiles_full : a.tiles_full, j = half ? id - a.tiles_full : id;      // walk each kind's list
        const int xcd = j & 7, qd = n >> 3, rm = n & 7;
        const int L = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (j >> 3);
        // column panels of a.gn N tiles (all of them: one panel), N fastest inside a panel, the panel's row tiles top to bottom
        const int mt = half ? n / a.ntiles : (a.m_full >> 8);
        const int per = mt * a.gn, p = L / per, r = L - p * per;
        const int gh = a.ntiles - p * a.gn < a.gn ? a.ntiles - p * a.gn : a.gn;
        const int tm = r / gh;
        bm0 = half ? a.m_full + tm * 128 : tm * 256;
        bn0 = (p * a.gn + r - tm * gh) * 256;
        return true;
    };

    const __amdgpu_buffer_rsrc_t xsrd = gs_srd(a.x, a.x_bytes), wsrd = gs_srd(a.w, a.w_bytes);
    const __amdgpu_buffer_rsrc_t ysrd = gs_srd(a.y, a.y_bytes);
    const __amdgpu_buffer_rsrc_t rsrd = gs_srd(a.res ? a.res : a.y, a.res ? a.res_bytes : 0u);
    const __amdgpu_buffer_rsrc_t hsrd = gs_srd(a.shift, a.shift ? (unsigned)a.Cout * 4u : 0u);   // null: zero fill
    const __amdgpu_buffer_rsrc_t ssrd = a.scale ? gs_srd(a.scale, (unsigned)a.Cout * 4u) : gs_srd(g_ones4, 16u);
    const __amdgpu_buffer_rsrc_t rowsrd = gs_srd(a.rowstats, (ROWAFF && a.rowstats) ? (unsigned)a.M * 32u : 0u);   // [M][4 planes][2]; null: zero fill
    const __amdgpu_buffer_rsrc_t stsrd = gs_srd(a.stats_out, (STATS && a.stats_out) ? (unsigned)a.M * 32u : 0u);   // [M][4 planes][2]; null: dropped

    // ---- loader (gemm_pp.hip): piece = 8 rows x 128 B; wave w fills pieces w, w+8 of a half tile
    const int lrow = lane >> 3;
    const int lc = (lane & 7) ^ ((4 * (wid & 1) + (lane >> 4)) & 7);
    // Row offsets of cursor A (halves X1, W1) and cursor B (X0, W0) for piece `wid`; piece `wid + 8` is 64 rows
    // further (a wave-uniform stride).  Rows past M and filter rows past the padded Cout fail the
    // descriptors' range checks (zero fill); a cursor past the last tile is out of range altogether.
    int xa, wa, xb, wb;
    const int x64 = 64 * a.x_ld * ES, w64 = 64 * a.Kp_bytes;
    auto set_rows = [&](int i, int half, int& xo, int& wo) {
        int bm0 = 0, bn0 = 0;
        bool th;
        const bool ok = tile_origin(i, bm0, bn0, th);
        const int row = 128 * half + 8 * wid + lrow;
        const int n = (row & ~31) | (((row >> 2) & 3) << 3) | (((row >> 4) & 1) << 2) | (row & 3);
        const int xrow = GS_DBG(a, 8) ? ((bm0 + row) & 2047) : bm0 + row;      // (ablation bit 8: every X row from the first 2048 — L2-resident operand, timing only)
        xo = (ok && !(th && half)) ? xrow * a.x_ld * ES : OOB;      // a half-height tile has no X1
        wo = ok ? ((GS_DBG(a, 32) ? 0 : bn0) + n) * a.Kp_bytes : OOB;      // (ablation bit 32: every tile multiplies filter rows 0 .. 255 — L2-resident filter, timing only)
    };
    char* const lbase = smem + wid * 1024;
    // (an offset that is out of range stays out of range after the small additions: x_bytes, w_bytes < 2^31)
    auto dma_x = [&](int region, int par, int xo, int kt) {
        const int q = kt * 8 + lc;
        const int off = (q < a.kchunks && xo >= 0) ? xo + q * 16 : OOB;
        char* b = lbase + (par << 16) + region;
        gs_dma16(xsrd, b, off);
        gs_dma16(xsrd, b + 8192, off >= 0 ? off + x64 : OOB);
    };
    auto dma_w = [&](int region, int par, int wo, int kt) {
        const int q = kt * 8 + lc;
        const int off = (q * 16 < a.Kp_bytes && wo >= 0) ? wo + q * 16 : OOB;
        char* b = lbase + (par << 16) + region;
        gs_dma16(wsrd, b, off);
        gs_dma16(wsrd, b + 8192, off >= 0 ? off + w64 : OOB);
    };
    // scale / shift table of tile i -> table (i & 1): wave w brings channels 32w..32w+31, [shift 32][scale 32]
    auto dma_table = [&](int i) {
        int bm0 = 0, bn0 = 0;
        bool th;
        const bool ok = tile_origin(i, bm0, bn0, th);
        char* dst = smem + TABLE + (i & 1) * 2048 + wid * 256;
        const int lane = lane_now();
        if (lane < 8) {
            const int off = ok ? (bn0 + 32 * wid + 4 * lane) * 4 : OOB;
            gs_dma16(hsrd, dst, off);
            gs_dma16(ssrd, dst + 128, a.scale ? off : 0);
        }
        // ROWAFF: the statistics of rows 32w .. 32w+31 of the tile, [row][4 planes][2] = 32 bytes a row, ONE contiguous kilobyte per wave
        // (rows past M: zero fill; planes past ln_planes were never written and are left out by rowab_convert)
        if constexpr (ROWAFF) gs_dma16(rowsrd, smem + ROWTAB + (i & 1) * 8192 + wid * 1024, ok ? (bm0 + 32 * wid) * 32 + lane * 16 : OOB);
    };

    // ---- fragment reads (gemm_pp.hip)
    const int frow = lane & 15, fg = lane >> 4;
    const int foff = frow * 128 + ((fg ^ ((frow >> 1) & 7)) << 4);
    const int xf0 = wr * 64 * 128 + foff, wf0 = wc * 32 * 128 + foff;

    f32x4 acc[4][8];   // [2*g + ci][4*h + pi]
    u32x4 xf[4][2], w0f[2][2], w1f[2][2];
    u32x4 rr[R];       // residual step in flight (RES)
    auto read_x = [&](const char* kb, int region) {
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            xf[pi][0] = *reinterpret_cast<const u32x4*>(kb + region + pi * 2048 + xf0);
            xf[pi][1] = *reinterpret_cast<const u32x4*>(kb + region + pi * 2048 + (xf0 ^ 64));
        }
    };
    auto read_w = [&](const char* kb, int region, u32x4 (&wf)[2][2]) {
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
            wf[ci][0] = *reinterpret_cast<const u32x4*>(kb + region + ci * 2048 + wf0);
            wf[ci][1] = *reinterpret_cast<const u32x4*>(kb + region + ci * 2048 + (wf0 ^ 64));
        }
    };

    // the S stores of a quadrant that does not exist (H = 1 of a half-height tile), dropped by the range check: the counts stay
    auto dead_stores = [&]() {
#pragma unroll
        for (int q = 0; q < SY; ++q) gs_store16_wb(ysrd, u32x4{0u, 0u, 0u, 0u}, OOB);
    };

    // ROWAFF: the (a, b) pairs of this lane's four rows (sub-tiles pi = 0..3 of half h) are read from the row table in the LOAD segment of
    // the phase whose MFMA segment runs the first epilogue of that half — p1 of the last K tile for h = 0 (E00, E01), p3 for h = 1 (E11, and
    // E10 one phase later) — by ONE asm block with its own lgkmcnt(0): eight registers live across two phases.  (Read inside the MFMA
    // segment, one ds_read_b64 right in front of its use, the pairs of sub-tile 1 of h = 1 came back with b = 0 in lanes 48 - 63 for the
    // low halves of the packed FMAs on some launches — tools/dbg notes in DESIGN 5.4; the table itself was right.)
    f32x2 rab[4];
    auto rowab_fetch = [&](int h, int tpar) {
        if constexpr (ROWAFF) {
            // row r of the tile lives at (r / 32) * 1024 + (r % 32) * 32 (rowab_convert): rows 128 h + 64 wr + 16 pi + px -> offsets 0, 512, 1024, 1536
            const int ln = lane_now();
            const unsigned la = (unsigned)(uintptr_t)(lds_ptr_gs_t)(smem + ROWTAB + tpar * 8192 + (4 * h + 2 * wr) * 1024 + (ln & 15) * 32);
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(rab[0]), "=&v"(rab[1]), "=&v"(rab[2]), "=&v"(rab[3])
                         : "v"(la)
                         : "memory");
        }
    };
    // ROWAFF: the statistics of a tile's rows -> its (a, b) table, in place over each row's first pair.  Wave w converts the 32 rows it
    // brought itself (dma_table: complete since the p3 wait of the tile's first K tile), one row per lane of its lower half, in the load
    // segment of p0 of the LAST K tile; the barrier of that phase stands between this write and the reads of rowab_fetch (p1, p3) by the
    // other waves.
    auto rowab_convert = [&](int tpar) {
        if constexpr (ROWAFF) {
            const int ln = lane_now();
            if (ln < 32) {
                const unsigned la = (unsigned)(uintptr_t)(lds_ptr_gs_t)(smem + ROWTAB + tpar * 8192 + wid * 1024 + ln * 32);
                f32x4 t01, t23;
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t01), "=&v"(t23) : "v"(la) : "memory");
                const int np = a.ln_planes;
                const float sm = (t01[0] + (np > 1 ? t01[2] : 0.f)) + ((np > 2 ? t23[0] : 0.f) + (np > 3 ? t23[2] : 0.f));
                const float sq = (t01[1] + (np > 1 ? t01[3] : 0.f)) + ((np > 2 ? t23[1] : 0.f) + (np > 3 ? t23[3] : 0.f));
                const float mean = sm * a.ln_inv_c;
                const float var = fmaxf(__builtin_fmaf(-mean, mean, sq * a.ln_inv_c), 0.f);
                const float rstd = 1.f / sqrtf(var + a.ln_eps);
                const f32x2 ab = f32x2{rstd, -mean * rstd};
                asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(la), "v"(ab) : "memory");
            }
        }
    };

    int bm0 = 0, bn0 = 0, pbm0 = 0, pbn0 = 0;      // origin of the tile being multiplied, of the one before
    bool hcur = false;      // the tile being multiplied is half-height

    // STATS, stage 2: the eight pairs (2 column halves x 4 wc waves) of this lane's row of half `hp` of the tile at (sbm, sbn), read from
    // the scratch by asm blocks (the same precaution as rowab_fetch), added in a fixed order, stored by wave wc == 0.  Runs in a LOAD
    // segment, behind that segment's DMA and in front of its counted wait: the store stands where the quadrant's (S + 1)-th store stood
    // relative to every LATER wait's target, and the wait right behind it only gets one operation more conservative.
    bool have_prev = false;                            // the tile before this one left its half 1 in the scratch
    auto stats_stage2 = [&](int hp, int sbm, int sbn, bool pend) {
        if constexpr (STATS) {
            const int ln = lane_now();
            const unsigned la = (unsigned)(uintptr_t)(lds_ptr_gs_t)(smem + STATSCR + hp * 8192 + wr * 2048 + ln * 8);
            // (all eight pairs in flight: ONE LDS round trip in this load segment — four blocks of two cost the producers 6 %)
            f32x2 p0, p1, p2, p3, p4, p5, p6, p7;
            asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\tds_read_b64 %3, %8 offset:1536\n\t"
                         "ds_read_b64 %4, %8 offset:4096\n\tds_read_b64 %5, %8 offset:4608\n\tds_read_b64 %6, %8 offset:5120\n\tds_read_b64 %7, %8 offset:5632\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7)
                         : "v"(la)
                         : "memory");
            const float s4 = ((p0[0] + p1[0]) + (p2[0] + p3[0])) + ((p4[0] + p5[0]) + (p6[0] + p7[0]));
            const float q4 = ((p0[1] + p1[1]) + (p2[1] + p3[1])) + ((p4[1] + p5[1]) + (p6[1] + p7[1]));
            const int m = sbm + 128 * hp + 64 * wr + ln;
            const int okm = (pend && wc == 0) ? ((m - a.M) >> 31) : 0;
            const int so = (((m * 4 + (sbn >> 8)) * 8) & okm) | (OOB & ~okm);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, s4), __builtin_bit_cast(unsigned, q4)}, stsrd, so, 0, 0);
        }
    };

    // ---- quadrant epilogue: lane (fg, px) owns channels 128g + 32wc + 8fg .. +7 of pixel rows
    // 128h + 64wr + 16pi + px.  Always S stores (suppressed ones go to an out-of-range offset).
    auto epi = [&](auto h_tag, auto g_tag, int bm0, int bn0, int tpar, bool live) {
        constexpr int H = decltype(h_tag)::value, G = decltype(g_tag)::value;
        const int ln = lane_now();     // offsets are recomputed here, not kept live across the K loop (common.h: lane_now)
        const int px = ln & 15, fg = ln >> 4;
        const int col = 128 * G + 32 * wc + 8 * fg;
        const int ch0 = bn0 + col;
        // (branch-free selects on purpose: an exec-masked branch would cut the MFMA segment this runs in into basic blocks and
        //  the scheduler could no longer put the arithmetic between the MFMAs)
        const int chm = (live && !GS_DBG(a, 2)) ? ((ch0 - a.Cout) >> 31) : 0;      // -1: a real channel (Cout is a multiple of 8 on this path)
        if GS_DBG(a, 1) {   // ablation: stores without the arithmetic
#pragma unroll
            for (int pi = 0; pi < S; ++pi) gs_store16_nt(ysrd, __builtin_bit_cast(u32x4, acc[2 * G][4 * H + (pi & 3)]), OOB);
            return;
        }
        const float* tb = reinterpret_cast<const float*>(smem + TABLE + tpar * 2048 + (4 * G + wc) * 256) + 8 * fg;
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(tb), h1 = *reinterpret_cast<const f32x4*>(tb + 4);
        f32x4 s0 = f32x4{1.f, 1.f, 1.f, 1.f}, s1 = s0;
        if constexpr (!RES) {   // a residual implies scale == nullptr
            s0 = *reinterpret_cast<const f32x4*>(tb + 32);
            s1 = *reinterpret_cast<const f32x4*>(tb + 36);
        }
        float st_s[4], st_q[4];      // STATS: this lane's 8 channels of row pi
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            const int m = bm0 + 128 * H + 64 * wr + 16 * pi + px;
            float v[8];
            if constexpr (ROWAFF) {   // y = a[m] * acc + b[m] * c1[n] + c2[n]   (rab: rowab_fetch, one or two phases ago)
                const f32x2 ab = rab[pi];
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    v[bb] = acc[2 * G][4 * H + pi][bb] * ab[0] + (ab[1] * s0[bb] + h0[bb]);
                    v[4 + bb] = acc[2 * G + 1][4 * H + pi][bb] * ab[0] + (ab[1] * s1[bb] + h1[bb]);
                }
            } else {
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    if constexpr (RES) {
                        v[bb] = acc[2 * G][4 * H + pi][bb] + h0[bb];
                        v[4 + bb] = acc[2 * G + 1][4 * H + pi][bb] + h1[bb];
                    } else {
                        v[bb] = acc[2 * G][4 * H + pi][bb] * s0[bb] + h0[bb];
                        v[4 + bb] = acc[2 * G + 1][4 * H + pi][bb] * s1[bb] + h1[bb];
                    }
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
            if constexpr (STATS) {      // channels past Cout are zeros here (zero filter rows, zero-filled shift / residual)
                float ss = v[0], qq = v[0] * v[0];
#pragma unroll
                for (int e = 1; e < 8; ++e) { ss += v[e]; qq = __builtin_fmaf(v[e], v[e], qq); }
                st_s[pi] = ss;
                st_q[pi] = qq;
            }
            const int okm = chm & ((m - a.M) >> 31);
            int yo = (((m * a.y_ld + ch0) * ES) & okm) | (OOB & ~okm);   // out-of-range stores are dropped
            if GS_DBG(a, 16) {   // ablation (wrong data, right byte count): this store instruction writes 8 whole 128-byte lines
                const int mm = bm0 + 128 * H + 64 * wr + 16 * pi + 8 * (col >> 7) + (ln >> 3), cc = bn0 + (128 * (G ^ (col >> 7)) + 64 * (wc >> 1) + 8 * (ln & 7));
                yo = (mm < a.M && cc < a.Cout && live) ? (mm * a.y_ld + cc) * ES : OOB;
            }
            if constexpr (ES == 2) {
                half8v hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
                if (!GS_DBG(a, 4)) gs_store16_wb(ysrd, __builtin_bit_cast(u32x4, hv), yo);      // (bit 4, A/B: non-temporal)
                else gs_store16_nt(ysrd, __builtin_bit_cast(u32x4, hv), yo);
            } else {
                f32x4 f0, f1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { f0[e] = v[e]; f1[e] = v[4 + e]; }
                gs_store16_nt(ysrd, __builtin_bit_cast(u32x4, f0), yo);
                gs_store16_nt(ysrd, __builtin_bit_cast(u32x4, f1), yo + 16);
            }
        }
        if constexpr (STATS) {
            // lanes (fg = 0..3, px) hold the four 8-channel parts of row px of sub-tile pi.  v_permlane16_swap(A, B) trades A's odd lane
            // rows with B's even ones, so A + B afterwards = [A0+A1, B0+B1, A2+A3, B2+B3] by lane row: one swap and one add take TWO
            // quantities one level up; v_permlane32_swap joins the halves: lane row g ends with the 32-channel total of sub-tile pi = g,
            // i.e. lane l with the total of row 128 H + 64 wr + l.
            auto tree = [&](const float (&x)[4]) -> float { return ln_row_tree(x[0], x[1], x[2], x[3]); };
            const float ts = tree(st_s), tq = tree(st_q);
            // this wave's pair -> scratch [H][G][wr][wc][lane].  The quadrant's extra store: E01 / E10 send a dummy; for E00 / E11 it is the
            // stage-2 store of the half completed one epilogue ago, issued by the load segment of this same phase (stats_stage2)
            // (visible to the other waves from the barrier behind this segment: GS_SYNC_E waits for lgkmcnt(0) in front of it)
            *reinterpret_cast<f32x2*>(smem + STATSCR + H * 8192 + G * 4096 + wr * 2048 + wc * 512 + ln * 8) = f32x2{ts, tq};
            if constexpr (H != G) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, stsrd, OOB, 0, 0);
        }
    };
    // Residual step r = 0..7 covers quadrant (r>>1) in the phase order (0,0) (0,1) (1,1) (1,0), pixel sub-tiles
    // 2*(r&1), 2*(r&1)+1: R loads now, added into the accumulators one K tile later.
    auto res_load = [&](auto r_tag, int bm0, int bn0, bool thalf) {
        constexpr int RS = decltype(r_tag)::value, Q = RS >> 1, H = (Q >> 1), G = (Q == 1 || Q == 2) ? 1 : 0, P0 = 2 * (RS & 1);
        const int ln = lane_now();
        const int px = ln & 15, fg = ln >> 4;
        const int ch0 = bn0 + 128 * G + 32 * wc + 8 * fg;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int m = bm0 + 128 * H + 64 * wr + 16 * (P0 + p) + px;
            const int okm = (H == 1 && thalf) ? 0 : (((m - a.M) & (ch0 - a.Cout)) >> 31);      // (H = 1 of a half-height tile: zeros)
            const int ro = (((m * a.res_ld + ch0) * ES) & okm) | (OOB & ~okm);
#pragma unroll
            for (int hh = 0; hh < ES / 2; ++hh) rr[p * (ES / 2) + hh] = gs_load16(rsrd, ro + 16 * hh);
        }
    };
    auto res_add = [&](auto r_tag) {
        constexpr int RS = decltype(r_tag)::value, Q = RS >> 1, H = (Q >> 1), G = (Q == 1 || Q == 2) ? 1 : 0, P0 = 2 * (RS & 1);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float rv[8];
            if constexpr (ES == 2) {
                const half8v hv = __builtin_bit_cast(half8v, rr[p]);
#pragma unroll
                for (int e = 0; e < 8; ++e) rv[e] = (float)hv[e];
            } else {
                const f32x4 r0 = __builtin_bit_cast(f32x4, rr[2 * p]), r1 = __builtin_bit_cast(f32x4, rr[2 * p + (ES / 2 - 1)]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { rv[e] = r0[e]; rv[4 + e] = r1[e]; }
            }
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                acc[2 * G][4 * H + P0 + p][bb] += rv[bb];
                acc[2 * G + 1][4 * H + P0 + p][bb] += rv[4 + bb];
            }
        }
    };

#define GS_SYNC()                          \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);
// the barrier behind an MFMA segment that carried a quadrant epilogue: a STATS kernel first waits for that epilogue's LDS write (long
// landed by now; a wait right behind the write would stall the segment), so that the other waves may read it after the next barrier
#define GS_SYNC_E()                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                               \
    if constexpr (STATS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
    __builtin_amdgcn_s_barrier();                                                    \
    __builtin_amdgcn_sched_barrier(0);
// one quadrant x one K tile; ZERO: the accumulators start from 0 (first K tile of an output tile).  EPI: a statement (the
// epilogue of the quadrant finished one phase earlier) whose instructions the scheduler spreads between the MFMAs: 3 vector
// instructions behind each MFMA, the rest (and the stores) behind the last one.
#define GS_MMA_E(H, G, WF, ZERO, EPI)                                                                       \
    {                                                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        _Pragma("unroll") for (int pi = 0; pi < 4; ++pi)                                                    \
        _Pragma("unroll") for (int ci = 0; ci < 2; ++ci)                                                    \
            acc[2 * G + ci][4 * H + pi] = MmaGS<T>::run(WF[ci][0], xf[pi][0],                               \
                                                        (ZERO) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[2 * G + ci][4 * H + pi]); \
        _Pragma("unroll") for (int pi = 0; pi < 4; ++pi)                                                    \
        _Pragma("unroll") for (int ci = 0; ci < 2; ++ci)                                                    \
            acc[2 * G + ci][4 * H + pi] = MmaGS<T>::run(WF[ci][1], xf[pi][1], acc[2 * G + ci][4 * H + pi]); \
        EPI;                                                                                                \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                    \
            __builtin_amdgcn_sched_group_barrier(0x008, sizeof(T) == 2 ? 1 : 4, 0);                         \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                              \
        }                                                                                                   \
        __builtin_amdgcn_s_setprio(0);                                                                      \
    }
#define GS_MMA(H, G, WF, ZERO) GS_MMA_E(H, G, WF, ZERO, (void)0)

    const int ks = a.ksteps;
    // stream state: compute side (tile ordinal ci_, buffer parity cpar) and the two DMA cursors
    int cpar = 0;
    int ia = 0, kta = 1, para = 1;    // cursor A: next (W1, X1) to stage = stream position 1
    int ib = 0, ktb = 2, parb = 0;    // cursor B: next (X0, W0) to stage = stream position 2
    auto adv_a = [&]() {
        para ^= 1;
        if (++kta == ks) { kta = 0; ++ia; set_rows(ia, 1, xa, wa); }
    };
    auto adv_b = [&]() {
        parb ^= 1;
        if (++ktb >= ks) { ktb -= ks; ++ib; set_rows(ib, 0, xb, wb); }
    };

    // ---- prologue: table of tile 0; X0 W0 W1 X1 of K tile 0; X0 W0 of K tile 1 (ks >= 2)
    set_rows(0, 1, xa, wa);
    set_rows(0, 0, xb, wb);
    dma_table(0);
    dma_x(RX0, 0, xb, 0);
    dma_w(RW0, 0, wb, 0);
    dma_w(RW1, 0, wa, 0);
    dma_x(RX1, 0, xa, 0);
    dma_x(RX0, 1, xb, 1);
    dma_w(RW0, 1, wb, 1);
    if (ks == 2) { ktb = 0; ib = 1; set_rows(1, 0, xb, wb); }
    gs_vmcnt<8>();
    GS_SYNC();
    if (wr == 1) { GS_SYNC(); }   // group 1 runs one barrier behind


    // HF: the tile is half-height (compile time: a run-time test around the MFMA blocks makes the register allocator split the
    // accumulators' live ranges at every join — 80 - 120 spilled registers; a half-height tile is always a workgroup's last)
    auto ktile = [&](auto mode_tag, auto half_tag, int i) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool HF = decltype(half_tag)::value != 0;
        constexpr bool K0 = MODE == GS_K0_FIRST || MODE == GS_K0_AFTER;
        constexpr bool RI = MODE >= GS_R0 && MODE < GS_RC;   // issues residual loads
        constexpr int RQ = RI ? R : 0;
        const char* kb = smem + (cpar << 16);
        // ---- p0: quadrant (0,0)
        read_x(kb, RX0);
        read_w(kb, RW0, w0f);
        if constexpr (MODE == GS_LAST) rowab_convert(i & 1);
        if constexpr (MODE > GS_R0 && MODE <= GS_RC) res_add(IntTag<MODE - GS_R0 - 1>{});
        if constexpr (RI) {
            __builtin_amdgcn_sched_barrier(0);   // the new loads re-use the registers just consumed
            res_load(IntTag<MODE - GS_R0>{}, bm0, bn0, HF);
        }
        dma_w(RW1, para, wa, kta);
        gs_vmcnt<(MODE == GS_K0_AFTER ? 8 + 3 * S + TT : 8 + RQ)>();
        GS_SYNC();
        if constexpr (MODE == GS_K0_AFTER) { GS_MMA_E(0, 0, w0f, K0, epi(IntTag<1>{}, IntTag<0>{}, pbm0, pbn0, (i - 1) & 1, true)); }      // (the tile before is never half-height)
        else { GS_MMA(0, 0, w0f, K0); }
        if constexpr (MODE == GS_K0_AFTER) { GS_SYNC_E(); } else { GS_SYNC(); }
        // ---- p1: quadrant (0,1)
        read_w(kb, RW1, w1f);
        if constexpr (MODE == GS_LAST) rowab_fetch(0, i & 1);
        dma_x(RX1, para, xa, kta);
        if constexpr (MODE == GS_LAST) stats_stage2(1, pbm0, pbn0, have_prev);      // half 1 of the tile before (completed by its E10): E00's extra store
        adv_a();
        gs_vmcnt<(MODE == GS_K0_AFTER ? 8 + 4 * S + TT : 8 + RQ)>();
        GS_SYNC();
        if constexpr (MODE == GS_LAST) { GS_MMA_E(0, 1, w1f, K0, epi(IntTag<0>{}, IntTag<0>{}, bm0, bn0, i & 1, true)); }
        else { GS_MMA(0, 1, w1f, K0); }
        if constexpr (MODE == GS_LAST) { GS_SYNC_E(); } else { GS_SYNC(); }
        // ---- p2: quadrant (1,1) — idle in a half-height tile (its MFMA segment still carries the epilogue of (0,1))
        if constexpr (!HF) read_x(kb, RX1);
        dma_x(RX0, parb, xb, ktb);
        if constexpr (MODE == GS_LAST) dma_table(i + 1);
        GS_SYNC();
        if constexpr (!HF) {
            if constexpr (MODE == GS_LAST) { GS_MMA_E(1, 1, w1f, K0, epi(IntTag<0>{}, IntTag<1>{}, bm0, bn0, i & 1, true)); }
            else { GS_MMA(1, 1, w1f, K0); }
        } else {
            if constexpr (MODE == GS_LAST) epi(IntTag<0>{}, IntTag<1>{}, bm0, bn0, i & 1, true);
        }
        if constexpr (MODE == GS_LAST) { GS_SYNC_E(); } else { GS_SYNC(); }
        // ---- p3: quadrant (1,0)
        if constexpr (MODE == GS_LAST && !HF) rowab_fetch(1, i & 1);
        dma_w(RW0, parb, wb, ktb);
        if constexpr (MODE == GS_LAST) stats_stage2(0, bm0, bn0, true);      // half 0 of this tile (completed by E01): E11's extra store (also of a half-height tile)
        adv_b();
        gs_vmcnt<(MODE == GS_K0_AFTER ? 8 + 2 * S : MODE == GS_LAST ? 8 + 2 * S + TT : 8 + RQ)>();
        GS_SYNC();
        if constexpr (!HF) {
            if constexpr (MODE == GS_LAST) { GS_MMA_E(1, 0, w0f, K0, epi(IntTag<1>{}, IntTag<1>{}, bm0, bn0, i & 1, true)); }
            else { GS_MMA(1, 0, w0f, K0); }
        } else {
            if constexpr (MODE == GS_LAST) dead_stores();
        }
        if constexpr (MODE == GS_LAST) { GS_SYNC_E(); } else { GS_SYNC(); }
        cpar ^= 1;
    };

    auto tile = [&](auto half_tag, int i) {
        if (i == 0) ktile(IntTag<GS_K0_FIRST>{}, half_tag, i);
        else ktile(IntTag<GS_K0_AFTER>{}, half_tag, i);
        int kt = 1;
        if constexpr (RES) {
            ktile(IntTag<GS_R0>{}, half_tag, i);
            ktile(IntTag<GS_R0 + 1>{}, half_tag, i);
            ktile(IntTag<GS_R0 + 2>{}, half_tag, i);
            ktile(IntTag<GS_R0 + 3>{}, half_tag, i);
            ktile(IntTag<GS_R0 + 4>{}, half_tag, i);
            ktile(IntTag<GS_R0 + 5>{}, half_tag, i);
            ktile(IntTag<GS_R0 + 6>{}, half_tag, i);
            ktile(IntTag<GS_R0 + 7>{}, half_tag, i);
            ktile(IntTag<GS_RC>{}, half_tag, i);
            kt = 10;
        }
        for (; kt < ks - 1; ++kt) ktile(IntTag<GS_INTERIOR>{}, half_tag, i);
        ktile(IntTag<GS_LAST>{}, half_tag, i);
    };
    // this workgroup's tiles: n_full whole tiles, then at most one half-height tile (its last: launch_gs) — run behind the loop,
    // not inside it, so that the two code paths meet at no loop-carried join
    bool last_half = false;
    {
        int b0, b1;
        if (n_mine > 0) tile_origin(n_mine - 1, b0, b1, last_half);
    }
    const int n_full = n_mine - (last_half ? 1 : 0);
    for (int i = 0; i < n_full; ++i) {
        pbm0 = bm0;
        pbn0 = bn0;
        have_prev = i > 0;
        tile_origin(i, bm0, bn0, hcur);
        tile(IntTag<0>{}, i);
    }
    if (last_half) {
        pbm0 = bm0;
        pbn0 = bn0;
        have_prev = n_full > 0;
        tile_origin(n_full, bm0, bn0, hcur);
        tile(IntTag<1>{}, n_full);
    }
    if (wr == 0) { GS_SYNC(); }   // barrier counts match again
    if (n_mine > 0 && !hcur) {
        epi(IntTag<1>{}, IntTag<0>{}, bm0, bn0, (n_mine - 1) & 1, true);
        if constexpr (STATS) {      // ... and half 1 of the last tile, which no later epilogue carries
            GS_SYNC_E();
            stats_stage2(1, bm0, bn0, true);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // zero-fill DMAs of the stream's tail
#undef GS_MMA
#undef GS_MMA_E
#undef GS_SYNC
#undef GS_SYNC_E
#undef GS_DBG
}

// Preconditions as launch_gemm256 (conv_igemm.hip's dispatcher) plus: a.ksteps = packed pitch / 128 >= 2;
// with a residual: fp16, a.scale == nullptr, residual added before the activation, a.ksteps >= 11.
template <typename T, int ACT, bool RES, bool ROWAFF = false, bool STATS = false> static int launch_gs(const Gemm256Args& a0, hipStream_t st, int cus) {
    Gemm256Args a = a0;
    a.debug = (int)tune_int("TLXMI_DEBUG", 0);     // ablation bits: tuning flavour only (TLXMI_DBG is `false` in the product)
    if (const long only = tune_int("TLXMI_DEBUG_COUT", 0); only > 0 && only != a.Cout) a.debug = 0;      // ... on the launches with this Cout only
    a.mtiles = (a.M + 255) / 256;
    a.ntiles = (a.Cout + 255) / 256;
    a.gn = a.ntiles;
    if (const long g = tune_int("TLXMI_GS_PANEL", 3); g > 0 && g < a.ntiles) a.gn = (int)g;
    const size_t lds = (size_t)8 * 128 * 128 + 2 * 2048 + (ROWAFF ? 2 * 8192 : 0) + (STATS ? 16384 : 0);   // two K tiles, channel tables, row tables / statistics scratch
    const void* fn = reinterpret_cast<const void*>(&gemm_stream_kernel<T, ACT, RES, ROWAFF, STATS>);
    if (int rc = raise_lds_limit(fn, (int)lds, "gemm_stream")) return rc;
    int maxgrid = cus & ~7;         // one workgroup per CU; a multiple of 8 keeps a virtual block on its XCD
    if (maxgrid < 8) maxgrid = 8;
    const int tiles = a.mtiles * a.ntiles;
    int grid = maxgrid < tiles ? maxgrid : tiles;   // fewer tiles than CUs: one tile each (ids < tiles, mapping still bijective)
    a.tiles_full = a.tiles_total = tiles;
    a.m_full = a.mtiles * 256;
    // Balanced tail (file header): r whole rounds + a last round at most half full -> the rows behind the whole row tiles of
    // the r rounds become half-height tiles; fewer tiles than half the CUs -> every tile is half-height.  Taken when the
    // half-height tiles are at most one per workgroup (consecutive ids behind the full tiles: it is then that workgroup's LAST tile,
    // which is what the kernel's compile-time half-height path relies on).
    {
        const int r = tiles / maxgrid, rem = tiles % maxgrid;
        const int mt_full = r == 0 ? 0 : (r * maxgrid) / a.ntiles;
        const int halves = ((a.M - mt_full * 256 + 127) / 128) * a.ntiles;
        const bool tail = r >= 1 && rem != 0 && 2 * rem <= maxgrid && halves <= maxgrid;      // (<= one per workgroup, its last tile)
        const bool small = tune_int("TLXMI_HALFTAIL", 1) == 2 && r == 0 && halves <= maxgrid && halves > tiles;      // (measured: a loss on Swin-B — every half-height tile re-reads its whole filter panel; tuning flavour only)
        if ((tail || small) && tune_int("TLXMI_HALFTAIL", 1)) {
            a.tiles_full = mt_full * a.ntiles;
            a.m_full = mt_full * 256;
            a.tiles_total = a.tiles_full + halves;
            grid = maxgrid < a.tiles_total ? maxgrid : a.tiles_total;
        }
    }
    void* args[] = {&a};
    hipError_t e = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(512), args, lds, st);
    if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "gemm_stream: HIP launch failed: %s", hipGetErrorString(e));
    return TLXMI_OK;
}

template <typename T> static int launch_gs_t(const Gemm256Args& a, hipStream_t st, int cus) {
    if constexpr (sizeof(T) == 2) {
        // LayerNorm folded around the Linear layers (gemm_stream_ok: fp16; consumer: no residual, NONE / GELU; producer: NONE)
        if (a.rowstats) {
            if (a.act == TLXMI_ACT_GELU) return launch_gs<T, TLXMI_ACT_GELU, false, true, false>(a, st, cus);
            return launch_gs<T, TLXMI_ACT_NONE, false, true, false>(a, st, cus);
        }
        if (a.stats_out) {
            if (a.res) return launch_gs<T, TLXMI_ACT_NONE, true, false, true>(a, st, cus);
            return launch_gs<T, TLXMI_ACT_NONE, false, false, true>(a, st, cus);
        }
        if (a.res) {
            if (a.act == TLXMI_ACT_RELU) return launch_gs<T, TLXMI_ACT_RELU, true>(a, st, cus);
            return launch_gs<T, TLXMI_ACT_NONE, true>(a, st, cus);
        }
    }
    if (a.act == TLXMI_ACT_RELU) return launch_gs<T, TLXMI_ACT_RELU, false>(a, st, cus);
    if constexpr (sizeof(T) == 2) {
        if (a.act == TLXMI_ACT_GELU) return launch_gs<T, TLXMI_ACT_GELU, false>(a, st, cus);
    }
    return launch_gs<T, TLXMI_ACT_NONE, false>(a, st, cus);
}

bool gemm_stream_ok(int dtype, const Gemm256Args& a) {
    if (a.ksteps < 2) return false;
    if (a.act != TLXMI_ACT_NONE && a.act != TLXMI_ACT_RELU && !(a.act == TLXMI_ACT_GELU && dtype == TLXMI_F16 && !a.res)) return false;
    if (a.res && (dtype != TLXMI_F16 || a.scale != nullptr || (a.flags & TLXMI_EPI_RES_AFTER_ACT) || a.ksteps < 11)) return false;
    if (a.rowstats && (dtype != TLXMI_F16 || a.res || a.stats_out || !a.scale || !a.shift || (a.act != TLXMI_ACT_NONE && a.act != TLXMI_ACT_GELU))) return false;
    if (a.stats_out && (dtype != TLXMI_F16 || a.act != TLXMI_ACT_NONE || (a.Cout & 31))) return false;
    return true;
}

int launch_gemm_stream(int dtype, const Gemm256Args& a, hipStream_t st, int cus) {
    if (dtype == TLXMI_F16) return launch_gs_t<half_t>(a, st, cus);
    return launch_gs_t<float>(a, st, cus);
}

}  // namespace tlxmi

// MEASURED DEAD END, kept as a reference (tools/probe, not built into the library): round 4's second form of gemm_w4.hip.
// Swapped in as tlxcv_amd/csrc/gemm_w4.hip it passes tests/test_gemm_gpu.py -k w4 and the ViT-B/16 Linear shapes against torch
// (DESIGN 5.3), and measures slower than both the first form and the 8-wave kernels: K-loop-only forward 10.7 ms vs 9.1 / 8.5.
// Persistent 256 x 256 tile GEMM on FOUR waves — one per SIMD, each with the whole 512-register file — for the
// MFMA-bound Linear layers of ViT / Swin in fp16 (reference vision_transformer.py:81-87, 112-123; swin_transformer.py:192-229):
//     Y[m][n] = act( sum_k X[m][k] * Wp[n][k] + shift[n] + R[m][n] )           (scale == nullptr: Linear layers)
// Same operands, filter packing and LDS K-tile image as gemm_pp.hip / gemm_stream.hip; its own filter-row permutation.
//
// Why another schedule.  Measured on the ViT-B/16 forward (tools/ab_w4dbg.sh, hipGraph replay, all four Linear layers): with
// the epilogue of every output tile removed the forward takes 9.1 ms instead of 11.0 — arithmetic, residual loads and the
// store burst of 128 KB per CU at every tile end cost a fifth of the forward.  The 8-wave kernels cannot hide it: a wave's
// 128 accumulators + 64 fragment registers leave nothing (238 of 256), so the epilogue runs inside the load segments and
// stretches every barrier interval it touches, and fc1's GELU (6.5 VALU instructions an element) is not hidden at all.
// Here a wave owns 128 x 128 outputs (16 blocks of v_mfma_f32_32x32x16_f16 = 256 accumulators, AGPRs) and the other 256
// registers hold: two sets of 8 fragments (k-step s + 1 is read from LDS while k-step s multiplies) and the finished tile,
// rounded to fp16, in 128 PENDING registers.  Nothing of a tile's epilogue runs at the tile's end except that conversion:
//   * the bias (from a double-buffered LDS table, staged by LDS-DMA) is added as the block is converted;
//   * the residual is added BY THE MATRIX PIPE: acc(32 ch x 32 px) += I(32 x 16) . R(16 ch x 32 px) for the two 16-channel
//     groups of a block — the residual's rows in HBM are exactly B fragments (8 consecutive channels of a pixel), the identity
//     slices are two constant A fragments; 2 MFMAs per block (+4 % at K = 768), no VALU, one rounding (exact fp32 add);
//   * activation (ReLU, or GELU on the fp16-rounded pre-activation) and the 32 stores of a tile are DRAINED from the pending
//     registers two units (2 x 32 bytes of one pixel) per K tile under the NEXT tile's MFMAs: an MFMA 32x32x16 holds the
//     matrix pipe for 32 cycles and the wave's issue for 8 of them, so ~5 other instructions per MFMA are free, and the
//     stores leave the CU as a trickle of 4 per K tile instead of a burst of 128 KB.
// One s_barrier per K tile (64 MFMAs per wave).
//
// K tile = 128 bytes of K (64 halves) = 4 k-steps of 16; in LDS: X rows 0-255 (32 KiB) then W rows 0-255 (32 KiB), 128-byte
// rows, chunk c of row r in slot c ^ ((r >> 1) & 7) (conflict-free for the ds_read_b128 of a 32-row x 16-byte fragment
// column: the 16 lanes of a b128 group cover all 8 swizzles at both row parities); two K tiles resident + 3 KiB of tables.
// Wave w = (wr, wc) = (w >> 1, w & 1): pixel rows 128 wr .. +127, filter rows 128 wc .. +127; acc[ci][pi] = 32 x 32 block.
// LDS filter row 32 ci + 4h + 8q + j of a wave's 128 holds channel 64h + 16 ci + 4q + j, so lane (px, h) of the MFMA result
// owns channels 64h + 16 ci + reg: 64 consecutive channels of one pixel per lane, a full 128-byte line.
//
// Schedule of K tile t (buffer t & 1), per wave:
//     k-steps 0, 1, 2: 16 MFMAs each | 8 ds_read_b128 for the next k-step | drain work (VALU, 4 stores), residual loads / MFMAs
//     s_waitcnt lgkmcnt(0) + vmcnt(NV) + s_barrier      (my reads of buffer t & 1 are done; my pieces of K tile t + 1 have landed)
//     k-step 3: 16 MFMAs | 8 ds_read_b128: k-step 0 of K tile t + 1 | 17 LDS-DMA pieces: K tile t + 2 -> buffer t & 1 (+ table)
// A piece is in flight for >= 3 k-steps before anybody waits for it.  The stream of K tiles runs across output tiles (DMA
// cursor = compute position + 2); K tiles past the end are fetched at an out-of-range offset (zero fill, no traffic).
#include "common.h"
#include "gemm256.h"
#include <utility>

namespace tlxmi {

typedef __attribute__((address_space(3))) void* lds_ptr_w4_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
static __device__ __forceinline__ void w4_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_w4_t)lds, 16, voff, 0, 0, 0);
}
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t w4_srd(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
static __device__ __forceinline__ u32x4 w4_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
}
// POL: cache policy bits of the store (gfx950: 1 = sc0, 2 = nt, 16 = sc1); 0 = write-back (the consumer launch finds the rows
// in L2 / the Infinity Cache: measured on whole forwards)
template <int POL> static __device__ __forceinline__ void w4_store16(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, POL);
}
// The 256 accumulators are a[0:255], owned by the inline-asm statements below (block b = ci * 4 + pi is a[16b : 16b + 15]): left to
// hipcc, the 16 accumulator blocks are re-assigned between the K-tile variants of the loop and every control-flow join copies
// blocks through v_accvgpr_read / _mov / _write (hundreds of instructions per K tile, and 60 - 350 spilled registers).  Every
// statement lists ALL accumulator registers as clobbers, so the compiler keeps nothing of its own in them (and the kernel
// descriptor allocates them); audit after every edit: .vgpr_spill_count 0 and no compiler v_accvgpr_* outside #ASMSTART / #ASMEND.
#define W4_A4(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define W4_AGPRS                                                                                                                   \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", W4_A4(1), W4_A4(2), W4_A4(3), W4_A4(4), W4_A4(5), W4_A4(6), W4_A4(7),  \
        W4_A4(8), W4_A4(9), W4_A4(10), W4_A4(11), W4_A4(12), W4_A4(13), W4_A4(14), W4_A4(15), W4_A4(16), W4_A4(17), W4_A4(18),     \
        W4_A4(19), W4_A4(20), W4_A4(21), W4_A4(22), W4_A4(23), W4_A4(24), "a250", "a251", "a252", "a253", "a254", "a255"
// acc[B] += a . b   (s_nop 1: the wait states between a VALU write of an operand and the MFMA that reads it — hipcc pads nothing
// inside an asm statement; an accumulate chain on the same 16 registers needs none)
template <int B> static __device__ __forceinline__ void w4_mma(u32x4 a, u32x4 b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(16 * B), "i"(16 * B + 15) : W4_AGPRS);
}
// acc[B] = a . b   (first MFMA of an output tile; the C / D operands of an MFMA are both accumulator registers or both not,
// so the bias cannot enter here as a VGPR block: it is added when the finished block leaves the accumulators)
template <int B> static __device__ __forceinline__ void w4_mma_zero(u32x4 a, u32x4 b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, 0" ::"v"(a), "v"(b), "i"(16 * B), "i"(16 * B + 15) : W4_AGPRS);
}
// acc[B] -> 16 VGPRs (the wait states after the last MFMA that wrote the block: a 16-pass MFMA's result may be read 18 states later)
template <int B> static __device__ __forceinline__ f32x16 w4_acc_read() {
    float r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;
    asm volatile("s_nop 15\n\ts_nop 3\n\t"
                 "v_accvgpr_read_b32 %0, a[%c16]\n\tv_accvgpr_read_b32 %1, a[%c16+1]\n\tv_accvgpr_read_b32 %2, a[%c16+2]\n\tv_accvgpr_read_b32 %3, a[%c16+3]\n\t"
                 "v_accvgpr_read_b32 %4, a[%c16+4]\n\tv_accvgpr_read_b32 %5, a[%c16+5]\n\tv_accvgpr_read_b32 %6, a[%c16+6]\n\tv_accvgpr_read_b32 %7, a[%c16+7]\n\t"
                 "v_accvgpr_read_b32 %8, a[%c16+8]\n\tv_accvgpr_read_b32 %9, a[%c16+9]\n\tv_accvgpr_read_b32 %10, a[%c16+10]\n\tv_accvgpr_read_b32 %11, a[%c16+11]\n\t"
                 "v_accvgpr_read_b32 %12, a[%c16+12]\n\tv_accvgpr_read_b32 %13, a[%c16+13]\n\tv_accvgpr_read_b32 %14, a[%c16+14]\n\tv_accvgpr_read_b32 %15, a[%c16+15]"
                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7), "=v"(r8), "=v"(r9), "=v"(r10), "=v"(r11),
                   "=v"(r12), "=v"(r13), "=v"(r14), "=v"(r15)
                 : "i"(16 * B)
                 : W4_AGPRS);
    return f32x16{r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15};
}
// f(IntTag<0>{}), ..., f(IntTag<N - 1>{}): the accumulator block of an MFMA statement is an immediate operand
template <typename F, int... I> __device__ __forceinline__ void w4_unroll_impl(F&& f, std::integer_sequence<int, I...>) { (f(IntTag<I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void w4_unroll(F&& f) { w4_unroll_impl(f, std::make_integer_sequence<int, N>{}); }
template <int N> __device__ __forceinline__ void w4_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ACT: TLXMI_ACT_NONE / RELU / GELU.  RES: a.res is added (before the activation).  a.scale must be null.
// DBG (tuning flavour only, timing ablations; results are wrong): 16 stores dropped, 32 no drain work at all, 128 nt stores
template <int ACT, bool RES, int DBG = 0>
__global__ __launch_bounds__(256) void gemm_w4_kernel(const Gemm256Args a) {
    constexpr int XR = 0, WR = 32768, KTB = 65536;      // regions of a K tile, bytes of a K tile
    constexpr int TAB = 2 * KTB;                        // two shift tables of 256 floats, then 4 x 256 B of dump space
    constexpr int OOB = (int)0x80000000;
    constexpr int STP = (DBG & 128) ? 2 : 0;
    constexpr int NV = ((DBG & 32) ? 0 : 4) + (RES ? 4 : 0);      // vector-memory operations of a drain / residual step (all before the barrier)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int nb = a.mtiles * a.ntiles;
    const int n_mine = ((int)blockIdx.x < nb) ? (nb - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int ks = a.ksteps;

    // i-th tile of this workgroup -> origin (gemm_stream.hip: ids sharing an XCD walk consecutive tiles, N tiles fastest)
    auto tile_origin = [&](int i, int& bm0, int& bn0) -> bool {
        if (i >= n_mine) return false;
        const int id = (int)blockIdx.x + i * (int)gridDim.x;
        const int xcd = id & 7, qd = nb >> 3, rm = nb & 7;
        const int L = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (id >> 3);
        const int tm = L / a.ntiles;
        bm0 = tm * 256;
        bn0 = (L - tm * a.ntiles) * 256;
        return true;
    };

    const __amdgpu_buffer_rsrc_t xsrd = w4_srd(a.x, a.x_bytes), wsrd = w4_srd(a.w, a.w_bytes);
    const __amdgpu_buffer_rsrc_t ysrd = w4_srd(a.y, a.y_bytes);
    const __amdgpu_buffer_rsrc_t rsrd = w4_srd(a.res ? a.res : a.y, a.res ? a.res_bytes : 0u);
    const __amdgpu_buffer_rsrc_t hsrd = w4_srd(a.shift, a.shift ? (unsigned)a.Cout * 4u : 0u);      // null: zero fill

    // ---- loader: a piece = 8 rows x 128 B (one wave instruction).  Wave w fills pieces w + 4j (j = 0..7) of the X region and of
    // the W region: LDS rows 32j + 8w + (lane >> 3), slot lane & 7; (row >> 1) & 7 = (4 (w & 1) + (lane >> 4)) & 7 for all of them,
    // so a lane fetches the same K chunk `lc` for its 16 pieces and piece j is a wave-uniform stride further.
    // Lane constants are (re)derived at the top of every output tile (lane_now(), common.h): nothing lane-constant lives
    // across a whole tile, where the register allocator would push it to scratch — and a scratch reload waits vmcnt(0).
    int lc, xf0, wf0, lrow;
    auto lane_consts = [&]() {
        const int ln = lane_now();
        lc = (ln & 7) ^ ((4 * (wid & 1) + (ln >> 4)) & 7);
        lrow = ln >> 3;
        // fragment reads: lane (row = ln & 31, h = ln >> 5) reads row `row` of a 32-row block, 16-byte chunk 2 * kstep + h
        const int row = ln & 31;
        const int foff = row * 128 + (((ln >> 5) ^ ((row >> 1) & 7)) << 4);      // k-step s: ^ (32 * s)
        xf0 = XR + wr * 128 * 128 + foff;
        wf0 = WR + wc * 128 * 128 + foff;
    };
    const int xs32 = 32 * a.x_ld * 2;
    int xo, wo;                                    // cursor tile: byte offsets of X row 8w + lrow / of this lane's filter row (OOB: no tile)
    auto set_rows = [&](int i) {
        int bm0 = 0, bn0 = 0;
        const bool ok = tile_origin(i, bm0, bn0);
        // LDS filter row 32j + (8w + lrow) holds channel 128 (j >> 2) + 16 (j & 3) + 64 h + 4 q + jj with 8w + lrow = 4h + 8q + jj
        const int n = 64 * ((lrow >> 2) & 1) + 4 * wid + (lrow & 3);
        xo = ok ? (bm0 + 8 * wid + lrow) * a.x_ld * 2 : OOB;
        wo = ok ? (bn0 + n) * a.Kp_bytes : OOB;
    };
    char* const lbase = smem + wid * 1024;
    // Offsets of piece 0 of both operands for the cursor's K tile (once per K tile; an offset with bit 31 set is out of range for
    // every descriptor here — all tensors are < 2 GiB — and stays so under the small positive piece strides added below)
    int xb, wb;
    auto dma_prep = [&](int kt) {
        const int q = kt * 8 + lc;
        // (branch-free on purpose: a select that hipcc turns into an exec-masked branch cuts the K tile's scheduling region in two)
        const int xok = ((q - a.kchunks) & ~xo) >> 31, wok = ((q * 16 - a.Kp_bytes) & ~wo) >> 31;      // -1: inside K and a real tile
        xb = ((xo + q * 16) & xok) | (OOB & ~xok);
        wb = ((wo + q * 16) & wok) | (OOB & ~wok);
    };
    auto dma_pair = [&](int j, int par) {
        char* b = lbase + par * KTB + j * 4096;
        w4_dma16(xsrd, b + XR, xb + j * xs32);
        w4_dma16(wsrd, b + WR, wb + (128 * (j >> 2) + 16 * (j & 3)) * a.Kp_bytes);
    };
    // shift table of tile i -> slot i & 1 (wave w brings channels 64w .. 64w + 63 with its first 16 lanes); `live` false: the same
    // instruction into the dump space at an out-of-range offset (keeps the operation count of a K tile uniform)
    auto dma_table = [&](int i, bool live) {
        int bm0 = 0, bn0 = 0;
        const bool ok = tile_origin(i, bm0, bn0) && live;
        const int ln = lane_now();
        char* dst = smem + TAB + (live ? (i & 1) * 1024 : 2048) + wid * 256;
        if (ln < 16) w4_dma16(hsrd, dst, ok ? (bn0 + 64 * wid + 4 * ln) * 4 : OOB);
    };

    // accumulators: a[0:255] (w4_mma<ci * 4 + pi>)
    u32x4 wfA[4], xfA[4], wfB[4], xfB[4];
    u32x4 pend[4][4][2];       // [pi][ci]: the finished tile in fp16, channels 64h + 16 ci .. + 15 of pixel 32 pi + px
    u32x4 rfr[2][2];           // residual B fragments in flight: [block of the step][channel group]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) pend[i][j][0] = pend[i][j][1] = u32x4{0u, 0u, 0u, 0u};

    // ---- drain: unit (pi, ci) of the PREVIOUS tile: activation, two 16-byte stores
    auto drain_unit = [&](auto u_tag, int pbm0, int pbn0, bool live) {
        constexpr int U = decltype(u_tag)::value, PI = U >> 2, CI = U & 3;
        if constexpr ((DBG & 32) != 0) {
            asm volatile("" ::"v"(pend[PI][CI][0]), "v"(pend[PI][CI][1]));
            return;
        }
        const int ln = lane_now();
        const int m = pbm0 + 128 * wr + 32 * PI + (ln & 31);
        const int ch = pbn0 + 128 * wc + 64 * (ln >> 5) + 16 * CI;
        u32x4 o[2] = {pend[PI][CI][0], pend[PI][CI][1]};
        if constexpr (ACT == TLXMI_ACT_RELU) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                half8v v = __builtin_bit_cast(half8v, o[hh]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > (half_t)0 ? v[e] : (half_t)0;
                o[hh] = __builtin_bit_cast(u32x4, v);
            }
        } else if constexpr (ACT == TLXMI_ACT_GELU) {
            // one pair at a time: each pair's input is tied to the previous pair's result by an empty asm, so the scheduler cannot
            // open all 8 polynomial chains of a unit at once (80 live registers; the kernel has ~40 to spare)
            unsigned chain = 0;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned w = o[hh][e];
                    asm volatile("" : "+v"(w) : "v"(chain));
                    const half2v h2 = __builtin_bit_cast(half2v, w);
                    const f32x2v g2 = gelu_fast2(f32x2v{(float)h2[0], (float)h2[1]});
                    const half2v r2 = half2v{(half_t)g2[0], (half_t)g2[1]};
                    chain = __builtin_bit_cast(unsigned, r2);
                    o[hh][e] = chain;
                }
            }
        }
        const int rowok = (live && m < a.M && !(DBG & 16)) ? 0 : OOB;
        const int yo = (m * a.y_ld + ch) * 2;
        w4_store16<STP>(ysrd, o[0], (ch < a.Cout ? yo : OOB) | rowok);            // Cout is a multiple of 8 on this path
        w4_store16<STP>(ysrd, o[1], (ch + 8 < a.Cout ? yo + 16 : OOB) | rowok);
    };
    // ---- residual step: B fragments of block (ci, pi) = channels 64 s + 16 ci + 8 h .. + 7 of pixel 32 pi + px, s = 0, 1
    auto res_load = [&](auto b_tag, int slot, int bm0, int bn0) {
        constexpr int B = decltype(b_tag)::value, CI = B >> 2, PI = B & 3;
        const int ln = lane_now();
        const int m = bm0 + 128 * wr + 32 * PI + (ln & 31);
        const int ch = bn0 + 128 * wc + 16 * CI + 8 * (ln >> 5);
        const int ro = (m * a.res_ld + ch) * 2;
        const int rowok = m < a.M ? 0 : OOB;
        rfr[slot][0] = w4_load16(rsrd, (ch < a.Cout ? ro : OOB) | rowok);
        rfr[slot][1] = w4_load16(rsrd, (ch + 64 < a.Cout ? ro + 128 : OOB) | rowok);
    };
    // identity slices: A[row i][k] = 1 where row i = 4 s + 8 q + j of a block holds the channel of k = 4 q + j (channel group s)
    auto res_add = [&](auto b_tag, int slot) {
        constexpr int B = decltype(b_tag)::value, CI = B >> 2, PI = B & 3;
        const int ln = lane_now();
        const int i = ln & 31, k0 = 8 * (ln >> 5);
        const int kk = 4 * (i >> 3) + (i & 3) - k0;      // this lane's one-hot position inside its 8 k values, if 0 .. 7
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bool mine = ((i >> 2) & 1) == s && kk >= 0 && kk < 8;
            u32x4 id;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                id[e] = (mine && (kk >> 1) == e) ? ((kk & 1) ? 0x3c000000u : 0x00003c00u) : 0u;      // fp16 1.0 in the high / low half
            w4_mma<CI * 4 + PI>(id, rfr[slot][s]);
        }
    };

    // ---- stream state
    int ic = 0, kc = 0;      // DMA cursor: tile ordinal, K tile inside it
    auto advance = [&]() {
        if (++kc == ks) { kc = 0; ++ic; set_rows(ic); }
    };

    // ---- prologue: K tiles 0 and 1 of the stream, the table of tile 0; fragment set A of K tile 0
    lane_consts();
    set_rows(0);
    dma_prep(kc);
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_pair(j, 0);
    dma_table(0, true);
    advance();
    dma_prep(kc);
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_pair(j, 1);
    dma_table(0, false);
    advance();
    w4_vmcnt<17>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        wfA[r] = *reinterpret_cast<const u32x4*>(smem + wf0 + r * 4096);
        xfA[r] = *reinterpret_cast<const u32x4*>(smem + xf0 + r * 4096);
    }

    int par = 0;      // buffer of the K tile being multiplied
    int bm0 = 0, bn0 = 0, pbm0 = 0, pbn0 = 0;
    bool have_prev = false;
    // One K tile.  V = 0..8: step V of the tile's drain / residual schedule; V = 9: none.
    //   V = 0      : first K tile of an output tile — the accumulators start from 0 (C operand of the first MFMAs)
    //   V = 0 .. 7 : drain units 2V, 2V + 1 of the previous tile; with a residual, load the fragments of blocks 2V, 2V + 1
    //   V = 1 .. 8 : with a residual, add blocks 2V - 2, 2V - 1 (loaded one K tile earlier)
    auto ktile = [&](auto v_tag, int i) {
        constexpr int V = decltype(v_tag)::value;
        const char* kb = smem + par * KTB;
        const char* kn = smem + (par ^ 1) * KTB;
        dma_prep(kc);      // (a few VALU instructions: they ride in the issue slots of the MFMAs)
        if constexpr (RES && V >= 1 && V <= 8) {
            res_add(IntTag<2 * V - 2>{}, 0);
            res_add(IntTag<2 * V - 1>{}, 1);
        }
        if constexpr (RES && V <= 7) {
            __builtin_amdgcn_sched_barrier(0);      // the new loads re-use the registers just consumed
            res_load(IntTag<2 * V>{}, 0, bm0, bn0);
            res_load(IntTag<2 * V + 1>{}, 1, bm0, bn0);
        }
        w4_unroll<3>([&](auto s_tag) {
            // k-step s: set (s & 1) multiplies, the other set (k-step s + 1 of this K tile) arrives.  The statements stay in
            // this order (asm volatile MFMAs, memory operations): per row of blocks 2 fragment reads, 4 MFMAs.
            constexpr int s = decltype(s_tag)::value;
            u32x4(&wf)[4] = (s & 1) ? wfB : wfA;
            u32x4(&xf)[4] = (s & 1) ? xfB : xfA;
            u32x4(&wn)[4] = (s & 1) ? wfA : wfB;
            u32x4(&xn)[4] = (s & 1) ? xfA : xfB;
            w4_unroll<4>([&](auto r_tag) {
                constexpr int r = decltype(r_tag)::value;
                wn[r] = *reinterpret_cast<const u32x4*>(kb + (wf0 ^ (32 * (s + 1))) + r * 4096);
                xn[r] = *reinterpret_cast<const u32x4*>(kb + (xf0 ^ (32 * (s + 1))) + r * 4096);
                if constexpr (V == 0 && s == 0) {
                    w4_unroll<4>([&](auto p_tag) { w4_mma_zero<r * 4 + decltype(p_tag)::value>(wf[r], xf[decltype(p_tag)::value]); });
                } else {
                    w4_unroll<4>([&](auto p_tag) { w4_mma<r * 4 + decltype(p_tag)::value>(wf[r], xf[decltype(p_tag)::value]); });
                }
                if constexpr (V <= 7 && r == 1) {
                    if constexpr (s == 1) drain_unit(IntTag<2 * V>{}, pbm0, pbn0, have_prev);
                    if constexpr (s == 2) drain_unit(IntTag<2 * V + 1>{}, pbm0, pbn0, have_prev);
                }
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads of buffer `par` are done ...
        w4_vmcnt<(V <= 7 ? NV : 0)>();                            // ... and its pieces of the next K tile have landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // k-step 3: set B multiplies, set A (k-step 0 of the next K tile) arrives, buffer `par` is refilled two K tiles ahead:
        // per row of blocks 2 fragment reads, then one LDS-DMA piece behind each MFMA
        w4_unroll<4>([&](auto r_tag) {
            constexpr int r = decltype(r_tag)::value;
            wfA[r] = *reinterpret_cast<const u32x4*>(kn + wf0 + r * 4096);
            xfA[r] = *reinterpret_cast<const u32x4*>(kn + xf0 + r * 4096);
            w4_mma<r * 4 + 0>(wfB[r], xfB[0]);
            w4_dma16(xsrd, lbase + par * KTB + (2 * r) * 4096 + XR, xb + (2 * r) * xs32);
            w4_mma<r * 4 + 1>(wfB[r], xfB[1]);
            w4_dma16(wsrd, lbase + par * KTB + (2 * r) * 4096 + WR, wb + (128 * ((2 * r) >> 2) + 16 * ((2 * r) & 3)) * a.Kp_bytes);
            w4_mma<r * 4 + 2>(wfB[r], xfB[2]);
            w4_dma16(xsrd, lbase + par * KTB + (2 * r + 1) * 4096 + XR, xb + (2 * r + 1) * xs32);
            w4_mma<r * 4 + 3>(wfB[r], xfB[3]);
            w4_dma16(wsrd, lbase + par * KTB + (2 * r + 1) * 4096 + WR, wb + (128 * ((2 * r + 1) >> 2) + 16 * ((2 * r + 1) & 3)) * a.Kp_bytes);
        });
        dma_table(ic, kc == 0);
        advance();
        par ^= 1;
    };

    for (int i = 0; i < n_mine; ++i) {
        pbm0 = bm0;
        pbn0 = bn0;
        tile_origin(i, bm0, bn0);
        if (i > 0) lane_consts();
        ktile(IntTag<0>{}, i);
        ktile(IntTag<1>{}, i);
        ktile(IntTag<2>{}, i);
        ktile(IntTag<3>{}, i);
        ktile(IntTag<4>{}, i);
        ktile(IntTag<5>{}, i);
        ktile(IntTag<6>{}, i);
        ktile(IntTag<7>{}, i);
        ktile(IntTag<8>{}, i);
        for (int kt = 9; kt < ks; ++kt) ktile(IntTag<9>{}, i);
        // the finished tile -> fp16 pending registers (the only epilogue work that is not hidden: 256 register reads, the bias, 128 conversions)
        w4_unroll<4>([&](auto c_tag) {
            constexpr int ci = decltype(c_tag)::value;
            // the bias of channels 64h + 16 ci .. + 15 (table slot i & 1: staged two K tiles before the tile began)
            const int ln = lane_now();
            const float* tb = reinterpret_cast<const float*>(smem + TAB + (i & 1) * 1024) + 128 * wc + 64 * (ln >> 5) + 16 * ci;
            f32x4 b4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) b4[e] = *reinterpret_cast<const f32x4*>(tb + 4 * e);
            w4_unroll<4>([&](auto p_tag) {
                constexpr int pi = decltype(p_tag)::value;
                const f32x16 v = w4_acc_read<ci * 4 + pi>();
                half8v h0, h1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    h0[e] = (half_t)(v[e] + b4[e >> 2][e & 3]);
                    h1[e] = (half_t)(v[8 + e] + b4[2 + (e >> 2)][e & 3]);
                }
                pend[pi][ci][0] = __builtin_bit_cast(u32x4, h0);
                pend[pi][ci][1] = __builtin_bit_cast(u32x4, h1);
            });
        });
        have_prev = true;
    }
    // ---- the last tile's drain, in the open
    if (n_mine > 0) {
        auto tail = [&](auto u_tag) { drain_unit(u_tag, bm0, bn0, true); };
        tail(IntTag<0>{}); tail(IntTag<1>{}); tail(IntTag<2>{}); tail(IntTag<3>{});
        tail(IntTag<4>{}); tail(IntTag<5>{}); tail(IntTag<6>{}); tail(IntTag<7>{});
        tail(IntTag<8>{}); tail(IntTag<9>{}); tail(IntTag<10>{}); tail(IntTag<11>{});
        tail(IntTag<12>{}); tail(IntTag<13>{}); tail(IntTag<14>{}); tail(IntTag<15>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // zero-fill DMAs of the stream's tail
}

// Preconditions as launch_gemm256 (conv_igemm.hip's dispatcher) plus gemm_w4_ok().
template <int ACT, bool RES, int DBG = 0> static int launch_w4(const Gemm256Args& a0, hipStream_t st, int cus) {
    Gemm256Args a = a0;
    a.mtiles = (a.M + 255) / 256;
    a.ntiles = (a.Cout + 255) / 256;
    a.gn = a.ntiles;
    const size_t lds = (size_t)2 * 65536 + 2 * 1024 + 4 * 256;
    const void* fn = reinterpret_cast<const void*>(&gemm_w4_kernel<ACT, RES, DBG>);
    if (int rc = raise_lds_limit(fn, (int)lds, "gemm_w4")) return rc;
    const int tiles = a.mtiles * a.ntiles;
    int grid = cus & ~7;            // one workgroup per CU; a multiple of 8 keeps a virtual block on its XCD
    if (grid < 8) grid = 8;
    if (grid > tiles) grid = tiles;
    void* args[] = {&a};
    hipError_t e = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(256), args, lds, st);
    if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "gemm_w4: HIP launch failed: %s", hipGetErrorString(e));
    return TLXMI_OK;
}

// fp16 Linear layers (no BatchNorm scale) with >= 9 K tiles: the drain / residual schedule of a tile takes its first nine
bool gemm_w4_ok(int dtype, const Gemm256Args& a) {
    if (dtype != TLXMI_F16 || a.ksteps < 9 || a.conv || a.kslices > 1 || a.scale) return false;
    if (a.res && (a.flags & TLXMI_EPI_RES_AFTER_ACT)) return false;
    if (a.act != TLXMI_ACT_NONE && a.act != TLXMI_ACT_RELU && a.act != TLXMI_ACT_GELU) return false;
    return true;
}

template <int DBG> static int launch_w4_d(const Gemm256Args& a, hipStream_t st, int cus) {
    if (a.res) {
        if (a.act == TLXMI_ACT_RELU) return launch_w4<TLXMI_ACT_RELU, true, DBG>(a, st, cus);
        if (a.act == TLXMI_ACT_GELU) return launch_w4<TLXMI_ACT_GELU, true, DBG>(a, st, cus);
        return launch_w4<TLXMI_ACT_NONE, true, DBG>(a, st, cus);
    }
    if (a.act == TLXMI_ACT_RELU) return launch_w4<TLXMI_ACT_RELU, false, DBG>(a, st, cus);
    if (a.act == TLXMI_ACT_GELU) return launch_w4<TLXMI_ACT_GELU, false, DBG>(a, st, cus);
    return launch_w4<TLXMI_ACT_NONE, false, DBG>(a, st, cus);
}

int launch_gemm_w4(int dtype, const Gemm256Args& a, hipStream_t st, int cus) {
    (void)dtype;
#ifdef TLXMI_TUNING
    switch ((int)tune_int("TLXMI_W4_DBG", 0)) {      // timing ablations (results are wrong)
        case 16: return launch_w4_d<16>(a, st, cus);
        case 32: return launch_w4_d<32>(a, st, cus);
        case 128: return launch_w4_d<128>(a, st, cus);
        default: break;
    }
#endif
    return launch_w4_d<0>(a, st, cus);
}

}  // namespace tlxmi

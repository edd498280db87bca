// Persistent 256 x 256 tile GEMM on FOUR waves — one per SIMD, each with the whole 512-register file — for the
// MFMA-bound Linear layers of ViT / Swin (reference vision_transformer.py:81-87, 112-123; swin_transformer.py:192-229):
//     Y[m][n] = act( (sum_k X[m][k] * Wp[n][k]) * scale[n] + shift[n] (+ R[m][n]) )
// Same operands, filter packing, LDS K-tile image and channel permutation as gemm_pp.hip / gemm_stream.hip.
//
// Why another schedule.  The 8-wave kernels give a wave a 128 x 64 tile in 128 accumulators and leave no register for
// anything else (238 of 256): a finished tile's epilogue has to run inside the load segments, where it stretches
// every barrier interval it touches (measured: 4 - 6 us of a 26-us tile at K = 768; fc1's GELU costs another 48 us a
// launch), and each MFMA needs one ds_read_b128 per 1.3 MFMAs.  Here a wave owns 128 x 128 outputs:
//   * 256 accumulators + two sets of 16 fragments (k-step s + 1 is read from LDS while k-step s multiplies) = 384 registers,
//     and room to spare for the epilogue of the PREVIOUS tile to drain under the MFMAs of the next one;
//   * 32 ds_read_b128 per 128 MFMAs (a quarter of the LDS traffic per FLOP);
//   * one s_barrier per K tile instead of eight; nothing is overlapped between waves — every wave interleaves its own
//     fragment reads, its 16 LDS-DMA pieces and the leftover epilogue work in the issue slots its MFMAs leave
//     (an MFMA 16x16x32 occupies the matrix pipe for 16 cycles and the wave's issue for 8 of them).
//
// K tile = 128 bytes of K (64 halves / 32 floats); in LDS: X rows 0-255 (32 KiB) then W rows 0-255 (32 KiB), 128-byte rows,
// chunk c of row r in slot c ^ ((r >> 1) & 7); two K tiles resident.  Wave w = (wr, wc) = (w >> 1, w & 1) owns pixel rows
// 128 wr .. +127 and filter rows (channels) 128 wc .. +127; acc[ci][pi] = channels 16 ci.. x pixels 16 pi.. (transposed
// product: a lane ends with 8 consecutive channels of one pixel per (ci pair, pi)).
//
// Schedule of one K tile t (buffer t & 1), per wave:
//     k-step 0: 64 MFMAs on fragment set A  |  16 ds_read_b128: k-step 1 of tile t -> set B
//     s_waitcnt lgkmcnt(0) (my reads of buffer t & 1 are done)  +  vmcnt (my DMA pieces of tile t + 1 have landed)  +  s_barrier
//     k-step 1: 64 MFMAs on set B  |  16 ds_read_b128: k-step 0 of tile t + 1 -> set A  |  16 LDS-DMA pieces: tile t + 2 -> buffer t & 1
// i.e. a buffer is refilled right after the barrier behind its last read, and a piece is in flight for >= one k-step
// (>= 1000 cycles) + the rest of the k-step it was issued in before anybody waits for it.  The K-tile stream runs across
// output tiles (DMA cursor = compute position + 2), K tiles past the end are fetched at an out-of-range offset (zero fill).
#include "common.h"
#include "gemm256.h"

namespace tlxmi {

typedef __attribute__((address_space(3))) void* lds_ptr_w4_t;
static __device__ __forceinline__ void w4_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_w4_t)lds, 16, voff, 0, 0, 0);
}
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t w4_srd(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
static __device__ __forceinline__ u32x4 w4_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
}
// POL: cache policy bits of the store (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
template <int POL> static __device__ __forceinline__ void w4_store16(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, POL);
}

__device__ __attribute__((aligned(16))) float g_w4_ones[4] = {1.f, 1.f, 1.f, 1.f};

template <typename T> struct MmaW4;
template <> struct MmaW4<half_t> {
    static constexpr int N = 1;
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, a), __builtin_bit_cast(half8v, b), c, 0, 0, 0);
    }
};
template <> struct MmaW4<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], c, 0, 0, 0);
        return c;
    }
};

template <int N> __device__ __forceinline__ void w4_vmcnt() {
    constexpr int C = N > 63 ? 63 : N;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C) : "memory");
}

// ACT: compile-time activation (TLXMI_ACT_*; GELU in fp16 = gelu_fast2, in fp32 = erff).  RES: a.res added before the activation.
// DBG (tuning flavour only, timing ablations; results are wrong): 1 no LDS-DMA in the loop, 2 no fragment reads in the loop, 4 no
// mid-tile wait + barrier, 8 no MFMAs.
template <typename T, int ACT, bool RES, int DBG = 0>
__global__ __launch_bounds__(256) void gemm_w4_kernel(const Gemm256Args a) {
    constexpr int ES = (int)sizeof(T);
    constexpr int XR = 0, WR = 32768, KTB = 65536;      // regions of a K tile, bytes of a K tile
    constexpr int OOB = (int)0x80000000;
    constexpr int SPT = 32 * (ES / 2);                  // 16-byte stores of a wave's tile per lane
    constexpr int MPF = MmaW4<T>::N;                    // MFMA instructions per fragment pair
    constexpr int STP = (DBG & 64 ? 16 : 0) | (DBG & 128 ? 2 : 0) | (DBG & 256 ? 1 : 0);      // store policy under test
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
    const __amdgpu_buffer_rsrc_t ssrd = a.scale ? w4_srd(a.scale, (unsigned)a.Cout * 4u) : w4_srd(g_w4_ones, 16u);

    // ---- loader: a piece = 8 rows x 128 B (one wave instruction).  Wave w fills pieces w + 4j (j = 0..7) of the X region and of
    // the W region: rows 32j + 8w + (lane >> 3), slot lane & 7; (row >> 1) & 7 = (4 (w & 1) + (lane >> 4)) & 7 for all of them, so
    // a lane fetches the same K chunk `lc` for its 16 pieces and piece j is a wave-uniform stride further.
    // Lane constants are (re)derived at the top of every output tile (lane_now(), common.h): nothing lane-constant then lives
    // across the epilogue, whose register demand would otherwise push them to scratch — and a scratch reload waits vmcnt(0).
    int lc, xf0, wf0;                              // this lane's K chunk; fragment offsets of the X / W region (k-step 1: ^ 64)
    int lrow_w;                                    // 8 w + (lane >> 3): this lane's row inside a group of 32
    auto lane_consts = [&]() {
        const int ln = lane_now();
        lc = (ln & 7) ^ ((4 * (wid & 1) + (ln >> 4)) & 7);
        lrow_w = 8 * wid + (ln >> 3);
        // fragment reads: lane (frow, fg) reads row frow of a 16-row sub-tile, 16-byte chunk 4 * ksub + fg
        const int frow = ln & 15, fg = ln >> 4;
        const int foff = frow * 128 + ((fg ^ ((frow >> 1) & 7)) << 4);
        xf0 = XR + wr * 128 * 128 + foff;
        wf0 = WR + wc * 128 * 128 + foff;
    };
    const int xs32 = 32 * a.x_ld * ES, ws32 = 32 * a.Kp_bytes;
    int xo, wo;                                    // cursor tile: byte offsets of X row 8w + lrow / of its filter row (OOB: no tile)
    auto set_rows = [&](int i) {
        int bm0 = 0, bn0 = 0;
        const bool ok = tile_origin(i, bm0, bn0);
        const int row = lrow_w;                    // < 32: the channel permutation acts inside groups of 32 rows
        const int n = (((row >> 2) & 3) << 3) | (((row >> 4) & 1) << 2) | (row & 3);
        xo = ok ? (bm0 + row) * a.x_ld * ES : OOB;
        wo = ok ? (bn0 + n) * a.Kp_bytes : OOB;
    };
    char* const lbase = smem + wid * 1024;
    // Offsets of piece 0 of both operands for the cursor's K tile (once per K tile; an offset with bit 31 set is out of range for
    // every descriptor here — all tensors are < 2 GiB — and stays so under the small positive piece strides added below, so the
    // pieces need no select of their own: zero fill, no memory traffic)
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
        w4_dma16(wsrd, b + WR, wb + j * ws32);
    };

    f32x4 acc[8][8];          // [ci][pi]
    u32x4 wfA[8], xfA[8], wfB[8], xfB[8];

    // ---- epilogue of a finished tile, from registers: lane (fg, px) owns channels 128 wc + 32 cp + 8 fg .. +7 of pixel rows
    // 128 wr + 16 pi + px (the filter rows are permuted so that the sub-tiles 2 cp, 2 cp + 1 give 8 neighbours)
    auto epilogue = [&](int bm0, int bn0) {
        if constexpr ((DBG & 32) != 0) {      // ablation: no epilogue at all
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
            return;
        }
        const int ln = lane_now();
        const int px = ln & 15, fq = ln >> 4;
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
            const int ch0 = bn0 + 128 * wc + 32 * cp + 8 * fq;
            const bool chok = ch0 < a.Cout;                      // Cout is a multiple of 8 on this path
            const f32x4 h0 = __builtin_bit_cast(f32x4, w4_load16(hsrd, ch0 * 4)), h1 = __builtin_bit_cast(f32x4, w4_load16(hsrd, ch0 * 4 + 16));
            const f32x4 s0 = __builtin_bit_cast(f32x4, w4_load16(ssrd, a.scale ? ch0 * 4 : 0)), s1 = __builtin_bit_cast(f32x4, w4_load16(ssrd, a.scale ? ch0 * 4 + 16 : 0));
            u32x4 rr[8][ES / 2];
            if constexpr (RES) {
#pragma unroll
                for (int pi = 0; pi < 8; ++pi) {
                    const int m = bm0 + 128 * wr + 16 * pi + px;
                    const int ro = (m < a.M && chok) ? (m * a.res_ld + ch0) * ES : OOB;
#pragma unroll
                    for (int hh = 0; hh < ES / 2; ++hh) rr[pi][hh] = w4_load16(rsrd, ro + 16 * hh);
                }
            }
#pragma unroll
            for (int pi = 0; pi < 8; ++pi) {
                const int m = bm0 + 128 * wr + 16 * pi + px;
                float v[8];
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    v[bb] = acc[2 * cp][pi][bb] * s0[bb] + h0[bb];
                    v[4 + bb] = acc[2 * cp + 1][pi][bb] * s1[bb] + h1[bb];
                }
                if constexpr (RES) {
                    if constexpr (ES == 2) {
                        const half8v hv = __builtin_bit_cast(half8v, rr[pi][0]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)hv[e];
                    } else {
                        const f32x4 r0 = __builtin_bit_cast(f32x4, rr[pi][0]), r1 = __builtin_bit_cast(f32x4, rr[pi][ES / 2 - 1]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
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
                const int yo = (m < a.M && chok && !(DBG & 16)) ? (m * a.y_ld + ch0) * ES : OOB;      // out-of-range stores are dropped
                if constexpr (ES == 2) {
                    half8v hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
                    w4_store16<STP>(ysrd, __builtin_bit_cast(u32x4, hv), yo);
                } else {
                    f32x4 f0, f1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { f0[e] = v[e]; f1[e] = v[4 + e]; }
                    w4_store16<STP>(ysrd, __builtin_bit_cast(u32x4, f0), yo);
                    w4_store16<STP>(ysrd, __builtin_bit_cast(u32x4, f1), yo + 16);
                }
            }
        }
    };

    // ---- stream state
    int ic = 0, kc = 0, cpar = 0;      // DMA cursor: tile ordinal, K tile inside it, buffer
    auto advance = [&]() {
        cpar ^= 1;
        if (++kc == ks) { kc = 0; ++ic; set_rows(ic); }
    };

    // ---- prologue: K tiles 0 and 1 of the stream; fragment set A of K tile 0
    lane_consts();
    set_rows(0);
    dma_prep(kc);
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_pair(j, 0);
    advance();
    dma_prep(kc);
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_pair(j, 1);
    advance();
    w4_vmcnt<16>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        wfA[r] = *reinterpret_cast<const u32x4*>(smem + wf0 + r * 2048);
        xfA[r] = *reinterpret_cast<const u32x4*>(smem + xf0 + r * 2048);
    }

    int par = 0;      // buffer of the K tile being multiplied
    // One K tile.  ZERO: first K tile of an output tile (the accumulators start from 0).  NST: vector-memory operations this wave
    // has issued since the DMA pieces the mid-tile wait is for (the stores of the previous tile's epilogue; else 0).
    auto ktile = [&](auto zero_tag, auto nst_tag) {
        constexpr bool ZERO = decltype(zero_tag)::value != 0;
        constexpr int NST = decltype(nst_tag)::value;
        const char* kb = smem + par * KTB;
        const char* kn = smem + (par ^ 1) * KTB;
        dma_prep(kc);      // (a few VALU instructions: they ride in the issue slots of k-step 0's MFMAs)
        // k-step 0: set A multiplies, set B (k-step 1 of this K tile) arrives
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if constexpr (!(DBG & 2)) {
                wfB[r] = *reinterpret_cast<const u32x4*>(kb + (wf0 ^ 64) + r * 2048);
                xfB[r] = *reinterpret_cast<const u32x4*>(kb + (xf0 ^ 64) + r * 2048);
            }
            if constexpr (!(DBG & 8)) {
#pragma unroll
                for (int pi = 0; pi < 8; ++pi)
                    acc[r][pi] = MmaW4<T>::run(wfA[r], xfA[pi], ZERO ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[r][pi]);
            } else {
                asm volatile("" ::"v"(wfA[r]), "v"(xfA[r]));
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);          // 2 LDS reads
            __builtin_amdgcn_sched_group_barrier(0x008, 8 * MPF, 0);    // 8 fragment pairs of MFMAs
        }
        if constexpr (!(DBG & 4)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads of buffer `par` are done ...
            w4_vmcnt<NST>();                                          // ... and its pieces of the next K tile have landed
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        // k-step 1: set B multiplies, set A (k-step 0 of the next K tile) arrives, buffer `par` is refilled two K tiles ahead
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if constexpr (!(DBG & 2)) {
                wfA[r] = *reinterpret_cast<const u32x4*>(kn + wf0 + r * 2048);
                xfA[r] = *reinterpret_cast<const u32x4*>(kn + xf0 + r * 2048);
            }
            if constexpr (!(DBG & 1)) dma_pair(r, par);
            if constexpr (!(DBG & 8)) {
#pragma unroll
                for (int pi = 0; pi < 8; ++pi) acc[r][pi] = MmaW4<T>::run(wfB[r], xfB[pi], acc[r][pi]);
            } else {
                asm volatile("" ::"v"(wfB[r]), "v"(xfB[r]));
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);          // 2 LDS reads
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);          // 2 LDS-DMA pieces (VMEM reads)
            __builtin_amdgcn_sched_group_barrier(0x008, 8 * MPF, 0);
        }
        advance();
        par ^= 1;
    };

    for (int i = 0; i < n_mine; ++i) {
        int bm0, bn0;
        tile_origin(i, bm0, bn0);
        if (i > 0) lane_consts();
        if (i == 0) ktile(IntTag<1>{}, IntTag<0>{});
        else ktile(IntTag<1>{}, IntTag<SPT>{});
        for (int kt = 1; kt < ks; ++kt) ktile(IntTag<0>{}, IntTag<0>{});
        epilogue(bm0, bn0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // zero-fill DMAs of the stream's tail
}

// Preconditions as launch_gemm256 (conv_igemm.hip's dispatcher) plus: a.ksteps = packed pitch / 128 >= 2; a residual is
// added before the activation.
template <typename T, int ACT, bool RES, int DBG = 0> static int launch_w4(const Gemm256Args& a0, hipStream_t st, int cus) {
    Gemm256Args a = a0;
    a.mtiles = (a.M + 255) / 256;
    a.ntiles = (a.Cout + 255) / 256;
    a.gn = a.ntiles;
    const size_t lds = (size_t)2 * 65536;
    const void* fn = reinterpret_cast<const void*>(&gemm_w4_kernel<T, ACT, RES, DBG>);
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

bool gemm_w4_ok(int dtype, const Gemm256Args& a) {
    if (a.ksteps < 2 || a.conv || a.kslices > 1) return false;
    if (a.res && (a.flags & TLXMI_EPI_RES_AFTER_ACT)) return false;
    if (a.act != TLXMI_ACT_NONE && a.act != TLXMI_ACT_RELU && a.act != TLXMI_ACT_GELU) return false;
    return true;
}

template <typename T, int DBG> static int launch_w4_d(const Gemm256Args& a, hipStream_t st, int cus) {
    if (a.res) {
        if (a.act == TLXMI_ACT_RELU) return launch_w4<T, TLXMI_ACT_RELU, true, DBG>(a, st, cus);
        if (a.act == TLXMI_ACT_GELU) return launch_w4<T, TLXMI_ACT_GELU, true, DBG>(a, st, cus);
        return launch_w4<T, TLXMI_ACT_NONE, true, DBG>(a, st, cus);
    }
    if (a.act == TLXMI_ACT_RELU) return launch_w4<T, TLXMI_ACT_RELU, false, DBG>(a, st, cus);
    if (a.act == TLXMI_ACT_GELU) return launch_w4<T, TLXMI_ACT_GELU, false, DBG>(a, st, cus);
    return launch_w4<T, TLXMI_ACT_NONE, false, DBG>(a, st, cus);
}

template <typename T> static int launch_w4_t(const Gemm256Args& a, hipStream_t st, int cus) {
#ifdef TLXMI_TUNING
    if constexpr (sizeof(T) == 2) {
        switch ((int)tune_int("TLXMI_W4_DBG", 0)) {      // timing ablations (results are wrong)
            case 1: return launch_w4<T, TLXMI_ACT_NONE, false, 1>(a, st, cus);
            case 4: return launch_w4<T, TLXMI_ACT_NONE, false, 4>(a, st, cus);
            case 5: return launch_w4<T, TLXMI_ACT_NONE, false, 5>(a, st, cus);
            case 7: return launch_w4<T, TLXMI_ACT_NONE, false, 7>(a, st, cus);
            case 16: return launch_w4_d<T, 16>(a, st, cus);
            case 32: return launch_w4_d<T, 32>(a, st, cus);
            case 128: return launch_w4_d<T, 128>(a, st, cus);
            default: break;
        }
    }
#endif
    return launch_w4_d<T, 0>(a, st, cus);
}

int launch_gemm_w4(int dtype, const Gemm256Args& a, hipStream_t st, int cus) {
    if (dtype == TLXMI_F16) return launch_w4_t<half_t>(a, st, cus);
    return launch_w4_t<float>(a, st, cus);
}

}  // namespace tlxmi

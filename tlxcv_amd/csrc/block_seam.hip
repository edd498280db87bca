// The seam between two ResNet bottleneck blocks as ONE launch (fp16):
//
//     y  = relu( bn3(conv3(t2)) + skip )                  the "expand" 1x1 conv of block b      (resnet.py:151-155)
//     t1 = relu( bn1'(conv1'(y)) )                        the "reduce" 1x1 conv of block b + 1  (resnet.py:143-145)
//
// Layer by layer the wide map y (256 x 56 x 56 x 256 channels = 411 MB at batch 256) crosses HBM three times per block:
// written by the expand conv, read back by the next block's reduce conv, read again as that block's skip.  Here the
// reduce conv consumes y while it is still in registers, so y is written once and read once: per seam at 56 x 56 the
// traffic drops from 1439 MB to 1028 MB (t2 103 + skip 411 + y 411 + t1 103).
//
// Both convs are 1x1, so a pixel's t1 depends on that pixel's y only: a wave owns 32 pixels (two MFMA column blocks of
// 16) and needs no other wave.  D[channel][pixel] orientation (weights are the A operand), 64 output channels of the
// expand conv per step:
//   GEMM1  acc1[ci][pw] (4 sub-tiles of 16 channels) = W3[64c .. 64c+63][:] . t2[pixels][:]
//          sub-tile ci, MFMA row i  <->  channel 64c + 16*(i >> 2) + 4*ci + (i & 3), so that the lane of pixel p and lane
//          group g ends up with the 16 CONSECUTIVE channels 64c + 16g .. + 15 of its pixel (32 contiguous bytes of y);
//   epilogue: * scale3 + shift3 + skip, ReLU, -> fp16: stored to y AND — packed 8 + 8 — exactly the B-operand fragments
//          of the next product's two k-steps (k-step s of lane group g = channels 64c + 16g + 8s .. + 7): the accumulator
//          tile is the next MFMA's operand, no LDS, no shuffle;
//   GEMM2  acc2[t][pw] += W1'[:][64c .. 64c+63] . y-chunk   (all N2 output channels, accumulated over the steps)
// and after the last step acc2 * scale1 + shift1, ReLU -> t1.
// The activation streams (t2, skip in; y, t1 out) are 16-byte-per-lane accesses whose four lane groups cover whole
// 128-byte lines; the filters are staged through LDS per step (below).
// Bound: HBM.  Algorithmic bytes per pixel: (K1 + 2*N1 + N2) * 2.
#include "common.h"
#include <stdio.h>
#include "block_seam.h"
#ifdef TLXMI_TUNING
#define TLXMI_DBG_NT(x) (x)
#else
#define TLXMI_DBG_NT(x) (false)     // product: plain stores of y and t1 (the non-temporal forms are A/B candidates only)
#endif

namespace tlxmi {

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t bs_srd(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
static __device__ __forceinline__ u32x4 bs_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
}
static __device__ __forceinline__ void bs_store16_nt(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 2);
}
static __device__ __forceinline__ void bs_store16(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 0);
}
static __device__ __forceinline__ f32x4 bs_mma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, a), __builtin_bit_cast(half8v, b), c, 0, 0, 0);
}

// Weight staging.  Every wave needs every filter element, so the filters go through LDS once per workgroup instead of
// once per wave (as A fragments straight from L2 the fragment-shaped reads — 16 rows x 64 bytes per instruction — ran at
// ~4.8 TB/s chip-wide and bound the kernel: 333 us against 280 us for the two separate convolutions at 56 x 56).  A
// step's filters are (K1/64 + N2/64) panels of [64 rows][128 bytes]: W3 rows 64c .. 64c+63 cut into 64-channel column
// blocks, and the columns 64c .. 64c+63 of W1' cut into 64-row blocks.  All 512 threads load one 16-byte chunk of every
// panel of step c + 1 (plain loads: no LDS-DMA in this kernel, so hipcc's counted waits stay exact) while step c is
// computed, and write them into the other buffer before the step's single barrier.  16-byte chunk k of panel row r sits
// at slot k ^ f(r): with the lane -> row map of the A fragments (row = 16*(i >> 2) + 4*ci + (i & 3), i = lane & 15)
//     f1(r) = ((r >> 1) & 1) | (((r >> 4) & 3) << 1)          W3 panels,  chunk = 4*(k-step & 1) + lane group
//     f2(r) = (((r >> 1) & 1) << 2) | ((r >> 5) & 1)          W1' panels, chunk = 2*lane group + k-step
// make every ds_read_b128 lane group (4 x 16 lanes: {0-3,12-15,20-27} ...) hit 16 distinct 16-byte bank slots.
static __device__ __forceinline__ int bs_f1(int r) { return ((r >> 1) & 1) | (((r >> 4) & 3) << 1); }
static __device__ __forceinline__ int bs_f2(int r) { return (((r >> 1) & 1) << 2) | ((r >> 5) & 1); }

// K1: input channels of the expand conv (t2), N2: output channels of the reduce conv (t1); N1 (channels of y) is a
// runtime multiple of 64.  PW: MFMA pixel blocks (16 pixels) per wave; 8 waves per workgroup.
// relu(acc * scale + skip + shift) of 4 channels -> 4 packed halves, 2.5 VALU instructions a value: the skip enters the FMA as
// its fp16 operand (v_fma_mix_f32), the shift is a packed fp32 add, ReLU runs on the packed halves after the one rounding
// (max(x, 0) commutes with round-to-nearest).  One rounding to fp16, as in the two-launch path.
static __device__ __forceinline__ unsigned bs_pkrelu(unsigned v) {
    unsigned r;
    asm("v_pk_max_f16 %0, %1, 0" : "=v"(r) : "v"(v));
    return r;
}
static __device__ __forceinline__ u32x2 bs_bn_skip_relu4(f32x4 acc, f32x4 sc, f32x4 sh, half4v skip) {
    f32x2 t0 = f32x2{__builtin_fmaf(acc[0], sc[0], (float)skip[0]), __builtin_fmaf(acc[1], sc[1], (float)skip[1])};
    f32x2 t1 = f32x2{__builtin_fmaf(acc[2], sc[2], (float)skip[2]), __builtin_fmaf(acc[3], sc[3], (float)skip[3])};
    t0 += f32x2{sh[0], sh[1]};
    t1 += f32x2{sh[2], sh[3]};
    const half2v p0 = half2v{(half_t)t0[0], (half_t)t0[1]}, p1 = half2v{(half_t)t1[0], (half_t)t1[1]};
    return u32x2{bs_pkrelu(__builtin_bit_cast(unsigned, p0)), bs_pkrelu(__builtin_bit_cast(unsigned, p1))};
}
// MLP form (below): gelu(acc * scale + shift) of 4 channels -> 4 packed halves (the fp16-mode GELU of the GEMM epilogues, common.h)
static __device__ __forceinline__ u32x2 bs_bn_gelu4(f32x4 acc, f32x4 sc, f32x4 sh) {
    const f32x2v g0 = gelu_fast2(f32x2v{__builtin_fmaf(acc[0], sc[0], sh[0]), __builtin_fmaf(acc[1], sc[1], sh[1])});
    const f32x2v g1 = gelu_fast2(f32x2v{__builtin_fmaf(acc[2], sc[2], sh[2]), __builtin_fmaf(acc[3], sc[3], sh[3])});
    const half2v p0 = half2v{(half_t)g0[0], (half_t)g0[1]}, p1 = half2v{(half_t)g1[0], (half_t)g1[1]};
    return u32x2{__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
}
// ... and acc * scale + shift + res, no activation
static __device__ __forceinline__ u32x2 bs_bn_res4(f32x4 acc, f32x4 sc, f32x4 sh, half4v res) {
    const float t0 = __builtin_fmaf(acc[0], sc[0], sh[0]) + (float)res[0], t1 = __builtin_fmaf(acc[1], sc[1], sh[1]) + (float)res[1];
    const float t2 = __builtin_fmaf(acc[2], sc[2], sh[2]) + (float)res[2], t3 = __builtin_fmaf(acc[3], sc[3], sh[3]) + (float)res[3];
    const half2v p0 = half2v{(half_t)t0, (half_t)t1}, p1 = half2v{(half_t)t2, (half_t)t3};
    return u32x2{__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
}
static __device__ __forceinline__ u32x2 bs_bn_relu4(f32x4 acc, f32x4 sc, f32x4 sh) {
    const f32x2 t0 = f32x2{__builtin_fmaf(acc[0], sc[0], sh[0]), __builtin_fmaf(acc[1], sc[1], sh[1])};
    const f32x2 t1 = f32x2{__builtin_fmaf(acc[2], sc[2], sh[2]), __builtin_fmaf(acc[3], sc[3], sh[3])};
    const half2v p0 = half2v{(half_t)t0[0], (half_t)t0[1]}, p1 = half2v{(half_t)t1[0], (half_t)t1[1]};
    return u32x2{bs_pkrelu(__builtin_bit_cast(unsigned, p0)), bs_pkrelu(__builtin_bit_cast(unsigned, p1))};
}

// PROJ: the skip is not a stored map but the projection shortcut bn_d(conv_d(xp)) of the block (resnet.py:246-261, a 1x1 conv
// on the block's K1-channel input at stride 1: ResNet's layer1.0), computed here as a second product per step — the 411 MB
// shortcut map is neither written nor read.
// WPS: waves per SIMD the register budget is cut for (0: the compiler's choice — with 4-wave workgroups it spreads into the
// AGPRs and a CU then holds ONE workgroup however little LDS it takes)
// MLP (round 5): the same two chained products as a transformer MLP whose hidden activations never leave the CU —
//     out = fc2(gelu(fc1(x))) + res          (swin_transformer.py:62-82 Mlp, :335 `x + mlp(norm2(x))`; stage 1 of Swin-B: 128 -> 512 -> 128)
// x = the K1-channel rows (a.x), W3 / shift3 = fc1's filter / bias, W1 / shift1 = fc2's; epilogue 1 is bias + GELU with NO store (y is
// never written) and no skip, epilogue 2 is bias + the residual rows a.res [M][N2] with no activation.  Per row 3 * 2 * K1 bytes cross
// HBM instead of (2 K1 + 4 N1) * 2: the hidden map (205 MB per block at half batch 64) is neither written nor read.
template <int K1, int N2, int PW, int NW, bool PROJ = false, int WPS = 0, bool MLP = false>
__global__ __launch_bounds__(64 * NW, WPS ? WPS : 1) void seam_kernel(const SeamArgs a) {
    static_assert(!(PROJ && MLP), "one form at a time");
    constexpr int NT = 64 * NW;          // threads
    constexpr int IPT = 512 / NT;        // 16-byte chunks of a panel each thread stages
    constexpr int KS = K1 / 32;          // k-steps of GEMM1
    constexpr int CB = K1 / 64;          // W3 panels of a step
    constexpr int Q2 = N2 / 64;          // W1' panels of a step
    constexpr int CD = PROJ ? CB : 0;    // Wd panels of a step (the shortcut's input has K1 channels too)
    constexpr int NP = CB + Q2 + CD;
    constexpr int T2 = N2 / 16;          // MFMA row tiles of GEMM2
    constexpr int PANEL = 64 * 128;
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wbuf = smem;                                                   // two step buffers of NP panels
    float* const tab3 = reinterpret_cast<float*>(smem + 2 * NP * PANEL);       // scale3[N1], shift3[N1]
    float* const tab1 = tab3 + 2 * a.N1;                                       // scale1[N2], shift1[N2]
    float* const tabd = tab1 + 2 * N2;                                         // PROJ: scale_d[N1], shift_d[N1]

    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, g = lane >> 4;

    const __amdgpu_buffer_rsrc_t xsrd = bs_srd(a.x, a.x_bytes), w3srd = bs_srd(a.w3, a.w3_bytes), w1srd = bs_srd(a.w1, a.w1_bytes);
    const __amdgpu_buffer_rsrc_t rsrd = bs_srd(a.res, a.res_bytes), ysrd = bs_srd(a.y, a.y_bytes), zsrd = bs_srd(a.z, a.z_bytes);
    const __amdgpu_buffer_rsrc_t wdsrd = bs_srd(PROJ ? a.wd : a.w3, PROJ ? a.wd_bytes : 0u);

    // staging: chunk k of this thread in every panel = (row, physical slot) of index k*NT + t; the logical chunks it fetches
    auto stage_load = [&](int c, u32x4 (&st)[NP][IPT]) {
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int idx = k * NT + t, srow = idx >> 3, sslot = idx & 7;
            const int sl1 = sslot ^ bs_f1(srow), sl2 = sslot ^ bs_f2(srow);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) st[cb][k] = bs_load16(w3srd, ((64 * c + srow) * K1 + 64 * cb + 8 * sl1) * 2);
#pragma unroll
            for (int q = 0; q < Q2; ++q) st[CB + q][k] = bs_load16(w1srd, ((64 * q + srow) * a.N1 + 64 * c + 8 * sl2) * 2);
#pragma unroll
            for (int cb = 0; cb < CD; ++cb) st[CB + Q2 + cb][k] = bs_load16(wdsrd, ((64 * c + srow) * K1 + 64 * cb + 8 * sl1) * 2);
        }
    };
    auto stage_write = [&](int buf, const u32x4 (&st)[NP][IPT]) {
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int k = 0; k < IPT; ++k) *reinterpret_cast<u32x4*>(wbuf + (buf * NP + j) * PANEL + (k * NT + t) * 16) = st[j][k];
    };

    // this wave's pixels: block pw covers pixels m0 + 16*pw + fr (waves past the end still stage filters and keep the barriers)
    const int m0 = ((int)blockIdx.x * NW + wid) * (16 * PW);
    int pix[PW];
    bool pok[PW];
#pragma unroll
    for (int pw = 0; pw < PW; ++pw) {
        pix[pw] = m0 + 16 * pw + fr;
        pok[pw] = pix[pw] < a.M;
    }

    // t2 fragments (B operand of GEMM1): lane (pixel fr, group g) holds channels 32*ks + 8g .. + 7
    u32x4 xf[KS][PW];
#pragma unroll
    for (int pw = 0; pw < PW; ++pw)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            xf[ks][pw] = bs_load16(xsrd, pok[pw] ? (pix[pw] * a.x_ld + 32 * ks + 8 * g) * 2 : OOB);
    u32x4 xpf[PROJ ? KS : 1][PW];      // PROJ: fragments of the block input (B operand of the shortcut product)
    if constexpr (PROJ) {
#pragma unroll
        for (int pw = 0; pw < PW; ++pw)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                xpf[ks][pw] = bs_load16(rsrd, pok[pw] ? (pix[pw] * a.res_ld + 32 * ks + 8 * g) * 2 : OOB);
    }
    auto skip_load = [&](int c, u32x4 (&sk)[PW][2]) {
        if constexpr (PROJ || MLP) return;
#pragma unroll
        for (int pw = 0; pw < PW; ++pw)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                sk[pw][h] = bs_load16(rsrd, pok[pw] ? (pix[pw] * a.res_ld + 64 * c + 16 * g + 8 * h) * 2 : OOB);
    };
    u32x4 sk[PW][2];
    skip_load(0, sk);
    {
        u32x4 st[NP][IPT];
        stage_load(0, st);
        for (int i = t; i < a.N1; i += NT) {
            tab3[i] = a.scale3 ? a.scale3[i] : 1.f;
            tab3[a.N1 + i] = a.shift3 ? a.shift3[i] : 0.f;
        }
        for (int i = t; i < N2; i += NT) {
            tab1[i] = a.scale1 ? a.scale1[i] : 1.f;
            tab1[N2 + i] = a.shift1 ? a.shift1[i] : 0.f;
        }
        if constexpr (PROJ) {
            for (int i = t; i < a.N1; i += NT) {
                tabd[i] = a.scale_d ? a.scale_d[i] : 1.f;
                tabd[a.N1 + i] = a.shift_d ? a.shift_d[i] : 0.f;
            }
        }
        stage_write(0, st);
    }
#ifdef TLXMI_TUNING
    if (a.debug >> 8) {      // experiment: the workgroup in the odd wave slot of its SIMD starts late (two workgroups of a CU out of phase)
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);    // HW_ID.wave_id
        if (slot & 1)
            for (int i = 0; i < (a.debug >> 8); ++i) __builtin_amdgcn_s_sleep(8);
    }
#endif
    __syncthreads();

    // A-fragment addresses inside a panel: row arow + 4*ci; W3: chunk 4*h + g (h = k-step & 1); W1': chunk 2*g + s
    const int arow = 16 * (fr >> 2) + (fr & 3);
    int a1off[4], a2off[4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
        const int r = arow + 4 * ci;
        a1off[ci] = r * 128 + ((g ^ bs_f1(r)) << 4);          // k-step parity h: XOR 64 (chunk bit 2; f1 < 8 keeps it separate)
        a2off[ci] = r * 128 + (((2 * g) ^ bs_f2(r)) << 4);    // k-step s: XOR 16 (chunk bit 0)
    }

    f32x4 acc2[T2][PW];
#pragma unroll
    for (int tt = 0; tt < T2; ++tt)
#pragma unroll
        for (int pw = 0; pw < PW; ++pw) acc2[tt][pw] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nch = a.N1 >> 6;
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        const char* const wb = wbuf + (c & 1) * NP * PANEL;
        const bool more = c + 1 < nch;
        u32x4 st[NP][IPT], skn[PW][2];
        if (more) {                                   // next step's filters and skip: in flight under this step's MFMAs
            if (!TLXMI_DBG(a, 8)) stage_load(c + 1, st);
            if (!TLXMI_DBG(a, 2)) skip_load(c + 1, skn);
        }

        // ---- GEMM1: 64 channels of the expand conv
        f32x4 acc1[4][PW];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
#pragma unroll
            for (int pw = 0; pw < PW; ++pw) acc1[ci][pw] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (TLXMI_DBG(a, 16)) continue;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 af = *reinterpret_cast<const u32x4*>(wb + (ks >> 1) * PANEL + (a1off[ci] ^ ((ks & 1) << 6)));
#pragma unroll
                for (int pw = 0; pw < PW; ++pw) acc1[ci][pw] = bs_mma(af, xf[ks][pw], acc1[ci][pw]);
            }
        }
        f32x4 accd[PROJ ? 4 : 1][PW];
        if constexpr (PROJ) {
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
#pragma unroll
                for (int pw = 0; pw < PW; ++pw) accd[ci][pw] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 af = *reinterpret_cast<const u32x4*>(wb + (CB + Q2 + (ks >> 1)) * PANEL + (a1off[ci] ^ ((ks & 1) << 6)));
#pragma unroll
                    for (int pw = 0; pw < PW; ++pw) accd[ci][pw] = bs_mma(af, xpf[ks][pw], accd[ci][pw]);
                }
            }
        }

        // ---- epilogue 1: folded BatchNorm, + skip, ReLU; y out; the rounded values are GEMM2's B fragments
        u32x4 yf[PW][2];
        {
            const float* sc = tab3 + 64 * c + 16 * g;
            const float* sh = sc + a.N1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc + 8 * h), s1 = *reinterpret_cast<const f32x4*>(sc + 8 * h + 4);
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh + 8 * h), h1 = *reinterpret_cast<const f32x4*>(sh + 8 * h + 4);
                f32x4 d0 = s0, d1 = s0, e0 = s0, e1 = s0;
                if constexpr (PROJ) {
                    const float* scd = tabd + 64 * c + 16 * g + 8 * h;
                    d0 = *reinterpret_cast<const f32x4*>(scd); d1 = *reinterpret_cast<const f32x4*>(scd + 4);
                    e0 = *reinterpret_cast<const f32x4*>(scd + a.N1); e1 = *reinterpret_cast<const f32x4*>(scd + a.N1 + 4);
                }
#pragma unroll
                for (int pw = 0; pw < PW; ++pw) {
                    half8v rv;
                    if constexpr (PROJ) {       // the shortcut as the two-launch path stores it: rounded to fp16 once
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            rv[e] = (half_t)(accd[2 * h][pw][e] * d0[e] + e0[e]);
                            rv[4 + e] = (half_t)(accd[2 * h + 1][pw][e] * d1[e] + e1[e]);
                        }
                    } else {
                        rv = __builtin_bit_cast(half8v, sk[pw][h]);
                    }
                    if constexpr (MLP) {      // bias + GELU; the hidden activations stay in registers
                        const u32x2 o0 = bs_bn_gelu4(acc1[2 * h][pw], s0, h0), o1 = bs_bn_gelu4(acc1[2 * h + 1][pw], s1, h1);
                        yf[pw][h] = u32x4{o0[0], o0[1], o1[0], o1[1]};
                        continue;
                    }
                    const u32x2 o0 = bs_bn_skip_relu4(acc1[2 * h][pw], s0, h0, half4v{rv[0], rv[1], rv[2], rv[3]});
                    const u32x2 o1 = bs_bn_skip_relu4(acc1[2 * h + 1][pw], s1, h1, half4v{rv[4], rv[5], rv[6], rv[7]});
                    const u32x4 o = u32x4{o0[0], o0[1], o1[0], o1[1]};
                    yf[pw][h] = o;
                    if (TLXMI_DBG(a, 1)) continue;
                    if (TLXMI_DBG_NT(a.y_nt)) bs_store16_nt(ysrd, yf[pw][h], pok[pw] ? (pix[pw] * a.y_ld + 64 * c + 16 * g + 8 * h) * 2 : OOB);
                    else bs_store16(ysrd, yf[pw][h], pok[pw] ? (pix[pw] * a.y_ld + 64 * c + 16 * g + 8 * h) * 2 : OOB);
                }
            }
        }

        // ---- GEMM2: this step's 64 channels are two k-steps of the reduce conv
#pragma unroll
        for (int tt = 0; tt < T2; ++tt) {
            if (TLXMI_DBG(a, 16)) continue;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const u32x4 af = *reinterpret_cast<const u32x4*>(wb + (CB + (tt >> 2)) * PANEL + (a2off[tt & 3] ^ (s << 4)));
#pragma unroll
                for (int pw = 0; pw < PW; ++pw) acc2[tt][pw] = bs_mma(af, yf[pw][s], acc2[tt][pw]);
            }
        }

        if (more) {
            stage_write((c + 1) & 1, st);
#pragma unroll
            for (int pw = 0; pw < PW; ++pw) { sk[pw][0] = skn[pw][0]; sk[pw][1] = skn[pw][1]; }
        }
        __syncthreads();      // every wave is done with this step's buffer; the next step's is written
    }

    // ---- epilogue 2: t1 = relu(acc2 * scale1 + shift1); the lane holds 16 consecutive channels per block of 64
#pragma unroll
    for (int q = 0; q < N2 / 64; ++q) {
        const float* sc = tab1 + 64 * q + 16 * g;
        const float* sh = sc + N2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc + 8 * h), s1 = *reinterpret_cast<const f32x4*>(sc + 8 * h + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh + 8 * h), h1 = *reinterpret_cast<const f32x4*>(sh + 8 * h + 4);
#pragma unroll
            for (int pw = 0; pw < PW; ++pw) {
                u32x2 o0, o1;
                if constexpr (MLP) {      // + the residual rows, no activation
                    const half8v rv = __builtin_bit_cast(half8v, bs_load16(rsrd, pok[pw] ? (pix[pw] * a.res_ld + 64 * q + 16 * g + 8 * h) * 2 : OOB));
                    o0 = bs_bn_res4(acc2[4 * q + 2 * h][pw], s0, h0, half4v{rv[0], rv[1], rv[2], rv[3]});
                    o1 = bs_bn_res4(acc2[4 * q + 2 * h + 1][pw], s1, h1, half4v{rv[4], rv[5], rv[6], rv[7]});
                } else {
                    o0 = bs_bn_relu4(acc2[4 * q + 2 * h][pw], s0, h0);
                    o1 = bs_bn_relu4(acc2[4 * q + 2 * h + 1][pw], s1, h1);
                }
                const u32x4 o = u32x4{o0[0], o0[1], o1[0], o1[1]};
                if (TLXMI_DBG(a, 4)) continue;
                if (TLXMI_DBG_NT(a.z_nt)) bs_store16_nt(zsrd, o, pok[pw] ? (pix[pw] * a.z_ld + 64 * q + 16 * g + 8 * h) * 2 : OOB);
                else bs_store16(zsrd, o, pok[pw] ? (pix[pw] * a.z_ld + 64 * q + 16 * g + 8 * h) * 2 : OOB);
            }
        }
    }
}

// The same seam for wide filters (256 -> 1024 -> 256 at 14 x 14: 1 MB of filters per 128 pixels), two waves per 32 pixels.
// With 16 pixels a wave every MFMA above needs a fresh A fragment from LDS (64 x ds_read_b128 per wave and step: as many LDS
// cycles as MFMA cycles, and they add).  Here the waves 2p and 2p + 1 share 32 pixels (two pixel blocks: every fragment read
// feeds two MFMAs) and split the CHANNELS: wave h computes the sub-tiles ci = 2h, 2h + 1 of GEMM1 — with the row permutation
// above that is exactly the 8-channel half h of every lane's 16 channels, i.e. GEMM2's k-step h — stores that half of y, hands
// it to its partner through LDS (2 KB a wave), and accumulates the row tiles 8h .. 8h + 7 of GEMM2 (channels 128h .. of t1)
// over both k-steps.  Per wave and step: 16 + 16 fragment reads for 32 + 32 MFMAs, half the skip registers; one more barrier
// per step (the exchange).
template <int K1, int N2, int NW>
__global__ __launch_bounds__(64 * NW, 1) void seam_pair_kernel(const SeamArgs a) {
    static_assert(N2 % 128 == 0 && NW % 2 == 0, "pairs of waves split N2 in halves of whole 64-channel blocks");
    constexpr int NT = 64 * NW;
    constexpr int IPT = 512 / NT;
    constexpr int PW = 2;
    constexpr int KS = K1 / 32;
    constexpr int CB = K1 / 64;
    constexpr int Q2 = N2 / 64;
    constexpr int NP = CB + Q2;
    constexpr int T2H = N2 / 32;         // row tiles of GEMM2 per wave (half of N2 / 16)
    constexpr int PANEL = 64 * 128;
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wbuf = smem;
    float* const tab3 = reinterpret_cast<float*>(smem + 2 * NP * PANEL);
    float* const tab1 = tab3 + 2 * a.N1;
    char* const ybuf = reinterpret_cast<char*>(tab1 + 2 * N2);                  // [wave][pw][64 lanes][16 bytes]

    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int hh = wid & 1, pp = wid >> 1;
    const int fr = lane & 15, g = lane >> 4;

    const __amdgpu_buffer_rsrc_t xsrd = bs_srd(a.x, a.x_bytes), w3srd = bs_srd(a.w3, a.w3_bytes), w1srd = bs_srd(a.w1, a.w1_bytes);
    const __amdgpu_buffer_rsrc_t rsrd = bs_srd(a.res, a.res_bytes), ysrd = bs_srd(a.y, a.y_bytes), zsrd = bs_srd(a.z, a.z_bytes);

    auto stage_load = [&](int c, u32x4 (&st)[NP][IPT]) {
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int idx = k * NT + t, srow = idx >> 3, sslot = idx & 7;
            const int sl1 = sslot ^ bs_f1(srow), sl2 = sslot ^ bs_f2(srow);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) st[cb][k] = bs_load16(w3srd, ((64 * c + srow) * K1 + 64 * cb + 8 * sl1) * 2);
#pragma unroll
            for (int q = 0; q < Q2; ++q) st[CB + q][k] = bs_load16(w1srd, ((64 * q + srow) * a.N1 + 64 * c + 8 * sl2) * 2);
        }
    };
    auto stage_write = [&](int buf, const u32x4 (&st)[NP][IPT]) {
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int k = 0; k < IPT; ++k) *reinterpret_cast<u32x4*>(wbuf + (buf * NP + j) * PANEL + (k * NT + t) * 16) = st[j][k];
    };

    const int m0 = ((int)blockIdx.x * (NW / 2) + pp) * (16 * PW);
    int pix[PW];
    bool pok[PW];
#pragma unroll
    for (int pw = 0; pw < PW; ++pw) {
        pix[pw] = m0 + 16 * pw + fr;
        pok[pw] = pix[pw] < a.M;
    }
    u32x4 xf[KS][PW];
#pragma unroll
    for (int pw = 0; pw < PW; ++pw)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            xf[ks][pw] = bs_load16(xsrd, pok[pw] ? (pix[pw] * a.x_ld + 32 * ks + 8 * g) * 2 : OOB);
    auto skip_load = [&](int c, u32x4 (&sk)[PW]) {
#pragma unroll
        for (int pw = 0; pw < PW; ++pw) sk[pw] = bs_load16(rsrd, pok[pw] ? (pix[pw] * a.res_ld + 64 * c + 16 * g + 8 * hh) * 2 : OOB);
    };
    u32x4 sk[PW];
    skip_load(0, sk);
    {
        u32x4 st[NP][IPT];
        stage_load(0, st);
        for (int i = t; i < a.N1; i += NT) {
            tab3[i] = a.scale3 ? a.scale3[i] : 1.f;
            tab3[a.N1 + i] = a.shift3 ? a.shift3[i] : 0.f;
        }
        for (int i = t; i < N2; i += NT) {
            tab1[i] = a.scale1 ? a.scale1[i] : 1.f;
            tab1[N2 + i] = a.shift1 ? a.shift1[i] : 0.f;
        }
        stage_write(0, st);
    }
    __syncthreads();

    const int arow = 16 * (fr >> 2) + (fr & 3);
    int a1off[2], a2off[4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = arow + 4 * (2 * hh + e);
        a1off[e] = r * 128 + ((g ^ bs_f1(r)) << 4);
    }
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
        const int r = arow + 4 * ci;
        a2off[ci] = r * 128 + (((2 * g) ^ bs_f2(r)) << 4);
    }
    char* const ymine = ybuf + (wid * PW * 64 + lane) * 16;
    const char* const ypart = ybuf + ((wid ^ 1) * PW * 64 + lane) * 16;

    f32x4 acc2[T2H][PW];
#pragma unroll
    for (int tt = 0; tt < T2H; ++tt)
#pragma unroll
        for (int pw = 0; pw < PW; ++pw) acc2[tt][pw] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nch = a.N1 >> 6;
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        const char* const wb = wbuf + (c & 1) * NP * PANEL;
        const bool more = c + 1 < nch;
        u32x4 st[NP][IPT], skn[PW];
        if (more) {
            stage_load(c + 1, st);
            skip_load(c + 1, skn);
        }
        // ---- GEMM1: this wave's 32 of the step's 64 expand channels, both pixel blocks
        f32x4 acc1[2][PW];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int pw = 0; pw < PW; ++pw) acc1[e][pw] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 af = *reinterpret_cast<const u32x4*>(wb + (ks >> 1) * PANEL + (a1off[e] ^ ((ks & 1) << 6)));
#pragma unroll
                for (int pw = 0; pw < PW; ++pw) acc1[e][pw] = bs_mma(af, xf[ks][pw], acc1[e][pw]);
            }
        }
        // ---- epilogue 1 of the half: BN, + skip, ReLU; stored to y and handed to the partner wave
        u32x4 yf[PW][2];
        {
            const float* sc = tab3 + 64 * c + 16 * g + 8 * hh;
            const float* sh = sc + a.N1;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh), h1 = *reinterpret_cast<const f32x4*>(sh + 4);
#pragma unroll
            for (int pw = 0; pw < PW; ++pw) {
                const half8v rv = __builtin_bit_cast(half8v, sk[pw]);
                const u32x2 o0 = bs_bn_skip_relu4(acc1[0][pw], s0, h0, half4v{rv[0], rv[1], rv[2], rv[3]});
                const u32x2 o1 = bs_bn_skip_relu4(acc1[1][pw], s1, h1, half4v{rv[4], rv[5], rv[6], rv[7]});
                const u32x4 o = u32x4{o0[0], o0[1], o1[0], o1[1]};
                bs_store16(ysrd, o, pok[pw] ? (pix[pw] * a.y_ld + 64 * c + 16 * g + 8 * hh) * 2 : OOB);
                *reinterpret_cast<u32x4*>(ymine + pw * 1024) = o;
                if (hh == 0) yf[pw][0] = o; else yf[pw][1] = o;
            }
        }
        __syncthreads();          // both halves of the step's y chunk are in LDS
#pragma unroll
        for (int pw = 0; pw < PW; ++pw) {
            const u32x4 o = *reinterpret_cast<const u32x4*>(ypart + pw * 1024);
            if (hh == 0) yf[pw][1] = o; else yf[pw][0] = o;
        }
        // ---- GEMM2: row tiles 8 hh .. of the reduce conv, both k-steps of the step's 64 channels
#pragma unroll
        for (int tt = 0; tt < T2H; ++tt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const u32x4 af = *reinterpret_cast<const u32x4*>(wb + (CB + hh * (Q2 / 2) + (tt >> 2)) * PANEL + (a2off[tt & 3] ^ (s << 4)));
#pragma unroll
                for (int pw = 0; pw < PW; ++pw) acc2[tt][pw] = bs_mma(af, yf[pw][s], acc2[tt][pw]);
            }
        }
        if (more) {
            stage_write((c + 1) & 1, st);
#pragma unroll
            for (int pw = 0; pw < PW; ++pw) sk[pw] = skn[pw];
        }
        __syncthreads();          // done with this step's filter buffer and exchange slots; the next step's filters are written
    }

    // ---- epilogue 2: this wave's half of t1
#pragma unroll
    for (int q = 0; q < Q2 / 2; ++q) {
        const float* sc = tab1 + 64 * (hh * (Q2 / 2) + q) + 16 * g;
        const float* sh = sc + N2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc + 8 * h), s1 = *reinterpret_cast<const f32x4*>(sc + 8 * h + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh + 8 * h), h1 = *reinterpret_cast<const f32x4*>(sh + 8 * h + 4);
#pragma unroll
            for (int pw = 0; pw < PW; ++pw) {
                const u32x2 o0 = bs_bn_relu4(acc2[4 * q + 2 * h][pw], s0, h0), o1 = bs_bn_relu4(acc2[4 * q + 2 * h + 1][pw], s1, h1);
                const u32x4 o = u32x4{o0[0], o0[1], o1[0], o1[1]};
                bs_store16(zsrd, o, pok[pw] ? (pix[pw] * a.z_ld + 64 * (hh * (Q2 / 2) + q) + 16 * g + 8 * h) * 2 : OOB);
            }
        }
    }
}

template <int K1, int N2, int NW> static int launch_seam_pair_t(const SeamArgs& a, hipStream_t st) {
    constexpr int NP = K1 / 64 + N2 / 64;
    const size_t lds = (size_t)2 * NP * 64 * 128 + (size_t)(2 * a.N1 + 2 * N2) * sizeof(float) + (size_t)NW * 2 * 1024;
    if (lds > 160 * 1024) return fail(TLXMI_ERR_UNSUPPORTED, "block_seam: %zu bytes of LDS", lds);
    const void* fn = reinterpret_cast<const void*>(&seam_pair_kernel<K1, N2, NW>);
    if (int rc = raise_lds_limit(fn, 160 * 1024, "block_seam")) return rc;
    const long grid = ((long)a.M + NW * 16 - 1) / (NW * 16);
    if (grid >= (1l << 31)) return fail(TLXMI_ERR_UNSUPPORTED, "block_seam: too many rows");
    SeamArgs b = a;
    void* args[] = {&b};
    hipError_t e = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(64 * NW), args, lds, st);
    if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "block_seam: HIP launch failed: %s", hipGetErrorString(e));
    return TLXMI_OK;
}

template <int K1, int N2, int PW, int NW, bool PROJ = false, int WPS = 0, bool MLP = false> static int launch_seam_t(const SeamArgs& a, hipStream_t st) {
    constexpr int NP = K1 / 64 + N2 / 64 + (PROJ ? K1 / 64 : 0);
    const size_t lds = (size_t)2 * NP * 64 * 128 + (size_t)(2 * a.N1 + 2 * N2 + (PROJ ? 2 * a.N1 : 0)) * sizeof(float);
    if (lds > 160 * 1024) return fail(TLXMI_ERR_UNSUPPORTED, "block_seam: %zu bytes of LDS", lds);
    const void* fn = reinterpret_cast<const void*>(&seam_kernel<K1, N2, PW, NW, PROJ, WPS, MLP>);
    if (lds > 64 * 1024)
        if (int rc = raise_lds_limit(fn, 160 * 1024, "block_seam")) return rc;
    const long grid = ((long)a.M + NW * 16 * PW - 1) / (NW * 16 * PW);
    if (grid >= (1l << 31)) return fail(TLXMI_ERR_UNSUPPORTED, "block_seam: too many rows");
#ifdef TLXMI_TUNING
    if (tune_int("TLXMI_SEAM_OCC", 0)) {
        int nb = -1;
        hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * NW, lds);
        fprintf(stderr, "seam<%d,%d,%d,%d,%d>: lds %zu, grid %ld, occupancy %d workgroups / CU (%s)\n", K1, N2, PW, NW, (int)PROJ, lds, grid, nb, hipGetErrorString(oe));
    }
#endif
    SeamArgs b = a;
    void* args[] = {&b};
    hipError_t e = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(64 * NW), args, lds, st);
    if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "block_seam: HIP launch failed: %s", hipGetErrorString(e));
    return TLXMI_OK;
}

bool block_seam_shape_ok(int K1, int N1, int N2) {
    if (N1 % 64 || N1 < 64 || N1 > 2048) return false;
    if (!((K1 == 64 && (N2 == 64 || N2 == 128)) || (K1 == 128 && (N2 == 128 || N2 == 256)) || (K1 == 256 && N2 == 256))) return false;
    // the same LDS arithmetic as the launchers below (two buffers of K1/64 + N2/64 filter panels, the scale / shift tables of both
    // convolutions, and — wave-pair form of 256 -> N1 -> 256 — 2 KiB of exchange slots per wave): "supported" must never
    // answer yes to a launch that is then refused (256 -> 2048 -> 256 needs 165888 bytes; found by tests/test_seam_gpu.py)
    const size_t lds = (size_t)2 * (K1 / 64 + N2 / 64) * 64 * 128 + (size_t)(2 * N1 + 2 * N2) * sizeof(float) + (K1 == 256 ? (size_t)8 * 2 * 1024 : 0);
    return lds <= 160 * 1024;
}

// the MLP form (a.y == nullptr): 128 -> N1 -> 128 (Swin-B stage 1)
int launch_mlp_seam(const SeamArgs& a0, int K1, int N2, hipStream_t st) {
    SeamArgs b = a0;
    b.z_nt = b.y_nt = 0;
    b.debug = 0;
    if (K1 == 128 && N2 == 128) return launch_seam_t<128, 128, 2, 8, false, 0, true>(b, st);
    return fail(TLXMI_ERR_UNSUPPORTED, "mlp_seam: no instantiation for %d -> N1 -> %d channels", K1, N2);
}

int launch_block_seam(const SeamArgs& a0, int K1, int N2, hipStream_t st) {
    const int vv = (int)tune_int("TLXMI_SEAM", 0);     // tuning flavour: bit 0 = the other workgroup shape, bit 1 / 2 = flip the t1 / y store policy (A/B)
    const int v = vv & 1;
    SeamArgs b = a0;
    b.z_nt = (vv & 2) ? 1 : 0;        // t1 is small and read back at once by the next conv: a plain store keeps it in the Infinity Cache (-124 us per ResNet-50 forward vs nt)
    b.y_nt = (vv & 4) ? 1 : 0;
    b.debug = (int)tune_int("TLXMI_SEAM_DBG", 0);      // ablation bits (tuning flavour only; see the kernel)
    const SeamArgs& a = b;
    if (a.wd) {
        if (K1 == 64 && N2 == 64) return launch_seam_t<64, 64, 2, 4, true>(a, st);
        return fail(TLXMI_ERR_UNSUPPORTED, "block_seam: the projection-shortcut form is compiled for 64 -> N1 -> 64 channels");
    }
    if (vv & 16) {      // A/B: 4-wave workgroups with the register budget of 2 / 3 waves per SIMD (2 / 3 workgroups per CU)
        if (K1 == 64 && N2 == 64) return launch_seam_t<64, 64, 2, 4, false, 3>(a, st);
        if (K1 == 64 && N2 == 128) return launch_seam_t<64, 128, 2, 4, false, 3>(a, st);
        if (K1 == 128 && N2 == 128) return launch_seam_t<128, 128, 2, 4, false, 2>(a, st);
    }
    // (64 -> 256 -> 64 with the register budget of 3 waves per SIMD — three workgroups a CU instead of two — is 224 -> 210 us in
    //  the micro-benchmark and nothing in the ResNet-50 forward, 3.76 vs 3.78 ms: the compiler's choice stays)
    if (K1 == 64 && N2 == 64) return v ? launch_seam_t<64, 64, 2, 8>(a, st) : launch_seam_t<64, 64, 2, 4>(a, st);
    if (K1 == 64 && N2 == 128) return v ? launch_seam_t<64, 128, 2, 8>(a, st) : launch_seam_t<64, 128, 2, 4>(a, st);
    // (measured, batch 256: 56 x 56 seams 226 / 268 us with 4 waves vs 233 / 288 with 8; 28 x 28 seams 153 / 194 us with 8 waves
    //  vs 185 / 266 with 4 — the 128-channel filters are 256 - 384 KB per pass and want more pixels per staging)
    if (K1 == 128 && N2 == 128) return v ? launch_seam_t<128, 128, 2, 4>(a, st) : launch_seam_t<128, 128, 2, 8>(a, st);
    if (K1 == 128 && N2 == 256) return v ? launch_seam_t<128, 256, 1, 4>(a, st) : launch_seam_t<128, 256, 1, 8>(a, st);
    if (K1 == 256 && N2 == 256) {
        if (!(vv & 32)) return launch_seam_pair_t<256, 256, 8>(a, st);      // TLXMI_SEAM bit 5 (A/B): the one-wave-per-16-pixels form
        return v ? launch_seam_t<256, 256, 1, 4>(a, st) : launch_seam_t<256, 256, 1, 8>(a, st);
    }
    return fail(TLXMI_ERR_UNSUPPORTED, "block_seam: no instantiation for %d -> N1 -> %d channels", K1, N2);
}

}  // namespace tlxmi

using namespace tlxmi;

extern "C" int tlxmi_bottleneck_seam_supported(int dtype, int K1, int N1, int N2) {
    return dtype == TLXMI_F16 && block_seam_shape_ok(K1, N1, N2) ? 1 : 0;
}

static int seam_impl(const tlxmi_seam_desc* d, const void* t2, const void* w3_packed, const float* scale3,
                     const float* shift3, const void* skip, void* y, const void* w1_packed, const float* scale1,
                     const float* shift1, void* t1, void* stream, const void* wd_packed, const float* scale_d, const float* shift_d) {
    TLXMI_REQUIRE(d && t2 && w3_packed && skip && y && w1_packed && t1, TLXMI_ERR_BAD_ARG, "bottleneck_seam: null argument");
    TLXMI_REQUIRE(d->dtype == TLXMI_F16, TLXMI_ERR_UNSUPPORTED, "bottleneck_seam: fp16 only (the fp32 parity mode runs the two convolutions)");
    TLXMI_REQUIRE(d->rows > 0 && d->K1 > 0 && d->N1 > 0 && d->N2 > 0, TLXMI_ERR_BAD_ARG, "bottleneck_seam: bad extent");
    if (!block_seam_shape_ok(d->K1, d->N1, d->N2))
        return fail(TLXMI_ERR_UNSUPPORTED, "bottleneck_seam: no kernel for %d -> %d -> %d channels", d->K1, d->N1, d->N2);
    TLXMI_REQUIRE(d->t2_ld >= d->K1 && d->skip_ld >= (wd_packed ? d->K1 : d->N1) && d->y_ld >= d->N1 && d->t1_ld >= d->N2, TLXMI_ERR_BAD_ARG, "bottleneck_seam: row stride below the channel count");
    TLXMI_REQUIRE(!wd_packed || (aligned16(wd_packed) && d->K1 == 64 && d->N2 == 64), TLXMI_ERR_UNSUPPORTED, "bottleneck_seam_proj: 64 -> N1 -> 64 channels only");
    TLXMI_REQUIRE(d->t2_ld % 8 == 0 && d->skip_ld % 8 == 0 && d->y_ld % 8 == 0 && d->t1_ld % 8 == 0 && aligned16(t2) && aligned16(skip) && aligned16(y) &&
                      aligned16(t1) && aligned16(w3_packed) && aligned16(w1_packed),
                  TLXMI_ERR_ALIGNMENT, "bottleneck_seam: rows must be whole 16-byte chunks");
    TLXMI_REQUIRE(d->act == TLXMI_ACT_RELU, TLXMI_ERR_UNSUPPORTED, "bottleneck_seam: ReLU after both convolutions only");
    const long long rows = d->rows;
    const long long big = (1ll << 31);
    TLXMI_REQUIRE(rows * d->t2_ld * 2 < big && rows * d->skip_ld * 2 < big && rows * d->y_ld * 2 < big && rows * d->t1_ld * 2 < big && rows < big,
                  TLXMI_ERR_UNSUPPORTED, "bottleneck_seam: a tensor exceeds the 2 GiB the 32-bit buffer offsets address");
    SeamArgs a;
    a.x = (const char*)t2; a.w3 = (const char*)w3_packed; a.res = (const char*)skip; a.w1 = (const char*)w1_packed;
    a.y = (char*)y; a.z = (char*)t1;
    a.scale3 = scale3; a.shift3 = shift3; a.scale1 = scale1; a.shift1 = shift1;
    a.M = (int)rows; a.N1 = d->N1;
    a.x_ld = d->t2_ld; a.res_ld = d->skip_ld; a.y_ld = d->y_ld; a.z_ld = d->t1_ld;
    a.x_bytes = (unsigned)(rows * d->t2_ld * 2); a.res_bytes = (unsigned)(rows * d->skip_ld * 2);
    a.y_bytes = (unsigned)(rows * d->y_ld * 2); a.z_bytes = (unsigned)(rows * d->t1_ld * 2);
    // packed filters (tlxmi_pack_filter, 1x1): [Cout rounded up to 128][K] row-major; K1 and N1 are multiples of 64
    a.wd = (const char*)wd_packed; a.scale_d = scale_d; a.shift_d = shift_d;
    a.wd_bytes = wd_packed ? (unsigned)(((size_t)(d->N1 + 127) / 128 * 128) * (size_t)d->K1 * 2) : 0u;
    a.w3_bytes = (unsigned)(((size_t)(d->N1 + 127) / 128 * 128) * (size_t)d->K1 * 2);
    a.w1_bytes = (unsigned)(((size_t)(d->N2 + 127) / 128 * 128) * (size_t)d->N1 * 2);
    const int rc = launch_block_seam(a, d->K1, d->N2, as_stream(stream));
    if (rc != TLXMI_OK) return rc;
    return check_launch("bottleneck_seam");
}

extern "C" int tlxmi_bottleneck_seam(const tlxmi_seam_desc* d, const void* t2, const void* w3_packed, const float* scale3,
                                     const float* shift3, const void* skip, void* y, const void* w1_packed, const float* scale1,
                                     const float* shift1, void* t1, void* stream) {
    return seam_impl(d, t2, w3_packed, scale3, shift3, skip, y, w1_packed, scale1, shift1, t1, stream, nullptr, nullptr, nullptr);
}

// The same with the block's projection shortcut computed in place of a stored skip map: skip = (x . Wd^T) * scale_d + shift_d,
// rounded to fp16 as the stand-alone convolution would store it (resnet.py:246-261 downsample of layer1.0: 1x1, stride 1).
// x: [rows][skip_ld] with K1 channels (the block's input).  fp16, K1 = N2 = 64.
extern "C" int tlxmi_bottleneck_seam_proj(const tlxmi_seam_desc* d, const void* t2, const void* w3_packed, const float* scale3,
                                          const float* shift3, const void* x, const void* wd_packed, const float* scale_d,
                                          const float* shift_d, void* y, const void* w1_packed, const float* scale1,
                                          const float* shift1, void* t1, void* stream) {
    TLXMI_REQUIRE(wd_packed, TLXMI_ERR_BAD_ARG, "bottleneck_seam_proj: null shortcut filter");
    return seam_impl(d, t2, w3_packed, scale3, shift3, x, y, w1_packed, scale1, shift1, t1, stream, wd_packed, scale_d, shift_d);
}

// A transformer MLP as ONE launch (round 5): out = fc2(gelu(fc1(x) + b1)) + b2 + res, the hidden activations never leave the CU
// (swin_transformer.py:62-82, :335; the seam kernel above in its MLP form).  fp16; x [rows][x_ld] with K channels, res / out [rows][ld]
// with N = K output channels; hidden = the middle width (a multiple of 64).  Compiled for K = N = 128 (stage 1 of Swin-B);
// tlxmi_mlp_seam_supported asks first.
extern "C" int tlxmi_mlp_seam_supported(int dtype, int K, int hidden, int N) {
    return (dtype == TLXMI_F16 && K == 128 && N == 128 && hidden % 64 == 0 && hidden >= 64 && hidden <= 2048 && block_seam_shape_ok(K, hidden, N)) ? 1 : 0;
}

extern "C" int tlxmi_mlp_seam(int dtype, int64_t rows, int K, int hidden, int N, const void* x, int x_ld, const void* w1_packed, const float* bias1,
                              const void* w2_packed, const float* bias2, const void* res, int res_ld, void* out, int out_ld, void* stream) {
    TLXMI_REQUIRE(x && w1_packed && w2_packed && res && out, TLXMI_ERR_BAD_ARG, "mlp_seam: null argument");
    if (!tlxmi_mlp_seam_supported(dtype, K, hidden, N)) return fail(TLXMI_ERR_UNSUPPORTED, "mlp_seam: no kernel for %d -> %d -> %d", K, hidden, N);
    TLXMI_REQUIRE(rows > 0 && x_ld >= K && res_ld >= N && out_ld >= N, TLXMI_ERR_BAD_ARG, "mlp_seam: bad extent");
    TLXMI_REQUIRE(x_ld % 8 == 0 && res_ld % 8 == 0 && out_ld % 8 == 0 && aligned16(x) && aligned16(res) && aligned16(out) && aligned16(w1_packed) && aligned16(w2_packed),
                  TLXMI_ERR_ALIGNMENT, "mlp_seam: rows must be whole 16-byte chunks");
    const long long big = (1ll << 31);
    TLXMI_REQUIRE(rows * x_ld * 2 < big && rows * res_ld * 2 < big && rows * out_ld * 2 < big, TLXMI_ERR_UNSUPPORTED, "mlp_seam: a tensor exceeds 2 GiB");
    SeamArgs a;
    a.x = (const char*)x; a.w3 = (const char*)w1_packed; a.res = (const char*)res; a.w1 = (const char*)w2_packed;
    a.y = nullptr; a.z = (char*)out;
    a.scale3 = nullptr; a.shift3 = bias1; a.scale1 = nullptr; a.shift1 = bias2;
    a.M = (int)rows; a.N1 = hidden;
    a.x_ld = x_ld; a.res_ld = res_ld; a.y_ld = 0; a.z_ld = out_ld;
    a.x_bytes = (unsigned)(rows * x_ld * 2); a.res_bytes = (unsigned)(rows * res_ld * 2); a.y_bytes = 0; a.z_bytes = (unsigned)(rows * out_ld * 2);
    a.wd = nullptr; a.scale_d = a.shift_d = nullptr; a.wd_bytes = 0;
    a.w3_bytes = (unsigned)(((size_t)(hidden + 127) / 128 * 128) * (size_t)K * 2);
    a.w1_bytes = (unsigned)(((size_t)(N + 127) / 128 * 128) * (size_t)hidden * 2);
    a.y_nt = a.z_nt = 0; a.debug = 0;
    const int rc = launch_mlp_seam(a, K, N, as_stream(stream));
    if (rc != TLXMI_OK) return rc;
    return check_launch("mlp_seam");
}

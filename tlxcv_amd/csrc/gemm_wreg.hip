// Linear / 1x1 convolution with a SHORT reduction axis (K = 128 input channels, fp16): the filter lives in registers, the
// activations stream through once.
//
// Swin-B's first stage (swin_transformer.py:192-229, 37-50: qkv 128 -> 384, proj 128 -> 128, fc1 128 -> 512 on 401 408 tokens at
// batch 128) and the other K = 128 pointwise layers are HBM-bound, and the tiled GEMM kernels pay for their generality there: a
// 256 x 128 tile is two K tiles of MFMA work against an epilogue of the same size, and the activation rows are re-read once per
// column tile (fc1: 4 x; 2.9 TB/s algorithmic).  Here a workgroup covers ALL N output channels: wave (column group cg, row
// group rg) keeps the filter rows of its 64 channels as MFMA A fragments in registers for the whole launch (64 channels x 128 K
// x 2 B = 64 VGPRs) and the workgroup walks row tiles: the X rows of a tile arrive once by LDS-DMA (double-buffered: the next
// tile lands while this one is computed), every wave of a row group reads them as B fragments (XOR-swizzled 256-byte rows:
// conflict-free ds_read_b128), 16 MFMAs per 16-row block, and the epilogue (scale / bias, activation) ends in
// two 16-byte stores per lane whose four lane groups cover 64 contiguous bytes of the row each (an MFMA row <-> channel
// permutation gives a lane 8 consecutive channels of each 32-channel half).  X is read once, Y written once.
// Bound: HBM.  Algorithmic bytes per row: (K + N) * 2.  Plain (cacheable) stores: with the non-temporal policy and 16-byte pieces
// at a 32-byte stride the same kernel ran 131 us on Swin-B's qkv (80.5 us as it is: 5.1 TB/s).
#include "common.h"
#include "gemm256.h"

namespace tlxmi {

typedef __attribute__((address_space(3))) void* wr_lds_ptr_t;
static __device__ __forceinline__ void wr_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (wr_lds_ptr_t)lds, 16, voff, 0, 0, 0);
}
#ifndef WR_AUX
#define WR_AUX 0        // store policy of the output rows (0 plain, 2 non-temporal)
#endif
static __device__ __forceinline__ f32x4 wr_mma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, a), __builtin_bit_cast(half8v, b), c, 0, 0, 0);
}

// K: input channels (128 or 256); NCG: column groups of 64 output channels (N = 64 * NCG); NRG: row groups; a wave = (cg, rg) owns
// 64 channels x PB 16-row blocks of every tile; tile = 16 * PB * NRG rows.
// Registers: 64 filter + 16 X fragments + 16 accumulators + the epilogue values = 152 VGPRs (three waves per SIMD; the budget of
// four spills).  No residual input: with loads in flight next to the LDS-DMAs hipcc drains vmcnt(0) before every DMA
// (pitfall 2) and the layer runs slower than on the tiled kernels, which keep those layers.
template <int K, int NCG, int NRG, int PB>
__global__ __launch_bounds__(64 * NCG * NRG, K == 128 ? 3 : 2) void gemm_wreg_kernel(const Gemm256Args a, const int ntiles) {
    constexpr int NW = NCG * NRG, NT = 64 * NW;
    constexpr int KS = K / 32;                  // k-steps
    constexpr int RB = K * 2;                   // bytes of an X row
    constexpr int CPR = RB / 16;                // 16-byte chunks per row
    constexpr int TR = 16 * PB * NRG;           // rows per tile
    constexpr int TBYTES = TR * RB;
    constexpr int OOB = (int)0x80000000;
    constexpr int N = 64 * NCG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const tab = reinterpret_cast<float*>(smem + 2 * TBYTES);      // scale[N], shift[N]

    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int cg = wid % NCG, rg = wid / NCG;
    const int cbase = (int)blockIdx.y * N;      // wide layers: column slices of N channels as blockIdx.y (X is re-read per slice, from L2)
    const int fr = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);

    // X rows of a tile -> LDS: row r at r * RB, slot s holds chunk s ^ (r & 15) (low four bits): the 16 lanes of a
    // ds_read_b128 group (16 rows, one chunk) hit 16 distinct 16-byte bank slots
    auto fill = [&](int tile, int buf) {
        const int m0 = tile * TR;
#pragma unroll
        for (int base = 0; base < TR * CPR; base += NT) {
            const int idx = base + t;
            if (TR * CPR % NT == 0 || idx < TR * CPR) {
                const int r = idx / CPR, s = idx % CPR;
                const int c = s ^ (r & 15);
                wr_dma16(xsrd, smem + buf * TBYTES + (base + wid * 64) * 16, m0 + r < a.M ? ((m0 + r) * a.x_ld) * 2 + c * 16 : OOB);
            }
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) fill(tile, 0);

    // the filter rows of this wave's 64 channels as A fragments: sub-tile ci, MFMA row i <-> channel
    // 64 cg + 32 (ci >> 1) + 8 (i >> 2) + 4 (ci & 1) + (i & 3): lane group g then holds channels 8g .. 8g + 7 of both 32-channel
    // halves, and each of the two stores of a 16-row block writes 64 contiguous bytes per row (four lanes x 16 bytes)
    u32x4 wf[4][KS];
    {
        const int chbase = cbase + 64 * cg + 8 * (fr >> 2) + (fr & 3);
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                wf[ci][ks] = *reinterpret_cast<const u32x4*>(a.w + (size_t)(chbase + 32 * (ci >> 1) + 4 * (ci & 1)) * a.Kp_bytes + (32 * ks + 8 * g) * 2);
    }
    for (int i = t; i < N; i += NT) {
        tab[i] = a.scale ? a.scale[cbase + i] : 1.f;
        tab[N + i] = a.shift ? a.shift[cbase + i] : 0.f;
    }
    const int ch0 = 64 * cg + 8 * g;            // this lane's channels: ch0 .. ch0 + 7 and ch0 + 32 .. ch0 + 39

    auto body = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        // gfx950 counts loads, stores and LDS-DMA in ONE in-order counter: per tile a wave issues [the next tile's DMAs] ...
        // [2 PB stores], so "all but the 2 PB youngest" at the top of the next tile = that tile's DMAs have landed while this
        // tile's stores may still be in flight (the stores are unconditional buffer stores — rows past M go to an out-of-range
        // offset and are dropped — so the count is a constant).
        for (int n = 0; tile < ntiles; tile += gridDim.x, ++n) {
            const int buf = n & 1;
            if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PB) : "memory");
            __syncthreads();
            const int mrow = tile * TR + rg * PB * 16 + fr;
            if (tile + (int)gridDim.x < ntiles) fill(tile + gridDim.x, buf ^ 1);
            const char* const xb = smem + buf * TBYTES + (rg * PB * 16 + fr) * RB;
#pragma unroll 1
            for (int pb = 0; pb < PB; ++pb) {
                const int m = mrow + 16 * pb;
                u32x4 xf[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    xf[ks] = *reinterpret_cast<const u32x4*>(xb + pb * 16 * RB + ((((4 * ks + g) ^ fr) & 15) | ((4 * ks + g) & ~15)) * 16);
                f32x4 acc[4];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    acc[ci] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) acc[ci] = wr_mma(wf[ci][ks], xf[ks], acc[ci]);
                }
                // epilogue: lane (row fr, group g) holds channels ch0 + 32 (ci >> 1) + 4 (ci & 1) + r
                float v[16];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    const int co = ch0 + 32 * (ci >> 1) + 4 * (ci & 1);
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(tab + co), sh = *reinterpret_cast<const f32x4*>(tab + N + co);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[4 * ci + r] = acc[ci][r] * sc[r] + sh[r];
                }
                if constexpr (ACT == TLXMI_ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        const f32x2v g2 = gelu_fast2(f32x2v{v[e], v[e + 1]});
                        v[e] = g2[0];
                        v[e + 1] = g2[1];
                    }
                } else if constexpr (ACT != TLXMI_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = apply_act_t<ACT>(v[e], a.act_param);
                }
                half8v o0, o1;
#pragma unroll
                for (int e = 0; e < 8; ++e) { o0[e] = (half_t)v[e]; o1[e] = (half_t)v[8 + e]; }
                const int yo = m < a.M ? (m * a.y_ld + cbase + ch0) * 2 : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), ysrd, yo, 0, WR_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), ysrd, yo, 64, WR_AUX);
            }
        }
    };
    TLXMI_DISPATCH_ACT(a.act, body);
}

// fp16 dense rows, K = 128 (one 256-byte packed filter row), N = 64 * {2 .. 8}, 16-byte aligned rows
bool gemm_wreg_ok(int dtype, const Gemm256Args& a) {
    if (dtype != TLXMI_F16 || a.conv || a.kslices > 1 || a.res) return false;
    const int n64 = a.Cout / 64;
    if (a.Cout % 64) return false;
    if (a.kchunks == 16 && a.Kp_bytes == 256) {               // K = 128
        if (n64 != 2 && n64 != 4 && n64 != 6 && n64 != 8) return false;
    } else if (a.kchunks == 32 && a.Kp_bytes == 512) {        // K = 256: 128 filter registers a wave, slices of <= 512 channels
        if (n64 != 4 && n64 != 8 && n64 != 12 && n64 != 16) return false;
    } else {
        return false;
    }
    if (a.x_ld % 8 || a.y_ld % 8) return false;
    if (((uintptr_t)a.x | (uintptr_t)a.y | (uintptr_t)a.w) & 15u) return false;
    if (a.y_bytes == 0 || (long long)a.M * a.y_ld * 2 >= (1ll << 31)) return false;
    return a.M >= 1 && (long long)a.M * a.x_ld * 2 < (1ll << 31);
}

template <int K, int NCG, int NRG, int PB> static int launch_wreg_t(const Gemm256Args& a, hipStream_t st, int slices = 1) {
    constexpr int TR = 16 * PB * NRG, NT = 64 * NCG * NRG;
    const size_t lds = (size_t)2 * TR * K * 2 + (size_t)2 * 64 * NCG * sizeof(float);
    const int ntiles = (a.M + TR - 1) / TR;
    const void* fn = reinterpret_cast<const void*>(&gemm_wreg_kernel<K, NCG, NRG, PB>);
    if (lds > 64 * 1024)
        if (int rc = raise_lds_limit(fn, 160 * 1024, "linear (filter in registers)")) return rc;
    // one persistent workgroup per CU measured best (qkv of Swin-B stage 1 at batch 128: 80.5 / 81.5 / 84.9 us with 1 / 2 / 3)
    long per_cu = tune_int("TLXMI_WREG_WGS", 1);
    long grid = ((long)device_cus() * per_cu + slices - 1) / slices;
    if (grid > ntiles) grid = ntiles;
    Gemm256Args b = a;
    int nt = ntiles;
    void* args[] = {&b, &nt};
    hipError_t e = hipLaunchKernel(fn, dim3((unsigned)grid, (unsigned)slices), dim3(NT), args, lds, st);
    if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "linear (filter in registers): HIP launch failed: %s", hipGetErrorString(e));
    return check_launch("linear (filter in registers)");
}

int launch_gemm_wreg(const Gemm256Args& a, hipStream_t st) {
    if (a.kchunks == 32) {
        switch (a.Cout / 64) {
            case 4: return launch_wreg_t<256, 4, 2, 2>(a, st);          // 256 channels
            case 8: return launch_wreg_t<256, 8, 1, 4>(a, st);          // 512
            case 12: return launch_wreg_t<256, 6, 1, 4>(a, st, 2);      // 768 (qkv of Swin-B stage 2): two slices of 384
            default: return launch_wreg_t<256, 8, 1, 4>(a, st, 2);      // 1024 (fc1): two slices of 512
        }
    }
    switch (a.Cout / 64) {
        case 2: return launch_wreg_t<128, 2, 4, 2>(a, st);      // 128 channels: 2 column groups x 4 row groups, 128-row tiles
        case 4: return launch_wreg_t<128, 4, 2, 4>(a, st);      // 256
        case 6: return launch_wreg_t<128, 6, 1, 4>(a, st);      // 384 (qkv of Swin-B stage 1)
        default: return launch_wreg_t<128, 8, 1, 4>(a, st);     // 512 (fc1)
    }
}

}  // namespace tlxmi

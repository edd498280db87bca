// 256 x 256 tile GEMM for the MFMA-bound Linear layers (ViT / Swin qkv, proj, fc1, fc2):
//     Y[m][n] = act( (sum_k X[m][k] * Wp[n][k]) * scale[n] + shift[n] (+ R[m][n]) )
// Same operands, packing and epilogue semantics as conv_igemm.hip's 1x1 path; different shape of work:
// the 128 x 128 tile needs 64 B of LDS fill per clock per CU at full MFMA rate, about twice what the
// L2 -> LDS path sustains (MI355X_MICROARCH "Indexed rows: gather into LDS": 66-73 GB/s per CU), so it
// tops out near 30 % MFMA utilisation.  A 256 x 256 tile halves the fill per FLOP.
//
//   512 threads = 8 waves as 2 (pixels) x 4 (channels); wave tile 128 x 64 = 8 x 4 MFMA sub-tiles
//   (128 accumulator registers).  K advances 64 bytes per step (32 halves); one step = 16 KiB of X rows
//   + 16 KiB of filter rows brought in by LDS-DMA (4 wave-instructions per wave), 32 MFMAs per wave.
//   LDS = ring of four 32-KiB steps (128 KiB, one workgroup per CU); the DMA runs three steps ahead
//   behind counted s_waitcnt vmcnt(8/4/0); one raw s_barrier per step.
//   LDS rows are 64 B, chunk c of row r at slot c ^ ((-(r>>2))&3) (conflict-free for ds_read_b128
//   fragment reads); applied on the DMA source side, where it is the same for every piece of a lane.
//   Epilogue straight from registers: each lane owns 8 consecutive channels of a pixel per sub-tile
//   pair (filter rows are permuted inside groups of 32 as in conv_igemm.hip), 16-byte non-temporal stores.
#include "common.h"
#include "gemm256.h"
#include <stdlib.h>

namespace tlxmi {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
static __device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds, 16, voff, 0, 0, 0);
}
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t srd(const char* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, bytes, 0x00020000);
}
static __device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
}
static __device__ __forceinline__ void buf_store16_nt(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 2);
}
static __device__ __forceinline__ void buf_store16_wb(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 0);
}

template <typename T> struct Mma256;
template <> struct Mma256<half_t> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, a), __builtin_bit_cast(half8v, b), c, 0, 0, 0);
    }
};
template <> struct Mma256<float> {
    static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
        f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], c, 0, 0, 0);
        return c;
    }
};

// BN = 256 with WGN = 4 (8 waves, ring of 4, one workgroup per CU) or BN = 128 with WGN = 2 (4 waves, ring
// of 3 x 24 KiB, two independent workgroups per CU: their barriers drift apart and one's epilogue overlaps
// the other's MFMAs).
template <typename T, int BN, int WGN, int NSLOT>
__global__ __launch_bounds__(WGN * 128, 2) void gemm256_kernel(const Gemm256Args a) {
    constexpr int ES = (int)sizeof(T);
    constexpr int BM = 256, NW = 2 * WGN;
    constexpr int WM = 128, WN = 64, PI = WM / 16, CI = WN / 16;
    constexpr int STEP = (BM + BN) * 64;          // bytes per K step
    constexpr int XPW = (BM / 16) / NW, WPW = (BN / 16) / NW, DPS = XPW + WPW;   // DMA pieces per wave and step
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);

    // block -> tile: blocks sharing an XCD (id % 8) take consecutive tiles, N tiles fastest
    int tile_m, tile_n;
    {
        const int nb = a.mtiles * a.ntiles, id = blockIdx.x;
        const int xcd = id & 7, qd = nb >> 3, rm = nb & 7;
        const int L = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (id >> 3);
        // tiles are walked in column panels of `gn` N-tiles (filter panel <= ~1.5 MB, stays in the XCD's 4 MB
        // L2 while the activation rows stream through), M fastest across a panel's rows
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
    const __amdgpu_buffer_rsrc_t xsrd = srd(a.x, a.x_bytes), wsrd = srd(a.w, a.w_bytes);

    // loader: one piece = 16 rows x 64 B; lane l -> row (l>>2), LDS slot (l&3); wave w fills pieces w, w+8
    const int lr = lane >> 2;
    const int lchunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);   // logical chunk fetched into this lane's slot
    int xo[XPW], wo[WPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int m = bm0 + 16 * (wid + NW * i) + lr;
        xo[i] = m < a.M ? m * a.x_ld * ES : OOB;
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int rho = 16 * (wid + NW * i) + lr;      // LDS row; holds channel perm(rho) (see conv_igemm.hip)
        const int n = (rho & ~31) | (((rho >> 2) & 3) << 3) | (((rho >> 4) & 1) << 2) | (rho & 3);
        wo[i] = (bn0 + n) * a.Kp_bytes;
    }
    int q = lchunk;   // chunk index along K of the next step to stage
    auto stage = [&](int slot) {
        char* b = smem + slot * STEP;
        const int d = q < a.kchunks ? q * 16 : OOB;
#pragma unroll
        for (int i = 0; i < XPW; ++i) dma16(xsrd, b + (wid + NW * i) * 1024, xo[i] + d);
#pragma unroll
        for (int i = 0; i < WPW; ++i) dma16(wsrd, b + BM * 64 + (wid + NW * i) * 1024, wo[i] + q * 16);
        q += 4;
    };

    // fragment read offsets: row (lane&15) of a 16-row sub-tile, chunk (lane>>4), swizzled
    const int frow = lane & 15, fg = lane >> 4;
    const int foff = frow * 64 + ((fg ^ ((0 - (frow >> 2)) & 3)) << 4);
    const int wave_m0 = (wid & 1) * WM, wave_n0 = (wid >> 1) * WN;   // 2 x WGN waves
    const int xfrag = wave_m0 * 64 + foff;
    const int wfrag = BM * 64 + wave_n0 * 64 + foff;

    f32x4 acc[CI][PI];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int pi = 0; pi < PI; ++pi) acc[ci][pi] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ks = a.ksteps;
    stage(0);
    if (ks > 1) stage(1);
    if (NSLOT == 4 && ks > 2) stage(2);
    // scale / shift of this block's BN channels -> LDS behind the ring (read back in the epilogue)
    float* sbuf = reinterpret_cast<float*>(smem + NSLOT * STEP);
    if (t < BN) {
        const int ch = bn0 + t < a.Cout ? bn0 + t : a.Cout - 1;
        sbuf[t] = a.scale ? a.scale[ch] : 1.f;
        sbuf[BN + t] = a.shift ? a.shift[ch] : 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS writes are done before this wave's next barrier
    }
    int slot = 0;
    for (int kt = 0; kt < ks; ++kt) {
        // DPS DMA instructions per wave and step: leave the younger steps in flight
        if (NSLOT == 4 && kt + 2 < ks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPS) : "memory");
        else if (kt + 1 < ks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + NSLOT - 1 < ks) stage(slot == 0 ? NSLOT - 1 : slot - 1);   // buffer of step kt-1: every wave is past it
        const char* b = smem + slot * STEP;
        u32x4 wf[CI], xf[PI];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) wf[ci] = *reinterpret_cast<const u32x4*>(b + wfrag + ci * 1024);
#pragma unroll
        for (int pi = 0; pi < PI; ++pi) xf[pi] = *reinterpret_cast<const u32x4*>(b + xfrag + pi * 1024);
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int pi = 0; pi < PI; ++pi) acc[ci][pi] = Mma256<T>::run(wf[ci], xf[pi], acc[ci][pi]);
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }

    // ---- epilogue from registers: lane (g, px) owns channels 32cp + 8g .. +7 of pixel 16pi + px
    const int g = lane >> 4, px = lane & 15;
    const bool res_after = (a.flags & TLXMI_EPI_RES_AFTER_ACT) != 0;
    const __amdgpu_buffer_rsrc_t ysrd = srd(a.y, a.y_bytes), rsrd = srd(a.res ? a.res : a.y, a.res ? a.res_bytes : 0u);
    auto epi = [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int cp = 0; cp < CI / 2; ++cp) {
        const int ch0 = bn0 + wave_n0 + 32 * cp + 8 * g;
        if (ch0 >= a.Cout) continue;      // Cout is a multiple of 8 on this path
        float sc[8], sf[8];
        {
            const int col = wave_n0 + 32 * cp + 8 * g;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sbuf + col), s1 = *reinterpret_cast<const f32x4*>(sbuf + col + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(sbuf + BN + col), h1 = *reinterpret_cast<const f32x4*>(sbuf + BN + col + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[e] = s0[e]; sc[4 + e] = s1[e]; sf[e] = h0[e]; sf[4 + e] = h1[e]; }
        }
        u32x4 rr[PI][ES / 2];
        if (a.res) {
#pragma unroll
            for (int pi = 0; pi < PI; ++pi) {
                const int m = bm0 + wave_m0 + pi * 16 + px;
                const int ro = m < a.M ? (m * a.res_ld + ch0) * ES : OOB;
#pragma unroll
                for (int hh = 0; hh < ES / 2; ++hh) rr[pi][hh] = buf_load16(rsrd, ro + 16 * hh);
            }
        }
#pragma unroll
        for (int pi = 0; pi < PI; ++pi) {
            const int m = bm0 + wave_m0 + pi * 16 + px;
            float v[8], rv[8];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                v[bb] = acc[2 * cp][pi][bb] * sc[bb] + sf[bb];
                v[4 + bb] = acc[2 * cp + 1][pi][bb] * sc[4 + bb] + sf[4 + bb];
            }
            if (a.res) {
                if constexpr (ES == 2) {
                    const half8v h = __builtin_bit_cast(half8v, rr[pi][0]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) rv[e] = (float)h[e];
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
            const int yo = m < a.M ? (m * a.y_ld + ch0) * ES : OOB;   // OOB stores are dropped by the range check
            if constexpr (ES == 2) {
                half8v h;
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = (half_t)v[e];
                if (!TLXMI_NT_STORES(a)) buf_store16_wb(ysrd, __builtin_bit_cast(u32x4, h), yo);
                else buf_store16_nt(ysrd, __builtin_bit_cast(u32x4, h), yo);
            } else {
                f32x4 f0, f1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { f0[e] = v[e]; f1[e] = v[4 + e]; }
                if (!TLXMI_NT_STORES(a)) buf_store16_wb(ysrd, __builtin_bit_cast(u32x4, f0), yo);
                else buf_store16_nt(ysrd, __builtin_bit_cast(u32x4, f0), yo);
                if (!TLXMI_NT_STORES(a)) buf_store16_wb(ysrd, __builtin_bit_cast(u32x4, f1), yo + 16);
                else buf_store16_nt(ysrd, __builtin_bit_cast(u32x4, f1), yo + 16);
            }
        }
    }
    };
    TLXMI_DISPATCH_ACT(a.act, epi)
}

// Called by conv_igemm.hip's dispatcher.  Preconditions (checked there): 1x1, stride 1, no padding, dense
// batch strides, Cout % 8 == 0, 16-byte aligned y / res rows, every tensor < 2 GiB.  variant 0: 256 x 256,
// variant 1: 256 x 128.
template <typename T, int BN, int WGN, int NSLOT> static int launch_v(const Gemm256Args& a0, hipStream_t st) {
    Gemm256Args a = a0;
    a.debug = (int)tune_int("TLXMI_DEBUG", 0);     // A/B bits: tuning flavour only
    a.mtiles = (a.M + 255) / 256;
    a.ntiles = (a.Cout + BN - 1) / BN;
    {
        const long budget = tune_int("TLXMI_PANEL_KB", 1l << 20) * 1024;   // default: no panels (N fastest); measured neutral
        long gn = budget / ((long)BN * a.Kp_bytes);
        a.gn = (int)(gn < 1 ? 1 : (gn > a.ntiles ? a.ntiles : gn));
    }
    const size_t lds = (size_t)NSLOT * (256 + BN) * 64 + 2 * BN * sizeof(float);
    const void* fn = reinterpret_cast<const void*>(&gemm256_kernel<T, BN, WGN, NSLOT>);
    if (int rc = raise_lds_limit(fn, (int)lds, "gemm256")) return rc;
    void* args[] = {&a};
    hipError_t e = hipLaunchKernel(fn, dim3((unsigned)(a.mtiles * a.ntiles)), dim3(WGN * 128), args, lds, st);
    if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "gemm256: HIP launch failed: %s", hipGetErrorString(e));
    return TLXMI_OK;
}

int launch_gemm256(int dtype, int variant, const Gemm256Args& a, hipStream_t st) {
    if (variant == 0) {
        if (dtype == TLXMI_F16) return launch_v<half_t, 256, 4, 4>(a, st);
        return launch_v<float, 256, 4, 4>(a, st);
    }
    if (dtype == TLXMI_F16) return launch_v<half_t, 128, 2, 3>(a, st);
    return launch_v<float, 128, 2, 3>(a, st);
}

}  // namespace tlxmi

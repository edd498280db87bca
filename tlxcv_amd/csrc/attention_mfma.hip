// fp16 fused multi-head attention on MFMA for short sequences (N <= 256, hd in {32, 64}):
// ViT-B/16 (N=197, hd=64; vision_transformer.py:112-123) and Swin windows (N=49, hd=32, +relative
// position bias +shift mask; swin_transformer.py:192-229).
//
// One workgroup (4 waves) per (batch/window, head).  K and V of that head are staged once in LDS
// (rows padded by 32 B so that both the ds_read_b128 K-fragment reads and the ds_read_b64_tr_b16
// V reads are bank-conflict free); every wave then owns whole 16-query tiles:
//   S^T = K . Q^T      v_mfma_f32_16x16x32_f16, A = K rows from LDS, B = Q rows straight from HBM
//                      -> the lane that owns query (lane&15) holds keys 16t + 4(lane>>4) + r
//   softmax            in registers; row max / sum need only two cross-lane steps (xor 16, 32)
//   O^T = V^T . P^T    the S^T accumulators, converted to fp16, ARE the B operand (keys of two
//                      16-key tiles interleaved as k-slot 8g+j <-> key 32p + 16(j>>2) + 4g + (j&3));
//                      the matching V^T A-fragments come from the row-major V image through the
//                      transposing LDS read (two ds_read_b64_tr_b16 per fragment).
// The N x N score matrix never leaves registers.  HBM traffic per (b, head): Q, K, V read once,
// O written once.
#include "common.h"

namespace tlxmi {

#ifndef ATTN_STAGE_BATCH
#define ATTN_STAGE_BATCH 4
#endif

struct AttnArgs {
    const void* qkv;
    const float* bias;
    const float* mask;
    void* out;
    int B, N, heads, hd, nW;
    float scale;
    const float* comb;   // bias + mask pre-summed and padded: [max(nW,1)][heads][NP][NP], NP = 32 * ceil(N / 32)
    int debug = 0;       // tuning flavour only (TLXMI_ATTN_DBG; results are wrong): 1 no K / V staging, 2 one query tile per wave only
    // Window map (round 5; Swin, tlxmi_attention_windows): wm_ws > 0 -> qkv and out are IMAGE-order token matrices [B / wpi][wm_H * wm_W][...]
    // and item b = img * wpi + w is the wm_ws x wm_ws window w of image img after the cyclic shift (token (iy, ix) of window (wy, wx) is
    // pixel ((wy * ws + iy + shift) % H, (wx * ws + ix + shift) % W): swin_transformer.py:316-324, and :327-333 on the way back)
    int wm_ws = 0, wm_H = 0, wm_W = 0, wm_shift = 0;
};

typedef __fp16 fp16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) fp16x4 lds_fp16x4;

// NT = number of 16-key tiles (even), keys padded to 16*NT.  ADD = how bias + mask reach the scores:
// 0 none (ViT: no code at all in the score loop), 1 summed once into an LDS table (Swin-sized windows),
// 2 read from global memory per score (generic fallback), 3 one 16-byte load per key tile from the caller's
// pre-summed, padded table (a.comb): no LDS table, no extra barrier dependency, 4 loads per lane instead of 20.
// KF = key tiles known to lie wholly inside the sequence (NT - 2 when N > 16 * (NT - 2), else 0): no padding mask there
template <int HD, int NT, int ADD, int KF>
__global__ __launch_bounds__(256, 2) void attn_mfma_kernel(const AttnArgs a) {
    constexpr int SR = HD * 2 + 32;          // padded LDS row stride in bytes (160 / 96)
    constexpr int NP = 16 * NT;              // padded key count
    constexpr int KS = HD / 32;              // k-steps of the QK^T product
    constexpr int DT = HD / 16;              // 16-wide d tiles of the output
    constexpr int CPR = HD / 8;              // 16-byte chunks per row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + NP * SR;

    const int N = a.N, heads = a.heads;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int g = lane >> 4, li = lane & 15;
    const size_t tok_ld = (size_t)3 * heads * HD;   // elements between tokens of the packed qkv
    const half_t* qbase = reinterpret_cast<const half_t*>(a.qkv) + (size_t)b * N * tok_ld + (size_t)h * HD;
    const half_t* kbase = qbase + (size_t)heads * HD;
    const half_t* vbase = qbase + (size_t)2 * heads * HD;

    // ---- Q fragments (B operand) of this wave's first query tile, fetched before the K/V staging so that
    // their HBM latency hides behind it: lane's query row, d = 32*ks + 8g .. +7
    const int nqt = (N + 15) >> 4;
    u32x4 qcur[KS];
    {
        const int query = wv * 16 + li;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qcur[ks] = u32x4{0u, 0u, 0u, 0u};
            if (query < N) qcur[ks] = *reinterpret_cast<const u32x4*>(qbase + (size_t)query * tok_ld + ks * 32 + g * 8);
        }
    }

    // Swin-sized windows: the (N, N) relative-position bias of this head and the shift mask of this window are
    // summed once into an LDS table (coalesced reads) instead of two dependent global gathers per score.  Their
    // loads are issued here, ahead of the K / V staging, so that a workgroup pays ONE memory latency before its
    // first barrier, not two (these workgroups are tiny and latency-bound: time was proportional to their count).
    constexpr int TPT = ADD == 1 ? (NP * NP + 255) / 256 : 1;      // table values per thread (upper bound)
    float tv[TPT];
    const float* bias = a.bias ? a.bias + (size_t)h * N * N : nullptr;
    const float* mask = (a.mask && a.nW > 0) ? a.mask + (size_t)(b % a.nW) * N * N : nullptr;
    if constexpr (ADD == 1) {
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
            const int i = t + u * 256;
            tv[u] = 0.f;
            if (i < N * N) tv[u] = (bias ? bias[i] : 0.f) + (mask ? mask[i] : 0.f);
        }
    }

    // ---- stage K and V (zero rows for padded keys); loads are issued in batches ahead of the LDS writes
    // (TLXMI_ATTN_BATCH, tuning flavour: loads in flight per thread and operand; all PER of them = one memory latency per workgroup)
    constexpr int ITEMS = NP * CPR, PER = (ITEMS + 255) / 256, BATCH = PER < ATTN_STAGE_BATCH ? PER : ATTN_STAGE_BATCH;
    for (int b0 = 0; b0 < (TLXMI_DBG(a, 1) ? 0 : PER); b0 += BATCH) {
        u32x4 kv[BATCH], vv[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int i = t + (b0 + u) * 256;
            const int key = i / CPR, c = i - key * CPR;
            kv[u] = u32x4{0u, 0u, 0u, 0u};
            vv[u] = u32x4{0u, 0u, 0u, 0u};
            if (i < ITEMS && key < N) {
                kv[u] = *reinterpret_cast<const u32x4*>(kbase + (size_t)key * tok_ld + c * 8);
                vv[u] = *reinterpret_cast<const u32x4*>(vbase + (size_t)key * tok_ld + c * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int i = t + (b0 + u) * 256;
            const int key = i / CPR, c = i - key * CPR;
            if (i < ITEMS) {
                *reinterpret_cast<u32x4*>(Ks + key * SR + c * 16) = kv[u];
                *reinterpret_cast<u32x4*>(Vs + key * SR + c * 16) = vv[u];
            }
        }
    }
    float* Ts = reinterpret_cast<float*>(smem + 2 * NP * SR);
    if constexpr (ADD == 1) {
#pragma unroll
        for (int u = 0; u < TPT; ++u) {
            const int i = t + u * 256;
            if (i < N * N) Ts[i] = tv[u];
        }
    }
    __syncthreads();

    half_t* obase = reinterpret_cast<half_t*>(a.out) + (size_t)b * N * heads * HD + (size_t)h * HD;

    // transposing V reads: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3
    const int vlane = (4 * g + (li >> 2)) * SR + (li & 3) * 8;
    auto load_v = [&](int pr, fp16x4 (&lo)[DT], fp16x4 (&hi)[DT]) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const char* a0 = Vs + pr * 32 * SR + vlane + dt * 32;
            lo[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(a0));
            hi[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(a0 + 16 * SR));
        }
    };
    const int klane = li * SR + g * 16;

#pragma unroll 1
    for (int qt = wv; qt < (TLXMI_DBG(a, 2) ? (nqt < 4 ? nqt : 4) : nqt); qt += 4) {
        const int query = qt * 16 + li;
        const bool qok = query < N;
        // next tile's Q rows travel while this tile computes
        u32x4 qnext[KS];
        {
            const int nq = query + 64;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                qnext[ks] = u32x4{0u, 0u, 0u, 0u};
                if (qt + 4 < nqt && nq < N)
                    qnext[ks] = *reinterpret_cast<const u32x4*>(qbase + (size_t)nq * tok_ld + ks * 32 + g * 8);
            }
        }
        // ---- scores: s[kt][r] = S[query][key = 16kt + 4g + r]; K fragments of tile kt+1 are read from LDS
        // while the MFMAs of tile kt run
        float s[NT][4];
        float mx = -INFINITY;
        f32x4 tb[ADD == 3 ? NT : 1];
        if constexpr (ADD == 3) {      // rows / columns up to npc = 32 * ceil(N / 32) exist in the padded table
            const int npc = (N + 31) & ~31;
            const float* crow = a.comb + (((size_t)(a.nW > 0 ? b % a.nW : 0) * heads + h) * npc + query) * npc + 4 * g;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
                tb[kt] = 16 * kt < npc ? *reinterpret_cast<const f32x4*>(crow + 16 * kt) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        u32x4 kf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const u32x4*>(Ks + klane + ks * 64);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            u32x4 kn[KS];
            if (kt + 1 < NT) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    kn[ks] = *reinterpret_cast<const u32x4*>(Ks + (kt + 1) * 16 * SR + klane + ks * 64);
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, kf[ks]),
                                                             __builtin_bit_cast(half8v, qcur[ks]), acc, 0, 0, 0);
            // ADD == 0 keeps the raw dot products (the scale is folded into the exponent below); key tiles that lie
            // wholly inside the sequence need no padding mask (wave-uniform test)
            const bool full_tile = kt < KF || kt * 16 + 16 <= N;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + 4 * g + r;
                float v = ADD == 0 ? acc[r] : acc[r] * a.scale;
                if constexpr (ADD == 1) {
                    if (qok && (full_tile || key < N)) v += Ts[query * N + key];
                } else if constexpr (ADD == 3) {
                    v += tb[kt][r];
                } else if constexpr (ADD == 2) {
                    if (qok && (full_tile || key < N)) {
                        if (bias) v += bias[(size_t)query * N + key];
                        if (mask) v += mask[(size_t)query * N + key];
                    }
                }
                if (kt >= KF && !full_tile && key >= N) v = -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
            if (kt + 1 < NT) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kf[ks] = kn[ks];
            }
        }
        // V fragments of the first key pair: in flight during the softmax
        fp16x4 vlo[DT], vhi[DT];
        load_v(0, vlo, vhi);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float ec = a.scale * 1.44269504088896340736f;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // ADD == 0: softmax(scale * x) = exp2((x - max x) * scale * log2 e), one FMA + v_exp_f32
                const float p = ADD == 0 ? __builtin_amdgcn_exp2f(fmaf(s[kt][r], ec, -mx * ec)) : __expf(s[kt][r] - mx);
                s[kt][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;

        // ---- O^T = V^T . P^T, V fragments of pair pr+1 read while pair pr multiplies
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pr = 0; pr < NT / 2; ++pr) {
            fp16x4 nlo[DT], nhi[DT];
            if (pr + 1 < NT / 2) load_v(pr + 1, nlo, nhi);
            half8v pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (half_t)s[2 * pr][r];
                pf[4 + r] = (half_t)s[2 * pr + 1][r];
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                half8v vf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vf[r] = (half_t)vlo[dt][r];
                    vf[4 + r] = (half_t)vhi[dt][r];
                }
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[dt], 0, 0, 0);
            }
            if (pr + 1 < NT / 2) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) { vlo[dt] = nlo[dt]; vhi[dt] = nhi[dt]; }
            }
        }
        if (qok) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                half4v ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[dt][r] * inv);
                *reinterpret_cast<half4v*>(obase + (size_t)query * heads * HD + dt * 16 + 4 * g) = ov;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qcur[ks] = qnext[ks];
    }
}

// The same computation for head dim 64 without bias / mask (ViT-B/16: 197 tokens, vision_transformer.py:112-123) with K and V
// staged by LDS-DMA (round 4).  Whole-forward ablation (tools/ab_graph.py TLXMI_ATTN_DBG): the forward loses 1.26 ms of 10.8 when
// the K / V staging of attn_mfma_kernel is removed — a workgroup there runs [load batch 1, write, load batch 2, write | barrier |
// compute]: two exposed memory latencies and 57 KB through VGPRs and ds_write_b128 before its first MFMA.  Here every wave issues
// its 14 pieces (8 keys x 128 B each, K then V) at once, straight into LDS, after its first Q fragments: one latency, no register
// round trip.  LDS-DMA writes lane-linear kilobytes, so the rows cannot be padded; they are 128 B with chunk c of row r in slot
// c ^ f(r), f(r) = ((r >> 1) & 3) << 1 (applied to the SOURCE chunk a lane fetches): conflict-free both for the ds_read_b128
// K-fragment reads (a 16-lane group holds the even slots c ^ {0,2,4,6} of one chunk family and the odd ones of the other, at both
// row parities) and for the ds_read_b64_tr_b16 V reads (a 32-lane half covers rows 4g + q, g = 0, 1: the four same-parity rows
// carry f = 0, 2, 4, 6, so {2dt, 2dt + 1} ^ f are 8 distinct slots x 2 halves x 2 parities = all 64 banks).  56 KB per
// workgroup instead of 72: still two per CU.  NTL = key tiles kept in LDS (NT, or NT - 1 when the last one is wholly padding).
// Measured on the forward (two half batches): equal to the register-staged kernel within 0.3 % — the 1.26 ms are the
// BYTES (they compete with the other half batch's GEMM for HBM), not the staging mechanics; K / V as one contiguous block per
// (image, head) instead of 128-B pieces of the packed qkv rows: equal too.
template <int NT, int KF, int NTL>
__global__ __launch_bounds__(256, 2) void attn_dma_kernel(const AttnArgs a) {
    constexpr int HD = 64, SR = 128, NP = 16 * NTL, KS = 2, DT = 4;
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + NP * SR;

    const int N = a.N, heads = a.heads;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = lane >> 4, li = lane & 15;
    const size_t tok_ld = (size_t)3 * heads * HD;   // elements between tokens of the packed qkv
    const half_t* qbase = reinterpret_cast<const half_t*>(a.qkv) + (size_t)b * N * tok_ld + (size_t)h * HD;

    // ---- Q fragments of this wave's first query tile: issued BEFORE the DMA pieces (the vector-memory counter is in order: a
    // load behind them would wait for them)
    const int nqt = (N + 15) >> 4;
    u32x4 qcur[KS];
    {
        const int query = wv * 16 + li;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qcur[ks] = u32x4{0u, 0u, 0u, 0u};
            if (query < N) qcur[ks] = *reinterpret_cast<const u32x4*>(qbase + (size_t)query * tok_ld + ks * 32 + g * 8);
        }
    }
    // ---- K and V: piece p = keys 8p .. 8p + 7; lane l -> key 8p + (l >> 3), slot l & 7, source chunk (l & 7) ^ f(key); wave w
    // takes pieces w, w + 4, ...; keys past the sequence are an out-of-range offset (zero rows, no traffic)
    {
        const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(a.qkv), 0, (unsigned)((size_t)a.B * N * tok_ld * 2), 0x00020000);
        const int lc = (lane & 7) ^ (((lane >> 4) & 3) << 1);
        const int base = (int)(((size_t)b * N * tok_ld + (size_t)h * HD) * 2) + lc * 16;
        const int krow = lane >> 3;
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
#pragma unroll
        for (int which = 1; which <= 2; ++which)      // 1: K, 2: V (the packed qkv row is [q | k | v] x [heads][hd])
#pragma unroll
            for (int j = 0; j < (NP / 8 + 3) / 4; ++j) {
                const int p = wv + 4 * j;
                const int key = 8 * p + krow;
                const int off = (p < NP / 8 && key < N && !TLXMI_DBG(a, 1)) ? base + (int)(key * tok_ld * 2) + which * heads * HD * 2 : OOB;      // (ablation bit 1, tuning flavour: no K / V traffic — zero rows)
                if (p < NP / 8)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)((which == 1 ? Ks : Vs) + p * 1024), 16, off, 0, 0, 0);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    half_t* obase = reinterpret_cast<half_t*>(a.out) + (size_t)b * N * heads * HD + (size_t)h * HD;

    // transposing V reads: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 16-column block
    const int vrow = 4 * g + (li >> 2);
    const int vf = ((2 * g + (li >> 3)) & 3) << 1;                       // f(32 pr + 4g + q [+ 16])
    auto load_v = [&](int pr, fp16x4 (&lo)[DT], fp16x4 (&hi)[DT]) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const char* a0 = Vs + (pr * 32 + vrow) * SR + (((2 * dt + ((li & 3) >> 1)) ^ vf) << 4) + (li & 1) * 8;
            lo[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(a0));
            hi[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(a0 + (2 * pr + 1 < NTL ? 16 * SR : 0)));   // P is 0 there
        }
    };
    const int kf_ = ((li >> 1) & 3) << 1;                                  // f(16 kt + li)
    auto k_addr = [&](int kt, int ks) { return Ks + (kt * 16 + li) * SR + (((4 * ks + g) ^ kf_) << 4); };

#pragma unroll 1
    for (int qt = wv; qt < nqt; qt += 4) {
        const int query = qt * 16 + li;
        const bool qok = query < N;
        u32x4 qnext[KS];
        {
            const int nq = query + 64;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                qnext[ks] = u32x4{0u, 0u, 0u, 0u};
                if (qt + 4 < nqt && nq < N)
                    qnext[ks] = *reinterpret_cast<const u32x4*>(qbase + (size_t)nq * tok_ld + ks * 32 + g * 8);
            }
        }
        float s[NT][4];
        float mx = -INFINITY;
        u32x4 kf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const u32x4*>(k_addr(0, ks));
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if (kt >= NTL) {      // a key tile wholly past the sequence, not in LDS (NTL < NT)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt][r] = -INFINITY;
                continue;
            }
            u32x4 kn[KS];
            if (kt + 1 < NTL) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kn[ks] = *reinterpret_cast<const u32x4*>(k_addr(kt + 1, ks));
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, kf[ks]),
                                                             __builtin_bit_cast(half8v, qcur[ks]), acc, 0, 0, 0);
            const bool full_tile = kt < KF || kt * 16 + 16 <= N;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + 4 * g + r;
                float v = acc[r];
                if (kt >= KF && !full_tile && key >= N) v = -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
            if (kt + 1 < NTL) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kf[ks] = kn[ks];
            }
        }
        fp16x4 vlo[DT], vhi[DT];
        load_v(0, vlo, vhi);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float ec = a.scale * 1.44269504088896340736f;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], ec, -mx * ec));
                s[kt][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pr = 0; pr < NT / 2; ++pr) {
            fp16x4 nlo[DT], nhi[DT];
            if (pr + 1 < NT / 2) load_v(pr + 1, nlo, nhi);
            half8v pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (half_t)s[2 * pr][r];
                pf[4 + r] = (half_t)s[2 * pr + 1][r];
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                half8v vfr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vfr[r] = (half_t)vlo[dt][r];
                    vfr[4 + r] = (half_t)vhi[dt][r];
                }
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfr, pf, o[dt], 0, 0, 0);
            }
            if (pr + 1 < NT / 2) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) { vlo[dt] = nlo[dt]; vhi[dt] = nhi[dt]; }
            }
        }
        if (qok) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                half4v ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[dt][r] * inv);
                *reinterpret_cast<half4v*>(obase + (size_t)query * heads * HD + dt * 16 + 4 * g) = ov;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qcur[ks] = qnext[ks];
    }
}

// Long sequences (more than 256 tokens: ViT at 384 x 384 = 577 tokens, vision_transformer.py:358-): the same S^T = K . Q^T /
// O^T = V^T . P^T fragments as attn_mfma_kernel, but the keys go through LDS in chunks of 16 * NTC and the softmax is
// the online one — per query tile a running maximum m, a running sum l and the output accumulators, rescaled by
// exp2((m - m') * c) when a chunk raises the maximum — so neither the score row nor K / V need fit anywhere at once.
// One workgroup (4 waves) per (batch, head, group of 4 * QPW query tiles): every wave owns QPW query tiles, all of whose
// state stays in registers while the chunks pass (QPW * (4 * DT + 2) registers); a chunk is staged by all 256 threads
// (loads of chunk c + 1 issued into registers before chunk c is computed, written to LDS after the barrier that retires its
// readers).  No bias / mask (ViT): the scale is folded into the exponent.  K / V are re-read once per query group (L2).
template <int HD, int NTC, int QPW>
__global__ __launch_bounds__(256, 2) void attn_flash_kernel(const AttnArgs a) {
    constexpr int SR = HD * 2 + 32;
    constexpr int CK = 16 * NTC;             // keys per chunk
    constexpr int KS = HD / 32;
    constexpr int DT = HD / 16;
    constexpr int CPR = HD / 8;
    constexpr int ITEMS = CK * CPR, PER = (ITEMS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + CK * SR;

    const int N = a.N, heads = a.heads;
    const int nqt = (N + 15) >> 4, ngroups = (nqt + 4 * QPW - 1) / (4 * QPW);
    const int bh = blockIdx.x / ngroups, qg = blockIdx.x - bh * ngroups;
    const int b = bh / heads, h = bh - b * heads;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int g = lane >> 4, li = lane & 15;
    const size_t tok_ld = (size_t)3 * heads * HD;
    const half_t* qbase = reinterpret_cast<const half_t*>(a.qkv) + (size_t)b * N * tok_ld + (size_t)h * HD;
    const half_t* kbase = qbase + (size_t)heads * HD;
    const half_t* vbase = qbase + (size_t)2 * heads * HD;
    half_t* obase = reinterpret_cast<half_t*>(a.out) + (size_t)b * N * heads * HD + (size_t)h * HD;
    const int nchunks = (N + CK - 1) / CK;

    // ---- Q fragments of this wave's tiles (tile j: query tile qg * 4 * QPW + 4 * j + wv)
    u32x4 qf[QPW][KS];
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
        const int query = (qg * 4 * QPW + 4 * j + wv) * 16 + li;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[j][ks] = u32x4{0u, 0u, 0u, 0u};
            if (query < N) qf[j][ks] = *reinterpret_cast<const u32x4*>(qbase + (size_t)query * tok_ld + ks * 32 + g * 8);
        }
    }
    f32x4 o[QPW][DT];
    float mrun[QPW], lrun[QPW];
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
        mrun[j] = -INFINITY;
        lrun[j] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    u32x4 kreg[PER], vreg[PER];
    auto fetch = [&](int c) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = t + u * 256;
            const int key = c * CK + i / CPR, cc = i % CPR;
            kreg[u] = u32x4{0u, 0u, 0u, 0u};
            vreg[u] = u32x4{0u, 0u, 0u, 0u};
            if (i < ITEMS && key < N) {
                kreg[u] = *reinterpret_cast<const u32x4*>(kbase + (size_t)key * tok_ld + cc * 8);
                vreg[u] = *reinterpret_cast<const u32x4*>(vbase + (size_t)key * tok_ld + cc * 8);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = t + u * 256;
            if (i < ITEMS) {
                *reinterpret_cast<u32x4*>(Ks + (i / CPR) * SR + (i % CPR) * 16) = kreg[u];
                *reinterpret_cast<u32x4*>(Vs + (i / CPR) * SR + (i % CPR) * 16) = vreg[u];
            }
        }
    };
    const int vlane = (4 * g + (li >> 2)) * SR + (li & 3) * 8;
    const int klane = li * SR + g * 16;
    const float ec = a.scale * 1.44269504088896340736f;

    fetch(0);
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        commit();
        __syncthreads();
        if (c + 1 < nchunks) fetch(c + 1);           // in flight under this chunk's arithmetic
        const bool tail = (c + 1) * CK > N;         // wave-uniform: only the last chunk can hold padded keys
#pragma unroll
        for (int j = 0; j < QPW; ++j) {
            // ---- scores of query tile j against the chunk's keys
            float s[NTC][4];
            float mx = mrun[j];
#pragma unroll
            for (int kt = 0; kt < NTC; ++kt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + kt * 16 * SR + klane + ks * 64);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, kf), __builtin_bit_cast(half8v, qf[j][ks]), acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[r];
                    if (tail && c * CK + kt * 16 + 4 * g + r >= N) v = -INFINITY;
                    s[kt][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // every chunk holds at least one real key, so mx is finite from the first chunk on; mrun = -inf gives alpha = 0
            const float alpha = __builtin_amdgcn_exp2f((mrun[j] - mx) * ec);
            mrun[j] = mx;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NTC; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][r], ec, -mx * ec));
                    s[kt][r] = p;
                    sum += p;
                }
            lrun[j] = lrun[j] * alpha + sum;      // per-lane partial sums: alpha is the same in the 4 lanes of a query
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[j][dt] *= alpha;
            // ---- O^T += V^T . P^T
#pragma unroll
            for (int pr = 0; pr < NTC / 2; ++pr) {
                half8v pf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[r] = (half_t)s[2 * pr][r];
                    pf[4 + r] = (half_t)s[2 * pr + 1][r];
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const char* a0 = Vs + pr * 32 * SR + vlane + dt * 32;
                    const fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(a0));
                    const fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(a0 + 16 * SR));
                    half8v vf;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        vf[r] = (half_t)lo[r];
                        vf[4 + r] = (half_t)hi[r];
                    }
                    o[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[j][dt], 0, 0, 0);
                }
            }
        }
        __syncthreads();                             // every wave is done with this chunk before the next one overwrites it
    }
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
        float l = lrun[j];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.f / l;
        const int query = (qg * 4 * QPW + 4 * j + wv) * 16 + li;
        if (query < N) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                half4v ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[j][dt][r] * inv);
                *reinterpret_cast<half4v*>(obase + (size_t)query * heads * HD + dt * 16 + 4 * g) = ov;
            }
        }
    }
}

// Persistent variant for windows of at most 64 tokens (Swin: 49 tokens, thousands of (window, head) items per
// launch).  With one workgroup per item the launch was latency-bound — time proportional to the item count,
// 175 / 92 / 51 / 29 us for 32768 / 16384 / 8192 / 4096 items — every workgroup paying its own global-memory
// latency before its first barrier.  Here a workgroup walks items blockIdx, blockIdx + grid, ...; the K / V rows,
// the Q fragments and the bias-table values of the NEXT item are fetched into registers while the current
// item computes, so only the first item of a workgroup waits for memory.  Each wave owns one 16-query tile.
// Scores: scale * q.k + comb (a.comb, pre-summed bias + mask, or nothing).
// TAB (round 3): the table stays in LDS.  Read per item it is 16 KB (64 x 64 fp32) against 9 KB of q, k, v — 20 % of the kernel's
// time at 56 x 56 — and every item of one (window position, head) uses the same one.  A workgroup is therefore bound to one
// (window position w, head or head pair): it copies that table into LDS once, re-ordered so that wave wv's fragment kt is the
// 1 KiB run ((wv * NT + kt) * 64 + lane) * 16 (conflict-free ds_read_b128), and walks the images img = j, j + S, ... of its
// share (item b = img * nW + w); `spc` = S, the workgroups per (w, head pair).
template <int HD, int NT, int KF, bool TAB = false>
__global__ __launch_bounds__(256) void attn_win_kernel(const AttnArgs a, const int nitems, const int spc) {
    constexpr int SR = HD * 2 + 32;
    constexpr int NP = 16 * NT;
    constexpr int KS = HD / 32;
    constexpr int DT = HD / 16;
    constexpr int CPR = HD / 8;
    constexpr int ITEMS = NP * CPR, PER = (ITEMS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + NP * SR;
    char* Ts = smem + 2 * NP * SR;          // TAB: G tables of [4 waves][NT][64 lanes][16 bytes]
    constexpr int TBYTES = 4 * NT * 1024;

    const int N = a.N, heads = a.heads;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int g = lane >> 4, li = lane & 15;
    const size_t tok_ld = (size_t)3 * heads * HD;
    const int nqt = (N + 15) >> 4;
    const int query = wv * 16 + li;
    const bool qok = wv < nqt && query < N;
    const int npc = (N + 31) & ~31;

    u32x4 kreg[PER], vreg[PER], qreg[KS];
    f32x4 treg[NT];
    // Window map (a.wm_ws > 0, TAB form only: the workgroup is bound to ONE window position, so the image rows of its query and key
    // tokens are per-thread constants): token tok of window wpos -> row of the image-order token matrix
    const bool mapped = TAB && a.wm_ws > 0;
    const int wm_nwx = mapped ? a.wm_W / a.wm_ws : 1, wm_wpi = mapped ? (a.wm_H / a.wm_ws) * wm_nwx : 1, wm_L = a.wm_H * a.wm_W;
    auto map_row = [&](int wpos, int tok) -> int {
        const int wy = wpos / wm_nwx, wx = wpos - wy * wm_nwx;
        const int iy = tok / a.wm_ws, ix = tok - iy * a.wm_ws;
        int y = wy * a.wm_ws + iy + a.wm_shift, x = wx * a.wm_ws + ix + a.wm_shift;
        if (y >= a.wm_H) y -= a.wm_H;
        if (x >= a.wm_W) x -= a.wm_W;
        return y * a.wm_W + x;
    };
    int mq = 0, mk[PER];      // filled below once `tw` (the workgroup's window position) is known
#pragma unroll
    for (int u = 0; u < PER; ++u) mk[u] = 0;
    auto fetch = [&](int item) {
        const bool live = item < nitems;
        const int b = live ? item / heads : 0, h = live ? item - b * heads : 0;
        // mapped: b = img * wpi + wpos, rows of the image; else the tokens of window b are rows b * N ...
        const size_t row0 = mapped ? (size_t)(b / wm_wpi) * wm_L : (size_t)b * N;
        const half_t* qbase = reinterpret_cast<const half_t*>(a.qkv) + row0 * tok_ld + (size_t)h * HD;
        const half_t* kbase = qbase + (size_t)heads * HD;
        const half_t* vbase = qbase + (size_t)2 * heads * HD;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qreg[ks] = u32x4{0u, 0u, 0u, 0u};
            if (live && qok) qreg[ks] = *reinterpret_cast<const u32x4*>(qbase + (size_t)(mapped ? mq : query) * tok_ld + ks * 32 + g * 8);
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = t + u * 256;
            const int key = i / CPR, c = i - key * CPR;
            kreg[u] = u32x4{0u, 0u, 0u, 0u};
            vreg[u] = u32x4{0u, 0u, 0u, 0u};
            if (live && i < ITEMS && key < N) {
                const size_t kr = mapped ? mk[u] : key;
                kreg[u] = *reinterpret_cast<const u32x4*>(kbase + kr * tok_ld + c * 8);
                vreg[u] = *reinterpret_cast<const u32x4*>(vbase + kr * tok_ld + c * 8);
            }
        }
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) treg[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!TAB && a.comb && live && wv < nqt) {      // rows / columns up to npc exist in the padded table
            const float* crow = a.comb + (((size_t)(a.nW > 0 ? b % a.nW : 0) * heads + h) * npc + query) * npc + 4 * g;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
                if (16 * kt < npc) treg[kt] = *reinterpret_cast<const f32x4*>(crow + 16 * kt);
        }
    };

    const int vlane = (4 * g + (li >> 2)) * SR + (li & 3) * 8;
    const int klane = li * SR + g * 16;
    // Item order of a workgroup.  With hd = 32 a token's 128-byte line holds the q (k, v) rows of TWO adjacent heads: a workgroup
    // takes head pairs — items 2p, 2p + 1 back to back, p = blockIdx, blockIdx + grid, ... — so that the second head finds the
    // line in this CU's cache; with one item per workgroup the two halves went to workgroups on different XCDs (two L2s fetched
    // every line).  G = 1: item = blockIdx + n * grid.
    constexpr int G = HD == 32 ? 2 : 1;
    const bool pairs = G == 2 && !(heads & 1);
    // TAB: combo = (window position w, head group hg) = blockIdx % combos, j = blockIdx / combos
    const int nWe = mapped ? wm_wpi : (a.nW > 0 ? a.nW : 1);      // (mapped: one workgroup per window POSITION also without a mask)
    const int hgs = pairs ? heads >> 1 : heads, combos = nWe * hgs;
    const int combo = TAB ? (int)blockIdx.x % combos : 0, tj = TAB ? (int)blockIdx.x / combos : 0;
    const int tw = combo / hgs, thg = combo - tw * hgs;
    const int nimg = a.B / nWe;
    if (mapped) {
        mq = map_row(tw, query < N ? query : 0);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int key = (t + u * 256) / CPR;
            mk[u] = map_row(tw, key < N ? key : 0);
        }
    }
    auto item_at = [&](int n) -> int {
        if constexpr (TAB) {
            const int img = tj + (pairs ? n >> 1 : n) * spc;
            if (img >= nimg) return nitems;
            return (img * nWe + tw) * heads + (pairs ? 2 * thg + (n & 1) : thg);
        }
        if (G == 1 || (heads & 1)) return (int)blockIdx.x + n * (int)gridDim.x;
        return 2 * ((int)blockIdx.x + (n >> 1) * (int)gridDim.x) + (n & 1);
    };
    if constexpr (TAB) {
        // the table(s) of this workgroup: row-major [npc][npc] fp32 in memory -> fragment order in LDS, zeros beyond npc
        const int ntab = pairs ? 2 : 1;
        for (int tb = 0; tb < ntab; ++tb) {
            const float* src = a.comb + ((size_t)(a.nW > 0 ? tw : 0) * heads + (pairs ? 2 * thg + tb : thg)) * npc * npc;
            for (int i = t; i < 64 * NT * 4; i += 256) {
                const int row = i / (NT * 4), c4 = i - row * (NT * 4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < npc && 4 * c4 < npc) v = *reinterpret_cast<const f32x4*>(src + (size_t)row * npc + 4 * c4);
                const int fl = ((c4 & 3) << 4) | (row & 15);
                *reinterpret_cast<f32x4*>(Ts + tb * TBYTES + ((((row >> 4) * NT + (c4 >> 2)) * 64 + fl) << 4)) = v;
            }
        }
    }
    fetch(item_at(0));
    for (int n = 0, item = item_at(0); item < nitems; item = item_at(++n)) {
        // ---- this item's K / V rows from the prefetch registers to LDS (zero rows for padded keys)
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = t + u * 256;
            const int key = i / CPR, c = i - key * CPR;
            if (i < ITEMS) {
                *reinterpret_cast<u32x4*>(Ks + key * SR + c * 16) = kreg[u];
                *reinterpret_cast<u32x4*>(Vs + key * SR + c * 16) = vreg[u];
            }
        }
        u32x4 qcur[KS];
        f32x4 tb[NT];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qcur[ks] = qreg[ks];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) tb[kt] = treg[kt];
        __syncthreads();
        fetch(item_at(n + 1));         // travels while this item computes
        if constexpr (TAB) {
            const char* tp = Ts + (pairs ? (n & 1) * TBYTES : 0) + ((wv * NT * 64 + lane) << 4);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) tb[kt] = *reinterpret_cast<const f32x4*>(tp + kt * 1024);
        }

        if (wv < nqt) {
            const int b = item / heads, h = item - b * heads;
            // ---- scores s[kt][r] = S[query][key = 16kt + 4g + r]
            float s[NT][4];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + kt * 16 * SR + klane + ks * 64);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, kf), __builtin_bit_cast(half8v, qcur[ks]), acc, 0, 0, 0);
                }
                const bool full_tile = kt < KF || kt * 16 + 16 <= N;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + 4 * g + r;
                    float v = acc[r] * a.scale + tb[kt][r];
                    if (kt >= KF && !full_tile && key >= N) v = -INFINITY;
                    s[kt][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
            fp16x4 vlo[NT / 2][DT], vhi[NT / 2][DT];
#pragma unroll
            for (int pr = 0; pr < NT / 2; ++pr)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const char* a0 = Vs + pr * 32 * SR + vlane + dt * 32;
                    vlo[pr][dt] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(a0));
                    vhi[pr][dt] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(a0 + 16 * SR));
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __expf(s[kt][r] - mx);
                    s[kt][r] = p;
                    sum += p;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.f / sum;
            f32x4 o[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pr = 0; pr < NT / 2; ++pr) {
                half8v pf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[r] = (half_t)s[2 * pr][r];
                    pf[4 + r] = (half_t)s[2 * pr + 1][r];
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    half8v vf;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        vf[r] = (half_t)vlo[pr][dt][r];
                        vf[4 + r] = (half_t)vhi[pr][dt][r];
                    }
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[dt], 0, 0, 0);
                }
            }
            if (query < N) {
                const size_t orow = mapped ? (size_t)(b / wm_wpi) * wm_L + mq : (size_t)b * N + query;      // (mapped: back to the token's image row)
                half_t* obase = reinterpret_cast<half_t*>(a.out) + orow * heads * HD + (size_t)h * HD;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    half4v ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = (half_t)(o[dt][r] * inv);
                    *reinterpret_cast<half4v*>(obase + dt * 16 + 4 * g) = ov;
                }
            }
        }
        __syncthreads();               // every wave is done with this item's K / V before the next overwrite
    }
}

template <int HD, int NT, int KF> static int launch_win_kf(const AttnArgs& a, hipStream_t st) {
    constexpr int SR = HD * 2 + 32;
    const size_t lds = (size_t)2 * 16 * NT * SR;
    const int cus = device_cus();
    const long nitems = (long)a.B * a.heads;
    const bool pairs = HD == 32 && !(a.heads & 1);
    if (a.comb) {
        // table-resident form: workgroups bound to one (window position, head / head pair).  Grid = what is resident at once
        // (44 KB of LDS with two tables: 3 workgroups per CU; measured at half batch 64, stage 1..4: 66.6 / 34.9 / 20.7 / 11.0 us
        // streaming -> 45.4 / 27.3 / 17.0 / 10.7 us; 2, 4 or 6 per CU lose 10 - 25 % to a partial second round).  A workgroup
        // with a single image still reads no more table bytes than the streaming form does for its items.
        const int nWe = a.wm_ws > 0 ? (a.wm_H / a.wm_ws) * (a.wm_W / a.wm_ws) : (a.nW > 0 ? a.nW : 1), nimg = a.B / nWe;      // (kernel: nWe)
        const long combos = (long)nWe * (pairs ? a.heads / 2 : a.heads);
        const int wpc = (int)tune_int("TLXMI_WIN_WPC", 3);
        long spc = ((long)cus * wpc) / combos;
        if (spc < 1) spc = 1;
        if (spc > nimg) spc = nimg;
        const size_t tlds = lds + (size_t)(pairs ? 2 : 1) * 4 * NT * 1024;
        if (combos * spc < (1l << 31) && !tune_int("TLXMI_WIN_STREAM", 0)) {
            hipLaunchKernelGGL((attn_win_kernel<HD, NT, KF, true>), dim3((unsigned)(combos * spc)), dim3(256), tlds, st, a, (int)nitems, (int)spc);
            return check_launch("attention(windows, resident table)");
        }
    }
    if (a.wm_ws > 0) return fail(TLXMI_ERR_UNSUPPORTED, "attention_windows: the image-order form needs the resident-table kernel (a pre-summed table)");
    const long units = pairs ? nitems / 2 : nitems;      // head pairs (kernel: item_at)
    const long grid = units < (long)cus * 6 ? units : (long)cus * 6;
    hipLaunchKernelGGL((attn_win_kernel<HD, NT, KF, false>), dim3((unsigned)grid), dim3(256), lds, st, a, (int)nitems, 0);
    return check_launch("attention(windows)");
}
template <int HD, int NT> static int launch_win(const AttnArgs& a, hipStream_t st) {
    if (a.N > 16 * (NT - 2)) return launch_win_kf<HD, NT, NT - 2>(a, st);
    return launch_win_kf<HD, NT, 0>(a, st);
}

template <int HD, int NT, int ADD, int KF> static int launch_kf(const AttnArgs& a, hipStream_t st) {
    if constexpr (HD == 64 && ADD == 0 && NT >= 8) {      // ViT: K / V by LDS-DMA (TLXMI_ATTN_DMA=0, tuning flavour: the register-staged kernel)
        if (tune_int("TLXMI_ATTN_DMA", 1) && (size_t)a.B * a.N * 3 * a.heads * HD * 2 < (1ull << 31)) {
            AttnArgs b = a;
            b.debug = (int)tune_int("TLXMI_ATTN_DBG", 0);
            // an odd count of key tiles (ViT: 197 tokens = 13) keeps and multiplies only those (-0.4 % of the ViT-B/16 forward).
            // 52 KB would fit three workgroups per CU: measured +6 % (168 registers: 148 B of scratch in the query loop)
            auto go = [&](auto ntl) -> int {
                constexpr int NTL = decltype(ntl)::value;
                const size_t lds = (size_t)2 * 16 * NTL * 128;
                if (int rc = raise_lds_limit(reinterpret_cast<const void*>(&attn_dma_kernel<NT, KF, NTL>), 160 * 1024, "attention")) return rc;
                hipLaunchKernelGGL((attn_dma_kernel<NT, KF, NTL>), dim3(a.B * a.heads), dim3(256), lds, st, b);
                return 0;
            };
            if (int rc = (a.N <= 16 * (NT - 1) && !tune_int("TLXMI_ATTN_EVEN", 0)) ? go(IntTag<NT - 1>{}) : go(IntTag<NT>{})) return rc;
            return check_launch("attention(dma)");
        }
    }
    constexpr int SR = HD * 2 + 32;
    const size_t lds = (size_t)2 * 16 * NT * SR + (ADD == 1 ? (size_t)a.N * a.N * sizeof(float) : 0);
    if (lds > 64 * 1024)
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(&attn_mfma_kernel<HD, NT, ADD, KF>), 160 * 1024, "attention")) return rc;
    AttnArgs b = a;
    b.debug = (int)tune_int("TLXMI_ATTN_DBG", 0);
    hipLaunchKernelGGL((attn_mfma_kernel<HD, NT, ADD, KF>), dim3(a.B * a.heads), dim3(256), lds, st, b);
    return check_launch("attention(mfma)");
}

template <int HD, int NT, int ADD> static int launch_add(const AttnArgs& a, hipStream_t st) {
    if (a.N > 16 * (NT - 2)) return launch_kf<HD, NT, ADD, NT - 2>(a, st);
    return launch_kf<HD, NT, ADD, 0>(a, st);
}

template <int HD, int NT> static int launch_one(const AttnArgs& a, hipStream_t st) {
    if (a.comb) return launch_add<HD, NT, 3>(a, st);
    if (!a.bias && !(a.mask && a.nW > 0) && a.scale > 0.f) return launch_add<HD, NT, 0>(a, st);   // scale folded into exp2: needs scale > 0
    if (NT <= 4) return launch_add<HD, NT, 1>(a, st);
    return launch_add<HD, NT, 2>(a, st);
}

template <int HD> static int launch_hd(const AttnArgs& a, hipStream_t st) {
    const int nt = ((a.N + 31) / 32) * 2;
    // windows of at most 64 tokens whose bias is absent or pre-summed: the persistent kernel (many tiny items)
    if (nt <= 4 && (a.comb || (!a.bias && !(a.mask && a.nW > 0))) && (long)a.B * a.heads < (1l << 31)) {
        if (nt <= 2) return launch_win<HD, 2>(a, st);
        return launch_win<HD, 4>(a, st);
    }
    if (nt <= 2) return launch_one<HD, 2>(a, st);
    if (nt <= 4) return launch_one<HD, 4>(a, st);
    if (nt <= 8) return launch_one<HD, 8>(a, st);
    if (nt <= 10) return launch_one<HD, 10>(a, st);      // 144-token windows (Swin window 12, swin_transformer.py:641-650)
    if (nt <= 14) return launch_one<HD, 14>(a, st);
    return launch_one<HD, 16>(a, st);
}

// fp16, no bias / mask, scale > 0, hd 64 or 32: the chunked online-softmax kernel for sequences of more than 256 tokens
bool attn_flash_ok(const AttnArgs& a) {
    return (a.hd == 64 || a.hd == 32) && !a.bias && !(a.mask && a.nW > 0) && !a.comb && a.scale > 0.f;
}
template <int HD> static int launch_flash_hd(const AttnArgs& a, hipStream_t st) {
    constexpr int NTC = 8, QPW = HD == 64 ? 2 : 4;      // registers: QPW * (4 * DT + 2) state + Q fragments (hd 64 spills at 4)
    constexpr int SR = HD * 2 + 32;
    const size_t lds = (size_t)2 * 16 * NTC * SR;
    const int nqt = (a.N + 15) / 16, ngroups = (nqt + 4 * QPW - 1) / (4 * QPW);
    const long grid = (long)a.B * a.heads * ngroups;
    if (grid >= (1l << 31)) return fail(TLXMI_ERR_UNSUPPORTED, "attention: too many (batch, head, query group) items");
    hipLaunchKernelGGL((attn_flash_kernel<HD, NTC, QPW>), dim3((unsigned)grid), dim3(256), lds, st, a);
    return check_launch("attention(flash)");
}
int launch_attn_flash(const AttnArgs& a, hipStream_t st) {
    return a.hd == 64 ? launch_flash_hd<64>(a, st) : launch_flash_hd<32>(a, st);
}

int launch_attn_mfma(const AttnArgs& a, hipStream_t st) {
    if (a.hd == 64) return launch_hd<64>(a, st);
    if (a.hd == 96) return launch_hd<96>(a, st);      // vit_small_patch16_224: 768 / 8 heads (vision_transformer.py:208)
    return launch_hd<32>(a, st);
}

}  // namespace tlxmi

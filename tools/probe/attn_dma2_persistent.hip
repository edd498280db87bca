// RECORD of a round-5 experiment, not compiled into any library: attn_dma_kernel (tlxcv_amd/csrc/attention_mfma.hip) as a persistent
// 8-wave workgroup per CU with the K / V image double-buffered — the next (image, head) item's LDS-DMA pieces land in the second image
// while the current item is computed.  Bit-identical to attn_dma_kernel (tests at 65 / 130 / 197 / 256 tokens, 804 items), and
// SLOWER on the ViT-B/16 forward (hipGraph replay, tools/ab_graph.py TLXMI_ATTN_PERSIST 1,0, one box): batch 256 9.989 vs 9.910 ms,
// batch 128 5.232 vs 5.105 ms.  Why: (i) hipcc puts `s_waitcnt vmcnt(0)` in front of the first ds_read_b64_tr_b16 of every query
// tile (it cannot tell the V image being read from the image the DMA writes), so the prefetch is drained after the first QK^T
// anyway; (ii) 106 KB of LDS = ONE workgroup per CU, so a half batch's attention no longer shares a CU with anything of the other
// half (the one-item form runs two 52-KB workgroups per CU); (iii) round 4's finding stands — in the two-stream forward the
// attention launch is bound by its bytes competing with the other half's GEMM, not by how they are staged.
// Kernel body as it was in attention_mfma.hip (needs that file's AttnArgs / typedefs):

// attn_dma_kernel as a PERSISTENT workgroup of 8 waves with the K / V image double-buffered (round 5): one workgroup per CU walks
// items b * heads + h = blockIdx, blockIdx + grid, ...; while item i is computed out of buffer i & 1, the LDS-DMA pieces of item i + 1
// land in the other buffer (issued right behind the Q fragments of item i + 1, which travel in registers: a plain load issued
// BEHIND the pieces would wait for them, the vector-memory counter being in order).  One vmcnt(0) + barrier per item.  The
// one-item workgroup ran [Q loads, 26 - 28 pieces | wait | compute]: at two workgroups per CU the exposed wait was a third of the
// launch (64.5 of 90.7 us without the staging at batch 256).  Same arithmetic, same LDS image, same fragment reads:
// bit-identical to attn_dma_kernel.  106 KB of LDS (two images of 2 x 16 NTL x 128 B): one workgroup per CU, two waves per SIMD,
// up to two query tiles per wave (<= 256 tokens).
template <int NT, int KF, int NTL>
__global__ __launch_bounds__(512) void attn_dma2_kernel(const AttnArgs a, const int nitems) {
    constexpr int HD = 64, SR = 128, NP = 16 * NTL, KS = 2, DT = 4, QW = 2;
    constexpr int IMG = 2 * NP * SR;      // K image + V image of one item
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int N = a.N, heads = a.heads;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);      // 0 .. 7
    const int g = lane >> 4, li = lane & 15;
    const size_t tok_ld = (size_t)3 * heads * HD;
    const int nqt = (N + 15) >> 4;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.qkv), 0, (unsigned)((size_t)a.B * N * tok_ld * 2), 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    auto load_q = [&](int item, u32x4 (&q)[QW][KS]) {
        const int b = item / heads, h = item - b * heads;
        const half_t* qbase = reinterpret_cast<const half_t*>(a.qkv) + (size_t)b * N * tok_ld + (size_t)h * HD;
#pragma unroll
        for (int j = 0; j < QW; ++j) {
            const int query = (wv + 8 * j) * 16 + li;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                q[j][ks] = u32x4{0u, 0u, 0u, 0u};
                if (wv + 8 * j < nqt && query < N) q[j][ks] = *reinterpret_cast<const u32x4*>(qbase + (size_t)query * tok_ld + ks * 32 + g * 8);
            }
        }
    };
    // piece p = keys 8p .. 8p + 7 (attn_dma_kernel); wave w takes pieces w, w + 8, ...
    auto issue_dma = [&](int item, int buf) {
        const int b = item / heads, h = item - b * heads;
        const int lc = (lane & 7) ^ (((lane >> 4) & 3) << 1);
        const int base = (int)(((size_t)b * N * tok_ld + (size_t)h * HD) * 2) + lc * 16;
        const int krow = lane >> 3;
        char* img = smem + buf * IMG;
#pragma unroll
        for (int which = 1; which <= 2; ++which)
#pragma unroll
            for (int j = 0; j < (NP / 8 + 7) / 8; ++j) {
                const int p = wv + 8 * j;
                const int key = 8 * p + krow;
                const int off = (p < NP / 8 && key < N) ? base + (int)(key * tok_ld * 2) + which * heads * HD * 2 : OOB;
                if (p < NP / 8)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lds_ptr_t)(img + (which - 1) * NP * SR + p * 1024), 16, off, 0, 0, 0);
            }
    };
    const int vrow = 4 * g + (li >> 2);
    const int vf = ((2 * g + (li >> 3)) & 3) << 1;
    const int kf_ = ((li >> 1) & 3) << 1;

    u32x4 qc[QW][KS], qn[QW][KS];
    int item = (int)blockIdx.x;
    if (item < nitems) {
        load_q(item, qc);
        issue_dma(item, 0);
    }
#pragma unroll 1
    for (int it = 0; item < nitems; ++it, item += (int)gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this item's K / V pieces and Q fragments (and the previous item's stores)
        __syncthreads();                                       // ... of every wave; every wave is done reading the OTHER image
        const int nxt = item + (int)gridDim.x;
        if (nxt < nitems) {
            load_q(nxt, qn);
            issue_dma(nxt, (it + 1) & 1);
        }
        const char* Ks = smem + (it & 1) * IMG;
        const char* Vs = Ks + NP * SR;
        const int b = item / heads, h = item - b * heads;
        half_t* obase = reinterpret_cast<half_t*>(a.out) + (size_t)b * N * heads * HD + (size_t)h * HD;
        auto load_v = [&](int pr, fp16x4 (&lo)[DT], fp16x4 (&hi)[DT]) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const char* a0 = Vs + (pr * 32 + vrow) * SR + (((2 * dt + ((li & 3) >> 1)) ^ vf) << 4) + (li & 1) * 8;
                lo[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(a0));
                hi[dt] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(a0 + (2 * pr + 1 < NTL ? 16 * SR : 0)));
            }
        };
        auto k_addr = [&](int kt, int ks) { return Ks + (kt * 16 + li) * SR + (((4 * ks + g) ^ kf_) << 4); };
#pragma unroll
        for (int j = 0; j < QW; ++j) {
            const int qt = wv + 8 * j;
            if (qt >= nqt) continue;          // (wave-uniform)
            const int query = qt * 16 + li;
            const bool qok = query < N;
            float s[NT][4];
            float mx = -INFINITY;
            u32x4 kf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const u32x4*>(k_addr(0, ks));
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                if (kt >= NTL) {
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
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, kf[ks]), __builtin_bit_cast(half8v, qc[j][ks]), acc, 0, 0, 0);
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
                    const float pv = __builtin_amdgcn_exp2f(fmaf(s[kt][r], ec, -mx * ec));
                    s[kt][r] = pv;
                    sum += pv;
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
        }
#pragma unroll
        for (int j = 0; j < QW; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qc[j][ks] = qn[j][ks];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


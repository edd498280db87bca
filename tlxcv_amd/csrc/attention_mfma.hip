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

struct AttnArgs {
    const void* qkv;
    const float* bias;
    const float* mask;
    void* out;
    int B, N, heads, hd, nW;
    float scale;
};

typedef __fp16 fp16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) fp16x4 lds_fp16x4;

template <int HD, int NT>  // NT = number of 16-key tiles (even), keys padded to 16*NT
__global__ __launch_bounds__(256) void attn_mfma_kernel(const AttnArgs a) {
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

    // ---- stage K and V (zero rows for padded keys)
    for (int i = t; i < NP * CPR; i += 256) {
        const int key = i / CPR, c = i - key * CPR;
        u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
        if (key < N) {
            kv = *reinterpret_cast<const u32x4*>(kbase + (size_t)key * tok_ld + c * 8);
            vv = *reinterpret_cast<const u32x4*>(vbase + (size_t)key * tok_ld + c * 8);
        }
        *reinterpret_cast<u32x4*>(Ks + key * SR + c * 16) = kv;
        *reinterpret_cast<u32x4*>(Vs + key * SR + c * 16) = vv;
    }
    __syncthreads();

    const float* bias = a.bias ? a.bias + (size_t)h * N * N : nullptr;
    const float* mask = (a.mask && a.nW > 0) ? a.mask + (size_t)(b % a.nW) * N * N : nullptr;
    half_t* obase = reinterpret_cast<half_t*>(a.out) + (size_t)b * N * heads * HD + (size_t)h * HD;

    const int nqt = (N + 15) >> 4;
    for (int qt = wv; qt < nqt; qt += 4) {
        const int query = qt * 16 + li;
        const bool qok = query < N;
        // Q fragments (B operand): this lane's query row, d = 32*ks + 8g .. +7
        u32x4 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 z = {0u, 0u, 0u, 0u};
            qf[ks] = qok ? *reinterpret_cast<const u32x4*>(qbase + (size_t)query * tok_ld + ks * 32 + g * 8) : z;
        }
        // ---- scores: s[kt][r] = S[query][key = 16kt + 4g + r]
        float s[NT][4];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + (kt * 16 + li) * SR + (ks * 4 + g) * 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, kf),
                                                             __builtin_bit_cast(half8v, qf[ks]), acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + 4 * g + r;
                float v = acc[r] * a.scale;
                if (key < N) {
                    if (bias && qok) v += bias[(size_t)query * N + key];
                    if (mask && qok) v += mask[(size_t)query * N + key];
                } else {
                    v = -INFINITY;
                }
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
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

        // ---- O^T = V^T . P^T
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
            // transposing reads: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3
            const int q4 = li >> 2, p4 = li & 3;
            const int row0 = pr * 32 + 4 * g + q4;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const char* a0 = Vs + row0 * SR + (dt * 16 + 4 * p4) * 2;
                fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(a0));
                fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)(a0 + 16 * SR));
                half8v vf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vf[r] = (half_t)lo[r];
                    vf[4 + r] = (half_t)hi[r];
                }
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[dt], 0, 0, 0);
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
}

template <int HD, int NT> static int launch_one(const AttnArgs& a, hipStream_t st) {
    constexpr int SR = HD * 2 + 32;
    const size_t lds = (size_t)2 * 16 * NT * SR;
    if (lds > 64 * 1024) {
        static thread_local bool raised = false;
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<HD, NT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "attention: cannot raise LDS limit: %s", hipGetErrorString(e));
            raised = true;
        }
    }
    hipLaunchKernelGGL((attn_mfma_kernel<HD, NT>), dim3(a.B * a.heads), dim3(256), lds, st, a);
    return check_launch("attention(mfma)");
}

template <int HD> static int launch_hd(const AttnArgs& a, hipStream_t st) {
    const int nt = ((a.N + 31) / 32) * 2;
    if (nt <= 2) return launch_one<HD, 2>(a, st);
    if (nt <= 4) return launch_one<HD, 4>(a, st);
    if (nt <= 8) return launch_one<HD, 8>(a, st);
    if (nt <= 14) return launch_one<HD, 14>(a, st);
    return launch_one<HD, 16>(a, st);
}

int launch_attn_mfma(const AttnArgs& a, hipStream_t st) {
    if (a.hd == 64) return launch_hd<64>(a, st);
    return launch_hd<32>(a, st);
}

}  // namespace tlxmi

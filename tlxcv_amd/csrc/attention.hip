// Fused multi-head self attention on the packed qkv matrix.
// Reference: Attention.forward vision_transformer.py:112-123 (scale applied to q k^T, :117);
// WindowAttention.forward swin_transformer.py:192-229 (q scaled first :202, + relative position
// bias :205-215, + shift mask viewed as (B, nW, heads, N, N) :216-220, softmax, @v).
//
//   qkv [B][N][3][heads][hd]  ->  out [B][N][heads*hd]
//   out[b,i,h,:] = sum_j softmax_j(scale * q_i.k_j + bias[h,i,j] + mask[b % nW,i,j]) v_j
//
// Three kernels:
//   attn_rows_kernel  — exact-fp32 arithmetic, any hd <= 128, N <= 256: K and V of one (b, head)
//                       staged once in LDS as fp32, one wave per query row, wave-shuffle softmax.
//                       This is the fp32 parity path and the generic fallback.
//   attn_long_kernel  — any N (ViT at 384x384: 577 tokens, vision_transformer.py:209-215): one wave per
//                       query row, keys in tiles of 64 straight from global memory / L2, running max and
//                       sum (the usual online softmax); a coverage path, not a tuned one.
//   attn_mfma_kernel  — fp16 throughput path (see below), hd in {32, 64, 96}, N <= 256.
#include "common.h"
#include <stdlib.h>

namespace tlxmi {

struct AttnArgs {
    const void* qkv;
    const float* bias;
    const float* mask;
    void* out;
    int B, N, heads, hd, nW;
    float scale;
    const float* comb;   // bias + mask pre-summed and padded: [max(nW,1)][heads][NP][NP], NP = 32 * ceil(N / 32)
    int debug = 0;       // (the same struct as in attention_mfma.hip: keep the two definitions identical)
    // Window map (round 5; Swin, tlxmi_attention_windows): wm_ws > 0 -> qkv and out are IMAGE-order token matrices [B / wpi][wm_H * wm_W][...]
    // and item b = img * wpi + w is the wm_ws x wm_ws window w of image img after the cyclic shift (token (iy, ix) of window (wy, wx) is
    // pixel ((wy * ws + iy + shift) % H, (wx * ws + ix + shift) % W): swin_transformer.py:316-324, and :327-333 on the way back)
    int wm_ws = 0, wm_H = 0, wm_W = 0, wm_shift = 0;
};

template <typename T>
__global__ __launch_bounds__(256) void attn_rows_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Ks = reinterpret_cast<float*>(smem_raw);
    const int N = a.N, hd = a.hd, ldk = hd + 1;
    float* Vs = Ks + N * ldk;
    float* ps = Vs + N * ldk;   // [4][256]
    float* qs = ps + 4 * 256;   // [4][128]
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long tok_ld = 3L * a.heads * hd;
    const T* base = reinterpret_cast<const T*>(a.qkv) + (long)b * N * tok_ld + (long)h * hd;
    for (int i = t; i < N * hd; i += 256) {
        const int n = i / hd, d = i - n * hd;
        Ks[n * ldk + d] = (float)base[n * tok_ld + (long)a.heads * hd + d];
        Vs[n * ldk + d] = (float)base[n * tok_ld + 2L * a.heads * hd + d];
    }
    __syncthreads();
    const float* bias = a.bias ? a.bias + (long)h * N * N : nullptr;
    const float* mask = (a.mask && a.nW > 0) ? a.mask + (long)(b % a.nW) * N * N : nullptr;
    T* out = reinterpret_cast<T*>(a.out) + (long)b * N * a.heads * hd + (long)h * hd;
    float* myq = qs + wv * 128;
    float* myp = ps + wv * 256;
    for (int row = wv; row < N; row += 4) {
        for (int d = lane; d < hd; d += 64) myq[d] = (float)base[row * tok_ld + d];
        __builtin_amdgcn_wave_barrier();
        float s[4];
        float mx = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = lane + 64 * jj;
            s[jj] = -INFINITY;
            if (j < N) {
                float acc = 0.f;
                for (int d = 0; d < hd; ++d) acc = fmaf(myq[d], Ks[j * ldk + d], acc);
                acc *= a.scale;
                if (bias) acc += bias[(long)row * N + j];
                if (mask) acc += mask[(long)row * N + j];
                s[jj] = acc;
                mx = fmaxf(mx, acc);
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = lane + 64 * jj;
            if (j < N) {
                s[jj] = expf(s[jj] - mx);
                sum += s[jj];
            }
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = lane + 64 * jj;
            if (j < N) myp[j] = s[jj] * inv;
        }
        __builtin_amdgcn_wave_barrier();
        for (int d = lane; d < hd; d += 64) {
            float o = 0.f;
            for (int j = 0; j < N; ++j) o = fmaf(myp[j], Vs[j * ldk + d], o);
            out[(long)row * a.heads * hd + d] = (T)o;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_long_kernel(const AttnArgs a) {
    __shared__ float ps[4][64];
    __shared__ float qs[4][128];
    const int N = a.N, hd = a.hd;
    const int bh = blockIdx.x, b = bh / a.heads, h = bh - b * a.heads;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.y * 4 + wv;
    if (row >= N) return;       // wave-uniform; no block-wide barrier below
    const long tok_ld = 3L * a.heads * hd;
    const T* base = reinterpret_cast<const T*>(a.qkv) + (long)b * N * tok_ld + (long)h * hd;
    const T* kb = base + (long)a.heads * hd;
    const T* vb = base + 2L * a.heads * hd;
    const float* bias = a.bias ? a.bias + (long)h * N * N + (long)row * N : nullptr;
    const float* mask = (a.mask && a.nW > 0) ? a.mask + (long)(b % a.nW) * N * N + (long)row * N : nullptr;
    for (int d = lane; d < hd; d += 64) qs[wv][d] = (float)base[row * tok_ld + d];
    __builtin_amdgcn_wave_barrier();
    float m = -INFINITY, l = 0.f, o0 = 0.f, o1 = 0.f;   // running max / sum; output dims lane, lane + 64
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        float s = -INFINITY;
        if (j < N) {
            float acc = 0.f;
            const T* kr = kb + j * tok_ld;
            for (int d = 0; d < hd; ++d) acc = fmaf(qs[wv][d], (float)kr[d], acc);
            acc *= a.scale;
            if (bias) acc += bias[j];
            if (mask) acc += mask[j];
            s = acc;
        }
        const float mn = fmaxf(m, wave_max(s));
        const float p = j < N ? expf(s - mn) : 0.f;
        const float corr = expf(m - mn);     // 0 on the first tile (m = -inf)
        l = l * corr + wave_sum(p);
        ps[wv][lane] = p;
        __builtin_amdgcn_wave_barrier();
        o0 *= corr;
        o1 *= corr;
        const int nj = N - j0 < 64 ? N - j0 : 64;
        for (int jj = 0; jj < nj; ++jj) {
            const float pj = ps[wv][jj];
            const T* vr = vb + (long)(j0 + jj) * tok_ld;
            if (lane < hd) o0 = fmaf(pj, (float)vr[lane], o0);
            if (lane + 64 < hd) o1 = fmaf(pj, (float)vr[lane + 64], o1);
        }
        __builtin_amdgcn_wave_barrier();
        m = mn;
    }
    T* out = reinterpret_cast<T*>(a.out) + (long)b * N * a.heads * hd + (long)h * hd + (long)row * a.heads * hd;
    const float inv = 1.f / l;
    if (lane < hd) out[lane] = (T)(o0 * inv);
    if (lane + 64 < hd) out[lane + 64] = (T)(o1 * inv);
}

template <typename T> static int launch_long(const AttnArgs& a, hipStream_t st) {
    hipLaunchKernelGGL((attn_long_kernel<T>), dim3(a.B * a.heads, (a.N + 3) / 4), dim3(256), 0, st, a);
    return check_launch("attention(long)");
}

template <typename T> static int launch_rows(const AttnArgs& a, hipStream_t st) {
    const size_t lds = ((size_t)2 * a.N * (a.hd + 1) + 4 * 256 + 4 * 128) * sizeof(float);
    if (lds > 160 * 1024) return fail(TLXMI_ERR_UNSUPPORTED, "attention: N=%d hd=%d needs %zu B of LDS", a.N, a.hd, lds);
    if (lds > 64 * 1024)
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(&attn_rows_kernel<T>), 160 * 1024, "attention")) return rc;
    hipLaunchKernelGGL((attn_rows_kernel<T>), dim3(a.B * a.heads), dim3(256), lds, st, a);
    return check_launch("attention(rows)");
}

int launch_attn_mfma(const AttnArgs& a, hipStream_t st);  // attention_mfma.hip
bool attn_flash_ok(const AttnArgs& a);                     // attention_mfma.hip: chunked online-softmax kernel (long sequences)
int launch_attn_flash(const AttnArgs& a, hipStream_t st);

// General multi-head attention: separate, strided Q / K / V (sequence-first or batch-first, packed or not), query and
// key lengths may differ (cross attention), optional additive mask, optional head-averaged weights.
// Reference: MultiHeadAttention.forward, detection/detr.py:1003-1062 (q scaled by hd^-0.5 before q k^T :1023, + attn_mask
// :1038-1039, softmax :1041, @ v :1044, weights averaged over the heads :1054-1060); tlx.nn.MultiheadAttention.
// A coverage kernel (one wave per query row, keys in tiles of 64 with the usual running max / sum; with WEIGHTS the
// wave walks all heads of its row and re-derives the probabilities in a second pass so it can add p / heads into
// avgw[b][row][:] without atomics), not a tuned one: self attention of <= 256 tokens is routed to tlxmi_attention.
struct MhaArgs {
    const void *q, *k, *v;
    void* out;
    const float* mask;
    float* avgw;
    int B, Lq, Lk, heads, hd;
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;   // element strides: batch, row
    long mask_bs;                                          // 0: one [Lq][Lk] mask; Lq*Lk: one per (batch, head)
    float scale;
};

template <typename T, bool WEIGHTS>
__global__ __launch_bounds__(256) void mha_rows_kernel(const MhaArgs a) {
    __shared__ float ps[4][64];
    __shared__ float qs[4][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.y * 4 + wv;
    if (row >= a.Lq) return;       // wave-uniform; no block-wide barrier below
    const int b = WEIGHTS ? (int)blockIdx.x : (int)blockIdx.x / a.heads;
    const int h0 = WEIGHTS ? 0 : (int)blockIdx.x - b * a.heads, h1 = WEIGHTS ? a.heads : h0 + 1;
    const int hd = a.hd, Lk = a.Lk;
    float* wrow = WEIGHTS ? a.avgw + ((long)b * a.Lq + row) * Lk : nullptr;
    const float invh = 1.f / (float)a.heads;
    for (int h = h0; h < h1; ++h) {
        const T* qb = reinterpret_cast<const T*>(a.q) + b * a.q_bs + row * a.q_rs + (long)h * hd;
        const T* kb = reinterpret_cast<const T*>(a.k) + b * a.k_bs + (long)h * hd;
        const T* vb = reinterpret_cast<const T*>(a.v) + b * a.v_bs + (long)h * hd;
        const float* mask = a.mask ? a.mask + ((long)b * a.heads + h) * a.mask_bs + (long)row * Lk : nullptr;
        for (int d = lane; d < hd; d += 64) qs[wv][d] = (float)qb[d] * a.scale;     // q scaled first (detr.py:1023)
        __builtin_amdgcn_wave_barrier();
        float m = -INFINITY, l = 0.f, o0 = 0.f, o1 = 0.f;   // running max / sum; output dims lane, lane + 64
        for (int j0 = 0; j0 < Lk; j0 += 64) {
            const int j = j0 + lane;
            float s = -INFINITY;
            if (j < Lk) {
                float acc = 0.f;
                const T* kr = kb + j * a.k_rs;
                for (int d = 0; d < hd; ++d) acc = fmaf(qs[wv][d], (float)kr[d], acc);
                if (mask) acc += mask[j];
                s = acc;
            }
            const float mn = fmaxf(m, wave_max(s));
            // a row whose every key so far is masked with -inf keeps m = mn = -inf: exp(-inf - -inf) must not poison it
            const float p = (j < Lk && mn > -INFINITY) ? expf(s - mn) : 0.f;
            const float corr = mn > -INFINITY ? expf(m - mn) : 1.f;     // 0 on the first live tile (m = -inf)
            l = l * corr + wave_sum(p);
            ps[wv][lane] = p;
            __builtin_amdgcn_wave_barrier();
            o0 *= corr;
            o1 *= corr;
            const int nj = Lk - j0 < 64 ? Lk - j0 : 64;
            for (int jj = 0; jj < nj; ++jj) {
                const float pj = ps[wv][jj];
                const T* vr = vb + (long)(j0 + jj) * a.v_rs;
                if (lane < hd) o0 = fmaf(pj, (float)vr[lane], o0);
                if (lane + 64 < hd) o1 = fmaf(pj, (float)vr[lane + 64], o1);
            }
            __builtin_amdgcn_wave_barrier();
            m = mn;
        }
        T* out = reinterpret_cast<T*>(a.out) + b * a.o_bs + row * a.o_rs + (long)h * hd;
        const float inv = 1.f / l;
        if (lane < hd) out[lane] = (T)(o0 * inv);
        if (lane + 64 < hd) out[lane + 64] = (T)(o1 * inv);
        if constexpr (WEIGHTS) {
            for (int j = lane; j < Lk; j += 64) {
                float acc = 0.f;
                const T* kr = kb + j * a.k_rs;
                for (int d = 0; d < hd; ++d) acc = fmaf(qs[wv][d], (float)kr[d], acc);
                if (mask) acc += mask[j];
                const float pw = expf(acc - m) * inv * invh;
                wrow[j] = h == 0 ? pw : wrow[j] + pw;          // this wave owns the row: plain accumulate
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace tlxmi

using namespace tlxmi;

extern "C" int tlxmi_attention(const tlxmi_attn_desc* d, const void* qkv, const float* bias, const float* mask,
                               void* out, void* stream) {
    TLXMI_REQUIRE(d && qkv && out, TLXMI_ERR_BAD_ARG, "attention: null argument");
    TLXMI_REQUIRE(d->dtype == TLXMI_F16 || d->dtype == TLXMI_F32, TLXMI_ERR_BAD_ARG, "attention: bad dtype");
    TLXMI_REQUIRE(d->B > 0 && d->Ntok > 0 && d->heads > 0 && d->hd > 0, TLXMI_ERR_BAD_ARG, "attention: bad extent");
    TLXMI_REQUIRE(d->Ntok <= 65535 * 4 && d->hd <= 128 && (long long)d->B * d->heads < (1ll << 31), TLXMI_ERR_UNSUPPORTED,
                  "attention: Ntok=%d hd=%d (<=128)", d->Ntok, d->hd);
    TLXMI_REQUIRE(!mask || d->nW > 0, TLXMI_ERR_BAD_ARG, "attention: mask given but nW=%d", d->nW);
    TLXMI_REQUIRE(!mask || d->B % d->nW == 0, TLXMI_ERR_BAD_ARG, "attention: B=%d not a multiple of nW=%d", d->B, d->nW);
    AttnArgs a;
    a.qkv = qkv; a.bias = bias; a.mask = mask; a.out = out;
    a.B = d->B; a.N = d->Ntok; a.heads = d->heads; a.hd = d->hd; a.nW = mask ? d->nW : 0; a.scale = d->scale;
    a.comb = nullptr;
    hipStream_t st = as_stream(stream);
    if (d->Ntok > 256) {
        if (d->dtype == TLXMI_F16 && attn_flash_ok(a) && aligned16(qkv) && aligned16(out)) return launch_attn_flash(a, st);   // ViT at 384 x 384
        return d->dtype == TLXMI_F32 ? launch_long<float>(a, st) : launch_long<half_t>(a, st);
    }
    if (d->dtype == TLXMI_F32) return launch_rows<float>(a, st);
    if ((d->hd == 64 || d->hd == 32 || d->hd == 96) && aligned16(qkv) && aligned16(out)) return launch_attn_mfma(a, st);
    return launch_rows<half_t>(a, st);
}

extern "C" int tlxmi_mha(const tlxmi_mha_desc* d, const void* q, const void* k, const void* v, const float* mask,
                         void* out, float* avg_weights, void* stream) {
    TLXMI_REQUIRE(d && q && k && v && out, TLXMI_ERR_BAD_ARG, "mha: null argument");
    TLXMI_REQUIRE(d->dtype == TLXMI_F16 || d->dtype == TLXMI_F32, TLXMI_ERR_BAD_ARG, "mha: bad dtype");
    TLXMI_REQUIRE(d->B > 0 && d->Lq > 0 && d->Lk > 0 && d->heads > 0 && d->hd > 0, TLXMI_ERR_BAD_ARG, "mha: bad extent");
    TLXMI_REQUIRE(d->hd <= 128 && (long long)d->B * d->heads < (1ll << 31) && d->Lq <= 65535 * 4, TLXMI_ERR_UNSUPPORTED,
                  "mha: hd=%d (<= 128), Lq=%d", d->hd, d->Lq);
    TLXMI_REQUIRE(d->mask_mode >= 0 && d->mask_mode <= 2 && (d->mask_mode == 0) == (mask == nullptr), TLXMI_ERR_BAD_ARG,
                  "mha: mask_mode %d does not match the mask pointer", d->mask_mode);
    const long need = (long)d->heads * d->hd;
    TLXMI_REQUIRE(d->q_row_stride >= need || d->B == 1 || d->q_batch_stride >= need, TLXMI_ERR_BAD_ARG, "mha: q strides");
    MhaArgs a;
    a.q = q; a.k = k; a.v = v; a.out = out; a.mask = mask; a.avgw = avg_weights;
    a.B = d->B; a.Lq = d->Lq; a.Lk = d->Lk; a.heads = d->heads; a.hd = d->hd; a.scale = d->scale;
    a.q_bs = d->q_batch_stride; a.q_rs = d->q_row_stride; a.k_bs = d->k_batch_stride; a.k_rs = d->k_row_stride;
    a.v_bs = d->v_batch_stride; a.v_rs = d->v_row_stride; a.o_bs = d->out_batch_stride; a.o_rs = d->out_row_stride;
    a.mask_bs = d->mask_mode == 2 ? (long)d->Lq * d->Lk : 0;
    hipStream_t st = as_stream(stream);
    const dim3 blk(256);
    if (avg_weights) {
        const dim3 g(d->B, (d->Lq + 3) / 4);
        if (d->dtype == TLXMI_F32) hipLaunchKernelGGL((mha_rows_kernel<float, true>), g, blk, 0, st, a);
        else hipLaunchKernelGGL((mha_rows_kernel<half_t, true>), g, blk, 0, st, a);
    } else {
        const dim3 g(d->B * d->heads, (d->Lq + 3) / 4);
        if (d->dtype == TLXMI_F32) hipLaunchKernelGGL((mha_rows_kernel<float, false>), g, blk, 0, st, a);
        else hipLaunchKernelGGL((mha_rows_kernel<half_t, false>), g, blk, 0, st, a);
    }
    return check_launch("mha");
}

// tlxmi_attention with the relative-position bias and the shift mask handed over pre-summed and padded:
// comb[w][h][i][j] = bias[h][i][j] + mask[w][i][j] for i, j < Ntok, 0 elsewhere, rows and columns padded to
// NP = 32 * ceil(Ntok / 32); nW = 0 means one table per head ([1][heads][NP][NP]).  The caller builds it once per
// layer (swin_transformer.py:205-220 computes the same sum on every forward).  fp16, hd in {32, 64, 96},
// Ntok <= 256 only (the MFMA kernel); anything else: TLXMI_ERR_UNSUPPORTED, use tlxmi_attention.
extern "C" int tlxmi_attention_comb(const tlxmi_attn_desc* d, const void* qkv, const float* comb, void* out, void* stream) {
    TLXMI_REQUIRE(d && qkv && out && comb, TLXMI_ERR_BAD_ARG, "attention_comb: null argument");
    TLXMI_REQUIRE(d->B > 0 && d->Ntok > 0 && d->heads > 0 && d->hd > 0 && d->nW >= 0, TLXMI_ERR_BAD_ARG, "attention_comb: bad extent");
    TLXMI_REQUIRE(d->nW == 0 || d->B % d->nW == 0, TLXMI_ERR_BAD_ARG, "attention_comb: B=%d not a multiple of nW=%d", d->B, d->nW);
    if (!(d->dtype == TLXMI_F16 && (d->hd == 64 || d->hd == 32 || d->hd == 96) && d->Ntok <= 256 && aligned16(qkv) && aligned16(out) && aligned16(comb)))
        return fail(TLXMI_ERR_UNSUPPORTED, "attention_comb: fp16, hd in {32,64,96}, Ntok <= 256, 16-byte aligned buffers only");
    AttnArgs a;
    a.qkv = qkv; a.bias = nullptr; a.mask = nullptr; a.out = out; a.comb = comb;
    a.B = d->B; a.N = d->Ntok; a.heads = d->heads; a.hd = d->hd; a.nW = d->nW; a.scale = d->scale;
    return launch_attn_mfma(a, as_stream(stream));
}

// Swin's windowed attention on IMAGE-order token matrices (round 5): qkv [Bimg][H * W][3][heads][hd] as the qkv Linear wrote it for the
// rows of the residual stream, out [Bimg][H * W][heads * hd] in the same row order — roll(-shift) + window_partition on the way in and
// window_reverse + roll(+shift) on the way out (swin_transformer.py:316-333) are the kernel's row arithmetic, so no launch moves
// tokens between image order and window order.  comb: the pre-summed bias (+ mask) table of tlxmi_attention_comb, [max(nW,1)][heads]
// [NP][NP]; nW = 0 (no shift mask) or the number of windows per image.  fp16, ws * ws <= 64 tokens, hd in {32, 64, 96}.
extern "C" int tlxmi_attention_windows(const tlxmi_attn_desc* d, const void* qkv, const float* comb, void* out, int H, int W, int ws,
                                       int shift, void* stream) {
    TLXMI_REQUIRE(d && qkv && out && comb, TLXMI_ERR_BAD_ARG, "attention_windows: null argument");
    TLXMI_REQUIRE(H > 0 && W > 0 && ws > 0 && H % ws == 0 && W % ws == 0 && shift >= 0 && shift < ws, TLXMI_ERR_BAD_ARG,
                  "attention_windows: %d x %d tokens in windows of %d, shift %d", H, W, ws, shift);
    const int wpi = (H / ws) * (W / ws);
    TLXMI_REQUIRE(d->B > 0 && d->B % wpi == 0 && d->Ntok == ws * ws && d->heads > 0 && d->hd > 0 && (d->nW == 0 || d->nW == wpi), TLXMI_ERR_BAD_ARG,
                  "attention_windows: B = %d windows of %d tokens, nW = %d for %d windows per image", d->B, d->Ntok, d->nW, wpi);
    if (!(d->dtype == TLXMI_F16 && (d->hd == 64 || d->hd == 32 || d->hd == 96) && d->Ntok <= 64 && aligned16(qkv) && aligned16(out) && aligned16(comb)))
        return fail(TLXMI_ERR_UNSUPPORTED, "attention_windows: fp16, hd in {32,64,96}, windows of <= 64 tokens, 16-byte aligned buffers only");
    if ((long long)(d->B / wpi) * H * W * 3 * d->heads * d->hd * 2 >= (1ll << 40)) return fail(TLXMI_ERR_UNSUPPORTED, "attention_windows: too large");
    AttnArgs a;
    a.qkv = qkv; a.bias = nullptr; a.mask = nullptr; a.out = out; a.comb = comb;
    a.B = d->B; a.N = d->Ntok; a.heads = d->heads; a.hd = d->hd; a.nW = d->nW; a.scale = d->scale;
    a.wm_ws = ws; a.wm_H = H; a.wm_W = W; a.wm_shift = shift;
    return launch_attn_mfma(a, as_stream(stream));
}

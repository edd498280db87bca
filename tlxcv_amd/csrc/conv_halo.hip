// Stride-1 convolution for thin inputs (C*2 bytes = 32 / 64 / 128 per pixel), fp16: the ResNet stem after
// space-to-depth (4x4 taps over 16 channels -> 64; reference resnet.py:213-218), the 3x3 64 -> 64 convs of
// ResNet layer1 (resnet.py:142-156), DarkNet's 3x3 32 -> 64 (darknet.py:54-58).
//
// Why not the implicit GEMM: conv_igemm.hip gathers every tap of every pixel from L2 into LDS, R*S times
// the input (16x for the stem, 9x for a 3x3), and with only 64 output channels per input byte those layers
// are bound by that L2 -> LDS fill (measured 434 / 535 TFLOP/s, 19 B/clk/CU of fill).  Here
//   * a tile is 256 or 512 consecutive output pixels of one image x 64 channels; a workgroup walks a contiguous
//     range of tiles and keeps the input rows it needs in an LDS ring of NRING rows ((W + 2*pad) pixels, zero
//     padding included): every input row is brought in ONCE per workgroup by LDS-DMA (the R-1 halo rows two
//     consecutive tiles share stay in the ring), the rows the next tile adds are fetched while this tile is
//     computed.  Every tap is read from the ring: consecutive pixels of a row are consecutive in LDS, so the
//     K slice (tap s .. s+S-1, all channels) of filter row r is one contiguous span starting at pixel (y+r, x);
//   * the filters live in registers: a wave owns 32 output channels, 2 x (R*S*C*2/64) MFMA A-fragments
//     (144 VGPRs for 3x3x64), loaded once per workgroup — no filter traffic, no LDS reads for A;
//   * one workgroup per CU, two wave groups in alternation: tile i is computed by group i & 1 (4 waves = 2
//     pixel halves x 2 channel halves, one wave per SIMD) while the other group runs the epilogue of tile i-1
//     and issues the row DMAs of tile i+2 — the MFMA pipe is fed by one group while the other does its VALU /
//     memory work (with all 8 waves in lock step the pipe idled through every epilogue and staging phase:
//     7.5 us per 256 pixels of which 4.4 in the MFMA loop).  One barrier per tile.
// A ring row is padded to a whole number of 1-KiB DMA pieces (PWp pixels), so a piece never straddles rows:
// row / validity / base offset of a piece are wave-uniform, a lane adds constants.  LDS lines of 128 B are
// XOR-swizzled: physical 16-byte slot = slot ^ (line & 7), applied on the DMA source side and on the
// fragment reads; any 16 consecutive pixels of 128 B read conflict-free whatever the first pixel.
// Algorithmic bytes: input once + output once (+ residual); FLOPs 2*M*Cout*R*S*C.
//
// POOL variant (TLXMI_EPI_MAXPOOL_3S2P1: the ResNet stem, resnet.py:287-290 conv1 -> bn1 -> relu -> maxpool(3, 2, 1)):
// the 112 x 112 x 64 conv map (411 MB at batch 256) is never written.  A tile is exactly two conv rows (PI = 7: a
// pixel half of a tile = 16 * PI = Wo = 112 pixels = one row), so tile j of an image holds conv rows 2j, 2j+1.  The
// epilogue of a tile takes the horizontal 3-tap / stride-2 maximum in registers (the left / right neighbours of an even
// pixel come by DPP row rotates; the pixel left of a 16-pixel sub-tile is lane 15 of the previous sub-tile) and leaves
// two half-resolution rows (56 pixels x 64 channels, 7 KB each) in a 3-tile LDS ring; the epilogue phase after that
// — the other wave group, one barrier later — combines rows 2j-1, 2j, 2j+1 (the first from tile j-1's ring slot) and
// stores pooled row j with full-line 16-byte stores.  A workgroup whose range starts inside an image first recomputes
// the tile before it (no store) to obtain row 2j-1.  Padding of the pool is "skip", which equals -inf padding.
#include "common.h"
#include "conv_halo.h"
#include <stdlib.h>

namespace tlxmi {

typedef __attribute__((address_space(3))) void* lds_ptr_ch_t;
static __device__ __forceinline__ void ch_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_ch_t)lds, 16, voff, 0, 0, 0);
}
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t ch_srd(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
static __device__ __forceinline__ u32x4 ch_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
}
static __device__ __forceinline__ void ch_store16_nt(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 2);
}
static __device__ __forceinline__ void ch_store16_wb(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 0);
}

// R x S taps, PB = bytes per input pixel (C * 2); S * PB must be a multiple of 64 (one MFMA K slice)
// PI = MFMA pixel sub-tiles per wave: a tile is TP = 32 * PI consecutive pixels (8 where the filter registers
// leave room)
template <int CTRL> static __device__ __forceinline__ unsigned ch_dpp(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
// packed fp16 maximum as ONE instruction (the builtin maximum canonicalises both operands first: three instructions)
static __device__ __forceinline__ unsigned ch_pkmax(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
static __device__ __forceinline__ u32x4 ch_pkmax4(u32x4 a, u32x4 b) {
    return u32x4{ch_pkmax(a[0], b[0]), ch_pkmax(a[1], b[1]), ch_pkmax(a[2], b[2]), ch_pkmax(a[3], b[3])};
}

// PACT: the activation of the POOL epilogue, compiled in (a runtime switch per sub-tile doubled its instruction count)
template <int R, int S, int PB, int PI, bool RES, bool POOL = false, int PACT = TLXMI_ACT_NONE>
__global__ __launch_bounds__(512) void conv_halo_kernel(const HaloArgs a) {
    static_assert(!POOL || !RES, "the pooled epilogue takes no residual");
    static_assert(PACT == TLXMI_ACT_NONE || PACT == TLXMI_ACT_RELU, "POOL epilogue: ReLU or none");
    constexpr int TP = 32 * PI;          // pixels of a tile (2 pixel halves x PI sub-tiles of 16)
    constexpr int KR = S * PB / 64;      // K slices per filter row
    constexpr int NKK = R * KR;          // K slices in all
    constexpr int OOB = (int)0x80000000;
    constexpr int PSH = PB == 128 ? 7 : PB == 64 ? 6 : 5;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wid >> 2;                   // wave group: computes the tiles i with (i & 1) == grp
    const int pg = wid & 1, cg = (wid >> 1) & 1;  // pixel half, channel half inside the group
    const int frow = lane & 15, fg = lane >> 4;

    const __amdgpu_buffer_rsrc_t xsrd = ch_srd(a.x, a.x_bytes), wsrd = ch_srd(a.w, a.w_bytes);
    const __amdgpu_buffer_rsrc_t ysrd = ch_srd(a.y, a.y_bytes);
    const __amdgpu_buffer_rsrc_t rsrd = ch_srd(a.res ? a.res : a.y, a.res ? a.res_bytes : 0u);

    // one channel tile (a.nt) per launch: the filters are loaded once, before the tile loop.  Workgroup b takes
    // the contiguous tile range [b*T/G, (b+1)*T/G) in (image, first pixel) order: consecutive tiles of an image.
    const int ntiles = a.N * a.tpi;
    const int s_lo = (int)(((long)blockIdx.x * ntiles) / gridDim.x), t_hi = (int)(((long)(blockIdx.x + 1) * ntiles) / gridDim.x);
    // POOL: pooled row j needs conv row 2j-1 of the tile before; a range that starts inside an image recomputes it
    const int warm = (POOL && s_lo < t_hi && s_lo - (s_lo / a.tpi) * a.tpi != 0) ? 1 : 0;
    const int t_lo = s_lo - warm;
    const int n_mine = t_hi - t_lo;
    const float inv_wo = 1.0f / (float)a.Wo;

    // Tile geometry without integer division (a scalar s32 division is ~20 dependent instructions and a tile needs
    // three): x / d = floor((x + 0.5) * (1 / d)) is exact for the sizes here (x < 2^20, d < 2^12).
    const float inv_tpi = 1.0f / (float)a.tpi;
    struct Tile { int n, m0, npx, oy0, nr; bool ok; };
    auto tile_at = [&](int i) {
        Tile tl;
        tl.ok = i < n_mine;
        const int L = tl.ok ? t_lo + i : t_lo;
        tl.n = (int)(((float)L + 0.5f) * inv_tpi);
        const int tt = L - tl.n * a.tpi;
        tl.m0 = tt * TP;
        tl.npx = a.HoWo - tl.m0 < TP ? a.HoWo - tl.m0 : TP;
        tl.oy0 = (int)(((float)tl.m0 + 0.5f) * inv_wo);
        tl.nr = (int)(((float)(tl.m0 + tl.npx - 1) + 0.5f) * inv_wo) - tl.oy0 + R;
        return tl;
    };

    // ---- row DMA: piece = 1 KiB = PPP pixels of one ring row per wave instruction.
    // Lane -> physical slot -> logical slot (the swizzle mask line & 7 = (lane >> 3) & 7 is the same for every
    // piece) -> pixel inside the piece and byte inside the pixel: constants of the lane.
    constexpr int PPP = 1024 / PB;                    // pixels per piece
    const int lbyte = ((lane >> 3) << 7) + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);   // logical byte inside the piece
    const int lpix = lbyte >> PSH;                    // pixel inside the piece
    const int loff = lpix * a.x_ld * 2 + (lbyte & (PB - 1));
    const int PR = a.PWp / PPP;                       // pieces per ring row
    const int row_bytes = a.PWp << PSH;
    const int rmask = a.nring - 1;                    // nring is a power of two
    // rows [lo, hi) of image n -> ring slots slot0, slot0+1, ... (mod nring); rows outside the image are zero rows
    const int w4 = wid & 3, py_w = w4 / PR, pc_w = w4 - py_w * PR;
    auto load_rows = [&](int n, int lo, int hi, int slot0) {
        int py = py_w, pc = pc_w;                     // wave w of the staging group takes pieces w, w + 4, ...
        while (py < hi - lo) {
            const int iy = lo + py;
            const int x0 = pc * PPP - a.pw;           // input column of the piece's first pixel
            const int ix = x0 + lpix;
            const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const int rowoff = ((n * a.H + iy) * a.W + x0) * a.x_ld * 2;
            ch_dma16(xsrd, smem + ((slot0 + py) & rmask) * row_bytes + (pc << 10), in ? rowoff + loff : OOB);
            pc += 4;
            while (pc >= PR) { pc -= PR; ++py; }
        }
    };
    // ring state: `cursor` = next free slot; rows [.., have_hi) of image `img_ld` are loaded, row base_iy_ld sits in
    // slot base_slot_ld.  The tile being computed may still belong to the image before (cur_base_*).
    int cursor = 0, img_ld = -1, have_hi = 0, base_iy_ld = 0, base_slot_ld = 0;
    auto prefetch = [&](const Tile& tl, bool issue) {   // every wave tracks the ring; only the staging group issues
        if (!tl.ok) return;
        const int iy0 = tl.oy0 - a.ph;
        int lo = have_hi;
        if (tl.n != img_ld) {
            lo = iy0;
            img_ld = tl.n;
            base_iy_ld = iy0;
            base_slot_ld = cursor;
        }
        const int hi = iy0 + tl.nr;
        if (hi > lo) {
            if (issue) load_rows(tl.n, lo, hi, cursor);
            cursor = (cursor + hi - lo) & rmask;
            have_hi = hi;
        }
    };

    // ---- filters and scale / shift of a channel tile -> registers.  MFMA D row i of sub-tile ci is channel
    // 8*(i>>2) + 4*ci + (i&3) of the wave's 32, so that lane group g owns channels 8g .. 8g+7 (epilogue).
    u32x4 wreg[2][NKK];
    auto load_filters = [&](int nt) {
        const int cbase = nt * 64 + cg * 32;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
            const int ch = cbase + 8 * (frow >> 2) + 4 * ci + (frow & 3);
            const int woff = ch * a.Kp_bytes + fg * 16;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) wreg[ci][kk] = ch_load16(wsrd, woff + kk * 64);
        }
        // scale / shift of the launch's 64 channels -> LDS table behind the ring (read back in the epilogues)
        if (t < 64) {
            const int ch = nt * 64 + t < a.Cout ? nt * 64 + t : a.Cout - 1;
            float* tb = reinterpret_cast<float*>(smem + a.nring * row_bytes);
            tb[t] = a.scale ? a.scale[ch] : 1.f;
            tb[64 + t] = a.shift ? a.shift[ch] : 0.f;
        }
    };

    // ---- prologue: rows of tiles 0 (group 0 issues) and 1 (group 1), filters; everything landed before the loop
    // (a wait on a filter register inside it would also drain row DMAs and stores: one in-order counter; the
    // builtin, not inline asm, so that the compiler's own wait insertion sees it)
    int q0_iy, q0_slot, q1_iy = 0, q1_slot = 0, q2_iy = 0, q2_slot = 0;   // ring mapping of tiles p, p+1, p+2
    Tile tprev = tile_at(0), tcur = tprev, tnx1 = tile_at(1);   // tiles p-1, p, p+1 of phase p (each computed once)
    prefetch(tcur, grp == 0);
    q0_iy = base_iy_ld; q0_slot = base_slot_ld;
    prefetch(tnx1, grp == 1);
    q1_iy = base_iy_ld; q1_slot = base_slot_ld;
    load_filters(a.nt);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __builtin_amdgcn_s_barrier();

    const bool res_after = (a.flags & TLXMI_EPI_RES_AFTER_ACT) != 0;
    const int ch0 = a.nt * 64 + cg * 32 + 8 * fg;     // lane (g = fg, px = frow) owns channels ch0 .. ch0+7
    const bool chok = ch0 < a.Cout;                   // Cout is a multiple of 8 on this path
    f32x4 acc[2][PI];

    // phase p: group p & 1 computes tile p; the other group finishes tile p-1 and stages the rows of tile p+2
    // half-resolution rows of the POOL epilogue: [3 tile slots][2 rows][Wo/2 pixels][128 B], 16-byte chunk c of pixel
    // x' at slot c ^ (x' & 7) (the four even pixels a ds_write_b128 lane group stores would share banks otherwise)
    char* const hb = smem + a.nring * row_bytes + 512;
    constexpr int HROW = 16 * PI / 2 * 128;       // bytes of one half-resolution row (POOL: PI = 7 -> 56 pixels)
    Tile tpp = tprev;                             // tile p-2 (POOL)
    auto pooled_row = [&](int p) {
        if constexpr (POOL) {
                if (p >= 2) {
                    // pooled row of tile p-2 = max(half-res rows 2j-1 [tile p-3's second row], 2j, 2j+1); two 16-byte
                    // chunks per thread of the group (448 chunks = 56 pixels x 8); every lane issues both stores
                    // (out-of-range offset when it has nothing to store) so that the counted wait below stays exact
                    const Tile tj = tpp;
                    const int L = t_lo + p - 2;                       // global tile index
                    const int jrow = L - tj.n * a.tpi;                // tile inside its image = pooled row
                    const bool st_ok = L >= s_lo;                     // (the warm-up tile stores nothing)
                    const char* r0 = hb + (((p - 2) % 3) * 2) * HROW;
                    const char* rp = hb + (((p + 0) % 3) * 2 + 1) * HROW;   // slot (p-3) % 3, second row
                    const int t4 = t & 255;
#pragma unroll
                    for (int rd = 0; rd < 2; ++rd) {
                        const int c = t4 + 256 * rd;
                        const int xh = c >> 3, ch = c & 7;
                        const bool in = c < (16 * PI / 2) * 8;
                        const int off = xh * 128 + ((ch ^ (xh & 7)) << 4);
                        u32x4 m = u32x4{0u, 0u, 0u, 0u};
                        if (in) {
                            m = ch_pkmax4(*reinterpret_cast<const u32x4*>(r0 + off), *reinterpret_cast<const u32x4*>(r0 + HROW + off));
                            if (jrow > 0) m = ch_pkmax4(m, *reinterpret_cast<const u32x4*>(rp + off));
                        }
                        const int opix = (tj.n * a.tpi + jrow) * (16 * PI / 2) + xh;     // pooled pixel index (N, Ho/2, Wo/2)
                        if (!TLXMI_DBG(a, 0x8000)) ch_store16_wb(ysrd, m, (in && st_ok && ch * 8 < a.Cout && !TLXMI_DBG(a, 2)) ? (opix * a.y_ld + a.nt * 64 + ch * 8) * 2 : OOB);
                        else ch_store16_nt(ysrd, m, (in && st_ok && ch * 8 < a.Cout && !TLXMI_DBG(a, 2)) ? (opix * a.y_ld + a.nt * 64 + ch * 8) * 2 : OOB);
                    }
                }
        }
    };
    for (int p = 0; p <= n_mine + (POOL ? 1 : 0); ++p) {
        const Tile nx2 = tile_at(p + 2);
        if (grp == (p & 1)) {
            prefetch(nx2, false);
            if (p < n_mine && !TLXMI_DBG(a, 32)) {
                const Tile cur = tcur;
                // ring slot (for filter row 0) and byte inside the row of this lane's pixel of each sub-tile
                int sq[PI], bx[PI];
#pragma unroll
                for (int pi = 0; pi < PI; ++pi) {
                    int pt = pg * (16 * PI) + pi * 16 + frow;
                    pt = pt < cur.npx ? pt : cur.npx - 1;      // pixels past the tile read a valid row, never stored
                    const int m = cur.m0 + pt;
                    const int oy = (int)(((float)m + 0.5f) * inv_wo);
                    const int ox = m - oy * a.Wo;
                    sq[pi] = q0_slot + (oy - a.ph - q0_iy);
                    bx[pi] = (ox << PSH) + fg * 16;
                }
                const char* pb = smem;
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int pi = 0; pi < PI; ++pi) acc[ci][pi] = f32x4{0.f, 0.f, 0.f, 0.f};
                // K slices in order (filter row r, slice kr of its S*PB bytes), software-pipelined over two fragment
                // buffers: only one wave per SIMD computes, so the reads of slice k+1 are in flight under the MFMAs
                // of slice k.  128-byte pixels: the two slices of a line differ by an XOR of 64 in the address.
                u32x4 xf[2][PI];
                int phys[PI];
                auto fetch = [&](auto k_tag, u32x4 (&dst)[PI]) {
                    constexpr int K = decltype(k_tag)::value, r = K / KR, kr = K % KR;
                    if constexpr (PB == 128) {
                        if constexpr ((kr & 1) == 0) {
#pragma unroll
                            for (int pi = 0; pi < PI; ++pi) {
                                const int b = ((sq[pi] + r) & rmask) * row_bytes + bx[pi] + (kr >> 1) * 128;
                                phys[pi] = b ^ (((b >> 7) & 7) << 4);
                                dst[pi] = *reinterpret_cast<const u32x4*>(pb + phys[pi]);
                            }
                        } else {
#pragma unroll
                            for (int pi = 0; pi < PI; ++pi) dst[pi] = *reinterpret_cast<const u32x4*>(pb + (phys[pi] ^ 64));
                        }
                    } else {
#pragma unroll
                        for (int pi = 0; pi < PI; ++pi) {
                            const int b = ((sq[pi] + r) & rmask) * row_bytes + bx[pi] + kr * 64;
                            dst[pi] = *reinterpret_cast<const u32x4*>(pb + (b ^ (((b >> 7) & 7) << 4)));
                        }
                    }
                };
                auto mma = [&](auto k_tag, const u32x4 (&src)[PI]) {
                    constexpr int K = decltype(k_tag)::value;
#pragma unroll
                    for (int pi = 0; pi < PI; ++pi)
#pragma unroll
                        for (int ci = 0; ci < 2; ++ci)
                            acc[ci][pi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                __builtin_bit_cast(half8v, wreg[ci][K]), __builtin_bit_cast(half8v, src[pi]), acc[ci][pi], 0, 0, 0);
                };
                fetch(IntTag<0>{}, xf[0]);
                auto steps = [&](auto self, auto k_tag) -> void {
                    constexpr int K = decltype(k_tag)::value;
                    if constexpr (K < NKK) {
                        if constexpr (K + 1 < NKK) fetch(IntTag<K + 1>{}, xf[(K + 1) & 1]);
                        mma(k_tag, xf[K & 1]);
                        self(self, IntTag<K + 1>{});
                    }
                };
                steps(steps, IntTag<0>{});
            }
            if constexpr (POOL) {
                if TLXMI_DBG(a, 4) {
                    pooled_row(p);
                    if TLXMI_DBG(a, 8) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // my row DMAs of the phase before have landed
                } else if TLXMI_DBG(a, 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            // rows of tile p+2 first (its DMAs are the oldest operations of this phase), then the epilogue
            if (!TLXMI_DBG(a, 1)) prefetch(nx2, true); else prefetch(nx2, false);
            if constexpr (POOL) {
                if (p >= 1 && p <= n_mine && !TLXMI_DBG(a, 16)) {
                    // tile p-1: scale / shift / activation, fp16, horizontal 3-tap stride-2 maximum -> ring slot (p-1) % 3
                    const float* tb = reinterpret_cast<const float*>(smem + a.nring * row_bytes) + cg * 32 + 8 * fg;
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(tb), s1 = *reinterpret_cast<const f32x4*>(tb + 4);
                    const f32x4 h0 = *reinterpret_cast<const f32x4*>(tb + 64), h1 = *reinterpret_cast<const f32x4*>(tb + 68);
                    char* const hrow = hb + (((p - 1) % 3) * 2 + pg) * HROW;     // this wave's conv row (pg) of the tile
                    u32x4 prev = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int pi = 0; pi < PI; ++pi) {
                        float v[8];
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb) {
                            v[bb] = acc[0][pi][bb] * s0[bb] + h0[bb];
                            v[4 + bb] = acc[1][pi][bb] * s1[bb] + h1[bb];
                        }
                        half8v hv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
                        u32x4 cur4 = __builtin_bit_cast(u32x4, hv);
                        if constexpr (PACT == TLXMI_ACT_RELU) {      // on the rounded pairs: max(fp16(v), 0) == fp16(max(v, 0))
#pragma unroll
                            for (int q = 0; q < 4; ++q) cur4[q] = ch_pkmax(cur4[q], 0u);
                        }
                        u32x4 hm;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            // row_ror:1 — lane l takes lane l-1 of its 16-lane row, lane 0 takes lane 15: feed lane 15 the
                            // previous sub-tile's value, so that lane 0 receives pixel x-1 across the sub-tile seam
                            const unsigned left = ch_dpp<0x121>(frow == 15 ? prev[q] : cur4[q]);
                            const unsigned right = ch_dpp<0x12F>(cur4[q]);      // row_ror:15 — lane l takes lane l+1
                            const unsigned m = ch_pkmax(cur4[q], right);
                            hm[q] = (pi == 0 && frow == 0) ? m : ch_pkmax(m, left);   // x = 0: the left tap is padding
                        }
                        prev = cur4;
                        if ((frow & 1) == 0) {
                            const int xh = pi * 8 + (frow >> 1);
                            *reinterpret_cast<u32x4*>(hrow + xh * 128 + (((cg * 4 + fg) ^ (xh & 7)) << 4)) = hm;
                        }
                    }
                }
                if (!TLXMI_DBG(a, 4)) pooled_row(p);
            } else
            if (p >= 1) {
                const Tile cur = tprev;
                // residual loads (four sub-tiles at a time) before the stores of those sub-tiles: a load behind a
                // store would wait for it (in-order counter)
                u32x4 rr[4];
#pragma unroll
                for (int pi = 0; pi < PI; ++pi) {
                    if (RES && (pi & 3) == 0) {
#pragma unroll
                        for (int p2 = 0; p2 < 4; ++p2) {
                            const int pt2 = pg * (16 * PI) + (pi + p2) * 16 + frow;
                            const int mg2 = cur.n * a.HoWo + cur.m0 + pt2;
                            rr[p2] = ch_load16(rsrd, (chok && pt2 < cur.npx) ? (mg2 * a.res_ld + ch0) * 2 : OOB);
                        }
                    }
                    const int pt = pg * (16 * PI) + pi * 16 + frow;
                    const bool ok = chok && pt < cur.npx;
                    const int mg = cur.n * a.HoWo + cur.m0 + pt;
                    float v[8];
                    {
                        const float* tb = reinterpret_cast<const float*>(smem + a.nring * row_bytes) + cg * 32 + 8 * fg;
                        const f32x4 s0 = *reinterpret_cast<const f32x4*>(tb), s1 = *reinterpret_cast<const f32x4*>(tb + 4);
                        const f32x4 h0 = *reinterpret_cast<const f32x4*>(tb + 64), h1 = *reinterpret_cast<const f32x4*>(tb + 68);
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb) {
                            v[bb] = acc[0][pi][bb] * s0[bb] + h0[bb];
                            v[4 + bb] = acc[1][pi][bb] * s1[bb] + h1[bb];
                        }
                    }
                    float rv[8];
                    if constexpr (RES) {
                        const half8v hv = __builtin_bit_cast(half8v, rr[pi & 3]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) rv[e] = (float)hv[e];
                        if (!res_after) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += rv[e];
                        }
                    }
                    if (a.act == TLXMI_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    } else if (a.act == TLXMI_ACT_LEAKY) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * a.act_param;
                    } else if (a.act == TLXMI_ACT_RELU6) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(v[e], 0.f), 6.f);
                    } else if (a.act == TLXMI_ACT_HARDSWISH) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] * fminf(fmaxf(v[e] + 3.f, 0.f), 6.f) * (1.f / 6.f);
                    }
                    if (RES && res_after) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rv[e];
                    }
                    half8v hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
                    if (!TLXMI_DBG(a, 0x8000)) ch_store16_wb(ysrd, __builtin_bit_cast(u32x4, hv), (ok && !TLXMI_DBG(a, 2)) ? (mg * a.y_ld + ch0) * 2 : OOB);
                    else ch_store16_nt(ysrd, __builtin_bit_cast(u32x4, hv), (ok && !TLXMI_DBG(a, 2)) ? (mg * a.y_ld + ch0) * 2 : OOB);
                }
            }
            // the rows of tile p+2 have landed; the PI stores just issued (and nothing else) may stay in flight
            if constexpr (POOL) {
                if TLXMI_DBG(a, 8) { if (p == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // relaxed: waited for at the end of this group's next (compute) phase
                else if (p >= 2 && !TLXMI_DBG(a, 4)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (p >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PI) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        q2_iy = base_iy_ld; q2_slot = base_slot_ld;
        // after the barrier: nobody reads the rows only tile p needed, tile p+2's rows are visible to every wave
        if constexpr (POOL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the half-resolution rows are in LDS
        __builtin_amdgcn_s_barrier();
        q0_iy = q1_iy; q0_slot = q1_slot;
        q1_iy = q2_iy; q1_slot = q2_slot;
        tpp = tprev;
        tprev = tcur; tcur = tnx1; tnx1 = nx2;
    }
}

// The activations compiled into the epilogue above; the dispatcher sends others to conv_igemm.hip.
bool conv_halo_act_ok(int act) {
    return act == TLXMI_ACT_NONE || act == TLXMI_ACT_RELU || act == TLXMI_ACT_LEAKY || act == TLXMI_ACT_RELU6 ||
           act == TLXMI_ACT_HARDSWISH;
}

template <int R, int S, int PB, int PI, bool RES, bool POOL = false, int PACT = TLXMI_ACT_NONE> static int launch_halo_r(const HaloArgs& a, hipStream_t st, int cus) {
    const void* fn = reinterpret_cast<const void*>(&conv_halo_kernel<R, S, PB, PI, RES, POOL, PACT>);
    const size_t lds = (size_t)a.nring * a.PWp * PB + 512 + (POOL ? 3 * 2 * (16 * PI / 2) * 128 : 0);   // ring + scale / shift table (+ half-resolution rows)
    if (int rc = raise_lds_limit(fn, 160 * 1024, "conv_halo")) return rc;
    const int tiles = a.N * a.tpi;
    int grid = cus & ~7;
    if (grid < 8) grid = 8;
    if (grid > tiles) grid = tiles;
    const int dbg = (int)tune_int("TLXMI_DEBUG", 0);   // ablation bits: tuning flavour only
    for (int nt = 0; nt < a.ntn; ++nt) {     // one launch per tile of 64 output channels
        HaloArgs b = a;
        b.debug = dbg;
        b.nt = nt;
        void* args[] = {&b};
        hipError_t e = hipLaunchKernel(fn, dim3((unsigned)grid), dim3(512), args, lds, st);
        if (e != hipSuccess) return fail(TLXMI_ERR_LAUNCH, "conv_halo: HIP launch failed: %s", hipGetErrorString(e));
    }
    return TLXMI_OK;
}

template <int R, int S, int PB, int PI> static int launch_halo_t(const HaloArgs& a, hipStream_t st, int cus) {
    return a.res ? launch_halo_r<R, S, PB, PI, true>(a, st, cus) : launch_halo_r<R, S, PB, PI, false>(a, st, cus);
}

// Shapes with a compiled instantiation (R, S, bytes per pixel); returns the pixels of a tile, 0 if none
int conv_halo_tile_pixels(int R, int S, int PB) {
    if (R == 3 && S == 3 && PB == 128) return 128;
    if ((R == 3 && S == 3 && PB == 64) || (R == 4 && S == 4 && PB == 32) || (R == 2 && S == 2 && PB == 32)) return 256;
    return 0;
}

// The fused max-pool epilogue: a pixel half of a tile must be exactly one conv row (16 * PI = Wo) and tiles pair rows
bool conv_halo_pool_act_ok(int act) { return act == TLXMI_ACT_RELU || act == TLXMI_ACT_NONE; }
bool conv_halo_pool_ok(int R, int S, int PB, int Ho, int Wo) {
    return R == 4 && S == 4 && PB == 32 && Wo == 112 && Ho % 2 == 0 && Ho >= 2;
}

int launch_conv_halo(const HaloArgs& a, hipStream_t st, int cus) {
    if (a.pool) {
        if (!conv_halo_pool_ok(a.R, a.S, a.PB, a.Ho, a.Wo) || a.res)
            return fail(TLXMI_ERR_UNSUPPORTED, "conv_halo: no fused max-pool instantiation for this geometry");
        if (a.act == TLXMI_ACT_RELU) return launch_halo_r<4, 4, 32, 7, false, true, TLXMI_ACT_RELU>(a, st, cus);
        if (a.act == TLXMI_ACT_NONE) return launch_halo_r<4, 4, 32, 7, false, true, TLXMI_ACT_NONE>(a, st, cus);
        return fail(TLXMI_ERR_UNSUPPORTED, "conv_halo: the fused max-pool epilogue is compiled for ReLU / no activation");
    }
    if (a.R == 3 && a.S == 3 && a.PB == 128) return launch_halo_t<3, 3, 128, 4>(a, st, cus);
    if (a.R == 3 && a.S == 3 && a.PB == 64) return launch_halo_t<3, 3, 64, 8>(a, st, cus);
    if (a.R == 4 && a.S == 4 && a.PB == 32) return launch_halo_t<4, 4, 32, 8>(a, st, cus);
    if (a.R == 2 && a.S == 2 && a.PB == 32) return launch_halo_t<2, 2, 32, 8>(a, st, cus);
    return fail(TLXMI_ERR_UNSUPPORTED, "conv_halo: no instantiation for %dx%d taps, %d bytes per pixel", a.R, a.S, a.PB);
}

}  // namespace tlxmi

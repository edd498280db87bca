// Arguments of the 256 x 256 tile GEMM (gemm256.hip), filled by conv_igemm.hip's dispatcher.
#pragma once
#include <hip/hip_runtime.h>

namespace tlxmi {

struct Gemm256Args {
    const char* x;
    const char* w;
    char* y;
    const float* scale;
    const float* shift;
    const char* res;
    // LayerNorm folded around the Linear layers (gemm_stream.hip only; fp16):
    //   stats_out [M][4][2]: per row (sum y, sum y^2) over each 256-channel tile column of the outputs this launch stores (pairs past
    //             ceil(Cout / 256) are not written) — the PRODUCER side: the next LayerNorm's statistics without a pass over y;
    //   rowstats  such rows of pairs for the rows this launch READS (ln_planes of them valid, <= 4; ln_inv_c = 1 / row width, ln_eps): the CONSUMER of a
    //             LayerNorm whose gamma is folded into the packed filter forms (a, b) = (rstd, -mean * rstd) per row from them and stores
    //             y = act(a * acc + b * scale[n] + shift[n])  (scale = c1[n] = sum_k W'[n][k], shift = c2[n] = bias + W beta).
    const float* rowstats = nullptr;
    float* stats_out = nullptr;
    int ln_planes = 0;
    float ln_inv_c = 0.f, ln_eps = 0.f;
    int M, Cout, x_ld, y_ld, res_ld;
    int kchunks;   // true 16-byte chunks per row
    int ksteps;    // 64-byte steps (packed pitch / 64)
    int Kp_bytes;  // packed filter row pitch
    int act;
    float act_param;
    unsigned flags;
    int mtiles, ntiles;
    // gemm_stream.hip only: tiles [0, tiles_full) are 256-row tiles of rows [0, m_full), tiles [tiles_full, tiles_total) are
    // HALF-HEIGHT tiles (128 rows, the H = 1 phases idle) of rows [m_full, M) — the rows of a short last round (launch_gs)
    int tiles_full = 0, tiles_total = 0, m_full = 0;
    int gn;        // N-tiles per column panel of the tile walk
    unsigned x_bytes, w_bytes, y_bytes, res_bytes;
    int debug;     // tuning builds only
    // gemm_pp.hip only — 3-wide-filter convolution as implicit GEMM (conv != 0): X rows are gathered per filter tap.
    // One K tile (128 bytes) never straddles taps: C * sizeof(T) is 128 << ctshift bytes.
    int conv;
    int cH, cW, cWo, cHoWo, csh, csw, cph, cpw;   // input extent, output extent, stride, padding (dilation 1)
    int ctshift, ctaps;                           // log2(K tiles per tap), R * 3 taps
    // gemm_pp.hip only — split K (tlxmi_conv2d_splitk): the grid holds kslices copies of the tile grid, copy s multiplies K tiles
    // [s * kt_slice, (s + 1) * kt_slice) and stores its fp32 accumulators, unscaled, at y + s * slice_bytes ([M][y_ld] floats)
    int kslices = 1, kt_slice = 0;
    long long slice_bytes = 0;
};

int launch_gemm256(int dtype, int variant, const Gemm256Args& a, hipStream_t st);
// gemm_pp.hip: 256 x 256 tile, two wave groups in antiphase; a.ksteps = packed pitch / 128
int launch_gemm_pp(int dtype, const Gemm256Args& a, hipStream_t st);
int launch_gemm_pp128(int dtype, const Gemm256Args& a, hipStream_t st);   // 128 x 256 tiles (tails)
int launch_gemm_pp_n128(int dtype, const Gemm256Args& a, hipStream_t st); // 256 x 128 tiles (128 output channels)
// gemm_stream.hip: persistent version (one workgroup per CU walks its tiles as one K-tile stream)
bool gemm_stream_ok(int dtype, const Gemm256Args& a);
int launch_gemm_stream(int dtype, const Gemm256Args& a, hipStream_t st, int cus);

// gemm_w4.hip: 256 x 256 tiles on four waves (one per SIMD, 128 x 128 wave tiles, the whole register file), persistent
bool gemm_w4_ok(int dtype, const Gemm256Args& a);
int launch_gemm_w4(int dtype, const Gemm256Args& a, hipStream_t st, int cus);

// gemm_wreg.hip: K = 128 rows, all N channels per workgroup, the filter in registers (HBM-bound pointwise layers)
bool gemm_wreg_ok(int dtype, const Gemm256Args& a);
int launch_gemm_wreg(const Gemm256Args& a, hipStream_t st);

}  // namespace tlxmi

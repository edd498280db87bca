// Arguments of the fused bottleneck seam (block_seam.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace tlxmi {

struct SeamArgs {
    const char *x, *w3, *res, *w1;     // t2 [M][x_ld], packed expand filter [N1][K1], skip [M][res_ld], packed reduce filter [N2][N1]
    char *y, *z;                       // y [M][y_ld] (N1 channels), t1 [M][z_ld] (N2 channels)
    const float *scale3, *shift3, *scale1, *shift1;
    int M, N1;
    int x_ld, res_ld, y_ld, z_ld;      // elements between rows
    unsigned x_bytes, w3_bytes, res_bytes, y_bytes, w1_bytes, z_bytes;
    int y_nt, z_nt;                    // non-temporal stores of y / t1
    const char* wd;                    // PROJ form: packed shortcut filter [N1][K1]; `res` is then the block input [M][res_ld]
    const float *scale_d, *shift_d;
    unsigned wd_bytes;
    int debug;                         // tuning flavour: ablation bits
};

bool block_seam_shape_ok(int K1, int N1, int N2);
int launch_block_seam(const SeamArgs& a, int K1, int N2, hipStream_t st);
int launch_mlp_seam(const SeamArgs& a, int K1, int N2, hipStream_t st);      // the MLP form: fc1 + GELU + fc2 + residual

}  // namespace tlxmi

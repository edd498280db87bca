// Arguments of the thin-input stride-1 convolution (conv_halo.hip), filled by conv_igemm.hip's dispatcher.
#pragma once
#include <hip/hip_runtime.h>

namespace tlxmi {

struct HaloArgs {
    const char* x;
    const char* w;
    char* y;
    const float* scale;
    const float* shift;
    const char* res;
    int N, H, W, Cout, R, S, ph, pw, Ho, Wo, HoWo;
    int x_ld, y_ld, res_ld;   // elements between pixels
    int PB;                   // bytes per input pixel that take part (C * 2)
    int Kp_bytes;             // packed filter row pitch
    int act;
    float act_param;
    unsigned flags;
    int tpi;                  // tiles (conv_halo_tile_pixels consecutive output pixels) per image
    int ntn;                  // channel tiles (64 channels), one launch each
    int nt;                   // channel tile of this launch
    int PW;                   // patch width in pixels = Wo + S - 1
    int PWp;                  // patch row pitch in pixels: PW rounded up to whole 1-KiB pieces (1024 / PB pixels)
    int nring;                // rows of the LDS ring (power of two)
    unsigned x_bytes, w_bytes, y_bytes, res_bytes;
    int debug;                // tuning builds only
    int pool;                 // 1: y receives maxpool(3, 2, 1) of the conv + epilogue result (conv_halo.hip, POOL variant)
};

int conv_halo_tile_pixels(int R, int S, int PB);   // 0: no instantiation
bool conv_halo_act_ok(int act);
int launch_conv_halo(const HaloArgs& a, hipStream_t st, int cus);
bool conv_halo_pool_ok(int R, int S, int PB, int Ho, int Wo);
bool conv_halo_pool_act_ok(int act);   // geometry of the fused max-pool epilogue

}  // namespace tlxmi

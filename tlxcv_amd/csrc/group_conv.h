// Grouped 3x3 convolution on the small-block MFMA (group_conv.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/tlxmi.h"

namespace tlxmi {

// fp16, Cin == Cout, 4 / 8 / 16 / 32 channels per group, 3x3, padding 1, stride 1 or 2, no residual, C a multiple of 64
bool gconv_small_ok(const tlxmi_conv2d_desc* d, int groups, const void* res);
// w_packed: tlxmi_pack_group_filter's buffer; Kp_bytes: bytes of one packed filter row
int launch_gconv_small(const tlxmi_conv2d_desc* d, int groups, const void* x, const void* w_packed, const float* scale,
                       const float* shift, void* y, int Kp_bytes, hipStream_t st);

}  // namespace tlxmi

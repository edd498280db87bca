// Debug aid: fill the whole LDS of every CU with a pattern, so that a kernel launched next shows any read of LDS it has not
// written itself (stale data) as a NaN / huge value in its output instead of as "the same numbers as the previous launch".
#include <hip/hip_runtime.h>
extern "C" __global__ void poison_kernel(unsigned pattern) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    if (lds[(threadIdx.x * 7) % (160 * 1024 / 4)] != pattern) __builtin_trap();
}
extern "C" int poison_lds(unsigned pattern, void* stream) {
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute((const void*)poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
        raised = true;
    }
    hipLaunchKernelGGL(poison_kernel, dim3(1024), dim3(1024), 160 * 1024, (hipStream_t)stream, pattern);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// Test infrastructure (tests/test_lds_poison_gpu.py), not product: fill the whole LDS of every CU with a pattern, so that a
// kernel launched next shows any read of LDS bytes it has not written itself (a short wait count, a read before the barrier that
// orders it, a table entry nobody staged) as a NaN / infinity / changed bit in its output.  Back-to-back launches of one kernel
// hide that class of fault: the stale bytes are then the previous launch's identical data.
// 1024 workgroups of 160 KiB: one is resident per CU at a time, so every CU is swept (4 passes on 256 CUs).
#include <hip/hip_runtime.h>
extern "C" __global__ void __launch_bounds__(1024) poison_kernel(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    if (lds[(threadIdx.x * 7) % (160 * 1024 / 4)] != pattern && sink) *sink = 1u;   // keeps the stores alive
}
extern "C" int poison_lds(unsigned pattern, void* stream) {
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute((const void*)poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
        raised = true;
    }
    hipLaunchKernelGGL(poison_kernel, dim3(1024), dim3(1024), 160 * 1024, (hipStream_t)stream, pattern, (unsigned*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// How many workgroups share a CU as a function of dynamic LDS and VGPRs: a spin kernel of fixed duration, grid = k * CUs.
// build: hipcc --offload-arch=gfx950 -O2 occ_probe.hip -o occ_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int REGS> __global__ __launch_bounds__(256) void spin(long cycles, float* out) {
    extern __shared__ char smem[];
    float v[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) v[i] = (float)(threadIdx.x + i);
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s + smem[threadIdx.x];
}
template <int REGS> static void run(int cus, size_t lds) {
    const void* fn = reinterpret_cast<const void*>(&spin<REGS>);
    hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, fn);
    int occ = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 256, lds);
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const long cyc = 100000;   // 1 ms at 100 MHz wall clock
    printf("regs %3d (numRegs %d) lds %6zu occ-api %d:", REGS, fa.numRegs, lds, occ);
    for (int k = 1; k <= 8; ++k) {
        hipLaunchKernelGGL(spin<REGS>, dim3(cus * k), dim3(256), lds, 0, cyc, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin<REGS>, dim3(cus * k), dim3(256), lds, 0, cyc, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf(" k=%d %.2f", k, ms);
    }
    printf(" ms\n");
    hipFree(out);
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("CUs %d, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, regsPerBlock %d\n", p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock);
    for (size_t lds : {(size_t)0, (size_t)16384, (size_t)32768, (size_t)35328, (size_t)40960, (size_t)53248, (size_t)65536, (size_t)70656, (size_t)81920})
        run<16>(p.multiProcessorCount, lds);
    run<100>(p.multiProcessorCount, 0);
    run<150>(p.multiProcessorCount, 0);
    run<200>(p.multiProcessorCount, 0);
    return 0;
}

// Probe: what does an exec-masked LDS-DMA (buffer_load_dwordx4 ... lds) write?  Fill 4 KiB of LDS with a pattern, issue ONE 16-byte LDS-DMA with only
// lanes < NACT active, dump the LDS.  Build: hipcc --offload-arch=gfx950 -O2 -o ldsdma_mask ldsdma_mask.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void probe(const uint32_t* src, uint32_t* out, int nact, int base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* w = reinterpret_cast<uint32_t*>(smem + base);
    for (int i = threadIdx.x; i < 1024; i += 64) w[i] = 0xAAAA0000u | i;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, 4096, 0x00020000);
    const int lane = threadIdx.x;
    if (lane < nact) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + base + 256), 16, lane * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = w[i];
}
int main() {
    uint32_t *src, *out, h[1024];
    hipMalloc(&src, 4096); hipMalloc(&out, 4096);
    for (int i = 0; i < 1024; ++i) h[i] = 0x55550000u | i;
    hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int base : {0, 131072, 135168}) for (int nact : {8, 16, 64}) {
        probe<<<1, 64, 160 * 1024>>>(src, out, nact, base);
        hipMemcpy(h, out, 4096, hipMemcpyDeviceToHost);
        int changed = 0, first = -1, last = -1, zeros = 0;
        for (int i = 0; i < 1024; ++i) if (h[i] != (0xAAAA0000u | i)) { ++changed; if (first < 0) first = i; last = i; if (h[i] == 0) ++zeros; }
        printf("base %6d  active lanes %2d: %4d dwords changed, first %d last %d (expected %d .. %d), zeros %d, sample %08x %08x\n", base, nact, changed, first, last, 64, 64 + nact * 4 - 1, zeros, h[64], h[64 + nact * 4 - 1]);
    }
    return 0;
}

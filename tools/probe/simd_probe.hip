// Which SIMD does wave w of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8],
// sh [12], se [15:13] on gfx9)   build: hipcc -O3 --offload-arch=gfx950 tools/probe/simd_probe.hip -o tools/probe/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out)
{
    extern __shared__ char smem[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main()
{
    unsigned *d, h[8 * 16];
    hipMalloc(&d, sizeof(h));
    for (int threads : {512, 576, 768}) {
        hipMemset(d, 0, sizeof(h));
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(k, dim3(8), dim3(threads), 160 * 1024, 0, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 8; ++b) {
            printf("%d threads, wg %d: simd of wave 0..: ", threads, b);
            for (int w = 0; w < threads / 64; ++w) printf("%u ", (h[b * 16 + w] >> 4) & 3);
            printf("  (cu %u se %u)\n", (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7);
        }
    }
    return 0;
}

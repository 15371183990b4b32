// Does hipExtAnyOrderLaunch let a kernel start while its predecessor IN THE SAME STREAM still runs?  (a packet without the
// barrier bit; the packet behind it with the bit waits for both)   build: hipcc -O3 --offload-arch=gfx950 tools/probe/anyorder_probe.hip -o tools/probe/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long cycles, unsigned long long *stamp)
{
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < cycles) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}
int main()
{
    unsigned long long *st, h[6];
    hipMalloc(&st, sizeof(h));
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long long us100 = 100 * 100;       // wall_clock64: 100 MHz
    for (int flags = 0; flags < 2; ++flags) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, s);
            hipExtLaunchKernelGGL(spin, dim3(52), dim3(256), 0, s, nullptr, nullptr, 0, us100, st);             // A: 100 us on 52 CUs
            hipExtLaunchKernelGGL(spin, dim3(100), dim3(256), 0, s, nullptr, nullptr, flags, us100 / 2, st + 2);  // B: 50 us, any order?
            hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, nullptr, nullptr, 0, 100, st + 4);               // C: ordinary
            hipEventRecord(e1, s); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
        printf("flags=%d: A + B + C took %.1f us; B started %.1f us after A started (A ended at %.1f), C started at %.1f\n", flags, ms * 1e3,
               (double)(long long)(h[2] - h[0]) / 100.0, (double)(long long)(h[1] - h[0]) / 100.0, (double)(long long)(h[4] - h[0]) / 100.0);
    }
    return 0;
}

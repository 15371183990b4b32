// What does it cost the MAIN stream to let a side stream start behind one of its kernels?  (round 4)
//   A  plain: k1 -> k2 back to back
//   B  k1 carries a stop event (hipExtLaunchKernelGGL), the side stream waits for it
//   C  hipEventRecord between k1 and k2, the side stream waits for it
//   D  hipStreamWriteValue32 between k1 and k2, the side stream hipStreamWaitValue32
//   E  k1 itself stores the flag (last thing it does), the side stream hipStreamWaitValue32: nothing extra on the main stream
// build: hipcc -O2 --offload-arch=gfx950 tools/probe/sync_probe.cpp -o tools/probe/sync_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void spin(unsigned long long cycles, unsigned *flag, unsigned value)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(4);
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void side_work(float *x) { x[threadIdx.x] += 1.f; }

int main()
{
    hipStream_t main_s, side; CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t e0, e1, ev; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    float *x; CK(hipMalloc(&x, 4096)); CK(hipMemset(x, 0, 4096));
    unsigned *flag; CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
    // hipStreamWaitValue32 wants memory the CP can poll; device memory from hipMalloc is accepted on ROCm >= 5 (signal memory is the documented choice)
    unsigned *sig = nullptr;
    if (hipExtMallocWithFlags((void **)&sig, 64, hipMallocSignalMemory) != hipSuccess) { sig = flag; (void)hipGetLastError(); printf("(no signal memory: plain device memory)\n"); }
    CK(hipMemset(sig, 0, 8));
    const unsigned long long cyc = 100 * 100;     // s_memtime runs at 100 MHz: 100 us
    const int reps = 200;
    unsigned epoch = 0;
    for (int mode = 0; mode < 5; ++mode) {
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, main_s));
            for (int i = 0; i < reps; ++i) {
                ++epoch;
                switch (mode) {
                case 0:
                    hipLaunchKernelGGL(spin, dim3(52), dim3(256), 0, main_s, cyc / 10, nullptr, 0u);
                    break;
                case 1:
                    hipExtLaunchKernelGGL(spin, dim3(52), dim3(256), 0, main_s, nullptr, ev, 0, cyc / 10, nullptr, 0u);
                    CK(hipStreamWaitEvent(side, ev, 0));
                    hipLaunchKernelGGL(side_work, dim3(1), dim3(64), 0, side, x);
                    break;
                case 2:
                    hipLaunchKernelGGL(spin, dim3(52), dim3(256), 0, main_s, cyc / 10, nullptr, 0u);
                    CK(hipEventRecord(ev, main_s));
                    CK(hipStreamWaitEvent(side, ev, 0));
                    hipLaunchKernelGGL(side_work, dim3(1), dim3(64), 0, side, x);
                    break;
                case 3:
                    hipLaunchKernelGGL(spin, dim3(52), dim3(256), 0, main_s, cyc / 10, nullptr, 0u);
                    CK(hipStreamWriteValue32(main_s, sig, epoch, 0));
                    CK(hipStreamWaitValue32(side, sig, epoch, hipStreamWaitValueGte, 0xffffffffu));
                    hipLaunchKernelGGL(side_work, dim3(1), dim3(64), 0, side, x);
                    break;
                case 4:
                    hipLaunchKernelGGL(spin, dim3(52), dim3(256), 0, main_s, cyc / 10, sig, epoch);
                    CK(hipStreamWaitValue32(side, sig, epoch, hipStreamWaitValueGte, 0xffffffffu));
                    hipLaunchKernelGGL(side_work, dim3(1), dim3(64), 0, side, x);
                    break;
                }
                hipLaunchKernelGGL(spin, dim3(52), dim3(256), 0, main_s, cyc / 10, nullptr, 0u);
            }
            CK(hipEventRecord(e1, main_s));
            CK(hipEventSynchronize(e1));
            CK(hipStreamSynchronize(side));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass) printf("mode %c: %.2f us per pair of 10 us kernels on the main stream\n", "ABCDE"[mode], ms * 1e3 / reps);
        }
    }
    float h[64]; CK(hipMemcpy(h, x, sizeof(h), hipMemcpyDeviceToHost));
    printf("side work ran %.0f times (expected %d)\n", h[0], 2 * reps * 4);
    return 0;
}

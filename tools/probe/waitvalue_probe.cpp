// Does hipStreamWaitValue32 let a second stream start a kernel while a long kernel of the first stream is still running, gated
// by a word that kernel writes half-way?  (round 5: next layer's input projection beside the recurrent kernel.)
// Kernel A: spins `half` ticks of the 100 MHz clock, stores its clock into out[0], publishes flag = 1 (system-scope atomic after
// a drained store), spins another `half`, stores its clock into out[1].  Stream B: hipStreamWaitValue32(flag >= 1), kernel B stores
// its clock into out[2].  Gated and concurrent: out[0] <= out[2] <= out[1]; the lag out[2] - out[0] is the command processor's
// reaction time.  Memory kinds tried for the flag: plain hipMalloc, fine-grained, hipMallocSignalMemory, pinned host.
// build: hipcc -O2 --offload-arch=gfx950 tools/probe/waitvalue_probe.cpp -o tools/probe/waitvalue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned long long u64;

__global__ void producer(unsigned *flag, u64 *out, u64 half)
{
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < half) __builtin_amdgcn_s_sleep(8);
    out[0] = __builtin_amdgcn_s_memrealtime();
    __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2 * half) __builtin_amdgcn_s_sleep(8);
    out[1] = __builtin_amdgcn_s_memrealtime();
}
__global__ void consumer(u64 *out) { out[2] = __builtin_amdgcn_s_memrealtime(); }

int main()
{
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    u64 *out; CK(hipMalloc(&out, 64));
    const char *kinds[4] = {"hipMalloc", "fine-grained", "signal memory", "pinned host"};
    for (int kind = 0; kind < 4; ++kind) {
        unsigned *flag = nullptr;
        hipError_t e = hipSuccess;
        if (kind == 0) e = hipMalloc(&flag, 64);
        else if (kind == 1) e = hipExtMallocWithFlags((void **)&flag, 64, hipDeviceMallocFinegrained);
        else if (kind == 2) e = hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory);
        else e = hipHostMalloc((void **)&flag, 64, hipHostMallocDefault);
        if (e != hipSuccess) { printf("%-14s allocation failed: %s\n", kinds[kind], hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        for (u64 half : {5000ull, 20000ull}) {          // 50 us, 200 us
            CK(hipMemset(flag, 0, 4)); CK(hipMemset(out, 0, 64)); CK(hipDeviceSynchronize());
            e = hipStreamWaitValue32(b, flag, 1, hipStreamWaitValueGte, 0xffffffffu);
            if (e != hipSuccess) { printf("%-14s hipStreamWaitValue32 failed: %s\n", kinds[kind], hipGetErrorString(e)); (void)hipGetLastError(); break; }
            hipLaunchKernelGGL(consumer, dim3(1), dim3(1), 0, b, out);
            hipLaunchKernelGGL(producer, dim3(1), dim3(1), 0, a, flag, out, half);
            CK(hipDeviceSynchronize());
            u64 h[3]; CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
            printf("%-14s half %3llu us: flag at 0, consumer ran at %+7.1f us, producer ended at %+7.1f us  %s\n", kinds[kind], half / 100,
                   ((double)h[2] - (double)h[0]) / 100.0, ((double)h[1] - (double)h[0]) / 100.0,
                   h[2] >= h[0] && h[2] <= h[1] ? "GATED, CONCURRENT" : h[2] > h[1] ? "gated, but only after the producer ended" : "NOT gated");
        }
        if (kind == 3) (void)hipHostFree(flag); else (void)hipFree(flag);
    }
    return 0;
}

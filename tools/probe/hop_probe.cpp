// How long does one hand-off through L2 take between two workgroups?  (round 4; the cluster kernels' step is ~60 % hop)
// Ping-pong between workgroup 0 and workgroup P of one launch: 0 stores tag i (relaxed agent-scope atomic, as cn_lstm_cluster.hip's
// publish), P polls it (relaxed agent-scope atomic loads) and answers, 0 polls the answer.  Round trip / 2 = one hop.
// P = 8: same XCD under round-robin placement; P = 1: the neighbouring XCD.  Prints the XCC id of both.
// build: hipcc -O2 --offload-arch=gfx950 tools/probe/hop_probe.cpp -o tools/probe/hop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;

template <int POLLS>
__global__ void pingpong(u64 *slots, int partner, int iters, unsigned long long *out, unsigned *xcc)
{
    const int b = blockIdx.x;
    if (threadIdx.x == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[b] = id & 0xf; }
    if (b != 0 && b != partner) return;
    u64 *mine = slots + (b == 0 ? 0 : 64) + threadIdx.x, *theirs = slots + (b == 0 ? 64 : 0) + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= iters; ++i) {
        if (b == 0) __hip_atomic_store(mine, ((u64)i << 32) | 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // wait for the partner's tag i
        if (POLLS == 1) {
            for (;;) { u64 x = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((unsigned)(x >> 32) == (unsigned)i) break; }
        } else {
            u64 xa = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_sleep(4);
            for (;;) {
                u64 xb = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(xa >> 32) == (unsigned)i) break;
                xa = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(xb >> 32) == (unsigned)i) break;
            }
        }
        if (b != 0) __hip_atomic_store(mine, ((u64)i << 32) | 9u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (b == 0 && threadIdx.x == 0) *out = __builtin_amdgcn_s_memrealtime() - t0;
}

int main()
{
    u64 *slots; CK(hipMalloc(&slots, 128 * 8));
    unsigned long long *out; CK(hipMalloc(&out, 8));
    unsigned *xcc; CK(hipMalloc(&xcc, 64 * 4));
    const int iters = 2000;
    for (int polls = 1; polls <= 2; ++polls)
        for (int partner : {8, 1, 2, 16, 32}) {
            CK(hipMemset(slots, 0, 128 * 8));
            if (polls == 1) hipLaunchKernelGGL(pingpong<1>, dim3(64), dim3(64), 0, 0, slots, partner, iters, out, xcc);
            else            hipLaunchKernelGGL(pingpong<2>, dim3(64), dim3(64), 0, 0, slots, partner, iters, out, xcc);
            CK(hipDeviceSynchronize());
            unsigned long long t; unsigned x[64];
            CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(x, xcc, sizeof(x), hipMemcpyDeviceToHost));
            printf("polls in flight %d, partner wg %2d (xcc %u <-> %u): round trip %.0f ns, one hop %.0f ns\n", polls, partner, x[0], x[partner], t * 10.0 / iters, t * 5.0 / iters);
        }
    return 0;
}

// How long does one hand-off through L2 take between two workgroups?  (round 4; the cluster kernels' step is ~60 % hop)
// Ping-pong between workgroup 0 and workgroup P of one launch: 0 stores tag i, P polls it and answers, 0 polls the answer.
// Round trip / 2 = one hop.  P = 8: same XCD under round-robin placement; P = 1: the neighbouring XCD.  Prints the XCC id of both.
// Round 5: the primitives are a template parameter --
//   mode 0  relaxed AGENT-scope atomic store, polled by relaxed agent-scope atomic loads (sc1: what cn_lstm_cluster.hip shipped with)
//   mode 1  relaxed WORKGROUP-scope atomic store / loads (sc0 only; may be served by the polling CU's own L1 for ever: bounded)
//   mode 2  agent-scope store, polled by a returning agent-scope RMW (atomic OR 0: executes in the L2, sc1 = 0)
//   mode 3  workgroup-scope store (no sc1: the line stays in the XCD's L2), polled by the RMW of mode 2
// Every spin is bounded (a mode that never sees the partner's value reports `stuck` instead of hanging the device).
// build: hipcc -O2 --offload-arch=gfx950 tools/probe/hop_probe.cpp -o tools/probe/hop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;

template <int MODE> __device__ __forceinline__ void put(u64 *p, u64 v)
{
    if (MODE == 0 || MODE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int MODE> __device__ __forceinline__ u64 get(u64 *p)
{
    if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // (hipcc folds an idempotent RMW such as fetch_or(p, 0) into an agent-scope LOAD: the instruction is written out)
    u64 old, zero = 0;
    asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(p), "v"(zero) : "memory");
    return old;
}

template <int MODE>
__global__ void pingpong(u64 *slots, int partner, int iters, unsigned long long *out, unsigned *xcc, int *stuck)
{
    const int b = blockIdx.x;
    if (threadIdx.x == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[b] = id & 0xf; }
    if (b != 0 && b != partner) return;
    u64 *mine = slots + (b == 0 ? 0 : 64) + threadIdx.x, *theirs = slots + (b == 0 ? 64 : 0) + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool dead = false;
    for (int i = 1; i <= iters && !dead; ++i) {
        if (b == 0) put<MODE>(mine, ((u64)i << 32) | 7u);
        int spins = 0;
        for (;;) {
            u64 x = get<MODE>(theirs);
            if ((unsigned)(x >> 32) >= (unsigned)i) break;
            if (++spins > 200000) { dead = true; *stuck = 1; break; }
        }
        if (b != 0) put<MODE>(mine, ((u64)i << 32) | 9u);
    }
    // a stuck side releases the other one (agent scope: seen by every mode's poll sooner or later)
    if (dead) __hip_atomic_store(mine, ((u64)0x7fffffff << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b == 0 && threadIdx.x == 0) *out = __builtin_amdgcn_s_memrealtime() - t0;
}

int main()
{
    u64 *slots; CK(hipMalloc(&slots, 128 * 8));
    unsigned long long *out; CK(hipMalloc(&out, 8));
    unsigned *xcc; CK(hipMalloc(&xcc, 64 * 4));
    int *stuck; CK(hipMalloc(&stuck, 4));
    const int iters = 2000;
    const char *names[4] = {"agent store / agent load", "workgroup store / workgroup load", "agent store / RMW poll", "workgroup store / RMW poll"};
    for (int mode = 0; mode < 4; ++mode)
        for (int partner : {8, 1, 16}) {
            CK(hipMemset(slots, 0, 128 * 8)); CK(hipMemset(stuck, 0, 4));
            switch (mode) {
            case 0: hipLaunchKernelGGL(pingpong<0>, dim3(64), dim3(64), 0, 0, slots, partner, iters, out, xcc, stuck); break;
            case 1: hipLaunchKernelGGL(pingpong<1>, dim3(64), dim3(64), 0, 0, slots, partner, iters, out, xcc, stuck); break;
            case 2: hipLaunchKernelGGL(pingpong<2>, dim3(64), dim3(64), 0, 0, slots, partner, iters, out, xcc, stuck); break;
            default: hipLaunchKernelGGL(pingpong<3>, dim3(64), dim3(64), 0, 0, slots, partner, iters, out, xcc, stuck); break;
            }
            CK(hipDeviceSynchronize());
            unsigned long long t; unsigned x[64]; int s;
            CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(x, xcc, sizeof(x), hipMemcpyDeviceToHost)); CK(hipMemcpy(&s, stuck, 4, hipMemcpyDeviceToHost));
            printf("%-34s partner wg %2d (xcc %u <-> %u): round trip %6.0f ns, one hop %5.0f ns%s\n", names[mode], partner, x[0], x[partner], t * 10.0 / iters, t * 5.0 / iters,
                   s ? "   STUCK (never saw the partner's value)" : "");
        }
    return 0;
}

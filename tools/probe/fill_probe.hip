// How fast can a CU take an L2-resident operand into LDS?  (round 6: sizing a short-M gemm_nt: the headline's N-wide products
// re-read the 512 KB weight matrix once per 64-row panel, 235 panels -- if the L2 -> LDS path gives a CU 16 B/clk they are bound
// by it, if it gives 50+ they are not.)
// Every workgroup (one per CU, 4 or 8 waves) fills `stages` of 40 KB: 8 KB of its own A panel (M x K, K = 1024 bf16) and 32 KB of a
// shared B (256 x 1024 bf16), in 128-byte rows, round after round, keeping `depth` stages in flight:
//   mode 0: buffer_load_dwordx4 ... lds (1 KB per wave instruction: 8 rows x 128 B)
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: like 0 with 64-byte rows (16 rows x 64 B per instruction)
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/fill_probe.hip -o tools/probe/fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int K = 1024, BN = 256, BM = 64, STAGE = (BM + BN) * 128;   // 40 KB per k-tile of 64

template <int MODE, int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void fill_kernel(const char *A, const char *B, int M, int rounds, unsigned *sink)
{
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass of this hipcc drops the stub of a kernel TEMPLATE whose body names a buffer resource)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int PIECES = STAGE / 1024, PPW = PIECES / NW;          // 40 pieces of 1 KB per stage
    const __amdgpu_buffer_rsrc_t resA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(A), (short)0, (int)((long)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t resB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(B), (short)0, BN * K * 2, 0x00020000);
    const int m0 = blockIdx.x * BM;
    // piece q of a stage: rows [8 q, 8 q + 8) of the (A; B) image (MODE 2: 16 rows of 64 B -- two k-halves side by side)
    unsigned voff[PPW];
    bool isA[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int q = wave * PPW + j;
        int row, chunk;
        if (MODE == 2) { row = (q % 20) * 16 + (lane >> 2); chunk = (lane & 3) + 4 * (q / 20); }
        else { row = q * 8 + (lane >> 3); chunk = (lane & 7) ^ ((row >> 1) & 7); }
        isA[j] = MODE == 2 ? (q % 20) < 4 : q < 8;
        const int r = isA[j] ? min(m0 + row, M - 1) : row - BM;
        voff[j] = (unsigned)((long)r * K * 2 + chunk * 16);
    }
    unsigned acc = 0;
    auto issue = [&](int kt) {
        char *st = smem + (kt % (DEPTH + 1)) * STAGE;
        const int koff = (kt % (K / 64)) * 128;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            if (MODE == 1) {
                const u32x4 v = *(const u32x4 *)((isA[j] ? A : B) + voff[j] + koff);
                *(u32x4 *)(st + (wave * PPW + j) * 1024 + lane * 16) = v;
            } else if (isA[j])
                __builtin_amdgcn_raw_ptr_buffer_load_lds(resA, (__attribute__((address_space(3))) void *)(st + (wave * PPW + j) * 1024), 16, voff[j], koff, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(resB, (__attribute__((address_space(3))) void *)(st + (wave * PPW + j) * 1024), 16, voff[j], koff, 0, 0);
        }
    };
    if (MODE == 1) {
        for (int kt = 0; kt < rounds; ++kt) { issue(kt); __syncthreads(); acc += *(const unsigned *)(smem + (kt % (DEPTH + 1)) * STAGE + tid * 4); }
    } else {
        for (int kt = 0; kt < DEPTH; ++kt) issue(kt);
        for (int kt = 0; kt < rounds; ++kt) {
            // k-tile kt has landed when all but the (DEPTH - 1) younger stages' pieces are done
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            else if (DEPTH == 2) { if (PPW == 10) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory"); }
            else { if (PPW == 10) asm volatile("s_waitcnt vmcnt(20)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory"); }
            issue(kt + DEPTH);
            unsigned v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)((kt % (DEPTH + 1)) * STAGE + tid * 4)) : "memory");
            acc += v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc == 0x12345678u) sink[0] = acc;
#endif
}


template <int MODE, int NW, int DEPTH>
void run(const char *A, const char *B, int M, unsigned *sink, const char *what)
{
    const int rounds = 1600, nwg = M / BM;
    const int lds = (DEPTH + 1) * STAGE;
    auto kern = fill_kernel<MODE, NW, DEPTH>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NW * 64), lds, 0, A, B, M, 64, sink);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NW * 64), lds, 0, A, B, M, rounds, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)nwg * rounds * STAGE;
    printf("%-46s %3d wgs x %d waves, %d stage(s) in flight: %7.3f ms  %6.2f TB/s chip  %6.1f GB/s per CU (= %5.1f B/clk at 2.4 GHz)\n", what, nwg, NW, DEPTH, ms, bytes / ms * 1e-9, bytes / nwg / ms * 1e-6, bytes / nwg / ms * 1e-6 / 2.4);
}

int main()
{
    const int M = 256 * BM;        // one panel per CU
    char *A, *B; unsigned *sink;
    CK(hipMalloc((void **)&A, (size_t)M * K * 2)); CK(hipMalloc((void **)&B, (size_t)BN * K * 2)); CK(hipMalloc((void **)&sink, 64));
    CK(hipMemset(A, 0, (size_t)M * K * 2)); CK(hipMemset(B, 0, (size_t)BN * K * 2));
    run<0, 4, 1>(A, B, M, sink, "LDS-DMA b128, 128-byte rows");
    run<0, 4, 2>(A, B, M, sink, "LDS-DMA b128, 128-byte rows");
    run<0, 4, 3>(A, B, M, sink, "LDS-DMA b128, 128-byte rows");
    run<0, 8, 2>(A, B, M, sink, "LDS-DMA b128, 128-byte rows");
    run<0, 8, 3>(A, B, M, sink, "LDS-DMA b128, 128-byte rows");
    run<2, 4, 2>(A, B, M, sink, "LDS-DMA b128, 64-byte rows");
    run<2, 4, 3>(A, B, M, sink, "LDS-DMA b128, 64-byte rows");
    run<1, 4, 1>(A, B, M, sink, "global_load_dwordx4 + ds_write_b128 (compiled)");
    run<1, 8, 1>(A, B, M, sink, "global_load_dwordx4 + ds_write_b128 (compiled)");
    return 0;
}

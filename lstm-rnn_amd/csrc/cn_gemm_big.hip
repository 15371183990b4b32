// gemm_nt for the MFMA-bound shapes of the path (LVCSR output layer, 512- and 1024-wide BLSTM gate products):
// C[m][n] = sum_k A[m][k] B[n][k] (+ bias) -> act, same contract as gemm_nt_kernel (cn_gemm.hip), which stays the
// kernel of the small-K, output-bound products of the headline workload.
//
// 256 x 256 tile, 8 waves as 2 (M) x 4 (N), each wave 128 x 64 in 32x32 MFMA tiles (128 accumulator registers).
// Operands go global -> LDS directly (global_load_lds_dwordx4, no staging registers): one instruction fills 1 KB
// = 8 tile rows of 128 B, so the LDS image is lane-linear and cannot be padded; the bank spread comes from an XOR
// swizzle applied on the SOURCE address instead -- LDS slot s of row r holds the row's 16-byte chunk s ^ ((r >> 1) & 7),
// which makes every 16-lane service group of the fragment ds_read_b128 hit 16 different 16-byte bank slots.
// Two LDS buffers (128 KB): the fill of k-tile t+1 is in flight while k-tile t is multiplied; one barrier per k-tile.
// (Measured and rejected: four k-tiles of 64-byte rows with counted vmcnt waits and a raw barrier -- fills never drain,
// but a barrier every 16 MFMAs per wave instead of every 32: 646 / 770 / 624 vs 694 / 828 / 693 TFLOP/s on the LVCSR shapes.)
// (Round 2, measured and rejected as well: a four-phase k-tile after cdna_hip_programming.md section 5 -- 8 MFMAs per wave and
// phase, one 16 KB half-tile filled per phase up to 1.75 k-tiles ahead, ONE counted `s_waitcnt vmcnt(6)` per k-tile that leaves
// three half-tiles in flight across every raw s_barrier, s_setprio around the MFMA clusters; 230 VGPRs, no compiler-inserted
// vmcnt(0) in the loop, bit-identical results: 600 / 671 / 690 / 719 / 795 / 664 TFLOP/s against 635 / 669 / 697 / 730 / 796 /
// 686 of this kernel on the six MFMA-bound shapes of tools/probe/gemm_bench.  PMC (tools/pmc_l2.sh): L2 hit rate 0.63-0.72,
// fabric reads 9x the unique operand bytes at ~1.1 TB/s, L2 requests ~8.6 TB/s -- neither is a roof; a k-tile takes 3x its
// MFMA time with either schedule.  The guide's own example source, with its two staggered wave groups, is not available here.)
// The epilogue transposes the accumulators through the (then free) LDS in four passes of 64 rows and writes whole
// 1 KB rows, 16 B per lane, as gemm_nt_kernel does.
#include "cn_internal.h"

namespace cn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

namespace {

__device__ __forceinline__ float big_act(int act, float x)
{
    // activation_functions/Logistic.cuh:33-44, Tanh.cuh:33-36 (as act_apply in cn_gemm.hip)
    if (act == ACT_IDENTITY) return x;
    float z = (act == ACT_TANH) ? 2.0f * x : x;
    float s;
    if (z < 88.722839f) s = (z > -88.722839f) ? 1.0f / (1.0f + __expf(-z)) : 0.0f;
    else s = 1.0f;
    return (act == ACT_TANH) ? 2.0f * s - 1.0f : s;
}

template <bool F32>
__device__ __forceinline__ void big_mma(f32x16 &acc, const u32x4 &a, const u32x4 &b)
{
    if constexpr (F32) {
        const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[i], acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
}

constexpr int BG_BM = 256, BG_BN = 256, BG_ROWB = 128;          // tile, bytes of K per tile row and k-tile
constexpr int BG_OPER = BG_BM * BG_ROWB;                        // one operand k-tile: 32 KB
constexpr int BG_LDS = 4 * BG_OPER;                             // (A, B) x 2 buffers = 128 KB
constexpr int BG_EP = BG_BN * 4 + 16;                           // epilogue staging row pitch
constexpr int BG_GROUP_M = 4;                                   // tile rows per L2 group (1 / 4 / 8 measured: 681 / 710 / 717 TFLOP/s at N = 8000)
static_assert(64 * BG_EP <= BG_LDS, "epilogue staging does not fit");

template <bool F32>
__global__ __launch_bounds__(512) void gemm_nt_big_kernel(GemmNT p, int tiles_n, int nwg)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ELT = F32 ? 4 : 2;
    constexpr int KB = BG_ROWB / ELT;                           // k elements per k-tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 31, fh = lane >> 5;

    int bid = blockIdx.x;
    {   // XCD-aware bijective tile order (see gemm_nt_kernel)
        int q = nwg / 8, r = nwg % 8, x = bid % 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
    }
    // ... and inside an XCD's run, tiles in groups of BG_GROUP_M tile rows walked column by column: the ~32 workgroups
    // an XCD runs at a time then share 4 A panels and 8 B panels in its L2 instead of 1 and 32 (B is the whole weight
    // matrix and does not fit the 4 MB L2: with row-major order every tile row streamed it from the Infinity Cache again)
    const int tiles_m = (p.M + BG_BM - 1) / BG_BM;
    const int per_group = BG_GROUP_M * tiles_n, grp = bid / per_group, first_m = grp * BG_GROUP_M;
    const int gm = min(BG_GROUP_M, tiles_m - first_m), in_grp = bid % per_group;
    const int m0 = (first_m + in_grp % gm) * BG_BM, n0 = (in_grp / gm) * BG_BN;
    const int nk = p.K / KB;                                    // the launcher guarantees K % KB == 0

    // fill: instruction q = 4 * wave + j (j < 4) of an operand covers tile rows [8q, 8q + 8); lane l brings the chunk
    // that belongs in LDS slot l & 7 of row 8q + (l >> 3)
    const char *srcA[4], *srcB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (4 * wave + j) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int ma = min(m0 + row, p.M - 1), nb = min(n0 + row, p.N - 1);      // rows past the edge: results are not stored
        srcA[j] = (const char *)p.A + (long)ma * p.lda * ELT + chunk * 16;
        srcB[j] = (const char *)p.B + (long)nb * p.ldb * ELT + chunk * 16;
    }
    auto fill = [&](int kt, int buf) {                          // 8 LDS-DMA instructions per wave
        char *la = smem + buf * 2 * BG_OPER + (4 * wave) * 1024, *lb = la + BG_OPER;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcA[j] + (long)kt * BG_ROWB),
                                             (__attribute__((address_space(3))) void *)(la + j * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcB[j] + (long)kt * BG_ROWB),
                                             (__attribute__((address_space(3))) void *)(lb + j * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses inside an operand k-tile (bytes): row * 128 + ((chunk ^ swizzle(row)) << 4), chunk = 2 g + fh
    int offA[4], offB[2], swA[4], swB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int row = wm * 128 + i * 32 + fr; offA[i] = row * BG_ROWB; swA[i] = (row >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int row = wn * 64 + j * 32 + fr; offB[j] = row * BG_ROWB; swB[j] = (row >> 1) & 7; }

    fill(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every wave's part of the fill has landed ...
    __syncthreads();                                            // ... before any wave reads the tile
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) fill(kt + 1, (kt + 1) & 1);
        const char *sa = smem + (kt & 1) * 2 * BG_OPER, *sb = sa + BG_OPER;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x4 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *(const u32x4 *)(sa + offA[i] + (((2 * g + fh) ^ swA[i]) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *(const u32x4 *)(sb + offB[j] + (((2 * g + fh) ^ swB[j]) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) big_mma<F32>(acc[i][j], a[i], b[j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                        // next fill landed; this buffer is free for the fill after it
    }

    // epilogue: four passes of 64 rows through LDS (C/D map of the 32x32 MFMA: col = lane & 31,
    // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5))
    const int c4 = lane, n = n0 + c4 * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && n < p.N) bv = *(const f32x4 *)(p.bias + n);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        if (h) __syncthreads();
        if (wm == (h >> 1)) {
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        *(float *)(smem + (i2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * BG_EP + (wn * 64 + j * 32 + fr) * 4) = acc[2 * (h & 1) + i2][j][r];
        }
        __syncthreads();
        if (n < p.N) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = wave + 8 * k, m = m0 + 64 * h + row;
                if (m >= p.M) break;
                f32x4 v = *(const f32x4 *)(smem + row * BG_EP + c4 * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = big_act(p.act, v[e] + bv[e]);
                if (p.C) *(f32x4 *)(p.C + (long)m * p.ldc + n) = v;
                if (p.C2) {
                    if constexpr (F32) *(f32x4 *)((float *)p.C2 + (long)m * p.ldc2 + n) = v;
                    else {
                        const bf16x4 hh = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        *(bf16x4 *)((__bf16 *)p.C2 + (long)m * p.ldc2 + n) = hh;
                    }
                }
            }
        }
    }
}

}  // namespace

// Does this product go to the 256 x 256 kernel?  K in whole k-tiles, and enough tiles to fill the chip about twice
// (smaller outputs keep the 128 x 128 kernel: more, shorter workgroups).
bool gemm_nt_big_applies(int prec, const GemmNT &g)
{
    const bool f32 = prec != P_BF16;     // the LDS-DMA kernel is bf16 only (P_F32 / P_X3 operands are fp32 in memory)
    static const bool off = getenv("CN_NO_BIG_GEMM") != nullptr;
    const int KB = BG_ROWB / (f32 ? 4 : 2);
    if (off || g.K % KB != 0 || g.K < 4 * KB) return false;
    const long tiles = (long)((g.M + BG_BM - 1) / BG_BM) * ((g.N + BG_BN - 1) / BG_BN);
    return tiles >= 384;
}

void launch_gemm_nt_big(hipStream_t s, int prec, const GemmNT &g, hipEvent_t done)
{
    const bool f32 = prec != P_BF16;
    const int tiles_m = (g.M + BG_BM - 1) / BG_BM, tiles_n = (g.N + BG_BN - 1) / BG_BN, nwg = tiles_m * tiles_n;
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void *)gemm_nt_big_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, BG_LDS);
        (void)hipFuncSetAttribute((const void *)gemm_nt_big_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, BG_LDS);
    }
    if (f32) hipExtLaunchKernelGGL(gemm_nt_big_kernel<true>, dim3(nwg), dim3(512), BG_LDS, s, nullptr, done, 0, g, tiles_n, nwg);
    else     hipExtLaunchKernelGGL(gemm_nt_big_kernel<false>, dim3(nwg), dim3(512), BG_LDS, s, nullptr, done, 0, g, tiles_n, nwg);
}

}  // namespace cn

// gemm_nt for the OUTPUT-bound products of the wide layers: C[m][n] = sum_k A[m][k] B[n][k] (+ bias) -> act with a short K
// (the input projections of the 256-wide layers, K = 512, and of the 8000-class output layer: LstmLayer.cu:771-786,
// FeedForwardLayer.cu:143-160 through Matrix.cu:218-239), same contract as gemm_nt_kernel / gemm_nt_big_kernel.
//
// Why another kernel: with K = 512 a 256 x 256 tile is 8 k-tiles of fill (64 KB each: ~13 us at a CU's share of the L2 -> LDS
// path) followed by 256 KB of fp32 result (~16 us at a CU's share of the HBM write rate), and gemm_nt_big_kernel -- ONE workgroup
// per CU, 128 KB of LDS -- does the two one after the other (the persistent kernel that overlaps them needs a dozen k-tiles per
// tile to pay for its seams).  Here a tile is 128 x 256 and a workgroup FOUR waves (1 x 4, each 128 x 64: the same 128 accumulator
// registers per wave), three fill stages of 24 KB: 72 KB of LDS, TWO workgroups per CU -- while one stores its result the other
// multiplies.  The rest is the structure of gemm_tn_big_kernel (cn_gemm_tn_big.hip): fills global -> LDS directly (buffer loads,
// 1 KB = 16 tile rows of 64 bytes per wave instruction, XOR on the source chunk: slot s of row r holds chunk s ^ ((r >> 2) & 3),
// which makes every 16-lane service group of the fragment ds_read_b128 hit 16 different bank slots), counted vmcnt waits that
// leave one fill in flight across the barrier, the k-tile body (12 fragment reads, 16 MFMAs) as ONE asm statement with counted
// lgkmcnt waits and fixed fragment registers (left to hipcc, LDS reads "may alias" the fills in flight and it drains vmcnt in
// front of them).  The epilogue transposes the accumulators through the (then free) LDS in two passes of 64 rows and writes
// whole 1 KB rows, 16 B per lane, bias / activation / operand-type copy on the way out, as gemm_nt_big_kernel does.
#include "cn_internal.h"
#include <algorithm>
#include <cstdint>

namespace cn {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int NM_BM = 128, NM_BN = 256, NM_BK = 32, NM_ST = 3;
constexpr int NM_ROWB = 64;                                     // bytes of K per tile row and k-tile
constexpr int NM_A = NM_BM * NM_ROWB, NM_B = NM_BN * NM_ROWB;   // 8 KB, 16 KB
constexpr int NM_STAGE = NM_A + NM_B;                           // 24 KB
constexpr int NM_LDS = NM_ST * NM_STAGE;                        // 72 KB: two workgroups per CU
constexpr int NM_EP = NM_BN * 4 + 16;                           // epilogue staging row pitch
constexpr int NM_GROUP_M = 8;                                   // tile rows per L2 group
static_assert(64 * NM_EP <= NM_LDS, "epilogue staging does not fit");

__device__ __forceinline__ float mid_act(int act, float x)
{
    // activation_functions/Logistic.cuh:33-44, Tanh.cuh:33-36 (as act_apply in cn_gemm.hip)
    if (act == ACT_IDENTITY) return x;
    float z = (act == ACT_TANH) ? 2.0f * x : x;
    float s;
    if (z < 88.722839f) s = (z > -88.722839f) ? 1.0f / (1.0f + __expf(-z)) : 0.0f;
    else s = 1.0f;
    return (act == ACT_TANH) ? 2.0f * s - 1.0f : s;
}

__global__ __launch_bounds__(256, 2) void gemm_nt_mid_kernel(GemmNT p, int tiles_n, int nwg)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;       // wave = wn: its 64 columns of the tile
    const int fr = lane & 31, fh = lane >> 5;

    int bid = blockIdx.x;
    {   // XCD-aware bijective tile order (see gemm_nt_kernel)
        int q = nwg / 8, r = nwg % 8, x = bid % 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
    }
    const int tiles_m = (p.M + NM_BM - 1) / NM_BM;
    const int per_group = NM_GROUP_M * tiles_n, grp = bid / per_group, first_m = grp * NM_GROUP_M;
    const int gm = min(NM_GROUP_M, tiles_m - first_m), in_grp = bid % per_group;
    const int m0 = (first_m + in_grp % gm) * NM_BM, n0 = (in_grp / gm) * NM_BN;
    const int nk = p.K / NM_BK;                                 // the launcher guarantees K % 32 == 0

    // fill: instruction q of an operand covers tile rows [16 q, 16 q + 16); lane l brings the chunk that belongs in LDS slot l & 3
    // of row 16 q + (l >> 2).  A: 8 instructions (wave w: 2w, 2w + 1), B: 16 (wave w: 4w .. 4w + 3).  Rows past the edge: clamped
    // (their results are not stored).
    auto resource = [](const void *base, long bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, (int)(unsigned)bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t resA = resource(p.A, (long)p.M * p.lda * 2), resB = resource(p.B, (long)p.N * p.ldb * 2);
    unsigned voffA[2], voffB[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 16 * (2 * wave + j) + (lane >> 2), chunk = (lane & 3) ^ ((row >> 2) & 3);
        voffA[j] = (unsigned)((long)min(m0 + row, p.M - 1) * p.lda * 2 + chunk * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 16 * (4 * wave + j) + (lane >> 2), chunk = (lane & 3) ^ ((row >> 2) & 3);
        voffB[j] = (unsigned)((long)min(n0 + row, p.N - 1) * p.ldb * 2 + chunk * 16);
    }
    auto fill = [&](int kt) {                                   // 6 LDS-DMA instructions per wave
        char *la = smem + (kt % NM_ST) * NM_STAGE + (2 * wave) * 1024, *lb = smem + (kt % NM_ST) * NM_STAGE + NM_A + (4 * wave) * 1024;
        const int koff = kt < nk ? kt * NM_ROWB : 0;            // (fills past the last k-tile re-read the first: never consumed)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(resA, (__attribute__((address_space(3))) void *)(la + j * 1024), 16, voffA[j], koff, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(resB, (__attribute__((address_space(3))) void *)(lb + j * 1024), 16, voffB[j], koff, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses inside a stage (bytes): row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4), chunk = 2 s + fh for k-step s;
    // the second k-step's address is the first one's with bit 5 flipped
    int offA[4], offB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int row = i * 32 + fr; offA[i] = row * NM_ROWB + ((fh ^ ((row >> 2) & 3)) << 4); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int row = wave * 64 + j * 32 + fr; offB[j] = NM_A + row * NM_ROWB + ((fh ^ ((row >> 2) & 3)) << 4); }

    fill(0);
    fill(1);
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's part of k-tile kt has landed (the younger fill, 6 instructions, stays in flight); behind the barrier
        // everybody's has, and everybody is done reading k-tile kt - 1, whose stage the next fill overwrites
        asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        fill(kt + 2);
        const int st = (kt % NM_ST) * NM_STAGE;
        const int a0 = st + offA[0], a1 = st + offA[1], a2 = st + offA[2], a3 = st + offA[3], b0 = st + offB[0], b1 = st + offB[1];
        const int a0x = a0 ^ 32, a1x = a1 ^ 32, a2x = a2 ^ 32, a3x = a3 ^ 32, b0x = b0 ^ 32, b1x = b1 ^ 32;
        // fragments (fixed registers, declared clobbered): first k-step b0 200 b1 204 a0 208 a1 212 a2 216 a3 220 | second k-step
        // d0 224 d1 228 c0 232 c1 236 c2 240 c3 244; LDS returns in order
        asm volatile(
            "ds_read_b128 v[200:203], %[b0]\n\t"  "ds_read_b128 v[204:207], %[b1]\n\t"  "ds_read_b128 v[208:211], %[a0]\n\t"
            "ds_read_b128 v[212:215], %[a1]\n\t"  "ds_read_b128 v[216:219], %[a2]\n\t"  "ds_read_b128 v[220:223], %[a3]\n\t"
            "s_waitcnt lgkmcnt(3)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c00], v[208:211], v[200:203], %[c00]\n\t"
            "ds_read_b128 v[224:227], %[b0x]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c01], v[208:211], v[204:207], %[c01]\n\t"
            "ds_read_b128 v[228:231], %[b1x]\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c10], v[212:215], v[200:203], %[c10]\n\t"
            "ds_read_b128 v[232:235], %[a0x]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c11], v[212:215], v[204:207], %[c11]\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c20], v[216:219], v[200:203], %[c20]\n\t"
            "ds_read_b128 v[236:239], %[a1x]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c21], v[216:219], v[204:207], %[c21]\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c30], v[220:223], v[200:203], %[c30]\n\t"
            "ds_read_b128 v[240:243], %[a2x]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c31], v[220:223], v[204:207], %[c31]\n\t"
            "ds_read_b128 v[244:247], %[a3x]\n\t"
            "s_waitcnt lgkmcnt(3)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c00], v[232:235], v[224:227], %[c00]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c01], v[232:235], v[228:231], %[c01]\n\t"
            "s_waitcnt lgkmcnt(2)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c10], v[236:239], v[224:227], %[c10]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c11], v[236:239], v[228:231], %[c11]\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c20], v[240:243], v[224:227], %[c20]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c21], v[240:243], v[228:231], %[c21]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c30], v[244:247], v[224:227], %[c30]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c31], v[244:247], v[228:231], %[c31]\n\t"
            : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]),
              [c20] "+v"(acc[2][0]), [c21] "+v"(acc[2][1]), [c30] "+v"(acc[3][0]), [c31] "+v"(acc[3][1])
            : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [b0] "v"(b0), [b1] "v"(b1),
              [a0x] "v"(a0x), [a1x] "v"(a1x), [a2x] "v"(a2x), [a3x] "v"(a3x), [b0x] "v"(b0x), [b1x] "v"(b1x)
            : "memory", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215",
              "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231",
              "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247");
    }
    // (the accumulators were last written inside an asm statement: the compiler's hazard recognizer has not seen those MFMAs;
    // the fills issued past the end must not outlive the stage they target, which the epilogue reuses)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    // epilogue: two passes of 64 rows through LDS (C/D map of the 32x32 MFMA: col = lane & 31,
    // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5))
    const int c4 = lane, n = n0 + c4 * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && n < p.N) bv = *(const f32x4 *)(p.bias + n);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *(float *)(smem + (i2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * NM_EP + (wave * 64 + j * 32 + fr) * 4) = acc[2 * h + i2][j][r];
        __syncthreads();
        if (n < p.N) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int row = wave + 4 * k, m = m0 + 64 * h + row;
                if (m >= p.M) break;
                f32x4 v = *(const f32x4 *)(smem + row * NM_EP + c4 * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = mid_act(p.act, v[e] + bv[e]);
                if (p.C) *(f32x4 *)(p.C + (long)m * p.ldc + n) = v;
                if (p.C2) {
                    const bf16x4 hh = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *(bf16x4 *)((__bf16 *)p.C2 + (long)m * p.ldc2 + n) = hh;
                }
            }
        }
    }
}

}  // namespace

// bf16 products with a short K (whole k-tiles of 32, below 768: from a dozen 64-wide k-tiles on the persistent 256 x 256 kernel
// overlaps its stores by itself) and an output large enough for the 256 x 256 path today (>= 384 such tiles), operands addressable
// with 32-bit offsets, 16-byte result rows.  Not the 8000-class output layer: a 128 x 256 tile fills 1.5 x the bytes of a
// 256 x 256 one per result, and at N = 8000, K = 512 the L2 -> LDS path is what bounds it (745 us against 702 us; the layer
// products, N = 2048: 133 against 142 us at M = 35200, 205 against 208 us at M = 51200 -- profiles/r05c_gemm_nt_mid.md)
bool gemm_nt_mid_applies(int prec, const GemmNT &g)
{
    const bool off = opt().no_nt_mid;
    if (off || prec != P_BF16) return false;
    if (g.K % NM_BK != 0 || g.K < 4 * NM_BK || g.K >= 768 || g.N > 4096) return false;
    if (g.N % 4 != 0 || (g.C && (g.ldc % 4 || (uintptr_t)g.C % 16)) || (g.C2 && (g.ldc2 % 4 || (uintptr_t)g.C2 % 8)) || (uintptr_t)g.A % 16 || (uintptr_t)g.B % 16 || (g.bias && (uintptr_t)g.bias % 16) || !(g.C || g.C2) || g.lda % 8 || g.ldb % 8) return false;
    if ((unsigned long long)g.M * g.lda * 2 >= 0xfffffff0ull || (unsigned long long)g.N * g.ldb * 2 >= 0xfffffff0ull) return false;
    // (from 1000 tiles on and K >= 512 the persistent 256 x 256 kernel takes the product: launch_gemm_nt_big)
    const long tiles = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
    return tiles >= opt().nt_mid_min_tiles && (tiles < 1000 || g.K < 512);
}

void launch_gemm_nt_mid(hipStream_t s, const GemmNT &g, hipEvent_t done)
{
    const int tiles_m = (g.M + NM_BM - 1) / NM_BM, tiles_n = (g.N + NM_BN - 1) / NM_BN, nwg = tiles_m * tiles_n;
    static DeviceOnce attr_once;
    if (attr_once.first()) (void)hipFuncSetAttribute((const void *)gemm_nt_mid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NM_LDS);
    hipExtLaunchKernelGGL(gemm_nt_mid_kernel, dim3(nwg), dim3(256), NM_LDS, s, nullptr, done, 0, g, tiles_n, nwg);
}

}  // namespace cn

// gemm_tn for the weight-gradient products whose operands do not fit the caches (LVCSR, reading B, long utterances):
// C[m][n] += sum_k A[k][m] B[k][n], k = frames (ComputeWeightUpdateFn, LstmLayer.cu:289-512; FeedForwardLayer.cu:200-207),
// same contract as gemm_tn_kernel (cn_gemm.hip), which keeps the small products of the headline workload.
//
// Why another kernel: with 64 x 64 (128 x 128) tiles every operand byte is requested M/64 resp. N/64 times -- the LVCSR layer
// product (delta [35 200][2048], x [35 200][512]) moved 3.4 GB through the L2 -> LDS fill path for 216 MB of operands and ran at
// that path's ceiling (~7 TB/s), 186 TFLOP/s.  256 x 256 tiles ask for a quarter of that.
//
// 256 x 256 tile, 8 waves as 2 (M) x 4 (N), each wave 128 x 64 in 32x32x16 MFMA tiles (128 accumulator registers), bf16 only.
// Both operands are K-MAJOR in memory (a tile row = one frame = 512 contiguous bytes) and stay K-major in LDS: they go
// global -> LDS directly (global_load_lds_dwordx4: one wave instruction fills 1 KB = two tile rows, lane-linear, so the image
// cannot be padded) and the MFMA fragments are read with ds_read_b64_tr_b16 (hardware transpose: a 16-lane group reads 4 k-rows
// x 64 B).  Bank spread by an XOR applied to the SOURCE address: LDS slot s (16 bytes) of tile row r holds the row's chunk
// s ^ ((r & 3) << 2), so the four rows a 32-lane half reads land in four different 64-byte bank windows (conflict-free; without
// it the read is 4-way).  k-tiles of 32 frames (32 KB for both operands), FOUR stages: three fills in flight while one k-tile
// is multiplied, one counted `s_waitcnt vmcnt(8)` + one raw barrier per k-tile, fills never drained.
// Fills are buffer loads whose resource ends at the split's last frame: rows past the end of K come back as zeros (the fill
// cannot mask lanes), so every k-tile is whole; columns past M / N are clamped (their products land in rows / columns of the
// tile that are not stored).  Split-K over frames; an XCD runs a contiguous run of the (split, tile) pairs, so the tiles it works
// on side by side walk the same frames: a panel is fetched into that L2 once and served to the tiles of its row / column from
// there.  fp32 atomics into the pre-zeroed gradient like gemm_tn_kernel.
#include "cn_internal.h"
#include <algorithm>
#include <cstdint>

namespace cn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int TB_BM = 256, TB_BN = 256, TB_BK = 32, TB_ST = 4;
constexpr int TB_ROWB = 512;                                   // bytes of a tile row (256 bf16)
constexpr int TB_OPER = TB_BK * TB_ROWB;                       // one operand k-tile: 16 KB
constexpr int TB_STAGE = 2 * TB_OPER;                          // A + B
constexpr int TB_LDS = TB_ST * TB_STAGE;                       // 128 KB
constexpr int TB_GROUP = 3;

struct TnBigGroup {
    GemmTN p[TB_GROUP];
    int tiles_n[TB_GROUP], ntiles[TB_GROUP], kchunk[TB_GROUP], splits[TB_GROUP], first_block[TB_GROUP + 1];
};

__global__ __launch_bounds__(512) void gemm_tn_big_kernel(TnBigGroup grp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int gi = 0;
#pragma unroll
    for (int i = 1; i < TB_GROUP; ++i) if ((int)blockIdx.x >= grp.first_block[i]) gi = i;
    const GemmTN p = grp.p[gi];
    const int ntiles = grp.ntiles[gi], tiles_n = grp.tiles_n[gi], kchunk = grp.kchunk[gi];
    const int local = blockIdx.x - grp.first_block[gi];
    // The (split, tile) pairs in split-major order, cut into eight equal runs, one per XCD (a product's block count is a multiple
    // of 8, so blockIdx % 8 -- the XCD under round-robin placement -- is local % 8): every XCD gets the same number of tiles, and
    // the tiles it runs side by side belong to one or two splits, tile rows together.
    const int items = ntiles * grp.splits[gi], per = (items + 7) / 8;
    const int item = (local % 8) * per + local / 8;
    if (local / 8 >= per || item >= items) return;
    const int split = item / ntiles, tile = item % ntiles;
    const int m0 = (tile / tiles_n) * TB_BM, n0 = (tile % tiles_n) * TB_BN;
    const int kbeg = split * kchunk, kend = min(p.K, kbeg + kchunk);
    if (kbeg >= kend) return;
    const int nk = (kend - kbeg + TB_BK - 1) / TB_BK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;

    // fill: wave w brings tile rows 4w .. 4w+3 of both operands, two rows per instruction; lane l of instruction j brings the
    // chunk that belongs in LDS slot l & 31 of row 4w + 2j + (l >> 5).  Buffer loads: a 32-bit byte offset per lane off a
    // resource that ends behind frame kend - 1 (the launcher checks that the operands are smaller than 4 GB).
    auto resource = [](const void *base, long bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, (int)(unsigned)bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t resA = resource(p.A, (long)kend * p.lda * 2), resB = resource(p.B, (long)kend * p.ldb * 2);
    unsigned voffA[2], voffB[2];
    const unsigned stepA = (unsigned)(TB_BK * p.lda * 2), stepB = (unsigned)(TB_BK * p.ldb * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 4 * wave + 2 * j + (lane >> 5);
        const int chunk = (lane & 31) ^ ((row & 3) << 2);
        const int ca = m0 + 8 * chunk < p.M ? m0 + 8 * chunk : 0, cb = n0 + 8 * chunk < p.N ? n0 + 8 * chunk : 0;   // (M, N multiples of 8)
        voffA[j] = (unsigned)(((long)(kbeg + row) * p.lda + ca) * 2);
        voffB[j] = (unsigned)(((long)(kbeg + row) * p.ldb + cb) * 2);
    }
    auto fill = [&](int kt) {                                  // 4 LDS-DMA instructions per wave; k-tiles are filled in order
        char *la = smem + (kt % TB_ST) * TB_STAGE + (4 * wave) * TB_ROWB, *lb = la + TB_OPER;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(resA, (__attribute__((address_space(3))) void *)(la + j * 2 * TB_ROWB), 16, voffA[j], 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(resB, (__attribute__((address_space(3))) void *)(lb + j * 2 * TB_ROWB), 16, voffB[j], 0, 0, 0);
            voffA[j] += stepA; voffB[j] += stepB;              // (past the last frame: out of the resource's range, zeros)
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment reads (ds_read_b64_tr_b16): lane 4q+p of 16-lane group g16 supplies row (k step) + 8 (g16 >> 1) + 4 jj + q,
    // columns base + 16 (g16 & 1) + 4p .. +3 and receives column base + 16 (g16 & 1) + idx of those four rows = its MFMA
    // operand elements 4 jj .. 4 jj + 3 (A: m = lane & 31, k half = lane >> 5; B likewise).  Byte offset of a row's chunk ch:
    // row * 512 + ((ch ^ ((row & 3) << 2)) << 4); (row & 3) = q for every read of a lane.
    const int g16 = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
    const int rbase = (8 * (g16 >> 1) + q) * TB_ROWB + 8 * (pp & 1);
    int offA[4], offB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = (wm * 128 + i * 32 + 16 * (g16 & 1) + 4 * pp) / 8;
        offA[i] = rbase + ((ch ^ (q << 2)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = (wn * 64 + j * 32 + 16 * (g16 & 1) + 4 * pp) / 8;
        offB[j] = TB_OPER + rbase + ((ch ^ (q << 2)) << 4);
    }

#pragma unroll
    for (int s = 0; s < TB_ST - 1; ++s) fill(s);
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's part of k-tile kt has landed (the two younger fills, 8 instructions, stay in flight); behind the barrier
        // everybody's has, and everybody is done reading k-tile kt - 1, whose stage the next fill overwrites
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        fill(kt + TB_ST - 1);
        const int st = (kt % TB_ST) * TB_STAGE;
        int adA[4], adB[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) adA[i] = st + offA[i];
#pragma unroll
        for (int j = 0; j < 2; ++j) adB[j] = st + offB[j];
        // One k-tile = two k-steps of 16: 24 transposed fragment reads, 16 MFMAs, written out as ONE statement.  Left to hipcc the
        // reads are LDS accesses that "may alias" the fills in flight, and it drains vmcnt in front of them (the three-deep fill
        // pipeline became a fill-and-wait loop); inside the statement the reads carry counted lgkmcnt waits (LDS returns in order;
        // at most 12 in flight): the first k-step's B fragments and A fragment 0 open the MFMAs, every later fragment is waited
        // for right in front of its first MFMA, and the second k-step's reads are issued between the MFMAs of the first.
        // Offsets: k rows 0 / 4 of a k-step are 0 / 2048 bytes, the second k-step starts at 8192.
        // Fragment registers are fixed (v200 .. v247, declared clobbered): an MFMA operand is a 4-register tuple filled by two
        // 2-register reads, and inline-asm operands cannot be addressed by halves.
        // b0 200 b1 204 a0 208 a1 212 a2 216 a3 220 (first k-step) | d0 224 d1 228 c0 232 c1 236 c2 240 c3 244 (second)
        asm volatile(
            "ds_read_b64_tr_b16 v[200:201], %[pB0]\n\t" "ds_read_b64_tr_b16 v[202:203], %[pB0] offset:2048\n\t"
            "ds_read_b64_tr_b16 v[204:205], %[pB1]\n\t" "ds_read_b64_tr_b16 v[206:207], %[pB1] offset:2048\n\t"
            "ds_read_b64_tr_b16 v[208:209], %[pA0]\n\t" "ds_read_b64_tr_b16 v[210:211], %[pA0] offset:2048\n\t"
            "ds_read_b64_tr_b16 v[212:213], %[pA1]\n\t" "ds_read_b64_tr_b16 v[214:215], %[pA1] offset:2048\n\t"
            "ds_read_b64_tr_b16 v[216:217], %[pA2]\n\t" "ds_read_b64_tr_b16 v[218:219], %[pA2] offset:2048\n\t"
            "ds_read_b64_tr_b16 v[220:221], %[pA3]\n\t" "ds_read_b64_tr_b16 v[222:223], %[pA3] offset:2048\n\t"
            "s_waitcnt lgkmcnt(6)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c00], v[208:211], v[200:203], %[c00]\n\t"
            "ds_read_b64_tr_b16 v[224:225], %[pB0] offset:8192\n\t" "ds_read_b64_tr_b16 v[226:227], %[pB0] offset:10240\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c01], v[208:211], v[204:207], %[c01]\n\t"
            "ds_read_b64_tr_b16 v[228:229], %[pB1] offset:8192\n\t" "ds_read_b64_tr_b16 v[230:231], %[pB1] offset:10240\n\t"
            "s_waitcnt lgkmcnt(8)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c10], v[212:215], v[200:203], %[c10]\n\t"
            "ds_read_b64_tr_b16 v[232:233], %[pA0] offset:8192\n\t" "ds_read_b64_tr_b16 v[234:235], %[pA0] offset:10240\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c11], v[212:215], v[204:207], %[c11]\n\t"
            "s_waitcnt lgkmcnt(8)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c20], v[216:219], v[200:203], %[c20]\n\t"
            "ds_read_b64_tr_b16 v[236:237], %[pA1] offset:8192\n\t" "ds_read_b64_tr_b16 v[238:239], %[pA1] offset:10240\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c21], v[216:219], v[204:207], %[c21]\n\t"
            "s_waitcnt lgkmcnt(8)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c30], v[220:223], v[200:203], %[c30]\n\t"
            "ds_read_b64_tr_b16 v[240:241], %[pA2] offset:8192\n\t" "ds_read_b64_tr_b16 v[242:243], %[pA2] offset:10240\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c31], v[220:223], v[204:207], %[c31]\n\t"
            "ds_read_b64_tr_b16 v[244:245], %[pA3] offset:8192\n\t" "ds_read_b64_tr_b16 v[246:247], %[pA3] offset:10240\n\t"
            "s_waitcnt lgkmcnt(6)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c00], v[232:235], v[224:227], %[c00]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c01], v[232:235], v[228:231], %[c01]\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c10], v[236:239], v[224:227], %[c10]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c11], v[236:239], v[228:231], %[c11]\n\t"
            "s_waitcnt lgkmcnt(2)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c20], v[240:243], v[224:227], %[c20]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c21], v[240:243], v[228:231], %[c21]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c30], v[244:247], v[224:227], %[c30]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c31], v[244:247], v[228:231], %[c31]\n\t"
            : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]),
              [c20] "+v"(acc[2][0]), [c21] "+v"(acc[2][1]), [c30] "+v"(acc[3][0]), [c31] "+v"(acc[3][1])
            : [pA0] "v"(adA[0]), [pA1] "v"(adA[1]), [pA2] "v"(adA[2]), [pA3] "v"(adA[3]), [pB0] "v"(adB[0]), [pB1] "v"(adB[1])
            : "memory", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247");
    }
    // (the accumulators were last written inside an asm statement: the compiler's hazard recognizer has not seen those MFMAs)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the fills issued past the end (zeros) must not outlive the workgroup's LDS

    // split-K: fp32 atomics (C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
    const int fr = lane & 31, fh = lane >> 5;
    const bool whole = m0 + TB_BM <= p.M;
    // deterministic mode: this split's partial is stored to its own copy of C and folded in split order afterwards (launch_fold)
    float *const cbase = p.ws ? p.ws + (long)split * p.M * p.ldc : p.C;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mb = m0 + wm * 128 + i * 32 + 4 * fh;
            float *c0 = cbase + (long)mb * p.ldc + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = (r & 3) + 8 * (r >> 2);
                if (whole || mb + dm < p.M) {
                    if (p.ws) c0[(long)dm * p.ldc] = acc[i][j][r];
                    else atomicAdd(c0 + (long)dm * p.ldc, acc[i][j][r]);
                }
            }
        }
    }
}

}  // namespace

// Can this product run on the 256 x 256 kernel?  bf16 operands, 16-byte aligned rows, 32-bit fill offsets, an output of at least two
// tiles whose last tile column is at least three quarters full (N = 64 / 128 products keep the small tiles), and enough frames
// for the operands to be a fill-path problem at all.
bool gemm_tn_big_can(int prec, const GemmTN &g)
{
    const bool off = opt().no_big_tn;
    if (off || prec != P_BF16) return false;
    if (g.lda % 8 || g.ldb % 8 || g.M % 8 || g.N % 8 || (uintptr_t)g.A % 16 || (uintptr_t)g.B % 16) return false;
    if (g.K < 4096 || g.M < 512) return false;
    if ((unsigned long long)g.K * g.lda * 2 >= 0xfffffff0ull || (unsigned long long)g.K * g.ldb * 2 >= 0xfffffff0ull) return false;   // 32-bit fill offsets
    const int rem = g.N % TB_BN;
    return g.N >= 192 && (rem == 0 || rem >= 192);
}
// ... and should it, on its own?  From 2^20 outputs on (2048 x 512: the 256-wide layers' dW_in): with fewer tiles the chip is
// filled by splits, every split pays M*N atomics, and the small tiles' re-reads still fit the caches (headline dW_in 1024 x 256:
// 61.6 us here against 46.0 us in the grouped 64 x 64 launch; dW_rec 1024 x 256 of the 256-wide layers: 72 against 68 us --
// such a product only rides along in the launch of a larger one, launch_gemm_tn_group).
bool gemm_tn_big_applies(int prec, const GemmTN &g)
{
    return gemm_tn_big_can(prec, g) && (long)g.M * g.N >= (1L << 20);
}

void launch_gemm_tn_big_group(hipStream_t s, const GemmTN *gs, int n, int cu_budget, const FoldItem *extra)
{
    if (n <= 0) { if (extra) launch_fold(s, extra, 1); return; }
    static DeviceOnce attr_once;
    static int cus = 256;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void *)gemm_tn_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS);
        int dev = 0; (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    }
    TnBigGroup grp{};
    FoldItem fold[TB_GROUP + 1]; int nfold = 0;
    long all_tiles = 0;
    for (int i = 0; i < n; ++i) all_tiles += (long)((gs[i].M + TB_BM - 1) / TB_BM) * ((gs[i].N + TB_BN - 1) / TB_BN);
    // The budget is a promise to the recurrent kernel beside this launch (cn_api.cpp: on_side): its cluster grid must find its CUs
    // free, and a 128 KB-LDS workgroup of this kernel owns a CU.  Splits are cut to the budget below; a group whose TILES alone
    // exceed it would still put a workgroup on every CU of the chip, so it runs on the small tiles instead (several per CU, no
    // whole-CU claim).  (Does not happen for the shipped workloads: 24 / 64 tiles against budgets of 100+.)
    if (cu_budget > 0 && all_tiles > std::min(cu_budget, cus)) { launch_gemm_tn_small_group(s, P_BF16, gs, n, extra); return; }
    int blocks = 0;
    for (int i = 0; i < TB_GROUP; ++i) {
        grp.first_block[i] = blocks;
        if (i >= n) { grp.first_block[i] = 0x7fffffff; continue; }
        const GemmTN &g = gs[i];
        const int tiles_m = (g.M + TB_BM - 1) / TB_BM, tiles_n = (g.N + TB_BN - 1) / TB_BN, ntiles = tiles_m * tiles_n;
        // one workgroup per CU (128 KB of LDS): splits so that the group fills the chip once; every split ends in M*N fp32
        // atomics (~1.3 TB/s chip-wide): at most 64 MB of them per product, and at least 16 k-tiles per split
        const int target_env = (int)opt().tnbig_blocks;
        const int target = target_env ? target_env : (cu_budget > 0 ? std::min(cu_budget, cus) : cus);
        int splits = (int)std::max(1L, target / all_tiles);
        const long cap_atomic = std::max(1L, (64L << 20) / ((long)g.M * g.N * 4));
        if (!g.ws) splits = (int)std::min<long>(splits, cap_atomic);
        else splits = std::min(splits, g.ws_splits);
        splits = std::min(splits, std::max(1, g.K / (16 * TB_BK)));
        int kchunk = ((g.K + splits - 1) / splits + TB_BK - 1) / TB_BK * TB_BK;
        splits = (g.K + kchunk - 1) / kchunk;
        grp.p[i] = g; grp.tiles_n[i] = tiles_n; grp.ntiles[i] = ntiles; grp.kchunk[i] = kchunk; grp.splits[i] = splits;
        blocks += 8 * ((ntiles * splits + 7) / 8);
        if (g.ws && g.ws_used) *g.ws_used = splits;
        else if (g.ws) fold[nfold++] = FoldItem{g.C, g.ws, (long)g.M * g.ldc, splits, g.M, g.N, (int)g.ldc, 0, 0};
    }
    grp.first_block[TB_GROUP] = blocks;
    hipLaunchKernelGGL(gemm_tn_big_kernel, dim3(blocks), dim3(512), TB_LDS, s, grp);
    if (extra) fold[nfold++] = *extra;
    if (nfold) launch_fold(s, fold, nfold);
}

}  // namespace cn

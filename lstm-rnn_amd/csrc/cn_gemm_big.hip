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
#include <type_traits>
#include <algorithm>
#include <cstdint>

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


// ---------------------------------------------------------------------------------------------
// bf16: persistent workgroups, four phases per k-tile, two wave groups half a phase apart
// ---------------------------------------------------------------------------------------------
// Same tile and wave layout as above.  What differs:
//  * a k-tile is four PHASES, each {fragment ds_reads + fill instructions, barrier, 8 MFMAs, barrier}; the waves of the
//    second tile row (wr = 1: the second wave of every SIMD) run one barrier behind the first, so on every SIMD one wave
//    multiplies while the other reads and fills.  Fills are never drained: counted `s_waitcnt vmcnt(N)` retire a piece four to
//    five phases after its issue and leave three or four pieces in flight across every barrier (cdna_hip_programming.md
//    section 5, "The 256^2 8-phase template", re-derived: its source is not here).
//  * one workgroup per CU walks its tiles (the XCD-aware order of the kernel above, 32 slots per XCD) and the fill stream runs
//    on across tile seams: the first k-tiles of the next tile are in flight while the last ones of this tile are multiplied.
//  * the result of a tile leaves during the FIRST k-tile of the next one: its MFMAs start from a zero C operand, and each phase
//    first stores the two 32 x 32 accumulator blocks it is about to overwrite -- transposed through 4 KB of wave-private LDS
//    (rows swapped in pairs, r ^ (r >> 2 & 1), conflict-free both ways) into 16-byte stores of 8 rows x 128 B.  The stores count
//    in vmcnt like the fills, in order, so the three k-tiles behind a seam wait with larger counts (`B8_W*` below).
//
// A wave's 128 x 64 output is four quadrants of 64 x 32 walked (lo,lo) (lo,hi) (hi,hi) (hi,lo):
//    phase 0: wait B-hi(T);          read A-lo (8 ds_read_b128), B-lo (4: kept for phase 3)
//    phase 1: wait A-hi(T);          read B-hi (4);   fill A-hi of k-tile T+1
//    phase 2:                        read A-hi (8);   fill A-lo of k-tile T+2
//    phase 3: wait A-lo, B-lo(T+1);                   fill B-lo, B-hi of k-tile T+2
// The fill pieces are those four row sets over all waves (128 tile rows, 16 KB each): A-lo = rows [0,64) + [128,192), A-hi the
// rest; B-lo = the first 32 rows of every wave column's 64.  A piece is overwritten two or three phases after its last read
// (reads retire at the head of the reading phase's multiply section; the other group's fill is issued at least a barrier
// later); a piece is read one phase after BOTH groups waited for it (waits in the read section, barrier, next read section).
// LDS: [A buf 0][A buf 1][B buf 0][B buf 1] 32 KB each, rows of 128 B with the XOR swizzle described at the top, then 8 x 4 KB
// of transposition staging.
constexpr int B8_BUF = BG_OPER, B8_BREG = 2 * BG_OPER, B8_XP = 4 * BG_OPER;
constexpr int B8_LDS = B8_XP + 8 * 4096;
}  // namespace
#ifdef B8_STAMP
// tools/probe/gemm_bench -DB8_STAMP: cycles per phase and segment, summed over the k loop, for waves 0 and 4 of two workgroups:
// [wg][wave group][phase][segment: counted wait, read issue + barrier, operand wait, multiply section, barrier]
__device__ unsigned g_b8_stamps[2][2][4][5];
__device__ unsigned g_b8_clock[3];
__device__ unsigned g_b8_kind[2][4];      // cycles per k-tile kind (normal, first, second, third), wave groups 0 / 1 of workgroup 0      // s_memtime cycles and s_memrealtime ticks (100 MHz) over the k loop of workgroup 0
#define B8_T(x) const unsigned x = (unsigned)__builtin_amdgcn_s_memtime()
#else
#define B8_T(x)
#endif
namespace {

constexpr int b8_cap(int n) { return n > 63 ? 63 : n; }         // vmcnt is six bits: a smaller count only waits longer
// What load_bias() fetches for a product without a bias (K8, K13: bias == nullptr): the load must exist (it counts in vmcnt), so
// it reads a word that holds 0.f.
__device__ const float b8_zero_word[1] = {0.f};

template <int OUTS>                                             // 1: C or C2, 2: both
__global__ __launch_bounds__(512) void gemm_nt_big8_kernel(GemmNT p, int tiles_n, int nwg)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 31, fh = lane >> 5;
    const int nk = p.K / 64;

    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    // this workgroup's tiles: slot s of XCD x takes every nslots-th tile of the XCD's run (see gemm_nt_big_kernel)
    const int tiles_m = (p.M + BG_BM - 1) / BG_BM, per_group = BG_GROUP_M * tiles_n;
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8, nslots = gridDim.x / 8;
    const int run_q = nwg / 8, run_r = nwg % 8;
    const int run_n = run_q + (xcd < run_r ? 1 : 0);
    const int run_0 = xcd < run_r ? xcd * (run_q + 1) : run_r * (run_q + 1) + (xcd - run_r) * run_q;
    if (slot >= run_n) return;
    const int ntl = (run_n - slot + nslots - 1) / nslots;
    auto tile_at = [&](int i, int &m0, int &n0) {
        const int bid = run_0 + slot + i * nslots;
        const int grp = bid / per_group, first_m = grp * BG_GROUP_M;
        const int gm = min(BG_GROUP_M, tiles_m - first_m), in_grp = bid - grp * per_group;
        m0 = (first_m + in_grp % gm) * BG_BM; n0 = (in_grp / gm) * BG_BN;
    };

    // fill: instruction jj (0, 1) of this wave covers rows [8 g8, 8 g8 + 8) of a piece, g8 = 2 * wave + jj; buffer loads:
    // a 32-bit byte offset per lane off the operand's resource, the k-tile as the scalar offset.
    // pieces: 0 A-lo, 1 B-hi, 2 A-hi, 3 B-lo
    auto resource = [](const void *base, long bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, (int)(unsigned)(bytes > 0xfffffff0l ? 0xfffffff0l : bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t resA = resource(p.A, (long)p.M * p.lda * 2), resB = resource(p.B, (long)p.N * p.ldb * 2);
    int prow[4][2], dst[4][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int pr0 = 8 * (2 * wave + jj);
        const int ra = pr0 < 64 ? pr0 : pr0 + 64;                 // A-lo tile row (A-hi: + 64)
        const int rb = (pr0 >> 5) * 64 + (pr0 & 31);              // B-lo tile row (B-hi: + 32)
        const int rows[4] = {ra, rb + 32, ra + 64, rb};
#pragma unroll
        for (int o = 0; o < 4; ++o) { prow[o][jj] = rows[o]; dst[o][jj] = ((o & 1) ? B8_BREG : 0) + rows[o] * BG_ROWB; }
    }
    // (the lane-derived constants of the rarely run paths are recomputed where they are used: kept live across the k loop they spill)
    auto fresh_lane = []() { int l; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l)); return l; };
    int voff[4][2];                                             // per piece: the tile its NEXT fill belongs to
    auto fill_offsets = [&](auto Oc, int m0, int n0) {
        constexpr int O = decltype(Oc)::value;
        const int l = fresh_lane(), lrow = l >> 3, lchunk = l & 7;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int row = prow[O][jj] + lrow;
            const int chunk = lchunk ^ ((row >> 1) & 7);
            if (O & 1) voff[O][jj] = (int)((unsigned)min(n0 + row, p.N - 1) * (unsigned)p.ldb * 2u + chunk * 16);
            else       voff[O][jj] = (int)((unsigned)min(m0 + row, p.M - 1) * (unsigned)p.lda * 2u + chunk * 16);
        }
    };
    int nm0 = 0, nn0 = 0;                                       // the tile the fill stream crosses into next
    // piece O of k-tile kt + lead into buffer buf; the first fill past the end of this tile's k range moves the piece to the next tile
    auto stage = [&](auto Oc, int buf, int ktl) {
        constexpr int O = decltype(Oc)::value;
        if (ktl == nk) fill_offsets(Oc, nm0, nn0);
        const int kt = ktl >= nk ? ktl - nk : ktl;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds((O & 1) ? resB : resA, (__attribute__((address_space(3))) void *)(smem + dst[O][jj] + buf * B8_BUF),
                                                     16, voff[O][jj], kt * BG_ROWB, 0, 0);
    };

    f32x16 acc[4][2];
    // fragment addresses: the swizzle term (row >> 1) & 7 is the same for every 32-row fragment of a lane; the buffer is the
    // 32 KB bit, toggled after every k-tile
    const int sw = (fr >> 1) & 7;
    int adrA[4], adrB[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        adrA[g] = (wr * 128 + fr) * BG_ROWB + (((2 * g + fh) ^ sw) << 4);
        adrB[g] = B8_BREG + (wc * 64 + fr) * BG_ROWB + (((2 * g + fh) ^ sw) << 4);
    }
    u32x4 aF[2][4], bLo[4], bHi[4];

    // ---- result path
    const __amdgpu_buffer_rsrc_t resC = resource(p.C, p.C ? (long)p.M * p.ldc * 4 : 0), resC2 = resource(p.C2, p.C2 ? (long)p.M * p.ldc2 * 2 : 0);
    char *const xp = smem + B8_XP + wave * 4096;
    // accumulator register r of a lane is tile row (r & 3) + 8 (r >> 2) + 4 fh, column fr; staged at row ^ fh
    float biasv[2] = {0.f, 0.f};
    const bool has_bias = p.bias != nullptr;
    // the two accumulator blocks acc[ib][j], acc[ib + 1][j] of the tile at (pm0, pn0); !live: same instruction count, nothing written
    auto store_blocks = [&](auto IBc, auto Jc, int pm0, int pn0, bool live) {
        constexpr int ib = decltype(IBc)::value, j = decltype(Jc)::value;
        const int l = fresh_lane(), lrow = l >> 3, lchunk = l & 7, xfr = l & 31, xfh = l >> 5;
        const int xw_e = xfr * 4 + xfh * 5 * 128, xw_o = xfr * 4 + xfh * 3 * 128;   // even / odd r: (row + 4) ^ 1 = row + 5 / row + 3
        const int xr = ((lrow ^ xfh) * 128) + lchunk * 16;                          // read back: staged row (lrow + 8 k) ^ fh
        const unsigned vC = (unsigned)(lrow * (int)p.ldc + lchunk * 4) * 4u, vC2 = (unsigned)(lrow * (int)p.ldc2 + lchunk * 4) * 2u;
        const bool ok = live && pn0 + wc * 64 + j * 32 + lchunk * 4 < p.N;
#pragma unroll
        for (int i = ib; i < ib + 2; ++i) {
            if (p.act == ACT_IDENTITY) {                          // (one uniform branch per block instead of one per element)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *(float *)(xp + ((r & 1) ? xw_o : xw_e) + ((r & 3) + 8 * (r >> 2)) * 128) = acc[i][j][r] + biasv[j];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *(float *)(xp + ((r & 1) ? xw_o : xw_e) + ((r & 3) + 8 * (r >> 2)) * 128) = big_act(p.act, acc[i][j][r] + biasv[j]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 v = *(const f32x4 *)(xp + xr + k * 1024);
                const unsigned mrow = (unsigned)(pm0 + wr * 128 + i * 32 + 8 * k), ncol = (unsigned)(pn0 + wc * 64 + j * 32);
                if (p.C) {
                    const unsigned off = vC + (mrow * (unsigned)p.ldc + ncol) * 4u;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), resC, (int)(ok ? off : 0xfffffff0u), 0, 0);
                }
                if (p.C2) {
                    const unsigned off = vC2 + (mrow * (unsigned)p.ldc2 + ncol) * 2u;
                    const bf16x4 hh = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hh), resC2, (int)(ok ? off : 0xfffffff0u), 0, 0);
                }
            }
        }
    };
    // Two loads, always (they count in vmcnt).  Issued through asm so that hipcc does not know of them: a load it tracks would
    // get a compiler-inserted vmcnt wait at its first use in the next tile's seam (vmcnt(0) across the loop's back edge: the fill
    // stream drained once per tile).  Sound only while the compiled code does not touch biasv[] before one of the kernel's own
    // counted waits has retired the load; tests/test_abi_and_host.py checks that on the ISA.
    auto load_bias = [&](int n0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = min(n0 + wc * 64 + j * 32 + (fresh_lane() & 31), p.N - 1);
            const float *src = has_bias ? p.bias + n : b8_zero_word;   // no bias: the counted load fetches a word that IS 0.f
            float b;
            asm volatile("global_load_dword %0, %1, off" : "=v"(b) : "v"(src) : "memory");
            biasv[j] = b;
        }
    };

    constexpr int S = 8 * OUTS;                                 // stores per phase of a seam k-tile
    // vmcnt counts [kind][wait]: kind 0 normal, 1..3 the first three k-tiles of a tile; waits of phase 0, 1, 3
    constexpr int B8_W[4][3] = {{8, 6, 6},
                                {8, b8_cap(6 + S), b8_cap(6 + 3 * S)},
                                {b8_cap(8 + 4 * S + 2), b8_cap(6 + 3 * S + 2), b8_cap(6 + S + 2)},
                                {b8_cap(8 + S + 2), 6, 6}};
#define B8_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

#ifdef B8_STAMP
    unsigned st[4][5] = {};
#endif
    int par = 0;                                                // buffer of the k-tile being multiplied
    int pm0 = 0, pn0 = 0;                                       // the tile whose result is still in the accumulators
    bool live = false;
    auto phase = [&](auto PHc, auto KINDc, int kt) {
        constexpr int PH = decltype(PHc)::value, KIND = decltype(KINDc)::value;
        B8_T(t0);
        if constexpr (PH == 0) B8_WAIT(B8_W[KIND][0]);
        if constexpr (PH == 1) B8_WAIT(B8_W[KIND][1]);
        if constexpr (PH == 3) B8_WAIT(B8_W[KIND][2]);
        B8_T(t1);
        constexpr int ib = PH >= 2 ? 2 : 0, j = (PH == 1 || PH == 2) ? 1 : 0;
        if constexpr (KIND == 1) {
            store_blocks(std::integral_constant<int, ib>{}, std::integral_constant<int, j>{}, pm0, pn0, live);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PH == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bLo[g] = *(const u32x4 *)(smem + adrB[g]);
        }
        if constexpr (PH == 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bHi[g] = *(const u32x4 *)(smem + adrB[g] + 32 * BG_ROWB);
        }
        if constexpr (PH == 0 || PH == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) aF[i][g] = *(const u32x4 *)(smem + adrA[g] + (PH + i) * 32 * BG_ROWB);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifndef B8_DIAG_NOSTAGE
        if constexpr (PH == 1) stage(I2{}, par ^ 1, kt + 1);
        if constexpr (PH == 2) stage(I0{}, par, kt + 2);
        if constexpr (PH == 3) { stage(I3{}, par, kt + 2); stage(I1{}, par, kt + 2); }
        __builtin_amdgcn_sched_barrier(0);
#endif
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        B8_T(t2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        B8_T(t3);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#ifdef B8_DIAG_NOMFMA
                asm volatile("" : "+v"(acc[ib + i][j]) : "v"(aF[i][g]), "v"(j ? bHi[g] : bLo[g]));
#else
                if (KIND == 1 && g == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[ib + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aF[i][g]), __builtin_bit_cast(bf16x8, j ? bHi[g] : bLo[g]), zero, 0, 0, 0);
                } else {
                    big_mma<false>(acc[ib + i][j], aF[i][g], j ? bHi[g] : bLo[g]);
                }
#endif
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        B8_T(t4);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef B8_STAMP
        B8_T(t5);
        st[PH][0] += t1 - t0; st[PH][1] += t2 - t1; st[PH][2] += t3 - t2; st[PH][3] += t4 - t3; st[PH][4] += t5 - t4;
#endif
    };
#ifdef B8_STAMP
    unsigned kind_cycles[4] = {};
#endif
    auto ktile = [&](auto KINDc, int kt) {
        B8_T(k0);
        phase(I0{}, KINDc, kt); phase(I1{}, KINDc, kt); phase(I2{}, KINDc, kt); phase(I3{}, KINDc, kt);
#pragma unroll
        for (int g = 0; g < 4; ++g) { adrA[g] ^= B8_BUF; adrB[g] ^= B8_BUF; }
        par ^= 1;
#ifdef B8_STAMP
        B8_T(k1);
        kind_cycles[decltype(KINDc)::value] += k1 - k0;
#endif
    };

    // prologue: k-tile 0 whole and A-lo, B-lo, B-hi of k-tile 1, in the loop's issue order (A-lo, B-lo, B-hi, A-hi per k-tile);
    // the launcher guarantees nk >= 4
    int m0, n0;
    tile_at(0, m0, n0);
    fill_offsets(I0{}, m0, n0); fill_offsets(I1{}, m0, n0); fill_offsets(I2{}, m0, n0); fill_offsets(I3{}, m0, n0);
    stage(I0{}, 0, 0); stage(I3{}, 0, 0); stage(I1{}, 0, 0); stage(I2{}, 0, 0);
    stage(I0{}, 1, 1); stage(I3{}, 1, 1); stage(I1{}, 1, 1);
    B8_WAIT(10);                                                // A-lo, B-lo of k-tile 0 (B-hi, A-hi: the loop's own waits)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                  // the second group runs one barrier behind from here on
#ifdef B8_STAMP
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int i = 0; i < ntl; ++i) {
        // the tile after this one (the fill stream crosses into it two k-tiles before the seam); past the end: this tile again
        tile_at(i + 1 < ntl ? i + 1 : i, nm0, nn0);
        ktile(I1{}, 0);                                         // stores tile i - 1 ...
        load_bias(n0);                                          // ... and fetches this tile's bias behind the last store
        pm0 = m0; pn0 = n0; live = true;
        ktile(I2{}, 1);
        ktile(I3{}, 2);
        for (int kt = 3; kt < nk; ++kt) ktile(I0{}, kt);
        m0 = nm0; n0 = nn0;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
#ifdef B8_STAMP
    if (blockIdx.x == 0 && wave == 0 && lane == 0) {
        g_b8_clock[0] = (unsigned)(__builtin_amdgcn_s_memtime() - c0); g_b8_clock[1] = (unsigned)(__builtin_amdgcn_s_memrealtime() - r0);
        g_b8_clock[2] = (unsigned)(ntl * nk);
    }
    if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0) {
        for (int a = 0; a < 4; ++a) g_b8_kind[wr][a] = kind_cycles[a];
    }
    if ((blockIdx.x == 0 || blockIdx.x == 100) && (wave & 3) == 0 && lane == 0)
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 5; ++b) g_b8_stamps[blockIdx.x ? 1 : 0][wr][a][b] = st[a][b];
#endif
    // the last tile's result (and the fills issued past the end of the stream)
    B8_WAIT(0);
#ifndef B8_DIAG_NOEPI
    store_blocks(I0{}, I0{}, pm0, pn0, true); store_blocks(I0{}, I1{}, pm0, pn0, true);
    store_blocks(I2{}, I1{}, pm0, pn0, true); store_blocks(I2{}, I0{}, pm0, pn0, true);
#endif
    B8_WAIT(0);
}

}  // namespace

// Does this product go to the 256 x 256 kernel?  K in whole k-tiles, and enough tiles to fill the chip about twice
// (smaller outputs keep the 128 x 128 kernel: more, shorter workgroups).
bool gemm_nt_big_applies(int prec, const GemmNT &g)
{
    const bool f32 = prec != P_BF16;     // the LDS-DMA kernel is bf16 only (P_F32 / P_X3 operands are fp32 in memory)
    // split-bf16: the fp32 instance of this kernel multiplies with EXACT fp32 MFMAs (1/16 of the bf16 rate), the 128-column
    // kernel with three bf16 MFMAs per product (3/16): until round 5 the mode's large products came here and ran at a third of the
    // speed they have there (reading B at tolerance: gemm_wide 7.89 -> 5.03 ms per six fractions, 3.07 -> 3.41 M frames/s)
    if (prec == P_X3) return false;
    const bool off = opt().no_big_gemm;
    const int KB = BG_ROWB / (f32 ? 4 : 2);
    if (off || g.K % KB != 0 || g.K < 4 * KB) return false;
    const long tiles = (long)((g.M + BG_BM - 1) / BG_BM) * ((g.N + BG_BN - 1) / BG_BN);
    return tiles >= 384;
}

#ifdef B8_STAMP
void b8_read_stamps(unsigned *h, unsigned *clk)
{
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_b8_stamps), sizeof(g_b8_stamps));
    (void)hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_b8_clock), sizeof(g_b8_clock));
    (void)hipMemcpyFromSymbol(clk + 3, HIP_SYMBOL(g_b8_kind), sizeof(g_b8_kind));
}
#endif

void launch_gemm_nt_big(hipStream_t s, int prec, const GemmNT &g, hipEvent_t done)
{
    const bool f32 = prec != P_BF16;
    const int tiles_m = (g.M + BG_BM - 1) / BG_BM, tiles_n = (g.N + BG_BN - 1) / BG_BN, nwg = tiles_m * tiles_n;
    static DeviceOnce attr_once;
    static int cus = 0;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void *)gemm_nt_big_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, BG_LDS);
        (void)hipFuncSetAttribute((const void *)gemm_nt_big_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, BG_LDS);
        (void)hipFuncSetAttribute((const void *)gemm_nt_big8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, B8_LDS);
        (void)hipFuncSetAttribute((const void *)gemm_nt_big8_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, B8_LDS);
        int dev = 0; (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    }
    const bool no8 = opt().no_big8;
    // the persistent kernel: 32-bit byte offsets into every operand, whole 16-byte stores, and at least a dozen k-tiles per tile
    // (a seam costs about five k-tiles of time; option big8_min_k lets the tests run it on short K)
    // ... or eight, when the launch is many tiles per CU long (>= 1000 tiles: the LVCSR layer and output products at K = 512 --
    // tools/probe/gemm_bench, us non-persistent / persistent: 51 200 x 2048 203.4 / 193.8, 35 200 x 2048 138.8 / 134.9,
    // 51 200 x 8000 710.8 / 652.7; reading B's 15 600 x 2048, 488 tiles: 51.5 / 54.7 and stays)
    const int min_k = opt().big8_min_k > 0 ? (int)opt().big8_min_k : (nwg >= 1000 ? 8 * 64 : 12 * 64);
    const bool fits = (unsigned long long)g.M * g.lda * 2 < 0xfffffff0ull && (unsigned long long)g.N * g.ldb * 2 < 0xfffffff0ull &&
                      (!g.C || ((unsigned long long)g.M * g.ldc * 4 < 0xfffffff0ull && g.ldc % 4 == 0 && (uintptr_t)g.C % 16 == 0)) &&
                      (!g.C2 || ((unsigned long long)g.M * g.ldc2 * 2 < 0xfffffff0ull && g.ldc2 % 4 == 0 && (uintptr_t)g.C2 % 8 == 0)) &&
                      g.N % 4 == 0 && (g.C || g.C2) &&
                      g.K >= min_k;
    if (f32) hipExtLaunchKernelGGL(gemm_nt_big_kernel<true>, dim3(nwg), dim3(512), BG_LDS, s, nullptr, done, 0, g, tiles_n, nwg);
    else if (!no8 && fits) {
        const int grid = std::min(nwg, std::max(8, cus / 8 * 8)) / 8 * 8;
        if (g.C && g.C2) hipExtLaunchKernelGGL(gemm_nt_big8_kernel<2>, dim3(grid), dim3(512), B8_LDS, s, nullptr, done, 0, g, tiles_n, nwg);
        else             hipExtLaunchKernelGGL(gemm_nt_big8_kernel<1>, dim3(grid), dim3(512), B8_LDS, s, nullptr, done, 0, g, tiles_n, nwg);
    }
    else     hipExtLaunchKernelGGL(gemm_nt_big_kernel<false>, dim3(nwg), dim3(512), BG_LDS, s, nullptr, done, 0, g, tiles_n, nwg);
}

}  // namespace cn

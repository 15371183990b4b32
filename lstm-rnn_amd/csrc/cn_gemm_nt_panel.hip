// gemm_nt for the long-K, narrow-N products of SHORT fractions: C[m][n] = sum_k A[m][k] B[n][k] (+ bias), identity activation,
// with at most one 64-row panel of (real) rows per CU -- the error to the preceding layer (K8, LstmLayer.cu:990-1009, through
// Matrix.cu:218-239) of the headline and reading-B steps: M ~ 15 000 frames, N = 256 / 512, K = 1024 / 2048.  Same contract as
// gemm_nt_kernel, results bit-equal to it (same k order, same MFMA).
//
// Why another kernel (round 6): on the headline's N-wide products gemm_nt_kernel sits at 0.25 of its roof (bench.py roofline_gemm)
// and neither bytes nor MFMAs bound it: a workgroup's life is load -> LDS -> MFMA -> LDS -> store once per tile with ONE k-tile in
// flight, and the chip takes in 9 TB/s from its L2s while tools/probe/fill_probe.hip measures 25 TB/s for LDS-DMA fills with two
// 40 KB stages in flight per CU.  Here ONE workgroup per CU owns a PANEL of 64 rows of C for the whole launch and walks the
// N / 256 column tiles x K / 64 k-tiles as one continuous pipeline:
//   * fills global -> LDS directly (buffer_load ... lds, 1 KB = 8 tile rows of 128 bytes per wave instruction, XOR on the source
//     chunk: slot s of row r holds chunk s ^ ((r >> 1) & 7), so the 16 lanes of a service group of the fragment ds_read_b128 hit
//     16 different bank slots), rings of three k-tiles (A 8 KB, B 32 KB each), two in flight ACROSS tile boundaries;
//   * EIGHT waves: four multiply (one per SIMD, 64 columns of the tile each), four only fill.  The texture path takes ~24 cycles
//     per 1 KB piece and CU (fill_probe), 960 per k-tile against 512 of MFMA: a wave that issues fills stalls in that queue, and
//     with the fills in the multiplying waves (first version: 4 waves, fills in front of the k-tile body) the MFMA pipe idled
//     through every stall -- 2 500 cycles per k-tile, no faster than gemm_nt_kernel.  A loader wave's vmcnt holds fills only
//     (the stores belong to the multiplying waves and are never waited for): its wait is the same counted one every k-tile;
//   * the MFMA operands are SWAPPED (W fragment as A, activation fragment as B): a lane's four consecutive accumulator registers
//     are four consecutive COLUMNS of one row of C -- 16-byte LDS writes into a per-wave staging block, read back as rows and
//     stored 4 rows x 256 contiguous bytes per instruction, no barrier (each wave turns its own 64 x 64 block).  (Stored as they
//     stand -- 32 pieces of 32 bytes per instruction -- the stores of the headline's input projection alone took 23-28 us.)
//   * the fraction's ROW MAP (GemmNT::rowmap): only the rows of real frames are multiplied -- the headline's fraction is
//     17 984 rows = 281 panels on 256 CUs, a second round for 25 panels that doubled every launch; its 14 800 real rows are 232 --
//     and the loader waves write 0 + bias into their share of the dummy rows at the end;
//   * the multiplying waves touch both operands ahead of the fills (L2 warming, below).
// The k-tile body (16 fragment reads, 16 MFMAs) is one asm statement with fixed fragment registers, like gemm_nt_mid_kernel.
//
// Measured (MI355X; NOTEBOOK A.7 has the versions): warm, back to back (tools/probe/gemm_bench): M = 15 600, N = 256, K = 1024
// 20.7 -> 14.1 us, N = 512, K = 2048 49 -> 41 us, N = 1024, K = 256 22.6 -> 22.3 (61 MB of fp32 result: store-bound either way),
// N = 1024, K = 64 14.7 -> 17.1 (slower).  Inside the headline step (cold operands; tools/gemm_in_step.sh): the two K8 products
// 25.5 -> 20-21 us each, the input projections 27 -> 28-33, the softmax products 10.3 -> 12-13: the dispatch takes K >= 512 on at
// most two column tiles (options nt_panel_min_ktiles / nt_panel_max_ntiles).  Reading B (N = 512, K = 2048): gemm_wide -11 %.
#include "cn_internal.h"
#include <algorithm>
#include <cstdint>

namespace cn {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

namespace {

constexpr int NP_BM = 64, NP_BN = 256, NP_BK = 64;
constexpr int NP_ROWB = 128;                                    // bytes of K per tile row and k-tile
constexpr int NP_A = NP_BM * NP_ROWB, NP_B = NP_BN * NP_ROWB;   // 8 KB, 32 KB per k-tile
constexpr int NP_ST = 3;                                        // ring slots: two k-tiles in flight
constexpr int NP_RING_A = NP_ST * NP_B;                         // A ring behind the B ring
constexpr int NP_BIAS = NP_RING_A + NP_ST * NP_A;               // bias behind the rings
constexpr int NP_MAX_N = 1792;                                  // seven column tiles
constexpr int NP_TOUCH_A = 8;                                   // k-tiles the A panel is touched ahead of its use
constexpr int NP_ROWS = NP_BIAS + NP_MAX_N * 4;                 // the panel's 64 destination rows (ints; -1: none) + spare, 1 KB
constexpr int NP_STG = NP_ROWS + 1024;                          // output staging: 32 rows x 64 columns of fp32 per multiplying wave
constexpr int NP_STG_WAVE = 32 * 256;
constexpr int NP_LDS = NP_STG + 4 * NP_STG_WAVE;                // 160 KB: one workgroup per CU

__global__ __launch_bounds__(512) void gemm_nt_panel_kernel(GemmNT p, int tiles_n, int nk, int no_touch)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3;                                 // multiplying wave: its 64 columns of a tile; loader: its quarter of the pieces
    const int fr = lane & 31, fh = lane >> 5;
    const int m0 = blockIdx.x * NP_BM;
    // row map (GemmNT::rowmap): panel row m is row rowmap[m] of A and C, m < nreal; the dummy rows get bias / 0 at the end
    const int nreal = p.rowcnt ? __builtin_amdgcn_readfirstlane(p.rowcnt[0]) : p.M, ndummy = p.rowcnt ? __builtin_amdgcn_readfirstlane(p.rowcnt[1]) : 0;
    const int total = m0 < nreal ? tiles_n * nk : 0;            // (a panel behind the last real row has nothing to multiply)

    // bias -> LDS (zeros past N and without a bias), the panel's rows -> LDS, before the first fill is issued
    for (int i = tid; i < tiles_n * NP_BN; i += 512) ((float *)(smem + NP_BIAS))[i] = (p.bias && i < p.N) ? p.bias[i] : 0.f;
    if (tid < NP_BM) ((int *)(smem + NP_ROWS))[tid] = m0 + tid < nreal ? (p.rowmap ? p.rowmap[m0 + tid] : m0 + tid) : -1;
    __syncthreads();

    auto resource = [](const void *base, long bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, (int)(unsigned)bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t resA = resource(p.A, (long)p.M * p.lda * 2), resB = resource(p.B, (long)p.N * p.ldb * 2);
    const __amdgpu_buffer_rsrc_t resC = resource(p.C, p.C ? (long)p.M * p.ldc * 4 : 0), resC2 = resource(p.C2, p.C2 ? (long)p.M * p.ldc2 * 2 : 0);

    // fill: piece q of an operand covers its tile rows [8 q, 8 q + 8); lane l brings the chunk that belongs in LDS slot l & 7 of
    // row 8 q + (l >> 3).  A: 8 pieces (loader w: 2w, 2w + 1), B: 32 (loader w: 8w .. 8w + 7).  Rows past the edge: clamped (their
    // results are never stored).
    unsigned voffA[2], voffB[8];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 8 * (2 * wave + j) + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int m = max(min(m0 + row, nreal - 1), 0);
        voffA[j] = (unsigned)((long)(p.rowmap ? p.rowmap[m] : m) * p.lda * 2 + chunk * 16);
    }
    auto tile_columns = [&](int n0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = 8 * (8 * wave + j) + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
            voffB[j] = (unsigned)((long)min(n0 + row, p.N - 1) * p.ldb * 2 + chunk * 16);
        }
    };
    auto fill = [&](int f, int kt) {                            // 10 LDS-DMA instructions per loader wave
        char *la = smem + NP_RING_A + (f % NP_ST) * NP_A + (2 * wave) * 1024, *lb = smem + (f % NP_ST) * NP_B + (8 * wave) * 1024;
        const int koff = kt * NP_ROWB;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(resA, (__attribute__((address_space(3))) void *)(la + j * 1024), 16, voffA[j], koff, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(resB, (__attribute__((address_space(3))) void *)(lb + j * 1024), 16, voffB[j], koff, 0, 0);
    };

    f32x16 acc[2][2];                                           // [32 rows of the panel][32 columns of the wave's 64]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses inside a stage (bytes): row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4) with chunk = 2 g + fh for k-group g:
    // = base ^ (g << 5), base = row * 128 + ((fh ^ (s & 1)) << 4) + ((s >> 1) << 5), s = (row >> 1) & 7 (the same for every
    // fragment row of a lane: they are multiples of 32 apart)
    const int s = (fr >> 1) & 7, inrow = ((fh ^ (s & 1)) << 4) + ((s >> 1) << 5);
    const int offX0 = fr * NP_ROWB + inrow, offX1 = (32 + fr) * NP_ROWB + inrow;
    const int offW0 = (wave * 64 + fr) * NP_ROWB + inrow, offW1 = (wave * 64 + 32 + fr) * NP_ROWB + inrow;

    if (loader) {
        int my_dummy = -1;
        if (ndummy > 0) {
            const int d0 = (int)((long)blockIdx.x * ndummy / gridDim.x), d1 = (int)((long)(blockIdx.x + 1) * ndummy / gridDim.x);
            const int d = d0 + wave + 4 * lane;                 // (a share of more than 256 rows would leave rows out: the launcher keeps M / grid below that)
            if (d < d1) my_dummy = p.dummymap[d];
        }
        // the fill counter runs two k-tiles ahead of the multiply counter
        int f_nt = 0, f_kt = 0;
        auto next_fill = [&](int f) {
            if (f_kt == 0) tile_columns(f_nt * NP_BN);
            fill(f, f_kt);
            if (++f_kt == nk) { f_kt = 0; ++f_nt; }
        };
        next_fill(0);
        if (total > 1) next_fill(1);
        for (int i = 0; i < total; ++i) {
            // this wave's part of k-tile i has landed (the younger fill stays in flight); behind the barrier every loader's has,
            // and the multiplying waves are done reading k-tile i - 1, whose slots the next fill overwrites
            if (i + 1 < total) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (i + 2 < total) next_fill(i + 2);
        }
        // the dummy rows of the result: 0 + bias (identity activation; every operand row of a dummy frame is zero), this
        // workgroup's share of them, by the waves that have nothing left to fill.  Loader wave w takes every fourth row of the
        // share; lane i holds the index of the wave's i-th row, loaded in front of the first fill (one latency, not one per row).
        for (int i = 0; i < 64; ++i) {
            const int row = __builtin_amdgcn_readlane(my_dummy, i);
            if (row < 0) break;
            for (int c4 = lane * 4; c4 < p.N; c4 += 256) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) v = *(const f32x4 *)(p.bias + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 0.f + v[e];                  // (what the epilogue computes for a zero sum)
                if (p.C) *(f32x4 *)(p.C + (long)row * p.ldc + c4) = v;
                if (p.C2) *(bf16x4 *)((__bf16 *)p.C2 + (long)row * p.ldc2 + c4) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            }
        }
        return;
    }

    // L2 warming.  Inside a training step both operands are cold (1.4 GB pass between two uses of anything): all panels walk B in
    // step, so every k-tile of B is a first touch for every CU of an XCD at once, and a panel's own A stream has two k-tiles
    // (16 KB) in flight against an HBM latency (tools/probe/gemm_bench GEMM_BENCH_COLD=1: the headline's error product 14 us
    // warm, 25 us cold; 16 us cold without the fills).  The multiplying waves never wait on vmcnt, so THEY touch ahead: one
    // dword per 128-byte line, loaded into a spare LDS word by LDS-DMA (no destination register to keep alive, nothing the
    // compiler sees).  Waves 1 .. 3: this workgroup's share of the whole of B, once, at the start (block ids 8 apart share an
    // XCD under round-robin placement -- speed only); wave 0: the panel's k-tiles NP_TOUCH_A ahead, one instruction per k-tile.
    auto words = [](const void *base, long bytes) {            // the same descriptor as resource(), as four dwords for the asm
        const unsigned long long a = (unsigned long long)(uintptr_t)base;
        return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, (unsigned)bytes, 0x00020000u};
    };
    const u32x4 rA = words(p.A, (long)p.M * p.lda * 2), rB = words(p.B, (long)p.N * p.ldb * 2);
    const unsigned touch_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)(smem + NP_ROWS + 256);
    auto touch = [&](const u32x4 &rsrc, unsigned voff, int soff) {
        unsigned keep;
        asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tbuffer_load_dword %[voff], %[rsrc], %[soff] offen lds\n\ts_mov_b32 m0, %[keep]"
                     : [keep] "=&s"(keep) : [lds] "s"(touch_lds), [voff] "v"(voff), [rsrc] "s"(rsrc), [soff] "s"(soff) : "memory");
    };
    unsigned touchA = 0;
    if (!no_touch) {
        if (wave == 0) {
            const int dst = ((const int *)(smem + NP_ROWS))[lane];
            const int last = max(nreal - 1, 0);
            touchA = (unsigned)((long)(dst >= 0 ? dst : (p.rowmap ? p.rowmap[last] : last)) * p.lda * 2);
            for (int f = 0; f < min(nk, NP_TOUCH_A); ++f) touch(rA, touchA, f * NP_ROWB);
        } else if (total > 0) {
            // line L = row * nk + k-tile of B; this workgroup takes L = share, share + shares, ...
            const int shares = max(1, min(32, (int)gridDim.x / 8)), share = (blockIdx.x >> 3) % shares, lines = p.N * nk;
            for (int L = share + shares * (tid - 64); L < lines; L += shares * 192)
                touch(rB, (unsigned)((long)(L / nk) * p.ldb * 2 + (L % nk) * NP_ROWB), 0);
        }
    }

    int nt = 0, kt = 0;
    for (int i = 0; i < total; ++i) {
        asm volatile("s_barrier" ::: "memory");                 // k-tile i is in LDS
        if (wave == 0 && i + NP_TOUCH_A < nk && !no_touch) touch(rA, touchA, (i + NP_TOUCH_A) * NP_ROWB);
        const int sta = NP_RING_A + (i % NP_ST) * NP_A, stb = (i % NP_ST) * NP_B;
        const int x0 = sta + offX0, x1 = sta + offX1, w0 = stb + offW0, w1 = stb + offW1;
        // fragments (fixed registers, declared clobbered), k-group g: x0 128 + 16 g, x1 132 + 16 g, w0 136 + 16 g, w1 140 + 16 g;
        // LDS returns in order; at most 12 reads outstanding (lgkmcnt counts to 15)
        asm volatile(
            "ds_read_b128 v[128:131], %[x0]\n\t"     "ds_read_b128 v[132:135], %[x1]\n\t"     "ds_read_b128 v[136:139], %[w0]\n\t"     "ds_read_b128 v[140:143], %[w1]\n\t"
            "ds_read_b128 v[144:147], %[x0a]\n\t"    "ds_read_b128 v[148:151], %[x1a]\n\t"    "ds_read_b128 v[152:155], %[w0a]\n\t"    "ds_read_b128 v[156:159], %[w1a]\n\t"
            "ds_read_b128 v[160:163], %[x0b]\n\t"    "ds_read_b128 v[164:167], %[x1b]\n\t"    "ds_read_b128 v[168:171], %[w0b]\n\t"    "ds_read_b128 v[172:175], %[w1b]\n\t"
            "s_waitcnt lgkmcnt(8)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c00], v[136:139], v[128:131], %[c00]\n\t"
            "ds_read_b128 v[176:179], %[x0c]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c01], v[140:143], v[128:131], %[c01]\n\t"
            "ds_read_b128 v[180:183], %[x1c]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c10], v[136:139], v[132:135], %[c10]\n\t"
            "ds_read_b128 v[184:187], %[w0c]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c11], v[140:143], v[132:135], %[c11]\n\t"
            "ds_read_b128 v[188:191], %[w1c]\n\t"
            "s_waitcnt lgkmcnt(8)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c00], v[152:155], v[144:147], %[c00]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c01], v[156:159], v[144:147], %[c01]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c10], v[152:155], v[148:151], %[c10]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c11], v[156:159], v[148:151], %[c11]\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c00], v[168:171], v[160:163], %[c00]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c01], v[172:175], v[160:163], %[c01]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c10], v[168:171], v[164:167], %[c10]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c11], v[172:175], v[164:167], %[c11]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c00], v[184:187], v[176:179], %[c00]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c01], v[188:191], v[176:179], %[c01]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c10], v[184:187], v[180:183], %[c10]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c11], v[188:191], v[180:183], %[c11]\n\t"
            : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1])
            : [x0] "v"(x0), [x1] "v"(x1), [w0] "v"(w0), [w1] "v"(w1), [x0a] "v"(x0 ^ 32), [x1a] "v"(x1 ^ 32), [w0a] "v"(w0 ^ 32), [w1a] "v"(w1 ^ 32),
              [x0b] "v"(x0 ^ 64), [x1b] "v"(x1 ^ 64), [w0b] "v"(w0 ^ 64), [w1b] "v"(w1 ^ 64), [x0c] "v"(x0 ^ 96), [x1c] "v"(x1 ^ 96), [w0c] "v"(w0 ^ 96), [w1c] "v"(w1 ^ 96)
            : "memory", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143",
              "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159",
              "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175",
              "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191");

        if (++kt == nk) {
            // the tile is complete.  A lane holds four consecutive columns of 32 rows: stored as they stand, a wave instruction
            // writes 32 pieces of 32 bytes (measured: the stores of the headline's input projection ALONE took 23-28 us, 2.7 TB/s).
            // Each multiplying wave turns its 64 x 64 block through its own 8 KB of LDS instead, 32 rows at a time (chunks of 16
            // bytes XORed with the row: conflict-free both ways; no barrier -- a wave reads what it wrote), and stores 4 rows x
            // 256 contiguous bytes per instruction, the bias added on the way out.
            // (the accumulators were last written inside an asm statement: the hazard recognizer has not seen those MFMAs)
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            char *stg = smem + NP_STG + wave * NP_STG_WAVE;
            const int c16 = lane & 15, r4 = lane >> 4;
            const int ncol = nt * NP_BN + wave * 64 + c16 * 4;
            const f32x4 bv = *(const f32x4 *)(smem + NP_BIAS + ncol * 4);
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i2][j][4 * g + e];
                        *(f32x4 *)(stg + fr * 256 + (((j * 8 + g * 2 + fh) ^ (fr & 15)) << 4)) = v;
                    }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int row = 4 * t + r4;
                    f32x4 v = *(const f32x4 *)(stg + row * 256 + ((c16 ^ (row & 15)) << 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bv[e];
                    const int dst = ((const int *)(smem + NP_ROWS))[i2 * 32 + row];
                    const bool in = ncol < p.N && dst >= 0;
                    if (p.C) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), resC, in ? (unsigned)((long)dst * p.ldc * 4) + (unsigned)ncol * 4u : 0x80000000u, 0, 0);
                    if (p.C2) {
                        const bf16x4 hh = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hh), resC2, in ? (unsigned)((long)dst * p.ldc2 * 2) + (unsigned)ncol * 2u : 0x80000000u, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i2][j][r] = 0.f;
            kt = 0; ++nt;
        }
    }
    // (a touch lands in LDS: none may be in flight when the workgroup's LDS goes back to the CU -- with a dozen k-tiles behind the
    // last one they have landed long ago, with one or two they may not have)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

// bf16 products with the identity activation whose row panels of 64 are at most a round and a half of the chip's CUs (beyond that
// the tiled kernels' rounds even out and their larger tiles re-read less), K in whole k-tiles of 64, N <= 2048, operands and
// outputs addressable with 31-bit byte offsets, 16-byte rows.
bool gemm_nt_panel_applies(int prec, const GemmNT &g, int cus)
{
    if (opt().no_nt_panel || prec != P_BF16 || g.act != ACT_IDENTITY) return false;
    if (g.K % NP_BK != 0 || g.K < NP_BK || g.N > NP_MAX_N || g.N % 4 != 0 || !(g.C || g.C2)) return false;
    // long K, narrow N (the error to the preceding layer): where the fill pipeline pays.  Measured inside the headline step (cold
    // operands; tools/gemm_in_step.sh): K = 1024, N = 256: 25.5 -> 21-23 us; N = 1024, K = 256 (61 MB of result, store-bound either
    // way): 27 -> 28-33 us; N = 192 / 256, K = 256 / 192: 10.3 -> 12-13 us.
    if (g.K / NP_BK < opt().nt_panel_min_ktiles || (g.N + NP_BN - 1) / NP_BN > opt().nt_panel_max_ntiles) return false;
    if ((g.C && (g.ldc % 4 || (uintptr_t)g.C % 16)) || (g.C2 && (g.ldc2 % 4 || (uintptr_t)g.C2 % 8)) || (uintptr_t)g.A % 16 || (uintptr_t)g.B % 16 || g.lda % 8 || g.ldb % 8) return false;
    const unsigned long long lim = 0x7fff0000ull, rows = (unsigned long long)g.M + NP_BM;
    if (rows * g.lda * 2 >= lim || (unsigned long long)g.N * g.ldb * 2 >= lim || (g.C && rows * g.ldc * 4 >= lim) || (g.C2 && rows * g.ldc2 * 2 >= lim)) return false;
    // one round of workgroups: the (estimated) real rows in panels of 64 must not outnumber the CUs -- a second round of a
    // few panels doubles the launch (seen on the headline: 281 panels on 256 CUs, every product slower than the tiled kernel)
    const long rows_est = g.rowmap && g.m_est > 0 ? std::min(g.m_est, g.M) : g.M, panels = (rows_est + NP_BM - 1) / NP_BM;
    return panels <= (opt().nt_panel_max_panels > 0 ? opt().nt_panel_max_panels : cus);
}

void launch_gemm_nt_panel(hipStream_t s, const GemmNT &g, hipEvent_t done)
{
    const int panels = (g.M + NP_BM - 1) / NP_BM, tiles_n = (g.N + NP_BN - 1) / NP_BN;
    static DeviceOnce attr_once;
    if (attr_once.first()) (void)hipFuncSetAttribute((const void *)gemm_nt_panel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NP_LDS);
    hipExtLaunchKernelGGL(gemm_nt_panel_kernel, dim3(panels), dim3(512), NP_LDS, s, nullptr, done, 0, g, tiles_n, g.K / NP_BK, (int)opt().nt_panel_no_touch);
}

}  // namespace cn

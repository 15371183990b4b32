// MFMA GEMM kernels for the N-wide gate products and the weight-gradient products.
//
// Replaces helpers::Matrix<TDevice>::assignProduct/addProduct (helpers/Matrix.cu:218-349, Cpu
// functors :41-183) and helpers::cublas::multiplyMatrices (helpers/cublas.cu:52-84) at the call
// sites LstmLayer.cu:774-785 (K1), :996-1006 (K8), the GEMM parts of ComputeWeightUpdateFn
// :289-512 (K9), and FeedForwardLayer.cu:148-152,190-197,202-206 (K10,K13,K14).
//
// gfx950 only.  Two kernels:
//   gemm_nt : C[m][n]  = sum_k A[m][k] B[n][k]   both operands K-contiguous (activations x packed
//             weights).  128x128 tile, 4 waves of 64x64 (2x2 v_mfma 32x32), 128 bytes of K per
//             LDS row (+16 pad -> conflict-free ds_read_b128), next k-tile staged in registers.
//   gemm_tn : C[m][n] += sum_k A[k][m] B[k][n]   the reduction runs over frames (K = T*PS), so the
//             operands arrive K-strided; tiles are kept K-major in LDS and the bf16 fragments are
//             read with ds_read_b64_tr_b16 (hardware transpose), fp32 ones with ds_read_b32.
//             Split-K over frames, fp32 atomics into the pre-zeroed gradient.
// Operand type: bf16 (v_mfma_f32_32x32x16_bf16) or fp32 (v_mfma_f32_32x32x2_f32, exact fp32).
#include "cn_internal.h"
#include "cn_lstm_device.h"      // vector types, split_bf16

#include <cstdlib>

namespace cn {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ float act_apply(int act, float x)
{
    // activation_functions/Logistic.cuh:33-44, Tanh.cuh:33-36 (tanh(x) = 2*logistic(2x) - 1)
    if (act == ACT_IDENTITY) return x;
    float z = (act == ACT_TANH) ? 2.0f * x : x;
    float s;
    if (z < 88.722839f) s = (z > -88.722839f) ? 1.0f / (1.0f + __expf(-z)) : 0.0f;
    else s = 1.0f;
    return (act == ACT_TANH) ? 2.0f * s - 1.0f : s;
}

// one K-group (32 bytes of K per row: 16 bf16 or 8 fp32) of a 32x32 tile product
template <bool F32>
__device__ __forceinline__ void mma32(f32x16 &acc, const u32x4 &a, const u32x4 &b)
{
    if constexpr (F32) {
        // K order inside the group is permuted identically for A and B (lane half h holds
        // k = 4h..4h+3), which leaves the dot product unchanged.
        // (bit_cast the whole vector: a bit_cast of a single ext_vector element picks element 0)
        const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[i], acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                      __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
}

// split-bf16 product of one 16-element K-group of a 32x32 tile (P_X3): three bf16 MFMAs, small terms first
__device__ __forceinline__ void mma32_x3(f32x16 &acc, const u32x4 &ah, const u32x4 &al, const u32x4 &bh, const u32x4 &bl)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
}
// four fp32 -> 4 bf16 hi and 4 bf16 lo (8 bytes each)
__device__ __forceinline__ void split4(const u32x4 &x, u32x2 &hi, u32x2 &lo)
{
    const f32x4 f = __builtin_bit_cast(f32x4, x);
    bf16x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) { __bf16 a, b; split_bf16(f[i], a, b); h[i] = a; l[i] = b; }
    hi = __builtin_bit_cast(u32x2, h); lo = __builtin_bit_cast(u32x2, l);
}

// ---------------------------------------------------------------------------------------------
// gemm_nt
// ---------------------------------------------------------------------------------------------
constexpr int NT_BN = 128, NT_ROWB = 128, NT_PITCH = 144;
// One LDS buffer per operand (the next k-tile waits in registers) and the epilogue staged in two halves:
// 36.9 KB per workgroup at 128 rows, three workgroups per CU (VGPR-bound) instead of two with two buffers + a whole-tile
// epilogue (73.7 KB).  Measured (tools/probe/gemm_bench): the small-K products of the headline step are unchanged to
// -3 %, the MFMA-bound ones gain 7-9 % (M = 25 600: N = 8000, K = 1024 550 -> 589 TFLOP/s; N = 1024, K = 8000 607 -> 661).
constexpr int NT_NBUF = 1, NT_EPI_HALVES = 2;
// BM = 128 or 64 rows of C per workgroup (BN = 128 columns, 4 waves as 2 x 2).  The 64-row tile is for products whose 128-row
// grid leaves the chip short of workgroups (N = 256: 244 tiles on 256 CUs, one 4-wave workgroup per CU and nothing to overlap
// its load -> LDS -> MFMA -> store chain with).
template <int BM> struct NtGeom {
    static constexpr int A_BYTES = BM * NT_PITCH, B_BYTES = NT_BN * NT_PITCH;
    static constexpr int EPI_BYTES = BM / NT_EPI_HALVES * (NT_BN * 4 + 16);
    static constexpr int OPS = NT_NBUF * (A_BYTES + B_BYTES);
    static constexpr int LDS_BYTES = OPS > EPI_BYTES ? OPS : EPI_BYTES;
    static constexpr int TI = BM / 64;                      // 32-row MFMA tiles per wave (rows); 2 across the columns
    static constexpr int NLD_A = BM * 8 / 256;              // 16-byte chunks per thread and k-tile
};

template <int PREC, int BM>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNT p, int tiles_n, int nwg, int ndw)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = NtGeom<BM>;
    constexpr bool F32 = PREC == P_F32, X3 = PREC == P_X3;
    constexpr int ELT = PREC == P_BF16 ? 2 : 4;          // operand element in MEMORY (P_X3: fp32, split when it enters the LDS)
    constexpr int KB = NT_ROWB / ELT;          // k elements per tile row
    constexpr int CH = 16 / ELT;               // k elements per 16-byte chunk
    constexpr int TI = G::TI, RW = BM / 2;     // rows of C a wave row owns

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // Row map of the fraction (GemmNT::rowmap): tile row m is row rowmap[m] of A and C for m < nreal; the rows of the dummy
    // frames are not multiplied at all: they get act(0 + bias[n]) -- what the product gives for an all-zero operand row -- from
    // the first `ndw` workgroups of the grid, which do nothing else (they start first and run beside the first round of tiles;
    // in every tile's workgroup instead, the chain count -> row index -> store in front of its k loop cost the headline's input
    // projection 5 us).  The grid still covers M rows: the host does not know nreal.
    if ((int)blockIdx.x < ndw) {
        __shared__ int drow[256];
        const int ndummy = __builtin_amdgcn_readfirstlane(p.rowcnt[1]);
        for (int base = blockIdx.x; base < ndummy; base += ndw * 256) {
            __syncthreads();
            const int d = base + ndw * tid;
            drow[tid] = d < ndummy ? p.dummymap[d] : -1;       // this workgroup's next 256 dummy rows: one round of loads
            __syncthreads();
            for (int i = 0; i < 256 && drow[i] >= 0; ++i) {
                const long m = drow[i];
                for (int n = tid * 4; n < p.N; n += 1024) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias) v = *(const f32x4 *)(p.bias + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_apply(p.act, 0.f + v[e]);       // (0 + bias: what the epilogue computes for a zero sum)
                    if (p.C) *(f32x4 *)(p.C + m * p.ldc + n) = v;
                    if (p.C2) {
                        if constexpr (PREC != P_BF16) *(f32x4 *)((float *)p.C2 + m * p.ldc2 + n) = v;
                        else *(bf16x4 *)((__bf16 *)p.C2 + m * p.ldc2 + n) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    }
                }
            }
        }
        return;
    }

    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous
    // run of tiles so the N-tiles of one A panel hit the same L2 (bijective remap).
    int bid = blockIdx.x - ndw;
    {
        int q = nwg / 8, r = nwg % 8, x = bid % 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * NT_BN;

    const char *Ab = (const char *)p.A, *Bb = (const char *)p.B;
    const int nk = (p.K + KB - 1) / KB;

    // (count and row indices are loaded side by side -- one latency in front of the k loop; entries behind nreal are stale, valid
    // rows -- and kept in LDS for the epilogue: looked up there row by row, each store waited for its own index)
    __shared__ int crow[128];
    long arow[G::NLD_A];                       // this thread's rows of A (elements), -1: behind the last row
    int nreal = p.M;
    {
        if (tid < BM) crow[tid] = p.rowmap ? p.rowmap[min(m0 + tid, p.M - 1)] : m0 + tid;
        if (p.rowcnt) nreal = __builtin_amdgcn_readfirstlane(p.rowcnt[0]);
        if (m0 >= nreal) return;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < G::NLD_A; ++j) { const int r = (tid + 256 * j) >> 3; arow[j] = m0 + r < nreal ? (long)crow[r] * p.lda : -1; }
    }

    u32x4 ra[G::NLD_A], rb[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = tid + 256 * j, row = c >> 3, kc = c & 7;
            int k = kt * KB + kc * CH;
            u32x4 z = {0u, 0u, 0u, 0u};
            if (j < G::NLD_A) ra[j < G::NLD_A ? j : 0] = z;
            rb[j] = z;
            if (k < p.K) {
                if (j < G::NLD_A && arow[j < G::NLD_A ? j : 0] >= 0) ra[j < G::NLD_A ? j : 0] = *(const u32x4 *)(Ab + (arow[j < G::NLD_A ? j : 0] + k) * ELT);
                if (n0 + row < p.N) rb[j] = *(const u32x4 *)(Bb + ((long)(n0 + row) * p.ldb + k) * ELT);
            }
        }
    };
    auto lwrite = [&](int buf) {
        char *sa = smem + buf * (G::A_BYTES + G::B_BYTES), *sb = sa + G::A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = tid + 256 * j, row = c >> 3, kc = c & 7;
            if constexpr (X3) {
                // a tile row holds 32 k: [32 bf16 hi | 32 bf16 lo] in the same 128 bytes the fp32 row would take
                u32x2 h, l;
                if (j < G::NLD_A) {
                    split4(ra[j < G::NLD_A ? j : 0], h, l);
                    *(u32x2 *)(sa + row * NT_PITCH + kc * 8) = h; *(u32x2 *)(sa + row * NT_PITCH + 64 + kc * 8) = l;
                }
                split4(rb[j], h, l);
                *(u32x2 *)(sb + row * NT_PITCH + kc * 8) = h; *(u32x2 *)(sb + row * NT_PITCH + 64 + kc * 8) = l;
            } else {
                if (j < G::NLD_A) *(u32x4 *)(sa + row * NT_PITCH + kc * 16) = ra[j < G::NLD_A ? j : 0];
                *(u32x4 *)(sb + row * NT_PITCH + kc * 16) = rb[j];
            }
        }
    };

    f32x16 acc[TI][2];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    gload(0);
    lwrite(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
        const char *sa = smem + (kt % NT_NBUF) * (G::A_BYTES + G::B_BYTES), *sb = sa + G::A_BYTES;
        if constexpr (X3) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                u32x4 ah[TI], al[TI], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    ah[i] = *(const u32x4 *)(sa + (wm * RW + i * 32 + fr) * NT_PITCH + g * 32 + fh * 16);
                    al[i] = *(const u32x4 *)(sa + (wm * RW + i * 32 + fr) * NT_PITCH + 64 + g * 32 + fh * 16);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    bh[i] = *(const u32x4 *)(sb + (wn * 64 + i * 32 + fr) * NT_PITCH + g * 32 + fh * 16);
                    bl[i] = *(const u32x4 *)(sb + (wn * 64 + i * 32 + fr) * NT_PITCH + 64 + g * 32 + fh * 16);
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma32_x3(acc[i][j], ah[i], al[i], bh[j], bl[j]);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x4 a[TI], b[2];
#pragma unroll
                for (int i = 0; i < TI; ++i) a[i] = *(const u32x4 *)(sa + (wm * RW + i * 32 + fr) * NT_PITCH + g * 32 + fh * 16);
#pragma unroll
                for (int i = 0; i < 2; ++i) b[i] = *(const u32x4 *)(sb + (wn * 64 + i * 32 + fr) * NT_PITCH + g * 32 + fh * 16);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma32<F32>(acc[i][j], a[i], b[j]);
            }
        }
        if (NT_NBUF == 1) __syncthreads();
        if (kt + 1 < nk) lwrite((kt + 1) % NT_NBUF);
        __syncthreads();
    }

    // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5): a lane owns
    // single dwords of 16 different rows, and stored straight from the registers the tile leaves the CU as 64
    // dword stores per lane (2.3 TB/s of output at best, measured with K = 64).  The tile is transposed through
    // the operand LDS instead (free after the last k step) and written as whole 512-byte rows,
    // 16 B per lane; bias and activation are applied on the way out, where a thread's four columns are fixed.
    constexpr int EP = NT_BN * 4 + 16;                 // staging row pitch (bytes)
    constexpr int EH = NT_EPI_HALVES, ROWS = BM / EH;           // staged rows per pass (= the rows of one wave row)
    static_assert(ROWS * EP <= G::LDS_BYTES, "epilogue staging does not fit the operand buffers");
    const int c4 = tid & 31, n = n0 + c4 * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && n < p.N) bv = *(const f32x4 *)(p.bias + n);
#pragma unroll
    for (int h = 0; h < EH; ++h) {
        if (h) __syncthreads();
        if (wm == h) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        *(float *)(smem + (i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * EP + (wn * 64 + j * 32 + fr) * 4) = acc[i][j][r];
        }
        __syncthreads();
        if (n < p.N) {
#pragma unroll
            for (int k = 0; k < ROWS / 8; ++k) {
                const int row = (tid >> 5) + 8 * k, mt = m0 + h * ROWS + row;
                if (mt >= nreal) break;
                const long m = crow[h * ROWS + row];
                f32x4 v = *(const f32x4 *)(smem + row * EP + c4 * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_apply(p.act, v[e] + bv[e]);
                if (p.C) *(f32x4 *)(p.C + m * p.ldc + n) = v;
                if (p.C2) {
                    if constexpr (ELT == 4) *(f32x4 *)((float *)p.C2 + m * p.ldc2 + n) = v;
                    else {
                        const bf16x4 hh = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        *(bf16x4 *)((__bf16 *)p.C2 + m * p.ldc2 + n) = hh;
                    }
                }
            }
        }
    }
}

template <int PREC, int BM>
static void launch_nt(hipStream_t s, const GemmNT &g, hipEvent_t done)
{
    using G = NtGeom<BM>;
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + NT_BN - 1) / NT_BN, nwg = tiles_m * tiles_n;
    auto kern = gemm_nt_kernel<PREC, BM>;
    static DeviceOnce attr_once;
    if (attr_once.first()) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    // (with a row map: dummy-row workgroups in front, a fraction's dummy rows x N / 4 chunks over them)
    const int ndw = g.rowcnt ? 96 : 0;
    hipExtLaunchKernelGGL(kern, dim3(ndw + nwg), dim3(256), G::LDS_BYTES, s, nullptr, done, 0, g, tiles_n, nwg, ndw);
}

void launch_gemm_nt(hipStream_t s, int prec, const GemmNT &g_in, hipEvent_t done)
{
    if (g_in.M <= 0 || g_in.N <= 0) return;
    GemmNT g = g_in;
    {
        // CUs of the current device, once per device
        static int cus_of[64] = {0};
        int dev = 0; (void)hipGetDevice(&dev);
        int &cus = cus_of[dev & 63];
        if (!cus) { hipDeviceProp_t prop; cus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256; }
        if (gemm_nt_panel_applies(prec, g, cus)) { launch_gemm_nt_panel(s, g, done); return; }
    }
    // The tiled kernels below multiply every row unless asked otherwise: gemm_nt_kernel takes the row map too (option
    // nt_rowmap_tiled; tests/test_gpu_rowmap.py runs it), but 15 % fewer rows buy it nothing inside the headline step (input
    // projections 27 -> 27 us: store-bound, and the dummy rows are written all the same; softmax products 10.3 -> 11-13)
    if (!opt().nt_rowmap_tiled) { g.rowmap = g.dummymap = g.rowcnt = nullptr; g.m_est = 0; }
    if (gemm_nt_mid_applies(prec, g)) { launch_gemm_nt_mid(s, g, done); return; }
    if (gemm_nt_big_applies(prec, g)) { launch_gemm_nt_big(s, prec, g, done); return; }
    // 64-row tiles when the 128-row grid leaves the chip short of workgroups (< 400 tiles: the N = 256 / 192 products of the
    // headline step, 244 tiles on 256 CUs) or the K loop is at most four k-tiles long (the input projections: a workgroup's
    // fixed costs dominate).  tools/probe/gemm_bench, 128 -> 64 rows: error to the preceding layer 22.2 -> 20.7 us, softmax
    // products 10.1 / 9.7 -> 9.0 / 8.8, input projections 16.2 / 23.3 -> 14.7 / 22.7; long-K products with 400+ tiles lose
    // (N = 512, K = 2048: 48.6 -> 57.0 us) and keep 128 rows.  CN_NT_BM64_BELOW=<tiles> overrides the first threshold.
    const long below = opt().nt_bm64_below;
    const long tiles128 = (long)((g.M + 127) / 128) * ((g.N + NT_BN - 1) / NT_BN);
    const int elt = prec == P_BF16 ? 2 : 4;
    const bool small = tiles128 < below || ((long)g.K * elt <= 4 * NT_ROWB && tiles128 < opt().nt_bm64_shortk_tiles);
#define CN_NT_DISPATCH(P) { if (small) launch_nt<P, 64>(s, g, done); else launch_nt<P, 128>(s, g, done); }
    if (prec == P_F32) CN_NT_DISPATCH(P_F32)
    else if (prec == P_X3) CN_NT_DISPATCH(P_X3)
    else CN_NT_DISPATCH(P_BF16)
#undef CN_NT_DISPATCH
}

// ---------------------------------------------------------------------------------------------
// gemm_tn
// ---------------------------------------------------------------------------------------------
// BM x BN output tile, WM x WN waves, each wave (BM/WM) x (BN/WN) in 32x32 MFMA tiles.  Shapes:
//   64 x 64   (2 x 2 waves)  the small split-K products when the output is narrow (M < 256)
//   128 x 128 (2 x 2 waves)  once the output alone fills the chip
//   (256-row tiles with 4 x 2 waves were measured and rejected: see tn_shape)
// rows of K per k-step: 64 for the 64 x 64 tiles (the small split-K products: 4-19 % faster than 32, fewer barriers per
// byte), 32 for the larger tiles (64 measured 3 % slower at 128 x 128)
constexpr int tn_bk(int bm) { return bm == 64 ? 64 : 32; }
constexpr int tn_threads(int bm) { return bm == 256 ? 512 : 256; }
// one LDS buffer per operand, next k-tile in registers (as gemm_nt): more workgroups per CU; the 8000 x 1024 x 25 600
// product gains 32 % (342 -> 450 TFLOP/s), the small split-K products are unchanged
constexpr int TN_NBUF = 1;
template <int PREC, int BM, int BN> struct TnGeom {
    static constexpr int ELT = PREC == P_BF16 ? 2 : 4;     // operand element in memory
    static constexpr int LELT = PREC == P_F32 ? 4 : 2;     // element of an LDS plane (P_X3: two bf16 planes, hi and lo)
    static constexpr int PLANES = PREC == P_X3 ? 2 : 1;
    static constexpr int WM = BM == 256 ? 4 : 2, WN = 2, NT = 64 * WM * WN;
    static constexpr int BK = tn_bk(BM);
    static constexpr int PITCH_A = BM * LELT + 64, PITCH_B = BN * LELT + 64;   // K-major rows; 4 consecutive k rows hit distinct bank quarters
    static constexpr int PLANE_A = BK * PITCH_A, PLANE_B = BK * PITCH_B;
    static constexpr int TILE_A = PLANES * PLANE_A, TILE_B = PLANES * PLANE_B;
    static constexpr int LDS = TN_NBUF * (TILE_A + TILE_B);
    static constexpr int CPR_A = BM * ELT / 16, CPR_B = BN * ELT / 16;         // 16-byte chunks per tile row (in memory)
    static constexpr int NCH_A = BK * CPR_A, NCH_B = BK * CPR_B;                // chunks of a k-tile
    static constexpr int NLD_A = (NCH_A + NT - 1) / NT, NLD_B = (NCH_B + NT - 1) / NT;     // chunks per thread (the last round may be partial)
    static constexpr int TI = BM / WM / 32, TJ = BN / WN / 32;                 // 32x32 MFMA tiles per wave
};

// Up to TN_GROUP independent products in one launch (the three weight-gradient products of an LSTM layer): each
// of them alone is bound by the latency of its short per-workgroup K loops and by its split-K atomics, not by
// bandwidth, so side by side they take about as long as the largest one.
constexpr int TN_GROUP = 3;
struct GemmTNGroup {
    GemmTN p[TN_GROUP];
    int tiles_n[TN_GROUP], ntiles[TN_GROUP], kchunk[TN_GROUP], first_block[TN_GROUP + 1];
};

template <int PREC, int BM, int BN>
__global__ __launch_bounds__(tn_threads(BM)) void gemm_tn_kernel(GemmTNGroup grp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = TnGeom<PREC, BM, BN>;
    constexpr bool F32 = PREC == P_F32, X3 = PREC == P_X3;
    constexpr int ELT = G::ELT, PA = G::PITCH_A, PB = G::PITCH_B, TI = G::TI, TJ = G::TJ, NT = G::NT;
    constexpr int CH = 16 / ELT, RM = BM / G::WM, RN = BN / G::WN;      // rows / columns of the output a wave owns

    int gi = 0;
#pragma unroll
    for (int i = 1; i < TN_GROUP; ++i) if ((int)blockIdx.x >= grp.first_block[i]) gi = i;
    // a COPY of the product's description: through a reference into the kernel-argument struct (the index is dynamic) hipcc re-loaded
    // M / N / lda / ldb with an s_load + wait in front of every global load of the K loop (ISA, round 5)
    const GemmTN p = grp.p[gi];
    const int tiles_n = grp.tiles_n[gi], ntiles = grp.ntiles[gi], kchunk = grp.kchunk[gi];
    const int bid = blockIdx.x - grp.first_block[gi];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / G::WN, wn = wave % G::WN;
    const int tile = bid % ntiles, split = bid / ntiles;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int kbeg = split * kchunk;
    const int kend = min(p.K, kbeg + kchunk);
    if (kbeg >= kend) return;
    const int nk = (kend - kbeg + G::BK - 1) / G::BK;

    const char *Ab = (const char *)p.A, *Bb = (const char *)p.B;
    u32x4 ra[G::NLD_A], rb[G::NLD_B];
    auto gload = [&](int kt) {
#pragma unroll
        for (int j = 0; j < G::NLD_A; ++j) {
            int c = tid + NT * j, kr = c / G::CPR_A, cc = c % G::CPR_A;
            int k = kbeg + kt * G::BK + kr;
            u32x4 z = {0u, 0u, 0u, 0u};
            ra[j] = z;
            if (c < G::NCH_A && k < kend && m0 + cc * CH < p.M) ra[j] = *(const u32x4 *)(Ab + ((long)k * p.lda + m0 + cc * CH) * ELT);
        }
#pragma unroll
        for (int j = 0; j < G::NLD_B; ++j) {
            int c = tid + NT * j, kr = c / G::CPR_B, cc = c % G::CPR_B;
            int k = kbeg + kt * G::BK + kr;
            u32x4 z = {0u, 0u, 0u, 0u};
            rb[j] = z;
            if (c < G::NCH_B && k < kend && n0 + cc * CH < p.N) rb[j] = *(const u32x4 *)(Bb + ((long)k * p.ldb + n0 + cc * CH) * ELT);
        }
    };
    auto lwrite = [&](int buf) {
        char *sa = smem + buf * (G::TILE_A + G::TILE_B), *sb = sa + G::TILE_A;
#pragma unroll
        for (int j = 0; j < G::NLD_A; ++j) {
            int c = tid + NT * j, kr = c / G::CPR_A, cc = c % G::CPR_A;
            if (G::NCH_A % NT != 0 && c >= G::NCH_A) continue;
            if constexpr (X3) {
                u32x2 h, l;
                split4(ra[j], h, l);
                *(u32x2 *)(sa + kr * PA + cc * 8) = h; *(u32x2 *)(sa + G::PLANE_A + kr * PA + cc * 8) = l;
            } else *(u32x4 *)(sa + kr * PA + cc * 16) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < G::NLD_B; ++j) {
            int c = tid + NT * j, kr = c / G::CPR_B, cc = c % G::CPR_B;
            if (G::NCH_B % NT != 0 && c >= G::NCH_B) continue;
            if constexpr (X3) {
                u32x2 h, l;
                split4(rb[j], h, l);
                *(u32x2 *)(sb + kr * PB + cc * 8) = h; *(u32x2 *)(sb + G::PLANE_B + kr * PB + cc * 8) = l;
            } else *(u32x4 *)(sb + kr * PB + cc * 16) = rb[j];
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    gload(0);
    lwrite(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
        const char *sa = smem + (kt % TN_NBUF) * (G::TILE_A + G::TILE_B), *sb = sa + G::TILE_A;
        if constexpr (F32) {
            // v_mfma_f32_32x32x2_f32: lane (r,h) holds A[m=r][k=2s+h] / B[k=2s+h][n=r]
#pragma unroll 4
            for (int s2 = 0; s2 < G::BK / 2; ++s2) {
                float a[TI], b[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) a[i] = *(const float *)(sa + (2 * s2 + fh) * PA + (wm * RM + i * 32 + fr) * 4);
#pragma unroll
                for (int j = 0; j < TJ; ++j) b[j] = *(const float *)(sb + (2 * s2 + fh) * PB + (wn * RN + j * 32 + fr) * 4);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(m) block, lane 4q+p supplies the
            // address of row q, columns 4p..4p+3, lane i receives column i of the 4 rows.
            const int g16 = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
#pragma unroll
            for (int ks = 0; ks < G::BK / 16; ++ks) {
                bf16x8 a[G::PLANES][TI], b[G::PLANES][TJ];
#pragma unroll
                for (int pl = 0; pl < G::PLANES; ++pl)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int k = ks * 16 + 8 * (g16 >> 1) + 4 * jj + q;
#pragma unroll
                        for (int i = 0; i < TI; ++i) {
                            const int ma = wm * RM + i * 32 + 16 * (g16 & 1) + 4 * pp;
                            s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (s16x4 __attribute__((address_space(3))) *)(sa + pl * G::PLANE_A + k * PA + ma * 2));
                            bf16x4 ta = __builtin_bit_cast(bf16x4, va);
#pragma unroll
                            for (int e = 0; e < 4; ++e) a[pl][i][4 * jj + e] = ta[e];
                        }
#pragma unroll
                        for (int j = 0; j < TJ; ++j) {
                            const int nb = wn * RN + j * 32 + 16 * (g16 & 1) + 4 * pp;
                            s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (s16x4 __attribute__((address_space(3))) *)(sb + pl * G::PLANE_B + k * PB + nb * 2));
                            bf16x4 tb = __builtin_bit_cast(bf16x4, vb);
#pragma unroll
                            for (int e = 0; e < 4; ++e) b[pl][j][4 * jj + e] = tb[e];
                        }
                    }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        if constexpr (X3) {          // plane 0 = hi, plane 1 = lo; small terms first
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
                    }
            }
        }
        if (TN_NBUF == 1) __syncthreads();
        if (kt + 1 < nk) lwrite((kt + 1) % TN_NBUF);
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int n = n0 + wn * RN + j * 32 + fr;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * RM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m >= p.M) continue;
                // deterministic mode: this split's partial goes to its own copy of C, folded in split order afterwards (launch_fold)
                if (p.ws) p.ws[(long)split * p.M * p.ldc + (long)m * p.ldc + n] = acc[i][j][r];
                else atomicAdd(&p.C[(long)m * p.ldc + n], acc[i][j][r]);
            }
        }
    }
}

template <int PREC, int BM, int BN>
static void launch_tn(hipStream_t s, const GemmTN *gs, int n, const FoldItem *extra)
{
    using G = TnGeom<PREC, BM, BN>;
    GemmTNGroup grp{};
    FoldItem fold[TN_GROUP + 1]; int nfold = 0;
    int blocks = 0;
    long all_tiles = 0;
    for (int i = 0; i < n; ++i) all_tiles += (long)((gs[i].M + BM - 1) / BM) * ((gs[i].N + BN - 1) / BN);
    for (int i = 0; i < TN_GROUP; ++i) {
        grp.first_block[i] = blocks;
        if (i >= n) { grp.first_block[i] = 0x7fffffff; continue; }
        const GemmTN &g = gs[i];
        const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
        // K splits: enough workgroups (2-3 per CU at 64 rows of K per k-step) to hide the latency of the short per-workgroup K loops, but every
        // split ends in M*N fp32 atomics and the chip adds only ~1.3 TB/s of atomic bytes (MI355X_MICROARCH.md,
        // global float atomics): keep the atomic volume of one launch under ~32 MB and every split >= 4 K-tiles.
        // (a group shares the workgroup budget: its products run side by side, and their atomics add up)
        long cap_atomic = (32L << 20) / ((long)g.M * g.N * 4);
        // 576 four-wave workgroups swept best (256..2048) on the headline step for the 64 x 64 tiles
        const int target_env = (int)opt().tn_blocks;
        const int target = target_env ? target_env : 576;
        int splits = (int)((target + all_tiles - 1) / all_tiles);
        int maxsplit = (g.K + 4 * G::BK - 1) / (4 * G::BK);
        if (splits > maxsplit) splits = maxsplit;
        if (splits > cap_atomic && !g.ws) splits = (int)cap_atomic;
        if (g.ws && splits > g.ws_splits) splits = g.ws_splits;
        if (splits < 1) splits = 1;
        int kchunk = ((g.K + splits - 1) / splits + G::BK - 1) / G::BK * G::BK;
        splits = (g.K + kchunk - 1) / kchunk;
        grp.p[i] = g; grp.tiles_n[i] = tiles_n; grp.ntiles[i] = ntiles; grp.kchunk[i] = kchunk;
        blocks += ntiles * splits;
        if (g.ws && g.ws_used) *g.ws_used = splits;        // (the caller's consumer adds the partials itself)
        else if (g.ws) fold[nfold++] = FoldItem{g.C, g.ws, (long)g.M * g.ldc, splits, g.M, g.N, (int)g.ldc, 0, 0};
    }
    grp.first_block[TN_GROUP] = blocks;
    if (extra) fold[nfold++] = *extra;
    if (blocks == 0) { if (nfold) launch_fold(s, fold, nfold); return; }
    auto kern = gemm_tn_kernel<PREC, BM, BN>;
    constexpr int lds = G::LDS;
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(G::NT), lds, s, grp);
    if (nfold) launch_fold(s, fold, nfold);          // deterministic mode: C = the partials of the splits, added in split order
}

// tile shape of one launch (all products of a group share it)
// Measured and rejected (round 2, tools/probe/gemm_bench): 256 x 128 / 256 x 64 tiles (4 x 2 waves) for the LSTM gradient products.
// They cut the operand bytes a launch requests from ~3.7x to ~1.6x the unique bytes (profiles/r01k_pmc.md: 180 MB for 48 MB), and
// they are SLOWER at every workgroup budget (96..384): layer 2/3 group 48.0 vs 45.9 us, layer 1 group 35.6 vs 29.7 us.  The
// re-reads of the 64 x 64 tiles are served by L2 / the Infinity Cache (the operands of one layer are 48 MB); what bounds these
// products is the latency of their short per-workgroup K loops and the split-K atomics, and fewer, fatter workgroups have less
// of both to overlap.  The template keeps rectangular tiles (BM x BN, WM x WN waves); only 64 x 64 and 128 x 128 are built.
enum TnShape { TN_64, TN_128 };
static TnShape tn_shape(const GemmTN *gs, int n)
{
    // once the output alone fills the chip, 128 x 128 tiles and few splits (weight matrices of the LVCSR output layer)
    for (int i = 0; i < n; ++i)
        if ((long)((gs[i].M + 63) / 64) * ((gs[i].N + 63) / 64) >= 2048) return TN_128;
    return TN_64;
}

static void launch_tn_any(hipStream_t s, int prec, const GemmTN *gs, int n, const FoldItem *extra)
{
    const TnShape sh = tn_shape(gs, n);
#define CN_TN_DISPATCH(P)                                                      \
    switch (sh) {                                                              \
    case TN_64:      launch_tn<P, 64, 64>(s, gs, n, extra); break;             \
    case TN_128:     launch_tn<P, 128, 128>(s, gs, n, extra); break;           \
    }
    if (prec == P_F32) { CN_TN_DISPATCH(P_F32) }
    else if (prec == P_X3) { CN_TN_DISPATCH(P_X3) }
    else { CN_TN_DISPATCH(P_BF16) }
#undef CN_TN_DISPATCH
}

void launch_gemm_tn(hipStream_t s, int prec, const GemmTN &g, int cu_budget, const FoldItem *extra)
{
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) { if (extra) launch_fold(s, extra, 1); return; }
    if (gemm_tn_big_applies(prec, g)) { launch_gemm_tn_big_group(s, &g, 1, cu_budget, extra); return; }
    launch_tn_any(s, prec, &g, 1, extra);
}

void launch_gemm_tn_small_group(hipStream_t s, int prec, const GemmTN *gs, int n, const FoldItem *extra)
{
    launch_tn_any(s, prec, gs, n, extra);
}

void launch_gemm_tn_group(hipStream_t s, int prec, const GemmTN *gs, int n, int cu_budget, const FoldItem *extra)
{
    GemmTN grp[TN_GROUP], big[TN_GROUP]; int ng = 0, nb = 0;
    bool any_big = false;
    for (int i = 0; i < n; ++i) any_big = any_big || (gs[i].M > 0 && gs[i].N > 0 && gs[i].K > 0 && gemm_tn_big_applies(prec, gs[i]));
    {   // ... and the layer-1 group of the 256-wide layers (dW_in 2048 x 64 stays on the small tiles, the two dW_rec 1024 x 256 can go):
        // nothing in it is large enough on its own, but it is the exposed tail of the step, and over many frames the pair does better
        // on the 256 x 256 kernel -- tools/probe/gemm_bench, group alone, us small tiles / pair on the large kernel + dW_in behind it:
        // K = 15 600 67.2 / 84.9, K = 35 200 142.5 / 121.0, K = 51 200 230.7 / 146.6.  From half a million outputs and 28 000 frames on.
        const long gmink = opt().tnbig_group_mink;
        long outs = 0, kmin = 1L << 40;
        for (int i = 0; i < n; ++i)
            if (gs[i].M > 0 && gs[i].N > 0 && gs[i].K > 0 && gemm_tn_big_can(prec, gs[i])) { outs += (long)gs[i].M * gs[i].N; kmin = std::min<long>(kmin, gs[i].K); }
        if (outs >= (1L << 19) && kmin >= gmink) any_big = true;
    }
    for (int i = 0; i < n; ++i) {
        if (gs[i].M <= 0 || gs[i].N <= 0 || gs[i].K <= 0) continue;
        // one grouped launch of 256 x 256 tiles for the products large enough to want them and those that can ride along
        if (any_big && nb < TN_GROUP && gemm_tn_big_can(prec, gs[i])) { big[nb++] = gs[i]; continue; }
        const bool huge = (long)((gs[i].M + 63) / 64) * ((gs[i].N + 63) / 64) >= 2048;
        if (huge || ng == TN_GROUP) launch_gemm_tn(s, prec, gs[i]);      // (not grouped)
        else grp[ng++] = gs[i];
    }
    // (the extra fold rides on the last launch of the call)
    if (nb) launch_gemm_tn_big_group(s, big, nb, cu_budget, ng ? nullptr : extra);
    if (ng) launch_tn_any(s, prec, grp, ng, extra);
    else if (!nb && extra) launch_fold(s, extra, 1);
}

}  // namespace cn

// Persistent recurrent LSTM kernels, "s2" shape: TWO sequences per workgroup, one wave per SIMD.
//
// Same reference functions as cn_lstm.hip (LstmLayer.cu:812-829 / :847-864 forward with ComputeBlockOutputFn :47-138,
// :936-951 / :970-985 backward with ComputeBlockErrorsFn :190-287, the Resort functors :140-188 and the bias / peephole
// parts of ComputeWeightUpdateFn :289-512), same buffers, same arithmetic (row-pair sparse MFMAs of cn_lstm_device.h,
// P_BF16 and P_X3) -- a different cut of the work over the chip.
//
// Why.  The step of the 4-sequences-per-workgroup kernels (cn_lstm.hip) is bound by vector-instruction issue, not by the
// MFMA pipe and not by memory: 8 waves x (43 VALU + 10 transcendental + 8 MFMA issues) = 2 waves per SIMD x ~316 issue
// cycles = ~630 of the ~1080 cycles of a forward step, the rest is the LDS hand-off, and at PS = 50 the grid is 26
// workgroups on a 256-CU chip.  Halving the sequences per workgroup halves the cell updates per CU and doubles the CUs
// in use -- but only if every lane of every wave still owns a (unit, sequence) pair, because instruction issue is per
// wave, not per lane.  So a wave owns 32 units (two 16-column MFMA tiles, "unit groups" 0 and 1) x 2 sequences:
//   lane (c = lane & 15, q = lane >> 4)  ->  unit 32*wave + 16*(q >> 1) + c,  sequence s0 + (q & 1).
// The 16x16 MFMA hands row 4*q' + r of the output tile to lane quarter q', so group 0's sums must come out in tile rows
// 0..7 and group 1's in rows 8..15.  Both groups accumulate into the SAME accumulators: group 0's product reads the
// operand tile through a view whose rows 8..15 are a row of zeros, group 1's through the mirrored view (rows 0..7 zero),
// so each adds exactly nothing to the other's lanes.  The views are per-lane LDS addresses (a lane of the A operand IS a
// tile row): the tile itself holds only the 4 (forward) / 8 (backward) data rows of the two sequences plus one zero row.
// Cost: 16 instead of 8 sparse MFMAs per wave and step on half as many waves -- the MFMA pipe of a SIMD is busy for the
// same 256 cycles per step as before, while its VALU issue drops from two waves' cell updates to one.
//
// Everything else follows cn_lstm.hip: W_rec fragments register resident for the whole pass (128 VGPRs per lane at
// Hp = 128; 256 with the hi/lo fragments of P_X3), y[t] / deltas handed over through a double-buffered LDS tile, operands
// prefetched two steps ahead through staged registers, LDS-only barrier, fw/bw halves written in place.
#include "cn_internal.h"
#include "cn_lstm_device.h"

#include <cstdio>
#include <cstdlib>

namespace cn {

#define KEEP_TUPLE(tuple, after) asm volatile("" :: "v"(tuple), "v"(after))

template <typename T> __device__ __forceinline__ T &at32(const void *base, unsigned elem)
{
    return *(T *)((char *)base + elem * (unsigned)sizeof(T));
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int PREC, int HP>
__global__ __launch_bounds__(HP * 2) void lstm_fwd_s2_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(PREC != P_F32 && HP % 64 == 0, "row-pair products: bf16 operands, whole 64-value K chunks");
    constexpr bool X3 = PREC == P_X3;
    constexpr int MELT = X3 ? 4 : 2;                 // operand element in memory (y, W_rec)
    constexpr int KCS = HP / 64;                     // 64-value K chunks
    constexpr int pitch = lds_pitch(HP);             // a tile row holds half of the K values of a sequence (bf16)
    // bf16: rows 2*seq + parity; split-bf16: "row quads" 4*seq + 2*part + parity with part 0 = hi, 1 = lo halves of y -- the two
    // rows of a quad that the bf16 tile leaves empty.  One MFMA against W_hi then yields [y_hi; y_lo] * W_hi in the four
    // registers of the lane that owns the quad, a second one against W_lo the same with W_lo: their sum over the four
    // registers is the full product (y_hi + y_lo)(W_hi + W_lo) -- TWO MFMAs per product (three for hi*hi + lo*hi + hi*lo from
    // separate hi and lo tiles, which also drops lo*lo), one tile, half the LDS operand reads.  + one row of zeros.
    constexpr int DROWS = X3 ? 9 : 5;
    constexpr int plane = DROWS * pitch;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int ug = q >> 1, sq = q & 1;               // this lane's unit group and sequence
    const int d = blockIdx.x % p.dirs, s0 = (blockIdx.x / p.dirs) * 2;
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    const long arow = (long)dirs * 4 * HP, crow = (long)dirs * HP;

    for (int i = threadIdx.x * 4; i < 2 * plane; i += blockDim.x * 4) *(unsigned *)(smem + i) = 0u;

    // W_rec fragments of both unit groups: B lane (col = c, k slice = q) of gate g, chunk kc
    u32x8 wsp[2][4][KCS];
    [[maybe_unused]] u32x8 wsl[X3 ? 2 : 1][4][KCS];
    const char *Wd = (const char *)p.Wrec + (long)d * 4 * HP * HP * MELT;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kc = 0; kc < KCS; ++kc) {
                const long w0 = (long)(g * HP + 32 * wave + 16 * j + c) * HP + kc * 64 + q * 16;
                if constexpr (X3) sp_load_split((const float *)Wd + w0, wsp[j][g][kc], wsl[j][g][kc]);
                else wsp[j][g][kc] = sp_load_bf16(Wd + w0 * 2);
            }
    const int spidx = sp_index(c);
    // the two views of the tile: as A-operand lane, c is the tile row and q the k slice
    const int vrow0 = X3 ? (c < 8 ? c : 8) : ((c & 10) == 0 ? 2 * (c >> 2) + (c & 1) : 4);          // bf16: rows 0,1,4,5 = (seq 0, seq 1) x (even, odd)
    const int vrow1 = X3 ? (c >= 8 ? c - 8 : 8) : ((c & 10) == 8 ? 2 * ((c >> 2) & 1) + (c & 1) : 4);   // bf16: rows 8,9,12,13
    const int av0 = vrow0 * pitch + q * 16, av1 = vrow1 * pitch + q * 16;

    const int unit = 32 * wave + 16 * ug + c;
    const float pi = p.peep[(d * 3 + 0) * HP + unit], pf = p.peep[(d * 3 + 1) * HP + unit], po = p.peep[(d * 3 + 2) * HP + unit];
    const int sv = s0 + sq;
    const unsigned oA = sv * (int)arow + (d * HP + unit) * 4;
    const unsigned oC = sv * (int)crow + d * HP + unit;
    const unsigned oP = sv;
    const long stepA = (long)PS * arow, stepC = (long)PS * crow;
    const int k = unit & 63;
    const int oT = ((X3 ? 4 : 2) * sq + sp_parity(k)) * pitch + ((unit >> 6) * 32 + sp_pos(k)) * 2;      // (split-bf16: the hi row; lo two rows on)

    float cst = 0.f;
    f32x4 preA, preB;
    int ptA, ptB;
    auto prefetch = [&](int t, f32x4 &pre, int &pt) {
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        pt = (int)at32<unsigned char>(p.pat + (long)t * PS, oP);
        if (!X3 && p.pre16) {          // bf16 pre-activations: {n, i} and {f, o} as two dwords, widened by shift / mask like the hand-written loop
            const uint2 w = *(const uint2 *)((const char *)p.pre16 + ((long)t * stepA + oA) * 2);
            pre = f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
        } else pre = *(const f32x4 *)&at32<float>(p.acts + t * stepA, oA);
    };

    // accumulators live across steps (the sparse MFMA accumulates in place): register 0 = even tile row, seeded with the
    // staged pre-activation; register 1 = odd row, cleared; registers 2, 3 belong to zero rows in both views and stay 0
    // (split-bf16: the lo rows, cleared every step like register 1)
    f32x4 accp[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) accp[g] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto step = [&](int it, f32x4 &pre, int &pt) {
        const int t = d ? T - 1 - it : it;
        const char *ycur = smem + (it & 1) * plane;
        char *ynxt = smem + ((it + 1) & 1) * plane;
        const bool check = t >= p.Tmin;              // LstmLayer.cu:825,860
        float *actsT = p.acts + t * stepA;
        float *cellT = p.cell + t * stepC;
        float *thT = p.th + t * stepC;
        char *yT = (char *)p.y_op + t * stepC * MELT;

        int ptc;
        asm volatile("v_mov_b32 %0, %1" : "=&v"(ptc) : "v"(pt));
        const bool dummy = check && ptc == 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v_;
            asm volatile("v_mov_b32 %0, %1" : "=&v"(v_) : "v"(pre[g]));
            accp[g][0] = v_; accp[g][1] = 0.f;
            if constexpr (X3) { accp[g][2] = 0.f; accp[g][3] = 0.f; }
        }
        prefetch(d ? t - 2 : t + 2, pre, pt);

        // recurrent product (LstmLayer.cu:815-818 / :850-853), all four gates of both unit groups
        u32x4 a0[KCS], a1[KCS];
#pragma unroll
        for (int kc = 0; kc < KCS; ++kc) {
            a0[kc] = *(const u32x4 *)(ycur + av0 + kc * 64);
            a1[kc] = *(const u32x4 *)(ycur + av1 + kc * 64);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kc = 0; kc < KCS; ++kc) {
                if constexpr (X3) {          // small terms first
                    smma16(accp[g], a0[kc], wsl[0][g][kc], spidx);
                    smma16(accp[g], a0[kc], wsp[0][g][kc], spidx);
                    smma16(accp[g], a1[kc], wsl[1][g][kc], spidx);
                    smma16(accp[g], a1[kc], wsp[1][g][kc], spidx);
                } else {
                    smma16(accp[g], a0[kc], wsp[0][g][kc], spidx);
                    smma16(accp[g], a1[kc], wsp[1][g][kc], spidx);
                }
            }

        // ComputeBlockOutputFn, LstmLayer.cu:87-136 (bias is already inside the pre-activation)
        float s_[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (X3) s_[g] = (accp[g][0] + accp[g][1]) + (accp[g][2] + accp[g][3]);
            else s_[g] = accp[g][0] + accp[g][1];
            KEEP_TUPLE(accp[g], s_[g]);
        }
        const float cp = cst;
        const float ni = tanh_ref<false>(s_[0]);
        const float ig = logistic<false>(s_[1] + cp * pi);
        const float fg = logistic<false>(s_[2] + cp * pf);
        const float cs = __builtin_fmaf(ni, ig, cp * fg);     // (written out: the hand-written loop below rounds the same way)
        const float og = logistic<false>(s_[3] + cs * po);
        const float th = tanh_ref<false>(cs);
        const float y = th * og;
        const float yo = dummy ? 0.f : y;
        const float co = dummy ? 0.f : cs;            // :78-85 (zeroed in both directions here)
        cst = co;
        if constexpr (X3) {
            __bf16 yh, yl;
            split_bf16(yo, yh, yl);
            *(__bf16 *)(ynxt + oT) = yh;
            *(__bf16 *)(ynxt + oT + 2 * pitch) = yl;
        } else *(__bf16 *)(ynxt + oT) = (__bf16)yo;
        const f32x4 av = {ni, ig, fg, og};           // (dummy slots: never read back)
        *(f32x4 *)&at32<float>(actsT, oA) = av;
        at32<float>(cellT, oC) = co;
        at32<float>(thT, oC) = th;
        if constexpr (MELT == 4) at32<float>(yT, oC) = yo;
        else at32<__bf16>(yT, oC) = (__bf16)yo;
        lds_barrier();
    };

    prefetch(d ? T - 1 : 0, preA, ptA);
    prefetch(d ? T - 2 : 1, preB, ptB);
    lds_barrier();
    // branch-free pairs of steps, first pair peeled (see cn_lstm.hip)
    if (T >= 2) {
        step(0, preA, ptA);
        step(1, preB, ptB);
        int it = 2;
        for (; it + 1 < T; it += 2) {
            step(it, preA, ptA);
            step(it + 1, preB, ptB);
        }
        if (it < T) step(it, preA, ptA);
    } else {
        step(0, preA, ptA);
    }
}

// ---------------------------------------------------------------------------------------------
// forward, bf16, Hp = 128: the time loop written by hand
// ---------------------------------------------------------------------------------------------
// Same layout, same operand order and the same arithmetic as lstm_fwd_s2_kernel<P_BF16, 128> above (tests hold the two
// bit-equal on every real slot); only the instruction stream of the loop is ours.  With one wave per SIMD nothing overlaps
// a wave's own issue slots, so a step costs what its instructions cost: hipcc's loop issues ~180 instructions per step
// (50 scalar: the time index is turned into 64-bit addresses again every step; 8 64-bit VALU address adds; register
// shuffles around the v_pk_* pairs it forms); this one issues 86:
//   * addresses are a constant SGPR base + a 32-bit VGPR byte offset that moves by one time step per step (4 v_add_u32);
//     the prefetch two steps ahead uses the same offsets on bases shifted by two steps, so it needs no address of its own;
//     the last two steps have no prefetch (their copies of the body simply lack the loads: no clamping);
//   * the stage of step t+2 lands in the registers step t has just copied into the accumulators: two stages, loop body of
//     two steps, no copies besides the accumulator seeding the sparse MFMA needs anyway;
//   * MFMAs gate-major (n, i, f, o; per accumulator in the order of the C++ kernel), the activations of a finished gate
//     in the issue gaps of the following gates' MFMAs -- 8 of each MFMA's 16 cycles, one transcendental or two plain
//     VALU instructions -- and never sooner than four MFMAs after the gate's last one (the matrix pipe's result is not
//     interlocked against VALU reads: 11 wait states after an 8-pass MFMA);
//   * a transcendental's consumer never issues right behind it (trans forwarding hazard of gfx940+).
// Dummy slots: a slot is treated as a dummy whenever its pattern type is NONE; the reference (LstmLayer.cu:825,860) and the
// C++ kernels only look at the pattern type from t >= minSeqLength on, so for t < minSeqLength the UNUSED slots of a
// partial fraction (never real slots) hold zeros here and functor output there.  Nothing reads them: their errors are
// zero in the backward pass either way.
//
// Fixed registers (clobbered): v[224:239] the four accumulators, v[240:243] / v[244:247] the two stages, v[248:251]
// n, i, f, o of the step (one 16-byte store), v252 = cell state before the dummy select, v253 = tanh(cell state).
#ifdef CN_S2_STAMP_F
__device__ unsigned cn_s2_stamp_buf_f[4][8];
#define S2A_ST(i) "s_memtime s[98:99]\n\ts_waitcnt lgkmcnt(0)\n\ts_sub_u32 %[tq], s98, %[tl]\n\ts_add_u32 %[st" #i "], %[st" #i "], %[tq]\n\ts_mov_b32 %[tl], s98\n\t"
#else
#define S2A_ST(i) ""
#endif
#define S2A_K1 "0xbfb8aa3b"     /* -log2(e)   */
#define S2A_K2 "0xc038aa3b"     /* -2 log2(e) */
#define S2A_MF(acc, a, w) "v_smfmac_f32_16x16x64_bf16 " acc ", %[" a "], %[" w "], %[spidx]\n\t"
// Two forms of the loop text, one operand list (lstm_fwd_s2_asm_kernel<PRE16>):
//   32: fp32 pre-activations (the default): the stage is four dwords, copied into the accumulators;
//   16: bf16 pre-activations (option pre16, LstmRec::pre16): the stage is two dwords {n, i} / {f, o}, widened by the shift / mask
//       that takes the place of the copy (no extra instruction); the load sits at half the fp32 offset (x6 is free there).
#ifdef CN_S2_DIAG_HOT
#define S2A_PF32(PX4, PX2, PT) \
    "global_load_ubyte %[" PT "], %[oT], %[pat]\n\t" \
    "global_load_dwordx4 " PX4 ", %[oT], %[acts]\n\t"
#define S2A_PF16(PX4, PX2, PT) \
    "global_load_ubyte %[" PT "], %[oT], %[pat]\n\t" \
    "global_load_dwordx2 " PX2 ", %[oT], %[pre]\n\t"
#else
#define S2A_PF32(PX4, PX2, PT) \
    "global_load_ubyte %[" PT "], %[oP], %[patpf]\n\t" \
    "global_load_dwordx4 " PX4 ", %[oA], %[actspf]\n\t"
#define S2A_PF16(PX4, PX2, PT) \
    "global_load_ubyte %[" PT "], %[oP], %[patpf]\n\t" \
    "v_lshrrev_b32 %[x6], 1, %[oA]\n\t" \
    "global_load_dwordx2 " PX2 ", %[x6], %[prepf]\n\t"
#endif
// the four accumulator seeds of a step (n, i, f, o) from its stage registers P0 .. P3 (form 16 uses P0, P1 only)
#define S2A_SEED32_N(P0, P1, P2, P3) "v_mov_b32 v224, " P0 "\n\t"
#define S2A_SEED32_I(P0, P1, P2, P3) "v_mov_b32 v228, " P1 "\n\t"
#define S2A_SEED32_F(P0, P1, P2, P3) "v_mov_b32 v232, " P2 "\n\t"
#define S2A_SEED32_O(P0, P1, P2, P3) "v_mov_b32 v236, " P3 "\n\t"
#define S2A_SEED16_N(P0, P1, P2, P3) "v_lshlrev_b32 v224, 16, " P0 "\n\t"
#define S2A_SEED16_I(P0, P1, P2, P3) "v_and_b32 v228, 0xffff0000, " P0 "\n\t"
#define S2A_SEED16_F(P0, P1, P2, P3) "v_lshlrev_b32 v232, 16, " P1 "\n\t"
#define S2A_SEED16_O(P0, P1, P2, P3) "v_and_b32 v236, 0xffff0000, " P1 "\n\t"
#define S2A_NOPF "s_nop 1\n\t"
// FORM: 32 / 16 (above); P0 .. P3: the stage's registers; PT: its pattern-type operand; R0 / R1: LDS byte offsets of K chunk 0 / 1 of
// the tile read, WO: of the tile written; VM: outstanding vector-memory operations that may stay in flight at the top
#define S2A_STEP(FORM, P0, P1, P2, P3, PT, R0, R1, WO, VM, PFCODE) \
    "s_waitcnt vmcnt(" VM ")\n\t" \
    S2A_ST(0) \
    "ds_read_b128 %[a0], %[av0] offset:" R0 "\n\t" \
    "ds_read_b128 %[a1], %[av1] offset:" R0 "\n\t" \
    "ds_read_b128 %[a2], %[av0] offset:" R1 "\n\t" \
    "ds_read_b128 %[a3], %[av1] offset:" R1 "\n\t" \
    S2A_SEED##FORM##_N(P0, P1, P2, P3) \
    "v_mov_b32 v225, 0\n\t" \
    S2A_SEED##FORM##_I(P0, P1, P2, P3) \
    "v_mov_b32 v229, 0\n\t" \
    "v_cmp_eq_u32 vcc, 0, %[" PT "]\n\t" \
    "s_waitcnt lgkmcnt(3)\n\t" \
    S2A_MF("v[224:227]", "a0", "w0n0") \
    S2A_SEED##FORM##_F(P0, P1, P2, P3) \
    "v_mov_b32 v233, 0\n\t" \
    "s_waitcnt lgkmcnt(2)\n\t" \
    S2A_MF("v[224:227]", "a1", "w1n0") \
    S2A_SEED##FORM##_O(P0, P1, P2, P3) \
    "v_mov_b32 v237, 0\n\t" \
    "s_waitcnt lgkmcnt(1)\n\t" \
    S2A_MF("v[224:227]", "a2", "w0n1") \
    "s_waitcnt lgkmcnt(0)\n\t" \
    S2A_MF("v[224:227]", "a3", "w1n1") \
    S2A_ST(1) \
    PFCODE \
    S2A_MF("v[228:231]", "a0", "w0i0") \
    "v_add_u32 %[oA], %[oA], %[sA]\n\t" \
    S2A_MF("v[228:231]", "a1", "w1i0") \
    "v_add_u32 %[oP], %[oP], %[sP]\n\t" \
    S2A_MF("v[228:231]", "a2", "w0i1") \
    "v_add_u32 %[oC], %[oC], %[sC]\n\t" \
    S2A_MF("v[228:231]", "a3", "w1i1") \
    "v_add_u32 %[oY], %[oY], %[sY]\n\t" \
    "v_add_f32 %[x0], v224, v225\n\t" \
    S2A_MF("v[232:235]", "a0", "w0f0") \
    "v_mul_f32 %[x0], " S2A_K2 ", %[x0]\n\t" \
    S2A_MF("v[232:235]", "a1", "w1f0") \
    "v_exp_f32 %[x0], %[x0]\n\t" \
    S2A_MF("v[232:235]", "a2", "w0f1") \
    "v_add_f32 %[x1], v228, v229\n\t" \
    "v_add_f32 %[x0], 1.0, %[x0]\n\t" \
    S2A_MF("v[232:235]", "a3", "w1f1") \
    "v_fmac_f32 %[x1], %[pi], %[cst]\n\t" \
    "v_rcp_f32 %[x0], %[x0]\n\t" \
    S2A_MF("v[236:239]", "a0", "w0o0") \
    "v_mul_f32 %[x1], " S2A_K1 ", %[x1]\n\t" \
    S2A_MF("v[236:239]", "a1", "w1o0") \
    "v_exp_f32 %[x1], %[x1]\n\t" \
    S2A_MF("v[236:239]", "a2", "w0o1") \
    "v_fma_f32 v248, %[x0], 2.0, -1.0\n\t" \
    "v_add_f32 %[x1], 1.0, %[x1]\n\t" \
    S2A_MF("v[236:239]", "a3", "w1o1") \
    S2A_ST(2) \
    "v_add_f32 %[x2], v232, v233\n\t" \
    "v_rcp_f32 v249, %[x1]\n\t" \
    "v_fmac_f32 %[x2], %[pf], %[cst]\n\t" \
    "v_mul_f32 %[x2], " S2A_K1 ", %[x2]\n\t" \
    "v_exp_f32 %[x2], %[x2]\n\t" \
    "s_nop 0\n\t" \
    "v_add_f32 %[x2], 1.0, %[x2]\n\t" \
    "v_rcp_f32 v250, %[x2]\n\t" \
    "s_nop 0\n\t" \
    "v_mul_f32 %[x3], %[cst], v250\n\t" \
    "v_fma_f32 v252, v248, v249, %[x3]\n\t" \
    "v_add_f32 %[x4], v236, v237\n\t" \
    "v_fmac_f32 %[x4], %[po], v252\n\t" \
    "v_mul_f32 %[x5], " S2A_K2 ", v252\n\t" \
    "v_mul_f32 %[x4], " S2A_K1 ", %[x4]\n\t" \
    "v_exp_f32 %[x5], %[x5]\n\t" \
    "v_exp_f32 %[x4], %[x4]\n\t" \
    "v_add_f32 %[x5], 1.0, %[x5]\n\t" \
    "v_add_f32 %[x4], 1.0, %[x4]\n\t" \
    "v_rcp_f32 %[x5], %[x5]\n\t" \
    "v_rcp_f32 v251, %[x4]\n\t" \
    "v_fma_f32 v253, %[x5], 2.0, -1.0\n\t" \
    "v_cndmask_b32_e64 %[cst], v252, 0, vcc\n\t" \
    "v_mul_f32 %[x6], v253, v251\n\t" \
    "v_cvt_pk_bf16_f32 %[x6], %[x6], %[x6]\n\t" \
    "v_cndmask_b32_e64 %[x6], %[x6], 0, vcc\n\t" \
    S2A_ST(3) \
    "ds_write_b16 %[oT], %[x6] offset:" WO "\n\t" \
    "global_store_dwordx4 %[oA], v[248:251], %[acts1]\n\t" \
    "global_store_dword %[oC], %[cst], %[cell1]\n\t" \
    "global_store_dword %[oC], v253, %[th1]\n\t" \
    "global_store_short %[oY], %[x6], %[yop1]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    S2A_ST(4) \
    "s_barrier\n\t" \
    S2A_ST(5)
#define S2A_STEP_A(FORM, R0, R1, WO, VM, PFCODE) S2A_STEP(FORM, "v240", "v241", "v242", "v243", "ptA", R0, R1, WO, VM, PFCODE)
#define S2A_STEP_B(FORM, R0, R1, WO, VM, PFCODE) S2A_STEP(FORM, "v244", "v245", "v246", "v247", "ptB", R0, R1, WO, VM, PFCODE)
// first stages of the pass
#define S2A_INIT32 \
    "global_load_ubyte %[ptA], %[oP], %[pat]\n\t" \
    "global_load_dwordx4 v[240:243], %[oA], %[acts]\n\t" \
    "v_add_u32 %[x0], %[oP], %[sP]\n\t" \
    "v_add_u32 %[x1], %[oA], %[sA]\n\t" \
    "global_load_ubyte %[ptB], %[x0], %[pat]\n\t" \
    "global_load_dwordx4 v[244:247], %[x1], %[acts]\n\t"
#define S2A_INIT16 \
    "global_load_ubyte %[ptA], %[oP], %[pat]\n\t" \
    "v_lshrrev_b32 %[x2], 1, %[oA]\n\t" \
    "global_load_dwordx2 v[240:241], %[x2], %[pre]\n\t" \
    "v_add_u32 %[x0], %[oP], %[sP]\n\t" \
    "v_add_u32 %[x1], %[oA], %[sA]\n\t" \
    "global_load_ubyte %[ptB], %[x0], %[pat]\n\t" \
    "v_lshrrev_b32 %[x1], 1, %[x1]\n\t" \
    "global_load_dwordx2 v[244:245], %[x1], %[pre]\n\t"
// the whole loop text in one of the two forms
#define S2A_LOOP_TEXT(FORM) \
        /* accumulator rows 2, 3 belong to rows of zeros in both views and stay 0 for the whole pass */ \
        "v_mov_b32 v226, 0\n\tv_mov_b32 v227, 0\n\tv_mov_b32 v230, 0\n\tv_mov_b32 v231, 0\n\t" \
        "v_mov_b32 v234, 0\n\tv_mov_b32 v235, 0\n\tv_mov_b32 v238, 0\n\tv_mov_b32 v239, 0\n\t" \
        S2A_INIT##FORM \
        "s_waitcnt vmcnt(0)\n\t" \
        "1:\n\t" \
        S2A_STEP_A(FORM, "0", "64", "800", "10", S2A_PF##FORM("v[240:243]", "v[240:241]", "ptA")) \
        S2A_STEP_B(FORM, "800", "864", "0", "10", S2A_PF##FORM("v[244:247]", "v[244:245]", "ptB")) \
        "s_sub_u32 %[np], %[np], 1\n\t" \
        "s_cmp_lg_u32 %[np], 0\n\t" \
        "s_cbranch_scc1 1b\n\t" \
        "s_cmp_eq_u32 %[rem], 3\n\t" \
        "s_cbranch_scc0 2f\n\t" \
        S2A_STEP_A(FORM, "0", "64", "800", "10", S2A_PF##FORM("v[240:243]", "v[240:241]", "ptA")) \
        S2A_STEP_B(FORM, "800", "864", "0", "8", S2A_NOPF) \
        S2A_STEP_A(FORM, "0", "64", "800", "8", S2A_NOPF) \
        "s_branch 3f\n\t" \
        "2:\n\t" \
        S2A_STEP_A(FORM, "0", "64", "800", "8", S2A_NOPF) \
        S2A_STEP_B(FORM, "800", "864", "0", "8", S2A_NOPF) \
        "3:\n\t" \
        "s_waitcnt vmcnt(0)\n\t"               /* (nothing the compiler does not know of may be in flight when the statement ends) */

template <bool PRE16>
__global__ __launch_bounds__(256) void lstm_fwd_s2_asm_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HP = 128, KCS = 2;
    constexpr int pitch = lds_pitch(HP);             // 160
    constexpr int plane = 5 * pitch;                 // 800: the asm below carries it as literal offsets
    static_assert(plane == 800, "LDS offsets of the hand-written loop");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int ug = q >> 1, sq = q & 1;
    const int d = blockIdx.x % p.dirs, s0 = (blockIdx.x / p.dirs) * 2;
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    const long arow = (long)dirs * 4 * HP, crow = (long)dirs * HP;

    for (int i = threadIdx.x * 4; i < 2 * plane; i += blockDim.x * 4) *(unsigned *)(smem + i) = 0u;

    u32x8 w[2][4][KCS];
    const char *Wd = (const char *)p.Wrec + (long)d * 4 * HP * HP * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kc = 0; kc < KCS; ++kc)
                w[j][g][kc] = sp_load_bf16(Wd + ((long)(g * HP + 32 * wave + 16 * j + c) * HP + kc * 64 + q * 16) * 2);
    const int spidx = sp_index(c);
    const int vrow0 = (c & 10) == 0 ? 2 * (c >> 2) + (c & 1) : 4;
    const int vrow1 = (c & 10) == 8 ? 2 * ((c >> 2) & 1) + (c & 1) : 4;
    const unsigned av0 = vrow0 * pitch + q * 16, av1 = vrow1 * pitch + q * 16;

    const int unit = 32 * wave + 16 * ug + c;
    const float pi = p.peep[(d * 3 + 0) * HP + unit], pf = p.peep[(d * 3 + 1) * HP + unit], po = p.peep[(d * 3 + 2) * HP + unit];
    const int sv = s0 + sq;
    const int k = unit & 63;
    const unsigned oT = (2 * sq + sp_parity(k)) * pitch + ((unit >> 6) * 32 + sp_pos(k)) * 2;
    // Byte offsets of this lane at the first processed step, and what one step adds (mod 2^32: backwards for d = 1).  The
    // offsets are kept BIAS steps ahead of the time index and every base BIAS steps behind, so that no offset ever passes
    // zero (a negative one would be 4 GB up): they move on in MFMA gaps behind the step's prefetch, also in the last step, and
    // the stores of the step, issued after that, use bases one step back.
    constexpr long BIAS = 8;
    const long t0 = d ? T - 1 : 0, dt = d ? -1 : 1;
    const long stepA = (long)PS * arow, stepC = (long)PS * crow;
    const unsigned lC = (unsigned)(sv * (int)crow + d * HP + unit);          // elements into a cell row block (x 16 / 4 / 2 bytes)
    unsigned oA = (unsigned)((t0 + BIAS) * stepA * 4) + lC * 16, oC = (unsigned)((t0 + BIAS) * stepC * 4) + lC * 4, oY = (unsigned)((t0 + BIAS) * stepC * 2) + lC * 2;
    unsigned oP = (unsigned)((t0 + BIAS) * PS) + (unsigned)sv;
    const unsigned sA = (unsigned)(dt * stepA * 4), sC = (unsigned)(dt * stepC * 4), sY = (unsigned)(dt * stepC * 2), sP = (unsigned)(dt * PS);
    // pre-activations in: fp32 out of `acts` (stages two steps ahead; the activations overwrite them one step back), or -- PRE16 -- bf16
    // out of p.pre16, 8 bytes per unit and frame: the fp32 offsets halved
    const char *acts = (const char *)p.acts - BIAS * stepA * 4, *actspf = acts + 2 * dt * stepA * 4, *acts1 = acts - dt * stepA * 4;
    const char *pre = PRE16 ? (const char *)p.pre16 - BIAS * stepA * 2 : acts, *prepf = pre + 2 * dt * stepA * 2;
    const char *pat = p.pat - BIAS * PS, *patpf = pat + 2 * dt * PS;
    const char *cell1 = (const char *)p.cell - (BIAS + dt) * stepC * 4, *th1 = (const char *)p.th - (BIAS + dt) * stepC * 4;
    const char *yop1 = (const char *)p.y_op - (BIAS + dt) * stepC * 2;
    unsigned np = (unsigned)(T - 2) / 2;             // pairs of steps with a prefetch (T >= 4: at least one)
    const unsigned rem = (unsigned)T - 2 * np;       // 2 or 3 steps behind them

    float cst = 0.f;
    int ptA, ptB;
    u32x4 a0, a1, a2, a3;
    float x0, x1, x2, x3, x4, x5, x6;
#ifdef CN_S2_STAMP_F
    unsigned st[6] = {0, 0, 0, 0, 0, 0}, tq;
    unsigned long long tm;
    unsigned tl = (unsigned)__builtin_amdgcn_s_memtime();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    lds_barrier();
#ifdef CN_S2_STAMP_F
#define S2A_STAMP_OPS , [st0] "+s"(st[0]), [st1] "+s"(st[1]), [st2] "+s"(st[2]), [st3] "+s"(st[3]), [st4] "+s"(st[4]), [st5] "+s"(st[5]), [tq] "=&s"(tq), [tl] "+s"(tl), "={s[98:99]}"(tm)
#else
#define S2A_STAMP_OPS
#endif
    // (one operand list for both forms of the text; an operand a form does not name is simply unused)
#define S2A_OPERANDS \
        : [cst] "+v"(cst), [oA] "+v"(oA), [oC] "+v"(oC), [oY] "+v"(oY), [oP] "+v"(oP), [np] "+s"(np), \
          [ptA] "=&v"(ptA), [ptB] "=&v"(ptB), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), \
          [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3), [x4] "=&v"(x4), [x5] "=&v"(x5), [x6] "=&v"(x6) \
          S2A_STAMP_OPS \
        : [w0n0] "v"(w[0][0][0]), [w0n1] "v"(w[0][0][1]), [w1n0] "v"(w[1][0][0]), [w1n1] "v"(w[1][0][1]), \
          [w0i0] "v"(w[0][1][0]), [w0i1] "v"(w[0][1][1]), [w1i0] "v"(w[1][1][0]), [w1i1] "v"(w[1][1][1]), \
          [w0f0] "v"(w[0][2][0]), [w0f1] "v"(w[0][2][1]), [w1f0] "v"(w[1][2][0]), [w1f1] "v"(w[1][2][1]), \
          [w0o0] "v"(w[0][3][0]), [w0o1] "v"(w[0][3][1]), [w1o0] "v"(w[1][3][0]), [w1o1] "v"(w[1][3][1]), \
          [spidx] "v"(spidx), [av0] "v"(av0), [av1] "v"(av1), [oT] "v"(oT), [pi] "v"(pi), [pf] "v"(pf), [po] "v"(po), \
          [acts] "s"(acts), [actspf] "s"(actspf), [pre] "s"(pre), [prepf] "s"(prepf), [acts1] "s"(acts1), [cell1] "s"(cell1), [th1] "s"(th1), [yop1] "s"(yop1), \
          [pat] "s"(pat), [patpf] "s"(patpf), [sA] "s"(sA), [sC] "s"(sC), [sY] "s"(sY), [sP] "s"(sP), [rem] "s"(rem) \
        : "memory", "vcc", "scc", \
          "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", \
          "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253"
    if constexpr (PRE16) asm volatile(S2A_LOOP_TEXT(16) S2A_OPERANDS);
    else                 asm volatile(S2A_LOOP_TEXT(32) S2A_OPERANDS);
#undef S2A_OPERANDS
#undef S2A_STAMP_OPS
#ifdef CN_S2_STAMP_F
    if (blockIdx.x == 0 && lane == 0) {
        for (int i = 0; i < 6; ++i) cn_s2_stamp_buf_f[wave][i] = st[i];
        cn_s2_stamp_buf_f[wave][7] = (unsigned)(__builtin_amdgcn_s_memrealtime() - rt0);
    }
#endif
}
#ifdef CN_S2_STAMP_F
extern "C" int cn_dbg_read_stamps_s2f(unsigned *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cn_s2_stamp_buf_f), sizeof(cn_s2_stamp_buf_f)); }
#endif

// ---------------------------------------------------------------------------------------------
// forward, split-bf16 (P_X3), Hp = 128: the time loop written by hand
// ---------------------------------------------------------------------------------------------
// The parity mode in the same cut: fp32 y / W_rec in memory, every operand split into bf16 hi + lo (cn_lstm_device.h); the
// tile carries a sequence's hi and lo halves in the four rows of its quad, and a product is TWO sparse MFMAs ([hi; lo] x W_lo,
// [hi; lo] x W_hi) whose four accumulator registers are summed (lstm_fwd_s2_kernel<P_X3>).  32 MFMAs per wave and step: this
// loop is bound by the matrix pipe, and everything else (activations, prefetch, address updates) rides in its issue gaps;
// compiled, the same cut lost to the 8-wave kernel (one wave per SIMD has to issue its ~190 other instructions somewhere),
// which is why the mode takes this cut only where the hand-written loops apply.  W_rec hi and lo fragments: 256 registers,
// all in AGPRs.  Every step is the same code (guard steps, cn_api.cpp: dalloc_guarded): two stages, loop body of two steps,
// left after any step.  Arithmetic, operand order and layout of lstm_fwd_s2_kernel<P_X3, 128> (bit-equal on real slots);
// dummy slots as in the bf16 loop.  LDS: 9 rows of 160 bytes per tile buffer, buffers at 0 and 1440.
#define X3A_MF(acc, a, w) "v_smfmac_f32_16x16x64_bf16 " acc ", %[" a "], %[" w "], %[spidx]\n\t"
#define X3A_MF2(acc, A, BH, BL) X3A_MF(acc, A, BL) X3A_MF(acc, A, BH)
// the eight MFMAs of one gate (views 0 / 1 x K chunks 0 / 1 x W_lo, W_hi); F0..F3: fillers behind each pair
#define X3A_GATE(acc, G, F0, F1, F2, F3) \
    X3A_MF2(acc, "a0", "h0" G "0", "l0" G "0") F0 \
    X3A_MF2(acc, "a1", "h1" G "0", "l1" G "0") F1 \
    X3A_MF2(acc, "a2", "h0" G "1", "l0" G "1") F2 \
    X3A_MF2(acc, "a3", "h1" G "1", "l1" G "1") F3
#define X3A_INIT(R0, R1, R2, R3, PX) "v_mov_b32 " R0 ", " PX "\n\tv_mov_b32 " R1 ", 0\n\tv_mov_b32 " R2 ", 0\n\tv_mov_b32 " R3 ", 0\n\t"
#define X3A_STEP(PXT, PX0, PX1, PX2, PX3, PT, R, WO) \
    "s_waitcnt vmcnt(10)\n\t" \
    "ds_read_b128 %[a0], %[av0] offset:" R "\n\t" \
    "ds_read_b128 %[a1], %[av1] offset:" R "\n\t" \
    "ds_read_b128 %[a2], %[av0] offset:" R "+64\n\t" \
    "ds_read_b128 %[a3], %[av1] offset:" R "+64\n\t" \
    X3A_INIT("v224", "v225", "v226", "v227", PX0) \
    "v_cmp_eq_u32 vcc, 0, %[" PT "]\n\t" \
    "s_waitcnt lgkmcnt(3)\n\t" \
    X3A_MF2("v[224:227]", "a0", "h0n0", "l0n0") \
    X3A_INIT("v228", "v229", "v230", "v231", PX1) \
    "s_waitcnt lgkmcnt(2)\n\t" \
    X3A_MF2("v[224:227]", "a1", "h1n0", "l1n0") \
    X3A_INIT("v232", "v233", "v234", "v235", PX2) \
    "s_waitcnt lgkmcnt(1)\n\t" \
    X3A_MF2("v[224:227]", "a2", "h0n1", "l0n1") \
    X3A_INIT("v236", "v237", "v238", "v239", PX3) \
    "s_waitcnt lgkmcnt(0)\n\t" \
    X3A_MF2("v[224:227]", "a3", "h1n1", "l1n1") \
    "global_load_ubyte %[" PT "], %[oP], %[patpf]\n\t" \
    "global_load_dwordx4 " PXT ", %[oA], %[actspf]\n\t" \
    X3A_GATE("v[228:231]", "i", \
        "v_add_u32 %[oA], %[oA], %[sA]\n\tv_add_u32 %[oP], %[oP], %[sP]\n\t", \
        "v_add_u32 %[oC], %[oC], %[sC]\n\t", \
        "v_add_f32 %[x0], v224, v225\n\tv_add_f32 %[x7], v226, v227\n\t", \
        "v_add_f32 %[x0], %[x0], %[x7]\n\tv_mul_f32 %[x0], " S2A_K2 ", %[x0]\n\t") \
    X3A_GATE("v[232:235]", "f", \
        "v_exp_f32 %[x0], %[x0]\n\t", \
        "v_add_f32 %[x1], v228, v229\n\tv_add_f32 %[x7], v230, v231\n\t", \
        "v_add_f32 %[x1], %[x1], %[x7]\n\tv_add_f32 %[x0], 1.0, %[x0]\n\t", \
        "v_fmac_f32 %[x1], %[pi], %[cst]\n\tv_rcp_f32 %[x0], %[x0]\n\t") \
    X3A_GATE("v[236:239]", "o", \
        "v_mul_f32 %[x1], " S2A_K1 ", %[x1]\n\t", \
        "v_exp_f32 %[x1], %[x1]\n\tv_fma_f32 v248, %[x0], 2.0, -1.0\n\t", \
        "v_add_f32 %[x2], v232, v233\n\tv_add_f32 %[x7], v234, v235\n\tv_add_f32 %[x1], 1.0, %[x1]\n\t", \
        "v_add_f32 %[x2], %[x2], %[x7]\n\tv_rcp_f32 v249, %[x1]\n\t") \
    "v_fmac_f32 %[x2], %[pf], %[cst]\n\t" \
    "v_mul_f32 %[x2], " S2A_K1 ", %[x2]\n\t" \
    "v_exp_f32 %[x2], %[x2]\n\t" \
    "s_nop 0\n\t" \
    "v_add_f32 %[x2], 1.0, %[x2]\n\t" \
    "v_rcp_f32 v250, %[x2]\n\t" \
    "s_nop 0\n\t" \
    "v_mul_f32 %[x3], %[cst], v250\n\t" \
    "v_fma_f32 v252, v248, v249, %[x3]\n\t" \
    "v_mul_f32 %[x5], " S2A_K2 ", v252\n\t" \
    "v_exp_f32 %[x5], %[x5]\n\t" \
    "s_nop 1\n\t" \
    "v_add_f32 %[x4], v236, v237\n\t" \
    "v_add_f32 %[x7], v238, v239\n\t" \
    "v_add_f32 %[x5], 1.0, %[x5]\n\t" \
    "v_add_f32 %[x4], %[x4], %[x7]\n\t" \
    "v_rcp_f32 %[x5], %[x5]\n\t" \
    "v_fmac_f32 %[x4], %[po], v252\n\t" \
    "v_cndmask_b32_e64 %[cst], v252, 0, vcc\n\t" \
    "v_mul_f32 %[x4], " S2A_K1 ", %[x4]\n\t" \
    "v_fma_f32 v253, %[x5], 2.0, -1.0\n\t" \
    "v_exp_f32 %[x4], %[x4]\n\t" \
    "s_nop 0\n\t" \
    "v_add_f32 %[x4], 1.0, %[x4]\n\t" \
    "v_rcp_f32 v251, %[x4]\n\t" \
    "s_nop 0\n\t" \
    "v_mul_f32 %[x6], v253, v251\n\t" \
    "v_cndmask_b32_e64 %[x6], %[x6], 0, vcc\n\t" \
    "v_cvt_pk_bf16_f32 %[x0], %[x6], %[x6]\n\t" \
    "v_lshlrev_b32 %[x1], 16, %[x0]\n\t" \
    "v_sub_f32 %[x1], %[x6], %[x1]\n\t" \
    "v_cvt_pk_bf16_f32 %[x1], %[x1], %[x1]\n\t" \
    "ds_write_b16 %[oT], %[x0] offset:" WO "\n\t" \
    "ds_write_b16 %[oT], %[x1] offset:" WO "+320\n\t" \
    "global_store_dwordx4 %[oA], v[248:251], %[acts1]\n\t" \
    "global_store_dword %[oC], %[cst], %[cell1]\n\t" \
    "global_store_dword %[oC], v253, %[th1]\n\t" \
    "global_store_dword %[oC], %[x6], %[yop1]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "s_barrier\n\t" \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t" \
    "s_cbranch_scc1 9f\n\t"

__global__ __launch_bounds__(256) void lstm_fwd_s2_x3_asm_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HP = 128, KCS = 2;
    constexpr int pitch = lds_pitch(HP);             // 160
    constexpr int plane = 9 * pitch;                 // 1440: the asm carries it (and 2 * pitch = 320) as literals
    static_assert(plane == 1440 && CN_GUARD_STEPS >= 3, "LDS offsets / prefetch distance of the hand-written loop");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int ug = q >> 1, sq = q & 1;
    const int d = blockIdx.x % p.dirs, s0 = (blockIdx.x / p.dirs) * 2;
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    const long arow = (long)dirs * 4 * HP, crow = (long)dirs * HP;

    for (int i = threadIdx.x * 4; i < 2 * plane; i += blockDim.x * 4) *(unsigned *)(smem + i) = 0u;

    u32x8 wh[2][4][KCS], wl[2][4][KCS];
    const float *Wd = (const float *)p.Wrec + (long)d * 4 * HP * HP;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kc = 0; kc < KCS; ++kc)
                sp_load_split(Wd + (long)(g * HP + 32 * wave + 16 * j + c) * HP + kc * 64 + q * 16, wh[j][g][kc], wl[j][g][kc]);
    const int spidx = sp_index(c);
    const unsigned av0 = (c < 8 ? c : 8) * pitch + q * 16, av1 = (c >= 8 ? c - 8 : 8) * pitch + q * 16;

    const int unit = 32 * wave + 16 * ug + c;
    const float pi = p.peep[(d * 3 + 0) * HP + unit], pf = p.peep[(d * 3 + 1) * HP + unit], po = p.peep[(d * 3 + 2) * HP + unit];
    const int sv = s0 + sq;
    const int k = unit & 63;
    const unsigned oT = (4 * sq + sp_parity(k)) * pitch + ((unit >> 6) * 32 + sp_pos(k)) * 2;
    // offsets BIAS steps ahead, bases BIAS steps behind (see lstm_fwd_s2_asm_kernel); y is fp32 here: it shares the cell offset
    constexpr long BIAS = 8;
    const long t0 = d ? T - 1 : 0, dt = d ? -1 : 1;
    const long stepA = (long)PS * arow, stepC = (long)PS * crow;
    const unsigned lC = (unsigned)(sv * (int)crow + d * HP + unit);
    unsigned oA = (unsigned)((t0 + BIAS) * stepA * 4) + lC * 16, oC = (unsigned)((t0 + BIAS) * stepC * 4) + lC * 4;
    unsigned oP = (unsigned)((t0 + BIAS) * PS) + (unsigned)sv;
    const unsigned sA = (unsigned)(dt * stepA * 4), sC = (unsigned)(dt * stepC * 4), sP = (unsigned)(dt * PS);
    const char *acts = (const char *)p.acts - BIAS * stepA * 4, *actspf = acts + 2 * dt * stepA * 4, *acts1 = acts - dt * stepA * 4;
    const char *pat = p.pat - BIAS * PS, *patpf = pat + 2 * dt * PS;
    const char *cell1 = (const char *)p.cell - (BIAS + dt) * stepC * 4, *th1 = (const char *)p.th - (BIAS + dt) * stepC * 4;
    const char *yop1 = (const char *)p.y_op - (BIAS + dt) * stepC * 4;
    unsigned cnt = (unsigned)T - 1;                  // steps behind the current one

    float cst = 0.f;
    int ptA, ptB;
    u32x4 a0, a1, a2, a3;
    float x0, x1, x2, x3, x4, x5, x6, x7;
    lds_barrier();
    asm volatile(
        "global_load_ubyte %[ptA], %[oP], %[pat]\n\t"
        "global_load_dwordx4 v[240:243], %[oA], %[acts]\n\t"
        "v_add_u32 %[x0], %[oP], %[sP]\n\t"
        "v_add_u32 %[x1], %[oA], %[sA]\n\t"
        "global_load_ubyte %[ptB], %[x0], %[pat]\n\t"
        "global_load_dwordx4 v[244:247], %[x1], %[acts]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "1:\n\t"
        X3A_STEP("v[240:243]", "v240", "v241", "v242", "v243", "ptA", "0", "1440")
        X3A_STEP("v[244:247]", "v244", "v245", "v246", "v247", "ptB", "1440", "0")
        "s_branch 1b\n\t"
        "9:\n\t"
        "s_waitcnt vmcnt(0)\n\t"               // (prefetches of the last two steps: see lstm_bwd_s2_asm_kernel)
        : [cst] "+v"(cst), [oA] "+v"(oA), [oC] "+v"(oC), [oP] "+v"(oP), [cnt] "+s"(cnt),
          [ptA] "=&v"(ptA), [ptB] "=&v"(ptB), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3),
          [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3), [x4] "=&v"(x4), [x5] "=&v"(x5), [x6] "=&v"(x6), [x7] "=&v"(x7)
        : [h0n0] "a"(wh[0][0][0]), [h0n1] "a"(wh[0][0][1]), [h1n0] "a"(wh[1][0][0]), [h1n1] "a"(wh[1][0][1]),
          [h0i0] "a"(wh[0][1][0]), [h0i1] "a"(wh[0][1][1]), [h1i0] "a"(wh[1][1][0]), [h1i1] "a"(wh[1][1][1]),
          [h0f0] "a"(wh[0][2][0]), [h0f1] "a"(wh[0][2][1]), [h1f0] "a"(wh[1][2][0]), [h1f1] "a"(wh[1][2][1]),
          [h0o0] "a"(wh[0][3][0]), [h0o1] "a"(wh[0][3][1]), [h1o0] "a"(wh[1][3][0]), [h1o1] "a"(wh[1][3][1]),
          [l0n0] "a"(wl[0][0][0]), [l0n1] "a"(wl[0][0][1]), [l1n0] "a"(wl[1][0][0]), [l1n1] "a"(wl[1][0][1]),
          [l0i0] "a"(wl[0][1][0]), [l0i1] "a"(wl[0][1][1]), [l1i0] "a"(wl[1][1][0]), [l1i1] "a"(wl[1][1][1]),
          [l0f0] "a"(wl[0][2][0]), [l0f1] "a"(wl[0][2][1]), [l1f0] "a"(wl[1][2][0]), [l1f1] "a"(wl[1][2][1]),
          [l0o0] "a"(wl[0][3][0]), [l0o1] "a"(wl[0][3][1]), [l1o0] "a"(wl[1][3][0]), [l1o1] "a"(wl[1][3][1]),
          [spidx] "v"(spidx), [av0] "v"(av0), [av1] "v"(av1), [oT] "v"(oT), [pi] "v"(pi), [pf] "v"(pf), [po] "v"(po),
          [acts] "s"(acts), [actspf] "s"(actspf), [acts1] "s"(acts1), [cell1] "s"(cell1), [th1] "s"(th1), [yop1] "s"(yop1),
          [pat] "s"(pat), [patpf] "s"(patpf), [sA] "s"(sA), [sC] "s"(sC), [sP] "s"(sP)
        : "memory", "vcc", "scc",
          "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239",
          "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253");
}

// ---------------------------------------------------------------------------------------------
// forward, bf16, Hp = 256 on ONE CU ("s2w"): the compiled twin of the hand-written loop below (CN_NO_S2W_ASM=1 selects it)
// ---------------------------------------------------------------------------------------------
// W_rec of a 256-unit direction is 512 KB of bf16 -- the whole register file of a CU.  The cluster kernels split the units
// over two CUs and pay an L2 hop per step (0.8-1.0 us of their 1.3 us).  Here one CU keeps three of the four K chunks of
// every fragment in registers (384 per lane) and streams the fourth from LDS every step (128 KB: 16 fragments per wave as
// two lane-linear 16-byte planes), so a step is bound by its 64 sparse MFMAs per wave (~0.6 us) instead of the hop.
// Cut as in the s2 kernels: two sequences per workgroup, 4 waves, a wave owns 64 units = two PAIRS of unit groups; each
// pair accumulates through the two zero-padded views of the tile; a lane owns one (unit, sequence) of each pair.
__global__ __launch_bounds__(256) void lstm_fwd_s2w_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HP = 256, KCS = 4, KR = 3;         // K chunks in all / register resident
    constexpr int pitch = lds_pitch(HP);             // 288
    constexpr int plane = 5 * pitch;
    constexpr int WL = 2 * plane + 64;               // (16-byte aligned) base of the streamed fragments: [wave][tile*4 + gate][half][lane][16 B]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int ug = q >> 1, sq = q & 1;
    const int d = blockIdx.x % p.dirs, s0 = (blockIdx.x / p.dirs) * 2;
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    const long arow = (long)dirs * 4 * HP, crow = (long)dirs * HP;

    for (int i = threadIdx.x * 4; i < 2 * plane; i += blockDim.x * 4) *(unsigned *)(smem + i) = 0u;

    u32x8 wreg[4][4][KR];
    const char *Wd = (const char *)p.Wrec + (long)d * 4 * HP * HP * 2;
    char *wl = smem + WL + wave * 32768 + lane * 16;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const long w0 = (long)(g * HP + 64 * wave + 16 * u + c) * HP + q * 16;
#pragma unroll
            for (int kc = 0; kc < KR; ++kc) wreg[u][g][kc] = sp_load_bf16(Wd + (w0 + kc * 64) * 2);
            const u32x8 f = sp_load_bf16(Wd + (w0 + KR * 64) * 2);
            *(u32x4 *)(wl + (u * 4 + g) * 2048) = __builtin_shufflevector(f, f, 0, 1, 2, 3);
            *(u32x4 *)(wl + (u * 4 + g) * 2048 + 1024) = __builtin_shufflevector(f, f, 4, 5, 6, 7);
        }
    const int spidx = sp_index(c);
    const int vrow0 = (c & 10) == 0 ? 2 * (c >> 2) + (c & 1) : 4;
    const int vrow1 = (c & 10) == 8 ? 2 * ((c >> 2) & 1) + (c & 1) : 4;
    const int av0 = vrow0 * pitch + q * 16, av1 = vrow1 * pitch + q * 16;

    int unit[2], oT[2];
    float pi[2], pf[2], po[2], cst[2] = {0.f, 0.f};
    unsigned oA[2], oC[2];
    const int sv = s0 + sq;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        unit[s] = 64 * wave + 32 * s + 16 * ug + c;
        pi[s] = p.peep[(d * 3 + 0) * HP + unit[s]]; pf[s] = p.peep[(d * 3 + 1) * HP + unit[s]]; po[s] = p.peep[(d * 3 + 2) * HP + unit[s]];
        oA[s] = sv * (int)arow + (d * HP + unit[s]) * 4;
        oC[s] = sv * (int)crow + d * HP + unit[s];
        const int k = unit[s] & 63;
        oT[s] = (2 * sq + sp_parity(k)) * pitch + ((unit[s] >> 6) * 32 + sp_pos(k)) * 2;
    }
    const unsigned oP = sv;
    const long stepA = (long)PS * arow, stepC = (long)PS * crow;

    f32x4 preA[2], preB[2];
    int ptA, ptB;
    auto prefetch = [&](int t, f32x4 (&pre)[2], int &pt) {
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        pt = (int)at32<unsigned char>(p.pat + (long)t * PS, oP);
#pragma unroll
        for (int s = 0; s < 2; ++s) pre[s] = *(const f32x4 *)&at32<float>(p.acts + t * stepA, oA[s]);
    };

    auto step = [&](int it, f32x4 (&pre)[2], int &pt) {
        const int t = d ? T - 1 - it : it;
        const char *ycur = smem + (it & 1) * plane;
        char *ynxt = smem + ((it + 1) & 1) * plane;
        const bool check = t >= p.Tmin;
        float *actsT = p.acts + t * stepA;
        float *cellT = p.cell + t * stepC;
        float *thT = p.th + t * stepC;
        char *yT = (char *)p.y_op + t * stepC * 2;
        int ptc;
        asm volatile("v_mov_b32 %0, %1" : "=&v"(ptc) : "v"(pt));
        const bool dummy = check && ptc == 0;
        f32x4 g_[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) asm volatile("v_mov_b32 %0, %1" : "=&v"(g_[s][g]) : "v"(pre[s][g]));
        prefetch(d ? t - 2 : t + 2, pre, pt);

#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = f32x4{g_[s][g], 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < KCS; ++kc) {
                const u32x4 a0 = *(const u32x4 *)(ycur + av0 + kc * 64), a1 = *(const u32x4 *)(ycur + av1 + kc * 64);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (kc < KR) {
                        smma16(acc[g], a0, wreg[2 * s][g][kc], spidx);
                        smma16(acc[g], a1, wreg[2 * s + 1][g][kc], spidx);
                    } else {
                        const char *f0 = wl + ((2 * s) * 4 + g) * 2048, *f1 = wl + ((2 * s + 1) * 4 + g) * 2048;
                        smma16(acc[g], a0, sp_join(*(const u32x4 *)f0, *(const u32x4 *)(f0 + 1024)), spidx);
                        smma16(acc[g], a1, sp_join(*(const u32x4 *)f1, *(const u32x4 *)(f1 + 1024)), spidx);
                    }
                }
            }
            float s_[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) { s_[g] = acc[g][0] + acc[g][1]; KEEP_TUPLE(acc[g], s_[g]); }
            const float cp = cst[s];
            const float ni = tanh_ref<false>(s_[0]);
            const float ig = logistic<false>(s_[1] + cp * pi[s]);
            const float fg = logistic<false>(s_[2] + cp * pf[s]);
            const float cs = __builtin_fmaf(ni, ig, cp * fg);
            const float og = logistic<false>(s_[3] + cs * po[s]);
            const float th = tanh_ref<false>(cs);
            const float y = th * og;
            const float yo = dummy ? 0.f : y;
            const float co = dummy ? 0.f : cs;
            cst[s] = co;
            *(__bf16 *)(ynxt + oT[s]) = (__bf16)yo;
            const f32x4 av = {ni, ig, fg, og};
            *(f32x4 *)&at32<float>(actsT, oA[s]) = av;
            at32<float>(cellT, oC[s]) = co;
            at32<float>(thT, oC[s]) = th;
            at32<__bf16>(yT, oC[s]) = (__bf16)yo;
        }
        lds_barrier();
    };

    prefetch(d ? T - 1 : 0, preA, ptA);
    prefetch(d ? T - 2 : 1, preB, ptB);
    lds_barrier();
    if (T >= 2) {
        step(0, preA, ptA);
        step(1, preB, ptB);
        int it = 2;
        for (; it + 1 < T; it += 2) {
            step(it, preA, ptA);
            step(it + 1, preB, ptB);
        }
        if (it < T) step(it, preA, ptA);
    } else {
        step(0, preA, ptA);
    }
}

// ---------------------------------------------------------------------------------------------
// forward, bf16, Hp = 256 on ONE CU: the time loop written by hand
// ---------------------------------------------------------------------------------------------
// The cut of lstm_fwd_s2w_kernel above (layout, operand order per accumulator and arithmetic: bit-equal on real slots) with the
// instruction stream of the hand-written Hp = 128 loops: one `asm volatile`, constant SGPR bases + 32-bit VGPR offsets that
// move by a step per step, two stages two steps ahead, every step the same code (guard steps, cn_api.cpp: dalloc_guarded), the
// loop left after any step.  W_rec: K chunks 0 and 1 of all 16 fragments of a wave in the 256 AGPRs, chunk 2 in VGPRs except
// two fragments, those two and chunk 3 (18 fragments, 36 KB per wave) streamed from LDS every step through three 8-register
// buffers, each refilled right behind the MFMA that consumed it (nine or more MFMAs before its next use).  The 64 MFMAs of a
// step run gate-major, pair A then pair B, on ONE set of accumulators: pair A's activations fill the gaps of the following gates'
// and of pair B's MFMAs, pair B's accumulators are seeded as pair A's sums are read, only pair B's output gate is exposed
// behind the last MFMA.  The step body is generated (tools/gen_s2w_loop.py -> cn_lstm_s2w_loop.inc): the schedule is a table
// there, and the `s_waitcnt lgkmcnt` counts of the 44 LDS reads per step are derived from it.
// vmcnt: a step issues, in order, pair A's prefetch (2 loads), pair B's (1), pair A's stores (4), pair B's (4).
#include "cn_lstm_s2w_loop.inc"
#ifdef CN_S2W_STAMP
__device__ unsigned cn_s2w_stamp_buf[4][8];
#endif

__global__ __launch_bounds__(256) void lstm_fwd_s2w_asm_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HP = 256;
    constexpr int pitch = lds_pitch(HP);             // 288
    constexpr int plane = 5 * pitch;                 // 1440: the generated loop carries it as literal offsets
    constexpr int WL = 2 * plane + 64;               // streamed fragments: [wave][fragment j][half][lane][16 B]
    static_assert(plane == 1440 && CN_GUARD_STEPS >= 3, "LDS offsets / prefetch distance of the hand-written loop");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int ug = q >> 1, sq = q & 1;
    const int d = blockIdx.x % p.dirs, s0 = (blockIdx.x / p.dirs) * 2;
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    const long arow = (long)dirs * 4 * HP, crow = (long)dirs * HP;

    for (int i = threadIdx.x * 4; i < 2 * plane; i += blockDim.x * 4) *(unsigned *)(smem + i) = 0u;

    // fragment (pair S, view V, gate g, chunk K): rows g * HP + 64 * wave + 16 * (2 S + V) + c of W_rec, K values [64 K, 64 K + 64)
    const char *Wd = (const char *)p.Wrec + (long)d * 4 * HP * HP * 2;
    auto frag = [&](int S, int V, int g, int K) { return sp_load_bf16(Wd + ((long)(g * HP + 64 * wave + 16 * (2 * S + V) + c) * HP + K * 64 + q * 16) * 2); };
    u32x8 wa[2][2][4][2], wv[2][2][4];
#pragma unroll
    for (int S = 0; S < 2; ++S)
#pragma unroll
        for (int V = 0; V < 2; ++V)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                wa[S][V][g][0] = frag(S, V, g, 0); wa[S][V][g][1] = frag(S, V, g, 1);
                if (!(V == 1 && g == 0)) wv[S][V][g] = frag(S, V, g, 2);
            }
    const unsigned wl = WL + wave * (S2W_STREAM_COUNT * 2048) + lane * 16;
    {
        constexpr int tab[S2W_STREAM_COUNT][4] = S2W_STREAM_TABLE;
#pragma unroll
        for (int j = 0; j < S2W_STREAM_COUNT; ++j) {
            const u32x8 f = frag(tab[j][0], tab[j][1], tab[j][2], tab[j][3]);
            *(u32x4 *)(smem + wl + j * 2048) = __builtin_shufflevector(f, f, 0, 1, 2, 3);
            *(u32x4 *)(smem + wl + j * 2048 + 1024) = __builtin_shufflevector(f, f, 4, 5, 6, 7);
        }
    }
    const int spidx = sp_index(c);
    const int vrow0 = (c & 10) == 0 ? 2 * (c >> 2) + (c & 1) : 4;
    const int vrow1 = (c & 10) == 8 ? 2 * ((c >> 2) & 1) + (c & 1) : 4;
    const unsigned av0 = vrow0 * pitch + q * 16, av1 = vrow1 * pitch + q * 16;

    // pair A: unit 64 wave + 16 ug + c, pair B: 32 units on (acts + 512 B, cell / tanh + 128 B, y + 64 B)
    const int unit = 64 * wave + 16 * ug + c;
    const float piA = p.peep[(d * 3 + 0) * HP + unit], pfA = p.peep[(d * 3 + 1) * HP + unit], poA = p.peep[(d * 3 + 2) * HP + unit];
    const float piB = p.peep[(d * 3 + 0) * HP + unit + 32], pfB = p.peep[(d * 3 + 1) * HP + unit + 32], poB = p.peep[(d * 3 + 2) * HP + unit + 32];
    const int sv = s0 + sq;
    const int kA = unit & 63, kB = (unit + 32) & 63;
    const unsigned oTA = (2 * sq + sp_parity(kA)) * pitch + ((unit >> 6) * 32 + sp_pos(kA)) * 2;
    const unsigned oTB = (2 * sq + sp_parity(kB)) * pitch + (((unit + 32) >> 6) * 32 + sp_pos(kB)) * 2;
    // offsets BIAS steps ahead, bases BIAS steps behind (see lstm_fwd_s2_asm_kernel)
    constexpr long BIAS = 8;
    const long t0 = d ? T - 1 : 0, dt = d ? -1 : 1;
    const long stepA = (long)PS * arow, stepC = (long)PS * crow;
    const unsigned lC = (unsigned)(sv * (int)crow + d * HP + unit);
    unsigned oA = (unsigned)((t0 + BIAS) * stepA * 4) + lC * 16, oC = (unsigned)((t0 + BIAS) * stepC * 4) + lC * 4, oY = (unsigned)((t0 + BIAS) * stepC * 2) + lC * 2;
    unsigned oP = (unsigned)((t0 + BIAS) * PS) + (unsigned)sv;
    const unsigned sA = (unsigned)(dt * stepA * 4), sC = (unsigned)(dt * stepC * 4), sY = (unsigned)(dt * stepC * 2), sP = (unsigned)(dt * PS);
    const char *acts = (const char *)p.acts - BIAS * stepA * 4, *actspf = acts + 2 * dt * stepA * 4, *actspf1 = acts + dt * stepA * 4, *acts1 = acts - dt * stepA * 4;
    const char *pat = p.pat - BIAS * PS, *patpf = pat + 2 * dt * PS;
    const char *cell1 = (const char *)p.cell - (BIAS + dt) * stepC * 4, *th1 = (const char *)p.th - (BIAS + dt) * stepC * 4;
    const char *yop1 = (const char *)p.y_op - (BIAS + dt) * stepC * 2;
    unsigned cnt = (unsigned)T - 1;                  // steps behind the current one

    float cstA = 0.f, cstB = 0.f;
    int ptP, ptQ;
    u32x4 a00, a01, a10, a11, a20, a21, a30, a31;
    float x4, x5;
#ifdef CN_S2W_STAMP
    unsigned st[6] = {0, 0, 0, 0, 0, 0}, tq;
    unsigned long long tm;
    unsigned tl = (unsigned)__builtin_amdgcn_s_memtime();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    lds_barrier();
    asm volatile(
        // accumulator rows 2, 3 belong to rows of zeros in both views and stay 0 for the whole pass
        "v_mov_b32 v226, 0\n\tv_mov_b32 v227, 0\n\tv_mov_b32 v230, 0\n\tv_mov_b32 v231, 0\n\t"
        "v_mov_b32 v234, 0\n\tv_mov_b32 v235, 0\n\tv_mov_b32 v238, 0\n\tv_mov_b32 v239, 0\n\t"
        // stages of the first two steps, the first three streamed fragments
        "global_load_ubyte %[ptP], %[oP], %[pat]\n\t"
        "global_load_dwordx4 v[208:211], %[oA], %[acts]\n\t"
        "global_load_dwordx4 v[212:215], %[oA], %[acts] offset:512\n\t"
        "v_add_u32 %[x4], %[oP], %[sP]\n\t"
        "v_add_u32 %[x5], %[oA], %[sA]\n\t"
        "global_load_ubyte %[ptQ], %[x4], %[pat]\n\t"
        "global_load_dwordx4 v[216:219], %[x5], %[acts]\n\t"
        "global_load_dwordx4 v[220:223], %[x5], %[acts] offset:512\n\t"
        "ds_read_b128 v[176:179], %[wl]\n\t"
        "ds_read_b128 v[180:183], %[wl] offset:1024\n\t"
        "ds_read_b128 v[184:187], %[wl] offset:2048\n\t"
        "ds_read_b128 v[188:191], %[wl] offset:3072\n\t"
        "ds_read_b128 v[192:195], %[wl] offset:4096\n\t"
        "ds_read_b128 v[196:199], %[wl] offset:5120\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        "1:\n\t"
        S2W_STEP_P
        S2W_STEP_Q
        "s_branch 1b\n\t"
        "9:\n\t"
        "s_waitcnt vmcnt(0)\n\t"               // (nothing the compiler does not know of may be in flight when the statement ends)
        : [cstA] "+v"(cstA), [cstB] "+v"(cstB), [oA] "+v"(oA), [oC] "+v"(oC), [oY] "+v"(oY), [oP] "+v"(oP), [cnt] "+s"(cnt),
          [ptP] "=&v"(ptP), [ptQ] "=&v"(ptQ), [a00] "=&v"(a00), [a01] "=&v"(a01), [a10] "=&v"(a10), [a11] "=&v"(a11),
          [a20] "=&v"(a20), [a21] "=&v"(a21), [a30] "=&v"(a30), [a31] "=&v"(a31), [x4] "=&v"(x4), [x5] "=&v"(x5)
#ifdef CN_S2W_STAMP
          , [st0] "+s"(st[0]), [st1] "+s"(st[1]), [st2] "+s"(st[2]), [st3] "+s"(st[3]), [st4] "+s"(st[4]), [st5] "+s"(st[5]),
          [tq] "=&s"(tq), [tl] "+s"(tl), "={s[98:99]}"(tm)
#endif
        : S2W_W_OPERANDS,
          [spidx] "v"(spidx), [av0] "v"(av0), [av1] "v"(av1), [wl] "v"(wl), [oTA] "v"(oTA), [oTB] "v"(oTB),
          [piA] "v"(piA), [pfA] "v"(pfA), [poA] "v"(poA), [piB] "v"(piB), [pfB] "v"(pfB), [poB] "v"(poB),
          [acts] "s"(acts), [actspf] "s"(actspf), [actspf1] "s"(actspf1), [acts1] "s"(acts1), [cell1] "s"(cell1), [th1] "s"(th1), [yop1] "s"(yop1),
          [pat] "s"(pat), [patpf] "s"(patpf), [sA] "s"(sA), [sC] "s"(sC), [sY] "s"(sY), [sP] "s"(sP)
        : "memory", "vcc", "scc",
          "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191",
          "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207",
          "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223",
          "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239",
          "v248", "v249", "v250", "v251", "v252", "v253");
#ifdef CN_S2W_STAMP
    if (blockIdx.x == 0 && lane == 0) {
        for (int i = 0; i < 6; ++i) cn_s2w_stamp_buf[wave][i] = st[i];
        cn_s2w_stamp_buf[wave][7] = (unsigned)(__builtin_amdgcn_s_memrealtime() - rt0);
    }
#endif
}
#ifdef CN_S2W_STAMP
extern "C" int cn_dbg_read_stamps_s2f(unsigned *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cn_s2w_stamp_buf), sizeof(cn_s2w_stamp_buf)); }
#endif

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
struct BwdStage {
    f32x4 a;        // n, i, f, o of step t
    float e;        // outputErrors of step t
    float cp;       // cell state of prev(t)
    float th;       // tanh(cell state) of step t
};

template <int PREC, int HP>
__global__ __launch_bounds__(HP * 2) void lstm_bwd_s2_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(PREC != P_F32 && HP % 32 == 0, "row-pair products: bf16 operands");
    constexpr bool X3 = PREC == P_X3;
    constexpr int MELT = X3 ? 4 : 2;                 // operand element in memory (delta, W_rec^T)
    // K = 4*HP with k = 4*unit + gate.  A sequence takes four tile rows.  bf16: (K half, parity); the product of K half h uses
    // accumulator h and is read from the rows of that half (cn_lstm.hip "KHS"): e = accA[0] + accA[1] + accB[2] + accB[3].
    // split-bf16: (part, parity) with part 0 = hi, 1 = lo halves of the deltas, rows of the whole K: one MFMA against W_hi and
    // one against W_lo per chunk, e = the sum of the accumulator's four registers (see lstm_fwd_s2_kernel).
    constexpr int KCS = 4 * HP / 64, KCH = X3 ? KCS : KCS / 2;        // chunks in all / per tile row
    constexpr int pitch = lds_pitch(KCH * 64);
    constexpr int DROWS = 9;                         // rows 4*seq + 2*(half | part) + parity, + one row of zeros
    constexpr int plane = DROWS * pitch;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int ug = q >> 1, sq = q & 1;
    const int d = blockIdx.x % p.dirs, s0 = (blockIdx.x / p.dirs) * 2;
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    const long arow = (long)dirs * 4 * HP, crow = (long)dirs * HP;

    for (int i = threadIdx.x * 4; i < 2 * plane; i += blockDim.x * 4) *(unsigned *)(smem + i) = 0u;
    // dummy-slot table: dtab[t][s] = (t >= Tmin && patTypes[t][s0 + s] == NONE) (LstmLayer.cu:224-234 with :949,983)
    unsigned char *dtab = (unsigned char *)smem + 2 * plane;
    for (int i = threadIdx.x; i < 2 * T; i += blockDim.x) {
        const int tt = i >> 1;
        dtab[i] = tt >= p.Tmin && p.pat[(long)tt * PS + s0 + (i & 1)] == 0;
    }

    u32x8 wsp[2][KCS];
    [[maybe_unused]] u32x8 wsl[X3 ? 2 : 1][KCS];
    const char *Wd = (const char *)p.WrecT + (long)d * 4 * HP * HP * MELT;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kc = 0; kc < KCS; ++kc) {
            const long w0 = (long)(32 * wave + 16 * j + c) * 4 * HP + kc * 64 + q * 16;
            if constexpr (X3) sp_load_split((const float *)Wd + w0, wsp[j][kc], wsl[j][kc]);
            else wsp[j][kc] = sp_load_bf16(Wd + w0 * 2);
        }
    const int spidx = sp_index(c);
    const int av0 = (c < 8 ? c : 8) * pitch + q * 16, av1 = (c >= 8 ? c - 8 : 8) * pitch + q * 16;

    const int unit = 32 * wave + 16 * ug + c;
    const float pi = p.peep[(d * 3 + 0) * HP + unit], pf = p.peep[(d * 3 + 1) * HP + unit], po = p.peep[(d * 3 + 2) * HP + unit];
    const int sv = s0 + sq;
    const unsigned oA = (unsigned)(sv * (int)arow + (d * HP + unit) * 4);
    const unsigned oC = (unsigned)(sv * (int)crow + d * HP + unit);
    const unsigned stepA = (unsigned)PS * (unsigned)arow, stepC = (unsigned)PS * (unsigned)crow;
    const int uh = X3 ? unit : unit % (HP / 2), half = X3 ? 0 : unit / (HP / 2);
    const int oT = (4 * sq + 2 * half) * pitch + ((uh >> 4) * 32 + sp_pos(4 * (uh & 15))) * 2;      // (split-bf16: the hi rows; lo two rows on)

    float fgn = 0.f, ecn = 0.f, dign = 0.f, dfgn = 0.f, ccur;
    float sb[4] = {0.f, 0.f, 0.f, 0.f}, spi = 0.f, spf = 0.f, spo = 0.f;

    const int tfirst = d ? 0 : T - 1;
    BwdStage preA, preB;
    auto prefetch = [&](int t, BwdStage &pre) {
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        const int tprev = d ? t + 1 : t - 1;
        const bool hasprev = tprev >= 0 && tprev < T;
        const unsigned bA = (unsigned)t * stepA, bC = (unsigned)t * stepC;
        const unsigned bCp = (unsigned)(hasprev ? tprev : t) * stepC;
        pre.e = at32<float>(p.err, bC + oC);
        pre.a = *(const f32x4 *)&at32<float>(p.acts, bA + oA);
        pre.cp = at32<float>(p.cell, bCp + oC);
        pre.th = at32<float>(p.th, bC + oC);
    };

    auto step = [&](int it, BwdStage &pre) {
        const int t = d ? it : T - 1 - it;
        const char *dcur = smem + (it & 1) * plane;
        char *dnxt = smem + ((it + 1) & 1) * plane;
        const int tprev_ = d ? t + 1 : t - 1;
        const bool hasprev_ = tprev_ >= 0 && tprev_ < T;       // !lastCall, LstmLayer.cu:947,981
        const unsigned bD = (unsigned)t * stepA;

        const unsigned char dmy = dtab[2 * t + sq];
        float e_, c_, th_;
        f32x4 a_;
        asm volatile("v_mov_b32 %0, %1" : "=&v"(e_) : "v"(pre.e));
        asm volatile("v_mov_b32 %0, %1" : "=&v"(c_) : "v"(pre.cp));
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("v_mov_b32 %0, %1" : "=&v"(a_[g]) : "v"(pre.a[g]));
        asm volatile("v_mov_b32 %0, %1" : "=&v"(th_) : "v"(pre.th));
        const float cp = hasprev_ ? c_ : 0.f;
        prefetch(d ? t + 2 : t - 2, pre);

        // BPTT product (LstmLayer.cu:939-942 / :973-976): the four gates contract into one K = 4*HP
        u32x4 a0[KCH], a1[KCH];
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
            a0[kc] = *(const u32x4 *)(dcur + av0 + kc * 64);
            a1[kc] = *(const u32x4 *)(dcur + av1 + kc * 64);
        }
        f32x4 accA = {e_, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};      // err enters as the C operand
        float e;
        if constexpr (X3) {
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) {          // small terms first
                smma16(accA, a0[kc], wsl[0][kc], spidx);
                smma16(accA, a0[kc], wsp[0][kc], spidx);
                smma16(accA, a1[kc], wsl[1][kc], spidx);
                smma16(accA, a1[kc], wsp[1][kc], spidx);
            }
            e = (accA[0] + accA[1]) + (accA[2] + accA[3]);
            KEEP_TUPLE(accA, e);
        } else {
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) {
                smma16(accA, a0[kc], wsp[0][kc], spidx);
                smma16(accB, a0[kc], wsp[0][KCH + kc], spidx);
                smma16(accA, a1[kc], wsp[1][kc], spidx);
                smma16(accB, a1[kc], wsp[1][KCH + kc], spidx);
            }
            e = (accA[0] + accA[1]) + (accB[2] + accB[3]);
            KEEP_TUPLE(accA, e); KEEP_TUPLE(accB, e);
        }

        // ComputeBlockErrorsFn, LstmLayer.cu:236-285, as an explicit operation sequence (no contraction left to the compiler:
        // the hand-written loop below issues exactly these operations and is held bit-equal).  Everything that does not
        // depend on e is formed first -- in the hand-written loop one step early, in the shadow of the LDS hand-off --, and
        // the dummy-slot rule (:224-234: all deltas and the cell state error are 0) enters as a factor m = 0 / 1, so that
        // behind the product only
        //   dog = [m og (1 - og) th] e,   ec = e [m (og (1 - th^2) + po og (1 - og) th)] + m (fg' ec' + pi dig' + pf dfg')
        //   dni = [m ig (1 - ni^2)] ec,   dfg = [m fg (1 - fg) cp] ec,   dig = [m ig (1 - ig) ni] ec      (' = carried from t+1)
        // remain: one multiply or fma each, then the clips.  (ec uses the unclipped dog, :262-263, as the reference does.)
        const bool dummy = dmy != 0;
        const float ni = a_[0], ig = a_[1], fg = a_[2], og = a_[3];
        const float cs = ccur, th = th_;
        float dog, ec, dni, dfg, dig;
        {
#pragma clang fp contract(off)
            const float m = dummy ? 0.f : 1.f;
            const float t2p = __builtin_fmaf(-og, og, og) * th;
            const float vp = og * __builtin_fmaf(-th, th, 1.0f);
            const float w = __builtin_fmaf(po, t2p, vp);
            const float d2p = ig * __builtin_fmaf(-ni, ni, 1.0f);
            const float d3p = __builtin_fmaf(-fg, fg, fg) * cp;              // cp = 0 at lastCall
            const float d4p = __builtin_fmaf(-ig, ig, ig) * ni;
            float car = fgn * ecn;                                          // zero carry at firstCall
            car = __builtin_fmaf(pi, dign, car);
            car = __builtin_fmaf(pf, dfgn, car);
            dog = (t2p * m) * e;
            ec = __builtin_fmaf(e, w * m, car * m);
            dni = (d2p * m) * ec; dfg = (d3p * m) * ec; dig = (d4p * m) * ec;
            fgn = fg * m;
        }
        dni = clip1(dni); dig = clip1(dig); dfg = clip1(dfg); dog = clip1(dog);
        ecn = ec; dign = dig; dfgn = dfg;
        ccur = cp;
        // gradient sums (ComputeWeightUpdateFn bias / peephole cases, :392-408, :440-475)
        sb[0] += dni; sb[1] += dig; sb[2] += dfg; sb[3] += dog;
        spi = __builtin_fmaf(cp, dig, spi); spf = __builtin_fmaf(cp, dfg, spf); spo = __builtin_fmaf(cs, dog, spo);
        if constexpr (X3) {
            const f32x4 dv = {dni, dig, dfg, dog};
            bf16x4 dh, dl;
#pragma unroll
            for (int g = 0; g < 4; ++g) { __bf16 h_, l_; split_bf16(dv[g], h_, l_); dh[g] = h_; dl[g] = l_; }
            *(bf16x2 *)(dnxt + oT) = bf16x2{dh[0], dh[1]};
            *(bf16x2 *)(dnxt + oT + pitch) = bf16x2{dh[2], dh[3]};
            *(bf16x2 *)(dnxt + oT + 2 * pitch) = bf16x2{dl[0], dl[1]};
            *(bf16x2 *)(dnxt + oT + 3 * pitch) = bf16x2{dl[2], dl[3]};
            *(f32x4 *)&at32<float>(p.delta_op, bD + oA) = dv;
        } else {
            const bf16x4 dv = {(__bf16)dni, (__bf16)dig, (__bf16)dfg, (__bf16)dog};
            *(bf16x2 *)(dnxt + oT) = bf16x2{dv[0], dv[1]};
            *(bf16x2 *)(dnxt + oT + pitch) = bf16x2{dv[2], dv[3]};
            *(bf16x4 *)&at32<__bf16>(p.delta_op, bD + oA) = dv;
        }
        lds_barrier();
    };

    ccur = at32<float>(p.cell, (unsigned)tfirst * stepC + oC);
    prefetch(tfirst, preA);
    prefetch(d ? 1 : T - 2, preB);
    lds_barrier();
    if (T >= 2) {
        step(0, preA);
        step(1, preB);
        int it = 2;
        for (; it + 1 < T; it += 2) {
            step(it, preA);
            step(it + 1, preB);
        }
        if (it < T) step(it, preA);
    } else {
        step(0, preA);
    }

    // fold the two sequences of each unit column, then one atomic per (gate, unit) and workgroup
    float v[7] = {sb[0], sb[1], sb[2], sb[3], spi, spf, spo};
#pragma unroll
    for (int i = 0; i < 7; ++i) v[i] += __shfl_xor(v[i], 16);
    if (sq == 0) {
        lstm_grad_sums_out(p, HP, d, unit, v);
    }
}

// ---------------------------------------------------------------------------------------------
// backward, bf16, Hp = 128: the time loop written by hand
// ---------------------------------------------------------------------------------------------
// As lstm_fwd_s2_asm_kernel: layout, operand order and arithmetic of lstm_bwd_s2_kernel<P_BF16, 128> (bit-equal on real
// slots), our own instruction stream, ~100 issued instructions per step against hipcc's ~190 -- and software-pipelined:
// in-kernel stamps (tools/stamps_s2.py) showed a step of one wave per SIMD to be the SUM of its stalls (LDS operands landed
// 170 cycles after the barrier, 16 MFMAs with two fillers per gap 410, the e-dependent chain of dependent VALU instructions
// at 6.6 cycles each 125, LDS write + tail 125), so the work is placed where the wave waits anyway:
//   * "block": everything of ComputeBlockErrorsFn that does not depend on the product -- the activation derivatives, the
//     carried terms, the dummy-slot factor m -- is formed for step t+1 at the END of step t, between the LDS write and the
//     barrier; the MFMA phase carries only one cheap filler per gap (a gap hides one VALU instruction, not two);
//   * the stage registers are therefore dead when their step begins, and the prefetch is issued at the TOP of a step,
//     straight into them (outputErrors into register 0 of the stage's accumulator, the MFMA's C operand, once the sums are
//     read): no copies.  FOUR stages, four steps of distance: what the backward pass reads of a lower layer (gate
//     activations, cell states, tanh(c): 90 MB written a whole forward pass ago) has left the Infinity Cache, and with two
//     steps (0.8 us) in flight every workgroup of such a launch waited for HBM every step -- 121 us against 103 us per
//     300 steps with the same loads served from cache (tools/wgtime_s2.py, diag build CN_S2_DIAG_HOT).  The 128 registers
//     of W_rec^T fragments live in AGPRs (an MFMA reads its B operand from either file), which is what makes room;
//   * every step is the same code: the activation buffers carry CN_GUARD_STEPS steps of mapped zeros on both sides
//     (cn_api.cpp: dalloc_guarded), so the prefetch of the last steps needs no special case; lastCall (LstmLayer.cu:947,981:
//     no cell state behind the last processed step) is a scalar select on the one register that carries c[prev]; the loop
//     body is four steps and may be left after any of them;
//   * the 11 wait states between the last MFMA and the first read of its result are the previous step's gradient sums;
//   * behind the product: three adds, five multiplies / fmas, four clips, two conversions; no selects (the factor m);
//   * both LDS writes of a lane are one ds_write2_b32, its delta_op store one 8-byte store.
// Dummy slots as in the forward loop: the pattern type alone decides (for t < minSeqLength the unused slots of a partial
// fraction carry zero errors, so their deltas are zero either way).
//
// Round 5: ONE view of the operand tile.  The two zero-padded views (one per unit group) cost two LDS reads per K chunk; with
// lanes c >= 8 reading the rows of lanes c - 8 the A operand's rows 8 .. 15 repeat rows 0 .. 7, ONE read serves both unit
// groups' MFMAs, and the MFMA of group j leaves that group's sums for both sequences in all four lane quarters -- in an
// accumulator of its own; a lane keeps the sums of its own group (one select per step).  Four reads instead of eight per step,
// four accumulators instead of two per stage, the same sixteen MFMAs in the same order per accumulator (bit-equal).
// Fixed registers (clobbered): stage k = 0..3 at b = 168 + 20k: v[b : b+3] n,i,f,o; v[b+4 : b+7] / v[b+8 : b+11] the accumulators
// of K half 0 / 1 of unit group 0; v[b+12 : b+15] / v[b+16 : b+19] of unit group 1; v[248:249] the four bf16 deltas of the step.
#ifdef CN_S2_DIAG_NOMFMA
#define S2B_MF(acc, a, w) ""
#else
#define S2B_MF(acc, a, w) "v_smfmac_f32_16x16x64_bf16 " acc ", %[" a "], %[" w "], %[spidx]\n\t"
#endif
#ifdef CN_S2_WGTIME
__device__ unsigned cn_s2_wg_time[256][2];        // diag: loop time of every workgroup of the last backward launch (100 MHz ticks), HW_ID
extern "C" int cn_dbg_read_wgtime_s2(unsigned *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cn_s2_wg_time), sizeof(cn_s2_wg_time)); }
#endif
// CN_S2_STAMP (tools/stamps_s2.py; never in the shipped build): every wave of workgroup 0 sums s_memtime deltas per step segment
#ifdef CN_S2_STAMP
__device__ unsigned cn_s2_stamp_buf[4][8];
#define S2B_ST(i) "s_memtime s[98:99]\n\ts_waitcnt lgkmcnt(0)\n\ts_sub_u32 %[tq], s98, %[tl]\n\ts_add_u32 %[st" #i "], %[st" #i "], %[tq]\n\ts_mov_b32 %[tl], s98\n\t"
#else
#define S2B_ST(i) ""
#endif
// prefetch of step t+4 into stage X: activations, tanh(c), c[prev], pattern type (outputErrors follow behind the e sum)
#ifdef CN_S2_DIAG_HOT
#define S2B_PF(AXT, TH, CP, PT) \
    "global_load_dwordx4 " AXT ", %[oT], %[acts]\n\t" \
    "global_load_dword %[" TH "], %[oT], %[th]\n\t" \
    "global_load_dword %[" CP "], %[oT], %[cell]\n\t" \
    "global_load_ubyte %[" PT "], %[oT], %[pat]\n\t"
#define S2B_PFE(ACCA0) "global_load_dword " ACCA0 ", %[oT], %[err]\n\t"
#else
#define S2B_PF(AXT, TH, CP, PT) \
    "global_load_dwordx4 " AXT ", %[oA], %[actspf]\n\t" \
    "global_load_dword %[" TH "], %[oC], %[thpf]\n\t" \
    "global_load_dword %[" CP "], %[oC], %[cellpf]\n\t" \
    "global_load_ubyte %[" PT "], %[oP], %[patpf]\n\t"
#define S2B_PFE(ACCA0) "global_load_dword " ACCA0 ", %[oC], %[errpf]\n\t"
#endif
// the e-independent terms of the step that stage X holds; CN: register that receives c[prev] of that step (the CS of the
// step after it), or 0 when that step is the last one (s[last] = all ones then).  Reads the carries of the step before it.
#define S2B_BLOCK(NI, IG, FG, OG, TH, CP, PT, CN) \
    "v_cmp_eq_u32 vcc, 0, %[" PT "]\n\t" \
    "v_fma_f32 %[x0], -" OG ", " OG ", " OG "\n\t" \
    "v_fma_f32 %[x1], -%[" TH "], %[" TH "], 1.0\n\t" \
    "v_cndmask_b32_e64 %[" CN "], %[" CP "], 0, %[last]\n\t" \
    "v_cndmask_b32_e64 %[m], 1.0, 0, vcc\n\t" \
    "v_mul_f32 %[t2m], %[x0], %[" TH "]\n\t" \
    "v_mul_f32 %[x1], " OG ", %[x1]\n\t" \
    "v_fma_f32 %[x0], -" NI ", " NI ", 1.0\n\t" \
    "v_mul_f32 %[car], %[fgn], %[ecn]\n\t" \
    "v_fma_f32 %[wm], %[po], %[t2m], %[x1]\n\t" \
    "v_mul_f32 %[d2m], " IG ", %[x0]\n\t" \
    "v_fma_f32 %[x0], -" FG ", " FG ", " FG "\n\t" \
    "v_fmac_f32 %[car], %[pi], %[dign]\n\t" \
    "v_fma_f32 %[x1], -" IG ", " IG ", " IG "\n\t" \
    "v_mul_f32 %[d3m], %[x0], %[" CN "]\n\t" \
    "v_fmac_f32 %[car], %[pf], %[dfgn]\n\t" \
    "v_mul_f32 %[d4m], %[x1], " NI "\n\t" \
    "v_mul_f32 %[fgn], " FG ", %[m]\n\t"
// ... and their products with the dummy-slot factor m: one per MFMA gap of the step itself (S2B_STEP), where a VALU
// instruction costs ~3.5 cycles instead of the 6.6 of the dependent stream behind the LDS write
#define S2B_MASK(x) "v_mul_f32 %[" x "], %[" x "], %[m]\n\t"
// ACCA / ACCB: the stage's accumulators (A0, A1 / B2, B3: the registers that are read); CS: register holding the cell state
// of this step; R: LDS byte offset of the tile read; WT: operand holding the lane's address in the tile written (the two
// dword offsets of ds_write2_b32 are 8-bit fields: its second row, one pitch = 72 dwords on, fits; the tile base does not);
// PFCODE / PFECODE: the prefetch into this stage; BLOCKCODE: the block of the NEXT stage (behind a wait for its loads)
#define S2B_STEP(ACCA, A0, A1, ACCB, B2, B3, ACCC, C0, C1, ACCD, D2, D3, CS, R, WT, PFCODE, PFECODE, BLOCKCODE) \
    S2B_ST(0) \
    "ds_read_b128 %[r00], %[av0] offset:" R "\n\t" \
    "ds_read_b128 %[r01], %[av0] offset:" R "+64\n\t" \
    "ds_read_b128 %[r02], %[av0] offset:" R "+128\n\t" \
    "ds_read_b128 %[r03], %[av0] offset:" R "+192\n\t" \
    "v_mov_b32 " C0 ", " A0 "\n\t" \
    "v_mov_b32 " A1 ", 0\n\t" \
    "v_mov_b32 " B2 ", 0\n\t" \
    "v_mov_b32 " B3 ", 0\n\t" \
    "v_mov_b32 " C1 ", 0\n\t" \
    "v_mov_b32 " D2 ", 0\n\t" \
    "v_mov_b32 " D3 ", 0\n\t" \
    PFCODE \
    "s_waitcnt lgkmcnt(3)\n\t" \
    S2B_ST(1) \
    S2B_MF(ACCA, "r00", "w0k0") \
    "v_add_u32 %[oA], %[oA], %[sA]\n\t" \
    S2B_MF(ACCB, "r00", "w0k4") \
    "v_add_u32 %[oC], %[oC], %[sC]\n\t" \
    S2B_MF(ACCC, "r00", "w1k0") \
    "v_add_u32 %[oD], %[oD], %[sD]\n\t" \
    S2B_MF(ACCD, "r00", "w1k4") \
    "v_add_u32 %[oP], %[oP], %[sP]\n\t" \
    "s_waitcnt lgkmcnt(2)\n\t" \
    S2B_MF(ACCA, "r01", "w0k1") \
    S2B_MASK("t2m") \
    S2B_MF(ACCB, "r01", "w0k5") \
    S2B_MASK("wm") \
    S2B_MF(ACCC, "r01", "w1k1") \
    "v_mul_f32 %[carm], %[car], %[m]\n\t" \
    S2B_MF(ACCD, "r01", "w1k5") \
    S2B_MASK("d2m") \
    "s_waitcnt lgkmcnt(1)\n\t" \
    S2B_MF(ACCA, "r02", "w0k2") \
    S2B_MASK("d3m") \
    S2B_MF(ACCB, "r02", "w0k6") \
    S2B_MASK("d4m") \
    S2B_MF(ACCC, "r02", "w1k2") \
    S2B_MF(ACCD, "r02", "w1k6") \
    "s_waitcnt lgkmcnt(0)\n\t" \
    S2B_MF(ACCA, "r03", "w0k3") \
    S2B_MF(ACCB, "r03", "w0k7") \
    S2B_MF(ACCC, "r03", "w1k3") \
    S2B_MF(ACCD, "r03", "w1k7") \
    S2B_ST(2) \
    "v_add_f32 %[sb0], %[sb0], %[dni]\n\t" \
    "v_add_f32 %[sb1], %[sb1], %[dign]\n\t" \
    "v_add_f32 %[sb2], %[sb2], %[dfgn]\n\t" \
    "v_add_f32 %[sb3], %[sb3], %[dog]\n\t" \
    "v_fmac_f32 %[spi], %[" CS "], %[dign]\n\t" \
    "v_fmac_f32 %[spf], %[" CS "], %[dfgn]\n\t" \
    "s_cmp_eq_u32 %[cnt], 1\n\t" \
    "s_cselect_b64 %[last], -1, 0\n\t" \
    "s_nop 2\n\t" \
    "v_add_f32 %[x0], " A0 ", " A1 "\n\t" \
    "v_add_f32 %[x1], " B2 ", " B3 "\n\t" \
    "v_add_f32 %[car], " C0 ", " C1 "\n\t" \
    "v_add_f32 %[x0], %[x0], %[x1]\n\t" \
    "v_add_f32 %[x1], " D2 ", " D3 "\n\t" \
    "v_add_f32 %[car], %[car], %[x1]\n\t" \
    "v_cndmask_b32_e64 %[x0], %[x0], %[car], %[ugm]\n\t" \
    PFECODE \
    S2B_ST(3) \
    "v_mul_f32 %[dog], %[t2m], %[x0]\n\t" \
    "v_fma_f32 %[ecn], %[x0], %[wm], %[carm]\n\t" \
    "v_med3_f32 %[dog], %[dog], -1.0, 1.0\n\t" \
    "v_mul_f32 %[dni], %[d2m], %[ecn]\n\t" \
    "v_mul_f32 %[dfgn], %[d3m], %[ecn]\n\t" \
    "v_mul_f32 %[dign], %[d4m], %[ecn]\n\t" \
    "v_med3_f32 %[dni], %[dni], -1.0, 1.0\n\t" \
    "v_med3_f32 %[dfgn], %[dfgn], -1.0, 1.0\n\t" \
    "v_med3_f32 %[dign], %[dign], -1.0, 1.0\n\t" \
    "v_cvt_pk_bf16_f32 v249, %[dfgn], %[dog]\n\t" \
    "v_cvt_pk_bf16_f32 v248, %[dni], %[dign]\n\t" \
    S2B_ST(4) \
    "ds_write2_b32 %[" WT "], v248, v249 offset0:0 offset1:72\n\t" \
    "global_store_dwordx2 %[oD], v[248:249], %[delta1]\n\t" \
    "v_fmac_f32 %[spo], %[" CS "], %[dog]\n\t" \
    "s_waitcnt vmcnt(19)\n\t" \
    BLOCKCODE \
    "s_waitcnt lgkmcnt(0)\n\t" \
    S2B_ST(5) \
    "s_barrier\n\t" \
    S2B_ST(6) \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t" \
    "s_cbranch_scc1 9f\n\t"
// tiles: plane = 9 * 288 = 2592 bytes; tile 0 at 0, tile 1 at 2592.  Stage k: registers 200 + 12 k ...; cell-state registers
// alternate with the step parity (ccA holds the cell state of even steps)
#define S2B_STEP_0 S2B_STEP("v[172:175]", "v172", "v173", "v[176:179]", "v178", "v179", "v[180:183]", "v180", "v181", "v[184:187]", "v186", "v187", "ccA", "0", "oT1", S2B_PF("v[168:171]", "th0", "cp0", "pt0"), S2B_PFE("v172"), \
                            S2B_BLOCK("v188", "v189", "v190", "v191", "th1", "cp1", "pt1", "ccA"))
#define S2B_STEP_1 S2B_STEP("v[192:195]", "v192", "v193", "v[196:199]", "v198", "v199", "v[200:203]", "v200", "v201", "v[204:207]", "v206", "v207", "ccB", "2592", "oT", S2B_PF("v[188:191]", "th1", "cp1", "pt1"), S2B_PFE("v192"), \
                            S2B_BLOCK("v208", "v209", "v210", "v211", "th2", "cp2", "pt2", "ccB"))
#define S2B_STEP_2 S2B_STEP("v[212:215]", "v212", "v213", "v[216:219]", "v218", "v219", "v[220:223]", "v220", "v221", "v[224:227]", "v226", "v227", "ccA", "0", "oT1", S2B_PF("v[208:211]", "th2", "cp2", "pt2"), S2B_PFE("v212"), \
                            S2B_BLOCK("v228", "v229", "v230", "v231", "th3", "cp3", "pt3", "ccA"))
#define S2B_STEP_3 S2B_STEP("v[232:235]", "v232", "v233", "v[236:239]", "v238", "v239", "v[240:243]", "v240", "v241", "v[244:247]", "v246", "v247", "ccB", "2592", "oT", S2B_PF("v[228:231]", "th3", "cp3", "pt3"), S2B_PFE("v232"), \
                            S2B_BLOCK("v168", "v169", "v170", "v171", "th0", "cp0", "pt0", "ccB"))
// the stages of the first four steps: step k at offsets advanced by k steps (x0 / x1 / m: scratch offsets)
#define S2B_FIRST(AXT, A0, TH, CP, PT) \
    "global_load_dwordx4 " AXT ", %[x0], %[acts]\n\t" \
    "global_load_dword %[" TH "], %[x1], %[th]\n\t" \
    "global_load_dword %[" CP "], %[x1], %[cell1]\n\t" \
    "global_load_ubyte %[" PT "], %[m], %[pat]\n\t" \
    "global_load_dword " A0 ", %[x1], %[err]\n\t" \
    "v_add_u32 %[x0], %[x0], %[sA]\n\t" \
    "v_add_u32 %[x1], %[x1], %[sC]\n\t" \
    "v_add_u32 %[m], %[m], %[sP]\n\t"

__global__ __launch_bounds__(256) void lstm_bwd_s2_asm_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HP = 128, KCS = 8, KCH = 4;
    constexpr int pitch = lds_pitch(KCH * 64);       // 288
    constexpr int plane = 9 * pitch;                 // 2592: the asm carries it (and pitch / 4 = 72) as literals
    static_assert(pitch == 288 && plane == 2592, "LDS offsets of the hand-written loop");
    static_assert(CN_GUARD_STEPS >= 5, "prefetch four steps ahead, c[prev] five");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int ug = q >> 1, sq = q & 1;
    const int d = blockIdx.x % p.dirs, s0 = (blockIdx.x / p.dirs) * 2;
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    const long arow = (long)dirs * 4 * HP, crow = (long)dirs * HP;

    for (int i = threadIdx.x * 4; i < 2 * plane; i += blockDim.x * 4) *(unsigned *)(smem + i) = 0u;

    u32x8 w[2][KCS];
    const char *Wd = (const char *)p.WrecT + (long)d * 4 * HP * HP * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kc = 0; kc < KCS; ++kc)
            w[j][kc] = sp_load_bf16(Wd + ((long)(32 * wave + 16 * j + c) * 4 * HP + kc * 64 + q * 16) * 2);
    const int spidx = sp_index(c);
    // ONE view of the tile: lanes c >= 8 read the rows of lanes c - 8; a lane keeps the sums of its own unit group (ugm: group 1's lanes)
    const unsigned av0 = (c & 7) * pitch + q * 16;
    const unsigned long long ugm = 0xFFFFFFFF00000000ull;

    const int unit = 32 * wave + 16 * ug + c;
    const float pi = p.peep[(d * 3 + 0) * HP + unit], pf = p.peep[(d * 3 + 1) * HP + unit], po = p.peep[(d * 3 + 2) * HP + unit];
    const int sv = s0 + sq;
    const int uh = unit % (HP / 2), half = unit / (HP / 2);
    const unsigned oT = (4 * sq + 2 * half) * pitch + ((uh >> 4) * 32 + sp_pos(4 * (uh & 15))) * 2, oT1 = oT + plane;
    // Byte offsets of this lane and what one step adds (mod 2^32).  The offsets are kept BIAS steps ahead of the time index
    // and every base pointer BIAS steps behind, so that no offset ever passes zero: the loop moves them on under the first
    // MFMAs of a step, behind its prefetch, also in the last step (a negative offset would be 4 GB up).
    constexpr long BIAS = 8;
    const long t0 = d ? 0 : T - 1, dt = d ? 1 : -1;
    const long stepA = (long)PS * arow, stepC = (long)PS * crow;
    const unsigned lC = (unsigned)(sv * (int)crow + d * HP + unit);
    unsigned oA = (unsigned)((t0 + BIAS) * stepA * 4) + lC * 16, oC = (unsigned)((t0 + BIAS) * stepC * 4) + lC * 4;
    unsigned oD = (unsigned)((t0 + BIAS) * stepA * 2) + lC * 8, oP = (unsigned)((t0 + BIAS) * PS) + (unsigned)sv;
    const unsigned sA = (unsigned)(dt * stepA * 4), sC = (unsigned)(dt * stepC * 4), sD = (unsigned)(dt * stepA * 2), sP = (unsigned)(dt * PS);
    const char *acts = (const char *)p.acts - BIAS * stepA * 4, *cell = (const char *)p.cell - BIAS * stepC * 4, *th = (const char *)p.th - BIAS * stepC * 4;
    const char *err = (const char *)p.err - BIAS * stepC * 4, *pat = p.pat - BIAS * PS;
    // bases of the prefetch four steps ahead; c[prev(t)] is the cell state of the step processed after t: five steps ahead.
    // outputErrors and the delta_op store are issued behind the offsets' move to the next step: their bases are one step back.
    const char *actspf = acts + 4 * dt * stepA * 4, *thpf = th + 4 * dt * stepC * 4, *errpf = err + (4 - 1) * dt * stepC * 4;
    const char *cell1 = cell + dt * stepC * 4, *cellpf = cell + 5 * dt * stepC * 4, *patpf = pat + 4 * dt * PS;
    const char *delta1 = (const char *)p.delta_op - BIAS * stepA * 2 - dt * stepA * 2;
    unsigned cnt = (unsigned)T - 1;                  // steps behind the current one

    float fgn = 0.f, ecn = 0.f, dign = 0.f, dfgn = 0.f, dni = 0.f, dog = 0.f;
    float sb0 = 0.f, sb1 = 0.f, sb2 = 0.f, sb3 = 0.f, spi = 0.f, spf = 0.f, spo = 0.f;
    float ccA, ccB, th0, th1, th2, th3, cp0, cp1, cp2, cp3;
    int pt0, pt1, pt2, pt3;
    u32x4 r00, r01, r02, r03;
    float x0, x1, m, t2m, wm, carm, d2m, d3m, d4m, car;
    unsigned long long last;
#ifdef CN_S2_STAMP
    unsigned st[7] = {0, 0, 0, 0, 0, 0, 0}, tq;
    unsigned long long tm;
    unsigned tl = (unsigned)__builtin_amdgcn_s_memtime();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef CN_S2_WGTIME
    const unsigned long long wg0 = __builtin_amdgcn_s_memrealtime();
#endif
    lds_barrier();
    asm volatile(
        // (registers 2, 3 of the K-half-0 accumulators and 0, 1 of the K-half-1 accumulators -- of either unit group -- collect products
        // of rows that belong to the other half; they are never read -- cleared once so that they hold numbers)
        "v_mov_b32 v174, 0\n\tv_mov_b32 v175, 0\n\tv_mov_b32 v176, 0\n\tv_mov_b32 v177, 0\n\tv_mov_b32 v182, 0\n\tv_mov_b32 v183, 0\n\tv_mov_b32 v184, 0\n\tv_mov_b32 v185, 0\n\t"
        "v_mov_b32 v194, 0\n\tv_mov_b32 v195, 0\n\tv_mov_b32 v196, 0\n\tv_mov_b32 v197, 0\n\tv_mov_b32 v202, 0\n\tv_mov_b32 v203, 0\n\tv_mov_b32 v204, 0\n\tv_mov_b32 v205, 0\n\t"
        "v_mov_b32 v214, 0\n\tv_mov_b32 v215, 0\n\tv_mov_b32 v216, 0\n\tv_mov_b32 v217, 0\n\tv_mov_b32 v222, 0\n\tv_mov_b32 v223, 0\n\tv_mov_b32 v224, 0\n\tv_mov_b32 v225, 0\n\t"
        "v_mov_b32 v234, 0\n\tv_mov_b32 v235, 0\n\tv_mov_b32 v236, 0\n\tv_mov_b32 v237, 0\n\tv_mov_b32 v242, 0\n\tv_mov_b32 v243, 0\n\tv_mov_b32 v244, 0\n\tv_mov_b32 v245, 0\n\t"
        // cell state of the first processed step; the stages of the first four steps (beyond T: guard steps, never used)
        "global_load_dword %[ccA], %[oC], %[cell]\n\t"
        "v_mov_b32 %[x0], %[oA]\n\tv_mov_b32 %[x1], %[oC]\n\tv_mov_b32 %[m], %[oP]\n\t"
        S2B_FIRST("v[168:171]", "v172", "th0", "cp0", "pt0")
        S2B_FIRST("v[188:191]", "v192", "th1", "cp1", "pt1")
        S2B_FIRST("v[208:211]", "v212", "th2", "cp2", "pt2")
        S2B_FIRST("v[228:231]", "v232", "th3", "cp3", "pt3")
        "s_waitcnt vmcnt(0)\n\t"
        "s_cmp_eq_u32 %[cnt], 0\n\t"
        "s_cselect_b64 %[last], -1, 0\n\t"
        "s_nop 1\n\t"
        S2B_BLOCK("v168", "v169", "v170", "v171", "th0", "cp0", "pt0", "ccB")
        "1:\n\t"
        S2B_STEP_0
        S2B_STEP_1
        S2B_STEP_2
        S2B_STEP_3
        "s_branch 1b\n\t"
        "9:\n\t"
        // the prefetches of the last steps are still in flight; hipcc counts an asm load's destination as written when the
        // statement ends and reuses those registers in the epilogue (the atomics' addresses -- seen as an intermittent memory
        // fault of the first, cold backward pass beside the gradient GEMMs): drain before leaving
        "s_waitcnt vmcnt(0)\n\t"
        // gradient sums of the last step (its c[prev] is 0: lastCall)
        "v_add_f32 %[sb0], %[sb0], %[dni]\n\tv_add_f32 %[sb1], %[sb1], %[dign]\n\tv_add_f32 %[sb2], %[sb2], %[dfgn]\n\tv_add_f32 %[sb3], %[sb3], %[dog]\n\t"
        : [fgn] "+v"(fgn), [ecn] "+v"(ecn), [dign] "+v"(dign), [dfgn] "+v"(dfgn), [dni] "+v"(dni), [dog] "+v"(dog),
          [sb0] "+v"(sb0), [sb1] "+v"(sb1), [sb2] "+v"(sb2), [sb3] "+v"(sb3), [spi] "+v"(spi), [spf] "+v"(spf), [spo] "+v"(spo),
          [oA] "+v"(oA), [oC] "+v"(oC), [oD] "+v"(oD), [oP] "+v"(oP), [cnt] "+s"(cnt), [last] "=&s"(last),
          [ccA] "=&v"(ccA), [ccB] "=&v"(ccB), [th0] "=&v"(th0), [th1] "=&v"(th1), [th2] "=&v"(th2), [th3] "=&v"(th3),
          [cp0] "=&v"(cp0), [cp1] "=&v"(cp1), [cp2] "=&v"(cp2), [cp3] "=&v"(cp3),
          [pt0] "=&v"(pt0), [pt1] "=&v"(pt1), [pt2] "=&v"(pt2), [pt3] "=&v"(pt3),
          [r00] "=&v"(r00), [r01] "=&v"(r01), [r02] "=&v"(r02), [r03] "=&v"(r03),
          [x0] "=&v"(x0), [x1] "=&v"(x1), [m] "=&v"(m), [t2m] "=&v"(t2m), [wm] "=&v"(wm), [carm] "=&v"(carm),
          [d2m] "=&v"(d2m), [d3m] "=&v"(d3m), [d4m] "=&v"(d4m), [car] "=&v"(car)
#ifdef CN_S2_STAMP
          , [st0] "+s"(st[0]), [st1] "+s"(st[1]), [st2] "+s"(st[2]), [st3] "+s"(st[3]), [st4] "+s"(st[4]), [st5] "+s"(st[5]), [st6] "+s"(st[6]),
          [tq] "=&s"(tq), [tl] "+s"(tl), "={s[98:99]}"(tm)
#endif
        : [w0k0] "a"(w[0][0]), [w0k1] "a"(w[0][1]), [w0k2] "a"(w[0][2]), [w0k3] "a"(w[0][3]), [w0k4] "a"(w[0][4]), [w0k5] "a"(w[0][5]), [w0k6] "a"(w[0][6]), [w0k7] "a"(w[0][7]),
          [w1k0] "a"(w[1][0]), [w1k1] "a"(w[1][1]), [w1k2] "a"(w[1][2]), [w1k3] "a"(w[1][3]), [w1k4] "a"(w[1][4]), [w1k5] "a"(w[1][5]), [w1k6] "a"(w[1][6]), [w1k7] "a"(w[1][7]),
          [spidx] "v"(spidx), [av0] "v"(av0), [ugm] "s"(ugm), [oT] "v"(oT), [oT1] "v"(oT1), [pi] "v"(pi), [pf] "v"(pf), [po] "v"(po),
          [acts] "s"(acts), [actspf] "s"(actspf), [cell] "s"(cell), [cell1] "s"(cell1), [cellpf] "s"(cellpf), [th] "s"(th), [thpf] "s"(thpf),
          [err] "s"(err), [errpf] "s"(errpf), [pat] "s"(pat), [patpf] "s"(patpf), [delta1] "s"(delta1),
          [sA] "s"(sA), [sC] "s"(sC), [sD] "s"(sD), [sP] "s"(sP)
        : "memory", "vcc", "scc",
          "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249");

#ifdef CN_S2_STAMP
    if (blockIdx.x == 0 && lane == 0) {
        for (int i = 0; i < 7; ++i) cn_s2_stamp_buf[wave][i] = st[i];
        cn_s2_stamp_buf[wave][7] = (unsigned)(__builtin_amdgcn_s_memrealtime() - rt0);      // 100 MHz ticks over the loop
    }
#endif
#ifdef CN_S2_WGTIME
    if (threadIdx.x == 0 && blockIdx.x < 256) { cn_s2_wg_time[blockIdx.x][0] = (unsigned)(__builtin_amdgcn_s_memrealtime() - wg0); cn_s2_wg_time[blockIdx.x][1] = (__builtin_amdgcn_s_getreg(63492) & 0xffff) | (__builtin_amdgcn_s_getreg(63508) << 16) /* HW_ID | XCC_ID << 16 */; }
#endif
    // fold the two sequences of each unit column, then one atomic per (gate, unit) and workgroup
    float v[7] = {sb0, sb1, sb2, sb3, spi, spf, spo};
#pragma unroll
    for (int i = 0; i < 7; ++i) v[i] += __shfl_xor(v[i], 16);
    if (sq == 0) {
        lstm_grad_sums_out(p, HP, d, unit, v);
    }
}

#ifdef CN_S2_STAMP
extern "C" int cn_dbg_read_stamps_s2(unsigned *host)      // [4][8]
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cn_s2_stamp_buf), sizeof(cn_s2_stamp_buf));
}
#endif

// ---------------------------------------------------------------------------------------------
// backward, split-bf16 (P_X3), Hp = 128: the time loop written by hand
// ---------------------------------------------------------------------------------------------
// lstm_bwd_s2_asm_kernel's structure (four stages, block of step t+1 behind the LDS write of step t, uniform steps over the
// guard steps, loop left after any step) with the operands of lstm_bwd_s2_kernel<P_X3, 128>: fp32 deltas in memory, the tile
// rows of a sequence's quad = hi / lo halves over the whole K = 512 (8 chunks per row), 32 MFMAs per step ([hi; lo] x W_lo,
// [hi; lo] x W_hi per chunk and view) into ONE accumulator whose four registers are summed; W_rec^T hi and lo fragments: 256
// AGPRs.  Bit-equal to the compiled kernel on real slots.  LDS: 9 rows of 544 bytes per tile buffer, buffers at 0 and 4896.
// Fixed registers: stage k = 0..3: v[200+8k : 203+8k] n,i,f,o, v[204+8k : 207+8k] the accumulator; v[232:235] the four fp32
// deltas of the step (n, i, f, o: one 16-byte store; i and f double as the carried deltas).
#define X3B_MF(acc, a, w) "v_smfmac_f32_16x16x64_bf16 " acc ", %[" a "], %[" w "], %[spidx]\n\t"
#ifdef CN_X3B_NOSTORE
#define X3B_STORE ""
#else
#define X3B_STORE "global_store_dwordx4 %[oA], v[232:235], %[delta1]\n\t"
#endif
#define X3B_MF4(acc, KC, F0, F1) \
    X3B_MF(acc, "r0" KC, "l0k" KC) X3B_MF(acc, "r0" KC, "h0k" KC) F0 \
    X3B_MF(acc, "r1" KC, "l1k" KC) X3B_MF(acc, "r1" KC, "h1k" KC) F1
#ifdef CN_X3B_NOPF
#define X3B_PF(AXT, TH, CP, PT) ""
#else
#define X3B_PF(AXT, TH, CP, PT) \
    "global_load_dwordx4 " AXT ", %[oA], %[actspf]\n\t" \
    "global_load_dword %[" TH "], %[oC], %[thpf]\n\t" \
    "global_load_dword %[" CP "], %[oC], %[cellpf]\n\t" \
    "global_load_ubyte %[" PT "], %[oP], %[patpf]\n\t"
#endif
#define X3B_BLOCK(NI, IG, FG, OG, TH, CP, PT, CN) \
    "v_cmp_eq_u32 vcc, 0, %[" PT "]\n\t" \
    "v_fma_f32 %[x0], -" OG ", " OG ", " OG "\n\t" \
    "v_fma_f32 %[x1], -%[" TH "], %[" TH "], 1.0\n\t" \
    "v_cndmask_b32_e64 %[" CN "], %[" CP "], 0, %[last]\n\t" \
    "v_cndmask_b32_e64 %[m], 1.0, 0, vcc\n\t" \
    "v_mul_f32 %[t2m], %[x0], %[" TH "]\n\t" \
    "v_mul_f32 %[x1], " OG ", %[x1]\n\t" \
    "v_fma_f32 %[x0], -" NI ", " NI ", 1.0\n\t" \
    "v_mul_f32 %[car], %[fgn], %[ecn]\n\t" \
    "v_fma_f32 %[wm], %[po], %[t2m], %[x1]\n\t" \
    "v_mul_f32 %[d2m], " IG ", %[x0]\n\t" \
    "v_fma_f32 %[x0], -" FG ", " FG ", " FG "\n\t" \
    "v_fmac_f32 %[car], %[pi], v233\n\t" \
    "v_fma_f32 %[x1], -" IG ", " IG ", " IG "\n\t" \
    "v_mul_f32 %[d3m], %[x0], %[" CN "]\n\t" \
    "v_fmac_f32 %[car], %[pf], v234\n\t" \
    "v_mul_f32 %[d4m], %[x1], " NI "\n\t" \
    "v_mul_f32 %[fgn], " FG ", %[m]\n\t"
#define X3B_MASK(x) "v_mul_f32 %[" x "], %[" x "], %[m]\n\t"
#define X3B_RD(KC, OFF) "ds_read_b128 %[r0" KC "], %[av0] offset:" OFF "\n\tds_read_b128 %[r1" KC "], %[av1] offset:" OFF "\n\t"
// ACC: the stage's accumulator (A0..A3 its registers); CS: cell state of this step; R0..R7: LDS byte offsets of the eight K
// chunks of the tile read; WT / WL: operands with the lane's address of its hi / lo rows in the tile written
#define X3B_STEP(ACC, A0, A1, A2, A3, CS, R0, R1, R2, R3, R4, R5, R6, R7, WT, WL, PFCODE, PFECODE, BLOCKCODE) \
    X3B_RD("0", R0) X3B_RD("1", R1) X3B_RD("2", R2) X3B_RD("3", R3) X3B_RD("4", R4) X3B_RD("5", R5) X3B_RD("6", R6) X3B_RD("7", R7) \
    "v_mov_b32 " A1 ", 0\n\t" \
    "v_mov_b32 " A2 ", 0\n\t" \
    "v_mov_b32 " A3 ", 0\n\t" \
    PFCODE \
    "s_waitcnt lgkmcnt(14)\n\t" \
    X3B_MF4(ACC, "0", "v_add_u32 %[oA], %[oA], %[sA]\n\t", "v_add_u32 %[oC], %[oC], %[sC]\n\t") \
    "s_waitcnt lgkmcnt(12)\n\t" \
    X3B_MF4(ACC, "1", "v_add_u32 %[oP], %[oP], %[sP]\n\t", X3B_MASK("t2m")) \
    "s_waitcnt lgkmcnt(10)\n\t" \
    X3B_MF4(ACC, "2", X3B_MASK("wm"), "v_mul_f32 %[carm], %[car], %[m]\n\t") \
    "s_waitcnt lgkmcnt(8)\n\t" \
    X3B_MF4(ACC, "3", X3B_MASK("d2m"), X3B_MASK("d3m")) \
    "s_waitcnt lgkmcnt(6)\n\t" \
    X3B_MF4(ACC, "4", X3B_MASK("d4m"), "") \
    "s_waitcnt lgkmcnt(4)\n\t" \
    X3B_MF4(ACC, "5", "", "") \
    "s_waitcnt lgkmcnt(2)\n\t" \
    X3B_MF4(ACC, "6", "", "") \
    "s_waitcnt lgkmcnt(0)\n\t" \
    X3B_MF4(ACC, "7", "", "") \
    "v_add_f32 %[sb0], %[sb0], v232\n\t" \
    "v_add_f32 %[sb1], %[sb1], v233\n\t" \
    "v_add_f32 %[sb2], %[sb2], v234\n\t" \
    "v_add_f32 %[sb3], %[sb3], v235\n\t" \
    "v_fmac_f32 %[spi], %[" CS "], v233\n\t" \
    "v_fmac_f32 %[spf], %[" CS "], v234\n\t" \
    "s_cmp_eq_u32 %[cnt], 1\n\t" \
    "s_cselect_b64 %[last], -1, 0\n\t" \
    "s_nop 2\n\t" \
    "v_add_f32 %[x0], " A0 ", " A1 "\n\t" \
    "v_add_f32 %[x1], " A2 ", " A3 "\n\t" \
    "v_add_f32 %[x0], %[x0], %[x1]\n\t" \
    PFECODE \
    "v_mul_f32 v235, %[t2m], %[x0]\n\t" \
    "v_fma_f32 %[ecn], %[x0], %[wm], %[carm]\n\t" \
    "v_med3_f32 v235, v235, -1.0, 1.0\n\t" \
    "v_mul_f32 v232, %[d2m], %[ecn]\n\t" \
    "v_mul_f32 v234, %[d3m], %[ecn]\n\t" \
    "v_mul_f32 v233, %[d4m], %[ecn]\n\t" \
    "v_med3_f32 v232, v232, -1.0, 1.0\n\t" \
    "v_med3_f32 v234, v234, -1.0, 1.0\n\t" \
    "v_med3_f32 v233, v233, -1.0, 1.0\n\t" \
    "v_cvt_pk_bf16_f32 %[hb], v234, v235\n\t" \
    "v_cvt_pk_bf16_f32 %[ha], v232, v233\n\t" \
    "v_and_b32 %[x1], 0xffff0000, %[hb]\n\t" \
    "v_lshlrev_b32 %[x0], 16, %[hb]\n\t" \
    "v_sub_f32 %[x1], v235, %[x1]\n\t" \
    "v_sub_f32 %[x0], v234, %[x0]\n\t" \
    "v_cvt_pk_bf16_f32 %[lb], %[x0], %[x1]\n\t" \
    "v_and_b32 %[x1], 0xffff0000, %[ha]\n\t" \
    "v_lshlrev_b32 %[x0], 16, %[ha]\n\t" \
    "v_sub_f32 %[x1], v233, %[x1]\n\t" \
    "v_sub_f32 %[x0], v232, %[x0]\n\t" \
    "v_cvt_pk_bf16_f32 %[la], %[x0], %[x1]\n\t" \
    "ds_write2_b32 %[" WT "], %[ha], %[hb] offset0:0 offset1:136\n\t" \
    "ds_write2_b32 %[" WL "], %[la], %[lb] offset0:0 offset1:136\n\t" \
    X3B_STORE \
    "v_fmac_f32 %[spo], %[" CS "], v235\n\t" \
    "s_waitcnt vmcnt(19)\n\t" \
    BLOCKCODE \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "s_barrier\n\t" \
    "s_sub_u32 %[cnt], %[cnt], 1\n\t" \
    "s_cbranch_scc1 9f\n\t"
#define X3B_APPLY(M, ...) M(__VA_ARGS__)
#define X3B_EVEN "0", "64", "128", "192", "256", "320", "384", "448", "oT1", "oT1l"
#define X3B_ODD "4896", "4960", "5024", "5088", "5152", "5216", "5280", "5344", "oT", "oTl"
#define X3B_STEP_0 X3B_APPLY(X3B_STEP, "v[204:207]", "v204", "v205", "v206", "v207", "ccA", X3B_EVEN, X3B_PF("v[200:203]", "th0", "cp0", "pt0"), S2B_PFE("v204"), \
                            X3B_BLOCK("v208", "v209", "v210", "v211", "th1", "cp1", "pt1", "ccA"))
#define X3B_STEP_1 X3B_APPLY(X3B_STEP, "v[212:215]", "v212", "v213", "v214", "v215", "ccB", X3B_ODD,  X3B_PF("v[208:211]", "th1", "cp1", "pt1"), S2B_PFE("v212"), \
                            X3B_BLOCK("v216", "v217", "v218", "v219", "th2", "cp2", "pt2", "ccB"))
#define X3B_STEP_2 X3B_APPLY(X3B_STEP, "v[220:223]", "v220", "v221", "v222", "v223", "ccA", X3B_EVEN, X3B_PF("v[216:219]", "th2", "cp2", "pt2"), S2B_PFE("v220"), \
                            X3B_BLOCK("v224", "v225", "v226", "v227", "th3", "cp3", "pt3", "ccA"))
#define X3B_STEP_3 X3B_APPLY(X3B_STEP, "v[228:231]", "v228", "v229", "v230", "v231", "ccB", X3B_ODD,  X3B_PF("v[224:227]", "th3", "cp3", "pt3"), S2B_PFE("v228"), \
                            X3B_BLOCK("v200", "v201", "v202", "v203", "th0", "cp0", "pt0", "ccB"))

__global__ __launch_bounds__(256) void lstm_bwd_s2_x3_asm_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HP = 128, KCS = 8;
    constexpr int pitch = lds_pitch(KCS * 64);       // 544
    constexpr int plane = 9 * pitch;                 // 4896: the asm carries it (and pitch / 4 = 136) as literals
    static_assert(pitch == 544 && plane == 4896 && CN_GUARD_STEPS >= 5, "LDS offsets / prefetch distance of the hand-written loop");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int ug = q >> 1, sq = q & 1;
    const int d = blockIdx.x % p.dirs, s0 = (blockIdx.x / p.dirs) * 2;
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    const long arow = (long)dirs * 4 * HP, crow = (long)dirs * HP;

    for (int i = threadIdx.x * 4; i < 2 * plane; i += blockDim.x * 4) *(unsigned *)(smem + i) = 0u;

    u32x8 wh[2][KCS], wl[2][KCS];
    const float *Wd = (const float *)p.WrecT + (long)d * 4 * HP * HP;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kc = 0; kc < KCS; ++kc)
            sp_load_split(Wd + (long)(32 * wave + 16 * j + c) * 4 * HP + kc * 64 + q * 16, wh[j][kc], wl[j][kc]);
    const int spidx = sp_index(c);
    const unsigned av0 = (c < 8 ? c : 8) * pitch + q * 16, av1 = (c >= 8 ? c - 8 : 8) * pitch + q * 16;

    const int unit = 32 * wave + 16 * ug + c;
    const float pi = p.peep[(d * 3 + 0) * HP + unit], pf = p.peep[(d * 3 + 1) * HP + unit], po = p.peep[(d * 3 + 2) * HP + unit];
    const int sv = s0 + sq;
    const unsigned oT = (4 * sq) * pitch + ((unit >> 4) * 32 + sp_pos(4 * (unit & 15))) * 2, oT1 = oT + plane;
    const unsigned oTl = oT + 2 * pitch, oT1l = oT1 + 2 * pitch;
    // offsets BIAS steps ahead, bases BIAS steps behind (see lstm_bwd_s2_asm_kernel); the fp32 deltas share the activations' offset
    constexpr long BIAS = 8;
    const long t0 = d ? 0 : T - 1, dt = d ? 1 : -1;
    const long stepA = (long)PS * arow, stepC = (long)PS * crow;
    const unsigned lC = (unsigned)(sv * (int)crow + d * HP + unit);
    unsigned oA = (unsigned)((t0 + BIAS) * stepA * 4) + lC * 16, oC = (unsigned)((t0 + BIAS) * stepC * 4) + lC * 4;
    unsigned oP = (unsigned)((t0 + BIAS) * PS) + (unsigned)sv;
    const unsigned sA = (unsigned)(dt * stepA * 4), sC = (unsigned)(dt * stepC * 4), sP = (unsigned)(dt * PS);
    const char *acts = (const char *)p.acts - BIAS * stepA * 4, *cell = (const char *)p.cell - BIAS * stepC * 4, *th = (const char *)p.th - BIAS * stepC * 4;
    const char *err = (const char *)p.err - BIAS * stepC * 4, *pat = p.pat - BIAS * PS;
    const char *actspf = acts + 4 * dt * stepA * 4, *thpf = th + 4 * dt * stepC * 4, *errpf = err + (4 - 1) * dt * stepC * 4;
    const char *cell1 = cell + dt * stepC * 4, *cellpf = cell + 5 * dt * stepC * 4, *patpf = pat + 4 * dt * PS;
    const char *delta1 = (const char *)p.delta_op - (BIAS + dt) * stepA * 4;
    unsigned cnt = (unsigned)T - 1;

    float fgn = 0.f, ecn = 0.f;
    float sb0 = 0.f, sb1 = 0.f, sb2 = 0.f, sb3 = 0.f, spi = 0.f, spf = 0.f, spo = 0.f;
    float ccA, ccB, th0, th1, th2, th3, cp0, cp1, cp2, cp3;
    int pt0, pt1, pt2, pt3;
    u32x4 r00, r01, r02, r03, r04, r05, r06, r07, r10, r11, r12, r13, r14, r15, r16, r17;
    float x0, x1, m, t2m, wm, carm, d2m, d3m, d4m, car;
    unsigned ha, hb, la, lb;
    unsigned long long last;
    lds_barrier();
    asm volatile(
        // the carried deltas of the step before the first one are zero
        "v_mov_b32 v232, 0\n\tv_mov_b32 v233, 0\n\tv_mov_b32 v234, 0\n\tv_mov_b32 v235, 0\n\t"
        "global_load_dword %[ccA], %[oC], %[cell]\n\t"
        "v_mov_b32 %[x0], %[oA]\n\tv_mov_b32 %[x1], %[oC]\n\tv_mov_b32 %[m], %[oP]\n\t"
        S2B_FIRST("v[200:203]", "v204", "th0", "cp0", "pt0")
        S2B_FIRST("v[208:211]", "v212", "th1", "cp1", "pt1")
        S2B_FIRST("v[216:219]", "v220", "th2", "cp2", "pt2")
        S2B_FIRST("v[224:227]", "v228", "th3", "cp3", "pt3")
        "s_waitcnt vmcnt(0)\n\t"
        "s_cmp_eq_u32 %[cnt], 0\n\t"
        "s_cselect_b64 %[last], -1, 0\n\t"
        "s_nop 1\n\t"
        X3B_BLOCK("v200", "v201", "v202", "v203", "th0", "cp0", "pt0", "ccB")
        "1:\n\t"
        X3B_STEP_0
        X3B_STEP_1
        X3B_STEP_2
        X3B_STEP_3
        "s_branch 1b\n\t"
        "9:\n\t"
        // the prefetches of the last steps are still in flight; hipcc counts an asm load's destination as written when the
        // statement ends and reuses those registers in the epilogue (the atomics' addresses): drain before leaving
        "s_waitcnt vmcnt(0)\n\t"
        "v_add_f32 %[sb0], %[sb0], v232\n\tv_add_f32 %[sb1], %[sb1], v233\n\tv_add_f32 %[sb2], %[sb2], v234\n\tv_add_f32 %[sb3], %[sb3], v235\n\t"
        : [fgn] "+v"(fgn), [ecn] "+v"(ecn),
          [sb0] "+v"(sb0), [sb1] "+v"(sb1), [sb2] "+v"(sb2), [sb3] "+v"(sb3), [spi] "+v"(spi), [spf] "+v"(spf), [spo] "+v"(spo),
          [oA] "+v"(oA), [oC] "+v"(oC), [oP] "+v"(oP), [cnt] "+s"(cnt), [last] "=&s"(last),
          [ccA] "=&v"(ccA), [ccB] "=&v"(ccB), [th0] "=&v"(th0), [th1] "=&v"(th1), [th2] "=&v"(th2), [th3] "=&v"(th3),
          [cp0] "=&v"(cp0), [cp1] "=&v"(cp1), [cp2] "=&v"(cp2), [cp3] "=&v"(cp3),
          [pt0] "=&v"(pt0), [pt1] "=&v"(pt1), [pt2] "=&v"(pt2), [pt3] "=&v"(pt3),
          [r00] "=&v"(r00), [r01] "=&v"(r01), [r02] "=&v"(r02), [r03] "=&v"(r03), [r04] "=&v"(r04), [r05] "=&v"(r05), [r06] "=&v"(r06), [r07] "=&v"(r07),
          [r10] "=&v"(r10), [r11] "=&v"(r11), [r12] "=&v"(r12), [r13] "=&v"(r13), [r14] "=&v"(r14), [r15] "=&v"(r15), [r16] "=&v"(r16), [r17] "=&v"(r17),
          [x0] "=&v"(x0), [x1] "=&v"(x1), [m] "=&v"(m), [t2m] "=&v"(t2m), [wm] "=&v"(wm), [carm] "=&v"(carm),
          [d2m] "=&v"(d2m), [d3m] "=&v"(d3m), [d4m] "=&v"(d4m), [car] "=&v"(car),
          [ha] "=&v"(ha), [hb] "=&v"(hb), [la] "=&v"(la), [lb] "=&v"(lb)
        : [h0k0] "a"(wh[0][0]), [h0k1] "a"(wh[0][1]), [h0k2] "a"(wh[0][2]), [h0k3] "a"(wh[0][3]), [h0k4] "a"(wh[0][4]), [h0k5] "a"(wh[0][5]), [h0k6] "a"(wh[0][6]), [h0k7] "a"(wh[0][7]),
          [h1k0] "a"(wh[1][0]), [h1k1] "a"(wh[1][1]), [h1k2] "a"(wh[1][2]), [h1k3] "a"(wh[1][3]), [h1k4] "a"(wh[1][4]), [h1k5] "a"(wh[1][5]), [h1k6] "a"(wh[1][6]), [h1k7] "a"(wh[1][7]),
          [l0k0] "a"(wl[0][0]), [l0k1] "a"(wl[0][1]), [l0k2] "a"(wl[0][2]), [l0k3] "a"(wl[0][3]), [l0k4] "a"(wl[0][4]), [l0k5] "a"(wl[0][5]), [l0k6] "a"(wl[0][6]), [l0k7] "a"(wl[0][7]),
          [l1k0] "a"(wl[1][0]), [l1k1] "a"(wl[1][1]), [l1k2] "a"(wl[1][2]), [l1k3] "a"(wl[1][3]), [l1k4] "a"(wl[1][4]), [l1k5] "a"(wl[1][5]), [l1k6] "a"(wl[1][6]), [l1k7] "a"(wl[1][7]),
          [spidx] "v"(spidx), [av0] "v"(av0), [av1] "v"(av1), [oT] "v"(oT), [oT1] "v"(oT1), [oTl] "v"(oTl), [oT1l] "v"(oT1l), [pi] "v"(pi), [pf] "v"(pf), [po] "v"(po),
          [acts] "s"(acts), [actspf] "s"(actspf), [cell] "s"(cell), [cell1] "s"(cell1), [cellpf] "s"(cellpf), [th] "s"(th), [thpf] "s"(thpf),
          [err] "s"(err), [errpf] "s"(errpf), [pat] "s"(pat), [patpf] "s"(patpf), [delta1] "s"(delta1),
          [sA] "s"(sA), [sC] "s"(sC), [sP] "s"(sP)
        : "memory", "vcc", "scc",
          "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211",
          "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223",
          "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235");

    float v[7] = {sb0, sb1, sb2, sb3, spi, spf, spo};
#pragma unroll
    for (int i = 0; i < 7; ++i) v[i] += __shfl_xor(v[i], 16);
    if (sq == 0) {
        lstm_grad_sums_out(p, HP, d, unit, v);
    }
}

// ---------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------
static size_t s2_lds_bytes(int prec, bool bwd, int Hp, int T)
{
    const bool x3 = prec == P_X3;                                   // split-bf16: row quads (hi, lo) -- 9 rows, backward rows of the whole K
    const size_t pitch = (size_t)lds_pitch(bwd ? (x3 ? 4 : 2) * Hp : Hp);      // bwd: KCH * 64 bytes per row
    return 2 * (size_t)(bwd || x3 ? 9 : 5) * pitch + (bwd ? (((size_t)T * 2 + 15) & ~(size_t)15) : 0);
}

// the hand-written loops cover Hp = 128 in bf16 (both passes) and split-bf16 (see s2_x3_asm), at least four time steps in the
// bf16 forward loop (its tail copies), and activations addressable with 32-bit byte offsets
static bool s2_asm_applies(int prec, bool bwd, const LstmRec &p)
{
    if (opt().no_s2_asm || prec == P_F32 || p.Hp != 128) return false;
    if (prec == P_BF16 && !bwd && p.T < 4) return false;
    if (bwd && opt().no_s2_asm_bwd) return false;
    return (unsigned long long)(p.T + 16) * p.PS * p.dirs * 4 * p.Hp * 4 < 0xF0000000ull;     // (+16: offsets run 8 steps ahead)
}

// The shape applies when the layer is one the row-pair products cover (bf16, Hp = 64 or 128), the caller
// chose one sequence per lane (PS * dirs / 4 workgroups fit the chip: rpl == 1), and twice that grid still leaves every
// workgroup a CU of its own.
bool lstm_s2_applies(int prec, const LstmRec &p, bool bwd)
{
    // split-bf16: only where the hand-written loops exist (Hp = 128).  The mode is bound by the MFMA pipe (32 MFMAs per SIMD and
    // step), which this cut does not relieve, while it takes away the second wave that hides the first one's VALU work: the
    // COMPILED kernels of the cut lose to the 8-wave kernels (CN_S2_X3=1 selects them anyway, for the tests).
    if (prec == P_X3 && !opt().s2_x3 && !s2_asm_applies(prec, bwd, p)) return false;
    if (opt().no_s2 || prec == P_F32 || p.rpl != 1 || (p.Hp != 64 && p.Hp != 128) || p.PS % 2) return false;
    if (p.dirs * (p.PS / 2) > p.num_cus) return false;
    return s2_lds_bytes(prec, bwd, p.Hp, p.T) <= 160 * 1024;
}

// Hp = 256 on one CU (forward pass, bf16): before the 2-CU cluster kernels wherever every pair of sequences gets a CU
bool lstm_s2w_applies(int prec, const LstmRec &p, bool bwd)
{
    if (opt().no_s2w || bwd || prec != P_BF16 || p.Hp != 256 || p.rpl != 1 || p.PS % 2) return false;
    if (p.dirs * (p.PS / 2) > p.num_cus) return false;
    return (unsigned long long)(p.T + 16) * p.PS * p.dirs * 4 * p.Hp * 4 < 0xF0000000ull;     // 32-bit byte offsets, 8 steps ahead
}

void launch_lstm_s2w(hipStream_t s, bool bwd, const LstmRec &p, hipEvent_t done)
{
    static DeviceOnce once;
    if (once.first()) {
        (void)hipFuncSetAttribute((const void *)lstm_fwd_s2w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)lstm_fwd_s2w_asm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const bool hand = !opt().no_s2w_asm;
    const size_t lds = 2 * 5 * (size_t)lds_pitch(256) + 64 + 4 * (size_t)(hand ? S2W_STREAM_COUNT * 2048 : 32768);
    lstm_note_grid(p, p.dirs * (p.PS / 2));
    hipExtLaunchKernelGGL(hand ? lstm_fwd_s2w_asm_kernel : lstm_fwd_s2w_kernel, dim3(p.dirs * (p.PS / 2)), dim3(256), lds, s, nullptr, done, 0, p);
    if (p.kname) snprintf(p.kname, CN_KNAME_LEN, hand ? "lstm_fwd_s2w_asm_kernel" : "lstm_fwd_s2w_kernel");
}

template <int PREC, bool BWD, int HP>
static void launch_s2(hipStream_t s, const LstmRec &p, hipEvent_t done)
{
    if constexpr (HP == 128) {
        if (s2_asm_applies(PREC, BWD, p)) {
            // (the bf16 forward loop has two forms: fp32 pre-activations out of `acts`, or bf16 ones out of LstmRec::pre16)
            void (*akern)(LstmRec) = PREC == P_X3 ? (BWD ? lstm_bwd_s2_x3_asm_kernel : lstm_fwd_s2_x3_asm_kernel)
                                                  : (BWD ? lstm_bwd_s2_asm_kernel : (p.pre16 ? lstm_fwd_s2_asm_kernel<true> : lstm_fwd_s2_asm_kernel<false>));
            static DeviceOnce once[2];
            if (once[p.pre16 ? 1 : 0].first()) (void)hipFuncSetAttribute((const void *)akern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            const size_t lds = s2_lds_bytes(PREC, BWD, HP, p.T);
            size_t lds_claim = lds;
            if (p.dirs * (p.PS / 2) <= 128 && !opt().no_lds_claim) lds_claim = 160 * 1024 - 1024;
            lstm_note_grid(p, p.dirs * (p.PS / 2));
            hipExtLaunchKernelGGL(akern, dim3(p.dirs * (p.PS / 2)), dim3(256), lds_claim < lds ? lds : lds_claim, s, nullptr, done, 0, p);
            if (p.kname) snprintf(p.kname, CN_KNAME_LEN, "lstm_%s_s2%s_asm_kernel", BWD ? "bwd" : "fwd", PREC == P_X3 ? "_x3" : "");
            return;
        }
    }
    auto kern = BWD ? lstm_bwd_s2_kernel<PREC, HP> : lstm_fwd_s2_kernel<PREC, HP>;
    static DeviceOnce attr_once;
    if (attr_once.first()) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = s2_lds_bytes(PREC, BWD, HP, p.T);
    // claim the CU's whole LDS so that no workgroup of a concurrently running kernel is placed beside it (cn_lstm.hip)
    size_t lds_claim = lds;
    if (p.dirs * (p.PS / 2) <= 128 && !opt().no_lds_claim) lds_claim = 160 * 1024 - 1024;
    lstm_note_grid(p, p.dirs * (p.PS / 2));
    hipExtLaunchKernelGGL(kern, dim3(p.dirs * (p.PS / 2)), dim3(HP * 2), lds_claim < lds ? lds : lds_claim, s, nullptr, done, 0, p);
    if (p.kname) snprintf(p.kname, CN_KNAME_LEN, "lstm_%s_s2_kernel<%d,%d>", BWD ? "bwd" : "fwd", PREC, HP);
}

void launch_lstm_s2(hipStream_t s, int prec, bool bwd, const LstmRec &p, hipEvent_t done)
{
    if (prec == P_X3) {
        if (p.Hp == 128) { if (bwd) launch_s2<P_X3, true, 128>(s, p, done); else launch_s2<P_X3, false, 128>(s, p, done); }
        else             { if (bwd) launch_s2<P_X3, true, 64>(s, p, done);  else launch_s2<P_X3, false, 64>(s, p, done); }
    } else {
        if (p.Hp == 128) { if (bwd) launch_s2<P_BF16, true, 128>(s, p, done); else launch_s2<P_BF16, false, 128>(s, p, done); }
        else             { if (bwd) launch_s2<P_BF16, true, 64>(s, p, done);  else launch_s2<P_BF16, false, 64>(s, p, done); }
    }
}

}  // namespace cn

// Persistent recurrent LSTM kernels: the whole time loop of one layer pass in ONE launch.
//
// Replaces the per-timestep launch storm of LstmLayer<TDevice>::computeForwardPass
// (LstmLayer.cu:812-829 fw, :847-864 bw: 4 x addProduct + ComputeBlockOutputFn :47-138 per step)
// and computeBackwardPass (:936-951, :970-985: 4 x addProduct + ComputeBlockErrorsFn :190-287 per
// step), plus ResortOutputsFn/ResortOutputErrorsFn (:140-188) and the bias / peephole parts of
// ComputeWeightUpdateFn (:289-512).
//
// MI355X design:
//   * parallel sequences never interact inside a layer pass, so the PS sequences are cut into
//     small groups (4, 8 or 16 sequences, picked so that the groups spread over the 256 CUs) and
//     every (direction, sequence group) pair is one workgroup that walks all T steps on its own:
//     no inter-workgroup communication, no grid barrier.  The per-step cost is a dependency chain
//     (LDS hand-off -> MFMA -> cell update -> LDS hand-off), so fewer sequences per workgroup
//     means fewer cell updates per lane and a shorter step.
//   * the four gates are packed into one tile row: wave w owns hidden units [16w, 16w+16) and
//     the 4 gate tiles of those units, so after the MFMAs every lane holds n/i/f/o of one unit
//     and the cell update needs no cross-lane traffic.  Gate pre-activations from the N-wide input
//     GEMM (bias already added) are stored [frame][unit][gate] so a lane moves them with one 16-byte
//     access; they are added after the MFMAs (dense path) or seed the accumulators (row-pair path).
//   * bf16 / split-bf16 modes with at most two sequences per lane: the recurrent products run on the 2:4 sparse
//     MFMA with every sequence spread over two tile rows ("row pairs", cn_lstm_device.h) -- half the MFMA-pipe
//     time and half the operand reads of the dense tile, nothing pruned.
//   * W_rec fragments stay in registers for the whole pass when 4*Hp*Hp operands fit one CU's
//     register file (Hp <= 128), otherwise they are streamed from L2 every step.
//   * y[t] (forward) / the four deltas (backward) are exchanged between the waves of the workgroup
//     through a double-buffered LDS tile in MFMA A-operand order; cell state, forget-gate carry and
//     the peephole/bias gradient sums never leave registers.  Operands of the next two steps are
//     prefetched from HBM while the current step computes; the LDS barrier does not drain them.
//     (Measured and rejected: issuing the loop's loads as LDS-DMA into a per-wave ring plus asm stores with
//     exact s_waitcnt vmcnt(2*ST+LD) counts -- bit-identical results, 5-8 % SLOWER: the step is bound by
//     the MFMA -> cell-update -> LDS hand-off chain, not by memory waits.  A register-destination asm
//     load is not an option at all: hipcc rotates the staged registers with v_mov while the load is in
//     flight.)
//   * fw/bw halves are written straight into the interleaved [N][2*Hp] layer output.
#include "cn_internal.h"
#include <stdexcept>
#include <string>
#include "cn_lstm_device.h"

#include <cstdio>
#include <cstdlib>

#ifndef CN_KQ_STACK
#define CN_KQ_STACK 1
#endif
#ifndef CN_SPARSE
#define CN_SPARSE 1
#endif
#ifndef CN_TH_STORE
#define CN_TH_STORE 1
#endif
#ifndef CN_X3_ACCURATE_ACT
#define CN_X3_ACCURATE_ACT 0
#endif

namespace cn {

// In-kernel segment timing (tools/stamps.py builds a second library with -DCN_STAMP; never defined in the
// shipped build): every wave of workgroup 0 sums s_memtime deltas per step segment.
#ifdef CN_STAMP
__device__ unsigned long long cn_stamp_buf[2][16][8];
#define STAMP_DECL unsigned long long st_prev = __builtin_amdgcn_s_memtime(), st_acc[6] = {0, 0, 0, 0, 0, 0};
#define STAMP(i) { unsigned long long st_now = __builtin_amdgcn_s_memtime(); st_acc[i] += st_now - st_prev; st_prev = st_now; }
#define STAMP_FORCE(x) asm volatile("v_mov_b32 %0, %0" : "+v"(x));
#define STAMP_RESET st_prev = __builtin_amdgcn_s_memtime();
#define STAMP_STORE(k) if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 6; ++i_) cn_stamp_buf[k][threadIdx.x >> 6][i_] = st_acc[i_]; }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FORCE(x)
#define STAMP_RESET
#define STAMP_STORE(k)
#endif

// element `elem` of an array of T at a wave-uniform base: the byte offset stays a 32-bit VGPR (saddr form)
// Keeps every lane of an MFMA accumulator tuple allocated until `after` (a value computed from the tuple's results) exists.
// Lanes whose rows are never read are dead to the register allocator, which may hand them to another value while the MFMA
// that will still write them is in flight.  For ordinary instructions the hazard recognizer then inserts the wait states; the
// staged-operand copies of these kernels are inline asm, which it does not look into: observed once (an experimental build,
// T = 1 path): `v_smfmac v[140:143]` followed by the asm copy `v_mov_b32 v142, ...`, overwritten when the MFMA retired.
#define KEEP_TUPLE(tuple, after) asm volatile("" :: "v"(tuple), "v"(after))

template <typename T> __device__ __forceinline__ T &at32(const void *base, unsigned elem)
{
    return *(T *)((char *)base + elem * (unsigned)sizeof(T));
}

// dtab[t][j] = (t >= Tmin && patTypes[t][s0 + j] == NONE) for the 4*RPL sequences of a workgroup.  One dword
// (four sequences) per thread and round, so a pass of up to blockDim.x / RPL time steps is a single round
// trip; PS and s0 are multiples of 4, the rows are dword aligned.
template <int RPL>
__device__ __forceinline__ void build_dummy_table(unsigned char *dtab, const char *pat, int T, int Tmin, int PS, int s0)
{
    for (int i = threadIdx.x; i < T * RPL; i += blockDim.x) {
        const int tt = i / RPL, jj = i % RPL;
        const unsigned w = *(const unsigned *)(pat + (long)tt * PS + s0 + 4 * jj);
        unsigned f = 0;
        if (tt >= Tmin) {
#pragma unroll
            for (int b = 0; b < 4; ++b) f |= (((w >> (8 * b)) & 0xffu) == 0u ? 1u : 0u) << (8 * b);
        }
        *(unsigned *)(dtab + 4 * i) = f;
    }
}

// ---------------------------------------------------------------------------------------------
// forward: a[t] = G[t] + Wrec^T y[prev(t)]; ComputeBlockOutputFn
// ---------------------------------------------------------------------------------------------
// HP  : padded units per direction when the W_rec fragments are register resident, 0 = stream W_rec
// UG  : unit groups (of 16) per wave
// RPL : sequences per lane; a workgroup handles 4*RPL sequences: sequence s0 + 4*r + q sits in MFMA
//       tile row 4*q + r (q = lane>>4, r < RPL), rows r >= RPL are zero padding
// PREC : P_BF16 / P_F32 / P_X3 (cn_internal.h).  P_X3 keeps fp32 in memory like P_F32 and two bf16 planes (hi, lo) of the y
//        tile in LDS; W_rec is split once, in the prologue, into hi and lo fragments (the registers the fp32 fragments take
//        in P_F32), and a K chunk of the product is three bf16 MFMAs instead of eight fp32 ones.
template <int PREC, int HP, int UG, int RPL>
__global__ __launch_bounds__(HP ? HP * 4 / UG : 1024) void lstm_fwd_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ACC: libm-grade expf and IEEE division in the activations (exact-fp32 mode only: v_exp_f32 / v_rcp_f32 are good to
    // ~1 ulp = 2^-23, far below the 2^-16 of the split products, and the accurate forms cost the P_X3 forward step 25 %)
    constexpr bool F32 = PREC == P_F32, X3 = PREC == P_X3, ACC = PREC == P_F32 || (X3 && CN_X3_ACCURATE_ACT);
    constexpr int MELT = PREC == P_BF16 ? 2 : 4;     // operand element in memory (y, W_rec)
    constexpr int ELT = F32 ? 4 : 2;                 // operand element in LDS / MFMA fragments
    constexpr int PLANES = X3 ? 2 : 1;
    constexpr bool RES = HP != 0;
    // SP: 2:4 row-pair products (cn_lstm_device.h): with at most two sequences per lane every sequence takes two tile rows and
    // a K = 64 chunk is one v_smfmac_f32_16x16x64_bf16 instead of two dense MFMAs; the tile rows are half as long
    constexpr bool SP = RES && UG == 1 && RPL <= 2 && !F32 && HP % 64 == 0 && CN_SPARSE;
    constexpr int KCS = SP ? HP / 64 : 1;            // 64-value K chunks of the sparse product
    const int Hp = RES ? HP : p.Hp;
    const int KC = Hp * ELT / 64;                    // 64-byte K chunks
    constexpr int KCR = RES ? HP * ELT / 64 : 1;
    const int pitch = lds_pitch(SP ? Hp * ELT / 2 : Hp * ELT);   // LDS row pitch of the y tile (bytes)
    const int nw = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    // LDS tile rows: the resident kernels keep the full 16-row MFMA operand tile; the streaming fallback (W_rec read
    // from L2 every step, HP = 0: large layers) keeps the 4*RPL real rows plus ONE shared zero row, which is what
    // lets a 4*Hp-wide fp32 / bf16 delta row of 8 KB (Hp = 512 fp32, Hp = 1024 bf16) fit the 160 KB LDS twice
    constexpr int TROWS = RES ? 16 : 4 * RPL + 1;
    [[maybe_unused]] const int rrow = RES ? c : ((c & 3) < RPL ? (c >> 2) * RPL + (c & 3) : 4 * RPL);
    const int d = blockIdx.x % p.dirs, s0 = (blockIdx.x / p.dirs) * (4 * RPL);
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    const long arow = (long)dirs * 4 * Hp;           // acts row stride (floats)
    const long crow = (long)dirs * Hp;               // cell / y row stride (elements)

    // zero both y tiles (y[prev] of the first processed step is 0; padding rows stay 0)
    const int plane = TROWS * pitch;                 // P_X3: hi plane, then lo plane
    for (int i = threadIdx.x * 4; i < 2 * PLANES * plane; i += blockDim.x * 4) *(unsigned *)(smem + i) = 0u;

    int unit[UG];
    float pi[UG], pf[UG], po[UG];
    [[maybe_unused]] u32x4 wreg[UG][4][SP ? 1 : KCR];
    [[maybe_unused]] u32x4 wlo[X3 && !SP ? UG : 1][4][SP ? 1 : KCR];
    [[maybe_unused]] u32x8 wsp[SP ? 4 : 1][KCS], wsl[SP && X3 ? 4 : 1][KCS];
    [[maybe_unused]] const int spidx = sp_index(c);
    const char *Wd = (const char *)p.Wrec + (long)d * 4 * Hp * Hp * MELT;
#pragma unroll
    for (int u = 0; u < UG; ++u) {
        unit[u] = 16 * (wave + u * nw) + c;
        pi[u] = p.peep[(d * 3 + 0) * Hp + unit[u]];
        pf[u] = p.peep[(d * 3 + 1) * Hp + unit[u]];
        po[u] = p.peep[(d * 3 + 2) * Hp + unit[u]];
        if constexpr (SP) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int kc = 0; kc < KCS; ++kc) {
                    const long w0 = (long)(g * Hp + unit[u]) * Hp + kc * 64 + q * 16;     // 16 consecutive K values per lane
                    if constexpr (X3) sp_load_split((const float *)Wd + w0, wsp[g][kc], wsl[g][kc]);
                    else wsp[g][kc] = sp_load_bf16(Wd + w0 * 2);
                }
        } else if constexpr (RES) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int kc = 0; kc < KCR; ++kc) {
                    if constexpr (X3) {
                        const float *wp = (const float *)Wd + (long)(g * Hp + unit[u]) * Hp + kc * 32 + q * 8;
                        split8(*(const f32x4 *)wp, *(const f32x4 *)(wp + 4), wreg[u][g][kc], wlo[u][g][kc]);
                    } else
                        wreg[u][g][kc] = *(const u32x4 *)(Wd + ((long)(g * Hp + unit[u]) * Hp) * ELT + kc * 64 + q * 16);
                }
        }
    }

    // per-lane element offsets inside one time step (32-bit; the time step itself moves a wave-uniform
    // base pointer, so a step costs no 64-bit per-lane address arithmetic)
    // PS is padded to a multiple of 4*RPL on the device (pad slots are permanent dummies), so every lane
    // owns real memory: no lane predication, no divergent branch inside the time loop.
    unsigned oP[RPL], oA[UG][RPL], oC[UG][RPL];     // unsigned: base (SGPR pair) + 32-bit lane offset, no 64-bit VALU add per access
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int sv = s0 + 4 * r + q;
        // pattern types travel as the aligned dword of the four sequences of a tile row group (PS and s0 are multiples of 4);
        // the offset is formally per lane (vzero) so that it stays a vector load: a scalar load would share lgkmcnt with the LDS
        unsigned vzero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
        oP[r] = (unsigned)(s0 + 4 * r) / 4 + vzero;
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            oA[u][r] = sv * (int)arow + (d * Hp + unit[u]) * 4;
            oC[u][r] = sv * (int)crow + d * Hp + unit[u];
        }
    }
    const long stepA = (long)PS * arow, stepC = (long)PS * crow;
    // tile position of this lane's y (byte offset of the bf16 / fp32 value inside one tile plane)
    int oT[UG][RPL];
#pragma unroll
    for (int u = 0; u < UG; ++u)
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            if constexpr (SP) {
                const int k = unit[u] & 63;
                oT[u][r] = (4 * q + 2 * r + sp_parity(k)) * pitch + ((unit[u] >> 6) * 32 + sp_pos(k)) * 2;
            } else oT[u][r] = (RES ? 4 * q + r : q * RPL + r) * pitch + unit[u] * ELT;
        }

    float cst[UG][RPL];
#pragma unroll
    for (int u = 0; u < UG; ++u)
#pragma unroll
        for (int r = 0; r < RPL; ++r) cst[u][r] = 0.f;

    // two prefetch stages (steps it and it+1), addressed statically through the lambda parameters.
    // Prefetches are unconditional (time index clamped) so the staged registers are plain load
    // results: no select against an old value, no wait at the loop back edge.
    f32x4 preA[UG][RPL], preB[UG][RPL];
    int ptA[RPL], ptB[RPL];         // pattern types, kept as whole dwords (a byte-typed stage made hipcc copy the
                                    // freshly loaded register, which waits for the load in the middle of the step)
    auto prefetch = [&](int t, f32x4 (&pre)[UG][RPL], int (&pt)[RPL]) {
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        const float *actsT = p.acts + t * stepA;
        const char *patT = p.pat + (long)t * PS;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            pt[r] = (int)at32<unsigned>(patT, oP[r]);
#pragma unroll
            for (int u = 0; u < UG; ++u) pre[u][r] = *(const f32x4 *)&at32<float>(actsT, oA[u][r]);
        }
    };

    // SP: the accumulators live across steps.  The sparse MFMA accumulates in place (no C operand that could be the inline
    // constant 0), so every step would clear sixteen registers; kept alive, the even row of a sequence is loaded with the
    // staged pre-activation (the copy the dense path makes anyway), its odd row is cleared, and with one sequence per lane the
    // rows of the unused pair are never touched again: their tile rows are zero, they stay what they are.
    [[maybe_unused]] f32x4 accp[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) accp[g] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned ptmask = 0xffu << (8 * q);        // this lane's byte of a pattern-type dword
    STAMP_DECL
    auto step = [&](int it, f32x4 (&pre)[UG][RPL], int (&pt)[RPL]) {
        const int t = d ? T - 1 - it : it;
        const char *ycur = smem + (it & 1) * PLANES * plane;
        char *ynxt = smem + ((it + 1) & 1) * PLANES * plane;
        const bool check = t >= p.Tmin;              // LstmLayer.cu:825,860
        float *actsT = p.acts + t * stepA;
        float *cellT = p.cell + t * stepC;
        [[maybe_unused]] float *thT = p.th + t * stepC;
        char *yT = (char *)p.y_op + t * stepC * MELT;

        // accumulators start at 0 (inline constant, nothing to set up before the first MFMA); the gate
        // pre-activation of the N-wide GEMM is added to the real rows afterwards
        f32x4 acc[UG][4];
        f32x4 g_[UG][RPL];
        bool dummy_[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            // (staged like the pre-activations, through an asm instruction that stays in this step and picks this lane's byte:
            // hipcc otherwise evaluates the OTHER stage's pattern type one step early, behind an s_waitcnt for a load that was
            // issued only a step before)
            int ptc;
            asm volatile("v_and_b32 %0, %1, %2" : "=&v"(ptc) : "v"(pt[r]), "v"(ptmask));
            dummy_[r] = check && ptc == 0;
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            // the staged values move to registers of their own (early-clobber asm copy) so that the prefetch
            // below can land in the stage's registers again; left to the register coalescer, the stage keeps
            // being read by the cell update, the load goes elsewhere and is copied back right behind its issue
#pragma unroll
            for (int r = 0; r < RPL; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v_;
                    asm volatile("v_mov_b32 %0, %1" : "=&v"(v_) : "v"(pre[u][r][g]));
                    if constexpr (SP) { accp[g][2 * r] = v_; accp[g][2 * r + 1] = 0.f; g_[u][r][g] = 0.f; }
                    else g_[u][r][g] = v_;
                }
            if constexpr (!SP) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[u][g][r] = 0.f;
            }
        }

        prefetch(d ? t - 2 : t + 2, pre, pt);
        STAMP(0)

        // recurrent product (LstmLayer.cu:815-818 / :850-853), all four gates at once
        if constexpr (SP) {
#pragma unroll
            for (int kc = 0; kc < KCS; ++kc) {
                u32x4 a = *(const u32x4 *)(ycur + c * pitch + kc * 64 + q * 16);
                [[maybe_unused]] u32x4 al;
                if constexpr (X3) al = *(const u32x4 *)(ycur + plane + c * pitch + kc * 64 + q * 16);
#ifdef CN_STAMP
                if (kc == 0) { STAMP_FORCE(a[0]) STAMP(1) }
#endif
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (X3) smma16_x3(accp[g], a, al, wsp[g][kc], wsl[g][kc], spidx);
                    else smma16(accp[g], a, wsp[g][kc], spidx);
                }
            }
        } else if constexpr (RES && UG > 1 && KCR <= 8 && !X3) {
            // unit group after unit group: the cell update of group u only needs that group's sums, so it can
            // run on the VALU while the MFMAs of group u+1 are in flight (one wave per SIMD in this shape)
            u32x4 a[KCR];
#pragma unroll
            for (int kc = 0; kc < KCR; ++kc) a[kc] = *(const u32x4 *)(ycur + c * pitch + kc * 64 + q * 16);
#ifdef CN_STAMP
            STAMP_FORCE(a[0][0]) STAMP(1)
#endif
#pragma unroll
            for (int u = 0; u < UG; ++u)
#pragma unroll
                for (int kc = 0; kc < KCR; ++kc)
#pragma unroll
                    for (int g = 0; g < 4; ++g) mma16<F32>(acc[u][g], a[kc], wreg[u][g][kc]);
#ifndef CN_STAMP
            // issue order: all LDS reads, the MFMAs of group 0, then every MFMA of the later groups followed by
            // three VALU instructions (the cell update of the group before it)
            __builtin_amdgcn_sched_group_barrier(0x100, KCR, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * KCR, 0);
#pragma unroll
            for (int i = 0; i < 4 * KCR * (UG - 1); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
#endif
        } else if constexpr (RES) {
#pragma unroll
            for (int kc = 0; kc < KCR; ++kc) {
                u32x4 a = *(const u32x4 *)(ycur + c * pitch + kc * 64 + q * 16);
                [[maybe_unused]] u32x4 al;
                if constexpr (X3) al = *(const u32x4 *)(ycur + plane + c * pitch + kc * 64 + q * 16);
#ifdef CN_STAMP
                if (kc == 0) { STAMP_FORCE(a[0]) STAMP(1) }
#endif
#pragma unroll
                for (int u = 0; u < UG; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if constexpr (X3) mma16_x3(acc[u][g], a, al, wreg[u][g][kc], wlo[u][g][kc]);
                        else mma16<F32>(acc[u][g], a, wreg[u][g][kc]);
                    }
            }
        } else {
            for (int kc = 0; kc < KC; ++kc) {
                const u32x4 a = *(const u32x4 *)(ycur + rrow * pitch + kc * 64 + q * 16);
                [[maybe_unused]] u32x4 al;
                if constexpr (X3) al = *(const u32x4 *)(ycur + plane + rrow * pitch + kc * 64 + q * 16);
#pragma unroll
                for (int u = 0; u < UG; ++u) {
                    u32x4 b[4];
                    [[maybe_unused]] u32x4 bl[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if constexpr (X3) {       // fp32 W_rec from L2, split on the fly (functional fallback for wide layers)
                            const float *wp = (const float *)Wd + (long)(g * Hp + unit[u]) * Hp + kc * 32 + q * 8;
                            split8(*(const f32x4 *)wp, *(const f32x4 *)(wp + 4), b[g], bl[g]);
                        } else
                            b[g] = *(const u32x4 *)(Wd + ((long)(g * Hp + unit[u]) * Hp) * ELT + kc * 64 + q * 16);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if constexpr (X3) mma16_x3(acc[u][g], a, al, b[g], bl[g]);
                        else mma16<F32>(acc[u][g], a, b[g]);
                    }
                }
            }
        }

#ifdef CN_STAMP
        { float f_ = SP ? accp[3][0] : acc[UG - 1][3][0]; STAMP_FORCE(f_) STAMP(2) }
#endif
        // cell update: C/D map of the 16x16 MFMA: col = lane&15 (unit), row = 4*(lane>>4)+reg
#pragma unroll
        for (int u = 0; u < UG; ++u) {
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const bool dummy = dummy_[r];
                const float cp = cst[u][r];
                // ComputeBlockOutputFn, LstmLayer.cu:87-136 (bias is already inside the pre-activation)
                float s_[4];                             // recurrent sums of this sequence (SP: its two tile rows)
#pragma unroll
                for (int g = 0; g < 4; ++g)          // SP: the pre-activation entered through the even row's accumulator
                    s_[g] = SP ? accp[g][(2 * r) & 3] + accp[g][(2 * r + 1) & 3] : acc[u][g][r] + g_[u][r][g];
                if (r == RPL - 1) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) { if constexpr (SP) KEEP_TUPLE(accp[g], s_[g]); else KEEP_TUPLE(acc[u][g], s_[g]); }
                }
                const float ni = tanh_ref<ACC>(s_[0]);
                const float ig = logistic<ACC>(s_[1] + cp * pi[u]);
                const float fg = logistic<ACC>(s_[2] + cp * pf[u]);
                const float cs = ni * ig + cp * fg;
                const float og = logistic<ACC>(s_[3] + cs * po[u]);
                const float th = tanh_ref<ACC>(cs);
                const float y = th * og;
                float yo = dummy ? 0.f : y;
                const float co = dummy ? 0.f : cs;     // :78-85 (zeroed in both directions here)
                cst[u][r] = co;
#ifdef CN_STAMP
                if (u == UG - 1 && r == RPL - 1) { STAMP_FORCE(yo) STAMP(3) }
#endif
                if constexpr (F32) *(float *)(ynxt + oT[u][r]) = yo;
                else if constexpr (X3) {
                    __bf16 yh, yl;
                    split_bf16(yo, yh, yl);
                    *(__bf16 *)(ynxt + oT[u][r]) = yh;
                    *(__bf16 *)(ynxt + plane + oT[u][r]) = yl;
                } else *(__bf16 *)(ynxt + oT[u][r]) = (__bf16)yo;
                const f32x4 av = {ni, ig, fg, og};       // (dummy slots: never read back)
                *(f32x4 *)&at32<float>(actsT, oA[u][r]) = av;
                at32<float>(cellT, oC[u][r]) = co;
                at32<float>(thT, oC[u][r]) = th;          // the backward pass reads it back instead of recomputing it (dummy slots: never used)
                if constexpr (MELT == 4) at32<float>(yT, oC[u][r]) = yo;
                else at32<__bf16>(yT, oC[u][r]) = (__bf16)yo;
            }
        }
        STAMP(4)
        lds_barrier();
        STAMP(5)
    };

    prefetch(d ? T - 1 : 0, preA, ptA);
    prefetch(d ? T - 2 : 1, preB, ptB);
    lds_barrier();
    STAMP_RESET
    // The first two steps are peeled so that the loop is entered with the same memory operations in flight
    // as at its back edge.  Entered straight from the prologue, the compiler's s_waitcnt at the loop head
    // has to cover the prologue's load order and becomes vmcnt(1): every other step then waits for the
    // previous step's stores to be acknowledged (~250 cycles of a ~1500 cycle step).
    // The loop body is exactly two steps with no branch between them (an odd last step follows the loop): a
    // conditional second step makes the staged registers phi values, which hipcc resolves with copies right
    // behind the prefetch loads, i.e. with waits for loads it has just issued.
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
    STAMP_STORE(0)
}

// ---------------------------------------------------------------------------------------------
// backward: e[t] = err[t] + Wrec delta[next(t)]; ComputeBlockErrorsFn; bias/peephole gradient sums
// ---------------------------------------------------------------------------------------------
template <int UG, int RPL> struct BwdPre {
    f32x4 a[UG][RPL];        // n, i, f, o of step t
    float e[UG][RPL];        // outputErrors of step t
    float cp[UG][RPL];       // cell state of prev(t)
    float th[UG][RPL];       // tanh(cell state) of step t (CN_TH_STORE)
};

template <int PREC, int HP, int UG, int RPL>
__global__ __launch_bounds__(HP ? HP * 4 / UG : 1024) void lstm_bwd_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool F32 = PREC == P_F32, X3 = PREC == P_X3, ACC = PREC == P_F32 || (X3 && CN_X3_ACCURATE_ACT);
    constexpr int MELT = PREC == P_BF16 ? 2 : 4;     // operand element in memory (delta, W_rec^T)
    constexpr int ELT = F32 ? 4 : 2;                 // operand element in LDS / MFMA fragments (P_X3: two bf16 planes)
    constexpr int PLANES = X3 ? 2 : 1;
    constexpr bool RES = HP != 0;
    const int Hp = RES ? HP : p.Hp;
    const int KC = 4 * Hp * ELT / 64;
    constexpr int KCR = RES ? 4 * HP * ELT / 64 : 1;
    // K-quarter stacking (KQS).  With one sequence per lane (RPL = 1) only rows 0, 4, 8, 12 of the 16-row MFMA operand tile
    // are sequences; the other twelve were zero padding that every wave read from LDS every step.  Stacked, row 4q + r holds
    // K-quarter r of sequence q: the tile is a quarter as wide and has no padding, a step reads a quarter of the operand
    // bytes (Hp = 128: 4 instead of 16 ds_read_b128 per wave, 128 instead of 512 LDS cycles per CU and step), and the product
    // for quarter r is A . W[quarter r] of which only rows 4q + r are wanted: the MFMA count is unchanged, quarter r goes to an
    // accumulator of its own, and a lane finds its sequence's sum as acc_0[0] + acc_1[1] + acc_2[2] + acc_3[3] -- in its own
    // registers (C/D layout: row 4q + r = lane quarter q, register r).
    // Here K = 4*Hp with k = 4*unit + gate, so quarter r = units [r*Hp/4, (r+1)*Hp/4).  Measured: backward step 0.54 -> 0.50 us
    // (rec_bwd 6.04 -> 5.67 ms per 30 launches, headline +2.7 %).  The same stacking in the FORWARD kernel (K = Hp, one read
    // instead of four, sixteen accumulators) loses 13-20 % there, quarter-major or gate-major: its four reads were no
    // bottleneck, sixteen accumulators and twelve extra adds are; it was taken out again.
    // SP: 2:4 row-pair products (cn_lstm_device.h) in the bf16 and split-bf16 modes: a sequence takes two tile rows, a K = 64
    // chunk is one sparse MFMA.  KHS: with one sequence per lane the other two rows of its quad hold the second K half (the
    // stacking idea above with two accumulators instead of four): e = accA[0] + accA[1] + accB[2] + accB[3].
    // Hp = 128: 4 ds_read_b128 and 8 MFMA-pipe slots per wave and step (KQS: 4 and 16; plain: 16 and 16).
    constexpr bool SP = RES && UG == 1 && RPL <= 2 && !F32 && CN_SPARSE;
    constexpr bool KHS = SP && RPL == 1;
    constexpr int KCS = SP ? 4 * HP / 64 : 1;        // 64-value K chunks of the sparse product
    constexpr int KCH = KHS ? KCS / 2 : KCS;         // chunks per tile row
    constexpr bool KQS = !SP && RES && UG == 1 && RPL == 1 && (KCR % 4 == 0) && CN_KQ_STACK;
    constexpr int KCQ = KCR / 4;
    const int pitch = lds_pitch(SP ? KCH * 64 : (KQS ? Hp : 4 * Hp) * ELT);       // LDS row pitch of the delta tile
    const int nw = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    // LDS tile rows: the resident kernels keep the full 16-row MFMA operand tile; the streaming fallback (W_rec read
    // from L2 every step, HP = 0: large layers) keeps the 4*RPL real rows plus ONE shared zero row, which is what
    // lets a 4*Hp-wide fp32 / bf16 delta row of 8 KB (Hp = 512 fp32, Hp = 1024 bf16) fit the 160 KB LDS twice
    constexpr int TROWS = RES ? 16 : 4 * RPL + 1;
    [[maybe_unused]] const int rrow = RES ? c : ((c & 3) < RPL ? (c >> 2) * RPL + (c & 3) : 4 * RPL);
    const int d = blockIdx.x % p.dirs, s0 = (blockIdx.x / p.dirs) * (4 * RPL);
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    const long arow = (long)dirs * 4 * Hp;
    const long crow = (long)dirs * Hp;

    const int plane = TROWS * pitch;
    for (int i = threadIdx.x * 4; i < 2 * PLANES * plane; i += blockDim.x * 4) *(unsigned *)(smem + i) = 0u;
    // dummy-slot table of this workgroup's sequences, read per step from LDS (LstmLayer.cu:224-234 with
    // checkPatType of :949,983); the forward kernel stages the pattern type through its register prefetch
    // instead, which measured faster there (0.47 vs 0.49 us per step) and slower here
    unsigned char *dtab = (unsigned char *)smem + 2 * PLANES * plane;
    build_dummy_table<RPL>(dtab, p.pat, T, p.Tmin, p.PS, (blockIdx.x / p.dirs) * (4 * RPL));

    int unit[UG];
    float pi[UG], pf[UG], po[UG];
    [[maybe_unused]] u32x4 wreg[UG][SP ? 1 : KCR];
    [[maybe_unused]] u32x4 wlo[X3 && !SP ? UG : 1][SP ? 1 : KCR];
    [[maybe_unused]] u32x8 wsp[KCS], wsl[SP && X3 ? KCS : 1];
    [[maybe_unused]] const int spidx = sp_index(c);
    const char *Wd = (const char *)p.WrecT + (long)d * 4 * Hp * Hp * MELT;
#pragma unroll
    for (int u = 0; u < UG; ++u) {
        unit[u] = 16 * (wave + u * nw) + c;
        pi[u] = p.peep[(d * 3 + 0) * Hp + unit[u]];
        pf[u] = p.peep[(d * 3 + 1) * Hp + unit[u]];
        po[u] = p.peep[(d * 3 + 2) * Hp + unit[u]];
        if constexpr (SP) {
#pragma unroll
            for (int kc = 0; kc < KCS; ++kc) {
                const long w0 = (long)unit[u] * 4 * Hp + kc * 64 + q * 16;             // 16 consecutive K values per lane
                if constexpr (X3) sp_load_split((const float *)Wd + w0, wsp[kc], wsl[kc]);
                else wsp[kc] = sp_load_bf16(Wd + w0 * 2);
            }
        } else if constexpr (RES) {
#pragma unroll
            for (int kc = 0; kc < KCR; ++kc) {
                if constexpr (X3) {
                    const float *wp = (const float *)Wd + (long)unit[u] * 4 * Hp + kc * 32 + q * 8;
                    split8(*(const f32x4 *)wp, *(const f32x4 *)(wp + 4), wreg[u][kc], wlo[u][kc]);
                } else
                    wreg[u][kc] = *(const u32x4 *)(Wd + ((long)unit[u] * 4 * Hp) * ELT + kc * 64 + q * 16);
            }
        }
    }

    // uniform base + 32-bit offsets, as in the forward kernel
    unsigned oA[UG][RPL], oC[UG][RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int sv = s0 + 4 * r + q;
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            oA[u][r] = (unsigned)(sv * (int)arow + (d * Hp + unit[u]) * 4);          // elements into an acts / delta row block
            oC[u][r] = (unsigned)(sv * (int)crow + d * Hp + unit[u]);                // elements into a cell / err row block
        }
    }
    const unsigned stepA = (unsigned)PS * (unsigned)arow, stepC = (unsigned)PS * (unsigned)crow;   // elements per time step
    // SP: tile position of this lane's deltas.  K index k = 4*unit + gate, so the gates (n, i) of a unit are two neighbouring
    // stored values of the even row of its sequence and (f, o) the same two positions of the odd row.
    [[maybe_unused]] int oT[RPL];
    if constexpr (SP) {
        const int uh = KHS ? unit[0] % (Hp / 2) : unit[0], half = KHS ? unit[0] / (Hp / 2) : 0;
#pragma unroll
        for (int r = 0; r < RPL; ++r)
            oT[r] = (4 * q + 2 * (KHS ? half : r)) * pitch + ((uh >> 4) * 32 + sp_pos(4 * (uh & 15))) * 2;
    }

    // carried across steps (values of the step processed just before = next(t) in time)
    float fgn[UG][RPL], ecn[UG][RPL], dign[UG][RPL], dfgn[UG][RPL], ccur[UG][RPL];
    // gradient sums: bias (4 gates) and peepholes (i, f, o)
    float sb[UG][4], spi[UG], spf[UG], spo[UG];
#pragma unroll
    for (int u = 0; u < UG; ++u) {
        spi[u] = spf[u] = spo[u] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) sb[u][g] = 0.f;
#pragma unroll
        for (int r = 0; r < RPL; ++r) fgn[u][r] = ecn[u][r] = dign[u][r] = dfgn[u][r] = 0.f;
    }

    // processing order is the reverse of the forward pass of this direction
    const int tfirst = d ? 0 : T - 1;
    BwdPre<UG, RPL> preA, preB;
    auto prefetch = [&](int t, BwdPre<UG, RPL> &pre) {
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);         // unconditional, clamped (see the forward kernel)
        const int tprev = d ? t + 1 : t - 1;          // prev(t) in the forward processing order
        const bool hasprev = tprev >= 0 && tprev < T; // lastCall, LstmLayer.cu:947,981
        const unsigned bA = (unsigned)t * stepA, bC = (unsigned)t * stepC;
        const unsigned bCp = (unsigned)(hasprev ? tprev : t) * stepC;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                pre.e[u][r] = at32<float>(p.err, bC + oC[u][r]);
                pre.a[u][r] = *(const f32x4 *)&at32<float>(p.acts, bA + oA[u][r]);
                pre.cp[u][r] = at32<float>(p.cell, bCp + oC[u][r]);     // raw; masked with lastCall when consumed (no wait here)
                if constexpr (CN_TH_STORE) pre.th[u][r] = at32<float>(p.th, bC + oC[u][r]);
            }
        }
    };

    STAMP_DECL
    auto step = [&](int it, BwdPre<UG, RPL> &pre) {
        const int t = d ? it : T - 1 - it;
        const char *dcur = smem + (it & 1) * PLANES * plane;
        char *dnxt = smem + ((it + 1) & 1) * PLANES * plane;
        const int tprev_ = d ? t + 1 : t - 1;
        const bool hasprev_ = tprev_ >= 0 && tprev_ < T;       // !lastCall, LstmLayer.cu:947,981
        const unsigned bD = (unsigned)t * stepA;


        f32x4 acc[UG];
        f32x4 a_[UG][RPL];
        float cp_[UG][RPL];
        [[maybe_unused]] float th_[UG][RPL];
        unsigned char dmy[RPL];                      // dummy-slot flags of this step (LDS table, see the forward kernel)
#pragma unroll
        for (int r = 0; r < RPL; ++r) dmy[r] = dtab[t * (4 * RPL) + 4 * r + q];
        // The stage moves to registers of its own through early-clobber asm copies (as in the forward kernel), so
        // the prefetch below lands in the stage registers again.  Left to the register coalescer the loads went
        // elsewhere and were copied back at the loop latch behind `s_waitcnt vmcnt(3)`, i.e. every second step
        // waited for loads issued half a step earlier: free when they hit the Infinity Cache (the top layer,
        // 217 us per launch), ~300 cycles from HBM (252 us), more beside the gradient GEMMs (291 us).
#pragma unroll
        for (int u = 0; u < UG; ++u) {
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                float e_, c_;
                asm volatile("v_mov_b32 %0, %1" : "=&v"(e_) : "v"(pre.e[u][r]));
                asm volatile("v_mov_b32 %0, %1" : "=&v"(c_) : "v"(pre.cp[u][r]));
#pragma unroll
                for (int g = 0; g < 4; ++g) asm volatile("v_mov_b32 %0, %1" : "=&v"(a_[u][r][g]) : "v"(pre.a[u][r][g]));
                if constexpr (CN_TH_STORE) asm volatile("v_mov_b32 %0, %1" : "=&v"(th_[u][r]) : "v"(pre.th[u][r]));
                acc[u][r] = e_;                                                             // err enters as the MFMA C operand
                cp_[u][r] = hasprev_ ? c_ : 0.f;
            }
#pragma unroll
            for (int r = RPL; r < 4; ++r) acc[u][r] = 0.f;
        }
        prefetch(d ? t + 2 : t - 2, pre);
        STAMP(0)

        // BPTT product (LstmLayer.cu:939-942 / :973-976): the four gates contract into one K = 4*Hp
        if constexpr (SP) {
            u32x4 a[KCH];
            [[maybe_unused]] u32x4 al[X3 ? KCH : 1];
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) {
                a[kc] = *(const u32x4 *)(dcur + c * pitch + kc * 64 + q * 16);
                if constexpr (X3) al[kc] = *(const u32x4 *)(dcur + plane + c * pitch + kc * 64 + q * 16);
            }
#ifdef CN_STAMP
            STAMP_FORCE(a[0][0]) STAMP(1)
#endif
            if constexpr (KHS) {
                f32x4 accA = {acc[0][0], 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};      // err enters as the C operand
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc) {
                    if constexpr (X3) {
                        smma16_x3(accA, a[kc], al[kc], wsp[kc], wsl[kc], spidx);
                        smma16_x3(accB, a[kc], al[kc], wsp[KCH + kc], wsl[KCH + kc], spidx);
                    } else {
                        smma16(accA, a[kc], wsp[kc], spidx);
                        smma16(accB, a[kc], wsp[KCH + kc], spidx);
                    }
                }
                acc[0][0] = (accA[0] + accA[1]) + (accB[2] + accB[3]);
                KEEP_TUPLE(accA, acc[0][0]); KEEP_TUPLE(accB, acc[0][0]);
            } else {
                f32x4 accs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < RPL; ++r) accs[2 * r] = acc[0][r];
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc) {
                    if constexpr (X3) smma16_x3(accs, a[kc], al[kc], wsp[kc], wsl[kc], spidx);
                    else smma16(accs, a[kc], wsp[kc], spidx);
                }
#pragma unroll
                for (int r = 0; r < RPL; ++r) acc[0][r] = accs[2 * r] + accs[2 * r + 1];
                KEEP_TUPLE(accs, acc[0][RPL - 1]);
            }
        } else if constexpr (KQS) {
            f32x4 accq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) accq[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            accq[0][0] = acc[0][0];                  // err enters as the C operand of quarter 0
            u32x4 a[KCQ];
            [[maybe_unused]] u32x4 al[X3 ? KCQ : 1];
#pragma unroll
            for (int kc = 0; kc < KCQ; ++kc) {
                a[kc] = *(const u32x4 *)(dcur + c * pitch + kc * 64 + q * 16);
                if constexpr (X3) al[kc] = *(const u32x4 *)(dcur + plane + c * pitch + kc * 64 + q * 16);
            }
#ifdef CN_STAMP
            STAMP_FORCE(a[0][0]) STAMP(1)
#endif
#pragma unroll
            for (int kc = 0; kc < KCQ; ++kc)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (X3) mma16_x3(accq[r], a[kc], al[kc], wreg[0][r * KCQ + kc], wlo[0][r * KCQ + kc]);
                    else mma16<F32>(accq[r], a[kc], wreg[0][r * KCQ + kc]);
                }
            acc[0][0] = (accq[0][0] + accq[1][1]) + (accq[2][2] + accq[3][3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) KEEP_TUPLE(accq[r], acc[0][0]);
        } else if constexpr (RES) {
            // A-operand reads run LDS_AHEAD chunks ahead of the MFMAs that consume them.  Left alone, the
            // scheduler issues each read one chunk ahead, so every MFMA group waits most of an LDS round trip
            // (16 of them per step); reading everything up front costs 64 VGPRs and pushes W_rec into AGPRs.
            constexpr int LDS_AHEAD = KCR < 4 ? KCR : 4;
            constexpr int MPC = (X3 ? 3 : (F32 ? 4 : 1)) * UG;      // MFMAs per K chunk
            u32x4 a[KCR];
            [[maybe_unused]] u32x4 al[X3 ? KCR : 1];
#pragma unroll
            for (int kc = 0; kc < KCR; ++kc) {
                a[kc] = *(const u32x4 *)(dcur + c * pitch + kc * 64 + q * 16);
                if constexpr (X3) al[kc] = *(const u32x4 *)(dcur + plane + c * pitch + kc * 64 + q * 16);
            }
#ifdef CN_STAMP
            STAMP_FORCE(a[0][0]) STAMP(1)
#endif
#pragma unroll
            for (int kc = 0; kc < KCR; ++kc) {
#pragma unroll
                for (int u = 0; u < UG; ++u) {
                    if constexpr (X3) mma16_x3(acc[u], a[kc], al[kc], wreg[u][kc], wlo[u][kc]);
                    else mma16<F32>(acc[u], a[kc], wreg[u][kc]);
                }
            }
#ifndef CN_STAMP
            __builtin_amdgcn_sched_group_barrier(0x100, PLANES * LDS_AHEAD, 0);
#pragma unroll
            for (int kc = 0; kc < KCR; ++kc) {
                __builtin_amdgcn_sched_group_barrier(0x008, MPC, 0);
                if (kc + LDS_AHEAD < KCR) __builtin_amdgcn_sched_group_barrier(0x100, PLANES, 0);
            }
#endif
        } else {
            for (int kc = 0; kc < KC; ++kc) {
                const u32x4 a = *(const u32x4 *)(dcur + rrow * pitch + kc * 64 + q * 16);
                [[maybe_unused]] u32x4 al;
                if constexpr (X3) al = *(const u32x4 *)(dcur + plane + rrow * pitch + kc * 64 + q * 16);
#pragma unroll
                for (int u = 0; u < UG; ++u) {
                    if constexpr (X3) {
                        u32x4 b, bl;
                        const float *wp = (const float *)Wd + (long)unit[u] * 4 * Hp + kc * 32 + q * 8;
                        split8(*(const f32x4 *)wp, *(const f32x4 *)(wp + 4), b, bl);
                        mma16_x3(acc[u], a, al, b, bl);
                    } else {
                        const u32x4 b = *(const u32x4 *)(Wd + ((long)unit[u] * 4 * Hp) * ELT + kc * 64 + q * 16);
                        mma16<F32>(acc[u], a, b);
                    }
                }
            }
        }

        STAMP_FORCE(acc[UG - 1][0]) STAMP(2)
#pragma unroll
        for (int u = 0; u < UG; ++u) {
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const bool dummy = dmy[r] != 0;
                // ComputeBlockErrorsFn, LstmLayer.cu:236-285
                const float e = acc[u][r];
                const float ni = a_[u][r][0], ig = a_[u][r][1], fg = a_[u][r][2], og = a_[u][r][3];
                const float cs = ccur[u][r], cp = cp_[u][r];
                const float th = CN_TH_STORE ? th_[u][r] : tanh_ref<ACC>(cs);      // (the forward pass's value, same function of the same cs)
                float dog = og * (1.0f - og) * th * e;
                if constexpr (!SP && !KQS) { if (r == RPL - 1) KEEP_TUPLE(acc[u], dog); }      // (the dense tile's padding rows)
                float ec = og * (1.0f - th * th) * e + po[u] * dog;
                ec += fgn[u][r] * ecn[u][r] + pi[u] * dign[u][r] + pf[u] * dfgn[u][r];   // zero carry at firstCall
                float dni = ig * (1.0f - ni * ni) * ec;
                float dfg = fg * (1.0f - fg) * cp * ec;                                     // cp = 0 at lastCall
                float dig = ig * (1.0f - ig) * ni * ec;
                dni = clip1(dni); dig = clip1(dig); dfg = clip1(dfg); dog = clip1(dog);
                // :224-234, as selects: a branch here splits the step into basic blocks and fences the scheduler in
                dni = dummy ? 0.f : dni; dig = dummy ? 0.f : dig; dfg = dummy ? 0.f : dfg; dog = dummy ? 0.f : dog;
                ec = dummy ? 0.f : ec;
#ifdef CN_STAMP
                if (u == UG - 1 && r == RPL - 1) { STAMP_FORCE(dog) STAMP(3) }
#endif
                fgn[u][r] = dummy ? 0.f : fg;
                ecn[u][r] = ec; dign[u][r] = dig; dfgn[u][r] = dfg;
                ccur[u][r] = cp;
                // gradient sums (ComputeWeightUpdateFn bias / peephole cases, :392-408, :440-475)
                sb[u][0] += dni; sb[u][1] += dig; sb[u][2] += dfg; sb[u][3] += dog;
                spi[u] += cp * dig; spf[u] += cp * dfg; spo[u] += cs * dog;
                // tile position of this lane's deltas: row of its sequence (KQS: row 4q + K-quarter of its unit), column of its unit
                const int trow = KQS ? 4 * q + unit[u] / (Hp / 4) : (RES ? 4 * q + r : q * RPL + r);
                const int tcol = KQS ? unit[u] % (Hp / 4) : unit[u];
                if constexpr (F32) {
                    const f32x4 dv = {dni, dig, dfg, dog};
                    *(f32x4 *)(dnxt + trow * pitch + tcol * 16) = dv;
                    *(f32x4 *)&at32<float>(p.delta_op, bD + oA[u][r]) = dv;
                } else if constexpr (SP && X3) {
                    const f32x4 dv = {dni, dig, dfg, dog};
                    bf16x4 dh, dl;
#pragma unroll
                    for (int g = 0; g < 4; ++g) { __bf16 h_, l_; split_bf16(dv[g], h_, l_); dh[g] = h_; dl[g] = l_; }
                    *(bf16x2 *)(dnxt + oT[r]) = bf16x2{dh[0], dh[1]};
                    *(bf16x2 *)(dnxt + oT[r] + pitch) = bf16x2{dh[2], dh[3]};
                    *(bf16x2 *)(dnxt + plane + oT[r]) = bf16x2{dl[0], dl[1]};
                    *(bf16x2 *)(dnxt + plane + oT[r] + pitch) = bf16x2{dl[2], dl[3]};
                    *(f32x4 *)&at32<float>(p.delta_op, bD + oA[u][r]) = dv;
                } else if constexpr (SP) {
                    const bf16x4 dv = {(__bf16)dni, (__bf16)dig, (__bf16)dfg, (__bf16)dog};
                    *(bf16x2 *)(dnxt + oT[r]) = bf16x2{dv[0], dv[1]};
                    *(bf16x2 *)(dnxt + oT[r] + pitch) = bf16x2{dv[2], dv[3]};
                    *(bf16x4 *)&at32<__bf16>(p.delta_op, bD + oA[u][r]) = dv;
                } else if constexpr (X3) {
                    const f32x4 dv = {dni, dig, dfg, dog};
                    bf16x4 dh, dl;
#pragma unroll
                    for (int g = 0; g < 4; ++g) { __bf16 h_, l_; split_bf16(dv[g], h_, l_); dh[g] = h_; dl[g] = l_; }
                    *(bf16x4 *)(dnxt + trow * pitch + tcol * 8) = dh;
                    *(bf16x4 *)(dnxt + plane + trow * pitch + tcol * 8) = dl;
                    *(f32x4 *)&at32<float>(p.delta_op, bD + oA[u][r]) = dv;
                } else {
                    const bf16x4 dv = {(__bf16)dni, (__bf16)dig, (__bf16)dfg, (__bf16)dog};
                    *(bf16x4 *)(dnxt + trow * pitch + tcol * 8) = dv;
                    *(bf16x4 *)&at32<__bf16>(p.delta_op, bD + oA[u][r]) = dv;
                }
            }
        }
        STAMP(4)
        lds_barrier();
        STAMP(5)
    };

#pragma unroll
    for (int r = 0; r < RPL; ++r)
#pragma unroll
        for (int u = 0; u < UG; ++u)
            ccur[u][r] = at32<float>(p.cell, (unsigned)tfirst * stepC + oC[u][r]);
    prefetch(tfirst, preA);
    prefetch(d ? 1 : T - 2, preB);
    lds_barrier();
    STAMP_RESET
    // branch-free pairs of steps, first pair peeled (see the forward kernel)
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
    STAMP_STORE(1)

    // fold the 4 sequence quads of each unit column, then one atomic per (gate, unit) and workgroup
#pragma unroll
    for (int u = 0; u < UG; ++u) {
        float v[7] = {sb[u][0], sb[u][1], sb[u][2], sb[u][3], spi[u], spf[u], spo[u]};
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            v[i] += __shfl_xor(v[i], 16);
            v[i] += __shfl_xor(v[i], 32);
        }
        if (q == 0) {
            lstm_grad_sums_out(p, Hp, d, unit[u], v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <int PREC, bool BWD, int HP, int UG, int RPL>
static void launch_one(hipStream_t s, const LstmRec &p, int nwaves, hipEvent_t done)
{
    const int ELT = PREC == P_F32 ? 4 : 2, PLANES = PREC == P_X3 ? 2 : 1;
    const int nsg = p.PS / (4 * RPL);                // PS is padded to whole sequence groups
    const int pitch = lds_pitch((BWD ? 4 : 1) * p.Hp * ELT);
    const size_t lds = 2 * (size_t)PLANES * (HP ? 16 : 4 * RPL + 1) * pitch + (BWD ? (((size_t)p.T * 4 * RPL + 15) & ~(size_t)15) : 0);   // tiles (+ dummy-slot table)
    auto kern = BWD ? lstm_bwd_kernel<PREC, HP, UG, RPL> : lstm_fwd_kernel<PREC, HP, UG, RPL>;
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    // When the grid leaves CUs free (one workgroup per CU at most), each workgroup claims the CU's whole LDS so that
    // no workgroup of a concurrently running kernel (the gradient GEMMs of the side stream) can be placed beside
    // it: sharing its SIMDs' issue slots with GEMM waves cost the latency-bound recurrent kernel 27 % (291 vs
    // 229 us per backward launch in the step timeline); the GEMMs lose 26 of 256 CUs instead.
    size_t lds_claim = lds;
    if (p.dirs * nsg <= 128 && !opt().no_lds_claim) lds_claim = 160 * 1024 - 1024;
    lstm_note_grid(p, p.dirs * nsg);
    hipExtLaunchKernelGGL(kern, dim3(p.dirs * nsg), dim3(64 * nwaves), lds_claim < lds ? lds : lds_claim, s, nullptr, done, 0, p);
    if (p.kname) snprintf(p.kname, CN_KNAME_LEN, "lstm_%s_kernel<%d,%d,%d,%d>", BWD ? "bwd" : "fwd", PREC, HP, UG, RPL);
}

void lstm_note_grid(const LstmRec &p, int grid)
{
    if (p.det_grid) *p.det_grid = grid;
    if (p.gpart && grid > p.gpart_slots)
        throw std::runtime_error("deterministic mode: a recurrent kernel of " + std::to_string(grid) + " workgroups, " + std::to_string(p.gpart_slots) + " partial-sum slots");
}

template <int PREC, bool BWD, int HP, int UG>
static void launch_rpl(hipStream_t s, const LstmRec &p, int nwaves, hipEvent_t done)
{
    if (p.rpl == 1)      launch_one<PREC, BWD, HP, UG, 1>(s, p, nwaves, done);
    else if (p.rpl == 2) launch_one<PREC, BWD, HP, UG, 2>(s, p, nwaves, done);
    else               launch_one<PREC, BWD, HP, UG, 4>(s, p, nwaves, done);
}

template <int PREC, bool BWD>
static void launch_rec(hipStream_t s, const LstmRec &p, hipEvent_t done = nullptr)
{
    constexpr bool F32 = PREC != P_BF16;             // (register budget of the fragments: P_X3 = P_F32)
    const int groups = p.Hp / 16;
    switch (p.Hp) {
    case 32:  launch_rpl<PREC, BWD, 32, 1>(s, p, 2, done); return;
    case 64:  launch_rpl<PREC, BWD, 64, 1>(s, p, 4, done); return;
    case 96:  launch_rpl<PREC, BWD, 96, 1>(s, p, 6, done); return;
    case 128:
        // 8 waves x 1 unit group (two waves per SIMD overlap each other's MFMA, VALU and scalar issue).  For the
        // backward kernel 4 waves x 2 unit groups (half the LDS operand reads: every wave reads the whole
        // 16 x 4Hp delta tile) used to win by 4 %; since the stage copies removed the latch stall it loses:
        // 0.64 vs 0.56 us per step (CN_BWD_UG2 keeps it selectable)
        if constexpr (PREC == P_BF16) {
            if (BWD && opt().bwd_ug2) { launch_rpl<PREC, BWD, 128, 2>(s, p, 4, done); return; }
            if (!BWD && opt().fwd_ug2) { launch_rpl<PREC, BWD, 128, 2>(s, p, 4, done); return; }   // measured slower: 0.63 vs 0.47 us per step
        }
        launch_rpl<PREC, BWD, 128, 1>(s, p, 8, done);
        return;
    case 160:     // bf16 only: 200 / 288 KB of W_rec still fit one CU's registers (10 / 12 waves, three on some SIMDs)
        if constexpr (!F32) { launch_rpl<PREC, BWD, 160, 1>(s, p, 10, done); return; }
        break;
    case 192:
        if constexpr (!F32) { launch_rpl<PREC, BWD, 192, 1>(s, p, 12, done); return; }
        break;
    default: break;
    }
    if (groups <= 16)      launch_rpl<PREC, BWD, 0, 1>(s, p, groups, done);
    else if (groups <= 32) launch_rpl<PREC, BWD, 0, 2>(s, p, groups / 2, done);
    else                   launch_rpl<PREC, BWD, 0, 4>(s, p, groups / 4, done);
}

#ifdef CN_STAMP
extern "C" int cn_dbg_read_stamps(unsigned long long *host)      // [2][16][8]
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cn_stamp_buf), sizeof(cn_stamp_buf));
}
#endif

// dynamic LDS of one workgroup of the single-CU kernels (launch_one): two operand tiles (+ the backward kernel's
// dummy-slot table); resident shapes are those launch_rec dispatches on
bool lstm_rec_resident(int prec, int Hp)
{
    return Hp == 32 || Hp == 64 || Hp == 96 || Hp == 128 || (prec == P_BF16 && (Hp == 160 || Hp == 192));
}
size_t lstm_rec_lds_bytes(int prec, bool bwd, int Hp, int rpl, int T)
{
    const int ELT = prec == P_F32 ? 4 : 2, PLANES = prec == P_X3 ? 2 : 1;
    const bool resident = lstm_rec_resident(prec, Hp);
    const size_t pitch = (size_t)lds_pitch((bwd ? 4 : 1) * Hp * ELT);
    return 2 * (size_t)PLANES * (resident ? 16 : 4 * rpl + 1) * pitch + (bwd ? (((size_t)T * 4 * rpl + 15) & ~(size_t)15) : 0);
}

bool lstm_fwd_takes_pre16(int prec, const LstmRec &p)
{
    // the two-sequence kernels of cn_lstm_s2.hip (hand-written loop and its compiled twin), bf16 mode
    return prec == P_BF16 && opt().pre16 && !lstm_s2w_applies(prec, p, false) && lstm_s2_applies(prec, p, false);
}

void launch_lstm_forward(hipStream_t s, int prec, const LstmRec &p)
{
    if (lstm_s2w_applies(prec, p, false)) { launch_lstm_s2w(s, false, p); return; }
    if (lstm_s2_applies(prec, p, false)) { launch_lstm_s2(s, prec, false, p); return; }
    if (prec == P_F32) launch_rec<P_F32, false>(s, p);
    else if (prec == P_X3) launch_rec<P_X3, false>(s, p);
    else launch_rec<P_BF16, false>(s, p);
}
void launch_lstm_backward(hipStream_t s, int prec, const LstmRec &p, hipEvent_t done)
{
    if (lstm_s2_applies(prec, p, true)) { launch_lstm_s2(s, prec, true, p, done); return; }
    if (prec == P_F32) launch_rec<P_F32, true>(s, p, done);
    else if (prec == P_X3) launch_rec<P_X3, true>(s, p, done);
    else launch_rec<P_BF16, true>(s, p, done);
}

}  // namespace cn

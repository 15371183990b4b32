// Recurrent LSTM kernels for layers whose W_rec does not fit one CU: a CLUSTER of CUs per (direction,
// sequence group), each CU keeping the W_rec fragments of its own slice of hidden units in registers for the
// whole pass and exchanging only y[t] (forward) / the four deltas (backward) of its units with its partners
// once per time step.
//
// Why: with Hp = 256 (the Graves ASRU'13 reading of "3x250 BLSTM", CURRENNT "size": 500) the bf16 W_rec of one
// direction is 512 KB -- more than the 160 KB LDS or the usable part of the 512 KB register file of a CU.
// Streaming it from L2 every step (cn_lstm.hip, HP = 0) costs ~17 us per step; the hand-off below costs ~1 us.
//
// Hand-off (cdna_hip_programming.md Guideline 16, form R2 "the data IS the flag"): every lane publishes its
// value as ONE naturally aligned 8-byte granule {tag = step + 1, value} with a relaxed agent-scope atomic store
// (sc1, write-through); the consumer re-reads the granule with relaxed agent-scope atomic loads (sc1, bypasses
// the per-CU L1) until the tag matches -- no fences, placement independent.  Granule slots alternate with the
// step parity (tags count on across launches, LstmRec::xch_epoch): a producer can only overwrite slot p at step t+2 after it has consumed its partners' step t+1,
// which they published after consuming its step t from that very slot.  Tags count on across launches (the buffer is cleared
// only at allocation and long before the 32-bit tags would wrap); every spin is bounded and reports through a fault word instead
// of hanging.
// Cluster members sit 8 block ids apart (same XCD under round-robin placement: speed only, never correctness);
// the launcher only uses this path when the whole grid is resident (<= one workgroup per CU of the device, whose CU count
// it is given), and launches of different contexts on one device are serialised (cluster_gate) so that two half-resident
// grids cannot wait for each other.
//
// Arithmetic is identical to cn_lstm.hip (same MFMA tiles -- the 2:4 row-pair sparse products included --, same cell update).
// Modes: CN_PREC_BF16 (Hp = 256 as 2 x 128 units, Hp = 512 as 8 x 64) and CN_PREC_BF16X3 (Hp = 256 as 4 x 64: hi and lo fragments).
#include "cn_internal.h"
#include "cn_lstm_device.h"

#include <cstdio>

#ifndef CN_KQ_STACK
#define CN_KQ_STACK 1
#endif
#ifndef CN_SPARSE
#define CN_SPARSE 1
#endif
#ifndef CN_TH_STORE
#define CN_TH_STORE 1
#endif
#ifndef CN_POLL2
#define CN_POLL2 0      // round 4: two polls in flight measured 3-5 % SLOWER per step than one (A.5: every extra sample costs more than it finds)
#endif
#ifndef CN_KHS_READ_AHEAD
#define CN_KHS_READ_AHEAD 1
#endif
#ifndef CN_POLL_DELAY_MANY
#define CN_POLL_DELAY_MANY 2
#endif
#ifndef CN_POLL_DELAY_PSUM
#define CN_POLL_DELAY_PSUM 8        // partial-sum backward kernel (bf16x3, 4 CUs): reading B at tolerance, rec_bwd 12.63 -> 12.02 ms per six fractions (4: 12.25, 6: 12.10, 9: 12.02, 12: 12.17)
#endif
#ifndef CN_POLL_DELAY_7
#define CN_POLL_DELAY_7 1
#endif
#ifndef CN_BWD_READ_FENCE
#define CN_BWD_READ_FENCE 1
#endif
#ifndef CN_FWD_READ_AHEAD
#define CN_FWD_READ_AHEAD 1
#endif
#ifndef CN_POLL_SLEEP
#define CN_POLL_SLEEP 1
#endif
#ifndef CN_POLL2_STAGGER
#define CN_POLL2_STAGGER 4
#endif
#ifndef CN_PACKED_XCH
#define CN_PACKED_XCH 1      // round 6: the 8-CU bf16 backward kernel exchanges ONE granule per lane and partner (four bf16 deltas with the tag in their spare exponent bits)
#endif
#ifndef CN_POLL_DELAY_PACKED
#define CN_POLL_DELAY_PACKED 0     // swept on the long-utterance workload (ms per fraction, two rounds): 0 30.11 / 30.13, 1 30.24 / 30.31, 2 30.42 / 30.47
#endif

#include <cstdlib>
#include <map>
#include <type_traits>
#include <mutex>

#define KEEP_TUPLE(tuple, after) asm volatile("" :: "v"(tuple), "v"(after))

// In-kernel segment timing of the delta-exchange backward kernel (tools/stamps_cl.py; `make variantc NAME=clstamp DEFS=-DCN_CL_STAMP`;
// never defined in the shipped build): wave w of cluster 0, member 0 sums s_memtime deltas per step segment
#ifdef CN_CL_STAMP
__device__ unsigned long long cn_cl_stamp_buf[8][8], cn_cl_stamp_buf_f[8][8];
extern "C" int cn_dbg_read_stamps_cl(unsigned long long *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cn_cl_stamp_buf), sizeof(cn_cl_stamp_buf)); }
extern "C" int cn_dbg_read_stamps_cl_fwd(unsigned long long *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cn_cl_stamp_buf_f), sizeof(cn_cl_stamp_buf_f)); }
#define CLS_STORE_F if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 8; ++i_) cn_cl_stamp_buf_f[threadIdx.x >> 6][i_] = st_acc[i_]; }
#define CLS_DECL unsigned long long st_prev = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define CLS(i) { unsigned long long st_now = __builtin_amdgcn_s_memtime(); st_acc[i] += st_now - st_prev; st_prev = st_now; }
#define CLS_FORCE(x) asm volatile("v_mov_b32 %0, %0" : "+v"(x));
#define CLS_STORE if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 8; ++i_) cn_cl_stamp_buf[threadIdx.x >> 6][i_] = st_acc[i_]; }
#else
#define CLS_DECL
#define CLS(i)
#define CLS_FORCE(x)
#define CLS_STORE
#define CLS_STORE_F
#endif

namespace cn {

typedef unsigned long long u64;

#ifndef CN_POLL_RMW
#define CN_POLL_RMW 0
#endif
// K samples of granules, all in flight together: relaxed agent-scope loads (sc1), or -- CN_POLL_RMW -- returning atomic ORs of zero,
// which execute in the L2 (hipcc folds an idempotent fetch_or into a load, so the instructions are written out: issued without a
// wait, then ONE wait, then every result is re-defined behind it so that no use can be scheduled in front of the wait)
template <int K>
__device__ __forceinline__ void sample_all(const u64 *const (&slot)[K], u64 (&x)[K])
{
#if CN_POLL_RMW
    const u64 zero = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0" : "=v"(x[i]) : "v"(slot[i]), "v"(zero) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < K; ++i) asm volatile("" : "+v"(x[i]));
#else
#pragma unroll
    for (int i = 0; i < K; ++i) x[i] = __hip_atomic_load(slot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void publish(u64 *slot, unsigned epoch, unsigned value)
{
    __hip_atomic_store(slot, ((u64)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned consume(const u64 *slot, unsigned epoch, int *fault)
{
    u64 x;
    int spins = 0;
    for (;;) {
        x = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) == epoch) break;
        if (++spins > (1 << 21)) { *fault = 1; break; }      // ~1 s: a partner never arrived (not resident / faulted)
        __builtin_amdgcn_s_sleep(1);
    }
    return (unsigned)x;
}

// K granules at once: all loads are in flight together, so a member waits one L2 round trip for its partners' values
// instead of one per granule (7 partners x RPL granules in the 8-CU shape)
// A poll that times out (a partner never arrived: not resident, or faulted) sets the fault word AND `gaveup`: the thread
// does not wait again, so a broken launch ends after one time-out (~1 s) instead of one per time step.
template <int K, int DELAY = -1>
__device__ __forceinline__ void consume_all(const u64 *const (&slot)[K], unsigned epoch, int *fault, unsigned (&val)[K], bool &gaveup)
{
    int spins = gaveup ? (1 << 21) : 0;
    // Two polls in flight, half a round trip apart (2-CU clusters): a poll samples L2 once per round trip, so a value that lands
    // just behind a sample waits most of a round trip for the next one; the second, staggered poll halves that residual.
    // Measured: reading B backward kernel -3 %; with the 14 granules per thread of the 8-CU clusters the doubled poll traffic
    // costs 70 % (long-utterance config 37 -> 54 ms per fraction), so only for few granules.
    if constexpr (CN_POLL2 && K <= 4) {
    u64 xa[K], xb[K];
#pragma unroll
    for (int i = 0; i < K; ++i) xa[i] = __hip_atomic_load(slot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_sleep(CN_POLL2_STAGGER);
    for (;;) {
#pragma unroll
        for (int i = 0; i < K; ++i) xb[i] = __hip_atomic_load(slot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < K; ++i) { ok = ok && (unsigned)(xa[i] >> 32) == epoch; val[i] = (unsigned)xa[i]; }
        if (ok) break;
#pragma unroll
        for (int i = 0; i < K; ++i) xa[i] = __hip_atomic_load(slot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = true;
#pragma unroll
        for (int i = 0; i < K; ++i) { ok = ok && (unsigned)(xb[i] >> 32) == epoch; val[i] = (unsigned)xb[i]; }
        if (ok) break;
        if (++spins > (1 << 20)) { *fault = 1; gaveup = true; break; }
    }
    } else {
    // many granules per thread (8-CU shapes: 7 forward, 14 backward): a sample that leaves before the partners' values are in L2
    // costs a whole extra round trip, so the FIRST one waits s_sleep(n) = n x 64 cycles.  Swept on the long-utterance workload
    // (ms per fraction; backward n / forward n): 0 / 0 33.1; 2 / 2 31.85; 4 / 4 32.7; 6 / 6 33.8; 9 / 9 35.3; 2 / 1 31.71;
    // 1 / 1 31.70; 3 / 1 31.8; 2 / 3 32.0 -- two for the backward kernel's 14 granules, one for the forward kernel's 7.
    if constexpr (DELAY > 0) __builtin_amdgcn_s_sleep(DELAY);
    else if constexpr (DELAY < 0 && K > 8 && CN_POLL_DELAY_MANY > 0) __builtin_amdgcn_s_sleep(CN_POLL_DELAY_MANY);
    else if constexpr (DELAY < 0 && K > 4 && CN_POLL_DELAY_7 > 0) __builtin_amdgcn_s_sleep(CN_POLL_DELAY_7);
    for (;;) {
        u64 x[K];
        sample_all<K>(slot, x);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < K; ++i) { ok = ok && (unsigned)(x[i] >> 32) == epoch; val[i] = (unsigned)x[i]; }
        if (ok) break;
        if (++spins > (1 << 21)) { *fault = 1; gaveup = true; break; }
        if (CN_POLL_SLEEP) __builtin_amdgcn_s_sleep(CN_POLL_SLEEP);
    }
    }
}

// ---- packed granules (round 6; the 8-CU bf16 backward kernel) ------------------------------------------------------------------
// A lane's four deltas are bf16 values clipped to [-1, 1] (ComputeBlockErrorsFn, LstmLayer.cu:281-285): bit 14 -- the top bit of
// the exponent -- is 0 in every one of them.  Those four bits carry the tag, so the whole hand-off of a lane is ONE naturally
// aligned 8-byte granule instead of two {32-bit tag, 32-bit value} ones: half the polled bytes and lines (14 -> 7 granules per
// lane and step; the poll was 1 700-1 900 of the 4 100 cycles of a step and every extra sample cost more than it found: rec_bwd of
// the long-utterance step 66.5 -> 60.1 ms per four fractions, the fraction 31.7 -> 30.1 ms).  The same for the FORWARD kernel --
// y is below 1 too; the four lanes of a quad hand their values to the quad's first lane by DPP, which publishes and polls one
// granule for the four -- was built, is correct and measured 30 % SLOWER (rec_fwd 50.2 -> 66 ms, whatever the first sample's
// delay; no scratch): not kept.  Four
// bits do not count launches: tags are 1 + (step mod 15) -- two consecutive writes of a slot (two steps apart) always differ --,
// 0 is "nothing yet"; the packed slots live in a region of their own behind the 32-bit-tag schemes' (no other kernel ever
// interprets them), every member zeroes ITS slots when the launch begins, drains, and the cluster meets once through ordinary
// granules tagged with the launch's epoch (a value no step uses) before anybody reads a packed slot.
__device__ __forceinline__ void packed_masks(unsigned tag, unsigned &lo, unsigned &hi)
{
    lo = ((tag & 1u) << 14) | (((tag >> 1) & 1u) << 30);
    hi = (((tag >> 2) & 1u) << 14) | (((tag >> 3) & 1u) << 30);
}
template <int K, int DELAY>
__device__ __forceinline__ void consume_all_packed(const u64 *const (&slot)[K], unsigned tag, int *fault, unsigned (&lo)[K], unsigned (&hi)[K], bool &gaveup)
{
    unsigned elo, ehi;
    packed_masks(tag, elo, ehi);
    int spins = gaveup ? (1 << 21) : 0;
    // (the first sample waits DELAY x 64 cycles: one that leaves before the partners' granules are in L2 costs a whole round trip more)
    if constexpr (DELAY > 0) __builtin_amdgcn_s_sleep(DELAY);
    for (;;) {
        u64 x[K];
        sample_all<K>(slot, x);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const unsigned l = (unsigned)x[i], h = (unsigned)(x[i] >> 32);
            ok = ok && (l & 0x40004000u) == elo && (h & 0x40004000u) == ehi;
            lo[i] = l & ~0x40004000u; hi[i] = h & ~0x40004000u;
        }
        if (ok) break;
        if (++spins > (1 << 21)) { *fault = 1; gaveup = true; break; }
        if (CN_POLL_SLEEP) __builtin_amdgcn_s_sleep(CN_POLL_SLEEP);
    }
}

// Round 4: the poll's OWN round trip was the largest segment of a step (in-kernel stamps, tools/stamps_cl.py: ~1000 of the 3400
// cycles of a 2-CU backward step, ~1850 of 4800 in the 8-CU shape, although the partners' values had reached L2 long before the
// first sample came back): a poll issued behind the own part of the product only starts its trip then.  The samples are
// therefore taken at the TOP of the step -- the partners published at the end of their previous step, one hop (~270-320 ns,
// tools/probe/hop_probe.cpp) before -- and looked at behind the own part; a thread whose early samples missed falls back to
// the polling loop above.  Measured (variants of CN_EARLY_MASK / CN_POLL2, reading B per fraction): no early sample + two polls
// in flight (rounds 2-3) 2.928 ms; no early sample, one poll 2.836; ONE early sample behind the stage copies + one poll 2.796;
// at the top 2.837; behind the own part 2.841; two early samples 2.89; three 2.96: every extra sample costs more in L2 traffic
// than it finds.  Shipped: one early sample for threads that wait for at most two granules (2-CU bf16), none otherwise
// (split-bf16 4-CU: 5.01 ms without against 5.06 with; 8-CU: 7 / 14 granules per thread).
#ifndef CN_EARLY_MASK
#define CN_EARLY_MASK 2          // which of the three sample points of a step are taken (bit 0: top, 1: behind the stage copies, 2: behind the own part)
#endif
#ifndef CN_EARLY_MAXK
#define CN_EARLY_MAXK 2          // ... by threads that wait for at most this many granules (2-CU bf16 shapes)
#endif
template <int K> struct EarlyPoll { static constexpr bool ON = K <= CN_EARLY_MAXK; u64 s[ON ? 3 : 1][ON ? K : 1]; };
template <int N, int K>
__device__ __forceinline__ void poll_early(const u64 *const (&slot)[K], EarlyPoll<K> &e)
{
    if constexpr (EarlyPoll<K>::ON && ((CN_EARLY_MASK >> N) & 1)) {
#pragma unroll
        for (int i = 0; i < K; ++i) e.s[N][i] = __hip_atomic_load(slot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int K>
__device__ __forceinline__ void poll_finish(const u64 *const (&slot)[K], unsigned epoch, int *fault, unsigned (&val)[K], bool &gaveup, const EarlyPoll<K> &e)
{
    if constexpr (EarlyPoll<K>::ON) {
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            if (!((CN_EARLY_MASK >> n) & 1)) continue;
            bool ok = true;
#pragma unroll
            for (int i = 0; i < K; ++i) { ok = ok && (unsigned)(e.s[n][i] >> 32) == epoch; val[i] = (unsigned)e.s[n][i]; }
            if (ok) return;
        }
    }
    consume_all<K>(slot, epoch, fault, val, gaveup);
}

// two granules at once: both loads are in flight together (one L2 round trip instead of two when the data is there)
__device__ __forceinline__ uint2 consume2(const u64 *slot0, const u64 *slot1, unsigned epoch, int *fault)
{
    u64 x, y;
    int spins = 0;
    for (;;) {
        x = __hip_atomic_load(slot0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        y = __hip_atomic_load(slot1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) == epoch && (unsigned)(y >> 32) == epoch) break;
        if (++spins > (1 << 21)) { *fault = 1; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    return make_uint2((unsigned)x, (unsigned)y);
}

// block id -> (cluster, member): the CS members of a cluster are 8 ids apart
template <int CS>
__device__ __forceinline__ void cluster_of(int &cluster, int &member)
{
    member = (blockIdx.x / 8) % CS;
    cluster = (blockIdx.x / (8 * CS)) * 8 + blockIdx.x % 8;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// PREC: P_BF16, or P_X3 (fp32 in memory and in the exchange granules; two bf16 planes (hi, lo) of the tile in LDS, W_rec split once
// into hi and lo fragments, three sparse MFMAs per chunk -- cn_lstm.hip; 64-unit members only: the fragments take 256 VGPRs)
template <int PREC, int HP, int UPC, int RPL>
__global__ __launch_bounds__(UPC * 4) void lstm_fwd_cluster_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool X3 = PREC == P_X3;
    // QUAD (split-bf16, one sequence per lane): the two spare rows of a sequence's row quad carry the lo halves of y under the hi
    // halves (cn_lstm_s2.hip, "row quads"): [hi; lo] x W_lo + [hi; lo] x W_hi = TWO sparse MFMAs per chunk and gate instead of
    // three (and the lo x lo term on top), one tile plane instead of two; the sum is over all four accumulator registers.
    constexpr bool QUAD = X3 && RPL == 1;
    constexpr int MELT = X3 ? 4 : 2, PLANES = (X3 && !QUAD) ? 2 : 1;
    // SP: 2:4 row-pair products (cn_lstm_device.h): a sequence takes two tile rows, a K = 64 chunk is one sparse MFMA and the
    // tile rows are half as long; a member part is UPC / 64 chunks of 32 stored values per row
    constexpr bool SP = (CN_SPARSE || X3) && UPC % 64 == 0;
    static_assert(!X3 || SP, "the split-bf16 cluster kernels exist in the row-pair form only");
    constexpr int CS = HP / UPC, NT = UPC * 4, KC = SP ? HP / 64 : HP / 32;
    constexpr int pitch = lds_pitch(SP ? HP : HP * 2);
    int cluster, member;
    cluster_of<CS>(cluster, member);
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    if (cluster >= dirs * (PS / (4 * RPL))) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int d = cluster % dirs, s0 = (cluster / dirs) * (4 * RPL);
    const int lunit = 16 * wave + c, unit = member * UPC + lunit;      // unit inside the slice / the direction
    const long arow = (long)dirs * 4 * HP, crow = (long)dirs * HP;
    // tile position of y of local unit `lunit` of member part `part` (0 = own) for sequence r of lane quarter q
    auto tile_off = [&](int part, int r) {
        return SP ? (4 * q + 2 * r + sp_parity(lunit & 63)) * pitch + part * UPC + ((lunit >> 6) * 32 + sp_pos(lunit & 63)) * 2
                  : (4 * q + r) * pitch + (part * UPC + lunit) * 2;
    };
    [[maybe_unused]] const int spidx = sp_index(c);

    constexpr int plane = 16 * pitch;                // P_X3: hi plane, then lo plane
    for (int i = tid * 4; i < 2 * PLANES * plane; i += NT * 4) *(unsigned *)(smem + i) = 0u;
    bool gaveup = false;

    // K chunks in the order they are used: first the KCO chunks of this member's own units (their y is in LDS as soon
    // as the step starts), then the partners' (which have to cross L2 first).  kch[j] is the chunk behind wreg[.][j].
    constexpr int KCO = KC / CS;
    [[maybe_unused]] u32x4 wreg[4][SP ? 1 : KC];
    [[maybe_unused]] u32x8 wsp[4][SP ? KC : 1], wsl[4][X3 ? KC : 1];
    int kch[KC];
    const char *Wd = (const char *)p.Wrec + (long)d * 4 * HP * HP * MELT;
#pragma unroll
    for (int j = 0; j < KC; ++j) kch[j] = ((member + j / KCO) % CS) * KCO + j % KCO;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            if constexpr (X3) sp_load_split((const float *)Wd + (long)(g * HP + unit) * HP + kch[j] * 64 + q * 16, wsp[g][j], wsl[g][j]);
            else if constexpr (SP) wsp[g][j] = sp_load_bf16(Wd + ((long)(g * HP + unit) * HP + kch[j] * 64 + q * 16) * 2);
            else wreg[g][j] = *(const u32x4 *)(Wd + ((long)(g * HP + unit) * HP) * 2 + kch[j] * 64 + q * 16);
        }
    const float pi = p.peep[(d * 3 + 0) * HP + unit], pf = p.peep[(d * 3 + 1) * HP + unit], po = p.peep[(d * 3 + 2) * HP + unit];

    int oP[RPL], oA[RPL], oC[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int sv = s0 + 4 * r + q;
        oP[r] = sv; oA[r] = sv * (int)arow + (d * HP + unit) * 4; oC[r] = sv * (int)crow + d * HP + unit;
    }
    const long stepA = (long)PS * arow, stepC = (long)PS * crow;
    u64 *xbase = p.xch + (long)cluster * 2 * CS * (RPL * NT);

    float cst[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) cst[r] = 0.f;

    f32x4 preA[RPL], preB[RPL];
    char ptA[RPL], ptB[RPL];
    CLS_DECL
    auto prefetch = [&](int t, f32x4 (&pre)[RPL], char (&pt)[RPL]) {
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        const float *actsT = p.acts + t * stepA;
        const char *patT = p.pat + (long)t * PS;
#pragma unroll
        for (int r = 0; r < RPL; ++r) { pt[r] = patT[oP[r]]; pre[r] = *(const f32x4 *)(actsT + oA[r]); }
    };

    auto step = [&](int it, f32x4 (&pre)[RPL], char (&pt)[RPL]) {
        const int t = d ? T - 1 - it : it;
        const char *ycur = smem + (it & 1) * PLANES * plane;
        char *ynxt = smem + ((it + 1) & 1) * PLANES * plane;
        const bool check = t >= p.Tmin;
        float *actsT = p.acts + t * stepA, *cellT = p.cell + t * stepC, *thT = p.th + t * stepC;
        char *yT = (char *)p.y_op + t * stepC * MELT;
        u64 *xslot = xbase + (long)(it & 1) * CS * (RPL * NT);

        f32x4 acc[4], g_[RPL];
        char ptc[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) { ptc[r] = pt[r]; g_[r] = pre[r]; }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[g][r] = 0.f;
        // own units' part of the recurrent product: needs nothing from the partners
        auto product = [&](auto j0_, auto j1_) {          // K chunks [j0, j1) of the member-relative order
            constexpr int j0 = decltype(j0_)::value, j1 = decltype(j1_)::value;
            if constexpr (SP && (!X3 || QUAD) && CN_FWD_READ_AHEAD && (j1 - j0) > 2) {
                // the partners' chunks of the 8-CU shape (7 reads, 28 MFMAs): left to hipcc the reads are issued in pairs, each
                // pair behind the MFMAs of the pair before it (two operand buffers), and every pair's LDS latency is exposed
                // (1 250 cycles for 28 MFMAs by the stamps).  All reads first, then the MFMAs.
                u32x4 a[j1 - j0];
#pragma unroll
                for (int j = j0; j < j1; ++j) a[j - j0] = *(const u32x4 *)(ycur + c * pitch + j * 64 + q * 16);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = j0; j < j1; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if constexpr (QUAD) { smma16(acc[g], a[j - j0], wsl[g][j], spidx); smma16(acc[g], a[j - j0], wsp[g][j], spidx); }
                        else smma16(acc[g], a[j - j0], wsp[g][j], spidx);
                    }
                return;
            }
#pragma unroll
            for (int j = j0; j < j1; ++j) {
                const u32x4 a = *(const u32x4 *)(ycur + c * pitch + j * 64 + q * 16);
                [[maybe_unused]] u32x4 al;
                if constexpr (X3 && !QUAD) al = *(const u32x4 *)(ycur + plane + c * pitch + j * 64 + q * 16);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (QUAD) { smma16(acc[g], a, wsl[g][j], spidx); smma16(acc[g], a, wsp[g][j], spidx); }
                    else if constexpr (X3) smma16_x3(acc[g], a, al, wsp[g][j], wsl[g][j], spidx);
                    else if constexpr (SP) smma16(acc[g], a, wsp[g][j], spidx);
                    else mma16<false>(acc[g], a, wreg[g][j]);
                }
            }
        };
        // y[t-1] of the partners' units: published at the end of their previous step.  The samples are taken NOW (their round trip
        // runs beside the own part of the product) and looked at behind it
        // partner j = member + 1 + j (mod CS) sits in column block j + 1 of the member-relative tile
        const u64 *slots[(CS - 1) * RPL];
        EarlyPoll<(CS - 1) * RPL> early;
        {
            u64 *xprev = xbase + (long)((it + 1) & 1) * CS * (RPL * NT);
#pragma unroll
            for (int j = 0; j < CS - 1; ++j)
#pragma unroll
                for (int r = 0; r < RPL; ++r) slots[j * RPL + r] = xprev + (long)((member + 1 + j) % CS) * (RPL * NT) + r * NT + tid;
        }
        if (it > 0) poll_early<0>(slots, early);
        CLS(0)
        product(std::integral_constant<int, 0>(), std::integral_constant<int, KCO / 2>());
        if (it > 0) poll_early<1>(slots, early);
        product(std::integral_constant<int, KCO / 2>(), std::integral_constant<int, KCO>());
        if (it > 0) poll_early<2>(slots, early);
        CLS_FORCE(acc[3][0]) CLS(1)
        // (the scheduler must not pull the first look at the samples -- and with it the wait for them -- up in front of the MFMAs)
        __builtin_amdgcn_sched_barrier(0);
        if (it > 0) {
            unsigned vals[(CS - 1) * RPL];
            poll_finish<(CS - 1) * RPL>(slots, p.xch_epoch + it, p.fault, vals, gaveup, early);
            CLS_FORCE(vals[0]) CLS(2)
#pragma unroll
            for (int j = 0; j < CS - 1; ++j)
#pragma unroll
                for (int r = 0; r < RPL; ++r)
                {
                    char *dst = const_cast<char *>(ycur) + tile_off(j + 1, r);
                    if constexpr (X3) {
                        __bf16 yh, yl;
                        split_bf16(__builtin_bit_cast(float, vals[j * RPL + r]), yh, yl);
                        *(__bf16 *)dst = yh; *(__bf16 *)(dst + (QUAD ? 2 * pitch : plane)) = yl;
                    } else *(unsigned short *)dst = (unsigned short)vals[j * RPL + r];
                }
            lds_barrier();
            CLS(3)
        }
        // (the poll above drains vmcnt: the prefetch is issued behind it so that it has a whole step to land)
        prefetch(d ? t - 2 : t + 2, pre, pt);
        product(std::integral_constant<int, KCO>(), std::integral_constant<int, KC>());
        CLS_FORCE(acc[3][0]) CLS(4)

#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const bool dummy = check && ptc[r] == 0;
            const float cp = cst[r];
            float s_[4];                                 // recurrent sums of this sequence (SP: its two tile rows)
#pragma unroll
            for (int g = 0; g < 4; ++g) s_[g] = QUAD ? (acc[g][0] + acc[g][1]) + (acc[g][2] + acc[g][3]) : (SP ? acc[g][(2 * r) & 3] + acc[g][(2 * r + 1) & 3] : acc[g][r]);
            // ComputeBlockOutputFn, LstmLayer.cu:87-136
            const float ni = tanh_ref<false>(s_[0] + g_[r][0]);
            const float ig = logistic<false>(s_[1] + g_[r][1] + cp * pi);
            const float fg = logistic<false>(s_[2] + g_[r][2] + cp * pf);
            const float cs = ni * ig + cp * fg;
            const float og = logistic<false>(s_[3] + g_[r][3] + cs * po);
            const float th = tanh_ref<false>(cs);
            const float y = th * og;
            const float co = dummy ? 0.f : cs;
            const float yo = dummy ? 0.f : y;
            const __bf16 yb = (__bf16)yo;
            cst[r] = co;
            // hand y[t] of this unit to the partners first (it is on their critical path), then keep it here
            u64 *mine = xslot + (long)member * (RPL * NT) + r * NT + tid;
            if constexpr (X3) {
                publish(mine, p.xch_epoch + it + 1, __builtin_bit_cast(unsigned, yo));
                __bf16 yh, yl;
                split_bf16(yo, yh, yl);
                *(__bf16 *)(ynxt + tile_off(0, r)) = yh; *(__bf16 *)(ynxt + (QUAD ? 2 * pitch : plane) + tile_off(0, r)) = yl;
            } else {
                publish(mine, p.xch_epoch + it + 1, __builtin_bit_cast(unsigned short, yb));
                *(__bf16 *)(ynxt + tile_off(0, r)) = yb;       // the tile is member-relative: own units first
            }
            const f32x4 av = {ni, ig, fg, og};
            *(f32x4 *)(actsT + oA[r]) = av;
            cellT[oC[r]] = co;
            thT[oC[r]] = th;                                 // for the backward pass (cn_lstm.hip)
            if constexpr (X3) ((float *)yT)[oC[r]] = yo; else ((__bf16 *)yT)[oC[r]] = yb;
        }
        CLS(5)
        lds_barrier();
        CLS(6)
    };

    prefetch(d ? T - 1 : 0, preA, ptA);
    prefetch(d ? T - 2 : 1, preB, ptB);
    lds_barrier();
#ifdef CN_CL_STAMP
    st_prev = __builtin_amdgcn_s_memtime();
#endif
    for (int it = 0; it < T; it += 2) {
        step(it, preA, ptA);
        if (it + 1 < T) step(it + 1, preB, ptB);
    }
    CLS_STORE_F
}

// ---------------------------------------------------------------------------------------------
// backward: helper workgroups that keep the L2 ahead of a cluster (round 5)
// ---------------------------------------------------------------------------------------------
// What a backward step reads -- gate activations, cell states, tanh(c), outputErrors: 28 bytes per unit-frame -- was written a
// whole forward pass ago and comes from HBM (LVCSR: 400 MB per layer, more than the Infinity Cache).  The cluster kernels poll
// with vector loads, vmcnt retires in order, so every prefetch a wave has in flight must land before its next poll does: however
// far ahead it is issued, a prefetch has ONE step (1.1 us) to arrive, and with the loads served from cache-hot lines instead
// (timing-only build CN_CL_DIAG_HOT) the 2-CU backward kernel is 17-19 % faster (reading B 2.76 -> 2.52 ms, LVCSR 9.89 -> 9.10 ms
// per fraction -- an upper bound: those lines sit in the CU's own L1).  A ninth wave per CU cannot do the fetching (226 registers
// per wave: two waves per SIMD is all that fits), so extra WORKGROUPS do: helper h, on a free CU of the XCD that runs cluster h (block ids congruent mod 8: placement is speed only),
// reads the cluster's progress off its exchange granules (tag = epoch + step + 1) and touches one dword of every 128-byte line the
// cluster will read 3 ... 10 steps later; the members' own prefetches then find their lines in that XCD's L2.  A helper never
// writes, never makes anybody wait, and gives up by itself (bounded) -- results cannot depend on it.
template <int CS, int RPL, int GNT>
__device__ __forceinline__ void bwd_cluster_touch_ahead(const LstmRec &p, int cluster, int nclusters, int HP)
{
    if (cluster >= nclusters) return;
    const int T = p.T, PS = p.PS, dirs = p.dirs, SEQ = 4 * RPL;
    const int d = cluster % dirs, s0 = (cluster / dirs) * SEQ;
    const long arow = (long)dirs * 4 * HP, crow = (long)dirs * HP, stepA = (long)PS * arow, stepC = (long)PS * crow;
    const u64 *xbase = p.xch + (long)cluster * 2 * CS * (RPL * GNT);          // member 0, granule 0, thread 0 of either parity
    const u64 *tag0 = xbase, *tag1 = xbase + (long)CS * (RPL * GNT);
    // lines of one (step, sequence): acts HP * 16 bytes, err / cell / th HP * 4 bytes each
    const int la = HP * 16 / 128, lc = HP * 4 / 128, per_seq = la + 3 * lc, per_step = SEQ * per_seq;
    constexpr int AHEAD = 3, WINDOW = 8;
    int done = -1;                       // every step up to `done` has been touched
    for (int spin = 0; spin < 64 * 1024; ++spin) {
        const u64 a = __hip_atomic_load(tag0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(tag1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int pa = (int)((unsigned)(a >> 32) - p.xch_epoch) - 1, pb = (int)((unsigned)(b >> 32) - p.xch_epoch) - 1;
        int prog = -1;                   // last step member 0 has published (tags of earlier launches are <= xch_epoch)
        if (pa >= 0 && pa < T) prog = pa;
        if (pb >= 0 && pb < T && pb > prog) prog = pb;
        if (prog >= T - 1 - AHEAD) return;
        const int first = max(done + 1, prog + AHEAD), last = min(T - 1, prog + AHEAD + WINDOW - 1);
        // up to four lines per thread and round, issued together and awaited INSIDE the statement: an asm load's destination counts
        // as written when its statement ends, and a sink that is reused while its load is still in flight would be overwritten
        // by it (lanes without a line re-read the tag)
        const int count = (last - first + 1) * per_step;
        for (int i0 = threadIdx.x; i0 < count; i0 += 4 * blockDim.x) {
            const char *line[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * blockDim.x;
                line[k] = (const char *)tag0;
                if (i < count) {
                    const int it = first + i / per_step, r = i % per_step, sv = s0 + r / per_seq, l = r % per_seq;
                    const int t = d ? it : T - 1 - it;
                    if (l < la) line[k] = (const char *)(p.acts + t * stepA + sv * arow + (long)d * HP * 4) + l * 128;
                    else {
                        const int kk = (l - la) / lc, ll = (l - la) % lc;
                        const float *base = kk == 0 ? p.err : (kk == 1 ? p.cell : p.th);
                        line[k] = (const char *)(base + t * stepC + sv * crow + (long)d * HP) + ll * 128;
                    }
                }
            }
            unsigned k0, k1, k2, k3;
            asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %5, off\n\tglobal_load_dword %2, %6, off\n\t"
                         "global_load_dword %3, %7, off\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3) : "v"(line[0]), "v"(line[1]), "v"(line[2]), "v"(line[3]) : "memory");
        }
        if (last > done) done = last;
        __builtin_amdgcn_s_sleep(8);
    }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
template <int RPL> struct ClBwdPre { f32x4 a[RPL]; float e[RPL], cp[RPL], th[RPL]; char pt[RPL]; };

template <int PREC, int HP, int UPC, int RPL>
__global__ __launch_bounds__(UPC * 4) void lstm_bwd_cluster_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool X3 = PREC == P_X3;
    constexpr int MELT = X3 ? 4 : 2, PLANES = X3 ? 2 : 1;
    constexpr int G = X3 ? 4 : 2;                    // exchange granules per lane and sequence: four fp32 deltas, or two bf16 pairs
    // tanh(cell state) read back from the forward pass instead of recomputed: on the 64-unit members; the 128-unit members run two
    // waves per SIMD at the register limit, where the extra staged value costs more than the five instructions (measured: +7 %)
    constexpr bool TH = CN_TH_STORE && UPC <= 64;
    // SP: 2:4 row-pair products (cn_lstm_device.h); KHS: with one sequence per lane the other row pair of its quad holds the
    // second half of every member part (two accumulators, half the operand reads), see cn_lstm.hip.  KHS on the 64-unit
    // members only: measured on the 128-unit members (reading B / LVCSR) the backward kernel is 5 % slower with it.
    constexpr bool SP = CN_SPARSE || X3;
    constexpr bool KHS = SP && RPL == 1 && UPC <= 64;
    constexpr int CS = HP / UPC, NT = UPC * 4, KC = SP ? 4 * HP / 64 : 4 * HP / 32;
    // K-quarter stacking (cn_lstm.hip, backward kernel): with one sequence per lane the twelve padding rows of the operand tile
    // carry the other three quarters of every member's part of K; a wave reads KC / 4 instead of KC chunks per step (Hp = 256:
    // 8 instead of 32 ds_read_b128, 256 instead of 1024 LDS cycles per CU and step).  Member part p, quarter r, chunk kq of the
    // tile is chunk p*KCO + r*KCO/4 + kq of the member-relative K order below.
    // Measured: 8 CUs x 64 units (Hp = 512) backward step -20 % (longutt_5x1024: 51.0 -> 45.3 ms per fraction); 2 CUs x 128 units
    // (Hp = 256) +35 %: that shape holds 128 VGPRs of W_rec per lane at two waves per SIMD and the three extra accumulators
    // push the time loop into scratch.  On for the 64-unit members only.
    constexpr bool KQS = !SP && RPL == 1 && UPC <= 64 && (KC / CS) % 4 == 0 && CN_KQ_STACK;
    // packed granules (above): the 8-CU bf16 shape, where a lane polls 14 granules per step otherwise
    constexpr bool PACK = CN_PACKED_XCH && !X3 && SP && CS == 8;
    constexpr int pitch = lds_pitch(SP ? (KHS ? 2 * HP : 4 * HP) : (KQS ? HP : 4 * HP) * 2);     // delta tile row: k = 4*unit + gate
    int cluster, member;
    cluster_of<CS>(cluster, member);
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    const int nclusters = dirs * (PS / (4 * RPL)), ncblocks = (nclusters + 7) / 8 * 8 * CS;
    if ((int)blockIdx.x >= ncblocks) {       // a helper workgroup (launch_cluster appends them): warms the L2 ahead of cluster blockIdx - ncblocks
        bwd_cluster_touch_ahead<CS, RPL, G * UPC * 4>(p, (int)blockIdx.x - ncblocks, nclusters, HP);
        return;
    }
    if (cluster >= nclusters) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int d = cluster % dirs, s0 = (cluster / dirs) * (4 * RPL);
    const int lunit = 16 * wave + c, unit = member * UPC + lunit;
    const long arow = (long)dirs * 4 * HP, crow = (long)dirs * HP;
    // tile position of the deltas of local unit `lunit` of member part `part` (0 = own) for the sequence in lane quarter q
    [[maybe_unused]] const int krow = KQS ? lunit / (UPC / 4) : 0;
    // (SP: offset of the gate pair (n, i) in the even row of the sequence; (f, o) sit at the same offset one row further)
    auto tile_off = [&](int part, int r) {
        if constexpr (SP) {
            const int uh = KHS ? lunit % (UPC / 2) : lunit, rp = KHS ? lunit / (UPC / 2) : r;
            return (4 * q + 2 * rp) * pitch + (part * (KHS ? UPC / 32 : UPC / 16) + (uh >> 4)) * 64 + sp_pos(4 * (uh & 15)) * 2;
        } else
            return KQS ? (4 * q + krow) * pitch + (part * (UPC / 4) + lunit % (UPC / 4)) * 8
                       : (4 * q + r) * pitch + (part * UPC + lunit) * 8;
    };
    [[maybe_unused]] const int spidx = sp_index(c);

    constexpr int plane = 16 * pitch;                // P_X3: hi plane, then lo plane
    for (int i = tid * 4; i < 2 * PLANES * plane; i += NT * 4) *(unsigned *)(smem + i) = 0u;
    bool gaveup = false;

    constexpr int KCO = KC / CS;       // own units' K chunks first, see the forward kernel
    [[maybe_unused]] u32x4 wreg[SP ? 1 : KC];
    [[maybe_unused]] u32x8 wsp[SP ? KC : 1], wsl[X3 ? KC : 1];
    int kch[KC];
    const char *Wd = (const char *)p.WrecT + (long)d * 4 * HP * HP * MELT;
#pragma unroll
    for (int j = 0; j < KC; ++j) kch[j] = ((member + j / KCO) % CS) * KCO + j % KCO;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
        if constexpr (X3) sp_load_split((const float *)Wd + (long)unit * 4 * HP + kch[j] * 64 + q * 16, wsp[j], wsl[j]);
        else if constexpr (SP) wsp[j] = sp_load_bf16(Wd + ((long)unit * 4 * HP + kch[j] * 64 + q * 16) * 2);
        else wreg[j] = *(const u32x4 *)(Wd + ((long)unit * 4 * HP) * 2 + kch[j] * 64 + q * 16);
    }
    const float pi = p.peep[(d * 3 + 0) * HP + unit], pf = p.peep[(d * 3 + 1) * HP + unit], po = p.peep[(d * 3 + 2) * HP + unit];

    int oP[RPL], oA[RPL], oC[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int sv = s0 + 4 * r + q;
        oP[r] = sv; oA[r] = sv * (int)arow + (d * HP + unit) * 4; oC[r] = sv * (int)crow + d * HP + unit;
    }
    const long stepA = (long)PS * arow, stepC = (long)PS * crow;
    u64 *xbase = p.xch + (long)cluster * 2 * CS * (RPL * G * NT);
    [[maybe_unused]] u64 *xpack = PACK ? p.xch_packed + (long)cluster * 2 * CS * (RPL * NT) : nullptr;
    if constexpr (PACK) {
        // my packed slots of both parities say "nothing yet"; once that is acknowledged the cluster meets (granules of the ordinary
        // kind in parity slot 1, tagged with the launch's epoch: no step of any launch uses that value), and only then is a packed
        // slot of anybody read
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int r = 0; r < RPL; ++r)
                __hip_atomic_store(xpack + (long)par * CS * (RPL * NT) + (long)member * (RPL * NT) + r * NT + tid, (u64)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        u64 *meet = xbase + (long)CS * (RPL * G * NT);
        publish(meet + (long)member * (RPL * G * NT) + tid, p.xch_epoch, 0u);
        const u64 *ms[CS - 1];
#pragma unroll
        for (int j = 0; j < CS - 1; ++j) ms[j] = meet + (long)((member + 1 + j) % CS) * (RPL * G * NT) + tid;
        unsigned mv[CS - 1];
        consume_all<CS - 1>(ms, p.xch_epoch, p.fault, mv, gaveup);
    }

    float fgn[RPL], ecn[RPL], dign[RPL], dfgn[RPL], ccur[RPL];
    float sb[4] = {0.f, 0.f, 0.f, 0.f}, spi = 0.f, spf = 0.f, spo = 0.f;
#pragma unroll
    for (int r = 0; r < RPL; ++r) fgn[r] = ecn[r] = dign[r] = dfgn[r] = 0.f;

    const int tfirst = d ? 0 : T - 1;
    ClBwdPre<RPL> preA, preB;
    CLS_DECL
    auto prefetch = [&](int t, ClBwdPre<RPL> &pre) {
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
#ifdef CN_CL_DIAG_HOT
        t = tfirst;                  // timing-only build (results wrong): every prefetch from the same, cache-hot lines
#endif
        const int tprev = d ? t + 1 : t - 1;
        const bool hasprev = tprev >= 0 && tprev < T;
        const float *actsT = p.acts + t * stepA, *errT = p.err + t * stepC;
        const float *cellP = p.cell + (hasprev ? tprev : t) * stepC;
        const char *patT = p.pat + (long)t * PS;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            pre.pt[r] = patT[oP[r]];
            pre.e[r] = errT[oC[r]];
            pre.a[r] = *(const f32x4 *)(actsT + oA[r]);
            pre.cp[r] = cellP[oC[r]];
            if constexpr (TH) pre.th[r] = (p.th + t * stepC)[oC[r]];
        }
    };

    auto step = [&](int it, ClBwdPre<RPL> &pre) {
        const int t = d ? it : T - 1 - it;
        const char *dcur = smem + (it & 1) * PLANES * plane;
        char *dnxt = smem + ((it + 1) & 1) * PLANES * plane;
        const bool check = t >= p.Tmin;
        const int tprev_ = d ? t + 1 : t - 1;
        const bool hasprev_ = tprev_ >= 0 && tprev_ < T;
        char *deltaT = (char *)p.delta_op + t * stepA * MELT;
        u64 *xslot = xbase + (long)(it & 1) * CS * (RPL * G * NT);

        // the partners' deltas of the previous step: sampled first thing (EarlyPoll), looked at behind the own part of the product
        const u64 *slots[(CS - 1) * RPL * G];
        [[maybe_unused]] const u64 *pslots[(CS - 1) * RPL];
        EarlyPoll<(CS - 1) * RPL * G> early;
        {
            u64 *xprev = xbase + (long)((it + 1) & 1) * CS * (RPL * G * NT);
#pragma unroll
            for (int j = 0; j < CS - 1; ++j)
#pragma unroll
                for (int r = 0; r < RPL; ++r) {
                    const u64 *theirs = xprev + (long)((member + 1 + j) % CS) * (RPL * G * NT) + (r * G) * NT + tid;
#pragma unroll
                    for (int i = 0; i < G; ++i) slots[(j * RPL + r) * G + i] = theirs + i * NT;
                    if constexpr (PACK)
                        pslots[j * RPL + r] = xpack + (long)((it + 1) & 1) * CS * (RPL * NT) + (long)((member + 1 + j) % CS) * (RPL * NT) + r * NT + tid;
                }
        }
        if (!PACK && it > 0) poll_early<0>(slots, early);
        f32x4 acc, a_[RPL];
        [[maybe_unused]] f32x4 accq[4];              // KQS: one accumulator per K-quarter
        [[maybe_unused]] f32x4 accs, acch;           // SP: the row-pair accumulator (KHS: first half) and the second half's
        float cp_[RPL];
        [[maybe_unused]] float th_[RPL];
        char ptc[RPL];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (r < RPL) ? pre.e[r < RPL ? r : 0] : 0.f;
#pragma unroll
        for (int r = 0; r < RPL; ++r) { ptc[r] = pre.pt[r]; cp_[r] = hasprev_ ? pre.cp[r] : 0.f; a_[r] = pre.a[r]; if constexpr (TH) th_[r] = pre.th[r]; }
        // the product over member parts [p0, p1) of K
        auto product = [&](auto p0_, auto p1_) {      // (compile-time bounds: wreg must stay in registers)
            constexpr int p0 = decltype(p0_)::value, p1 = decltype(p1_)::value;
            if constexpr (KHS && !X3 && CN_KHS_READ_AHEAD) {
                // all operand reads of the parts first, then the MFMAs: left to hipcc each MFMA pair waited for its own read (second part
                // of the product in the 8-CU shape: 1 290 cycles for 28 MFMAs; long utterances 36.6 -> 35.4 ms per fraction).  The same
                // in the forward kernel and in the 2-CU backward kernel measured nothing and was not kept.
                constexpr int KH = KCO / 2, NR = (p1 - p0) * KH;
                u32x4 a[NR];
#pragma unroll
                for (int j = 0; j < NR; ++j) a[j] = *(const u32x4 *)(dcur + c * pitch + (p0 * KH + j) * 64 + q * 16);
                // (round 5: without this hipcc sinks the reads back between the MFMAs, one or two ahead of their use)
                if constexpr (CN_BWD_READ_FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pp = p0; pp < p1; ++pp)
#pragma unroll
                    for (int kq = 0; kq < KH; ++kq) {
                        smma16(accs, a[(pp - p0) * KH + kq], wsp[pp * KCO + kq], spidx);
                        smma16(acch, a[(pp - p0) * KH + kq], wsp[pp * KCO + KH + kq], spidx);
                    }
            } else if constexpr (KHS) {
                constexpr int KH = KCO / 2;
#pragma unroll
                for (int pp = p0; pp < p1; ++pp)
#pragma unroll
                    for (int kq = 0; kq < KH; ++kq) {
                        const u32x4 a = *(const u32x4 *)(dcur + c * pitch + (pp * KH + kq) * 64 + q * 16);
                        if constexpr (X3) {
                            const u32x4 al = *(const u32x4 *)(dcur + plane + c * pitch + (pp * KH + kq) * 64 + q * 16);
                            smma16_x3(accs, a, al, wsp[pp * KCO + kq], wsl[pp * KCO + kq], spidx);
                            smma16_x3(acch, a, al, wsp[pp * KCO + KH + kq], wsl[pp * KCO + KH + kq], spidx);
                        } else {
                            smma16(accs, a, wsp[pp * KCO + kq], spidx);
                            smma16(acch, a, wsp[pp * KCO + KH + kq], spidx);
                        }
                    }
            } else if constexpr (SP) {
#pragma unroll
                for (int j = p0 * KCO; j < p1 * KCO; ++j) {
                    const u32x4 a = *(const u32x4 *)(dcur + c * pitch + j * 64 + q * 16);
                    if constexpr (X3) {
                        const u32x4 al = *(const u32x4 *)(dcur + plane + c * pitch + j * 64 + q * 16);
                        smma16_x3(accs, a, al, wsp[j], wsl[j], spidx);
                    } else smma16(accs, a, wsp[j], spidx);
                }
            } else if constexpr (KQS) {
                constexpr int KQ = KCO / 4;
#pragma unroll
                for (int pp = p0; pp < p1; ++pp)
#pragma unroll
                    for (int kq = 0; kq < KQ; ++kq) {
                        const u32x4 a = *(const u32x4 *)(dcur + c * pitch + (pp * KQ + kq) * 64 + q * 16);
#pragma unroll
                        for (int r = 0; r < 4; ++r) mma16<false>(accq[r], a, wreg[pp * KCO + r * KQ + kq]);
                    }
            } else {
#pragma unroll
                for (int j = p0 * KCO; j < p1 * KCO; ++j) {
                    const u32x4 a = *(const u32x4 *)(dcur + c * pitch + j * 64 + q * 16);
                    mma16<false>(acc, a, wreg[j]);
                }
            }
        };
        if constexpr (KQS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) accq[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            accq[0][0] = acc[0];                     // err enters as the C operand of quarter 0
        }
        if constexpr (SP) {                          // err enters through the even row of each sequence
            accs = f32x4{0.f, 0.f, 0.f, 0.f}; acch = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < RPL; ++r) accs[2 * r] = acc[r];
        }
        if (!PACK && it > 0) poll_early<1>(slots, early);
        CLS(0)
        product(std::integral_constant<int, 0>(), std::integral_constant<int, 1>());
        if (!PACK && it > 0) poll_early<2>(slots, early);
#ifdef CN_CL_STAMP
        if constexpr (SP) { CLS_FORCE(accs[0]) } else { CLS_FORCE(acc[0]) }
#endif
        CLS(1)
        __builtin_amdgcn_sched_barrier(0);     // (the first look at the samples, and the wait for them, stays behind the MFMAs above)
        if (it > 0) {      // the partners' deltas of the previous step (sampled at the top of the step, see EarlyPoll)
            unsigned vals[(CS - 1) * RPL * G];
            if constexpr (PACK) {
                unsigned plo[(CS - 1) * RPL], phi[(CS - 1) * RPL];
                consume_all_packed<(CS - 1) * RPL, CN_POLL_DELAY_PACKED>(pslots, 1u + (unsigned)(it - 1) % 15u, p.fault, plo, phi, gaveup);
#pragma unroll
                for (int i = 0; i < (CS - 1) * RPL; ++i) { vals[2 * i] = plo[i]; vals[2 * i + 1] = phi[i]; }
            } else
            poll_finish<(CS - 1) * RPL * G>(slots, p.xch_epoch + it, p.fault, vals, gaveup, early);
            CLS_FORCE(vals[0]) CLS(2)
#pragma unroll
            for (int j = 0; j < CS - 1; ++j)
#pragma unroll
                for (int r = 0; r < RPL; ++r)
                {
                    char *dst = const_cast<char *>(dcur) + tile_off(j + 1, r);
                    if constexpr (X3) {
                        bf16x4 dh, dl;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            __bf16 h_, l_;
                            split_bf16(__builtin_bit_cast(float, vals[(j * RPL + r) * G + g]), h_, l_);
                            dh[g] = h_; dl[g] = l_;
                        }
                        const uint2 hb = __builtin_bit_cast(uint2, dh), lb = __builtin_bit_cast(uint2, dl);
                        *(unsigned *)dst = hb.x; *(unsigned *)(dst + pitch) = hb.y;
                        *(unsigned *)(dst + plane) = lb.x; *(unsigned *)(dst + plane + pitch) = lb.y;
                    } else if constexpr (SP) {
                        *(unsigned *)dst = vals[(j * RPL + r) * 2];                     // (n, i): even row
                        *(unsigned *)(dst + pitch) = vals[(j * RPL + r) * 2 + 1];       // (f, o): odd row
                    } else *(uint2 *)dst = make_uint2(vals[(j * RPL + r) * 2], vals[(j * RPL + r) * 2 + 1]);
                }
            lds_barrier();
            CLS(3)
        }
        prefetch(d ? t + 2 : t - 2, pre);      // behind the poll (it drains vmcnt), see the forward kernel
        product(std::integral_constant<int, 1>(), std::integral_constant<int, CS>());
#ifdef CN_CL_STAMP
        if constexpr (SP) { CLS_FORCE(accs[0]) } else { CLS_FORCE(acc[0]) }
#endif
        CLS(4)
        if constexpr (KQS) acc[0] = (accq[0][0] + accq[1][1]) + (accq[2][2] + accq[3][3]);
        if constexpr (KHS) acc[0] = (accs[0] + accs[1]) + (acch[2] + acch[3]);
        else if constexpr (SP) {
#pragma unroll
            for (int r = 0; r < RPL; ++r) acc[r] = accs[2 * r] + accs[2 * r + 1];
        }

#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const bool dummy = check && ptc[r] == 0;
            // ComputeBlockErrorsFn, LstmLayer.cu:236-285
            const float e = acc[r];
            const float ni = a_[r][0], ig = a_[r][1], fg = a_[r][2], og = a_[r][3];
            const float cs = ccur[r], cp = cp_[r];
            const float th = TH ? th_[r] : tanh_ref<false>(cs);
            float dog = og * (1.0f - og) * th * e;
            float ec = og * (1.0f - th * th) * e + po * dog;
            ec += fgn[r] * ecn[r] + pi * dign[r] + pf * dfgn[r];
            float dni = ig * (1.0f - ni * ni) * ec;
            float dfg = fg * (1.0f - fg) * cp * ec;
            float dig = ig * (1.0f - ig) * ni * ec;
            dni = clip1(dni); dig = clip1(dig); dfg = clip1(dfg); dog = clip1(dog);
            dni = dummy ? 0.f : dni; dig = dummy ? 0.f : dig; dfg = dummy ? 0.f : dfg; dog = dummy ? 0.f : dog;
            ec = dummy ? 0.f : ec;      // selects, not a branch (see cn_lstm.hip)
            fgn[r] = dummy ? 0.f : fg;
            ecn[r] = ec; dign[r] = dig; dfgn[r] = dfg;
            ccur[r] = cp;
            sb[0] += dni; sb[1] += dig; sb[2] += dfg; sb[3] += dog;
            spi += cp * dig; spf += cp * dfg; spo += cs * dog;
            u64 *mine = xslot + (long)member * (RPL * G * NT) + (r * G) * NT + tid;
            if constexpr (X3) {
                const f32x4 dv = {dni, dig, dfg, dog};
                const float ds[4] = {dni, dig, dfg, dog};       // (scalars: a bit_cast of an ext_vector element picks element 0)
                bf16x4 dh, dl;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    publish(mine + g * NT, p.xch_epoch + it + 1, __float_as_uint(ds[g]));
                    __bf16 h_, l_;
                    split_bf16(ds[g], h_, l_);
                    dh[g] = h_; dl[g] = l_;
                }
                const uint2 hb = __builtin_bit_cast(uint2, dh), lb = __builtin_bit_cast(uint2, dl);
                char *dst = dnxt + tile_off(0, r);
                *(unsigned *)dst = hb.x; *(unsigned *)(dst + pitch) = hb.y;
                *(unsigned *)(dst + plane) = lb.x; *(unsigned *)(dst + plane + pitch) = lb.y;
                *(f32x4 *)((float *)deltaT + oA[r]) = dv;
            } else {
                const bf16x4 dv = {(__bf16)dni, (__bf16)dig, (__bf16)dfg, (__bf16)dog};
                const u64 bits = __builtin_bit_cast(u64, dv);
                if constexpr (PACK) {
                    unsigned mlo, mhi;
                    packed_masks(1u + (unsigned)it % 15u, mlo, mhi);
                    __hip_atomic_store(xpack + (long)(it & 1) * CS * (RPL * NT) + (long)member * (RPL * NT) + r * NT + tid,
                                       bits | (u64)mlo | ((u64)mhi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // (the helper workgroup reads the cluster's progress off member 0's ordinary granule)
                    if (member == 0 && tid == 0 && r == 0) publish(mine, p.xch_epoch + it + 1, 0u);
                } else {
                publish(mine, p.xch_epoch + it + 1, (unsigned)bits);
                publish(mine + NT, p.xch_epoch + it + 1, (unsigned)(bits >> 32));
                }
                if constexpr (SP) {                              // member-relative tile: own units first
                    *(unsigned *)(dnxt + tile_off(0, r)) = (unsigned)bits;
                    *(unsigned *)(dnxt + tile_off(0, r) + pitch) = (unsigned)(bits >> 32);
                } else *(bf16x4 *)(dnxt + tile_off(0, r)) = dv;
                *(bf16x4 *)((__bf16 *)deltaT + oA[r]) = dv;
            }
        }
        CLS(5)
        lds_barrier();
        CLS(6)
    };

#pragma unroll
    for (int r = 0; r < RPL; ++r) ccur[r] = (p.cell + tfirst * stepC)[oC[r]];
    prefetch(tfirst, preA);
    prefetch(d ? 1 : T - 2, preB);
    lds_barrier();
#ifdef CN_CL_STAMP
    st_prev = __builtin_amdgcn_s_memtime();
#endif
    for (int it = 0; it < T; it += 2) {
        step(it, preA);
        if (it + 1 < T) step(it + 1, preB);
    }
    CLS_STORE

    float v[7] = {sb[0], sb[1], sb[2], sb[3], spi, spf, spo};
#pragma unroll
    for (int i = 0; i < 7; ++i) { v[i] += __shfl_xor(v[i], 16); v[i] += __shfl_xor(v[i], 32); }
    if (q == 0) {
        lstm_grad_sums_out(p, HP, d, unit, v);
    }
}

// ---------------------------------------------------------------------------------------------
// backward, PARTIAL-SUM exchange (round 4; one sequence per lane)
// ---------------------------------------------------------------------------------------------
// The kernel above splits the BPTT product e[i] = err[i] + sum_k WrecT[i][k] delta[k] (LstmLayer.cu:936-943,970-977) by OUTPUT
// unit: a member needs the deltas of ALL units, so every lane publishes its four deltas (2 granules in bf16, 4 in the split
// mode) and polls (CS - 1) x that many, writes the partners' deltas into its LDS tile and passes a second barrier before the
// second part of the product.  This kernel splits by K instead: a member multiplies ITS OWN deltas (k in its units x 4 gates,
// already in its LDS tile) with the rows of WrecT of ALL units -- the same number of MFMAs and the same number of W_rec
// registers -- and hands each partner the partial sums of the partner's units: ONE fp32 granule per lane and partner, the
// partners' parts first (their trip through L2 runs beside the own part's MFMAs), no partner data in LDS, ONE barrier per
// step.  e = (err + own part) + partner parts in member order: fp32 sums, the same terms in another order.
// Split-bf16 mode: the two spare rows of a sequence's row quad carry the lo halves of its deltas (cn_lstm_s2.hip, "row
// quads"): [hi; lo] x W_hi + [hi; lo] x W_lo = two MFMAs per chunk instead of three, one tile plane instead of two.
// Granule slot of (destination member, source member, thread): ((dest * CS + src) * NT + tid), two step parities.
template <int PREC, int HP, int UPC>
__global__ __launch_bounds__(UPC * 4) void lstm_bwd_cluster_psum_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool X3 = PREC == P_X3;
    constexpr int MELT = X3 ? 4 : 2;
    constexpr bool TH = CN_TH_STORE && UPC <= 64;
    constexpr int CS = HP / UPC, NT = UPC * 4;
    constexpr int KCO = 4 * UPC / 64;                // K = 64 chunks of a member's own deltas (k = 4*unit + gate)
    constexpr int pitch = lds_pitch(KCO * 64);       // a tile row: 32 stored values per chunk
    constexpr int plane = 16 * pitch;
    int cluster, member;
    cluster_of<CS>(cluster, member);
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    if (cluster >= dirs * (PS / 4)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int d = cluster % dirs, s0 = (cluster / dirs) * 4;
    const int lunit = 16 * wave + c, unit = member * UPC + lunit;
    const long arow = (long)dirs * 4 * HP, crow = (long)dirs * HP;
    // the (n, i) pair of this lane's unit in the even row of its sequence's quad; (f, o) one row further; lo halves two rows further
    const int toff = (4 * q) * pitch + (lunit >> 4) * 64 + sp_pos(4 * (lunit & 15)) * 2;
    const int spidx = sp_index(c);

    for (int i = tid * 4; i < 2 * plane; i += NT * 4) *(unsigned *)(smem + i) = 0u;
    bool gaveup = false;

    // W_rec^T fragments: part jp = the rows of member (member + jp) % CS's units, columns = the K chunks of THIS member's deltas
    u32x8 wsp[CS * KCO];
    [[maybe_unused]] u32x8 wsl[X3 ? CS * KCO : 1];
    const char *Wd = (const char *)p.WrecT + (long)d * 4 * HP * HP * MELT;
#pragma unroll
    for (int jp = 0; jp < CS; ++jp)
#pragma unroll
        for (int kq = 0; kq < KCO; ++kq) {
            const long w0 = (long)(((member + jp) % CS) * UPC + lunit) * 4 * HP + (member * KCO + kq) * 64 + q * 16;
            if constexpr (X3) sp_load_split((const float *)Wd + w0, wsp[jp * KCO + kq], wsl[jp * KCO + kq]);
            else wsp[jp * KCO + kq] = sp_load_bf16(Wd + w0 * 2);
        }
    const float pi = p.peep[(d * 3 + 0) * HP + unit], pf = p.peep[(d * 3 + 1) * HP + unit], po = p.peep[(d * 3 + 2) * HP + unit];

    const int sv = s0 + q;
    const int oP = sv, oA = sv * (int)arow + (d * HP + unit) * 4, oC = sv * (int)crow + d * HP + unit;
    const long stepA = (long)PS * arow, stepC = (long)PS * crow;
    u64 *xbase = p.xch + (long)cluster * 2 * CS * CS * NT;

    float fgn = 0.f, ecn = 0.f, dign = 0.f, dfgn = 0.f, ccur;
    float sb[4] = {0.f, 0.f, 0.f, 0.f}, spi = 0.f, spf = 0.f, spo = 0.f;

    const int tfirst = d ? 0 : T - 1;
    ClBwdPre<1> preA, preB;
    auto prefetch = [&](int t, ClBwdPre<1> &pre) {
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        const int tprev = d ? t + 1 : t - 1;
        const bool hasprev = tprev >= 0 && tprev < T;
        pre.pt[0] = (p.pat + (long)t * PS)[oP];
        pre.e[0] = (p.err + t * stepC)[oC];
        pre.a[0] = *(const f32x4 *)(p.acts + t * stepA + oA);
        pre.cp[0] = (p.cell + (hasprev ? tprev : t) * stepC)[oC];
        if constexpr (TH) pre.th[0] = (p.th + t * stepC)[oC];
    };

    auto step = [&](int it, ClBwdPre<1> &pre) {
        const int t = d ? it : T - 1 - it;
        const char *dcur = smem + (it & 1) * plane;
        char *dnxt = smem + ((it + 1) & 1) * plane;
        const bool check = t >= p.Tmin;
        const int tprev_ = d ? t + 1 : t - 1;
        const bool hasprev_ = tprev_ >= 0 && tprev_ < T;
        char *deltaT = (char *)p.delta_op + t * stepA * MELT;
        u64 *xslot = xbase + (long)(it & 1) * CS * CS * NT;

        const char ptc = pre.pt[0];
        const float cp = hasprev_ ? pre.cp[0] : 0.f;
        const f32x4 a_ = pre.a[0];
        [[maybe_unused]] float th_ = 0.f;
        if constexpr (TH) th_ = pre.th[0];
        const float e_in = pre.e[0];

        // this member's deltas of the previous step (zero tile at it = 0): read once, used by every part
        u32x4 a[KCO];
#pragma unroll
        for (int kq = 0; kq < KCO; ++kq) a[kq] = *(const u32x4 *)(dcur + c * pitch + kq * 64 + q * 16);
        auto part = [&](int jp, float c0) {
            f32x4 acc = {c0, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kq = 0; kq < KCO; ++kq) {
                if constexpr (X3) smma16(acc, a[kq], wsl[jp * KCO + kq], spidx);      // small terms first
                smma16(acc, a[kq], wsp[jp * KCO + kq], spidx);
            }
            const float v = X3 ? (acc[0] + acc[1]) + (acc[2] + acc[3]) : acc[0] + acc[1];
            KEEP_TUPLE(acc, v);
            return v;
        };
        // the partners' parts first: published at once, their trip through L2 runs beside the own part below
        if (it > 0) {
#pragma unroll
            for (int jp = 1; jp < CS; ++jp) {
                const float v = part(jp, 0.f);
                const int dest = (member + jp) % CS;
                publish(xslot + ((long)dest * CS + member) * NT + tid, p.xch_epoch + it + 1, __float_as_uint(v));
            }
        }
        float e = part(0, e_in);                     // err enters as the C operand (LstmLayer.cu:939: addProduct into tmpOutputErrors)
        if (it > 0) {
            const u64 *slots[CS - 1];
            unsigned vals[CS - 1];
#pragma unroll
            for (int j = 0; j < CS - 1; ++j) slots[j] = xslot + ((long)member * CS + (member + 1 + j) % CS) * NT + tid;
            consume_all<CS - 1, CN_POLL_DELAY_PSUM>(slots, p.xch_epoch + it + 1, p.fault, vals, gaveup);
#pragma unroll
            for (int j = 0; j < CS - 1; ++j) e += __uint_as_float(vals[j]);
        }
        prefetch(d ? t + 2 : t - 2, pre);            // behind the poll (it drains vmcnt)

        const bool dummy = check && ptc == 0;
        // ComputeBlockErrorsFn, LstmLayer.cu:236-285
        const float ni = a_[0], ig = a_[1], fg = a_[2], og = a_[3];
        const float cs = ccur;
        const float th = TH ? th_ : tanh_ref<false>(cs);
        float dog = og * (1.0f - og) * th * e;
        float ec = og * (1.0f - th * th) * e + po * dog;
        ec += fgn * ecn + pi * dign + pf * dfgn;
        float dni = ig * (1.0f - ni * ni) * ec;
        float dfg = fg * (1.0f - fg) * cp * ec;
        float dig = ig * (1.0f - ig) * ni * ec;
        dni = clip1(dni); dig = clip1(dig); dfg = clip1(dfg); dog = clip1(dog);
        dni = dummy ? 0.f : dni; dig = dummy ? 0.f : dig; dfg = dummy ? 0.f : dfg; dog = dummy ? 0.f : dog;
        ec = dummy ? 0.f : ec;
        fgn = dummy ? 0.f : fg;
        ecn = ec; dign = dig; dfgn = dfg;
        ccur = cp;
        sb[0] += dni; sb[1] += dig; sb[2] += dfg; sb[3] += dog;
        spi += cp * dig; spf += cp * dfg; spo += cs * dog;
        if constexpr (X3) {
            const f32x4 dv = {dni, dig, dfg, dog};
            const float ds[4] = {dni, dig, dfg, dog};
            bf16x4 dh, dl;
#pragma unroll
            for (int g = 0; g < 4; ++g) { __bf16 h_, l_; split_bf16(ds[g], h_, l_); dh[g] = h_; dl[g] = l_; }
            const uint2 hb = __builtin_bit_cast(uint2, dh), lb = __builtin_bit_cast(uint2, dl);
            *(unsigned *)(dnxt + toff) = hb.x; *(unsigned *)(dnxt + toff + pitch) = hb.y;
            *(unsigned *)(dnxt + toff + 2 * pitch) = lb.x; *(unsigned *)(dnxt + toff + 3 * pitch) = lb.y;
            *(f32x4 *)((float *)deltaT + oA) = dv;
        } else {
            const bf16x4 dv = {(__bf16)dni, (__bf16)dig, (__bf16)dfg, (__bf16)dog};
            const u64 bits = __builtin_bit_cast(u64, dv);
            *(unsigned *)(dnxt + toff) = (unsigned)bits;
            *(unsigned *)(dnxt + toff + pitch) = (unsigned)(bits >> 32);
            *(bf16x4 *)((__bf16 *)deltaT + oA) = dv;
        }
        lds_barrier();
    };

    ccur = (p.cell + tfirst * stepC)[oC];
    prefetch(tfirst, preA);
    prefetch(d ? 1 : T - 2, preB);
    lds_barrier();
    for (int it = 0; it < T; it += 2) {
        step(it, preA);
        if (it + 1 < T) step(it + 1, preB);
    }

    float v[7] = {sb[0], sb[1], sb[2], sb[3], spi, spf, spo};
#pragma unroll
    for (int i = 0; i < 7; ++i) { v[i] += __shfl_xor(v[i], 16); v[i] += __shfl_xor(v[i], 32); }
    if (q == 0) {
        lstm_grad_sums_out(p, HP, d, unit, v);
    }
}

// ---------------------------------------------------------------------------------------------
// backward, bf16, Hp = 256: TWO sequences per cluster of two CUs, one wave per SIMD ("s2c"; round 5)
// ---------------------------------------------------------------------------------------------
// The cut of cn_lstm_s2.hip (a wave owns 32 units x 2 sequences; both unit groups accumulate through two zero-padded views of the
// operand tile; row-pair sparse MFMAs) on the two-CU split of the kernel above: member m owns output units 128 m .. 128 m + 127
// and keeps their rows of W_rec^T -- all K = 1024 of them, 256 registers per lane -- resident.  Why: the step of the 4-sequence
// cluster kernel is a chain (deltas published -> hop through L2 -> partner's deltas into LDS -> barrier -> their half of the
// product -> block errors -> deltas) that eight waves per CU walk with two waves per SIMD and four sequences' cell arithmetic
// each; this cut halves the vector work per CU (a lane owns ONE (unit, sequence)), doubles the CUs in use (PS = 64: 128 of 256)
// and is the layout the hand-written loop below schedules.
// Tile: a sequence's quad of rows = (K half, parity) as in lstm_bwd_s2_kernel, K half 0 = the member's OWN 128 units (written by
// the member itself at the end of a step), K half 1 = the partner's (written behind the poll of the next step); e = accA[0] +
// accA[1] (own half, own rows of W) + accB[2] + accB[3] (partner half).  The own half is multiplied while the partner's
// deltas are on their way.  Exchange: two 8-byte {tag, value} granules per lane and step (cn_lstm_cluster.hip protocol: tags
// count on across launches, slots alternate with the step parity, bounded spins).
__global__ __launch_bounds__(256) void lstm_bwd_s2c_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HP = 256, UPC = 128, CS = 2, NT = 256, G = 2;
    constexpr int KCS = 4 * HP / 64, KCH = KCS / 2;          // 16 chunks of K in all, 8 per tile row (= per member)
    constexpr int pitch = lds_pitch(KCH * 64);
    constexpr int DROWS = 9, plane = DROWS * pitch;
    int cluster, member;
    cluster_of<CS>(cluster, member);
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    if (cluster >= dirs * (PS / 2)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int ug = q >> 1, sq = q & 1;
    const int d = cluster % dirs, s0 = (cluster / dirs) * 2;
    const long arow = (long)dirs * 4 * HP, crow = (long)dirs * HP;

    for (int i = tid * 4; i < 2 * plane; i += NT * 4) *(unsigned *)(smem + i) = 0u;
    // dummy-slot table: dtab[t][s] = (t >= Tmin && patTypes[t][s0 + s] == NONE) (LstmLayer.cu:224-234 with :949,983)
    unsigned char *dtab = (unsigned char *)smem + 2 * plane;
    for (int i = tid; i < 2 * T; i += NT) {
        const int tt = i >> 1;
        dtab[i] = tt >= p.Tmin && p.pat[(long)tt * PS + s0 + (i & 1)] == 0;
    }

    // W_rec^T rows of this lane's two output units (unit group j), K chunks in member-relative order: own units first
    u32x8 wsp[2][KCS];
    const char *Wd = (const char *)p.WrecT + (long)d * 4 * HP * HP * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kc = 0; kc < KCS; ++kc) {
            const int kch = ((member + kc / KCH) % CS) * KCH + kc % KCH;
            wsp[j][kc] = sp_load_bf16(Wd + ((long)(member * UPC + 32 * wave + 16 * j + c) * 4 * HP + kch * 64 + q * 16) * 2);
        }
    const int spidx = sp_index(c);
    const int av0 = (c < 8 ? c : 8) * pitch + q * 16, av1 = (c >= 8 ? c - 8 : 8) * pitch + q * 16;

    const int lunit = 32 * wave + 16 * ug + c, unit = member * UPC + lunit;
    const float pi = p.peep[(d * 3 + 0) * HP + unit], pf = p.peep[(d * 3 + 1) * HP + unit], po = p.peep[(d * 3 + 2) * HP + unit];
    const int sv = s0 + sq;
    const unsigned oA = (unsigned)(sv * (int)arow + (d * HP + unit) * 4);
    const unsigned oC = (unsigned)(sv * (int)crow + d * HP + unit);
    const unsigned stepA = (unsigned)PS * (unsigned)arow, stepC = (unsigned)PS * (unsigned)crow;
    // the lane's column in a tile row (32 stored values per 64-k chunk) and the rows of its own / its partner twin's deltas
    const int col = ((lunit >> 4) * 32 + sp_pos(4 * (lunit & 15))) * 2;
    const int oOwn = (4 * sq) * pitch + col, oPar = (4 * sq + 2) * pitch + col;
    u64 *xbase = p.xch + (long)cluster * 2 * CS * (G * NT);
    bool gaveup = false;

    float fgn = 0.f, ecn = 0.f, dign = 0.f, dfgn = 0.f, ccur;
    float sb[4] = {0.f, 0.f, 0.f, 0.f}, spi = 0.f, spf = 0.f, spo = 0.f;

    const int tfirst = d ? 0 : T - 1;
    struct Stage { f32x4 a; float e, cp, th; } preA, preB;
    auto prefetch = [&](int t, Stage &pre) {
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
        const int tprev = d ? t + 1 : t - 1;
        const bool hasprev = tprev >= 0 && tprev < T;
        const unsigned bA = (unsigned)t * stepA, bC = (unsigned)t * stepC;
        const unsigned bCp = (unsigned)(hasprev ? tprev : t) * stepC;
        pre.e = p.err[bC + oC];
        pre.a = *(const f32x4 *)(p.acts + bA + oA);
        pre.cp = p.cell[bCp + oC];
        pre.th = p.th[bC + oC];
    };

    auto step = [&](int it, Stage &pre) {
        const int t = d ? it : T - 1 - it;
        char *dcur = smem + (it & 1) * plane;
        char *dnxt = smem + ((it + 1) & 1) * plane;
        const int tprev_ = d ? t + 1 : t - 1;
        const bool hasprev_ = tprev_ >= 0 && tprev_ < T;       // !lastCall, LstmLayer.cu:947,981
        const unsigned bD = (unsigned)t * stepA;
        u64 *xslot = xbase + (long)(it & 1) * CS * (G * NT), *xprev = xbase + (long)((it + 1) & 1) * CS * (G * NT);
        const u64 *theirs = xprev + (long)((member + 1) % CS) * (G * NT) + tid;
        u64 *mine = xslot + (long)member * (G * NT) + tid;

        const unsigned char dmy = dtab[2 * t + sq];
        const float e_ = pre.e, c_ = pre.cp, th_ = pre.th;
        const f32x4 a_ = pre.a;
        const float cp = hasprev_ ? c_ : 0.f;

        // own half of the BPTT product (LstmLayer.cu:939-942 / :973-976), while the partner's deltas travel
        f32x4 accA = {e_, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};      // err enters as the C operand
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
            const u32x4 a0 = *(const u32x4 *)(dcur + av0 + kc * 64), a1 = *(const u32x4 *)(dcur + av1 + kc * 64);
            smma16(accA, a0, wsp[0][kc], spidx);
            smma16(accA, a1, wsp[1][kc], spidx);
        }
        if (it > 0) {      // the partner's deltas of the previous step: rows (K half 1) of the tile being read
            const u64 *slots[G] = {theirs, theirs + NT};
            unsigned vals[G];
            consume_all<G>(slots, p.xch_epoch + it, p.fault, vals, gaveup);
            *(unsigned *)(dcur + oPar) = vals[0];                // (n, i): even row
            *(unsigned *)(dcur + oPar + pitch) = vals[1];        // (f, o): odd row
            lds_barrier();
        }
        prefetch(d ? t + 2 : t - 2, pre);      // behind the poll (it drains vmcnt)
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
            const u32x4 a0 = *(const u32x4 *)(dcur + av0 + kc * 64), a1 = *(const u32x4 *)(dcur + av1 + kc * 64);
            smma16(accB, a0, wsp[0][KCH + kc], spidx);
            smma16(accB, a1, wsp[1][KCH + kc], spidx);
        }
        const float e = (accA[0] + accA[1]) + (accB[2] + accB[3]);
        KEEP_TUPLE(accA, e); KEEP_TUPLE(accB, e);

        // ComputeBlockErrorsFn, LstmLayer.cu:236-285, the explicit operation sequence of lstm_bwd_s2_kernel
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
        sb[0] += dni; sb[1] += dig; sb[2] += dfg; sb[3] += dog;
        spi = __builtin_fmaf(cp, dig, spi); spf = __builtin_fmaf(cp, dfg, spf); spo = __builtin_fmaf(cs, dog, spo);
        const bf16x4 dv = {(__bf16)dni, (__bf16)dig, (__bf16)dfg, (__bf16)dog};
        const u64 bits = __builtin_bit_cast(u64, dv);
        publish(mine, p.xch_epoch + it + 1, (unsigned)bits);
        publish(mine + NT, p.xch_epoch + it + 1, (unsigned)(bits >> 32));
        *(unsigned *)(dnxt + oOwn) = (unsigned)bits;
        *(unsigned *)(dnxt + oOwn + pitch) = (unsigned)(bits >> 32);
        *(bf16x4 *)((__bf16 *)p.delta_op + bD + oA) = dv;
        lds_barrier();
    };

    ccur = p.cell[(unsigned)tfirst * stepC + oC];
    prefetch(tfirst, preA);
    prefetch(d ? 1 : T - 2, preB);
    lds_barrier();
    for (int it = 0; it < T; it += 2) {
        step(it, preA);
        if (it + 1 < T) step(it + 1, preB);
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
// backward, bf16, Hp = 256, two sequences per cluster of two CUs: the time loop written by hand (round 5)
// ---------------------------------------------------------------------------------------------
// The cut, the layout, the operand order per accumulator and the arithmetic of lstm_bwd_s2c_kernel above (bit-equal on real
// slots), the instruction stream of lstm_bwd_s2_asm_kernel (cn_lstm_s2.hip) with the exchange in the middle of the step; the
// text of the loop is generated (tools/gen_s2c_loop.py -> cn_lstm_s2c_loop.inc; the step is described there).  W_rec^T: 32
// fragments = all 256 AGPRs.  Members publish a granule of zeros for "step -1" in front of the loop, so that step 0 polls like
// every other step.  Dummy slots as in the other hand-written loops: the pattern type alone decides.
#include "cn_lstm_s2c_loop.inc"
// CN_S2C_STAMP (tools/stamps_s2c.py; `make variantc NAME=s2cstamp DEFS=-DCN_S2C_STAMP`; never in the shipped build): every wave of
// workgroup 0 sums s_memtime deltas per step segment (the segments are named in tools/gen_s2c_loop.py)
#ifdef CN_S2C_STAMP
__device__ unsigned cn_s2c_stamp_buf[4][8];
extern "C" int cn_dbg_read_stamps_s2c(unsigned *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cn_s2c_stamp_buf), sizeof(cn_s2c_stamp_buf)); }
#endif
__global__ __launch_bounds__(256) void lstm_bwd_s2c_asm_kernel(LstmRec p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HP = 256, UPC = 128, CS = 2, NT = 256, G = 2, KCS = 16, KCH = 8;
    constexpr int pitch = lds_pitch(KCH * 64), plane = 9 * pitch;
    static_assert(pitch == 544 && plane == 4896 && G * NT * 4 == 2048, "LDS / granule offsets of the generated loop");
    static_assert(CN_GUARD_STEPS >= 5, "prefetch four steps ahead, c[prev] five");
    int cluster, member;
    cluster_of<CS>(cluster, member);
    const int PS = p.PS, T = p.T, dirs = p.dirs;
    if (cluster >= dirs * (PS / 2)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int ug = q >> 1, sq = q & 1;
    const int d = cluster % dirs, s0 = (cluster / dirs) * 2;
    const long arow = (long)dirs * 4 * HP, crow = (long)dirs * HP;

    for (int i = tid * 4; i < 2 * plane; i += NT * 4) *(unsigned *)(smem + i) = 0u;

    u32x8 w[2][KCS];
    const char *Wd = (const char *)p.WrecT + (long)d * 4 * HP * HP * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kc = 0; kc < KCS; ++kc) {
            const int kch = ((member + kc / KCH) % CS) * KCH + kc % KCH;
            w[j][kc] = sp_load_bf16(Wd + ((long)(member * UPC + 32 * wave + 16 * j + c) * 4 * HP + kch * 64 + q * 16) * 2);
        }
    const int spidx = sp_index(c);
    // ONE view of the tile: lanes c >= 8 read the rows of lanes c - 8 (tools/gen_s2c_loop.py); a lane keeps the sums of its own unit group
    const unsigned av = (c & 7) * pitch + q * 16;
    const unsigned long long ugm = 0xFFFFFFFF00000000ull;      // lanes of unit group 1 (q >= 2)

    const int lunit = 32 * wave + 16 * ug + c, unit = member * UPC + lunit;
    const float pi = p.peep[(d * 3 + 0) * HP + unit], pf = p.peep[(d * 3 + 1) * HP + unit], po = p.peep[(d * 3 + 2) * HP + unit];
    const int sv = s0 + sq;
    const unsigned col = ((lunit >> 4) * 32 + sp_pos(4 * (lunit & 15))) * 2;
    const unsigned oT = (4 * sq) * pitch + col, oT1 = oT + plane, oTp = (4 * sq + 2) * pitch + col, oTp1 = oTp + plane;
    // exchange granules: byte offsets of this lane's / its partner twin's first granule in slot set 0 / 1
    const char *xch = (const char *)(p.xch + (long)cluster * 2 * CS * (G * NT));
    const unsigned oXm0 = (unsigned)(((0 * CS + member) * (G * NT) + tid) * 8), oXm1 = (unsigned)(((1 * CS + member) * (G * NT) + tid) * 8);
    const unsigned oXt0 = (unsigned)(((0 * CS + (member ^ 1)) * (G * NT) + tid) * 8), oXt1 = (unsigned)(((1 * CS + (member ^ 1)) * (G * NT) + tid) * 8);
    // "step -1": zeros, tagged with this launch's first tag, in the slot set step 0 polls
    publish((u64 *)(xch + oXm1), p.xch_epoch, 0u);
    publish((u64 *)(xch + oXm1) + NT, p.xch_epoch, 0u);
    unsigned tagc = p.xch_epoch;

    // byte offsets of this lane, kept BIAS steps ahead of the time index, bases BIAS steps behind (lstm_bwd_s2_asm_kernel)
    constexpr long BIAS = 8;
    const long t0 = d ? 0 : T - 1, dt = d ? 1 : -1;
    const long stepA = (long)PS * arow, stepC = (long)PS * crow;
    const unsigned lC = (unsigned)(sv * (int)crow + d * HP + unit);
    unsigned oA = (unsigned)((t0 + BIAS) * stepA * 4) + lC * 16, oC = (unsigned)((t0 + BIAS) * stepC * 4) + lC * 4;
    unsigned oD = (unsigned)((t0 + BIAS) * stepA * 2) + lC * 8, oP = (unsigned)((t0 + BIAS) * PS) + (unsigned)sv;
    const unsigned sA = (unsigned)(dt * stepA * 4), sC = (unsigned)(dt * stepC * 4), sD = (unsigned)(dt * stepA * 2), sP = (unsigned)(dt * PS);
    const char *acts = (const char *)p.acts - BIAS * stepA * 4, *cell = (const char *)p.cell - BIAS * stepC * 4, *th = (const char *)p.th - BIAS * stepC * 4;
    const char *err = (const char *)p.err - BIAS * stepC * 4, *pat = p.pat - BIAS * PS;
    const char *actspf = acts + 4 * dt * stepA * 4, *thpf = th + 4 * dt * stepC * 4, *errpf = err + (4 - 1) * dt * stepC * 4;
    const char *cell1 = cell + dt * stepC * 4, *cellpf = cell + 5 * dt * stepC * 4, *patpf = pat + 4 * dt * PS;
    const char *delta1 = (const char *)p.delta_op - BIAS * stepA * 2 - dt * stepA * 2;
    unsigned cnt = (unsigned)T - 1;                  // steps behind the current one

    float fgn = 0.f, ecn = 0.f, dign = 0.f, dfgn = 0.f, dni = 0.f, dog = 0.f;
    float sb0 = 0.f, sb1 = 0.f, sb2 = 0.f, sb3 = 0.f, spi = 0.f, spf = 0.f, spo = 0.f;
    float ccA, ccB, th0, th1, th2, th3, cp0, cp1, cp2, cp3;
    int pt0, pt1, pt2, pt3;
    float x0, x1, m, t2m, wm, carm, d2m, d3m, d4m, car;
    unsigned long long last, tq;
    unsigned spin = 0, gave = 0;
#ifdef CN_S2C_STAMP
    unsigned st[7] = {0, 0, 0, 0, 0, 0, 0}, tz;
    unsigned long long tm;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    lds_barrier();
#ifdef CN_S2C_STAMP
    unsigned tl = (unsigned)__builtin_amdgcn_s_memtime();
    asm volatile(S2C_ASM_TEXT_STAMP
#else
    asm volatile(S2C_ASM_TEXT
#endif
        : [fgn] "+v"(fgn), [ecn] "+v"(ecn), [dign] "+v"(dign), [dfgn] "+v"(dfgn), [dni] "+v"(dni), [dog] "+v"(dog),
          [sb0] "+v"(sb0), [sb1] "+v"(sb1), [sb2] "+v"(sb2), [sb3] "+v"(sb3), [spi] "+v"(spi), [spf] "+v"(spf), [spo] "+v"(spo),
          [oA] "+v"(oA), [oC] "+v"(oC), [oD] "+v"(oD), [oP] "+v"(oP), [tagc] "+v"(tagc), [cnt] "+s"(cnt), [spin] "+s"(spin), [gave] "+s"(gave),
          [last] "=&s"(last), [tq] "=&s"(tq),
          [ccA] "=&v"(ccA), [ccB] "=&v"(ccB), [th0] "=&v"(th0), [th1] "=&v"(th1), [th2] "=&v"(th2), [th3] "=&v"(th3),
          [cp0] "=&v"(cp0), [cp1] "=&v"(cp1), [cp2] "=&v"(cp2), [cp3] "=&v"(cp3),
          [pt0] "=&v"(pt0), [pt1] "=&v"(pt1), [pt2] "=&v"(pt2), [pt3] "=&v"(pt3),
          [x0] "=&v"(x0), [x1] "=&v"(x1), [m] "=&v"(m), [t2m] "=&v"(t2m), [wm] "=&v"(wm), [carm] "=&v"(carm),
          [d2m] "=&v"(d2m), [d3m] "=&v"(d3m), [d4m] "=&v"(d4m), [car] "=&v"(car)
#ifdef CN_S2C_STAMP
          , [st0] "+s"(st[0]), [st1] "+s"(st[1]), [st2] "+s"(st[2]), [st3] "+s"(st[3]), [st4] "+s"(st[4]), [st5] "+s"(st[5]), [st6] "+s"(st[6]),
          [tz] "=&s"(tz), [tl] "+s"(tl), "={s[98:99]}"(tm)
#endif
        : [w0k0] "a"(w[0][0]), [w0k1] "a"(w[0][1]), [w0k2] "a"(w[0][2]), [w0k3] "a"(w[0][3]), [w0k4] "a"(w[0][4]), [w0k5] "a"(w[0][5]), [w0k6] "a"(w[0][6]), [w0k7] "a"(w[0][7]), [w0k8] "a"(w[0][8]), [w0k9] "a"(w[0][9]), [w0k10] "a"(w[0][10]), [w0k11] "a"(w[0][11]), [w0k12] "a"(w[0][12]), [w0k13] "a"(w[0][13]), [w0k14] "a"(w[0][14]), [w0k15] "a"(w[0][15]), [w1k0] "a"(w[1][0]), [w1k1] "a"(w[1][1]), [w1k2] "a"(w[1][2]), [w1k3] "a"(w[1][3]), [w1k4] "a"(w[1][4]), [w1k5] "a"(w[1][5]), [w1k6] "a"(w[1][6]), [w1k7] "a"(w[1][7]), [w1k8] "a"(w[1][8]), [w1k9] "a"(w[1][9]), [w1k10] "a"(w[1][10]), [w1k11] "a"(w[1][11]), [w1k12] "a"(w[1][12]), [w1k13] "a"(w[1][13]), [w1k14] "a"(w[1][14]), [w1k15] "a"(w[1][15]),
          [spidx] "v"(spidx), [av] "v"(av), [ugm] "s"(ugm), [oT] "v"(oT), [oT1] "v"(oT1), [oTp] "v"(oTp), [oTp1] "v"(oTp1),
          [oXm0] "v"(oXm0), [oXm1] "v"(oXm1), [oXt0] "v"(oXt0), [oXt1] "v"(oXt1), [pi] "v"(pi), [pf] "v"(pf), [po] "v"(po),
          [acts] "s"(acts), [actspf] "s"(actspf), [cell] "s"(cell), [cell1] "s"(cell1), [cellpf] "s"(cellpf), [th] "s"(th), [thpf] "s"(thpf),
          [err] "s"(err), [errpf] "s"(errpf), [pat] "s"(pat), [patpf] "s"(patpf), [delta1] "s"(delta1), [xch] "s"(xch), [fault] "s"(p.fault),
          [sA] "s"(sA), [sC] "s"(sC), [sD] "s"(sD), [sP] "s"(sP)
        : "memory", "vcc", "scc",
          "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");

#ifdef CN_S2C_STAMP
    if (blockIdx.x == 0 && lane == 0) {
        for (int i = 0; i < 7; ++i) cn_s2c_stamp_buf[wave][i] = st[i];
        cn_s2c_stamp_buf[wave][7] = (unsigned)(__builtin_amdgcn_s_memrealtime() - rt0);      // 100 MHz ticks over the loop
    }
#endif
    // fold the two sequences of each unit column, then one atomic per (gate, unit) and workgroup
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
template <int PREC, int HP, int UPC, int RPL, bool BWD>
static void launch_cluster(hipStream_t s, const LstmRec &p)
{
    constexpr int CS = HP / UPC, NT = UPC * 4;
    const int nclusters = p.dirs * (p.PS / (4 * RPL));
    const int grid = (nclusters + 7) / 8 * 8 * CS;
    const size_t lds = 2 * (PREC == P_X3 ? 2 : 1) * 16 * (size_t)lds_pitch((BWD ? 4 : 1) * HP * 2);     // (upper bound: the row-pair tiles are narrower)
    // (no clearing per launch: tags continue from LstmRec::xch_epoch, which the caller advances by T + 1 per launch, so
    // the granules a previous launch left behind never match; the memset kernel and its stream bubble cost ~7 us per
    // layer pass)
    if constexpr (BWD && RPL == 1) {
        // One sequence per lane: the partial-sum exchange.  Measured (round 4, DESIGN A.5): the split-bf16 mode gains 5 % per step of
        // reading B (two MFMAs per chunk instead of three, 3 instead of 12 polled granules per thread); in bf16 the critical path
        // of a step is the same as with the delta exchange (delta -> LDS -> barrier -> MFMAs -> hop, in another order) and the
        // kernel measures 3-4 % SLOWER (2-CU and 8-CU shapes alike), so bf16 keeps the delta exchange.
        // CN_BWD_PSUM=1 / CN_NO_BWD_PSUM=1 force either one (A/B, tests).
        const bool psum = opt().bwd_psum ? true : (opt().no_bwd_psum ? false : PREC == P_X3);
        if (psum) {
            auto kp = lstm_bwd_cluster_psum_kernel<PREC, HP, UPC>;
            static DeviceOnce attr_once_p;
            if (attr_once_p.first()) (void)hipFuncSetAttribute((const void *)kp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            const size_t lds_p = 2 * 16 * (size_t)lds_pitch(4 * UPC);
            lstm_note_grid(p, grid);
            hipLaunchKernelGGL(kp, dim3(grid), dim3(NT), lds_p, s, p);
            if (p.kname) snprintf(p.kname, CN_KNAME_LEN, "lstm_bwd_cluster_psum_kernel<%d,%d,%d>", PREC, HP, UPC);
            return;
        }
    }
    auto kern = BWD ? lstm_bwd_cluster_kernel<PREC, HP, UPC, RPL> : lstm_fwd_cluster_kernel<PREC, HP, UPC, RPL>;
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    // backward: one helper workgroup per cluster behind the members (bwd_cluster_touch_ahead), while the chip has CUs to spare.
    // Measured (same box, interleaved, ms per fraction with / without): long utterances (8 CUs x 64 units, T = 2000) 35.66 / 36.27
    // and 35.79 / 36.16 (-1.4 %); reading B 2.819 / 2.811, LVCSR 10.19 / 10.10 (2 CUs x 128 units: +0.3 ... +0.8 %: an L2 hit
    // is still longer than the half step hipcc's latch copies leave a prefetch there) -- so only the 8-CU shape gets them.
    // CN_CLUSTER_HELPERS=1 / CN_NO_CLUSTER_HELPERS=1 force either way (A/B, tests).
    int helpers = 0;
    const bool want = opt().cluster_helpers ? true : (opt().no_cluster_helpers ? false : CS == 8);
    if (BWD && want && grid + (nclusters + 7) / 8 * 8 <= p.cluster_cus) helpers = (nclusters + 7) / 8 * 8;
    lstm_note_grid(p, grid);                       // (the helper workgroups form no sums)
    hipLaunchKernelGGL(kern, dim3(grid + helpers), dim3(NT), lds, s, p);
    if (p.kname) snprintf(p.kname, CN_KNAME_LEN, "lstm_%s_cluster_kernel<%d,%d,%d,%d>", BWD ? "bwd" : "fwd", PREC, HP, UPC, RPL);
}

// cluster shapes: Hp = 256 -> 2 CUs x 128 units (8 waves each), Hp = 512 -> 8 CUs x 64 units (4 waves each; the slice of
// W_rec a CU keeps in registers is 4*UPC*Hp operands = 256 KB in both cases)
// CN_CLUSTER4=1: Hp = 256 as 4 CUs x 64 units (A/B; the 64-unit members have registers to spare and take the stacked
// backward operand tile)
// P_X3 (fp32 tolerances): Hp = 256 as 4 CUs x 64 units (hi and lo fragments: 256 VGPRs per lane at one wave per SIMD); wider
// layers stream W_rec in that mode
static int cluster_size(int prec, int Hp)
{
    const bool four = opt().cluster4;
    if (prec == P_X3) return Hp == 256 ? 4 : 0;
    if (prec != P_BF16) return 0;
    return Hp == 256 ? (four ? 4 : 2) : (Hp == 512 ? 8 : 0);
}

// bytes of exchange buffer a layer of this shape needs (0 = the cluster path does not apply)
int lstm_cluster_size(int prec, int Hp, int dirs, int PS, int rpl, int num_cus)
{
    const int CS = cluster_size(prec, Hp);
    if (CS == 0 || rpl > 2 || opt().no_cluster) return 0;
    const int nclusters = dirs * (PS / (4 * rpl));
    if ((nclusters + 7) / 8 * 8 * CS > num_cus) return 0;      // every member must be resident (one workgroup per CU)
    return CS;
}
size_t lstm_cluster_xch_bytes(int prec, int Hp, int dirs, int PS, int rpl, int num_cus)
{
    const int CS = lstm_cluster_size(prec, Hp, dirs, PS, rpl, num_cus);
    if (CS == 0) return 0;
    const int nclusters = dirs * (PS / (4 * rpl));
    const int NT = (Hp / CS) * 4;
    const size_t delta_scheme = (size_t)nclusters * 2 * CS * rpl * (prec == P_X3 ? 4 : 2) * NT * sizeof(u64);     // (the delta-exchange backward kernel's granules)
    const size_t psum_scheme = (size_t)nclusters * 2 * CS * CS * NT * sizeof(u64);                                  // lstm_bwd_cluster_psum_kernel
    // ... and behind them the packed granules of the 8-CU bf16 backward kernel (a region no 32-bit-tag scheme ever reads)
    const size_t packed = (size_t)nclusters * 2 * CS * rpl * NT * sizeof(u64);
    return (delta_scheme > psum_scheme ? delta_scheme : psum_scheme) + packed;
}
size_t lstm_cluster_xch_packed_offset(int prec, int Hp, int dirs, int PS, int rpl, int num_cus)
{
    const int CS = lstm_cluster_size(prec, Hp, dirs, PS, rpl, num_cus);
    if (CS == 0) return 0;
    const size_t packed = (size_t)(dirs * (PS / (4 * rpl))) * 2 * CS * rpl * ((Hp / CS) * 4) * sizeof(u64);
    return lstm_cluster_xch_bytes(prec, Hp, dirs, PS, rpl, num_cus) - packed;
}

// Cluster launches of one device go through one gate: a cluster kernel needs ALL its workgroups resident, and two such
// grids of different contexts (streams) started together can each hold part of the CUs and spin for partners that never
// get one.  While a single stream launches cluster kernels the gate costs nothing; from the moment a second stream shows up
// every cluster launch waits (device side) for the completion event of the one before it.
struct ClusterGate { std::mutex mu; hipEvent_t last = nullptr; hipStream_t last_stream = nullptr; bool multi = false; };
static ClusterGate &cluster_gate()
{
    static std::mutex mu;
    static std::map<int, ClusterGate *> gates;
    int d = 0;
    (void)hipGetDevice(&d);
    std::lock_guard<std::mutex> lock(mu);
    ClusterGate *&g = gates[d];
    if (!g) g = new ClusterGate;
    return *g;
}

static void launch_cluster_shape(hipStream_t s, int prec, bool bwd, const LstmRec &p);

// a context is going away (its stream has been synchronised): the gate must not name the stream any more
void lstm_cluster_stream_gone(hipStream_t s)
{
    ClusterGate &gate = cluster_gate();
    std::lock_guard<std::mutex> lock(gate.mu);
    if (gate.last_stream == s) gate.last_stream = nullptr;
}

bool launch_lstm_cluster(hipStream_t s, int prec, bool bwd, LstmRec &p, unsigned *epoch)
{
    if (!p.xch || lstm_cluster_size(prec, p.Hp, p.dirs, p.PS, p.rpl, p.cluster_cus) == 0) return false;
    if (lstm_s2w_applies(prec, p, bwd)) return false;          // one CU per pair of sequences, no hop (cn_lstm_s2.hip)
    p.xch_epoch = *epoch;
    *epoch += (unsigned)p.T + 1;
    ClusterGate &gate = cluster_gate();
    std::lock_guard<std::mutex> lock(gate.mu);
    if (opt().cluster_gate_off) {          // probe only (tools/split_probe.py): two resident grids that together fit the chip
        launch_cluster_shape(s, prec, bwd, p);
        return true;
    }
    if (gate.last_stream && gate.last_stream != s && !gate.multi) {
        // first launch from a second stream: the earlier launches carry no event yet, wait for them on the host once
        (void)hipStreamSynchronize(gate.last_stream);
        (void)hipEventCreateWithFlags(&gate.last, hipEventDisableTiming);
        gate.multi = true;
    } else if (gate.multi && gate.last_stream != s) {
        (void)hipStreamWaitEvent(s, gate.last, 0);
    }
    launch_cluster_shape(s, prec, bwd, p);
    if (gate.multi) (void)hipEventRecord(gate.last, s);
    gate.last_stream = s;
    return true;
}

template <int PREC, int HP, int UPC>
static void launch_cluster_rpl(hipStream_t s, bool bwd, const LstmRec &p)
{
    if (p.rpl == 1) { if (bwd) launch_cluster<PREC, HP, UPC, 1, true>(s, p); else launch_cluster<PREC, HP, UPC, 1, false>(s, p); }
    else            { if (bwd) launch_cluster<PREC, HP, UPC, 2, true>(s, p); else launch_cluster<PREC, HP, UPC, 2, false>(s, p); }
}

// the s2c cut: bf16, Hp = 256, backward, one sequence per lane, an even number of sequences, both members of every cluster resident
static bool s2c_applies(int prec, bool bwd, const LstmRec &p)
{
    // (the COMPILED kernel of the cut measures 25 % slower per step than the 8-wave cluster kernel -- reading B 2.77 -> 3.05 ms, LVCSR
    // 9.88 -> 10.68 ms per fraction: 154 AGPR copies per step and one wave per SIMD to issue them --; it is the twin the
    // hand-written loop is held against, selected with CN_S2C=1.  CN_NO_S2C=1: the 8-wave kernel.)
    if (!bwd || prec != P_BF16 || p.Hp != 256 || p.rpl != 1 || p.PS % 2 || opt().no_s2c || opt().bwd_psum) return false;
    // the hand-written loop addresses activations with 32-bit byte offsets that run 8 steps ahead (as the other hand-written loops);
    // its compiled twin (CN_S2C=1) has no such limit but loses to the 8-wave kernel
    if (!opt().s2c && (unsigned long long)(p.T + 16) * p.PS * p.dirs * 4 * p.Hp * 4 >= 0xF0000000ull) return false;
    const int nclusters = p.dirs * (p.PS / 2);
    return (nclusters + 7) / 8 * 8 * 2 <= p.cluster_cus;
}
static void launch_s2c(hipStream_t s, const LstmRec &p)
{
    const int nclusters = p.dirs * (p.PS / 2), grid = (nclusters + 7) / 8 * 8 * 2;
    constexpr int pitch = lds_pitch(8 * 64);
    const size_t lds = 2 * 9 * (size_t)pitch + (((size_t)p.T * 2 + 15) & ~(size_t)15);
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void *)lstm_bwd_s2c_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)lstm_bwd_s2c_asm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const bool hand = !opt().s2c;
    // (one workgroup per CU: each claims the CU's whole LDS so that no gradient-GEMM workgroup is placed beside it, cn_lstm.hip)
    size_t lds_claim = opt().no_lds_claim ? lds : (size_t)(160 * 1024 - 1024);
    lstm_note_grid(p, grid);
    hipLaunchKernelGGL(hand ? lstm_bwd_s2c_asm_kernel : lstm_bwd_s2c_kernel, dim3(grid), dim3(256), lds_claim < lds ? lds : lds_claim, s, p);
    if (p.kname) snprintf(p.kname, CN_KNAME_LEN, hand ? "lstm_bwd_s2c_asm_kernel" : "lstm_bwd_s2c_kernel");
}

// CUs the backward cluster launch of this shape occupies (0: not a cluster shape) -- what a gradient GEMM beside it must leave free
int lstm_cluster_bwd_cus(int prec, const LstmRec &p)
{
    const int CS = lstm_cluster_size(prec, p.Hp, p.dirs, p.PS, p.rpl, p.cluster_cus);
    if (CS == 0) return 0;
    if (s2c_applies(prec, true, p)) return (p.dirs * (p.PS / 2) + 7) / 8 * 8 * 2;
    const int nclusters = p.dirs * (p.PS / (4 * p.rpl));
    return (nclusters + 7) / 8 * 8 * (CS + (CS == 8 ? 1 : 0));      // (+ the helper workgroups of the 8-CU shape)
}

static void launch_cluster_shape(hipStream_t s, int prec, bool bwd, const LstmRec &p)
{
    if (s2c_applies(prec, bwd, p)) { launch_s2c(s, p); return; }
    if (prec == P_X3)                                       launch_cluster_rpl<P_X3, 256, 64>(s, bwd, p);
    else if (p.Hp == 256 && cluster_size(prec, 256) == 4)   launch_cluster_rpl<P_BF16, 256, 64>(s, bwd, p);
    else if (p.Hp == 256)                                   launch_cluster_rpl<P_BF16, 256, 128>(s, bwd, p);
    else                                                    launch_cluster_rpl<P_BF16, 512, 64>(s, bwd, p);
}

}  // namespace cn

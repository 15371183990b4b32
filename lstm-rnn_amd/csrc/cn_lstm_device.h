// Device helpers shared by the recurrent LSTM kernels (cn_lstm.hip, cn_lstm_cluster.hip): the reference's
// activation functions, the 16x16 MFMA step and the LDS-only workgroup barrier.
#pragma once

#include <hip/hip_runtime.h>

namespace cn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) unsigned u32x8;
typedef __attribute__((ext_vector_type(16))) __bf16 bf16x16;

#define LOG2E 1.4426950408889634f

// Logistic::fn (Logistic.cuh:33-44).  The reference clamps to exactly 0 / 1 beyond |x| >= 88.72; both
// forms below reach the same limits without a branch (exp overflows to +inf -> 1/inf = 0; exp underflows
// -> 1/(1+0) = 1).  F32: libm-grade expf and IEEE division; bf16 mode: v_exp_f32 / v_rcp_f32.
template <bool F32>
__device__ __forceinline__ float logistic(float x)
{
    if constexpr (F32) return 1.0f / (1.0f + expf(-x));
    else return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-LOG2E * x));
}
// Tanh::fn = Maxmin1::fn(2x) = 2*Logistic::fn(2x) - 1 (Tanh.cuh:33-36, Maxmin1.cuh:33-36)
template <bool F32>
__device__ __forceinline__ float tanh_ref(float x)
{
    if constexpr (F32) return 2.0f * (1.0f / (1.0f + expf(-2.0f * x))) - 1.0f;
    else return __builtin_fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.0f * LOG2E * x)), -1.0f);
}
__device__ __forceinline__ float clip1(float e) { return fminf(fmaxf(e, -1.0f), 1.0f); }   // limitedError.cuh:31-34

// one 64-byte K chunk of a 16x16 tile product: 32 bf16 (one MFMA) or 16 fp32 (four MFMAs; the K order
// inside the chunk is permuted identically for A and B)
template <bool F32>
__device__ __forceinline__ void mma16(f32x4 &acc, const u32x4 &a, const u32x4 &b)
{
    if constexpr (F32) {
        // (bit_cast the whole vector: a bit_cast of a single ext_vector element picks element 0)
        const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[i], acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                      __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
}

// Row pitch of an MFMA A-operand tile in LDS (16 rows read as ds_read_b128 at row*pitch + 16*(lane>>4)).
// ds_read_b128 is served in groups of 16 lanes ({0-3,12-15,20-27}, ...) that mix rows of two k-quarters,
// so it is conflict-free exactly when pitch/16 = 2 (mod 4); a pitch of row + 16 B is a 2-way conflict on
// every read (SQ_LDS_BANK_CONFLICT = 4 cycles per read, measured).
__host__ __device__ constexpr int lds_pitch(int row_bytes)
{
    return row_bytes + 16 * ((2 - (row_bytes / 16) % 4 + 4) % 4);
}

// ---- split-bf16 ("bf16x3") operands -----------------------------------------------------------------------------
// x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-17 |x| (two 8-bit significands, round to nearest):
// a product a*b ~ ah*bh + al*bh + ah*bl drops al*bl (2^-18) and the two r terms (2^-17 each): ~2^-16 relative per term
// with fp32 accumulation, against 2^-9 for plain bf16 operands and 2^-24 for fp32.
__device__ __forceinline__ void split_bf16(float x, __bf16 &hi, __bf16 &lo)
{
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}
// eight consecutive fp32 (two 16-byte vectors) -> one MFMA fragment of 8 bf16 hi and one of 8 bf16 lo
__device__ __forceinline__ void split8(const f32x4 &x0, const f32x4 &x1, u32x4 &hi, u32x4 &lo)
{
    bf16x8 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __bf16 a, b;
        split_bf16(x0[i], a, b); h[i] = a; l[i] = b;
        split_bf16(x1[i], a, b); h[4 + i] = a; l[4 + i] = b;
    }
    hi = __builtin_bit_cast(u32x4, h); lo = __builtin_bit_cast(u32x4, l);
}
// three bf16 MFMAs of one 32-element K chunk of a 16x16 tile product (small terms first)
__device__ __forceinline__ void mma16_x3(f32x4 &acc, const u32x4 &ah, const u32x4 &al, const u32x4 &bh, const u32x4 &bl)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
}

// ---- 2:4 "row pair" products (v_smfmac_f32_16x16x64_bf16) -----------------------------------------------------------
// The 16-row A tile of a recurrent step holds 4*RPL sequences, so with RPL <= 2 at least half of its rows are padding and
// the dense MFMA spends its cycles on them.  The sparse MFMA multiplies a 16 x 64 A whose rows keep 2 of every 4 K
// positions (8 stored values + eight 2-bit positions per lane) with a dense 64 x 16 B in the cycles of the dense
// 16x16x32 (measured, tools/probe/smfmac_probe.hip: 17-18 cycles per instruction on 1, 2 or 8 accumulators).  Two tile
// rows share one sequence: the even row keeps positions {0,1} of every group of four, the odd row {2,3}; both are fully
// dense in their stored values, so nothing is pruned -- the product is exact, the K = 64 chunk costs one instruction
// instead of two, and the sequence's sum is D[even row] + D[odd row], both in the lane's own registers.
// Operand pairing (measured with unit impulses, same probe): A lane (row = lane & 15, j = lane >> 4), stored slot s with
// position field v meets B lane (col = lane & 15, j' = 2*(j & 1) + (s >> 2)) slot 8*(j >> 1) + 4*((s >> 1) & 1) + v; the
// position fields of a lane are bits [15:0] of the index register (ABID = 0).  With B lane j' slot i holding k' = 16*j' + i
// (16 consecutive K values per lane, two 16-byte loads), value k' of a sequence belongs in tile row parity (k' >> 1) & 1 at
// stored position sp_pos(k') of its 32-value row chunk; a reader lane takes its 8 slots with one ds_read_b128 at
// row*pitch + 64*chunk + 16*j, the same expression as for the dense tile.
__device__ __forceinline__ constexpr int sp_parity(int k) { return (k >> 1) & 1; }
__device__ __forceinline__ constexpr int sp_pos(int k)
{
    return 16 * ((k >> 3) & 1) + 8 * ((k >> 5) & 1) + 4 * ((k >> 4) & 1) + 2 * ((k >> 2) & 1) + (k & 1);
}
// index register of a reader lane: even tile rows keep positions {0,1}, odd rows {2,3} of every group
__device__ __forceinline__ int sp_index(int row) { return (row & 1) ? (int)0xEEEEEEEEu : 0x44444444; }
__device__ __forceinline__ u32x8 sp_join(const u32x4 &lo, const u32x4 &hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ void smma16(f32x4 &acc, const u32x4 &a, const u32x8 &b, int idx)
{
    acc = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x16, b), acc, idx, 0, 0);
}
// split operands (P_X3): small terms first, as mma16_x3
__device__ __forceinline__ void smma16_x3(f32x4 &acc, const u32x4 &ah, const u32x4 &al, const u32x8 &bh, const u32x8 &bl, int idx)
{
    smma16(acc, al, bh, idx);
    smma16(acc, ah, bl, idx);
    smma16(acc, ah, bh, idx);
}
// 16 consecutive K values of one B column as a sparse-product fragment: bf16 in memory, or fp32 split into hi and lo
__device__ __forceinline__ u32x8 sp_load_bf16(const void *p) { return sp_join(*(const u32x4 *)p, *((const u32x4 *)p + 1)); }
__device__ __forceinline__ void sp_load_split(const float *p, u32x8 &hi, u32x8 &lo)
{
    u32x4 h0, l0, h1, l1;
    split8(*(const f32x4 *)p, *(const f32x4 *)(p + 4), h0, l0);
    split8(*(const f32x4 *)(p + 8), *(const f32x4 *)(p + 12), h1, l1);
    hi = sp_join(h0, h1); lo = sp_join(l0, l1);
}

// The bias / peephole sums a backward workgroup has formed over its time steps and sequences for unit `unit` of direction d
// (v[0..3]: gate deltas n, i, f, o; v[4..6]: peephole terms i, f, o) leave for the gradient: one atomic per (gate, unit) and
// workgroup, or -- deterministic mode -- one plain store into the workgroup's slot of p.gpart (LstmRec, cn_internal.h).
template <typename REC>
__device__ __forceinline__ void lstm_grad_sums_out(const REC &p, int HP, int d, int unit, const float (&v)[7])
{
    if (p.gpart) {
        float *slot = p.gpart + (size_t)blockIdx.x * (size_t)(7 * p.dirs * HP);
#pragma unroll
        for (int g = 0; g < 4; ++g) slot[(d * HP + unit) * 4 + g] = p.bias * v[g];
#pragma unroll
        for (int g = 0; g < 3; ++g) slot[4 * p.dirs * HP + (d * 3 + g) * HP + unit] = v[4 + g];
        return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) atomicAdd(&p.dbias[(d * HP + unit) * 4 + g], p.bias * v[g]);
#pragma unroll
    for (int g = 0; g < 3; ++g) atomicAdd(&p.dpeep[(d * 3 + g) * HP + unit], v[4 + g]);
}

// workgroup barrier that orders LDS traffic only: global prefetch loads and the activation stores
// stay in flight across it (a __syncthreads() would drain vmcnt every step)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace cn

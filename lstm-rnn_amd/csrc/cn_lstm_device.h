// Device helpers shared by the recurrent LSTM kernels (cn_lstm.hip, cn_lstm_cluster.hip): the reference's
// activation functions, the 16x16 MFMA step and the LDS-only workgroup barrier.
#pragma once

#include <hip/hip_runtime.h>

namespace cn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

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

// workgroup barrier that orders LDS traffic only: global prefetch loads and the activation stores
// stay in flight across it (a __syncthreads() would drain vmcnt every step)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace cn

/*
 * currennt_hip_debug.h -- kernel-level test hooks of libcurrennt_hip.so.
 *
 * Not part of the drop-in boundary: these run one GEMM kernel on host-provided fp32 matrices (converted to
 * the context's operand type on the device) so the tests can check the MFMA kernels in isolation against
 * the oracle's helpers::Matrix restatement (helpers/Matrix.cu:41-183).
 */
#ifndef CURRENNT_HIP_DEBUG_H
#define CURRENNT_HIP_DEBUG_H

#include "currennt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* C[M][N] = act(A[M][K] * B[N][K]^T + bias[N]);  K % 8 == 0, N % 32 == 0;  act: 0 tanh, 1 logistic, 2 identity;
 * act | 0x100: the kernel writes the fp32 result AND its operand-type copy (what a hidden feed-forward layer asks for) and the
 * copy is returned (widened to float); act | 0x200: the copy alone */
int cn_dbg_gemm_nt(cn_ctx *ctx, const float *A, const float *B, float *C, int M, int N, int K,
                   const float *bias, int act);
/* C[M][N] = A[K][M]^T * B[K][N];  M % 32 == 0, N % 32 == 0 */
int cn_dbg_gemm_tn(cn_ctx *ctx, const float *A, const float *B, float *C, int M, int N, int K);
/* The row map of the fraction that is loaded (no counterpart in the reference, which multiplies every frame of a fraction:
 * LstmLayer.cu:771-786): out[0] frames whose rows the N-wide products compute, out[1] dummy frames whose rows they fill with
 * bias / 0 instead, out[2] T x (padded) parallel sequences.  [sync] */
int cn_dbg_row_map_counts(cn_ctx *ctx, int out[3]);
/* Loads of this context that found their fraction announced and re-laid out (cn_fraction_prefetch / _resident) and only exchanged
 * buffers. */
int cn_dbg_prefetch_hits(cn_ctx *ctx, int *hits);

#ifdef __cplusplus
}
#endif
#endif

// Bandwidth-bound helper kernels around the GEMMs and the recurrent kernels: weight packing,
// gradient unpacking, softmax / loss rows, optimizer step, layout conversion.  gfx950 only.
//
// Reference functors restated here (currennt_lib/src):
//   layers/FeedForwardLayer.cu:69-80   ComputeDeltaFn            -> ff_delta_kernel
//   layers/FeedForwardLayer.cu:82-102  ComputeBiasWeightUpdateFn -> colsum_kernel
//   layers/SoftmaxLayer.cu:45-160      offset/exp/sum/normalise  -> softmax_fwd_kernel (one pass)
//   layers/SoftmaxLayer.cu:162-219     error offset / errors     -> softmax_bwd_kernel (one pass)
//   layers/MulticlassClassificationLayer.cu:48-135               -> mcc_rows_kernel, mcc_backward_kernel
//   layers/SsePostOutputLayer.cu:39-88 and the other post output layers -> post_rows_kernel, post_backward_kernel
//   optimizers/SteepestDescentOptimizer.cu:39-59 UpdateWeightFn  -> sgd_kernel
#include "cn_internal.h"

#include <float.h>

namespace cn {

#define NL_MIN 1.1754944e-038f
#define NL_MAX 3.4028235e+038f
#define NL_EXPLIMIT 88.722839f
#define NL_LOGZERO (-1e30f)

template <bool F32> __device__ __forceinline__ void st_op(void *base, long idx, float v)
{
    if constexpr (F32) ((float *)base)[idx] = v; else ((__bf16 *)base)[idx] = (__bf16)v;
}

// reference feature index of a padded column of the preceding layer's output, -1 for padding.
// An LSTM layer stores direction d at columns [d*Hp, d*Hp+H); every other layer is dense.
__device__ __forceinline__ int unpad_col(int pc, int P, int prevH, int prevHp, int prevDirs)
{
    if (prevH == 0) return pc < P ? pc : -1;
    int dd = pc / prevHp, jj = pc % prevHp;
    return (dd < prevDirs && jj < prevH) ? dd * prevH + jj : -1;
}
__device__ __forceinline__ int pad_col(int i, int prevH, int prevHp)
{
    if (prevH == 0) return i;
    return (i / prevH) * prevHp + (i % prevH);
}

// ---------------------------------------------------------------------------------------------
// LSTM weight packing (flat layout: LstmLayer.hpp:36-55, LstmLayer.cu:535-541,583-596)
// ---------------------------------------------------------------------------------------------
// The weight at flat index fi as the operand copies should see it: as stored, or (upd) after the momentum-SGD step, which this
// thread then also writes back (same arithmetic as sgd_kernel: separate multiplies and adds, no contraction).
// UPD = 2: the gradient of this weight is still in its PACKED accumulator `gp` (scaled by gscale: the bias gradient of a
// feed-forward layer is bias * column sum, FeedForwardLayer.cu:94-100): it is read, cleared for the next backward pass and
// written to the flat weightUpdates on the way (what lstm_unpack_kernel / ff_unpack_kernel do in the unfused sequence).
struct PackUpd { float *w_rw; const float *wu; float *wd; float lr, mom; float *wu_rw; };
// deterministic mode: the partial sums of this entry, added in order -- ((p0 + p1) + p2) + ..., exactly fold_kernel's sum
// (sixteen loads in flight per batch: the bias / peephole sums have one partial per backward workgroup -- 52 on the headline --
// and four at a time made the launch 11 us longer, a chain of thirteen L2 round trips per thread)
__device__ __forceinline__ float pack_fold(const PackFold &f, long off)
{
    float *p = f.part + off;
    float t = 0.f;
    for (int s = 0; s < f.nparts; s += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = s + k < f.nparts ? p[(long)k * f.stride] : 0.f;
        if (f.clear) {
#pragma unroll
            for (int k = 0; k < 16; ++k) if (s + k < f.nparts) p[(long)k * f.stride] = 0.f;
        }
        if (s == 0) t = v[0]; else t += v[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) if (s + k < f.nparts) t += v[k];      // (never adds a padding zero: -0 + 0 would change the sign)
        p += 16 * f.stride;
    }
    return t;
}
template <int UPD>
__device__ __forceinline__ float pack_fetch(const float *w, const PackUpd &u, long fi, float *gp = nullptr, float gscale = 1.0f,
                                            const PackFold *fold = nullptr, long foff = 0)
{
    if constexpr (UPD != 0) {
        float g;
        if constexpr (UPD == 3) {          // deterministic mode: this launch adds the stored partial sums itself (a variant of its own:
            if (fold->nparts) g = pack_fold(*fold, foff);        // with the loop inside the UPD = 2 code every launch of the default mode
            else { g = *gp; *gp = 0.f; }                          // paid ~11 us for it); the packed accumulator itself was never written
            if (gscale != 1.0f) g = __fmul_rn(gscale, g);
            u.wu_rw[fi] = g;
        } else if constexpr (UPD == 2) { g = *gp; *gp = 0.f; if (gscale != 1.0f) g = __fmul_rn(gscale, g); u.wu_rw[fi] = g; }
        else g = u.wu[fi];
        const float dl = __fsub_rn(__fmul_rn(u.mom, u.wd[fi]), __fmul_rn(u.lr, g));          // SteepestDescentOptimizer.cu:51
        u.wd[fi] = dl;
        const float v = __fadd_rn(u.w_rw[fi], dl);                                            // :55
        u.w_rw[fi] = v;
        return v;
    } else return w[fi];
}
struct PackGrad { float *g_in, *g_rec, *g_bias, *g_peep; PackFold f_in, f_rec[2], f_bias; };

// (first / count: the workgroups [first, first + count) of the launch work on this layer: pack_group_kernel)
template <bool F32, int UPD = 0>
__device__ __forceinline__ void lstm_pack_body(const LstmGeom &g, float bias, const float *w, void *Win, void *WinT,
                                               void *Wrec, void *WrecT, float *bias_p, float *peep_p, int first, int count,
                                               const PackUpd &upd = PackUpd{}, const PackGrad &pg = PackGrad{})
{
    const int P = g.P, Pp = g.Pp, L = g.L, H = g.H, Hp = g.Hp, dirs = g.dirs;
    const long R = (long)dirs * 4 * Hp;                 // packed gate rows
    const long nIn = R * Pp, nRec = (long)dirs * 4 * Hp * Hp, nB = (long)dirs * 4 * Hp, nPe = (long)dirs * 3 * Hp;
    const long total = nIn + nRec + nB + nPe;
    for (long idx = (blockIdx.x - first) * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)count * blockDim.x) {
        if (idx < nIn) {
            // packed gate row r = (d*Hp + j)*4 + gate: a lane of the recurrent kernels moves n/i/f/o of
            // one unit with a single 16-byte access
            const int r = idx / Pp, pc = idx % Pp;
            const int d = r / (4 * Hp), j = (r / 4) % Hp, gg = r % 4;
            const int i = unpad_col(pc, P, g.prevH, g.prevHp, g.prevDirs);
            float v = 0.f;
            if (j < H && i >= 0) v = pack_fetch<UPD>(w, upd, (long)gg * L * P + (long)d * H * P + (long)j * P + i, pg.g_in + (long)r * Pp + pc, 1.0f,
                                                     &pg.f_in, (long)r * Pp + pc);   // dWin[r][pc]
            st_op<F32>(Win, (long)r * Pp + pc, v);
            st_op<F32>(WinT, (long)pc * R + r, v);
        } else if (idx < nIn + nRec) {
            const long k = idx - nIn;
            const int d = k / (4L * Hp * Hp), rem = k % (4L * Hp * Hp);
            const int gg = rem / (Hp * Hp), j = (rem / Hp) % Hp, i = rem % Hp;
            float v = 0.f;
            if (j < H && i < H)
                v = pack_fetch<UPD>(w, upd, 4L * L * (P + 1) + (long)gg * L * H + (long)d * H * H + (long)j * H + i,
                                    pg.g_rec + ((long)d * 4 * Hp + 4 * j + gg) * Hp + i, 1.0f,
                                    &pg.f_rec[d & 1], (long)(4 * j + gg) * Hp + i);                                          // dWrec[d][4j + g][i]
            st_op<F32>(Wrec, ((long)d * 4 * Hp + gg * Hp + j) * Hp + i, v);
            st_op<F32>(WrecT, ((long)d * Hp + i) * 4 * Hp + 4 * j + gg, v);
        } else if (idx < nIn + nRec + nB) {
            const int k = idx - nIn - nRec;
            const int d = k / (4 * Hp), j = (k / 4) % Hp, gg = k % 4;
            bias_p[k] = (j < H) ? bias * pack_fetch<UPD>(w, upd, 4L * L * P + gg * L + d * H + j, pg.g_bias + k, 1.0f, &pg.f_bias, k) : 0.f;   // LstmLayer.cu:97-100
        } else {
            const int k = idx - nIn - nRec - nB;
            const int d = k / (3 * Hp), pp = (k / Hp) % 3, j = k % Hp;
            peep_p[k] = (j < H) ? pack_fetch<UPD>(w, upd, 4L * L * (P + 1) + 4L * L * H + pp * L + d * H + j, pg.g_peep + k, 1.0f,
                                                  &pg.f_bias, nB + k) : 0.f;      // (a slot: the bias sums, then the peephole sums)
        }
    }
}
template <bool F32>
__global__ void lstm_pack_kernel(LstmGeom g, float bias, const float *w, void *Win, void *WinT,
                                 void *Wrec, void *WrecT, float *bias_p, float *peep_p)
{
    lstm_pack_body<F32>(g, bias, w, Win, WinT, Wrec, WrecT, bias_p, peep_p, 0, gridDim.x);
}

void launch_lstm_pack(hipStream_t s, bool f32, const LstmGeom &g, float bias, const float *w,
                      void *Win, void *WinT, void *Wrec, void *WrecT, float *bias_p, float *peep_p)
{
    long total = (long)g.dirs * 4 * g.Hp * (g.Pp + g.Hp + 1) + (long)g.dirs * 3 * g.Hp;
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    if (f32) hipLaunchKernelGGL(lstm_pack_kernel<true>, dim3(blocks), dim3(256), 0, s, g, bias, w, Win, WinT, Wrec, WrecT, bias_p, peep_p);
    else     hipLaunchKernelGGL(lstm_pack_kernel<false>, dim3(blocks), dim3(256), 0, s, g, bias, w, Win, WinT, Wrec, WrecT, bias_p, peep_p);
}

// packed fp32 gradients -> flat weightUpdates (same layout as the weights, LstmLayer.cu:577-581)
// Every packed accumulator is cleared right after it is read, so the next backward pass finds zeros without
// a separate memset (padded entries only ever receive exact zeros).
__global__ void lstm_unpack_kernel(LstmGeom g, float *dWin, float *dWrec, float *dbias, float *dpeep, float *wu)
{
    const int P = g.P, Pp = g.Pp, L = g.L, H = g.H, Hp = g.Hp;
    const long nIn = 4L * L * P, nB = 4L * L, nRec = 4L * L * H, nPe = 3L * L;
    const long total = nIn + nB + nRec + nPe;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        float v;
        if (idx < nIn) {
            const int gg = idx / ((long)L * P), rem = idx % ((long)L * P);
            const int blk = rem / P, i = rem % P, d = blk / H, j = blk % H;
            float *src = &dWin[(((long)d * Hp + j) * 4 + gg) * Pp + pad_col(i, g.prevH, g.prevHp)];
            v = *src; *src = 0.f;
        } else if (idx < nIn + nB) {
            const int k = idx - nIn, gg = k / L, blk = k % L, d = blk / H, j = blk % H;
            float *src = &dbias[(d * Hp + j) * 4 + gg];
            v = *src; *src = 0.f;
        } else if (idx < nIn + nB + nRec) {
            const long k = idx - nIn - nB;
            const int gg = k / ((long)L * H), rem = k % ((long)L * H);
            const int blk = rem / H, i = rem % H, d = blk / H, j = blk % H;
            float *src = &dWrec[((long)d * 4 * Hp + 4 * j + gg) * Hp + i];
            v = *src; *src = 0.f;
        } else {
            const int k = idx - nIn - nB - nRec, pp = k / L, blk = k % L, d = blk / H, j = blk % H;
            float *src = &dpeep[(d * 3 + pp) * Hp + j];
            v = *src; *src = 0.f;
        }
        wu[idx] = v;
    }
}

void launch_lstm_unpack_grads(hipStream_t s, const LstmGeom &g, float *dWin, float *dWrec, float *dbias, float *dpeep, float *wu, hipEvent_t done)
{
    long total = (long)g.L * (4 * (g.P + 1) + 4 * g.H + 3);
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipExtLaunchKernelGGL(lstm_unpack_kernel, dim3(blocks), dim3(256), 0, s, nullptr, done, 0, g, dWin, dWrec, dbias, dpeep, wu);
}

// ---------------------------------------------------------------------------------------------
// feed-forward weight packing (flat layout: [j][i] P x L column-major then L bias weights,
// FeedForwardLayer.cu:148,160)
// ---------------------------------------------------------------------------------------------
template <bool F32, int UPD = 0>
__device__ __forceinline__ void ff_pack_body(const FfGeom &g, float bias, const float *w, void *W, void *WT, float *bias_p, int first, int count,
                                             const PackUpd &upd = PackUpd{}, const PackGrad &pg = PackGrad{})
{
    const long nW = (long)g.Lp * g.Pp, total = nW + g.Lp;
    for (long idx = (blockIdx.x - first) * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)count * blockDim.x) {
        if (idx < nW) {
            const int j = idx / g.Pp, pc = idx % g.Pp;
            const int i = unpad_col(pc, g.P, g.prevH, g.prevHp, g.prevDirs);
            float v = (j < g.L && i >= 0) ? pack_fetch<UPD>(w, upd, (long)j * g.P + i, pg.g_in + (long)j * g.Pp + pc, 1.0f, &pg.f_in, (long)j * g.Pp + pc) : 0.f;
            st_op<F32>(W, (long)j * g.Pp + pc, v);
            st_op<F32>(WT, (long)pc * g.Lp + j, v);
        } else {
            const int j = idx - nW;
            bias_p[j] = (j < g.L) ? bias * pack_fetch<UPD>(w, upd, (long)g.L * g.P + j, pg.g_bias + j, bias, &pg.f_bias, j) : 0.f;            // FeedForwardLayer.cu:59
        }
    }
}
template <bool F32>
__global__ void ff_pack_kernel(FfGeom g, float bias, const float *w, void *W, void *WT, float *bias_p)
{
    ff_pack_body<F32>(g, bias, w, W, WT, bias_p, 0, gridDim.x);
}

// Every trainable layer's operand copies in ONE launch (after cn_sgd_update_all): four launches of 4-9 us each were
// either on the critical path (the first layer's) or cost a fork event, a side stream and a wait (the others').
template <bool F32>
__global__ void pack_group_kernel(PackGroup grp)
{
    int i = 0;
#pragma unroll
    for (int k = 1; k < PACK_GROUP_MAX; ++k) if (k < grp.n && (int)blockIdx.x >= grp.first[k]) i = k;
    const PackItem it = grp.item[i];     // (a copy: through a reference into the argument struct hipcc re-loads fields inside the loops)
    const int count = (i + 1 < grp.n ? grp.first[i + 1] : (int)gridDim.x) - grp.first[i];
    if (it.update) {      // cn_sgd_update_all: the weight update rides on the pack (one launch instead of two on the critical tail)
        const PackUpd upd{it.w_rw, it.wu, it.wd, it.lr, it.mom, it.wu_rw};
        if (it.update == 2) {   // armed update: ... and so does the unpacking of the gradient
            const PackGrad pg{it.g_in, it.g_rec, it.g_bias, it.g_peep, {}, {{}, {}}, {}};
            if (it.lstm) lstm_pack_body<F32, 2>(it.lg, it.bias, it.w, it.Win, it.WinT, it.Wrec, it.WrecT, it.bias_p, it.peep_p, grp.first[i], count, upd, pg);
            else         ff_pack_body<F32, 2>(it.fg, it.bias, it.w, it.Win, it.WinT, it.bias_p, grp.first[i], count, upd, pg);
            return;
        }
        if (it.update == 3) {   // ... in deterministic mode, from the partial sums the producers stored (PackFold)
            const PackGrad pg{it.g_in, it.g_rec, it.g_bias, it.g_peep, it.f_in, {it.f_rec[0], it.f_rec[1]}, it.f_bias};
            if (it.lstm) lstm_pack_body<F32, 3>(it.lg, it.bias, it.w, it.Win, it.WinT, it.Wrec, it.WrecT, it.bias_p, it.peep_p, grp.first[i], count, upd, pg);
            else         ff_pack_body<F32, 3>(it.fg, it.bias, it.w, it.Win, it.WinT, it.bias_p, grp.first[i], count, upd, pg);
            return;
        }
        if (it.lstm) lstm_pack_body<F32, 1>(it.lg, it.bias, it.w, it.Win, it.WinT, it.Wrec, it.WrecT, it.bias_p, it.peep_p, grp.first[i], count, upd);
        else         ff_pack_body<F32, 1>(it.fg, it.bias, it.w, it.Win, it.WinT, it.bias_p, grp.first[i], count, upd);
        return;
    }
    if (it.lstm) lstm_pack_body<F32>(it.lg, it.bias, it.w, it.Win, it.WinT, it.Wrec, it.WrecT, it.bias_p, it.peep_p, grp.first[i], count);
    else         ff_pack_body<F32>(it.fg, it.bias, it.w, it.Win, it.WinT, it.bias_p, grp.first[i], count);
}
void launch_pack_group(hipStream_t s, bool f32, PackGroup &grp, hipEvent_t done)
{
    int blocks = 0;
    for (int i = 0; i < grp.n; ++i) {
        const PackItem &it = grp.item[i];
        const long total = it.lstm ? (long)it.lg.dirs * 4 * it.lg.Hp * (it.lg.Pp + it.lg.Hp + 1) + (long)it.lg.dirs * 3 * it.lg.Hp
                                   : (long)it.fg.Lp * it.fg.Pp + it.fg.Lp;
        int b = (int)((total + 255) / 256); if (b > 1024) b = 1024;
        grp.first[i] = blocks; blocks += b;
    }
    if (blocks == 0) return;
    if (f32) hipExtLaunchKernelGGL(pack_group_kernel<true>, dim3(blocks), dim3(256), 0, s, nullptr, done, 0, grp);
    else     hipExtLaunchKernelGGL(pack_group_kernel<false>, dim3(blocks), dim3(256), 0, s, nullptr, done, 0, grp);
}
void launch_ff_pack(hipStream_t s, bool f32, const FfGeom &g, float bias, const float *w, void *W, void *WT, float *bias_p)
{
    long total = (long)g.Lp * g.Pp + g.Lp;
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    if (f32) hipLaunchKernelGGL(ff_pack_kernel<true>, dim3(blocks), dim3(256), 0, s, g, bias, w, W, WT, bias_p);
    else     hipLaunchKernelGGL(ff_pack_kernel<false>, dim3(blocks), dim3(256), 0, s, g, bias, w, W, WT, bias_p);
}

__global__ void ff_unpack_kernel(FfGeom g, float bias, float *dW, float *colsum, float *wu)
{
    const long nW = (long)g.L * g.P, total = nW + g.L;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        if (idx < nW) {
            const int j = idx / g.P, i = idx % g.P;
            float *src = &dW[(long)j * g.Pp + pad_col(i, g.prevH, g.prevHp)];
            wu[idx] = *src; *src = 0.f;
        } else {
            wu[idx] = bias * colsum[idx - nW];                                     // FeedForwardLayer.cu:94-100
            colsum[idx - nW] = 0.f;
        }
    }
}
void launch_ff_unpack_grads(hipStream_t s, const FfGeom &g, float bias, float *dW, float *colsum, float *wu, hipEvent_t done)
{
    long total = (long)g.L * (g.P + 1);
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipExtLaunchKernelGGL(ff_unpack_kernel, dim3(blocks), dim3(256), 0, s, nullptr, done, 0, g, bias, dW, colsum, wu);
}

// ---------------------------------------------------------------------------------------------
// layout conversion
// ---------------------------------------------------------------------------------------------
template <bool F32>
__global__ void pad_convert_kernel(const float *src, int N, int P, void *dst, int Pp)
{
    const long total = (long)N * Pp;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long n = idx / Pp; const int c = idx % Pp;
        st_op<F32>(dst, idx, c < P ? src[n * P + c] : 0.f);
    }
}
void launch_pad_convert(hipStream_t s, bool f32, const float *src, int N, int P, void *dst, int Pp)
{
    long total = (long)N * Pp; if (total <= 0) return;
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    if (f32) hipLaunchKernelGGL(pad_convert_kernel<true>, dim3(blocks), dim3(256), 0, s, src, N, P, dst, Pp);
    else     hipLaunchKernelGGL(pad_convert_kernel<false>, dim3(blocks), dim3(256), 0, s, src, N, P, dst, Pp);
}

// One launch for a fraction that is already in HBM (cn_fraction_load_resident): patTypes, target classes or
// target rows and the input patterns go from the caller's [T][PS] layout to the padded [T][PSp] device layout,
// the inputs converted to the operand type on the way (instead of three strided copies + pad_convert).
template <bool F32>
__global__ void fraction_load_kernel(int T, int PS, int PSp, const char *pat, char *dpat, const int *tcls, int *dtcls,
                                     const float *tgt, float *dtgt, int W, const float *in, int P, void *dst, int Pp)
{
    const long N = (long)T * PSp;
    const long tid = blockIdx.x * (long)blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
    for (long n = tid; n < N; n += nth) {
        const long t = n / PSp; const int sl = n % PSp;
        const bool real = sl < PS;
        dpat[n] = real ? pat[t * PS + sl] : 0;
        if (tcls) dtcls[n] = real ? tcls[t * PS + sl] : -1;
    }
    if (tgt)
        for (long idx = tid; idx < N * W; idx += nth) {
            const long n = idx / W; const int j = idx % W;
            const long t = n / PSp; const int sl = n % PSp;
            dtgt[idx] = sl < PS ? tgt[(t * PS + sl) * W + j] : 0.f;
        }
    for (long idx = tid; idx < N * Pp; idx += nth) {
        const long n = idx / Pp; const int c = idx % Pp;
        const long t = n / PSp; const int sl = n % PSp;
        st_op<F32>(dst, idx, (sl < PS && c < P) ? in[(t * PS + sl) * P + c] : 0.f);
    }
}
// The row map of a fraction (GemmNT::rowmap): the rows of the real frames and of the dummy ones, each in ascending order, and their
// counts.  Dummy = patType NONE at a time step >= the fraction's shortest sequence: only there do the recurrent kernels force
// y = 0 / deltas = 0 (checkPatType, LstmLayer.cu:796,868) -- in front of it an empty slot is computed like any other, by the
// reference too, and counts as real here.  One workgroup: every thread counts its chunk, a scan over the 1024 counts (shuffles inside a wave, 16 wave
// sums through LDS), every thread writes its chunk's rows.  Runs behind the re-layout on the same stream (beside the backward pass for a prefetched fraction).
__global__ __launch_bounds__(1024) void rowmap_kernel(const char *pat, int N, int *rm, int maxN, int unchecked, int staged)
{
    extern __shared__ __attribute__((aligned(16))) char spat[];
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = (N + 1023) / 1024, a = min(N, tid * chunk), b = min(N, a + chunk);
    // (a thread's chunk is a run of single bytes: read from memory one by one it is a chain of ~2 x 18 dependent loads, 15 us for
    // the headline's 18 000 frames; staged through LDS in 16-byte pieces first when the fraction fits)
    const char *src = pat;
    if (staged) {
        for (int i = tid * 16; i < N; i += 1024 * 16) {
            if (i + 16 <= N && ((size_t)(pat + i) & 15) == 0) *(uint4 *)(spat + i) = *(const uint4 *)(pat + i);
            else for (int j = i; j < min(N, i + 16); ++j) spat[j] = pat[j];
        }
        __syncthreads();
        src = spat;
    }
    int n = 0;
    for (int i = a; i < b; ++i) n += src[i] != 0 || i < unchecked;
    int x = n;                                 // inclusive scan over the wave's 64 counts, then over the 16 waves
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int v = wsum[w]; total += v; if (w < wave) base += v; }
    int r = base + x - n, d = a - r;           // real / dummy rows in front of this chunk
    int *real = rm + 4, *dummy = rm + 4 + maxN;
    for (int i = a; i < b; ++i) { if (src[i] != 0 || i < unchecked) real[r++] = i; else dummy[d++] = i; }
    if (tid == 0) { rm[0] = total; rm[1] = N - total; }
}
void launch_rowmap(hipStream_t s, const char *dpat, int N, int *rm, int maxN, int unchecked)
{
    if (!rm || N <= 0) return;
    const int staged = N <= 60 * 1024;
    hipLaunchKernelGGL(rowmap_kernel, dim3(1), dim3(1024), staged ? (N + 15) & ~15 : 0, s, dpat, N, rm, maxN, unchecked, staged);
}
void launch_fraction_load(hipStream_t s, bool f32, int T, int PS, int PSp, const char *pat, char *dpat, const int *tcls, int *dtcls,
                          const float *tgt, float *dtgt, int W, const float *in, int P, void *dst, int Pp, int *rm, int maxN, int Tmin)
{
    long total = (long)T * PSp * Pp; if (total <= 0) return;
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    if (f32) hipLaunchKernelGGL(fraction_load_kernel<true>, dim3(blocks), dim3(256), 0, s, T, PS, PSp, pat, dpat, tcls, dtcls, tgt, dtgt, W, in, P, dst, Pp);
    else     hipLaunchKernelGGL(fraction_load_kernel<false>, dim3(blocks), dim3(256), 0, s, T, PS, PSp, pat, dpat, tcls, dtcls, tgt, dtgt, W, in, P, dst, Pp);
    launch_rowmap(s, dpat, T * PSp, rm, maxN, Tmin * PSp);
}

template <bool BF16>
__global__ void unpad_kernel(const void *src, long ld, int col0, int cstride, int N, int L, float *dst, long ldd, int dcol0, int PS, int PSp)
{
    const long total = (long)N * L;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long n = idx / L; const int j = idx % L;
        const long ns = (n / PS) * PSp + n % PS;          // device row of host pattern n
        float v;
        if constexpr (BF16) v = (float)((const __bf16 *)src)[ns * ld + col0 + (long)j * cstride];
        else v = ((const float *)src)[ns * ld + col0 + (long)j * cstride];
        dst[n * ldd + dcol0 + j] = v;
    }
}
void launch_unpad(hipStream_t s, bool src_is_bf16, const void *src, long ld, int col0, int cstride, int N, int L, float *dst, long ldd, int dcol0, int PS, int PSp)
{
    long total = (long)N * L; if (total <= 0) return;
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    if (src_is_bf16) hipLaunchKernelGGL(unpad_kernel<true>, dim3(blocks), dim3(256), 0, s, src, ld, col0, cstride, N, L, dst, ldd, dcol0, PS, PSp);
    else             hipLaunchKernelGGL(unpad_kernel<false>, dim3(blocks), dim3(256), 0, s, src, ld, col0, cstride, N, L, dst, ldd, dcol0, PS, PSp);
}

__global__ void pad_f32_kernel(const float *src, int N, int L, float *dst, long ld, int prevH, int prevHp, int PS, int PSp)
{
    const long total = (long)N * L;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long n = idx / L; const int j = idx % L;
        dst[((n / PS) * PSp + n % PS) * ld + pad_col(j, prevH, prevHp)] = src[idx];
    }
}
void launch_pad_f32(hipStream_t s, const float *src, int N, int L, float *dst, long ld, int prevH, int prevHp, int PS, int PSp)
{
    long total = (long)N * L; if (total <= 0) return;
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pad_f32_kernel, dim3(blocks), dim3(256), 0, s, src, N, L, dst, ld, prevH, prevHp, PS, PSp);
}

// ---------------------------------------------------------------------------------------------
// feed-forward delta and bias gradient
// ---------------------------------------------------------------------------------------------
template <bool F32>
__global__ void ff_delta_kernel(int act, const float *y, float *err, void *delta_op, int N, int L, int Lp)
{
    const long total = (long)N * Lp;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int j = idx % Lp;
        float dl = 0.f;
        if (j < L) {
            const float yy = y[idx], e = err[idx];
            float dv = 1.0f;                                         // Identity.cuh:38-41
            if (act == ACT_TANH) dv = 1.0f - yy * yy;                // Tanh.cuh:38-41
            else if (act == ACT_LOGISTIC) dv = yy * (1.0f - yy);     // Logistic.cuh:46-49
            dl = dv * e;
        }
        err[idx] = dl;
        st_op<F32>(delta_op, idx, dl);
    }
}
void launch_ff_delta(hipStream_t s, bool f32, int act, const float *y, float *err, void *delta_op, int N, int L, int Lp)
{
    long total = (long)N * Lp; if (total <= 0) return;
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    if (f32) hipLaunchKernelGGL(ff_delta_kernel<true>, dim3(blocks), dim3(256), 0, s, act, y, err, delta_op, N, L, Lp);
    else     hipLaunchKernelGGL(ff_delta_kernel<false>, dim3(blocks), dim3(256), 0, s, act, y, err, delta_op, N, L, Lp);
}

// ---------------------------------------------------------------------------------------------
// deterministic mode: partial sums (split-K partial products, per-workgroup bias / peephole / column sums) are STORED by their
// producers and added here in a fixed order -- ((p0 + p1) + p2) + ... -- by one thread per output, so a gradient no longer
// depends on the order in which workgroups retire (the reference sums serially, one logical thread per weight:
// LstmLayer.cu:289-512, FeedForwardLayer.cu:82-102).
// ---------------------------------------------------------------------------------------------
struct FoldGroup { FoldItem it[FOLD_MAX]; int first_block[FOLD_MAX + 1]; };
__global__ __launch_bounds__(256) void fold_kernel(FoldGroup g)
{
    int gi = 0;
#pragma unroll
    for (int i = 1; i < FOLD_MAX; ++i) if ((int)blockIdx.x >= g.first_block[i]) gi = i;
    const FoldItem f = g.it[gi];
    const long total = (long)f.rows * f.cols;
    const long base = (long)(blockIdx.x - g.first_block[gi]) * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long e = base + 256 * k;
        if (e >= total) break;
        const long idx = (e / f.cols) * f.ld + e % f.cols;
        float *p = f.part + idx;
        float t = 0.f;
        int s = 0;
        for (; s + 4 <= f.nparts; s += 4) {            // four loads in flight, added in order
            const float a = p[0], b = p[f.stride], c = p[2 * f.stride], d = p[3 * f.stride];
            if (f.clear) { p[0] = 0.f; p[f.stride] = 0.f; p[2 * f.stride] = 0.f; p[3 * f.stride] = 0.f; }
            t = s ? t + a : a; t += b; t += c; t += d;
            p += 4 * f.stride;
        }
        for (; s < f.nparts; ++s) {
            const float a = *p;
            if (f.clear) *p = 0.f;
            t = s ? t + a : a;
            p += f.stride;
        }
        f.dst[idx] = f.accumulate ? f.dst[idx] + t : t;
    }
}
void launch_fold(hipStream_t s, const FoldItem *items, int n)
{
    while (n > 0) {
        FoldGroup g{};
        int blocks = 0, m = n < FOLD_MAX ? n : FOLD_MAX;
        for (int i = 0; i < FOLD_MAX; ++i) {
            g.first_block[i] = blocks;
            if (i >= m) { g.first_block[i] = 0x7fffffff; continue; }
            g.it[i] = items[i];
            blocks += (int)(((long)items[i].rows * items[i].cols + 1023) / 1024);
        }
        g.first_block[FOLD_MAX] = blocks;
        if (blocks) hipLaunchKernelGGL(fold_kernel, dim3(blocks), dim3(256), 0, s, g);
        items += m; n -= m;
    }
}

// colsum[j] += sum_n err[n][j]; block = 256 threads = 8 row lanes x 32 columns, rows strided over the grid
// (det_part: the workgroup's sums are stored to its row of det_part instead, launch_colsum folds the rows in workgroup order)
__global__ void colsum_kernel(const float *err, int N, int Lp, float *colsum, float *det_part)
{
    __shared__ float part[8][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    for (int c0 = 0; c0 < Lp; c0 += 32) {
        float sacc = 0.f;
        for (long n = (long)blockIdx.x * 8 + ry; n < N; n += (long)gridDim.x * 8) sacc += err[n * Lp + c0 + cx];
        part[ry][cx] = sacc;
        __syncthreads();
        if (ry == 0) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += part[r][cx];
            if (det_part) det_part[(long)blockIdx.x * Lp + c0 + cx] = t;
            else atomicAdd(&colsum[c0 + cx], t);
        }
        __syncthreads();
    }
}
static void fold_now_or_later(hipStream_t s, const FoldItem &f, FoldItem *fold_out)
{
    if (fold_out) *fold_out = f; else launch_fold(s, &f, 1);
}
void launch_colsum(hipStream_t s, const float *err, int N, int Lp, float *colsum, float *det_part, FoldItem *fold_out)
{
    if (fold_out) fold_out->nparts = 0;
    if (N <= 0) return;
    int blocks = (N + 63) / 64; if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, s, err, N, Lp, colsum, det_part);
    if (det_part) fold_now_or_later(s, FoldItem{colsum, det_part, (long)Lp, blocks, 1, Lp, Lp, 1, 0}, fold_out);
}

// ---------------------------------------------------------------------------------------------
// softmax rows: one 64-lane wave per pattern
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float safe_exp(float x)      // helpers/safeExp.cuh:31-40
{
    if (x <= NL_LOGZERO) return 0.f;
    if (x >= NL_EXPLIMIT) return NL_MAX;
    return expf(x);
}

// The wide-row kernels of the bf16 throughput mode (FAST): v_exp_f32 and one reciprocal per row instead of libm's expf and an
// IEEE division per element -- at 8000 classes x 35 000 patterns those are 0.3 ms of VALU work per pass, more than the row's
// memory time.  (The LSTM kernels of that mode evaluate their activations the same way.)  One definition of a posterior per mode:
// every kernel that produces or re-produces one calls softmax_term with the same arguments.
template <bool FAST> __device__ __forceinline__ float softmax_exp(float x)
{
    if constexpr (!FAST) return safe_exp(x);
    return __builtin_amdgcn_exp2f(fminf(x, NL_EXPLIMIT) * 1.44269504088896341f);      // (v_exp_f32 underflows to 0 by itself)
}
// `norm` = sum (exact modes: SoftmaxLayer.cu:152 divides) or 1 / sum (FAST)
template <bool FAST> __device__ __forceinline__ float softmax_norm(float sum) { return FAST ? 1.0f / sum : sum; }
template <bool FAST> __device__ __forceinline__ float softmax_scale(float e, float norm) { return FAST ? e * norm : e / norm; }

// the wave's index within its workgroup as a scalar: what is indexed with it (target class, pattern type, the target's
// posterior) then comes through the scalar cache instead of a vector memory round trip per link of the chain
#ifndef CN_EW_NO_UNIFORM
#define CN_WAVE_ID() __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))
#else
#define CN_WAVE_ID() ((int)(threadIdx.x >> 6))
#endif
// pat[row] for a wave-uniform row through the scalar cache (gfx950 has no scalar byte load: a plain pat[row] is a vector
// memory round trip in front of everything that depends on it)
__device__ __forceinline__ int pat_at(const char *pat, long row)
{
    const uintptr_t a = (uintptr_t)(pat + row);
    const unsigned w = *(const unsigned *)(a & ~(uintptr_t)3);
    return (w >> (8 * (a & 3))) & 0xff;
}

// rowstat (optional): per pattern {log max(FLT_MIN, p_target), 1 if argmax == target} for the multiclass
// post output layer, so that the loss evaluation is a fixed-order reduction of N pairs instead of a second
// pass over the posteriors (MulticlassClassificationLayer.cu:55-68, :77-105)
__global__ void softmax_fwd_kernel(float *y, const char *pat, int N, int L, int Lp, const int *tcls, float2 *rowstat)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + CN_WAVE_ID();
    if (row >= N) return;
    if (pat_at(pat, row) == 0) {                         // SKIP_MARKER path, SoftmaxLayer.cu:58-59
        if (rowstat && lane == 0) rowstat[row] = make_float2(0.f, 0.f);
        return;
    }
    float *r = y + row * Lp;
    float mx = NL_MIN, mn = NL_MAX;                      // :61-62 (max starts at FLT_MIN, quirk Q3)
    for (int j = lane; j < L; j += 64) { float v = r[j]; mx = fmaxf(mx, v); mn = fminf(mn, v); }
    mx = wave_max(mx); mn = wave_min(mn);
    const float offset = 0.5f * (mn + mx);               // :74
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) { float x = safe_exp(r[j] - offset); r[j] = x; sum += x; }
    sum = wave_sum(sum);
    float best = 0.f; int bi = 0;                        // CountCorrectClassificationsFn :89-99
    const int tc = rowstat ? tcls[row] : -1;
    float ptv = 0.f;                                     // posterior of the target class (one lane holds it)
    for (int j = lane; j < L; j += 64) {
        float v = r[j] / sum; r[j] = v;                  // :152
        if (v > best) { best = v; bi = j; }
        if (j == tc) ptv = v;
    }
    if (rowstat) {
        ptv = wave_sum(ptv);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o); int oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (best <= 0.f) bi = 0;
        if (lane == 0) rowstat[row] = tc < 0 ? make_float2(0.f, 0.f) : make_float2(logf(fmaxf(NL_MIN, ptv)), bi == tc ? 1.f : 0.f);
    }
}
// Narrow rows (Lp <= 256, the headline's 183 classes), round 4: SIXTEEN lanes per pattern, four patterns per wave.  Lane c of
// a pattern's group owns the 16-byte column groups c, c + 16, c + 32, c + 48 (v[4g + e] = column 4 (c + 16 g) + e, ascending):
// 16-byte loads and stores instead of dwords, four-step reductions instead of six, a quarter of the waves.  The loads of the row
// are issued beside the pattern's type and target class, not behind them (a padding pattern's row is read and dropped).
// Same expressions per element as softmax_fwd_kernel; the sums associate differently (per lane, then a 16-lane tree).
__device__ __forceinline__ float g16_sum(float v) { v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1); return v; }
__device__ __forceinline__ float g16_max(float v) { v = fmaxf(v, __shfl_xor(v, 8)); v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 2)); return fmaxf(v, __shfl_xor(v, 1)); }
__device__ __forceinline__ float g16_min(float v) { v = fminf(v, __shfl_xor(v, 8)); v = fminf(v, __shfl_xor(v, 4)); v = fminf(v, __shfl_xor(v, 2)); return fminf(v, __shfl_xor(v, 1)); }
template <bool FAST>
__global__ __launch_bounds__(256) void softmax_fwd_rows16_kernel(float *__restrict__ y, const char *__restrict__ pat, int N, int L, int Lp,
                                                                 const int *__restrict__ tcls, float2 *__restrict__ rowstat)
{
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    const int c = threadIdx.x & 15;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool in = row < N;
    const long rr = in ? row : N - 1;
    float *r = y + rr * Lp;
    float v[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int j0 = 4 * (c + 16 * g);
        const f32x4 x = j0 < Lp ? *(const f32x4 *)(r + j0) : f32x4{0.f, 0.f, 0.f, 0.f};
        v[4 * g] = x[0]; v[4 * g + 1] = x[1]; v[4 * g + 2] = x[2]; v[4 * g + 3] = x[3];
    }
    const bool real = in && pat[rr] != 0;                // SKIP_MARKER path, SoftmaxLayer.cu:58-59
    const int tc = rowstat ? tcls[rr] : -1;
    float mx = NL_MIN, mn = NL_MAX;                      // :61-62 (max starts at FLT_MIN, quirk Q3)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int j0 = 4 * (c + 16 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (j0 + e < L) { mx = fmaxf(mx, v[4 * g + e]); mn = fminf(mn, v[4 * g + e]); }
            else v[4 * g + e] = -__builtin_inff();       // (exp gives 0: nothing in the sum, never the argmax, written back as 0)
        }
    }
    mx = g16_max(mx); mn = g16_min(mn);
    const float offset = 0.5f * (mn + mx);               // :74
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { v[k] = softmax_exp<FAST>(v[k] - offset); sum += v[k]; }
    sum = g16_sum(sum);
    const float norm = softmax_norm<FAST>(sum);
    float best = 0.f, ptv = 0.f; int bi = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int j = 4 * (c + 16 * (k >> 2)) + (k & 3);
        const float w = softmax_scale<FAST>(v[k], norm); v[k] = w;            // :152
        if (w > best) { best = w; bi = j; }              // ascending j per lane: first maximum kept
        if (j == tc) ptv = w;
    }
    if (real) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int j0 = 4 * (c + 16 * g);
            if (j0 < Lp) *(f32x4 *)(r + j0) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
        }
    }
    if (rowstat) {
        ptv = g16_sum(ptv);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o); int oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (best <= 0.f) bi = 0;
        if (in && c == 0) rowstat[row] = (!real || tc < 0) ? make_float2(0.f, 0.f) : make_float2(logf(fmaxf(NL_MIN, ptv)), bi == tc ? 1.f : 0.f);
    }
}
// Wide rows (256 < L <= 8192, e.g. 8000 tied states): one workgroup per pattern, the row in registers (VPT values per
// thread), block reductions through LDS -- one read and one write of the posteriors instead of three and two.
constexpr int SMW_VPT = 32;
__device__ __forceinline__ float block_reduce(float v, int op, float *sh)      // op 0 sum, 1 max, 2 min; 256 threads
{
    v = op == 0 ? wave_sum(v) : (op == 1 ? wave_max(v) : wave_min(v));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float a = sh[0], b = sh[1], c = sh[2], d = sh[3];
    return op == 0 ? (a + b) + (c + d) : (op == 1 ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : fminf(fminf(a, b), fminf(c, d)));
}
// STORE = false (the lazy form): the posteriors stay unwritten -- the row keeps its logits and smstat[row] = {offset, sum} lets
// softmax_mcc_bwd_wide_kernel<.., true> / softmax_normalise_wide_kernel recompute exactly the values this kernel would have
// stored (same expression per element), while rowstat (log p_target, correct) is complete as always.
template <bool STORE, bool FAST>
__global__ __launch_bounds__(256) void softmax_fwd_wide_kernel(float *y, const char *pat, int N, int L, int Lp, const int *tcls, float2 *rowstat, float2 *smstat)
{
    __shared__ float sh[4]; __shared__ float shb[4]; __shared__ int shi[4];
    const long row = blockIdx.x;
    const int tid = threadIdx.x;
    if (pat_at(pat, row) == 0) {                         // SKIP_MARKER path, SoftmaxLayer.cu:58-59
        if (rowstat && tid == 0) rowstat[row] = make_float2(0.f, 0.f);
        return;
    }
    float *r = y + row * Lp;
    // a thread owns the four columns 4 (tid + 256 q) .. + 3 of every q (16-byte loads and stores; Lp is a multiple of 32):
    // v[k] is column col(k) = 4 (tid + 256 (k >> 2)) + (k & 3), ascending in k
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    auto col = [&](int k) { return 4 * (tid + 256 * (k >> 2)) + (k & 3); };
    float v[SMW_VPT];
#pragma unroll
    for (int q = 0; q < SMW_VPT / 4; ++q) {
        const int j0 = 4 * (tid + 256 * q);
        const f32x4 x = j0 < Lp ? *(const f32x4 *)(r + j0) : f32x4{0.f, 0.f, 0.f, 0.f};
        v[4 * q] = x[0]; v[4 * q + 1] = x[1]; v[4 * q + 2] = x[2]; v[4 * q + 3] = x[3];
    }
    const int tc = rowstat ? tcls[row] : -1;
    float mx = NL_MIN, mn = NL_MAX;                      // :61-62 (max starts at FLT_MIN, quirk Q3)
    // columns >= L (padding up to Lp, and the groups beyond Lp) become -inf behind the min / max pass: exp gives them 0, they add
    // nothing to the sum, never win the argmax and are written back as the 0 the GEMM left there
#pragma unroll
    for (int q = 0; q < SMW_VPT / 4; ++q) {
        const int j0 = 4 * (tid + 256 * q);
        if (j0 + 3 < L) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { mx = fmaxf(mx, v[4 * q + e]); mn = fminf(mn, v[4 * q + e]); }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (j0 + e < L) { mx = fmaxf(mx, v[4 * q + e]); mn = fminf(mn, v[4 * q + e]); }
                else v[4 * q + e] = -__builtin_inff();
            }
        }
    }
    mx = block_reduce(mx, 1, sh); mn = block_reduce(mn, 2, sh);
    const float offset = 0.5f * (mn + mx);               // :74
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < SMW_VPT; ++k) { v[k] = softmax_exp<FAST>(v[k] - offset); sum += v[k]; }
    sum = block_reduce(sum, 0, sh);
    const float norm = softmax_norm<FAST>(sum);
    float best = 0.f, ptv = 0.f; int bi = 0;
#pragma unroll
    for (int k = 0; k < SMW_VPT; ++k) {
        const int j = col(k);
        const float w = softmax_scale<FAST>(v[k], norm); v[k] = w;            // :152
        if (w > best) { best = w; bi = j; }              // ascending j per thread: first maximum kept
        if (j == tc) ptv = w;
    }
    if constexpr (STORE) {
#pragma unroll
        for (int q = 0; q < SMW_VPT / 4; ++q) {          // (columns L .. Lp - 1 get back what they held)
            const int j0 = 4 * (tid + 256 * q);
            if (j0 < Lp) *(f32x4 *)(r + j0) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        }
    } else if (tid == 0) smstat[row] = make_float2(offset, norm);
    if (rowstat) {
        ptv = block_reduce(ptv, 0, sh);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o); int oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { shb[tid >> 6] = best; shi[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w) if (shb[w] > best || (shb[w] == best && shi[w] < bi)) { best = shb[w]; bi = shi[w]; }
            if (best <= 0.f) bi = 0;
            rowstat[row] = tc < 0 ? make_float2(0.f, 0.f) : make_float2(logf(fmaxf(NL_MIN, ptv)), bi == tc ? 1.f : 0.f);
        }
    }
}

// the posteriors of a lazily forwarded wide softmax layer, on demand: y = exp(z - offset) / sum per element
template <bool FAST>
__global__ __launch_bounds__(256) void softmax_normalise_wide_kernel(float *y, const char *pat, int N, int L, int Lp, const float2 *smstat)
{
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    const long row = blockIdx.x;
    if (pat_at(pat, row) == 0) return;
    const float2 st = smstat[row];
    float *r = y + row * Lp;
    for (int j0 = 4 * threadIdx.x; j0 < Lp; j0 += 1024) {
        f32x4 x = *(const f32x4 *)(r + j0);
#pragma unroll
        for (int e = 0; e < 4; ++e) if (j0 + e < L) x[e] = softmax_scale<FAST>(softmax_exp<FAST>(x[e] - st.x), st.y);
        *(f32x4 *)(r + j0) = x;
    }
}
bool softmax_fwd_can_be_lazy(int L) { return L > 256 && L <= 256 * SMW_VPT; }
void launch_softmax_normalise(hipStream_t s, bool fast, float *y, const char *pat, int N, int L, int Lp, const float *smstat)
{
    if (N <= 0) return;
    if (fast) hipLaunchKernelGGL(softmax_normalise_wide_kernel<true>, dim3(N), dim3(256), 0, s, y, pat, N, L, Lp, (const float2 *)smstat);
    else      hipLaunchKernelGGL(softmax_normalise_wide_kernel<false>, dim3(N), dim3(256), 0, s, y, pat, N, L, Lp, (const float2 *)smstat);
}
void launch_softmax_fwd(hipStream_t s, float *y, const char *pat, int N, int L, int Lp, const int *tcls, float *rowstat, bool fast, float *smstat)
{
    if (N <= 0) return;
    if (L > 256 && L <= 256 * SMW_VPT) {
        float2 *rs = (float2 *)rowstat, *sm = (float2 *)smstat;
        if (fast) { if (sm) hipLaunchKernelGGL((softmax_fwd_wide_kernel<false, true>), dim3(N), dim3(256), 0, s, y, pat, N, L, Lp, tcls, rs, sm);
                    else    hipLaunchKernelGGL((softmax_fwd_wide_kernel<true, true>), dim3(N), dim3(256), 0, s, y, pat, N, L, Lp, tcls, rs, sm); }
        else      { if (sm) hipLaunchKernelGGL((softmax_fwd_wide_kernel<false, false>), dim3(N), dim3(256), 0, s, y, pat, N, L, Lp, tcls, rs, sm);
                    else    hipLaunchKernelGGL((softmax_fwd_wide_kernel<true, false>), dim3(N), dim3(256), 0, s, y, pat, N, L, Lp, tcls, rs, sm); }
    } else if (Lp <= 256) {
        if (fast) hipLaunchKernelGGL(softmax_fwd_rows16_kernel<true>, dim3((N + 15) / 16), dim3(256), 0, s, y, pat, N, L, Lp, tcls, (float2 *)rowstat);
        else      hipLaunchKernelGGL(softmax_fwd_rows16_kernel<false>, dim3((N + 15) / 16), dim3(256), 0, s, y, pat, N, L, Lp, tcls, (float2 *)rowstat);
    }
    else
        hipLaunchKernelGGL(softmax_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, s, y, pat, N, L, Lp, tcls, (float2 *)rowstat);
}

// fixed-order reduction of the row statistics: loss2[0] += -sum(log p), loss2[1] (int) += #correct
// One workgroup, fixed summation order (reproducible): 1024 "virtual threads" v take rows v, v + 1024, ... (four loads in flight
// at a time: a plain strided loop is a chain of N/1024 memory round trips, 10 us at N = 17 500), 16 virtual waves are folded with
// xor shuffles, their sums added in wave order.  A workgroup of fewer threads walks the virtual threads in passes and forms the
// SAME sums in the same order (softmax_mcc_bwd_kernel's extra workgroup: the reduction rides beside the backward rows).
__device__ __forceinline__ void rowstat_reduce_body(const float2 *rowstat, int N, float *loss2, float scale)
{
    __shared__ float sl[16]; __shared__ int sc[16];
    for (int vt = threadIdx.x; vt < 1024; vt += blockDim.x) {
        float l4[4] = {0.f, 0.f, 0.f, 0.f}; int c4[4] = {0, 0, 0, 0};
        for (int i0 = vt; i0 < N; i0 += 4 * 1024) {
            float2 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int i = i0 + k * 1024; v[k] = i < N ? rowstat[i] : make_float2(0.f, 0.f); }
#pragma unroll
            for (int k = 0; k < 4; ++k) { l4[k] += v[k].x; c4[k] += (int)v[k].y; }
        }
        float l = (l4[0] + l4[1]) + (l4[2] + l4[3]);
        int c = c4[0] + c4[1] + c4[2] + c4[3];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { l += __shfl_xor(l, o); c += __shfl_xor(c, o); }
        if ((vt & 63) == 0) { sl[vt >> 6] = l; sc[vt >> 6] = c; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float lt = 0.f; int ct = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { lt += sl[w]; ct += sc[w]; }
        loss2[0] += scale * lt; ((int *)loss2)[1] += ct;
    }
}
__global__ void rowstat_reduce_kernel(const float2 *rowstat, int N, float *loss2, float scale)
{
    rowstat_reduce_body(rowstat, N, loss2, scale);
}
void launch_rowstat_reduce(hipStream_t s, const float *rowstat, int N, float *loss2, bool reset, float scale)
{
    if (reset) (void)hipMemsetAsync(loss2, 0, 2 * sizeof(float), s);
    if (N <= 0) return;
    hipLaunchKernelGGL(rowstat_reduce_kernel, dim3(1), dim3(1024), 0, s, (const float2 *)rowstat, N, loss2, scale);
}

__global__ void softmax_bwd_kernel(const float *y, float *err, const char *pat, int N, int L, int Lp)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= N) return;
    if (pat[row] == 0) return;                           // SoftmaxLayer.cu:175-176
    const float *yr = y + row * Lp; float *er = err + row * Lp;
    float off = 0.f;
    for (int j = lane; j < L; j += 64) off += yr[j] * er[j];   // :183-185
    off = wave_sum(off);
    for (int j = lane; j < L; j += 64) er[j] = yr[j] * (er[j] - off);  // :214
}
void launch_softmax_bwd(hipStream_t s, const float *y, float *err, const char *pat, int N, int L, int Lp)
{
    if (N <= 0) return;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((N + 3) / 4), dim3(256), 0, s, y, err, pat, N, L, Lp);
}

// multiclass_classification::computeBackwardPass + SoftmaxLayer::computeBackwardPass + the Identity delta +
// ComputeBiasWeightUpdateFn in one pass over the posteriors (MulticlassClassificationLayer.cu:220-240,
// SoftmaxLayer.cu:317-349, FeedForwardLayer.cu:69-102).  The injected error is -1/max(FLT_MIN, p_t) at the
// target and 0 elsewhere, so sum_j y_j e_j has exactly one non-zero term: bit-identical to the three passes.
// Requires Lp <= 256 (column sums live in 4 registers per lane).
// `err` may be null in bf16 mode: the fp32 outputErrors of the layer are then not materialised (nothing on the training path reads
// them: the products take the bf16 operand copy), which is 4 of the 10 bytes per element this HBM-bound pass moves.
// `loss2` != null: SIXTEEN extra workgroups (the last ones) sum the row statistics of the forward pass into loss2
// (cn_loss_accumulate deferred into this launch: 9 us of a one-workgroup kernel off the critical path).  Workgroup w forms the sum
// of "virtual wave" w of rowstat_reduce_body -- the same loads, the same adds, the same shuffles --, the last one to arrive adds
// the sixteen sums in wave order: bit-identical to the one-workgroup kernel, a sixteenth of its time.
constexpr int MCC_LOSS_WGS = 16;
constexpr int MCC_COL_REPL = 8;           // replicas of the column sums (softmax_mcc_bwd_colpart_floats)
__device__ __forceinline__ void rowstat_reduce_wave(const float2 *rowstat, int N, float *loss2, float scale, int w, float *part)
{
    if (threadIdx.x >= 64) return;
    const int vt = 64 * w + threadIdx.x;
    float l4[4] = {0.f, 0.f, 0.f, 0.f}; int c4[4] = {0, 0, 0, 0};
    for (int i0 = vt; i0 < N; i0 += 4 * 1024) {
        float2 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = i0 + k * 1024; v[k] = i < N ? rowstat[i] : make_float2(0.f, 0.f); }
#pragma unroll
        for (int k = 0; k < 4; ++k) { l4[k] += v[k].x; c4[k] += (int)v[k].y; }
    }
    float l = (l4[0] + l4[1]) + (l4[2] + l4[3]);
    int c = c4[0] + c4[1] + c4[2] + c4[3];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { l += __shfl_xor(l, o); c += __shfl_xor(c, o); }
    // the last workgroup to arrive adds the sixteen sums in wave order; its lanes fetch them side by side (a chain of dependent
    // L2 round trips in one thread cost 6 us here)
    unsigned *cnt = (unsigned *)(part + 2 * MCC_LOSS_WGS);
    int last = 0;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&part[2 * w], l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((int *)&part[2 * w + 1], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The partials must have ARRIVED before this workgroup is counted.  A workgroup-scope release fence does not do that
        // (it compiles to no wait at all on gfx950: the stores and the counter add left unordered, other L2 channels): the one
        // counting thread drains its stores explicitly.  They are agent-scope (sc1, write-through) stores -- acknowledged is
        // visible to every CU --, so the wait is all a release has to add here: an agent-scope release RMW would also write the
        // XCD's whole dirty L2 back (buffer_wbl2: the rows this launch has stored; +1.6 us on the launch).  The last arriver
        // acquires before it reads (tests/test_abi_and_host.py checks the wait in the ISA).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == MCC_LOSS_WGS - 1;
    }
    last = __shfl(last, 0);
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int k = threadIdx.x < MCC_LOSS_WGS ? threadIdx.x : 0;
    const float pl = __hip_atomic_load(&part[2 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int pc = __hip_atomic_load((int *)&part[2 * k + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float lt = 0.f; int ct = 0;
#pragma unroll
    for (int j = 0; j < MCC_LOSS_WGS; ++j) { lt += __shfl(pl, j); ct += __shfl(pc, j); }
    if (threadIdx.x == 0) {
        loss2[0] += scale * lt; ((int *)loss2)[1] += ct;
        __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (zero again for the next launch)
    }
}
template <bool F32>
__global__ void softmax_mcc_bwd_kernel(const float *y, const int *tcls, const char *pat, int N, int L, int Lp,
                                       float *err, void *delta_op, float *colsum, const float2 *rowstat, float *loss2, float *loss_part, float *colpart,
                                       float *det_part)
{
    const unsigned nwg = gridDim.x - (loss2 ? (unsigned)MCC_LOSS_WGS : 0u);
    if (loss2 && blockIdx.x >= nwg) { rowstat_reduce_wave(rowstat, N, loss2, -1.0f, (int)(blockIdx.x - nwg), loss_part); return; }
    // SIXTEEN lanes per pattern, four patterns per wave, sixteen per workgroup and pass (round 4; before: a wave per pattern,
    // dword accesses, four patterns in flight per wave).  Lane c of a pattern's group owns the 16-byte column groups c, c + 16,
    // c + 32, c + 48.  A row used to be a chain of dependent loads (target class -> its posterior -> the row): the row is
    // loaded beside its class, the target's posterior is picked out of the registers (one select per element + one shuffle
    // from its owner lane), and the next sixteen patterns are fetched before the current ones are worked on.  The column
    // sums end in one atomic per column and WORKGROUP: they are same-address atomics, so the grid stays at one workgroup per
    // CU (2048 workgroups: 50 us).
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    __shared__ float part[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15;
    f32x4 cs[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) cs[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    struct Rows { f32x4 v[4]; int tc; bool in, real; };
    auto fetch = [&](long batch, Rows &r) {
        const long row = batch * 16 + (threadIdx.x >> 4);
        r.in = row < N;
        const long rr = r.in ? row : N - 1;
        r.tc = tcls[rr]; r.real = r.in && pat[rr] != 0;
        const float *yr = y + rr * Lp;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int j0 = 4 * (c + 16 * g);
            r.v[g] = j0 < Lp ? *(const f32x4 *)(yr + j0) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    const long nb = ((long)N + 15) / 16;
    Rows cur, nxt, nx2;                                  // two passes ahead: a workgroup makes four to five passes, each a memory round trip
    if ((long)blockIdx.x < nb) fetch(blockIdx.x, cur);
    if ((long)blockIdx.x + nwg < nb) fetch(blockIdx.x + nwg, nxt);
    for (long bt = blockIdx.x; bt < nb; bt += nwg) {
        const bool more = bt + nwg < nb;
        if (bt + 2 * (long)nwg < nb) fetch(bt + 2 * (long)nwg, nx2);
        const long row = bt * 16 + (threadIdx.x >> 4);
        // the target's posterior: v[og][oe] of lane ol of this pattern's group
        const int idx = cur.tc >= 0 ? cur.tc : 0, og = idx >> 6, oe = idx & 3, ol = (idx >> 2) & 15;
        float cand = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) if (g == og && e == oe) cand = cur.v[g][e];
        const float pt_ = __shfl(cand, (lane & 48) | ol);
        float et = 0.f, off = 0.f;
        if (cur.real && cur.tc >= 0) { et = -(1.0f / fmaxf(NL_MIN, pt_)); off = pt_ * et; }
        const float m0 = 0.f - off, mt = et - off;        // factor of every column but the target's / of the target's
        const int tgroup = cur.tc & ~3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int j0 = 4 * (c + 16 * g);
            if (j0 < Lp) {
                f32x4 dl = {0.f, 0.f, 0.f, 0.f};
                if (cur.real) {
                    dl = cur.v[g] * m0;
                    if (j0 == tgroup) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (j0 + e == cur.tc) dl[e] = cur.v[g][e] * mt;
                    }
                    if (j0 + 3 >= L) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (j0 + e >= L) dl[e] = 0.f;
                    }
                }
                if (cur.in) {
                    if (err) *(f32x4 *)(err + row * Lp + j0) = dl;    // (bf16 mode: only the operand copy is written; cn_layer_read converts it)
                    if constexpr (!F32) *(bf16x4 *)((__bf16 *)delta_op + row * Lp + j0) = bf16x4{(__bf16)dl[0], (__bf16)dl[1], (__bf16)dl[2], (__bf16)dl[3]};
                }
                cs[g] += dl;
            }
        }
        if (more) { cur = nxt; nxt = nx2; }
    }
    // the four patterns of a wave -> lanes 0..15, the four waves through LDS, one atomic per column
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = cs[g][e];
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            if (lane < 16) part[wv][4 * (c + 16 * g) + e] = v;
        }
    __syncthreads();
    if (det_part) {          // deterministic mode: this workgroup's row of partials, folded in workgroup order by the launcher
        for (int j = threadIdx.x; j < Lp; j += 256) det_part[(long)blockIdx.x * Lp + j] = (part[0][j] + part[1][j]) + (part[2][j] + part[3][j]);
        return;
    }
    if (!colpart) {
        for (int j = threadIdx.x; j < Lp; j += 256) atomicAdd(&colsum[j], (part[0][j] + part[1][j]) + (part[2][j] + part[3][j]));
        return;
    }
    // 256 workgroups adding to the same 192 words at the same moment are a queue at six L2 lines (4.5 of the launch's 17 us):
    // the adds go to MCC_COL_REPL replicas instead, and the last workgroup to arrive folds the replicas into colsum
    float *rep = colpart + (blockIdx.x % MCC_COL_REPL) * 256;
    for (int j = threadIdx.x; j < Lp; j += 256) atomicAdd(&rep[j], (part[0][j] + part[1][j]) + (part[2][j] + part[3][j]));
    // The adds are device-scope atomics, performed where all CUs meet: they must have been ACKNOWLEDGED before this workgroup
    // is counted.  Every wave drains its own (`s_waitcnt vmcnt(0)`; a workgroup-scope release fence compiles to no wait on
    // gfx950 and left a late add free to land behind the counter, or in a replica already folded and zeroed), the barrier
    // collects the waves, and thread 0 counts.  The wait is the whole release: what is handed over are atomics, performed where
    // the CUs meet, not plain stores in a write-back L2 -- an agent-scope release RMW by thread 0 measured +1.6 us per launch
    // (its buffer_wbl2 writes back the rows this launch stored; the same fence in every thread doubled the launch in round 4).
    // The last arriver acquires before it reads.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int last;
    unsigned *cnt = (unsigned *)(colpart + MCC_COL_REPL * 256);
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1;
    __syncthreads();
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int j = threadIdx.x; j < Lp; j += 256) {
        float v[MCC_COL_REPL];
#pragma unroll
        for (int r = 0; r < MCC_COL_REPL; ++r) v[r] = __hip_atomic_load(&colpart[r * 256 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < MCC_COL_REPL; ++r) { t += v[r]; __hip_atomic_store(&colpart[r * 256 + j], 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        colsum[j] += t;
    }
    if (threadIdx.x == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (zero again for the next launch)
}
// The same fusion for wide rows (256 < Lp <= 8192): one workgroup walks rows blockIdx.x, blockIdx.x + grid, ...; a thread
// owns the four columns 4 (tid + 256 k) .. + 3 of every k (16-byte loads and stores, 8-byte bf16 stores) and keeps their
// sums in registers; one atomic per column and workgroup at the end.  The loads of a row are issued together, and those of
// the next row before the current one is worked on (the row and its per-row scalars are a chain of three dependent round
// trips otherwise).  LAZY (1 exact, 2 FAST): y holds the logits of softmax_fwd_wide_kernel<false, ..>, the posteriors are
// recomputed from smstat.
template <bool F32, int LAZY>
__global__ __launch_bounds__(256) void softmax_mcc_bwd_wide_kernel(const float *__restrict__ y, const int *__restrict__ tcls, const char *__restrict__ pat,
                                                                   int N, int L, int Lp, float *__restrict__ err, void *__restrict__ delta_op,
                                                                   float *colsum, const float2 *__restrict__ smstat, float *det_part)
{
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    constexpr int NV = SMW_VPT / 4;                       // 4-column groups per thread
    const int tid = threadIdx.x;
    f32x4 cs[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) cs[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    struct Row { f32x4 v[NV]; float pt; int tc; bool real; float2 st; };
    auto fetch = [&](long row, Row &r) {
        r.tc = -1; r.real = false; r.pt = 0.f; r.st = make_float2(0.f, 1.f);
        if (row >= N) return;
        r.tc = tcls[row]; r.real = pat_at(pat, row) != 0;
        if (!r.real) return;
        const float *yr = y + row * Lp;
        if constexpr (LAZY != 0) r.st = smstat[row];
        if (r.tc >= 0) r.pt = yr[r.tc];
        // columns >= L hold what makes their posterior 0: -inf as a logit (LAZY), 0 as a posterior
        const float none = LAZY != 0 ? -__builtin_inff() : 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = 4 * (tid + 256 * k);
            r.v[k] = j < Lp ? *(const f32x4 *)(yr + j) : f32x4{none, none, none, none};
            if (j + 3 >= L) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (j + e >= L) r.v[k][e] = none;
            }
        }
    };
    Row cur, nxt;
    fetch(blockIdx.x, cur);
    for (long row = blockIdx.x; row < N; row += gridDim.x) {
        fetch(row + gridDim.x, nxt);
        float et = 0.f, off = 0.f;
        if (cur.real && cur.tc >= 0) {
            float pt_ = cur.pt;
            if constexpr (LAZY != 0) pt_ = softmax_scale<LAZY == 2>(softmax_exp<LAZY == 2>(pt_ - cur.st.x), cur.st.y);
            et = -(1.0f / fmaxf(NL_MIN, pt_)); off = pt_ * et;
        }
        const float m0 = 0.f - off, mt = et - off;        // factor of every column but the target's / of the target's
        const int tgroup = cur.tc & ~3;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = 4 * (tid + 256 * k);
            if (j >= Lp) break;
            f32x4 dl = {0.f, 0.f, 0.f, 0.f};
            if (cur.real) {
                f32x4 p = cur.v[k];
                if constexpr (LAZY != 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) p[e] = softmax_scale<LAZY == 2>(softmax_exp<LAZY == 2>(p[e] - cur.st.x), cur.st.y);
                }
                dl = p * m0;
                if (j == tgroup) {                       // (one lane of the workgroup per row)
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (j + e == cur.tc) dl[e] = p[e] * mt;
                }
            }
            if (err) *(f32x4 *)(err + row * Lp + j) = dl;
            if constexpr (!F32) *(bf16x4 *)((__bf16 *)delta_op + row * Lp + j) = bf16x4{(__bf16)dl[0], (__bf16)dl[1], (__bf16)dl[2], (__bf16)dl[3]};
            cs[k] += dl;
        }
        cur = nxt;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = 4 * (tid + 256 * k);
        if (j < Lp) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (det_part) det_part[(long)blockIdx.x * Lp + j + e] = cs[k][e];
                else atomicAdd(&colsum[j + e], cs[k][e]);
            }
        }
    }
}

bool softmax_mcc_bwd_takes_loss(int Lp) { return Lp <= 256; }
size_t softmax_mcc_bwd_colpart_floats() { return MCC_COL_REPL * 256 + 1; }
void launch_softmax_mcc_bwd(hipStream_t s, bool f32, const float *y, const int *tcls, const char *pat, int N, int L, int Lp,
                            float *err, void *delta_op, float *colsum, const float *rowstat, float *loss2, float *loss_part, const float *smstat, bool fast,
                            float *colpart, float *det_part, FoldItem *fold_out)
{
    if (fold_out) fold_out->nparts = 0;
    if (N <= 0) return;
    if (Lp > 256) {
        int blocks = N < 1536 ? N : 1536;      // (two to three workgroups per CU are resident, 170 VGPRs: 512 ... 2048 measured)
        const float2 *sm = (const float2 *)smstat;
#define CN_BWD_WIDE(F, Z) hipLaunchKernelGGL((softmax_mcc_bwd_wide_kernel<F, Z>), dim3(blocks), dim3(256), 0, s, y, tcls, pat, N, L, Lp, err, delta_op, colsum, sm, det_part)
        if (f32) { if (!sm) CN_BWD_WIDE(true, 0); else if (fast) CN_BWD_WIDE(true, 2); else CN_BWD_WIDE(true, 1); }
        else     { if (!sm) CN_BWD_WIDE(false, 0); else if (fast) CN_BWD_WIDE(false, 2); else CN_BWD_WIDE(false, 1); }
#undef CN_BWD_WIDE
        if (det_part) fold_now_or_later(s, FoldItem{colsum, det_part, (long)Lp, blocks, 1, Lp, Lp, 1, 0}, fold_out);
        return;
    }
    int blocks = (N + 15) / 16; if (blocks > 256) blocks = 256;
    const int nwg = blocks;
    if (loss2 && !loss_part) loss2 = nullptr;
    if (loss2) blocks += MCC_LOSS_WGS;
    if (f32) hipLaunchKernelGGL(softmax_mcc_bwd_kernel<true>, dim3(blocks), dim3(256), 0, s, y, tcls, pat, N, L, Lp, err, delta_op, colsum, (const float2 *)rowstat, loss2, loss_part, colpart, det_part);
    else     hipLaunchKernelGGL(softmax_mcc_bwd_kernel<false>, dim3(blocks), dim3(256), 0, s, y, tcls, pat, N, L, Lp, err, delta_op, colsum, (const float2 *)rowstat, loss2, loss_part, colpart, det_part);
    if (det_part) fold_now_or_later(s, FoldItem{colsum, det_part, (long)Lp, nwg, 1, Lp, Lp, 1, 0}, fold_out);
}
size_t det_colsum_part_floats(int Lp) { return (size_t)1536 * Lp; }      // the most workgroups any of the column-sum producers launches

// ---------------------------------------------------------------------------------------------
// post output layers.  loss2[0] = error (float), loss2[1] = #correct (int bits).  One wave per pattern; the per-pattern
// term lands in rowstat[N] = {log p_target, correct} -- what the softmax forward pass leaves there for its own rows -- and
// is summed in a fixed order by rowstat_reduce_kernel: reproducible, no float atomics (round 6; before: one atomic per
// workgroup, in arrival order).
// ---------------------------------------------------------------------------------------------
__global__ void mcc_rows_kernel(const float *y, const int *tcls, int N, int L, int Lp, float2 *rowstat)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (long row = (long)blockIdx.x * 4 + wv; row < N; row += (long)gridDim.x * 4) {
        const int tc = tcls[row];
        if (tc < 0) { if (lane == 0) rowstat[row] = make_float2(0.f, 0.f); continue; }   // MulticlassClassificationLayer.cu:61-62
        const float *r = y + row * Lp;
        // CountCorrectClassificationsFn :89-99: first strictly greater value wins, start (0, class 0)
        float best = 0.f; int bi = 0;
        for (int j = lane; j < L; j += 64) { float v = r[j]; if (v > best) { best = v; bi = j; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o); int oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (best <= 0.f) bi = 0;
        if (lane == 0) rowstat[row] = make_float2(logf(fmaxf(NL_MIN, r[tc])), (bi == tc) ? 1.f : 0.f);   // :65-66
    }
}
void launch_mcc_eval(hipStream_t s, const float *y, const int *tcls, int N, int L, int Lp, float *loss2, bool reset, float *rowstat)
{
    if (N <= 0) { if (reset) (void)hipMemsetAsync(loss2, 0, 2 * sizeof(float), s); return; }
    int blocks = (N + 3) / 4; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(mcc_rows_kernel, dim3(blocks), dim3(256), 0, s, y, tcls, N, L, Lp, (float2 *)rowstat);
    launch_rowstat_reduce(s, rowstat, N, loss2, reset);            // calculateError returns -sum, :212
}

__global__ void mcc_backward_kernel(const float *y, const int *tcls, int N, int L, int Lp, float *err)
{
    const long total = (long)N * Lp;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long n = idx / Lp; const int j = idx % Lp;
        float v = 0.f;                                            // fill_n 0, :227
        if (j < L && tcls[n] == j) v = -(1.0f / fmaxf(NL_MIN, y[idx]));   // :128-130
        err[idx] = v;
    }
}
void launch_mcc_backward(hipStream_t s, const float *y, const int *tcls, int N, int L, int Lp, float *err)
{
    long total = (long)N * Lp; if (total <= 0) return;
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mcc_backward_kernel, dim3(blocks), dim3(256), 0, s, y, tcls, N, L, Lp, err);
}

// ---------------------------------------------------------------------------------------------
// post output layers other than multiclass_classification (sse: SsePostOutputLayer.cu:39-88; the rest is
// SURVEY section 8 row f4).  L = size of the output layer, y pitch Lp;
// targets hold L values per pattern, or 2L interleaved (target, weight | filter input) pairs.
//   weightedsse  WeightedSsePostOutputLayer.cu:40-93, :119-167    wf   SseMaskPostOutputLayer.cu:40-93, :119-167
//   ce           CePostOutputLayer.cu:43-99, :125-168    rmse RmsePostOutputLayer.cu:40-97, :125-172
//   binary_classification  BinaryClassificationLayer.cu:44-111, :132-207
// One wave per pattern; the per-pattern term lands in rowstat[N] = {term, correct} and is summed
// in a fixed order by rowstat_reduce_kernel (reproducible, no float atomics).
// ---------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ float post_term(float y, const float *tg, long i)
{
    if (KIND == POST_SSE)         { float df = tg[i] - y; return df * df; }
    if (KIND == POST_WEIGHTEDSSE) { float df = (y - tg[2 * i]) * tg[2 * i + 1]; return df * df; }
    if (KIND == POST_SSE_MASK)    { float df = y * tg[2 * i + 1] - tg[2 * i]; return df * df; }
    if (KIND == POST_CE)          { float t = tg[i]; return t * logf(fmaxf(NL_MIN, t) / fmaxf(NL_MIN, y)); }
    if (KIND == POST_RMSE)        { float df = y - tg[i]; return df * df; }
    /* POST_BINARY */               float act = fmaxf(y, NL_MIN); return -logf(tg[i] > 0.f ? act : 1.f - act);
}
template <int KIND>
__device__ __forceinline__ float post_err(float y, const float *tg, long i, float rmse)
{
    if (KIND == POST_SSE)         return y - tg[i];
    if (KIND == POST_WEIGHTEDSSE) return (y - tg[2 * i]) * tg[2 * i + 1];
    if (KIND == POST_SSE_MASK)    return (y * tg[2 * i + 1] - tg[2 * i]) * tg[2 * i + 1];
    if (KIND == POST_CE)          return fminf(100.f, fmaxf(-100.f, -tg[i] / fmaxf(NL_MIN, y)));
    if (KIND == POST_RMSE)        return rmse * (y - tg[i]);
    /* POST_BINARY */               float act = fmaxf(y, NL_MIN); float tp = tg[i] > 0.f ? act : 1.f - act;
                                    return tg[i] > 0.f ? -(1.f / tp) : (1.f / tp);
}

template <int KIND>
__global__ void post_rows_kernel(const float *y, const float *tgt, const char *pat, int N, int L, int Lp, float2 *rowstat)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    float a = 0.f, corr = 0.f;
    if (pat[row] != 0) {
        for (int j = lane; j < L; j += 64) a += post_term<KIND>(y[row * Lp + j], tgt, row * L + j);
        a = wave_sum(a);
        if (KIND == POST_RMSE) a = sqrtf(a / (float)L);
        if (KIND == POST_BINARY) corr = ((tgt[row] > 0.5f) == (y[row * Lp] > 0.5f)) ? 1.f : 0.f;
    }
    if (lane == 0) rowstat[row] = make_float2(a, corr);
}

template <int KIND>
__global__ void post_backward_kernel(const float *y, const float *tgt, const char *pat, int N, int L, int Lp, float *err)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const bool real = pat[row] != 0;
    float rmse = 0.f;
    if (KIND == POST_RMSE && real) {
        float a = 0.f;
        for (int j = lane; j < L; j += 64) a += post_term<POST_RMSE>(y[row * Lp + j], tgt, row * L + j);
        rmse = sqrtf(wave_sum(a) / (float)L);
    }
    for (int j = lane; j < Lp; j += 64)
        err[row * Lp + j] = (real && j < L) ? post_err<KIND>(y[row * Lp + j], tgt, row * L + j, rmse) : 0.f;
}

#define POST_DISPATCH(KERN, ...)                                                                              \
    switch (kind) {                                                                                           \
    case POST_SSE:         hipLaunchKernelGGL(KERN<POST_SSE>, dim3((N + 3) / 4), dim3(256), 0, s, __VA_ARGS__); break;         \
    case POST_WEIGHTEDSSE: hipLaunchKernelGGL(KERN<POST_WEIGHTEDSSE>, dim3((N + 3) / 4), dim3(256), 0, s, __VA_ARGS__); break; \
    case POST_SSE_MASK:    hipLaunchKernelGGL(KERN<POST_SSE_MASK>, dim3((N + 3) / 4), dim3(256), 0, s, __VA_ARGS__); break;    \
    case POST_CE:          hipLaunchKernelGGL(KERN<POST_CE>, dim3((N + 3) / 4), dim3(256), 0, s, __VA_ARGS__); break;          \
    case POST_RMSE:        hipLaunchKernelGGL(KERN<POST_RMSE>, dim3((N + 3) / 4), dim3(256), 0, s, __VA_ARGS__); break;        \
    default:               hipLaunchKernelGGL(KERN<POST_BINARY>, dim3((N + 3) / 4), dim3(256), 0, s, __VA_ARGS__); break;      \
    }

void launch_post_eval(hipStream_t s, int kind, const float *y, const float *tgt, const char *pat, int N, int L, int Lp,
                      float *rowstat, float *loss2, bool reset)
{
    if (N > 0) { POST_DISPATCH(post_rows_kernel, y, tgt, pat, N, L, Lp, (float2 *)rowstat) }
    const bool half = kind == POST_SSE || kind == POST_WEIGHTEDSSE || kind == POST_SSE_MASK;
    launch_rowstat_reduce(s, rowstat, N, loss2, reset, half ? 0.5f : 1.0f);
}
void launch_post_backward(hipStream_t s, int kind, const float *y, const float *tgt, const char *pat, int N, int L, int Lp, float *err)
{
    if (N <= 0) return;
    POST_DISPATCH(post_backward_kernel, y, tgt, pat, N, L, Lp, err)
}

// BinaryClassificationLayer.cu:157-164: the targets are the target classes copied as floats
__global__ void classes_to_targets_kernel(const int *tcls, float *tgt, int N)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) tgt[i] = (float)tcls[i];
}
void launch_classes_to_targets(hipStream_t s, const int *tcls, float *tgt, int N)
{
    if (N <= 0) return;
    hipLaunchKernelGGL(classes_to_targets_kernel, dim3((N + 255) / 256), dim3(256), 0, s, tcls, tgt, N);
}

// ---------------------------------------------------------------------------------------------
// optimizer step
// ---------------------------------------------------------------------------------------------
// x *= a (test double of the gradient exchange, cn_allreduce_grads with CN_COMM_TEST_DOUBLE: what a two-rank all-reduce of
// equal shards does to a gradient)
__global__ void scale_kernel(float *x, size_t n, float a)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= a;
}
void launch_scale(hipStream_t s, float *x, size_t n, float a)
{
    if (n) hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 1023) / 1024 > 512 ? 512 : (n + 1023) / 1024)), dim3(256), 0, s, x, n, a);
}
// dst = src[0] + src[1] + ... (rank order: every rank forms the same sum) -- the test backend of the gradient exchange
__global__ void sum_ranks_kernel(float *dst, SumRanks sr, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = sr.src[0][i];
        for (int r = 1; r < sr.n; ++r) v += sr.src[r][i];
        dst[i] = v;
    }
}
void launch_sum_ranks(hipStream_t s, float *dst, const SumRanks &sr, size_t n)
{
    if (n) hipLaunchKernelGGL(sum_ranks_kernel, dim3((unsigned)((n + 1023) / 1024 > 512 ? 512 : (n + 1023) / 1024)), dim3(256), 0, s, dst, sr, n);
}

// Batch learning: the epoch sum of the fractions' weightUpdates (Optimizer.cu:72-85: thrust::copy for the first fraction,
// thrust::transform(plus) for the others -- one launch over the whole arena instead of one per layer).
__global__ void accumulate_kernel(float *acc, const float *wu, size_t n, int first)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc[i] = first ? wu[i] : __fadd_rn(acc[i], wu[i]);
}
void launch_accumulate(hipStream_t s, float *acc, const float *wu, size_t n, bool first)
{
    if (n == 0) return;
    int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(accumulate_kernel, dim3(blocks), dim3(256), 0, s, acc, wu, n, first ? 1 : 0);
}

__global__ void sgd_kernel(float *w, const float *wu, float *wd, size_t n, float lr, float mom)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        // separate multiplies and adds (no FMA contraction) so that equal gradients give bit-equal weights
        const float dl = __fsub_rn(__fmul_rn(mom, wd[i]), __fmul_rn(lr, wu[i]));   // SteepestDescentOptimizer.cu:51
        wd[i] = dl;
        w[i] = __fadd_rn(w[i], dl);                               // :55
    }
}
void launch_sgd(hipStream_t s, float *w, const float *wu, float *wd, size_t n, float lr, float mom, hipEvent_t done)
{
    if (n == 0) return;
    int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipExtLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, s, nullptr, done, 0, w, wu, wd, n, lr, mom);
}

}  // namespace cn

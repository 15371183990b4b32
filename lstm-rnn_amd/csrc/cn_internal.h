// Internal declarations shared by the translation units of libcurrennt_hip.so.
// Nothing in here is part of the ABI (include/currennt_hip.h is).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>
#include <stddef.h>
#include <stdint.h>

#include "../../include/currennt_hip.h"

namespace cn {

// ---- activation ids used by kernels --------------------------------------------------------
enum Act { ACT_TANH = 0, ACT_LOGISTIC = 1, ACT_IDENTITY = 2 };

// kernel classes for cn_ctx_timing_*
enum KClass { KC_REC_FWD = 0, KC_REC_BWD = 1, KC_GEMM_WIDE = 2, KC_GEMM_GRAD = 3, KC_OTHER = 4, KC_COMM = 5, KC_COUNT = 6 };

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Arithmetic of the MFMA products (`prec` arguments of the launchers below).  The values of P_BF16 / P_F32 equal
// false / true, so a caller that only knows "fp32 operands or not" may pass a bool.
//   P_BF16  bf16 operands in memory and in the MFMAs (throughput mode)
//   P_F32   fp32 operands in memory, exact-fp32 MFMAs (v_mfma_f32_*_f32, 1/16 of the bf16 rate)
//   P_X3    fp32 operands in memory; every operand is split in the kernel into bf16 hi + bf16 lo (x = hi + lo + O(2^-17 x))
//           and a product is three bf16 MFMAs (hi*hi + lo*hi + hi*lo, fp32 accumulation): ~2^-16 relative per term, a third
//           of the bf16 rate -- the parity mode that is fast (CN_PREC_BF16X3)
enum Prec { P_BF16 = 0, P_F32 = 1, P_X3 = 2 };

// ---- options -----------------------------------------------------------------------------------
// ONE block of A/B and test switches per context instead of environment look-ups scattered over the launch paths (round 6).  A context
// copies the process defaults -- read ONCE from the environment, variable CN_<NAME IN CAPITALS> -- at cn_ctx_create; after that
// only cn_ctx_set_option(ctx, "<name>", value) changes them, and no launch path reads the environment.  FLAG: set by the mere
// presence of the variable (as before); NUM: its integer value.  Options that size allocations (cluster4, no_cluster, rpl) must be
// set before the layers are created.  cn_layer_recurrent_kernel keeps reporting which kernel actually ran.
#define CN_OPTION_LIST(FLAG, NUM)                                                                                                  \
    /* GEMM selection */                                                                                                          \
    FLAG(no_big_gemm) FLAG(no_big8) NUM(big8_min_k, 0) FLAG(no_nt_mid) FLAG(no_big_tn) NUM(nt_bm64_below, 400) NUM(tn_blocks, 0) \
    NUM(tnbig_blocks, 0) NUM(tnbig_group_mink, 28000) NUM(nt_mid_min_tiles, 384) NUM(nt_bm64_shortk_tiles, 1100) FLAG(no_nt_panel) NUM(nt_panel_max_panels, 0) FLAG(nt_panel_no_touch) NUM(nt_panel_min_ktiles, 8) NUM(nt_panel_max_ntiles, 2) FLAG(no_nt_rowmap) FLAG(nt_rowmap_tiled)                                                                           \
    /* recurrent kernel selection */                                                                                              \
    FLAG(no_lds_claim) FLAG(bwd_ug2) FLAG(fwd_ug2) FLAG(bwd_psum) FLAG(no_bwd_psum) FLAG(cluster_helpers) FLAG(no_cluster_helpers) \
    FLAG(cluster4) FLAG(no_cluster) FLAG(cluster_gate_off) FLAG(no_s2c) FLAG(s2c) FLAG(no_s2_asm) FLAG(no_s2_asm_bwd) FLAG(s2_x3) \
    FLAG(no_s2) FLAG(no_s2w) FLAG(no_s2w_asm) FLAG(pre16)                                                                           \
    /* step structure */                                                                                                          \
    FLAG(softmax_exact) FLAG(tail_on_side) FLAG(no_side_rule) FLAG(no_lazy_softmax) FLAG(lazy_softmax) FLAG(comm_test_double)     \
    FLAG(no_loss_defer) FLAG(no_pack_group) FLAG(no_sgd_fuse) NUM(comm_cu_margin, 32)
struct Options {
#define CN_OPT_FIELD(name) int name = 0;
#define CN_OPT_FIELD_NUM(name, dflt) long name = dflt;
    CN_OPTION_LIST(CN_OPT_FIELD, CN_OPT_FIELD_NUM)
#undef CN_OPT_FIELD
#undef CN_OPT_FIELD_NUM
};
Options options_from_env();                              // (the environment is read here and nowhere else on the compute path)
bool option_set(Options &o, const char *name, long value);  // false: no such option
bool option_get(const Options &o, const char *name, long *value);
// the options of the context whose call is running on this thread (set by every entry point of the ABI that touches the
// device); the process defaults outside of one
const Options &opt();

// hipFuncSetAttribute (the > 64 KB dynamic LDS opt-in) is per device: a process may drive several GPUs
struct DeviceOnce {
    std::atomic<unsigned long long> seen{0};
    bool first()
    {
        int d = 0;
        (void)hipGetDevice(&d);
        const unsigned long long bit = 1ull << (d & 63);
        return (seen.fetch_or(bit) & bit) == 0;        // (contexts of several host threads may launch concurrently)
    }
};

// ---- GEMM ------------------------------------------------------------------------------------
// All matrices are row-major; "op" element type is float (CN_PREC_F32) or bf16 (CN_PREC_BF16).
struct GemmNT {            // C[m][n] = sum_k A[m][k] * B[n][k]   (+ bias[n]) -> act -> C / C2
    const void *A; long lda;      // [M][K] op
    const void *B; long ldb;      // [N][K] op
    float *C; long ldc;           // [M][N] fp32 (nullable)
    void *C2; long ldc2;          // [M][N] op copy (nullable)
    const float *bias;            // [N] fp32 added before the activation (nullable)
    int act;                      // Act
    int M, N, K;                  // K multiple of 8 (bf16) / 4 (f32); N multiple of 32
    // Row map of the fraction (launch_rowmap; nullable): rowcnt[0] real frames whose rows are rowmap[0 ..), rowcnt[1] dummy ones
    // (patType NONE at a time step where the kernels check it: pad slots, frames behind the end of a sequence) in dummymap[0 ..).  A kernel that takes the map computes the
    // real rows only and writes bias[n] (or 0) into the dummy rows -- what the full product gives there, because every operand row
    // of a dummy frame is zero (zero-padded inputs, y = 0 by ComputeBlockOutputFn's dummy rule, deltas = 0 by ComputeBlockErrorsFn's).
    // Kernels that do not take it compute all M rows.  m_est: the host's estimate of rowcnt[0] (0: M), for the dispatch only.
    const int *rowmap, *dummymap, *rowcnt; int m_est;
};
struct GemmTN {            // C[m][n] += sum_k A[k][m] * B[k][n]   k in [0,K), fp32 atomics (split-K)
    const void *A; long lda;      // [K][M] op
    const void *B; long ldb;      // [K][N] op
    float *C; long ldc;           // [M][N] fp32, pre-zeroed
    int M, N, K;                  // M, N multiples of 32
    // deterministic mode (cn_ctx option "deterministic"): split s STORES its partial product to ws + s * M * ldc (same pitch as
    // C) and a second launch adds the partials IN SPLIT ORDER into C -- the sum over the frames no longer depends on the order in
    // which workgroups retire (ComputeWeightUpdateFn, LstmLayer.cu:289-512, is one serial sum per weight).  nullptr: fp32 atomics.
    float *ws; int ws_splits;     // workspace of ws_splits * M * ldc floats; the launcher never cuts K into more splits than that
    // ws_used (host pointer, nullable): the launcher writes the number of splits it cut there and does NOT launch the fold -- the
    // caller's consumer adds the partials itself (pack_group_kernel in its update = 2 form: no extra launch behind the product)
    int *ws_used;
};
constexpr int DET_MAX_SPLITS = 8;
// dst[r][c] (+)= part[0][r][c] + part[1][r][c] + ... in that order, r < rows, c < cols (partials share dst's pitch `ld`, one every
// `stride` floats).  accumulate: onto what dst holds (else dst is overwritten); clear: the partials are zeroed behind the read.
struct FoldItem { float *dst; float *part; long stride; int nparts, rows, cols, ld, accumulate, clear; };
constexpr int FOLD_MAX = 4;
void launch_fold(hipStream_t s, const FoldItem *items, int n);
// `done`: optional event that completes with the kernel itself (hipExtLaunchKernelGGL stop event): a fork point for
// another stream without a marker packet on this stream (an hipEventRecord between two kernels costs the second one ~7 us)
void launch_gemm_nt(hipStream_t s, int prec, const GemmNT &g, hipEvent_t done = nullptr);
// 256 x 256 LDS-DMA variant for the MFMA-bound shapes (cn_gemm_big.hip); launch_gemm_nt dispatches to it
bool gemm_nt_big_applies(int prec, const GemmNT &g);
void launch_gemm_nt_big(hipStream_t s, int prec, const GemmNT &g, hipEvent_t done = nullptr);
// 128 x 256 tiles, two workgroups per CU, for the output-bound short-K products (cn_gemm_nt_mid.hip)
bool gemm_nt_mid_applies(int prec, const GemmNT &g);
// one 64-row panel per CU walked as one fill pipeline, for the N-wide products of short fractions (cn_gemm_nt_panel.hip)
bool gemm_nt_panel_applies(int prec, const GemmNT &g, int cus);
void launch_gemm_nt_panel(hipStream_t s, const GemmNT &g, hipEvent_t done = nullptr);
void launch_gemm_nt_mid(hipStream_t s, const GemmNT &g, hipEvent_t done = nullptr);
// cu_budget: CUs the launch may fill with its one-per-CU workgroups when it goes to the 256 x 256 kernel (0 = the chip)
// `extra` (deterministic mode, nullable): one more fold that rides on the launch that adds this product's split partials (the
// layer's bias / peephole / column partial sums): one launch behind the product instead of two
void launch_gemm_tn(hipStream_t s, int prec, const GemmTN &g, int cu_budget = 0, const FoldItem *extra = nullptr);
// 256 x 256 LDS-DMA variant for the products whose operands do not fit the caches (cn_gemm_tn_big.hip); launch_gemm_tn /
// launch_gemm_tn_group dispatch to it
bool gemm_tn_big_applies(int prec, const GemmTN &g);    // on its own
bool gemm_tn_big_can(int prec, const GemmTN &g);        // beside a product that applies (one grouped launch)
void launch_gemm_tn_big_group(hipStream_t s, const GemmTN *gs, int n, int cu_budget = 0, const FoldItem *extra = nullptr);   // n <= 3
void launch_gemm_tn_group(hipStream_t s, int prec, const GemmTN *gs, int n, int cu_budget = 0, const FoldItem *extra = nullptr);      // up to 3 small products in one launch
void launch_gemm_tn_small_group(hipStream_t s, int prec, const GemmTN *gs, int n, const FoldItem *extra = nullptr);   // ... on the 64 x 64 / 128 x 128 tiles whatever their size

// ---- recurrent LSTM kernels --------------------------------------------------------------------
struct LstmRec {
    int H, Hp, dirs, PS, T, Tmin;
    const char *pat;              // [T*PS]
    // forward
    float *acts;                  // [N][dirs][Hp][4] fp32: pre-activations in, n/i/f/o activations out (gate innermost)
    const void *pre16;            // nullable: the pre-activations as bf16 in the same [N][dirs][Hp][4] order (8 bytes per unit and frame);
                                  // the forward kernel then takes them from here and `acts` is output only (lstm_fwd_takes_pre16)
    float *cell;                  // [N][dirs][Hp]    fp32
    float *th;                    // [N][dirs][Hp]    fp32 tanh(cell state), kept by the forward pass for the backward pass
    void  *y_op;                  // [N][dirs*Hp]     op  (layer output, GEMM operand)
    const void *Wrec;             // [dirs][4*Hp][Hp] op  (k = source unit contiguous)
    const float *peep;            // [dirs][3][Hp]
    // backward
    const float *err;             // [N][dirs*Hp] fp32 outputErrors of this layer
    void  *delta_op;              // [N][dirs][Hp][4] op
    const void *WrecT;            // [dirs][Hp][4*Hp] op  (k = 4*target unit + gate contiguous)
    float *dbias;                 // [dirs][Hp][4] fp32 accumulators (pre-zeroed)
    float *dpeep;                 // [dirs][3][Hp]
    float bias;                   // JSON bias value (scales the bias gradient)
    // deterministic mode: workgroup b of the backward kernel STORES its bias / peephole sums into slot b of gpart (a slot: dbias's
    // [dirs][Hp][4] then dpeep's [dirs][3][Hp], 7 * dirs * Hp floats; entries a workgroup does not own stay zero) instead of adding
    // them to dbias / dpeep with atomics; the launcher writes the grid it used to *det_grid (<= gpart_slots or it refuses) and the
    // caller folds the slots in workgroup order (launch_fold, clear = 1).  nullptr: atomics.
    float *gpart; int gpart_slots; int *det_grid;
    int rpl;                      // sequences per lane (1/2/4); PS is a multiple of 4*rpl (padded slots are dummies)
    // multi-CU cluster kernels (cn_lstm_cluster.hip)
    unsigned long long *xch;      // exchange granules (nullable: cluster path off), zeroed at allocation
    unsigned long long *xch_packed;   // ... the packed granules' region of the same buffer (lstm_cluster_xch_packed_offset)
    unsigned xch_epoch;           // tags of this launch are xch_epoch + 1 ... xch_epoch + T (launch_lstm_cluster sets it and advances the counter)
    int *fault;                   // set to 1 by a bounded spin that gave up
    int num_cus;                  // CUs of the device (the one-CU kernels' "does the grid fit the chip in one wave" rule)
    int cluster_cus;              // CUs a cluster grid may count on: a cluster grid must be resident as a whole (spin-wait hand-off),
                                  // so with a communicator bound this is num_cus minus a margin for RCCL's channels
    char *kname;                  // nullable, CN_KNAME_LEN bytes: the launcher writes the name of the kernel it instantiated
};
constexpr int CN_KNAME_LEN = 64;
// every launcher of a recurrent kernel reports its grid (deterministic mode folds that many slots of gpart) and refuses one
// the slots do not cover
void lstm_note_grid(const LstmRec &p, int grid);
// time steps of zeros the library keeps in front of and behind acts / cell / th / err / pat of an LSTM layer (cn_api.cpp:
// dalloc_guarded): a recurrent loop may load up to this many steps outside [0, T)
constexpr int CN_GUARD_STEPS = 6;
size_t lstm_rec_lds_bytes(int prec, bool bwd, int Hp, int rpl, int T);        // dynamic LDS per workgroup of the single-CU kernels
bool lstm_rec_resident(int prec, int Hp);                                      // W_rec fragments register resident (single-CU kernels)
void launch_lstm_forward(hipStream_t s, int prec, const LstmRec &p);
// does the forward kernel launch_lstm_forward would pick for this shape read bf16 pre-activations (LstmRec::pre16)?  The input
// projection then writes 8 instead of 16 bytes per unit and frame -- the N-wide product is bound by that store (round 6).
bool lstm_fwd_takes_pre16(int prec, const LstmRec &p);
// "s2" shape (cn_lstm_s2.hip): two sequences per workgroup, one wave per SIMD, 32 units per wave; launch_lstm_forward /
// launch_lstm_backward dispatch to it when it applies
bool lstm_s2_applies(int prec, const LstmRec &p, bool bwd);
void launch_lstm_s2(hipStream_t s, int prec, bool bwd, const LstmRec &p, hipEvent_t done = nullptr);
bool lstm_s2w_applies(int prec, const LstmRec &p, bool bwd);
void launch_lstm_s2w(hipStream_t s, bool bwd, const LstmRec &p, hipEvent_t done = nullptr);
void launch_lstm_backward(hipStream_t s, int prec, const LstmRec &p, hipEvent_t done = nullptr);   // done: see launch_gemm_nt
// cluster variants for layers whose W_rec exceeds one CU; return false when the shape is not covered
// `num_cus`: the CU count of the device; the spin-wait hand-off needs every member workgroup resident, so a grid larger
// than the device (a partitioned or CU-masked part) does not take the cluster path
size_t lstm_cluster_xch_bytes(int prec, int Hp, int dirs, int PS, int rpl, int num_cus);
size_t lstm_cluster_xch_packed_offset(int prec, int Hp, int dirs, int PS, int rpl, int num_cus);   // bytes from the start of the buffer
int lstm_cluster_size(int prec, int Hp, int dirs, int PS, int rpl, int num_cus);     // CUs per cluster, 0 = path does not apply
// `epoch`: the context's granule-tag counter; the launcher hands tags epoch + 1 ... epoch + T to this launch and advances
// the counter by T + 1, so no caller can forget to (stale granules of an earlier launch never match)
bool launch_lstm_cluster(hipStream_t s, int prec, bool bwd, LstmRec &p, unsigned *epoch);
int lstm_cluster_bwd_cus(int prec, const LstmRec &p);     // CUs the backward cluster launch of this shape occupies (0 = no cluster shape)
void lstm_cluster_stream_gone(hipStream_t s);         // cn_ctx_destroy: the per-device launch gate forgets the stream

// ---- element-wise / packing kernels -----------------------------------------------------------
struct LstmGeom { int P, Pp, L, H, Hp, dirs; int prevH, prevHp, prevDirs; /* prevH=0: identity column map */ };

// flat fp32 reference weights -> packed op copies + fp32 bias/peephole vectors
void launch_lstm_pack(hipStream_t s, bool f32, const LstmGeom &g, float bias, const float *w,
                      void *Win, void *WinT, void *Wrec, void *WrecT, float *bias_p, float *peep_p);
// packed fp32 gradients -> flat reference layout
// (the packed accumulators are cleared as they are read)
void launch_lstm_unpack_grads(hipStream_t s, const LstmGeom &g, float *dWin, float *dWrec, float *dbias, float *dpeep, float *wu, hipEvent_t done = nullptr);
struct FfGeom { int P, Pp, L, Lp; int prevH, prevHp, prevDirs; };
// all trainable layers' operand copies in one launch
constexpr int PACK_GROUP_MAX = 8;
// deterministic mode, update == 2: where the partial sums of one packed gradient array lie (nparts == 0: read the packed
// accumulator itself).  The consumer adds part[0], part[stride], ... in that order -- the sum launch_fold would have formed.
struct PackFold { float *part; long stride; int nparts, clear; };
struct PackItem {
    int lstm; LstmGeom lg; FfGeom fg; float bias; const float *w;
    void *Win, *WinT, *Wrec, *WrecT; float *bias_p, *peep_p;
    // update == 1: the momentum-SGD step of SteepestDescentOptimizer.cu:39-59 is applied on the way (every flat weight is read by
    // exactly one packed position, so the thread that packs it also updates it): wd = mom*wd - lr*wu; w += wd
    int update; float *w_rw; const float *wu; float *wd; float lr, mom;
    // update == 2 (cn_ctx_arm_update, no communicator): the gradient is taken straight from the PACKED accumulators the gradient
    // GEMMs / recurrent kernel summed into (every packed position is visited by exactly one thread, which also clears it and
    // writes the flat weightUpdates entry): unpack + update + operand copies in ONE launch behind the layer's gradient GEMMs
    float *wu_rw; float *g_in, *g_rec, *g_bias, *g_peep;      // lstm: dWin, dWrec, dbias, dpeep; ff: dW (g_in), colsum (g_bias)
    // update == 3: ... in deterministic mode from the partial sums their producers stored (PackFold): dWin's splits, dWrec's per direction,
    // the backward workgroups' bias / peephole slots (peephole entries R floats into a slot) or the column-sum rows
    PackFold f_in, f_rec[2], f_bias;
};
struct PackGroup { PackItem item[PACK_GROUP_MAX]; int first[PACK_GROUP_MAX]; int n; };
void launch_pack_group(hipStream_t s, bool f32, PackGroup &grp, hipEvent_t done = nullptr);
void launch_ff_pack(hipStream_t s, bool f32, const FfGeom &g, float bias, const float *w,
                    void *W, void *WT, float *bias_p);
void launch_ff_unpack_grads(hipStream_t s, const FfGeom &g, float bias, float *dW, float *colsum, float *wu, hipEvent_t done = nullptr);

// inputs [N][P] fp32 (reference layout) -> [N][Pp] op, zero padded
// rm (nullable): the fraction's row map, built behind the re-layout: rm[0] real frames, rm[1] dummy ones, rm[4 ..) the real rows in
// ascending order, rm[4 + maxN ..) the dummy rows
void launch_rowmap(hipStream_t s, const char *dpat, int N, int *rm, int maxN, int unchecked);    // rows < unchecked count as real
void launch_fraction_load(hipStream_t s, bool f32, int T, int PS, int PSp, const char *pat, char *dpat, const int *tcls, int *dtcls,
                          const float *tgt, float *dtgt, int W, const float *in, int P, void *dst, int Pp, int *rm = nullptr, int maxN = 0, int Tmin = 0);
void launch_pad_convert(hipStream_t s, bool f32, const float *src, int N, int P, void *dst, int Pp);
// delta = act'(y) * err (in place on err, all N slots: FeedForwardLayer.cu:72-79), op copy for the GEMMs
void launch_ff_delta(hipStream_t s, bool f32, int act, const float *y, float *err, void *delta_op, int N, int L, int Lp);
// column sums of delta over the N slots (FeedForwardLayer.cu:82-102): colsum[j] += sum_n err[n][j]
// det_part (nullable; deterministic mode): det_colsum_part_floats(Lp) floats; the workgroups store their partial sums there and a
// second launch adds them in workgroup order
// fold_out (nullable): the fold is not launched but described there (nparts = 0: nothing to fold) for the caller to attach to
// the layer's gradient product (launch_gemm_tn's `extra`)
void launch_colsum(hipStream_t s, const float *err, int N, int Lp, float *colsum, float *det_part = nullptr, FoldItem *fold_out = nullptr);
size_t det_colsum_part_floats(int Lp);
// softmax rows in place (SoftmaxLayer.cu:250-315), dummies skipped
// optional: tcls + rowstat[N][2] = {log p_target, argmax == target} for the multiclass loss
// smstat (nullable, wide rows only: softmax_fwd_can_be_lazy): the LAZY form -- y keeps the logits, smstat[N][2] = {offset, sum};
// launch_softmax_mcc_bwd(.., smstat) and launch_softmax_normalise recompute the very same posteriors from them
// fast (wide rows only; the bf16 throughput mode): v_exp_f32 and one reciprocal per row in place of expf and a division per element
void launch_softmax_fwd(hipStream_t s, float *y, const char *pat, int N, int L, int Lp, const int *tcls, float *rowstat, bool fast = false, float *smstat = nullptr);
bool softmax_fwd_can_be_lazy(int L);
void launch_softmax_normalise(hipStream_t s, bool fast, float *y, const char *pat, int N, int L, int Lp, const float *smstat);
void launch_rowstat_reduce(hipStream_t s, const float *rowstat, int N, float *loss2, bool reset, float scale = -1.0f);
// remaining post output layers (row f4): per-pattern terms -> rowstat -> fixed-order sum
enum { POST_SSE = 0, POST_WEIGHTEDSSE, POST_SSE_MASK, POST_CE, POST_RMSE, POST_BINARY };
void launch_post_eval(hipStream_t s, int kind, const float *y, const float *tgt, const char *pat, int N, int L, int Lp,
                      float *rowstat, float *loss2, bool reset);
void launch_post_backward(hipStream_t s, int kind, const float *y, const float *tgt, const char *pat, int N, int L, int Lp, float *err);
void launch_classes_to_targets(hipStream_t s, const int *tcls, float *tgt, int N);
// multiclass error injection + softmax Jacobian + delta copy + bias column sums in one pass (Lp <= 256)
// rowstat / loss2 (nullable; narrow rows only, softmax_mcc_bwd_takes_loss): the launch also sums the forward pass's row
// statistics into loss2 like launch_rowstat_reduce(..., reset = false) would (the same sums in the same order), in sixteen extra workgroups
// `loss_part`: 16 x {float sum, int count} + one arrival counter (zero between launches) for the sixteen reduction workgroups
void launch_softmax_mcc_bwd(hipStream_t s, bool f32, const float *y, const int *tcls, const char *pat, int N, int L, int Lp,
                            float *err, void *delta_op, float *colsum, const float *rowstat = nullptr, float *loss2 = nullptr, float *loss_part = nullptr,
                            const float *smstat = nullptr, bool fast = false, float *colpart = nullptr, float *det_part = nullptr, FoldItem *fold_out = nullptr);
bool softmax_mcc_bwd_takes_loss(int Lp);
// `colpart` (nullable; narrow rows): softmax_mcc_bwd_colpart_floats() zeroed floats the launch spreads its column-sum atomics over
// (replicas folded into colsum by the last workgroup; zero again when the launch ends)
size_t softmax_mcc_bwd_colpart_floats();
// e_i <- y_i (e_i - sum_j y_j e_j) (SoftmaxLayer.cu:317-349), dummies skipped
void launch_softmax_bwd(hipStream_t s, const float *y, float *err, const char *pat, int N, int L, int Lp);
// multiclass_classification: loss/#correct reduction and error injection
// rowstat: [N][2] scratch the per-pattern terms pass through (summed in a fixed order: launch_rowstat_reduce)
void launch_mcc_eval(hipStream_t s, const float *y, const int *tcls, int N, int L, int Lp, float *loss2 /*[2]*/, bool reset, float *rowstat);
void launch_mcc_backward(hipStream_t s, const float *y, const int *tcls, int N, int L, int Lp, float *err);
// sse
// UpdateWeightFn over a flat range
void launch_scale(hipStream_t s, float *x, size_t n, float a);     // x *= a
// dst[i] = src[0][i] + src[1][i] + ... in that order (cn_comm_ipc.cpp: the test backend of the gradient exchange)
struct SumRanks { const float *src[8]; int n; };
void launch_sum_ranks(hipStream_t s, float *dst, const SumRanks &sr, size_t n);

// ---- CN_COMM_BACKEND=ipc: test backend of cn_comm_* for ranks that share a device (cn_comm_ipc.cpp) ----------------
struct IpcComm;
bool ipc_backend_selected();
void ipc_unique_id(char *id, size_t bytes);
IpcComm *ipc_comm_create(const char *id, int rank, int world);
void ipc_comm_destroy(IpcComm *c);
void ipc_comm_mark_failed(IpcComm *c);
void ipc_allreduce(IpcComm *c, float *buf, size_t n, hipStream_t st, size_t capacity_hint = 0);
bool ipc_comm_is_p2p(const IpcComm *c);
void ipc_comm_check(IpcComm *c, hipStream_t st);       // p2p: raise if a poll of the communicator timed out (synchronises st)
void ipc_comm_check_fast(IpcComm *c);                  // p2p: the same from the host-mapped word the kernel sets (no synchronisation)
// p2p: allocate and map the regions and run the first-contact self-check (flags and staged lines through every peer's mapping,
// both forms of the exchange, sums verified on the host).  False: some rank saw a wrong sum or a time-out -- every rank gets the
// same answer -- and `rccl_id` (CN_COMM_ID_BYTES) holds what rank 0's `make_id` produced: the caller fails over to RCCL.
bool ipc_comm_p2p_selfcheck(IpcComm *c, hipStream_t st, void (*make_id)(char *), char *rccl_id);
void ipc_allreduce_loss(IpcComm *c, float *err, int *correct);
// ---- CN_COMM_BACKEND=p2p: one stream-ordered kernel per bucket over peer-mapped regions (cn_comm_p2p.hip) -----------
constexpr int P2P_GROUPS = 64, P2P_THREADS = 256;
// a rank's region starts with 64-bit flag words: [word][source rank][workgroup]
constexpr int P2P_READY = 0, P2P_REDUCED = 8 * P2P_GROUPS, P2P_DONE = 16 * P2P_GROUPS, P2P_FAILED = 24 * P2P_GROUPS, P2P_FLAG_WORDS = 24 * P2P_GROUPS + 64;
struct P2pArgs {
    float *buf; size_t n, piece, slot;         // the bucket (in place); floats per piece (even): n <= world * P2P_GROUPS * piece; per staging slot (>= piece)
    float *stage[8];                           // staging half of this exchange, per rank (peers' mapped into this process)
    unsigned long long *flags[8];              // flag words, per rank
    int me, world, two_phase;
    unsigned long long seq, timeout_ticks;     // exchange counter (from 1); poll deadline in ticks of the 100 MHz clock
    unsigned long long *host_failed;           // host-mapped word: set when a wait of this exchange ended without its flag
};
void launch_p2p_allreduce(hipStream_t s, const P2pArgs &a);
void launch_sgd(hipStream_t s, float *w, const float *wu, float *wd, size_t n, float lr, float mom, hipEvent_t done = nullptr);
void launch_accumulate(hipStream_t s, float *acc, const float *wu, size_t n, bool first);
// gather a padded row-major fp32/op matrix into the reference layout [N][L]
// (host row n = t*PS + s maps to device row t*PSp + s)
void launch_unpad(hipStream_t s, bool src_is_bf16, const void *src, long ld, int col0, int cstride, int N, int L, float *dst, long ldd, int dcol0, int PS, int PSp);
// scatter host-provided [N][L] fp32 into a padded fp32 matrix (tests)
void launch_pad_f32(hipStream_t s, const float *src, int N, int L, float *dst, long ld, int prevH, int prevHp, int PS, int PSp);

}  // namespace cn

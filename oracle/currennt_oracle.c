/*
 * currennt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A scalar fp32 CPU restatement of the `Cpu` (Thrust-host) instantiation of the
 * CURRENNT LSTM training hot path as found in naxingyu/lstm-rnn.  It is used
 * only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
 * the checker / reported CPU baseline.  The shipped library (lstm-rnn_amd/csrc)
 * never links, loads or calls anything in this file.
 *
 * Every function cites the reference file:line (relative to
 * currennt_lib/src/) whose arithmetic and summation order it restates.
 *
 * PINNING: (1) oracle/_ref -- the reference's own object code for the arithmetic of this path: helpers/Matrix.cu
 * compiled as it lies under /root/reference, and the functors of layers/LstmLayer.cu, FeedForwardLayer.cu,
 * SoftmaxLayer.cu and MulticlassClassificationLayer.cu driven by a harness (the oracle/ref sources) in the reference's call
 * order.  tests/test_oracle_ref.py holds this file BIT-EQUAL to it (all LSTM internals, outputs, propagated errors,
 * every gradient, five training steps of the KAT-0 network); tests/golden/ref_golden.npz carries vectors generated from
 * it for machines without /root/reference (tests/test_oracle_golden.py).  The layer CLASSES of the reference
 * (TrainableLayer.cu and up) need Boost, absent in this image, and are not built: the time-loop call sequences
 * (LstmLayer.cu:763-1051) are restated in the harness, and (2) KAT-0 pins those end to end (error,
 * #correct and per-layer sums of tests/test1/network.jsn on the first 10 sequences of
 * examples/speech_recognition_chime/val_1_speaker.nc, recorded by the survey session from a shim-assisted build of the
 * full layer classes; tests/test_oracle_kat0.py).  (3) An independent fp64 autograd model of the same equations
 * (tests/test_oracle_autograd.py).  The reference's own test (tests/test1) pins nothing in this fork
 * (expected_network.jsn == network.jsn).
 *
 * Conventions (SURVEY.md section 7 "Memory layouts"):
 *   layer activations  a[(t*PS + ps)*L + unit]      (column-major L x N, N = T*PS)
 *   patTypes           char[N], 0 = NONE (dummy slot)
 *   LSTM per-direction internals  b[(t*PS + ps)*H + j]
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off keeps `x += a*b` a separate multiply and add like the
 * reference's x86-64 host build (no FMA target feature).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Threads (tests only; bench.py's cpu_baseline stays at 1 = the reference's single-threaded Thrust-host build,
 * CMakeLists.txt:11-13).  Work is split over OUTPUT elements only: every sum still runs serially over k on one
 * thread in the reference's order, so results are bit-identical for any thread count. */
static int g_threads = 1;
void orc_set_threads(int n)
{
    g_threads = n > 0 ? n : 1;
#ifdef _OPENMP
    omp_set_num_threads(g_threads);
#endif
}
int orc_get_threads(void) { return g_threads; }
#define ORC_PAR(work) const long orc_work_ = (long)(work); (void)orc_work_; \
    _Pragma("omp parallel for schedule(static) if (g_threads > 1 && orc_work_ > 400000)")

typedef float real_t;

#define PATTYPE_NONE 0

/* ------------------------------------------------------------------------- */
/* operand rounding (a MODEL of the product's bf16 mode, not reference code)   */
/* ------------------------------------------------------------------------- */
/*
 * orc_set_operand_rounding(1): every matrix product of the path -- Matrix.cu's assignProduct / addProduct call sites
 * (LstmLayer.cu:774-785,815-818,850-853,939-942,973-976,996-1006, FeedForwardLayer.cu:148-152,190-197,202-206) and the
 * input / internal weight cases of ComputeWeightUpdateFn (LstmLayer.cu:370-390,410-437) -- rounds BOTH operands to bf16
 * (round to nearest even) before multiplying, and an LSTM layer stores its outputs rounded (the HIP library keeps them in
 * bf16 only).  Everything else is untouched fp32: accumulation in the reference's order, cell states, gate activations
 * with libm expf, bias and peephole weights, the clipped deltas carried to the neighbouring time step, the bias / peephole
 * gradient sums (they read the unrounded deltas), softmax, losses, the update.  This is exactly the set of values
 * CN_PREC_BF16 rounds (`op` buffers and packed weights), so what remains between this mode and the
 * HIP path is summation order and v_exp_f32 / v_rcp_f32 -- tests/test_gpu_bf16_pinned.py holds that to 2e-4.
 * Mode 0 (default) is the reference's arithmetic; no statement of mode 0 changes when the mode exists: the rounded
 * copies are made in front of the unchanged loops.
 */
#include <stdint.h>
static int g_opround = 0;
void orc_set_operand_rounding(int mode) { g_opround = mode ? 1 : 0; }
int orc_get_operand_rounding(void) { return g_opround; }
/*
 * orc_set_preact_rounding(1) (round 6; only looked at in operand-rounding mode, by the NEXT orc_lstm_forward calls): the gate
 * pre-activations of the input projection (LstmLayer.cu:771-786) are kept in bf16 WITH the bias term the product's epilogue adds
 * (x W_in + bias w_b, rounded to nearest even) -- what CN_PREC_BF16 stores between its input-projection GEMM and the recurrent
 * kernels of the layers that take them in that form (cn_layer_recurrent_kernel: the "_s2_" forward kernels).  The functor
 * (block_output) keeps adding bias w_b itself, so the model subtracts it again behind the rounding: (bf16(a + b) - b) + b differs
 * from bf16(a + b) by an fp32 rounding at most, far inside the pinned tolerances.
 */
static int g_preround = 0;
void orc_set_preact_rounding(int on) { g_preround = on ? 1 : 0; }
int orc_get_preact_rounding(void) { return g_preround; }

static real_t bf16_rne(real_t v)
{
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return v;              /* inf / nan: as is */
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&v, &u, 4);
    return v;
}
static real_t *rounded_copy(const real_t *src, size_t n)
{
    real_t *q = (real_t *)malloc(sizeof(real_t) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) q[i] = bf16_rne(src[i]);
    return q;
}
/* operands of one product: rounded copies in mode 1, the caller's arrays otherwise */
#define MM_OPERANDS(na, nb) \
    real_t *aq_ = NULL, *bq_ = NULL; \
    if (g_opround) { aq_ = rounded_copy(a, (size_t)(na)); bq_ = rounded_copy(b, (size_t)(nb)); a = aq_; b = bq_; }
#define MM_RELEASE() do { free(aq_); free(bq_); } while (0)

/* helpers/NumericLimits.cuh:39-43 */
#define NL_MIN      1.1754944e-038f
#define NL_MAX      3.4028235e+038f
#define NL_EXPLIMIT 88.722839f
#define NL_LOGZERO  (-1e30f)
#define SKIP_MARKER NL_MAX            /* SoftmaxLayer.cu:37 */

/* ------------------------------------------------------------------------- */
/* activation functions                                                       */
/* ------------------------------------------------------------------------- */

/* activation_functions/Logistic.cuh:33-44 */
static real_t logistic_fn(real_t x)
{
    if (x < NL_EXPLIMIT) {
        if (x > -NL_EXPLIMIT)
            return (real_t)1.0 / ((real_t)1.0 + expf(-x));
        else
            return 0;
    }
    return 1;
}
/* Logistic.cuh:46-49 */
static real_t logistic_deriv(real_t y) { return y * ((real_t)1.0 - y); }

/* Maxmin1.cuh:33-36 */
static real_t maxmin1_fn(real_t x) { return ((real_t)2.0 * logistic_fn(x) - (real_t)1.0); }
/* Tanh.cuh:33-36: tanh(x) = maxmin1(2x) */
static real_t tanh_fn(real_t x) { return maxmin1_fn((real_t)2.0 * x); }
/* Tanh.cuh:38-41 */
static real_t tanh_deriv(real_t y) { return (real_t)1.0 - (y * y); }

/* helpers/boundRange.cuh:31-34, limitedError.cuh:31-34 */
static real_t limited_error(real_t e)
{
    return (e < -1.0f ? -1.0f : (e > +1.0f ? +1.0f : e));
}

/* helpers/safeExp.cuh:31-40 */
static real_t safe_exp(real_t x)
{
    if (x <= NL_LOGZERO)
        return 0;
    else if (x >= NL_EXPLIMIT)
        return NL_MAX;
    else
        return expf(x);
}

/* activation ids shared with the python wrapper */
enum { ACT_TANH = 0, ACT_LOGISTIC = 1, ACT_IDENTITY = 2 };

static real_t act_fn(int act, real_t x)
{
    switch (act) {
    case ACT_TANH:     return tanh_fn(x);
    case ACT_LOGISTIC: return logistic_fn(x);
    default:           return x;                 /* Identity.cuh:33-36 */
    }
}
static real_t act_deriv(int act, real_t y)
{
    switch (act) {
    case ACT_TANH:     return tanh_deriv(y);
    case ACT_LOGISTIC: return logistic_deriv(y);
    default:           return 1;                 /* Identity.cuh:38-41 */
    }
}

/* ------------------------------------------------------------------------- */
/* helpers/Matrix.cu naive products (column-major, ld = rows)                 */
/* ------------------------------------------------------------------------- */

/* C(rowsA x colsB) (+)= A(rowsA x colsA) * B(colsA x colsB)
 * Matrix.cu:41-62 (MatrixMultiplyFn), :64-85 (AddMatrixMultiplyFn) */
static void mm_nn(real_t *c, const real_t *a, int rowsA, int colsA,
                  const real_t *b, int rowsB, int colsB, int add)
{
    int total = rowsA * colsB;
    MM_OPERANDS((size_t)rowsA * colsA, (size_t)rowsB * colsB)
    ORC_PAR((long)total * colsA)
    for (int idx = 0; idx < total; ++idx) {
        const real_t *offRowA = a + (idx % rowsA);
        const real_t *offColB = b + (idx / rowsA) * rowsB;
        real_t x = 0;
        for (int i = 0; i < colsA; ++i)
            x += offRowA[i * rowsA] * offColB[i];
        c[idx] = add ? c[idx] + x : x;
    }
    MM_RELEASE();
}

/* C(colsA x colsB) (+)= A^T * B,  A(rowsA x colsA), B(rowsA x colsB)
 * Matrix.cu:87-108 (MatrixMultiplyTransposedAFn), :110-131 (Add...) */
static void mm_tn(real_t *c, const real_t *a, int rowsA, int colsA,
                  const real_t *b, int rowsB, int colsB, int add)
{
    int total = colsA * colsB;
    MM_OPERANDS((size_t)rowsA * colsA, (size_t)rowsB * colsB)
    ORC_PAR((long)total * rowsA)
    for (int idx = 0; idx < total; ++idx) {
        const real_t *offColA = a + (idx % colsA) * rowsA;
        const real_t *offColB = b + (idx / colsA) * rowsB;
        real_t x = 0;
        for (int i = 0; i < rowsA; ++i)
            x += offColA[i] * offColB[i];
        c[idx] = add ? c[idx] + x : x;
    }
    MM_RELEASE();
}

/* C(rowsA x rowsB) (+)= A * B^T,  A(rowsA x colsA), B(rowsB x colsA)
 * Matrix.cu:133-157 (MatrixMultiplyTransposedBFn), :159-183 (Add...) */
static void mm_nt(real_t *c, const real_t *a, int rowsA, int colsA,
                  const real_t *b, int rowsB, int colsB, int add)
{
    (void)colsB;
    int total = rowsA * rowsB;
    MM_OPERANDS((size_t)rowsA * colsA, (size_t)rowsB * colsA)
    ORC_PAR((long)total * colsA)
    for (int idx = 0; idx < total; ++idx) {
        const real_t *offRowA = a + (idx % rowsA);
        const real_t *offRowB = b + (idx / rowsA);
        real_t x = 0;
        for (int i = 0; i < colsA; ++i) {
            x += *offRowA * *offRowB;
            offRowA += rowsA;
            offRowB += rowsB;
        }
        c[idx] = add ? c[idx] + x : x;
    }
    MM_RELEASE();
}

/* exported for direct unit tests of the three product kinds */
void orc_matmul(int kind, real_t *c, const real_t *a, int rowsA, int colsA,
                const real_t *b, int rowsB, int colsB, int add)
{
    if (kind == 0)      mm_nn(c, a, rowsA, colsA, b, rowsB, colsB, add);
    else if (kind == 1) mm_tn(c, a, rowsA, colsA, b, rowsB, colsB, add);
    else                mm_nt(c, a, rowsA, colsA, b, rowsB, colsB, add);
}

/* ------------------------------------------------------------------------- */
/* LSTM layer                                                                 */
/* ------------------------------------------------------------------------- */

/* number of weights of an (b)lstm layer: LstmLayer.cu:525, TrainableLayer.cu:101 */
int orc_lstm_weight_count(int P, int L, int bidir)
{
    return L * (4 * (P + 1) + (bidir ? 2 : 4) * L + 3);
}

/*
 * Per-direction internal buffers, each PS*T*H floats, laid out back to back
 * in `bufs` in this order (LstmLayer.hpp:88-100):
 */
enum {
    B_TMPOUT = 0, B_TMPERR, B_CELL, B_CELLERR,
    B_NIACT, B_IGACT, B_FGACT, B_OGACT,
    B_NIDELTA, B_IGDELTA, B_FGDELTA, B_OGDELTA,
    B_COUNT
};
int orc_lstm_internal_count(void) { return B_COUNT; }

typedef struct {
    int P, L, H, dirs, PS, T, Tmin;
    real_t bias;
    const char *patTypes;
    const real_t *w;          /* flat weights, LstmLayer.hpp:36-55 */
    real_t *dir[2][B_COUNT];  /* per-direction internals */
    real_t *dq[2][4];         /* operand rounding mode: bf16 copies of the four delta vectors (NULL otherwise) */
} lstm_t;

static void lstm_bind(lstm_t *l, int P, int L, int bidir, real_t bias, int PS, int maxT,
                      int T, int Tmin, const char *patTypes, const real_t *w, real_t *bufs)
{
    l->P = P; l->L = L; l->dirs = bidir ? 2 : 1; l->H = L / l->dirs;
    l->PS = PS; l->T = T; l->Tmin = Tmin; l->bias = bias;
    l->patTypes = patTypes; l->w = w;
    memset(l->dq, 0, sizeof l->dq);
    size_t per = (size_t)PS * maxT * l->H;     /* LstmLayer.cu:554 */
    for (int d = 0; d < l->dirs; ++d)
        for (int b = 0; b < B_COUNT; ++b)
            l->dir[d][b] = bufs + ((size_t)d * B_COUNT + b) * per;
}

/* weight sub-matrix pointers: LstmLayer.cu:583-596 and :535-541 */
static const real_t *w_input(const lstm_t *l, const real_t *w, int g, int d)
{ return w + (size_t)g * l->L * l->P + (size_t)d * l->H * l->P; }
static const real_t *w_bias(const lstm_t *l, const real_t *w, int g, int d)
{ return w + (size_t)4 * l->L * l->P + (size_t)g * l->L + d * l->H; }
static const real_t *w_internal(const lstm_t *l, const real_t *w, int g, int d)
{ return w + (size_t)4 * l->L * (l->P + 1) + (size_t)g * l->L * l->H + (size_t)d * l->H * l->H; }
static const real_t *w_peep(const lstm_t *l, const real_t *w, int p, int d)
{ return w + (size_t)4 * l->L * (l->P + 1) + (size_t)4 * l->L * l->H + (size_t)p * l->L + d * l->H; }

/* ComputeBlockOutputFn::operator(), LstmLayer.cu:70-137 */
static real_t block_output(const lstm_t *l, int d, int prevOutputDistance,
                           int outputIdx, int firstCall, int checkPatType)
{
    int H = l->H;
    real_t *cellStates = l->dir[d][B_CELL];
    real_t *niActs = l->dir[d][B_NIACT], *igActs = l->dir[d][B_IGACT];
    real_t *fgActs = l->dir[d][B_FGACT], *ogActs = l->dir[d][B_OGACT];

    if (checkPatType) {
        int patIdx = outputIdx / H;
        if (l->patTypes[patIdx] == PATTYPE_NONE) {
            if (prevOutputDistance > 0)
                cellStates[outputIdx] = 0;
            return 0;
        }
    }
    int blockIdx = outputIdx % H;

    real_t niAct = niActs[outputIdx];
    real_t igAct = igActs[outputIdx];
    real_t fgAct = fgActs[outputIdx];
    real_t ogAct = ogActs[outputIdx];

    niAct += l->bias * w_bias(l, l->w, 0, d)[blockIdx];
    igAct += l->bias * w_bias(l, l->w, 1, d)[blockIdx];
    fgAct += l->bias * w_bias(l, l->w, 2, d)[blockIdx];
    ogAct += l->bias * w_bias(l, l->w, 3, d)[blockIdx];

    if (!firstCall) {
        real_t prevCellState = cellStates[outputIdx + prevOutputDistance];
        igAct += prevCellState * w_peep(l, l->w, 0, d)[blockIdx];
        fgAct += prevCellState * w_peep(l, l->w, 1, d)[blockIdx];
    }

    niAct = tanh_fn(niAct);
    igAct = logistic_fn(igAct);
    fgAct = logistic_fn(fgAct);

    niActs[outputIdx] = niAct;
    igActs[outputIdx] = igAct;
    fgActs[outputIdx] = fgAct;

    real_t cellState = niAct * igAct;
    if (!firstCall)
        cellState += cellStates[outputIdx + prevOutputDistance] * fgAct;
    cellStates[outputIdx] = cellState;

    ogAct += cellState * w_peep(l, l->w, 2, d)[blockIdx];
    ogAct = logistic_fn(ogAct);
    ogActs[outputIdx] = ogAct;

    return tanh_fn(cellState) * ogAct;
}

/* the value of a block output as the layer keeps it: fp32, or bf16 in the operand rounding mode */
#define STORE_Y(v) (g_opround ? bf16_rne(v) : (v))

/*
 * LstmLayer<Cpu>::computeForwardPass, LstmLayer.cu:763-886.
 *   x    : preceding layer outputs [N][P]
 *   y    : layer outputs [N][L]
 *   bufs : dirs*12 internal vectors of PS*maxT*H floats
 */
void orc_lstm_forward(int P, int L, int bidir, real_t bias, int PS, int maxT, int T, int Tmin,
                      const char *patTypes, const real_t *w, const real_t *x,
                      real_t *y, real_t *bufs)
{
    lstm_t l;
    lstm_bind(&l, P, L, bidir, bias, PS, maxT, T, Tmin, patTypes, w, bufs);
    int H = l.H, N = T * PS, n = PS * H;
    static const int actBuf[4] = { B_NIACT, B_IGACT, B_FGACT, B_OGACT };

    /* :771-786 input projection, assignProduct(W, true, X, false) */
    for (int d = 0; d < l.dirs; ++d)
        for (int g = 0; g < 4; ++g)
            mm_tn(l.dir[d][actBuf[g]], w_input(&l, w, g, d), P, H, x, P, N, 0);
    if (g_opround && g_preround)               /* model only: pre-activations (bias term included) stored in bf16 */
        for (int d = 0; d < l.dirs; ++d)
            for (int g = 0; g < 4; ++g) {
                real_t *a = l.dir[d][actBuf[g]];
                const real_t *wb = w_bias(&l, w, g, d);
                for (int i = 0; i < N * H; ++i) {
                    real_t b = l.bias * wb[i % H];
                    a[i] = bf16_rne(a[i] + b) - b;
                }
            }

    /* :812-829 forward states */
    for (int t = 0; t < T; ++t) {
        if (t != 0)
            for (int g = 0; g < 4; ++g)        /* :815-818 addProduct(W, true, y[t-1], false) */
                mm_tn(l.dir[0][actBuf[g]] + (size_t)t * n, w_internal(&l, w, g, 0), H, H,
                      l.dir[0][B_TMPOUT] + (size_t)(t - 1) * n, H, PS, 1);
        for (int i = 0; i < n; ++i)            /* :822-828 */
            l.dir[0][B_TMPOUT][(size_t)n * t + i] =
                STORE_Y(block_output(&l, 0, -n, n * t + i, t == 0, t >= Tmin));
    }

    /* :832-865 backward states */
    if (bidir) {
        for (int t = T - 1; t >= 0; --t) {
            if (t != T - 1)
                for (int g = 0; g < 4; ++g)    /* :850-853 */
                    mm_tn(l.dir[1][actBuf[g]] + (size_t)t * n, w_internal(&l, w, g, 1), H, H,
                          l.dir[1][B_TMPOUT] + (size_t)(t + 1) * n, H, PS, 1);
            for (int i = 0; i < n; ++i)        /* :857-863 */
                l.dir[1][B_TMPOUT][(size_t)n * t + i] =
                    STORE_Y(block_output(&l, 1, +n, n * t + i, t == T - 1, t >= Tmin));
        }
    }

    /* :869-885 resort outputs (ResortOutputsFn :148-160); uni: outputs alias tmpOutputs */
    if (bidir) {
        for (int outputIdx = 0; outputIdx < N * L; ++outputIdx) {
            int patIdx = outputIdx / L, valIdx = outputIdx % L;
            int offset = patIdx * H + valIdx;
            y[outputIdx] = (valIdx < H) ? l.dir[0][B_TMPOUT][offset]
                                        : l.dir[1][B_TMPOUT][offset - H];
        }
    } else {
        memcpy(y, l.dir[0][B_TMPOUT], sizeof(real_t) * (size_t)N * L);
    }
}

/* ComputeBlockErrorsFn::operator(), LstmLayer.cu:213-286 */
static void block_errors(const lstm_t *l, int d, int prevOutputDistance, int outputIdx,
                         int firstCall, int lastCall, int checkPatType)
{
    int H = l->H;
    real_t *const *b = l->dir[d];
    real_t outputErr = b[B_TMPERR][outputIdx];

    if (checkPatType) {
        int patIdx = outputIdx / H;
        if (l->patTypes[patIdx] == PATTYPE_NONE) {
            b[B_NIDELTA][outputIdx] = 0;
            b[B_IGDELTA][outputIdx] = 0;
            b[B_FGDELTA][outputIdx] = 0;
            b[B_OGDELTA][outputIdx] = 0;
            b[B_CELLERR][outputIdx] = 0;
            return;
        }
    }
    int blockIdx = outputIdx % H;

    real_t niAct = b[B_NIACT][outputIdx];
    real_t igAct = b[B_IGACT][outputIdx];
    real_t ogAct = b[B_OGACT][outputIdx];
    real_t cellState = b[B_CELL][outputIdx];

    real_t ogDelta = logistic_deriv(ogAct) * tanh_fn(cellState) * outputErr;

    real_t ogPeepWeight = w_peep(l, l->w, 2, d)[blockIdx];
    real_t cellStateErr = ogAct * tanh_deriv(tanh_fn(cellState)) * outputErr + ogPeepWeight * ogDelta;

    if (!firstCall) {
        real_t nextFgAct        = b[B_FGACT][outputIdx - prevOutputDistance];
        real_t nextCellStateErr = b[B_CELLERR][outputIdx - prevOutputDistance];
        real_t nextIgDelta      = b[B_IGDELTA][outputIdx - prevOutputDistance];
        real_t nextFgDelta      = b[B_FGDELTA][outputIdx - prevOutputDistance];
        real_t igPeepWeight = w_peep(l, l->w, 0, d)[blockIdx];
        real_t fgPeepWeight = w_peep(l, l->w, 1, d)[blockIdx];
        cellStateErr += nextFgAct * nextCellStateErr + igPeepWeight * nextIgDelta + fgPeepWeight * nextFgDelta;
    }

    real_t niDelta = igAct * tanh_deriv(niAct) * cellStateErr;

    real_t fgDelta = 0;
    if (!lastCall) {
        real_t fgAct = b[B_FGACT][outputIdx];
        real_t prevCellState = b[B_CELL][outputIdx + prevOutputDistance];
        fgDelta = logistic_deriv(fgAct) * prevCellState * cellStateErr;
    }

    real_t igDelta = logistic_deriv(igAct) * niAct * cellStateErr;

    b[B_NIDELTA][outputIdx] = limited_error(niDelta);
    b[B_IGDELTA][outputIdx] = limited_error(igDelta);
    b[B_FGDELTA][outputIdx] = limited_error(fgDelta);
    b[B_OGDELTA][outputIdx] = limited_error(ogDelta);
    b[B_CELLERR][outputIdx] = cellStateErr;
}

/* ComputeWeightUpdateFn::operator(), LstmLayer.cu:316-511 */
static real_t weight_update(const lstm_t *l, const real_t *plOutputs, int weightIdx)
{
    int layerSize = l->L, effLayerSize = l->H, precLayerSize = l->P;
    int parallelSequences = l->PS;
    int timestepDistance = l->PS * l->H;
    int patternsCount = l->T * l->PS;
    int biasWeightsOffset = layerSize * precLayerSize * 4;
    int internalWeightsOffset = biasWeightsOffset + layerSize * 4;
    int peepholeWeightsOffset = internalWeightsOffset + layerSize * effLayerSize * 4;

    int inwc = layerSize * precLayerSize;
    int biwc = layerSize;
    int itwc = layerSize * effLayerSize;
    int pewc = layerSize;

    int weightType = (int)(weightIdx >= 0                     + 1 * inwc) +
                     (int)(weightIdx >= 0                     + 2 * inwc) +
                     (int)(weightIdx >= 0                     + 3 * inwc) +
                     (int)(weightIdx >= 0                     + 4 * inwc) +
                     (int)(weightIdx >= biasWeightsOffset     + 1 * biwc) +
                     (int)(weightIdx >= biasWeightsOffset     + 2 * biwc) +
                     (int)(weightIdx >= biasWeightsOffset     + 3 * biwc) +
                     (int)(weightIdx >= biasWeightsOffset     + 4 * biwc) +
                     (int)(weightIdx >= internalWeightsOffset + 1 * itwc) +
                     (int)(weightIdx >= internalWeightsOffset + 2 * itwc) +
                     (int)(weightIdx >= internalWeightsOffset + 3 * itwc) +
                     (int)(weightIdx >= internalWeightsOffset + 4 * itwc) * 2 +
                     (int)(weightIdx >= peepholeWeightsOffset + 1 * pewc) +
                     (int)(weightIdx >= peepholeWeightsOffset + 2 * pewc);

    int weightTypeX = weightType & 0xC;
    int weightTypeY = weightType & 0x3;

    const real_t *offOutputs;
    int tgtBlockIdx, offOutputsInc;
    int skipFirstPattern = 0, skipLastPattern = 0, isBwStateWeight;

    switch (weightTypeX) {
    case 0x0: {   /* input weight */
        int plBlockIdx = weightIdx % precLayerSize;
        int blockIdx = (weightIdx - weightTypeY * (biasWeightsOffset / 4)) / precLayerSize;
        isBwStateWeight = (blockIdx >= effLayerSize);
        if (isBwStateWeight) blockIdx -= effLayerSize;
        tgtBlockIdx = blockIdx;
        offOutputs = &plOutputs[plBlockIdx];
        offOutputsInc = precLayerSize;
        break; }
    case 0x4: {   /* bias weight */
        int biasWeightIdx = weightIdx - biasWeightsOffset;
        int blockIdx = biasWeightIdx - weightTypeY * layerSize;
        isBwStateWeight = (blockIdx >= effLayerSize);
        if (isBwStateWeight) blockIdx -= effLayerSize;
        tgtBlockIdx = blockIdx;
        offOutputs = NULL;
        offOutputsInc = 0;
        break; }
    case 0x8: {   /* internal weight */
        int internalWeightIdx = weightIdx - internalWeightsOffset;
        int srcBlockIdx = internalWeightIdx % effLayerSize;
        int blockIdx = internalWeightIdx / effLayerSize - weightTypeY * layerSize;
        isBwStateWeight = (blockIdx >= effLayerSize);
        if (isBwStateWeight) blockIdx -= effLayerSize;
        tgtBlockIdx = blockIdx;
        offOutputs = isBwStateWeight ? &l->dir[1][B_TMPOUT][srcBlockIdx] : &l->dir[0][B_TMPOUT][srcBlockIdx];
        offOutputsInc = effLayerSize;
        if (isBwStateWeight) { offOutputs += timestepDistance; skipLastPattern = 1; }
        else                 { offOutputs -= timestepDistance; skipFirstPattern = 1; }
        break; }
    default: {    /* peephole weight */
        int peepholeWeightIdx = weightIdx - peepholeWeightsOffset;
        int blockIdx = peepholeWeightIdx - (weightTypeY - 1) * layerSize;
        isBwStateWeight = (blockIdx >= effLayerSize);
        if (isBwStateWeight) blockIdx -= effLayerSize;
        const real_t *cellStates = isBwStateWeight ? l->dir[1][B_CELL] : l->dir[0][B_CELL];
        int timeShift;
        if (weightTypeY == 0x3) {
            timeShift = 0;
        } else if (isBwStateWeight) {
            timeShift = timestepDistance; skipLastPattern = 1;
        } else {
            timeShift = -timestepDistance; skipFirstPattern = 1;
        }
        tgtBlockIdx = blockIdx;
        offOutputs = &cellStates[blockIdx + timeShift];
        offOutputsInc = effLayerSize;
        break; }
    }

    static const int deltaBuf[4] = { B_NIDELTA, B_IGDELTA, B_FGDELTA, B_OGDELTA };
    const real_t *offDeltas = &l->dir[isBwStateWeight ? 1 : 0][deltaBuf[weightTypeY]][tgtBlockIdx];
    /* operand rounding mode: the input / internal cases are matrix products (rounded operands: plOutputs arrives rounded,
     * tmpOutputs are stored rounded); the bias / peephole sums read the unrounded deltas */
    if (l->dq[0][0] && (weightTypeX == 0x0 || weightTypeX == 0x8))
        offDeltas = &l->dq[isBwStateWeight ? 1 : 0][weightTypeY][tgtBlockIdx];

    if (skipFirstPattern) {
        offOutputs += parallelSequences * offOutputsInc;
        offDeltas  += parallelSequences * effLayerSize;
    }
    int numPatterns = patternsCount;
    if (skipFirstPattern || skipLastPattern)
        numPatterns -= parallelSequences;

    real_t wu = 0;
    for (int i = 0; i < numPatterns; ++i) {
        wu += (offOutputs ? *offOutputs : l->bias) * *offDeltas;
        offOutputs += offOutputsInc;       /* NULL + 0 for bias weights */
        offDeltas  += effLayerSize;
    }
    return wu;
}

/*
 * LstmLayer<Cpu>::computeBackwardPass, LstmLayer.cu:888-1051.
 *   x        : preceding layer outputs [N][P]
 *   outErr   : this layer's outputErrors [N][L] (input)
 *   prevErr  : preceding layer's outputErrors [N][P]; NULL when the preceding
 *              layer is not trainable (:991-992)
 *   wu       : weightUpdates (same layout as w)
 */
void orc_lstm_backward(int P, int L, int bidir, real_t bias, int PS, int maxT, int T, int Tmin,
                       const char *patTypes, const real_t *w, const real_t *x,
                       const real_t *outErr, real_t *prevErr, real_t *wu, real_t *bufs)
{
    lstm_t l;
    lstm_bind(&l, P, L, bidir, bias, PS, maxT, T, Tmin, patTypes, w, bufs);
    int H = l.H, N = T * PS, n = PS * H;
    static const int deltaBuf[4] = { B_NIDELTA, B_IGDELTA, B_FGDELTA, B_OGDELTA };

    /* :892-910 ResortOutputErrorsFn (:171-187); uni: tmpOutputErrors alias outputErrors */
    if (bidir) {
        for (int outputIdx = 0; outputIdx < N * L; ++outputIdx) {
            int patIdx = outputIdx / L, valIdx = outputIdx % L;
            int offset = patIdx * H + valIdx;
            if (valIdx < H) l.dir[0][B_TMPERR][offset] = outErr[outputIdx];
            else            l.dir[1][B_TMPERR][offset - H] = outErr[outputIdx];
        }
    } else {
        memcpy(l.dir[0][B_TMPERR], outErr, sizeof(real_t) * (size_t)N * L);
    }

    /* :936-951 forward states, t = T-1 .. 0 */
    for (int t = T - 1; t >= 0; --t) {
        if (t != T - 1)
            for (int g = 0; g < 4; ++g)        /* :939-942 addProduct(W, false, delta[t+1], false) */
                mm_nn(l.dir[0][B_TMPERR] + (size_t)t * n, w_internal(&l, w, g, 0), H, H,
                      l.dir[0][deltaBuf[g]] + (size_t)(t + 1) * n, H, PS, 1);
        for (int i = 0; i < n; ++i)            /* :946-950 */
            block_errors(&l, 0, -n, n * t + i, t == T - 1, t == 0, t >= Tmin);
    }

    /* :954-986 backward states, t = 0 .. T-1 */
    if (bidir) {
        for (int t = 0; t < T; ++t) {
            if (t != 0)
                for (int g = 0; g < 4; ++g)    /* :973-976 */
                    mm_nn(l.dir[1][B_TMPERR] + (size_t)t * n, w_internal(&l, w, g, 1), H, H,
                          l.dir[1][deltaBuf[g]] + (size_t)(t - 1) * n, H, PS, 1);
            for (int i = 0; i < n; ++i)        /* :980-984 */
                block_errors(&l, 1, +n, n * t + i, t == 0, t == T - 1, t >= Tmin);
        }
    }

    /* :990-1009 error to the preceding layer */
    if (prevErr) {
        int first = 1;
        for (int d = 0; d < l.dirs; ++d)
            for (int g = 0; g < 4; ++g) {
                mm_nn(prevErr, w_input(&l, w, g, d), P, H, l.dir[d][deltaBuf[g]], H, N, !first);
                first = 0;
            }
    }

    /* :1012-1044 weight updates */
    int nw = orc_lstm_weight_count(P, L, bidir);
    real_t *xq = NULL;
    if (g_opround) {
        xq = rounded_copy(x, (size_t)N * P); x = xq;
        for (int d = 0; d < l.dirs; ++d)
            for (int g = 0; g < 4; ++g) l.dq[d][g] = rounded_copy(l.dir[d][deltaBuf[g]], (size_t)N * H);
    }
    ORC_PAR((long)nw * N)
    for (int i = 0; i < nw; ++i)
        wu[i] = weight_update(&l, x, i);
    if (g_opround) {
        free(xq);
        for (int d = 0; d < l.dirs; ++d)
            for (int g = 0; g < 4; ++g) free(l.dq[d][g]);
    }
}

/* ------------------------------------------------------------------------- */
/* feed-forward layer                                                         */
/* ------------------------------------------------------------------------- */

int orc_ff_weight_count(int P, int L) { return L * (P + 1); }   /* TrainableLayer.cu:101 */

/* FeedForwardLayer<Cpu,Act>::computeForwardPass, FeedForwardLayer.cu:143-170 */
void orc_ff_forward(int act, int P, int L, real_t bias, int N,
                    const real_t *w, const real_t *x, real_t *y)
{
    mm_tn(y, w, P, L, x, P, N, 0);                               /* :148-152 */
    const real_t *biasWeights = w + (size_t)L * P;
    for (int outputIdx = 0; outputIdx < N * L; ++outputIdx) {    /* ComputeOutputFn :53-66 */
        real_t a = y[outputIdx];
        a += bias * biasWeights[outputIdx % L];
        y[outputIdx] = act_fn(act, a);
    }
}

/* FeedForwardLayer<Cpu,Act>::computeBackwardPass, FeedForwardLayer.cu:172-224.
 * outErr is updated in place to the deltas (:177-185). */
void orc_ff_backward(int act, int P, int L, real_t bias, int N,
                     const real_t *w, const real_t *x, const real_t *y,
                     real_t *outErr, real_t *prevErr, real_t *wu)
{
    for (int i = 0; i < N * L; ++i)                              /* ComputeDeltaFn :74-78 */
        outErr[i] = act_deriv(act, y[i]) * outErr[i];
    if (prevErr)
        mm_nn(prevErr, w, P, L, outErr, L, N, 0);                /* :190-197 */
    mm_nt(wu, x, P, N, outErr, L, N, 0);                         /* :202-206 */
    for (int j = 0; j < L; ++j) {                                /* ComputeBiasWeightUpdateFn :90-101 */
        const real_t *offDeltas = outErr + j;
        real_t s = 0;
        for (int i = 0; i < N; ++i) {
            s += bias * *offDeltas;
            offDeltas += L;
        }
        wu[(size_t)P * L + j] = s;
    }
}

/* ------------------------------------------------------------------------- */
/* softmax layer (FeedForward<Identity> + normalisation)                      */
/* ------------------------------------------------------------------------- */

/* SoftmaxLayer<Cpu,Identity>::computeForwardPass, SoftmaxLayer.cu:250-315 */
void orc_softmax_forward(int P, int L, real_t bias, int N, const char *patTypes,
                         const real_t *w, const real_t *x, real_t *y, real_t *patTmp)
{
    orc_ff_forward(ACT_IDENTITY, P, L, bias, N, w, x, y);        /* :253 */
    for (int patIdx = 0; patIdx < N; ++patIdx) {                 /* CalculateOffsetFn :54-77 */
        if (patTypes[patIdx] == PATTYPE_NONE) { patTmp[patIdx] = SKIP_MARKER; continue; }
        real_t max = NL_MIN, min = NL_MAX;
        const real_t *off = &y[(size_t)patIdx * L];
        for (int i = 0; i < L; ++i) {
            real_t v = off[i];
            min = (min < v ? min : v);                           /* helpers/min.cuh */
            max = (max > v ? max : v);                           /* helpers/max.cuh */
        }
        patTmp[patIdx] = (real_t)0.5 * (min + max);
    }
    for (int outputIdx = 0; outputIdx < N * L; ++outputIdx) {    /* CalculateExpFn :86-105 */
        real_t offset = patTmp[outputIdx / L];
        if (offset == SKIP_MARKER) continue;
        y[outputIdx] = safe_exp(y[outputIdx] - offset);
    }
    for (int patIdx = 0; patIdx < N; ++patIdx) {                 /* SumUpOutputsFn :114-132 */
        if (patTmp[patIdx] == SKIP_MARKER) continue;
        const real_t *off = &y[(size_t)patIdx * L];
        real_t sum = 0;
        for (int i = 0; i < L; ++i) sum += off[i];
        patTmp[patIdx] = sum;
    }
    for (int outputIdx = 0; outputIdx < N * L; ++outputIdx) {    /* NormalizeOutputsFn :141-159 */
        real_t normFact = patTmp[outputIdx / L];
        if (normFact == SKIP_MARKER) continue;
        y[outputIdx] = y[outputIdx] / normFact;
    }
}

/* SoftmaxLayer<Cpu,Identity>::computeBackwardPass, SoftmaxLayer.cu:317-353 */
void orc_softmax_backward(int P, int L, real_t bias, int N, const char *patTypes,
                          const real_t *w, const real_t *x, const real_t *y,
                          real_t *outErr, real_t *prevErr, real_t *wu, real_t *patTmp)
{
    for (int patIdx = 0; patIdx < N; ++patIdx) {                 /* CalculateErrorOffsetFn :171-188 */
        if (patTypes[patIdx] == PATTYPE_NONE) { patTmp[patIdx] = SKIP_MARKER; continue; }
        const real_t *o = &y[(size_t)patIdx * L], *e = &outErr[(size_t)patIdx * L];
        real_t offset = 0;
        for (int i = 0; i < L; ++i) offset += o[i] * e[i];
        patTmp[patIdx] = offset;
    }
    for (int outputIdx = 0; outputIdx < N * L; ++outputIdx) {    /* CalculateErrorsFn :197-218 */
        real_t offset = patTmp[outputIdx / L];
        if (offset == SKIP_MARKER) continue;
        outErr[outputIdx] = y[outputIdx] * (outErr[outputIdx] - offset);
    }
    orc_ff_backward(ACT_IDENTITY, P, L, bias, N, w, x, y, outErr, prevErr, wu);   /* :352 */
}

/* ------------------------------------------------------------------------- */
/* post output layers                                                         */
/* ------------------------------------------------------------------------- */

/* MulticlassClassificationLayer<Cpu>::calculateError, .cu:194-213 (CE fn :55-68).
 * thrust::transform_reduce on the host backend reduces sequentially in fp32. */
real_t orc_mcc_error(int L, int N, const int *targetClasses, const real_t *outputs)
{
    real_t error = 0;
    for (int patIdx = 0; patIdx < N; ++patIdx) {
        int targetClass = targetClasses[patIdx];
        if (targetClass == -1) continue;
        real_t p = outputs[(size_t)patIdx * L + targetClass];
        real_t targetProb = (NL_MIN > p ? NL_MIN : p);
        error += logf(targetProb);
    }
    return -error;
}

/* MulticlassClassificationLayer<Cpu>::countCorrectClassifications, .cu:159-177 (fn :77-105) */
int orc_mcc_correct(int L, int N, const int *targetClasses, const real_t *outputs)
{
    int correct = 0;
    for (int patIdx = 0; patIdx < N; ++patIdx) {
        int targetClass = targetClasses[patIdx];
        if (targetClass == -1) continue;
        const real_t *off = outputs + (size_t)patIdx * L;
        real_t maxProb = 0; int estClass = 0;
        for (int i = 0; i < L; ++i) {
            real_t out = off[i];
            if (out > maxProb) { maxProb = out; estClass = i; }
        }
        if (targetClass == estClass) ++correct;
    }
    return correct;
}

/* MulticlassClassificationLayer<Cpu>::computeBackwardPass, .cu:220-240 (fn :115-134) */
void orc_mcc_backward(int L, int N, const int *targetClasses, const real_t *outputs, real_t *outErr)
{
    memset(outErr, 0, sizeof(real_t) * (size_t)N * L);           /* :227 */
    for (int patIdx = 0; patIdx < N; ++patIdx) {
        int targetClass = targetClasses[patIdx];
        if (targetClass == -1) continue;
        size_t outputIdx = (size_t)patIdx * L + targetClass;
        real_t p = outputs[outputIdx];
        real_t targetProb = (NL_MIN > p ? NL_MIN : p);
        outErr[outputIdx] = -(1 / targetProb);
    }
}

/* SsePostOutputLayer<Cpu>::calculateError, .cu:114-132 (fn :45-60) */
real_t orc_sse_error(int L, int N, const char *patTypes, const real_t *targets, const real_t *outputs)
{
    real_t s = 0;
    for (int outputIdx = 0; outputIdx < N * L; ++outputIdx) {
        if (patTypes[outputIdx / L] == PATTYPE_NONE) continue;
        real_t diff = targets[outputIdx] - outputs[outputIdx];
        s += diff * diff;
    }
    return (real_t)0.5 * s;
}

/* SsePostOutputLayer<Cpu>::computeBackwardPass, .cu:139-155 (fn :69-87) */
void orc_sse_backward(int L, int N, const char *patTypes, const real_t *targets,
                      const real_t *outputs, real_t *outErr)
{
    for (int outputIdx = 0; outputIdx < N * L; ++outputIdx) {
        if (patTypes[outputIdx / L] == PATTYPE_NONE) outErr[outputIdx] = 0;
        else outErr[outputIdx] = outputs[outputIdx] - targets[outputIdx];
    }
}

/* post output layer kinds shared with the python wrapper */
enum { POST_SSE = 0, POST_WEIGHTEDSSE = 1, POST_SSE_MASK = 2, POST_CE = 3, POST_RMSE = 4, POST_BINARY = 5 };

/*
 * calculateError() of the remaining post output layers.  L = size of the OUTPUT layer; targets hold
 * L values per pattern, or 2L interleaved (target, weight|filter input) pairs for weightedsse / wf.
 *   weightedsse : WeightedSsePostOutputLayer.cu:40-64, :119-139   0.5 * sum ((y - t) * w)^2
 *   wf/sse_mask : SseMaskPostOutputLayer.cu:40-64, :119-139       0.5 * sum (y * f - t)^2
 *   ce          : CePostOutputLayer.cu:43-71, :125-143            sum t * log(max(min,t) / max(min,y))
 *   rmse        : RmsePostOutputLayer.cu:40-71, :125-152          sum_rows sqrt(sum_j (y-t)^2 / L)
 *   binary      : BinaryClassificationLayer.cu:44-67, :166-183    sum -log(t > 0 ? act : 1 - act)
 */
real_t orc_post_error(int kind, int L, int N, const char *patTypes, const real_t *targets, const real_t *outputs)
{
    real_t s = 0;
    if (kind == POST_RMSE) {
        for (int patIdx = 0; patIdx < N; ++patIdx) {
            if (patTypes[patIdx] == PATTYPE_NONE) continue;
            real_t sum = 0;
            for (int i = 0; i < L; ++i) {
                real_t diff = outputs[(size_t)patIdx * L + i] - targets[(size_t)patIdx * L + i];
                sum += diff * diff;
            }
            s += sqrtf(sum / L);
        }
        return s;
    }
    for (int index = 0; index < N * L; ++index) {
        if (patTypes[index / L] == PATTYPE_NONE) continue;
        real_t output = outputs[index];
        if (kind == POST_SSE) {
            real_t diff = targets[index] - output; s += diff * diff;
        } else if (kind == POST_WEIGHTEDSSE) {
            real_t diff = (output - targets[index * 2]) * targets[index * 2 + 1]; s += diff * diff;
        } else if (kind == POST_SSE_MASK) {
            real_t diff = output * targets[index * 2 + 1] - targets[index * 2]; s += diff * diff;
        } else if (kind == POST_CE) {
            real_t target = targets[index];
            real_t ftarget = (NL_MIN > target ? NL_MIN : target);
            real_t o = (NL_MIN > output ? NL_MIN : output);
            s += target * logf(ftarget / o);
        } else {   /* POST_BINARY, L == 1 */
            real_t act = (output > NL_MIN ? output : NL_MIN);
            real_t targetProb = (targets[index] > 0 ? act : 1 - act);
            s += -logf(targetProb);
        }
    }
    if (kind == POST_SSE || kind == POST_WEIGHTEDSSE || kind == POST_SSE_MASK) s = (real_t)0.5 * s;
    return s;
}

/* BinaryClassificationLayer<Cpu>::countCorrectClassifications, .cu:69-85, :132-154 */
int orc_binary_correct(int N, const char *patTypes, const real_t *targets, const real_t *outputs)
{
    int c = 0;
    for (int i = 0; i < N; ++i) {
        int tgtClass = targets[i] > (real_t)0.5, estClass = outputs[i] > (real_t)0.5;
        c += (patTypes[i] != PATTYPE_NONE) && (tgtClass == estClass);
    }
    return c;
}

/*
 * computeBackwardPass() of the same layers (error written into the output layer's outputErrors):
 *   weightedsse .cu:66-93, :146-167   wf .cu:66-93, :146-167   ce .cu:73-99, :150-170 (clipped to +-100)
 *   rmse .cu:73-97, :154-174 (rmse * (y - t))   binary .cu:87-111, :190-207
 *   binary .cu:90-113 (dummy slots are left untouched by the reference; 0 here)
 */
void orc_post_backward(int kind, int L, int N, const char *patTypes, const real_t *targets,
                       const real_t *outputs, real_t *outErr)
{
    for (int patIdx = 0; patIdx < N; ++patIdx) {
        int real = patTypes[patIdx] != PATTYPE_NONE;
        real_t rmse = 0;
        if (kind == POST_RMSE && real) {
            real_t sum = 0;
            for (int i = 0; i < L; ++i) {
                real_t diff = outputs[(size_t)patIdx * L + i] - targets[(size_t)patIdx * L + i];
                sum += diff * diff;
            }
            rmse = sqrtf(sum / L);
        }
        for (int i = 0; i < L; ++i) {
            size_t index = (size_t)patIdx * L + i;
            real_t y = outputs[index], e = 0;
            if (real) {
                if (kind == POST_SSE) e = y - targets[index];
                else if (kind == POST_WEIGHTEDSSE) e = (y - targets[index * 2]) * targets[index * 2 + 1];
                else if (kind == POST_SSE_MASK) e = (y * targets[index * 2 + 1] - targets[index * 2]) * targets[index * 2 + 1];
                else if (kind == POST_CE) {
                    real_t a = (NL_MIN > y ? NL_MIN : y);
                    real_t b = -targets[index] / a;
                    e = (b < -100 ? -100 : (b > 100 ? 100 : b));
                } else if (kind == POST_RMSE) e = rmse * (y - targets[index]);
                else {
                    real_t act = (y > NL_MIN ? y : NL_MIN);
                    real_t targetProb = (targets[index] > 0 ? act : 1 - act);
                    e = (targets[index] > 0 ? -(1 / targetProb) : (1 / targetProb));
                }
            }
            outErr[index] = e;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* optimizer step                                                             */
/* ------------------------------------------------------------------------- */

/* UpdateWeightFn, optimizers/SteepestDescentOptimizer.cu:48-58 */
void orc_sgd_update(int n, real_t learningRate, real_t momentum,
                    real_t *weights, const real_t *weightUpdates, real_t *weightDeltas)
{
    for (int i = 0; i < n; ++i) {
        real_t delta = momentum * weightDeltas[i] - learningRate * weightUpdates[i];
        weightDeltas[i] = delta;
        weights[i] = weights[i] + delta;
    }
}

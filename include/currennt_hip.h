/*
 * currennt_hip.h -- C ABI of libcurrennt_hip.so, the MI355X (gfx950) implementation of the
 * CURRENNT LSTM training hot path (naxingyu/lstm-rnn, currennt_lib/src).
 *
 * The reference has no FFI layer: its seam is the virtual layers::Layer<TDevice> interface
 * (layers/Layer.hpp:40-179, layers/TrainableLayer.hpp:84-155, layers/PostOutputLayer.hpp:80)
 * plus the GEMM seam helpers::Matrix / helpers::cublas::multiplyMatrices
 * (helpers/Matrix.hpp:48-49, helpers/cublas.hpp:30-38).  Every entry point below names the
 * reference method it stands in for; a maintainer binds them from a `Hip` device policy next to
 * `Cpu`/`Gpu` (Types.hpp:45-67) as shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: opaque handles, ints, floats, raw pointers; no C++ or torch types.
 *   - every function returns 0 on success or a negative cn_status; the message is available
 *     from cn_last_error() (the reference throws std::runtime_error(msg), e.g. Matrix.cu:209-210;
 *     host wrappers turn a non-zero status back into an exception -> "FAILED: msg", exit code 2,
 *     main.cpp:492-495).
 *   - host arrays use the reference layouts: activations [t*PS + ps][unit]
 *     (DataSet.cpp:359,376,383,406), flat weight vectors as in LstmLayer.hpp:36-55 and
 *     FeedForwardLayer.cu:148,160.  Padded / packed / bf16 device copies are internal.
 *   - one cn_ctx per GPU, used from one host thread at a time (the reference is single-threaded,
 *     cublas.cu:33-43).  All work is enqueued on the ctx stream; only the functions marked
 *     [sync] wait for the device.
 */
#ifndef CURRENNT_HIP_H
#define CURRENNT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cn_ctx   cn_ctx;
typedef struct cn_layer cn_layer;

typedef enum cn_status {
    CN_OK            =  0,
    CN_ERR_BAD_ARG   = -1,   /* null handle, negative size, unknown enum                     */
    CN_ERR_SHAPE     = -2,   /* size mismatch (reference: "Invalid matrix dimensions", ...)  */
    CN_ERR_HIP       = -3,   /* a HIP runtime call failed                                    */
    CN_ERR_STATE     = -4,   /* call order violated (e.g. forward before a fraction is loaded)*/
    CN_ERR_NO_DEVICE = -5,   /* no gfx950 device / code object cannot run here               */
    CN_ERR_COMM      = -6    /* RCCL could not be loaded / a collective call failed            */
} cn_status;

/* arithmetic mode of the GEMM operands (accumulation and all state are always fp32) */
typedef enum cn_precision {
    CN_PREC_F32    = 0, /* fp32 operands on v_mfma_f32_*_f32: exact-fp32 parity mode (real_t = float, Types.hpp:39); the fp32
                           MFMAs run at 1/16 of the bf16 rate                                                              */
    CN_PREC_BF16   = 1, /* bf16 operands on v_mfma_f32_*_bf16: throughput mode                                            */
    CN_PREC_BF16X3 = 2  /* fp32 operands in memory, every operand split into bf16 hi + bf16 lo inside the kernels and a product
                           computed as three bf16 MFMAs (hi*hi + lo*hi + hi*lo, fp32 accumulation): ~2^-16 relative per term,
                           i.e. the fp32 tolerance of BASELINE.json (posterior max-abs < 1e-4) at a third of the bf16 MFMA rate --
                           the parity mode that is fast; state, activations and gradients sums are fp32 as in the other modes      */
} cn_precision;

/* layer kinds = the type strings of LayerFactory.cu:52-87 that are on the hot path */
typedef enum cn_layer_kind {
    CN_LAYER_INPUT = 0,                   /* "input"                      InputLayer.cpp            */
    CN_LAYER_LSTM,                        /* "lstm"                       LstmLayer.cu              */
    CN_LAYER_BLSTM,                       /* "blstm"                      LstmLayer.cu              */
    CN_LAYER_FF_TANH,                     /* "feedforward_tanh"           FeedForwardLayer.cu       */
    CN_LAYER_FF_LOGISTIC,                 /* "feedforward_logistic"                                 */
    CN_LAYER_FF_IDENTITY,                 /* "feedforward_identity"                                 */
    CN_LAYER_SOFTMAX,                     /* "softmax"                    SoftmaxLayer.cu           */
    CN_LAYER_SSE,                         /* "sse"                        SsePostOutputLayer.cu     */
    CN_LAYER_MULTICLASS_CLASSIFICATION,   /* "multiclass_classification"  MulticlassClassificationLayer.cu */
    /* remaining post output layers of LayerFactory.cu:52-87.  The weighted kinds have size == 2 * size of
     * the output layer and take interleaved (target, weight | filter input) pairs in cn_fraction.targets;
     * binary_classification has size 1 and reads cn_fraction.target_classes. */
    CN_LAYER_WEIGHTEDSSE,                 /* "weightedsse"                WeightedSsePostOutputLayer.cu    */
    CN_LAYER_SSE_MASK,                    /* "wf"                         SseMaskPostOutputLayer.cu        */
    CN_LAYER_CE,                          /* "ce"                         CePostOutputLayer.cu             */
    CN_LAYER_RMSE,                        /* "rmse"                       RmsePostOutputLayer.cu           */
    CN_LAYER_BINARY_CLASSIFICATION        /* "binary_classification"      BinaryClassificationLayer.cu     */
} cn_layer_kind;

/* which device vector cn_layer_read / cn_layer_device_ptr addresses */
typedef enum cn_buffer {
    CN_BUF_OUTPUTS = 0,        /* Layer::outputs()               [N][size]  Layer.hpp:132        */
    CN_BUF_OUTPUT_ERRORS,      /* Layer::outputErrors()          [N][size]  Layer.hpp:153        */
    CN_BUF_WEIGHTS,            /* TrainableLayer::weights()      flat       TrainableLayer.hpp:119*/
    CN_BUF_WEIGHT_UPDATES,     /* TrainableLayer::weightUpdates() flat      TrainableLayer.hpp:133*/
    CN_BUF_WEIGHT_DELTAS,      /* SteepestDescentOptimizer::m_weightDeltas  SteepestDescentOptimizer.cu:117-123 */
    /* LSTM per-direction internals, [N][H] each (LstmLayer.hpp:88-100,170-233); `dir` selects fw/bw */
    CN_BUF_LSTM_CELL_STATES,
    CN_BUF_LSTM_NI_ACTS,
    CN_BUF_LSTM_IG_ACTS,
    CN_BUF_LSTM_FG_ACTS,
    CN_BUF_LSTM_OG_ACTS,
    CN_BUF_LSTM_NI_DELTAS,
    CN_BUF_LSTM_IG_DELTAS,
    CN_BUF_LSTM_FG_DELTAS,
    CN_BUF_LSTM_OG_DELTAS,
    CN_BUF_LSTM_TMP_OUTPUTS    /* per-direction block outputs y */
} cn_buffer;

/* one parallel-sequence mini batch = data_sets::DataSetFraction (DataSetFraction.hpp:50-60) */
typedef struct cn_fraction {
    int          max_seq_length;    /* T     = fraction.maxSeqLength()                               */
    int          min_seq_length;    /* Tmin  = fraction.minSeqLength()                               */
    int          num_sequences;     /* fraction.numSequences() (<= parallel_sequences)               */
    int          input_pattern_size;/* fraction.inputPatternSize()                                   */
    int          output_pattern_size;/* fraction.outputPatternSize()                                  */
    const char  *pat_types;         /* [T*PS]            0 = PATTYPE_NONE  (Types.hpp:30-33)         */
    const float *inputs;            /* [T*PS][inputSize]                                              */
    const int   *target_classes;    /* [T*PS] or NULL    -1 on dummy slots (DataSet.cpp:336)         */
    const float *targets;           /* [T*PS][outputSize] or NULL                                     */
} cn_fraction;

/* ---- context ------------------------------------------------------------------------------- */

/* Replaces the device selection of main.cpp:526-541 and the lazy cuBLAS handle of cublas.cu:33-43.
 * `stream` is a hipStream_t the caller owns (e.g. torch.cuda.current_stream().cuda_stream) or NULL
 * for a stream owned by the context. */
int  cn_ctx_create(int device_id, cn_precision precision, void *stream, cn_ctx **out);
int  cn_ctx_destroy(cn_ctx *ctx);
int  cn_ctx_synchronize(cn_ctx *ctx);                                   /* [sync] */
/* Named integer options of a context (no counterpart in the reference, whose Cpu path has nothing to choose):
 *   "deterministic"  1: every sum over the patterns of a fraction -- weight gradients (ComputeWeightUpdateFn,
 *                    LstmLayer.cu:289-512; FeedForwardLayer.cu:82-102,200-207), bias / peephole / column sums, the error --
 *                    is formed in a fixed order (partials stored by their producers, added by one thread per output), so
 *                    two runs on the same inputs give BIT-IDENTICAL gradients and weights, like the reference's serial
 *                    sums.  0: fp32 atomics in arrival order.  Default: 1 for CN_PREC_F32 / CN_PREC_BF16X3 (the parity
 *                    modes), 0 for CN_PREC_BF16; the environment variable CN_DETERMINISTIC=0/1 sets the default.
 *   "overlap"        1 (default): weight-gradient products on side streams beside the recurrent kernels; 0: one stream.
 *   A/B and test switches of the kernel selection (csrc/cn_internal.h, CN_OPTION_LIST; e.g. "no_s2_asm", "no_cluster",
 *                    "no_big_tn", "tnbig_group_mink", "lazy_softmax"): every one of them was an environment variable read
 *                    somewhere on a launch path until round 5.  Now the environment (CN_<NAME>) is read ONCE, at
 *                    cn_ctx_create, into the context's options block; this call changes an entry afterwards; no launch path
 *                    calls getenv.  Options that size allocations ("cluster4", "no_cluster") belong before the layers.
 * Unknown names fail with CN_ERR_BAD_ARG.  May be changed between fractions.                                       */
int  cn_ctx_set_option(cn_ctx *ctx, const char *name, int value);
int  cn_ctx_get_option(const cn_ctx *ctx, const char *name, int *value);
/* The weight-gradient GEMMs of a backward pass run on an internal side stream beside the next layer's
 * recurrent kernel.  Every entry point of this library orders itself behind them; call cn_ctx_join before
 * ANOTHER library (e.g. RCCL through torch.distributed) reads weightUpdates on the ctx stream: it makes the
 * ctx stream wait (device-side, no host sync) for the side stream. */
int  cn_ctx_join(cn_ctx *ctx);
/* The same for the weightUpdates of ONE layer: the context's stream waits for that layer's gradient work only.
 * Lets a data-parallel caller all-reduce the gradient of layer k+1 while layer k is still in its backward
 * pass (SURVEY.md 8e "Overlap": bucket = layer).  No-op when nothing is pending for the layer.   [async] */
int  cn_layer_join(cn_layer *layer);
/* The HIP stream (hipStream_t) the context enqueues its work on: the one given to cn_ctx_create, or the
 * library's own.  For callers that order foreign work (a collective library's stream) against it. */
void *cn_ctx_stream(cn_ctx *ctx);
/* The same ordering for a stream of the CALLER's choice (a HIP stream handle): `stream` waits for the gradient work
 * of `layer` and for nothing else the context has enqueued since, so a collective issued on it can start while the
 * context's stream is still busy with the backward pass of the layers below.  The caller orders the context's
 * stream behind that collective before the next cn_sgd_update*.                                      [async] */
int  cn_layer_join_stream(cn_layer *layer, void *stream);
/* message of the last failed call on this thread (ctx may be NULL for creation failures) */
const char *cn_last_error(cn_ctx *ctx);
/* "gfx950" etc. of the bound device; version string of the library */
const char *cn_device_arch(cn_ctx *ctx);
/* Device enumeration for `--list_devices` (main.cpp:509-525: cudaGetDeviceCount / cudaGetDeviceProperties).
 * cn_device_count returns the number of HIP devices (0 when there is none); cn_device_name copies the name of
 * device `index` ("<marketing name> (<gfx arch>)") into `buf` and returns a status. */
int  cn_device_count(void);
int  cn_device_name(int index, char *buf, int buf_size);
const char *cn_version(void);

/* ---- layers (LayerFactory<TDevice>::createLayer, LayerFactory.hpp:49-56) -------------------- */

/* `preceding` is NULL only for CN_LAYER_INPUT.  parallel_sequences / max_seq_length size every
 * buffer once, as Layer.cpp:59-66 and LstmLayer.cu:553-574 do; they are taken from `preceding`
 * for all other layers (pass 0).  `bias` is the JSON "bias" value (TrainableLayer.cu:57). */
int  cn_layer_create(cn_ctx *ctx, cn_layer_kind kind, cn_layer *preceding, int size, float bias,
                     int parallel_sequences, int max_seq_length, cn_layer **out);
int  cn_layer_destroy(cn_layer *layer);

int  cn_layer_size(const cn_layer *layer);
int  cn_layer_kind_of(const cn_layer *layer);
/* number of weights: size*(inputWeightsPerBlock*(P+1)+internalWeightsPerBlock), TrainableLayer.cu:101 */
int  cn_layer_weight_count(const cn_layer *layer);

/* Layer::loadSequences(fraction) for every layer of the stack: uploads patTypes once (the
 * reference copies them per layer, Layer.cpp:140), inputs (InputLayer.cpp:49-60) and targets /
 * target classes (PostOutputLayer.cpp:67-79, MulticlassClassificationLayer.cu:186-192).
 * Error texts follow the reference ("Input layer size of X != data input pattern size of Y").
 * The host buffers of `fraction` may be reused as soon as the call returns (they are copied into pinned staging
 * memory; the upload itself and the re-layout kernel are asynchronous).                                [async] */
int  cn_fraction_load(cn_ctx *ctx, cn_layer *input, cn_layer *post_output, const cn_fraction *fraction);

/* The same with every pointer of `fraction` addressing DEVICE memory of the context's GPU (inputs already
 * resident in HBM: a prefetching data loader uploads fraction k+1 while fraction k trains).  Copies are
 * device-to-device on the ctx stream; nothing synchronises. */
int  cn_fraction_load_resident(cn_ctx *ctx, cn_layer *input, cn_layer *post_output, const cn_fraction *fraction);

/* Announces the fraction the NEXT cn_fraction_load_resident will load (same descriptor, same layers).  The library
 * re-lays it out into a second set of buffers on the side stream of the next gradient work -- beside the backward
 * pass of the current fraction -- and that load then only exchanges the buffers (no kernel, nothing on the critical
 * path).  Purely a hint: a load of any other fraction, a host-buffer load, or a backward pass that put nothing on
 * the side stream simply discards it and loads the ordinary way.  The device buffers of `fraction` must stay
 * unchanged until that load.  CN_ERR_STATE
 * when a prefetch that is already in flight has not been consumed.  The reference has no counterpart (its loader
 * thread prefetches HOST fractions, DataSet.cpp:546-552; Layer::loadSequences copies synchronously).   [async] */
int  cn_fraction_prefetch_resident(cn_ctx *ctx, cn_layer *input, cn_layer *post_output, const cn_fraction *fraction);
/* The same for HOST buffers: announces the fraction the NEXT cn_fraction_load will load.  The library packs it into pinned
 * staging memory and starts its upload at once (copy stream), re-lays it out beside the coming backward pass, and that load
 * only exchanges buffers -- what the reference's loader thread does for host fractions one fraction ahead (DataSet.cpp:202-
 * 240,546-552,589), moved across PCIe as well.  The caller's buffers are copied before this returns.  Purely a hint, as above.
 * CN_ERR_STATE when a prefetch that is already in flight has not been consumed.   [async] */
int  cn_fraction_prefetch(cn_ctx *ctx, cn_layer *input, cn_layer *post_output, const cn_fraction *fraction);

/* Layer::computeForwardPass / computeBackwardPass (Layer.hpp:165-170).  Backward of a trainable
 * layer consumes its outputErrors, writes the preceding trainable layer's outputErrors
 * (LstmLayer.cu:990-1009, FeedForwardLayer.cu:188-198) and its own weightUpdates. */
int  cn_layer_forward(cn_layer *layer);
int  cn_layer_backward(cn_layer *layer);

/* PostOutputLayer::calculateError() and, for multiclass_classification,
 * countCorrectClassifications() (MulticlassClassificationLayer.cu:159-177,194-213).
 * `correct` may be NULL; it is set to -1 for layers without a class count.        [sync] */
int  cn_loss_eval(cn_layer *post_output, float *error, int *correct);

/* Asynchronous form for the training loop: add this fraction's error / #correct to device-side
 * accumulators (Optimizer.cu:46-55 sums them per epoch on the host, two blocking D2H copies per
 * fraction) and read the sums once with cn_loss_read ([sync]; `reset` != 0 clears them). */
int  cn_loss_accumulate(cn_layer *post_output);
int  cn_loss_read(cn_ctx *ctx, float *error_sum, int64_t *correct_sum, int reset);

/* ---- weights (TrainableLayer.cu:65-101,211-248) --------------------------------------------- */

int  cn_layer_set_weights(cn_layer *layer, const float *host, int count);       /* flat reference layout */
/* copy one reference-layout vector to the host; `dir` = 0 fw / 1 bw for LSTM internals.   [sync] */
int  cn_layer_read(cn_layer *layer, cn_buffer which, int dir, float *host, size_t count);
/* overwrite a flat parameter vector (WEIGHTS / WEIGHT_UPDATES / WEIGHT_DELTAS) from the host: batch-mode
 * accumulation (Optimizer.cu:72-85) and optimizer-state restore (SteepestDescentOptimizer.cu:125-131) */
int  cn_layer_upload(cn_layer *layer, cn_buffer which, const float *host, size_t count);
/* overwrite Layer::outputErrors() from the host (tests of a single layer's backward pass) */
int  cn_layer_write_output_errors(cn_layer *layer, const float *host, size_t count);
/* raw fp32 device pointer of a flat parameter vector (WEIGHTS / WEIGHT_UPDATES / WEIGHT_DELTAS),
 * valid until the layer is destroyed; SteepestDescentOptimizer.cu:83-85 reads the same pointers */
void *cn_layer_device_ptr(cn_layer *layer, cn_buffer which);

/* All trainable layers of a context share one parameter arena [weights | weightUpdates | deltas];
 * the weightUpdates part is what the data-parallel all-reduce sums.  The arena is
 * laid out at the first call of this function / the first forward pass; creating a trainable layer
 * afterwards fails with CN_ERR_STATE. */
int  cn_ctx_param_arena(cn_ctx *ctx, void **weights, void **weight_updates, void **weight_deltas,
                        size_t *count);

/* Tell the library that the flat weights were modified through a raw device pointer (e.g. by an
 * external optimizer); the packed copies are rebuilt before the next forward pass. */
int  cn_ctx_weights_touched(cn_ctx *ctx);

/* UpdateWeightFn (SteepestDescentOptimizer.cu:39-59): delta = momentum*delta - lr*update;
 * w += delta; then refresh the packed device copies the kernels read. */
int  cn_sgd_update(cn_layer *layer, float learning_rate, float momentum);
/* per-layer learning rate (the JSON "learningRate" of TrainableLayer.cu:58; negative = none) that cn_sgd_update_all
 * uses for this layer instead of its argument, as SteepestDescentOptimizer.cu:78-80 does */
int  cn_layer_set_learning_rate(cn_layer *layer, float learning_rate);
/* the same update for every trainable layer of the context in one launch; layers with a learning rate of their own
 * (cn_layer_set_learning_rate) are updated with it */
int  cn_sgd_update_all(cn_ctx *ctx, float learning_rate, float momentum);
/* Arms the update of the COMING backward pass (the per-fraction protocol of Optimizer.cu:86-94, hybrid online/batch learning):
 * each layer applies UpdateWeightFn with these values -- or its own learning rate -- as soon as its own gradient is complete,
 * on the stream that computed it (behind its all-reduce when a communicator is bound), instead of for all layers behind the
 * last backward kernel.  The result is the one of cn_layer_backward on every layer followed by cn_sgd_update_all: a layer's
 * weights are last read by its own backward pass, so the layers below never see the difference; only a cn_layer_read of
 * CN_BUF_WEIGHTS between the two calls would.  The following cn_sgd_update_all(ctx, same values) -- or cn_sgd_update per
 * layer -- completes the step (waits, handles layers no backward call reached) and disarms; it is an error to start another
 * backward pass before it.  Not for batch learning (the epoch sum is formed first) nor with weight noise.        [async] */
int  cn_ctx_arm_update(cn_ctx *ctx, float learning_rate, float momentum);

/* Batch learning (`--stochastic false`): Optimizer.cu:72-85 sums the fractions' weightUpdates of an epoch on the DEVICE
 * (thrust::copy for the first fraction, thrust::transform(plus) for the others) and updates once per epoch (:95-97).
 * cn_ctx_accumulate_updates adds the weightUpdates of ALL layers, as the backward pass of the current fraction left them, to
 * the context's epoch accumulator in one launch (first != 0: copies instead, the reference's `firstFraction` branch);
 * cn_ctx_take_accumulated makes the epoch sum the weightUpdates of every layer again -- what SteepestDescentOptimizer.cu:83-85
 * reads through `_curWeightUpdates()` -- in front of cn_allreduce_grads(ctx, NULL, 0) / cn_sgd_update*.  Nothing leaves the
 * device and nothing synchronises.                                                                               [async] */
int  cn_ctx_accumulate_updates(cn_ctx *ctx, int first);
int  cn_ctx_take_accumulated(cn_ctx *ctx);

/* ---- data-parallel training over the GPUs of one node (no counterpart in the reference, which
 *      drives a single device: main.cpp:526-541) ------------------------------------------------------------
 * One process (or thread) per GPU, one cn_ctx each.  The parallel sequences of a fraction are independent, so
 * ranks take disjoint sequences and meet only in the SUM over patterns of the weight gradients
 * (LstmLayer.cu:502-510, FeedForwardLayer.cu:94-100,206) and in the scalar error / #correct: one all-reduce(SUM)
 * of weightUpdates per layer on RCCL over xGMI, then the identical UpdateWeightFn on every rank keeps the replicas
 * bit-identical without a broadcast.  librccl is opened at run time (dlopen "librccl.so.1"; a process that already
 * holds one, e.g. through PyTorch, shares it); the library has no link-time dependency on it. */
/* CN_COMM_BACKEND=p2p in the environment selects the library's NATIVE exchange in place of RCCL, behind the same entry points:
 * a shared-memory rendezvous named by the id, one region per rank (flag words + two staging halves) mapped into every peer
 * through hipIpc, and cn_allreduce_grads = ONE stream-ordered kernel per bucket that stages, signals, sums in rank order and
 * acknowledges through the peers' regions -- one shot for small buckets, reduce-scatter + all-gather over the full xGMI mesh
 * for large ones (csrc/cn_comm_p2p.hip, csrc/cn_comm_ipc.cpp).  No host barrier after the first bucket; a peer that does not
 * answer within CN_COMM_IPC_TIMEOUT seconds (default 120) fails the communicator and the next cn_loss_read_global raises
 * CN_ERR_COMM.  One node, at most 8 ranks.  RCCL stays the default. */
/* (Tests on a box with fewer GPUs than ranks: with CN_COMM_BACKEND=ipc in the environment the four entry points below keep their
 * contract on a host-blocking TEST backend for ranks that share a device -- a shared-memory rendezvous named by the id, hipIpc
 * staging buffers, sums in rank order (csrc/cn_comm_ipc.cpp).  Never a measurement.  The p2p backend runs with shared devices
 * too.) */
#define CN_COMM_ID_BYTES 128
/* rank 0: a fresh rendezvous id (ncclGetUniqueId); hand its 128 bytes to every rank out of band (pipe, file, MPI, ...) */
int  cn_comm_unique_id(char *id /* [CN_COMM_ID_BYTES] */);
/* collective over all `world` ranks: binds a communicator for ctx's device to the context (ncclCommInitRank) */
int  cn_comm_init(cn_ctx *ctx, const char *id /* [CN_COMM_ID_BYTES] */, int rank, int world);
int  cn_comm_destroy(cn_ctx *ctx);
/* rank / world size of the bound communicator; world = 0 when there is none.  Either pointer may be NULL. */
int  cn_comm_info(const cn_ctx *ctx, int *rank, int *world);
/* which exchange the bound communicator runs: "rccl", "p2p" or "ipc" ("" when there is none); *exchanges (may be NULL): the
 * number of all-reduces cn_allreduce_grads has enqueued on it so far */
const char *cn_comm_backend(const cn_ctx *ctx, int64_t *exchanges);
/* all-reduce(SUM, fp32, in place) of the weightUpdates of `n` layers, in the order given, on the context's
 * communication stream: that stream waits for each layer's gradient work only (not for what the context's stream has
 * enqueued since: the backward pass of the layers below keeps running beside the exchange), and the context's stream
 * waits for the reductions in front of the next cn_sgd_update* / cn_layer_read.  Call it right after
 * cn_layer_backward(layer) for "bucket = layer" overlap.  n == 0 (layers may be NULL): the whole weightUpdates arena
 * of the context in ONE all-reduce, after all gradient work.  Collective: every rank makes the same calls in the same
 * order.                                                                                              [async] */
int  cn_allreduce_grads(cn_ctx *ctx, cn_layer *const *layers, int n);
/* cn_loss_read over all ranks: all-reduce(SUM) of the device-side error / #correct sums, then read.   [sync] */
int  cn_loss_read_global(cn_ctx *ctx, float *error_sum, int64_t *correct_sum, int reset);

/* ---- measurement ---------------------------------------------------------------------------- */

/* Wrap hipEvents around the kernels of one class on the ctx stream and accumulate their device
 * time; used by bench.py for the live roofline figure.  kernel_class: 0 = recurrent forward,
 * 1 = recurrent backward, 2 = N-wide gate GEMMs, 3 = weight-gradient GEMMs, 4 = everything else,
 * 5 = the gradient all-reduces of cn_allreduce_grads (events on the communication stream). */
int  cn_ctx_timing_enable(cn_ctx *ctx, int enable);
int  cn_ctx_timing_read(cn_ctx *ctx, int kernel_class, double *total_ms, int64_t *launches); /* [sync] */
int  cn_ctx_timing_reset(cn_ctx *ctx);
/* name of the recurrent kernel the layer's last forward (backward != 0: backward) pass launched, recorded by the launcher
 * that instantiated it, e.g. "lstm_bwd_s2_kernel<0,128>", "lstm_fwd_kernel<2,128,1,1>" or
 * "lstm_bwd_cluster_kernel<0,256,128,1>"; "" for other layers and before the first pass */
const char *cn_layer_recurrent_kernel(cn_layer *layer, int backward);

#ifdef __cplusplus
}
#endif
#endif /* CURRENNT_HIP_H */

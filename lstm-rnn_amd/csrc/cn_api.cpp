// C ABI of libcurrennt_hip.so (include/currennt_hip.h): contexts, layer objects, buffer ownership
// and the per-layer kernel sequences.  Host-only code; all device work is in the .hip files.
//
// Layer call sequences follow the reference methods they replace:
//   LSTM   forward  LstmLayer.cu:763-886    backward LstmLayer.cu:888-1051
//   FF     forward  FeedForwardLayer.cu:143-170   backward :172-224
//   softmax forward SoftmaxLayer.cu:250-315 backward :317-353
//   post output     MulticlassClassificationLayer.cu:159-240, SsePostOutputLayer.cu:114-155
#include "cn_internal.h"
#include "../../include/currennt_hip_debug.h"

#include <dlfcn.h>
#include <rccl/rccl.h>       // types and prototypes only: librccl is opened with dlopen (struct Rccl below), never linked

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

using namespace cn;

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_last_error;

struct cn_error : std::runtime_error {
    int code;
    cn_error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

void hip_check(hipError_t e, const char *what)
{
    if (e != hipSuccess)
        throw cn_error(CN_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_CHECK(x) hip_check((x), #x)

template <typename F> int guarded(F &&f)
{
    try { f(); return CN_OK; }
    catch (const cn_error &e) { g_last_error = e.what(); return e.code; }
    catch (const std::exception &e) { g_last_error = e.what(); return CN_ERR_HIP; }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// objects
// ---------------------------------------------------------------------------------------------
struct cn_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // side stream: weight-gradient GEMMs run beside the next layer's (latency-bound, 26-CU) recurrent kernel
    hipStream_t side = nullptr;
    // CU-masked twin of the side stream: gradient GEMMs that run beside a recurrent kernel have its whole duration
    // to finish, and at full speed their memory traffic stalls the latency-bound recurrent kernel (291 vs 229 us
    // per backward launch); on a subset of the CUs they run longer but draw less bandwidth
    hipStream_t side_slow = nullptr; int side_cus = 0;    // CU-masked side stream and its CU count
    int tn_cus = 0;                       // CUs the gradient products of the current on_side() call may fill (0 = the chip)
    hipEvent_t ev_sgd = nullptr, ev_ext = nullptr;
    hipEvent_t ev_pack_last = nullptr;         // = ev_pack of the last layer whose operand copies cn_sgd_update_all rebuilt (not owned)
    std::vector<hipEvent_t> pending_joins;     // side-stream work the main stream has not waited for yet ...
    std::vector<hipStream_t> pending_join_streams;   // ... and the stream each event was recorded on (same index)
    bool overlap = true;
    bool attach_forks = true;                  // CN_NO_ATTACHED_FORKS=1: fork / join / update events recorded with hipEventRecord
    bool f32 = true;                           // operands are fp32 in memory (CN_PREC_F32 and CN_PREC_BF16X3)
    // option "deterministic": every sum over the patterns of a fraction (weight gradients, bias / peephole / column sums, the
    // error) is formed in a fixed order -- partials stored by their producers, added by one thread per output -- instead of
    // with fp32 atomics in arrival order: two runs give bit-identical weights, as the reference's Cpu path does (one logical
    // thread per weight, serial sum: LstmLayer.cu:289-512, FeedForwardLayer.cu:82-102).  On by default in the parity modes
    // (CN_PREC_F32, CN_PREC_BF16X3), opt-in for CN_PREC_BF16; CN_DETERMINISTIC=0/1 sets the default of new contexts.
    bool det = false;
    Options opt;                               // A/B and test switches (cn_internal.h): the environment's at creation, cn_ctx_set_option afterwards
    int prec = P_F32;                          // arithmetic of the MFMA products: P_F32 / P_BF16 / P_X3 (cn_internal.h)
    int num_cus = 256;                         // hipDeviceProp_t::multiProcessorCount of the bound device
    std::string arch;
    std::vector<cn_layer *> layers;

    // current fraction (Layer::loadSequences, Layer.cpp:134-141)
    // PS: parallel sequences as the caller sees them; PSp >= PS: sequence slots per time step on the
    // device (padded to the recurrent kernels' sequence-group size; pad slots are permanent dummies).
    // N = T*PSp frames on the device, Next = T*PS frames in host layouts.
    int PS = 0, PSp = 0, rpl = 1, maxT = 0, T = 0, Tmin = 0, N = 0, Next = 0, numSeqs = 0;
    bool loaded = false;
    char *d_pat = nullptr, *d_pat_raw = nullptr;      // [maxN] pattern types inside an allocation with guard steps
    // ... and behind them the fraction's row map (GemmNT::rowmap; launch_rowmap): one allocation, so that the map travels with the
    // pattern types when a prefetched fraction's buffers are swapped in
    size_t pat_maxN = 0, pat_guard = 0;
    static size_t pat_bytes(size_t maxN, size_t guard) { return ((maxN + 2 * guard + 15) & ~(size_t)15) + (4 + 2 * maxN) * sizeof(int); }
    int *rowmap_of(char *pat_raw) const { return pat_raw ? (int *)(pat_raw + ((pat_maxN + 2 * pat_guard + 15) & ~(size_t)15)) : nullptr; }
    int est_real = 0;                                 // estimate of the current fraction's real frames (dispatch only)
    int *d_tcls = nullptr;
    float *d_loss = nullptr;      // [2] per-call (error, #correct as int bits)
    float *d_loss_acc = nullptr;  // [2] running sums for cn_loss_accumulate
    unsigned long long *d_xch = nullptr; size_t xch_bytes = 0;   // cluster kernels' exchange granules
    unsigned xch_epoch = 0;       // granule tags handed out so far (LstmRec::xch_epoch)
    int *d_fault = nullptr;       // device fault word (bounded spins)
    float *d_colpart = nullptr;   // softmax_mcc_bwd_colpart_floats(): replicas of the output layer's column sums (zero between launches)
    float *d_rowstat = nullptr;   // [maxN][2] per-pattern {log p_target, correct} of the last softmax forward pass
    cn_layer *rowstat_of = nullptr;
    bool loss_deferred = false;   // cn_loss_accumulate of the current fraction has not been enqueued yet: it rides on the output
                                  // layer's backward launch (softmax_mcc_bwd_kernel) or is flushed by whoever needs the sums / the rows

    // host fractions (cn_fraction_load): packed into pinned memory, uploaded on a copy stream into one of two
    // device staging areas while the previous fraction still computes, re-laid out by fraction_load_kernel
    hipStream_t copy = nullptr;
    char *h_stage[2] = {nullptr, nullptr}, *d_stage[2] = {nullptr, nullptr};
    size_t stage_bytes = 0;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
    bool stage_used[2] = {false, false};
    unsigned upload_idx = 0;

    // cn_fraction_prefetch_resident: the announced fraction is re-laid out into the alternate buffers on the side stream of
    // the next gradient work (beside the backward pass); the load that follows swaps the buffers instead of running the kernel
    struct Prefetch {
        bool valid = false, launched = false;
        cn_fraction f{}; cn_layer *input = nullptr, *post = nullptr;
        int hits = 0;                                            // loads that found their fraction re-laid out (cn_dbg_prefetch_hits)
        bool host = false; cn_fraction f_host{}; int slot = 0;     // cn_fraction_prefetch: f points into staging area `slot`, f_host is what the caller announced
        char *pat = nullptr, *pat_raw = nullptr; int *tcls = nullptr; void *in_op = nullptr; float *targets = nullptr;   // alternates
        bool allocated = false;
    } pf;

    // data-parallel training: RCCL communicator of this rank, its stream and the newest reduction's event
    ncclComm_t comm = nullptr;
    IpcComm *ipc = nullptr;                    // CN_COMM_BACKEND=ipc: the test backend (cn_comm_ipc.cpp) in place of RCCL
    int comm_rank = 0, comm_world = 0;
    bool has_comm() const { return comm != nullptr || ipc != nullptr; }
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_comm = nullptr, ev_comm_fork = nullptr;
    bool comm_pending = false;
    int64_t comm_exchanges = 0;                // all-reduces enqueued by cn_allreduce_grads (cn_comm_backend)

    // cn_ctx_arm_update: the momentum-SGD step of the coming backward pass is applied layer by layer, as soon as a layer's own
    // gradient is complete (on the stream that computed it), instead of for all layers behind the last backward kernel
    bool armed = false;
    float arm_lr = 0.f, arm_mom = 0.f;

    // parameter arena [weights | weightUpdates | weightDeltas]
    bool finalized = false;
    float *arena = nullptr;
    float *acc = nullptr;                 // epoch sum of weightUpdates (batch learning, cn_ctx_accumulate_updates); [total]
    bool acc_valid = false;
    size_t total = 0;

    // timing
    bool timing = false;
    int rpl_override = 0;
    struct Span { hipEvent_t a, b; };
    std::vector<Span> spans[KC_COUNT];
    std::vector<hipEvent_t> free_events;
    double acc_ms[KC_COUNT] = {};
    long acc_n[KC_COUNT] = {};

    size_t esz() const { return f32 ? 4 : 2; }
};

struct cn_layer {
    cn_ctx *ctx = nullptr;
    cn_layer_kind kind = CN_LAYER_INPUT;
    cn_layer *prev = nullptr;
    int size = 0;
    float bias = 0.f;
    int PS = 0, PSp = 0, maxT = 0;
    bool trainable = false, post = false, lstm = false, has_follower = false;
    bool mcc_pending = false;             // softmax layer: multiclass error injection deferred into the fused backward kernel
    // wide softmax layer (softmax_fwd_can_be_lazy): the forward pass may leave the LOGITS in out_f32 and {offset, sum} per row in
    // sm_stat; the fused backward kernel recomputes the posteriors from them, anybody else gets them through posteriors()
    float *sm_stat = nullptr;
    bool sm_lazy = false;                 // out_f32 holds logits right now
    bool sm_lazy_next = true;             // the last forward pass's posteriors were only consumed by the fused backward kernel
    bool sm_read = false;                 // somebody else looked at them since the forward pass

    int dirs = 1, H = 0, Hp = 0;          // lstm geometry
    int Lp = 0;                           // padded output width (row stride of out/err)
    int P = 0, Pp = 0;                    // preceding layer size / padded width

    // activations
    void *out_op = nullptr;               // [maxN][Lp] operand copy of the outputs
    float *out_f32 = nullptr;             // [maxN][Lp] fp32 outputs (ff/softmax; aliases out_op in f32 mode)
    float *err = nullptr;                 // [maxN][Lp] outputErrors
    float *stage_in = nullptr;            // input layer: [maxN][size] fp32 as loaded
    float *targets = nullptr;             // sse: [maxN][size]

    // lstm internals
    float *acts = nullptr, *cell = nullptr, *th = nullptr;
    void *pre16 = nullptr;                // bf16 mode: the input projection's pre-activations as bf16 (LstmRec::pre16), where the forward kernel takes them
    bool err_in_delta = false;            // ff/softmax, bf16 mode: outputErrors of the last backward pass exist only as the bf16 operand copy
    void *delta_op = nullptr;

    // parameters (flat reference layout, inside the ctx arena)
    size_t woff = 0;
    int nw = 0;
    float *w = nullptr, *wu = nullptr, *wd = nullptr;
    float own_lr = -1.f;                  // JSON "learningRate" (TrainableLayer.cu:58); negative: the optimizer's
    std::vector<float> pending_w;         // set_weights before the arena exists
    bool dirty = true;                    // packed copies out of date
    bool updated = false;                 // armed update: this layer's step has been enqueued behind its gradient already

    // packed copies
    void *Win = nullptr, *WinT = nullptr, *Wrec = nullptr, *WrecT = nullptr;
    float *bias_p = nullptr, *peep_p = nullptr;
    float *grad_block = nullptr;          // dWin | dWrec | dbias | dpeep  (or dW | colsum), zeroed per backward
    size_t grad_block_floats = 0;
    float *dWin = nullptr, *dWrec = nullptr, *dbias = nullptr, *dpeep = nullptr;

    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_pack = nullptr;         // operand copies rebuilt on the side stream after cn_sgd_update_all
    bool pack_pending = false;

    char kname[2][CN_KNAME_LEN] = {{0}, {0}};   // recurrent kernels the last forward / backward pass launched (written by the launchers)

    // deterministic mode (allocated at the first backward pass that needs them)
    float *det_ws = nullptr;              // split-K partial products: DET_MAX_SPLITS copies of [dWin | dWrec]
    float *gpart = nullptr; int gpart_slots = 0;   // LSTM: per-workgroup bias / peephole sums (LstmRec::gpart)
    float *det_colpart = nullptr;         // dense layers: per-workgroup column sums (det_colsum_part_floats)

    std::vector<void *> owned;            // device allocations to free

    size_t maxN() const { return (size_t)PSp * maxT; }
};

// ---- options (cn_internal.h) -------------------------------------------------------------------
namespace cn {
static const Options &default_options() { static const Options o = options_from_env(); return o; }
static thread_local const Options *t_opt = nullptr;
const Options &opt() { return t_opt ? *t_opt : default_options(); }
Options options_from_env()
{
    Options o;
    auto env_name = [](const char *name) { std::string e = "CN_"; for (const char *p = name; *p; ++p) e += (char)toupper((unsigned char)*p); return e; };
#define CN_OPT_FLAG(name) o.name = getenv(env_name(#name).c_str()) != nullptr;
#define CN_OPT_NUM(name, dflt) if (const char *v = getenv(env_name(#name).c_str())) o.name = atol(v);
    CN_OPTION_LIST(CN_OPT_FLAG, CN_OPT_NUM)
#undef CN_OPT_FLAG
#undef CN_OPT_NUM
    return o;
}
bool option_set(Options &o, const char *name, long value)
{
#define CN_OPT_FLAG(n) if (!strcmp(name, #n)) { o.n = value != 0; return true; }
#define CN_OPT_NUM(n, dflt) if (!strcmp(name, #n)) { o.n = value; return true; }
    CN_OPTION_LIST(CN_OPT_FLAG, CN_OPT_NUM)
#undef CN_OPT_FLAG
#undef CN_OPT_NUM
    return false;
}
bool option_get(const Options &o, const char *name, long *value)
{
#define CN_OPT_FLAG(n) if (!strcmp(name, #n)) { *value = o.n; return true; }
#define CN_OPT_NUM(n, dflt) if (!strcmp(name, #n)) { *value = o.n; return true; }
    CN_OPTION_LIST(CN_OPT_FLAG, CN_OPT_NUM)
#undef CN_OPT_FLAG
#undef CN_OPT_NUM
    return false;
}
}  // namespace cn

namespace {

// every entry point that touches the device: bind it and make the context's options the ones the launch paths see
void enter(cn_ctx *c)
{
    HIP_CHECK(hipSetDevice(c->device));
    cn::t_opt = &c->opt;
}

void *dalloc(cn_layer *l, size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0) bytes = 16;
    HIP_CHECK(hipMalloc(&p, bytes));
    HIP_CHECK(hipMemsetAsync(p, 0, bytes, l->ctx->stream));
    l->owned.push_back(p);
    return p;
}

// A per-frame buffer of an LSTM layer with CN_GUARD_STEPS time steps of zeros in front of frame 0 and behind frame maxN: the
// hand-written recurrent loops (cn_lstm_s2.hip) prefetch several steps ahead in EVERY step, also in the last ones, where the
// target lies outside [0, T) -- those loads must hit mapped memory, their values are never used.  (Nobody writes the guards:
// every kernel addresses frames 0 .. N-1 from the returned pointer.)
void *dalloc_guarded(cn_layer *l, size_t bytes, size_t bytes_per_step)
{
    const size_t guard = (size_t)CN_GUARD_STEPS * bytes_per_step;
    return (char *)dalloc(l, bytes + 2 * guard) + guard;
}

// ---- CU-masked stream ------------------------------------------------------------------------
// One CU-masked stream per (device, CU count) for the life of the process, shared by every context on that device.
// hipStreamDestroy of such a stream is not reliable on ROCm 7.2: depending on the network (seen with 2, 4, 5 and 6
// layers of blstm1024, not with the headline topology) it never returned although hipStreamSynchronize and
// hipStreamQuery reported the stream idle, with or without a hipDeviceSynchronize in front.  Contexts order their own
// work on it with events, so sharing it costs at most some serialisation between contexts.
hipStream_t masked_stream(int device, int ncu, int total)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, hipStream_t> pool;
    if (ncu <= 0 || ncu >= total) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto it = pool.find({device, ncu});
    if (it != pool.end()) return it->second;
    std::vector<uint32_t> mask((total + 31) / 32, 0u);
    for (int i = 0; i < total; ++i)
        if ((long)(i + 1) * ncu / total != (long)i * ncu / total) mask[i / 32] |= 1u << (i % 32);
    hipStream_t st = nullptr;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { (void)hipGetLastError(); st = nullptr; }
    pool[{device, ncu}] = st;
    return st;
}

// ---- row map -------------------------------------------------------------------------------
// The N-wide products of a fraction may skip its dummy frames (GemmNT::rowmap): the map was built behind the re-layout of the
// fraction that is current now (launch_fraction_load / launch_rowmap)
static void use_rowmap(cn_ctx *c, GemmNT &g)
{
    if (opt().no_nt_rowmap || !c->d_pat_raw || !c->loaded) return;
    int *rm = c->rowmap_of(c->d_pat_raw);
    g.rowcnt = rm; g.rowmap = rm + 4; g.dummymap = rm + 4 + c->pat_maxN; g.m_est = c->est_real;
}

// ---- timing ---------------------------------------------------------------------------------
hipEvent_t get_event(cn_ctx *c)
{
    if (!c->free_events.empty()) { hipEvent_t e = c->free_events.back(); c->free_events.pop_back(); return e; }
    hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); return e;
}
struct Timed {
    cn_ctx *c; int cls; hipEvent_t a = nullptr; hipStream_t st;
    Timed(cn_ctx *ctx, int k, hipStream_t stream = nullptr) : c(ctx), cls(k), st(stream ? stream : ctx->stream)
    {
        if (c->timing) { a = get_event(c); hipEventRecord(a, st); }
    }
    ~Timed()
    {
        if (a) { hipEvent_t b = get_event(c); hipEventRecord(b, st); c->spans[cls].push_back({a, b}); }
    }
};

// wide softmax rows of the bf16 throughput mode: v_exp_f32 + one reciprocal per row (cn_elementwise.hip, softmax_exp<FAST>)
bool softmax_fast(cn_ctx *c)
{
    const bool exact = opt().softmax_exact;      // A/B switch
    return c->prec == P_BF16 && !exact;
}
// fp32 outputs of a feed-forward / softmax layer for a reader other than the fused softmax backward kernel
float *posteriors(cn_layer *o)
{
    cn_ctx *c = o->ctx;
    o->sm_read = true;
    if (o->sm_lazy) {
        launch_softmax_normalise(c->stream, softmax_fast(c), o->out_f32, c->d_pat, c->N, o->size, o->Lp, o->sm_stat);
        o->sm_lazy = false;
    }
    return o->out_f32;
}

// the main stream waits for everything the side stream still has in flight
void join_side(cn_ctx *c)
{
    // events of one stream complete in order: waiting for the newest of each stream covers the older ones (a wait on an
    // event that has long completed still costs the stream ~3 us, and these sit in front of the weight update)
    for (size_t i = 0; i < c->pending_joins.size(); ++i) {
        bool newest = true;
        for (size_t j = i + 1; j < c->pending_joins.size(); ++j) newest = newest && c->pending_join_streams[j] != c->pending_join_streams[i];
        if (newest) HIP_CHECK(hipStreamWaitEvent(c->stream, c->pending_joins[i], 0));
    }
    c->pending_joins.clear(); c->pending_join_streams.clear();
    // ... and for the gradient all-reduces of the communication stream (in order on that stream: the newest covers all)
    if (c->comm_pending) { HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_comm, 0)); c->comm_pending = false; }
}
// `st` waits for the gradient work of `layer` and for nothing else the context has enqueued since
void stream_wait_layer(cn_layer *layer, hipStream_t st)
{
    cn_ctx *c = layer->ctx;
    bool pending = false;
    for (hipEvent_t e : c->pending_joins) pending = pending || e == layer->ev_join;
    if (pending) { HIP_CHECK(hipStreamWaitEvent(st, layer->ev_join, 0)); return; }
    // nothing of this layer is on the side stream (CN_NO_OVERLAP, the first trainable layer, or already joined): its
    // gradient is ordered on the context's stream
    if (!c->ev_ext) HIP_CHECK(hipEventCreateWithFlags(&c->ev_ext, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(c->ev_ext, c->stream));
    HIP_CHECK(hipStreamWaitEvent(st, c->ev_ext, 0));
}

// ---- RCCL, opened at run time ----------------------------------------------------------------
// No link-time dependency: a host that never trains data-parallel needs no librccl, and a process that already holds
// one (PyTorch ships a librccl.so with the SONAME librccl.so.1) shares it instead of loading a second copy.
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};
Rccl &rccl()
{
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (r.handle) return r;
    const char *names[] = {getenv("CN_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    std::string tried;
    for (const char *n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        tried += std::string(tried.empty() ? "" : "; ") + dlerror();
    }
    if (!h) throw cn_error(CN_ERR_COMM, "cannot open librccl (data-parallel training needs RCCL): " + tried);
    auto sym = [&](const char *name) {
        void *f = dlsym(h, name);
        if (!f) throw cn_error(CN_ERR_COMM, std::string("librccl lacks ") + name);
        return f;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.handle = h;
    return r;
}
void rccl_check(ncclResult_t e, const char *what)
{
    if (e != ncclSuccess) throw cn_error(CN_ERR_COMM, std::string(what) + ": " + rccl().GetErrorString(e));
}
#define RCCL_CHECK(x) rccl_check((x), #x)
void require_comm(cn_ctx *c, const char *who)
{
    if (!c->has_comm()) throw cn_error(CN_ERR_STATE, std::string(who) + ": no communicator bound to this context (call cn_comm_init first)");
}
// cn_fraction_prefetch_resident: the re-layout of the announced fraction, into the alternate buffers
void launch_prefetch(cn_ctx *c, hipStream_t st)
{
    cn_ctx::Prefetch &p = c->pf;
    const cn_fraction &f = p.f;
    const bool cls = p.post && (p.post->kind == CN_LAYER_MULTICLASS_CLASSIFICATION || p.post->kind == CN_LAYER_BINARY_CLASSIFICATION);
    if (p.host) HIP_CHECK(hipStreamWaitEvent(st, c->ev_up[p.slot], 0));      // cn_fraction_prefetch: f points into a staging area whose upload is on the copy stream
    launch_fraction_load(st, c->f32, f.max_seq_length, c->PS, c->PSp, (const char *)f.pat_types, p.pat,
                         cls ? (const int *)f.target_classes : nullptr, p.tcls,
                         (p.post && !cls) ? (const float *)f.targets : nullptr, p.post ? p.targets : nullptr,
                         p.post ? p.post->size : 0, (const float *)f.inputs, p.input->size, p.in_op, p.input->Lp, c->rowmap_of(p.pat_raw), (int)c->pat_maxN, f.min_seq_length);
    if (p.post && p.post->kind == CN_LAYER_BINARY_CLASSIFICATION)
        launch_classes_to_targets(st, p.tcls, p.targets, f.max_seq_length * c->PSp);
    if (p.host) { HIP_CHECK(hipEventRecord(c->ev_free[p.slot], st)); c->stage_used[p.slot] = true; }
    p.launched = true;
}
// run `f(stream)` on the side stream after everything enqueued on the main stream so far
// (fork_attached: ev_fork already completes with the last main-stream kernel, see fork_event)
template <typename F> void on_side(cn_layer *l, F &&f, bool fork_attached = false)
{
    cn_ctx *c = l->ctx;
    c->tn_cus = 0;
    if (!c->overlap) { f(c->stream, nullptr); return; }
    // The first trainable layer's gradient work has nothing to run beside: only the weight update follows.  On the main
    // stream it saves the two cross-stream hand-offs (fork, join) in front of the update.
    const bool tail_on_main = !opt().tail_on_side;
    if (tail_on_main && l->prev && !l->prev->trainable && !fork_attached) { f(c->stream, nullptr); return; }
    if (!l->ev_fork) { HIP_CHECK(hipEventCreateWithFlags(&l->ev_fork, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&l->ev_join, hipEventDisableTiming)); }
    if (!fork_attached) HIP_CHECK(hipEventRecord(l->ev_fork, c->stream));
    // a recurrent kernel follows on the main stream when the preceding layer is an LSTM layer: slow lane (a CU-masked stream, so
    // that the gradient products leave the latency-bound recurrent kernel's CUs and memory path alone) -- unless the products
    // would outlast that kernel on the masked CUs and end up in front of the next N-wide product of the critical path
    // (LVCSR config: the softmax layer's 8000 x 1024 gradient took 2.0 ms on 80 CUs beside a 1.3 ms recurrent kernel and the
    // error product behind it 757 instead of 140 us; 13.3 -> 12.7 ms per fraction with the rule below)
    bool slow = c->side_slow && l->prev && l->prev->lstm;
    const bool side_rule = !opt().no_side_rule;
    // ~1.5 TFLOP/s per CU on the 64 x 64 tiles; the 256 x 256 kernel of the wide layers (cn_gemm_tn_big.hip) runs at ~3
    const bool big = c->prec == P_BF16 && c->N >= 4096 && (l->lstm ? l->Hp >= 256 && l->Pp >= 192 : l->Lp >= 512 && l->Pp >= 192);
    // CUs the recurrent kernel that follows on the main stream will occupy when it is a cluster launch (0: one-CU kernels)
    int rec_cus = 0;
    if (l->prev && l->prev->lstm) {
        LstmRec r{};
        r.Hp = l->prev->Hp; r.dirs = l->prev->dirs; r.PS = c->PSp; r.T = c->T; r.rpl = c->rpl; r.num_cus = r.cluster_cus = c->num_cus;
        rec_cus = c->d_xch ? lstm_cluster_bwd_cus(c->prec, r) : 0;
    }
    if (slow && side_rule) {
        const double flops = 2.0 * c->N * (l->lstm ? (double)l->dirs * 4 * l->Hp * (l->Pp + l->Hp) : (double)l->Lp * l->Pp);
        const double t_side = flops / (c->side_cus * (big ? 3.0e12 : 1.5e12));
        const double t_rec = c->T * (l->prev->Hp > 192 ? 1.2e-6 : 0.5e-6);                           // cluster / single-CU step
        if (t_side > 0.8 * t_rec) slow = false;
        // a cluster grid that covers a large part of the chip (two sequences per two-CU cluster: 112-128 CUs, each claiming its
        // CU's whole LDS) lands on the masked CUs too and leaves the masked stream a fraction of them: no slow lane then
        // (LVCSR with the masked stream beside a 128-CU grid: gradient products 10.4 -> 19.2 ms per six fractions)
        if (rec_cus > c->num_cus / 4) slow = false;
    }
    hipStream_t st = slow ? c->side_slow : c->side;
    // The 256 x 256 gradient kernel puts ONE long-lived workgroup on a CU (128 KB of LDS): beside a recurrent kernel it must leave
    // that kernel's CUs alone -- a cluster grid starts when ALL its workgroups are resident, and the recurrent workgroups claim a
    // whole CU's LDS.  Masked stream: its CUs; unmasked beside a recurrent kernel: the chip less that kernel's grid (at least 64
    // CUs: the one-CU kernels at PS = 50 / 64 take 52 - 64).  (reading B with the kernel filling the chip: 2.79 -> 2.83 ms per
    // fraction, the recurrent kernel waited.)
    c->tn_cus = slow ? c->side_cus : (l->prev && l->prev->lstm && c->num_cus > 128 ? std::max(64, c->num_cus - std::max(64, rec_cus + 8)) : 0);
    HIP_CHECK(hipStreamWaitEvent(st, l->ev_fork, 0));
    if (c->pf.valid && !c->pf.launched) launch_prefetch(c, st);      // covered by this layer's join event (same stream, in order)
    // the join event rides on the last kernel of the side work too (f returns true when it attached it)
    const bool join_attached = f(st, (c->attach_forks && !c->timing) ? l->ev_join : nullptr);
    if (!join_attached) HIP_CHECK(hipEventRecord(l->ev_join, st));
    c->pending_joins.push_back(l->ev_join); c->pending_join_streams.push_back(st);
}
// The event the side stream forks from, to be attached to the last main-stream kernel in front of the fork (a stop
// event of that launch, hipExtLaunchKernelGGL): a separate hipEventRecord is a marker packet between two kernels of the
// critical path and delays the second one by ~7 us (three to four forks per training step; 1.41 -> 1.39 ms per step with
// the fork, join and update events all riding on kernels).  nullptr: record the event the ordinary way.
hipEvent_t fork_event(cn_layer *l)
{
    cn_ctx *c = l->ctx;
    if (!c->overlap || !c->attach_forks) return nullptr;
    if (!l->ev_fork) { HIP_CHECK(hipEventCreateWithFlags(&l->ev_fork, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&l->ev_join, hipEventDisableTiming)); }
    return l->ev_fork;
}
void timing_collect(cn_ctx *c)
{
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->side) HIP_CHECK(hipStreamSynchronize(c->side));
    if (c->side_slow) HIP_CHECK(hipStreamSynchronize(c->side_slow));
    if (c->comm_stream) HIP_CHECK(hipStreamSynchronize(c->comm_stream));
    for (int k = 0; k < KC_COUNT; ++k) {
        for (auto &sp : c->spans[k]) {
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, sp.a, sp.b));
            c->acc_ms[k] += ms; c->acc_n[k] += 1;
            c->free_events.push_back(sp.a); c->free_events.push_back(sp.b);
        }
        c->spans[k].clear();
    }
}

// ---- geometry -------------------------------------------------------------------------------
int pad_units(int H) { return H > 512 ? round_up(H, 64) : round_up(H, 32); }

LstmGeom lstm_geom(const cn_layer *l)
{
    LstmGeom g;
    g.P = l->P; g.Pp = l->Pp; g.L = l->size; g.H = l->H; g.Hp = l->Hp; g.dirs = l->dirs;
    g.prevH = l->prev->lstm ? l->prev->H : 0;
    g.prevHp = l->prev->lstm ? l->prev->Hp : 0;
    g.prevDirs = l->prev->lstm ? l->prev->dirs : 0;
    return g;
}
FfGeom ff_geom(const cn_layer *l)
{
    FfGeom g;
    g.P = l->P; g.Pp = l->Pp; g.L = l->size; g.Lp = l->Lp;
    g.prevH = l->prev->lstm ? l->prev->H : 0;
    g.prevHp = l->prev->lstm ? l->prev->Hp : 0;
    g.prevDirs = l->prev->lstm ? l->prev->dirs : 0;
    return g;
}
int ff_act(cn_layer_kind k)
{
    switch (k) {
    case CN_LAYER_FF_TANH: return ACT_TANH;
    case CN_LAYER_FF_LOGISTIC: return ACT_LOGISTIC;
    default: return ACT_IDENTITY;
    }
}

// ---- parameter arena ------------------------------------------------------------------------
void finalize(cn_ctx *c)
{
    if (c->finalized) return;
    size_t total = 0;
    for (cn_layer *l : c->layers)
        if (l->trainable) { l->woff = total; total += (size_t)round_up(l->nw, 4); }
    c->total = total;
    if (total) {
        HIP_CHECK(hipMalloc((void **)&c->arena, 3 * total * sizeof(float)));
        HIP_CHECK(hipMemsetAsync(c->arena, 0, 3 * total * sizeof(float), c->stream));
    }
    for (cn_layer *l : c->layers) {
        if (!l->trainable) continue;
        l->w = c->arena + l->woff;
        l->wu = c->arena + total + l->woff;
        l->wd = c->arena + 2 * total + l->woff;
        if (!l->pending_w.empty()) {
            HIP_CHECK(hipMemcpyAsync(l->w, l->pending_w.data(), l->pending_w.size() * sizeof(float),
                                     hipMemcpyHostToDevice, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            l->pending_w.clear();
        }
        l->dirty = true;
    }
    c->finalized = true;
}

void repack(cn_layer *l)
{
    cn_ctx *c = l->ctx;
    if (l->pack_pending) {       // rebuilt on the side stream after the last update (cn_sgd_update_all)
        // ONE wait, for the last copy that was rebuilt (same stream, in order: it covers every layer's); they are all
        // done long before the second trainable layer's forward pass asks, and each wait costs the stream ~3 us
        HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_pack_last, 0));
        for (cn_layer *o : c->layers) o->pack_pending = false;
    }
    if (!l->dirty) return;
    Timed tm(c, KC_OTHER);
    if (l->lstm) launch_lstm_pack(c->stream, c->f32, lstm_geom(l), l->bias, l->w, l->Win, l->WinT, l->Wrec, l->WrecT, l->bias_p, l->peep_p);
    else         launch_ff_pack(c->stream, c->f32, ff_geom(l), l->bias, l->w, l->Win, l->WinT, l->bias_p);
    l->dirty = false;
}

// after a stream sync: did a bounded spin of a cluster kernel give up?
void check_fault(cn_ctx *c)
{
    int f = 0;
    HIP_CHECK(hipMemcpy(&f, c->d_fault, sizeof(int), hipMemcpyDeviceToHost));
    if (f) {
        HIP_CHECK(hipMemset(c->d_fault, 0, sizeof(int)));
        throw cn_error(CN_ERR_HIP, "recurrent cluster kernel: inter-workgroup hand-off timed out (results of this pass are invalid)");
    }
}

void require_loaded(cn_ctx *c)
{
    if (!c->loaded) throw cn_error(CN_ERR_STATE, "no fraction loaded (call cn_fraction_load first)");
}

// a deferred cn_loss_accumulate that did not find a backward launch to ride on: the one-workgroup reduction now
void flush_loss(cn_ctx *c)
{
    if (!c->loss_deferred) return;
    c->loss_deferred = false;
    Timed tm(c, KC_OTHER);
    launch_rowstat_reduce(c->stream, c->d_rowstat, c->N, c->d_loss_acc, false);
}

// ---- forward / backward sequences -----------------------------------------------------------
void lstm_rec_args(cn_layer *l, LstmRec &r)
{
    cn_ctx *c = l->ctx;
    r.H = l->H; r.Hp = l->Hp; r.dirs = l->dirs; r.PS = c->PSp; r.T = c->T; r.Tmin = c->Tmin;
    r.pat = c->d_pat;
    r.acts = l->acts; r.cell = l->cell; r.th = l->th; r.y_op = l->out_op; r.Wrec = l->Wrec; r.peep = l->peep_p;
    r.pre16 = nullptr;
    r.err = l->err; r.delta_op = l->delta_op; r.WrecT = l->WrecT; r.dbias = l->dbias; r.dpeep = l->dpeep;
    r.bias = l->bias;
    r.rpl = c->rpl;
    r.xch = c->d_xch; r.fault = c->d_fault; r.num_cus = r.cluster_cus = c->num_cus;
    r.xch_packed = nullptr;
    // With a communicator bound, RCCL's persistent workgroups hold CUs on the communication stream while they wait for peer
    // ranks, beside the recurrent kernel of the layer below.  A cluster grid needs ALL its members resident (spin-wait
    // hand-off), so it must fit what RCCL leaves: the grid is sized against num_cus minus a margin for RCCL's channels
    // (CN_COMM_CU_MARGIN, default 32 -- RCCL's MI300-class defaults stay at or below that many workgroups per collective);
    // a grid that no longer fits takes the streaming kernels, which make no residency assumption.  The one-CU kernels (s2, s2w,
    // 4-sequence) make none either and keep the full count, so that a data-parallel run picks the kernels of a one-GPU run.
    if (c->has_comm()) {
        const int margin = (int)opt().comm_cu_margin;
        r.cluster_cus = c->num_cus - margin > 0 ? c->num_cus - margin : 1;
    }
    r.gpart = nullptr; r.gpart_slots = 0; r.det_grid = nullptr;
    if (c->d_xch) {
        const size_t off = lstm_cluster_xch_packed_offset(c->prec, l->Hp, l->dirs, c->PSp, c->rpl, c->num_cus);
        if (off) r.xch_packed = (unsigned long long *)((char *)c->d_xch + off);
    }
    r.kname = nullptr;
    // tag range of a cluster launch (cn_lstm_cluster.hip); cleared and restarted long before the 32-bit tags wrap
    if (c->d_xch && c->xch_epoch > 0xF0000000u) { HIP_CHECK(hipMemsetAsync(c->d_xch, 0, c->xch_bytes, c->stream)); c->xch_epoch = 0; }
    r.xch_epoch = c->xch_epoch;
}

// the single-CU recurrent kernels keep two operand tiles (and, backward, one byte per time step and sequence) in LDS
// deterministic mode: the layer's workspaces for stored partial sums (zeroed; owned by the layer)
void ensure_det(cn_layer *l)
{
    cn_ctx *c = l->ctx;
    if (l->det_ws) return;
    if (l->lstm) {
        const size_t R = (size_t)l->dirs * 4 * l->Hp;
        l->det_ws = (float *)dalloc(l, (size_t)DET_MAX_SPLITS * (R * l->Pp + R * l->Hp) * sizeof(float));
        // one slot per workgroup of the backward kernel: at most one workgroup per sequence, direction and cluster member
        l->gpart_slots = l->dirs * c->PSp * 8 + 64;
        l->gpart = (float *)dalloc(l, (size_t)l->gpart_slots * 7 * l->dirs * l->Hp * sizeof(float));
    } else {
        l->det_ws = (float *)dalloc(l, (size_t)DET_MAX_SPLITS * l->Lp * l->Pp * sizeof(float));
        l->det_colpart = (float *)dalloc(l, det_colsum_part_floats(l->Lp) * sizeof(float));
    }
}

void check_rec_lds(const cn_layer *l, bool bwd)
{
    const cn_ctx *c = l->ctx;
    const size_t need = lstm_rec_lds_bytes(c->prec, bwd, l->Hp, c->rpl, c->T);
    if (need > 160 * 1024)
        throw cn_error(CN_ERR_SHAPE, "LSTM layer with " + std::to_string(l->H) + " units per direction, " + std::to_string(c->T) +
                       " time steps: the recurrent " + (bwd ? "backward" : "forward") + " kernel needs " + std::to_string(need / 1024) +
                       " KB of LDS per workgroup (160 KB available)" + (c->prec == P_F32 ? "; use CN_PREC_BF16 or CN_PREC_BF16X3 for layers this wide" : "") +
                       " or shorter fractions (truncate_seq)");
}

// One layer's weight update + operand copies as one launch of the grouped pack kernel on `st`.
// mode 1: gradient from the flat weightUpdates; mode 2: from the packed accumulators (unpack fused in, cn_elementwise.hip)
// folds (mode 2, deterministic mode; nullable): f_in, f_rec[0], f_rec[1], f_bias -- the partial sums the launch adds itself
void launch_layer_update(hipStream_t st, cn_layer *l, int mode, float lr, float mom, hipEvent_t done, const PackFold *folds = nullptr)
{
    cn_ctx *c = l->ctx;
    PackGroup grp{};
    PackItem &it = grp.item[grp.n++];
    it.lstm = l->lstm ? 1 : 0;
    if (l->lstm) it.lg = lstm_geom(l); else it.fg = ff_geom(l);
    it.bias = l->bias; it.w = l->w; it.Win = l->Win; it.WinT = l->WinT; it.Wrec = l->Wrec; it.WrecT = l->WrecT;
    it.bias_p = l->bias_p; it.peep_p = l->peep_p;
    it.update = mode; it.w_rw = l->w; it.wu = l->wu; it.wd = l->wd; it.wu_rw = l->wu;
    it.lr = l->own_lr >= 0.f ? l->own_lr : lr; it.mom = mom;
    it.g_in = l->dWin; it.g_rec = l->dWrec; it.g_bias = l->dbias; it.g_peep = l->dpeep;
    if (folds) { it.update = 3; it.f_in = folds[0]; it.f_rec[0] = folds[1]; it.f_rec[1] = folds[2]; it.f_bias = folds[3]; }
    launch_pack_group(st, c->f32, grp, done);
    l->dirty = false; l->pack_pending = false; l->updated = true;
}
// armed update without a communicator: unpack + update + operand copies ride on ONE launch behind the gradient GEMMs
bool armed_fused(const cn_layer *l) { return l->ctx->armed && !l->ctx->has_comm(); }

void lstm_forward(cn_layer *l)
{
    cn_ctx *c = l->ctx;
    const int R = l->dirs * 4 * l->Hp;
    repack(l);
    // bf16 mode, two-sequence forward kernels: the pre-activations leave the product as bf16 (8 instead of 16 bytes per unit and
    // frame -- the product is bound by that store: 72 of the 88 MB a headline launch moved) and the recurrent kernel widens them
    bool pre16 = false;
    if (l->pre16) {
        LstmRec probe; lstm_rec_args(l, probe);
        pre16 = lstm_cluster_size(c->prec, l->Hp, l->dirs, c->PSp, c->rpl, probe.cluster_cus) == 0 && lstm_fwd_takes_pre16(c->prec, probe);
    }
    {   // K1: gate pre-activations of all frames, 4 gates x dirs packed into one N = R product
        Timed tm(c, KC_GEMM_WIDE);
        GemmNT g{};
        g.A = l->prev->out_op; g.lda = l->Pp; g.B = l->Win; g.ldb = l->Pp;
        g.C = l->acts; g.ldc = R; g.C2 = nullptr; g.ldc2 = 0; g.bias = l->bias_p; g.act = ACT_IDENTITY;
        if (pre16) { g.C = nullptr; g.C2 = l->pre16; g.ldc2 = R; }
        g.M = c->N; g.N = R; g.K = l->Pp;
        if (l->prev->lstm) use_rowmap(c, g);          // (y = 0 on dummy frames; what the caller's inputs hold there is the caller's business)
        launch_gemm_nt(c->stream, c->prec, g);
    }
    {   // K2+K3+K4: the whole time loop
        Timed tm(c, KC_REC_FWD);
        LstmRec r; lstm_rec_args(l, r); r.kname = l->kname[0];
        if (pre16) r.pre16 = l->pre16;
        if (!launch_lstm_cluster(c->stream, c->prec, false, r, &c->xch_epoch)) { check_rec_lds(l, false); launch_lstm_forward(c->stream, c->prec, r); }
        HIP_CHECK(hipGetLastError());
    }
}

void lstm_backward(cn_layer *l)
{
    cn_ctx *c = l->ctx;
    const int R = l->dirs * 4 * l->Hp, Hp = l->Hp, PS = c->PSp, N = c->N;
    const size_t e = c->esz();
    repack(l);
    bool fork_attached = false;
    // (the packed gradient accumulators are zero here: allocation clears them, the unpack kernel re-clears them)
    int det_grid = 0;
    if (c->det) ensure_det(l);
    {   // K5+K6+K7 and the bias / peephole sums of K9
        Timed tm(c, KC_REC_BWD);
        LstmRec r; lstm_rec_args(l, r); r.kname = l->kname[1];
        if (c->det) { r.gpart = l->gpart; r.gpart_slots = l->gpart_slots; r.det_grid = &det_grid; }
        if (!launch_lstm_cluster(c->stream, c->prec, true, r, &c->xch_epoch)) {
            check_rec_lds(l, true);
            // no K8 behind this kernel (the preceding layer is the input layer): the side stream forks from it directly
            const bool tail_on_side = opt().tail_on_side;
            hipEvent_t fork = (tail_on_side && !l->prev->trainable && !c->timing) ? fork_event(l) : nullptr;
            launch_lstm_backward(c->stream, c->prec, r, fork);
            fork_attached = fork != nullptr;
        }
        HIP_CHECK(hipGetLastError());
    }
    if (l->prev->trainable) {   // K8 (LstmLayer.cu:990-1009): one K = R product instead of 4*dirs
        Timed tm(c, KC_GEMM_WIDE);
        GemmNT g{};
        g.A = l->delta_op; g.lda = R; g.B = l->WinT; g.ldb = R;
        g.C = l->prev->err; g.ldc = l->prev->Lp; g.bias = nullptr; g.act = ACT_IDENTITY;
        g.M = N; g.N = l->Pp; g.K = R;
        use_rowmap(c, g);
        hipEvent_t fork = c->timing ? nullptr : fork_event(l);     // (timing mode records its own events around the kernel)
        launch_gemm_nt(c->stream, c->prec, g, fork);
        fork_attached = fork != nullptr;
    }
    // K9 runs on the side stream: it only feeds weightUpdates, forked AFTER K8 so the critical-path GEMM has the chip to itself, and running beside the
    // preceding layer's recurrent kernel (which occupies ~10 % of the CUs)
    on_side(l, [&](hipStream_t st, hipEvent_t join) {
        // deterministic mode with the armed, fused update behind the products: that launch adds the partial sums itself (no fold
        // launch at all); otherwise one fold launch rides behind the grouped product
        const bool fused = armed_fused(l);
        const bool defer = c->det && fused;
        int used_in = 0, used_rec[2] = {0, 0};
        {   // K9 input weights: dWin[r][i] = sum_n delta[n][r] x[n][i]
            Timed tm(c, KC_GEMM_GRAD, st);
            GemmTN gs[3]; int ng = 0;
            GemmTN g{};
            g.A = l->delta_op; g.lda = R; g.B = l->prev->out_op; g.ldb = l->Pp;
            g.C = l->dWin; g.ldc = l->Pp; g.M = R; g.N = l->Pp; g.K = N;
            if (c->det) { g.ws = l->det_ws; g.ws_splits = DET_MAX_SPLITS; g.ws_used = defer ? &used_in : nullptr; }
            gs[ng++] = g;
            // K9 recurrent weights: dWrec[(j,g)][i] = sum_t delta[t][(j,g)] y[prev(t)][i]
            if (N > PS) {
                for (int d = 0; d < l->dirs; ++d) {
                    GemmTN r{};
                    const char *dl = (const char *)l->delta_op + (size_t)d * 4 * Hp * e;
                    const char *y = (const char *)l->out_op + (size_t)d * Hp * e;
                    if (d == 0) { r.A = dl + (size_t)PS * R * e; r.B = y; }                       // skipFirstPattern, LstmLayer.cu:432-435
                    else        { r.A = dl; r.B = y + (size_t)PS * l->Lp * e; }                   // skipLastPattern,  :428-431
                    r.lda = R; r.ldb = l->Lp;
                    r.C = l->dWrec + (size_t)d * 4 * Hp * Hp; r.ldc = Hp; r.M = 4 * Hp; r.N = Hp; r.K = N - PS;
                    if (c->det) { r.ws = l->det_ws + (size_t)DET_MAX_SPLITS * ((size_t)R * l->Pp + (size_t)d * 4 * Hp * Hp); r.ws_splits = DET_MAX_SPLITS; r.ws_used = defer ? &used_rec[d] : nullptr; }
                    gs[ng++] = r;
                }
            }
            // deterministic mode: the workgroups' bias / peephole sums, added in workgroup order (dbias and dpeep are neighbours, in
            // the gradient block and in a slot alike), in the launch that adds the products' split partials
            const int slot = 7 * l->dirs * Hp;
            const FoldItem f{l->dbias, l->gpart, (long)slot, det_grid, 1, slot, slot, 1, 1};
            launch_gemm_tn_group(st, c->prec, gs, ng, c->tn_cus, (c->det && !defer) ? &f : nullptr);      // the three products side by side in one launch
        }
        {
            Timed tm(c, KC_OTHER, st);
            if (fused) {
                const int slot = 7 * l->dirs * Hp;
                const PackFold folds[4] = {
                    {l->det_ws, (long)R * l->Pp, used_in, 0},
                    {l->det_ws ? l->det_ws + (size_t)DET_MAX_SPLITS * (size_t)R * l->Pp : nullptr, 4L * Hp * Hp, used_rec[0], 0},
                    {l->det_ws ? l->det_ws + (size_t)DET_MAX_SPLITS * ((size_t)R * l->Pp + (size_t)4 * Hp * Hp) : nullptr, 4L * Hp * Hp, used_rec[1], 0},
                    {l->gpart, (long)slot, det_grid, 1}};
                launch_layer_update(st, l, 2, c->arm_lr, c->arm_mom, join, defer ? folds : nullptr);
            } else launch_lstm_unpack_grads(st, lstm_geom(l), l->dWin, l->dWrec, l->dbias, l->dpeep, l->wu, join);
        }
        return join != nullptr;
    }, fork_attached);
}

void ff_forward(cn_layer *l)
{
    cn_ctx *c = l->ctx;
    repack(l);
    const bool softmax = l->kind == CN_LAYER_SOFTMAX;
    {
        Timed tm(c, KC_GEMM_WIDE);
        GemmNT g{};
        g.A = l->prev->out_op; g.lda = l->Pp; g.B = l->Win; g.ldb = l->Pp;
        g.C = l->out_f32; g.ldc = l->Lp;
        g.C2 = (!c->f32 && !softmax) ? l->out_op : nullptr; g.ldc2 = l->Lp;
        g.bias = l->bias_p; g.act = ff_act(l->kind);
        g.M = c->N; g.N = l->Lp; g.K = l->Pp;
        if (l->prev->lstm) use_rowmap(c, g);          // (y = 0 on dummy frames, ComputeBlockOutputFn; a feed-forward layer's output is act(bias) there)
        launch_gemm_nt(c->stream, c->prec, g);
    }
    if (softmax) {
        flush_loss(c);                         // (the row statistics are about to be overwritten)
        Timed tm(c, KC_OTHER);
        const bool stat = c->d_rowstat != nullptr;
        // wide rows in training: the posteriors are only ever read by the fused backward kernel, which can recompute them from the
        // logits -- 1.1 GB less to write per fraction of the 8000-class output layer.  A layer whose posteriors were asked for after
        // its last forward pass (forward-only use, another loss than multiclass) runs the eager kernel.  Only where recomputing is
        // cheap (softmax_fast): with expf and a division per element the two lazy passes take longer than the eager ones
        // (probe/softmax_bench: 363 + 633 against 454 + 427 us).
        const bool lazy_off = opt().no_lazy_softmax;
        const bool lazy_force = opt().lazy_softmax;                              // the exact lazy kernels
        l->sm_lazy = stat && l->sm_stat && l->sm_lazy_next && !l->has_follower && !lazy_off && (softmax_fast(c) || lazy_force);
        l->sm_read = false;
        launch_softmax_fwd(c->stream, l->out_f32, c->d_pat, c->N, l->size, l->Lp, stat ? c->d_tcls : nullptr, stat ? c->d_rowstat : nullptr,
                           softmax_fast(c), l->sm_lazy ? l->sm_stat : nullptr);
        c->rowstat_of = stat ? l : nullptr;
        if (!c->f32 && l->has_follower) launch_pad_convert(c->stream, false, l->out_f32, c->N, l->Lp, l->out_op, l->Lp);
    }
}

void ff_backward(cn_layer *l)
{
    cn_ctx *c = l->ctx;
    const int N = c->N;
    repack(l);
    if (c->det) ensure_det(l);
    FoldItem colfold{}; colfold.nparts = 0;       // deterministic mode: the column sums' fold rides on the gradient product's
    bool dummy_deltas_zero = false;               // (the row map's condition: every operand row of a dummy frame is zero)
    {
        Timed tm(c, KC_OTHER);
        if (l->kind == CN_LAYER_SOFTMAX && l->mcc_pending && l->Lp <= 8192) {
            dummy_deltas_zero = true;             // softmax_mcc_bwd_kernel stores zeros for every pattern that is not real
            // bf16 mode: the fp32 outputErrors stay unwritten (read back from the bf16 operand copy if anyone asks)
            const bool with_loss = c->loss_deferred && c->rowstat_of == l && softmax_mcc_bwd_takes_loss(l->Lp);
            launch_softmax_mcc_bwd(c->stream, c->f32, l->out_f32, c->d_tcls, c->d_pat, N, l->size, l->Lp, c->f32 ? l->err : nullptr, l->delta_op, l->dbias,
                                   with_loss ? c->d_rowstat : nullptr, with_loss ? c->d_loss_acc : nullptr, c->d_loss + 6, l->sm_lazy ? l->sm_stat : nullptr,
                                   softmax_fast(c), c->d_colpart, c->det ? l->det_colpart : nullptr, &colfold);
            l->sm_lazy_next = !l->sm_read;
            if (with_loss) c->loss_deferred = false;
            l->err_in_delta = !c->f32;
        } else {
            l->err_in_delta = false;
            if (l->kind == CN_LAYER_SOFTMAX) l->sm_lazy_next = false;   // no fused consumer: lazy rows would be normalised on demand every pass
            const float *y = posteriors(l);
            if (l->mcc_pending) launch_mcc_backward(c->stream, y, c->d_tcls, N, l->size, l->Lp, l->err);
            if (l->kind == CN_LAYER_SOFTMAX) launch_softmax_bwd(c->stream, y, l->err, c->d_pat, N, l->size, l->Lp);
            launch_ff_delta(c->stream, c->f32, ff_act(l->kind), y, l->err, l->delta_op, N, l->size, l->Lp);
            launch_colsum(c->stream, l->err, N, l->Lp, l->dbias, c->det ? l->det_colpart : nullptr, &colfold);
        }
        l->mcc_pending = false;
    }
    bool fork_attached = false;
    if (l->prev->trainable) {   // FeedForwardLayer.cu:188-198
        Timed tm(c, KC_GEMM_WIDE);
        GemmNT g{};
        g.A = l->delta_op; g.lda = l->Lp; g.B = l->WinT; g.ldb = l->Lp;
        g.C = l->prev->err; g.ldc = l->prev->Lp; g.bias = nullptr; g.act = ACT_IDENTITY;
        g.M = N; g.N = l->Pp; g.K = l->Lp;
        if (dummy_deltas_zero) use_rowmap(c, g);
        hipEvent_t fork = c->timing ? nullptr : fork_event(l);
        launch_gemm_nt(c->stream, c->prec, g, fork);
        fork_attached = fork != nullptr;
    }
    on_side(l, [&](hipStream_t st, hipEvent_t join) {
        const bool fused = armed_fused(l);
        const bool defer = c->det && fused;
        int used_in = 0;
        {   // FeedForwardLayer.cu:200-207
            Timed tm(c, KC_GEMM_GRAD, st);
            GemmTN g{};
            g.A = l->delta_op; g.lda = l->Lp; g.B = l->prev->out_op; g.ldb = l->Pp;
            g.C = l->dWin; g.ldc = l->Pp; g.M = l->Lp; g.N = l->Pp; g.K = N;
            if (c->det) { g.ws = l->det_ws; g.ws_splits = DET_MAX_SPLITS; g.ws_used = defer ? &used_in : nullptr; }
            launch_gemm_tn(st, c->prec, g, c->tn_cus, (colfold.nparts && !defer) ? &colfold : nullptr);
        }
        {
            Timed tm(c, KC_OTHER, st);
            if (fused) {
                // (the column sums' partial rows: colfold describes them; its target, the packed bias gradient, stays zero)
                const PackFold folds[4] = {{l->det_ws, (long)l->Lp * l->Pp, used_in, 0}, {nullptr, 0, 0, 0}, {nullptr, 0, 0, 0},
                                           {colfold.part, colfold.stride, colfold.nparts, 0}};
                launch_layer_update(st, l, 2, c->arm_lr, c->arm_mom, join, defer ? folds : nullptr);
            } else launch_ff_unpack_grads(st, ff_geom(l), l->bias, l->dWin, l->dbias, l->wu, join);
        }
        return join != nullptr;
    }, fork_attached);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" {

const char *cn_version(void) { return "currennt_hip 0.1 (gfx950)"; }

const char *cn_last_error(cn_ctx *) { return g_last_error.c_str(); }

const char *cn_device_arch(cn_ctx *ctx) { return ctx ? ctx->arch.c_str() : ""; }

int cn_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return count;
}

int cn_device_name(int index, char *buf, int buf_size)
{
    if (!buf || buf_size <= 0) { g_last_error = "cn_device_name: bad buffer"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, index));
        snprintf(buf, (size_t)buf_size, "%s (%s)", prop.name, prop.gcnArchName);
    });
}

int cn_ctx_create(int device_id, cn_precision precision, void *stream, cn_ctx **out)
{
    if (!out) { g_last_error = "cn_ctx_create: out is NULL"; return CN_ERR_BAD_ARG; }
    *out = nullptr;
    if (precision != CN_PREC_F32 && precision != CN_PREC_BF16 && precision != CN_PREC_BF16X3) { g_last_error = "cn_ctx_create: unknown precision"; return CN_ERR_BAD_ARG; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        g_last_error = "cn_ctx_create: no HIP device available (this library has no CPU fallback)";
        return CN_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= count) { g_last_error = "cn_ctx_create: device id out of range"; return CN_ERR_BAD_ARG; }
    cn_ctx *c = nullptr;
    int rc = guarded([&] {
        HIP_CHECK(hipSetDevice(device_id));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
        std::string arch = prop.gcnArchName;
        if (arch.rfind("gfx950", 0) != 0)
            throw cn_error(CN_ERR_NO_DEVICE, "cn_ctx_create: device is " + arch + ", this library is built for gfx950 only");
        c = new cn_ctx;
        c->device = device_id; c->arch = arch; c->f32 = (precision != CN_PREC_BF16);
        c->prec = precision == CN_PREC_BF16 ? P_BF16 : (precision == CN_PREC_BF16X3 ? P_X3 : P_F32);
        c->num_cus = prop.multiProcessorCount;
        if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
        else { HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
        HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        {
            // 5/16 of the CUs (80 of 256) measured best; counts that divide the CU numbering evenly (32, 64, 96) put the
            // mask on few XCDs and gain nothing.  CN_SIDE_CUS=0 turns the slow lane off.
            int ncu = prop.multiProcessorCount * 5 / 16;
            if (const char *e = getenv("CN_SIDE_CUS")) ncu = atoi(e);
            c->side_slow = masked_stream(device_id, ncu, prop.multiProcessorCount);
            c->side_cus = ncu;
        }
        c->opt = options_from_env();            // the one place the switches' environment variables are read for this context
        cn::t_opt = &c->opt;
        c->det = precision != CN_PREC_BF16;
        if (const char *e = getenv("CN_DETERMINISTIC")) c->det = atoi(e) != 0;
        if (const char *e = getenv("CN_NO_OVERLAP")) c->overlap = atoi(e) == 0;
        if (const char *e = getenv("CN_NO_ATTACHED_FORKS")) c->attach_forks = atoi(e) == 0;
        if (const char *e = getenv("CN_RPL")) c->rpl_override = atoi(e);   // experiments: force 4/8/16 sequences per workgroup
        // per call | running sums | sums over all ranks | 16 x {sum, count} partials + arrival counter of the loss sum that rides on
        // the output layer's backward launch (softmax_mcc_bwd_kernel)
        HIP_CHECK(hipMalloc((void **)&c->d_loss, (6 + 2 * 16 + 2) * sizeof(float)));
        HIP_CHECK(hipMemsetAsync(c->d_loss, 0, (6 + 2 * 16 + 2) * sizeof(float), c->stream));
        c->d_loss_acc = c->d_loss + 2;
        HIP_CHECK(hipMalloc((void **)&c->d_colpart, softmax_mcc_bwd_colpart_floats() * sizeof(float)));
        HIP_CHECK(hipMemsetAsync(c->d_colpart, 0, softmax_mcc_bwd_colpart_floats() * sizeof(float), c->stream));
        HIP_CHECK(hipMalloc((void **)&c->d_fault, sizeof(int)));
        HIP_CHECK(hipMemsetAsync(c->d_fault, 0, sizeof(int), c->stream));
    });
    if (rc != CN_OK) { delete c; return rc; }
    *out = c;
    return CN_OK;
}

int cn_ctx_destroy(cn_ctx *ctx)
{
    if (!ctx) return CN_OK;
    return guarded([&] {
        hipSetDevice(ctx->device);
        hipStreamSynchronize(ctx->stream);
        lstm_cluster_stream_gone(ctx->stream);
        if (ctx->side) { hipStreamSynchronize(ctx->side); hipStreamDestroy(ctx->side); }
        if (ctx->side_slow) hipStreamSynchronize(ctx->side_slow);     // shared per device, never destroyed: masked_stream()
        if (ctx->ev_sgd) hipEventDestroy(ctx->ev_sgd);
        if (ctx->ev_ext) hipEventDestroy(ctx->ev_ext);
        if (ctx->copy) { hipStreamSynchronize(ctx->copy); hipStreamDestroy(ctx->copy); }
        if (ctx->comm_stream) hipStreamSynchronize(ctx->comm_stream);
        if (ctx->comm) { (void)rccl().CommDestroy(ctx->comm); ctx->comm = nullptr; }
        if (ctx->ipc) { ipc_comm_destroy(ctx->ipc); ctx->ipc = nullptr; }
        if (ctx->comm_stream) { hipStreamDestroy(ctx->comm_stream); hipEventDestroy(ctx->ev_comm); hipEventDestroy(ctx->ev_comm_fork); }
        for (int i = 0; i < 2; ++i) {
            if (ctx->h_stage[i]) hipHostFree(ctx->h_stage[i]);
            if (ctx->d_stage[i]) hipFree(ctx->d_stage[i]);
            if (ctx->ev_up[i]) { hipEventDestroy(ctx->ev_up[i]); hipEventDestroy(ctx->ev_free[i]); }
        }
        std::vector<cn_layer *> ls = ctx->layers;
        for (cn_layer *l : ls) {
            for (void *p : l->owned) hipFree(p);
            if (l->ev_fork) { hipEventDestroy(l->ev_fork); hipEventDestroy(l->ev_join); }
            if (l->ev_pack) hipEventDestroy(l->ev_pack);
            delete l;
        }
        for (int k = 0; k < KC_COUNT; ++k) for (auto &sp : ctx->spans[k]) { hipEventDestroy(sp.a); hipEventDestroy(sp.b); }
        for (hipEvent_t e : ctx->free_events) hipEventDestroy(e);
        hipFree(ctx->pf.pat_raw); hipFree(ctx->pf.tcls); hipFree(ctx->d_colpart);
        hipFree(ctx->d_pat_raw); hipFree(ctx->d_tcls); hipFree(ctx->d_loss); hipFree(ctx->arena); hipFree(ctx->acc); hipFree(ctx->d_rowstat); hipFree(ctx->d_xch); hipFree(ctx->d_fault);
        if (ctx->own_stream) hipStreamDestroy(ctx->stream);
        if (cn::t_opt == &ctx->opt) cn::t_opt = nullptr;
        delete ctx;
    });
}

int cn_ctx_set_option(cn_ctx *ctx, const char *name, int value)
{
    if (!ctx || !name) { g_last_error = "cn_ctx_set_option: NULL argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        if (!strcmp(name, "deterministic")) { ctx->det = value != 0; return; }
        if (!strcmp(name, "overlap")) { ctx->overlap = value != 0; return; }
        if (option_set(ctx->opt, name, value)) return;
        throw cn_error(CN_ERR_BAD_ARG, std::string("cn_ctx_set_option: unknown option \"") + name + "\"");
    });
}

int cn_ctx_get_option(const cn_ctx *ctx, const char *name, int *value)
{
    if (!ctx || !name || !value) { g_last_error = "cn_ctx_get_option: NULL argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        if (!strcmp(name, "deterministic")) { *value = ctx->det ? 1 : 0; return; }
        if (!strcmp(name, "overlap")) { *value = ctx->overlap ? 1 : 0; return; }
        long v = 0;
        if (option_get(ctx->opt, name, &v)) { *value = (int)v; return; }
        throw cn_error(CN_ERR_BAD_ARG, std::string("cn_ctx_get_option: unknown option \"") + name + "\"");
    });
}

int cn_ctx_synchronize(cn_ctx *ctx)
{
    if (!ctx) { g_last_error = "cn_ctx_synchronize: ctx is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        enter(ctx);
        join_side(ctx);
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        HIP_CHECK(hipGetLastError());
        check_fault(ctx);
    });
}

int cn_ctx_join(cn_ctx *ctx)
{
    if (!ctx) { g_last_error = "cn_ctx_join: ctx is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] { enter(ctx); join_side(ctx); });
}

int cn_layer_join(cn_layer *layer)
{
    if (!layer) { g_last_error = "cn_layer_join: layer is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        cn_ctx *c = layer->ctx;
        enter(c);
        for (size_t i = 0; i < c->pending_joins.size(); ++i)
            if (c->pending_joins[i] == layer->ev_join) {
                HIP_CHECK(hipStreamWaitEvent(c->stream, layer->ev_join, 0));
                c->pending_joins.erase(c->pending_joins.begin() + i); c->pending_join_streams.erase(c->pending_join_streams.begin() + i);
                break;
            }
    });
}

void *cn_ctx_stream(cn_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int cn_layer_join_stream(cn_layer *layer, void *stream)
{
    if (!layer || !stream) { g_last_error = "cn_layer_join_stream: NULL argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        enter(layer->ctx);
        stream_wait_layer(layer, (hipStream_t)stream);
    });
}

// ---------------------------------------------------------------------------------------------
// data-parallel training (RCCL)
// ---------------------------------------------------------------------------------------------
int cn_comm_unique_id(char *id)
{
    if (!id) { g_last_error = "cn_comm_unique_id: id is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        static_assert(sizeof(ncclUniqueId) == CN_COMM_ID_BYTES, "CN_COMM_ID_BYTES must match ncclUniqueId");
        if (ipc_backend_selected()) { ipc_unique_id(id, CN_COMM_ID_BYTES); return; }
        ncclUniqueId u;
        RCCL_CHECK(rccl().GetUniqueId(&u));
        memcpy(id, &u, sizeof(u));
    });
}

int cn_comm_init(cn_ctx *ctx, const char *id, int rank, int world)
{
    if (!ctx || !id) { g_last_error = "cn_comm_init: NULL argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        if (world < 1 || rank < 0 || rank >= world) throw cn_error(CN_ERR_BAD_ARG, "cn_comm_init: rank " + std::to_string(rank) + " outside world of " + std::to_string(world));
        if (ctx->has_comm()) throw cn_error(CN_ERR_STATE, "cn_comm_init: this context already has a communicator");
        enter(ctx);
        if (!ctx->comm_stream) {
            HIP_CHECK(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_comm, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_comm_fork, hipEventDisableTiming));
        }
        if (ipc_backend_selected()) {
            char fallback[CN_COMM_ID_BYTES];
            bool p2p_ok = true;
            try {
                ctx->ipc = ipc_comm_create(id, rank, world);
                // p2p: first contact.  Regions mapped, both forms of the exchange on a known bucket, one shared verdict.
                if (ipc_comm_is_p2p(ctx->ipc))
                    p2p_ok = ipc_comm_p2p_selfcheck(ctx->ipc, ctx->comm_stream, [](char *out) {
                        ncclUniqueId u;
                        rccl_check(rccl().GetUniqueId(&u), "ncclGetUniqueId");
                        memcpy(out, &u, sizeof(u));
                    }, fallback);
            }
            catch (const cn_error &) { if (ctx->ipc) { ipc_comm_mark_failed(ctx->ipc); ipc_comm_destroy(ctx->ipc); ctx->ipc = nullptr; } throw; }
            catch (const std::exception &e) {
                if (ctx->ipc) { ipc_comm_mark_failed(ctx->ipc); ipc_comm_destroy(ctx->ipc); ctx->ipc = nullptr; }
                throw cn_error(CN_ERR_COMM, e.what());
            }
            if (!p2p_ok) {
                // some rank saw a wrong sum or a time-out through a peer mapping: nobody trains on this backend.  Every rank got the
                // same verdict and the id rank 0 made: the job goes on over RCCL and says so.
                ipc_comm_destroy(ctx->ipc); ctx->ipc = nullptr;
                fprintf(stderr, "cn_comm_init: rank %d of %d: CN_COMM_BACKEND=p2p failed its first-contact self-check; falling back to RCCL\n", rank, world);
                ncclUniqueId u;
                memcpy(&u, fallback, sizeof(u));
                RCCL_CHECK(rccl().CommInitRank(&ctx->comm, world, u, rank));
            }
        } else {
            ncclUniqueId u;
            memcpy(&u, id, sizeof(u));
            RCCL_CHECK(rccl().CommInitRank(&ctx->comm, world, u, rank));
        }
        ctx->comm_rank = rank; ctx->comm_world = world;
    });
}

int cn_comm_destroy(cn_ctx *ctx)
{
    if (!ctx) { g_last_error = "cn_comm_destroy: ctx is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        if (!ctx->has_comm()) return;
        enter(ctx);
        HIP_CHECK(hipStreamSynchronize(ctx->comm_stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        std::string failure;
        if (ctx->ipc) {
            // p2p: a poll that timed out on the device is reported here at the latest (the resources go either way)
            try { ipc_comm_check(ctx->ipc, ctx->comm_stream); }
            catch (const std::exception &e) { ipc_comm_mark_failed(ctx->ipc); failure = e.what(); }
            ipc_comm_destroy(ctx->ipc); ctx->ipc = nullptr;
        }
        else RCCL_CHECK(rccl().CommDestroy(ctx->comm));
        ctx->comm = nullptr; ctx->comm_world = 0; ctx->comm_rank = 0; ctx->comm_pending = false;
        if (!failure.empty()) throw cn_error(CN_ERR_COMM, failure);
    });
}

int cn_comm_info(const cn_ctx *ctx, int *rank, int *world)
{
    if (!ctx) { g_last_error = "cn_comm_info: ctx is NULL"; return CN_ERR_BAD_ARG; }
    if (rank) *rank = ctx->comm_rank;
    if (world) *world = ctx->comm_world;
    return CN_OK;
}

const char *cn_comm_backend(const cn_ctx *ctx, int64_t *exchanges)
{
    if (exchanges) *exchanges = ctx ? ctx->comm_exchanges : 0;
    if (!ctx || !ctx->has_comm()) return "";
    return ctx->comm ? "rccl" : ipc_comm_is_p2p(ctx->ipc) ? "p2p" : "ipc";
}

// p2p: a wait inside an exchange kernel timed out (host-mapped word, no synchronisation): no update on top of a NaN gradient
static void comm_check_fast(cn_ctx *ctx)
{
    if (!ctx->ipc) return;
    try { ipc_comm_check_fast(ctx->ipc); }
    catch (const std::exception &e) { throw cn_error(CN_ERR_COMM, e.what()); }
}

// the ipc / p2p backends' exchange (ipc: host-blocking); a failure marks the segment so that the peers leave their barriers at once
static void ipc_reduce(cn_ctx *ctx, float *buf, size_t n)
{
    try { ipc_allreduce(ctx->ipc, buf, n, ctx->comm_stream, ctx->total); }
    catch (const std::exception &e) { ipc_comm_mark_failed(ctx->ipc); throw cn_error(CN_ERR_COMM, e.what()); }
}

int cn_allreduce_grads(cn_ctx *ctx, cn_layer *const *layers, int n)
{
    if (!ctx || n < 0 || (n > 0 && !layers)) { g_last_error = "cn_allreduce_grads: bad argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        require_comm(ctx, "cn_allreduce_grads");
        enter(ctx);
        finalize(ctx);
        // CN_COMM_TEST_DOUBLE (tests/test_gpu_parallel.py): the collective is replaced by a kernel on the communication stream
        // that doubles the gradient -- what a two-rank all-reduce of equal shards does --, so that the ORDER of gradient work,
        // exchange and update can be checked on a one-GPU box: a reduction that starts early or an update that does not wait
        // shows in the trained weights
        const bool test_double = opt().comm_test_double;
        if (test_double && ctx->comm_world > 1)
            throw cn_error(CN_ERR_STATE, "cn_allreduce_grads: CN_COMM_TEST_DOUBLE is set in a job of more than one rank (it replaces the "
                                         "gradient exchange by a stand-in and is for one-rank ordering tests only)");
        if (n == 0) {
            // the whole arena in one exchange: every gradient GEMM first, then fork the communication stream
            join_side(ctx);
            HIP_CHECK(hipEventRecord(ctx->ev_comm_fork, ctx->stream));
            HIP_CHECK(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_comm_fork, 0));
            float *g = ctx->arena + ctx->total;
            Timed tm(ctx, KC_COMM, ctx->comm_stream);
            ++ctx->comm_exchanges;
            if (test_double) launch_scale(ctx->comm_stream, g, ctx->total, 2.0f);
            else if (ctx->ipc) ipc_reduce(ctx, g, ctx->total);
            else if (ctx->total) RCCL_CHECK(rccl().AllReduce(g, g, ctx->total, ncclFloat32, ncclSum, ctx->comm, ctx->comm_stream));
        } else {
            for (int i = 0; i < n; ++i) {
                cn_layer *l = layers[i];
                if (!l || l->ctx != ctx || !l->trainable) throw cn_error(CN_ERR_BAD_ARG, "cn_allreduce_grads: not a trainable layer of this context");
                stream_wait_layer(l, ctx->comm_stream);
                Timed tm(ctx, KC_COMM, ctx->comm_stream);          // (events on the communication stream: the exchange itself, not its wait)
                ++ctx->comm_exchanges;
                if (test_double) launch_scale(ctx->comm_stream, l->wu, (size_t)l->nw, 2.0f);
                else if (ctx->ipc) ipc_reduce(ctx, l->wu, (size_t)l->nw);
                else RCCL_CHECK(rccl().AllReduce(l->wu, l->wu, (size_t)l->nw, ncclFloat32, ncclSum, ctx->comm, ctx->comm_stream));
                // armed update: the layer's step follows its reduction on the communication stream
                if (ctx->armed && !l->updated) launch_layer_update(ctx->comm_stream, l, 1, ctx->arm_lr, ctx->arm_mom, nullptr);
            }
        }
        HIP_CHECK(hipEventRecord(ctx->ev_comm, ctx->comm_stream));
        ctx->comm_pending = true;
    });
}

// ---------------------------------------------------------------------------------------------
// layers
// ---------------------------------------------------------------------------------------------
int cn_layer_create(cn_ctx *ctx, cn_layer_kind kind, cn_layer *preceding, int size, float bias,
                    int parallel_sequences, int max_seq_length, cn_layer **out)
{
    if (!ctx || !out) { g_last_error = "cn_layer_create: NULL argument"; return CN_ERR_BAD_ARG; }
    *out = nullptr;
    cn_layer *l = nullptr;
    int rc = guarded([&] {
        enter(ctx);
        if (size <= 0) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_create: layer size must be positive");
        if (kind == CN_LAYER_INPUT) {
            if (preceding) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_create: the input layer has no preceding layer");
            if (parallel_sequences <= 0 || max_seq_length <= 0)
                throw cn_error(CN_ERR_BAD_ARG, "cn_layer_create: parallel_sequences and max_seq_length must be positive");
            if (ctx->PS) throw cn_error(CN_ERR_STATE, "cn_layer_create: this context already has an input layer");
        } else {
            if (!preceding) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_create: preceding layer required");
            if (preceding->ctx != ctx) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_create: preceding layer belongs to another context");
            if (preceding->post) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_create: a post output layer must be the last layer");
        }
        l = new cn_layer;
        l->ctx = ctx; l->kind = kind; l->prev = preceding; l->size = size; l->bias = bias;
        l->PS = preceding ? preceding->PS : parallel_sequences;
        l->maxT = preceding ? preceding->maxT : max_seq_length;
        if (!preceding) {
            // sequences per workgroup of the recurrent kernels: the fewest (4, 8, 16) that still give at most
            // one workgroup per CU for a bidirectional layer; PS is padded to a whole number of groups
            const int cus = ctx->num_cus;
            int rpl = ctx->rpl_override ? ctx->rpl_override : (2 * ((l->PS + 3) / 4) <= cus ? 1 : (2 * ((l->PS + 7) / 8) <= cus ? 2 : 4));
            if (rpl != 1 && rpl != 2 && rpl != 4) throw cn_error(CN_ERR_BAD_ARG, "CN_RPL must be 1, 2 or 4");
            ctx->rpl = rpl;
            ctx->PSp = round_up(l->PS, 4 * rpl);
        }
        l->PSp = ctx->PSp;
        const size_t maxN = l->maxN(), e = ctx->esz();
        if (preceding) { l->P = preceding->size; l->Pp = preceding->Lp; }

        switch (kind) {
        case CN_LAYER_INPUT: {
            l->Lp = round_up(size, 32);
            l->stage_in = (float *)dalloc(l, maxN * size * sizeof(float));
            l->out_op = dalloc(l, maxN * l->Lp * e);
            ctx->PS = l->PS; ctx->maxT = l->maxT;
            {   // pattern types, with guard steps of PATTYPE_NONE on both sides (dalloc_guarded)
                const size_t guard = (size_t)CN_GUARD_STEPS * ctx->PSp;
                HIP_CHECK(hipMalloc((void **)&ctx->d_pat_raw, cn_ctx::pat_bytes(maxN, guard)));
                HIP_CHECK(hipMemsetAsync(ctx->d_pat_raw, 0, cn_ctx::pat_bytes(maxN, guard), ctx->stream));   // pad slots: PATTYPE_NONE forever
                ctx->pat_maxN = maxN; ctx->pat_guard = guard;
                ctx->d_pat = ctx->d_pat_raw + guard;
            }
            HIP_CHECK(hipMalloc((void **)&ctx->d_tcls, maxN * sizeof(int)));
            HIP_CHECK(hipMemsetAsync(ctx->d_tcls, 0xFF, maxN * sizeof(int), ctx->stream));     // pad slots: target class -1
            break; }
        case CN_LAYER_LSTM:
        case CN_LAYER_BLSTM: {
            if (ctx->finalized) throw cn_error(CN_ERR_STATE, "cn_layer_create: parameter arena already laid out");
            l->lstm = true; l->trainable = true;
            l->dirs = (kind == CN_LAYER_BLSTM) ? 2 : 1;
            if (l->dirs == 2 && size % 2 != 0)
                throw cn_error(CN_ERR_SHAPE, "Cannot create a bidirectional layer with an odd layer size");   // LstmLayer.cu:528-529
            l->H = size / l->dirs; l->Hp = pad_units(l->H);
            // bf16 mode: widths between the register-resident shapes (<= 192) and a cluster shape (256, 512) are padded
            // up to the cluster shape when the cluster fits the chip: zero units cost GEMM work and memory, streaming
            // W_rec from L2 every time step costs an order of magnitude more
            if (!ctx->f32 && l->Hp > 192 && l->Hp < 512 && l->Hp != 256) {
                const int hc = l->Hp < 256 ? 256 : 512;
                if (lstm_cluster_xch_bytes(P_BF16, hc, l->dirs, ctx->PSp, ctx->rpl, ctx->num_cus) > 0) l->Hp = hc;
            }
            // split-bf16 mode: the same between the register-resident shapes (<= 128) and its one cluster shape (256)
            if (ctx->prec == P_X3 && l->Hp > 128 && l->Hp < 256 &&
                lstm_cluster_xch_bytes(P_X3, 256, l->dirs, ctx->PSp, ctx->rpl, ctx->num_cus) > 0) l->Hp = 256;
            l->Lp = l->dirs * l->Hp;
            const size_t R = (size_t)l->dirs * 4 * l->Hp;
            // the recurrent kernels address every per-frame buffer with 32-bit byte offsets from its base
            if (maxN * R * sizeof(float) >= (1ull << 32))
                throw cn_error(CN_ERR_SHAPE, "cn_layer_create: parallel_sequences * max_seq_length * 16 * layer size must stay below 4 GiB "
                                             "per LSTM layer; use fewer parallel sequences");
            l->nw = size * (4 * (l->P + 1) + 4 * l->H + 3);                                            // LstmLayer.cu:525
            l->out_op = dalloc(l, maxN * l->Lp * e);
            l->err = (float *)dalloc_guarded(l, maxN * l->Lp * sizeof(float), (size_t)ctx->PSp * l->Lp * sizeof(float));
            l->acts = (float *)dalloc_guarded(l, maxN * R * sizeof(float), (size_t)ctx->PSp * R * sizeof(float));
            if (ctx->prec == P_BF16) l->pre16 = dalloc_guarded(l, maxN * R * 2, (size_t)ctx->PSp * R * 2);
            l->cell = (float *)dalloc_guarded(l, maxN * l->Lp * sizeof(float), (size_t)ctx->PSp * l->Lp * sizeof(float));
            l->th = (float *)dalloc_guarded(l, maxN * l->Lp * sizeof(float), (size_t)ctx->PSp * l->Lp * sizeof(float));
            l->delta_op = dalloc(l, maxN * R * e);
            l->Win = dalloc(l, R * l->Pp * e); l->WinT = dalloc(l, R * l->Pp * e);
            l->Wrec = dalloc(l, R * l->Hp * e); l->WrecT = dalloc(l, R * l->Hp * e);
            l->bias_p = (float *)dalloc(l, R * sizeof(float));
            l->peep_p = (float *)dalloc(l, (size_t)l->dirs * 3 * l->Hp * sizeof(float));
            l->grad_block_floats = R * l->Pp + R * l->Hp + R + (size_t)l->dirs * 3 * l->Hp;
            l->grad_block = (float *)dalloc(l, l->grad_block_floats * sizeof(float));
            l->dWin = l->grad_block; l->dWrec = l->dWin + R * l->Pp; l->dbias = l->dWrec + R * l->Hp; l->dpeep = l->dbias + R;
            {   // exchange buffer of the multi-CU cluster kernels (layers whose W_rec exceeds one CU)
                const size_t xb = lstm_cluster_xch_bytes(ctx->prec, l->Hp, l->dirs, ctx->PSp, ctx->rpl, ctx->num_cus);
                if (xb > ctx->xch_bytes) {
                    if (ctx->d_xch) HIP_CHECK(hipFree(ctx->d_xch));
                    HIP_CHECK(hipMalloc((void **)&ctx->d_xch, xb));
                    HIP_CHECK(hipMemsetAsync(ctx->d_xch, 0, xb, ctx->stream));
                    ctx->xch_bytes = xb; ctx->xch_epoch = 0;
                }
            }
            break; }
        case CN_LAYER_FF_TANH:
        case CN_LAYER_FF_LOGISTIC:
        case CN_LAYER_FF_IDENTITY:
        case CN_LAYER_SOFTMAX: {
            if (ctx->finalized) throw cn_error(CN_ERR_STATE, "cn_layer_create: parameter arena already laid out");
            l->trainable = true;
            l->Lp = round_up(size, 32);
            l->nw = size * (l->P + 1);                                                                   // FeedForwardLayer.cu:130
            l->out_f32 = (float *)dalloc(l, maxN * l->Lp * sizeof(float));
            l->out_op = ctx->f32 ? (void *)l->out_f32 : dalloc(l, maxN * l->Lp * e);
            l->err = (float *)dalloc(l, maxN * l->Lp * sizeof(float));
            l->delta_op = ctx->f32 ? (void *)l->err : dalloc(l, maxN * l->Lp * e);
            l->Win = dalloc(l, (size_t)l->Lp * l->Pp * e); l->WinT = dalloc(l, (size_t)l->Lp * l->Pp * e);
            l->bias_p = (float *)dalloc(l, l->Lp * sizeof(float));
            l->grad_block_floats = (size_t)l->Lp * l->Pp + l->Lp;
            l->grad_block = (float *)dalloc(l, l->grad_block_floats * sizeof(float));
            l->dWin = l->grad_block; l->dbias = l->dWin + (size_t)l->Lp * l->Pp;
            if (kind == CN_LAYER_SOFTMAX && softmax_fwd_can_be_lazy(size)) l->sm_stat = (float *)dalloc(l, maxN * 2 * sizeof(float));
            break; }
        case CN_LAYER_SSE:
        case CN_LAYER_WEIGHTEDSSE:
        case CN_LAYER_SSE_MASK:
        case CN_LAYER_CE:
        case CN_LAYER_RMSE:
        case CN_LAYER_BINARY_CLASSIFICATION:
        case CN_LAYER_MULTICLASS_CLASSIFICATION: {
            if (!preceding->trainable) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_create: a post output layer needs a trainable preceding layer");
            if (preceding->lstm) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_create: post output layers directly after an LSTM layer are not supported");
            const bool paired = kind == CN_LAYER_WEIGHTEDSSE || kind == CN_LAYER_SSE_MASK;
            // PostOutputLayer.cpp:58-59; the weighted layers take (target, weight) pairs: WeightedSsePostOutputLayer.cu:103, SseMaskPostOutputLayer.cu:103
            if (paired ? size != 2 * preceding->size : size != preceding->size)
                throw cn_error(CN_ERR_SHAPE, "Size mismatch: " + std::to_string(size) + " vs. " + std::to_string(preceding->size));
            if (kind == CN_LAYER_BINARY_CLASSIFICATION && size != 1)                                      // BinaryClassificationLayer.cu:123-124
                throw cn_error(CN_ERR_SHAPE, "The binary classification post output layer cannot be used for an output layer size != 1");
            if (kind == CN_LAYER_MULTICLASS_CLASSIFICATION && size == 1)                                  // MulticlassClassificationLayer.cu:146-147
                throw cn_error(CN_ERR_SHAPE, "The multiclass classification post output layer cannot be used for an output layer size of 1");
            l->post = true; l->Lp = preceding->Lp;
            // (every post output layer: the per-pattern terms of its error pass through here and are summed in a fixed order)
            if (kind == CN_LAYER_MULTICLASS_CLASSIFICATION && !ctx->d_rowstat)
                HIP_CHECK(hipMalloc((void **)&ctx->d_rowstat, maxN * 2 * sizeof(float)));
            if (kind != CN_LAYER_MULTICLASS_CLASSIFICATION) {
                l->targets = (float *)dalloc(l, maxN * size * sizeof(float));
                if (!ctx->d_rowstat) HIP_CHECK(hipMalloc((void **)&ctx->d_rowstat, maxN * 2 * sizeof(float)));
            }
            break; }
        default:
            throw cn_error(CN_ERR_BAD_ARG, "cn_layer_create: unknown layer kind");
        }
        if (preceding && l->trainable) preceding->has_follower = true;
        ctx->layers.push_back(l);
    });
    if (rc != CN_OK) { if (l) { for (void *p : l->owned) hipFree(p); delete l; } return rc; }
    *out = l;
    return CN_OK;
}

int cn_layer_destroy(cn_layer *layer)
{
    if (!layer) return CN_OK;
    return guarded([&] {
        cn_ctx *c = layer->ctx;
        enter(c);
        join_side(c);                                   // side-stream work of this layer may still be in flight
        HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->side) HIP_CHECK(hipStreamSynchronize(c->side));      // (operand copies being rebuilt: ev_pack_last may be this layer's)
        for (cn_layer *o : c->layers) o->pack_pending = false;
        c->ev_pack_last = nullptr;
        for (void *p : layer->owned) hipFree(p);
        if (layer->ev_fork) { hipEventDestroy(layer->ev_fork); hipEventDestroy(layer->ev_join); }
        if (layer->ev_pack) hipEventDestroy(layer->ev_pack);
        if (c->rowstat_of == layer) c->rowstat_of = nullptr;
        for (size_t i = 0; i < c->layers.size(); ++i)
            if (c->layers[i] == layer) { c->layers.erase(c->layers.begin() + i); break; }
        delete layer;
    });
}

int cn_layer_size(const cn_layer *layer) { return layer ? layer->size : CN_ERR_BAD_ARG; }
int cn_layer_kind_of(const cn_layer *layer) { return layer ? (int)layer->kind : CN_ERR_BAD_ARG; }
int cn_layer_weight_count(const cn_layer *layer) { return layer ? layer->nw : CN_ERR_BAD_ARG; }

// ---------------------------------------------------------------------------------------------
// fraction
// ---------------------------------------------------------------------------------------------
static void check_fraction(cn_ctx *ctx, cn_layer *input, cn_layer *post_output, const cn_fraction *f)
{
    if (input->kind != CN_LAYER_INPUT) throw cn_error(CN_ERR_BAD_ARG, "cn_fraction_load: `input` is not an input layer");
    if (f->input_pattern_size != input->size)                                                     // InputLayer.cpp:52-55
        throw cn_error(CN_ERR_SHAPE, "Input layer size of " + std::to_string(input->size) +
                       " != data input pattern size of " + std::to_string(f->input_pattern_size));
    if (post_output) {
        if (!post_output->post) throw cn_error(CN_ERR_BAD_ARG, "cn_fraction_load: `post_output` is not a post output layer");
        if (f->output_pattern_size != post_output->size)                                          // PostOutputLayer.cpp:70-73
            throw cn_error(CN_ERR_SHAPE, "Output layer size of " + std::to_string(post_output->size) +
                           " != data target pattern size of " + std::to_string(f->output_pattern_size));
    }
    const int T = f->max_seq_length;
    if (T <= 0 || T > ctx->maxT) throw cn_error(CN_ERR_SHAPE, "cn_fraction_load: max_seq_length " + std::to_string(T) +
                                                " outside (0, " + std::to_string(ctx->maxT) + "]");
    if (f->min_seq_length < 0 || f->min_seq_length > T) throw cn_error(CN_ERR_SHAPE, "cn_fraction_load: bad min_seq_length");
    if (!f->pat_types || !f->inputs) throw cn_error(CN_ERR_BAD_ARG, "cn_fraction_load: pat_types / inputs missing");
}
// Host buffers of a fraction -> pinned staging area -> ONE contiguous upload on the copy stream ([patTypes | classes or targets |
// inputs]; the two staging areas alternate).  Returns the area and where its parts lie; ev_up[slot] completes with the upload.
struct Staged { int slot; char *dv; size_t b_pat, b_tgt_al, W; bool cls; };
static Staged stage_upload(cn_ctx *ctx, cn_layer *input, cn_layer *post_output, const cn_fraction *f)
{
    const int T = f->max_seq_length;
    const size_t PS = ctx->PS;
    const bool cls = post_output && (post_output->kind == CN_LAYER_MULTICLASS_CLASSIFICATION || post_output->kind == CN_LAYER_BINARY_CLASSIFICATION);
    const size_t rows = (size_t)T * PS;
    const size_t W = post_output ? (size_t)post_output->size : 0;
    const size_t b_pat = (rows + 15) & ~(size_t)15;
    const size_t b_tgt = !post_output ? 0 : (cls ? rows * sizeof(int) : rows * W * sizeof(float));
    const size_t b_tgt_al = (b_tgt + 15) & ~(size_t)15;
    const size_t b_in = rows * (size_t)input->size * sizeof(float);
    const size_t need = b_pat + b_tgt_al + b_in;
    if (need > ctx->stage_bytes || !ctx->h_stage[0]) {
        HIP_CHECK(hipDeviceSynchronize());
        for (int i = 0; i < 2; ++i) {
            if (ctx->h_stage[i]) HIP_CHECK(hipHostFree(ctx->h_stage[i]));
            if (ctx->d_stage[i]) HIP_CHECK(hipFree(ctx->d_stage[i]));
            ctx->stage_used[i] = false;
        }
        const size_t maxrows = (size_t)ctx->maxT * PS;
        size_t cap = ((maxrows + 15) & ~(size_t)15) + ((maxrows * std::max(W * sizeof(float), sizeof(int)) + 15) & ~(size_t)15) + maxrows * (size_t)input->size * sizeof(float);
        if (cap < need) cap = need;
        for (int i = 0; i < 2; ++i) {
            HIP_CHECK(hipHostMalloc((void **)&ctx->h_stage[i], cap, hipHostMallocDefault));
            HIP_CHECK(hipMalloc((void **)&ctx->d_stage[i], cap));
            if (!ctx->ev_up[i]) { HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_up[i], hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_free[i], hipEventDisableTiming)); }
        }
        if (!ctx->copy) HIP_CHECK(hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking));
        ctx->stage_bytes = cap;
    }
    const int slot = (int)(ctx->upload_idx++ & 1u);
    if (ctx->stage_used[slot]) HIP_CHECK(hipEventSynchronize(ctx->ev_free[slot]));    // the re-layout kernel two fractions ago has read it
    char *h = ctx->h_stage[slot], *dv = ctx->d_stage[slot];
    memcpy(h, f->pat_types, rows);
    if (post_output) memcpy(h + b_pat, cls ? (const void *)f->target_classes : (const void *)f->targets, b_tgt);
    memcpy(h + b_pat + b_tgt_al, f->inputs, b_in);
    HIP_CHECK(hipMemcpyAsync(dv, h, need, hipMemcpyHostToDevice, ctx->copy));
    HIP_CHECK(hipEventRecord(ctx->ev_up[slot], ctx->copy));
    return Staged{slot, dv, b_pat, b_tgt_al, W, cls};
}

static bool same_fraction(const cn_fraction &a, const cn_fraction &b)
{
    return a.max_seq_length == b.max_seq_length && a.min_seq_length == b.min_seq_length && a.num_sequences == b.num_sequences &&
           a.input_pattern_size == b.input_pattern_size && a.output_pattern_size == b.output_pattern_size &&
           a.pat_types == b.pat_types && a.inputs == b.inputs && a.target_classes == b.target_classes && a.targets == b.targets;
}
static int fraction_load(cn_ctx *ctx, cn_layer *input, cn_layer *post_output, const cn_fraction *f, bool resident)
{
    if (!ctx || !input || !f) { g_last_error = "cn_fraction_load: NULL argument"; return CN_ERR_BAD_ARG; }
    const hipMemcpyKind kind = resident ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    return guarded([&] {
        enter(ctx);
        check_fraction(ctx, input, post_output, f);
        const int T = f->max_seq_length;
        finalize(ctx);
        join_side(ctx);                      // gradient GEMMs of the previous fraction still read the activations
        flush_loss(ctx);                     // (a deferred loss sum counts the rows of the fraction that is being replaced)
        const size_t PS = ctx->PS, PSp = ctx->PSp;
        const size_t N = (size_t)T * PSp;
        Timed tm(ctx, KC_OTHER);
        if (resident) {      // everything is in HBM already: one kernel re-lays it out
            const bool cls = post_output && (post_output->kind == CN_LAYER_MULTICLASS_CLASSIFICATION || post_output->kind == CN_LAYER_BINARY_CLASSIFICATION);
            if (post_output && cls && !f->target_classes) throw cn_error(CN_ERR_BAD_ARG, "cn_fraction_load: target_classes missing");
            if (post_output && !cls && !f->targets) throw cn_error(CN_ERR_BAD_ARG, "cn_fraction_load: targets missing");
            cn_ctx::Prefetch &pf = ctx->pf;
            const bool hit = pf.valid && pf.launched && !pf.host && pf.input == input && pf.post == post_output && same_fraction(pf.f, *f);
            pf.valid = false;          // a prefetch serves the very next load or nothing
            if (hit) {
                ++pf.hits;
                // the side stream has re-laid this fraction out already (join_side above ordered this stream behind it): the
                // alternate buffers become the current ones
                std::swap(ctx->d_pat, pf.pat); std::swap(ctx->d_pat_raw, pf.pat_raw); std::swap(ctx->d_tcls, pf.tcls);
                std::swap(input->out_op, pf.in_op);
                if (post_output && post_output->targets) std::swap(post_output->targets, pf.targets);
            } else {
            launch_fraction_load(ctx->stream, ctx->f32, T, (int)PS, (int)PSp, f->pat_types, ctx->d_pat,
                                 cls ? f->target_classes : nullptr, ctx->d_tcls,
                                 (post_output && !cls) ? f->targets : nullptr, post_output ? post_output->targets : nullptr,
                                 post_output ? post_output->size : 0, f->inputs, input->size, input->out_op, input->Lp, ctx->rowmap_of(ctx->d_pat_raw), (int)ctx->pat_maxN, f->min_seq_length);
            if (post_output && post_output->kind == CN_LAYER_BINARY_CLASSIFICATION)
                launch_classes_to_targets(ctx->stream, ctx->d_tcls, post_output->targets, (int)N);
            }
            ctx->T = T; ctx->Tmin = f->min_seq_length; ctx->N = (int)N; ctx->est_real = f->num_sequences * ((f->min_seq_length + T + 1) / 2); ctx->Next = T * (int)PS; ctx->numSeqs = f->num_sequences;
            ctx->loaded = true;
            return;
        }
        {   // cn_fraction_prefetch announced exactly this fraction and the side stream has re-laid it out (join_side above ordered
            // this stream behind it): the alternate buffers become the current ones, nothing is copied or launched here
            cn_ctx::Prefetch &pf = ctx->pf;
            const bool hit = pf.valid && pf.launched && pf.host && pf.input == input && pf.post == post_output && same_fraction(pf.f_host, *f);
            pf.valid = false;          // a prefetch serves the very next load or nothing
            if (hit) {
                ++pf.hits;
                std::swap(ctx->d_pat, pf.pat); std::swap(ctx->d_pat_raw, pf.pat_raw); std::swap(ctx->d_tcls, pf.tcls);
                std::swap(input->out_op, pf.in_op);
                if (post_output && post_output->targets) std::swap(post_output->targets, pf.targets);
                ctx->T = T; ctx->Tmin = f->min_seq_length; ctx->N = (int)N; ctx->est_real = f->num_sequences * ((f->min_seq_length + T + 1) / 2); ctx->Next = T * (int)PS; ctx->numSeqs = f->num_sequences;
                ctx->loaded = true;
                return;
            }
        }
        if (ctx->overlap) {
            // Host buffers: pack [patTypes | classes or targets | inputs] into pinned memory, ONE contiguous upload
            // on the copy stream (it runs while the previous fraction still computes: the staging areas alternate),
            // then the same re-layout kernel as the resident path.  (Strided 2-D copies from pageable memory cost
            // one transfer per time step: 3.5 ms per fraction of 2.3 MB, twice the whole training step.)
            const bool cls = post_output && (post_output->kind == CN_LAYER_MULTICLASS_CLASSIFICATION || post_output->kind == CN_LAYER_BINARY_CLASSIFICATION);
            if (post_output && cls && !f->target_classes) throw cn_error(CN_ERR_BAD_ARG, "cn_fraction_load: target_classes missing");
            if (post_output && !cls && !f->targets) throw cn_error(CN_ERR_BAD_ARG, "cn_fraction_load: targets missing");
            const Staged sg = stage_upload(ctx, input, post_output, f);
            const int slot = sg.slot; char *dv = sg.dv; const size_t b_pat = sg.b_pat, b_tgt_al = sg.b_tgt_al, W = sg.W;
            HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_up[slot], 0));
            launch_fraction_load(ctx->stream, ctx->f32, T, (int)PS, (int)PSp, dv, ctx->d_pat,
                                 cls ? (const int *)(dv + b_pat) : nullptr, ctx->d_tcls,
                                 (post_output && !cls) ? (const float *)(dv + b_pat) : nullptr, post_output ? post_output->targets : nullptr,
                                 (int)W, (const float *)(dv + b_pat + b_tgt_al), input->size, input->out_op, input->Lp, ctx->rowmap_of(ctx->d_pat_raw), (int)ctx->pat_maxN, f->min_seq_length);
            HIP_CHECK(hipEventRecord(ctx->ev_free[slot], ctx->stream));
            ctx->stage_used[slot] = true;
            if (post_output && post_output->kind == CN_LAYER_BINARY_CLASSIFICATION)
                launch_classes_to_targets(ctx->stream, ctx->d_tcls, post_output->targets, (int)N);
            ctx->T = T; ctx->Tmin = f->min_seq_length; ctx->N = (int)N; ctx->est_real = f->num_sequences * ((f->min_seq_length + T + 1) / 2); ctx->Next = T * (int)PS; ctx->numSeqs = f->num_sequences;
            ctx->loaded = true;
            return;
        }
        // CN_NO_OVERLAP: strided copies [T][PS] -> [T][PSp]: pad slots keep their permanent NONE / -1 / 0 contents
        HIP_CHECK(hipMemcpy2DAsync(ctx->d_pat, PSp, f->pat_types, PS, PS, T, kind, ctx->stream));
        launch_rowmap(ctx->stream, ctx->d_pat, (int)N, ctx->rowmap_of(ctx->d_pat_raw), (int)ctx->pat_maxN, f->min_seq_length * (int)PSp);
        const size_t irow = (size_t)input->size * sizeof(float);
        HIP_CHECK(hipMemcpy2DAsync(input->stage_in, PSp * irow, f->inputs, PS * irow, PS * irow, T, kind, ctx->stream));
        if (post_output) {
            if (post_output->kind == CN_LAYER_MULTICLASS_CLASSIFICATION || post_output->kind == CN_LAYER_BINARY_CLASSIFICATION) {
                if (!f->target_classes) throw cn_error(CN_ERR_BAD_ARG, "cn_fraction_load: target_classes missing");
                HIP_CHECK(hipMemcpy2DAsync(ctx->d_tcls, PSp * sizeof(int), f->target_classes, PS * sizeof(int), PS * sizeof(int), T, kind, ctx->stream));
                if (post_output->kind == CN_LAYER_BINARY_CLASSIFICATION)
                    launch_classes_to_targets(ctx->stream, ctx->d_tcls, post_output->targets, (int)N);
            } else {
                if (!f->targets) throw cn_error(CN_ERR_BAD_ARG, "cn_fraction_load: targets missing");
                const size_t trow = (size_t)post_output->size * sizeof(float);
                HIP_CHECK(hipMemcpy2DAsync(post_output->targets, PSp * trow, f->targets, PS * trow, PS * trow, T, kind, ctx->stream));
            }
        }
        launch_pad_convert(ctx->stream, ctx->f32, input->stage_in, (int)N, input->size, input->out_op, input->Lp);
        ctx->T = T; ctx->Tmin = f->min_seq_length; ctx->N = (int)N; ctx->est_real = f->num_sequences * ((f->min_seq_length + T + 1) / 2); ctx->Next = T * (int)PS; ctx->numSeqs = f->num_sequences;
        ctx->loaded = true;
        if (!resident) HIP_CHECK(hipStreamSynchronize(ctx->stream));     // the caller may reuse its host buffers on return
    });
}

int cn_fraction_load(cn_ctx *ctx, cn_layer *input, cn_layer *post_output, const cn_fraction *f)
{
    return fraction_load(ctx, input, post_output, f, false);
}
int cn_fraction_load_resident(cn_ctx *ctx, cn_layer *input, cn_layer *post_output, const cn_fraction *f)
{
    return fraction_load(ctx, input, post_output, f, true);
}
static int fraction_prefetch(cn_ctx *ctx, cn_layer *input, cn_layer *post_output, const cn_fraction *f, bool host);
int cn_fraction_prefetch_resident(cn_ctx *ctx, cn_layer *input, cn_layer *post_output, const cn_fraction *f)
{
    return fraction_prefetch(ctx, input, post_output, f, false);
}
int cn_fraction_prefetch(cn_ctx *ctx, cn_layer *input, cn_layer *post_output, const cn_fraction *f)
{
    return fraction_prefetch(ctx, input, post_output, f, true);
}
static int fraction_prefetch(cn_ctx *ctx, cn_layer *input, cn_layer *post_output, const cn_fraction *f, bool host)
{
    if (!ctx || !input || !f) { g_last_error = "cn_fraction_prefetch_resident: NULL argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        enter(ctx);
        check_fraction(ctx, input, post_output, f);
        if (host && !ctx->overlap) return;            // CN_NO_OVERLAP: no copy stream, no side streams: the load does it all
        const bool cls = post_output && (post_output->kind == CN_LAYER_MULTICLASS_CLASSIFICATION || post_output->kind == CN_LAYER_BINARY_CLASSIFICATION);
        if (post_output && cls && !f->target_classes) throw cn_error(CN_ERR_BAD_ARG, "cn_fraction_prefetch_resident: target_classes missing");
        if (post_output && !cls && !f->targets) throw cn_error(CN_ERR_BAD_ARG, "cn_fraction_prefetch_resident: targets missing");
        cn_ctx::Prefetch &pf = ctx->pf;
        if (pf.valid && pf.launched) throw cn_error(CN_ERR_STATE, "cn_fraction_prefetch_resident: the previous prefetch has not been consumed by a cn_fraction_load_resident yet");
        if (!pf.allocated) {          // the alternates of the four buffers a fraction is re-laid out into (same initial contents)
            const size_t maxN = input->maxN(), guard = (size_t)CN_GUARD_STEPS * ctx->PSp;
            HIP_CHECK(hipMalloc((void **)&pf.pat_raw, cn_ctx::pat_bytes(maxN, guard)));
            HIP_CHECK(hipMemsetAsync(pf.pat_raw, 0, cn_ctx::pat_bytes(maxN, guard), ctx->stream));
            pf.pat = pf.pat_raw + guard;
            HIP_CHECK(hipMalloc((void **)&pf.tcls, maxN * sizeof(int)));
            HIP_CHECK(hipMemsetAsync(pf.tcls, 0xFF, maxN * sizeof(int), ctx->stream));
            pf.in_op = dalloc(input, maxN * input->Lp * ctx->esz());
            if (post_output && post_output->targets) pf.targets = (float *)dalloc(post_output, maxN * post_output->size * sizeof(float));
            pf.allocated = true;
        }
        if (post_output && post_output->targets && !pf.targets) pf.targets = (float *)dalloc(post_output, input->maxN() * post_output->size * sizeof(float));
        pf.f = *f; pf.input = input; pf.post = post_output; pf.launched = false; pf.host = host;
        if (host) {
            // pack and upload NOW (copy stream, beside whatever the device is doing); the re-layout follows on the side stream of the
            // next gradient work, reading the staging area
            const Staged sg = stage_upload(ctx, input, post_output, f);
            pf.f_host = *f; pf.slot = sg.slot;
            pf.f.pat_types = sg.dv;
            pf.f.target_classes = (post_output && sg.cls) ? (const int *)(sg.dv + sg.b_pat) : nullptr;
            pf.f.targets = (post_output && !sg.cls) ? (const float *)(sg.dv + sg.b_pat) : nullptr;
            pf.f.inputs = (const float *)(sg.dv + sg.b_pat + sg.b_tgt_al);
        }
        pf.valid = true;
    });
}

// ---------------------------------------------------------------------------------------------
// forward / backward / loss
// ---------------------------------------------------------------------------------------------
static int post_kind(const cn_layer *l)
{
    switch (l->kind) {
    case CN_LAYER_WEIGHTEDSSE:           return POST_WEIGHTEDSSE;
    case CN_LAYER_SSE_MASK:              return POST_SSE_MASK;
    case CN_LAYER_CE:                    return POST_CE;
    case CN_LAYER_RMSE:                  return POST_RMSE;
    case CN_LAYER_BINARY_CLASSIFICATION: return POST_BINARY;
    default:                             return POST_SSE;
    }
}

int cn_layer_forward(cn_layer *layer)
{
    if (!layer) { g_last_error = "cn_layer_forward: layer is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        enter(layer->ctx);
        require_loaded(layer->ctx);
        finalize(layer->ctx);
        join_side(layer->ctx);
        if (layer->lstm) lstm_forward(layer);
        else if (layer->trainable) ff_forward(layer);
        /* input and post output layers: no-op (InputLayer.cpp:62-65, SsePostOutputLayer.cu:134-137) */
    });
}

int cn_layer_backward(cn_layer *layer)
{
    if (!layer) { g_last_error = "cn_layer_backward: layer is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        cn_ctx *c = layer->ctx;
        enter(c);
        require_loaded(c);
        finalize(c);
        if (layer->trainable && layer->updated)
            throw cn_error(CN_ERR_STATE, "cn_layer_backward: the armed update of this layer's previous backward pass has not been completed (call cn_sgd_update_all / cn_sgd_update first)");
        if (layer->lstm) lstm_backward(layer);
        else if (layer->trainable) ff_backward(layer);
        else if (layer->post) {
            Timed tm(c, KC_OTHER);
            cn_layer *o = layer->prev;
            if (layer->kind == CN_LAYER_MULTICLASS_CLASSIFICATION)
                o->mcc_pending = true;         // injected inside the output layer's backward pass (fused kernel)
            else
                launch_post_backward(c->stream, post_kind(layer), posteriors(o), layer->targets, c->d_pat, c->N, o->size, o->Lp, o->err);
        }
    });
}

int cn_loss_eval(cn_layer *post, float *error, int *correct)
{
    if (!post || !error) { g_last_error = "cn_loss_eval: NULL argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        cn_ctx *c = post->ctx;
        enter(c);
        if (!post->post) throw cn_error(CN_ERR_BAD_ARG, "cn_loss_eval: not a post output layer");
        require_loaded(c);
        cn_layer *o = post->prev;
        {
            Timed tm(c, KC_OTHER);
            if (post->kind == CN_LAYER_MULTICLASS_CLASSIFICATION && c->rowstat_of == o)
                launch_rowstat_reduce(c->stream, c->d_rowstat, c->N, c->d_loss, true);
            else if (post->kind == CN_LAYER_MULTICLASS_CLASSIFICATION) {
                flush_loss(c);
                launch_mcc_eval(c->stream, posteriors(o), c->d_tcls, c->N, post->size, o->Lp, c->d_loss, true, c->d_rowstat);
                c->rowstat_of = nullptr;       // (the row statistics now are this evaluation's terms)
            } else {
                flush_loss(c);
                launch_post_eval(c->stream, post_kind(post), posteriors(o), post->targets, c->d_pat, c->N, o->size, o->Lp, c->d_rowstat, c->d_loss, true);
                c->rowstat_of = nullptr;       // the softmax row statistics were overwritten
            }
        }
        float h[2];
        HIP_CHECK(hipMemcpyAsync(h, c->d_loss, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        check_fault(c);
        *error = h[0];
        if (correct) {
            int cc; memcpy(&cc, &h[1], sizeof(int));
            *correct = (post->kind == CN_LAYER_MULTICLASS_CLASSIFICATION || post->kind == CN_LAYER_BINARY_CLASSIFICATION) ? cc : -1;
        }
    });
}

int cn_loss_accumulate(cn_layer *post)
{
    if (!post) { g_last_error = "cn_loss_accumulate: NULL argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        cn_ctx *c = post->ctx;
        enter(c);
        if (!post->post) throw cn_error(CN_ERR_BAD_ARG, "cn_loss_accumulate: not a post output layer");
        require_loaded(c);
        cn_layer *o = post->prev;
        flush_loss(c);
        // Training: the backward pass of the output layer follows; its launch (softmax_mcc_bwd_kernel) takes the sum along in one
        // extra workgroup, so nothing is enqueued here.  Whoever needs the sums or overwrites the rows first flushes (flush_loss).
        const bool defer_off = opt().no_loss_defer;
        if (post->kind == CN_LAYER_MULTICLASS_CLASSIFICATION && c->rowstat_of == o && !c->timing && !defer_off && softmax_mcc_bwd_takes_loss(o->Lp)) {
            c->loss_deferred = true;
            return;
        }
        Timed tm(c, KC_OTHER);
        if (post->kind == CN_LAYER_MULTICLASS_CLASSIFICATION && c->rowstat_of == o)
            launch_rowstat_reduce(c->stream, c->d_rowstat, c->N, c->d_loss_acc, false);
        else if (post->kind == CN_LAYER_MULTICLASS_CLASSIFICATION) {
            launch_mcc_eval(c->stream, posteriors(o), c->d_tcls, c->N, post->size, o->Lp, c->d_loss_acc, false, c->d_rowstat);
            c->rowstat_of = nullptr;
        } else {
            launch_post_eval(c->stream, post_kind(post), posteriors(o), post->targets, c->d_pat, c->N, o->size, o->Lp, c->d_rowstat, c->d_loss_acc, false);
            c->rowstat_of = nullptr;
        }
    });
}

int cn_loss_read(cn_ctx *ctx, float *error_sum, int64_t *correct_sum, int reset)
{
    if (!ctx) { g_last_error = "cn_loss_read: ctx is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        enter(ctx);
        flush_loss(ctx);
        float h[2];
        HIP_CHECK(hipMemcpyAsync(h, ctx->d_loss_acc, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        if (reset) HIP_CHECK(hipMemsetAsync(ctx->d_loss_acc, 0, sizeof(h), ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        check_fault(ctx);
        if (error_sum) *error_sum = h[0];
        if (correct_sum) { int cc; memcpy(&cc, &h[1], sizeof(int)); *correct_sum = cc; }
    });
}

int cn_loss_read_global(cn_ctx *ctx, float *error_sum, int64_t *correct_sum, int reset)
{
    if (!ctx) { g_last_error = "cn_loss_read_global: ctx is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        require_comm(ctx, "cn_loss_read_global");
        enter(ctx);
        flush_loss(ctx);
        float *g = ctx->d_loss + 4, h[2];
        if (ctx->ipc) {
            HIP_CHECK(hipMemcpyAsync(h, ctx->d_loss_acc, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
            if (reset) HIP_CHECK(hipMemsetAsync(ctx->d_loss_acc, 0, sizeof(h), ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            check_fault(ctx);
            int cc; memcpy(&cc, &h[1], sizeof(int));
            try { ipc_comm_check(ctx->ipc, ctx->comm_stream); ipc_allreduce_loss(ctx->ipc, &h[0], &cc); }
            catch (const std::exception &e) { ipc_comm_mark_failed(ctx->ipc); throw cn_error(CN_ERR_COMM, e.what()); }
            if (error_sum) *error_sum = h[0];
            if (correct_sum) *correct_sum = cc;
            return;
        }
        RCCL_CHECK(rccl().GroupStart());
        RCCL_CHECK(rccl().AllReduce(ctx->d_loss_acc, g, 1, ncclFloat32, ncclSum, ctx->comm, ctx->stream));
        RCCL_CHECK(rccl().AllReduce(ctx->d_loss_acc + 1, g + 1, 1, ncclInt32, ncclSum, ctx->comm, ctx->stream));
        RCCL_CHECK(rccl().GroupEnd());
        HIP_CHECK(hipMemcpyAsync(h, g, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        if (reset) HIP_CHECK(hipMemsetAsync(ctx->d_loss_acc, 0, sizeof(h), ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        check_fault(ctx);
        if (error_sum) *error_sum = h[0];
        if (correct_sum) { int cc; memcpy(&cc, &h[1], sizeof(int)); *correct_sum = cc; }
    });
}

// ---------------------------------------------------------------------------------------------
// weights and buffers
// ---------------------------------------------------------------------------------------------
int cn_layer_set_weights(cn_layer *layer, const float *host, int count)
{
    if (!layer || !host) { g_last_error = "cn_layer_set_weights: NULL argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        cn_ctx *c = layer->ctx;
        enter(c);
        if (!layer->trainable) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_set_weights: layer has no weights");
        if (count != layer->nw)
            throw cn_error(CN_ERR_SHAPE, "Invalid number of weights: " + std::to_string(count) + " given, " + std::to_string(layer->nw) + " expected");
        if (!c->finalized) { layer->pending_w.assign(host, host + count); return; }
        HIP_CHECK(hipMemcpyAsync(layer->w, host, (size_t)count * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        layer->dirty = true;
    });
}

int cn_layer_upload(cn_layer *layer, cn_buffer which, const float *host, size_t count)
{
    if (!layer || !host) { g_last_error = "cn_layer_upload: NULL argument"; return CN_ERR_BAD_ARG; }
    if (which == CN_BUF_WEIGHTS) return cn_layer_set_weights(layer, host, (int)count);
    return guarded([&] {
        cn_ctx *c = layer->ctx;
        enter(c);
        if (!layer->trainable) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_upload: layer has no weights");
        if (which != CN_BUF_WEIGHT_UPDATES && which != CN_BUF_WEIGHT_DELTAS) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_upload: not a parameter vector");
        if (count != (size_t)layer->nw) throw cn_error(CN_ERR_SHAPE, "cn_layer_upload: count != weight count");
        finalize(c);
        join_side(c);
        HIP_CHECK(hipMemcpyAsync(which == CN_BUF_WEIGHT_UPDATES ? layer->wu : layer->wd, host, count * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
    });
}

int cn_layer_write_output_errors(cn_layer *layer, const float *host, size_t count)
{
    if (!layer || !host) { g_last_error = "cn_layer_write_output_errors: NULL argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        cn_ctx *c = layer->ctx;
        enter(c);
        require_loaded(c);
        if (!layer->err) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_write_output_errors: layer has no outputErrors");
        if (count != (size_t)c->Next * layer->size) throw cn_error(CN_ERR_SHAPE, "cn_layer_write_output_errors: count != T*PS*size");
        float *tmp = nullptr;
        HIP_CHECK(hipMalloc((void **)&tmp, count * sizeof(float)));
        HIP_CHECK(hipMemcpyAsync(tmp, host, count * sizeof(float), hipMemcpyHostToDevice, c->stream));
        layer->mcc_pending = false;
        layer->err_in_delta = false;
        HIP_CHECK(hipMemsetAsync(layer->err, 0, (size_t)c->N * layer->Lp * sizeof(float), c->stream));
        launch_pad_f32(c->stream, tmp, c->Next, layer->size, layer->err, layer->Lp, layer->lstm ? layer->H : 0, layer->lstm ? layer->Hp : 0, c->PS, c->PSp);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        hipFree(tmp);
    });
}

int cn_layer_read(cn_layer *layer, cn_buffer which, int dir, float *host, size_t count)
{
    if (!layer || !host) { g_last_error = "cn_layer_read: NULL argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        cn_ctx *c = layer->ctx;
        enter(c);
        finalize(c);
        join_side(c);
        const size_t e = c->esz();
        const bool opbf = !c->f32;
        // flat parameter vectors
        if (which == CN_BUF_WEIGHTS || which == CN_BUF_WEIGHT_UPDATES || which == CN_BUF_WEIGHT_DELTAS) {
            if (!layer->trainable) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_read: layer has no weights");
            if (count != (size_t)layer->nw) throw cn_error(CN_ERR_SHAPE, "cn_layer_read: count != weight count");
            const float *src = which == CN_BUF_WEIGHTS ? layer->w : (which == CN_BUF_WEIGHT_UPDATES ? layer->wu : layer->wd);
            HIP_CHECK(hipMemcpyAsync(host, src, count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            return;
        }
        require_loaded(c);
        const int N = c->Next;                 // host layout: T*PS patterns
        int width = layer->size;
        if (which >= CN_BUF_LSTM_CELL_STATES) {
            if (!layer->lstm) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_read: not an LSTM layer");
            if (dir < 0 || dir >= layer->dirs) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_read: direction out of range");
            width = layer->H;
        }
        if (count != (size_t)N * width) throw cn_error(CN_ERR_SHAPE, "cn_layer_read: count != T*PS*width");
        float *tmp = nullptr;
        HIP_CHECK(hipMalloc((void **)&tmp, count * sizeof(float)));
        const int R = layer->dirs * 4 * layer->Hp, Hp = layer->Hp, H = layer->H;
        switch (which) {
        case CN_BUF_OUTPUTS:
            if (layer->kind == CN_LAYER_INPUT)        // the operand copy of the inputs (bf16 mode: rounded to bf16)
                launch_unpad(c->stream, opbf, layer->out_op, layer->Lp, 0, 1, N, layer->size, tmp, layer->size, 0, c->PS, c->PSp);
            else if (layer->lstm)
                for (int d = 0; d < layer->dirs; ++d) launch_unpad(c->stream, opbf, layer->out_op, layer->Lp, d * Hp, 1, N, H, tmp, layer->size, d * H, c->PS, c->PSp);
            else if (layer->trainable) launch_unpad(c->stream, false, posteriors(layer), layer->Lp, 0, 1, N, layer->size, tmp, layer->size, 0, c->PS, c->PSp);
            else throw cn_error(CN_ERR_BAD_ARG, "cn_layer_read: layer has no outputs");
            break;
        case CN_BUF_OUTPUT_ERRORS:
            if (!layer->err) throw cn_error(CN_ERR_BAD_ARG, "cn_layer_read: layer has no outputErrors");
            if (layer->mcc_pending) {     // the deferred multiclass error injection becomes visible here
                launch_mcc_backward(c->stream, posteriors(layer), c->d_tcls, c->N, layer->size, layer->Lp, layer->err);
                layer->mcc_pending = false;
            }
            if (layer->lstm)
                for (int d = 0; d < layer->dirs; ++d) launch_unpad(c->stream, false, layer->err, layer->Lp, d * Hp, 1, N, H, tmp, layer->size, d * H, c->PS, c->PSp);
            else if (layer->err_in_delta)      // fused softmax backward in bf16 mode: only the operand copy exists
                launch_unpad(c->stream, true, layer->delta_op, layer->Lp, 0, 1, N, layer->size, tmp, layer->size, 0, c->PS, c->PSp);
            else launch_unpad(c->stream, false, layer->err, layer->Lp, 0, 1, N, layer->size, tmp, layer->size, 0, c->PS, c->PSp);
            break;
        case CN_BUF_LSTM_CELL_STATES:
            launch_unpad(c->stream, false, layer->cell, layer->Lp, dir * Hp, 1, N, H, tmp, H, 0, c->PS, c->PSp); break;
        case CN_BUF_LSTM_TMP_OUTPUTS:
            launch_unpad(c->stream, opbf, layer->out_op, layer->Lp, dir * Hp, 1, N, H, tmp, H, 0, c->PS, c->PSp); break;
        case CN_BUF_LSTM_NI_ACTS: case CN_BUF_LSTM_IG_ACTS: case CN_BUF_LSTM_FG_ACTS: case CN_BUF_LSTM_OG_ACTS:
            launch_unpad(c->stream, false, layer->acts, R, dir * 4 * Hp + (which - CN_BUF_LSTM_NI_ACTS), 4, N, H, tmp, H, 0, c->PS, c->PSp); break;
        case CN_BUF_LSTM_NI_DELTAS: case CN_BUF_LSTM_IG_DELTAS: case CN_BUF_LSTM_FG_DELTAS: case CN_BUF_LSTM_OG_DELTAS:
            launch_unpad(c->stream, opbf, layer->delta_op, R, dir * 4 * Hp + (which - CN_BUF_LSTM_NI_DELTAS), 4, N, H, tmp, H, 0, c->PS, c->PSp); break;
        default:
            hipFree(tmp);
            throw cn_error(CN_ERR_BAD_ARG, "cn_layer_read: unknown buffer");
        }
        (void)e;
        HIP_CHECK(hipMemcpyAsync(host, tmp, count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        hipFree(tmp);
    });
}

void *cn_layer_device_ptr(cn_layer *layer, cn_buffer which)
{
    if (!layer || !layer->trainable) return nullptr;
    if (guarded([&] { enter(layer->ctx); finalize(layer->ctx); }) != CN_OK) return nullptr;
    switch (which) {
    case CN_BUF_WEIGHTS: return layer->w;
    case CN_BUF_WEIGHT_UPDATES: return layer->wu;
    case CN_BUF_WEIGHT_DELTAS: return layer->wd;
    default: return nullptr;
    }
}

int cn_ctx_param_arena(cn_ctx *ctx, void **weights, void **weight_updates, void **weight_deltas, size_t *count)
{
    if (!ctx) { g_last_error = "cn_ctx_param_arena: ctx is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        enter(ctx);
        finalize(ctx);
        if (weights) *weights = ctx->arena;
        if (weight_updates) *weight_updates = ctx->arena + ctx->total;
        if (weight_deltas) *weight_deltas = ctx->arena + 2 * ctx->total;
        if (count) *count = ctx->total;
    });
}

int cn_ctx_weights_touched(cn_ctx *ctx)
{
    if (!ctx) { g_last_error = "cn_ctx_weights_touched: ctx is NULL"; return CN_ERR_BAD_ARG; }
    for (cn_layer *l : ctx->layers) if (l->trainable) l->dirty = true;
    return CN_OK;
}

int cn_ctx_arm_update(cn_ctx *ctx, float learning_rate, float momentum)
{
    if (!ctx) { g_last_error = "cn_ctx_arm_update: ctx is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        for (cn_layer *l : ctx->layers)
            if (l->updated) throw cn_error(CN_ERR_STATE, "cn_ctx_arm_update: the previous armed update has not been completed (cn_sgd_update_all)");
        ctx->armed = true; ctx->arm_lr = learning_rate; ctx->arm_mom = momentum;
    });
}

int cn_sgd_update(cn_layer *layer, float learning_rate, float momentum)
{
    if (!layer) { g_last_error = "cn_sgd_update: layer is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        cn_ctx *c = layer->ctx;
        enter(c);
        if (!layer->trainable) throw cn_error(CN_ERR_BAD_ARG, "cn_sgd_update: layer has no weights");
        comm_check_fast(c);
        finalize(c);
        join_side(c);
        if (layer->updated) {            // cn_ctx_arm_update: this layer's step ran behind its gradient; nothing left but the wait above
            const float want = layer->own_lr >= 0.f ? layer->own_lr : c->arm_lr;
            if (learning_rate != want || momentum != c->arm_mom)
                throw cn_error(CN_ERR_STATE, "cn_sgd_update: learning rate / momentum differ from what cn_ctx_arm_update armed and applied");
            layer->updated = false;
            bool any = false;
            for (cn_layer *o : c->layers) any = any || o->updated;
            if (!any) c->armed = false;
            return;
        }
        Timed tm(c, KC_OTHER);
        launch_sgd(c->stream, layer->w, layer->wu, layer->wd, (size_t)layer->nw, learning_rate, momentum);
        layer->dirty = true;
    });
}

int cn_ctx_accumulate_updates(cn_ctx *ctx, int first)
{
    if (!ctx) { g_last_error = "cn_ctx_accumulate_updates: ctx is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        enter(ctx);
        finalize(ctx);
        if (ctx->armed) throw cn_error(CN_ERR_STATE, "cn_ctx_accumulate_updates: an armed per-fraction update is pending (batch learning sums first, cn_ctx_arm_update is not for it)");
        if (!first && !ctx->acc_valid) throw cn_error(CN_ERR_STATE, "cn_ctx_accumulate_updates: nothing accumulated yet (the first fraction of an epoch passes first != 0)");
        join_side(ctx);                     // the gradient GEMMs / unpack launches of the side streams write weightUpdates
        if (!ctx->acc) HIP_CHECK(hipMalloc((void **)&ctx->acc, ctx->total * sizeof(float)));
        Timed tm(ctx, KC_OTHER);
        launch_accumulate(ctx->stream, ctx->acc, ctx->arena + ctx->total, ctx->total, first != 0);
        ctx->acc_valid = true;
    });
}

int cn_ctx_take_accumulated(cn_ctx *ctx)
{
    if (!ctx) { g_last_error = "cn_ctx_take_accumulated: ctx is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        enter(ctx);
        finalize(ctx);
        if (!ctx->acc_valid) throw cn_error(CN_ERR_STATE, "cn_ctx_take_accumulated: nothing accumulated (cn_ctx_accumulate_updates)");
        join_side(ctx);
        HIP_CHECK(hipMemcpyAsync(ctx->arena + ctx->total, ctx->acc, ctx->total * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        ctx->acc_valid = false;
    });
}

int cn_layer_set_learning_rate(cn_layer *layer, float learning_rate)
{
    if (!layer) { g_last_error = "cn_layer_set_learning_rate: layer is NULL"; return CN_ERR_BAD_ARG; }
    if (!layer->trainable) { g_last_error = "cn_layer_set_learning_rate: layer has no weights"; return CN_ERR_BAD_ARG; }
    layer->own_lr = learning_rate;
    return CN_OK;
}

int cn_sgd_update_all(cn_ctx *ctx, float learning_rate, float momentum)
{
    if (!ctx) { g_last_error = "cn_sgd_update_all: ctx is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        enter(ctx);
        comm_check_fast(ctx);
        finalize(ctx);
        join_side(ctx);
        // cn_ctx_arm_update: layers whose step ran behind their gradient are complete (the wait above orders this stream behind
        // them); what follows handles the rest (none, normally)
        bool any_updated = false;
        for (cn_layer *l : ctx->layers) any_updated = any_updated || l->updated;
        if (any_updated && (learning_rate != ctx->arm_lr || momentum != ctx->arm_mom))
            throw cn_error(CN_ERR_STATE, "cn_sgd_update_all: learning rate / momentum differ from what cn_ctx_arm_update armed and applied");
        ctx->armed = false;
        struct ClearUpdated { cn_ctx *c; ~ClearUpdated() { for (cn_layer *l : c->layers) l->updated = false; } } clear_updated{ctx};
        Timed tm(ctx, KC_OTHER);
        int ntrain = 0;
        for (cn_layer *l : ctx->layers) if (l->trainable && !l->updated) ++ntrain;
        if (ntrain == 0) return;
        if (any_updated) {               // some layers were not reached by the armed pass (no backward call for them): one launch each
            for (cn_layer *l : ctx->layers)
                if (l->trainable && !l->updated) launch_layer_update(ctx->stream, l, 1, learning_rate, momentum, nullptr);
            return;
        }
        // The operand copies of the new weights are rebuilt right away, all layers in ONE launch on this stream
        // (pack_group_kernel): it costs about as much as the first layer's copy alone did, which was on the critical
        // path anyway, and the other layers' copies no longer need a fork event, the side stream and a wait.
        const bool group_off = opt().no_pack_group;
        const bool grouped = ctx->overlap && !group_off && ntrain <= PACK_GROUP_MAX;
        const bool attach = ctx->overlap && !grouped && ctx->attach_forks && !ctx->timing;
        if (ctx->overlap && !grouped && !ctx->ev_sgd) HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_sgd, hipEventDisableTiming));
        bool own_rates = false;
        for (cn_layer *l : ctx->layers) own_rates = own_rates || (l->trainable && l->own_lr >= 0.f);
        // grouped: the update itself rides on the pack launch (pack_fetch): one kernel instead of two behind the last gradient
        const bool fuse_off = opt().no_sgd_fuse;
        const bool fused = grouped && !fuse_off;
        if (fused) {
        } else if (!own_rates) {
            launch_sgd(ctx->stream, ctx->arena, ctx->arena + ctx->total, ctx->arena + 2 * ctx->total, ctx->total, learning_rate, momentum,
                       attach ? ctx->ev_sgd : nullptr);
        } else {
            // a layer with a "learningRate" of its own (SteepestDescentOptimizer.cu:78-80): one launch per layer
            cn_layer *last = nullptr;
            for (cn_layer *l : ctx->layers) if (l->trainable) last = l;
            for (cn_layer *l : ctx->layers)
                if (l->trainable)
                    launch_sgd(ctx->stream, l->w, l->wu, l->wd, (size_t)l->nw, l->own_lr >= 0.f ? l->own_lr : learning_rate, momentum,
                               (attach && l == last) ? ctx->ev_sgd : nullptr);
        }
        for (cn_layer *l : ctx->layers) if (l->trainable) l->dirty = true;
        if (grouped) {
            PackGroup grp{};
            for (cn_layer *l : ctx->layers) {
                if (!l->trainable) continue;
                PackItem &it = grp.item[grp.n++];
                it.lstm = l->lstm ? 1 : 0;
                if (l->lstm) it.lg = lstm_geom(l); else it.fg = ff_geom(l);
                it.bias = l->bias; it.w = l->w; it.Win = l->Win; it.WinT = l->WinT; it.Wrec = l->Wrec; it.WrecT = l->WrecT;
                it.bias_p = l->bias_p; it.peep_p = l->peep_p;
                if (fused) {
                    it.update = 1; it.w_rw = l->w; it.wu = l->wu; it.wd = l->wd;
                    it.lr = l->own_lr >= 0.f ? l->own_lr : learning_rate; it.mom = momentum;
                }
                l->dirty = false; l->pack_pending = false;
            }
            launch_pack_group(ctx->stream, ctx->f32, grp);
        } else if (ctx->overlap) {
            // (more layers than one group launch takes: the first trainable layer's copy on this stream, its forward pass
            // is next; the others on the side stream, each layer's forward pass waits for them in repack())
            if (!attach) HIP_CHECK(hipEventRecord(ctx->ev_sgd, ctx->stream));
            HIP_CHECK(hipStreamWaitEvent(ctx->side, ctx->ev_sgd, 0));
            bool first = true;
            for (cn_layer *l : ctx->layers) {
                if (!l->trainable) continue;
                if (first) { first = false; continue; }
                if (!l->ev_pack) HIP_CHECK(hipEventCreateWithFlags(&l->ev_pack, hipEventDisableTiming));
                {
                    Timed tp(ctx, KC_OTHER, ctx->side);
                    if (l->lstm) launch_lstm_pack(ctx->side, ctx->f32, lstm_geom(l), l->bias, l->w, l->Win, l->WinT, l->Wrec, l->WrecT, l->bias_p, l->peep_p);
                    else         launch_ff_pack(ctx->side, ctx->f32, ff_geom(l), l->bias, l->w, l->Win, l->WinT, l->bias_p);
                }
                HIP_CHECK(hipEventRecord(l->ev_pack, ctx->side));
                ctx->ev_pack_last = l->ev_pack;
                l->dirty = false; l->pack_pending = true;
            }
        }
    });
}

// ---------------------------------------------------------------------------------------------
// timing
// ---------------------------------------------------------------------------------------------
int cn_ctx_timing_enable(cn_ctx *ctx, int enable)
{
    if (!ctx) { g_last_error = "cn_ctx_timing_enable: ctx is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] { enter(ctx); timing_collect(ctx); ctx->timing = enable != 0; });
}
int cn_ctx_timing_read(cn_ctx *ctx, int kernel_class, double *total_ms, int64_t *launches)
{
    if (!ctx || kernel_class < 0 || kernel_class >= KC_COUNT) { g_last_error = "cn_ctx_timing_read: bad argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        enter(ctx);
        timing_collect(ctx);
        if (total_ms) *total_ms = ctx->acc_ms[kernel_class];
        if (launches) *launches = ctx->acc_n[kernel_class];
    });
}
int cn_ctx_timing_reset(cn_ctx *ctx)
{
    if (!ctx) { g_last_error = "cn_ctx_timing_reset: ctx is NULL"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        enter(ctx);
        timing_collect(ctx);
        for (int k = 0; k < KC_COUNT; ++k) { ctx->acc_ms[k] = 0; ctx->acc_n[k] = 0; }
    });
}

const char *cn_layer_recurrent_kernel(cn_layer *layer, int backward)
{
    // what the launchers instantiated on the layer's last forward / backward pass (empty before the first one)
    if (!layer || !layer->lstm) return "";
    return layer->kname[backward ? 1 : 0];
}

// ---------------------------------------------------------------------------------------------
// kernel-level test hooks (include/currennt_hip_debug.h)
// ---------------------------------------------------------------------------------------------
int cn_dbg_gemm_nt(cn_ctx *ctx, const float *A, const float *B, float *C, int M, int N, int K, const float *bias, int act)
{
    if (!ctx || !A || !B || !C) { g_last_error = "cn_dbg_gemm_nt: NULL argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        enter(ctx);
        if (K % 8 || N % 32) throw cn_error(CN_ERR_SHAPE, "cn_dbg_gemm_nt: K must be a multiple of 8 and N of 32");
        const size_t e = ctx->esz();
        float *dA, *dB, *dC, *dbias = nullptr; void *oA, *oB;
        HIP_CHECK(hipMalloc((void **)&dA, (size_t)M * K * 4)); HIP_CHECK(hipMalloc((void **)&dB, (size_t)N * K * 4));
        HIP_CHECK(hipMalloc((void **)&dC, (size_t)M * N * 4));
        HIP_CHECK(hipMalloc(&oA, (size_t)M * K * e)); HIP_CHECK(hipMalloc(&oB, (size_t)N * K * e));
        HIP_CHECK(hipMemcpyAsync(dA, A, (size_t)M * K * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(dB, B, (size_t)N * K * 4, hipMemcpyHostToDevice, ctx->stream));
        if (bias) { HIP_CHECK(hipMalloc((void **)&dbias, (size_t)N * 4)); HIP_CHECK(hipMemcpyAsync(dbias, bias, (size_t)N * 4, hipMemcpyHostToDevice, ctx->stream)); }
        launch_pad_convert(ctx->stream, ctx->f32, dA, M, K, oA, K);
        launch_pad_convert(ctx->stream, ctx->f32, dB, N, K, oB, K);
        // act | 0x100: both outputs (fp32 and operand-type copy), the COPY is returned; act | 0x200: the copy alone
        const bool both = act & 0x100, copy_only = act & 0x200;
        void *dC2 = nullptr;
        if (both || copy_only) HIP_CHECK(hipMalloc(&dC2, (size_t)M * N * e));
        GemmNT g{}; g.A = oA; g.lda = K; g.B = oB; g.ldb = K; g.C = copy_only ? nullptr : dC; g.ldc = N; g.C2 = dC2; g.ldc2 = N;
        g.bias = dbias; g.act = act & 0xff; g.M = M; g.N = N; g.K = K;
        launch_gemm_nt(ctx->stream, ctx->prec, g);
        HIP_CHECK(hipGetLastError());
        if (dC2 && e == 2) {
            std::vector<uint16_t> h((size_t)M * N);
            HIP_CHECK(hipMemcpyAsync(h.data(), dC2, h.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            for (size_t i = 0; i < h.size(); ++i) { const uint32_t u = (uint32_t)h[i] << 16; memcpy(&C[i], &u, 4); }
        } else {
            HIP_CHECK(hipMemcpyAsync(C, dC2 ? dC2 : dC, (size_t)M * N * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(oA); hipFree(oB); hipFree(dbias); hipFree(dC2);
    });
}

int cn_dbg_prefetch_hits(cn_ctx *ctx, int *hits)
{
    if (!ctx || !hits) { g_last_error = "cn_dbg_prefetch_hits: NULL argument"; return CN_ERR_BAD_ARG; }
    *hits = ctx->pf.hits;
    return CN_OK;
}

int cn_dbg_row_map_counts(cn_ctx *ctx, int out[3])
{
    if (!ctx || !out) { g_last_error = "cn_dbg_row_map_counts: NULL argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        enter(ctx);
        require_loaded(ctx);
        join_side(ctx);
        HIP_CHECK(hipMemcpyAsync(out, ctx->rowmap_of(ctx->d_pat_raw), 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        out[2] = ctx->N;
    });
}

int cn_dbg_gemm_tn(cn_ctx *ctx, const float *A, const float *B, float *C, int M, int N, int K)
{
    if (!ctx || !A || !B || !C) { g_last_error = "cn_dbg_gemm_tn: NULL argument"; return CN_ERR_BAD_ARG; }
    return guarded([&] {
        enter(ctx);
        if (M % 32 || N % 32) throw cn_error(CN_ERR_SHAPE, "cn_dbg_gemm_tn: M and N must be multiples of 32");
        const size_t e = ctx->esz();
        float *dA, *dB, *dC; void *oA, *oB;
        HIP_CHECK(hipMalloc((void **)&dA, (size_t)M * K * 4)); HIP_CHECK(hipMalloc((void **)&dB, (size_t)N * K * 4));
        HIP_CHECK(hipMalloc((void **)&dC, (size_t)M * N * 4));
        HIP_CHECK(hipMalloc(&oA, (size_t)M * K * e)); HIP_CHECK(hipMalloc(&oB, (size_t)N * K * e));
        HIP_CHECK(hipMemcpyAsync(dA, A, (size_t)M * K * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(dB, B, (size_t)N * K * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipMemsetAsync(dC, 0, (size_t)M * N * 4, ctx->stream));
        launch_pad_convert(ctx->stream, ctx->f32, dA, K, M, oA, M);
        launch_pad_convert(ctx->stream, ctx->f32, dB, K, N, oB, N);
        GemmTN g{}; g.A = oA; g.lda = M; g.B = oB; g.ldb = N; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.K = K;
        launch_gemm_tn(ctx->stream, ctx->prec, g);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(oA); hipFree(oB);
    });
}

}  // extern "C"

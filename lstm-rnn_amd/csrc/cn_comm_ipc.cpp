// CN_COMM_BACKEND=ipc -- a TEST backend of the gradient exchange for boxes with fewer GPUs than ranks.
//
// RCCL refuses two ranks on one device, so on a one-GPU box the world > 1 flow of the C++ driver (fork, rendezvous, shard of
// every fraction per rank, per-layer exchange behind the backward pass, global loss, rank-0 file writing) could never
// run.  With CN_COMM_BACKEND=ipc in the environment cn_comm_unique_id / cn_comm_init / cn_allreduce_grads /
// cn_loss_read_global keep their contract but exchange through the host: the ranks (processes that may share a device) meet in a
// POSIX shared-memory segment named by the id, publish hipIpc handles of a staging buffer each, and an all-reduce is
//   copy my gradient into my staging buffer -> barrier -> sum the staging buffers of ALL ranks in rank order into my gradient
//   (peers' buffers through hipIpcOpenMemHandle) -> barrier.
// Every rank adds the same numbers in the same order: the replicas stay bit-identical, as with RCCL's ring.  The call blocks the
// host (two barriers per exchange) -- it is a functional double, never a measurement, and bench.py refuses it for `value`.
//
// CN_COMM_BACKEND=p2p -- the NATIVE small-message backend, same rendezvous: every rank owns a region (flag words + two staging
// halves), publishes its hipIpc handle once (and again when a larger bucket arrives: the only host barriers of the backend), and
// an all-reduce is ONE kernel on the caller's stream that stages, signals, sums and acknowledges through the peers' mapped
// regions (cn_comm_p2p.hip).  Nothing blocks the host; a poll that times out on the device marks the communicator failed, turns
// the workgroup's share of the gradient into NaN and sets a host-mapped word: the next call into the communicator
// (cn_allreduce_grads, cn_sgd_update*, cn_loss_read_global, cn_comm_destroy) raises.  One rank per GPU over xGMI is the intended
// use; ranks that share a device (the tests on a one-GPU box) run the same code.  cn_comm_init runs a first-contact self-check
// through every peer's mapping (ipc_comm_p2p_selfcheck) and fails over to RCCL, with a message, when any rank sees a wrong sum:
// the first time two DEVICES meet must not be a silent wrong gradient.  cn_comm_destroy is collective in this mode (a host
// barrier in front of the unmapping: a peer may still be reading my staging half when my own kernel has finished).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "cn_internal.h"

namespace cn {

namespace {

constexpr int IPC_MAX_RANKS = 8;
constexpr unsigned IPC_MAGIC = 0x434e4950u;      // "CNIP"

struct Slot {
    hipIpcMemHandle_t handle;                    // staging buffer of this rank
    std::atomic<unsigned long long> generation;  // bumped when the buffer was reallocated (peers reopen the handle)
    std::atomic<unsigned long long> capacity;    // floats
    float err; int correct;                      // cn_loss_read_global
};
struct Shared {
    std::atomic<unsigned> magic;
    std::atomic<unsigned> arrived;               // sense-reversing barrier
    std::atomic<unsigned> sense;
    std::atomic<unsigned> failed;                // a rank gave up: everybody leaves
    std::atomic<unsigned> selfcheck_bad;         // p2p: some rank's first-contact self-check saw a wrong sum or a time-out
    char rccl_id[128];                           // ... then rank 0 leaves the id of the RCCL communicator to fail over to here
    Slot slot[IPC_MAX_RANKS];
};

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

void hipck(hipError_t e, const char *what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string("ipc communicator: ") + what + ": " + hipGetErrorString(e));
}

}  // namespace

struct IpcComm {
    Shared *sh = nullptr;
    std::string name;
    int rank = 0, world = 1;
    unsigned my_sense = 0;
    float *staging = nullptr; size_t capacity = 0;
    float *peer[IPC_MAX_RANKS] = {};
    unsigned long long peer_gen[IPC_MAX_RANKS] = {};
    double timeout_s = 120.0;
    // p2p mode (cn_comm_p2p.hip): my region = flag words, then two staging halves of half_cap floats; the peers' regions mapped
    bool p2p = false;
    unsigned long long *region = nullptr; size_t half_cap = 0;
    unsigned long long *peer_region[IPC_MAX_RANKS] = {};
    unsigned long long seq = 0;
    size_t oneshot_max = 256 * 1024;             // floats: larger buckets go reduce-scatter + all-gather
    unsigned long long *host_failed = nullptr, *host_failed_dev = nullptr;   // host-mapped word the kernel sets when a wait times out
    bool force_two_phase = false, coarse = false, verbose = false;           // CN_P2P_FORCE_TWO_PHASE / _COARSE / _VERBOSE, read once at create

    void barrier(const char *what)
    {
        my_sense ^= 1u;
        if (sh->arrived.fetch_add(1) + 1 == (unsigned)world) {
            sh->arrived.store(0);
            sh->sense.store(my_sense);
            return;
        }
        const double t0 = now_s();
        while (sh->sense.load() != my_sense) {
            if (sh->failed.load()) throw std::runtime_error(std::string("ipc communicator: another rank failed (rank ") + std::to_string(rank) + " was in " + what + ")");
            if (now_s() - t0 > timeout_s) {
                sh->failed.store(1);
                throw std::runtime_error("ipc communicator: rank " + std::to_string(rank) + " of " + std::to_string(world) + " waited " +
                                         std::to_string((int)timeout_s) + " s for its peers in " + what);
            }
            usleep(20);
        }
    }
};

static bool p2p_selected()
{
    const char *b = getenv("CN_COMM_BACKEND");
    return b && !strcmp(b, "p2p");
}
bool ipc_backend_selected()
{
    const char *b = getenv("CN_COMM_BACKEND");
    return b && (!strcmp(b, "ipc") || !strcmp(b, "p2p"));
}
bool ipc_comm_is_p2p(const IpcComm *c) { return c && c->p2p; }

// Segments this process created and nobody has unlinked yet (rank 0 unlinks inside ipc_comm_create): if the caller throws
// between cn_comm_unique_id and cn_comm_init, or exits without ever calling it, the name would stay in /dev/shm.
static std::mutex g_pending_mu;
static std::vector<std::string> g_pending;
static pid_t g_pending_pid = 0;               // (a forked child inherits the list but owns none of it)
static void ipc_unlink_pending()
{
    std::lock_guard<std::mutex> lk(g_pending_mu);
    if (g_pending_pid == getpid())
        for (const std::string &n : g_pending) shm_unlink(n.c_str());
    g_pending.clear();
}
static void ipc_forget_pending(const char *id)
{
    std::lock_guard<std::mutex> lk(g_pending_mu);
    for (size_t i = 0; i < g_pending.size(); ++i)
        if (g_pending[i] == id) { g_pending.erase(g_pending.begin() + i); break; }
}

// rank 0: create the segment; its name is the id
void ipc_unique_id(char *id, size_t bytes)
{
    static std::atomic<unsigned> serial{0};
    static std::once_flag once;
    std::call_once(once, [] { atexit(ipc_unlink_pending); });
    memset(id, 0, bytes);
    snprintf(id, bytes, "/cn_ipc_%d_%u_%lx", (int)getpid(), serial.fetch_add(1), (unsigned long)(now_s() * 1e6));
    int fd = shm_open(id, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) throw std::runtime_error(std::string("ipc communicator: shm_open(") + id + ") failed");
    if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); shm_unlink(id); throw std::runtime_error("ipc communicator: ftruncate failed"); }
    void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { shm_unlink(id); throw std::runtime_error("ipc communicator: mmap failed"); }
    memset(p, 0, sizeof(Shared));
    ((Shared *)p)->magic.store(IPC_MAGIC);
    munmap(p, sizeof(Shared));
    std::lock_guard<std::mutex> lk(g_pending_mu);
    if (g_pending_pid != getpid()) { g_pending.clear(); g_pending_pid = getpid(); }
    g_pending.push_back(id);
}

IpcComm *ipc_comm_create(const char *id, int rank, int world)
{
    if (world > IPC_MAX_RANKS) throw std::runtime_error("ipc communicator: at most " + std::to_string(IPC_MAX_RANKS) + " ranks");
    if (id[0] != '/' || strncmp(id, "/cn_ipc_", 8)) throw std::runtime_error("ipc communicator: the id does not come from cn_comm_unique_id with CN_COMM_BACKEND=ipc");
    IpcComm *c = new IpcComm;
    c->name = id; c->rank = rank; c->world = world; c->p2p = p2p_selected();
    if (const char *t = getenv("CN_P2P_ONESHOT_MAX")) c->oneshot_max = (size_t)atoll(t);
    c->force_two_phase = getenv("CN_P2P_FORCE_TWO_PHASE") != nullptr;
    c->coarse = getenv("CN_P2P_COARSE") != nullptr;
    c->verbose = getenv("CN_P2P_VERBOSE") != nullptr;
    if (const char *t = getenv("CN_COMM_IPC_TIMEOUT")) {
        const double v = atof(t);              // garbage or 0 would time every barrier out at once: keep the default then
        if (v >= 1.0) c->timeout_s = v;
    }
    int fd = shm_open(id, O_RDWR, 0600);
    if (fd < 0) { delete c; throw std::runtime_error(std::string("ipc communicator: rank ") + std::to_string(rank) + " cannot open " + id); }
    void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; throw std::runtime_error("ipc communicator: mmap failed"); }
    c->sh = (Shared *)p;
    if (c->sh->magic.load() != IPC_MAGIC) { munmap(p, sizeof(Shared)); delete c; throw std::runtime_error("ipc communicator: segment not initialised"); }
    try {
        c->barrier("cn_comm_init");
    } catch (...) {
        if (rank == 0) { shm_unlink(id); ipc_forget_pending(id); }
        munmap(p, sizeof(Shared)); delete c;
        throw;
    }
    if (rank == 0) { shm_unlink(id); ipc_forget_pending(id); }           // every rank holds a mapping now: nothing stays behind in /dev/shm
    if (c->p2p) {
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (h) (void)hipHostFree(h);
            munmap(p, sizeof(Shared)); delete c;
            throw std::runtime_error("p2p communicator: no host-mapped memory for the failure word");
        }
        memset(h, 0, 64);
        c->host_failed = (unsigned long long *)h; c->host_failed_dev = (unsigned long long *)d;
    }
    return c;
}

static bool p2p_failed_word(const IpcComm *c) { return c->host_failed && *(volatile unsigned long long *)c->host_failed != 0; }

// The caller has synchronised its own streams.  p2p: that only says MY kernels are through -- a peer may still be reading my
// staging half (one-shot step 2, all-gather step 4) and will still write its DONE words into my region, so the ranks meet on the
// host before anybody unmaps or frees (cn_comm_destroy is collective in this mode; a failed communicator skips the meeting).
void ipc_comm_destroy(IpcComm *c)
{
    if (!c) return;
    if (c->p2p && c->region && c->sh && !c->sh->failed.load() && !p2p_failed_word(c)) {
        try { c->barrier("cn_comm_destroy (p2p: nobody reads a peer's region any more)"); }
        catch (const std::exception &e) { fprintf(stderr, "%s\n", e.what()); }
    }
    if (c->host_failed) (void)hipHostFree(c->host_failed);
    for (int r = 0; r < c->world; ++r)
        if (r != c->rank && c->peer_region[r]) (void)hipIpcCloseMemHandle(c->peer_region[r]);
    if (c->region) (void)hipFree(c->region);
    for (int r = 0; r < c->world; ++r)
        if (r != c->rank && c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    if (c->staging) (void)hipFree(c->staging);
    if (c->sh) munmap(c->sh, sizeof(Shared));
    delete c;
}

void ipc_comm_mark_failed(IpcComm *c) { if (c && c->sh) c->sh->failed.store(1); }

// p2p: (re)allocate my region for buckets of up to `want` floats and map everybody's.  Every rank gets here in the same call
// (the networks are replicas: same bucket sizes in the same order), so the three host barriers pair up; the exchange counter
// starts over with the fresh (zeroed) flag words.
static void p2p_grow(IpcComm *c, size_t want, hipStream_t st)
{
    hipck(hipStreamSynchronize(st), "hipStreamSynchronize");
    c->barrier("cn_allreduce_grads (p2p: everybody idle)");           // nobody reads or writes the old regions any more
    for (int r = 0; r < c->world; ++r)
        if (r != c->rank && c->peer_region[r]) { hipck(hipIpcCloseMemHandle(c->peer_region[r]), "hipIpcCloseMemHandle"); c->peer_region[r] = nullptr; }
    if (c->region) hipck(hipFree(c->region), "hipFree");
    c->region = nullptr;
    // a half is world x P2P_GROUPS slots; a slot holds the largest piece (rounded up to 4 floats)
    const size_t slots = (size_t)c->world * P2P_GROUPS;
    const size_t cap = ((want + slots - 1) / slots + 4) * slots;
    const size_t bytes = (size_t)P2P_FLAG_WORDS * 8 + 2 * cap * sizeof(float);
    // fine-grained: flag words and staged gradients are read by other devices while kernels of both sides run (coarse-grained
    // memory is only coherent across devices at kernel boundaries); plain device memory if the runtime refuses
    void *p = nullptr;
    bool fine = true;
    if (c->coarse) {                      // asked for (an A/B switch for one-device boxes): never a silent degradation
        hipck(hipMalloc(&p, bytes), "hipMalloc(p2p region)");
        fine = false;
    } else {
        const hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            throw std::runtime_error(std::string("p2p communicator: no fine-grained device memory for the exchange region (") + hipGetErrorString(e) +
                                     "); coarse-grained memory is not coherent across devices while kernels run -- set CN_P2P_COARSE=1 only for ranks that share one device");
        }
    }
    if (c->verbose)
        fprintf(stderr, "p2p communicator: rank %d of %d: region of %zu bytes (%s), staging halves of %zu floats\n", c->rank, c->world, bytes,
                fine ? "fine-grained" : "coarse-grained", cap);
    c->region = (unsigned long long *)p; c->half_cap = cap; c->seq = 0;
    hipck(hipMemset(p, 0, (size_t)P2P_FLAG_WORDS * 8), "hipMemset");
    hipck(hipDeviceSynchronize(), "hipDeviceSynchronize");
    Slot &mine = c->sh->slot[c->rank];
    hipck(hipIpcGetMemHandle(&mine.handle, p), "hipIpcGetMemHandle");
    mine.capacity.store(cap);
    c->barrier("cn_allreduce_grads (p2p: regions published)");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) { c->peer_region[r] = c->region; continue; }
        Slot &s = c->sh->slot[r];
        if (s.capacity.load() != cap) throw std::runtime_error("ipc communicator: ranks disagree about the size of an exchange");
        hipIpcMemHandle_t h = s.handle;
        void *q = nullptr;
        hipck(hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
        c->peer_region[r] = (unsigned long long *)q;
    }
    c->barrier("cn_allreduce_grads (p2p: regions mapped)");
}

void ipc_comm_check_fast(IpcComm *c)
{
    if (c && c->p2p && p2p_failed_word(c)) {
        c->sh->failed.store(1);
        throw std::runtime_error("p2p communicator: rank " + std::to_string(c->rank) + " of " + std::to_string(c->world) + " waited more than " +
                                 std::to_string((int)c->timeout_s) + " s for a peer inside a gradient exchange (the gradient of that exchange was set to NaN)");
    }
}

// one exchange kernel; form: 0 = by size, 1 = one shot, 2 = reduce-scatter + all-gather
static void p2p_launch(IpcComm *c, float *buf, size_t n, hipStream_t st, size_t hint, int form, double timeout_s)
{
    const size_t pieces = (size_t)c->world * P2P_GROUPS;
    const size_t piece = ((n + pieces - 1) / pieces + 3) / 4 * 4;
    // the regrow test IS the launch condition (piece <= slot): slot rounds down, piece rounds up
    if (!c->region || piece > c->half_cap / pieces / 4 * 4) p2p_grow(c, n > hint ? n : hint, st);
    P2pArgs a{};
    a.buf = buf; a.n = n; a.me = c->rank; a.world = c->world;
    a.piece = piece;
    a.slot = c->half_cap / pieces / 4 * 4;
    if (a.piece > a.slot) throw std::runtime_error("p2p communicator: bucket larger than the staging half");
    a.two_phase = form ? form == 2 : (c->force_two_phase || (c->world > 2 && n > c->oneshot_max));
    a.seq = ++c->seq;
    a.timeout_ticks = (unsigned long long)(timeout_s * 1e8);
    a.host_failed = c->host_failed_dev;
    for (int r = 0; r < c->world; ++r) {
        a.flags[r] = c->peer_region[r];
        a.stage[r] = (float *)(c->peer_region[r] + P2P_FLAG_WORDS) + (a.seq & 1) * c->half_cap;
    }
    launch_p2p_allreduce(st, a);
    hipck(hipGetLastError(), "p2p all-reduce kernel");
}

static void p2p_allreduce(IpcComm *c, float *buf, size_t n, hipStream_t st, size_t hint)
{
    ipc_comm_check_fast(c);               // a communicator that failed does not take another gradient
    if (n == 0) return;
    p2p_launch(c, buf, n, st, hint, 0, c->timeout_s);
}

// First contact (cn_comm_init): regions allocated and mapped, then both forms of the exchange on a bucket with a known answer --
// every (rank, workgroup) flag word and one staged line per (peer, workgroup) cross every mapping, under the device-side
// time-out.  The verdict is shared: if ANY rank saw a wrong sum or a time-out, every rank returns false and carries the id rank 0
// made for the fall-back communicator.
bool ipc_comm_p2p_selfcheck(IpcComm *c, hipStream_t st, void (*make_id)(char *), char *rccl_id)
{
    const int W = c->world;
    const size_t n = (size_t)W * P2P_GROUPS * 16 + 3;            // 16 floats (one 64-byte line) per piece and an odd tail
    std::string why;
    float *d = nullptr;
    // (an exception in here leaves the ranks out of step inside the host barriers: it fails the communicator for everybody)
    try { p2p_grow(c, 64 * 1024, st); }
    catch (...) { c->sh->failed.store(1); throw; }
    try {
        std::vector<float> h(n), back(n);
        hipck(hipMalloc((void **)&d, n * sizeof(float)), "hipMalloc(self-check)");
        const double t_check = c->timeout_s < 15.0 ? c->timeout_s : 15.0;
        for (int form = 1; form <= 2 && why.empty(); ++form) {
            for (size_t i = 0; i < n; ++i) h[i] = (float)((c->rank + 1) * (int)(i % 251 + 1) + form);
            hipck(hipMemcpyAsync(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice, st), "hipMemcpyAsync");
            p2p_launch(c, d, n, st, 0, form, t_check);
            hipck(hipMemcpyAsync(back.data(), d, n * sizeof(float), hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
            hipck(hipStreamSynchronize(st), "hipStreamSynchronize");
            if (p2p_failed_word(c)) { why = "a peer's flag did not arrive"; break; }
            for (size_t i = 0; i < n; ++i) {
                const float want = (float)(W * (W + 1) / 2 * (int)(i % 251 + 1) + W * form);     // small integers: exact in fp32, any order
                if (back[i] != want) {
                    why = std::string(form == 1 ? "one-shot" : "two-phase") + " sum wrong at element " + std::to_string(i) + ": " + std::to_string(back[i]) + " instead of " + std::to_string(want);
                    break;
                }
            }
        }
    } catch (const std::exception &e) { why = e.what(); }
    if (d) (void)hipFree(d);
    if (const char *inj = getenv("CN_P2P_SELFCHECK_FAIL"))       // fault injection for the fail-over test: "all" or a rank
        if (!strcmp(inj, "all") || atoi(inj) == c->rank) why = "fault injected by CN_P2P_SELFCHECK_FAIL";
    if (!why.empty()) {
        fprintf(stderr, "p2p communicator: rank %d of %d: first-contact self-check FAILED (%s)\n", c->rank, W, why.c_str());
        c->sh->selfcheck_bad.store(1);
    }
    c->barrier("cn_comm_init (p2p: self-check verdicts)");
    if (!c->sh->selfcheck_bad.load()) return true;
    if (c->rank == 0) make_id(c->sh->rccl_id);
    c->barrier("cn_comm_init (p2p: fall-back id)");
    memcpy(rccl_id, c->sh->rccl_id, sizeof(c->sh->rccl_id));
    c->barrier("cn_comm_init (p2p: fall-back id read)");         // (also the meeting ipc_comm_destroy would otherwise hold)
    c->sh->failed.store(1);                                      // the p2p communicator is abandoned: its destroy does not wait for anybody
    return false;
}

// p2p: did a poll time out on the device?  (host-synchronising: called where the host waits anyway)
void ipc_comm_check(IpcComm *c, hipStream_t st)
{
    if (!c || !c->p2p || !c->region) return;
    unsigned long long failed = 0;
    hipck(hipMemcpyAsync(&failed, c->region + P2P_FAILED, 8, hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
    hipck(hipStreamSynchronize(st), "hipStreamSynchronize");
    if (failed || p2p_failed_word(c)) {
        c->sh->failed.store(1);
        throw std::runtime_error("p2p communicator: rank " + std::to_string(c->rank) + " of " + std::to_string(c->world) + " waited more than " +
                                 std::to_string((int)c->timeout_s) + " s for a peer inside a gradient exchange");
    }
}

// all-reduce(SUM, fp32, in place) of buf[0..n) over the ranks; `st` is the stream the caller ordered behind the producer of
// buf.  ipc mode blocks the host; p2p mode enqueues one kernel (capacity_hint: the largest bucket the caller will ever bring)
void ipc_allreduce(IpcComm *c, float *buf, size_t n, hipStream_t st, size_t capacity_hint)
{
    if (c->p2p) { p2p_allreduce(c, buf, n, st, capacity_hint); return; }
    if (n == 0) { c->barrier("cn_allreduce_grads"); c->barrier("cn_allreduce_grads"); return; }
    Slot &mine = c->sh->slot[c->rank];
    if (n > c->capacity) {
        if (c->staging) hipck(hipFree(c->staging), "hipFree");
        const size_t cap = n + n / 2 + 1024;
        hipck(hipMalloc((void **)&c->staging, cap * sizeof(float)), "hipMalloc(staging)");
        c->capacity = cap;
        hipck(hipIpcGetMemHandle(&mine.handle, c->staging), "hipIpcGetMemHandle");
        mine.capacity.store(cap);
        mine.generation.fetch_add(1);
    }
    hipck(hipMemcpyAsync(c->staging, buf, n * sizeof(float), hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
    hipck(hipStreamSynchronize(st), "hipStreamSynchronize");
    c->barrier("cn_allreduce_grads (gradients staged)");
    SumRanks sr{};
    sr.n = c->world;
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) { sr.src[r] = c->staging; continue; }
        Slot &s = c->sh->slot[r];
        const unsigned long long gen = s.generation.load();
        if (gen != c->peer_gen[r]) {
            if (c->peer[r]) hipck(hipIpcCloseMemHandle(c->peer[r]), "hipIpcCloseMemHandle");
            c->peer[r] = nullptr;
            hipIpcMemHandle_t h = s.handle;
            hipck(hipIpcOpenMemHandle((void **)&c->peer[r], h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
            c->peer_gen[r] = gen;
        }
        if (s.capacity.load() < n) throw std::runtime_error("ipc communicator: ranks disagree about the size of an exchange");
        sr.src[r] = c->peer[r];
    }
    launch_sum_ranks(st, buf, sr, n);
    hipck(hipGetLastError(), "sum kernel");
    hipck(hipStreamSynchronize(st), "hipStreamSynchronize");
    c->barrier("cn_allreduce_grads (sums formed)");      // nobody overwrites a staging buffer a peer still reads
}

// sums of (error, #correct) over the ranks, in rank order on every rank
void ipc_allreduce_loss(IpcComm *c, float *err, int *correct)
{
    c->sh->slot[c->rank].err = *err;
    c->sh->slot[c->rank].correct = *correct;
    c->barrier("cn_loss_read_global");
    float e = 0.f; int k = 0;
    for (int r = 0; r < c->world; ++r) { e += c->sh->slot[r].err; k += c->sh->slot[r].correct; }
    c->barrier("cn_loss_read_global");
    *err = e; *correct = k;
}

}  // namespace cn

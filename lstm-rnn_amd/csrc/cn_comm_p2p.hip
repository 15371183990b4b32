// CN_COMM_BACKEND=p2p -- the gradient exchange as ONE stream-ordered kernel per bucket over peer-mapped memory (no library, no
// host barrier): the second backend of cn_allreduce_grads beside RCCL ("direct reduce-scatter + all-gather over the
// full mesh" -- every GPU of an MI355X node has its own xGMI link to every other one, so a 0.66-1.5 MB per-layer bucket is
// W - 1 concurrent point-to-point reads, not a ring of 2 (W - 1) dependent hops).  What is summed: the weightUpdates of a
// layer over the ranks' shards of a fraction (LstmLayer.cu:502-510, FeedForwardLayer.cu:94-100 sum over ALL patterns).
//
// Every rank owns a REGION (cn_comm_ipc.cpp allocates it and maps the peers' through hipIpc handles): flag words + two staging
// halves (exchange k uses half k & 1).  The bucket is cut into W x G pieces (W ranks, G workgroups per rank): piece (s, b) is
// sub-slice b of slice s.  Workgroup b of every rank works on the pieces (*, b) only, so one flag word per (rank, workgroup)
// orders everything -- there is no grid-wide step inside the launch:
//   0. wait until workgroup b of every rank has finished exchange k - 2 (their `done` words, written into MY region: all polls
//      are local reads), then copy my pieces (*, b) into the slots (*, b) of my staging half -- a half is W x G slots of fixed
//      size, so whatever the bucket's length only the workgroups b ever touch the slots (*, b);
//   1. release, write `ready[me][b] = k` into every rank's region;
//   small buckets (one shot):  2. wait for ready[r][b] of all r, add the pieces (*, b) of all ranks IN RANK ORDER into my
//      gradient;
//   large buckets (reduce-scatter + all-gather):  2. wait as above, add piece (me, b) of all ranks in rank order, write it to my
//      gradient AND over piece (me, b) of my staging half (peers only read THEIR slices of it until step 3);  3. release, write
//      `reduced[me][b] = k` to everybody;  4. for every other slice s: wait for reduced[s][b], copy piece (s, b) from rank s;
//   last: write `done[me][b] = k` to everybody.
// Every rank forms (or receives) each sum from the same numbers in the same order: the replicas stay bit-identical, as with
// RCCL's ring and with the ipc test backend.
// Memory model: the regions are fine-grained allocations and EVERY access to a staging half or a flag word is a relaxed
// system-scope atomic (8 bytes; `sc0 sc1` on the instruction: written through to / read from memory, never from a non-coherent
// cache line), so "release" is: EVERY wave waits for its own outstanding stores with an explicit `s_waitcnt vmcnt(0)` (gfx9 counts
// stores in vmcnt), then the workgroup barrier, then the flag store; "acquire" is the barrier behind the poll.  The wait is
// written out in asm: a workgroup-scope release fence compiles to NO vmcnt wait on gfx950 (round 5 shipped exactly that, the ISA
// showed `s_waitcnt lgkmcnt(0); s_barrier` only, so the staging stores of waves 1-3 were unordered against wave 0's flag store
// into the peer's region; tests/test_abi_and_host.py now reads the ISA for the wait).  No L2 write-back or invalidate, which a
// system-scope fence costs every time while the backward kernels beside the exchange keep the L2s full of dirty lines (first
// version: 37 us per exchange on one rank, most of it in `buffer_wbl2`).  The gradient itself (a.buf) is ordinary stream-ordered
// memory.
// Flags only grow (k is the communicator's exchange counter); a poll that sees nothing for `timeout` ticks of the 100 MHz clock
// sets the region's `failed` word (and the peers'), after which no poll of the communicator waits any more.  A workgroup whose
// wait ended that way does NOT go on summing whatever the peers' halves hold: it overwrites its pieces of the gradient with NaN,
// sets the host-mapped word `a.host_failed` and leaves -- an update that follows cannot quietly apply a partial sum, and the host
// raises CN_ERR_COMM at its next call into the communicator (cn_allreduce_grads, cn_sgd_update*, cn_loss_read_global,
// cn_comm_destroy).
#include "cn_internal.h"

namespace cn {

typedef unsigned long long u64;

namespace {

__device__ __forceinline__ void put(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u64 get(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float2 as_f2(u64 v) { float2 f; __builtin_memcpy(&f, &v, 8); return f; }
__device__ __forceinline__ u64 as_u64(float2 f) { u64 v; __builtin_memcpy(&v, &f, 8); return v; }

// thread-level wait: *p >= want (returns false), or the communicator failed, or the deadline passed (then it fails the
// communicator); both of those return true
__device__ __forceinline__ bool flag_wait(const P2pArgs &a, const u64 *p, u64 want)
{
    const u64 *failed = a.flags[a.me] + P2P_FAILED;
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    int spins = 0;
    while (get(p) < want) {
        if ((++spins & 63) == 0) {
            if (get(failed)) return true;
            if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                for (int r = 0; r < a.world; ++r) put(a.flags[r] + P2P_FAILED, 1);
                return true;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// every wave: my stores (staging half, gradient) have been acknowledged and my loads have returned.  gfx9 counts both in vmcnt.
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// this workgroup's stores to its staging half have completed (and its loads from the peers' have returned) before thread
// r < world writes `word[me][b] = k` into rank r's region: drain in EVERY wave, then the barrier, then the flag
__device__ __forceinline__ void signal_all(const P2pArgs &a, int word, int b)
{
    drain_vmem();
    __syncthreads();
    if ((int)threadIdx.x < a.world) put(a.flags[threadIdx.x] + word + a.me * P2P_GROUPS + b, a.seq);
}
// wait for `word[r][b] >= want` of all ranks r (only >= 0: of that rank).  Block-uniform result: true = a wait of this workgroup
// ended without its flag (time-out here or a failed communicator)
__device__ __forceinline__ bool wait_ranks(const P2pArgs &a, int word, int b, int only, u64 want)
{
    const int t = threadIdx.x;
    int bad = 0;
    if (only >= 0) { if (t == 0) bad = flag_wait(a, a.flags[a.me] + word + only * P2P_GROUPS + b, want); }
    else if (t < a.world) bad = flag_wait(a, a.flags[a.me] + word + t * P2P_GROUPS + b, want);
    return __syncthreads_or(bad) != 0;
}

__global__ __launch_bounds__(P2P_THREADS) void p2p_allreduce_kernel(P2pArgs a)
{
    const int b = blockIdx.x, t = threadIdx.x, W = a.world, me = a.me;
    const size_t piece = a.piece, slot = a.slot;       // floats (even): piece <= slot
    u64 *mine = (u64 *)a.stage[me];
    // piece (s, b): bucket elements [lo, hi), staged in slot (s, b) of a half; pairs of floats, a last odd element rides in a
    // pair whose other half is never used
    #define P2P_PIECE(s) const size_t lo = ((size_t)(s) * P2P_GROUPS + b) * piece, hi = lo + piece < a.n ? lo + piece : a.n, \
                                      base = ((size_t)(s) * P2P_GROUPS + b) * slot / 2, pairs = hi > lo ? (hi - lo + 1) / 2 : 0

    // a wait that ended without its flag: pieces (*, b) of the gradient -- the ones this workgroup writes in either form -- become
    // NaN (some may hold sums already, some not: none of it may be used), the host-mapped word tells the host, nothing is signalled
    #define P2P_FAIL_IF(cond) if (cond) { \
        for (int s = 0; s < W; ++s) { P2P_PIECE(s); (void)base; (void)pairs; for (size_t e = lo + t; e < hi; e += P2P_THREADS) a.buf[e] = __builtin_nanf(""); } \
        if (t == 0 && a.host_failed) put(a.host_failed, 1); \
        return; }

    // 0. slot (*, b) of this half is free once the workgroups b of all ranks have finished exchange k - 2
    if (a.seq > 2) P2P_FAIL_IF(wait_ranks(a, P2P_DONE, b, -1, a.seq - 2))
    for (int s = 0; s < W; ++s) {
        P2P_PIECE(s);
        for (size_t i = t; i < pairs; i += P2P_THREADS) {
            const size_t e = lo + 2 * i;
            float2 v; v.x = a.buf[e]; v.y = e + 1 < hi ? a.buf[e + 1] : 0.f;
            put(mine + base + i, as_u64(v));
        }
    }
    // 1.
    signal_all(a, P2P_READY, b);
    // 2.
    P2P_FAIL_IF(wait_ranks(a, P2P_READY, b, -1, a.seq))
    const int s_first = a.two_phase ? me : 0, s_last = a.two_phase ? me + 1 : W;
    for (int s = s_first; s < s_last; ++s) {
        P2P_PIECE(s);
        // four pairs per thread and trip, all their W loads in flight at once (a peer's memory is microseconds away), added in
        // rank order afterwards
        for (size_t i0 = t; i0 < pairs; i0 += 4 * P2P_THREADS) {
            u64 x[4][8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t i = i0 + j * P2P_THREADS;
                if (i < pairs) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) if (r < W) x[j][r] = get((const u64 *)a.stage[r] + base + i);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t i = i0 + j * P2P_THREADS;
                if (i < pairs) {
                    float2 v = as_f2(x[j][0]);
#pragma unroll
                    for (int r = 1; r < 8; ++r) if (r < W) { const float2 y = as_f2(x[j][r]); v.x += y.x; v.y += y.y; }
                    const size_t e = lo + 2 * i;
                    a.buf[e] = v.x; if (e + 1 < hi) a.buf[e + 1] = v.y;
                    if (a.two_phase) put(mine + base + i, as_u64(v));
                }
            }
        }
    }
    if (a.two_phase) {
        // 3., 4.: the other slices, starting with my right-hand neighbour's (the ranks do not all pull from rank 0 first)
        signal_all(a, P2P_REDUCED, b);
        for (int d = 1; d < W; ++d) {
            const int s = (me + d) % W;
            P2P_FAIL_IF(wait_ranks(a, P2P_REDUCED, b, s, a.seq))
            const u64 *theirs = (const u64 *)a.stage[s];
            P2P_PIECE(s);
            for (size_t i0 = t; i0 < pairs; i0 += 4 * P2P_THREADS) {
                u64 x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) if (i0 + j * P2P_THREADS < pairs) x[j] = get(theirs + base + i0 + j * P2P_THREADS);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const size_t i = i0 + j * P2P_THREADS;
                    if (i < pairs) { const float2 v = as_f2(x[j]); const size_t e = lo + 2 * i; a.buf[e] = v.x; if (e + 1 < hi) a.buf[e + 1] = v.y; }
                }
            }
        }
    }
    // last: my reads of the peers' halves have returned
    signal_all(a, P2P_DONE, b);
    #undef P2P_FAIL_IF
    #undef P2P_PIECE
}

}  // namespace

void launch_p2p_allreduce(hipStream_t s, const P2pArgs &a)
{
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(P2P_GROUPS), dim3(P2P_THREADS), 0, s, a);
}

}  // namespace cn

// comm.hip -- data-parallel gradient all-reduce over RCCL / xGMI (new: the
// reference is single-process).  One process per GPU; the
// communicator is bootstrapped from a 128-byte unique id shipped out of band.
// The all-reduce runs on the ctx stream so it can sit inside the captured
// step graph between loss.backward() and optim.step().
#include "common.h"

#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include <unistd.h>

#include "adam_dev.h"
#include "dp_dev.h"

constexpr int P2P_MAX_RANKS = 8;           // one node: 8 x MI355X, full xGMI mesh
constexpr int P2P_FLAG_WORDS = 32;         // per rank: ready[8] | done[8] | pad (uint32 step values, monotonic)
constexpr int P2P_READY = 0, P2P_DONE = 8;
constexpr int P2P_MAX_BLOCKS = 256, P2P_QUADS = 8;   // in-place form: <= 8 float4 per thread held across the barrier
constexpr long P2P_DEFAULT_TIMEOUT_MS = 120000;       // spins are bounded by the 100 MHz wall clock: a lost peer ends in an error, not a hang
                                                      // (TAPER_P2P_TIMEOUT_MS / th_comm_set_timeout_ms; long enough for a peer that saves a checkpoint)

struct P2PBlob {                           // what a rank ships to its peers (TH_P2P_BLOB_BYTES)
    hipIpcMemHandle_t buf, flags;          // allocation bases
    uint64_t buf_offset, n;                // the reduced buffer inside its allocation; length in floats
    int32_t pid, device, rank, pad;
    char bus_id[16];                       // PCI bus id of the rank's device ("0000:05:00.0"): two ranks with the same id share a GPU
    uint8_t reserved[TH_P2P_BLOB_BYTES - 2 * sizeof(hipIpcMemHandle_t) - 2 * sizeof(uint64_t) - 4 * sizeof(int32_t) - 16];
};
static_assert(th::DP_DATA_OFFSET >= P2P_MAX_RANKS * P2P_FLAG_WORDS * sizeof(uint32_t), "the three-launch flag block sits in front of the exchange region");
static_assert(sizeof(P2PBlob) == TH_P2P_BLOB_BYTES, "P2PBlob layout");

struct P2PDev {                            // kernel argument
    const float *buf[P2P_MAX_RANKS];       // buf[r]: rank r's gradient buffer as mapped here (own = local pointer)
    uint32_t *flags[P2P_MAX_RANKS];        // flags[r]: rank r's flag block as mapped here (own = local pointer)
    uint32_t *state;                       // local: [0] steps completed, [1] arrival counter, [2] error (timeout; STICKY), [3] READY verdict of the step
                                           // in flight ((step << 1) | ok), [4] DONE verdict (in-place form), [5] "dead at entry": [2] as it stood when
                                           // the previous launch ended -- what a launch's workgroups consult, so that they all see the same value
    uint32_t *err_host;                    // the same error word in device-visible pinned host memory: the host reads it after any stream sync
    long spin_ticks;
    int n_ranks, rank;
};

struct th_comm {
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0;
    // ---- peer-to-peer form (th_comm_init_p2p) ----
    bool p2p = false, connected = false;
    int device = 0;
    float *reg_buf = nullptr;              // the registered buffer (this rank's flat gradient arena)
    size_t reg_n = 0;
    uint32_t *flags_local = nullptr, *state = nullptr, *err_host = nullptr;
    long timeout_ms = P2P_DEFAULT_TIMEOUT_MS;
    void *peer_base[P2P_MAX_RANKS] = {};   // hipIpcOpenMemHandle results (to close)
    void *peer_flag_base[P2P_MAX_RANKS] = {};
    P2PDev dev{};
    long launches_inplace = 0, launches_fused = 0;   // enqueued (or captured) one-shot launches, for tests
    // ---- exchange inside the gradient launch (dp_dev.h): the region behind flags_local, mapped by every peer with the flag block ----
    th::DpDev dp{};
    bool loopback = false;                 // two "ranks", both this one: every push, flag, poll and load of the protocol through local memory
    int sharing = 1;                       // ranks on this rank's device (itself included)
    char bus_id[16] = {};
    long launches_inkernel = 0;
};

namespace th {
const DpDev *comm_dp_dev(const th_comm *c) { return (c && c->p2p && c->connected) ? &c->dp : nullptr; }
int comm_dp_sharing(const th_comm *c) { return c ? c->sharing : 1; }
// Ranks that SHARE a device (a test box; a deployment runs one rank per GPU) wait in each other's way: a workgroup that waits for a peer's
// slice holds its place, and the rank that is behind still has to get the launches IN FRONT of its exchange launch dispatched.  Two conditions:
//   places   (ranks on the device - 1) x grid < occupancy x CUs: the late rank's own exchange workgroups always find a place, so they
//            never wait for anybody who is not already resident and the wait graph has no cycle (necessary);
//   engines  (ranks on the device - 1) x ceil(grid / 32) <= 7: measured on gfx950 (r06, device-side placement trace,
//            profiles/r06_dp_three_ranks_one_device.txt) -- a launch's workgroups are dealt to an XCD's four shader engines (8 CUs each) in
//            strict rotation, one waiting workgroup per CU, and a launch whose next workgroup is a whole-CU one (the MLP's first launch: 1 024
//            threads; the conv chain) STALLS, in order, when the engine whose turn it is has no free CU.  One waiting rank leaves every
//            engine 1 - 2 free CUs (its own workgroups were dealt evenly); two ranks of 13 - 14 workgroups per XCD each fill an engine
//            (4 + 4) in half the runs, and the late rank's first launch then never finishes: all three time out.  The bound keeps one CU
//            free in every engine, counting every waiting workgroup as a whole CU.
// TAPER_DP_SHARED_RULE=places drops the second condition (the four-rank protocol test: its waiting workgroups are small enough to share
// CUs with the first launch's, 48 of 48 runs).
bool comm_dp_shared_fits(const th_comm *c, int grid, int per_cu) {
    const int sharing = c ? c->sharing : 1;
    if (sharing <= 1) return true;
    if ((long)(sharing - 1) * grid >= (long)per_cu * kNumCU) return false;
    static const bool places_only = [] { const char *e = std::getenv("TAPER_DP_SHARED_RULE"); return e && std::strcmp(e, "places") == 0; }();
    if (places_only) return true;
    constexpr int kEngines = kNumXCD * 4, kCUsPerEngine = kNumCU / kEngines;
    return (long)(sharing - 1) * ((grid + kEngines - 1) / kEngines) <= kCUsPerEngine - 1;
}
void comm_dp_count_launch(th_comm *c) { if (c) ++c->launches_inkernel; }
}

namespace th {
int scale_inplace(th_ctx *ctx, float *d_x, size_t n, float scale);
}

#define TH_NCCL(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) {                                                                       \
            th::set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
            return 1;                                                                                  \
        }                                                                                              \
    } while (0)

namespace th {

// ------------------------------------------------------------------------------------------------
// One-shot peer-to-peer all-reduce for LATENCY-bound buffers (the MLP's 407 KB gradient arena: a ring spends
// 2(W-1) dependent hops on it).  Every rank's buffer and flag block are mapped into every peer (IPC handles);
// one launch per rank:
//   1. workgroup 0 PUSHES "my gradients are complete" (the launch sits behind the backward kernels in the
//      stream, whose writes reached memory at their kernel boundaries) into every peer's flag block;
//   2. every workgroup polls its OWN flag block (local memory, system-scope acquire) until all W ranks are ready;
//   3. each thread reads its elements from all W buffers over xGMI and adds them in RANK ORDER -- every rank
//      computes the same sum bit for bit, so replicas stay identical -- then scales;
//   4. FUSED: Adam (optim.rs:83-113) is applied to this rank's p / m / v straight from the registers: the
//      reduced gradient is never written.  IN PLACE: the sums wait in registers;
//   5. the last workgroup to finish pushes "done reading" to every peer and advances the local step count;
//      nobody may overwrite its buffer before every peer is done: the in-place form waits for that before it
//      stores, the fused form before workgroup 0 exits (the next backward launch is behind this one).
// Flags are monotonic step numbers, so nothing is ever reset and a captured graph replays the launch as is.
// Spins are bounded by the wall clock: a missing peer raises state[2] instead of hanging the GPU.
__device__ __forceinline__ bool p2p_wait_all(const P2PDev &c, int word0, uint32_t step) {
    // lanes 0..W-1 of the first wave each watch one rank's slot in the LOCAL flag block
    bool ok = true;
    if (threadIdx.x < (unsigned)c.n_ranks) {
        const uint32_t *slot = c.flags[c.rank] + word0 + threadIdx.x;
        const long t0 = wall_clock64();
        while ((int32_t)(__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - step) < 0) {
            if (wall_clock64() - t0 > c.spin_ticks) {
                __hip_atomic_store(&c.state[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(c.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    ok = __syncthreads_and(ok);                       // the verdict of the watching lanes, for the whole workgroup
    // No acquire fence here (system scope = an invalidate walk of this XCD's L2 in every workgroup: 3.3 of the launch's 13.6 us with one
    // rank): what must not be stale are the PEERS' arenas, and those are read with system-coherent loads (ld_sys: sc0 sc1, served by memory,
    // never by a cache of this device) issued after this point -- the compiler may not move them up (asm volatile behind the barrier).
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    return ok;
}

// ONE verdict per launch and phase.  Workgroup 0 alone watches the flag block against the clock and publishes (step << 1) | ok in
// state[word]; every other workgroup waits for that word.  (Each workgroup polling -- and timing out -- on its own let a peer that arrived
// at the bound be "in time" for some workgroups and "late" for others: a partly applied update.)  The word is local device memory read
// with agent-scope atomics: it costs the followers one L2 round trip after workgroup 0's store.  All workgroups of these launches are
// resident at once (<= 256), so workgroup 0 always runs.
__device__ __forceinline__ bool p2p_verdict(const P2PDev &c, int flag_word0, int state_word, uint32_t step) {
    if (blockIdx.x == 0) {
        const bool ok = p2p_wait_all(c, flag_word0, step);
        if (threadIdx.x == 0) __hip_atomic_store(&c.state[state_word], (step << 1) | (ok ? 1u : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return ok;
    }
    __shared__ uint32_t verdict;
    if (threadIdx.x == 0) {
        uint32_t v;
        while (((v = __hip_atomic_load(&c.state[state_word], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) >> 1) != (step & 0x7fffffffu)) __builtin_amdgcn_s_sleep(1);
        verdict = v & 1u;
    }
    __syncthreads();
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    return verdict != 0u;
}

// a time-out is final: once the error word is up every later launch of this communicator returns at once -- no flag is pushed, nothing is
// reduced, no parameter moves, Adam's counter stays -- and the host raises at its next look (th_comm_error / th_comm_error_peek).
// Consulted through state[5], the error word as the PREVIOUS launch left it: a raise in the middle of a launch cannot split that launch.
__device__ __forceinline__ bool p2p_dead(const P2PDev &c) {
    return __hip_atomic_load(&c.state[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}
// workgroup 0, thread 0, the last thing it does
__device__ __forceinline__ void p2p_seal(const P2PDev &c) {
    const uint32_t e = __hip_atomic_load(&c.state[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e) __hip_atomic_store(&c.state[5], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void p2p_push(const P2PDev &c, int word0, uint32_t step) {
    // lane r writes this rank's slot in rank r's flag block (its own included).  The store is a system-scope RELEASE: the compiler puts
    // `buffer_wbl2 sc0 sc1` + `s_waitcnt vmcnt(0)` in front of it, i.e. everything this XCD's L2 still holds dirty goes to memory first; the
    // gradient arena itself was written by EARLIER launches, whose end-of-kernel release wrote every XCD's L2 back (the XCD L2s of one device
    // are not coherent with each other, so HIP's kernel boundary has to) -- a peer's system-coherent load then finds it in HBM / MALL.
    if (threadIdx.x < (unsigned)c.n_ranks)
        __hip_atomic_store(c.flags[threadIdx.x] + word0 + c.rank, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// arrival of this workgroup; the last one publishes the step and tells the peers this rank is done reading
__device__ __forceinline__ void p2p_arrive(const P2PDev &c, uint32_t step) {
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0) {
        // (nothing to release: "arrived" says this workgroup's READS of the peers' arenas have returned -- every thread consumed its
        // values before the barrier above -- and no peer reads anything this launch writes)
        const uint32_t arrived = __hip_atomic_fetch_add(&c.state[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = arrived == gridDim.x - 1;
        if (last) {
            __hip_atomic_store(&c.state[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&c.state[0], step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (last) p2p_push(c, P2P_DONE, step);
}

// a system-coherent 16-byte load: answered by memory (over xGMI for a peer's arena), not by this device's L1 / L2
typedef float p2p_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ p2p_f4 ld_sys(const float *p) {
    p2p_f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// NR = 1 / 2 / 4 / 8 >= n_ranks, compiled in: the NR requests and their wait are ONE basic block -- no branch between an inline-asm load and
// the wait that covers it, so what may touch a destination register before its data has landed can be read off the listing line by line
// (tools/asm_hazard_audit.py; r04's form looped to the run-time rank count with a `break`, and took this rank's own arena through an
// ordinary load behind a rank test: correct, but neither the compiler's nor the audit's reasoning covers loads and waits in different
// blocks).  All requests go out before the first value is used.  Slots r >= n_ranks re-read this rank's own arena and are dropped by a
// select; this rank's own arena (written by the backward launches in front of this one in the stream, written back at their kernel
// boundaries) takes the same system-coherent load as the peers'.  Added in RANK ORDER: the same bits on every rank.
template <int NR>
__device__ __forceinline__ float4 p2p_sum_quad(const P2PDev &c, long i, float scale) {
    p2p_f4 v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = ld_sys((r < c.n_ranks ? c.buf[r] : c.buf[c.rank]) + i);
#pragma unroll
    for (int r = 0; r < NR; ++r) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[r]) : : "memory");
    p2p_f4 s = v[0];
#pragma unroll
    for (int r = 1; r < NR; ++r) s = r < c.n_ranks ? s + v[r] : s;
    s *= scale;
    return make_float4(s.x, s.y, s.z, s.w);
}

// in place: buf[rank][0..n) = scale * sum_r buf[r][0..n); n % 4 == 0, 16-byte aligned
template <int NR>
__global__ __launch_bounds__(256) void p2p_allreduce_kernel(P2PDev c, float *__restrict__ out, long n, float scale) {
    if (p2p_dead(c)) return;
    const uint32_t step = __hip_atomic_load(&c.state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (blockIdx.x == 0) p2p_push(c, P2P_READY, step);
    const bool ok = p2p_verdict(c, P2P_READY, 3, step);
    const long stride = (long)gridDim.x * 256 * 4;
    float4 acc[P2P_QUADS];
    if (ok) {
#pragma unroll
        for (int q = 0; q < P2P_QUADS; ++q) {
            const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4 + q * stride;
            if (i < n) acc[q] = p2p_sum_quad<NR>(c, i, scale);
        }
        p2p_arrive(c, step);
    }
    // every peer has read this rank's buffer: it may be overwritten -- by ALL workgroups or by none (one verdict)
    const bool done = ok && p2p_verdict(c, P2P_DONE, 4, step);
    if (blockIdx.x == 0 && threadIdx.x == 0) p2p_seal(c);
    if (!done) return;
#pragma unroll
    for (int q = 0; q < P2P_QUADS; ++q) {
        const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4 + q * stride;
        if (i < n) *reinterpret_cast<float4 *>(out + i) = acc[q];
    }
}

__device__ __forceinline__ int p2p_find_tensor(const int64_t *__restrict__ offsets, int n_tensors, int64_t i) {
    int lo = 0, hi = n_tensors;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// fused: the mean gradient goes straight from the W buffers into Adam (optim.rs:99-110) on this rank's p / m / v.
// Arena slices are padded to multiples of 4 floats, so a quad never straddles two tensors; grad-less tensors are
// skipped entirely (Q8; the mask is rank-invariant).  t: every workgroup forms t + 1 itself, workgroup 0 publishes it
// on its way out (after every other reader is long past the load: they all arrive before the done flags go out).
template <int NR>
__global__ __launch_bounds__(256) void p2p_allreduce_adam_kernel(P2PDev c, long n, float scale, float *__restrict__ p, float *__restrict__ m,
                                                                 float *__restrict__ v, const int64_t *__restrict__ offsets,
                                                                 const int32_t *__restrict__ has_grad, int n_tensors, int32_t *t_state,
                                                                 const float *__restrict__ lr, float beta1, float beta2, float eps, float wd,
                                                                 int pre_ticked) {
    if (p2p_dead(c)) return;
    const uint32_t step = __hip_atomic_load(&c.state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (blockIdx.x == 0) p2p_push(c, P2P_READY, step);
    const int t = __hip_atomic_load(&t_state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (pre_ticked ? 0 : 1);   // optim.rs:84
    const float step_size = adam_step_size(lr[0], beta1, beta2, t);
    const bool ok = p2p_verdict(c, P2P_READY, 3, step);   // all workgroups apply the update, or none does
    if (ok) {
        for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 256 * 4) {
            if (!has_grad[p2p_find_tensor(offsets, n_tensors, i)]) continue;
            // this rank's p / m / v are requested BEFORE the gradient loads and their wait (inline assembly the compiler moves nothing across):
            // behind it they were a second dependent round trip per quad -- 11.2 us for the launch with one rank against 8.6 (r05)
            const float4 pv = *reinterpret_cast<const float4 *>(p + i), mv = *reinterpret_cast<const float4 *>(m + i),
                         vv = *reinterpret_cast<const float4 *>(v + i);
            const float4 g = p2p_sum_quad<NR>(c, i, scale);
            float4 po, mo, vo;
#define TH_ADAM_LANE(k)                                                  \
            {                                                            \
                const float gg = g.k + wd * pv.k;                        \
                mo.k = beta1 * mv.k + (1.0f - beta1) * gg;               \
                vo.k = beta2 * vv.k + (1.0f - beta2) * gg * gg;          \
                po.k = pv.k - step_size * mo.k / (sqrtf(vo.k) + eps);    \
            }
            TH_ADAM_LANE(x) TH_ADAM_LANE(y) TH_ADAM_LANE(z) TH_ADAM_LANE(w)
#undef TH_ADAM_LANE
            *reinterpret_cast<float4 *>(m + i) = mo;
            *reinterpret_cast<float4 *>(v + i) = vo;
            *reinterpret_cast<float4 *>(p + i) = po;
        }
    }
    if (!ok) {      // READY timed out: nothing was read, nothing moved, the counter stays; the error word is up
        if (blockIdx.x == 0 && threadIdx.x == 0) p2p_seal(c);
        return;
    }
    p2p_arrive(c, step);
    if (blockIdx.x == 0) {
        // the next backward launch overwrites the buffer the peers are reading: wait for them.  A time-out HERE comes after a complete
        // update (every gradient was summed over all W ranks): p / m / v moved, so the counter moves with them (optim.rs:84) -- the state
        // stays consistent -- and the error word says that a peer may not have finished (the host raises: the replicas are out of step).
        (void)p2p_wait_all(c, P2P_DONE, step);
        if (threadIdx.x == 0) {
            if (!pre_ticked) __hip_atomic_store(&t_state[0], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p2p_seal(c);
        }
    }
}

// The in-launch exchange on its own: one workgroup per slot, thread t of slot s on rank r in round k holds the pattern below (small
// integers: every sum and mean is exact); each thread checks the mean it gets back.  Collective: every rank launches it, same slots and
// rounds.  The same flags / receive slots / step counter as a training step's launch (dp_dev.h); 2 values per thread like mlp_tail.hip.
__device__ __forceinline__ float dp_pattern(int r, int k, int s, int t, int j) { return (float)((r + 1) * (k + 1)) + (float)((s * 3 + t * 2 + j) % 7); }
__global__ void dp_bump_kernel(uint32_t *state) {
    if (state[DP_ST_DEAD] == 0u) dp_advance_step(state + DP_ST_STEP);
}
template <int NR>
__global__ __launch_bounds__(256) void dp_selftest_kernel(DpDev c, int round, int loopback, uint32_t *bad) {
    const DpTicket tk = dp_begin(c);
    const int s = blockIdx.x, t = threadIdx.x;
    float v[2] = {dp_pattern(c.rank, round, s, t, 0), dp_pattern(c.rank, round, s, t, 1)};
    const bool ok = dp_reduce<NR, 2>(c, tk, s, t, v);
    if (ok) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float want = 0.f;
            for (int r = 0; r < c.n_ranks; ++r) want += dp_pattern(loopback ? c.rank : r, round, s, t, j);
            want *= c.scale;
            if (v[j] != want) atomicAdd(bad, 1u);
        }
    } else if (t == 0) {
        atomicAdd(bad, 1u);
    }
}

static int p2p_grid(size_t n) {
    long g = (long)((n / 4 + 255) / 256);
    return (int)(g < 1 ? 1 : (g > 256 ? 256 : g));   // one workgroup per CU at most: every one of them polls the flag block
}

}  // namespace th

extern "C" {

int th_comm_unique_id(uint8_t out_id[128]) {
    TH_REQUIRE(out_id, "th_comm_unique_id: null out");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    TH_NCCL(ncclGetUniqueId(&id));
    memcpy(out_id, &id, sizeof(id));
    return 0;
}

int th_comm_init_rank(th_ctx *ctx, int n_ranks, int rank, const uint8_t id[128], th_comm **out) {
    TH_REQUIRE(ctx && id && out, "th_comm_init_rank: null argument");
    TH_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "th_comm_init_rank: bad rank %d of %d", rank, n_ranks);
    TH_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    th_comm *c = new th_comm();
    c->n_ranks = n_ranks;
    c->rank = rank;
    ncclResult_t r = ncclCommInitRank(&c->comm, n_ranks, uid, rank);
    if (r != ncclSuccess) {
        th::set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r));
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

int th_comm_destroy(th_comm *comm) {
    if (!comm) return 0;
    if (comm->comm) ncclCommDestroy(comm->comm);
    if (comm->p2p) {
        (void)hipSetDevice(comm->device);
        for (int r = 0; r < comm->n_ranks; ++r) {
            if (r == comm->rank) continue;
            if (comm->peer_base[r]) (void)hipIpcCloseMemHandle(comm->peer_base[r]);
            if (comm->peer_flag_base[r]) (void)hipIpcCloseMemHandle(comm->peer_flag_base[r]);
        }
        if (comm->flags_local) (void)hipFree(comm->flags_local);
        if (comm->state) (void)hipFree(comm->state);
        if (comm->err_host) (void)hipHostFree(comm->err_host);
    }
    delete comm;
    return 0;
}

int th_comm_init_p2p(th_ctx *ctx, int n_ranks, int rank, th_comm **out) {
    TH_REQUIRE(ctx && out, "th_comm_init_p2p: null argument");
    TH_REQUIRE(n_ranks >= 1 && n_ranks <= P2P_MAX_RANKS && rank >= 0 && rank < n_ranks, "th_comm_init_p2p: bad rank %d of %d (at most %d ranks: one node)",
               rank, n_ranks, P2P_MAX_RANKS);
    TH_HIP(hipSetDevice(ctx->device));
    th_comm *c = new th_comm();
    c->p2p = true;
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->device = ctx->device;
    // the flag block peers write into: fine-grained (uncached) device memory where the runtime offers it
    // (one allocation = one IPC handle: the three-launch form's flag block, then the in-launch exchange's flags and receive slots, dp_dev.h)
    void *f = nullptr;
    if (hipExtMallocWithFlags(&f, th::DP_REGION_BYTES, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc(&f, th::DP_REGION_BYTES) != hipSuccess) {
            delete c;
            TH_REQUIRE(false, "th_comm_init_p2p: cannot allocate the flag block");
        }
    }
    c->flags_local = (uint32_t *)f;
    if (const char *e = getenv("TAPER_P2P_TIMEOUT_MS")) {
        const long ms = atol(e);
        if (ms > 0) c->timeout_ms = ms;
    }
    if (hipHostMalloc((void **)&c->err_host, 64, hipHostMallocMapped) != hipSuccess) {
        c->err_host = nullptr;
        th_comm_destroy(c);
        TH_REQUIRE(false, "th_comm_init_p2p: cannot allocate the host-visible error word");
    }
    c->err_host[0] = 0;
    if (hipDeviceGetPCIBusId(c->bus_id, (int)sizeof(c->bus_id), ctx->device) != hipSuccess) {
        (void)hipGetLastError();
        snprintf(c->bus_id, sizeof(c->bus_id), "dev%d", ctx->device);
    }
    // (the three-launch form's flags start at zero; every word of the receive region starts EMPTY: all ones, dp_dev.h)
    if (hipMalloc((void **)&c->state, 64) != hipSuccess || hipMemset(c->flags_local, 0, th::DP_DATA_OFFSET) != hipSuccess ||
        hipMemset((char *)c->flags_local + th::DP_DATA_OFFSET, 0xFF, th::DP_REGION_BYTES - th::DP_DATA_OFFSET) != hipSuccess ||
        hipMemset(c->state, 0, 64) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        th_comm_destroy(c);
        TH_REQUIRE(false, "th_comm_init_p2p: cannot initialise the flag block");
    }
    *out = c;
    return 0;
}

int th_comm_p2p_export(th_comm *comm, float *d_buf, size_t n, uint8_t out_blob[TH_P2P_BLOB_BYTES]) {
    TH_REQUIRE(comm && comm->p2p && d_buf && out_blob, "th_comm_p2p_export: needs a peer-to-peer communicator and a buffer");
    TH_REQUIRE(n > 0 && n % 4 == 0 && ((uintptr_t)d_buf & 15) == 0, "th_comm_p2p_export: the buffer must be 16-byte aligned with a length that is a multiple of 4 floats");
    TH_REQUIRE(n <= (size_t)P2P_MAX_BLOCKS * 256 * 4 * P2P_QUADS, "th_comm_p2p_export: %zu floats is past the one-shot form (use the RCCL communicator)", n);
    TH_HIP(hipSetDevice(comm->device));
    P2PBlob b{};
    hipDeviceptr_t base = nullptr;
    size_t span = 0;
    TH_HIP(hipMemGetAddressRange(&base, &span, (hipDeviceptr_t)d_buf));     // the pool hands out slices of larger allocations
    TH_HIP(hipIpcGetMemHandle(&b.buf, (void *)base));
    TH_HIP(hipIpcGetMemHandle(&b.flags, comm->flags_local));
    b.buf_offset = (uint64_t)((char *)d_buf - (char *)base);
    b.n = n;
    b.pid = (int32_t)getpid();
    b.device = comm->device;
    b.rank = comm->rank;
    memcpy(b.bus_id, comm->bus_id, sizeof(b.bus_id));
    comm->reg_buf = d_buf;
    comm->reg_n = n;
    memcpy(out_blob, &b, sizeof(b));
    return 0;
}

int th_comm_p2p_connect(th_comm *comm, const uint8_t *blobs) {
    TH_REQUIRE(comm && comm->p2p && blobs && comm->reg_buf, "th_comm_p2p_connect: export this rank's buffer first");
    TH_REQUIRE(!comm->loopback, "th_comm_p2p_connect: a loopback communicator is connected from its start");
    TH_HIP(hipSetDevice(comm->device));
    comm->sharing = 0;
    for (int r = 0; r < comm->n_ranks; ++r) {
        P2PBlob b;
        memcpy(&b, blobs + (size_t)r * TH_P2P_BLOB_BYTES, sizeof(b));
        if (memcmp(b.bus_id, comm->bus_id, sizeof(b.bus_id)) == 0) ++comm->sharing;
        TH_REQUIRE(b.rank == r && b.n == comm->reg_n, "th_comm_p2p_connect: blob %d is from rank %d with %llu floats (expected rank %d, %zu floats)", r,
                   b.rank, (unsigned long long)b.n, r, comm->reg_n);
        if (r == comm->rank) {
            comm->dev.buf[r] = comm->reg_buf;
            comm->dev.flags[r] = comm->flags_local;
            continue;
        }
        if (b.device != comm->device) {
            int can = 0;
            TH_HIP(hipDeviceCanAccessPeer(&can, comm->device, b.device));
            TH_REQUIRE(can, "th_comm_p2p_connect: device %d cannot access device %d", comm->device, b.device);
            hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) TH_HIP(e);
            (void)hipGetLastError();
        }
        TH_HIP(hipIpcOpenMemHandle(&comm->peer_base[r], b.buf, hipIpcMemLazyEnablePeerAccess));
        TH_HIP(hipIpcOpenMemHandle(&comm->peer_flag_base[r], b.flags, hipIpcMemLazyEnablePeerAccess));
        comm->dev.buf[r] = (const float *)((char *)comm->peer_base[r] + b.buf_offset);
        comm->dev.flags[r] = (uint32_t *)comm->peer_flag_base[r];
    }
    comm->dev.state = comm->state;
    comm->dev.err_host = comm->err_host;
    comm->dev.spin_ticks = comm->timeout_ms * 100000L;   // 100 MHz wall clock
    comm->dev.n_ranks = comm->n_ranks;
    comm->dev.rank = comm->rank;
    // the in-launch exchange: this rank's source block in every peer's region, its own region to receive in
    th::DpDev &d = comm->dp;
    for (int r = 0; r < comm->n_ranks; ++r) {
        char *base = (char *)comm->dev.flags[r];
        d.peer_data[r] = (float *)(base + th::DP_DATA_OFFSET);
    }
    d.recv_data = (float *)((char *)comm->flags_local + th::DP_DATA_OFFSET);
    // four ranks and more: a slice is reduced by ONE rank (two hops, 2 / W of the bytes per link); below, by every rank (one hop).
    // TAPER_DP_TWO_SHOT = 0 | 1 forces the form (every rank must agree: it is part of the protocol)
    {
        const char *e = getenv("TAPER_DP_TWO_SHOT");
        d.two_shot = e ? (atoi(e) != 0 && comm->n_ranks >= 2 ? 1 : 0) : (comm->n_ranks >= 4 ? 1 : 0);
    }
    d.state = comm->state;
    d.err_host = comm->err_host;
    d.spin_ticks = comm->dev.spin_ticks;
    d.n_ranks = comm->n_ranks;
    d.rank = comm->rank;
    d.scale = 1.0f / (float)comm->n_ranks;
    if (comm->sharing < 1) comm->sharing = 1;
    comm->connected = true;
    return 0;
}

int th_comm_init_loopback(th_ctx *ctx, th_comm **out) {
    TH_REQUIRE(ctx && out, "th_comm_init_loopback: null argument");
    th_comm *c = nullptr;
    if (int rc = th_comm_init_p2p(ctx, 2, 0, &c)) return rc;
    c->loopback = true;
    th::DpDev &d = c->dp;
    char *base = (char *)c->flags_local;
    for (int r = 0; r < 2; ++r) {   // "rank 1" is this rank again: what it would push arrives in source block 1 of the local region
        d.peer_data[r] = (float *)(base + th::DP_DATA_OFFSET) + (size_t)r * th::DP_SRC_STRIDE;   // (+ r blocks: this rank is 0, its push to "rank 1" lands in block 1)
    }
    d.recv_data = (float *)(base + th::DP_DATA_OFFSET);
    d.two_shot = 0;                          // (the owner of a slice would be a rank that does not exist: the one-hop form only)
    d.state = c->state;
    d.err_host = c->err_host;
    d.spin_ticks = c->timeout_ms * 100000L;
    d.n_ranks = 2;
    d.rank = 0;
    d.scale = 0.5f;
    c->sharing = 1;
    c->connected = true;
    *out = c;
    return 0;
}

int th_comm_is_loopback(const th_comm *comm) { return comm && comm->loopback ? 1 : 0; }

int th_comm_exchange_form(const th_comm *comm, int *out_form) {
    TH_REQUIRE(comm && out_form, "th_comm_exchange_form: null argument");
    *out_form = !(comm->p2p && comm->connected) ? 0 : (comm->dp.two_shot ? 2 : 1);
    return 0;
}

int th_comm_sharing(const th_comm *comm, int *out_ranks_on_this_device) {
    TH_REQUIRE(comm && out_ranks_on_this_device, "th_comm_sharing: null argument");
    *out_ranks_on_this_device = comm->sharing;
    return 0;
}

int th_comm_is_p2p(const th_comm *comm) { return comm && comm->p2p ? 1 : 0; }

int th_comm_count(const th_comm *comm, int *out_ranks) {
    TH_REQUIRE(comm && out_ranks, "th_comm_count: null argument");
    *out_ranks = comm->n_ranks;
    if (!comm->p2p && comm->comm) TH_NCCL(ncclCommCount(comm->comm, out_ranks));   // what RCCL itself says the communicator spans
    return 0;
}

int th_comm_error(th_comm *comm, th_ctx *ctx, int *out_error) {
    TH_REQUIRE(comm && ctx && out_error, "th_comm_error: null argument");
    *out_error = 0;
    if (!comm->p2p) return 0;
    uint32_t st[3] = {0, 0, 0};
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpy(st, comm->state, sizeof(st), hipMemcpyDeviceToHost));
    *out_error = (int)st[2];
    return 0;
}

// post-mortem of the in-launch exchange's receive region (TAPER_DP_POSTMORTEM=1, Trainer::check_comm): per parity and source block, the
// slots whose first word is not the empty mark
int th_comm_debug_dump(th_comm *comm, th_ctx *ctx) {
    TH_REQUIRE(comm && ctx && comm->p2p, "th_comm_debug_dump: needs a peer-to-peer communicator");
    TH_HIP(hipStreamSynchronize(ctx->stream));
    uint32_t st[16];
    TH_HIP(hipMemcpy(st, comm->state, sizeof(st), hipMemcpyDeviceToHost));
    fprintf(stderr, "[dp post-mortem rank %d] state:", comm->rank);
    for (int i = 0; i < 16; ++i) fprintf(stderr, " %u", st[i]);
    fprintf(stderr, "\n");
    std::vector<uint64_t> w(th::DP_SRC_STRIDE / 2);
    const char *base = (const char *)comm->flags_local + th::DP_DATA_OFFSET;
    for (int par = 0; par < 2; ++par)
        for (int s = 0; s <= th::DP_MAX_RANKS; ++s) {
            TH_HIP(hipMemcpy(w.data(), base + ((size_t)par * th::DP_PARITY_STRIDE + (size_t)s * th::DP_SRC_STRIDE) * sizeof(float),
                             th::DP_SRC_STRIDE * sizeof(float), hipMemcpyDeviceToHost));
            std::string line;
            int n_set = 0;
            for (int slot = 0; slot < th::DP_MAX_SLOTS; ++slot) {
                int cnt = 0;
                for (int i = 0; i < th::DP_SLOT_FLOATS / 2; ++i) cnt += w[(size_t)slot * (th::DP_SLOT_FLOATS / 2) + i] != ~0ull;
                if (cnt) {
                    ++n_set;
                    if (line.size() < 400) line += " " + std::to_string(slot) + ":" + std::to_string(cnt);
                }
            }
            if (n_set) fprintf(stderr, "[dp post-mortem rank %d] parity %d block %d: %d slots hold words (slot:count)%s\n", comm->rank, par, s, n_set, line.c_str());
        }
    return 0;
}

int th_comm_timeout_detail(th_comm *comm, th_ctx *ctx, int out4[4]) {
    TH_REQUIRE(comm && ctx && out4 && comm->p2p, "th_comm_timeout_detail: needs a peer-to-peer communicator");
    uint32_t st[16];
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpy(st, comm->state, sizeof(st), hipMemcpyDeviceToHost));
    out4[0] = (int)st[9] - 1;
    out4[1] = (int)st[10];
    out4[2] = (int)st[11];
    out4[3] = (int)st[th::DP_ST_STEP];
    return 0;
}

int th_comm_error_peek(const th_comm *comm, int *out_error) {
    TH_REQUIRE(comm && out_error, "th_comm_error_peek: null argument");
    *out_error = (comm->p2p && comm->err_host) ? (int)__atomic_load_n(comm->err_host, __ATOMIC_RELAXED) : 0;
    return 0;
}

int th_comm_error_word(const th_comm *comm, const uint32_t **d_out) {
    TH_REQUIRE(comm && d_out, "th_comm_error_word: null argument");
    *d_out = comm->p2p ? comm->state + 2 : nullptr;
    return 0;
}

int th_comm_step_word(const th_comm *comm, uint32_t **d_out) {
    TH_REQUIRE(comm && d_out, "th_comm_step_word: null argument");
    *d_out = comm->p2p ? comm->state + th::DP_ST_STEP : nullptr;
    return 0;
}

int th_comm_set_timeout_ms(th_comm *comm, int64_t ms) {
    TH_REQUIRE(comm && comm->p2p && ms > 0, "th_comm_set_timeout_ms: needs a peer-to-peer communicator and a positive bound");
    comm->timeout_ms = (long)ms;
    comm->dev.spin_ticks = comm->timeout_ms * 100000L;   // launches enqueued (or captured) from here on
    comm->dp.spin_ticks = comm->dev.spin_ticks;
    return 0;
}

int th_allreduce_adam(th_comm *comm, th_ctx *ctx, const float *d_grads, size_t n, float scale, float *d_params, float *d_m, float *d_v,
                      const int64_t *d_offsets, const int32_t *d_has_grad, int n_tensors, int32_t *d_t, const float *d_lr, float beta1,
                      float beta2, float eps, float weight_decay, int pre_ticked) {
    TH_REQUIRE(comm && ctx && d_grads && d_params && d_m && d_v && d_offsets && d_has_grad && d_t && d_lr && n_tensors > 0,
               "th_allreduce_adam: null argument");
    TH_REQUIRE(comm->p2p && comm->connected && !comm->loopback, "th_allreduce_adam: needs a connected peer-to-peer communicator (th_comm_init_p2p / _export / _connect)");
    TH_REQUIRE(d_grads == comm->reg_buf && n == comm->reg_n, "th_allreduce_adam: not the buffer this communicator exported");
    TH_REQUIRE(((uintptr_t)d_params & 15) == 0 && ((uintptr_t)d_m & 15) == 0 && ((uintptr_t)d_v & 15) == 0, "th_allreduce_adam: arenas must be 16-byte aligned");
#define TH_P2P_ADAM(NR_)                                                                                                                       \
    hipLaunchKernelGGL(th::p2p_allreduce_adam_kernel<NR_>, dim3(th::p2p_grid(n)), dim3(256), 0, ctx->stream, comm->dev, (long)n, scale, d_params, \
                       d_m, d_v, d_offsets, d_has_grad, n_tensors, d_t, d_lr, beta1, beta2, eps, weight_decay, pre_ticked)
    // the instance with the smallest compiled-in slot count >= n_ranks (p2p_sum_quad)
    if (comm->n_ranks <= 1) TH_P2P_ADAM(1);
    else if (comm->n_ranks == 2) TH_P2P_ADAM(2);
    else if (comm->n_ranks <= 4) TH_P2P_ADAM(4);
    else TH_P2P_ADAM(8);
#undef TH_P2P_ADAM
    TH_LAUNCH_CHECK();
    ++comm->launches_fused;
    return 0;
}

int th_comm_stats(const th_comm *comm, int64_t out2[2]) {
    TH_REQUIRE(comm && out2, "th_comm_stats: null argument");
    out2[0] = comm->launches_inplace;
    out2[1] = comm->launches_fused;
    return 0;
}

int th_comm_stats_inkernel(const th_comm *comm, int64_t *out_launches) {
    TH_REQUIRE(comm && out_launches, "th_comm_stats_inkernel: null argument");
    *out_launches = comm->launches_inkernel;
    return 0;
}

int th_comm_exchange_selftest(th_comm *comm, th_ctx *ctx, int slots, int rounds, int *out_bad) {
    TH_REQUIRE(comm && ctx && out_bad, "th_comm_exchange_selftest: null argument");
    TH_REQUIRE(comm->p2p && comm->connected, "th_comm_exchange_selftest: needs a connected peer-to-peer communicator");
    TH_REQUIRE(slots >= 1 && slots <= th::DP_MAX_SLOTS && rounds >= 1, "th_comm_exchange_selftest: 1 .. %d slots", th::DP_MAX_SLOTS);
    // every rank's workgroups must be able to be resident together when ranks share a device: a waiting workgroup holds its slot
    TH_REQUIRE(comm->sharing <= 1 || (long)comm->sharing * slots <= 2L * th::kNumCU,
               "th_comm_exchange_selftest: %d ranks on one device x %d slots do not fit the device together", comm->sharing, slots);
    uint32_t *bad = nullptr;
    TH_HIP(hipMalloc((void **)&bad, sizeof(uint32_t)));
    TH_HIP(hipMemsetAsync(bad, 0, sizeof(uint32_t), ctx->stream));
    for (int k = 0; k < rounds; ++k) {
        hipLaunchKernelGGL(th::dp_bump_kernel, dim3(1), dim3(1), 0, ctx->stream, comm->state);
#define TH_DP_SELFTEST(NR_) hipLaunchKernelGGL(th::dp_selftest_kernel<NR_>, dim3(slots), dim3(256), 0, ctx->stream, comm->dp, k, comm->loopback ? 1 : 0, bad)
        if (comm->n_ranks <= 2) TH_DP_SELFTEST(2);
        else if (comm->n_ranks <= 4) TH_DP_SELFTEST(4);
        else TH_DP_SELFTEST(8);
#undef TH_DP_SELFTEST
    }
    uint32_t h = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&h, bad, sizeof(h), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(bad);
    TH_HIP(e);
    *out_bad = (int)h;
    return 0;
}

int th_allreduce_sum_scale(th_comm *comm, th_ctx *ctx, float *d_buf, size_t n, float scale) {
    TH_REQUIRE(comm && ctx && (n == 0 || d_buf), "th_allreduce_sum_scale: null argument");
    if (n == 0) return 0;
    if (comm->p2p) {   // one-shot over xGMI: every rank reads its peers' buffers directly and adds them in rank order
        TH_REQUIRE(comm->connected && !comm->loopback, "th_allreduce_sum_scale: peer-to-peer communicator is not connected");
        TH_REQUIRE(d_buf == comm->reg_buf && n == comm->reg_n, "th_allreduce_sum_scale: not the buffer this communicator exported");
        int g = (int)((n / 4 + 255) / 256);             // one float4 per thread while that fits, up to P2P_QUADS beyond
        if (g > P2P_MAX_BLOCKS) g = P2P_MAX_BLOCKS;     // (every workgroup stays resident until all peers are done reading)
#define TH_P2P_INPLACE(NR_) hipLaunchKernelGGL(th::p2p_allreduce_kernel<NR_>, dim3(g < 1 ? 1 : g), dim3(256), 0, ctx->stream, comm->dev, d_buf, (long)n, scale)
        if (comm->n_ranks <= 1) TH_P2P_INPLACE(1);
        else if (comm->n_ranks == 2) TH_P2P_INPLACE(2);
        else if (comm->n_ranks <= 4) TH_P2P_INPLACE(4);
        else TH_P2P_INPLACE(8);
#undef TH_P2P_INPLACE
        TH_LAUNCH_CHECK();
        ++comm->launches_inplace;
        return 0;
    }
    // the data-parallel mean (scale == 1/n_ranks) is RCCL's own ncclAvg: no scale launch behind the collective
    const float mean = 1.0f / (float)comm->n_ranks;
    if (scale == mean) {
        TH_NCCL(ncclAllReduce(d_buf, d_buf, n, ncclFloat, ncclAvg, comm->comm, ctx->stream));
        return 0;
    }
    TH_NCCL(ncclAllReduce(d_buf, d_buf, n, ncclFloat, ncclSum, comm->comm, ctx->stream));
    return th::scale_inplace(ctx, d_buf, n, scale);
}

}  // extern "C"

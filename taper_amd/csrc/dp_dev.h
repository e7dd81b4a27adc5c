// dp_dev.h -- the data-parallel gradient exchange INSIDE the launch that produces the gradients (new: the reference is one process,
// SURVEY.md 8e; what the exchange feeds is optim.rs:83-113).  Device side only; comm.hip owns the memory and the bootstrap.
//
// A workgroup that has just finished a slice of a gradient (mlp_tail.hip: a 16 x 32 block of dW1, or a head tile) calls dp_exchange():
//   1. PUSH   every participating thread stores its NV values as 8-byte words into its place in every peer's receive region
//             (system-coherent stores: over xGMI for a peer on another device) -- and waits for nothing;
//   2. POLL   the same thread watches ITS OWN words of every peer's block in the LOCAL receive region until none of them is the sentinel any
//             more (all ones: a NaN no arithmetic produces; an 8-byte store lands whole) -- bounded by the wall clock;
//   3. SUM    it adds the W values IN RANK ORDER, its own at its rank's place: the same bits on every rank; * 1/W;
//   4. RESET  it puts the sentinel back into the words it has just read (they are written again two steps later at the earliest, see below).
// The caller then applies Adam from registers.  ONE dependent trip through memory per exchange -- the value IS the flag (no acknowledgement to
// wait for before a flag may follow, no flag, no second trip to fetch what the flag announced: that form measured + 5.0 us on a 14.5 us
// step with both "ranks" on one device, this one + 1.6) -- no extra bytes on a link, no launch boundary, no arena-wide hand-shake: a
// slice is exchanged while other workgroups still compute theirs, and with one rank the function is empty.
//
// Reuse of the receive region needs no "done reading" hand-off: it is double-buffered on the step's parity.  A rank writes parity p of
// step n + 2 only after it finished step n + 1, which took every peer's step-(n + 1) values for every slot, which a peer pushes from a launch
// that is behind its own step-n launch in its stream -- the peer has read (and reset) all of step n by then.
// The step number is state[6], advanced by the launch IN FRONT of the gradient launch (the step's first launch ticks Adam's counter and this
// one with it: th_ctx_set_update_guard; the self-test has a one-thread launch for it) -- every workgroup reads the same value through the
// scalar cache at no cost, nobody counts arrivals.  Nothing else is ever reset: a captured graph replays the launches as they are.
#pragma once
#include "common.h"

namespace th {

constexpr int DP_MAX_RANKS = 8;
constexpr int DP_MAX_SLOTS = 512;        // slices per launch
constexpr int DP_SLOT_FLOATS = 1024;     // 256 threads x <= 4 values
constexpr int DP_ST_ERROR = 2, DP_ST_DEAD = 5, DP_ST_STEP = 6, DP_ST_ABORTS = 8;   // words of th_comm::state
// region layout (one fine-grained allocation per rank, mapped by every peer):
//   [0, 1 KB)            the flag block of the three-launch form (comm.hip)
//   DP_DATA_OFFSET       float  data[parity][src rank][slot][DP_SLOT_FLOATS], all ones (DP_EMPTY) wherever nothing has arrived
constexpr size_t DP_DATA_OFFSET = 4096;
constexpr uint64_t DP_EMPTY = ~0ull;
constexpr size_t DP_SRC_STRIDE = (size_t)DP_MAX_SLOTS * DP_SLOT_FLOATS;            // floats between two source ranks
constexpr int DP_RESULT_BLOCK = DP_MAX_RANKS;                                       // two-shot form: the block an owner's means arrive in
constexpr size_t DP_PARITY_STRIDE = (size_t)(DP_MAX_RANKS + 1) * DP_SRC_STRIDE;    // floats between the two parities
constexpr size_t DP_REGION_BYTES = DP_DATA_OFFSET + 2 * DP_PARITY_STRIDE * sizeof(float);

struct DpDev {                               // kernel argument (n_ranks == 0: no exchange)
    float *peer_data[DP_MAX_RANKS];          // [r]: rank r's region data as mapped here (this rank's slices go to its source block `rank` in it)
    float *recv_data;                        // the local region's data
    int two_shot;                            // 0: every rank reduces every slice (one hop); 1: slice s is reduced by rank s % n_ranks (two hops, 1 / n_ranks of the bytes per link)
    uint32_t *state;                         // th_comm::state (local)
    uint32_t *err_host;                      // the error word's host-visible copy
    long spin_ticks;                         // bound of a poll, 100 MHz wall-clock ticks
    int n_ranks, rank;
    float scale;                             // 1 / n_ranks
};

__device__ __forceinline__ uint32_t dp_load_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// what a workgroup needs to know about the step, requested at kernel entry through the scalar unit (both words were written by EARLIER
// launches) and first looked at when its slice is ready: no round trip in front of the kernel's own loads
struct DpTicket {
    uint32_t step, dead;
};
__device__ __forceinline__ DpTicket dp_begin(const DpDev &c) {
    const __attribute__((address_space(4))) uint32_t *st = reinterpret_cast<const __attribute__((address_space(4))) uint32_t *>(reinterpret_cast<uintptr_t>(c.state));
    return DpTicket{st[DP_ST_STEP], st[DP_ST_DEAD]};
}
// one thread, in a launch in front of the exchanging one
__device__ __forceinline__ void dp_advance_step(uint32_t *step_word) { step_word[0] += 1u; }

// (what the first workgroup to give up was waiting for: state[9] = slot + 1, [10] = step, [11] = bit s set: rank s's words had not arrived -- th_comm_timeout_detail)
__device__ __forceinline__ void dp_note(const DpDev &c, int slot, uint32_t step, uint32_t missing) {
    if (__hip_atomic_exchange(&c.state[9], (uint32_t)slot + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        c.state[10] = step;
        c.state[11] = missing;
    }
}
__device__ __forceinline__ void dp_raise(const DpDev &c) {
    __hip_atomic_store(&c.state[DP_ST_ERROR], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&c.state[DP_ST_DEAD], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(c.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_fetch_add(&c.state[DP_ST_ABORTS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 8-byte system-coherent stores / loads as relaxed system-scope atomics: the compiler emits global_store / global_load_dwordx2 ... sc0 sc1
// (written through to / served by memory, never a cache of this device) AND knows about them -- its own wait counts and data-register
// hazards cover them (tools/asm_hazard_audit.py).  An aligned 8-byte store is single-copy atomic: a reader sees all of it or none.
__device__ __forceinline__ void dp_st64(float *p, uint64_t q) {
    __hip_atomic_store(reinterpret_cast<uint64_t *>(p), q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint64_t dp_ld64(const float *p) {
    return __hip_atomic_load(reinterpret_cast<const uint64_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint64_t dp_pack(float a, float b) {
    const uint64_t q = (uint64_t)__float_as_uint(a) | ((uint64_t)__float_as_uint(b) << 32);
    return q == DP_EMPTY ? q ^ 1ull : q;     // (two all-ones NaNs: still two NaNs, no longer the sentinel)
}

// All threads of the workgroup call this (it holds a barrier); `tid` in [0, 256) for a thread that owns NV values of the slice, -1 otherwise.
// NV = 2 or 4 (8-byte words).  NR = 2 / 4 / 8 >= n_ranks, compiled in (the sum of step 3 is a fixed, unrolled chain).
// Returns false when the communicator was dead at entry or a peer's values did not arrive within the bound (v is then unchanged and the
// error word is up).
template <int NR, int NV>
__device__ __forceinline__ bool dp_exchange(const DpDev &c, const DpTicket &tk, int slot, int tid, float (&v)[NV]) {
    if (tk.dead != 0u) return false;      // (uniform over the launch: the word as the previous launch left it)
    const uint32_t step = tk.step;
    static_assert(NV == 2 || NV == 4, "dp_exchange: 2 or 4 values per thread");
    constexpr int NQ = NV / 2;
    const int W = c.n_ranks, me = c.rank;
    const size_t par = (size_t)(step & 1u) * DP_PARITY_STRIDE;
    const size_t off = (size_t)slot * DP_SLOT_FLOATS + (size_t)(tid < 0 ? 0 : tid) * NV;
    bool ok = true;
    uint64_t q[NR][NQ];
    if (tid >= 0) {
        // 1. push
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            const uint64_t mine = dp_pack(v[2 * h], v[2 * h + 1]);
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (r < W && r != me) dp_st64(c.peer_data[r] + (size_t)me * DP_SRC_STRIDE + par + off + 2 * h, mine);
        }
        // 2. poll: every peer's words for this thread, all requests of a round in flight together
        float *mine_in = c.recv_data + par + off;
        const long t0 = wall_clock64();
        while (true) {
            bool all = true;
#pragma unroll
            for (int s = 0; s < NR; ++s)
#pragma unroll
                for (int h = 0; h < NQ; ++h) {
                    const bool peer = s < W && s != me;
                    q[s][h] = peer ? dp_ld64(mine_in + (size_t)s * DP_SRC_STRIDE + 2 * h) : 0ull;
                    all = all && q[s][h] != DP_EMPTY;
                }
            if (all) break;
            if (wall_clock64() - t0 > c.spin_ticks) {
                uint32_t missing = 0u;
#pragma unroll
                for (int s = 0; s < NR; ++s)
                    if (s < W && s != me && q[s][0] == DP_EMPTY) missing |= 1u << s;
                dp_note(c, slot, step, missing);
                dp_raise(c);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        // 4. reset what was read (only once everything has arrived: a word that is still empty stays empty)
        if (ok) {
#pragma unroll
            for (int s = 0; s < NR; ++s)
                if (s < W && s != me) {
#pragma unroll
                    for (int h = 0; h < NQ; ++h) dp_st64(mine_in + (size_t)s * DP_SRC_STRIDE + 2 * h, DP_EMPTY);
                }
        }
    }
    ok = __syncthreads_and(ok);      // one verdict per workgroup: the slice is applied by all of its threads or by none
    if (!ok || tid < 0) return ok;
    // 3. sum in rank order
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < NR; ++s) {
            const float x = s == me ? v[j] : __uint_as_float((uint32_t)(q[s][j / 2] >> (32 * (j & 1))));
            acc = s == 0 ? x : (s < W ? acc + x : acc);
        }
        v[j] = acc * c.scale;
    }
    return true;
}

// The TWO-SHOT form of the same exchange, for rings of four ranks and more: slice `slot` is reduced by ONE rank, its owner slot % W.  A
// non-owner pushes its values to the owner only and waits for the mean in the result block of its own region; the owner polls the W - 1
// contributions, adds in rank order (its own at its rank's place: the bits every rank would have formed), scales, and pushes the mean to
// every peer.  Two dependent hops instead of one, but a link carries 2 / W of a rank's gradient instead of all of it: with the MLP's 407 KB
// and eight ranks 102 KB per link and direction instead of 407 -- ~1.4 us of a 76.8 GB/s xGMI direction instead of ~5.3, which is what
// decides the step there.  Same words, same empty mark, same parity halves; every rank applies Adam itself to the mean it
// received: p / m / v stay replicated, bit-identical.
template <int NR, int NV>
__device__ __forceinline__ bool dp_exchange_two_shot(const DpDev &c, const DpTicket &tk, int slot, int tid, float (&v)[NV]) {
    static_assert(NV == 2 || NV == 4, "dp_exchange: 2 or 4 values per thread");
    constexpr int NQ = NV / 2;
    if (tk.dead != 0u) return false;
    const int W = c.n_ranks, me = c.rank, owner = slot % W;
    const size_t par = (size_t)(tk.step & 1u) * DP_PARITY_STRIDE;
    const size_t off = (size_t)slot * DP_SLOT_FLOATS + (size_t)(tid < 0 ? 0 : tid) * NV;
    bool ok = true;
    if (tid >= 0) {
        const long t0 = wall_clock64();
        if (me == owner) {                      // (workgroup-uniform)
            uint64_t q[NR][NQ];
            float *in = c.recv_data + par + off;
            while (true) {
                bool all = true;
#pragma unroll
                for (int s = 0; s < NR; ++s)
#pragma unroll
                    for (int h = 0; h < NQ; ++h) {
                        q[s][h] = (s < W && s != me) ? dp_ld64(in + (size_t)s * DP_SRC_STRIDE + 2 * h) : 0ull;
                        all = all && q[s][h] != DP_EMPTY;
                    }
                if (all) break;
                if (wall_clock64() - t0 > c.spin_ticks) {
                    uint32_t missing = 0u;
#pragma unroll
                    for (int s = 0; s < NR; ++s)
                        if (s < W && s != me && q[s][0] == DP_EMPTY) missing |= 1u << s;
                    dp_note(c, slot, tk.step, missing);
                    dp_raise(c);
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (ok) {
#pragma unroll
                for (int s = 0; s < NR; ++s)
                    if (s < W && s != me) {
#pragma unroll
                        for (int h = 0; h < NQ; ++h) dp_st64(in + (size_t)s * DP_SRC_STRIDE + 2 * h, DP_EMPTY);
                    }
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    float acc = 0.f;
#pragma unroll
                    for (int s = 0; s < NR; ++s) {
                        const float x = s == me ? v[j] : __uint_as_float((uint32_t)(q[s][j / 2] >> (32 * (j & 1))));
                        acc = s == 0 ? x : (s < W ? acc + x : acc);
                    }
                    v[j] = acc * c.scale;
                }
#pragma unroll
                for (int h = 0; h < NQ; ++h) {
                    const uint64_t mean = dp_pack(v[2 * h], v[2 * h + 1]);
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        if (r < W && r != me) dp_st64(c.peer_data[r] + (size_t)DP_RESULT_BLOCK * DP_SRC_STRIDE + par + off + 2 * h, mean);
                }
            }
        } else {
#pragma unroll
            for (int h = 0; h < NQ; ++h)
                dp_st64(c.peer_data[owner] + (size_t)me * DP_SRC_STRIDE + par + off + 2 * h, dp_pack(v[2 * h], v[2 * h + 1]));
            float *res = c.recv_data + (size_t)DP_RESULT_BLOCK * DP_SRC_STRIDE + par + off;
            uint64_t m[NQ];
            while (true) {
                bool all = true;
#pragma unroll
                for (int h = 0; h < NQ; ++h) {
                    m[h] = dp_ld64(res + 2 * h);
                    all = all && m[h] != DP_EMPTY;
                }
                if (all) break;
                if (wall_clock64() - t0 > c.spin_ticks) {
                    dp_note(c, slot, tk.step, 1u << owner);
                    dp_raise(c);
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (ok) {
#pragma unroll
                for (int h = 0; h < NQ; ++h) {
                    dp_st64(res + 2 * h, DP_EMPTY);
                    v[2 * h] = __uint_as_float((uint32_t)m[h]);        // (the owner's bits: dp_pack only ever changes a pair of all-ones NaNs)
                    v[2 * h + 1] = __uint_as_float((uint32_t)(m[h] >> 32));
                }
            }
        }
    }
    return __syncthreads_and(ok);
}

// what the kernels call: the communicator's form
template <int NR, int NV>
__device__ __forceinline__ bool dp_reduce(const DpDev &c, const DpTicket &tk, int slot, int tid, float (&v)[NV]) {
    return c.two_shot ? dp_exchange_two_shot<NR, NV>(c, tk, slot, tid, v) : dp_exchange<NR, NV>(c, tk, slot, tid, v);
}

// host side (comm.hip): the communicator's exchange descriptor (nullptr: not a connected peer-to-peer communicator), how many ranks run on
// this rank's device, and the launch counter the tests read
const DpDev *comm_dp_dev(const th_comm *c);
int comm_dp_sharing(const th_comm *c);
bool comm_dp_shared_fits(const th_comm *c, int grid, int per_cu);     // ranks sharing a device: may `grid` waiting workgroups per rank (per_cu to a CU) be launched? (comm.hip)
void comm_dp_count_launch(th_comm *c);

}  // namespace th

// common.h -- shared internals of libtaper_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <array>
#include <atomic>
#include <map>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/taper_hip.h"
#include "../../include/taper_hip_debug.h"   // test hooks: declared apart from the boundary, defined in the same library

namespace th {

void set_error(const char *fmt, ...);

#define TH_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            th::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

#define TH_REQUIRE(cond, ...)              \
    do {                                   \
        if (!(cond)) {                     \
            th::set_error(__VA_ARGS__);    \
            return 2;                      \
        }                                  \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a (function, DEVICE) pair: set once per pair and size -- the statics are per expansion
// site, i.e. per kernel instance -- so that a process with contexts on several devices can launch > 64 KB of LDS on each of them.  Atomic
// (two host threads may launch the same instance), and a site whose size GROWS sets the attribute again (the largest size seen is kept).
#define TH_SET_MAX_LDS(ctx_, fn_, bytes_)                                                                                \
    do {                                                                                                                 \
        static std::atomic<unsigned long long> th_lds_done_{0ull};                                                       \
        static std::atomic<int> th_lds_max_{0};                                                                          \
        const unsigned long long th_lds_bit_ = 1ull << ((ctx_)->device & 63);                                            \
        const int th_lds_want_ = (int)(bytes_);                                                                          \
        if (!(th_lds_done_.load(std::memory_order_relaxed) & th_lds_bit_) || th_lds_want_ > th_lds_max_.load(std::memory_order_relaxed)) { \
            TH_HIP(hipFuncSetAttribute((const void *)(fn_), hipFuncAttributeMaxDynamicSharedMemorySize, th_lds_want_));  \
            int th_lds_prev_ = th_lds_max_.load(std::memory_order_relaxed);                                              \
            if (th_lds_want_ > th_lds_prev_) {                                                                           \
                th_lds_max_.store(th_lds_want_, std::memory_order_relaxed);                                              \
                th_lds_done_.store(th_lds_bit_, std::memory_order_relaxed);   /* a larger size: the other devices set it again */ \
            } else {                                                                                                     \
                th_lds_done_.fetch_or(th_lds_bit_, std::memory_order_relaxed);                                           \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)

// launch check: kernel launches are asynchronous; this catches bad configs.
#define TH_LAUNCH_CHECK() TH_HIP(hipGetLastError())

constexpr int kWave = 64;       // gfx950 wavefront
constexpr int kNumCU = 256;     // MI355X
constexpr int kNumXCD = 8;

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Workgroup barrier for LDS hand-offs only: global loads and stores in flight stay in flight across it -- __syncthreads() waits for
// them (its fence covers every address space), which exposes a global round trip at every barrier of a multi-stage kernel.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- device-raised errors (the reference panics inside an op -- e.g. loss.rs:161 "Target class {} out of bounds for {}" -- where a
// kernel can only leave a note).  One 64-byte word block per device in host-visible, device-mapped memory; every translation unit that
// raises holds its own copy of the block's address (g_err_word; no relocatable device code in this library) and registers a binder
// that th_ctx_create runs once per device.  A raise is a few stores on a path no valid input takes; th_ctx_sync / th_memcpy_d2h look at
// the block after their wait, turn a non-zero code into a non-zero return + th_last_error() and clear it (runtime.hip).
enum : int32_t { TH_DEVERR_NONE = 0, TH_DEVERR_TARGET_OOB = 1 };
static __device__ __attribute__((unused)) int32_t *g_err_word;
__device__ __forceinline__ void raise_target_oob(long cls, int classes) {   // loss.rs:160-161
    int32_t *w = g_err_word;
    if (!w) return;
    w[1] = (int32_t)(cls > 0x7fffffffL ? 0x7fffffffL : cls);
    w[2] = classes;
    __threadfence_system();
    w[0] = TH_DEVERR_TARGET_OOB;
}
typedef int (*err_bind_fn)(int32_t *);
struct ErrBindReg { explicit ErrBindReg(err_bind_fn f); };
#define TH_USES_DEVICE_ERRORS()                                                                                        \
    namespace {                                                                                                        \
    int th_err_bind_here(int32_t *p) { return hipMemcpyToSymbol(HIP_SYMBOL(th::g_err_word), &p, sizeof(p)) == hipSuccess ? 0 : 1; } \
    th::ErrBindReg th_err_reg_here(th_err_bind_here);                                                                  \
    }

// Memory-bound grid: cap at 8 blocks/CU and grid-stride the rest.
static inline int ew_grid(size_t n_items, int block) {
    long g = (long)((n_items + block - 1) / block);
    long cap = (long)kNumCU * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace th

struct th_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // pooled allocator state
    std::multimap<size_t, void *> free_blocks;        // size -> block
    std::unordered_map<void *, size_t> block_size;    // every block we own
    std::unordered_set<void *> graph_owned;           // blocks pinned by a live graph
    std::unordered_set<void *> graph_owned_freed;     // ...that the host has already th_free'd
    size_t bytes_reserved = 0, bytes_in_use = 0;
    // capture state
    bool capturing = false;
    std::vector<void *> capture_blocks;               // allocated during the current capture
    std::multimap<size_t, void *> capture_free;       // freed again during the capture
    // staging plans of the image-resident conv kernel, built on device once per geometry (conv_mfma.hip); plain hipMalloc, freed with the ctx
    std::map<std::array<int, 8>, void *> conv_plans;
    int32_t *err_word = nullptr;                      // this device's error block (host-visible; shared by the contexts of a device)
    int m2_max_ksplit = 8;                            // mlp2.hip: cap on the workgroups sharing a row block's k chunks (th_mlp2_set_max_ksplit; 1 = no split)
    const uint32_t *update_guard = nullptr;           // th_ctx_set_update_guard: deferred updates / counter ticks do nothing once this device word is non-zero
    uint32_t *update_step_word = nullptr;             // ... and a tick that happens advances this word too (the in-launch exchange's step number, dp_dev.h)
    unsigned *m2_arrive = nullptr;                    // mlp2.hip: arrival counters of k-split row blocks (zero between launches); plain hipMalloc, freed with the ctx
};

struct th_graph {
    th_ctx *ctx = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<void *> blocks;                       // memory the captured kernels reference
};

struct th_event {
    hipEvent_t ev = nullptr;
};

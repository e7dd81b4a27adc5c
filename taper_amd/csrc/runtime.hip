// runtime.hip -- context, stream-ordered pool allocator, copies, events and
// hipGraph capture for libtaper_hip.so.  Replaces the reference's L0 runtime
// (Vec<f32> on mimalloc, examples/train_mnist.rs:2-5) with device memory.
#include "common.h"

#include <cstring>

namespace th {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

// device-raised errors (common.h): the binders of the translation units that raise, and one host-visible block per device
static std::vector<err_bind_fn> &err_binders() {
    static std::vector<err_bind_fn> v;
    return v;
}
ErrBindReg::ErrBindReg(err_bind_fn f) { err_binders().push_back(f); }
static std::map<int, int32_t *> g_err_blocks;

static int err_block_for(int device, int32_t **out) {   // current device == device
    auto it = g_err_blocks.find(device);
    if (it == g_err_blocks.end()) {
        int32_t *w = nullptr;
        TH_HIP(hipHostMalloc((void **)&w, 64, hipHostMallocMapped | hipHostMallocCoherent));
        memset(w, 0, 64);
        for (err_bind_fn f : err_binders())
            if (f(w)) {
                set_error("th_ctx_create: binding the device error block failed");
                return 1;
            }
        it = g_err_blocks.emplace(device, w).first;
    }
    *out = it->second;
    return 0;
}

// after a wait: did a kernel of this device leave a note?  (the reference would have panicked inside the op)
static int err_block_check(th_ctx *ctx) {
    volatile int32_t *w = ctx->err_word;
    if (!w || w[0] == TH_DEVERR_NONE) return 0;
    const int32_t code = w[0], a = w[1], b = w[2];
    w[0] = TH_DEVERR_NONE;
    if (code == TH_DEVERR_TARGET_OOB) set_error("Target class %d out of bounds for %d", a, b);   // loss.rs:161
    else set_error("device error %d (%d, %d)", code, a, b);
    return 3;
}

// Size classes: 256 B granularity below 64 KiB, then 1/8-octave steps, so a
// training step's repeating allocation pattern hits the free lists exactly.
static size_t round_size(size_t bytes) {
    if (bytes < 256) return 256;
    if (bytes <= (64u << 10)) return (bytes + 255) & ~size_t(255);
    size_t p = 1;
    while (p < bytes) p <<= 1;
    size_t step = p >> 4;  // 1/8 of the lower octave
    return (bytes + step - 1) / step * step;
}

__global__ void fill_f32_kernel(float *__restrict__ p, float v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t n4 = n / 4;
    float4 v4 = make_float4(v, v, v, v);
    float4 *p4 = reinterpret_cast<float4 *>(p);
    for (size_t j = i; j < n4; j += stride) p4[j] = v4;
    for (size_t j = n4 * 4 + i; j < n; j += stride) p[j] = v;
}

}  // namespace th

using namespace th;

extern "C" {

const char *th_last_error(void) { return g_last_error.c_str(); }

int th_device_count(int *out) {
    TH_REQUIRE(out, "th_device_count: null out");
    TH_HIP(hipGetDeviceCount(out));
    return 0;
}

int th_ctx_create(int device_id, th_ctx **out) {
    TH_REQUIRE(out, "th_ctx_create: null out");
    int n = 0;
    TH_HIP(hipGetDeviceCount(&n));
    TH_REQUIRE(device_id >= 0 && device_id < n, "th_ctx_create: device %d not present (%d visible)", device_id, n);
    TH_HIP(hipSetDevice(device_id));
    th_ctx *c = new th_ctx();
    c->device = device_id;
    TH_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if (err_block_for(device_id, &c->err_word)) return 1;
    // arrival counters of th_mlp2_xent's k-split row blocks (2 KB, zero between launches): here, so that no launch has to allocate inside a capture
    TH_HIP(hipMalloc((void **)&c->m2_arrive, 512 * sizeof(unsigned)));
    TH_HIP(hipMemsetAsync(c->m2_arrive, 0, 512 * sizeof(unsigned), c->stream));   // on the context's own stream (non-blocking: the null stream does not order with it)
    TH_HIP(hipStreamSynchronize(c->stream));
    *out = c;
    return 0;
}

int th_ctx_destroy(th_ctx *ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->block_size) (void)hipFree(kv.first);
    for (auto &kv : ctx->conv_plans) (void)hipFree(kv.second);
    if (ctx->m2_arrive) (void)hipFree(ctx->m2_arrive);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

int th_ctx_set_update_guard(th_ctx *ctx, const uint32_t *d_skip_if_nonzero, uint32_t *d_step_word) {
    TH_REQUIRE(ctx, "th_ctx_set_update_guard: null ctx");
    ctx->update_guard = d_skip_if_nonzero;   // read when a launch is enqueued (or captured): launches already in a graph keep what they were given
    ctx->update_step_word = d_step_word;
    return 0;
}

int th_ctx_sync(th_ctx *ctx) {
    TH_REQUIRE(ctx, "th_ctx_sync: null ctx");
    TH_HIP(hipStreamSynchronize(ctx->stream));
    return err_block_check(ctx);
}

void *th_ctx_stream(th_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int th_ctx_device(th_ctx *ctx) { return ctx ? ctx->device : -1; }

int th_malloc(th_ctx *ctx, size_t bytes, void **d_out) {
    TH_REQUIRE(ctx && d_out, "th_malloc: null argument");
    size_t sz = round_size(bytes);
    void *p = nullptr;
    if (ctx->capturing) {
        auto it = ctx->capture_free.find(sz);
        if (it != ctx->capture_free.end()) {
            p = it->second;
            ctx->capture_free.erase(it);
        }
    }
    if (!p) {
        auto it = ctx->free_blocks.find(sz);
        if (it != ctx->free_blocks.end()) {
            p = it->second;
            ctx->free_blocks.erase(it);
        } else {
            TH_HIP(hipSetDevice(ctx->device));
            TH_HIP(hipMalloc(&p, sz));
            ctx->block_size[p] = sz;
            ctx->bytes_reserved += sz;
        }
        if (ctx->capturing) ctx->capture_blocks.push_back(p);
    }
    ctx->bytes_in_use += sz;
    *d_out = p;
    return 0;
}

int th_free(th_ctx *ctx, void *d_ptr) {
    if (!d_ptr) return 0;
    TH_REQUIRE(ctx, "th_free: null ctx");
    auto it = ctx->block_size.find(d_ptr);
    TH_REQUIRE(it != ctx->block_size.end(), "th_free: %p was not allocated by this ctx", d_ptr);
    size_t sz = it->second;
    ctx->bytes_in_use -= sz;
    if (ctx->graph_owned.count(d_ptr)) {  // stays pinned until th_graph_destroy
        ctx->graph_owned_freed.insert(d_ptr);
        return 0;
    }
    if (ctx->capturing) {
        // a block born in this capture may be recycled inside it, but must
        // never reach the general pool while the graph can still replay
        for (void *q : ctx->capture_blocks)
            if (q == d_ptr) {
                ctx->capture_free.emplace(sz, d_ptr);
                return 0;
            }
    }
    ctx->free_blocks.emplace(sz, d_ptr);
    return 0;
}

int th_pool_stats(th_ctx *ctx, size_t *bytes_reserved, size_t *bytes_in_use) {
    TH_REQUIRE(ctx, "th_pool_stats: null ctx");
    if (bytes_reserved) *bytes_reserved = ctx->bytes_reserved;
    if (bytes_in_use) *bytes_in_use = ctx->bytes_in_use;
    return 0;
}

int th_memcpy_h2d(th_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    TH_REQUIRE(ctx && (bytes == 0 || (d_dst && h_src)), "th_memcpy_h2d: null argument");
    if (bytes == 0) return 0;
    TH_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    // pageable host memory: the caller may reuse h_src as soon as we return
    TH_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int th_malloc_finegrained(th_ctx *ctx, size_t bytes, void **d_out) {
    TH_REQUIRE(ctx && d_out, "th_malloc_finegrained: null argument");
    TH_HIP(hipSetDevice(ctx->device));
    TH_HIP(hipExtMallocWithFlags(d_out, bytes ? bytes : 4, hipDeviceMallocFinegrained));
    return 0;
}

int th_free_finegrained(th_ctx *ctx, void *d_ptr) {
    TH_REQUIRE(ctx, "th_free_finegrained: null ctx");
    if (!d_ptr) return 0;
    TH_HIP(hipStreamSynchronize(ctx->stream));   // not stream-ordered: nothing in flight may still use it
    TH_HIP(hipFree(d_ptr));
    return 0;
}

int th_host_malloc(th_ctx *ctx, size_t bytes, void **h_out) {
    TH_REQUIRE(ctx && h_out, "th_host_malloc: null argument");
    TH_HIP(hipSetDevice(ctx->device));
    TH_HIP(hipHostMalloc(h_out, bytes ? bytes : 4, hipHostMallocMapped | hipHostMallocCoherent));
    return 0;
}

int th_host_free(th_ctx *ctx, void *h_ptr) {
    TH_REQUIRE(ctx, "th_host_free: null ctx");
    if (h_ptr) TH_HIP(hipHostFree(h_ptr));
    return 0;
}

int th_memcpy_d2h(th_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    TH_REQUIRE(ctx && (bytes == 0 || (h_dst && d_src)), "th_memcpy_d2h: null argument");
    if (bytes == 0) return 0;
    TH_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TH_HIP(hipStreamSynchronize(ctx->stream));
    return err_block_check(ctx);
}

int th_memcpy_d2d(th_ctx *ctx, void *d_dst, const void *d_src, size_t bytes) {
    TH_REQUIRE(ctx && (bytes == 0 || (d_dst && d_src)), "th_memcpy_d2d: null argument");
    if (bytes == 0) return 0;
    TH_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

int th_fill_f32(th_ctx *ctx, float *d_p, float v, size_t n) {
    TH_REQUIRE(ctx && (n == 0 || d_p), "th_fill_f32: null argument");
    if (n == 0) return 0;
    TH_REQUIRE(((uintptr_t)d_p & 15) == 0 || n < 4, "th_fill_f32: pointer must be 16-byte aligned");
    hipLaunchKernelGGL(fill_f32_kernel, dim3(ew_grid(n / 4 + 1, 256)), dim3(256), 0, ctx->stream, d_p, v, n);
    TH_LAUNCH_CHECK();
    return 0;
}

/* ---- events ----------------------------------------------------------- */
int th_event_create(th_event **out) {
    TH_REQUIRE(out, "th_event_create: null out");
    th_event *e = new th_event();
    TH_HIP(hipEventCreate(&e->ev));
    *out = e;
    return 0;
}

int th_event_destroy(th_event *ev) {
    if (!ev) return 0;
    (void)hipEventDestroy(ev->ev);
    delete ev;
    return 0;
}

int th_event_record(th_ctx *ctx, th_event *ev) {
    TH_REQUIRE(ctx && ev, "th_event_record: null argument");
    TH_HIP(hipEventRecord(ev->ev, ctx->stream));
    return 0;
}

int th_event_elapsed_ms(th_event *start, th_event *stop, float *ms_out) {
    TH_REQUIRE(start && stop && ms_out, "th_event_elapsed_ms: null argument");
    TH_HIP(hipEventSynchronize(stop->ev));
    TH_HIP(hipEventElapsedTime(ms_out, start->ev, stop->ev));
    return 0;
}

/* ---- graph capture ---------------------------------------------------- */
int th_graph_begin(th_ctx *ctx) {
    TH_REQUIRE(ctx, "th_graph_begin: null ctx");
    TH_REQUIRE(!ctx->capturing, "th_graph_begin: capture already in progress");
    // Relaxed: the pool may still hipMalloc while capturing (first step)
    TH_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    ctx->capturing = true;
    ctx->capture_blocks.clear();
    ctx->capture_free.clear();
    return 0;
}

int th_graph_end(th_ctx *ctx, th_graph **out) {
    TH_REQUIRE(ctx && out, "th_graph_end: null argument");
    TH_REQUIRE(ctx->capturing, "th_graph_end: no capture in progress");
    ctx->capturing = false;
    th_graph *g = new th_graph();
    g->ctx = ctx;
    hipError_t e = hipStreamEndCapture(ctx->stream, &g->graph);
    if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    // every block born in the capture stays pinned for the graph's lifetime
    g->blocks.swap(ctx->capture_blocks);
    for (void *p : g->blocks) ctx->graph_owned.insert(p);
    for (auto &kv : ctx->capture_free) ctx->graph_owned_freed.insert(kv.second);
    ctx->capture_free.clear();
    if (e != hipSuccess) {
        set_error("graph capture failed: %s", hipGetErrorString(e));
        th_graph_destroy(g);
        return 1;
    }
    *out = g;
    return 0;
}

int th_graph_launch(th_ctx *ctx, th_graph *g) {
    TH_REQUIRE(ctx && g && g->exec, "th_graph_launch: null argument");
    TH_HIP(hipGraphLaunch(g->exec, ctx->stream));
    return 0;
}

int th_graph_destroy(th_graph *g) {
    if (!g) return 0;
    th_ctx *ctx = g->ctx;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (ctx) {
        for (void *p : g->blocks) {
            ctx->graph_owned.erase(p);
            // blocks the host still holds are returned by its own th_free later;
            // blocks it already freed go back to the pool now.
            if (ctx->graph_owned_freed.erase(p)) ctx->free_blocks.emplace(ctx->block_size[p], p);
        }
    }
    delete g;
    return 0;
}

}  // extern "C"

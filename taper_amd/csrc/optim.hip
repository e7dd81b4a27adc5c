// optim.hip -- fused multi-tensor Adam / SGD over the flat parameter arenas
// (src/optim.rs:8-128), the batch gather of the data loader
// (src/data/mnist.rs:277-310) and the device-side step bookkeeping.
// All HBM-bound: Adam touches 28 B per parameter (p r/w, g r, m r/w, v r/w).
#include "adam_dev.h"

namespace th {

__device__ __forceinline__ int find_tensor(const int64_t *__restrict__ offsets, int n_tensors, int64_t i) {
    int lo = 0, hi = n_tensors;  // offsets[lo] <= i < offsets[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// One launch per step.  Every workgroup reads the step counter, forms this
// step's t and step_size = lr * sqrt(1 - b2^t) / (1 - b1^t) (optim.rs:84-90)
// itself.  pre_ticked == 0: t = old + 1 and the LAST workgroup to finish
// (agent-scope arrival counter in t_state[1]) publishes it -- by then every
// other workgroup has long read the old value, so a captured graph advances t
// on every replay without a separate tick kernel.  pre_ticked == 1: an earlier
// kernel of this step already advanced t_state[0] (fused-update steps).
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, const int64_t *__restrict__ offsets,
                                                   const int32_t *__restrict__ has_grad, int n_tensors, int64_t total,
                                                   int32_t *t_state, const float *__restrict__ lr, float beta1, float beta2,
                                                   float eps, float wd, int pre_ticked, int vec_ok, const uint32_t *__restrict__ guard) {
    // guard (nullable): the error word of the communicator whose all-reduce produced g.  Up = that all-reduce timed out and g still holds
    // THIS rank's gradients: no update, no tick (written by an earlier launch only: every workgroup reads the same value)
    if (guard && guard[0] != 0u) return;
    const int t = __hip_atomic_load(&t_state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (pre_ticked ? 0 : 1);  // optim.rs:84
    const float step = adam_step_size(lr[0], beta1, beta2, t);
    // four consecutive elements per thread (dwordx4 on all seven streams); a quad that straddles two tensors or the
    // end of the arena goes element by element
    const int64_t quads = (total + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (int64_t)gridDim.x * 256) {
        const int64_t i = q << 2;
        const int ti = find_tensor(offsets, n_tensors, i);
        if (vec_ok && i + 3 < total && (ti + 1 >= n_tensors || offsets[ti + 1] > i + 3)) {
            if (!has_grad[ti]) continue;  // grad None: skipped entirely (Q8)
            const float4 gv = *reinterpret_cast<const float4 *>(g + i);
            const float4 pv = *reinterpret_cast<const float4 *>(p + i);
            const float4 mv = *reinterpret_cast<const float4 *>(m + i);
            const float4 vv = *reinterpret_cast<const float4 *>(v + i);
            float4 po, mo, vo;
#define TH_ADAM_LANE(c)                                                      \
            {                                                                \
                const float gg = gv.c + wd * pv.c;                           \
                mo.c = beta1 * mv.c + (1.0f - beta1) * gg;                   \
                vo.c = beta2 * vv.c + (1.0f - beta2) * gg * gg;              \
                po.c = pv.c - step * mo.c / (sqrtf(vo.c) + eps);             \
            }
            TH_ADAM_LANE(x) TH_ADAM_LANE(y) TH_ADAM_LANE(z) TH_ADAM_LANE(w)
#undef TH_ADAM_LANE
            *reinterpret_cast<float4 *>(m + i) = mo;
            *reinterpret_cast<float4 *>(v + i) = vo;
            *reinterpret_cast<float4 *>(p + i) = po;
            continue;
        }
        for (int64_t j = i; j < i + 4 && j < total; ++j) {
            if (!has_grad[find_tensor(offsets, n_tensors, j)]) continue;
            adam_update(p, m, v, j, g[j], step, beta1, beta2, eps, wd);
        }
    }
    if (pre_ticked) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int arrived = __hip_atomic_fetch_add(&t_state[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (int)gridDim.x - 1) {
            __hip_atomic_store(&t_state[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&t_state[0], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void adam_tick_kernel(int32_t *t_state, const uint32_t *guard, uint32_t *step_word) {
    if (guard && guard[0] != 0u) return;   // th_ctx_set_update_guard
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        t_state[0] += 1;
        if (step_word) step_word[0] += 1u;
    }
}

// Adam on one contiguous slice with an already-ticked t (fallback of the fused
// entry points when the gradient came from the large-shape kernels)
__global__ __launch_bounds__(256) void adam_slice_kernel(AdamDev a, const float *__restrict__ g, int64_t n) {
    const float step = adam_dev_step(a);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        adam_update(a.p, a.m, a.v, i, g[i], step, a.beta1, a.beta2, a.eps, a.wd);
}

__global__ __launch_bounds__(256) void adam_slices_kernel(AdamSlices x) { adam_slices_block(x, blockIdx.x); }

__global__ __launch_bounds__(256) void sgd_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                  const int64_t *__restrict__ offsets, const int32_t *__restrict__ has_grad,
                                                  int n_tensors, int64_t total, const float *__restrict__ lr) {
    const float l = lr[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ti = find_tensor(offsets, n_tensors, i);
        if (!has_grad[ti]) continue;
        p[i] -= l * g[i];                                  // optim.rs:28-30
    }
}

// one workgroup per batch row; row_len floats copied as float4 when aligned
__global__ __launch_bounds__(256) void gather_batch_kernel(const float *__restrict__ images, const float *__restrict__ labels,
                                                           const int32_t *__restrict__ indices, int64_t n_indices,
                                                           const int64_t *__restrict__ cursor, int row_len,
                                                           float *__restrict__ out_images, float *__restrict__ out_labels) {
    const int64_t pos = ((cursor ? cursor[0] : 0) + blockIdx.x) % n_indices;
    const int64_t src = indices ? (int64_t)indices[pos] : pos;
    const float *s = images + src * row_len;
    float *d = out_images + (int64_t)blockIdx.x * row_len;
    if ((row_len & 3) == 0) {
        const float4 *s4 = reinterpret_cast<const float4 *>(s);
        float4 *d4 = reinterpret_cast<float4 *>(d);
        for (int i = threadIdx.x; i < row_len / 4; i += 256) d4[i] = s4[i];
    } else {
        for (int i = threadIdx.x; i < row_len; i += 256) d[i] = s[i];
    }
    if (threadIdx.x == 0) out_labels[blockIdx.x] = labels[src];
}

__global__ __launch_bounds__(256) void u8_to_unit_kernel(const uint8_t *__restrict__ in, float *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = (float)in[i] / 255.0f;  // data/mnist.rs:226
}

__global__ void log_step_kernel(const float *__restrict__ loss, const float *__restrict__ ncorrect, float *__restrict__ metrics,
                                int64_t capacity, int64_t *__restrict__ state, int64_t advance) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int64_t s = state[0] % capacity;
        metrics[2 * s] = loss ? loss[0] : 0.f;
        metrics[2 * s + 1] = ncorrect ? ncorrect[0] : 0.f;
        state[0] += 1;
        state[1] += advance;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(float *__restrict__ x, size_t n, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] *= scale;
}

}  // namespace th

namespace th {
int scale_inplace(th_ctx *ctx, float *d_x, size_t n, float scale);
}
using namespace th;

extern "C" {

int th_adam_step_guarded(th_ctx *ctx, float *d_params, const float *d_grads, float *d_m, float *d_v, const int64_t *d_offsets,
                         const int32_t *d_has_grad, int n_tensors, int64_t total, int32_t *d_t, const float *d_lr, float beta1,
                         float beta2, float eps, float weight_decay, int pre_ticked, const uint32_t *d_skip_if_nonzero) {
    TH_REQUIRE(ctx && d_params && d_grads && d_m && d_v && d_offsets && d_has_grad && d_t && d_lr, "th_adam_step: null argument");
    TH_REQUIRE(n_tensors > 0 && total >= 0, "th_adam_step: bad sizes");
    // the grid is never empty so that t always advances (optim.rs:84 increments even with no grads)
    const int grid = ew_grid((size_t)(total > 0 ? (total + 3) / 4 : 1), 256);
    const int vec_ok = (((uintptr_t)d_params | (uintptr_t)d_grads | (uintptr_t)d_m | (uintptr_t)d_v) & 15) == 0;   // dwordx4 streams
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_params, d_grads, d_m, d_v, d_offsets, d_has_grad,
                       n_tensors, total, d_t, d_lr, beta1, beta2, eps, weight_decay, pre_ticked, vec_ok, d_skip_if_nonzero);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_adam_step(th_ctx *ctx, float *d_params, const float *d_grads, float *d_m, float *d_v, const int64_t *d_offsets,
                 const int32_t *d_has_grad, int n_tensors, int64_t total, int32_t *d_t, const float *d_lr, float beta1,
                 float beta2, float eps, float weight_decay, int pre_ticked) {
    return th_adam_step_guarded(ctx, d_params, d_grads, d_m, d_v, d_offsets, d_has_grad, n_tensors, total, d_t, d_lr, beta1, beta2, eps,
                                weight_decay, pre_ticked, nullptr);
}

int th_adam_tick(th_ctx *ctx, int32_t *d_t) {
    TH_REQUIRE(ctx && d_t, "th_adam_tick: null argument");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, ctx->stream, d_t, ctx->update_guard, ctx->update_step_word);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_adam_slices(th_ctx *ctx, const th_adam_slice *slices, int n) {
    TH_REQUIRE(ctx && n >= 0 && n <= TH_MAX_ADAM_SLICES && (n == 0 || slices), "th_adam_slices: bad argument");
    const AdamSlices x = make_adam_slices(slices, n, ctx);
    if (x.blocks() == 0) return 0;
    hipLaunchKernelGGL(adam_slices_kernel, dim3(x.blocks()), dim3(256), 0, ctx->stream, x);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_scale(th_ctx *ctx, float *d_x, size_t n, float scale) {
    TH_REQUIRE(ctx && (n == 0 || d_x), "th_scale: null argument");
    return th::scale_inplace(ctx, d_x, n, scale);
}

int th_sgd_step(th_ctx *ctx, float *d_params, const float *d_grads, const int64_t *d_offsets, const int32_t *d_has_grad,
                int n_tensors, int64_t total, const float *d_lr) {
    TH_REQUIRE(ctx && d_params && d_grads && d_offsets && d_has_grad && d_lr, "th_sgd_step: null argument");
    if (total == 0) return 0;
    hipLaunchKernelGGL(sgd_kernel, dim3(ew_grid((size_t)total, 256)), dim3(256), 0, ctx->stream, d_params, d_grads, d_offsets,
                       d_has_grad, n_tensors, total, d_lr);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_gather_batch(th_ctx *ctx, const float *d_images, const float *d_labels, const int32_t *d_indices, int64_t n_indices,
                    const int64_t *d_cursor, int batch, int row_len, float *d_out_images, float *d_out_labels) {
    TH_REQUIRE(ctx && d_images && d_labels && d_out_images && d_out_labels, "th_gather_batch: null argument");
    TH_REQUIRE(batch >= 0 && row_len > 0 && n_indices > 0, "th_gather_batch: bad sizes");
    if (batch == 0) return 0;
    hipLaunchKernelGGL(gather_batch_kernel, dim3(batch), dim3(256), 0, ctx->stream, d_images, d_labels, d_indices, n_indices,
                       d_cursor, row_len, d_out_images, d_out_labels);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_u8_to_unit_f32(th_ctx *ctx, const uint8_t *d_in, float *d_out, size_t n) {
    TH_REQUIRE(ctx && (n == 0 || (d_in && d_out)), "th_u8_to_unit_f32: null argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(u8_to_unit_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ctx->stream, d_in, d_out, n);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_log_step(th_ctx *ctx, const float *d_loss, const float *d_ncorrect, float *d_metrics, int64_t capacity, int64_t *d_state,
                int64_t advance) {
    TH_REQUIRE(ctx && d_metrics && d_state && capacity > 0, "th_log_step: bad argument");
    hipLaunchKernelGGL(log_step_kernel, dim3(1), dim3(64), 0, ctx->stream, d_loss, d_ncorrect, d_metrics, capacity, d_state, advance);
    TH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

namespace th {
int adam_slice(th_ctx *ctx, const AdamDev &a, const float *d_g, int64_t n) {
    if (n == 0 || !a.p) return 0;
    hipLaunchKernelGGL(adam_slice_kernel, dim3(ew_grid((size_t)n, 256)), dim3(256), 0, ctx->stream, a, d_g, n);
    TH_LAUNCH_CHECK();
    return 0;
}

int scale_inplace(th_ctx *ctx, float *d_x, size_t n, float scale) {
    if (n == 0 || scale == 1.0f) return 0;
    hipLaunchKernelGGL(scale_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ctx->stream, d_x, n, scale);
    TH_LAUNCH_CHECK();
    return 0;
}
}  // namespace th

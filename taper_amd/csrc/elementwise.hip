// elementwise.hip -- HBM-bound element-wise kernels (dwordx4 loads, grid-
// stride, <= 8 blocks/CU).  Replaces the AVX/SSE/NEON loops of
// src/tensor.rs:36-161 and the scalar loops of src/ops.rs.
#include "common.h"

namespace th {

template <typename F>
__global__ __launch_bounds__(256) void ew1_kernel(const float *__restrict__ a, float *__restrict__ out, size_t n, F f) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n >> 2;
    const float4 *a4 = reinterpret_cast<const float4 *>(a);
    float4 *o4 = reinterpret_cast<float4 *>(out);
    for (size_t i = tid; i < n4; i += stride) {
        float4 x = a4[i], y = o4[i];
        y.x = f(x.x, y.x); y.y = f(x.y, y.y); y.z = f(x.z, y.z); y.w = f(x.w, y.w);
        o4[i] = y;
    }
    for (size_t i = (n4 << 2) + tid; i < n; i += stride) out[i] = f(a[i], out[i]);
}

// out = f(a, b, out_old)
template <typename F>
__global__ __launch_bounds__(256) void ew2_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                  float *__restrict__ out, size_t n, F f) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n >> 2;
    const float4 *a4 = reinterpret_cast<const float4 *>(a);
    const float4 *b4 = reinterpret_cast<const float4 *>(b);
    float4 *o4 = reinterpret_cast<float4 *>(out);
    for (size_t i = tid; i < n4; i += stride) {
        float4 x = a4[i], y = b4[i], z = o4[i];
        z.x = f(x.x, y.x, z.x); z.y = f(x.y, y.y, z.y); z.z = f(x.z, y.z, z.z); z.w = f(x.w, y.w, z.w);
        o4[i] = z;
    }
    for (size_t i = (n4 << 2) + tid; i < n; i += stride) out[i] = f(a[i], b[i], out[i]);
}

// write-only variants (no read of out): forward ops
template <typename F>
__global__ __launch_bounds__(256) void ew1w_kernel(const float *__restrict__ a, float *__restrict__ out, size_t n, F f) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n >> 2;
    const float4 *a4 = reinterpret_cast<const float4 *>(a);
    float4 *o4 = reinterpret_cast<float4 *>(out);
    for (size_t i = tid; i < n4; i += stride) {
        float4 x = a4[i], y;
        y.x = f(x.x); y.y = f(x.y); y.z = f(x.z); y.w = f(x.w);
        o4[i] = y;
    }
    for (size_t i = (n4 << 2) + tid; i < n; i += stride) out[i] = f(a[i]);
}

template <typename F>
__global__ __launch_bounds__(256) void ew2w_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                   float *__restrict__ out, size_t n, F f) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n >> 2;
    const float4 *a4 = reinterpret_cast<const float4 *>(a);
    const float4 *b4 = reinterpret_cast<const float4 *>(b);
    float4 *o4 = reinterpret_cast<float4 *>(out);
    for (size_t i = tid; i < n4; i += stride) {
        float4 x = a4[i], y = b4[i], z;
        z.x = f(x.x, y.x); z.y = f(x.y, y.y); z.z = f(x.z, y.z); z.w = f(x.w, y.w);
        o4[i] = z;
    }
    for (size_t i = (n4 << 2) + tid; i < n; i += stride) out[i] = f(a[i], b[i]);
}

// three inputs, two optional accumulate outputs (Div backward)
__global__ __launch_bounds__(256) void div_bwd_kernel(const float *__restrict__ g, const float *__restrict__ a,
                                                      const float *__restrict__ b, float *__restrict__ ga,
                                                      float *__restrict__ gb, size_t n) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n; i += stride) {
        float gv = g[i], bv = b[i];
        if (ga) ga[i] += gv / bv;                    // ops.rs:476-478
        if (gb) gb[i] -= gv * a[i] / (bv * bv);      // ops.rs:488-490
    }
}

__global__ __launch_bounds__(256) void add_scalar_dev_kernel(const float *__restrict__ s, float divisor,
                                                             float *__restrict__ g, size_t n) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const float v = s[0] / divisor;
    for (size_t i = tid; i < n; i += stride) g[i] += v;
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

template <typename K, typename... Args>
static int launch_ew(th_ctx *ctx, size_t n, K kernel, Args... args) {
    hipLaunchKernelGGL(kernel, dim3(ew_grid((n + 3) / 4, 256)), dim3(256), 0, ctx->stream, args...);
    TH_LAUNCH_CHECK();
    return 0;
}

}  // namespace th

using namespace th;

#define EW_ARGCHECK(name, ...)                                                        \
    TH_REQUIRE(ctx, name ": null ctx");                                               \
    if (n == 0) return 0;                                                             \
    {                                                                                 \
        const void *ptrs_[] = {__VA_ARGS__};                                          \
        for (const void *p_ : ptrs_) {                                                \
            TH_REQUIRE(p_, name ": null device pointer");                             \
            TH_REQUIRE(aligned16(p_), name ": device pointers must be 16-byte aligned"); \
        }                                                                             \
    }

extern "C" {

int th_add(th_ctx *ctx, const float *a, const float *b, float *out, size_t n) {
    EW_ARGCHECK("th_add", a, b, out);
    auto f = [] __device__(float x, float y) { return x + y; };
    return launch_ew(ctx, n, ew2w_kernel<decltype(f)>, a, b, out, n, f);
}

int th_sub(th_ctx *ctx, const float *a, const float *b, float *out, size_t n) {
    EW_ARGCHECK("th_sub", a, b, out);
    auto f = [] __device__(float x, float y) { return x - y; };
    return launch_ew(ctx, n, ew2w_kernel<decltype(f)>, a, b, out, n, f);
}

int th_mul(th_ctx *ctx, const float *a, const float *b, float *out, size_t n) {
    EW_ARGCHECK("th_mul", a, b, out);
    auto f = [] __device__(float x, float y) { return x * y; };
    return launch_ew(ctx, n, ew2w_kernel<decltype(f)>, a, b, out, n, f);
}

int th_div(th_ctx *ctx, const float *a, const float *b, float *out, size_t n) {
    EW_ARGCHECK("th_div", a, b, out);
    auto f = [] __device__(float x, float y) { return x / y; };
    return launch_ew(ctx, n, ew2w_kernel<decltype(f)>, a, b, out, n, f);
}

int th_axpy(th_ctx *ctx, float alpha, const float *x, float *y, size_t n) {
    EW_ARGCHECK("th_axpy", x, y);
    if (alpha == 1.0f) {  // accumulate_grad: g = g + src (ops.rs:131-136)
        auto f = [] __device__(float xv, float yv) { return yv + xv; };
        return launch_ew(ctx, n, ew1_kernel<decltype(f)>, x, y, n, f);
    }
    auto f = [alpha] __device__(float xv, float yv) { return yv + alpha * xv; };  // ops.rs:148-150
    return launch_ew(ctx, n, ew1_kernel<decltype(f)>, x, y, n, f);
}

int th_mul_bwd(th_ctx *ctx, const float *gout, const float *other, float *g, size_t n) {
    EW_ARGCHECK("th_mul_bwd", gout, other, g);
    auto f = [] __device__(float gv, float ov, float acc) { return acc + gv * ov; };
    return launch_ew(ctx, n, ew2_kernel<decltype(f)>, gout, other, g, n, f);
}

int th_div_bwd(th_ctx *ctx, const float *gout, const float *a, const float *b, float *ga, float *gb, size_t n) {
    TH_REQUIRE(ctx && gout && a && b, "th_div_bwd: null argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(div_bwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ctx->stream, gout, a, b, ga, gb, n);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_add_scalar_dev(th_ctx *ctx, const float *d_scalar, float divisor, float *g, size_t n) {
    TH_REQUIRE(ctx && d_scalar && g, "th_add_scalar_dev: null argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(add_scalar_dev_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ctx->stream, d_scalar, divisor, g, n);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_relu_fwd(th_ctx *ctx, const float *x, float *y, size_t n) {
    EW_ARGCHECK("th_relu_fwd", x, y);
    // _mm_max_ps(v, 0): NaN -> 0; `v > 0 ? v : 0` has the same table
    auto f = [] __device__(float v) { return v > 0.0f ? v : 0.0f; };
    return launch_ew(ctx, n, ew1w_kernel<decltype(f)>, x, y, n, f);
}

int th_relu_bwd(th_ctx *ctx, const float *x, const float *gout, float *gin, size_t n, int accumulate) {
    EW_ARGCHECK("th_relu_bwd", x, gout, gin);
    if (!accumulate) {
        auto f = [] __device__(float xv, float gv) { return xv > 0.0f ? gv : 0.0f; };
        return launch_ew(ctx, n, ew2w_kernel<decltype(f)>, x, gout, gin, n, f);
    }
    auto f = [] __device__(float xv, float gv, float acc) { return acc + (xv > 0.0f ? gv : 0.0f); };
    return launch_ew(ctx, n, ew2_kernel<decltype(f)>, x, gout, gin, n, f);
}

int th_sigmoid_fwd(th_ctx *ctx, const float *x, float *y, size_t n) {
    EW_ARGCHECK("th_sigmoid_fwd", x, y);
    auto f = [] __device__(float v) {
        if (v > 0.0f) {
            float e = expf(-v);
            return 1.0f / (1.0f + e);
        }
        float e = expf(v);
        return e / (1.0f + e);
    };
    return launch_ew(ctx, n, ew1w_kernel<decltype(f)>, x, y, n, f);
}

int th_sigmoid_bwd(th_ctx *ctx, const float *y, const float *gout, float *gin, size_t n) {
    EW_ARGCHECK("th_sigmoid_bwd", y, gout, gin);
    auto f = [] __device__(float s, float gv, float acc) { return acc + gv * s * (1.0f - s); };
    return launch_ew(ctx, n, ew2_kernel<decltype(f)>, y, gout, gin, n, f);
}

int th_exp_fwd(th_ctx *ctx, const float *x, float *y, size_t n) {
    EW_ARGCHECK("th_exp_fwd", x, y);
    auto f = [] __device__(float v) { return expf(v); };
    return launch_ew(ctx, n, ew1w_kernel<decltype(f)>, x, y, n, f);
}

int th_log_fwd(th_ctx *ctx, const float *x, float *y, size_t n) {
    EW_ARGCHECK("th_log_fwd", x, y);
    auto f = [] __device__(float v) { return logf(v); };
    return launch_ew(ctx, n, ew1w_kernel<decltype(f)>, x, y, n, f);
}

int th_log_bwd(th_ctx *ctx, const float *x, const float *gout, float *gin, size_t n) {
    EW_ARGCHECK("th_log_bwd", x, gout, gin);
    auto f = [] __device__(float xv, float gv, float acc) { return acc + gv / xv; };
    return launch_ew(ctx, n, ew2_kernel<decltype(f)>, x, gout, gin, n, f);
}

int th_pow_fwd(th_ctx *ctx, const float *x, float e, float *y, size_t n) {
    EW_ARGCHECK("th_pow_fwd", x, y);
    auto f = [e] __device__(float v) { return powf(v, e); };
    return launch_ew(ctx, n, ew1w_kernel<decltype(f)>, x, y, n, f);
}

int th_pow_bwd(th_ctx *ctx, const float *x, float e, const float *gout, float *gin, size_t n) {
    EW_ARGCHECK("th_pow_bwd", x, gout, gin);
    auto f = [e] __device__(float xv, float gv, float acc) { return acc + gv * e * powf(xv, e - 1.0f); };
    return launch_ew(ctx, n, ew2_kernel<decltype(f)>, x, gout, gin, n, f);
}

}  // extern "C"

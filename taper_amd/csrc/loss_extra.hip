// loss_extra.hip -- the losses and layers beside the two MNIST examples:
// binary cross-entropy (src/loss.rs:6-73, the XOR demo src/main.rs), the one-hot form of
// cross-entropy (src/loss.rs:201-245) and the Dropout mask (src/nn.rs:798-822).
#include "common.h"

namespace th {

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ float block_sum4(float v, float *sh) {  // 256 threads; valid in thread 0
    v = wave_sum64(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return threadIdx.x == 0 ? ((sh[0] + sh[1]) + sh[2]) + sh[3] : 0.f;
}

__device__ __forceinline__ float clamp_prob(float p) {  // f32::clamp(eps, 1 - eps), loss.rs:19 (NaN stays NaN)
    const float eps = 1e-7f;
    return p < eps ? eps : (p > 1.0f - eps ? 1.0f - eps : p);
}

// acc -= y ln(p) + (1 - y) ln(1 - p)   (loss.rs:18-22), fixed-order two-pass sum
__global__ __launch_bounds__(256) void bce_partial_kernel(const float *__restrict__ p, const float *__restrict__ y,
                                                          float *__restrict__ part, size_t n) {
    __shared__ float sh[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float pi = clamp_prob(p[i]), yi = y[i];
        s -= yi * logf(pi) + (1.0f - yi) * logf(1.0f - pi);
    }
    s = block_sum4(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void bce_final_kernel(const float *__restrict__ part, int nparts, float *__restrict__ out, float n) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = block_sum4(s, sh);
    if (threadIdx.x == 0) out[0] = s / n;   // loss.rs:23
}

// loss.rs:41-66: gp[i] += g * (-(y/p - (1-y)/(1-p))) / N ; gy[i] += g * (ln(1-p) - ln(p)) / N
__global__ __launch_bounds__(256) void bce_bwd_kernel(const float *__restrict__ p, const float *__restrict__ y,
                                                      const float *__restrict__ g0, float *__restrict__ gp,
                                                      float *__restrict__ gy, size_t n, int acc_p, int acc_y) {
    const float g = g0[0], fn = (float)n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float pi = clamp_prob(p[i]), yi = y[i];
        if (gp) {
            const float d = g * (-(yi / pi - (1.0f - yi) / (1.0f - pi))) / fn;
            gp[i] = acc_p ? gp[i] + d : d;
        }
        if (gy) {
            const float d = g * (logf(1.0f - pi) - logf(pi)) / fn;
            gy[i] = acc_y ? gy[i] + d : d;
        }
    }
}

// loss.rs:226-240: dlogits[i] (+)= (exp(logp[i]) - t[i]) * g / B
__global__ __launch_bounds__(256) void xent_onehot_bwd_kernel(const float *__restrict__ logp, const float *__restrict__ t,
                                                              const float *__restrict__ g0, float *__restrict__ d, size_t n,
                                                              float batch, int accumulate) {
    const float g = g0[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = (expf(logp[i]) - t[i]) * g / batch;
        d[i] = accumulate ? d[i] + v : v;
    }
}

// counter-based uniform in [0,1): splitmix64 of (seed, element index), top 24 bits.  The
// reference draws from an unseeded thread_rng (nn.rs:810), so any fixed generator is as faithful.
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(float *__restrict__ mask, size_t n, float p, float scale, uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        mask[i] = uniform01(seed, i) > p ? scale : 0.0f;   // nn.rs:814-818
}

}  // namespace th

using namespace th;

extern "C" {

int th_bce_fwd(th_ctx *ctx, const float *d_pred, const float *d_targets, size_t n, float *d_loss1) {
    TH_REQUIRE(ctx && d_loss1 && (n == 0 || (d_pred && d_targets)), "th_bce_fwd: bad argument");
    int nparts = (int)((n + 256 * 16 - 1) / (256 * 16));
    nparts = nparts < 1 ? 1 : (nparts > 1024 ? 1024 : nparts);
    void *part = nullptr;
    if (th_malloc(ctx, nparts * sizeof(float), &part)) return 1;
    hipLaunchKernelGGL(bce_partial_kernel, dim3(nparts), dim3(256), 0, ctx->stream, d_pred, d_targets, (float *)part, n);
    TH_LAUNCH_CHECK();
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, ctx->stream, (const float *)part, nparts, d_loss1, (float)n);
    TH_LAUNCH_CHECK();
    return th_free(ctx, part);
}

int th_bce_bwd(th_ctx *ctx, const float *d_pred, const float *d_targets, const float *d_g0, size_t n, float *d_gpred,
               float *d_gtargets, int accumulate_mask) {
    TH_REQUIRE(ctx && d_pred && d_targets && d_g0, "th_bce_bwd: null argument");
    if (n == 0 || (!d_gpred && !d_gtargets)) return 0;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ctx->stream, d_pred, d_targets, d_g0, d_gpred,
                       d_gtargets, n, accumulate_mask & 1, (accumulate_mask >> 1) & 1);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_xent_onehot_bwd(th_ctx *ctx, const float *d_logp, const float *d_targets, const float *d_g0, int batch, int classes,
                       float *d_dlogits, int accumulate) {
    TH_REQUIRE(ctx && d_logp && d_targets && d_g0 && d_dlogits && batch >= 0 && classes >= 0, "th_xent_onehot_bwd: bad argument");
    const size_t n = (size_t)batch * classes;
    if (n == 0) return 0;
    hipLaunchKernelGGL(xent_onehot_bwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ctx->stream, d_logp, d_targets, d_g0,
                       d_dlogits, n, (float)batch, accumulate);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_dropout_mask(th_ctx *ctx, float *d_mask, size_t n, float p, uint64_t seed) {
    TH_REQUIRE(ctx && (n == 0 || d_mask), "th_dropout_mask: null argument");
    TH_REQUIRE(p >= 0.0f && p <= 1.0f, "Dropout probability must be between 0 and 1");   // nn.rs:782-785
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ctx->stream, d_mask, n, p,
                       p < 1.0f ? 1.0f / (1.0f - p) : 0.0f, seed);
    TH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// quant.hip -- the reference's post-training-quantization STORAGE codecs that do real work (src/tensor.rs:2110-2288;
// Int4 / BFloat16 / NF4 are placeholders there): int8 affine quantisation over the finite min / max of the tensor, and the
// hand-rolled IEEE half conversion.  Integer / bit work: restated literally (including the half-up rounding whose mantissa
// carry is OR-ed, not added, into the exponent field) and held to bit-exact parity.  HBM-bound, one pass each.
#include "common.h"

namespace th {

// tensor.rs:2191-2238
__device__ __forceinline__ uint16_t f32_to_f16_bits(float value) {
    const uint32_t bits = __float_as_uint(value);
    const uint32_t sign = (bits >> 31) & 0x1, exponent = (bits >> 23) & 0xFF, mantissa = bits & 0x7FFFFF;
    if (exponent == 0xFF) return (uint16_t)((sign << 15) | (0x1Fu << 10) | (mantissa != 0 ? 0x200u : 0u));   // inf / NaN
    if (exponent == 0 && mantissa == 0) return (uint16_t)(sign << 15);                                          // +-0
    const int f16_exponent = (int)exponent - 127 + 15;
    if (f16_exponent >= 0x1F) return (uint16_t)((sign << 15) | (0x1Fu << 10));                                  // overflow -> inf
    if (f16_exponent <= 0) {
        if (f16_exponent < -10) return (uint16_t)(sign << 15);                                                  // underflow -> 0
        const int shift = 1 - f16_exponent;
        const uint32_t m = (mantissa | 0x800000u) >> (shift + 13);                                             // truncating
        return (uint16_t)((sign << 15) | m);
    }
    const uint32_t m = (mantissa + 0x1000u) >> 13;                                                              // round half up; a carry (0x400) is OR-ed below
    return (uint16_t)((sign << 15) | ((uint32_t)f16_exponent << 10) | m);
}

// tensor.rs:2241-2287
__device__ __forceinline__ float f16_bits_to_f32(uint16_t value) {
    const uint32_t bits = value, sign = (bits >> 15) & 0x1, exponent = (bits >> 10) & 0x1F, mantissa = bits & 0x3FF;
    if (exponent == 0x1F) return __uint_as_float((sign << 31) | (0xFFu << 23) | (mantissa != 0 ? mantissa << 13 : 0u));
    if (exponent == 0) {
        if (mantissa == 0) return __uint_as_float(sign << 31);
        int exp = -14;
        uint32_t mant = mantissa;
        while ((mant & 0x400) == 0) {
            mant <<= 1;
            exp -= 1;
        }
        mant &= 0x3FF;
        return __uint_as_float((sign << 31) | (((uint32_t)(exp + 127) & 0xFF) << 23) | (mant << 13));
    }
    return __uint_as_float((sign << 31) | (((exponent + 127 - 15) & 0xFF) << 23) | (mantissa << 13));
}

__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float *__restrict__ x, uint16_t *__restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = f32_to_f16_bits(x[i]);
}
__global__ __launch_bounds__(256) void f16_to_f32_kernel(const uint16_t *__restrict__ x, float *__restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = f16_bits_to_f32(x[i]);
}

// finite min / max (tensor.rs:2117-2125): per-block partials, then one block folds them; min / max are order-independent
__global__ __launch_bounds__(256) void minmax_finite_kernel(const float *__restrict__ x, size_t n, float *__restrict__ part) {
    __shared__ float smin[4], smax[4];
    float mn = INFINITY, mx = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        if (isfinite(v)) {
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_down(mn, off, 64));
        mx = fmaxf(mx, __shfl_down(mx, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = mn;
        smax[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        part[2 * blockIdx.x + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
}

// params = {min_val, scale} (tensor.rs:2127-2134), then q = round((x - min) / scale) as i32 + qmin, clamped (2136-2143)
__global__ __launch_bounds__(256) void int8_params_kernel(const float *__restrict__ part, int n_part, float *__restrict__ params) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = 0; i < n_part; ++i) {
        mn = fminf(mn, part[2 * i]);
        mx = fmaxf(mx, part[2 * i + 1]);
    }
    if (mn == mx) {
        mn -= 0.1f;
        mx += 0.1f;
    }
    params[0] = mn;
    params[1] = (mx - mn) / 255.0f;   // qrange = qmax - qmin = 255
}

__device__ __forceinline__ int rust_f32_as_i32(float v) {   // `as i32`: saturating, NaN -> 0
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (int)0x80000000;
    return (int)v;
}

__global__ __launch_bounds__(256) void quantize_int8_kernel(const float *__restrict__ x, int8_t *__restrict__ q, size_t n,
                                                            const float *__restrict__ params) {
    const float mn = params[0], scale = params[1];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int v = rust_f32_as_i32(roundf((x[i] - mn) / scale));                  // f32::round: half away from zero
        const long w = (long)v + (-128);                                              // i32 add (cannot overflow after the clamp range check below)
        q[i] = (int8_t)(w < -128 ? -128 : (w > 127 ? 127 : w));
    }
}

__global__ __launch_bounds__(256) void dequantize_int8_kernel(const int8_t *__restrict__ q, float *__restrict__ y, size_t n, float scale,
                                                              int zero_point, float min_val) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = (float)((int)q[i] - zero_point) * scale + min_val;                     // tensor.rs:357 (two roundings: contraction is off)
}

}  // namespace th

using namespace th;

extern "C" {

int th_f32_to_f16(th_ctx *ctx, const float *d_x, uint16_t *d_y, size_t n) {
    TH_REQUIRE(ctx && (n == 0 || (d_x && d_y)), "th_f32_to_f16: null argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ctx->stream, d_x, d_y, n);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_f16_to_f32(th_ctx *ctx, const uint16_t *d_x, float *d_y, size_t n) {
    TH_REQUIRE(ctx && (n == 0 || (d_x && d_y)), "th_f16_to_f32: null argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ctx->stream, d_x, d_y, n);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_quantize_int8(th_ctx *ctx, const float *d_x, int8_t *d_q, size_t n, float *d_params) {
    TH_REQUIRE(ctx && d_params && (n == 0 || (d_x && d_q)), "th_quantize_int8: null argument");
    const int blocks = n == 0 ? 1 : (int)std::min<size_t>(1024, (n + 255) / 256);
    void *part = nullptr;
    if (th_malloc(ctx, (size_t)blocks * 2 * sizeof(float), &part)) return 1;
    hipLaunchKernelGGL(minmax_finite_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_x, n, (float *)part);
    TH_LAUNCH_CHECK();
    hipLaunchKernelGGL(int8_params_kernel, dim3(1), dim3(64), 0, ctx->stream, (const float *)part, blocks, d_params);
    TH_LAUNCH_CHECK();
    if (n) {
        hipLaunchKernelGGL(quantize_int8_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ctx->stream, d_x, d_q, n, (const float *)d_params);
        TH_LAUNCH_CHECK();
    }
    return th_free(ctx, part);
}

int th_dequantize_int8(th_ctx *ctx, const int8_t *d_q, float *d_y, size_t n, float scale, int zero_point, float min_val) {
    TH_REQUIRE(ctx && (n == 0 || (d_q && d_y)), "th_dequantize_int8: null argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(dequantize_int8_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ctx->stream, d_q, d_y, n, scale, zero_point, min_val);
    TH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// adam_dev.h -- device-side Adam arithmetic shared by the stand-alone optimizer
// kernel (optim.hip) and the kernels that apply the update in their epilogue
// (gemm.hip: linear backward; head.hip: classifier head).  Literal restatement of
// src/optim.rs:83-113: eps is added to sqrt(v) before the bias
// correction is folded into the step size (quirk Q10).
#pragma once
#include "common.h"

namespace th {

// llvm.powi.f32 as lowered by compiler-rt __powisf2 (f32::powi, optim.rs:87-88)
__device__ __forceinline__ float powi_f32(float a, int b) {
    const bool recip = b < 0;
    float r = 1.0f;
    while (true) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0f / r : r;
}

// step_size = lr * sqrt(1 - b2^t) / (1 - b1^t)   (optim.rs:87-90)
__device__ __forceinline__ float adam_step_size(float lr, float beta1, float beta2, int t) {
    const float bc1 = 1.0f - powi_f32(beta1, t);
    const float bc2 = 1.0f - powi_f32(beta2, t);
    return lr * (sqrtf(bc2) / bc1);
}

// one element of optim.rs:99-110
__device__ __forceinline__ void adam_update(float *__restrict__ p, float *__restrict__ m, float *__restrict__ v, long i, float grad,
                                            float step, float beta1, float beta2, float eps, float wd) {
    const float pv = p[i];
    const float gv = grad + wd * pv;                       // optim.rs:101
    const float mv = beta1 * m[i] + (1.0f - beta1) * gv;   // optim.rs:104
    const float vv = beta2 * v[i] + (1.0f - beta2) * gv * gv;  // optim.rs:107
    m[i] = mv;
    v[i] = vv;
    p[i] = pv - step * mv / (sqrtf(vv) + eps);             // optim.rs:110
}

// th_adam_fuse on the device side (p == nullptr: no fused update)
struct AdamDev {
    float *p, *m, *v;
    const int32_t *t;     // ALREADY ticked for this step
    const float *lr;
    float beta1, beta2, eps, wd;
};

// A uniform value from global memory through the scalar unit (s_load): it travels on lgkmcnt, so waiting for it does
// not drain the vector loads in flight.  (As a vector load + readfirstlane the compiler parks an s_waitcnt vmcnt(0)
// right behind it: one full memory round trip per such value BEFORE the operand loads are even issued.)
// Safe for data written by an EARLIER launch only: the scalar cache is invalidated at kernel start.
template <class T>
__device__ __forceinline__ T sload(const T *p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>(reinterpret_cast<uintptr_t>(p));
}

// the counter was advanced by an earlier launch (th_adam_fuse: "ALREADY ticked"); nothing writes it while this one runs
__device__ __forceinline__ float adam_dev_step(const AdamDev &a) { return adam_step_size(sload(a.lr), a.beta1, a.beta2, sload(a.t)); }

static inline AdamDev make_adam_dev(const th_adam_fuse *f) {
    if (!f || !f->d_p) return AdamDev{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f};
    return AdamDev{f->d_p, f->d_m, f->d_v, f->d_t, f->d_lr, f->beta1, f->beta2, f->eps, f->weight_decay};
}

// Deferred updates carried by another launch (th_adam_slice): block b of the role handles 1024
// consecutive elements of one slice.
struct AdamSlices {
    AdamDev a[TH_MAX_ADAM_SLICES];
    const float *g[TH_MAX_ADAM_SLICES];
    int64_t n[TH_MAX_ADAM_SLICES];
    int first_block[TH_MAX_ADAM_SLICES + 1];   // prefix of blocks per slice
    int count;
    const uint32_t *guard;                     // nullable (th_ctx_set_update_guard): non-zero = a data-parallel exchange failed; apply nothing, tick nothing
    uint32_t *step_word;                       // nullable: advanced with every tick (the in-launch exchange's step number, dp_dev.h)
    int blocks() const { return first_block[count]; }
};

static inline AdamSlices make_adam_slices(const th_adam_slice *s, int n, const th_ctx *ctx) {
    AdamSlices r{};
    r.count = 0;
    r.guard = ctx ? ctx->update_guard : nullptr;
    r.step_word = ctx ? ctx->update_step_word : nullptr;
    for (int i = 0; i < n; ++i) {
        if (!s[i].f.d_p || s[i].n <= 0) continue;
        r.a[r.count] = make_adam_dev(&s[i].f);
        r.g[r.count] = s[i].d_g;
        r.n[r.count] = s[i].n;
        r.first_block[r.count + 1] = r.first_block[r.count] + (int)((s[i].n + 1023) / 1024);
        ++r.count;
    }
    for (int i = r.count; i < TH_MAX_ADAM_SLICES; ++i) r.first_block[i + 1] = r.first_block[r.count];
    return r;
}

// optim.rs:99-110 on elements [i0, i0 + 4) of one slice (i0 a multiple of 4)
__device__ __forceinline__ void adam_slice_quad(const AdamDev &a, const float *__restrict__ g, int64_t n, int64_t i0, float step) {
    if (i0 >= n) return;
    const bool vec = i0 + 4 <= n && ((((uintptr_t)a.p | (uintptr_t)a.m | (uintptr_t)a.v | (uintptr_t)g) & 15) == 0);
    float gv[4], pv[4], mv[4], vv[4];
    const int cnt = (int)(n - i0 < 4 ? n - i0 : 4);
    if (vec) {
        const float4 g4 = *reinterpret_cast<const float4 *>(g + i0), p4 = *reinterpret_cast<const float4 *>(a.p + i0);
        const float4 m4 = *reinterpret_cast<const float4 *>(a.m + i0), v4 = *reinterpret_cast<const float4 *>(a.v + i0);
        gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
        pv[0] = p4.x; pv[1] = p4.y; pv[2] = p4.z; pv[3] = p4.w;
        mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
        vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = i0 + (j < cnt ? j : 0);
            gv[j] = g[i]; pv[j] = a.p[i]; mv[j] = a.m[i]; vv[j] = a.v[i];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float gj = gv[j] + a.wd * pv[j];
        mv[j] = a.beta1 * mv[j] + (1.0f - a.beta1) * gj;
        vv[j] = a.beta2 * vv[j] + (1.0f - a.beta2) * gj * gj;
        pv[j] = pv[j] - step * mv[j] / (sqrtf(vv[j]) + a.eps);
    }
    if (vec) {
        *reinterpret_cast<float4 *>(a.p + i0) = make_float4(pv[0], pv[1], pv[2], pv[3]);
        *reinterpret_cast<float4 *>(a.m + i0) = make_float4(mv[0], mv[1], mv[2], mv[3]);
        *reinterpret_cast<float4 *>(a.v + i0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    } else {
        for (int j = 0; j < cnt; ++j) {
            a.p[i0 + j] = pv[j]; a.m[i0 + j] = mv[j]; a.v[i0 + j] = vv[j];
        }
    }
}

// 256 threads; elements [1024 b', 1024 b' + 1024) of the slice owning block b
__device__ __forceinline__ void adam_slices_block(const AdamSlices &x, int b) {
    if (x.guard && sload(x.guard) != 0u) return;   // (written by an earlier launch)
    int s = 0;
#pragma unroll
    for (int i = 1; i < TH_MAX_ADAM_SLICES; ++i)
        if (i < x.count && b >= x.first_block[i]) s = i;
    const AdamDev &a = x.a[s];
    const int64_t i0 = (int64_t)(b - x.first_block[s]) * 1024 + threadIdx.x * 4;
    if (i0 >= x.n[s]) return;
    adam_slice_quad(a, x.g[s], x.n[s], i0, adam_dev_step(a));
}

// ALL slices in ONE workgroup, with the step counter as it stands at entry; afterwards `tick`
// (nullable) is advanced by one (optim.rs:84).  For th_linear_fwd_ex: the slices belong to the
// PREVIOUS step, the tick opens the next one; nothing else in that launch touches the counter.
__device__ __forceinline__ void adam_slices_then_tick(const AdamSlices &x, int32_t *tick) {
    if (x.guard && sload(x.guard) != 0u) return;   // (uniform: the whole workgroup leaves before the barrier below)
#pragma unroll
    for (int s = 0; s < TH_MAX_ADAM_SLICES; ++s) {
        if (s >= x.count) break;
        const float step = adam_dev_step(x.a[s]);
        for (int64_t i0 = (int64_t)threadIdx.x * 4; i0 < x.n[s]; i0 += (int64_t)blockDim.x * 4) adam_slice_quad(x.a[s], x.g[s], x.n[s], i0, step);
    }
    __syncthreads();
    if (tick && threadIdx.x == 0) {
        tick[0] += 1;
        if (x.step_word) x.step_word[0] += 1u;
    }
}

}  // namespace th

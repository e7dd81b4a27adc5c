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

__device__ __forceinline__ float adam_dev_step(const AdamDev &a) {
    return adam_step_size(a.lr[0], a.beta1, a.beta2, __hip_atomic_load(a.t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

static inline AdamDev make_adam_dev(const th_adam_fuse *f) {
    if (!f || !f->d_p) return AdamDev{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f};
    return AdamDev{f->d_p, f->d_m, f->d_v, f->d_t, f->d_lr, f->beta1, f->beta2, f->eps, f->weight_decay};
}

}  // namespace th

// tail_dev.h -- device helpers shared by the fused classifier kernels (mlp_tail.hip, wide_head.hip): byte-offset loads from
// uniform bases, cross-lane pairs through v_permlane{16,32}_swap, and the row softmax held across the four lanes (r16, 0..3).
#pragma once
#include "adam_dev.h"

namespace th {

typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ldg_b(const float *base, unsigned byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ float4 ldg4_b(const float *base, unsigned byte_off) {
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + byte_off);
}
// lane l and lane l ^ 16 (resp. l ^ 32) both receive (x of the lower lane, x of the upper lane)
__device__ __forceinline__ void pair16(float x, float &lo, float &hi) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}
__device__ __forceinline__ void pair32(float x, float &lo, float &hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_over_g4(float x) {
    float lo, hi;
    pair16(x, lo, hi);
    x = lo + hi;
    pair32(x, lo, hi);
    return lo + hi;
}
__device__ __forceinline__ void argmax_step(float &best, int &bi, float v0, float v1, float i0, float i1) {
    const int j0 = __float_as_int(i0), j1 = __float_as_int(i1);
    const bool take1 = v1 > v0 || (v1 == v0 && j1 < j0);
    best = take1 ? v1 : v0;
    bi = take1 ? j1 : j0;
}

// Softmax cross-entropy of one row held as lg[i] = logit[class 4 g4 + i] over the four lanes (r16, 0..3)
// (loss.rs:101-195, 271-290).  lg is -inf for classes >= C.  Every lane of the row gets nll and argmax.
__device__ __forceinline__ void tail_row_softmax(const float (&lg)[4], int g4, int C, float tf, float inv_b, float (&dl)[4],
                                                 float &nll, int &argmax) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < 4; ++i) {          // first max; NaN / -inf never win (tensor.rs:1062)
        const bool w = lg[i] > best;
        best = w ? lg[i] : best;
        bi = w ? g4 * 4 + i : bi;
    }
    float v0, v1, i0, i1;
    pair16(best, v0, v1);
    pair16(__int_as_float(bi), i0, i1);
    argmax_step(best, bi, v0, v1, i0, i1);
    pair32(best, v0, v1);
    pair32(__int_as_float(bi), i0, i1);
    argmax_step(best, bi, v0, v1, i0, i1);
    if (bi == 0x7fffffff) bi = 0;
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) se += (g4 * 4 + i < C) ? expf(lg[i] - best) : 0.f;
    se = sum_over_g4(se);
    const float log_sum = logf(se);
    const int tc = (tf >= 0.f) ? (int)fminf(tf, 2147483520.f) : 0;   // Rust `as usize`: saturating, NaN -> 0
    float my_nll = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cls = g4 * 4 + i;
        const float lp = (lg[i] - best) - log_sum;   // loss.rs:117-125
        const float gv = expf(lp);                    // loss.rs:178
        const bool hit = cls == tc;
        my_nll = hit ? -lp : my_nll;
        dl[i] = (cls < C) ? (hit ? gv - 1.0f : gv) * inv_b : 0.f;   // loss.rs:185-188 with g0 = 1
    }
    my_nll = sum_over_g4(my_nll);
    nll = (tc >= C) ? NAN : my_nll;                   // the reference panics (loss.rs:161): NaN loss + a note the next wait turns into an error
    if (tc >= C) raise_target_oob(tc, C);
    argmax = bi;
}

}  // namespace th

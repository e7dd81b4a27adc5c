// experiments/mfma_overlap.hip -- does non-MFMA work of one wave hide under the MFMAs of its SIMD partner?  Every wave loops over
// { NACC back-to-back v_mfma_f32_16x16x4_f32 ; NV v_add_u32 (+ ND ds_read_b32) }, 1 or 2 waves per SIMD, all 256 CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC, int NV, int ND, bool PHASE>
__global__ __launch_bounds__(512, 1) void k(float *out, long long *clk, int iters) {
    __shared__ float lds[4096];
    floatx4 acc[NACC];
    float a = threadIdx.x * 0.001f, b[NACC];
    int v[NV > 0 ? NV : 1];
    for (int i = 0; i < NACC; ++i) { acc[i] = floatx4{0, 0, 0, 0}; b[i] = threadIdx.x * 0.002f - i; }
    for (int i = 0; i < NV; ++i) v[i] = threadIdx.x + i;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const bool late = PHASE && (threadIdx.x >> 8);   // waves 4-7: the SIMD partners of waves 0-3 start with the non-MFMA part
    const long long t0 = clock64(), w0 = wall_clock64();
    float ld[ND > 0 ? ND : 1];
    for (int i = 0; i < ND; ++i) ld[i] = 0.f;
    auto other = [&](int it) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += it;
#pragma unroll
        for (int i = 0; i < ND; ++i) ld[i] = lds[(v[i % (NV > 0 ? NV : 1)] + 64 * i) & 4095];
    };
    if (late) { other(0); __builtin_amdgcn_sched_barrier(0); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[i], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        other(it);
        __builtin_amdgcn_sched_barrier(0);
        if (ND) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) b[i] += ld[i % ND] * 1e-30f;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < NV; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { clk[2 * (threadIdx.x >> 6)] = t1 - t0; clk[2 * (threadIdx.x >> 6) + 1] = w1 - w0; }
}

template <class K>
static void run(const char *name, K kern, int threads, int nacc) {
    float *out; long long *clk, h[32] = {0};
    const int grid = 256, iters = 4000;
    hipMalloc(&out, (size_t)grid * threads * 4); hipMalloc(&clk, sizeof(h));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    long long cmax = 0, wmax = 0;
    for (int w = 0; w < waves; ++w) { if (h[2 * w] > cmax) cmax = h[2 * w]; if (h[2 * w + 1] > wmax) wmax = h[2 * w + 1]; }
    printf("%-58s waves/SIMD %d: %7.1f shader clk per loop iteration per SIMD (pure MFMA = %d), clock %.2f GHz\n", name, waves / 4,
           (double)cmax / iters, nacc * 32 * waves / 4, (double)cmax / ((double)wmax * 10.0));
    hipFree(out); hipFree(clk);
}

int main() {
    for (int th : {256, 512}) {
        run("13 MFMA", k<13, 0, 0, false>, th, 13);
        run("13 MFMA + 13 v_add", k<13, 13, 0, false>, th, 13);
        run("13 MFMA + 27 v_add", k<13, 27, 0, false>, th, 13);
        run("13 MFMA + 13 v_add + 14 ds_read (+13 v_fma on the results)", k<13, 13, 14, false>, th, 13);
    }
    run("13 MFMA + 27 v_add, partners in anti-phase", k<13, 27, 0, true>, 512, 13);
    run("13 MFMA + 13 v_add + 14 ds_read, partners in anti-phase", k<13, 13, 14, true>, 512, 13);
    run("7 MFMA + 7 v_add + 8 ds_read", k<7, 7, 8, false>, 512, 7);
    return 0;
}

// Experiment (not part of the library): what a k loop of independent fp32 16x16x4 MFMAs with one LDS operand read per two MFMAs (the conv
// chain's shape: 3 doubles + 1 single per k-step) costs per MFMA per SIMD (ideal 32 cycles), 1 or 2 waves per SIMD, the next step's reads
// issued (a) as a group in front of the step's MFMAs or (b) one behind each pair of MFMAs.  Time = the slowest wave of workgroup 0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: no reads; 1: grouped a step ahead; 2: spread a step ahead; 3: grouped, conflict-free addresses
__global__ __launch_bounds__(512, 1) void kloop(float *out, long long *clk, int iters) {
    __shared__ float lds[8192];
    __shared__ long long tmax;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1.0f;
    if (threadIdx.x == 0) tmax = 0;
    __syncthreads();
    floatx4 acc[7];
    for (int i = 0; i < 7; ++i) acc[i] = floatx4{0, 0, 0, 0};
    const float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f;
    const int lane = threadIdx.x & 63;
    // the chain's pattern: 16 consecutive floats per lane group, lane groups 272 floats apart (16 banks: the measured 0.37 conflict ratio)
    const float *p = lds + (MODE == 3 ? lane : (lane & 15) + 272 * (lane >> 4)) + 16 * (threadIdx.x >> 6);
    float b0[4], b1[4];
    for (int i = 0; i < 4; ++i) b0[i] = p[i * 1100];
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const float *q = p + (it & 7) * 32;
        if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b1[i] = q[i * 1100];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[i], acc[2 * i], 0, 0, 0);
            if (i < 3) acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0[i], acc[2 * i + 1], 0, 0, 0);
            if (MODE == 2) {
                __builtin_amdgcn_sched_barrier(0);
                b1[i] = q[i * 1100];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b0[i] = b1[i];
        }
    }
    long long t1 = clock64();
    atomicMax((unsigned long long *)&tmax, (unsigned long long)(t1 - t0));
    float s = 0;
    for (int i = 0; i < 7; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = tmax; clk[1] = t1 - t0; }
}
int main() {
    float *out; long long *clk, h[2];
    (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&clk, 64);
    const int iters = 2000;
    const char *names[4] = {"no reads", "reads grouped, a step ahead", "reads spread behind MFMA pairs", "grouped, conflict-free"};
#define RUN(MODE, threads)                                                                                                           \
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kloop<MODE>, dim3(256), dim3(threads), 0, 0, out, clk, iters); (void)hipDeviceSynchronize(); } \
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);                                                                              \
    printf("%-32s %d wave(s)/SIMD: %.1f cycles per MFMA per SIMD (slowest wave %lld clk, wave 0 %lld clk)\n", names[MODE], threads / 256, \
           (double)h[0] / (iters * 7 * (threads / 256.0)), h[0], h[1]);
    RUN(0, 256) RUN(0, 512) RUN(1, 256) RUN(1, 512) RUN(2, 256) RUN(2, 512) RUN(3, 256) RUN(3, 512)
    return 0;
}

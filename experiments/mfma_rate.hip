// experiments/mfma_rate.hip -- issue rate of v_mfma_f32_16x16x4_f32 / 32x32x2 on gfx950 as a function of how many independent
// accumulators a wave cycles through and of the waves per SIMD.  Times: shader clock (s_memtime) per wave, max over the workgroup's waves.
// build: hipcc -O3 --offload-arch=gfx950 -o experiments/_mfma_rate experiments/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(1024, 1) void k16(float *out, long long *clk, int iters) {
    floatx4 acc[NACC];
    float a[NACC], b[NACC];
    for (int i = 0; i < NACC; ++i) { acc[i] = floatx4{0, 0, 0, 0}; a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
    __syncthreads();
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { clk[2 * (threadIdx.x >> 6)] = t1 - t0; clk[2 * (threadIdx.x >> 6) + 1] = w1 - w0; }
}

template <int NACC>
__global__ __launch_bounds__(1024, 1) void k32(float *out, long long *clk, int iters) {
    floatx16 acc[NACC];
    float a[NACC], b[NACC];
    for (int i = 0; i < NACC; ++i) { for (int e = 0; e < 16; ++e) acc[i][e] = 0; a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
    __syncthreads();
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { clk[2 * (threadIdx.x >> 6)] = t1 - t0; clk[2 * (threadIdx.x >> 6) + 1] = w1 - w0; }
}

template <class K>
static void run(const char *name, K kern, int threads, int nacc, int grid, double flops_per_mfma) {
    float *out; long long *clk, h[32] = {0};
    hipMalloc(&out, (size_t)grid * threads * 4); hipMalloc(&clk, sizeof(h));
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, clk, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    long long cmax = 0, wmax = 0;
    for (int w = 0; w < waves; ++w) { if (h[2 * w] > cmax) cmax = h[2 * w]; if (h[2 * w + 1] > wmax) wmax = h[2 * w + 1]; }
    const double mf_per_simd = (double)iters * nacc * waves / 4.0;
    printf("%-10s acc %2d waves/SIMD %d grid %3d: %6.1f shader clk / MFMA / SIMD (slowest wave), %6.1f ns*100MHz-> %5.2f GHz, kernel %.1f us -> %6.1f TFLOP/s chip\n", name, nacc,
           waves / 4, grid, (double)cmax / mf_per_simd, (double)wmax * 10.0 / mf_per_simd, (double)cmax / ((double)wmax * 10.0), ms * 1e3,
           flops_per_mfma * iters * nacc * waves * grid / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(clk);
}

int main() {
    const int grid = 256;
#define R16(N) for (int th : {256, 512, 1024}) run("16x16x4", k16<N>, th, N, grid, 2048.0);
    R16(1) R16(2) R16(3) R16(4) R16(6) R16(7) R16(8) R16(10) R16(13) R16(16)
#define R32(N) for (int th : {256, 512}) run("32x32x2", k32<N>, th, N, grid, 4096.0);
    R32(1) R32(2) R32(4)
    return 0;
}

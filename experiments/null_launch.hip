// null_launch.hip -- what a launch costs before its first wave and behind its last one ("~5 us of every launch of the 512-thread /
// 35-140 KB-LDS conv kernels lie outside their workgroups").  A chain of dependent EMPTY kernels replayed as one hipGraph, by launch shape:
// threads per workgroup, dynamic LDS, registers (forced through an asm that touches v[R-1]), grid.  Stand-alone:
//   hipcc -O3 --offload-arch=gfx950 -o experiments/_null_launch experiments/null_launch.hip && experiments/_null_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int R>
__global__ __launch_bounds__(512, 1) void null_kernel(float *p) {
    extern __shared__ float lds[];
    if (R > 0) {
        if (R >= 250) asm volatile("v_mov_b32 v250, 0" ::: "v250");
        else if (R >= 120) asm volatile("v_mov_b32 v120, 0" ::: "v120");
    }
    if (p == (float *)1) { lds[threadIdx.x] = 1.f; p[0] = lds[0]; }   // never taken: keeps lds / p alive
}
// the same with a store per workgroup (a dirty line to write back at the kernel's end)
template <int R>
__global__ __launch_bounds__(512, 1) void store_kernel(float *p, int per_wg) {
    if (R >= 250) asm volatile("v_mov_b32 v250, 0" ::: "v250");
    for (int i = threadIdx.x; i < per_wg; i += blockDim.x) p[(long)blockIdx.x * per_wg + i] = 1.f;
}

template <typename F>
static double period_us(hipStream_t s, int n, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3 / (reps * n);
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float *p; CK(hipMalloc(&p, 64 << 20));
    CK(hipFuncSetAttribute((const void *)null_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10));
    CK(hipFuncSetAttribute((const void *)null_kernel<120>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10));
    CK(hipFuncSetAttribute((const void *)null_kernel<250>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10));
    const int N = 200;
    struct Cfg { int grid, threads, lds_kb, regs; } cfgs[] = {
        {1, 64, 0, 0}, {256, 64, 0, 0}, {256, 256, 0, 0}, {256, 512, 0, 0}, {256, 512, 53, 0}, {256, 512, 140, 0}, {256, 512, 0, 120},
        {256, 512, 0, 250}, {256, 512, 53, 250}, {256, 512, 140, 250}, {256, 1024, 0, 0}, {2048, 256, 0, 0}, {8192, 256, 0, 0}};
    printf("chain of %d dependent empty kernels in one hipGraph: us per launch\n", N);
    for (auto c : cfgs) {
        double us = 0;
        if (c.regs >= 250) us = period_us(s, N, [&] { hipLaunchKernelGGL(null_kernel<250>, dim3(c.grid), dim3(c.threads), c.lds_kb << 10, s, p); });
        else if (c.regs >= 120) us = period_us(s, N, [&] { hipLaunchKernelGGL(null_kernel<120>, dim3(c.grid), dim3(c.threads), c.lds_kb << 10, s, p); });
        else us = period_us(s, N, [&] { hipLaunchKernelGGL(null_kernel<0>, dim3(c.grid), dim3(c.threads), c.lds_kb << 10, s, p); });
        printf("  grid %5d x %4d threads, %3d KB LDS, >= %3d VGPRs: %6.2f us\n", c.grid, c.threads, c.lds_kb, c.regs, us);
    }
    printf("the same with each workgroup storing (dirty lines at the kernel's end), 256 x 512 threads, 250 VGPRs:\n");
    for (int kb : {0, 4, 16, 64}) {
        const int per = kb * 256;
        double us = period_us(s, N, [&] { hipLaunchKernelGGL(store_kernel<250>, dim3(256), dim3(512), 0, s, p, per); });
        printf("  %3d KB per workgroup (%5.1f MB per launch): %6.2f us\n", kb, kb * 256 / 1024.0, us);
    }
    return 0;
}

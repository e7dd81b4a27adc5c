// measures the shader clock actually in effect during (a) a lone tiny kernel chain and (b) a chip-filling load
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(long long *out, int spin) {
    long long w0 = wall_clock64(), c0 = clock64();
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; out[2] = (long long)x; }
}
__global__ void burn(float *p, int spin) {
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.f) p[0] = x;
}
int main() {
    long long *d, h[3];
    float *f;
    hipMalloc(&d, 64); hipMalloc(&f, 64);
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 2000);   // light load: chain of tiny kernels
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("light load: %lld wall ticks (100MHz) %lld shader cycles -> %.0f MHz\n", h[0], h[1], h[1] * 100.0 / h[0]);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(burn, dim3(256 * 8), dim3(256), 0, 0, f, 4000000);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 200000);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("after heavy load: %lld wall ticks %lld shader cycles -> %.0f MHz\n", h[0], h[1], h[1] * 100.0 / h[0]);
    }
    return 0;
}

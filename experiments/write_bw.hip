// experiments/write_bw.hip -- write-only bandwidth on MI355X: store width, temporal hint, grid size, bytes written.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float vfloat4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void wr(float4 *p, size_t n4, float v) {
    const float4 x = make_float4(v, v + 1, v + 2, v + 3);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if (MODE == 0) p[i] = x;
        else if (MODE == 1) __builtin_nontemporal_store(vfloat4{x.x, x.y, x.z, x.w}, reinterpret_cast<vfloat4 *>(&p[i]));
    }
}
__global__ __launch_bounds__(256) void wr1(float *p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
// each workgroup writes one contiguous chunk (like a conv workgroup writing its image's channel planes)
__global__ __launch_bounds__(256) void wr_chunk(float4 *p, size_t n4, float v) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x, b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n4 ? b0 + per : n4;
    const float4 x = make_float4(v, v + 1, v + 2, v + 3);
    for (size_t i = b0 + threadIdx.x; i < b1; i += 256) p[i] = x;
}
template <class F> static double timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3 / 20;
}
int main() {
    for (size_t mb : {6, 26, 188, 1024}) {
        size_t bytes = mb << 20; float *p; hipMalloc(&p, bytes); size_t n4 = bytes / 16;
        for (int grid : {256, 1024, 2048, 8192}) {
            double t0 = timeit([&] { hipLaunchKernelGGL(wr<0>, dim3(grid), dim3(256), 0, 0, (float4 *)p, n4, 1.f); });
            double t1 = timeit([&] { hipLaunchKernelGGL(wr<1>, dim3(grid), dim3(256), 0, 0, (float4 *)p, n4, 1.f); });
            double t2 = timeit([&] { hipLaunchKernelGGL(wr1, dim3(grid), dim3(256), 0, 0, p, n4 * 4, 1.f); });
            double t3 = timeit([&] { hipLaunchKernelGGL(wr_chunk, dim3(grid), dim3(256), 0, 0, (float4 *)p, n4, 1.f); });
            printf("%5zu MB grid %5d: float4 %7.1f us %5.2f TB/s | nontemporal %7.1f us %5.2f TB/s | dword %7.1f us %5.2f TB/s | chunked float4 %7.1f us %5.2f TB/s\n", mb, grid,
                   t0, bytes / t0 / 1e6, t1, bytes / t1 / 1e6, t2, bytes / t2 / 1e6, t3, bytes / t3 / 1e6);
        }
        hipFree(p);
    }
    return 0;
}

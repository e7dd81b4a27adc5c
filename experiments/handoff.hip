// experiments/handoff.hip -- what does a producer -> consumer hand-off INSIDE one launch cost against a kernel boundary?
// Models the MLP step: 32 producer workgroups (1024 threads) each finish a 1 KB slab of H after ~2.5 us of work; 216 consumer workgroups
// (256 threads) need ALL of H (32 KB) plus 8 KB of their own independent operands.
//   two launches : producers | boundary | consumers (load own operands, load H, reduce, store)
//   one launch   : consumers start with the producers, load their own operands, then wait for H:
//       flag form   : producer: stores, __threadfence(), one agent-scope atomic per workgroup; consumer: poll 32 flags, fence, plain loads of H
//       granule form: H travels as 8-byte {value, step tag} granules written / read with system-coherent (sc1) accesses: no fence, no flag
// Timed as 200-step chains (hipGraph would add nothing here: back-to-back launches in one stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int NP = 32, NC = 216, HN = 8192;   // producers, consumers, floats of H

__device__ __forceinline__ void busy_us(float us) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < (long long)(us * 100.f)) __builtin_amdgcn_s_sleep(1); }

__global__ __launch_bounds__(1024) void producer(float *h, float step, float work_us) {
    busy_us(work_us);
    if (threadIdx.x < 256) h[blockIdx.x * 256 + threadIdx.x] = step + threadIdx.x;
}
__global__ __launch_bounds__(256) void consumer(const float *__restrict__ h, const float *__restrict__ own, float *out) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < 2048; i += 256) acc += own[blockIdx.x * 2048 + i];
    for (int i = threadIdx.x; i < HN; i += 256) acc += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>   // 0: flags + fences, 1: granules
__global__ __launch_bounds__(1024) void merged(float *h, unsigned long long *g, unsigned *flags, const float *__restrict__ own, float *out, unsigned step,
                                               float work_us, int *err) {
    if (blockIdx.x < NP) {
        busy_us(work_us);
        if (MODE == 0) {
            if (threadIdx.x < 256) h[blockIdx.x * 256 + threadIdx.x] = (float)step + threadIdx.x;
            __syncthreads();
            if (threadIdx.x == 0) {
                __threadfence();
                __hip_atomic_store(&flags[blockIdx.x], step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (threadIdx.x < 256) {
            const float v = (float)step + threadIdx.x;
            const unsigned long long gran = ((unsigned long long)step << 32) | __float_as_uint(v);
            __hip_atomic_store(&g[blockIdx.x * 256 + threadIdx.x], gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (threadIdx.x >= 256) return;          // consumers use 256 threads of the block
    const int cb = blockIdx.x - NP;
    float acc = 0.f;
    for (int i = threadIdx.x; i < 2048; i += 256) acc += own[cb * 2048 + i];     // independent operands: under the producers' work
    const long long t0 = wall_clock64();
    if (MODE == 0) {
        if (threadIdx.x < NP) {
            while (__hip_atomic_load(&flags[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != step) {
                if (wall_clock64() - t0 > 2000000) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int i = threadIdx.x; i < HN; i += 256) acc += h[i];
    } else {
        for (int i = threadIdx.x; i < HN; i += 256) {
            unsigned long long gran;
            do {
                gran = __hip_atomic_load(&g[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(gran >> 32) == step) break;
                if (wall_clock64() - t0 > 2000000) { *err = 1; break; }
            } while (true);
            acc += __uint_as_float((unsigned)gran);
        }
    }
    out[cb * 256 + threadIdx.x] = acc;
}

int main() {
    float *h, *own, *out; unsigned long long *g; unsigned *flags; int *err;
    hipMalloc(&h, HN * 4); hipMalloc(&g, HN * 8); hipMalloc(&flags, 256); hipMalloc(&own, NC * 2048 * 4); hipMalloc(&out, NC * 256 * 4); hipMalloc(&err, 4);
    hipMemset(flags, 0, 256); hipMemset(g, 0, HN * 8); hipMemset(own, 0, NC * 2048 * 4); hipMemset(err, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int steps = 200;
    for (float work : {0.0f, 2.5f}) {
        float ms;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            for (int s = 1; s <= steps; ++s) {
                hipLaunchKernelGGL(producer, dim3(NP), dim3(1024), 0, 0, h, (float)s, work);
                hipLaunchKernelGGL(consumer, dim3(NC), dim3(256), 0, 0, h, own, out);
            }
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        }
        printf("producer work %.1f us: two launches      %6.2f us / step\n", work, ms * 1e3 / steps);
        unsigned base = 1000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            for (int s = 1; s <= steps; ++s) hipLaunchKernelGGL(merged<0>, dim3(NP + NC), dim3(1024), 0, 0, h, g, flags, own, out, base + s, work, err);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); base += 1000;
        }
        printf("producer work %.1f us: one launch, flags  %6.2f us / step\n", work, ms * 1e3 / steps);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            for (int s = 1; s <= steps; ++s) hipLaunchKernelGGL(merged<1>, dim3(NP + NC), dim3(1024), 0, 0, h, g, flags, own, out, base + s, work, err);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); base += 1000;
        }
        printf("producer work %.1f us: one launch, granules %5.2f us / step\n", work, ms * 1e3 / steps);
    }
    int e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
    printf("timeouts: %d\n", e);
    return 0;
}

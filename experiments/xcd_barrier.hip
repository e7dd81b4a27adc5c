// experiments/xcd_barrier.hip -- what does a barrier + data exchange cost between workgroups that all sit on ONE XCD (one L2)?
// A persistent step loop (forward tiles -> barrier -> backward tiles -> barrier) needs two such exchanges per training step; across XCDs
// every agent-scope fence is an L2 write-back / invalidate (handoff.hip: the kernel boundary wins).  Workgroups on the same XCD share their
// L2, so relaxed agent-scope accesses (sc1: past the CU's L1, served by the L2) with NO fence should do.
// Launch 8 * NWG workgroups; workgroup b runs on XCD b % 8 (checked: XCC_ID); only XCD 0's workgroups take part.  Per iteration every
// participant writes a 1 KB slab (relaxed agent stores), waits for its stores, adds 1 to a counter, spins until all NWG arrived, then reads
// all NWG slabs (relaxed agent loads) and checks them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); }   // HW_REG_XCC_ID[3:0]

template <int NWG>
__global__ __launch_bounds__(256) void persistent(float *slabs, unsigned *counter, int iters, int *err, long long *t_out, unsigned *xcc_seen) {
    if ((blockIdx.x & 7) != 0) return;
    const int me = blockIdx.x >> 3;
    if (threadIdx.x == 0) xcc_seen[me] = xcc_id();
    const long long t_begin = wall_clock64();
    float acc = 0.f;
    for (int it = 1; it <= iters; ++it) {
        // produce: slab[me][t] = it + t
        __hip_atomic_store(&slabs[(size_t)me * 256 + threadIdx.x], (float)it + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);            // vmcnt(0): the stores are in the L2
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it * NWG)) {
                if (wall_clock64() - t0 > 200000) { *err = 1; break; }   // 2 ms: never hang the GPU
            }
        }
        __syncthreads();
        if (*err) return;
        // consume: everybody's slab
        float s = 0.f;
#pragma unroll 8
        for (int w = 0; w < NWG; ++w) s += __hip_atomic_load(&slabs[(size_t)w * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s != (float)NWG * ((float)it + threadIdx.x)) *err = 2;
        acc += s;
        // (the next iteration overwrites the slabs: everybody must be done reading -> second barrier, as a real step loop would have)
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(counter + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it * NWG)) {
                if (wall_clock64() - t0 > 200000) { *err = 1; break; }
            }
        }
        __syncthreads();
        if (*err) return;
    }
    if (threadIdx.x == 0 && me == 0) t_out[0] = wall_clock64() - t_begin;
    if (acc == 12345.f) slabs[0] = acc;
}

template <int NWG> void run(int iters) {
    float *slabs; unsigned *counter, *xcc; int *err; long long *t;
    hipMalloc(&slabs, NWG * 256 * sizeof(float)); hipMalloc(&counter, 256); hipMalloc(&err, 4); hipMalloc(&t, 8); hipMalloc(&xcc, NWG * 4);
    hipMemset(counter, 0, 256); hipMemset(err, 0, 4); hipMemset(slabs, 0, NWG * 256 * sizeof(float));
    hipLaunchKernelGGL(persistent<NWG>, dim3(8 * NWG), dim3(256), 0, 0, slabs, counter, iters, err, t, xcc);
    hipDeviceSynchronize();
    int herr; long long ht; std::vector<unsigned> hx(NWG);
    hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), xcc, NWG * 4, hipMemcpyDeviceToHost);
    bool same = true; for (int i = 1; i < NWG; ++i) same = same && hx[i] == hx[0];
    printf("NWG %2d: %d iterations, err %d, all on XCC %u: %s, %.3f us per iteration (two barriers + one %d KB exchange)\n", NWG, iters, herr, hx[0], same ? "yes" : "NO",
           ht * 0.01 / iters, NWG);
    hipFree(slabs); hipFree(counter); hipFree(err); hipFree(t); hipFree(xcc);
}

int main() {
    run<8>(2000); run<16>(2000); run<32>(2000); run<32>(2000);
    return 0;
}

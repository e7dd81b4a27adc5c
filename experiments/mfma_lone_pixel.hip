// Experiment (not part of the library): a 7 x 7 conv layer is 49 pixels = 3 pixel tiles + ONE pixel; the chain spends a fourth tile of MFMAs on
// it (25 % of the k loop).  v_mfma_f32_16x16x4_f32 IS a chain of fp32 FMAs over k ascending (experiments/mfma_fma_chain.hip), so the lone pixel
// could ride on the vector ALU with the same bits.  What does a k-step then cost per SIMD (two waves per SIMD, 18 steps per pass)?
//   MODE 0: today's step -- 2 doubles: 4 MFMAs, 2 pixel reads, 2 weight loads
//   MODE 1: 3 MFMAs (one channel tile x 3 pixel tiles), 3 pixel reads, 1 weight load, + the lone pixel: one uniform ds_read_b128 (x), three
//           ds_bpermute (the other lane groups' weights), four dependent v_fma
//   MODE 2: as 1 without the lone pixel's work (the bound of the mapping alone)
//   MODE 3: as 1, the lone pixel's weights by three extra global loads instead of the bpermutes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void kloop(float *out, const float *wsrc, long long *clk, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    __shared__ long long tmax;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1.0f;
    if (threadIdx.x == 0) tmax = 0;
    __syncthreads();
    floatx4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = floatx4{0, 0, 0, 0};
    const int lane = threadIdx.x & 63, l16 = lane & 15, wave = threadIdx.x >> 6;
    const float *p = lds + l16 + 272 * (lane >> 4) + 16 * wave;
    const float *xq = lds + 4096 + 64 * wave;
    const float *wp = wsrc + lane + 64 * wave;
    float b0[3], b1[3], lone = 0.f;
    float wc[18], wd[18], wn[18], wm[18];      // this pass's weight operands (two channel tiles in MODE 0), the next pass's (requested a pass ahead)
    for (int i = 0; i < 18; ++i) { wc[i] = wp[i * 512]; wd[i] = wp[4096 + i * 512]; }
    for (int i = 0; i < 3; ++i) b0[i] = p[i * 1100];
    long long t0 = clock64();
    for (int pass = 0; pass < iters / 18; ++pass) {
        const float *wq = wp + (pass & 7) * 8192;
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            const float *q = p + (st & 7) * 32;
            const float wa = wc[st], wb = wd[st];
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, b0[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb, b0[0], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                b1[0] = q[0];
                wn[st] = wq[st * 256];
                __builtin_amdgcn_sched_barrier(0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, b0[1], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb, b0[1], acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                b1[1] = q[1100];
                wm[st] = wq[4096 + st * 256];
                __builtin_amdgcn_sched_barrier(0);
                b0[0] = b1[0]; b0[1] = b1[1];
            } else {
                float4 x4;
                float w1, w2, w3;
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, b0[0], acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                b1[0] = q[0];
                wn[st] = wq[st * 256];
                if (MODE == 1 || MODE == 3) x4 = *reinterpret_cast<const float4 *>(xq + 4 * (st & 7));
                if (MODE == 1) w1 = __shfl(wa, l16 + 16);
                if (MODE == 3) wm[st] = wq[4096 + st * 256];
                __builtin_amdgcn_sched_barrier(0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, b0[1], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                b1[1] = q[1100];
                if (MODE == 1) { w2 = __shfl(wa, l16 + 32); w3 = __shfl(wa, l16 + 48); }
                __builtin_amdgcn_sched_barrier(0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, b0[2], acc[2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                b1[2] = q[2200];
                if (MODE == 1) {
                    lone = __builtin_fmaf(wa, x4.x, lone);
                    lone = __builtin_fmaf(w1, x4.y, lone);
                    lone = __builtin_fmaf(w2, x4.z, lone);
                    lone = __builtin_fmaf(w3, x4.w, lone);
                }
                if (MODE == 3) {     // (the lone pixel's weights as one float4 per step, loaded a pass ahead like the operands: models a re-laid-out copy)
                    lone = __builtin_fmaf(wa, x4.x, lone);
                    lone = __builtin_fmaf(wb, x4.y, lone);
                    lone = __builtin_fmaf(wb, x4.z, lone);
                    lone = __builtin_fmaf(wb, x4.w, lone);
                }
                __builtin_amdgcn_sched_barrier(0);
                b0[0] = b1[0]; b0[1] = b1[1]; b0[2] = b1[2];
            }
        }
#pragma unroll
        for (int i = 0; i < 18; ++i) { wc[i] = wn[i]; if (MODE == 0 || MODE == 3) wd[i] = wm[i]; }
    }
    long long t1 = clock64();
    atomicMax((unsigned long long *)&tmax, (unsigned long long)(t1 - t0));
    float s = lone;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = tmax; clk[1] = t1 - t0; }
}
int main() {
    float *out, *w; long long *clk, h[2];
    (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&w, 1 << 20); (void)hipMalloc(&clk, 64);
    (void)hipMemset(w, 0, 1 << 20);
    const int iters = 2304;     // 16 passes of 144 steps
    const char *names[4] = {"today: 4 MFMA + 2 reads + 2 loads", "3 MFMA + lone pixel (bpermute)", "3 MFMA, no lone pixel", "3 MFMA + lone pixel (second load)"};
#define RUN(MODE)                                                                                                                    \
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kloop<MODE>, dim3(256), dim3(512), 0, 0, out, w, clk, iters); (void)hipDeviceSynchronize(); } \
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);                                                                              \
    printf("%-36s: %.1f cycles per k-step per SIMD (two waves; slowest wave %lld clk)\n", names[MODE], (double)h[0] / iters, h[0]);
    RUN(0) RUN(1) RUN(2) RUN(3)
    return 0;
}

// Does v_mfma_f32_16x16x4_f32 equal a chain of fp32 FMAs over its four k in order?  (Question behind computing a ragged pixel tile's
// last pixel on the vector ALU with the SAME bits as the matrix core: conv 7x7 = 49 pixels = 3 tiles + 1 pixel.)
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off experiments/mfma_fma_chain.hip -o /tmp/mfc && /tmp/mfc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float floatx4 __attribute__((ext_vector_type(4)));

// one wave: lane (l16, g4).  A[m = l16][k = g4], B[k = g4][n = l16]; D[m = 4 g4 + e][n = l16]
__global__ void k(const float *A, const float *B, const float *C, float *D, int steps) {
    const int lane = threadIdx.x, l16 = lane & 15, g4 = lane >> 4;
    floatx4 acc;
    for (int e = 0; e < 4; ++e) acc[e] = C[(4 * g4 + e) * 16 + l16];
    for (int s = 0; s < steps; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(s * 16 + l16) * 4 + g4], B[(s * 4 + g4) * 16 + l16], acc, 0, 0, 0);
    for (int e = 0; e < 4; ++e) D[(4 * g4 + e) * 16 + l16] = acc[e];
}

int main() {
    const int steps = 144, trials = 200;
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, steps * 64 * 4); hipMalloc(&dB, steps * 64 * 4); hipMalloc(&dC, 256 * 4); hipMalloc(&dD, 256 * 4);
    float *A = new float[steps * 64], *B = new float[steps * 64], C[256], D[256];
    long bad[4] = {0, 0, 0, 0}, total = 0;
    srand(1);
    for (int t = 0; t < trials; ++t) {
        const float sc = t % 3 == 0 ? 1.f : t % 3 == 1 ? 1e-3f : 37.f;
        for (int i = 0; i < steps * 64; ++i) { A[i] = sc * ((float)rand() / RAND_MAX * 2 - 1); B[i] = (float)rand() / RAND_MAX * 2 - 1; if (t % 5 == 4 && rand() % 3 == 0) A[i] = 0.f; }
        for (int i = 0; i < 256; ++i) C[i] = t % 2 ? 0.f : (float)rand() / RAND_MAX - 0.5f;
        hipMemcpy(dA, A, steps * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B, steps * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(dC, C, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, steps);
        hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                float r[4] = {C[m * 16 + n], C[m * 16 + n], C[m * 16 + n], C[m * 16 + n]};
                for (int s = 0; s < steps; ++s) {
                    float p[4];
                    for (int kk = 0; kk < 4; ++kk) p[kk] = 0;
                    for (int kk = 0; kk < 4; ++kk) r[0] = fmaf(A[(s * 16 + m) * 4 + kk], B[(s * 4 + kk) * 16 + n], r[0]);          // fma chain, k ascending
                    for (int kk = 3; kk >= 0; --kk) r[1] = fmaf(A[(s * 16 + m) * 4 + kk], B[(s * 4 + kk) * 16 + n], r[1]);         // descending
                    for (int kk = 0; kk < 4; ++kk) { volatile float pr = A[(s * 16 + m) * 4 + kk] * B[(s * 4 + kk) * 16 + n]; r[2] = r[2] + pr; }   // mul, add
                    { double d = r[3]; for (int kk = 0; kk < 4; ++kk) d += (double)A[(s * 16 + m) * 4 + kk] * B[(s * 4 + kk) * 16 + n]; r[3] = (float)d; }  // exact dot, one rounding
                }
                for (int v = 0; v < 4; ++v) { if (memcmp(&r[v], &D[m * 16 + n], 4)) ++bad[v]; }
                ++total;
            }
    }
    printf("outputs %ld; differing from the MFMA: fma chain k ascending %ld, descending %ld, mul+add %ld, exact 4-dot %ld\n", total, bad[0], bad[1], bad[2], bad[3]);
    return 0;
}

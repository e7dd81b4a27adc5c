// mlp3.hip -- th_mlp3_xent: a three-layer classifier (Linear + ReLU, Linear + ReLU, Linear, softmax cross-entropy: the tail of
// examples/train_mnist_cnn.rs:53-61 behind the global average pool, of examples/train_mnist.rs:40-48 behind the images) forward AND
// backward in TWO launches instead of five (two forwards, the fused head, the first layer's backward, ...).
//
// What makes it possible: everything except the parameter gradients is ROW-parallel.  The forward of a row, its loss term, its dlogits
// (softmax - onehot) / B, and the gradients of its activations dZ3 -> dZ2 -> dZ1 -> dX depend on that row only (nn.rs:54-60,
// loss.rs:101-195, ops.rs:254-265, 358-369); only dW_l = dZ_l^T . A_l and db_l add over the batch (ops.rs:266-294, tensor.rs:686-691).
//   launch 1  mlp3_rows_kernel: a workgroup of 8 waves owns 16 ROWS and walks them through all six products (three forward, three
//             backward) with the activations in LDS; weights are read from global memory (L2: the same 100-450 KB for every workgroup).
//             Writes what launch 2 needs -- H1, H2 and the ReLU-masked dZ1, dZ2, dlogits -- plus dX and the block's loss / hit sums.
//             Its barriers order LDS only (lds_barrier): the stores of a stage's results stay in flight under the next stage.
//             Its first thread also opens the optimizer step (t += 1, optim.rs:84) when asked to.
//   launch 2  mlp3_grads_kernel (gemm.hip): dW1, db1, dW2, db2, dW3, db3 as 16x16 tiles over the batch, Adam (optim.rs:99-110) in
//             every epilogue -- no workgroup of this launch reads a parameter, so nothing has to be deferred --, and the lead
//             workgroup's loss, hit count and step log.
// MFMA operand order: k-step s of a 16-deep block uses, for lane group g, element k = 4 g + s (one float4 per lane and block, as in
// sgemm_small16): every k once, sums within fp32 rounding of the k-ordered reference chain (tolerance 1e-4, like every GEMM here).
#include "tail_dev.h"

TH_USES_DEVICE_ERRORS()

namespace th {

int mlp3_grads_launch(th_ctx *ctx, const float *const dz[3], const float *const act[3], float *const dw[3], float *const db[3],
                      const int out_f[3], const int in_f[3], const th_adam_fuse *const wf[3], const th_adam_fuse *const bf[3], int batch,
                      const float *part, int n_blk, float *loss, float *ncorrect, float *metrics, int64_t capacity, int64_t *state,
                      int64_t advance, const float *gap_dx, const th_mlp3_gap *gap);   // gemm.hip

struct Mlp3RowsArgs {
    const float *x, *targets;
    const float *w1, *b1, *w2, *b2, *w3, *b3;   // [h1][in], [h1], [h2][h1], [h2], [c][h2], [c]
    int batch, in_f, h1, h2, c;
    float *a1, *a2;          // post-ReLU activations [B][h1], [B][h2]
    float *dz3, *dz2, *dz1;  // dlogits [B][c], masked dZ2 [B][h2], masked dZ1 [B][h1]
    float *dx;               // [B][in] (nullable)
    float *part;             // [blocks][2]
    int32_t *tick;           // nullable: t += 1
    int a1_given;            // a1 already holds relu(X W1^T + b1) (th_linear_fwd: a wide first layer is a launch of its own, see th_mlp3_xent)
};

constexpr int M3_NW = 8;     // waves per workgroup

#ifdef TH_PROFILE
__device__ long long g_m3_prof[16];     // wall clock (100 MHz) at the stage boundaries of workgroup 3
#define M3_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 3) g_m3_prof[i] = wall_clock64(); } while (0)
#else
#define M3_STAMP(i) do { } while (0)
#endif

// One 16 x 16 output tile over the workgroup's 16 rows: D[row 4 g4 + e][col n0 + l16] = sum_k A[row][k] * B(k, col).
//   A: LDS [16][lda] (lda a multiple of 4).  B_KC: B(k, col) = bg[col * ldb + k] (a weight matrix [out][in] used forward: a float4 along
//   k per 16-deep block); else B(k, col) = bg[k * ldb + col] (the same matrix used backward: four dword loads per block).
// The six products of the launch form a dependent chain, and a weight operand costs an L2 round trip (~1 us) -- so the first eight blocks
// of a stage's weights (all of them for K <= 128) are requested one stage AHEAD (m3_fetch), under the previous stage's MFMAs, LDS
// traffic and barrier; blocks beyond those are fetched in place.  Four accumulation chains (one per float4 component) keep the MFMAs
// of a tile independent of each other.
struct M3W { float4 b[8]; };

template <bool B_KC>
__device__ __forceinline__ const float *m3_bptr(const float *__restrict__ bg, int ldb, int n0, int lane) {
    const int l16 = lane & 15, g4 = lane >> 4;
    return B_KC ? bg + (long)(n0 + l16) * ldb + 4 * g4 : bg + (long)(4 * g4) * ldb + n0 + l16;
}
template <bool B_KC>
__device__ __forceinline__ float4 m3_bload(const float *bp, int ldb, int kk) {
    if (B_KC) return *reinterpret_cast<const float4 *>(bp + kk);
    const float *q = bp + (long)kk * ldb;
    return float4{q[0], q[ldb], q[2 * ldb], q[3 * ldb]};
}
template <bool B_KC>
__device__ __forceinline__ void m3_fetch(M3W &w, const float *__restrict__ bg, int ldb, int K, int n0, int lane, bool active) {
    if (!active) return;
    const float *bp = m3_bptr<B_KC>(bg, ldb, n0, lane);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (16 * u >= K) break;      // (K is a compile-time constant in the prefetching instances)
        w.b[u] = m3_bload<B_KC>(bp, ldb, 16 * u);
    }
}
// PRE: the first eight blocks are already in w (m3_fetch); else every block is fetched in place
template <bool B_KC, bool PRE>
__device__ __forceinline__ floatx4 m3_tile(const M3W &w, const float *as, int lda, const float *__restrict__ bg, int ldb, int K, int n0, int lane) {
    const int l16 = lane & 15, g4 = lane >> 4;
    const float *ap = as + l16 * lda + 4 * g4;
    floatx4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    if (PRE) {
        // (K is a compile-time constant here: the LDS reads of the whole tile are requested before its first MFMA)
        float4 av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (16 * u >= K) break;
            av[u] = *reinterpret_cast<const float4 *>(ap + 16 * u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (16 * u >= K) break;
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, w.b[u].x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, w.b[u].y, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, w.b[u].z, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, w.b[u].w, c3, 0, 0, 0);
        }
    } else {
        // runtime K: chunks of eight blocks, a chunk's weights requested before its first MFMA (a second register set for the next
        // chunk spills at this kernel's size: 35.7 us instead of 24.0 for the 784-128-64-10 classifier at batch 256)
        const float *bp = m3_bptr<B_KC>(bg, ldb, n0, lane);
        for (int k0 = 0; k0 < K; k0 += 128) {
            float4 b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + 16 * u >= K) break;
                b[u] = m3_bload<B_KC>(bp, ldb, k0 + 16 * u);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + 16 * u >= K) break;
                const float4 a = *reinterpret_cast<const float4 *>(ap + k0 + 16 * u);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[u].x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[u].y, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[u].z, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[u].w, c3, 0, 0, 0);
            }
        }
    }
    floatx4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (c0[e] + c1[e]) + (c2[e] + c3[e]);
    return r;
}

// PRE (INB, H1B, H2B = in / 16, h1 / 16, h2 / 16 <= 8 compiled in): one tile per wave and stage, weights requested one stage ahead.  Every
// wave issues the SAME loads whether it owns a tile of the stage or not (idle waves fetch tile wave % tiles): with loads behind
// wave-dependent branches the compiler can only wait for "all loads" (s_waitcnt vmcnt(0)) in front of a stage's first MFMA -- and
// that drains the NEXT stage's requests as well: one exposed L2 round trip per stage, 9.6 us for the six stages.
// !PRE: runtime sizes, loads in place.
template <bool PRE, int INB, int H1B, int H2B>
__global__ __launch_bounds__(64 * M3_NW) void mlp3_rows_kernel(Mlp3RowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l16 = lane & 15, g4 = lane >> 4;
    const int r0 = blockIdx.x * 16;
    const int in_f = PRE ? 16 * INB : a.in_f, H1 = PRE ? 16 * H1B : a.h1, H2 = PRE ? 16 * H2B : a.h2, C = a.c;
    const int ldx = in_f + 4, ld1 = H1 + 4, ld2 = H2 + 4;
    float *XS = lds, *A1S = XS + 16 * ldx, *A2S = A1S + 16 * ld1, *D3S = A2S + 16 * ld2, *D2S = D3S + 16 * 20, *D1S = D2S + 16 * ld2;
    if (a.tick && blockIdx.x == 0 && t == 0) a.tick[0] += 1;                       // optim.rs:84 (the launches that read t come later)
    const int T1 = H1 / 16, T2 = H2 / 16, TX = in_f / 16;
    M3_STAMP(0);

    M3W wa, wb;                                                                    // this stage's / the next stage's weight operands
    m3_fetch<true>(wa, a.w1, in_f, in_f, (wave % T1) * 16, lane, PRE);
    const float bias1 = a.b1 ? a.b1[(wave % T1) * 16 + l16] : 0.f, bias2 = a.b2 ? a.b2[(wave % T2) * 16 + l16] : 0.f;
    if (!PRE && a.a1_given) {
        // the first layer's output comes from its own launch: its rows -> LDS
        for (int q = t; q < 16 * (H1 >> 2); q += 64 * M3_NW) {
            const int r = q / (H1 >> 2), c4 = q % (H1 >> 2);
            *reinterpret_cast<float4 *>(A1S + r * ld1 + 4 * c4) = *reinterpret_cast<const float4 *>(a.a1 + (long)(r0 + r) * H1 + 4 * c4);
        }
    } else {
        // the block's rows of X -> LDS
        for (int q = t; q < 16 * (in_f >> 2); q += 64 * M3_NW) {
            const int r = q / (in_f >> 2), c4 = q % (in_f >> 2);
            *reinterpret_cast<float4 *>(XS + r * ldx + 4 * c4) = *reinterpret_cast<const float4 *>(a.x + (long)(r0 + r) * in_f + 4 * c4);
        }
    }
    m3_fetch<true>(wb, a.w2, H1, H1, (wave % T2) * 16, lane, PRE);
    lds_barrier();
    M3_STAMP(1);
    // ---- forward (nn.rs:54-60 + activation.rs:10-12): A1 = relu(X W1^T + b1), A2 = relu(A1 W2^T + b2) ----
    for (int tile = wave; tile < ((!PRE && a.a1_given) ? 0 : T1); tile += M3_NW) {
        const floatx4 acc = m3_tile<true, PRE>(wa, XS, ldx, a.w1, in_f, in_f, tile * 16, lane);
        const int col = tile * 16 + l16;
        const float bv = tile == wave ? bias1 : (a.b1 ? a.b1[col] : 0.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[e] + bv;
            v = v > 0.f ? v : 0.f;
            A1S[(4 * g4 + e) * ld1 + col] = v;
            a.a1[(long)(r0 + 4 * g4 + e) * H1 + col] = v;
        }
    }
    // (wave 0) the logits' weight operand: W3[class l16][k] (rows >= C: a copy of row C - 1, masked below)
    const float *w3p = a.w3 + (long)min(l16, C - 1) * H2 + 4 * g4;
    if (PRE) {       // (every wave: see above; only wave 0 uses them)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (16 * u >= H2) break;
            wa.b[u] = *reinterpret_cast<const float4 *>(w3p + 16 * u);
        }
    }
    float b3v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) b3v[e] = a.b3 ? a.b3[min(4 * g4 + e, C - 1)] : 0.f;
    const float tf = a.targets[r0 + l16];
    lds_barrier();
    M3_STAMP(2);
    for (int tile = wave; tile < T2; tile += M3_NW) {
        const floatx4 acc = m3_tile<true, PRE>(wb, A1S, ld1, a.w2, H1, H1, tile * 16, lane);
        const int col = tile * 16 + l16;
        const float bv = tile == wave ? bias2 : (a.b2 ? a.b2[col] : 0.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[e] + bv;
            v = v > 0.f ? v : 0.f;
            A2S[(4 * g4 + e) * ld2 + col] = v;
            a.a2[(long)(r0 + 4 * g4 + e) * H2 + col] = v;
        }
    }
    // dZ2's weight operand (K = the 16 classes): W3[class 4 g4 + s][col], class index clamped (dZ3 is zero beyond C)
    float w3b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) w3b[s] = PRE ? a.w3[(long)min(4 * g4 + s, C - 1) * H2 + (wave % T2) * 16 + l16] : 0.f;
    lds_barrier();
    M3_STAMP(3);
    // ---- logits^T[class 4 g4 + e][row l16] = W3 . A2^T + b3, softmax cross-entropy of the row (loss.rs:101-195, 271-290) ----
    if (wave == 0) {
        const float *hp = A2S + l16 * ld2 + 4 * g4;                           // B operand: row l16
        floatx4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
        for (int k0 = 0; k0 < H2; k0 += 128) {
            if (!PRE || k0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (k0 + 16 * u >= H2) break;
                    wa.b[u] = *reinterpret_cast<const float4 *>(w3p + k0 + 16 * u);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + 16 * u >= H2) break;
                const float4 h = *reinterpret_cast<const float4 *>(hp + k0 + 16 * u);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.b[u].x, h.x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.b[u].y, h.y, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.b[u].z, h.z, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.b[u].w, h.w, c3, 0, 0, 0);
            }
        }
        float lg[4], dl[4], nll_row;
        int bi;
#pragma unroll
        for (int e = 0; e < 4; ++e) lg[e] = 4 * g4 + e < C ? ((c0[e] + c1[e]) + (c2[e] + c3[e])) + b3v[e] : -INFINITY;
        tail_row_softmax(lg, g4, C, tf, 1.0f / (float)a.batch, dl, nll_row, bi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cls = 4 * g4 + e;
            D3S[l16 * 20 + cls] = dl[e];                                      // [row][class], zero for classes >= C
            if (cls < C) a.dz3[(long)(r0 + l16) * C + cls] = dl[e];
        }
        // the block's loss / hit sums (lanes 0..15 hold rows 0..15)
        const float nl = nll_row, ht = (fabsf((float)bi - tf) < 1e-6f) ? 1.f : 0.f;   // loss.rs:283
        float ns = nl, hs = ht;                       // a fixed tree over the 16 rows (lanes 0..15)
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            ns += __shfl_down(ns, off, 16);
            hs += __shfl_down(hs, off, 16);
        }
        if (lane == 0) {
            a.part[2 * blockIdx.x] = ns;
            a.part[2 * blockIdx.x + 1] = hs;
        }
    }
    // next but one: dZ1 = dZ2 . W2 reads W2[k][col]
    m3_fetch<false>(wa, a.w2, H1, H2, (wave % T1) * 16, lane, PRE);
    lds_barrier();
    M3_STAMP(4);
    // ---- backward of the activations: dZ2 = (dZ3 W3) * [A2 > 0], dZ1 = (dZ2 W2) * [A1 > 0], dX = dZ1 W1 (ops.rs:254-265, 358-369) ----
    for (int tile = wave; tile < T2; tile += M3_NW) {
        const int col = tile * 16 + l16;
        const float4 d = *reinterpret_cast<const float4 *>(D3S + l16 * 20 + 4 * g4);
        const float dv[4] = {d.x, d.y, d.z, d.w};
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float wv = (PRE && tile == wave) ? w3b[s] : a.w3[(long)min(4 * g4 + s, C - 1) * H2 + col];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[s], wv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 4 * g4 + e;
            const float v = A2S[row * ld2 + col] > 0.f ? acc[e] : 0.f;
            D2S[row * ld2 + col] = v;
            a.dz2[(long)(r0 + row) * H2 + col] = v;
        }
    }
    m3_fetch<false>(wb, a.w1, in_f, H1, (wave % TX) * 16, lane, PRE);      // (also without dX: the same loads on every path)
    lds_barrier();
    M3_STAMP(5);
    for (int tile = wave; tile < T1; tile += M3_NW) {
        const floatx4 acc = m3_tile<false, PRE>(wa, D2S, ld2, a.w2, H1, H2, tile * 16, lane);
        const int col = tile * 16 + l16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 4 * g4 + e;
            const float v = A1S[row * ld1 + col] > 0.f ? acc[e] : 0.f;
            D1S[row * ld1 + col] = v;
            a.dz1[(long)(r0 + row) * H1 + col] = v;
        }
    }
    M3_STAMP(7);
    if (!a.dx) return;
    lds_barrier();
    M3_STAMP(6);
    for (int tile = wave; tile < TX; tile += M3_NW) {
        const floatx4 acc = m3_tile<false, PRE>(wb, D1S, ld1, a.w1, in_f, H1, tile * 16, lane);
        const int col = tile * 16 + l16;
#pragma unroll
        for (int e = 0; e < 4; ++e) a.dx[(long)(r0 + 4 * g4 + e) * in_f + col] = acc[e];
    }
    M3_STAMP(9);
}

thread_local long t_mlp3_calls = 0;   // th_debug_mlp3_calls

static size_t mlp3_lds_bytes(int in_f, int h1, int h2) {
    return (size_t)16 * ((in_f + 4) + 2 * (h1 + 4) + 2 * (h2 + 4) + 20) * sizeof(float);
}

}  // namespace th

using namespace th;

extern "C" {

int th_mlp3_supported(int batch, int in_features, int h1, int h2, int classes) {
    return batch > 0 && batch % 16 == 0 && batch <= 4096 && in_features > 0 && in_features % 16 == 0 && in_features <= 1024 && h1 > 0 &&
                   h1 % 16 == 0 && h1 <= 256 && h2 > 0 && h2 % 16 == 0 && h2 <= 256 && classes >= 1 && classes <= 16 &&
                   mlp3_lds_bytes(in_features, h1, h2) <= (150u << 10)
               ? 1 : 0;
}

int th_mlp3_xent(th_ctx *ctx, const float *d_x, const float *d_targets, int batch, int in_features, const th_mlp3_layer *layers,
                 float *d_dx, float *d_loss, float *d_ncorrect, float *d_metrics, int64_t metrics_capacity, int64_t *d_state,
                 int64_t advance, int32_t *d_tick, const th_mlp3_gap *gap) {
    TH_REQUIRE(ctx && d_x && d_targets && layers && d_loss, "th_mlp3_xent: null argument");
    const int h1 = layers[0].out_features, h2 = layers[1].out_features, c = layers[2].out_features;
    TH_REQUIRE(th_mlp3_supported(batch, in_features, h1, h2, c),
               "th_mlp3_xent: needs batch, in_features, hidden sizes multiples of 16 (in <= 1024, hidden <= 256), classes <= 16 (got %d, %d, %d, %d, %d)",
               batch, in_features, h1, h2, c);
    for (int l = 0; l < 3; ++l) {
        TH_REQUIRE(layers[l].d_w && ((uintptr_t)layers[l].d_w & 15) == 0, "th_mlp3_xent: weights must be 16-byte aligned");
        TH_REQUIRE(!(layers[l].w_fuse && layers[l].w_fuse->d_p) || layers[l].d_dw, "th_mlp3_xent: a fused weight update needs d_dw");
        TH_REQUIRE(!(layers[l].b_fuse && layers[l].b_fuse->d_p) || layers[l].d_db, "th_mlp3_xent: a fused bias update needs d_db");
    }
    TH_REQUIRE(((uintptr_t)d_x & 15) == 0, "th_mlp3_xent: d_x must be 16-byte aligned");
    TH_REQUIRE(!d_metrics || (d_state && metrics_capacity > 0), "th_mlp3_xent: metrics need d_state and a capacity");
    TH_REQUIRE(!(gap && gap->d_cnt) || (d_dx && gap->d_gb && gap->hw > 0), "th_mlp3_xent: the gap bias finish needs d_dx, d_gb and hw > 0");
    // workspace: A1, A2, dZ1, dZ2, dZ3, the blocks' partial sums
    const int n_blk = batch / 16;
    const size_t n1 = (size_t)batch * h1, n2 = (size_t)batch * h2, n3 = (size_t)batch * c;
    const size_t n3p = (n3 + 3) & ~(size_t)3;
    void *ws = nullptr;
    if (th_malloc(ctx, (2 * n1 + 2 * n2 + n3p + 2 * (size_t)n_blk) * sizeof(float), &ws)) return 1;
    float *a1 = (float *)ws, *a2 = a1 + n1, *dz1 = a2 + n2, *dz2 = dz1 + n1, *dz3 = dz2 + n2, *part = dz3 + n3p;
    Mlp3RowsArgs r{};
    r.x = d_x; r.targets = d_targets;
    r.w1 = layers[0].d_w; r.b1 = layers[0].d_b; r.w2 = layers[1].d_w; r.b2 = layers[1].d_b; r.w3 = layers[2].d_w; r.b3 = layers[2].d_b;
    r.batch = batch; r.in_f = in_features; r.h1 = h1; r.h2 = h2; r.c = c;
    r.a1 = a1; r.a2 = a2; r.dz3 = dz3; r.dz2 = dz2; r.dz1 = dz1; r.dx = d_dx; r.part = part; r.tick = d_tick;
    // A first layer deeper than 128 is a latency chain of in / 128 weight round trips inside a 16-workgroup launch (18.7 us for 784-128-64);
    // as a launch of its own (th_linear_fwd: 128 workgroups, K split over waves) it is 5 us, and the rows launch starts from its output
    const bool split_l1 = in_features > 128;
    if (split_l1)
        if (int rc = th_linear_fwd(ctx, d_x, layers[0].d_w, layers[0].d_b, a1, batch, in_features, h1, 1)) return rc;
    r.a1_given = split_l1 ? 1 : 0;
    const size_t lds = mlp3_lds_bytes(in_features, h1, h2);
    if (in_features == 128 && h1 == 128 && h2 == 64) {   // the reference CNN's classifier (examples/train_mnist_cnn.rs:53-61): sizes compiled in
        (void)hipFuncSetAttribute((const void *)mlp3_rows_kernel<true, 8, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((mlp3_rows_kernel<true, 8, 8, 4>), dim3(n_blk), dim3(64 * M3_NW), lds, ctx->stream, r);
    } else {
        (void)hipFuncSetAttribute((const void *)mlp3_rows_kernel<false, 0, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((mlp3_rows_kernel<false, 0, 0, 0>), dim3(n_blk), dim3(64 * M3_NW), lds, ctx->stream, r);
    }
    TH_LAUNCH_CHECK();
    const float *dz[3] = {dz1, dz2, dz3}, *act[3] = {d_x, a1, a2};
    float *dw[3] = {layers[0].d_dw, layers[1].d_dw, layers[2].d_dw}, *db[3] = {layers[0].d_db, layers[1].d_db, layers[2].d_db};
    const int out_f[3] = {h1, h2, c}, in_f[3] = {in_features, h1, h2};
    const th_adam_fuse *wf[3] = {layers[0].w_fuse, layers[1].w_fuse, layers[2].w_fuse}, *bf[3] = {layers[0].b_fuse, layers[1].b_fuse, layers[2].b_fuse};
    if (int rc = mlp3_grads_launch(ctx, dz, act, dw, db, out_f, in_f, wf, bf, batch, part, n_blk, d_loss, d_ncorrect, d_metrics, metrics_capacity,
                                   d_state, advance, d_dx, gap)) {
        (void)th_free(ctx, ws);
        return rc;
    }
    ++t_mlp3_calls;
    return th_free(ctx, ws);
}

int th_debug_mlp3_calls(int64_t *out) {
    if (out) *out = t_mlp3_calls;
    return 0;
}

#ifdef TH_PROFILE
int th_debug_mlp3_prof(th_ctx *ctx, long long *h_out16) {
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(th::g_m3_prof), 16 * sizeof(long long)));
    return 0;
}
#endif

}  // extern "C"

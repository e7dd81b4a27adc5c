// mlp_epoch.hip -- many training steps of a 2-layer MLP in ONE launch: Linear(in, hid) + ReLU ->
// Linear(hid, classes) -> cross-entropy -> backward -> Adam, i.e. the whole loop body of
// examples/train_mnist.rs:89-121 / src/train.rs:98-124 for BASELINE configs[1] (784-128-10, batch 64).
//
// Why: at batch 64 a step is ~27 MFLOP and 3 MB -- the 3-launch fused step spends its 22 us on kernel
// boundaries (1.6 us each) and cold global round trips (every launch re-reads its operands from the
// fabric).  Here the state never leaves the chip: workgroup g of hid/16 owns hidden features
// [16g, 16g+16): its W1 rows live in LDS, their Adam moments in registers, W2 / b2 (1290 values) are
// replicated in every workgroup and updated identically.  Per step a workgroup
//   1. computes its H slab  relu(X . W1_slab^T + b1)          (X rows gathered straight from the dataset)
//   2. publishes the slab and collects the other slabs          (ONE all-gather, see below)
//   3. runs the classifier head on the full H (replicated)      logits, softmax-xent, dlogits, dW2, db2
//   4. dZ_slab = dlogits . W2[:, slab] * (H_slab > 0);  dW1_slab = dZ_slab^T . X;  db1
//   5. applies Adam (optim.rs:83-113) to its slab of W1 / b1 and to its replica of W2 / b2
// No kernel boundary, no gather launch, no parameter traffic; the only inter-workgroup traffic is the
// 4 KB H slab per step.
//
// All-gather (cdna_hip_programming.md Guideline 16, recipe R2): the payload is <= 4 KB per producer, so the
// data IS the flag: 8-byte {tag = step + 1, value} granules written with ONE agent-scope relaxed store each
// (sc1, write-through) and re-read with agent-scope relaxed loads until every tag matches -- no fence, no
// separate flag, placement-independent.  Two granule buffers alternate (a workgroup can be at most one step
// ahead of the slowest); the buffers are zeroed by a memset before every launch; every spin is bounded and a
// timeout makes all workgroups leave (the launch then reports an error instead of hanging the GPU).
#include "adam_dev.h"

namespace th {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int ME_T = 512;         // threads per workgroup: 8 waves, 2 per SIMD -> a 256-VGPR budget per lane
constexpr int ME_NW = ME_T / 64;
constexpr int ME_B = 64;          // batch rows per step (<=)
constexpr int ME_F = 16;          // hidden features per workgroup
constexpr int ME_CMAX = 16;       // classes <=
constexpr int ME_HMAX = 256;      // hidden <=  (16 workgroups)
constexpr int ME_KMAX = 1024;     // in_features <=
constexpr unsigned ME_SPIN_LIMIT = 1u << 22;

struct MlpEpochArgs {
    const float *images, *labels;     // [N][in_f], [N]
    const int32_t *indices;           // [n_indices] (nullable: identity)
    long n_indices, first_pos;
    int batch, n_steps, in_f, hid, classes;
    float *w1, *b1, *w2, *b2;         // parameters
    float *m_w1, *v_w1, *m_b1, *v_b1, *m_w2, *v_w2, *m_b2, *v_b2;
    int32_t *t_state;                 // Adam's t (th_adam_step's d_t)
    const float *lr;
    float beta1, beta2, eps, wd;
    float *metrics;                   // [capacity][2] {loss, n_correct}, slot = (state[0] + s) % capacity
    long capacity;
    long *state;                      // [2]: steps logged, samples consumed (advanced at the end)
    int xt_tiles;                     // column tiles of X per LDS chunk in the backward slab
    u64 *granules;                    // [2][n_wg][ME_B * ME_F]
    unsigned *timeout;                // != 0: a spin gave up
    int32_t *status;                  // caller-visible copy of the timeout word (nullable)
};

__device__ __forceinline__ long me_target_class(float tf) {  // Rust `as usize`: saturating, NaN -> 0
    return (tf >= 0.f) ? (long)fminf(tf, 2147483520.f) : 0;
}

#ifdef TH_PROFILE
__device__ long long g_me_prof[8];
#define ME_STAMP(i) do { const long long now_ = wall_clock64(); prof_acc[i] += now_ - prof_last; prof_last = now_; } while (0)
#else
#define ME_STAMP(i) do { } while (0)
#endif

template <int XS, int TQ>   // XS: k-steps (of 16 input columns) per wave in the forward slab = ceil(in_f/16 / 2);  TQ: column tiles of
                             // the W1 slab per wave in the backward slab = ceil(in_f/16 / 8)
__global__ __launch_bounds__(ME_T) void mlp2_epoch_kernel(MlpEpochArgs a) {
#ifdef TH_PROFILE
    long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_last = wall_clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    const int wg = blockIdx.x, n_wg = gridDim.x;
    const int K = a.in_f, H = a.hid, C = a.classes;
    const int LD1 = K + 4;                    // W1 slab pitch
    const int LDH = H + 4;                    // Hfull / W2 pitch
    // ---- LDS carve (floats) ----
    float *W1s = lds;                         // [16][LD1]          this workgroup's rows of W1
    float *Hf = W1s + ME_F * LD1;             // [64][LDH]          H of the current step, all features
    float *W2s = Hf + ME_B * LDH;             // [16][LDH]          W2 (rows >= C zero), replicated
    float *lg = W2s + ME_CMAX * LDH;          // [64][16]           logits
    float *dl = lg + ME_B * ME_CMAX;          // [64][16]           dlogits
    float *dz = dl + ME_B * ME_CMAX;          // [64][17]           dZ slab (pitch 17)
    float *part = dz + ME_B * 17;             // [4][64][4]         k-split partials (forward slab / logits)
    float *red = part + 4 * 64 * 4;           // [64]               block reductions
    float *dbp = red + 64;                    // [8][16]            per-wave column sums of dlogits
    float *b1s = dbp + ME_NW * ME_CMAX;       // [16] b1 slab, [16] b2
    float *b2s = b1s + 16;
    float *XT = b2s + 16;                     // [64][xt_tiles*16 + 4]  X columns of the current chunk (backward slab)
    const int XP = a.xt_tiles * 16 + 4;
    int *rowoff = reinterpret_cast<int *>(XT + ME_B * XP);       // [2][64] offset of each batch row in `images`, in float4 units (-1: none)
    float *tgt = reinterpret_cast<float *>(rowoff + 2 * ME_B);   // [2][64] targets

    // ---- load the persistent state ----
    const int f0 = wg * ME_F;
    for (int i = t; i < ME_F * (K / 4); i += ME_T) {
        const int f = i / (K / 4), kq = (i % (K / 4)) * 4;
        *reinterpret_cast<float4 *>(W1s + f * LD1 + kq) = *reinterpret_cast<const float4 *>(a.w1 + (long)(f0 + f) * K + kq);
    }
    for (int i = t; i < ME_CMAX * H; i += ME_T) {
        const int c = i / H, k = i % H;
        W2s[c * LDH + k] = c < C ? a.w2[c * H + k] : 0.f;
    }
    if (t < ME_F) b1s[t] = a.b1[f0 + t];
    if (t < ME_CMAX) b2s[t] = t < C ? a.b2[t] : 0.f;
    // Adam moments in registers, laid out like the accumulators that will meet them:
    //   W1 slab: wave owns column tiles ct = wave + 8 q; lane holds (feature 4 g4 + i, column ct*16 + r16)
    const int n_ct = K / 16;
    float m1[TQ][4], v1[TQ][4];
#pragma unroll
    for (int q = 0; q < TQ; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ct = wave + ME_NW * q;
            const long idx = (long)(f0 + 4 * g4 + i) * K + ct * 16 + r16;
            m1[q][i] = ct < n_ct ? a.m_w1[idx] : 0.f;
            v1[q][i] = ct < n_ct ? a.v_w1[idx] : 0.f;
        }
    //   W2: wave owns column tiles wave + 8 j (j < 2, H <= 256); lane holds (class 4 g4 + i, column tile*16 + r16)
    const int n_ht = H / 16;
    float m2[2][4], v2[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cls = 4 * g4 + i, ht = wave + ME_NW * j;
            const bool ok = ht < n_ht && cls < C;
            const long idx = ok ? (long)cls * H + ht * 16 + r16 : 0;
            m2[j][i] = ok ? a.m_w2[idx] : 0.f;
            v2[j][i] = ok ? a.v_w2[idx] : 0.f;
        }
    float mb1 = 0.f, vb1 = 0.f, mb2 = 0.f, vb2 = 0.f;     // threads < 16: b1 slab / b2
    if (t < ME_F) { mb1 = a.m_b1[f0 + t]; vb1 = a.v_b1[f0 + t]; }
    if (t < C) { mb2 = a.m_b2[t]; vb2 = a.v_b2[t]; }
    int adam_t = a.t_state[0];
    const float lr = a.lr[0];
    const long slot0 = a.metrics ? a.state[0] : 0;

    auto stage_rows = [&](int s, int par) {
        if (t < ME_B) {
            const long pos = a.first_pos + (long)s * a.batch + t;
            const bool ok = t < a.batch && pos < a.n_indices && s < a.n_steps;
            const long src = ok ? (a.indices ? (long)a.indices[pos] : pos) : -1;
            rowoff[par * ME_B + t] = ok ? (int)(src * (K / 4)) : -1;   // in float4 units: 2^31 quads = 34 GB of images
            tgt[par * ME_B + t] = ok ? a.labels[src] : 0.f;
        }
    };
    stage_rows(0, 0);
    __syncthreads();

    bool alive = true;
    for (int s = 0; s < a.n_steps && alive; ++s) {
        // The lane- / wave-derived indices are made opaque once per step: otherwise every LDS address built from them
        // is hoisted out of the step loop and kept live across it.
        int r16v = r16, g4v = g4, lanev = lane, wavev = __builtin_amdgcn_readfirstlane(wave);   // the wave index stays scalar:
        asm volatile("" : "+v"(r16v), "+v"(g4v), "+v"(lanev), "+s"(wavev));                      // its branches are s_cbranch, not exec masks
        const int par = s & 1;
        const long pos0 = a.first_pos + (long)s * a.batch;
        const int rows = (int)min((long)a.batch, a.n_indices - pos0);
        const unsigned epoch = (unsigned)s + 1u;
        const int *ro = rowoff + par * ME_B;
        ME_STAMP(7);
        stage_rows(s + 1, par ^ 1);                         // next step's rows: visible after this step's barriers

        // ================= 1. forward slab: H[64][16] = relu(X . W1s^T + b1) =================
        // Every X quad this lane needs is requested before the first MFMA: ONE round trip.
        const int rt1 = wavev & 3, ks1 = wavev >> 2;            // row tile, k half
        const int steps_per = (n_ct + 1) / 2;
        const int kb1 = ks1 * steps_per, ke1 = min(n_ct, kb1 + steps_per);
        {
            float4 xa[XS];
            const int xo = ro[rt1 * 16 + r16v];
            const float *xp = a.images + (long)(xo >= 0 ? xo : 0) * 4 + g4v * 4;
            const float *wp = W1s + r16v * LD1 + g4v * 4;
            floatx4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < XS; ++u) {
                const int kk = min(kb1 + u, n_ct - 1);
                xa[u] = *reinterpret_cast<const float4 *>(xp + kk * 16);
            }
#pragma unroll
            for (int u = 0; u < XS; ++u) {
                if (xo < 0 || kb1 + u >= ke1) xa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kb1 + u < ke1) {
                    const float4 w = *reinterpret_cast<const float4 *>(wp + (kb1 + u) * 16);
                    if (u & 1) {   // two accumulator chains: the dependent-MFMA latency (40 cycles) exceeds the issue time (32)
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].x, w.x, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].y, w.y, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].z, w.z, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].w, w.w, acc2, 0, 0, 0);
                    } else {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].x, w.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].y, w.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].z, w.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].w, w.w, acc, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += acc2[i];
            if (ks1 > 0) {
                float *pp = part + (rt1 * 64 + lanev) * 4;
                pp[0] = acc[0]; pp[1] = acc[1]; pp[2] = acc[2]; pp[3] = acc[3];
            }
            __syncthreads();
            if (ks1 == 0) {
                const float *pp = part + (rt1 * 64 + lanev) * 4;
                acc[0] += pp[0]; acc[1] += pp[1]; acc[2] += pp[2]; acc[3] += pp[3];
                // D map: row = rt*16 + 4 g4 + i (batch row), col = r16 (feature)
                u64 *gout = a.granules + ((long)par * n_wg + wg) * (ME_B * ME_F);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = rt1 * 16 + 4 * g4v + i;
                    float h = acc[i] + b1s[r16v];
                    h = h > 0.f ? h : 0.f;                       // ops.rs:312-349
                    Hf[row * LDH + f0 + r16v] = h;
                    __hip_atomic_store(gout + row * ME_F + r16v, ((u64)epoch << 32) | (u64)__float_as_uint(h), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        ME_STAMP(0);
        // ================= 2. all-gather of the H slabs (granules: the data is the flag) =================
        {
            const u64 *gin = a.granules + (long)par * n_wg * (ME_B * ME_F);
            bool done = true;
            for (int base = 0; base < n_wg && done; base += 4) {      // <= 4 slabs x 2 elements per sweep: all loads in flight at once
                unsigned spins = 0;
                done = false;
                while (!done) {
                    u64 gv[4][2];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int g = min(base + j, n_wg - 1);
                            gv[j][e] = __hip_atomic_load(gin + (long)g * (ME_B * ME_F) + t + ME_T * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int g = base + j, el = t + ME_T * e;
                            if (g >= n_wg || g == wg) continue;
                            if ((unsigned)(gv[j][e] >> 32) == epoch) Hf[(el >> 4) * LDH + g * ME_F + (el & 15)] = __uint_as_float((unsigned)gv[j][e]);
                            else ok = false;
                        }
                    done = __all(ok);
                    if (!done) {
                        if (++spins > ME_SPIN_LIMIT || __hip_atomic_load(a.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                            if (lanev == 0) {
                                __hip_atomic_store(a.timeout, 1u + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (a.status) a.status[0] = 1 + s;
                            }
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            // a wave that gave up must take the whole workgroup out: agree through LDS
            if (lanev == 0) red[wavev] = done ? 1.f : 0.f;
            __syncthreads();
            float all_ok = 1.f;
            for (int w = 0; w < ME_NW; ++w) all_ok = fminf(all_ok, red[w]);
            if (all_ok == 0.f) { alive = false; }
            __syncthreads();
            if (!alive) break;
        }
        ME_STAMP(1);
        // ================= 3. classifier head on the full H (replicated in every workgroup) =================
        const float inv_b = 1.0f / (float)rows;
        {   // logits: 4 row tiles x 2 halves of the hidden dimension
            const int tile = wavev & 3, kp = wavev >> 2;
            const int ksteps = H / 16;
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            const float *ap = Hf + (tile * 16 + r16v) * LDH;
            const float *bp = W2s + r16v * LDH;
            for (int ks = kp; ks < ksteps; ks += 2) {
                const float4 av = *reinterpret_cast<const float4 *>(ap + ks * 16 + g4v * 4);
                const float4 bv = *reinterpret_cast<const float4 *>(bp + ks * 16 + g4v * 4);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
            }
            if (kp > 0) {
                float *pp = part + (tile * 64 + lanev) * 4;
                pp[0] = acc[0]; pp[1] = acc[1]; pp[2] = acc[2]; pp[3] = acc[3];
            }
            __syncthreads();
            if (kp == 0) {
                const float *pp = part + (tile * 64 + lanev) * 4;
                acc[0] += pp[0]; acc[1] += pp[1]; acc[2] += pp[2]; acc[3] += pp[3];
#pragma unroll
                for (int i = 0; i < 4; ++i) lg[(tile * 16 + g4v * 4 + i) * ME_CMAX + r16v] = acc[i] + b2s[r16v];
            }
        }
        __syncthreads();
        ME_STAMP(2);
        {   // softmax cross-entropy of a row inside its 16 lanes (loss.rs:101-195, 271-290); two passes of 32 rows
            float nll = 0.f, hit = 0.f, cs = 0.f;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int row_l = (t >> 4) + 32 * pass, sub = t & 15;
                const bool active = row_l < rows, valid = active && sub < C;
                const float tf = tgt[par * ME_B + row_l];
                const float logit = valid ? lg[row_l * ME_CMAX + sub] : -INFINITY;
                float best = logit;
                int bi = (valid && logit > -INFINITY) ? sub : 0x7fffffff;     // NaN / -inf never win (tensor.rs:1062)
                if (bi == 0x7fffffff) best = -INFINITY;
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) {
                    const float ov = __shfl_xor(best, off, 64);
                    const int oi = __shfl_xor(bi, off, 64);
                    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
                }
                if (bi == 0x7fffffff) bi = 0;
                float se = valid ? expf(logit - best) : 0.f;
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) se += __shfl_xor(se, off, 64);
                const float log_sum = logf(se);
                float my_nll = 0.f, dlv = 0.f;
                const long cls = me_target_class(tf);
                if (valid) {
                    const float lp = (logit - best) - log_sum;       // loss.rs:117-125
                    float gv = expf(lp);                              // loss.rs:178
                    if (sub == cls) { my_nll = -lp; gv -= 1.0f; }
                    dlv = gv * inv_b;                                 // loss.rs:185-188 with g0 = 1
                }
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) my_nll += __shfl_xor(my_nll, off, 64);
                if (active && sub == 0) {
                    nll += (cls >= C) ? NAN : my_nll;
                    hit += (fabsf((float)bi - tf) < 1e-6f) ? 1.f : 0.f;   // loss.rs:283
                }
                dl[row_l * ME_CMAX + sub] = dlv;
                cs += dlv;                                             // db2 partial: this lane's rows
            }
            cs += __shfl_xor(cs, 16, 64);
            cs += __shfl_xor(cs, 32, 64);
            if (lanev < ME_CMAX) dbp[wavev * ME_CMAX + lanev] = cs;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                nll += __shfl_down(nll, off, 64);
                hit += __shfl_down(hit, off, 64);
            }
            if (lanev == 0) { red[wavev] = nll; red[16 + wavev] = hit; }
        }
        __syncthreads();
        ME_STAMP(3);
        // ---- dZ slab = (dlogits . W2[:, slab]) * (H_slab > 0): waves 0..3, one row tile each ----
        if (wavev < 4) {
            const int rt = wavev;
            const float4 av = *reinterpret_cast<const float4 *>(dl + (rt * 16 + r16v) * ME_CMAX + g4v * 4);   // A[row][class]
            const float *bp = W2s + f0 + r16v;                                                                // B[class][feature]
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bp[(g4v * 4 + 0) * LDH], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bp[(g4v * 4 + 1) * LDH], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bp[(g4v * 4 + 2) * LDH], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bp[(g4v * 4 + 3) * LDH], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rt * 16 + 4 * g4v + i;
                dz[row * 17 + r16v] = Hf[row * LDH + f0 + r16v] > 0.f ? acc[i] : 0.f;   // ops.rs:358-369 through the output (Q15)
            }
        }
        // ---- dW2[class][col] = dlogits^T . H (ops.rs:280-291): wave owns column tiles wave + 8 j ----
        floatx4 dw2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            dw2[j] = floatx4{0.f, 0.f, 0.f, 0.f};
            const int ht = wavev + ME_NW * j;
            if (ht >= n_ht) continue;
            const float *bp = Hf + ht * 16 + r16v;
            floatx4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < ME_B / 16; ks += 2)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r1 = ks * 16 + g4v * 4 + u, r2 = r1 + 16;
                    dw2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(dl[r1 * ME_CMAX + r16v], bp[r1 * LDH], dw2[j], 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(dl[r2 * ME_CMAX + r16v], bp[r2 * LDH], acc2, 0, 0, 0);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) dw2[j][i] += acc2[i];
        }
        // step log (workgroup 0)
        if (wg == 0 && t == 0 && a.metrics) {
            float n = 0.f, hs = 0.f;
            for (int w = 0; w < ME_NW; ++w) { n += red[w]; hs += red[16 + w]; }
            const long slot = (slot0 + s) % a.capacity;
            a.metrics[2 * slot] = n / (float)rows;               // loss.rs:164
            a.metrics[2 * slot + 1] = hs;
        }
        __syncthreads();   // dz complete; every read of W2s / Hf of this step is done
        ME_STAMP(4);
        // ================= 5a. Adam on the W2 / b2 replica (optim.rs:83-113) =================
        adam_t += 1;                                                       // optim.rs:84
        const float step_sz = adam_step_size(lr, a.beta1, a.beta2, adam_t);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ht = wavev + ME_NW * j;
            if (ht >= n_ht) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cls = 4 * g4v + i;
                if (cls < C) {
                    float *pw = W2s + cls * LDH + ht * 16 + r16v;
                    const float pv = *pw, gv = dw2[j][i] + a.wd * pv;
                    m2[j][i] = a.beta1 * m2[j][i] + (1.0f - a.beta1) * gv;
                    v2[j][i] = a.beta2 * v2[j][i] + (1.0f - a.beta2) * gv * gv;
                    *pw = pv - step_sz * m2[j][i] / (sqrtf(v2[j][i]) + a.eps);
                }
            }
        }
        if (t < C) {
            float g = 0.f;
#pragma unroll
            for (int w = 0; w < ME_NW; ++w) g += dbp[w * ME_CMAX + t];
            const float pv = b2s[t], gv = g + a.wd * pv;
            mb2 = a.beta1 * mb2 + (1.0f - a.beta1) * gv;
            vb2 = a.beta2 * vb2 + (1.0f - a.beta2) * gv * gv;
            b2s[t] = pv - step_sz * mb2 / (sqrtf(vb2) + a.eps);
        }
        ME_STAMP(5);
        // ================= 4 + 5b. dW1 slab = dZ^T . X, Adam on the slab =================
        // X is requested again in the forward slab's layout (row-contiguous quads, all in flight at once: one round trip; holding
        // it in registers across the head costs more VGPRs than a 512-thread workgroup has), then goes chunk by chunk through
        // XT[row][col] in LDS, where the owner of a column tile reads it as the B operand X[k = row][j = col].
        {
            float4 xa[XS];
            {
                const int xo = ro[rt1 * 16 + r16v];
                const float *xp = a.images + (long)(xo >= 0 ? xo : 0) * 4 + g4v * 4;
#pragma unroll
                for (int u = 0; u < XS; ++u) {
                    const int kk = min(kb1 + u, n_ct - 1);
                    xa[u] = *reinterpret_cast<const float4 *>(xp + kk * 16);
                }
#pragma unroll
                for (int u = 0; u < XS; ++u)
                    if (xo < 0 || kb1 + u >= ke1) xa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const int n_chunks = (n_ct + a.xt_tiles - 1) / a.xt_tiles;
            for (int c = 0; c < n_chunks; ++c) {
                const int t0c = c * a.xt_tiles, t1c = min(n_ct, t0c + a.xt_tiles);
                if (c > 0) __syncthreads();              // the previous chunk's readers are done with XT
#pragma unroll
                for (int u = 0; u < XS; ++u) {
                    const int kstep = kb1 + u;
                    if (kstep < ke1 && kstep >= t0c && kstep < t1c)
                        *reinterpret_cast<float4 *>(XT + (rt1 * 16 + r16v) * XP + (kstep - t0c) * 16 + g4v * 4) = xa[u];
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < TQ; ++q) {
                    const int ct = wavev + ME_NW * q;
                    if (ct < t0c || ct >= t1c) continue;
                    floatx4 acc = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
                    const float *xcol = XT + (ct - t0c) * 16 + r16v;
#pragma unroll
                    for (int uu = 0; uu < 16; uu += 2) {    // k = batch row (uu >> 2) * 16 + 4 g4 + (uu & 3)
                        const int b0 = (uu >> 2) * 16 + 4 * g4v + (uu & 3), b1r = b0 + 1;
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dz[b0 * 17 + r16v], xcol[b0 * XP], acc, 0, 0, 0);
                        accb = __builtin_amdgcn_mfma_f32_16x16x4f32(dz[b1r * 17 + r16v], xcol[b1r * XP], accb, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {        // D map: row = feature 4 g4 + i, col = ct*16 + r16
                        float *pw = W1s + (4 * g4v + i) * LD1 + ct * 16 + r16v;
                        const float pv = *pw, gv = (acc[i] + accb[i]) + a.wd * pv;
                        m1[q][i] = a.beta1 * m1[q][i] + (1.0f - a.beta1) * gv;
                        v1[q][i] = a.beta2 * v1[q][i] + (1.0f - a.beta2) * gv * gv;
                        *pw = pv - step_sz * m1[q][i] / (sqrtf(v1[q][i]) + a.eps);
                    }
                }
            }
        }
        if (t < ME_F) {                               // db1 (tensor.rs:686-691) + Adam
            float g = 0.f;
            for (int r = 0; r < ME_B; ++r) g += dz[r * 17 + t];
            const float pv = b1s[t], gv = g + a.wd * pv;
            mb1 = a.beta1 * mb1 + (1.0f - a.beta1) * gv;
            vb1 = a.beta2 * vb1 + (1.0f - a.beta2) * gv * gv;
            b1s[t] = pv - step_sz * mb1 / (sqrtf(vb1) + a.eps);
        }
        __syncthreads();   // W1s / b1s / W2s / b2s of step s+1 are in place; next step's rows are staged
        ME_STAMP(6);
    }
#ifdef TH_PROFILE
    if (wg == 0 && t == 0)
        for (int i = 0; i < 8; ++i) g_me_prof[i] = prof_acc[i];
#endif

    // ---- write the state back ----
    for (int i = t; i < ME_F * (K / 4); i += ME_T) {
        const int f = i / (K / 4), kq = (i % (K / 4)) * 4;
        *reinterpret_cast<float4 *>(a.w1 + (long)(f0 + f) * K + kq) = *reinterpret_cast<const float4 *>(W1s + f * LD1 + kq);
    }
#pragma unroll
    for (int q = 0; q < TQ; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ct = wave + ME_NW * q;
            if (ct < n_ct) {
                const long idx = (long)(f0 + 4 * g4 + i) * K + ct * 16 + r16;
                a.m_w1[idx] = m1[q][i];
                a.v_w1[idx] = v1[q][i];
            }
        }
    if (t < ME_F) { a.b1[f0 + t] = b1s[t]; a.m_b1[f0 + t] = mb1; a.v_b1[f0 + t] = vb1; }
    if (wg == 0) {
        for (int i = t; i < C * H; i += ME_T) a.w2[i] = W2s[(i / H) * LDH + i % H];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cls = 4 * g4 + i, ht = wave + ME_NW * j;
                if (ht < n_ht && cls < C) {
                    const long idx = (long)cls * H + ht * 16 + r16;
                    a.m_w2[idx] = m2[j][i];
                    a.v_w2[idx] = v2[j][i];
                }
            }
        if (t < C) { a.b2[t] = b2s[t]; a.m_b2[t] = mb2; a.v_b2[t] = vb2; }
        if (t == 0) {
            a.t_state[0] = adam_t;
            if (a.state) {
                a.state[0] += a.n_steps;
                a.state[1] += min((long)a.n_steps * a.batch, a.n_indices - a.first_pos);
            }
        }
    }
}

static size_t mlp_epoch_lds_bytes(int in_f, int hid, int xt_tiles) {
    const size_t floats = (size_t)ME_F * (in_f + 4) + (size_t)ME_B * (hid + 4) + (size_t)ME_CMAX * (hid + 4) + 2 * ME_B * ME_CMAX +
                          ME_B * 17 + 4 * 64 * 4 + 64 + ME_NW * ME_CMAX + 32 + (size_t)ME_B * (xt_tiles * 16 + 4);
    return floats * sizeof(float) + 2 * ME_B * sizeof(int) + 2 * ME_B * sizeof(float) + 16;
}

}  // namespace th

using namespace th;

extern "C" int th_mlp2_train_steps(th_ctx *ctx, const float *d_images, const float *d_labels, const int32_t *d_indices,
                                   int64_t n_indices, int64_t first_pos, int batch, int n_steps, int in_features, int hidden,
                                   int classes, float *d_w1, float *d_b1, float *d_w2, float *d_b2, float *d_m_w1, float *d_v_w1,
                                   float *d_m_b1, float *d_v_b1, float *d_m_w2, float *d_v_w2, float *d_m_b2, float *d_v_b2,
                                   int32_t *d_t, const float *d_lr, float beta1, float beta2, float eps, float weight_decay,
                                   float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int32_t *d_status) {
    TH_REQUIRE(ctx && d_images && d_labels && d_w1 && d_b1 && d_w2 && d_b2 && d_m_w1 && d_v_w1 && d_m_b1 && d_v_b1 && d_m_w2 &&
                   d_v_w2 && d_m_b2 && d_v_b2 && d_t && d_lr,
               "th_mlp2_train_steps: null argument");
    TH_REQUIRE(batch >= 1 && batch <= ME_B && classes >= 1 && classes <= ME_CMAX && hidden >= 16 && hidden <= ME_HMAX &&
                   hidden % 16 == 0 && in_features >= 16 && in_features <= ME_KMAX && in_features % 16 == 0,
               "th_mlp2_train_steps: needs batch <= 64, classes <= 16, hidden a multiple of 16 <= 256, in_features a multiple of "
               "16 <= 1024 (got %d, %d, %d, %d)", batch, classes, hidden, in_features);
    TH_REQUIRE(n_steps >= 0 && first_pos >= 0 && (n_steps == 0 || first_pos + (int64_t)(n_steps - 1) * batch < n_indices),
               "th_mlp2_train_steps: the steps run past the index list");
    TH_REQUIRE(!d_metrics || (d_state && metrics_capacity > 0), "th_mlp2_train_steps: metrics need d_state and a capacity");
    TH_REQUIRE(((uintptr_t)d_images & 15) == 0 && ((uintptr_t)d_w1 & 15) == 0, "th_mlp2_train_steps: images / w1 must be 16-byte aligned");
    if (n_steps == 0) return 0;
    const int n_wg = hidden / ME_F;
    // granule buffers + timeout word, zeroed before EVERY launch (the tags restart at 1)
    const size_t gran_bytes = (size_t)2 * n_wg * ME_B * ME_F * sizeof(u64);
    void *ws = nullptr;
    if (th_malloc(ctx, gran_bytes + 16, &ws)) return 1;
    TH_HIP(hipMemsetAsync(ws, 0, gran_bytes + 16, ctx->stream));
    MlpEpochArgs a{d_images, d_labels, d_indices, (long)n_indices, (long)first_pos, batch, n_steps, in_features, hidden, classes,
                   d_w1, d_b1, d_w2, d_b2, d_m_w1, d_v_w1, d_m_b1, d_v_b1, d_m_w2, d_v_w2, d_m_b2, d_v_b2, d_t, d_lr, beta1, beta2,
                   eps, weight_decay, d_metrics, (long)metrics_capacity, (long *)d_state, 1, (u64 *)ws,
                   (unsigned *)((char *)ws + gran_bytes), d_status};
    // X chunk of the backward slab: as many column tiles as the LDS left by the resident state holds (<= all of them)
    const int n_ct = in_features / 16;
    int xt_tiles = n_ct;
    while (xt_tiles > 1 && mlp_epoch_lds_bytes(in_features, hidden, xt_tiles) > (160u << 10)) --xt_tiles;
    a.xt_tiles = xt_tiles;
    const size_t lds = mlp_epoch_lds_bytes(in_features, hidden, xt_tiles);
    TH_REQUIRE(lds <= (160u << 10), "th_mlp2_train_steps: state does not fit LDS (%zu bytes)", lds);
    const int xs = (n_ct + 1) / 2, tq = (n_ct + ME_NW - 1) / ME_NW;    // k-steps per wave (forward) / column tiles per wave (backward)
#define TH_ME_LAUNCH(XSV, TQV)                                                                                             \
    {                                                                                                                      \
        auto kern = mlp2_epoch_kernel<XSV, TQV>;                                                                           \
        TH_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(ME_T), lds, ctx->stream, a);                                             \
    }
    (void)tq;
    if (xs <= 8) TH_ME_LAUNCH(8, 2) else if (xs <= 16) TH_ME_LAUNCH(16, 4) else if (xs <= 25) TH_ME_LAUNCH(25, 7) else TH_ME_LAUNCH(32, 8)
#undef TH_ME_LAUNCH
    TH_LAUNCH_CHECK();
    return th_free(ctx, ws);   // stream-ordered: the block stays the launch's until it has run
}

#ifdef TH_PROFILE
extern "C" int th_debug_mlp_epoch_prof(th_ctx *ctx, long long *h_out8) {
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpyFromSymbol(h_out8, HIP_SYMBOL(th::g_me_prof), 8 * sizeof(long long)));
    return 0;
}
#endif

// mlp_steps.hip -- EXPERIMENTAL (parity-tested in tests/test_gpu_mlp_steps.py, measured, not used by the Trainer: see "Where it stands" below).
// th_mlp2_steps: MANY training steps of the two-layer MLP (Linear + ReLU, Linear, softmax cross-entropy, Adam:
// examples/train_mnist.rs / train.rs:98-144 with the 784-128-10 model of BASELINE configs[1]) in ONE persistent launch.
//
// Why: as two launches per step the batch-64 step is two kernel boundaries (1.6 us each) plus two chains of dependent round trips: 11.9 us,
// whatever the kernels do.  A hand-off INSIDE a launch costs more than the boundary when producers and consumers sit on
// different XCDs (every agent-scope fence is an L2 write-back / invalidate there: experiments/handoff.hip) -- but workgroups of ONE XCD share
// their L2: a store that has been acknowledged (s_waitcnt vmcnt(0)) is visible to every CU of that XCD that does not hit a stale line in its
// own L1, i.e. after an L1 invalidate (the acquire half of an agent-scope fence: buffer_inv sc1).  experiments/xcd_barrier.hip: two such
// barriers + a 32 KB exchange between 32 workgroups take 3.1 us.
//
// So: launch 8 N workgroups; workgroup b runs on XCD b % 8 (checked: XCC_ID), the N of XCD 0 stay, the others exit at once.  N = (batch / 16)
// x (hidden / 16) <= 32 workgroups of 1024 threads, one per CU.  Per step:
//   phase A  workgroup (row tile, hidden tile): its 16 x 16 tile of H = relu(X W1^T + b1), K split over the 16 waves (nn.rs:54-60,
//            activation.rs:10-12); workgroup 0 first applies the PREVIOUS step's Adam update of W2 / b2 (every workgroup read them in that
//            step's phase B)                                                                                  -- barrier --
//   phase B  workgroup (hidden tile, column group): logits / softmax / dlogits / the dZ1 tile of its hidden tile for all rows (recomputed per
//            workgroup, in registers, like th_mlp_tail), then its share of the dW1 = dZ1^T X tiles with Adam (optim.rs:99-110) in the
//            epilogue; column group 0 also db1 + Adam(b1); workgroup 0 also dW2, db2, the loss, the hit count, the step log   -- barrier --
// Barrier = arrival counter in the L2 (relaxed agent-scope atomics), wall-clock bounded spin (never hangs the GPU: a time-out raises the
// error word and every workgroup leaves), then the L1 invalidate.
//
// Where it stands (tools/prof_mlp_steps.sh, batch 64, 784-128-10): correct on the first run, no hangs; 86 us per step with the agent-scope acquire
// fence after each barrier (buffer_inv sc1 also walks the L2: ~31 us per barrier), 29.8 us with buffer_inv sc0 (this CU's L1 only), 16.0 us with
// the head's gradients spread over the column-group-0 workgroups instead of workgroup 0, 14.2 us with the per-lane addresses kept from
// being hoisted out of the step loop (69 -> 22 spilled registers at 16 waves x 128 VGPRs), 15.7 us as 8 waves x 256 VGPRs (no spills) with the
// next step's X rows pulled into the L2 during phase B and one dW1 tile's operands requested ahead (per-workgroup stamps of a step: phase A 3.1 us,
// 4.4 us where a W2 tile is updated; barrier 1.2; phase B 6.7; barrier 1.9).  The two launches per
// step it would replace take 11.9 us: every phase here is still a chain of dependent L2 round trips on a quarter of the waves the launches
// spread the same work over, and a barrier with its skew (1.2 - 1.9 us) costs what the kernel boundary (1.6 us) it was meant to beat costs.
#include "tail_dev.h"

namespace th {

constexpr int MS_NW = 8;                    // waves per workgroup
constexpr long long MS_SPIN_TICKS = 400000; // 4 ms at 100 MHz

#ifdef TH_PROFILE
__device__ long long g_ms_wg[32][4];     // every workgroup, last step: phase A start, arrival at barrier A, release from it, arrival at barrier B
__device__ long long g_ms_prof[2][16];   // wall clock (100 MHz) at the phase boundaries of the LAST step: workgroup 0, workgroup 5
#define MS_STAMP(i) do { if (threadIdx.x == 0 && s == a.steps - 1) { if (me == 0 || me == 5) g_ms_prof[me == 0 ? 0 : 1][i] = wall_clock64(); if ((i) == 1 || (i) == 2 || (i) == 3 || (i) == 6) g_ms_wg[me][(i) == 6 ? 3 : (i) - 1] = wall_clock64(); } } while (0)
#else
#define MS_STAMP(i) do { } while (0)
#endif

struct Mlp2StepsArgs {
    const float *x, *targets;               // [steps][B][in], [steps][B]
    int steps, batch, in_f, c;
    AdamDev w1, b1, w2, b2;                  // p = the parameter; t = the step counter BEFORE this launch (it adds `steps`)
    float *h;                                // workspace [B][hid]
    float *dw2, *db2;                        // workspace [c][hid], [c]: gradients of the head, applied by workgroup 0 one phase later
    float *loss;                             // [1]: the last step's loss
    float *metrics;
    int64_t capacity;
    int64_t *state;
    int64_t advance;
    unsigned *sync;                          // [64] zeroed before the launch: [0] barrier arrivals
    int *err;                                // host-visible: 1 barrier time-out, 2 a workgroup is not on XCD 0
};

__device__ __forceinline__ unsigned ms_xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); }   // HW_REG_XCC_ID[3:0]

// all stores of this workgroup are in the L2, everybody has arrived, stale L1 lines are gone
__device__ __forceinline__ bool ms_grid_barrier(unsigned *ctr, unsigned target, int *err) {
    __shared__ int bad;
    __syncthreads();                         // (workgroup-scope release: s_waitcnt vmcnt(0) -- the stores are acknowledged by the L2)
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        int b = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            // (ctr[32]: the launch's abort word in device memory -- the host-visible error word is only ever written, never polled)
            const bool aborted = __hip_atomic_load(ctr + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (aborted || wall_clock64() - t0 > MS_SPIN_TICKS) {
                if (!aborted) {
                    __hip_atomic_store(ctr + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                b = 1;
                break;
            }
        }
        bad = b;
    }
    __syncthreads();
#ifdef MS_INV_SC1
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // buffer_inv sc1: L1 AND the L2's non-local lines -- measured ~31 us per barrier
#else
    asm volatile("buffer_inv sc0" ::: "memory");           // this CU's L1 only: the L2 is shared by every workgroup of the launch
#endif
    return bad == 0;
}

template <int KS, int NW>   // hidden = 16 KS; NW waves per workgroup (8: 256 VGPRs per wave -- at 16 waves the 128-register budget spills)
__global__ __launch_bounds__(64 * NW) void mlp2_steps_kernel(Mlp2StepsArgs a) {
    constexpr int HID = 16 * KS, NR = 64 / NW;     // k blocks of 16 per wave in phase A (in_features <= 1024)
    constexpr int MAXT = 1;                       // dW1 tiles per wave whose operands are requested ahead
    if ((blockIdx.x & 7) != 0) return;
    const int me = blockIdx.x >> 3;
    const int RT = a.batch / 16, NWG = RT * KS;
    const int t = threadIdx.x, wave = t >> 6;
    __shared__ float red[NW][64][4];         // cross-wave sums of phase A; phase B: the high-half partial logits
    __shared__ float dzs[64][17];            // phase B: the dZ1 tile [row][hidden col] (batch <= 64 rows)
    __shared__ float dls[64][17];            // column group 0: dlogits [row][class]
    __shared__ float rowv[2][64];            // workgroup 0: the rows' NLL and hits
    __shared__ float dw2s[16][17], db2s[16]; // column group 0: this workgroup's dW2 tile [class][hidden col] (workgroup 0: db2), applied in the next phase A
    if (ms_xcc_id() != 0) {                  // not where the scheme needs it: nothing has been written yet; the others time out of barrier 1 at once
        if (t == 0) {
            __hip_atomic_store(a.sync + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int in_f = a.in_f, B = a.batch, C = a.c;
    const int t0 = a.w1.t[0];                // (nobody writes the counter before the last barrier)
    const float lr = a.w1.lr[0];
    // phase A role: tile (row tile ra, hidden tile ha); phase B role: hidden tile hb, column group cg of RT groups
    const int ra = me / KS, ha = me % KS, hb = me % KS, cg = me / KS;
    const int ntile_n = in_f / 16, per = (ntile_n + RT - 1) / RT, ct0 = cg * per, ct1 = min(ntile_n, ct0 + per);
    unsigned bar = 0;

    for (int s = 0; s < a.steps; ++s) {
        // (per step and phase, opaque to the optimizer: with the lane id loop-invariant every per-lane address of every phase is hoisted out
        // of the step loop and lives through all of them)
        int lane = t & 63;
        asm volatile("" : "+v"(lane));
        int l16 = lane & 15, g4 = lane >> 4;
        const float *xs = a.x + (long)s * B * in_f, *ts = a.targets + (long)s * B;
        const int tcur = t0 + s + 1;         // optim.rs:84
        const float step_sz = adam_step_size(lr, a.w1.beta1, a.w1.beta2, tcur);
        MS_STAMP(0);
        // ---------------- phase A ----------------
        MS_STAMP(1);
        {
            // H tile = relu(X[rows ra] . W1[cols ha]^T + b1): 16-deep k blocks round-robin over the waves, four accumulation chains
            const float *xp = xs + (long)(ra * 16 + l16) * in_f + 4 * g4, *wp = a.w1.p + (long)(ha * 16 + l16) * in_f + 4 * g4;
            floatx4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
            float4 av[NR], bv[NR];
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const int kk = (wave + NW * u) * 16;
                if (kk < in_f) {
                    av[u] = *reinterpret_cast<const float4 *>(xp + kk);
                    bv[u] = *reinterpret_cast<const float4 *>(wp + kk);
                }
            }
            // the PREVIOUS step's update of this workgroup's W2 tile [16 classes][16 hidden columns] and of b2 (every workgroup read W2 / b2 in
            // that step's phase B, so they could not be updated there; the gradients waited in LDS): state requested here, applied after the MFMAs
            const bool upd_w2 = cg == 0 && s > 0 && t < 256 && (t >> 4) < C, upd_b2 = me == 0 && s > 0 && t >= 256 && t < 256 + C;
            const long uix = upd_w2 ? (long)(t >> 4) * HID + hb * 16 + (t & 15) : (upd_b2 ? t - 256 : 0);
            const AdamDev &ua = upd_b2 ? a.b2 : a.w2;
            float up = 0.f, um = 0.f, uv = 0.f;
            if (upd_w2 || upd_b2) {
                up = ua.p[uix];
                um = ua.m[uix];
                uv = ua.v[uix];
            }
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                const int kk = (wave + NW * u) * 16;
                if (kk < in_f) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u].x, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u].y, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u].z, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u].w, c3, 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave][lane][e] = (c0[e] + c1[e]) + (c2[e] + c3[e]);
            if (upd_w2 || upd_b2) {          // optim.rs:99-110 with the previous step's counter
                const float st_prev = adam_step_size(lr, ua.beta1, ua.beta2, tcur - 1);
                const float gv = (upd_b2 ? db2s[t - 256] : dw2s[t >> 4][t & 15]) + ua.wd * up;
                const float mn = ua.beta1 * um + (1.0f - ua.beta1) * gv, vn = ua.beta2 * uv + (1.0f - ua.beta2) * gv * gv;
                ua.m[uix] = mn;
                ua.v[uix] = vn;
                ua.p[uix] = up - st_prev * mn / (sqrtf(vn) + ua.eps);
            }
            __syncthreads();
            if (wave < 4) {                  // wave e finishes element e of every lane's quad: row 4 g4 + e, column l16
                float sum = red[0][lane][wave];
#pragma unroll
                for (int w = 1; w < NW; ++w) sum += red[w][lane][wave];
                const int col = ha * 16 + l16, row = ra * 16 + 4 * g4 + wave;
                float v = sum + a.b1.p[col];
                v = v > 0.f ? v : 0.f;
                a.h[(long)row * HID + col] = v;
            }
        }
        MS_STAMP(2);
        if (!ms_grid_barrier(a.sync, (bar += NWG), a.err)) return;
        MS_STAMP(3);
        // ---------------- phase B ----------------
        asm volatile("" : "+v"(lane));
        l16 = lane & 15, g4 = lane >> 4;
        // (1) logits^T, softmax, dlogits, the dZ1 tile of hidden tile hb for every row block: waves 0 .. RT-1 take the low half of the hidden
        //     dimension of row block `wave`, waves RT .. 2 RT - 1 the high half
        const bool lo_half = wave < RT, hi_half = wave >= RT && wave < 2 * RT;
        floatx4 lgp = {0.f, 0.f, 0.f, 0.f};
        float w2b[4], hm[4], b2v[4], tf = 0.f;
        constexpr int KH = KS / 2;
        float4 wv[KH], hv[KH];
        if (lo_half || hi_half) {            // the critical chain's operands go out first
            const int r0 = (wave % RT) * 16, kb = hi_half ? 16 * KH : 0;
            const float *wp = a.w2.p + (long)min(l16, C - 1) * HID + kb + 4 * g4;     // A: class l16
            const float *hp = a.h + (long)(r0 + l16) * HID + kb + 4 * g4;            // B: row l16
#pragma unroll
            for (int u = 0; u < KH; ++u) {
                wv[u] = *reinterpret_cast<const float4 *>(wp + 16 * u);
                hv[u] = *reinterpret_cast<const float4 *>(hp + 16 * u);
            }
            if (lo_half) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cls = min(4 * g4 + i, C - 1);
                    w2b[i] = a.w2.p[(long)cls * HID + hb * 16 + l16];
                    b2v[i] = a.b2.p[cls];
                    hm[i] = a.h[(long)(r0 + 4 * g4 + i) * HID + hb * 16 + l16];
                }
                tf = ts[r0 + l16];
            }
        }
        // the NEXT step's X rows of this workgroup's phase-A tile, an eighth of them per workgroup of the row tile (they share the L2): pulled in
        // now by the last wave, so that phase A does not start with a cold HBM / TLB round trip
        float4 pf = {0.f, 0.f, 0.f, 0.f};
        if (wave == NW - 1 && s + 1 < a.steps) {
            const float *xn = xs + (long)B * in_f + (long)(ra * 16) * in_f;
            const int c4n = in_f / 4, c4per = (c4n + KS - 1) / KS, c4a = ha * c4per, c4b = min(c4n, c4a + c4per);
            for (int q = lane; q < 16 * (c4b - c4a); q += 64) {
                const int r = q / (c4b - c4a), c4 = c4a + q % (c4b - c4a);
                const float4 v = *reinterpret_cast<const float4 *>(xn + (long)r * in_f + 4 * c4);
                pf.x += v.x; pf.y += v.y; pf.z += v.z; pf.w += v.w;
            }
        }
        // operands that do not depend on this phase's results are requested first, under the logits: the Adam state and the X columns of
        // this wave's first dW1 tiles
        float pv[MAXT][4], mv[MAXT][4], vv[MAXT][4], xv[MAXT][4][4];
#pragma unroll
        for (int j = 0; j < MAXT; ++j) {
            const int ct = ct0 + wave + NW * j;
            const bool on = ct < ct1;
            const long widx = (long)(hb * 16 + 4 * g4) * in_f + (on ? ct : ct0) * 16 + l16;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pv[j][e] = a.w1.p[widx + (long)e * in_f];
                mv[j][e] = a.w1.m[widx + (long)e * in_f];
                vv[j][e] = a.w1.v[widx + (long)e * in_f];
            }
#pragma unroll
            for (int rb4 = 0; rb4 < 4; ++rb4)
#pragma unroll
                for (int i = 0; i < 4; ++i) xv[j][rb4][i] = xs[(long)(min(16 * rb4, B - 16) + 4 * g4 + i) * in_f + (on ? ct : ct0) * 16 + l16];
        }
        if (lo_half || hi_half) {
            floatx4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
            for (int u = 0; u < KH; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].x, hv[u].x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].y, hv[u].y, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].z, hv[u].z, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].w, hv[u].w, c1, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) lgp[i] = c0[i] + c1[i];
            if (hi_half) {
#pragma unroll
                for (int i = 0; i < 4; ++i) red[wave % RT][lane][i] = lgp[i];
            }
        }
        __syncthreads();
        if (lo_half) {
            const int r0 = wave * 16;
            float lg[4], dl[4], nll_row;
            int bi;
#pragma unroll
            for (int i = 0; i < 4; ++i) lg[i] = 4 * g4 + i < C ? (lgp[i] + red[wave][lane][i]) + b2v[i] : -INFINITY;
            tail_row_softmax(lg, g4, C, tf, 1.0f / (float)B, dl, nll_row, bi);
            // dZ1 tile: D[row 4 g4 + e][hidden col l16] = sum_class dl[row][class] W2[class][col], masked by H > 0 (ops.rs:254-265, 358-369)
            floatx4 dh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) dh = __builtin_amdgcn_mfma_f32_16x16x4f32(dl[i], w2b[i], dh, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) dzs[r0 + 4 * g4 + e][l16] = hm[e] > 0.f ? dh[e] : 0.f;
            if (cg == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) dls[r0 + l16][4 * g4 + i] = dl[i];
            }
            if (me == 0 && g4 == 0) {
                rowv[0][r0 + l16] = nll_row;
                rowv[1][r0 + l16] = (fabsf((float)bi - tf) < 1e-6f) ? 1.f : 0.f;   // loss.rs:283
            }
        }
        __syncthreads();
        MS_STAMP(4);
        // (2) the last three waves (they own the fewest dW1 tiles): db1 + Adam(b1) and the dW2 tile (column group 0), the step's numbers (workgroup 0)
        if (cg == 0 && wave == NW - 3) {      // db1 (tensor.rs:686-691): lane (column l16, row group g4), then the four groups
            float sum = 0.f;
            for (int r = g4; r < B; r += 4) sum += dzs[r][l16];
            sum = sum_over_g4(sum);
            if (g4 == 0) adam_update(a.b1.p, a.b1.m, a.b1.v, hb * 16 + l16, sum, step_sz, a.b1.beta1, a.b1.beta2, a.b1.eps, a.b1.wd);
        }
        if (cg == 0 && wave == NW - 2) {
            // dW2 tile [class l16 -> rows 4 g4 + e][hidden col l16] = sum_rows dl[row][class] H[row][col] (ops.rs:280-291): A = dl^T from LDS, B = H
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int rb = 0; rb < B; rb += 16) {
                float av[4], bv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    av[i] = dls[rb + 4 * g4 + i][l16];
                    bv[i] = a.h[(long)(rb + 4 * g4 + i) * HID + hb * 16 + l16];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) dw2s[4 * g4 + e][l16] = acc[e];
        }
        if (me == 0 && wave == NW - 1) {
            if (lane < C) {
                float sum = 0.f;
                for (int r = 0; r < B; ++r) sum += dls[r][lane];
                db2s[lane] = sum;
            }
            if (lane == 32) {
                float n = 0.f, hsum = 0.f;
                for (int r = 0; r < B; ++r) {
                    n += rowv[0][r];
                    hsum += rowv[1][r];
                }
                const float l = n / (float)B;            // loss.rs:164
                a.loss[0] = l;
                if (a.metrics) {                         // the step log of th_log_step
                    const int64_t s0 = a.state[0], s1 = a.state[1];
                    const int64_t slot = s0 < a.capacity ? s0 : s0 % a.capacity;
                    a.metrics[2 * slot] = l;
                    a.metrics[2 * slot + 1] = hsum;
                    a.state[0] = s0 + 1;
                    a.state[1] = s1 + a.advance;
                }
            }
        }
        MS_STAMP(5);
        // (3) dW1 tiles [hidden tile hb][column tiles ct0 .. ct1) with Adam in the epilogue; tile j of a wave: ct0 + wave + NW j
        for (int j = 0, ct = ct0 + wave; ct < ct1; ++j, ct += NW) {
            const long widx = (long)(hb * 16 + 4 * g4) * in_f + ct * 16 + l16;      // D: rows = hidden 4 g4 + e, column l16
            float p4[4], m4[4], v4[4], x4[4][4];
            if (j < MAXT) {
#pragma unroll
                for (int jj = 0; jj < MAXT; ++jj)
                    if (jj == j) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { p4[e] = pv[jj][e]; m4[e] = mv[jj][e]; v4[e] = vv[jj][e]; }
#pragma unroll
                        for (int rb4 = 0; rb4 < 4; ++rb4)
#pragma unroll
                            for (int i = 0; i < 4; ++i) x4[rb4][i] = xv[jj][rb4][i];
                    }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p4[e] = a.w1.p[widx + (long)e * in_f];
                    m4[e] = a.w1.m[widx + (long)e * in_f];
                    v4[e] = a.w1.v[widx + (long)e * in_f];
                }
#pragma unroll
                for (int rb4 = 0; rb4 < 4; ++rb4)
#pragma unroll
                    for (int i = 0; i < 4; ++i) x4[rb4][i] = xs[(long)(min(16 * rb4, B - 16) + 4 * g4 + i) * in_f + ct * 16 + l16];
            }
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rb4 = 0; rb4 < 4; ++rb4) {
                if (16 * rb4 >= B) break;
                float dz[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) dz[i] = dzs[16 * rb4 + 4 * g4 + i][l16];        // A: hidden l16, k = row 4 g4 + i; B: X[row][column l16]
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dz[i], x4[rb4][i], acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gv = acc[e] + a.w1.wd * p4[e];
                const float mn = a.w1.beta1 * m4[e] + (1.0f - a.w1.beta1) * gv;
                const float vn = a.w1.beta2 * v4[e] + (1.0f - a.w1.beta2) * gv * gv;
                a.w1.m[widx + (long)e * in_f] = mn;
                a.w1.v[widx + (long)e * in_f] = vn;
                a.w1.p[widx + (long)e * in_f] = p4[e] - step_sz * mn / (sqrtf(vn) + a.w1.eps);
            }
        }
        if (pf.x + pf.y + pf.z + pf.w == 1.2345e38f) a.loss[0] = pf.x;   // (never: keeps the prefetch loads)
        MS_STAMP(6);
        if (!ms_grid_barrier(a.sync, (bar += NWG), a.err)) return;
        MS_STAMP(7);
    }
    if (cg == 0) {                           // the last step's W2 / b2 (the barrier's __syncthreads made the LDS tiles visible), then the counter
        const float st_last = adam_step_size(lr, a.w2.beta1, a.w2.beta2, t0 + a.steps);
        if (t < 256) {
            const int cls = t >> 4, col = hb * 16 + (t & 15);
            if (cls < C) adam_update(a.w2.p, a.w2.m, a.w2.v, (long)cls * HID + col, dw2s[cls][t & 15], st_last, a.w2.beta1, a.w2.beta2, a.w2.eps, a.w2.wd);
        } else if (me == 0 && t < 256 + C) {
            adam_update(a.b2.p, a.b2.m, a.b2.v, t - 256, db2s[t - 256], st_last, a.b2.beta1, a.b2.beta2, a.b2.eps, a.b2.wd);
        }
        if (me == 0 && t == 0) const_cast<int32_t *>(a.w1.t)[0] = t0 + a.steps;
    }
}

}  // namespace th

using namespace th;

extern "C" {

int th_mlp2_steps_supported(int batch, int in_features, int hidden, int classes) {
    if (!(batch == 16 || batch == 32 || batch == 48 || batch == 64)) return 0;
    if (!(hidden == 32 || hidden == 64 || hidden == 128)) return 0;
    if ((batch / 16) * (hidden / 16) > 32) return 0;          // one workgroup per CU of an XCD
    return in_features > 0 && in_features % 16 == 0 && in_features <= 1024 && classes >= 1 && classes <= 16 ? 1 : 0;
}

int th_mlp2_steps(th_ctx *ctx, const float *d_x, const float *d_targets, int steps, int batch, int in_features, int hidden, int classes,
                  const th_adam_fuse *fuse4, float *d_loss, float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance,
                  int *h_err) {
    TH_REQUIRE(ctx && d_x && d_targets && fuse4 && d_loss && h_err && steps > 0, "th_mlp2_steps: null argument");
    TH_REQUIRE(th_mlp2_steps_supported(batch, in_features, hidden, classes),
               "th_mlp2_steps: needs batch 16 / 32 / 48 / 64, hidden 32 / 64 / 128 with (batch / 16) (hidden / 16) <= 32, in_features %% 16 == 0 (<= 1024), classes <= 16");
    for (int i = 0; i < 4; ++i)
        TH_REQUIRE(fuse4[i].d_p && fuse4[i].d_m && fuse4[i].d_v && fuse4[i].d_t == fuse4[0].d_t && fuse4[i].d_lr, "th_mlp2_steps: four complete Adam slots with one shared counter");
    TH_REQUIRE((((uintptr_t)d_x | (uintptr_t)fuse4[0].d_p | (uintptr_t)fuse4[2].d_p) & 15) == 0, "th_mlp2_steps: x, W1 and W2 must be 16-byte aligned");
    TH_REQUIRE(!d_metrics || (d_state && metrics_capacity > 0), "th_mlp2_steps: metrics need d_state and a capacity");
    const size_t n_h = (size_t)batch * hidden, n_w2 = ((size_t)classes * hidden + 3) & ~(size_t)3;
    void *ws = nullptr;
    if (th_malloc(ctx, (n_h + n_w2 + 16 + 64) * sizeof(float), &ws)) return 1;
    Mlp2StepsArgs a{};
    a.x = d_x; a.targets = d_targets; a.steps = steps; a.batch = batch; a.in_f = in_features; a.c = classes;
    a.w1 = make_adam_dev(&fuse4[0]); a.b1 = make_adam_dev(&fuse4[1]); a.w2 = make_adam_dev(&fuse4[2]); a.b2 = make_adam_dev(&fuse4[3]);
    a.h = (float *)ws; a.dw2 = a.h + n_h; a.db2 = a.dw2 + n_w2; a.sync = reinterpret_cast<unsigned *>(a.db2 + 16);
    a.loss = d_loss; a.metrics = d_metrics; a.capacity = metrics_capacity; a.state = d_state; a.advance = advance; a.err = h_err;
    if (int rc = th_fill_f32(ctx, reinterpret_cast<float *>(a.sync), 0.f, 64)) return rc;
    const int nwg = (batch / 16) * (hidden / 16);
    const dim3 grid(8 * nwg), block(64 * MS_NW);
    if (hidden == 128) hipLaunchKernelGGL((mlp2_steps_kernel<8, MS_NW>), grid, block, 0, ctx->stream, a);
    else if (hidden == 64) hipLaunchKernelGGL((mlp2_steps_kernel<4, MS_NW>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((mlp2_steps_kernel<2, MS_NW>), grid, block, 0, ctx->stream, a);
    TH_LAUNCH_CHECK();
    return th_free(ctx, ws);
}

#ifdef TH_PROFILE
int th_debug_mlp_steps_prof(th_ctx *ctx, long long *h_out32) {
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpyFromSymbol(h_out32, HIP_SYMBOL(th::g_ms_prof), 32 * sizeof(long long)));
    return 0;
}
int th_debug_mlp_steps_wg(th_ctx *ctx, long long *h_out128) {
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpyFromSymbol(h_out128, HIP_SYMBOL(th::g_ms_wg), 128 * sizeof(long long)));
    return 0;
}
#endif

}  // extern "C"

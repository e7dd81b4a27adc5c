// mlp_tail.hip -- th_mlp_tail: the classifier head AND the whole backward of the Linear+ReLU layer
// in front of it, in ONE launch (what th_linear_xent_head followed by th_linear_bwd_adam_ex
// compute: nn.rs:54-60 + loss.rs:101-195,271-290 + the backward closures ops.rs:238-294,
// tensor.rs:574-587,674-694 of both Linear layers and the ReLU node ops.rs:358-369).
//
// Why: at batch 64 the MNIST MLP step is three dependent launches (layer-1 forward, head, layer-1
// backward) and the single-workgroup head alone costs 9 us of dependent round trips.  The head is
// 164 kFLOP: cheap enough that EVERY workgroup of the layer-1 backward recomputes it in registers
// and goes straight on to its dW1 tile -- no dH round trip through memory, no launch boundary.
//
// Each wave owns 16 batch rows (4 waves = one 64-row chunk; larger batches loop) and never
// exchanges data with the other waves before the final tile reduction, because every product is
// arranged so that the MFMA C/D layout of one step IS the operand layout of the next:
//   logits^T[class][row] = W2 . H^T          A = W2[class=l&15][k], B = H[row=l&15][k]
//                                            -> lane (r,g) holds logit[row r][class 4g+i], i = 0..3
//   softmax / NLL / argmax / dlogits         over i in the lane and over g by xor-shuffles 16, 32
//   dH[row][hid] = dl . W2                   A = dl[row r][class 4g+s] (= register s), B = W2[class 4g+s][hid r]
//                                            -> lane (r,g) holds dH[row 4g+i][hid r]; ReLU mask H[row 4g+i][hid r]
//   dW1[hid][in] += dH^T . X                 A = dH[row 4g+s][hid r] (= register s), B = X[row 4g+s][in r]
// Workgroup roles by block index:
//   [0, n_head)      "head" workgroups, one per 16 hidden units hm: db1[16 hm ..] (+ fused Adam of b1),
//                    dW2[:, 16 hm ..]; hm == 0 also db2, loss, hit count, the step log
//   next n_dw        one 16-hidden x (16*TN)-input block of dW1 (+ fused Adam of W1), XCD-aware order
// W2 / b2 are READ by every workgroup, so their update cannot run here: the caller defers it
// (th_adam_slice) to the next launch that does not read them (th_linear_fwd_ex of the next step).
#include "tail_dev.h"
#include "dp_dev.h"

TH_USES_DEVICE_ERRORS()

namespace th {

struct TailArgs {
    const float *x, *h, *w2, *b2, *targets;
    int batch, in_f, hid, c;
    float *loss, *ncorrect, *dw1, *db1, *dw2, *db2;
    float *metrics;
    int64_t capacity;
    int64_t *state;
    int64_t advance;
    AdamDev w1_adam, b1_adam;
    int n_dw, n_head, tiles_m, groups;   // n_head = tiles_m rounded up to 8 (keeps the XCD phase of the dW blocks)
    const float *w1;                     // dX role (whole-tile kernel only): dX[B,in] = dZ1 . W1
    float *dx;
    int xgroups;                         // 32-column groups of dX; its workgroups: xgroups x ceil(B / 64) behind the dW blocks
    DpDev dp;                            // th_mlp_tail_dp: the gradient exchange across ranks in the epilogues (dp_dev.h)
    int32_t *dp_tick;                    // ... the step counter this step's first launch ticked (taken back when the exchange fails)
};

#ifdef TH_PROFILE
__device__ long long g_tail_prof[2][16];   // [0] lead head workgroup, [1] first dW1 workgroup
#define TAIL_STAMP(i)                                                                         \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == a.n_head))                  \
            g_tail_prof[blockIdx.x == 0 ? 0 : 1][i] = wall_clock64();                         \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
__global__ void tail_prof_mark_kernel(int i) { g_tail_prof[0][i] = g_tail_prof[1][i] = wall_clock64(); }
#else
#define TAIL_STAMP(i) do { } while (0)
#endif

template <int KS, int TN>
__global__ __launch_bounds__(256) void mlp_tail_kernel(TailArgs a) {
    __shared__ float red[4][TN][64][4];   // cross-wave sums of the dW1 block (head role: slot 0 = dW2 tile)
    __shared__ float tr[4][16][17];       // head role: per-wave transpose of dlogits
    __shared__ float sc[4][36];           // head role: per-wave db1[16], db2[16], nll, hits
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r16 = lane & 15, g4 = lane >> 4;
    const int bid = blockIdx.x;
    TAIL_STAMP(0);
    const bool head_role = bid < a.n_head;   // dispatched first: they carry the most work per workgroup
    int tile_m, grp = 0;
    if (!head_role) {   // block b runs on XCD b % 8: XCD x takes the x-th eighth of the blocks, hidden tile innermost
        const int b2 = bid - a.n_head;
        const int tt = (b2 & 7) * (a.n_dw >> 3) + (b2 >> 3);
        if (tt >= a.tiles_m * a.groups) return;
        tile_m = tt % a.tiles_m;
        grp = tt / a.tiles_m;
    } else {
        tile_m = bid;
        if (tile_m >= a.tiles_m) return;
    }
    const int hid = a.hid, C = a.c, in_f = a.in_f, B = a.batch;
    const int hcol = tile_m * 16 + r16;          // this lane's hidden unit (B-operand / C-column position)
    const bool hcol_ok = hcol < hid;
    const int col0 = grp * 16 * TN;
    const bool lead = head_role && tile_m == 0;

    // ---- everything is requested up front in ONE global round trip, branch-free: a guarded load reads a
    //      clamped (always valid) address and the value is zeroed afterwards -- per-load exec-mask branches
    //      cost ~1 us of scalar overhead here, and a wait between two groups of loads costs a second round trip ----
    // the counters / learning rates of the fused updates: scalar loads, requested first (plain loads: nobody
    // writes the counter in this launch); the step sizes are formed while the vector loads are in flight
    const bool any_w = a.w1_adam.p != nullptr, any_b = a.db1 && a.b1_adam.p != nullptr;
    const int32_t w_t = any_w ? sload(a.w1_adam.t) : 1, b_t = any_b ? sload(a.b1_adam.t) : 1;
    const float w_lr = any_w ? sload(a.w1_adam.lr) : 0.f, b_lr = any_b ? sload(a.b1_adam.lr) : 0.f;
    const int Cm1 = C - 1, hid4 = hid - 4, hcol_c = min(hcol, hid - 1);
    float4 wv[KS];                                // W2[class r16][16 ks + 4 g4 ..]
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = ks * 16 + g4 * 4;
        wv[ks] = *reinterpret_cast<const float4 *>(a.w2 + (long)min(r16, Cm1) * hid + min(k, hid4));
    }
    float w2b[4], b2v[4];                         // W2[class 4 g4 + s][hcol], b2[class 4 g4 + i]
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int cls = min(g4 * 4 + s, Cm1);
        w2b[s] = a.w2[(long)cls * hid + hcol_c];
        b2v[s] = a.b2 ? a.b2[cls] : 0.f;
    }
    // epilogue operands: wave e finishes element e of every lane's C/D quad (row 4 g4 + e of the tile)
    const int erow = tile_m * 16 + g4 * 4 + wave;
    long e_ix[TN];
    bool e_ok[TN];
    float e_p[TN], e_m[TN], e_v[TN], w_step = 0.f;
    const bool fuse_w = !head_role && a.w1_adam.p != nullptr;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int ecol = col0 + tn * 16 + r16;
        e_ok[tn] = !head_role && erow < hid && ecol < in_f;
        e_ix[tn] = e_ok[tn] ? (long)erow * in_f + ecol : 0;
        e_p[tn] = e_m[tn] = e_v[tn] = 0.f;
        if (fuse_w) {
            e_p[tn] = a.w1_adam.p[e_ix[tn]];
            e_m[tn] = a.w1_adam.m[e_ix[tn]];
            e_v[tn] = a.w1_adam.v[e_ix[tn]];
        }
    }
    const bool own_b1 = head_role && a.db1 && t < 16 && tile_m * 16 + t < hid;
    const bool fuse_b1 = head_role && a.db1 && a.b1_adam.p != nullptr;
    const int b1_ix = own_b1 ? tile_m * 16 + t : 0;
    float bp_ = 0.f, bm_ = 0.f, bv_ = 0.f, b_step = 0.f;
    if (fuse_b1) {
        bp_ = a.b1_adam.p[b1_ix];
        bm_ = a.b1_adam.m[b1_ix];
        bv_ = a.b1_adam.v[b1_ix];
    }
    const bool fuse_b = own_b1 && fuse_b1;
    const int64_t state0 = (lead && a.metrics) ? sload(a.state) : 0, state1 = (lead && a.metrics) ? sload(a.state + 1) : 0;

    floatx4 accdw[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) accdw[tn] = floatx4{0.f, 0.f, 0.f, 0.f};
    floatx4 acc_dw2 = {0.f, 0.f, 0.f, 0.f};
    float db1_acc = 0.f, db2_acc[4] = {0.f, 0.f, 0.f, 0.f}, nll_acc = 0.f, hit_acc = 0.f;
    const float inv_b = 1.0f / (float)B;

    for (int c0 = 0; c0 < B; c0 += 64) {
        const int r0 = c0 + wave * 16;            // this wave's 16 rows
        const int row_a = r0 + r16;               // row in the lane's "r16" position
        const bool row_a_ok = row_a < B;
        const int row_ac = min(row_a, B - 1);
        // ---- this chunk's operands, all requested before the first MFMA ----
        float4 hv[KS];                            // H[row_a][16 ks + 4 g4 ..]
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = ks * 16 + g4 * 4;
            hv[ks] = *reinterpret_cast<const float4 *>(a.h + (long)row_ac * hid + min(k, hid4));
        }
        const float tf_raw = a.targets[row_ac];
        float hm[4], xv[TN][4];                   // H[r0 + 4 g4 + s][hcol] (ReLU mask / dW2 operand), X[same row][col0 + 16 tn + r16]
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row_c = min(r0 + g4 * 4 + s, B - 1);
            hm[s] = a.h[(long)row_c * hid + hcol_c];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) xv[tn][s] = head_role ? 0.f : a.x[(long)row_c * in_f + min(col0 + tn * 16 + r16, in_f - 1)];
        }
        TAIL_STAMP(1);
        if (c0 == 0) {
            // step sizes of the fused updates (optim.rs:87-90): uniform ALU work that runs while the vector loads are in flight
            if (fuse_w) w_step = adam_step_size(w_lr, a.w1_adam.beta1, a.w1_adam.beta2, w_t);
            if (fuse_b1) b_step = adam_step_size(b_lr, a.b1_adam.beta1, a.b1_adam.beta2, b_t);
        }
        TAIL_STAMP(2);
        // ---- zero what lies outside the problem ----
        const float tf = row_a_ok ? tf_raw : 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bool k_ok = ks * 16 + g4 * 4 < hid;
            if (c0 == 0 && !(r16 < C && k_ok)) wv[ks] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(row_a_ok && k_ok)) hv[ks] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool row_ok = r0 + g4 * 4 + s < B;
            if (c0 == 0) {
                if (!(g4 * 4 + s < C && hcol_ok)) w2b[s] = 0.f;
                if (!(g4 * 4 + s < C)) b2v[s] = 0.f;
            }
            if (!(row_ok && hcol_ok)) hm[s] = 0.f;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                xv[tn][s] = (!head_role && row_ok && col0 + tn * 16 + r16 < in_f) ? xv[tn][s] : 0.f;
        }

        TAIL_STAMP(3);
        // ---- logits^T (nn.rs:54-60): four independent accumulation chains, one per float4 component ----
        floatx4 ax = {0.f, 0.f, 0.f, 0.f}, ay = ax, az = ax, aw = ax;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            ax = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ks].x, hv[ks].x, ax, 0, 0, 0);
            ay = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ks].y, hv[ks].y, ay, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ks].z, hv[ks].z, az, 0, 0, 0);
            aw = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ks].w, hv[ks].w, aw, 0, 0, 0);
        }
        // lane (r16, g4): logit[row_a][class 4 g4 + i]
        float lg[4], dl[4], nll_row;
        int bi;
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[i] = (g4 * 4 + i < C) ? ((ax[i] + ay[i]) + (az[i] + aw[i])) + b2v[i] : -INFINITY;
        tail_row_softmax(lg, g4, C, tf, inv_b, dl, nll_row, bi);
#pragma unroll
        for (int i = 0; i < 4; ++i) dl[i] = row_a_ok ? dl[i] : 0.f;

        TAIL_STAMP(4);
        // ---- dH tile (ops.rs:254-265) and the ReLU mask of the hidden layer (ops.rs:358-369, Q15) ----
        floatx4 dh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) dh = __builtin_amdgcn_mfma_f32_16x16x4f32(dl[s], w2b[s], dh, 0, 0, 0);
        float dhm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dhm[i] = hm[i] > 0.f ? dh[i] : 0.f;

        if (!head_role) {
            // ---- dW1 block (ops.rs:280-291 through the W^T node, tensor.rs:574-587) ----
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    accdw[tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(dhm[s], xv[tn][s], accdw[tn], 0, 0, 0);
        } else {
            // ---- db1 (tensor.rs:686-691): rows 4 g4 + i in the lane, then over g4 ----
            float cs = (dhm[0] + dhm[1]) + (dhm[2] + dhm[3]);
            cs += __shfl_xor(cs, 16, 64);
            cs += __shfl_xor(cs, 32, 64);
            db1_acc += cs;
            // ---- dW2 tile: A = dl[row 4 g4 + s][class r16] (wave-local transpose), B = H[row 4 g4 + s][hcol] ----
            if (c0 > 0) __syncthreads();          // previous chunk's readers of tr are done
#pragma unroll
            for (int i = 0; i < 4; ++i) tr[wave][r16][g4 * 4 + i] = dl[i];
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc_dw2 = __builtin_amdgcn_mfma_f32_16x16x4f32(tr[wave][g4 * 4 + s][r16], hm[s], acc_dw2, 0, 0, 0);
            if (lead) {
                // db2 (column sums of dlogits), loss and hits: over the 16 rows (r16) by xor-shuffles
                float nl = (row_a_ok && g4 == 0) ? nll_row : 0.f;
                float ht = (row_a_ok && g4 == 0 && fabsf((float)bi - tf) < 1e-6f) ? 1.f : 0.f;   // loss.rs:283
                float d4[4] = {dl[0], dl[1], dl[2], dl[3]};
#pragma unroll
                for (int off = 1; off <= 8; off <<= 1) {
                    nl += __shfl_xor(nl, off, 64);
                    ht += __shfl_xor(ht, off, 64);
#pragma unroll
                    for (int i = 0; i < 4; ++i) d4[i] += __shfl_xor(d4[i], off, 64);
                }
                nll_acc += nl;
                hit_acc += ht;
#pragma unroll
                for (int i = 0; i < 4; ++i) db2_acc[i] += d4[i];
            }
        }
    }

    TAIL_STAMP(5);
    if (!head_role) {
        // ---- deterministic cross-wave sum; wave e finishes element e (+ fused Adam, optim.rs:99-110) ----
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wave][tn][lane][i] = accdw[tn][i];
        __syncthreads();
        TAIL_STAMP(6);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            if (!e_ok[tn]) continue;
            const float out = ((red[0][tn][lane][wave] + red[1][tn][lane][wave]) + red[2][tn][lane][wave]) + red[3][tn][lane][wave];
            a.dw1[e_ix[tn]] = out;
            if (fuse_w) {
                const AdamDev &ad = a.w1_adam;
                const float gv = out + ad.wd * e_p[tn];
                const float mn = ad.beta1 * e_m[tn] + (1.0f - ad.beta1) * gv;
                const float vn = ad.beta2 * e_v[tn] + (1.0f - ad.beta2) * gv * gv;
                ad.m[e_ix[tn]] = mn;
                ad.v[e_ix[tn]] = vn;
                ad.p[e_ix[tn]] = e_p[tn] - w_step * mn / (sqrtf(vn) + ad.eps);
            }
        }
        TAIL_STAMP(7);
        return;
    }

    // ---- head role: cross-wave sums in wave order ----
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][0][lane][i] = acc_dw2[i];
    if (g4 == 0) sc[wave][r16] = db1_acc;
    if (lead && r16 == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sc[wave][16 + g4 * 4 + i] = db2_acc[i];
    }
    if (lead && lane == 0) {
        sc[wave][32] = nll_acc;
        sc[wave][33] = hit_acc;
    }
    __syncthreads();
    TAIL_STAMP(6);
    if (a.dw2) {   // wave e: class 4 g4 + e
        const int cls = g4 * 4 + wave;
        if (cls < C && hcol_ok)
            a.dw2[(long)cls * hid + hcol] = ((red[0][0][lane][wave] + red[1][0][lane][wave]) + red[2][0][lane][wave]) + red[3][0][lane][wave];
    }
    if (own_b1) {
        const float out = ((sc[0][t] + sc[1][t]) + sc[2][t]) + sc[3][t];
        const int ix = tile_m * 16 + t;
        a.db1[ix] = out;
        if (fuse_b) {
            const AdamDev &ad = a.b1_adam;
            const float gv = out + ad.wd * bp_;
            const float mn = ad.beta1 * bm_ + (1.0f - ad.beta1) * gv;
            const float vn = ad.beta2 * bv_ + (1.0f - ad.beta2) * gv * gv;
            ad.m[ix] = mn;
            ad.v[ix] = vn;
            ad.p[ix] = bp_ - b_step * mn / (sqrtf(vn) + ad.eps);
        }
    }
    if (lead) {
        if (a.db2 && t < C) a.db2[t] = ((sc[0][16 + t] + sc[1][16 + t]) + sc[2][16 + t]) + sc[3][16 + t];
        if (t == 0) {
            const float n = ((sc[0][32] + sc[1][32]) + sc[2][32]) + sc[3][32];
            const float hsum = ((sc[0][33] + sc[1][33]) + sc[2][33]) + sc[3][33];
            const float l = n / (float)B;   // loss.rs:164
            a.loss[0] = l;
            if (a.ncorrect) a.ncorrect[0] = hsum;
            if (a.metrics) {                // the step log of th_log_step
                const int64_t log_slot = state0 < a.capacity ? state0 : state0 % a.capacity;
                a.metrics[2 * log_slot] = l;
                a.metrics[2 * log_slot + 1] = hsum;
                a.state[0] = state0 + 1;
                a.state[1] = state1 + a.advance;
            }
        }
    }
    TAIL_STAMP(7);
}

// ------------------------------------------------------------------------------------------------
// The same launch for whole tiles: hid == 16 KS, batch % 16 == 0, in_features % 16 == 0 (the MNIST MLP).
// A lone wave per SIMD issues roughly one instruction every 5 cycles whatever its kind, so at this size
// the kernel's duration IS its instruction count: no clamps or zero-fills (rows / hidden units / input
// tiles are valid or skipped wave-uniformly; classes >= C are masked once, at the logits), 32-bit byte
// offsets from uniform bases (one VALU op per address, immediates for the rest), cross-lane steps through
// v_permlane{16,32}_swap instead of LDS round trips, branch-free softmax.
// dX role of the whole-tile kernel (ops.rs:254-265 of the hidden layer, for MLPs with more layers in front):
// workgroup (32-column group xg, 64-row chunk) -- each wave owns 16 rows and needs no other wave.  After the
// same logits / softmax, per 16 hidden units th:
//   dH^T[hid][row] = W2^T . dl^T      A = W2[class 4g+s][16 th + r], B = dl[row r][class 4g+s] (= register s)
//                                     -> lane (r,g) holds dH[row r][hid 16 th + 4g + i]; its ReLU mask is the
//                                        H value the lane loaded for the logits (hv[th], component i)
//   dX[row][in] += dHm . W1           A = dHm[row r][hid 16 th + 4g + s] (= register s), B = W1[hid 16 th + 4g + s][in r]
template <int KS, int NW>
__device__ __forceinline__ void tail_dx_role(const TailArgs &a, int rb) {
    constexpr unsigned HID = 16 * KS;
    constexpr int TX = 2, TH_BLK = KS < 8 ? KS : 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g4 = lane >> 4;
    const int C = a.c, B = a.batch;
    const unsigned in_f = (unsigned)a.in_f;
    const int chunk = rb / a.xgroups, xg = rb - chunk * a.xgroups;
    const int r0 = chunk * 16 * NW + wave * 16;
    if (r0 >= B) return;                              // wave-uniform; this role has no barriers
    const unsigned col0 = xg * 16 * TX;
    bool tx_ok[TX];
#pragma unroll
    for (int tx = 0; tx < TX; ++tx) tx_ok[tx] = col0 + tx * 16 < in_f;
    float4 wv[KS], hv[KS];
    const unsigned w_off = ((unsigned)min(r16, C - 1) * HID + g4 * 4) * 4u;
    const unsigned h_off = ((unsigned)(r0 + r16) * HID + g4 * 4) * 4u;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        wv[ks] = ldg4_b(a.w2, w_off + ks * 64);
        hv[ks] = ldg4_b(a.h, h_off + ks * 64);
    }
    float b2v[4];
    unsigned cls_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const unsigned cls = (unsigned)min(g4 * 4 + s, C - 1);
        cls_off[s] = (cls * HID + r16) * 4u;          // W2[class][r16 (+ 16 th)]
        b2v[s] = a.b2 ? ldg_b(a.b2, cls * 4u) : 0.f;
    }
    const float tf = ldg_b(a.targets, (unsigned)(r0 + r16) * 4u);
    const unsigned w1_off = ((unsigned)(g4 * 4) * in_f + col0 + r16) * 4u;   // W1[4 g4 (+ 16 th + s)][col0 + r16 (+ 16 tx)]

    floatx4 ax = {0.f, 0.f, 0.f, 0.f}, ay = ax, az = ax, aw = ax;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        ax = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ks].x, hv[ks].x, ax, 0, 0, 0);
        ay = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ks].y, hv[ks].y, ay, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ks].z, hv[ks].z, az, 0, 0, 0);
        aw = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ks].w, hv[ks].w, aw, 0, 0, 0);
    }
    float lg[4], dl[4], nll_row;
    int bi;
#pragma unroll
    for (int i = 0; i < 4; ++i) lg[i] = (g4 * 4 + i < C) ? ((ax[i] + ay[i]) + (az[i] + aw[i])) + b2v[i] : -INFINITY;
    tail_row_softmax(lg, g4, C, tf, 1.0f / (float)B, dl, nll_row, bi);

    floatx4 accx[TX];
#pragma unroll
    for (int tx = 0; tx < TX; ++tx) accx[tx] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int th0 = 0; th0 < KS; th0 += TH_BLK) {
        float w2t[TH_BLK][4], w1v[TH_BLK][TX][4];
#pragma unroll
        for (int t = 0; t < TH_BLK; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                w2t[t][s] = ldg_b(a.w2, cls_off[s] + (th0 + t) * 64);
#pragma unroll
                for (int tx = 0; tx < TX; ++tx)
                    w1v[t][tx][s] = tx_ok[tx] ? ldg_b(a.w1, w1_off + ((th0 + t) * 16 + s) * in_f * 4u + tx * 64) : 0.f;
            }
#pragma unroll
        for (int t = 0; t < TH_BLK; ++t) {
            floatx4 dht = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) dht = __builtin_amdgcn_mfma_f32_16x16x4f32(w2t[t][s], dl[s], dht, 0, 0, 0);
            const float4 hq = hv[th0 + t];
            const float m[4] = {hq.x > 0.f ? dht[0] : 0.f, hq.y > 0.f ? dht[1] : 0.f, hq.z > 0.f ? dht[2] : 0.f, hq.w > 0.f ? dht[3] : 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int tx = 0; tx < TX; ++tx) accx[tx] = __builtin_amdgcn_mfma_f32_16x16x4f32(m[s], w1v[t][tx][s], accx[tx], 0, 0, 0);
        }
    }
    const unsigned x_off = ((unsigned)(r0 + g4 * 4) * in_f + col0 + r16) * 4u;
#pragma unroll
    for (int tx = 0; tx < TX; ++tx) {
        if (!tx_ok[tx]) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float *>(reinterpret_cast<char *>(a.dx) + x_off + i * in_f * 4u + tx * 64) = accx[tx][i];
    }
}

// NW waves per workgroup (4 / 8): the 64-row chunks of a batch are a serial chain per wave (2.7 us each), so batches
// above 64 rows get more waves instead of more iterations; waves 0..3 finish the tile.
// DPNR > 0 (th_mlp_tail_dp): data parallel, DPNR >= the communicator's rank count -- every finished slice (a dW1 block; a head workgroup's
// dW2 tile, db1 and, in the lead, db2) goes through dp_exchange() before it is stored or fed to Adam: what is applied is the MEAN over the
// ranks, formed in rank order.  Loss and hit count stay this rank's own (the host averages the step logs).
template <int KS, int TN, bool HAS_DX, int NW, int DPNR = 0>
__global__ __launch_bounds__(64 * NW) void mlp_tail_exact_kernel(TailArgs a) {
    constexpr unsigned HID = 16 * KS;
    __shared__ float red[NW][TN][64][4];
    __shared__ float tr[NW][16][17];
    __shared__ float rowv[NW][2][16];
    __shared__ float sc[NW][36];
    // ONE batch of scalar loads for the kernel arguments (left alone the compiler sinks each field's s_load into the block
    // that first uses it, and every later batch has to wait out whatever scalar loads are in flight with it)
    asm volatile("" ::"s"(a.x), "s"(a.h), "s"(a.w2), "s"(a.b2), "s"(a.targets), "s"(a.batch), "s"(a.in_f), "s"(a.c), "s"(a.dw1),
                 "s"(a.db1), "s"(a.w1_adam.p), "s"(a.w1_adam.m), "s"(a.w1_adam.v), "s"(a.w1_adam.t), "s"(a.w1_adam.lr),
                 "s"(a.b1_adam.p), "s"(a.b1_adam.t), "s"(a.b1_adam.lr), "s"(a.n_dw), "s"(a.n_head), "s"(a.groups), "s"(a.metrics),
                 "s"(a.state));
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r16 = lane & 15, g4 = lane >> 4;
    const int bid = blockIdx.x;
    TAIL_STAMP(0);
    DpTicket dp_tk{0u, 0u};
    if constexpr (DPNR > 0) dp_tk = dp_begin(a.dp);   // (two scalar loads, looked at when the slice is ready)
    if (HAS_DX && bid >= a.n_head + a.n_dw) {   // (its own instantiation: the code of a role nobody runs still costs instruction fetches)
        tail_dx_role<KS, NW>(a, bid - a.n_head - a.n_dw);
        return;
    }
    const bool head_role = bid < a.n_head;
    int tile_m, grp = 0, tt = 0;
    if (!head_role) {
        const int b2 = bid - a.n_head;
        // XCD x takes the x-th eighth of the blocks, hidden tile innermost: each L2 fetches only its own X columns.
        // (XCD x = hidden tile x for every column group was 0.2 us faster -- the next forward launch reads the same W1 rows
        // on the same XCD -- but every L2 then fetches all of X: 4.9 MB instead of 3.8 MB of fabric traffic per launch.)
        tt = (b2 & 7) * (a.n_dw >> 3) + (b2 >> 3);
        if (tt >= KS * a.groups) return;
        tile_m = tt % KS;
        grp = tt / KS;
    } else {
        tile_m = bid;
        if (tile_m >= KS) return;
    }
    const int C = a.c, B = a.batch;
    const unsigned in_f = (unsigned)a.in_f;
    const unsigned hcol = tile_m * 16 + r16;
    const unsigned col0 = grp * 16 * TN;
    const bool lead = head_role && tile_m == 0;
    const bool any_w = a.w1_adam.p != nullptr, any_b = a.db1 && a.b1_adam.p != nullptr;
    const int32_t w_t = any_w ? sload(a.w1_adam.t) : 1, b_t = any_b ? sload(a.b1_adam.t) : 1;
    const float w_lr = any_w ? sload(a.w1_adam.lr) : 0.f, b_lr = any_b ? sload(a.b1_adam.lr) : 0.f;

    // ---- chunk-independent operands ----
    float4 wv[KS];                                   // W2[class r16][16 ks + 4 g4 ..]; rows >= C: a copy of row C-1, masked at the logits
    const unsigned w_off = ((unsigned)min(r16, C - 1) * HID + g4 * 4) * 4u;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wv[ks] = ldg4_b(a.w2, w_off + ks * 64);
    float w2b[4], b2v[4];                            // W2[class 4 g4 + s][hcol] (classes >= C meet dl == 0), b2
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const unsigned cls = (unsigned)min(g4 * 4 + s, C - 1);
        w2b[s] = ldg_b(a.w2, (cls * HID + hcol) * 4u);
        b2v[s] = a.b2 ? ldg_b(a.b2, cls * 4u) : 0.f;
    }
    bool tn_ok[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) tn_ok[tn] = col0 + tn * 16 < in_f;   // workgroup-uniform
    // epilogue operands: wave e finishes element e of every lane's C/D quad (row 4 g4 + e of the tile)
    const unsigned e_off = ((tile_m * 16 + g4 * 4 + wave) * in_f + col0 + r16) * 4u;
    const bool fuse_w = !head_role && any_w;
    const bool finisher = NW == 4 || wave < 4;      // waves 0..3 finish element `wave` of every C/D quad
    float e_p[TN], e_m[TN], e_v[TN], w_step = 0.f, b_step = 0.f;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        e_p[tn] = e_m[tn] = e_v[tn] = 0.f;
        if (fuse_w && tn_ok[tn] && finisher) {
            e_p[tn] = ldg_b(a.w1_adam.p, e_off + tn * 64);
            e_m[tn] = ldg_b(a.w1_adam.m, e_off + tn * 64);
            e_v[tn] = ldg_b(a.w1_adam.v, e_off + tn * 64);
        }
    }
    const bool fuse_b1 = head_role && any_b;
    const unsigned b1_off = (tile_m * 16 + (t & 15)) * 4u;
    float bp_ = 0.f, bm_ = 0.f, bv_ = 0.f;
    if (fuse_b1) {
        bp_ = ldg_b(a.b1_adam.p, b1_off);
        bm_ = ldg_b(a.b1_adam.m, b1_off);
        bv_ = ldg_b(a.b1_adam.v, b1_off);
    }
    const int64_t state0 = (lead && a.metrics) ? sload(a.state) : 0, state1 = (lead && a.metrics) ? sload(a.state + 1) : 0;

    floatx4 accdw[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) accdw[tn] = floatx4{0.f, 0.f, 0.f, 0.f};
    floatx4 acc_dw2 = {0.f, 0.f, 0.f, 0.f};
    float db1_acc = 0.f, db2_acc = 0.f, nll_acc = 0.f, hit_acc = 0.f;
    const float inv_b = 1.0f / (float)B;

    // Chunk operands.  The first chunk's loads are issued HERE, in the same basic block as the chunk-independent ones
    // above: behind a loop header the register allocator reuses VGPRs that still have loads in flight and parks
    // s_waitcnt vmcnt(n) in front of the H loads -- a second, serialised memory round trip.
    float4 hv[KS];                               // H[r0 + r16][16 ks + 4 g4 ..]
    float tf = 0.f, hm[4], xv[TN][4];            // H[r0 + 4 g4 + s][hcol], X[same row][col0 + 16 tn + r16]
    auto load_chunk = [&](int c0) {
        const int r0 = c0 + wave * 16;
        if (r0 >= B) return;                     // wave-uniform
        const unsigned h_off = ((unsigned)(r0 + r16) * HID + g4 * 4) * 4u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) hv[ks] = ldg4_b(a.h, h_off + ks * 64);
        tf = ldg_b(a.targets, (unsigned)(r0 + r16) * 4u);
        const unsigned hm_off = ((unsigned)(r0 + g4 * 4) * HID + hcol) * 4u;
        const unsigned x_off = ((unsigned)(r0 + g4 * 4) * in_f + col0 + r16) * 4u;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            hm[s] = ldg_b(a.h, hm_off + s * HID * 4);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                xv[tn][s] = (!head_role && tn_ok[tn]) ? ldg_b(a.x, x_off + s * in_f * 4u + tn * 64) : 0.f;
        }
    };
    load_chunk(0);
    TAIL_STAMP(1);
    // step sizes of the fused updates (optim.rs:87-90): ALU work under the loads' latency (every wave: all finish elements)
    if (fuse_w) w_step = adam_step_size(w_lr, a.w1_adam.beta1, a.w1_adam.beta2, w_t);
    if (fuse_b1) b_step = adam_step_size(b_lr, a.b1_adam.beta1, a.b1_adam.beta2, b_t);
    TAIL_STAMP(2);

    for (int c0 = 0;;) {
        const int r0 = c0 + wave * 16;               // this wave's 16 rows: all valid or (wave-uniformly) all absent
        const bool rows_here = r0 < B;
        if (rows_here) {
            TAIL_STAMP(3);
            // ---- logits^T (nn.rs:54-60): four independent accumulation chains, one per float4 component ----
            floatx4 ax = {0.f, 0.f, 0.f, 0.f}, ay = ax, az = ax, aw = ax;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                ax = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ks].x, hv[ks].x, ax, 0, 0, 0);
                ay = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ks].y, hv[ks].y, ay, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ks].z, hv[ks].z, az, 0, 0, 0);
                aw = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[ks].w, hv[ks].w, aw, 0, 0, 0);
            }
            float lg[4], dl[4], nll_row;
            int bi;
#pragma unroll
            for (int i = 0; i < 4; ++i) lg[i] = (g4 * 4 + i < C) ? ((ax[i] + ay[i]) + (az[i] + aw[i])) + b2v[i] : -INFINITY;
            tail_row_softmax(lg, g4, C, tf, inv_b, dl, nll_row, bi);
            TAIL_STAMP(4);

            // ---- dH tile (ops.rs:254-265) and the ReLU mask of the hidden layer (ops.rs:358-369, Q15) ----
            floatx4 dh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) dh = __builtin_amdgcn_mfma_f32_16x16x4f32(dl[s], w2b[s], dh, 0, 0, 0);
            float dhm[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) dhm[i] = hm[i] > 0.f ? dh[i] : 0.f;

            if (!head_role) {
                // ---- dW1 block (ops.rs:280-291 through the W^T node, tensor.rs:574-587) ----
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        accdw[tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(dhm[s], xv[tn][s], accdw[tn], 0, 0, 0);
            } else {
                // ---- db1 (tensor.rs:686-691): rows 4 g4 + i in the lane, then over g4 ----
                db1_acc += sum_over_g4((dhm[0] + dhm[1]) + (dhm[2] + dhm[3]));
                // ---- dW2 tile: A = dl[row 4 g4 + s][class r16] (wave-private LDS transpose), B = H[row 4 g4 + s][hcol] ----
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 4; ++i) tr[wave][r16][g4 * 4 + i] = dl[i];
                if (lead && g4 == 0) {
                    rowv[wave][0][r16] = nll_row;
                    rowv[wave][1][r16] = (fabsf((float)bi - tf) < 1e-6f) ? 1.f : 0.f;   // loss.rs:283
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc_dw2 = __builtin_amdgcn_mfma_f32_16x16x4f32(tr[wave][g4 * 4 + s][r16], hm[s], acc_dw2, 0, 0, 0);
                if (lead) {   // db2 (column sums of dlogits), loss, hits over this wave's 16 rows, in row order
                    float cs = 0.f, nl = 0.f, ht = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        cs += tr[wave][r][r16];
                        nl += rowv[wave][0][r];
                        ht += rowv[wave][1][r];
                    }
                    db2_acc += cs;
                    nll_acc += nl;
                    hit_acc += ht;
                }
            }
        }
        c0 += 16 * NW;
        if (c0 >= B) break;
        load_chunk(c0);
    }

    TAIL_STAMP(5);
    if (!head_role) {
        // ---- deterministic cross-wave sum; wave e finishes element e (+ fused Adam, optim.rs:99-110) ----
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wave][tn][lane][i] = accdw[tn][i];
        __syncthreads();
        TAIL_STAMP(6);
        float outv[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            outv[tn] = 0.f;
            if (!tn_ok[tn] || !finisher) continue;
            float out = red[0][tn][lane][wave];
#pragma unroll
            for (int w = 1; w < NW; ++w) out += red[w][tn][lane][wave];
            outv[tn] = out;
        }
        if constexpr (DPNR > 0) {
            // slice KS + tt of the launch: the same block of dW1 on every rank (finisher thread (wave, lane) holds the same elements everywhere)
            if (!dp_reduce<DPNR, TN>(a.dp, dp_tk, KS + tt, finisher ? t : -1, outv)) return;   // nothing applied; the word is up
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            if (!tn_ok[tn] || !finisher) continue;
            const float out = outv[tn];
            // (data parallel with the update fused: the reduced gradient goes straight into Adam and is never written)
            if (DPNR == 0 || !fuse_w) *reinterpret_cast<float *>(reinterpret_cast<char *>(a.dw1) + e_off + tn * 64) = out;
            if (fuse_w) {
                const AdamDev &ad = a.w1_adam;
                const float gv = out + ad.wd * e_p[tn];
                const float mn = ad.beta1 * e_m[tn] + (1.0f - ad.beta1) * gv;
                const float vn = ad.beta2 * e_v[tn] + (1.0f - ad.beta2) * gv * gv;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(ad.m) + e_off + tn * 64) = mn;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(ad.v) + e_off + tn * 64) = vn;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(ad.p) + e_off + tn * 64) = e_p[tn] - w_step * mn / (sqrtf(vn) + ad.eps);
            }
        }
        TAIL_STAMP(7);
        return;
    }

    // ---- head role: cross-wave sums in wave order ----
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][0][lane][i] = acc_dw2[i];
    if (g4 == 0) sc[wave][r16] = db1_acc;
    if (lead && g4 == 0) sc[wave][16 + r16] = db2_acc;
    if (lead && lane == 0) {
        sc[wave][32] = nll_acc;
        sc[wave][33] = hit_acc;
    }
    __syncthreads();
    TAIL_STAMP(6);
    if constexpr (DPNR > 0) {
        // slice tile_m of the launch, two values per thread of waves 0..3: [0] this thread's dW2 element, [1] db1 (threads 0..15) or, in the
        // lead workgroup, db2 (threads 16..16 + C - 1).  Reduced over the ranks, then exactly the single-GPU epilogue on the means.
        float hv2[2] = {0.f, 0.f};
        const int cls = g4 * 4 + wave;
        if (a.dw2 && wave < 4 && cls < C) {
            float sum = red[0][0][lane][wave];
#pragma unroll
            for (int w = 1; w < NW; ++w) sum += red[w][0][lane][wave];
            hv2[0] = sum;
        }
        if (a.db1 && t < 16) {
            float out = sc[0][t];
#pragma unroll
            for (int w = 1; w < NW; ++w) out += sc[w][t];
            hv2[1] = out;
        } else if (lead && a.db2 && t >= 16 && t < 16 + C) {
            float sum = sc[0][t];
#pragma unroll
            for (int w = 1; w < NW; ++w) sum += sc[w][t];
            hv2[1] = sum;
        }
        const bool okx = dp_reduce<DPNR, 2>(a.dp, dp_tk, tile_m, wave < 4 ? t : -1, hv2);
        if (!okx) {   // a peer's slice never came: nothing is applied, nothing is logged; the lead takes this step's tick back (optim.rs:84)
            // (only when the exchange failed in THIS launch: behind a dead communicator the step's first launch has not ticked)
            if (lead && t == 0 && a.dp_tick && dp_tk.dead == 0u) atomicSub(a.dp_tick, 1);
            return;
        }
        if (a.dw2 && wave < 4 && cls < C) a.dw2[cls * HID + hcol] = hv2[0];
        if (a.db1 && t < 16) {
            const float out = hv2[1];
            const int ix = tile_m * 16 + t;
            a.db1[ix] = out;
            if (fuse_b1) {
                const AdamDev &ad = a.b1_adam;
                const float gv = out + ad.wd * bp_;
                const float mn = ad.beta1 * bm_ + (1.0f - ad.beta1) * gv;
                const float vn = ad.beta2 * bv_ + (1.0f - ad.beta2) * gv * gv;
                ad.m[ix] = mn;
                ad.v[ix] = vn;
                ad.p[ix] = bp_ - b_step * mn / (sqrtf(vn) + ad.eps);
            }
        }
        if (lead && a.db2 && t >= 16 && t < 16 + C) a.db2[t - 16] = hv2[1];
    }
    if (DPNR == 0 && a.dw2 && (NW == 4 || wave < 4)) {   // wave e: class 4 g4 + e
        const int cls = g4 * 4 + wave;
        float sum = red[0][0][lane][wave];
#pragma unroll
        for (int w = 1; w < NW; ++w) sum += red[w][0][lane][wave];
        if (cls < C) a.dw2[cls * HID + hcol] = sum;
    }
    if (DPNR == 0 && a.db1 && t < 16) {
        float out = sc[0][t];
#pragma unroll
        for (int w = 1; w < NW; ++w) out += sc[w][t];
        const int ix = tile_m * 16 + t;
        a.db1[ix] = out;
        if (fuse_b1) {
            const AdamDev &ad = a.b1_adam;
            const float gv = out + ad.wd * bp_;
            const float mn = ad.beta1 * bm_ + (1.0f - ad.beta1) * gv;
            const float vn = ad.beta2 * bv_ + (1.0f - ad.beta2) * gv * gv;
            ad.m[ix] = mn;
            ad.v[ix] = vn;
            ad.p[ix] = bp_ - b_step * mn / (sqrtf(vn) + ad.eps);
        }
    }
    if (lead) {
        if (DPNR == 0 && a.db2 && t < C) {
            float sum = sc[0][16 + t];
#pragma unroll
            for (int w = 1; w < NW; ++w) sum += sc[w][16 + t];
            a.db2[t] = sum;
        }
        if (t == 0) {
            float n = sc[0][32], hsum = sc[0][33];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                n += sc[w][32];
                hsum += sc[w][33];
            }
            const float l = n / (float)B;   // loss.rs:164
            a.loss[0] = l;
            if (a.ncorrect) a.ncorrect[0] = hsum;
            if (a.metrics) {                // the step log of th_log_step
                const int64_t log_slot = state0 < a.capacity ? state0 : state0 % a.capacity;
                a.metrics[2 * log_slot] = l;
                a.metrics[2 * log_slot + 1] = hsum;
                a.state[0] = state0 + 1;
                a.state[1] = state1 + a.advance;
            }
        }
    }
    TAIL_STAMP(7);
}

}  // namespace th

using namespace th;

static bool tail_whole_tiles(int batch, int in_features, int hidden) {
    return (hidden == 32 || hidden == 64 || hidden == 128 || hidden == 256) && batch % 16 == 0 && in_features % 16 == 0;
}

extern "C" int th_mlp_tail_supported(int batch, int in_features, int hidden, int classes, int need_dx) {
    const bool ok = batch > 0 && batch <= 512 && in_features > 0 && hidden > 0 && hidden <= 256 && (hidden % 4) == 0 && classes > 0 &&
                    classes <= 16;
    return ok && (!need_dx || tail_whole_tiles(batch, in_features, hidden));
}

// the data-parallel instances: whole tiles, two input tiles per dW1 workgroup, hidden 64 / 128 (the MNIST MLPs), no dX role
static bool tail_dp_shapes(int batch, int in_features, int hidden, int classes) {
    return th_mlp_tail_supported(batch, in_features, hidden, classes, 0) && tail_whole_tiles(batch, in_features, hidden) &&
           (hidden == 64 || hidden == 128);
}
static int tail_dp_grid(int in_features, int hidden) {
    const int tiles_m = hidden / 16, groups = ceil_div(in_features, 32);
    return ((tiles_m * groups + 7) & ~7) + ((tiles_m + 7) & ~7);
}
template <int KS, int NW>
static const void *tail_dp_instance(int n_ranks) {
    if (n_ranks <= 2) return (const void *)mlp_tail_exact_kernel<KS, 2, false, NW, 2>;
    if (n_ranks <= 4) return (const void *)mlp_tail_exact_kernel<KS, 2, false, NW, 4>;
    return (const void *)mlp_tail_exact_kernel<KS, 2, false, NW, 8>;
}
static const void *tail_dp_kernel(int batch, int hidden, int n_ranks) {
    const bool nw8 = batch > 64;
    if (hidden == 64) return nw8 ? tail_dp_instance<4, 8>(n_ranks) : tail_dp_instance<4, 4>(n_ranks);
    return nw8 ? tail_dp_instance<8, 8>(n_ranks) : tail_dp_instance<8, 4>(n_ranks);
}

extern "C" int th_mlp_tail_dp_supported(const th_comm *comm, th_ctx *ctx, int batch, int in_features, int hidden, int classes) {
    const DpDev *dp = comm_dp_dev(comm);
    if (!dp || !ctx || dp->n_ranks < 2 || !tail_dp_shapes(batch, in_features, hidden, classes)) return 0;
    const int grid = tail_dp_grid(in_features, hidden);
    if (grid > DP_MAX_SLOTS) return 0;
    if (comm_dp_sharing(comm) > 1) {       // ranks on ONE device (a test box): comm_dp_shared_fits (comm.hip) has the two conditions and why
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tail_dp_kernel(batch, hidden, dp->n_ranks), batch > 64 ? 512 : 256, 0) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        if (!comm_dp_shared_fits(comm, grid, per_cu)) return 0;
    }
    return 1;
}

static int mlp_tail_launch(th_comm *comm, th_ctx *ctx, const float *d_x, const float *d_h, const float *d_w2, const float *d_b2,
                           const float *d_targets, int batch, int in_features, int hidden, int classes, float *d_loss,
                           float *d_ncorrect, float *d_dw1, float *d_db1, float *d_dw2, float *d_db2, const float *d_w1, float *d_dx,
                           float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance,
                           const th_adam_fuse *w1_fuse, const th_adam_fuse *b1_fuse, int32_t *d_tick) {
    TH_REQUIRE(ctx && d_x && d_h && d_w2 && d_targets && d_loss && d_dw1, "th_mlp_tail: null argument");
    TH_REQUIRE(th_mlp_tail_supported(batch, in_features, hidden, classes, 0),
               "th_mlp_tail: needs batch <= 512, hidden <= 256 and a multiple of 4, classes <= 16 (got %d, %d, %d)", batch, hidden, classes);
    TH_REQUIRE(!d_dx || (d_w1 && tail_whole_tiles(batch, in_features, hidden)),
               "th_mlp_tail: d_dx needs d_w1 and whole tiles (hidden 32 / 64 / 128 / 256, batch and in_features multiples of 16)");
    TH_REQUIRE(!d_dx || !(w1_fuse && w1_fuse->d_p), "th_mlp_tail: with d_dx the launch reads W1, its update must be deferred (th_adam_slice)");
    TH_REQUIRE((((uintptr_t)d_h | (uintptr_t)d_w2) & 15) == 0, "th_mlp_tail: d_h and d_w2 must be 16-byte aligned");
    TH_REQUIRE(!d_metrics || (d_state && metrics_capacity > 0), "th_mlp_tail: metrics need d_state and a capacity");
    TH_REQUIRE(!(b1_fuse && b1_fuse->d_p) || d_db1, "th_mlp_tail: fused b1 update needs d_db1");
    static const int tn_env = [] {   // measurement probe: TAPER_TAIL_TN = 1 | 2 | 4 input tiles per dW1 workgroup
        const char *e = getenv("TAPER_TAIL_TN");
        const int v = e ? atoi(e) : 2;
        return (v == 1 || v == 4) ? v : 2;
    }();
    const int tn = comm ? 2 : tn_env;
    TailArgs a{};
    a.x = d_x; a.h = d_h; a.w2 = d_w2; a.b2 = d_b2; a.targets = d_targets;
    a.batch = batch; a.in_f = in_features; a.hid = hidden; a.c = classes;
    a.loss = d_loss; a.ncorrect = d_ncorrect; a.dw1 = d_dw1; a.db1 = d_db1; a.dw2 = d_dw2; a.db2 = d_db2;
    a.metrics = d_metrics; a.capacity = metrics_capacity; a.state = d_state; a.advance = advance;
    a.w1_adam = make_adam_dev(w1_fuse);
    a.b1_adam = make_adam_dev(b1_fuse);
    a.tiles_m = ceil_div(hidden, 16);
    a.groups = ceil_div(in_features, 16 * tn);
    a.n_dw = (a.tiles_m * a.groups + 7) & ~7;
    a.n_head = (a.tiles_m + 7) & ~7;
    a.w1 = d_w1;
    a.dx = d_dx;
    a.xgroups = ceil_div(in_features, 32);
    // waves per workgroup of the whole-tile kernel: one 16-row block per wave and pass -- 4 up to 64 rows, 8 above (16 waves need
    // <= 128 VGPRs: spills, and measured no better); TAPER_TAIL_NW = 4 | 8 forces a count (measurement probe)
    static const int nw_env = getenv("TAPER_TAIL_NW") ? atoi(getenv("TAPER_TAIL_NW")) : 0;
    const int nw_eff = (!comm && (nw_env == 4 || nw_env == 8)) ? nw_env : (batch <= 64 ? 4 : 8);
    const int nw = nw_eff;
    const int grid = a.n_dw + a.n_head + (d_dx ? a.xgroups * ceil_div(batch, 16 * nw_eff) : 0);
    // whole tiles everywhere (the MNIST MLP: 784-128-10, batches of 64 / 32): the short-instruction-stream kernel
    const bool exact = tail_whole_tiles(batch, in_features, hidden) &&
                       (d_dx || !(getenv("TAPER_TAIL_GENERAL") && getenv("TAPER_TAIL_GENERAL")[0] == '1'));
    if (comm) {
        const DpDev *dp = comm_dp_dev(comm);
        TH_REQUIRE(dp && th_mlp_tail_dp_supported(comm, ctx, batch, in_features, hidden, classes),
                   "th_mlp_tail_dp: this communicator / shape cannot take the in-launch exchange (th_mlp_tail_dp_supported)");
        TH_REQUIRE(!d_dx && exact && grid == tail_dp_grid(in_features, hidden), "th_mlp_tail_dp: internal: launch shape");
        a.dp = *dp;
        a.dp_tick = d_tick;
        const void *fn = tail_dp_kernel(batch, hidden, dp->n_ranks);
        void *args[] = {&a};
        TH_HIP(hipLaunchKernel(fn, dim3(grid), dim3(64 * nw), args, 0, ctx->stream));
        comm_dp_count_launch(comm);
        return 0;
    }
#define TH_TAIL_LAUNCH(KS, TN)                                                                                        \
    do {                                                                                                              \
        if (exact && d_dx && nw == 8) hipLaunchKernelGGL((mlp_tail_exact_kernel<KS, TN, true, 8>), dim3(grid), dim3(512), 0, ctx->stream, a);     \
        else if (exact && d_dx) hipLaunchKernelGGL((mlp_tail_exact_kernel<KS, TN, true, 4>), dim3(grid), dim3(256), 0, ctx->stream, a);     \
        else if (exact && nw == 8) hipLaunchKernelGGL((mlp_tail_exact_kernel<KS, TN, false, 8>), dim3(grid), dim3(512), 0, ctx->stream, a);     \
        else if (exact) hipLaunchKernelGGL((mlp_tail_exact_kernel<KS, TN, false, 4>), dim3(grid), dim3(256), 0, ctx->stream, a);     \
        else hipLaunchKernelGGL((mlp_tail_kernel<KS, TN>), dim3(grid), dim3(256), 0, ctx->stream, a);                 \
    } while (0)
#define TH_TAIL_KS(TN)                          \
    do {                                        \
        if (exact && hidden == 32) TH_TAIL_LAUNCH(2, TN); \
        else if (hidden <= 64) TH_TAIL_LAUNCH(4, TN);  \
        else if (hidden <= 128) TH_TAIL_LAUNCH(8, TN); \
        else TH_TAIL_LAUNCH(16, TN);                   \
    } while (0)
    if (tn == 1) TH_TAIL_KS(1);
    else if (tn == 4) TH_TAIL_KS(4);
    else TH_TAIL_KS(2);
#undef TH_TAIL_KS
#undef TH_TAIL_LAUNCH
    TH_LAUNCH_CHECK();
    return 0;
}

extern "C" int th_mlp_tail(th_ctx *ctx, const float *d_x, const float *d_h, const float *d_w2, const float *d_b2,
                           const float *d_targets, int batch, int in_features, int hidden, int classes, float *d_loss,
                           float *d_ncorrect, float *d_dw1, float *d_db1, float *d_dw2, float *d_db2, const float *d_w1, float *d_dx,
                           float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance,
                           const th_adam_fuse *w1_fuse, const th_adam_fuse *b1_fuse) {
    return mlp_tail_launch(nullptr, ctx, d_x, d_h, d_w2, d_b2, d_targets, batch, in_features, hidden, classes, d_loss, d_ncorrect, d_dw1, d_db1,
                           d_dw2, d_db2, d_w1, d_dx, d_metrics, metrics_capacity, d_state, advance, w1_fuse, b1_fuse, nullptr);
}

extern "C" int th_mlp_tail_dp(th_comm *comm, th_ctx *ctx, const float *d_x, const float *d_h, const float *d_w2, const float *d_b2,
                              const float *d_targets, int batch, int in_features, int hidden, int classes, float *d_loss,
                              float *d_ncorrect, float *d_dw1, float *d_db1, float *d_dw2, float *d_db2, float *d_metrics,
                              int64_t metrics_capacity, int64_t *d_state, int64_t advance, const th_adam_fuse *w1_fuse,
                              const th_adam_fuse *b1_fuse, int32_t *d_tick) {
    TH_REQUIRE(comm, "th_mlp_tail_dp: null communicator");
    return mlp_tail_launch(comm, ctx, d_x, d_h, d_w2, d_b2, d_targets, batch, in_features, hidden, classes, d_loss, d_ncorrect, d_dw1, d_db1,
                           d_dw2, d_db2, nullptr, nullptr, d_metrics, metrics_capacity, d_state, advance, w1_fuse, b1_fuse, d_tick);
}

#ifdef TH_PROFILE
extern "C" int th_debug_tail_mark(th_ctx *ctx, int i) {
    hipLaunchKernelGGL(tail_prof_mark_kernel, dim3(1), dim3(1), 0, ctx->stream, i);
    return 0;
}
extern "C" int th_debug_tail_prof(th_ctx *ctx, long long *h_out32) {
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpyFromSymbol(h_out32, HIP_SYMBOL(g_tail_prof), 32 * sizeof(long long)));
    return 0;
}
#endif

// head.hip -- fused classifier head: the last Linear layer + softmax
// cross-entropy + the backward products for a unit upstream gradient, in ONE
// workgroup launch (src/nn.rs:54-60 + src/loss.rs:101-195,271-290 + the
// backward closures ops.rs:238-294, tensor.rs:674-694 of that Linear).
//
// Why one workgroup: the MNIST head is 64x10x128 -- 164 kFLOP.  On this chip a
// dependent launch costs ~1.6 us of boundary plus ~1 us per cold global round
// trip, far more than the arithmetic; one launch that touches global memory
// once (H, W in; dH, dW, db, loss out) replaces four.
//
// 1024 threads = 16 waves.  Per 64-row chunk of the batch:
//   stage    H chunk [64][K] and W [16][K] -> LDS (one global round trip, float4)
//   logits   4 row tiles x (4 waves splitting K), v_mfma_f32_16x16x4_f32, partials
//            combined in LDS in a fixed order -> lg[64][16]
//   softmax  16 lanes per row (xor shuffles): first-max argmax, logp, NLL, hit,
//            dlogits = (softmax - onehot)/B -> dl[64][16]
//   dH       [64][K] = dl . W: 4 x K/16 tiles over the 16 waves, K(=classes) = 16 -> 4 MFMAs per tile
//   dW       [16][K] += dl^T . H: K/16 tiles, 64 rows = 16 MFMAs per tile, accumulated
//            in registers across chunks by the wave that owns the tile
//   db       [16] += column sums of dl
// Then: loss / count / step log / Adam tick by thread 0, dW / db out (+ fused
// Adam update of W and b: every read of W in this launch came from the LDS copy).
//
// Above 64 rows the same kernel runs as up to 256 workgroups, each owning a row range (logits,
// softmax, dlogits and dH are row-local); dW / db / {nll, hits} go to per-workgroup slots and
// head_finish_kernel (+ th_colsum for many slots) adds them in slot order: deterministic, no atomics.
#include "adam_dev.h"

TH_USES_DEVICE_ERRORS()

namespace th {

int adam_slice(th_ctx *ctx, const AdamDev &a, const float *d_g, int64_t n);  // optim.hip

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int HEAD_RC = 64;     // rows per chunk
constexpr int HEAD_CMAX = 16;   // classes <= 16
constexpr int HEAD_KMAX = 256;  // in_features <= 256
constexpr int HEAD_T = 1024;    // threads (16 waves)

struct HeadArgs {
    const float *h, *w, *bias, *targets;
    int batch, k, c;
    float *logits, *loss, *ncorrect, *dh, *dw, *db;
    float *metrics;
    int64_t capacity;
    int64_t *state;
    int64_t advance;
    int32_t *adam_tick;
    AdamDev w_adam, b_adam;
    // multi-workgroup mode (batch > 256): workgroup g owns rows [g*rows_per_wg, (g+1)*rows_per_wg) and
    // writes its dW / db / {nll sum, hits} to slot g of the partial buffers; head_finish_kernel adds the
    // slots in order.  rows_per_wg == 0: one workgroup, final results written directly.
    int rows_per_wg;
    float *part_scalar;
    int mask_dh;   // h is a ReLU output: write dH * (H > 0), i.e. the gradient of the PRE-activation (ops.rs:358-369, Q15)
};

__device__ __forceinline__ long head_target_class(float tf) {  // Rust `as usize`: saturating, NaN -> 0
    return (tf >= 0.f) ? (long)fminf(tf, 2147483520.f) : 0;
}

#ifdef TH_PROFILE
__device__ long long g_head_prof[16];
#define HEAD_STAMP(i) do { if (threadIdx.x == 0) g_head_prof[i] = wall_clock64(); } while (0)
#else
#define HEAD_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(HEAD_T) void linear_xent_head_kernel(HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    HEAD_STAMP(0);
    const int K = a.k, C = a.c;
    const int KP = (K + 15) & ~15;          // K padded to the MFMA k-step; the pad columns hold zeros
    const int LD = KP + 4;                  // row pitch: 16-B aligned rows, breaks the 32-bank period
    float *Hs = lds;                         // [64][LD]
    float *Ws = Hs + HEAD_RC * LD;           // [16][LD]   rows >= C are zero
    float *lg = Ws + HEAD_CMAX * LD;         // [64][16]   logits, then dlogits
    float *part = lg + HEAD_RC * HEAD_CMAX;  // [4 tiles][3][64 lanes][4]  k-split partials of the logits
    float *red = part + 4 * 3 * 64 * 4;      // [32] block reduction scratch
    float *dbp = red + 32;                   // [16 waves][16 classes] per-wave column sums of dlogits
    float *dl = dbp + 16 * HEAD_CMAX;        // [64][16]   dlogits (separate from lg: no barrier between read and write)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r16 = lane & 15, g4 = lane >> 4;
    const bool multi = a.rows_per_wg > 0;
    const int row_beg = multi ? blockIdx.x * a.rows_per_wg : 0;
    const int row_end = multi ? min(a.batch, row_beg + a.rows_per_wg) : a.batch;
    if (multi) {   // this workgroup's slots
        if (a.dw) a.dw += (long)blockIdx.x * a.c * a.k;
        if (a.db) a.db += (long)blockIdx.x * a.c;
    }

    // the tick comes first: every fused update of this step (here and in the
    // following backward launches) must see t+1 (optim.rs:84)
    // (it is PUBLISHED by thread 0 at the end of this launch; in here t+1 is formed in registers,
    // so that no thread waits for a second global round trip before its Adam epilogue)
    // Plain accesses: nothing else touches the counter while this launch runs, and an sc1 store would
    // drop the line from L2, turning the next launch's first load into a fabric round trip.
    const int32_t tick_old = (a.adam_tick && !multi) ? a.adam_tick[0] : 0;

    // ---- everything this launch needs from global memory is requested up front: after a kernel
    //      boundary each dependent round trip costs ~1 us, so none may hide behind a barrier ----
    const float bias_v = (a.bias && r16 < C) ? a.bias[r16] : 0.f;               // logits epilogue
    const int64_t log_slot = (t == 0 && a.metrics && !multi) ? a.state[0] % a.capacity : 0;  // step log
    const bool own_dw = a.dw && wave < (KP / 16);
    const bool fuse_w = own_dw && a.w_adam.p, fuse_b = a.db && t < C && a.b_adam.p;
    float wp_[4], wm_[4], wv_[4], w_step = 0.f, bp_ = 0.f, bm_ = 0.f, bv_ = 0.f, b_step = 0.f;
    if (fuse_w) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cls = g4 * 4 + i, col = wave * 16 + r16;
            const long idx = (cls < C && col < K) ? (long)cls * K + col : 0;
            wp_[i] = a.w_adam.p[idx];
            wm_[i] = a.w_adam.m[idx];
            wv_[i] = a.w_adam.v[idx];
        }
    }
    if (fuse_b) {
        bp_ = a.b_adam.p[t];
        bm_ = a.b_adam.m[t];
        bv_ = a.b_adam.v[t];
    }
    // step counters / learning rates of the fused updates: loaded now, bias correction formed after staging
    int32_t w_t = 0, b_t = 0;
    float w_lr = 0.f, b_lr = 0.f;
    if (a.w_adam.p) {
        w_t = (a.w_adam.t == a.adam_tick) ? tick_old + 1 : a.w_adam.t[0];
        w_lr = a.w_adam.lr[0];
    }
    if (a.b_adam.p) {
        b_t = (a.b_adam.t == a.adam_tick) ? tick_old + 1 : a.b_adam.t[0];
        b_lr = a.b_adam.lr[0];
    }
    const bool vec4 = (K & 3) == 0 && (((uintptr_t)a.h | (uintptr_t)a.w) & 15) == 0;

    // ---- stage W (zero-padded to [16][KP]) and, when KP is a power of two (the 128-wide MNIST head),
    //      the first H chunk in the SAME round trip: all loads are issued before any LDS store waits ----
    const int kq = KP / 4;
    const bool pow2 = vec4 && (kq & (kq - 1)) == 0;
    bool h0_staged = false;
    if (pow2) {
        const int lg2 = __ffs(kq) - 1, rstep = HEAD_T >> lg2;     // rows advanced per 1024 threads
        const int kk = (t & (kq - 1)) * 4, rr0 = t >> lg2;
        const int rows0 = min(HEAD_RC, row_end - row_beg);
        float4 wv = make_float4(0.f, 0.f, 0.f, 0.f), hv[4];
        if (rr0 < C && rr0 < HEAD_CMAX && kk < K) wv = *reinterpret_cast<const float4 *>(a.w + rr0 * K + kk);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rr = rr0 + j * rstep;
            hv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rr < rows0 && kk < K) hv[j] = *reinterpret_cast<const float4 *>(a.h + (long)(row_beg + rr) * K + kk);
        }
        if (rr0 < HEAD_CMAX) *reinterpret_cast<float4 *>(Ws + rr0 * LD + kk) = wv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rr = rr0 + j * rstep;
            if (rr < HEAD_RC) *reinterpret_cast<float4 *>(Hs + rr * LD + kk) = hv[j];
        }
        h0_staged = true;
    } else if (vec4) {
        for (int i = t; i < HEAD_CMAX * (KP / 4); i += HEAD_T) {
            const int cc = i / (KP / 4), kk = (i % (KP / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cc < C && kk < K) v = *reinterpret_cast<const float4 *>(a.w + cc * K + kk);
            *reinterpret_cast<float4 *>(Ws + cc * LD + kk) = v;
        }
    } else {
        for (int i = t; i < HEAD_CMAX * KP; i += HEAD_T) {
            const int cc = i / KP, kk = i % KP;
            Ws[cc * LD + kk] = (cc < C && kk < K) ? a.w[cc * K + kk] : 0.f;
        }
    }

    if (a.w_adam.p) w_step = adam_step_size(w_lr, a.w_adam.beta1, a.w_adam.beta2, w_t);   // optim.rs:87-90
    if (a.b_adam.p) b_step = adam_step_size(b_lr, a.b_adam.beta1, a.b_adam.beta2, b_t);
    HEAD_STAMP(1);
    // dW tiles: wave `wave` owns column tile `wave` (K <= 256 -> at most 16 tiles)
    floatx4 dw_acc = {0.f, 0.f, 0.f, 0.f};
    const int dw_tiles = KP / 16;
    float db_acc = 0.f;                      // threads 0..15
    float nll_acc = 0.f, hit_acc = 0.f;      // lane 0 of each 16-lane row group
    const int row_l = t >> 4, sub = t & 15;
    const float inv_b = 1.0f / (float)a.batch;

    for (int r0 = row_beg; r0 < row_end; r0 += HEAD_RC) {
        const int rows = min(HEAD_RC, row_end - r0);
        if (r0 > row_beg) __syncthreads();  // previous chunk's readers are done with Hs / lg
        const float tf = (row_l < rows) ? a.targets[r0 + row_l] : 0.f;   // requested with the H loads
        if (r0 == row_beg && h0_staged) {
            // chunk 0 is already in LDS
        } else if (vec4) {
            for (int i = t; i < HEAD_RC * (KP / 4); i += HEAD_T) {
                const int rr = i / (KP / 4), kk = (i % (KP / 4)) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rr < rows && kk < K) v = *reinterpret_cast<const float4 *>(a.h + (long)(r0 + rr) * K + kk);
                *reinterpret_cast<float4 *>(Hs + rr * LD + kk) = v;
            }
        } else {
            for (int i = t; i < HEAD_RC * KP; i += HEAD_T) {
                const int rr = i / KP, kk = i % KP;
                Hs[rr * LD + kk] = (rr < rows && kk < K) ? a.h[(long)(r0 + rr) * K + kk] : 0.f;
            }
        }
        __syncthreads();
        HEAD_STAMP(2);

        // ---- logits tile (nn.rs:54-60): waves 4*tile .. 4*tile+3 split K ----
        {
            const int tile = wave >> 2, kp = wave & 3;
            const int ksteps = KP / 16;                   // k-steps of 16
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            const float *ap = Hs + (tile * 16 + r16) * LD;   // A[i = row][k]
            const float *bp = Ws + r16 * LD;                 // B[k][j = class] = W[class][k]
            for (int ks = kp; ks < ksteps; ks += 4) {
                const float4 av = *reinterpret_cast<const float4 *>(ap + ks * 16 + g4 * 4);
                const float4 bv = *reinterpret_cast<const float4 *>(bp + ks * 16 + g4 * 4);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
            }
            if (kp > 0) {
                float *pp = part + ((tile * 3 + (kp - 1)) * 64 + lane) * 4;
                pp[0] = acc[0]; pp[1] = acc[1]; pp[2] = acc[2]; pp[3] = acc[3];
            }
            __syncthreads();
            if (kp == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float *pp = part + ((tile * 3 + q) * 64 + lane) * 4;
                    acc[0] += pp[0]; acc[1] += pp[1]; acc[2] += pp[2]; acc[3] += pp[3];
                }
                // C/D map: col = lane & 15 (class), row = (lane >> 4) * 4 + i
#pragma unroll
                for (int i = 0; i < 4; ++i) lg[(tile * 16 + g4 * 4 + i) * HEAD_CMAX + r16] = acc[i] + bias_v;
            }
        }
        __syncthreads();
        HEAD_STAMP(3);

        // ---- softmax cross-entropy of the row inside its 16 lanes (loss.rs:101-195) ----
        {
            const bool active = row_l < rows;
            const bool valid = active && sub < C;
            const float logit = valid ? lg[row_l * HEAD_CMAX + sub] : -INFINITY;
            if (valid && a.logits) a.logits[(long)(r0 + row_l) * C + sub] = logit;
            float best = logit;
            int bi = (valid && logit > -INFINITY) ? sub : 0x7fffffff;   // NaN / -inf never win (tensor.rs:1062)
            if (bi == 0x7fffffff) best = -INFINITY;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                const float ov = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (ov > best || (ov == best && oi < bi)) {
                    best = ov;
                    bi = oi;
                }
            }
            if (bi == 0x7fffffff) bi = 0;
            float se = valid ? expf(logit - best) : 0.f;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) se += __shfl_xor(se, off, 64);
            const float log_sum = logf(se);
            float my_nll = 0.f, dlv = 0.f;
            const long cls = head_target_class(tf);
            if (valid) {
                const float lp = (logit - best) - log_sum;   // loss.rs:117-125
                float gv = expf(lp);                          // loss.rs:178
                if (sub == cls) {
                    my_nll = -lp;
                    gv -= 1.0f;
                }
                dlv = gv * inv_b;                             // loss.rs:185-188 with g0 = 1
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) my_nll += __shfl_xor(my_nll, off, 64);
            if (active && sub == 0) {
                nll_acc += (cls >= C) ? NAN : my_nll;         // the reference panics (loss.rs:161): NaN + a note for the next wait
                if (cls >= C) raise_target_oob(cls, C);
                hit_acc += (fabsf((float)bi - tf) < 1e-6f) ? 1.f : 0.f;  // loss.rs:283
            }
            dl[row_l * HEAD_CMAX + sub] = dlv;                // rows >= `rows`, classes >= C hold 0
            // db: column sums of dlogits (tensor.rs:686-691) -- the wave's 4 rows by shuffles, waves via LDS
            float cs = dlv;
            cs += __shfl_xor(cs, 16, 64);
            cs += __shfl_xor(cs, 32, 64);
            if (lane < HEAD_CMAX) dbp[wave * HEAD_CMAX + lane] = cs;
        }
        __syncthreads();
        HEAD_STAMP(4);

        // ---- dH[row][k] = sum_c dl[row][c] * W[c][k]   (ops.rs:254-265): K_mfma = 16 classes ----
        if (a.dh) {
            const int ntiles = 4 * dw_tiles;                  // 4 row tiles x K/16 column tiles
            for (int tl = wave; tl < ntiles; tl += 16) {
                const int rt = tl / dw_tiles, ct = tl % dw_tiles;
                const float4 av = *reinterpret_cast<const float4 *>(dl + (rt * 16 + r16) * HEAD_CMAX + g4 * 4);  // A[row][class]
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
                const float *bp = Ws + ct * 16 + r16;         // B[k = class][j = col] = W[class][col]
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bp[(g4 * 4 + 0) * LD], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bp[(g4 * 4 + 1) * LD], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bp[(g4 * 4 + 2) * LD], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bp[(g4 * 4 + 3) * LD], acc, 0, 0, 0);
                const int col = ct * 16 + r16;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rr = rt * 16 + g4 * 4 + i;
                    if (rr < rows && col < K) a.dh[(long)(r0 + rr) * K + col] = (a.mask_dh && !(Hs[rr * LD + col] > 0.f)) ? 0.f : acc[i];
                }
            }
        }
        HEAD_STAMP(5);
        // ---- dW[class][k] += sum_row dl[row][class] * H[row][k]   (ops.rs:280-291): K_mfma = 64 rows ----
        if (own_dw) {
            const float *bp = Hs + wave * 16 + r16;           // B[k = row][j = col] = H[row][col]
            floatx4 acc2 = {0.f, 0.f, 0.f, 0.f};              // second chain: halves the dependent-MFMA latency
#pragma unroll
            for (int ks = 0; ks < HEAD_RC / 16; ks += 2) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int r1 = ks * 16 + g4 * 4 + s, r2 = r1 + 16;
                    dw_acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dl[r1 * HEAD_CMAX + r16], bp[r1 * LD], dw_acc, 0, 0, 0);  // A[class][row]
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(dl[r2 * HEAD_CMAX + r16], bp[r2 * LD], acc2, 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) dw_acc[i] += acc2[i];
        }
        // ---- db[class] += sum over the 16 waves' partial column sums, fixed order ----
        if (a.db && t < HEAD_CMAX) {
            float sdb = 0.f;
#pragma unroll
            for (int w = 0; w < HEAD_T / 64; ++w) sdb += dbp[w * HEAD_CMAX + t];
            db_acc += sdb;
        }
    }

    HEAD_STAMP(6);
    // ---- loss / count: fixed-order block reduction ----
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        nll_acc += __shfl_down(nll_acc, off, 64);
        hit_acc += __shfl_down(hit_acc, off, 64);
    }
    if (lane == 0) {
        red[wave] = nll_acc;
        red[16 + wave] = hit_acc;
    }
    __syncthreads();
    if (t == 0) {
        float n = 0.f, hsum = 0.f;
        for (int w = 0; w < HEAD_T / 64; ++w) {
            n += red[w];
            hsum += red[16 + w];
        }
        if (multi) {   // raw sums of this workgroup's rows; head_finish_kernel divides, logs and ticks
            a.part_scalar[2 * blockIdx.x] = n;
            a.part_scalar[2 * blockIdx.x + 1] = hsum;
        }
        const float l = n / (float)a.batch;  // loss.rs:164
        if (!multi) a.loss[0] = l;
        if (!multi && a.ncorrect) a.ncorrect[0] = hsum;
        if (!multi && a.adam_tick) a.adam_tick[0] = tick_old + 1;  // optim.rs:84
        if (!multi && a.metrics) {  // the step log of th_log_step
            a.metrics[2 * log_slot] = l;
            a.metrics[2 * log_slot + 1] = hsum;
            a.state[0] += 1;
            a.state[1] += a.advance;
        }
    }

    HEAD_STAMP(7);
    // ---- gradients out (+ fused Adam, optim.rs:99-110: every read of W above came from the LDS
    //      copy and the p / m / v values were requested at kernel entry) ----
    if (own_dw) {
        const int col = wave * 16 + r16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cls = g4 * 4 + i;                       // C/D map: row = class, col = k
            if (cls < C && col < K) {
                const long idx = (long)cls * K + col;
                a.dw[idx] = dw_acc[i];
                if (fuse_w) {
                    const float gv = dw_acc[i] + a.w_adam.wd * wp_[i];
                    const float mn = a.w_adam.beta1 * wm_[i] + (1.0f - a.w_adam.beta1) * gv;
                    const float vn = a.w_adam.beta2 * wv_[i] + (1.0f - a.w_adam.beta2) * gv * gv;
                    a.w_adam.m[idx] = mn;
                    a.w_adam.v[idx] = vn;
                    a.w_adam.p[idx] = wp_[i] - w_step * mn / (sqrtf(vn) + a.w_adam.eps);
                }
            }
        }
    }
    if (a.db && t < C) {
        a.db[t] = db_acc;
        if (fuse_b) {
            const float gv = db_acc + a.b_adam.wd * bp_;
            const float mn = a.b_adam.beta1 * bm_ + (1.0f - a.b_adam.beta1) * gv;
            const float vn = a.b_adam.beta2 * bv_ + (1.0f - a.b_adam.beta2) * gv * gv;
            a.b_adam.m[t] = mn;
            a.b_adam.v[t] = vn;
            a.b_adam.p[t] = bp_ - b_step * mn / (sqrtf(vn) + a.b_adam.eps);
        }
    }
    HEAD_STAMP(8);
}

// Last pass of the multi-workgroup head (dW was summed over the slots by th_colsum): db = sum of
// the workgroups' slots in slot order, then loss / n_correct, the step log and Adam's tick.
__global__ __launch_bounds__(256) void head_finish_kernel(HeadArgs a, const float *__restrict__ part_dw,
                                                          const float *__restrict__ part_db, int n_wg) {
    // part_dw != nullptr (few slots): every block also adds its share of dW over the slots, in slot order
    if (part_dw) {
        const int n_dw = a.c * a.k;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n_dw; i += gridDim.x * 256) {
            float s = 0.f;
            for (int g = 0; g < n_wg; ++g) s += part_dw[(long)g * n_dw + i];
            a.dw[i] = s;
        }
    }
    if (blockIdx.x != 0) return;
    __shared__ float sh[2][4];
    __shared__ float dbs[16][HEAD_CMAX];
    const int t = threadIdx.x;
    if (a.db) {   // thread (part, cls) adds slots part, part+16, ...; thread cls then adds the 16 parts in order
        const int cls = t & 15, part = t >> 4;
        float s = 0.f;
        if (cls < a.c)
            for (int g = part; g < n_wg; g += 16) s += part_db[(long)g * a.c + cls];
        dbs[part][cls] = s;
        __syncthreads();
        if (t < a.c) {
            float tot = 0.f;
#pragma unroll
            for (int p = 0; p < 16; ++p) tot += dbs[p][t];
            a.db[t] = tot;
        }
    }
    float n = 0.f, hsum = 0.f;
    for (int g = t; g < n_wg; g += 256) {
        n += a.part_scalar[2 * g];
        hsum += a.part_scalar[2 * g + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_down(n, off, 64);
        hsum += __shfl_down(hsum, off, 64);
    }
    if ((t & 63) == 0) {
        sh[0][t >> 6] = n;
        sh[1][t >> 6] = hsum;
    }
    __syncthreads();
    if (t == 0) {
        n = ((sh[0][0] + sh[0][1]) + sh[0][2]) + sh[0][3];
        hsum = ((sh[1][0] + sh[1][1]) + sh[1][2]) + sh[1][3];
        const float l = n / (float)a.batch;  // loss.rs:164
        a.loss[0] = l;
        if (a.ncorrect) a.ncorrect[0] = hsum;
        if (a.adam_tick) a.adam_tick[0] += 1;  // optim.rs:84
        if (a.metrics) {
            const int64_t slot = a.state[0] % a.capacity;
            a.metrics[2 * slot] = l;
            a.metrics[2 * slot + 1] = hsum;
            a.state[0] += 1;
            a.state[1] += a.advance;
        }
    }
}

#ifdef TH_PROFILE
__global__ void head_prof_end_kernel() { g_head_prof[15] = wall_clock64(); }
#endif

static size_t head_lds_bytes(int k) {
    const int kp = (k + 15) & ~15, ld = kp + 4;
    return ((size_t)(HEAD_RC + HEAD_CMAX) * ld + 2 * HEAD_RC * HEAD_CMAX + 4 * 3 * 64 * 4 + 32 + 16 * HEAD_CMAX) * sizeof(float);
}

}  // namespace th

using namespace th;

extern "C" int th_linear_xent_head(th_ctx *ctx, const float *d_h, const float *d_w, const float *d_bias, const float *d_targets,
                                   int batch, int in_features, int classes, float *d_logits, float *d_loss, float *d_ncorrect,
                                   float *d_dh, float *d_dw, float *d_db, float *d_metrics, int64_t metrics_capacity,
                                   int64_t *d_state, int64_t advance, int32_t *d_adam_tick, const th_adam_fuse *w_fuse,
                                   const th_adam_fuse *b_fuse) {
    return th_linear_xent_head_masked(ctx, d_h, d_w, d_bias, d_targets, batch, in_features, classes, d_logits, d_loss, d_ncorrect, d_dh,
                                      d_dw, d_db, d_metrics, metrics_capacity, d_state, advance, d_adam_tick, w_fuse, b_fuse, 0);
}

extern "C" int th_linear_xent_head_masked(th_ctx *ctx, const float *d_h, const float *d_w, const float *d_bias, const float *d_targets,
                                          int batch, int in_features, int classes, float *d_logits, float *d_loss, float *d_ncorrect,
                                          float *d_dh, float *d_dw, float *d_db, float *d_metrics, int64_t metrics_capacity,
                                          int64_t *d_state, int64_t advance, int32_t *d_adam_tick, const th_adam_fuse *w_fuse,
                                          const th_adam_fuse *b_fuse, int mask_dh_by_h) {
    TH_REQUIRE(ctx && d_h && d_w && d_targets && d_loss, "th_linear_xent_head: null argument");
    TH_REQUIRE(batch > 0 && batch <= (1 << 22) && classes > 0 && classes <= HEAD_CMAX && in_features > 0 && in_features <= HEAD_KMAX,
               "th_linear_xent_head: needs batch <= 4194304, classes <= 16, in_features <= 256 (got %d, %d, %d)", batch, classes,
               in_features);
    TH_REQUIRE(!d_metrics || (d_state && metrics_capacity > 0), "th_linear_xent_head: metrics need d_state and a capacity");
    TH_REQUIRE(!(w_fuse && w_fuse->d_p) || d_dw, "th_linear_xent_head: fused W update needs d_dw");
    TH_REQUIRE(!(b_fuse && b_fuse->d_p) || d_db, "th_linear_xent_head: fused b update needs d_db");
    HeadArgs a{d_h, d_w, d_bias, d_targets, batch, in_features, classes, d_logits, d_loss, d_ncorrect, d_dh, d_dw, d_db,
               d_metrics, metrics_capacity, d_state, advance, d_adam_tick, make_adam_dev(w_fuse), make_adam_dev(b_fuse), 0, nullptr,
               mask_dh_by_h ? 1 : 0};
    const size_t lds = head_lds_bytes(in_features);
    TH_SET_MAX_LDS(ctx, linear_xent_head_kernel, head_lds_bytes(HEAD_KMAX));   // (per device: ADVICE r04)
    if (batch <= 64) {   // latency-bound: one workgroup, one launch (a single 64-row chunk)
        hipLaunchKernelGGL(linear_xent_head_kernel, dim3(1), dim3(HEAD_T), lds, ctx->stream, a);
        TH_LAUNCH_CHECK();
        return 0;
    }
    // throughput-bound: up to 256 workgroups of 64-row chunks + a finish pass
    int rows_per_wg = ceil_div(ceil_div(batch, 256), HEAD_RC) * HEAD_RC;
    const int n_wg = ceil_div(batch, rows_per_wg);
    const size_t n_dw = d_dw ? (size_t)classes * in_features : 0, n_db = d_db ? (size_t)classes : 0;
    void *ws = nullptr;
    if (th_malloc(ctx, (size_t)n_wg * (n_dw + n_db + 2) * sizeof(float), &ws)) return 1;
    float *part_dw = (float *)ws, *part_db = part_dw + (size_t)n_wg * n_dw, *part_sc = part_db + (size_t)n_wg * n_db;
    HeadArgs rows = a;
    rows.rows_per_wg = rows_per_wg;
    rows.dw = d_dw ? part_dw : nullptr;
    rows.db = d_db ? part_db : nullptr;
    rows.part_scalar = part_sc;
    rows.w_adam.p = nullptr;      // fused updates need the complete gradient: they run behind the finish pass
    rows.b_adam.p = nullptr;
    hipLaunchKernelGGL(linear_xent_head_kernel, dim3(n_wg), dim3(HEAD_T), lds, ctx->stream, rows);
    TH_LAUNCH_CHECK();
    a.part_scalar = part_sc;
    if (d_dw && n_wg > 16) {   // dW[c][k] = sum over the workgroups' slots: a column sum of the [n_wg, c*k] slot matrix
        if (int rc = th_colsum(ctx, part_dw, d_dw, n_wg, (int)n_dw)) return rc;
        hipLaunchKernelGGL(head_finish_kernel, dim3(1), dim3(256), 0, ctx->stream, a, (const float *)nullptr, (const float *)part_db, n_wg);
    } else {                   // few slots: the finish launch adds them itself
        hipLaunchKernelGGL(head_finish_kernel, dim3(d_dw ? ceil_div((long)n_dw, 256) : 1), dim3(256), 0, ctx->stream, a,
                           d_dw ? (const float *)part_dw : (const float *)nullptr, (const float *)part_db, n_wg);
    }
    TH_LAUNCH_CHECK();
    if (th_free(ctx, ws)) return 1;
    if (int rc = adam_slice(ctx, a.w_adam, d_dw, (int64_t)n_dw)) return rc;
    return adam_slice(ctx, a.b_adam, d_db, (int64_t)n_db);
}

#ifdef TH_PROFILE
extern "C" int th_debug_head_prof(th_ctx *ctx, long long *h_out16) {
    hipLaunchKernelGGL(head_prof_end_kernel, dim3(1), dim3(1), 0, ctx->stream);
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(g_head_prof), 16 * sizeof(long long)));
    return 0;
}
#endif

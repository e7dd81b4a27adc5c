// wide_head.hip -- th_linear_xent_wide: the classifier head for a WIDE input (in_features > 256: the simple CNN's
// Linear(3136, 10) on the flattened feature map) -- last Linear + softmax cross-entropy + every backward product of that
// Linear (nn.rs:54-60, loss.rs:101-195, 271-290, ops.rs:238-294, tensor.rs:574-587, 674-694) in TWO launches instead of
// five (forward, K-slice reduce, loss, backward, ...):
//   launch 1  the K slices of logits = X . W^T                         (gemm.hip: linear_fwd_partials)
//   launch 2  this kernel: a workgroup (16 waves) owns 32 input columns; each wave takes 16-row blocks of the batch:
//             logits (slices added in order + bias) -> softmax / NLL / argmax / dlogits in registers (tail_row_softmax),
//             dX[rows][its columns] = dl . W      A = dl[row r][class 4g+s] (= register s), B = W[class 4g+s][col r]
//             dW[classes][its columns] += dl^T . X  A = dl^T through a wave-private LDS transpose, B = X[row 4g+s][col r]
//             one extra "lead" workgroup: db, loss, hit count, the step log and Adam's t += 1.
// No parameter is updated here (W is read by every workgroup's dX product): the caller defers W / b (th_adam_slice).
#include "tail_dev.h"
#include "dp_dev.h"
#include "adam_dev.h"

TH_USES_DEVICE_ERRORS()

namespace th {

int linear_fwd_partials(th_ctx *ctx, const float *x, const float *w, int m, int n, int k, float **partial_out, int *kz_out);   // gemm.hip

struct WideArgs {
    const float *x, *w, *bias, *targets, *partial;
    int batch, k, c, kz;
    float *loss, *ncorrect, *dx, *dw, *db;
    float *metrics;
    int64_t capacity;
    int64_t *state;
    int64_t advance;
    int32_t *adam_tick;
    int n_col;           // column blocks (32 columns each); block n_col is the lead
    float *colsum;       // optional [k]: sum over the rows of dX[row][col] * [x[row][col] > 0] (see th_linear_xent_wide_ex)
    // th_linear_xent_wide_fused: Adam (optim.rs:99-110) for W (every workgroup: the columns it owns -- nobody else reads them in this launch)
    // and b (lead), and the finish of the conv bias in front (the LAST workgroup to arrive: db[ch] = sum of ch's hw column sums, + Adam).
    // The step counter is ticked HERE: every workgroup forms t + 1 itself, the last arriver publishes it (adam_tick[1] = arrival counter).
    AdamDev fw, fb, fcb;
    float *conv_gb;
    int conv_c, conv_hw, fused;
};

#ifndef TH_WH_TX
#define TH_WH_TX 2
#endif
constexpr int WH_TX = TH_WH_TX;   // 16-column tiles per workgroup (tuning probe: -DTH_WH_TX=1: 196 workgroups of 16 columns)
#ifdef TH_PROFILE
__device__ long long g_wh_prof[16];   // wall clock (100 MHz) at the phase boundaries of workgroup 5, wave 0
#define WH_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x == 5) g_wh_prof[i] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WH_STAMP(i) do { } while (0)
#endif

constexpr int WH_KZ_MAX = 8;   // K slices of the logits (linear_fwd_partials: <= k / 512, capped here)
constexpr int WH_NW = 16;  // waves per workgroup: 256 rows per pass (the chunks of a batch are a serial chain per wave, so go wide)

__global__ __launch_bounds__(64 * WH_NW) void wide_head_kernel(WideArgs a) {
    __shared__ float red[WH_NW][WH_TX][64][4];
    __shared__ float tr[WH_NW][16][17];
    __shared__ float rowv[WH_NW][2][16];
    __shared__ float sc[WH_NW][20];
    __shared__ float colred[WH_NW][WH_TX][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r16 = lane & 15, g4 = lane >> 4;
    WH_STAMP(0);
    const bool lead = (int)blockIdx.x == a.n_col;
    const int B = a.batch, K = a.k, C = a.c;
    const int col0 = blockIdx.x * 16 * WH_TX;
    const float inv_b = 1.0f / (float)B;
    const long bc = (long)B * C;

    // chunk-independent operands: W[class 4 g4 + s][col0 + 16 tx + r16], bias
    float wv[WH_TX][4], b4[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int cls = min(g4 * 4 + s, C - 1);
        b4[s] = a.bias ? a.bias[cls] : 0.f;
#pragma unroll
        for (int tx = 0; tx < WH_TX; ++tx) {
            const int col = col0 + tx * 16 + r16;
            wv[tx][s] = (!lead && g4 * 4 + s < C && col < K) ? a.w[(long)cls * K + col] : 0.f;
        }
    }
    const int64_t state0 = (lead && a.metrics) ? a.state[0] : 0, state1 = (lead && a.metrics) ? a.state[1] : 0;
    const int32_t tick_old = ((lead || a.fused) && a.adam_tick) ? __hip_atomic_load(&a.adam_tick[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    __shared__ int last_flag;

    floatx4 accw[WH_TX];
#pragma unroll
    for (int tx = 0; tx < WH_TX; ++tx) accw[tx] = floatx4{0.f, 0.f, 0.f, 0.f};
    float db_acc = 0.f, nll_acc = 0.f, hit_acc = 0.f;
    float cs[WH_TX];     // this lane's share of the masked column sums (rows 4 g4 + i of every block the wave takes, column r16 of each tile)
#pragma unroll
    for (int tx = 0; tx < WH_TX; ++tx) cs[tx] = 0.f;

    for (int c0 = 0; c0 < B; c0 += 16 * WH_NW) {
        const int r0 = c0 + wave * 16;
        const bool rows_here = r0 < B;                 // wave-uniform
        if (rows_here) {
            const int row_a = r0 + r16, row_ac = min(row_a, B - 1);
            const bool row_ok = row_a < B;
            // logits of (row_a, classes 4 g4 ..): the K slices in slice order, then the bias (nn.rs:54-60)
            // (every slice of every class is requested before the first add: as a loop of load + add the compiler waited for each load --
            // 4 kz dependent L2 round trips, 24 at kz = 6: most of this kernel's 12 us)
            float lg[4], pv[4][WH_KZ_MAX];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cls = min(g4 * 4 + i, C - 1);
                const float *pp = a.partial + (long)row_ac * C + cls;
#pragma unroll
                for (int z = 0; z < WH_KZ_MAX; ++z) pv[i][z] = pp[(long)min(z, a.kz - 1) * bc];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float sum = 0.f;
#pragma unroll
                for (int z = 0; z < WH_KZ_MAX; ++z) sum += z < a.kz ? pv[i][z] : 0.f;
                lg[i] = (g4 * 4 + i < C) ? sum + b4[i] : -INFINITY;
            }
            const float tf = a.targets[row_ac];
            float xv[WH_TX][4];                        // X[r0 + 4 g4 + s][col0 + 16 tx + r16]
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int row = r0 + g4 * 4 + s;
#pragma unroll
                for (int tx = 0; tx < WH_TX; ++tx) {
                    const int col = col0 + tx * 16 + r16;
                    xv[tx][s] = (!lead && row < B && col < K) ? a.x[(long)row * K + col] : 0.f;
                }
            }
            WH_STAMP(1);
            float dl[4], nll_row;
            int bi;
            tail_row_softmax(lg, g4, C, tf, inv_b, dl, nll_row, bi);
            WH_STAMP(2);
#pragma unroll
            for (int i = 0; i < 4; ++i) dl[i] = row_ok ? dl[i] : 0.f;

            // wave-private transpose of dlogits: A operand of the dW product, column sums for db
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) tr[wave][r16][g4 * 4 + i] = dl[i];
            if (lead && g4 == 0) {
                rowv[wave][0][r16] = row_ok ? nll_row : 0.f;
                rowv[wave][1][r16] = (row_ok && fabsf((float)bi - tf) < 1e-6f) ? 1.f : 0.f;   // loss.rs:283
            }
            __builtin_amdgcn_wave_barrier();
            if (!lead) {
                // dX block (ops.rs:254-265): rows 4 g4 + i of this wave's block, columns r16 of each tile
                if (a.dx || a.colsum) {
#pragma unroll
                    for (int tx = 0; tx < WH_TX; ++tx) {
                        floatx4 ax = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < 4; ++s) ax = __builtin_amdgcn_mfma_f32_16x16x4f32(dl[s], wv[tx][s], ax, 0, 0, 0);
                        const int col = col0 + tx * 16 + r16;
                        if (a.dx) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int row = r0 + g4 * 4 + i;
                                if (row < B && col < K) a.dx[(long)row * K + col] = ax[i];
                            }
                        }
                        // x[row 4 g4 + i][col] sits in xv[tx][i] (the dW operand): the D tile and that operand share their (row, column) map
                        if (a.colsum) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) cs[tx] += xv[tx][i] > 0.f ? ax[i] : 0.f;
                        }
                    }
                }
                // dW tile (ops.rs:280-291 through the W^T node): A[class r16][row 4 g4 + s], B = X[row 4 g4 + s][col r16]
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float at = tr[wave][g4 * 4 + s][r16];
#pragma unroll
                    for (int tx = 0; tx < WH_TX; ++tx) accw[tx] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, xv[tx][s], accw[tx], 0, 0, 0);
                }
            } else {
                float cs = 0.f, nl = 0.f, ht = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    cs += tr[wave][r][r16];
                    nl += rowv[wave][0][r];
                    ht += rowv[wave][1][r];
                }
                db_acc += cs;
                nll_acc += nl;
                hit_acc += ht;
            }
        }
    }

    WH_STAMP(3);
    if (!lead && a.colsum) {   // the four row groups of a wave in a fixed tree, then the waves in order: deterministic
#pragma unroll
        for (int tx = 0; tx < WH_TX; ++tx) {
            float v = cs[tx];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g4 == 0) colred[wave][tx][r16] = v;
        }
        __syncthreads();
        if (t < 16 * WH_TX) {
            const int tx = t >> 4, r = t & 15, col = col0 + tx * 16 + r;
            float sum = colred[0][tx][r];
#pragma unroll
            for (int w = 1; w < WH_NW; ++w) sum += colred[w][tx][r];
            if (col < K) a.colsum[col] = sum;
        }
    }
    WH_STAMP(4);
    if (!lead) {   // deterministic cross-wave sum; wave e finishes class 4 g4 + e
#pragma unroll
        for (int tx = 0; tx < WH_TX; ++tx)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wave][tx][lane][i] = accw[tx][i];
        __syncthreads();
        if (wave < 4) {
            const int cls = g4 * 4 + wave;
#pragma unroll
            for (int tx = 0; tx < WH_TX; ++tx) {
                const int col = col0 + tx * 16 + r16;
                float sum = red[0][tx][lane][wave];
#pragma unroll
                for (int w = 1; w < WH_NW; ++w) sum += red[w][tx][lane][wave];
                if (cls < C && col < K) {
                    a.dw[(long)cls * K + col] = sum;
                    if (a.fused && a.fw.p)
                        adam_update(a.fw.p, a.fw.m, a.fw.v, (long)cls * K + col, sum, adam_step_size(a.fw.lr[0], a.fw.beta1, a.fw.beta2, tick_old + 1),
                                    a.fw.beta1, a.fw.beta2, a.fw.eps, a.fw.wd);
                }
            }
        }
        WH_STAMP(5);
        if (!a.fused) return;
    }
    if (a.fused) {
        // arrival: the last workgroup finishes the conv bias from everybody's column sums and publishes the step counter
        if (lead) {
            if (g4 == 0) sc[wave][r16] = db_acc;
            if (lane == 0) {
                sc[wave][16] = nll_acc;
                sc[wave][17] = hit_acc;
            }
            __syncthreads();
            if (a.db && t < C) {
                float sum = sc[0][t];
                for (int w = 1; w < WH_NW; ++w) sum += sc[w][t];
                a.db[t] = sum;      // (b itself moves in the LAST workgroup to arrive, below: every workgroup of this launch reads b at its start)
            }
            if (t == 0) {
                float n = sc[0][16], hsum = sc[0][17];
                for (int w = 1; w < WH_NW; ++w) {
                    n += sc[w][16];
                    hsum += sc[w][17];
                }
                const float l = n / (float)B;   // loss.rs:164
                a.loss[0] = l;
                if (a.ncorrect) a.ncorrect[0] = hsum;
                if (a.metrics) {
                    const int64_t slot = state0 < a.capacity ? state0 : state0 % a.capacity;
                    a.metrics[2 * slot] = l;
                    a.metrics[2 * slot + 1] = hsum;
                    a.state[0] = state0 + 1;
                    a.state[1] = state1 + a.advance;
                }
            }
        }
        __syncthreads();
        if (t == 0) {
            __threadfence();                                   // this workgroup's column sums / updates are out before it is counted
            const int arrived = __hip_atomic_fetch_add(&a.adam_tick[1], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = arrived == (int)gridDim.x - 1;
        }
        __syncthreads();
        if (!last_flag) return;
        __threadfence();                                       // ... and everybody else's are visible here
        // b's update (optim.rs:99-110) from the lead's db.  NOT in the lead workgroup itself: every workgroup reads b for its logits when it
        // starts, and a workgroup that is dispatched late -- other processes on the GPU -- would read the moved bias (r05: found by
        // tests/test_gpu_repro.py beside two HBM-streaming processes: W off by ~0.1 lr on a replay; alone on the GPU every workgroup is
        // resident at once and none was ever late).  Here every workgroup has arrived, i.e. is past its reads.
        if (a.db && a.fb.p && t < C) {
            const float g = __hip_atomic_load(&a.db[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            adam_update(a.fb.p, a.fb.m, a.fb.v, t, g, adam_step_size(a.fb.lr[0], a.fb.beta1, a.fb.beta2, tick_old + 1), a.fb.beta1, a.fb.beta2,
                        a.fb.eps, a.fb.wd);
        }
        if (a.colsum && a.conv_gb) {
            // 16 lanes per channel of the conv in front: lane j adds column sums j, j + 16, ..., then a fixed shuffle tree (th_bias_from_colsum_adam)
            const int sub = t & 15;
            const float cstep = a.fcb.p ? adam_step_size(a.fcb.lr[0], a.fcb.beta1, a.fcb.beta2, tick_old + 1) : 0.f;
            for (int ch0 = 0; ch0 < a.conv_c; ch0 += 64 * WH_NW / 16) {
                const int ch = ch0 + (t >> 4);
                const float *row = a.colsum + (long)(ch < a.conv_c ? ch : 0) * a.conv_hw;
                float part = 0.f;
                for (int j = sub; j < a.conv_hw; j += 64) {
                    const float v0 = __builtin_nontemporal_load(row + j), v1 = j + 16 < a.conv_hw ? __builtin_nontemporal_load(row + j + 16) : 0.f,
                                v2 = j + 32 < a.conv_hw ? __builtin_nontemporal_load(row + j + 32) : 0.f,
                                v3 = j + 48 < a.conv_hw ? __builtin_nontemporal_load(row + j + 48) : 0.f;
                    part += (v0 + v1) + (v2 + v3);
                }
                part += __shfl_xor(part, 8, 64);
                part += __shfl_xor(part, 4, 64);
                part += __shfl_xor(part, 2, 64);
                part += __shfl_xor(part, 1, 64);
                if (sub == 0 && ch < a.conv_c) {
                    a.conv_gb[ch] = part;
                    if (a.fcb.p) adam_update(a.fcb.p, a.fcb.m, a.fcb.v, ch, part, cstep, a.fcb.beta1, a.fcb.beta2, a.fcb.eps, a.fcb.wd);
                }
            }
        }
        if (t == 0) {
            __hip_atomic_store(&a.adam_tick[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.adam_tick[0], tick_old + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // optim.rs:84
        }
        return;
    }
    if (g4 == 0) sc[wave][r16] = db_acc;
    if (lane == 0) {
        sc[wave][16] = nll_acc;
        sc[wave][17] = hit_acc;
    }
    __syncthreads();
    if (a.db && t < C) {
        float sum = sc[0][t];
        for (int w = 1; w < WH_NW; ++w) sum += sc[w][t];
        a.db[t] = sum;
    }
    if (t == 0) {
        float n = sc[0][16], hsum = sc[0][17];
        for (int w = 1; w < WH_NW; ++w) {
            n += sc[w][16];
            hsum += sc[w][17];
        }
        const float l = n / (float)B;   // loss.rs:164
        a.loss[0] = l;
        if (a.ncorrect) a.ncorrect[0] = hsum;
        if (a.adam_tick) a.adam_tick[0] = tick_old + 1;   // optim.rs:84 (nobody else touches the counter in this launch)
        if (a.metrics) {
            const int64_t slot = state0 < a.capacity ? state0 : state0 % a.capacity;
            a.metrics[2 * slot] = l;
            a.metrics[2 * slot + 1] = hsum;
            a.state[0] = state0 + 1;
            a.state[1] = state1 + a.advance;
        }
    }
}

// ---- th_wide_head_grads: the batch sums behind a chain launch that already did the classifier ROW by row (conv_chain.hip:
// chain_head_rows) -- dW = dl^T X, db, the loss, the hit count, the conv bias in front -- with every Adam update in the epilogue of the
// workgroup that owns the gradient: no workgroup of this launch reads a parameter, so nothing is deferred.
//   blocks [0, n_col)      32 input columns each: 16 waves x 16-row blocks of dl^T X on v_mfma_f32_16x16x4_f32 (A = dl^T straight from the
//                          [n][16] row records, B = X[row][col]), cross-wave sum in fixed order, Adam(W)
//   block n_col            db (+ Adam), loss, hit count, step log
//   blocks > n_col         16 conv channels each: sum over the images of the per-image channel sums (+ Adam)
struct WideGradArgs {
    const float *x, *dl, *rowstat, *cbpart;
    int batch, k, c, conv_c, n_col;
    float *dw, *db, *conv_gb, *loss, *ncorrect, *metrics;
    int64_t capacity;
    int64_t *state;
    int64_t advance;
    AdamDev fw, fb, fcb;
    DpDev dp;            // th_wide_head_grads_dp: the gradient exchange across ranks between a finished sum and its store / Adam (dp_dev.h)
    int32_t *dp_tick;    // ... Adam's counter, ticked by the chain launch in front: taken back when the exchange fails
};

// column sums of a [rows][ld] slab, columns col0 .. col0 + 15 (ncols live): thread t < 16 returns the sum of column col0 + t, rows in
// groups of 64 (thread (g, c) adds rows g, g + 64, ...), groups added in order.  1024 threads.
__device__ __forceinline__ float slab_colsum16(const float *__restrict__ p, int ld, int rows, int col0, int ncols, float (*red)[16], int t) {
    const int g = t >> 4, c = t & 15;
    // (a thread's rows are added in order, but requested eight at a time: one row per trip was a chain of rows / 64 memory round trips --
    // at 4 096 rows the ONE workgroup that sums the dlogits set the whole launch's 31 us, r06)
    float s = 0.f;
    if (c < ncols) {
        int r = g;
        for (; r + 15 * 64 < rows; r += 16 * 64) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p[(long)(r + 64 * u) * ld + col0 + c];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; r + 3 * 64 < rows; r += 4 * 64) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p[(long)(r + 64 * u) * ld + col0 + c];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        for (; r < rows; r += 64) s += p[(long)r * ld + col0 + c];
    }
    red[g][c] = s;
    __syncthreads();
    float tot = 0.f;
    if (t < 16)
        for (int g2 = 0; g2 < 64; ++g2) tot += red[g2][t];
    __syncthreads();
    return tot;
}

// DPNR > 0 (th_wide_head_grads_dp): data parallel -- every sum this launch finishes (a dW block: two columns per finishing lane; db; a block of
// the conv bias) goes through dp_reduce() before it is stored and fed to Adam: the MEAN over the ranks, formed in rank order.  Loss and hit
// count stay this rank's own.  One slot per workgroup (blockIdx.x).
template <int DPNR>
__global__ __launch_bounds__(64 * WH_NW) void wide_grads_kernel(WideGradArgs a) {
    __shared__ float red[WH_NW][WH_TX][64][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r16 = lane & 15, g4 = lane >> 4;
    const int B = a.batch, K = a.k, C = a.c;
    DpTicket dp_tk{0u, 0u};
    if constexpr (DPNR > 0) dp_tk = dp_begin(a.dp);
    static_assert(DPNR == 0 || WH_TX == 2, "the exchange carries two values per finishing lane");
    if ((int)blockIdx.x < a.n_col) {
        const int col0 = blockIdx.x * 16 * WH_TX;
        // the finishing lanes (wave e < 4: class 4 g4 + e, column r16 of each tile) request their Adam state before anything else
        const int fcls = g4 * 4 + wave;
        float fp[WH_TX], fm[WH_TX], fv[WH_TX];
        const bool fin = wave < 4 && fcls < C && a.fw.p;
#pragma unroll
        for (int tx = 0; tx < WH_TX; ++tx) {
            const int col = col0 + tx * 16 + r16;
            const bool ok = fin && col < K;
            const long i = (long)fcls * K + col;
            fp[tx] = ok ? a.fw.p[i] : 0.f;
            fm[tx] = ok ? a.fw.m[i] : 0.f;
            fv[tx] = ok ? a.fw.v[i] : 0.f;
        }
        floatx4 accw[WH_TX];
#pragma unroll
        for (int tx = 0; tx < WH_TX; ++tx) accw[tx] = floatx4{0.f, 0.f, 0.f, 0.f};
        // A wave's 16-row blocks (wave, wave + 16, ...) are a serial chain of accumulations -- and, one block per trip, a serial chain of
        // memory round trips: ~2 us each under the launch's own load, 31 us for 4 096 rows whatever the workgroup count or the width of the
        // loads (r06: 196 workgroups, 16-byte loads, a one-block prefetch all measured 30 - 44 us).  The operands of WG_U blocks are
        // requested together, then their MFMAs run in the same block order: the same bits, a quarter of the round trips.
        constexpr int WG_U = 4;
        for (int c0 = 0; c0 < B; c0 += 16 * WH_NW * WG_U) {
            if (c0 + wave * 16 >= B) break;            // wave-uniform
            float at[WG_U][4], xv[WG_U][WH_TX][4];
#pragma unroll
            for (int u = 0; u < WG_U; ++u) {
                const int r0 = c0 + (u * WH_NW + wave) * 16;
#pragma unroll
                for (int s = 0; s < 4; ++s) {          // A[class r16][row 4 g4 + s] = dl[row][class]; B = X[row 4 g4 + s][col r16]
                    const int row = r0 + g4 * 4 + s;
                    at[u][s] = row < B ? a.dl[(long)row * 16 + r16] : 0.f;
#pragma unroll
                    for (int tx = 0; tx < WH_TX; ++tx) {
                        const int col = col0 + tx * 16 + r16;
                        xv[u][tx][s] = (row < B && col < K) ? a.x[(long)row * K + col] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < WG_U; ++u) {
                if (c0 + (u * WH_NW + wave) * 16 >= B) break;   // wave-uniform: blocks past the batch add nothing (not even a signed zero)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int tx = 0; tx < WH_TX; ++tx) accw[tx] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[u][s], xv[u][tx][s], accw[tx], 0, 0, 0);
            }
        }
#pragma unroll
        for (int tx = 0; tx < WH_TX; ++tx)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wave][tx][lane][i] = accw[tx][i];
        __syncthreads();
        float sums[WH_TX];
#pragma unroll
        for (int tx = 0; tx < WH_TX; ++tx) {
            sums[tx] = 0.f;
            if (wave < 4) {                            // deterministic cross-wave sum; wave e finishes class 4 g4 + e
                float sum = red[0][tx][lane][wave];
#pragma unroll
                for (int w = 1; w < WH_NW; ++w) sum += red[w][tx][lane][wave];
                sums[tx] = sum;
            }
        }
        if constexpr (DPNR > 0) {
            if (!dp_reduce<DPNR, WH_TX>(a.dp, dp_tk, (int)blockIdx.x, wave < 4 ? t : -1, sums)) return;   // nothing applied; the word is up
        }
        if (wave < 4) {
            const float step = a.fw.p ? adam_dev_step(a.fw) : 0.f;
#pragma unroll
            for (int tx = 0; tx < WH_TX; ++tx) {
                const int col = col0 + tx * 16 + r16;
                const float sum = sums[tx];
                if (fcls < C && col < K) {
                    const long i = (long)fcls * K + col;
                    a.dw[i] = sum;
                    if (a.fw.p) {                      // optim.rs:99-110 on the state requested at entry
                        const float gv = sum + a.fw.wd * fp[tx];
                        const float mv = a.fw.beta1 * fm[tx] + (1.0f - a.fw.beta1) * gv;
                        const float vv = a.fw.beta2 * fv[tx] + (1.0f - a.fw.beta2) * gv * gv;
                        a.fw.m[i] = mv;
                        a.fw.v[i] = vv;
                        a.fw.p[i] = fp[tx] - step * mv / (sqrtf(vv) + a.fw.eps);
                    }
                }
            }
        }
        return;
    }
    float(*cred)[16] = reinterpret_cast<float(*)[16]>(&red[0][0][0][0]);   // [64][16]
    if ((int)blockIdx.x == a.n_col) {
        const int64_t state0 = a.metrics ? a.state[0] : 0, state1 = a.metrics ? a.state[1] : 0;
        float dbs = slab_colsum16(a.dl, 16, B, 0, C, cred, t);          // tensor.rs:686-691
        if constexpr (DPNR > 0) {
            float pair[2] = {t < C ? dbs : 0.f, 0.f};
            if (!dp_reduce<DPNR, 2>(a.dp, dp_tk, (int)blockIdx.x, t < 256 ? t : -1, pair)) {
                // a peer's sums never came: nothing is applied, nothing is logged; this workgroup takes the step's tick back (optim.rs:84) --
                // only when the exchange failed in THIS launch: behind a dead communicator the chain launch has not ticked
                if (t == 0 && a.dp_tick && dp_tk.dead == 0u) atomicSub(a.dp_tick, 1);
                return;
            }
            dbs = pair[0];
        }
        if (t < C && a.db) {
            a.db[t] = dbs;
            if (a.fb.p) adam_update(a.fb.p, a.fb.m, a.fb.v, t, dbs, adam_dev_step(a.fb), a.fb.beta1, a.fb.beta2, a.fb.eps, a.fb.wd);
        }
        const float st = slab_colsum16(a.rowstat, 2, B, 0, 2, cred, t);       // {sum nll, hits}
        __shared__ float st2[2];
        if (t < 2) st2[t] = st;
        __syncthreads();
        if (t == 0) {
            const float l = st2[0] / (float)B;   // loss.rs:164
            a.loss[0] = l;
            if (a.ncorrect) a.ncorrect[0] = st2[1];
            if (a.metrics) {
                const int64_t slot = state0 < a.capacity ? state0 : state0 % a.capacity;
                a.metrics[2 * slot] = l;
                a.metrics[2 * slot + 1] = st2[1];
                a.state[0] = state0 + 1;
                a.state[1] = state1 + a.advance;
            }
        }
        return;
    }
    const int ch0 = ((int)blockIdx.x - a.n_col - 1) * 16;
    float cb = slab_colsum16(a.cbpart, a.conv_c, B, ch0, min(16, a.conv_c - ch0), cred, t);
    if constexpr (DPNR > 0) {
        float pair[2] = {(t < 16 && ch0 + t < a.conv_c) ? cb : 0.f, 0.f};
        if (!dp_reduce<DPNR, 2>(a.dp, dp_tk, (int)blockIdx.x, t < 256 ? t : -1, pair)) return;
        cb = pair[0];
    }
    if (t < 16 && ch0 + t < a.conv_c) {
        a.conv_gb[ch0 + t] = cb;
        if (a.fcb.p) adam_update(a.fcb.p, a.fcb.m, a.fcb.v, ch0 + t, cb, adam_dev_step(a.fcb), a.fcb.beta1, a.fcb.beta2, a.fcb.eps, a.fcb.wd);
    }
}

}  // namespace th

using namespace th;

static int wide_launch(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, const float *d_targets, int batch, int in_features,
                       int classes, float *d_loss, float *d_ncorrect, float *d_dx, float *d_dw, float *d_db, float *d_metrics,
                       int64_t metrics_capacity, int64_t *d_state, int64_t advance, int32_t *d_adam_tick, float *d_colsum_masked,
                       const th_wide_fuse *fuse) {
    TH_REQUIRE(ctx && d_x && d_w && d_targets && d_loss && d_dw, "th_linear_xent_wide: null argument");
    TH_REQUIRE(batch > 0 && batch <= 4096 && in_features > 0 && classes > 0 && classes <= 16,
               "th_linear_xent_wide: needs batch <= 4096, classes <= 16 (got %d, %d)", batch, classes);
    TH_REQUIRE(!d_metrics || (d_state && metrics_capacity > 0), "th_linear_xent_wide: metrics need d_state and a capacity");
    float *partial = nullptr;
    int kz = 0;
    if (int rc = linear_fwd_partials(ctx, d_x, d_w, batch, classes, in_features, &partial, &kz)) return rc;
    WideArgs a{d_x, d_w, d_bias, d_targets, partial, batch, in_features, classes, kz, d_loss, d_ncorrect, d_dx, d_dw, d_db,
               d_metrics, metrics_capacity, d_state, advance, d_adam_tick, ceil_div(in_features, 16 * WH_TX), d_colsum_masked,
               make_adam_dev(fuse ? &fuse->w : nullptr), make_adam_dev(fuse ? &fuse->b : nullptr), make_adam_dev(fuse ? &fuse->conv_b : nullptr),
               fuse ? fuse->d_conv_gb : nullptr, fuse ? fuse->conv_c : 0, fuse ? fuse->conv_hw : 0, fuse ? 1 : 0};
    hipLaunchKernelGGL(wide_head_kernel, dim3(a.n_col + 1), dim3(64 * WH_NW), 0, ctx->stream, a);
    TH_LAUNCH_CHECK();
    return th_free(ctx, partial);
}

extern "C" int th_linear_xent_wide_fused(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, const float *d_targets, int batch,
                                         int in_features, int classes, float *d_loss, float *d_ncorrect, float *d_dw, float *d_db,
                                         float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance, int32_t *d_adam_tick,
                                         float *d_colsum_masked, const th_wide_fuse *fuse) {
    TH_REQUIRE(fuse && d_adam_tick, "th_linear_xent_wide_fused: needs the fuse descriptor and Adam's counter block");
    TH_REQUIRE(!fuse->d_conv_gb || (d_colsum_masked && fuse->conv_c > 0 && fuse->conv_hw > 0 && (long)fuse->conv_c * fuse->conv_hw == in_features),
               "th_linear_xent_wide_fused: the conv bias finish needs the column sums and conv_c * conv_hw == in_features");
    TH_REQUIRE(!fuse->b.d_p || d_db, "th_linear_xent_wide_fused: a fused bias update needs d_db");
    return wide_launch(ctx, d_x, d_w, d_bias, d_targets, batch, in_features, classes, d_loss, d_ncorrect, nullptr, d_dw, d_db, d_metrics,
                       metrics_capacity, d_state, advance, d_adam_tick, d_colsum_masked, fuse);
}

extern "C" int th_linear_xent_wide_ex(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, const float *d_targets, int batch,
                                      int in_features, int classes, float *d_loss, float *d_ncorrect, float *d_dx, float *d_dw, float *d_db,
                                      float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance, int32_t *d_adam_tick,
                                      float *d_colsum_masked) {
    return wide_launch(ctx, d_x, d_w, d_bias, d_targets, batch, in_features, classes, d_loss, d_ncorrect, d_dx, d_dw, d_db, d_metrics,
                       metrics_capacity, d_state, advance, d_adam_tick, d_colsum_masked, nullptr);
}

extern "C" int th_linear_xent_wide(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, const float *d_targets, int batch,
                                   int in_features, int classes, float *d_loss, float *d_ncorrect, float *d_dx, float *d_dw, float *d_db,
                                   float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance, int32_t *d_adam_tick) {
    return wide_launch(ctx, d_x, d_w, d_bias, d_targets, batch, in_features, classes, d_loss, d_ncorrect, d_dx, d_dw, d_db, d_metrics,
                       metrics_capacity, d_state, advance, d_adam_tick, nullptr, nullptr);
}

#ifdef TH_PROFILE
extern "C" int th_debug_wide_prof(th_ctx *ctx, long long *h_out16) {
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(th::g_wh_prof), 16 * sizeof(long long)));
    return 0;
}
#endif

static int wide_grads_launch(th_comm *comm, th_ctx *ctx, const float *d_x, const float *d_dl, const float *d_rowstat, const float *d_cbpart, int batch,
                             int in_features, int classes, int conv_c, float *d_dw, float *d_db, float *d_conv_gb, float *d_loss,
                             float *d_ncorrect, float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance,
                             const th_adam_fuse *w_fuse, const th_adam_fuse *b_fuse, const th_adam_fuse *cb_fuse, int32_t *d_tick) {
    TH_REQUIRE(ctx && d_x && d_dl && d_rowstat && d_dw && d_loss, "th_wide_head_grads: null argument");
    TH_REQUIRE(batch > 0 && in_features > 0 && classes > 0 && classes <= 16, "th_wide_head_grads: needs classes <= 16 (got %d)", classes);
    TH_REQUIRE(!d_metrics || (d_state && metrics_capacity > 0), "th_wide_head_grads: metrics need d_state and a capacity");
    TH_REQUIRE((d_cbpart != nullptr) == (d_conv_gb != nullptr) && (!d_cbpart || conv_c > 0),
               "th_wide_head_grads: the conv bias needs d_cbpart, d_conv_gb and conv_c together");
    TH_REQUIRE(!(b_fuse && b_fuse->d_p) || d_db, "th_wide_head_grads: a fused bias update needs d_db");
    TH_REQUIRE(!(cb_fuse && cb_fuse->d_p) || d_conv_gb, "th_wide_head_grads: a fused conv bias update needs d_conv_gb");
    WideGradArgs a{d_x, d_dl, d_rowstat, d_cbpart, batch, in_features, classes, d_cbpart ? conv_c : 0, ceil_div(in_features, 16 * WH_TX),
                   d_dw, d_db, d_conv_gb, d_loss, d_ncorrect, d_metrics, metrics_capacity, d_state, advance,
                   make_adam_dev(w_fuse), make_adam_dev(b_fuse), make_adam_dev(cb_fuse), DpDev{}, nullptr};
    const int grid = a.n_col + 1 + ceil_div(a.conv_c, 16);
    if (comm) {
        const DpDev *dp = comm_dp_dev(comm);
        TH_REQUIRE(dp && th_wide_head_grads_dp_supported(comm, ctx, batch, in_features, classes, conv_c),
                   "th_wide_head_grads_dp: this communicator / shape cannot take the in-launch exchange (th_wide_head_grads_dp_supported)");
        a.dp = *dp;
        a.dp_tick = d_tick;
        if (dp->n_ranks <= 2) hipLaunchKernelGGL(wide_grads_kernel<2>, dim3(grid), dim3(64 * WH_NW), 0, ctx->stream, a);
        else if (dp->n_ranks <= 4) hipLaunchKernelGGL(wide_grads_kernel<4>, dim3(grid), dim3(64 * WH_NW), 0, ctx->stream, a);
        else hipLaunchKernelGGL(wide_grads_kernel<8>, dim3(grid), dim3(64 * WH_NW), 0, ctx->stream, a);
        comm_dp_count_launch(comm);
    } else {
        hipLaunchKernelGGL(wide_grads_kernel<0>, dim3(grid), dim3(64 * WH_NW), 0, ctx->stream, a);
    }
    TH_LAUNCH_CHECK();
    return 0;
}

extern "C" int th_wide_head_grads(th_ctx *ctx, const float *d_x, const float *d_dl, const float *d_rowstat, const float *d_cbpart, int batch,
                                  int in_features, int classes, int conv_c, float *d_dw, float *d_db, float *d_conv_gb, float *d_loss,
                                  float *d_ncorrect, float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance,
                                  const th_adam_fuse *w_fuse, const th_adam_fuse *b_fuse, const th_adam_fuse *cb_fuse) {
    return wide_grads_launch(nullptr, ctx, d_x, d_dl, d_rowstat, d_cbpart, batch, in_features, classes, conv_c, d_dw, d_db, d_conv_gb, d_loss, d_ncorrect,
                             d_metrics, metrics_capacity, d_state, advance, w_fuse, b_fuse, cb_fuse, nullptr);
}

// data parallel: the same launch with every finished sum reduced over the ranks before its store / Adam (csrc/dp_dev.h)
extern "C" int th_wide_head_grads_dp_supported(const th_comm *comm, th_ctx *ctx, int batch, int in_features, int classes, int conv_c) {
    const DpDev *dp = comm_dp_dev(comm);
    if (!dp || !ctx || dp->n_ranks < 2 || batch <= 0 || in_features <= 0 || classes <= 0 || classes > 16 || WH_TX != 2) return 0;
    const int grid = ceil_div(in_features, 16 * WH_TX) + 1 + ceil_div(conv_c, 16);
    if (grid > DP_MAX_SLOTS) return 0;
    if (comm_dp_sharing(comm) > 1) {       // ranks on ONE device: comm_dp_shared_fits (comm.hip)
        int per_cu = 0;
        const void *fn = dp->n_ranks <= 2 ? (const void *)wide_grads_kernel<2> : dp->n_ranks <= 4 ? (const void *)wide_grads_kernel<4> : (const void *)wide_grads_kernel<8>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * WH_NW, 0) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        if (!comm_dp_shared_fits(comm, grid, per_cu)) return 0;
    }
    return 1;
}

extern "C" int th_wide_head_grads_dp(th_comm *comm, th_ctx *ctx, const float *d_x, const float *d_dl, const float *d_rowstat, const float *d_cbpart,
                                     int batch, int in_features, int classes, int conv_c, float *d_dw, float *d_db, float *d_conv_gb, float *d_loss,
                                     float *d_ncorrect, float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance,
                                     const th_adam_fuse *w_fuse, const th_adam_fuse *b_fuse, const th_adam_fuse *cb_fuse, int32_t *d_tick) {
    TH_REQUIRE(comm, "th_wide_head_grads_dp: null communicator");
    return wide_grads_launch(comm, ctx, d_x, d_dl, d_rowstat, d_cbpart, batch, in_features, classes, conv_c, d_dw, d_db, d_conv_gb, d_loss, d_ncorrect,
                             d_metrics, metrics_capacity, d_state, advance, w_fuse, b_fuse, cb_fuse, d_tick);
}

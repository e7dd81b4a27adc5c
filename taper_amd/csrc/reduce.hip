// reduce.hip -- broadcast / reduction / layout kernels and the fused softmax
// cross-entropy (src/tensor.rs:544-591,636-770,890-1088; src/loss.rs:101-195,
// 271-290).  All HBM-bound; reductions use wave64 shuffles + a fixed-order LDS
// combine (no atomics -> deterministic).
#include "common.h"

TH_USES_DEVICE_ERRORS()

namespace th {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}

__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block-wide sum for 256 threads; result valid in thread 0
__device__ __forceinline__ float block_sum_256(float v, float *sh /* >= 4 floats */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    __syncthreads();
    return r;
}

// ---- 2-D transpose through a padded LDS tile (tensor.rs:544-566) ---------
template <bool ACCUM>
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int rows,
                                                        int cols) {
    // out[j*rows + i] (=|+=) in[i*cols + j]; 64x64 tile, 65-float pitch
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int r = ty + 4 * rr;
        const int i = i0 + r, j = j0 + tx;
        tile[r][tx] = (i < rows && j < cols) ? in[(long)i * cols + j] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int r = ty + 4 * rr;
        const int j = j0 + r, i = i0 + tx;
        if (i < rows && j < cols) {
            const long o = (long)j * rows + i;
            if (ACCUM) out[o] += tile[tx][r];
            else out[o] = tile[tx][r];
        }
    }
}

// ---- y[r,c] = x[r,c] + bias[c] (+relu)  (tensor.rs:658-663) -------------
__global__ __launch_bounds__(256) void bias_add_rows_kernel(const float *__restrict__ x, const float *__restrict__ bias,
                                                            float *__restrict__ y, long total, int cols, int relu) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = tid; i < total; i += stride) {
        float v = x[i] + bias[i % cols];
        if (relu) v = v > 0.f ? v : 0.f;
        y[i] = v;
    }
}

// ---- column sums of a [rows, cols] matrix --------------------------------
// grid.x covers columns in chunks of 64; each block has 4 waves striding the
// rows; lanes are consecutive columns (coalesced); fixed-order LDS combine.
// Tall inputs are cut into row slabs (blockIdx.y) whose sums land in out[slab][cols]; a second
// launch of the same kernel adds the slabs in order (deterministic, no atomics).
template <bool ACCUM, bool NEGATE>
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ g, float *__restrict__ out, int rows,
                                                     int cols, int rows_per_slab) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r_beg = blockIdx.y * rows_per_slab, r_end = min(rows, r_beg + rows_per_slab);
    // eight independent chains, their loads requested together (r06: two chains had two 256-byte requests per wave in flight -- [16 384][256]
    // took 20 us, 0.8 TB/s)
    constexpr int U = 8;
    float s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = 0.f;
    if (c < cols) {
        int r = r_beg + wave;
        for (; r + 4 * (U - 1) < r_end; r += 4 * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = g[(long)(r + 4 * u) * cols + c];
#pragma unroll
            for (int u = 0; u < U; ++u) s[u] += v[u];
        }
        for (; r < r_end; r += 4) s[0] += g[(long)r * cols + c];
    }
    part[wave][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (wave == 0 && c < cols) {
        float tot = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
        if (NEGATE) tot = -tot;
        float *o = out + (long)blockIdx.y * cols + c;
        if (ACCUM) *o += tot;
        else *o = tot;
    }
}

// Narrow matrices (cols <= 16: the [batch][classes] gradient of a classifier's logits -> its bias gradient): the kernel above would use
// `cols` of every wave's 64 lanes.  Here a workgroup's 256 threads are 16 row groups x 16 column slots over a contiguous run of rows
// (consecutive lanes = consecutive addresses within a row, consecutive row groups = consecutive rows), 8 loads in flight per thread, then
// the 16 row groups of a column are added in order; grid.x row slabs, a second launch of colsum_kernel adds the slabs.
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const float *__restrict__ g, float *__restrict__ part, int rows, int cols, int rows_per_slab) {
    __shared__ float sh[16][17];
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int r_beg = blockIdx.x * rows_per_slab, r_end = min(rows, r_beg + rows_per_slab);
    constexpr int U = 8;
    float s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = 0.f;
    if (c < cols) {
        int r = r_beg + rg;
        for (; r + 16 * (U - 1) < r_end; r += 16 * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = g[(long)(r + 16 * u) * cols + c];
#pragma unroll
            for (int u = 0; u < U; ++u) s[u] += v[u];
        }
        for (; r < r_end; r += 16) s[0] += g[(long)r * cols + c];
    }
    sh[rg][c] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (rg == 0 && c < cols) {
        float tot = sh[0][c];
#pragma unroll
        for (int i = 1; i < 16; ++i) tot += sh[i][c];
        part[(long)blockIdx.x * cols + c] = tot;
    }
}

template <bool ACCUM, bool NEGATE>
static int colsum_launch(th_ctx *ctx, const float *g, float *out, int rows, int cols) {
    if (cols <= 16 && rows >= 2048) {
        const int slabs = std::min(256, ceil_div(rows, 512)), rps = ceil_div(rows, slabs), ns = ceil_div(rows, rps);
        void *part = nullptr;
        if (th_malloc(ctx, (size_t)ns * cols * sizeof(float), &part)) return 1;
        hipLaunchKernelGGL(colsum_narrow_kernel, dim3(ns), dim3(256), 0, ctx->stream, g, (float *)part, rows, cols, rps);
        TH_LAUNCH_CHECK();
        hipLaunchKernelGGL((colsum_kernel<ACCUM, NEGATE>), dim3(1), dim3(256), 0, ctx->stream, (const float *)part, out, ns, cols, ns);
        TH_LAUNCH_CHECK();
        return th_free(ctx, part);
    }
    const int gx = ceil_div(cols, 64);
    int slabs = 1;
    if (rows >= 256 && gx < 128) {   // too few column blocks to fill the chip: split the rows (>= 32 per slab)
        slabs = ceil_div(rows, 32);
        const int cap = ceil_div(512, gx);
        if (slabs > cap) slabs = cap;
    }
    if (slabs <= 1) {
        hipLaunchKernelGGL((colsum_kernel<ACCUM, NEGATE>), dim3(gx), dim3(256), 0, ctx->stream, g, out, rows, cols, rows);
        TH_LAUNCH_CHECK();
        return 0;
    }
    const int rps = ceil_div(rows, slabs);
    slabs = ceil_div(rows, rps);
    void *part = nullptr;
    if (th_malloc(ctx, (size_t)slabs * cols * sizeof(float), &part)) return 1;
    hipLaunchKernelGGL((colsum_kernel<false, false>), dim3(gx, slabs), dim3(256), 0, ctx->stream, g, (float *)part, rows, cols, rps);
    TH_LAUNCH_CHECK();
    hipLaunchKernelGGL((colsum_kernel<ACCUM, NEGATE>), dim3(gx), dim3(256), 0, ctx->stream, (const float *)part, out, slabs, cols, slabs);
    TH_LAUNCH_CHECK();
    return th_free(ctx, part);
}

// ---- row-wise ops on [rows, cols]: one wave per row ----------------------
enum RowOp { ROW_SUM = 0, ROW_SUM_NEG_ACCUM = 1 };

template <int OP>
__global__ __launch_bounds__(256) void rowsum_kernel(const float *__restrict__ x, float *__restrict__ out, int rows,
                                                     int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += x[(long)row * cols + c];
    s = wave_sum(s);
    if (lane == 0) {
        if (OP == ROW_SUM) out[row] = s;
        else out[row] -= s;  // tensor.rs:752-764: grad_r[row] -= sum
    }
}

__global__ __launch_bounds__(256) void sub_rows_kernel(const float *__restrict__ x, const float *__restrict__ r,
                                                       float *__restrict__ y, long total, int cols) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = tid; i < total; i += stride) y[i] = x[i] - r[i / cols];
}

// gin[r,c] += gout[r]  (BY_ROW) or gout[c]
template <bool BY_ROW>
__global__ __launch_bounds__(256) void bcast_accum_kernel(const float *__restrict__ gout, float *__restrict__ gin,
                                                          long total, int cols) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = tid; i < total; i += stride) gin[i] += BY_ROW ? gout[i / cols] : gout[i % cols];
}

// ---- full reduction: two deterministic passes ---------------------------
__global__ __launch_bounds__(256) void sum_partial_kernel(const float *__restrict__ x, float *__restrict__ part, size_t n) {
    __shared__ float sh[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += x[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sum_final_kernel(const float *__restrict__ part, int nparts, float *__restrict__ out,
                                                        float divisor) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) out[0] = s / divisor;
}

// ---- max / first-argmax along rows or columns (tensor.rs:1042-1066) ------
// strict '>' starting from -inf: first maximum wins, NaN never wins, an
// all-NaN (or all -inf) slice reports value -inf / index 0.
__device__ __forceinline__ void argmax_combine(float &v, int &i, float ov, int oi) {
    // keep the larger value; on ties keep the smaller index (first max)
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

__global__ __launch_bounds__(256) void rowmax_kernel(const float *__restrict__ x, float *__restrict__ vmax,
                                                     float *__restrict__ imax, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < cols; c += 64) {
        const float v = x[(long)row * cols + c];
        if (v > best) {
            best = v;
            bi = c;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        argmax_combine(best, bi, ov, oi);
    }
    if (lane == 0) {
        if (vmax) vmax[row] = best;
        if (imax) imax[row] = (bi == 0x7fffffff) ? 0.f : (float)bi;
    }
}

__global__ __launch_bounds__(256) void colmax_kernel(const float *__restrict__ x, float *__restrict__ vmax,
                                                     float *__restrict__ imax, int rows, int cols) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float best = -INFINITY;
    int bi = 0;
    for (int r = 0; r < rows; ++r) {
        const float v = x[(long)r * cols + c];
        if (v > best) {
            best = v;
            bi = r;
        }
    }
    if (vmax) vmax[c] = best;
    if (imax) imax[c] = (float)bi;
}

// ---- global max (tensor.rs:1072-1083): `max_by(partial_cmp)` keeps the LAST of equal maxima; a NaN anywhere makes the reference's
// `partial_cmp(..).unwrap()` panic (flag[0] = 1 here, the host raises).  Two passes: per-block (value, index) pairs, then one block.
__device__ __forceinline__ void lastmax_combine(float &v, long &i, float ov, long oi) {
    if (oi >= 0 && (i < 0 || ov > v || (ov == v && oi > i))) {
        v = ov;
        i = oi;
    }
}

__device__ __forceinline__ void lastmax_block(float &best, long &bi, int &nan_seen, float *shv, long *shi, int *shn) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const long oi = __shfl_xor(bi, off, 64);
        nan_seen |= __shfl_xor(nan_seen, off, 64);
        lastmax_combine(best, bi, ov, oi);
    }
    if (lane == 0) {
        shv[wave] = best;
        shi[wave] = bi;
        shn[wave] = nan_seen;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            lastmax_combine(best, bi, shv[w], shi[w]);
            nan_seen |= shn[w];
        }
    }
}

__global__ __launch_bounds__(256) void global_max_partial_kernel(const float *__restrict__ x, long n, float *__restrict__ pv, long *__restrict__ pi,
                                                                 int *__restrict__ pn) {
    __shared__ float shv[4];
    __shared__ long shi[4];
    __shared__ int shn[4];
    float best = 0.f;
    long bi = -1;
    int nan_seen = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        nan_seen |= (v != v) ? 1 : 0;
        if (bi < 0 || v >= best) {   // ascending i inside a thread: >= keeps the later one
            best = v;
            bi = i;
        }
    }
    lastmax_block(best, bi, nan_seen, shv, shi, shn);
    if (threadIdx.x == 0) {
        pv[blockIdx.x] = best;
        pi[blockIdx.x] = bi;
        pn[blockIdx.x] = nan_seen;
    }
}

__global__ __launch_bounds__(256) void global_max_final_kernel(const float *__restrict__ pv, const long *__restrict__ pi, const int *__restrict__ pn,
                                                               int nparts, float *__restrict__ vmax, float *__restrict__ imax, int *__restrict__ flag) {
    __shared__ float shv[4];
    __shared__ long shi[4];
    __shared__ int shn[4];
    float best = 0.f;
    long bi = -1;
    int nan_seen = 0;
    for (int p = threadIdx.x; p < nparts; p += 256) {
        lastmax_combine(best, bi, pv[p], pi[p]);
        nan_seen |= pn[p];
    }
    lastmax_block(best, bi, nan_seen, shv, shi, shn);
    if (threadIdx.x == 0) {
        if (vmax) vmax[0] = bi < 0 ? 0.f : best;          // empty input: unwrap_or((0.0, 0))
        if (imax) imax[0] = bi < 0 ? 0.f : (float)bi;      // `max_idx as f32`
        if (flag) flag[0] = nan_seen;
    }
}

// ---- fused log-softmax + NLL + argmax (loss.rs:101-195, 271-290) ---------
// A row is owned by LPR lanes (16 when classes <= 16 -- four rows per wave for
// the 10-class MNIST head -- else a whole wave); reductions are xor-shuffles
// inside the lane group.  Per row: first-max argmax (tensor.rs:1062), logp,
// the NLL term, the accuracy hit, and optionally the unit-upstream gradient
// (softmax - onehot) * (1/B) so that backward() from the loss needs no launch.
struct XentArgs {
    const float *logits, *targets;
    int batch, classes;
    float *logp, *argmax_out, *dunit;   // nullable
};

__device__ __forceinline__ long target_class(float tf) {  // Rust `as usize`: saturating, NaN -> 0
    return (tf >= 0.f) ? (long)fminf(tf, 2147483520.f) : 0;
}

// returns (nll, hit) of `row`, valid in the group's first lane
template <int LPR>
__device__ __forceinline__ void xent_row(const XentArgs &a, int row, int sub, float &nll, float &hit) {
    const float *x = a.logits + (long)row * a.classes;
    const float tf = a.targets ? a.targets[row] : 0.f;   // requested together with the logits
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = sub; c < a.classes; c += LPR) {
        const float v = x[c];
        if (v > best) {
            best = v;
            bi = c;
        }
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        argmax_combine(best, bi, ov, oi);
    }
    if (bi == 0x7fffffff) bi = 0;
    float s = 0.f;
    for (int c = sub; c < a.classes; c += LPR) s += expf(x[c] - best);
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float log_sum = logf(s);
    const long cls = target_class(tf);
    const float inv_b = 1.0f / (float)a.batch;
    float my_nll = 0.f;
    for (int c = sub; c < a.classes; c += LPR) {
        const float lp = (x[c] - best) - log_sum;  // loss.rs:117-125
        if (a.logp) a.logp[(long)row * a.classes + c] = lp;
        if (c == cls) my_nll = -lp;
        if (a.dunit) {  // loss.rs:174-191 with g0 = 1: (exp(logp) - onehot) * (1 / B)
            float gv = expf(lp);
            if (c == cls) gv -= 1.0f;
            a.dunit[(long)row * a.classes + c] = gv * inv_b;
        }
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) my_nll += __shfl_xor(my_nll, off, 64);  // exactly one lane holds it
    nll = (cls >= a.classes) ? NAN : my_nll;  // the reference panics (loss.rs:161): NaN + a note the next wait turns into an error
    if (cls >= a.classes && sub == 0) raise_target_oob(cls, a.classes);
    hit = (fabsf((float)bi - tf) < 1e-6f) ? 1.f : 0.f;  // loss.rs:283
    if (sub == 0 && a.argmax_out) a.argmax_out[row] = (float)bi;
}

struct StepLog {  // th_log_step folded into the loss kernel (nullable metrics)
    float *metrics;
    int64_t capacity;
    int64_t *state;
    int64_t advance;
    int32_t *adam_tick;  // nullable: Adam's t += 1 (optim.rs:84) for a step whose updates run fused
};

__device__ __forceinline__ void step_log(const StepLog &lg, float loss, float ncorrect) {
    if (lg.adam_tick) lg.adam_tick[0] += 1;
    if (!lg.metrics) return;
    const int64_t s = lg.state[0] % lg.capacity;
    lg.metrics[2 * s] = loss;
    lg.metrics[2 * s + 1] = ncorrect;
    lg.state[0] += 1;
    lg.state[1] += lg.advance;
}

// whole batch in ONE workgroup (batch <= 1024): rows -> block reduce -> loss, count, log
template <int LPR>
__global__ __launch_bounds__(1024) void softmax_xent_fused_kernel(XentArgs a, float *__restrict__ loss, float *__restrict__ ncorrect,
                                                                  StepLog lg) {
    __shared__ float sh_n[16], sh_h[16];
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR, ngrp = blockDim.x / LPR;
    float acc_n = 0.f, acc_h = 0.f;
    for (int row = grp; row < a.batch; row += ngrp) {  // whole lane groups iterate together
        float nll, hit;
        xent_row<LPR>(a, row, sub, nll, hit);
        if (sub == 0) {
            acc_n += nll;
            acc_h += hit;
        }
    }
    acc_n = wave_sum(acc_n);
    acc_h = wave_sum(acc_h);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) {
        sh_n[wave] = acc_n;
        sh_h[wave] = acc_h;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float n = 0.f, h = 0.f;
        for (int w = 0; w < nw; ++w) {
            n += sh_n[w];
            h += sh_h[w];
        }
        const float l = n / (float)a.batch;  // loss.rs:164
        loss[0] = l;
        if (ncorrect) ncorrect[0] = h;
        step_log(lg, l, h);
    }
}

// large batches: rows spread over the chip, per-row terms to scratch, then a finish kernel
template <int LPR>
__global__ __launch_bounds__(256) void softmax_xent_rows_kernel(XentArgs a, float *__restrict__ row_nll, float *__restrict__ row_hit) {
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR, ngrp = 256 / LPR;
    const int row = blockIdx.x * ngrp + grp;
    if (row >= a.batch) return;  // whole lane groups exit together; shuffles stay inside a group
    float nll, hit;
    xent_row<LPR>(a, row, sub, nll, hit);
    if (sub == 0) {
        if (row_nll) row_nll[row] = nll;
        if (row_hit) row_hit[row] = hit;
    }
}

__global__ __launch_bounds__(256) void xent_finish_kernel(const float *__restrict__ row_nll, const float *__restrict__ row_hit,
                                                          int batch, float *__restrict__ loss, float *__restrict__ ncorrect,
                                                          StepLog lg) {
    __shared__ float sh[4];
    float s = 0.f, h = 0.f;
    for (int i = threadIdx.x; i < batch; i += 256) {
        s += row_nll[i];
        h += row_hit[i];
    }
    s = block_sum_256(s, sh);
    h = block_sum_256(h, sh);
    if (threadIdx.x == 0) {
        const float l = s / (float)batch;  // loss.rs:164
        loss[0] = l;
        if (ncorrect) ncorrect[0] = h;
        step_log(lg, l, h);
    }
}

// dlogits[i,c] += (exp(logp[i,c]) - [c == t_i]) * (g0 / B)   (loss.rs:174-191)
__global__ __launch_bounds__(256) void softmax_xent_bwd_kernel(const float *__restrict__ logp, const float *__restrict__ targets,
                                                               const float *__restrict__ g0, int batch, int classes,
                                                               float *__restrict__ dlogits, int accumulate) {
    const long total = (long)batch * classes;
    const float scale = g0[0] / (float)batch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int row = (int)(i / classes), c = (int)(i % classes);
        const float tf = targets[row];
        const long cls = target_class(tf);
        float gval = expf(logp[i]);
        if (c == cls) gval -= 1.0f;
        dlogits[i] = accumulate ? dlogits[i] + gval * scale : gval * scale;
    }
}

__global__ __launch_bounds__(256) void accuracy_count_kernel(const float *__restrict__ am, const float *__restrict__ t, int n,
                                                             float *__restrict__ out) {
    __shared__ float sh[4];
    float h = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) h += (fabsf(am[i] - t[i]) < 1e-6f) ? 1.f : 0.f;
    h = block_sum_256(h, sh);
    if (threadIdx.x == 0) out[0] = h;
}

// rows x cols block copy between two pitched fp32 matrices (slice_channels / cat, nn.rs:862-1014)
__global__ __launch_bounds__(256) void copy2d_kernel(const float *__restrict__ src, float *__restrict__ dst, long total, int cols,
                                                     long src_ld, long dst_ld) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / cols, c = i - r * cols;
        dst[r * dst_ld + c] = src[r * src_ld + c];
    }
}

}  // namespace th

using namespace th;

extern "C" {

int th_transpose2d(th_ctx *ctx, const float *d_in, float *d_out, int rows, int cols) {
    TH_REQUIRE(ctx && d_in && d_out && rows >= 0 && cols >= 0, "th_transpose2d: bad argument");
    if (rows == 0 || cols == 0) return 0;
    hipLaunchKernelGGL(transpose_kernel<false>, dim3(ceil_div(cols, 64), ceil_div(rows, 64)), dim3(256), 0, ctx->stream,
                       d_in, d_out, rows, cols);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_transpose2d_bwd(th_ctx *ctx, const float *d_gout, float *d_gin, int rows, int cols) {
    // gin[i*cols + j] += gout[j*rows + i]: the transpose of a [cols, rows] matrix, accumulated
    TH_REQUIRE(ctx && d_gout && d_gin && rows >= 0 && cols >= 0, "th_transpose2d_bwd: bad argument");
    if (rows == 0 || cols == 0) return 0;
    hipLaunchKernelGGL(transpose_kernel<true>, dim3(ceil_div(rows, 64), ceil_div(cols, 64)), dim3(256), 0, ctx->stream,
                       d_gout, d_gin, cols, rows);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_bias_add_rows(th_ctx *ctx, const float *d_x, const float *d_bias, float *d_y, int rows, int cols, int relu) {
    TH_REQUIRE(ctx && d_x && d_bias && d_y && rows >= 0 && cols >= 0, "th_bias_add_rows: bad argument");
    const long total = (long)rows * cols;
    if (total == 0) return 0;
    hipLaunchKernelGGL(bias_add_rows_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, ctx->stream, d_x, d_bias, d_y, total,
                       cols, relu);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_colsum_accum(th_ctx *ctx, const float *d_g, float *d_gb, int rows, int cols) {
    TH_REQUIRE(ctx && d_g && d_gb && rows >= 0 && cols >= 0, "th_colsum_accum: bad argument");
    if (cols == 0) return 0;
    return colsum_launch<true, false>(ctx, d_g, d_gb, rows, cols);
}

int th_colsum(th_ctx *ctx, const float *d_x, float *d_y, int rows, int cols) {
    TH_REQUIRE(ctx && d_x && d_y && rows >= 0 && cols >= 0, "th_colsum: bad argument");
    if (cols == 0) return 0;
    return colsum_launch<false, false>(ctx, d_x, d_y, rows, cols);
}

int th_sub_rows(th_ctx *ctx, const float *d_x, const float *d_r, float *d_y, int rows, int cols) {
    TH_REQUIRE(ctx && d_x && d_r && d_y && rows >= 0 && cols >= 0, "th_sub_rows: bad argument");
    const long total = (long)rows * cols;
    if (total == 0) return 0;
    hipLaunchKernelGGL(sub_rows_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, ctx->stream, d_x, d_r, d_y, total, cols);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_rowsum_neg_accum(th_ctx *ctx, const float *d_g, float *d_gr, int rows, int cols) {
    TH_REQUIRE(ctx && d_g && d_gr && rows >= 0 && cols >= 0, "th_rowsum_neg_accum: bad argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(rowsum_kernel<ROW_SUM_NEG_ACCUM>, dim3(ceil_div(rows, 4)), dim3(256), 0, ctx->stream, d_g, d_gr, rows, cols);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_rowsum(th_ctx *ctx, const float *d_x, float *d_y, int rows, int cols) {
    TH_REQUIRE(ctx && d_x && d_y && rows >= 0 && cols >= 0, "th_rowsum: bad argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(rowsum_kernel<ROW_SUM>, dim3(ceil_div(rows, 4)), dim3(256), 0, ctx->stream, d_x, d_y, rows, cols);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_rowsum_bwd(th_ctx *ctx, const float *d_gout, float *d_gin, int rows, int cols) {
    TH_REQUIRE(ctx && d_gout && d_gin && rows >= 0 && cols >= 0, "th_rowsum_bwd: bad argument");
    const long total = (long)rows * cols;
    if (total == 0) return 0;
    hipLaunchKernelGGL(bcast_accum_kernel<true>, dim3(ew_grid(total, 256)), dim3(256), 0, ctx->stream, d_gout, d_gin, total, cols);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_colsum_bwd(th_ctx *ctx, const float *d_gout, float *d_gin, int rows, int cols) {
    TH_REQUIRE(ctx && d_gout && d_gin && rows >= 0 && cols >= 0, "th_colsum_bwd: bad argument");
    const long total = (long)rows * cols;
    if (total == 0) return 0;
    hipLaunchKernelGGL(bcast_accum_kernel<false>, dim3(ew_grid(total, 256)), dim3(256), 0, ctx->stream, d_gout, d_gin, total, cols);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_sum_all(th_ctx *ctx, const float *d_x, float *d_out1, size_t n, float divisor) {
    TH_REQUIRE(ctx && d_out1 && (n == 0 || d_x), "th_sum_all: bad argument");
    int nparts = (int)((n + 256 * 16 - 1) / (256 * 16));
    if (nparts < 1) nparts = 1;
    if (nparts > 1024) nparts = 1024;
    void *part = nullptr;
    if (th_malloc(ctx, nparts * sizeof(float), &part)) return 1;
    hipLaunchKernelGGL(sum_partial_kernel, dim3(nparts), dim3(256), 0, ctx->stream, d_x, (float *)part, n);
    TH_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, ctx->stream, (const float *)part, nparts, d_out1, divisor);
    TH_LAUNCH_CHECK();
    return th_free(ctx, part);
}

}  // extern "C" (reopened below)

namespace th {

// ---- sum(dim) / max(dim) for ANY rank (<= 4), restating the reference's per-element index arithmetic (tensor.rs:917-937, 960-994, 1042-1066):
// off the training step (the loss reduces 2-D tensors over their first or last dimension: the wave-shuffle kernels above), here so that the
// host mirror accepts what the reference accepts -- and gives what it gives, quirk Q14 included: max(dim)'s output index only steps its
// multiplier for dimensions IN FRONT of dim, so on rank > 2 several (row, column) pairs share an output element; sum's backward skips a
// dimension when its position is not below the output's ELEMENT count.
struct DimShape {
    int64_t shape[4];
    int ndim, dim, keepdim;
    int64_t n, out_n;
};

// forward: one thread per output element, its terms added in the order the reference meets them (ascending input index): bit-exact
__global__ __launch_bounds__(256) void sum_dim_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t outer, int64_t d, int64_t inner) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= outer * inner) return;
    const int64_t a = o / inner, c = o - a * inner;
    const float *p = x + a * d * inner + c;
    float s = 0.f;
    for (int64_t k = 0; k < d; ++k) s += p[k * inner];
    y[o] = s;
}

// backward: gin[i] += gout[min(out_idx(i), len - 1)], out_idx by tensor.rs:966-990 literally
__global__ __launch_bounds__(256) void sum_dim_bwd_kernel(const float *__restrict__ gout, float *__restrict__ gin, DimShape q) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= q.n) return;
    int64_t idx = i, out_idx = 0, mult = 1;
    for (int j = q.ndim - 1; j >= 0; --j) {
        const int64_t coord = idx % q.shape[j];
        idx /= q.shape[j];
        if (j != q.dim) {
            const int out_j = (j > q.dim && !q.keepdim) ? j - 1 : j;
            if ((int64_t)out_j < q.out_n) {                // (the reference compares with gout.len(): the element count)
                out_idx += coord * mult;
                mult *= q.shape[j];                         // (every branch of tensor.rs:976-988 with j != d is in_shape[j])
            }
        }
    }
    gin[i] += gout[out_idx < q.out_n - 1 ? out_idx : q.out_n - 1];
}

// max(dim): every input element offers {value, its index} to the output element the reference's arithmetic names; the winner of an output
// element is the largest value, the FIRST of equals (strict > in ascending input order), NaN and -inf never win (nothing is > them first):
// one 64-bit atomicMax on {order-preserving bits of the value, ~index} is exactly that, whatever order the threads run in.
__device__ __forceinline__ unsigned long long max_key(float v, int64_t i) {
    // (-0.0 and +0.0 compare EQUAL under the reference's strict > (tensor.rs:1062): the first of them wins -- both take +0.0's key; the
    // value itself is decoded from x[index], sign included)
    const unsigned b = __float_as_uint(v + 0.0f);
    const unsigned ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)ord << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
}
__device__ __forceinline__ int64_t max_out_index(int64_t i, const DimShape &q, int64_t *dim_idx) {
    int64_t idx = i, out_idx = 0, mult = 1;
    for (int j = q.ndim - 1; j >= 0; --j) {
        const int64_t coord = idx % q.shape[j];
        idx /= q.shape[j];
        if (j == q.dim) *dim_idx = coord;
        else {
            out_idx += coord * mult;
            mult *= (j < q.dim) ? q.shape[j] : 1;           // tensor.rs:1056 (Q14)
        }
    }
    return out_idx < q.out_n - 1 ? out_idx : q.out_n - 1;
}
__global__ __launch_bounds__(256) void max_dim_offer_kernel(const float *__restrict__ x, unsigned long long *__restrict__ keys, DimShape q) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= q.n) return;
    const float v = x[i];
    if (!(v > -INFINITY)) return;                           // NaN, -inf: `data[i] > max_values[..]` is false against the initial -inf
    int64_t dim_idx = 0;
    const int64_t o = max_out_index(i, q, &dim_idx);
    atomicMax(&keys[o], max_key(v, i));
}
__global__ __launch_bounds__(256) void max_dim_decode_kernel(const float *__restrict__ x, const unsigned long long *__restrict__ keys, float *__restrict__ vals,
                                                             float *__restrict__ idxs, DimShape q) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= q.out_n) return;
    const unsigned long long k = keys[o];
    if (k == 0ull) {                                        // nothing won: the initial {-inf, 0.0}
        vals[o] = -INFINITY;
        if (idxs) idxs[o] = 0.f;
        return;
    }
    const int64_t i = (int64_t)(0xffffffffu - (unsigned)(k & 0xffffffffull));
    int64_t dim_idx = 0;
    (void)max_out_index(i, q, &dim_idx);
    vals[o] = x[i];
    if (idxs) idxs[o] = (float)dim_idx;
}

static int dim_shape(const int64_t *shape, int ndim, int dim, int keepdim, DimShape *q, const char *who) {
    TH_REQUIRE(shape && ndim >= 1 && ndim <= 4 && dim >= 0 && dim < ndim, "%s: 1..4 dimensions, 0 <= dim < ndim", who);
    q->ndim = ndim; q->dim = dim; q->keepdim = keepdim; q->n = 1; q->out_n = 1;
    for (int j = 0; j < ndim; ++j) {
        TH_REQUIRE(shape[j] >= 1, "%s: empty dimension", who);
        q->shape[j] = shape[j];
        q->n *= shape[j];
        if (j != dim) q->out_n *= shape[j];
    }
    TH_REQUIRE(q->n < (1LL << 32), "%s: fewer than 2^32 elements", who);
    return 0;
}

}  // namespace th

extern "C" {

int th_sum_dim(th_ctx *ctx, const float *d_x, float *d_y, const int64_t *shape, int ndim, int dim) {
    TH_REQUIRE(ctx && d_x && d_y, "th_sum_dim: null argument");
    th::DimShape q;
    if (int rc = th::dim_shape(shape, ndim, dim, 0, &q, "th_sum_dim")) return rc;
    int64_t outer = 1, inner = 1;
    for (int j = 0; j < dim; ++j) outer *= shape[j];
    for (int j = dim + 1; j < ndim; ++j) inner *= shape[j];
    hipLaunchKernelGGL(th::sum_dim_kernel, dim3(th::ceil_div(outer * inner, 256)), dim3(256), 0, ctx->stream, d_x, d_y, outer, shape[dim], inner);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_sum_dim_bwd(th_ctx *ctx, const float *d_gout, float *d_gin, const int64_t *shape, int ndim, int dim, int keepdim) {
    TH_REQUIRE(ctx && d_gout && d_gin, "th_sum_dim_bwd: null argument");
    th::DimShape q;
    if (int rc = th::dim_shape(shape, ndim, dim, keepdim, &q, "th_sum_dim_bwd")) return rc;
    hipLaunchKernelGGL(th::sum_dim_bwd_kernel, dim3(th::ceil_div(q.n, 256)), dim3(256), 0, ctx->stream, d_gout, d_gin, q);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_max_dim(th_ctx *ctx, const float *d_x, float *d_max, float *d_argmax_f32, const int64_t *shape, int ndim, int dim) {
    TH_REQUIRE(ctx && d_x && d_max, "th_max_dim: null argument");
    th::DimShape q;
    if (int rc = th::dim_shape(shape, ndim, dim, 1, &q, "th_max_dim")) return rc;
    void *keys = nullptr;
    if (th_malloc(ctx, (size_t)q.out_n * sizeof(unsigned long long), &keys)) return 1;
    TH_HIP(hipMemsetAsync(keys, 0, (size_t)q.out_n * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(th::max_dim_offer_kernel, dim3(th::ceil_div(q.n, 256)), dim3(256), 0, ctx->stream, d_x, (unsigned long long *)keys, q);
    hipLaunchKernelGGL(th::max_dim_decode_kernel, dim3(th::ceil_div(q.out_n, 256)), dim3(256), 0, ctx->stream, d_x, (const unsigned long long *)keys, d_max,
                       d_argmax_f32, q);
    TH_LAUNCH_CHECK();
    return th_free(ctx, keys);
}

int th_rowmax(th_ctx *ctx, const float *d_x, float *d_max, float *d_argmax_f32, int rows, int cols) {
    TH_REQUIRE(ctx && d_x && rows >= 0 && cols >= 0, "th_rowmax: bad argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(rowmax_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, ctx->stream, d_x, d_max, d_argmax_f32, rows, cols);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_colmax(th_ctx *ctx, const float *d_x, float *d_max, float *d_argmax_f32, int rows, int cols) {
    TH_REQUIRE(ctx && d_x && rows >= 0 && cols >= 0, "th_colmax: bad argument");
    if (cols == 0) return 0;
    hipLaunchKernelGGL(colmax_kernel, dim3(ceil_div(cols, 256)), dim3(256), 0, ctx->stream, d_x, d_max, d_argmax_f32, rows, cols);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_global_max(th_ctx *ctx, const float *d_x, size_t n, float *d_max, float *d_argmax_f32, int *d_nan_flag) {
    TH_REQUIRE(ctx && (d_x || n == 0), "th_global_max: bad argument");
    int nparts = (int)((n + 256 * 16 - 1) / (256 * 16));
    nparts = nparts < 1 ? 1 : (nparts > 1024 ? 1024 : nparts);
    void *ws = nullptr;
    if (th_malloc(ctx, (size_t)nparts * 16, &ws)) return 1;
    long *pi = (long *)ws;
    float *pv = (float *)(pi + nparts);
    int *pn = (int *)(pv + nparts);
    hipLaunchKernelGGL(global_max_partial_kernel, dim3(nparts), dim3(256), 0, ctx->stream, d_x, (long)n, pv, pi, pn);
    TH_LAUNCH_CHECK();
    hipLaunchKernelGGL(global_max_final_kernel, dim3(1), dim3(256), 0, ctx->stream, (const float *)pv, (const long *)pi, (const int *)pn, nparts, d_max,
                       d_argmax_f32, d_nan_flag);
    TH_LAUNCH_CHECK();
    return th_free(ctx, ws);
}

int th_softmax_xent_fwd(th_ctx *ctx, const float *d_logits, const float *d_targets, int batch, int classes, float *d_logp,
                        float *d_loss, float *d_argmax, float *d_ncorrect, float *d_dlogits_unit, float *d_metrics,
                        int64_t metrics_capacity, int64_t *d_state, int64_t advance, int32_t *d_adam_tick) {
    TH_REQUIRE(ctx && d_logits && d_targets && d_loss && batch > 0 && classes > 0, "th_softmax_xent_fwd: bad argument");
    TH_REQUIRE(!d_metrics || (d_state && metrics_capacity > 0), "th_softmax_xent_fwd: metrics need d_state and a capacity");
    XentArgs a{d_logits, d_targets, batch, classes, d_logp, d_argmax, d_dlogits_unit};
    StepLog lg{d_metrics, metrics_capacity, d_state, advance, d_adam_tick};
    const bool narrow = classes <= 16;
    if (batch <= 1024) {  // one launch: rows, reduction, loss, count and the step log
        const int lpr = narrow ? 16 : 64;
        int threads = ((batch * lpr + 63) / 64) * 64;
        threads = threads < 64 ? 64 : (threads > 1024 ? 1024 : threads);
        if (narrow) hipLaunchKernelGGL(softmax_xent_fused_kernel<16>, dim3(1), dim3(threads), 0, ctx->stream, a, d_loss, d_ncorrect, lg);
        else hipLaunchKernelGGL(softmax_xent_fused_kernel<64>, dim3(1), dim3(threads), 0, ctx->stream, a, d_loss, d_ncorrect, lg);
        TH_LAUNCH_CHECK();
        return 0;
    }
    void *tmp = nullptr;
    if (th_malloc(ctx, 2 * (size_t)batch * sizeof(float), &tmp)) return 1;
    float *row_nll = (float *)tmp, *row_hit = row_nll + batch;
    if (narrow) hipLaunchKernelGGL(softmax_xent_rows_kernel<16>, dim3(ceil_div(batch, 16)), dim3(256), 0, ctx->stream, a, row_nll, row_hit);
    else hipLaunchKernelGGL(softmax_xent_rows_kernel<64>, dim3(ceil_div(batch, 4)), dim3(256), 0, ctx->stream, a, row_nll, row_hit);
    TH_LAUNCH_CHECK();
    hipLaunchKernelGGL(xent_finish_kernel, dim3(1), dim3(256), 0, ctx->stream, (const float *)row_nll, (const float *)row_hit, batch,
                       d_loss, d_ncorrect, lg);
    TH_LAUNCH_CHECK();
    return th_free(ctx, tmp);
}

int th_softmax_xent_bwd(th_ctx *ctx, const float *d_logp, const float *d_targets, const float *d_g0, int batch, int classes,
                        float *d_dlogits, int accumulate) {
    TH_REQUIRE(ctx && d_logp && d_targets && d_g0 && d_dlogits && batch > 0 && classes > 0, "th_softmax_xent_bwd: bad argument");
    const long total = (long)batch * classes;
    hipLaunchKernelGGL(softmax_xent_bwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, ctx->stream, d_logp, d_targets, d_g0,
                       batch, classes, d_dlogits, accumulate);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_log_softmax_fwd(th_ctx *ctx, const float *d_x, float *d_logp, int rows, int cols) {
    TH_REQUIRE(ctx && d_x && d_logp && rows >= 0 && cols > 0, "th_log_softmax_fwd: bad argument");
    if (rows == 0) return 0;
    XentArgs a{d_x, nullptr, rows, cols, d_logp, nullptr, nullptr};
    if (cols <= 16)
        hipLaunchKernelGGL(softmax_xent_rows_kernel<16>, dim3(ceil_div(rows, 16)), dim3(256), 0, ctx->stream, a, (float *)nullptr, (float *)nullptr);
    else
        hipLaunchKernelGGL(softmax_xent_rows_kernel<64>, dim3(ceil_div(rows, 4)), dim3(256), 0, ctx->stream, a, (float *)nullptr, (float *)nullptr);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_accuracy_count(th_ctx *ctx, const float *d_argmax, const float *d_targets, int n, float *d_ncorrect) {
    TH_REQUIRE(ctx && d_argmax && d_targets && d_ncorrect && n >= 0, "th_accuracy_count: bad argument");
    hipLaunchKernelGGL(accuracy_count_kernel, dim3(1), dim3(256), 0, ctx->stream, d_argmax, d_targets, n, d_ncorrect);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_copy2d(th_ctx *ctx, const float *d_src, float *d_dst, int64_t rows, int cols, int64_t src_ld, int64_t dst_ld) {
    TH_REQUIRE(ctx && rows >= 0 && cols >= 0 && src_ld >= cols && dst_ld >= cols, "th_copy2d: bad geometry");
    const long total = (long)rows * cols;
    if (total == 0) return 0;
    TH_REQUIRE(d_src && d_dst, "th_copy2d: null argument");
    hipLaunchKernelGGL(copy2d_kernel, dim3(ew_grid((size_t)total, 256)), dim3(256), 0, ctx->stream, d_src, d_dst, total, cols, (long)src_ld,
                       (long)dst_ld);
    TH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

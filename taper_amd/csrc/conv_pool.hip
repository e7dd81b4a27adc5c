// conv_pool.hip -- direct 3x3 convolution, 1x1 convolution, max/avg pooling
// and the NCHW bias kernels (src/tensor.rs:1221-2081).
//
// conv3x3 (this file): VALU direct convolution, the path for C_in < 8 (conv1 of both CNNs: K = 9,
// HBM-bound) and for planes the matrix-core kernel does not take; layers with C_in >= 8 run the
// direct convolution of conv_mfma.hip (measured 1.8-3.7x faster on the reference CNN's conv2-5;
// the north_star's "MFMA only for the dense GEMMs" plan was written before that measurement --
// DESIGN.md section 3).  One workgroup owns
// IMG images x 16 output channels; the input planes (+1-pixel halo) of 8
// input channels at a time are staged in LDS; each thread keeps a 2x4 pixel
// tile x 8 channels of accumulators in registers; the weights are re-laid-out once
// per call into [ci][kh][kw][co] so that every weight read in the inner loop
// is wave-uniform and is served by scalar loads (s_load_dwordx*), leaving the
// LDS pipe to the input taps only.  The im2col buffer of the reference (up to
// 231 MB at batch 256) is never materialised.
#include "adam_dev.h"

namespace th {

constexpr int CO_T = 8;   // output channels per thread (9 x 8 = 72 weight SGPRs per input channel: no spills)
constexpr int CI_T = 8;   // input channels staged per LDS pass

// w_t[((ci*3 + kh)*3 + kw) * co_pad + co] = w_eff[co][ci][kh][kw]
// layout 0 (taper, tensor.rs:1262): w_eff[co][k] = w[k * c_out + co], k = ci*9 + kh*3 + kw
// layout 1 (standard, tensor.rs:1329): w_eff[co][k] = w[co * c_in*9 + k]
// flip != 0 builds the bwd-input filter: roles of ci/co swapped, taps mirrored.
__global__ __launch_bounds__(256) void conv3x3_prep_weights(const float *__restrict__ w, float *__restrict__ w_t, int c_in,
                                                            int c_out, int layout, int flip, int out_ch, int out_ch_pad,
                                                            int in_ch) {
    // output filter has `in_ch` input channels and `out_ch` output channels
    const int total = in_ch * 9 * out_ch_pad;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int oc = i % out_ch_pad, tap = (i / out_ch_pad) % 9, ic = i / (out_ch_pad * 9);
        float v = 0.f;
        if (oc < out_ch) {
            int co, ci, kh = tap / 3, kw = tap % 3;
            if (!flip) {
                co = oc; ci = ic;
            } else {  // gx[ci] = sum_co gy[co] * w_eff[co][ci][2-kh][2-kw]
                co = ic; ci = oc; kh = 2 - kh; kw = 2 - kw;
            }
            const int k = ci * 9 + kh * 3 + kw;
            v = layout == 0 ? w[(long)k * c_out + co] : w[(long)co * c_in * 9 + k];
        }
        w_t[i] = v;
    }
}

// grid = (ceil(n / IMG), out_ch_pad / CO_T); block = 256.
// Thread item -> (image in group, 2x4 output pixel tile); CO_T = 8 output channels.
// Per input channel a thread reads its 4x6 input window from LDS once (24 reads)
// and issues 9 taps x 8 pixels x 8 channels = 576 FMAs against 72 wave-uniform
// weights held in SGPRs (scalar loads): 24 FMAs per LDS read, 8 per weight dword.
constexpr int TPH = 2, TPW = 4;   // pixel tile

template <bool ACCUM>
__global__ __launch_bounds__(256) void conv3x3_kernel(const float *__restrict__ x, const float *__restrict__ w_t,
                                                      const float *__restrict__ bias, float *__restrict__ y, int n, int c_in,
                                                      int h, int w, int c_out, int co_pad, int pad, int h_out, int w_out,
                                                      int img_per_wg, int relu) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [img][CI_T][hp][wp]
    const int tiles_w = (w_out + TPW - 1) / TPW, tiles_h = (h_out + TPH - 1) / TPH;
    // the LDS plane carries a zero halo and is padded so that edge tiles read in bounds
    const int hp = tiles_h * TPH + 2 + (1 - pad);
    const int wp = (tiles_w * TPW + 2 + (1 - pad)) | 1;   // odd pitch: window rows fall on different banks
    const int plane = hp * wp;
    const int tiles_per_img = tiles_w * tiles_h;
    const int items = img_per_wg * tiles_per_img;
    const int img0 = blockIdx.x * img_per_wg;
    const int co0 = blockIdx.y * CO_T;
    const int shift = 1 - pad;            // pad=1: window starts on the halo; pad=0: one pixel in
    const int tid = threadIdx.x;

    // the halo (and everything else) is zeroed once; the staging passes only rewrite interiors
    for (int i = tid; i < img_per_wg * CI_T * plane; i += 256) xs[i] = 0.f;

    for (int item0 = 0; item0 < items; item0 += 256) {
        const int item = item0 + tid;
        const bool active = item < items;
        const int li = active ? item / tiles_per_img : 0;
        const int rem = active ? item % tiles_per_img : 0;
        const int oh0 = (rem / tiles_w) * TPH, ow0 = (rem % tiles_w) * TPW;
        const int img = img0 + li;
        const bool img_ok = active && img < n;

        float acc[TPH * TPW][CO_T];   // pixel p = (dy, dx) = (p / TPW, p % TPW)
#pragma unroll
        for (int p = 0; p < TPH * TPW; ++p)
#pragma unroll
            for (int j = 0; j < CO_T; ++j) acc[p][j] = 0.f;

        for (int cb = 0; cb < c_in; cb += CI_T) {
            const int nci = min(CI_T, c_in - cb);
            __syncthreads();  // previous pass (and the zero fill) done with xs
            // stage interiors.  Per image the nci planes are ONE contiguous run of nci*h*w floats:
            // consecutive threads take consecutive floats (fully coalesced) and track their
            // (channel, row, col) incrementally -- no integer division in the loop.
            const int run = nci * h * w;
            const int step_y = 256 / w, step_x = 256 % w;
            for (int si = 0; si < img_per_wg; ++si) {
                const int gi = img0 + si;
                if (gi >= n) break;
                const float *src = x + ((long)gi * c_in + cb) * h * w;
                float *dst = xs + (si * CI_T) * plane + wp + 1;
                int sy = tid / w, sx = tid % w, sc = 0;
                while (sy >= h) { sy -= h; ++sc; }
                for (int e = tid; e < run; e += 256) {
                    dst[sc * plane + sy * wp + sx] = src[e];
                    sx += step_x;
                    sy += step_y;
                    if (sx >= w) { sx -= w; ++sy; }
                    while (sy >= h) { sy -= h; ++sc; }
                }
            }
            __syncthreads();
            if (active) {
                const float *xbase = xs + (li * CI_T) * plane + (oh0 + shift) * wp + ow0 + shift;
                for (int c = 0; c < nci; ++c) {
                    const float *xp = xbase + c * plane;
                    float in[TPH + 2][TPW + 2];
#pragma unroll
                    for (int yy = 0; yy < TPH + 2; ++yy)
#pragma unroll
                        for (int xx = 0; xx < TPW + 2; ++xx) in[yy][xx] = xp[yy * wp + xx];
                    const float *wc = w_t + (long)(cb + c) * 9 * co_pad + co0;  // wave-uniform -> scalar loads
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const float *wk = wc + (kh * 3 + kw) * co_pad;
#pragma unroll
                            for (int j = 0; j < CO_T; ++j) {
                                const float wv = wk[j];
#pragma unroll
                                for (int p = 0; p < TPH * TPW; ++p)
                                    acc[p][j] = fmaf(in[p / TPW + kh][p % TPW + kw], wv, acc[p][j]);
                            }
                        }
                }
            }
        }

        if (img_ok) {
#pragma unroll
            for (int j = 0; j < CO_T; ++j) {
                const int co = co0 + j;
                if (co >= c_out) break;
                const float bv = bias ? bias[co] : 0.f;
                float *yp = y + ((long)img * c_out + co) * h_out * w_out;
#pragma unroll
                for (int p = 0; p < TPH * TPW; ++p) {
                    const int oh = oh0 + p / TPW, ow = ow0 + p % TPW;
                    if (oh < h_out && ow < w_out) {
                        float v = acc[p][j] + bv;
                        if (relu) v = v > 0.f ? v : 0.f;
                        float *q = yp + oh * w_out + ow;
                        if (ACCUM) *q += v;
                        else *q = v;
                    }
                }
            }
        }
    }
}

// Row-tile variant for small / odd planes (14x14, 7x7, ...): thread item -> (image, output row,
// column group of PX pixels) x CO_R = 16 output channels.  More, finer work items than the 2x4
// tile kernel, which matters when an image only has 49 or 196 pixels.
constexpr int CO_R = 16;

template <int PX, bool ACCUM>
__global__ __launch_bounds__(256) void conv3x3_rows_kernel(const float *__restrict__ x, const float *__restrict__ w_t,
                                                      const float *__restrict__ bias, float *__restrict__ y, int n, int c_in,
                                                      int h, int w, int c_out, int co_pad, int pad, int h_out, int w_out,
                                                      int img_per_wg, int relu) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [img][CI_T][h+2][w+2 (+pad to even)]
    const int hp = h + 2, wp = w + 2;     // LDS plane always carries a 1-px halo; pad==0 just shifts the window
    const int plane = hp * wp;
    const int groups_per_row = (w_out + PX - 1) / PX;
    const int items_per_img = h_out * groups_per_row;
    const int items = img_per_wg * items_per_img;
    const int img0 = blockIdx.x * img_per_wg;
    const int co0 = blockIdx.y * CO_R;
    const int shift = 1 - pad;            // pad=1: window starts at halo; pad=0: starts one pixel in

    for (int i = threadIdx.x; i < img_per_wg * CI_T * plane; i += 256) xs[i] = 0.f;   // halo zeroed once

    for (int item0 = 0; item0 < items; item0 += 256) {
        const int item = item0 + threadIdx.x;
        const bool active = item < items;
        const int li = active ? item / items_per_img : 0;
        const int rem = active ? item % items_per_img : 0;
        const int oh = rem / groups_per_row, ow0 = (rem % groups_per_row) * PX;
        const int img = img0 + li;
        const bool img_ok = active && img < n;

        float acc[PX][CO_R];
#pragma unroll
        for (int p = 0; p < PX; ++p)
#pragma unroll
            for (int j = 0; j < CO_R; ++j) acc[p][j] = 0.f;

        for (int cb = 0; cb < c_in; cb += CI_T) {
            const int nci = min(CI_T, c_in - cb);
            __syncthreads();  // previous pass done with xs
            // stage interiors (the halo was zeroed once): per image the nci planes are ONE contiguous run;
            // threads take consecutive floats and track (channel, row, col) incrementally
            const int run = nci * h * w;
            const int step_y = 256 / w, step_x = 256 % w;
            for (int si = 0; si < img_per_wg; ++si) {
                const int gi = img0 + si;
                if (gi >= n) break;
                const float *src = x + ((long)gi * c_in + cb) * h * w;
                float *dst = xs + (si * CI_T) * plane + wp + 1;
                int sy = threadIdx.x / w, sx = threadIdx.x % w, sc = 0;
                while (sy >= h) { sy -= h; ++sc; }
                for (int e = threadIdx.x; e < run; e += 256) {
                    dst[sc * plane + sy * wp + sx] = src[e];
                    sx += step_x;
                    sy += step_y;
                    if (sx >= w) { sx -= w; ++sy; }
                    while (sy >= h) { sy -= h; ++sc; }
                }
            }
            __syncthreads();
            if (active) {
                for (int c = 0; c < nci; ++c) {
                    const float *xp = xs + (li * CI_T + c) * plane;
                    const float *wc = w_t + (long)(cb + c) * 9 * co_pad + co0;  // wave-uniform
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        float in[PX + 2];
                        const float *row = xp + (oh + kh + shift) * wp + ow0 + shift;
#pragma unroll
                        for (int q = 0; q < PX + 2; ++q) in[q] = (ow0 + q < w_out + 2) ? row[q] : 0.f;
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const float *wk = wc + (kh * 3 + kw) * co_pad;
#pragma unroll
                            for (int j = 0; j < CO_R; ++j) {
                                const float wv = wk[j];
#pragma unroll
                                for (int p = 0; p < PX; ++p) acc[p][j] = fmaf(in[p + kw], wv, acc[p][j]);
                            }
                        }
                    }
                }
            }
        }

        if (img_ok) {
#pragma unroll
            for (int j = 0; j < CO_R; ++j) {
                const int co = co0 + j;
                if (co >= c_out) break;
                const float bv = bias ? bias[co] : 0.f;
                float *yp = y + (((long)img * c_out + co) * h_out + oh) * w_out + ow0;
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    if (ow0 + p < w_out) {
                        float v = acc[p][j] + bv;
                        if (relu) v = v > 0.f ? v : 0.f;
                        if (ACCUM) yp[p] += v;
                        else yp[p] = v;
                    }
                }
            }
        }
    }
}

// gw_eff[co][ci][kh][kw] += sum_{n,h,w} x[n,ci,h+kh-pad,w+kw-pad] * gy[n,co,h,w]
// One workgroup per (co, ci) pair and image slab (blockIdx.z); 256 threads stride over the slab's
// n*h_out*w_out pixels; 9 block reductions.  Without slabs the result is added to gw through the weight
// layout map; with `part` set, slab z writes part[z][k][c_out] and wgrad_reduce adds the slabs in order.
__global__ __launch_bounds__(256) void conv3x3_bwd_weight_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                 float *__restrict__ gw, float *__restrict__ part, int n, int c_in,
                                                                 int h, int w, int c_out, int pad, int h_out, int w_out, int layout,
                                                                 int img_per_slab, int accumulate) {
    __shared__ float sh[9][4];
    const int co = blockIdx.x, ci = blockIdx.y;
    const int b0 = blockIdx.z * img_per_slab, b1 = min(n, b0 + img_per_slab);
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    const int osp = h_out * w_out;
    const long total = (long)(b1 - b0) * osp;
    for (long i = threadIdx.x; i < total; i += 256) {
        const int img = b0 + (int)(i / osp), p = (int)(i % osp);
        const int oh = p / w_out, ow = p % w_out;
        const float g = gy[((long)img * c_out + co) * osp + p];
        const float *xp = x + ((long)img * c_in + ci) * h * w;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh + kh - pad;
            if (ih < 0 || ih >= h) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow + kw - pad;
                if (iw < 0 || iw >= w) continue;
                acc[kh * 3 + kw] = fmaf(xp[ih * w + iw], g, acc[kh * 3 + kw]);
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float v = acc[t];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) sh[t][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const int t = threadIdx.x;
        const float tot = ((sh[t][0] + sh[t][1]) + sh[t][2]) + sh[t][3];
        const int k = ci * 9 + t;
        if (part) {
            part[((long)blockIdx.z * c_in * 9 + k) * c_out + co] = tot;
        } else {
            const long idx = layout == 0 ? (long)k * c_out + co : (long)co * c_in * 9 + k;
            gw[idx] = accumulate ? gw[idx] + tot : tot;
        }
    }
}

// The same gradient for a SINGLE input channel (conv1 of both CNNs: 1 -> 32 on 28 x 28), where the kernel above is one workgroup per
// (co, slab) striding over 6 272 pixels with nine guarded loads each (50 us at batch 256 for 26 MB of traffic).  Here a workgroup owns one
// IMAGE: its zero-haloed input plane and its [c_out_blk <= 32][h_out w_out] gradient planes are staged in LDS with coalesced loads (all of a
// thread's loads ahead of its first LDS store), then thread (channel, pixel slice) walks its slice with the nine tap sums in registers --
// one gradient read and nine window reads from LDS per nine FMAs -- and the slices are added in slice order.  One partial [9][c_out] slab
// per image; wgrad_reduce adds the images in order (deterministic).  grid = (n, ceil(c_out / 32)), 512 threads.
constexpr int C1W_CO = 32, C1W_NT = 512, C1W_NS = C1W_NT / C1W_CO;
__global__ __launch_bounds__(C1W_NT) void conv1_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ part,
                                                            int h, int w, int c_out, int pad, int h_out, int w_out) {
    extern __shared__ float c1w_lds[];
    const int wp = w + 2, hp = h + 2, osp = h_out * w_out, gp = osp + 1;      // (odd plane pitch: the 32 channels of a wave half spread over the banks)
    float *img = c1w_lds, *g = img + ((hp * wp + 3) & ~3), *red = g + C1W_CO * gp;     // red: [C1W_NS][C1W_CO][9]
    const int t = threadIdx.x, b = blockIdx.x, co0 = blockIdx.y * C1W_CO, nco = min(C1W_CO, c_out - co0);
    const float *xi = x + (long)b * h * w, *gi = gy + ((long)b * c_out + co0) * osp;
    for (int e = t; e < hp * wp; e += C1W_NT) {
        const int r = e / wp - 1, c = e % wp - 1;
        img[e] = (r >= 0 && r < h && c >= 0 && c < w) ? xi[r * w + c] : 0.f;
    }
    {
        const int total = nco * osp;           // the block's gradient planes are contiguous in memory
        constexpr int U = 8;
        for (int e0 = t; e0 < total; e0 += C1W_NT * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * C1W_NT;
                v[u] = e < total ? gi[e] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * C1W_NT;
                if (e < total) {
                    const int c = e / osp;
                    g[c * gp + e - c * osp] = v[u];
                }
            }
        }
    }
    __syncthreads();
    const int cl = t % C1W_CO, sl = t / C1W_CO;
    const int per = (osp + C1W_NS - 1) / C1W_NS, p0 = sl * per, p1 = min(osp, p0 + per);
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    if (cl < nco && p0 < p1) {
        int oh = p0 / w_out, ow = p0 - oh * w_out;
        const float *gc = g + cl * gp;
        const int shift = 1 - pad;             // halo coordinates of the window's corner: (oh + shift, ow + shift)
        float win[9];                          // the window slides along the row: three new values per pixel, nine at a row's start
        bool fresh = true;
        for (int p = p0; p < p1; ++p) {
            const float gv = gc[p];
            const float *wc = img + (oh + shift) * wp + ow + shift;
            if (fresh) {
#pragma unroll
                for (int k = 0; k < 9; ++k) win[k] = wc[(k / 3) * wp + k % 3];
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    win[3 * r] = win[3 * r + 1];
                    win[3 * r + 1] = win[3 * r + 2];
                    win[3 * r + 2] = wc[r * wp + 2];
                }
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] = fmaf(win[k], gv, acc[k]);
            fresh = ++ow == w_out;
            if (fresh) { ow = 0; ++oh; }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) red[(sl * C1W_CO + cl) * 9 + k] = acc[k];
    __syncthreads();
    if (t < C1W_CO * 9) {
        const int c = t % C1W_CO, k = t / C1W_CO;
        if (c < nco) {
            float v = red[c * 9 + k];
#pragma unroll
            for (int q = 1; q < C1W_NS; ++q) v += red[(q * C1W_CO + c) * 9 + k];
            part[((long)b * 9 + k) * c_out + co0 + c] = v;
        }
    }
}

// ---- NCHW bias (tensor.rs:1983-1992, 2017-2024) ---------------------------
__global__ __launch_bounds__(256) void bias_add_nchw_kernel(const float *__restrict__ x, const float *__restrict__ bias,
                                                            float *__restrict__ y, long total, int c, int hw, int relu) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float v = x[i] + bias[(i / hw) % c];
        if (relu) v = v > 0.f ? v : 0.f;
        y[i] = v;
    }
}

// grid = (c, slabs): block (ch, s) sums channel ch over its slab of images; with `part` set the slab sums land
// in part[s][c] and th_colsum_accum adds them to gb in slab order (deterministic), else gb[ch] += the sum.
// mask (nullable): only elements with mask[same index] > 0 count -- the ReLU backward folded in; overwrite: gb = sum
// pooled != 0: g is [n][c] -- the gradient of a GLOBAL average pool's output -- and every element of plane (b, ch)
// receives g[b][ch] / hw (tensor.rs:1626-1628)
__global__ __launch_bounds__(256) void bias_grad_nchw_kernel(const float *__restrict__ g, const float *__restrict__ mask,
                                                             float *__restrict__ gb, float *__restrict__ part, int n, int c, int hw,
                                                             int img_per_slab, int overwrite, int pooled) {
    __shared__ float sh[4];
    const int ch = blockIdx.x;
    const int b0 = blockIdx.y * img_per_slab, b1 = min(n, b0 + img_per_slab);
    float s = 0.f;
    const long total = (long)(b1 - b0) * hw;
    // (image, pixel) tracked incrementally: no integer division per element
    const int step_b = 256 / hw, step_sp = 256 % hw;
    int b = b0 + threadIdx.x / hw, sp = threadIdx.x % hw;
    for (long i = threadIdx.x; i < total; i += 256) {
        const long ix = ((long)b * c + ch) * hw + sp;
        const float v = pooled ? g[(long)b * c + ch] / (float)hw : g[ix];
        s += (mask && !(mask[ix] > 0.f)) ? 0.f : v;
        sp += step_sp;
        b += step_b;
        if (sp >= hw) { sp -= hw; ++b; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = ((sh[0] + sh[1]) + sh[2]) + sh[3];
        if (part) part[(long)blockIdx.y * c + ch] = tot;
        else gb[ch] = overwrite ? tot : gb[ch] + tot;
    }
}

// tmp[n][p][co] -> y[n][co][p] + bias[co] (+relu): the reshape + transpose_4d
// (tensor.rs:1275-1276, 2034-2076) + add_bias_4d of the 1x1 path
__global__ __launch_bounds__(256) void nhwc_to_nchw_bias_kernel(const float *__restrict__ t, const float *__restrict__ bias,
                                                                float *__restrict__ y, int hw, int c, int relu) {
    __shared__ float tile[64][65];
    const int img = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int p0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const float *tin = t + (long)img * hw * c;
    float *yout = y + (long)img * hw * c;
    for (int r = ty; r < 64; r += 4) {
        const int p = p0 + r, cc = c0 + tx;
        tile[r][tx] = (p < hw && cc < c) ? tin[(long)p * c + cc] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int cc = c0 + r, p = p0 + tx;
        if (cc < c && p < hw) {
            float v = tile[tx][r] + (bias ? bias[cc] : 0.f);
            if (relu) v = v > 0.f ? v : 0.f;
            yout[(long)cc * hw + p] = v;
        }
    }
}

// ---- pooling --------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                          int64_t *__restrict__ argmax, long total, int h, int w, int h_out,
                                                          int w_out, int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w) {
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const int ow = (int)(o % w_out), oh = (int)((o / w_out) % h_out);
        const long bc = o / ((long)w_out * h_out);
        const long in_base = bc * h * w;
        float best = -INFINITY;      // tensor.rs:1431
        long best_idx = in_base;     // tensor.rs:1432 "any valid default"
        for (int kh = 0; kh < k_h; ++kh) {          // kh outer, kw inner (1435-1455)
            const int ihp = oh * s_h + kh;
            if (ihp < pad_h || ihp >= h + pad_h) continue;
            for (int kw = 0; kw < k_w; ++kw) {
                const int iwp = ow * s_w + kw;
                if (iwp < pad_w || iwp >= w + pad_w) continue;
                const long idx = in_base + (long)(ihp - pad_h) * w + (iwp - pad_w);
                const float v = x[idx];
                if (v > best) {  // strict: first max wins, NaN never wins
                    best = v;
                    best_idx = idx;
                }
            }
        }
        y[o] = best;
        if (argmax) argmax[o] = best_idx;
    }
}

// 2x2 / stride 2 / unpadded windows on even maps: a thread owns a window and reads it as two float2 (consecutive lanes = consecutive
// windows of a row: coalesced), same scan (kh outer, kw inner, strict >, -inf start, default index = the plane's first pixel)
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t *__restrict__ argmax,
                                                           long total, int h, int w) {
    const int w_out = w >> 1, h_out = h >> 1;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const int ow = (int)(o % w_out), oh = (int)((o / w_out) % h_out);
        const long in_base = (o / ((long)w_out * h_out)) * h * w, p00 = in_base + (long)(2 * oh) * w + 2 * ow;
        const float2 a = *reinterpret_cast<const float2 *>(x + p00), b = *reinterpret_cast<const float2 *>(x + p00 + w);
        float best = -INFINITY;
        long idx = in_base;
        if (a.x > best) { best = a.x; idx = p00; }
        if (a.y > best) { best = a.y; idx = p00 + 1; }
        if (b.x > best) { best = b.x; idx = p00 + w; }
        if (b.y > best) { best = b.y; idx = p00 + w + 1; }
        y[o] = best;
        if (argmax) argmax[o] = idx;
    }
}

// Gather form of the scatter-add of tensor.rs:1504-1514: every input pixel
// walks the windows that can contain it in (oh, ow) ascending order -- the same
// order the reference's sequential `for o in 0..out_spatial` adds them, so
// the result is bit-identical, with no atomics.  A pixel whose padded
// coordinate lies outside a window can never be that window's argmax -- except
// pixel (0,0) of a plane: a window with no element > -inf (all NaN / -inf, or
// all padding) keeps the `best_idx = in_base` default (tensor.rs:1432) and its
// gradient lands there (1504-1514), wherever the window is.  So the thread of
// pixel (0,0) walks ALL windows of its plane, in the same ascending order.
__global__ __launch_bounds__(256) void maxpool_bwd_geo_kernel(const float *__restrict__ gout, const int64_t *__restrict__ argmax,
                                                              float *__restrict__ gin, long total, int h, int w, int h_out,
                                                              int w_out, int k_h, int k_w, int s_h, int s_w, int pad_h,
                                                              int pad_w, int zero_first) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int iw = (int)(i % w), ih = (int)((i / w) % h);
        const long bc = i / ((long)h * w);
        const long obase = bc * h_out * w_out;
        float v = zero_first ? 0.f : gin[i];
        const int ihp = ih + pad_h, iwp = iw + pad_w;
        const int oh_lo = ihp >= k_h ? (ihp - k_h) / s_h + 1 : 0, oh_hi = min(h_out - 1, ihp / s_h);
        const int ow_lo = iwp >= k_w ? (iwp - k_w) / s_w + 1 : 0, ow_hi = min(w_out - 1, iwp / s_w);
        if (ih == 0 && iw == 0) {
            const int n_out = h_out * w_out;
#pragma unroll 4
            for (int o = 0; o < n_out; ++o)
                if (argmax[obase + o] == i) v += gout[obase + o];
        } else {
            for (int oh = oh_lo; oh <= oh_hi; ++oh)
                for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                    const long o = obase + (long)oh * w_out + ow;
                    if (argmax[o] == i) v += gout[o];
                }
        }
        gin[i] = v;
    }
}

// The same scatter for the geometry the CNNs use -- 2x2 windows, stride 2, no padding, even height and width: every input pixel lies in
// exactly ONE window, so a window's lane writes its four pixels (two float2 stores) and nothing is gathered.  The general kernel's
// thread of pixel (0,0) walks all of its plane's windows one dependent load after the other (196 for a 28 x 28 plane: 82 us per launch
// at batch 256 for 45 MB of traffic); here a WAVE owns a plane, its lanes read the plane's indices / gradients coalesced, and the
// windows that kept the default index (tensor.rs:1432: none of their elements is > -inf) are found by a ballot -- only then does lane 0
// walk the plane, adding in ascending window order like tensor.rs:1504-1514.  Same bits as the general kernel.
// MASKED: the pool's input is the output of a ReLU (a Conv2dReLU row): the gradient that reaches pixel p is kept only where that
// output is > 0 (ops.rs:358-369) -- for the pixel a window's maximum came from that is `pooled value > 0`, so the ReLU's backward
// costs no pass over the full-resolution map (y_full is read for pixel (0,0) of planes with a default-index window only).
template <bool MASKED>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float *__restrict__ gout, const int64_t *__restrict__ argmax,
                                                           const float *__restrict__ y_pooled, const float *__restrict__ y_full,
                                                           float *__restrict__ gin, float *__restrict__ plane_sums, int planes, int h, int w,
                                                           int zero_first) {
    const int lane = threadIdx.x & 63, w_out = w >> 1, hw_out = (h >> 1) * w_out;
    const int wave0 = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    for (int pl = wave0; pl < planes; pl += n_waves) {
        const long obase = (long)pl * hw_out, in_base = (long)pl * h * w;
        bool deg = false;
        float first = 0.f;                                 // what lane 0 wrote to pixel (0,0)
        float wsum = 0.f;                                  // plane_sums (zero_first forms): what this lane's windows scattered
        for (int o0 = 0; o0 < hw_out; o0 += 256) {         // four windows per lane and round: the loads of a round go out together
            int64_t am[4];
            float g[4], m[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int o = o0 + 64 * u + lane;
                const bool in = o < hw_out;
                am[u] = in ? argmax[obase + o] : -1;
                g[u] = in ? gout[obase + o] : 0.f;
                m[u] = MASKED && in ? y_pooled[obase + o] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int o = o0 + 64 * u + lane;
                if (o >= hw_out) continue;
                const int oh = o / w_out, ow = o - oh * w_out;
                const long p00 = in_base + (long)(2 * oh) * w + 2 * ow;
                const long rel = am[u] - p00;
                const float gv = (!MASKED || m[u] > 0.f) ? g[u] : 0.f;
                float2 top = make_float2(0.f, 0.f), bot = top;
                if (!zero_first) {
                    top = *reinterpret_cast<const float2 *>(gin + p00);
                    bot = *reinterpret_cast<const float2 *>(gin + p00 + w);
                }
                if (rel == 0) top.x += gv;
                else if (rel == 1) top.y += gv;
                else if (rel == w) bot.x += gv;
                else if (rel == w + 1) bot.y += gv;
                if (rel == 0 || rel == 1 || rel == w || rel == w + 1) wsum += gv;
                deg |= o != 0 && am[u] == in_base;
                if (o == 0) first = top.x;
                *reinterpret_cast<float2 *>(gin + p00) = top;
                *reinterpret_cast<float2 *>(gin + p00 + w) = bot;
            }
        }
        if (__ballot(deg) != 0ull && lane == 0) {
            // rare: some window away from the origin kept the default index.  Pixel (0,0) = what lane 0 left there (the old value + window
            // 0's share) + every other window that points at it, in ascending order.  Under MASKED (zero_first only) the ReLU mask of THAT
            // pixel applies to the whole sum, so it is formed again from the unmasked gradients.
            float v = MASKED ? 0.f : first;
            for (int o = MASKED ? 0 : 1; o < hw_out; ++o)
                if (argmax[obase + o] == in_base) v += gout[obase + o];
            if (MASKED && !(y_full[in_base] > 0.f)) v = 0.f;
            gin[in_base] = v;
            wsum += v - first;
        }
        if (plane_sums) {      // the sum of the plane's scattered gradient: the bias gradient of the Conv2dReLU in front needs nothing else of it
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wsum += __shfl_down(wsum, off, 64);
            if (lane == 0) plane_sums[pl] = wsum;
        }
    }
}

// ReLU backward on an NCHW map with the sum of every plane of the result on the way (ops.rs:358-369 + the rows of tensor.rs:2017-2024's
// bias sum): a wave per plane, float4 where the plane allows.  The bias gradient of a Conv2dReLU is then a sum over [n][c] instead of a
// second pass over [n][c][h][w] (bias_grad_nchw_kernel: 15 us per layer at batch 256).
__global__ __launch_bounds__(256) void relu_bwd_planes_kernel(const float *__restrict__ y, const float *__restrict__ gout, float *__restrict__ gin,
                                                              float *__restrict__ plane_sums, int planes, int hw) {
    const int lane = threadIdx.x & 63;
    const int wave0 = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    const bool vec = (hw & 3) == 0;
    for (int pl = wave0; pl < planes; pl += n_waves) {
        const long base = (long)pl * hw;
        float s = 0.f;
        if (vec) {
            const float4 *y4 = reinterpret_cast<const float4 *>(y + base), *g4 = reinterpret_cast<const float4 *>(gout + base);
            float4 *o4 = reinterpret_cast<float4 *>(gin + base);
            for (int i = lane; i < hw / 4; i += 64) {
                const float4 yv = y4[i], gv = g4[i];
                float4 o;
                o.x = yv.x > 0.f ? gv.x : 0.f; o.y = yv.y > 0.f ? gv.y : 0.f; o.z = yv.z > 0.f ? gv.z : 0.f; o.w = yv.w > 0.f ? gv.w : 0.f;
                o4[i] = o;
                s += (o.x + o.y) + (o.z + o.w);
            }
        } else {
            for (int i = lane; i < hw; i += 64) {
                const float o = y[base + i] > 0.f ? gout[base + i] : 0.f;
                gin[base + i] = o;
                s += o;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) plane_sums[pl] = s;
    }
}

// Backward of a GLOBAL average pool (tensor.rs:1626-1628: every element of plane (b, ch) receives g[b][ch] / hw) whose input is the output
// of a ReLU, with that ReLU's backward and the plane sums in the same pass: gin = (y > 0) ? 0 + g / hw : 0.  A wave per plane.
__global__ __launch_bounds__(256) void gap_relu_bwd_planes_kernel(const float *__restrict__ y, const float *__restrict__ g, float *__restrict__ gin,
                                                                  float *__restrict__ plane_sums, int planes, int hw) {
    const int lane = threadIdx.x & 63;
    const int wave0 = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    for (int pl = wave0; pl < planes; pl += n_waves) {
        const long base = (long)pl * hw;
        const float gv = 0.f + g[pl] / (float)hw;      // (the pool's backward adds it to a zeroed slot)
        float s = 0.f;
        for (int i = lane; i < hw; i += 64) {
            const float o = y[base + i] > 0.f ? gv : 0.f;
            gin[base + i] = o;
            s += o;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0 && plane_sums) plane_sums[pl] = s;
    }
}

// gb[ch] (+)= sum over images of plane_sums[b][ch], images in ascending order within each of 16 interleaved chains (deterministic);
// one workgroup per 16 channels: thread (chain q, channel) adds images q, q + 16, ...; the 16 chains are added in order
__global__ __launch_bounds__(256) void bias_from_plane_sums_kernel(const float *__restrict__ ps, float *__restrict__ gb, int n, int c, int accumulate) {
    __shared__ float sh[16][16];
    const int cl = threadIdx.x & 15, q = threadIdx.x >> 4, ch = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (ch < c) {
#pragma unroll 8
        for (int b = q; b < n; b += 16) s += ps[(long)b * c + ch];
    }
    sh[q][cl] = s;
    __syncthreads();
    if (q == 0 && ch < c) {
        float tot = sh[0][cl];
#pragma unroll
        for (int u = 1; u < 16; ++u) tot += sh[u][cl];
        gb[ch] = accumulate ? gb[ch] + tot : tot;
    }
}

static bool maxpool2_fast(int n, int c, int h, int w, int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w) {
    return k_h == 2 && k_w == 2 && s_h == 2 && s_w == 2 && pad_h == 0 && pad_w == 0 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0 &&
           (long)n * c < (1L << 31);
}

__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, long total, int h,
                                                          int w, int h_out, int w_out, int k_h, int k_w, int s_h, int s_w,
                                                          int pad_h, int pad_w) {
    const float pool_size = (float)(k_h * k_w);  // tensor.rs:1548 (Q6)
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const int ow = (int)(o % w_out), oh = (int)((o / w_out) % h_out);
        const long in_base = (o / ((long)w_out * h_out)) * h * w;
        float sum = 0.f;
        for (int kh = 0; kh < k_h; ++kh) {
            const int ihp = oh * s_h + kh;
            if (ihp < pad_h || ihp >= h + pad_h) continue;
            for (int kw = 0; kw < k_w; ++kw) {
                const int iwp = ow * s_w + kw;
                if (iwp < pad_w || iwp >= w + pad_w) continue;
                sum += x[in_base + (long)(ihp - pad_h) * w + (iwp - pad_w)];
            }
        }
        y[o] = sum / pool_size;
    }
}

// one wave per (b,c) plane when the window is the whole plane (global pool); cnt (nullable): number of elements > 0 in
// the plane -- for a post-ReLU input that is all its backward needs to know about the plane (th_bias_grad_counts_adam)
__global__ __launch_bounds__(256) void avgpool_global_kernel(const float *__restrict__ x, float *__restrict__ y, float *__restrict__ cnt,
                                                             long planes, int hw, float pool_size) {
    const int lane = threadIdx.x & 63;
    const long pl = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pl >= planes) return;
    float s = 0.f, k = 0.f;
    for (int i = lane; i < hw; i += 64) {
        const float v = x[pl * hw + i];
        s += v;
        k += v > 0.f ? 1.f : 0.f;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        k += __shfl_down(k, off, 64);
    }
    if (lane == 0) {
        y[pl] = s / pool_size;
        if (cnt) cnt[pl] = k;
    }
}

// small planes (hw <= 256): a workgroup takes 16 consecutive planes = 16 hw contiguous floats, staged through LDS with every
// thread's loads independent (one wave per 49-element plane is a single dependent load per lane: latency-bound at 1 TB/s);
// 16 lanes then sum a plane in the same ascending-stride order and finish with a 16-lane shuffle tree
__global__ __launch_bounds__(256) void avgpool_global16_kernel(const float *__restrict__ x, float *__restrict__ y, float *__restrict__ cnt,
                                                               long planes, int hw, float pool_size) {
    extern __shared__ float pl_sh[];                      // [16][hw]
    const long p0 = (long)blockIdx.x * 16;
    const int n_here = (int)min((long)16, planes - p0), total = n_here * hw;
    const float *src = x + p0 * hw;
    for (int e = threadIdx.x; e < total; e += 256) pl_sh[e] = src[e];
    __syncthreads();
    const int pl = threadIdx.x >> 4, l = threadIdx.x & 15;
    float s = 0.f, k = 0.f;
    if (pl < n_here)
        for (int i = l; i < hw; i += 16) {
            const float v = pl_sh[pl * hw + i];
            s += v;
            k += v > 0.f ? 1.f : 0.f;
        }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 16);
        k += __shfl_down(k, off, 16);
    }
    if (l == 0 && pl < n_here) {
        y[p0 + pl] = s / pool_size;
        if (cnt) cnt[p0 + pl] = k;
    }
}

static void avgpool_global_launch(th_ctx *ctx, const float *d_x, float *d_y, float *d_cnt, long planes, int hw) {
    if (hw <= 256 && planes >= 64)
        hipLaunchKernelGGL(avgpool_global16_kernel, dim3(ceil_div(planes, 16)), dim3(256), (size_t)16 * hw * sizeof(float), ctx->stream, d_x, d_y,
                           d_cnt, planes, hw, (float)hw);
    else
        hipLaunchKernelGGL(avgpool_global_kernel, dim3(ceil_div(planes, 4)), dim3(256), 0, ctx->stream, d_x, d_y, d_cnt, planes, hw, (float)hw);
}

// db[ch] = sum_n (g[n][ch] / hw) * cnt[n][ch] (+ the Adam update of that bias element, + carried deferred updates): the bias
// gradient of Conv2dReLU -> global average pool from 2 n c floats instead of the n c hw conv outputs.
__global__ __launch_bounds__(1024) void bias_grad_counts_adam_kernel(const float *__restrict__ g, const float *__restrict__ cnt,
                                                                     float *__restrict__ gb, int n, int c, int hw, AdamDev ad,
                                                                     AdamSlices extra) {
    const int groups = (c + 15) / 16;
    if ((int)blockIdx.x >= groups) {
        if (threadIdx.x < 256) adam_slices_block(extra, blockIdx.x - groups);
        return;
    }
    // 16 channels per workgroup: thread (r, q) sums images r, r + 64, ... of channel 16 blockIdx + q (64-byte segments,
    // every load of a thread independent of the others: the launch is one or two round trips long)
    __shared__ float sh[64][17];
    const int q = threadIdx.x & 15, r = threadIdx.x >> 4, ch = blockIdx.x * 16 + q;
    float s = 0.f;
    if (ch < c) {
#pragma unroll 4
        for (int b = r; b < n; b += 64) s += g[(long)b * c + ch] / (float)hw * cnt[(long)b * c + ch];
    }
    sh[r][q] = s;
    __syncthreads();
    if (threadIdx.x < 16 && ch < c) {
        float tot = sh[0][q];
#pragma unroll
        for (int i = 1; i < 64; ++i) tot += sh[i][q];
        gb[ch] = tot;
        if (ad.p) {
            const float step = adam_dev_step(ad);
            adam_update(ad.p, ad.m, ad.v, ch, tot, step, ad.beta1, ad.beta2, ad.eps, ad.wd);
        }
    }
}

// gather form of tensor.rs:1624-1653, (oh, ow) ascending like the reference
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float *__restrict__ gout, float *__restrict__ gin, long total,
                                                          int h, int w, int h_out, int w_out, int k_h, int k_w, int s_h,
                                                          int s_w, int pad_h, int pad_w) {
    const float pool_size = (float)(k_h * k_w);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int iw = (int)(i % w), ih = (int)((i / w) % h);
        const long obase = (i / ((long)h * w)) * h_out * w_out;
        const int ihp = ih + pad_h, iwp = iw + pad_w;
        const int oh_lo = ihp >= k_h ? (ihp - k_h) / s_h + 1 : 0, oh_hi = min(h_out - 1, ihp / s_h);
        const int ow_lo = iwp >= k_w ? (iwp - k_w) / s_w + 1 : 0, ow_hi = min(w_out - 1, iwp / s_w);
        float v = gin[i];
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow) v += gout[obase + (long)oh * w_out + ow] / pool_size;
        gin[i] = v;
    }
}

static int conv3x3_launch(th_ctx *ctx, const float *x, const float *w_t, const float *bias, float *y, int n, int in_ch, int h,
                          int w, int out_ch, int co_pad, int pad, int relu, bool accum) {
    const int h_out = h + 2 * pad - 2, w_out = w + 2 * pad - 2;
    if (w_out >= 16 && w_out % TPW == 0 && h_out % TPH == 0) {
        // large planes: 2x4 pixel tiles x 8 channels, packed-FMA inner loop
        const int tiles_h = h_out / TPH, tiles_w = w_out / TPW;
        const int tiles_per_img = tiles_h * tiles_w;
        int img_per_wg = 256 / tiles_per_img;
        if (img_per_wg < 1) img_per_wg = 1;
        if (img_per_wg > n) img_per_wg = n;
        const int plane = (tiles_h * TPH + 2 + (1 - pad)) * ((tiles_w * TPW + 2 + (1 - pad)) | 1);
        while (img_per_wg > 1 && (size_t)img_per_wg * CI_T * plane * sizeof(float) > (64u << 10)) --img_per_wg;  // 2 workgroups per CU
        const size_t lds = (size_t)img_per_wg * CI_T * plane * sizeof(float);
        TH_REQUIRE(lds <= (160u << 10), "th_conv3x3: %dx%d plane does not fit the LDS tile", h, w);
        dim3 grid(ceil_div(n, img_per_wg), co_pad / CO_T);
#define TH_CONV_TILE(ACC)                                                                                                       \
    {                                                                                                                           \
        auto kern = conv3x3_kernel<ACC>;                                                                                        \
        if (lds > (64u << 10)) TH_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, x, w_t, bias, y, n, in_ch, h, w, out_ch, co_pad, pad, h_out, \
                           w_out, img_per_wg, relu);                                                                            \
    }
        if (accum) TH_CONV_TILE(true) else TH_CONV_TILE(false)
#undef TH_CONV_TILE
        TH_LAUNCH_CHECK();
        return 0;
    }
    // small / odd planes: row tiles of PX pixels x 16 channels
    const int px = (w_out % 4 == 0) ? 4 : ((w_out % 2 == 0) ? 2 : 1);
    const int items_per_img = h_out * ((w_out + px - 1) / px);
    int img_per_wg = 256 / items_per_img;
    if (img_per_wg < 1) img_per_wg = 1;
    if (img_per_wg > n) img_per_wg = n;
    const int plane = (h + 2) * (w + 2);
    while (img_per_wg > 1 && (size_t)img_per_wg * CI_T * plane * sizeof(float) > (64u << 10)) --img_per_wg;
    const size_t lds = (size_t)img_per_wg * CI_T * plane * sizeof(float);
    TH_REQUIRE(lds <= (160u << 10), "th_conv3x3: %dx%d plane does not fit the LDS tile", h, w);
    dim3 grid(ceil_div(n, img_per_wg), co_pad / CO_R);
#define TH_CONV_ROWS(PXV, ACC)                                                                                                  \
    {                                                                                                                           \
        auto kern = conv3x3_rows_kernel<PXV, ACC>;                                                                              \
        if (lds > (64u << 10)) TH_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, x, w_t, bias, y, n, in_ch, h, w, out_ch, co_pad, pad, h_out, \
                           w_out, img_per_wg, relu);                                                                            \
    }
    if (!accum) {
        if (px == 4) TH_CONV_ROWS(4, false) else if (px == 2) TH_CONV_ROWS(2, false) else TH_CONV_ROWS(1, false)
    } else {
        if (px == 4) TH_CONV_ROWS(4, true) else if (px == 2) TH_CONV_ROWS(2, true) else TH_CONV_ROWS(1, true)
    }
#undef TH_CONV_ROWS
    TH_LAUNCH_CHECK();
    return 0;
}

// conv_mfma.hip: the matrix-core path for C_in >= 8
bool conv3x3_mfma_supported(int c_in, int h, int w, int pad);
bool conv3x3_mfma_pool_supported(int c_in, int h, int w, int pad);
bool conv3x3_gap_supported(int n, int c_in, int h, int w_in, int c_out, int pad);
int conv3x3_mfma_launch(th_ctx *ctx, const float *x, const float *w, int w_ld, int w_cols, const float *bias, float *y, int n,
                        int c_in, int h, int w_in, int c_out, int pad, int relu, bool accum, bool pool = false, float *gap_cnt = nullptr,
                        bool gap = false);
int conv3x3_wgrad_mfma_launch(th_ctx *ctx, const float *x, const float *gy, float *gw, int n, int c_in, int h, int w_in, int c_out,
                              int pad, int layout, int accumulate);
int wgrad_reduce(th_ctx *ctx, const float *part, float *gw, int G, int kt, int c_out, int co_ld, int layout, int accumulate);

}  // namespace th

using namespace th;

namespace th {
// Bias gradient behind a GLOBAL average pool with the ReLU mask folded in, planes of <= 64 elements (7x7 here): one wave
// per plane -- lanes over the plane, the mask count by ballot -- adds (g[b][ch] / hw) * count; waves stride the slab's
// images.  grid = (c, slabs); with `part` the slab sums land in part[s][c], else gb[ch] (+)= the sum.
__global__ __launch_bounds__(256) void bias_grad_avgpool_small_kernel(const float *__restrict__ gp, const float *__restrict__ y,
                                                                      float *__restrict__ gb, float *__restrict__ part, int n, int c,
                                                                      int hw, int img_per_slab, int overwrite) {
    __shared__ float sh[4];
    const int ch = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b0 = blockIdx.y * img_per_slab, b1 = min(n, b0 + img_per_slab);
    float s = 0.f;
    for (int b = b0 + wave; b < b1; b += 4) {
        const long plane = (long)b * c + ch;
        const bool on = lane < hw && y[plane * hw + lane] > 0.f;
        const int cnt = __popcll(__ballot(on));
        s += (gp[plane] / (float)hw) * (float)cnt;      // wave-uniform
    }
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = ((sh[0] + sh[1]) + sh[2]) + sh[3];
        if (part) part[(long)blockIdx.y * c + ch] = tot;
        else gb[ch] = overwrite ? tot : gb[ch] + tot;
    }
}

// The bias gradient of a conv layer as the LAST backward launch of a fused training step: one 16-wave workgroup per channel
// sums its (masked / pooled) plane gradients in a fixed order -- no slabs, no second launch -- and applies the bias's Adam
// update (optim.rs:99-110) in place; the blocks behind the channels carry the deferred updates of other parameters
// (th_adam_slice), so the step needs no optimizer launch at all.  mode 0: g is [n][c][hw]; 1: g is [n][c], the gradient of a
// global average pool's output (every element of the plane receives g / hw, tensor.rs:1626-1628).
__global__ __launch_bounds__(1024) void bias_grad_adam_kernel(const float *__restrict__ g, const float *__restrict__ mask,
                                                              float *__restrict__ gb, int n, int c, int hw, int mode, AdamDev ad,
                                                              AdamSlices extra) {
    if ((int)blockIdx.x >= c) {
        if (threadIdx.x < 256) adam_slices_block(extra, blockIdx.x - c);
        return;
    }
    __shared__ float sh[16];
    const int ch = blockIdx.x, t = threadIdx.x;
    float pv = 0.f, mv = 0.f, vv = 0.f, step = 0.f;
    if (t == 0 && ad.p) {   // requested before the sum, not after it
        pv = ad.p[ch];
        mv = ad.m[ch];
        vv = ad.v[ch];
        step = adam_dev_step(ad);
    }
    float s = 0.f;
    const long total = (long)n * hw;
    const int step_b = 1024 / hw, step_sp = 1024 % hw;   // (image, pixel) tracked incrementally: no division per element
    int b = t / hw, sp = t % hw;
    for (long i = t; i < total; i += 1024) {
        const long ix = ((long)b * c + ch) * hw + sp;
        const float v = mode ? g[(long)b * c + ch] / (float)hw : g[ix];
        s += (mask && !(mask[ix] > 0.f)) ? 0.f : v;
        sp += step_sp;
        b += step_b;
        if (sp >= hw) { sp -= hw; ++b; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((t & 63) == 0) sh[t >> 6] = s;
    __syncthreads();
    if (t == 0) {
        float tot = sh[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) tot += sh[w];
        gb[ch] = tot;
        if (ad.p) {
            const float gv = tot + ad.wd * pv;
            const float mn = ad.beta1 * mv + (1.0f - ad.beta1) * gv;
            const float vn = ad.beta2 * vv + (1.0f - ad.beta2) * gv * gv;
            ad.m[ch] = mn;
            ad.v[ch] = vn;
            ad.p[ch] = pv - step * mn / (sqrtf(vn) + ad.eps);
        }
    }
}

// The same bias gradient from per-COLUMN sums: when the consumer of a bias-only Conv2dReLU (+ pool) is flatten -> Linear, the
// classifier head (th_linear_xent_wide_ex) has every dX[row][col] and x[row][col] in registers and hands over
// colsum[col] = sum_rows dX * [x > 0]; channel ch owns columns [ch hw, (ch + 1) hw) of the flattened map, so db[ch] is hw additions
// instead of a pass over two [n][c][hw] tensors, and dX is never written.  Block 0: one thread per channel (+ its Adam update);
// the other blocks apply carried Adam slices.
__global__ __launch_bounds__(1024) void bias_from_colsum_adam_kernel(const float *__restrict__ colsum, float *__restrict__ gb, int c, int hw, AdamDev ad,
                                                                     AdamSlices extra) {
    if (blockIdx.x > 0) {
        if (threadIdx.x < 256) adam_slices_block(extra, blockIdx.x - 1);
        return;
    }
    // 16 lanes per channel: lane j adds columns j, j + 16, ... (requested four at a time: a serial walk of hw dependent loads took 9 us),
    // then the 16 partial sums meet in a fixed shuffle tree
    const int sub = threadIdx.x & 15;
    for (int ch0 = 0; ch0 < c; ch0 += 64) {
        const int ch = ch0 + (threadIdx.x >> 4);
        const float *row = colsum + (long)(ch < c ? ch : 0) * hw;
        float part = 0.f;
        for (int j = sub; j < hw; j += 64) {
            const float v0 = row[j], v1 = j + 16 < hw ? row[j + 16] : 0.f, v2 = j + 32 < hw ? row[j + 32] : 0.f, v3 = j + 48 < hw ? row[j + 48] : 0.f;
            part += (v0 + v1) + (v2 + v3);
        }
        part += __shfl_xor(part, 8, 64);
        part += __shfl_xor(part, 4, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 1, 64);
        if (sub == 0 && ch < c) {
            const float tot = part;
            gb[ch] = tot;
            if (ad.p) {
                const float step = adam_dev_step(ad), pv = ad.p[ch];
                const float gv = tot + ad.wd * pv;
                const float mn = ad.beta1 * ad.m[ch] + (1.0f - ad.beta1) * gv;
                const float vn = ad.beta2 * ad.v[ch] + (1.0f - ad.beta2) * gv * gv;
                ad.m[ch] = mn;
                ad.v[ch] = vn;
                ad.p[ch] = pv - step * mn / (sqrtf(vn) + ad.eps);
            }
        }
    }
}

// im2col of the reference's GENERAL path (tensor.rs:1805-1906 + copy_consecutive_elements 1910-1969), one thread per
// element of col[window][ch][k_row][k_col].  Restated as a closed form of the loops: within one kernel row the taps that
// fall inside the padded width form ONE run starting at k_col = first; the run is copied from CONSECUTIVE input columns
// starting at the first tap's column (dilation is not applied inside a run, 1866/1891), and the plane it reads is
// batch*ch + ch, not batch*c + ch (1931/1964, quirk Q9).  Elements outside the image stay 0 (col is zero-initialised, 1681).
__global__ __launch_bounds__(256) void im2col_general_kernel(const float *__restrict__ x, float *__restrict__ col, long total, int c,
                                                             int h_in, int w_in, int h_out, int w_out, int k_h, int k_w, int s_h,
                                                             int s_w, int pad_h, int pad_w, int dil_h, int dil_w) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        long r = e;
        const int k_col = (int)(r % k_w); r /= k_w;
        const int k_row = (int)(r % k_h); r /= k_h;
        const int ch = (int)(r % c); r /= c;
        const int ow = (int)(r % w_out); r /= w_out;
        const int oh = (int)(r % h_out);
        const long batch = r / h_out;
        float v = 0.f;
        const int in_h = oh * s_h + k_row * dil_h, in_w_tap = ow * s_w + k_col * dil_w;
        if (in_h >= pad_h && in_h < h_in + pad_h && in_w_tap >= pad_w && in_w_tap < w_in + pad_w) {
            const int lack = pad_w - ow * s_w;                                    // first tap of the run: smallest k with ow*s_w + k*dil_w >= pad_w
            const int first = lack > 0 ? (lack + dil_w - 1) / dil_w : 0;
            const int in_w = ow * s_w + first * dil_w - pad_w + (k_col - first);  // consecutive columns from the run's start
            if (in_w < w_in) v = x[((batch * ch + ch) * h_in + (in_h - pad_h)) * (long)w_in + in_w];
        }
        col[e] = v;
    }
}

int bias_grad_launch(th_ctx *ctx, const float *d_gout, const float *d_mask_y, float *d_gb, int n, int c, int hw, int accumulate, int pooled) {
    TH_REQUIRE(ctx && d_gout && d_gb, "th_bias_grad_nchw: null argument");
    if (c == 0) return 0;
    int slabs = 1;
    if ((long)n * hw >= 4096 && c < 256) {   // one workgroup per channel cannot fill the chip: split the images
        slabs = ceil_div(512, c);
        if (slabs > n) slabs = n;
    }
    const bool small_planes = pooled && d_mask_y && hw <= 64;
    if (small_planes && slabs > 1) {   // one plane per wave and iteration: latency-bound, so more, shorter slabs (<= 16 images each)
        slabs = std::min(ceil_div(n, 4), std::max(slabs, ceil_div(2048, c)));
    }
    if (slabs <= 1) {
        if (small_planes)
            hipLaunchKernelGGL(bias_grad_avgpool_small_kernel, dim3(c), dim3(256), 0, ctx->stream, d_gout, d_mask_y, d_gb, (float *)nullptr, n,
                               c, hw, n, accumulate ? 0 : 1);
        else
        hipLaunchKernelGGL(bias_grad_nchw_kernel, dim3(c), dim3(256), 0, ctx->stream, d_gout, d_mask_y, d_gb, (float *)nullptr, n, c, hw, n,
                           accumulate ? 0 : 1, pooled);
        TH_LAUNCH_CHECK();
        return 0;
    }
    const int ips = ceil_div(n, slabs);
    slabs = ceil_div(n, ips);
    void *part = nullptr;
    if (th_malloc(ctx, (size_t)slabs * c * sizeof(float), &part)) return 1;
    if (small_planes)
        hipLaunchKernelGGL(bias_grad_avgpool_small_kernel, dim3(c, slabs), dim3(256), 0, ctx->stream, d_gout, d_mask_y, d_gb, (float *)part, n,
                           c, hw, ips, 0);
    else
    hipLaunchKernelGGL(bias_grad_nchw_kernel, dim3(c, slabs), dim3(256), 0, ctx->stream, d_gout, d_mask_y, d_gb, (float *)part, n, c, hw,
                       ips, 0, pooled);
    TH_LAUNCH_CHECK();
    if (int rc = accumulate ? th_colsum_accum(ctx, (const float *)part, d_gb, slabs, c) : th_colsum(ctx, (const float *)part, d_gb, slabs, c))
        return rc;
    return th_free(ctx, part);
}
}  // namespace th

namespace th {
// Single-channel 3x3 convolution + bias + ReLU + 2x2 max pool (conv1 of both CNNs: 1 -> 32 channels on 28x28).
// K = 9: nothing for the matrix cores to do, and a workgroup of the general kernel spends its life in set-up.  Here a
// workgroup takes one image: the zero-haloed image sits in LDS, a thread owns one POOLED pixel -- its 4x4 input
// window stays in registers -- and walks the output channels (weights through uniform loads); 36 FMAs per channel
// in tap order (the k-ordered fmaf chain of the matrix-core kernel: same bits), only the pooled tensor is written.
// grid = (n, channel blocks of CO_W); weights [9][c_out] (the taper layout as is, tensor.rs:1262).
constexpr int C1_CO_W = 16;
__global__ __launch_bounds__(256) void conv1_pool2_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ bias, float *__restrict__ y, int h, int w_in,
                                                          int c_out, int pad, int relu) {
    extern __shared__ float img[];                       // [(h + 2)][(w_in + 2)], zero halo
    __shared__ float wsm[C1_CO_W][12];                   // this block's filters (9 taps) + bias
    const int wp = w_in + 2, hp = h + 2;
    const float *xi = x + (long)blockIdx.x * h * w_in;
    if (threadIdx.x < C1_CO_W * 10) {
        const int cl = threadIdx.x / 10, k = threadIdx.x % 10, co = blockIdx.y * C1_CO_W + cl;
        wsm[cl][k] = co < c_out ? (k < 9 ? w[k * c_out + co] : (bias ? bias[co] : 0.f)) : 0.f;
    }
    for (int e = threadIdx.x; e < hp * wp; e += 256) {
        const int r = e / wp - 1, c = e % wp - 1;
        img[e] = (r >= 0 && r < h && c >= 0 && c < w_in) ? xi[r * w_in + c] : 0.f;
    }
    __syncthreads();
    const int h_out = h + 2 * pad - 2, w_out = w_in + 2 * pad - 2, ph = h_out >> 1, pw = w_out >> 1;
    const int co0 = blockIdx.y * C1_CO_W, co1 = min(c_out, co0 + C1_CO_W);
    const int shift = 1 - pad;                           // pad = 0: the window starts one pixel in
    for (int p = threadIdx.x; p < ph * pw; p += 256) {
        const int pr = p / pw, pc = p % pw;
        float win[4][4];                                 // input rows 2 pr + shift .. + 3 (halo coordinates), cols 2 pc + shift ..
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) win[i][j] = img[(2 * pr + shift + i) * wp + 2 * pc + shift + j];
        float *yo = y + ((long)blockIdx.x * c_out + co0) * ph * pw + p;
#pragma unroll 4
        for (int co = co0; co < co1; ++co, yo += ph * pw) {
            float wk[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) wk[k] = wsm[co - co0][k];    // broadcast reads
            const float b = wsm[co - co0][9];
            float m = -INFINITY;                         // strict >: NaN never wins (tensor.rs:1449-1461)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 9; ++k) acc = fmaf(wk[k], win[dy + k / 3][dx + k % 3], acc);
                    float v = acc + b;
                    if (relu) v = v > 0.f ? v : 0.f;
                    m = v > m ? v : m;
                }
            *yo = m;
        }
    }
}
// the same without the pool (conv1 of the reference CNN feeds conv2 directly): a thread owns output pixels, 9 FMAs per channel
__global__ __launch_bounds__(256) void conv1_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                    float *__restrict__ y, int h, int w_in, int c_out, int pad, int relu) {
    extern __shared__ float img[];                       // [(h + 2)][(w_in + 2)], zero halo
    __shared__ float wsm[C1_CO_W][12];
    const int wp = w_in + 2, hp = h + 2;
    const float *xi = x + (long)blockIdx.x * h * w_in;
    if (threadIdx.x < C1_CO_W * 10) {
        const int cl = threadIdx.x / 10, k = threadIdx.x % 10, co = blockIdx.y * C1_CO_W + cl;
        wsm[cl][k] = co < c_out ? (k < 9 ? w[k * c_out + co] : (bias ? bias[co] : 0.f)) : 0.f;
    }
    for (int e = threadIdx.x; e < hp * wp; e += 256) {
        const int r = e / wp - 1, c = e % wp - 1;
        img[e] = (r >= 0 && r < h && c >= 0 && c < w_in) ? xi[r * w_in + c] : 0.f;
    }
    __syncthreads();
    const int h_out = h + 2 * pad - 2, w_out = w_in + 2 * pad - 2;
    const int co0 = blockIdx.y * C1_CO_W, co1 = min(c_out, co0 + C1_CO_W);
    const int shift = 1 - pad;
    for (int p = threadIdx.x; p < h_out * w_out; p += 256) {
        const int r = p / w_out, c = p % w_out;
        float win[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) win[k] = img[(r + shift + k / 3) * wp + c + shift + k % 3];
        float *yo = y + ((long)blockIdx.x * c_out + co0) * h_out * w_out + p;
#pragma unroll 4
        for (int co = co0; co < co1; ++co, yo += h_out * w_out) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = fmaf(wsm[co - co0][k], win[k], acc);
            float v = acc + wsm[co - co0][9];
            if (relu) v = v > 0.f ? v : 0.f;
            *yo = v;
        }
    }
}
}  // namespace th

extern "C" {

int th_conv3x3_fwd(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, float *d_y, int n, int c_in, int h,
                   int w, int c_out, int pad, int weight_layout, int relu) {
    TH_REQUIRE(ctx && d_x && d_w && d_y, "th_conv3x3_fwd: null argument");
    TH_REQUIRE(n > 0 && c_in > 0 && c_out > 0 && h + 2 * pad >= 3 && w + 2 * pad >= 3, "th_conv3x3_fwd: bad geometry");
    TH_REQUIRE(pad == 0 || pad == 1, "th_conv3x3_fwd: pad must be 0 or 1 (got %d)", pad);
    TH_REQUIRE(weight_layout == 0 || weight_layout == 1, "th_conv3x3_fwd: weight_layout must be 0 (taper) or 1 (standard)");
    if (c_in == 1 && weight_layout == 0 && (size_t)(h + 2) * (w + 2) * sizeof(float) <= (48u << 10)) {
        // conv1: K = 9 -- one image per workgroup on the vector ALUs (17.3 -> 8 us at batch 256 against the matrix-core kernel)
        const size_t lds = (size_t)(h + 2) * (w + 2) * sizeof(float);
        hipLaunchKernelGGL(conv1_kernel, dim3(n, ceil_div(c_out, C1_CO_W)), dim3(256), lds, ctx->stream, d_x, d_w, d_bias, d_y, h, w, c_out,
                           pad, relu);
        TH_LAUNCH_CHECK();
        return 0;
    }
    const bool mfma = conv3x3_mfma_supported(c_in, h, w, pad);
    // the taper layout IS the [k][co] slab the matrix-core kernel stages (tensor.rs:1262): read it in place
    if (mfma && weight_layout == 0 && c_out % 4 == 0 && ((uintptr_t)d_w & 15) == 0)
        return conv3x3_mfma_launch(ctx, d_x, d_w, c_out, c_out, d_bias, d_y, n, c_in, h, w, c_out, pad, relu, false);
    const int co_pad = (c_out + CO_R - 1) / CO_R * CO_R;   // multiple of both kernels' channel blocks
    void *wt = nullptr;
    if (th_malloc(ctx, (size_t)c_in * 9 * co_pad * sizeof(float), &wt)) return 1;
    hipLaunchKernelGGL(conv3x3_prep_weights, dim3(ew_grid((size_t)c_in * 9 * co_pad, 256)), dim3(256), 0, ctx->stream, d_w,
                       (float *)wt, c_in, c_out, weight_layout, 0, c_out, co_pad, c_in);
    TH_LAUNCH_CHECK();
    if (mfma) {
        if (int rc = conv3x3_mfma_launch(ctx, d_x, (const float *)wt, co_pad, co_pad, d_bias, d_y, n, c_in, h, w, c_out, pad, relu, false)) return rc;
    } else if (int rc = conv3x3_launch(ctx, d_x, (const float *)wt, d_bias, d_y, n, c_in, h, w, c_out, co_pad, pad, relu, false)) {
        return rc;
    }
    return th_free(ctx, wt);
}

int th_conv3x3_pool2_supported(int c_in, int h, int w, int c_out, int pad) {
    return (pad == 0 || pad == 1) && c_in > 0 && c_out > 0 && c_out % 4 == 0 && h + 2 * pad >= 3 && w + 2 * pad >= 3 &&
           conv3x3_mfma_pool_supported(c_in, h, w, pad);
}

int th_conv3x3_pool2_fwd(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, float *d_y_pooled, int n, int c_in, int h,
                         int w, int c_out, int pad, int relu) {
    TH_REQUIRE(ctx && d_x && d_w && d_y_pooled && n > 0, "th_conv3x3_pool2_fwd: null argument");
    TH_REQUIRE(th_conv3x3_pool2_supported(c_in, h, w, c_out, pad) && ((uintptr_t)d_w & 15) == 0,
               "th_conv3x3_pool2_fwd: needs the matrix-core path (c_in >= 8 or == 1), c_out %% 4 == 0, even output height / width, "
               "rows of <= 64 outputs and 16-byte aligned weights");
    if (c_in == 1 && (size_t)(h + 2) * (w + 2) * sizeof(float) <= (48u << 10)) {   // conv1: one image per workgroup, VALU
        const size_t lds = (size_t)(h + 2) * (w + 2) * sizeof(float);
        hipLaunchKernelGGL(conv1_pool2_kernel, dim3(n, ceil_div(c_out, C1_CO_W)), dim3(256), lds, ctx->stream, d_x, d_w, d_bias, d_y_pooled,
                           h, w, c_out, pad, relu);
        TH_LAUNCH_CHECK();
        return 0;
    }
    // the taper layout IS the [k][co] slab the kernel stages (tensor.rs:1262)
    return conv3x3_mfma_launch(ctx, d_x, d_w, c_out, c_out, d_bias, d_y_pooled, n, c_in, h, w, c_out, pad, relu, false, true);
}

int th_conv3x3_gap_supported(int n, int c_in, int h, int w, int c_out, int pad) {
    return (pad == 0 || pad == 1) && n > 0 && c_in > 0 && c_out > 0 && h + 2 * pad >= 3 && w + 2 * pad >= 3 &&
           conv3x3_gap_supported(n, c_in, h, w, c_out, pad) ? 1 : 0;
}

int th_conv3x3_gap_fwd(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, float *d_y_mean, float *d_cnt, int n, int c_in,
                       int h, int w, int c_out, int pad, int relu) {
    TH_REQUIRE(ctx && d_x && d_w && d_y_mean && n > 0, "th_conv3x3_gap_fwd: null argument");
    TH_REQUIRE(th_conv3x3_gap_supported(n, c_in, h, w, c_out, pad) && ((uintptr_t)d_w & 15) == 0,
               "th_conv3x3_gap_fwd: needs c_in %% 8 == 0, c_out %% 4 == 0, whole images that fit a workgroup's LDS and 16-byte aligned weights");
    return conv3x3_mfma_launch(ctx, d_x, d_w, c_out, c_out, d_bias, d_y_mean, n, c_in, h, w, c_out, pad, relu, false, true, d_cnt, true);
}

int th_conv3x3_bwd_input(th_ctx *ctx, const float *d_gy, const float *d_w, float *d_gx, int n, int c_in, int h, int w,
                         int c_out, int pad, int weight_layout, int accumulate) {
    TH_REQUIRE(ctx && d_gy && d_w && d_gx, "th_conv3x3_bwd_input: null argument");
    TH_REQUIRE(pad == 1, "th_conv3x3_bwd_input: only pad=1 (same-size) convolutions are supported");
    // gx (+)= conv3x3(gy, mirrored filter with ci/co swapped), pad 1
    const int ci_pad = (c_in + CO_R - 1) / CO_R * CO_R;
    void *wt = nullptr;
    if (th_malloc(ctx, (size_t)c_out * 9 * ci_pad * sizeof(float), &wt)) return 1;
    hipLaunchKernelGGL(conv3x3_prep_weights, dim3(ew_grid((size_t)c_out * 9 * ci_pad, 256)), dim3(256), 0, ctx->stream, d_w,
                       (float *)wt, c_in, c_out, weight_layout, 1, c_in, ci_pad, c_out);
    TH_LAUNCH_CHECK();
    if (conv3x3_mfma_supported(c_out, h, w, 1)) {   // the mirrored filter [k = (co, kh, kw)][ci] feeds the matrix-core kernel as is
        if (int rc = conv3x3_mfma_launch(ctx, d_gy, (const float *)wt, ci_pad, ci_pad, nullptr, d_gx, n, c_out, h, w, c_in, 1, 0, accumulate != 0)) return rc;
    } else if (int rc = conv3x3_launch(ctx, d_gy, (const float *)wt, nullptr, d_gx, n, c_out, h, w, c_in, ci_pad, 1, 0, accumulate != 0)) {
        return rc;
    }
    return th_free(ctx, wt);
}

int th_conv3x3_bwd_weight(th_ctx *ctx, const float *d_x, const float *d_gy, float *d_gw, int n, int c_in, int h, int w,
                          int c_out, int pad, int weight_layout, int accumulate) {
    TH_REQUIRE(ctx && d_x && d_gy && d_gw, "th_conv3x3_bwd_weight: null argument");
    TH_REQUIRE(pad == 0 || pad == 1, "th_conv3x3_bwd_weight: pad must be 0 or 1");
    const int h_out = h + 2 * pad - 2, w_out = w + 2 * pad - 2;
    if (c_in >= 8 && conv3x3_mfma_supported(c_in, h, w, pad) && (long)n * h_out * w_out >= 2048)   // enough channels and pixels to contract over
        return conv3x3_wgrad_mfma_launch(ctx, d_x, d_gy, d_gw, n, c_in, h, w, c_out, pad, weight_layout, accumulate);
    {
        // one input channel, planes that fit the LDS (conv1): one image per workgroup, partial slabs per image
        const size_t lds = ((size_t)(((h + 2) * (w + 2) + 3) & ~3) + (size_t)C1W_CO * (h_out * w_out + 1) + (size_t)C1W_NS * C1W_CO * 9) * sizeof(float);
        static const bool off = getenv("TAPER_CONV1_WGRAD") && getenv("TAPER_CONV1_WGRAD")[0] == '0';   // measurement / parity knob
        if (!off && c_in == 1 && n >= 32 && h_out > 0 && w_out > 0 && lds <= (160u << 10)) {
            void *part = nullptr;
            if (th_malloc(ctx, (size_t)n * 9 * c_out * sizeof(float), &part)) return 1;
            TH_SET_MAX_LDS(ctx, conv1_wgrad_kernel, 160 << 10);
            hipLaunchKernelGGL(conv1_wgrad_kernel, dim3(n, ceil_div(c_out, C1W_CO)), dim3(C1W_NT), lds, ctx->stream, d_x, d_gy, (float *)part, h, w,
                               c_out, pad, h_out, w_out);
            TH_LAUNCH_CHECK();
            if (int rc = wgrad_reduce(ctx, (const float *)part, d_gw, n, 9, c_out, c_out, weight_layout, accumulate)) return rc;
            return th_free(ctx, part);
        }
    }
    int slabs = 1;
    if ((long)c_out * c_in < 512 && (long)n * h_out * w_out >= 8192) {   // few (co, ci) pairs, many pixels (conv1): split the images
        slabs = ceil_div(1024, c_out * c_in);
        if (slabs > n) slabs = n;
    }
    if (slabs <= 1) {
        hipLaunchKernelGGL(conv3x3_bwd_weight_kernel, dim3(c_out, c_in), dim3(256), 0, ctx->stream, d_x, d_gy, d_gw, (float *)nullptr, n,
                           c_in, h, w, c_out, pad, h_out, w_out, weight_layout, n, accumulate);
        TH_LAUNCH_CHECK();
        return 0;
    }
    const int ips = ceil_div(n, slabs);
    slabs = ceil_div(n, ips);
    void *part = nullptr;
    if (th_malloc(ctx, (size_t)slabs * c_in * 9 * c_out * sizeof(float), &part)) return 1;
    hipLaunchKernelGGL(conv3x3_bwd_weight_kernel, dim3(c_out, c_in, slabs), dim3(256), 0, ctx->stream, d_x, d_gy, d_gw, (float *)part, n,
                       c_in, h, w, c_out, pad, h_out, w_out, weight_layout, ips, 0);
    TH_LAUNCH_CHECK();
    if (int rc = wgrad_reduce(ctx, (const float *)part, d_gw, slabs, c_in * 9, c_out, c_out, weight_layout, accumulate)) return rc;
    return th_free(ctx, part);
}

int th_conv1x1_fwd(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, float *d_y, int n, int c_in, int h,
                   int w, int c_out, int weight_layout, int relu) {
    TH_REQUIRE(ctx && d_x && d_w && d_y, "th_conv1x1_fwd: null argument");
    const int hw = h * w;
    if (weight_layout == 0) {
        // taper: col = raw NCHW buffer viewed [N*H*W, C_in] (tensor.rs:1799-1801, Q4);
        // w viewed [C_in, C_out] (tensor.rs:1262, Q3); out2d = col . w2d -> NHWC -> NCHW (+bias)
        void *tmp = nullptr;
        if (th_malloc(ctx, (size_t)n * hw * c_out * sizeof(float), &tmp)) return 1;
        if (int rc = th_sgemm(ctx, 0, 0, n * hw, c_out, c_in, 1.0f, d_x, d_w, 0.0f, (float *)tmp)) return rc;
        hipLaunchKernelGGL(nhwc_to_nchw_bias_kernel, dim3(ceil_div(c_out, 64), ceil_div(hw, 64), n), dim3(256), 0, ctx->stream,
                           (const float *)tmp, d_bias, d_y, hw, c_out, relu);
        TH_LAUNCH_CHECK();
        return th_free(ctx, tmp);
    }
    // standard: Y_n[C_out, HW] = W[C_out, C_in] . X_n[C_in, HW] per image
    for (int i = 0; i < n; ++i) {
        if (int rc = th_sgemm(ctx, 0, 0, c_out, hw, c_in, 1.0f, d_w, d_x + (size_t)i * c_in * hw, 0.0f,
                              d_y + (size_t)i * c_out * hw))
            return rc;
    }
    if (d_bias || relu) {
        const long total = (long)n * c_out * hw;
        if (d_bias) {
            hipLaunchKernelGGL(bias_add_nchw_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, ctx->stream, (const float *)d_y,
                               d_bias, d_y, total, c_out, hw, relu);
            TH_LAUNCH_CHECK();
        } else {
            return th_relu_fwd(ctx, d_y, d_y, (size_t)total);
        }
    }
    return 0;
}

int th_conv2d_general_fwd(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, float *d_y, int n, int c_in, int h,
                          int w, int c_out, int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w, int dil_h, int dil_w, int relu) {
    TH_REQUIRE(ctx && d_x && d_w && d_y, "th_conv2d_general_fwd: null argument");
    TH_REQUIRE(n > 0 && c_in > 0 && c_out > 0 && k_h > 0 && k_w > 0 && s_h > 0 && s_w > 0 && dil_h > 0 && dil_w > 0 && pad_h >= 0 && pad_w >= 0,
               "th_conv2d_general_fwd: bad geometry");
    TH_REQUIRE(h + 2 * pad_h >= dil_h * (k_h - 1) + 1 && w + 2 * pad_w >= dil_w * (k_w - 1) + 1, "th_conv2d_general_fwd: kernel larger than the padded input");
    const int h_out = (h + 2 * pad_h - dil_h * (k_h - 1) - 1) / s_h + 1, w_out = (w + 2 * pad_w - dil_w * (k_w - 1) - 1) / s_w + 1;
    const long windows = (long)n * h_out * w_out, k = (long)c_in * k_h * k_w;
    TH_REQUIRE(windows < (1L << 31) && k < (1L << 31), "th_conv2d_general_fwd: col matrix too large");
    void *col = nullptr, *tmp = nullptr;
    if (th_malloc(ctx, (size_t)(windows * k) * sizeof(float), &col)) return 1;
    if (th_malloc(ctx, (size_t)windows * c_out * sizeof(float), &tmp)) return 1;
    hipLaunchKernelGGL(im2col_general_kernel, dim3(ew_grid((size_t)(windows * k), 256)), dim3(256), 0, ctx->stream, d_x, (float *)col, windows * k,
                       c_in, h, w, h_out, w_out, k_h, k_w, s_h, s_w, pad_h, pad_w, dil_h, dil_w);
    TH_LAUNCH_CHECK();
    // out2d[windows, c_out] = col . w viewed [k, c_out] (tensor.rs:1262, Q3) -> NHWC -> NCHW (+bias, tensor.rs:1272-1279)
    if (int rc = th_sgemm(ctx, 0, 0, (int)windows, c_out, (int)k, 1.0f, (const float *)col, d_w, 0.0f, (float *)tmp)) return rc;
    hipLaunchKernelGGL(nhwc_to_nchw_bias_kernel, dim3(ceil_div(c_out, 64), ceil_div(h_out * w_out, 64), n), dim3(256), 0, ctx->stream,
                       (const float *)tmp, d_bias, d_y, h_out * w_out, c_out, relu);
    TH_LAUNCH_CHECK();
    if (int rc = th_free(ctx, col)) return rc;
    return th_free(ctx, tmp);
}

int th_bias_add_nchw(th_ctx *ctx, const float *d_x, const float *d_bias, float *d_y, int n, int c, int hw, int relu) {
    TH_REQUIRE(ctx && d_x && d_bias && d_y, "th_bias_add_nchw: null argument");
    const long total = (long)n * c * hw;
    if (total == 0) return 0;
    hipLaunchKernelGGL(bias_add_nchw_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, ctx->stream, d_x, d_bias, d_y, total, c, hw, relu);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_bias_grad_nchw(th_ctx *ctx, const float *d_gout, float *d_gb, int n, int c, int hw) {
    return th_bias_grad_nchw_masked(ctx, d_gout, nullptr, d_gb, n, c, hw, 1);
}

int th_bias_grad_nchw_masked(th_ctx *ctx, const float *d_gout, const float *d_mask_y, float *d_gb, int n, int c, int hw, int accumulate) {
    return th::bias_grad_launch(ctx, d_gout, d_mask_y, d_gb, n, c, hw, accumulate, 0);
}

int th_bias_grad_masked_adam(th_ctx *ctx, const float *d_gout, const float *d_mask_y, float *d_gb, int n, int c, int hw, int pooled_avg,
                             const th_adam_fuse *b_fuse, const th_adam_slice *extra, int n_extra) {
    TH_REQUIRE(ctx && d_gout && d_gb && n > 0 && c > 0 && hw > 0, "th_bias_grad_masked_adam: null argument / empty tensor");
    TH_REQUIRE(!pooled_avg || d_mask_y, "th_bias_grad_masked_adam: the average-pool form needs the conv output as mask");
    TH_REQUIRE(n_extra >= 0 && n_extra <= TH_MAX_ADAM_SLICES && (n_extra == 0 || extra), "th_bias_grad_masked_adam: bad extra slices");
    const AdamSlices x = make_adam_slices(extra, n_extra, ctx);
    hipLaunchKernelGGL(bias_grad_adam_kernel, dim3(c + x.blocks()), dim3(1024), 0, ctx->stream, d_gout, d_mask_y, d_gb, n, c, hw,
                       pooled_avg ? 1 : 0, make_adam_dev(b_fuse), x);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_bias_from_colsum_adam(th_ctx *ctx, const float *d_colsum, float *d_gb, int c, int hw, const th_adam_fuse *b_fuse,
                             const th_adam_slice *extra, int n_extra) {
    TH_REQUIRE(ctx && d_colsum && d_gb && c > 0 && hw > 0, "th_bias_from_colsum_adam: null argument / empty tensor");
    TH_REQUIRE(n_extra >= 0 && n_extra <= TH_MAX_ADAM_SLICES && (n_extra == 0 || extra), "th_bias_from_colsum_adam: bad extra slices");
    const AdamSlices x = make_adam_slices(extra, n_extra, ctx);
    hipLaunchKernelGGL(bias_from_colsum_adam_kernel, dim3(1 + x.blocks()), dim3(1024), 0, ctx->stream, d_colsum, d_gb, c, hw, make_adam_dev(b_fuse), x);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_bias_grad_avgpool_masked(th_ctx *ctx, const float *d_gout_pooled, const float *d_mask_y, float *d_gb, int n, int c, int hw,
                                int accumulate) {
    TH_REQUIRE(d_mask_y, "th_bias_grad_avgpool_masked: null argument");
    return th::bias_grad_launch(ctx, d_gout_pooled, d_mask_y, d_gb, n, c, hw, accumulate, 1);
}


int th_maxpool2d_fwd(th_ctx *ctx, const float *d_x, float *d_y, int64_t *d_argmax, int n, int c, int h, int w, int k_h, int k_w,
                     int s_h, int s_w, int pad_h, int pad_w) {
    TH_REQUIRE(ctx && d_x && d_y, "th_maxpool2d_fwd: null argument");
    if (s_h == 0) { s_h = k_h; s_w = k_w; }
    TH_REQUIRE(k_h > 0 && k_w > 0 && s_h > 0 && s_w > 0 && h + 2 * pad_h >= k_h && w + 2 * pad_w >= k_w, "th_maxpool2d_fwd: bad geometry");
    const int h_out = (h + 2 * pad_h - k_h) / s_h + 1, w_out = (w + 2 * pad_w - k_w) / s_w + 1;
    const long total = (long)n * c * h_out * w_out;
    if (total == 0) return 0;
    if (maxpool2_fast(n, c, h, w, k_h, k_w, s_h, s_w, pad_h, pad_w) && (((uintptr_t)d_x) & 7) == 0) {
        hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, ctx->stream, d_x, d_y, d_argmax, total, h, w);
        TH_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, ctx->stream, d_x, d_y, d_argmax, total, h, w,
                       h_out, w_out, k_h, k_w, s_h, s_w, pad_h, pad_w);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_maxpool2d_bwd(th_ctx *ctx, const float *d_gout, const int64_t *d_argmax, float *d_gin, int n, int c, int h, int w,
                     int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w, int zero_first) {
    TH_REQUIRE(ctx && d_gout && d_argmax && d_gin, "th_maxpool2d_bwd: null argument");
    if (s_h == 0) { s_h = k_h; s_w = k_w; }
    TH_REQUIRE(k_h > 0 && k_w > 0 && s_h > 0 && s_w > 0, "th_maxpool2d_bwd: bad geometry");
    const int h_out = (h + 2 * pad_h - k_h) / s_h + 1, w_out = (w + 2 * pad_w - k_w) / s_w + 1;
    const long total = (long)n * c * h * w;
    if (total == 0) return 0;
    static const bool fast_off = getenv("TAPER_POOL_BWD_GENERAL") && getenv("TAPER_POOL_BWD_GENERAL")[0] == '1';   // measurement / parity knob
    if (!fast_off && maxpool2_fast(n, c, h, w, k_h, k_w, s_h, s_w, pad_h, pad_w) && (((uintptr_t)d_gin) & 7) == 0) {
        const int planes = n * c;
        hipLaunchKernelGGL(maxpool2_bwd_kernel<false>, dim3(std::min(ceil_div(planes, 4), 8 * kNumCU)), dim3(256), 0, ctx->stream, d_gout, d_argmax,
                           (const float *)nullptr, (const float *)nullptr, d_gin, (float *)nullptr, planes, h, w, zero_first);
        TH_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(maxpool_bwd_geo_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, ctx->stream, d_gout, d_argmax, d_gin,
                       total, h, w, h_out, w_out, k_h, k_w, s_h, s_w, pad_h, pad_w, zero_first);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_maxpool2d_relu_bwd_supported(int n, int c, int h, int w, int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w) {
    if (s_h == 0) { s_h = k_h; s_w = k_w; }
    return maxpool2_fast(n, c, h, w, k_h, k_w, s_h, s_w, pad_h, pad_w) ? 1 : 0;
}

int th_maxpool2d_relu_bwd(th_ctx *ctx, const float *d_gout, const int64_t *d_argmax, const float *d_y_pooled, const float *d_y_full,
                          float *d_gin, float *d_plane_sums, int n, int c, int h, int w) {
    TH_REQUIRE(ctx && d_gout && d_argmax && d_y_pooled && d_y_full && d_gin, "th_maxpool2d_relu_bwd: null argument");
    TH_REQUIRE(maxpool2_fast(n, c, h, w, 2, 2, 2, 2, 0, 0) && (((uintptr_t)d_gin) & 7) == 0,
               "th_maxpool2d_relu_bwd: 2x2 windows, stride 2, no padding, even height and width, 8-byte aligned gradient (got %d x %d)", h, w);
    const int planes = n * c;
    hipLaunchKernelGGL(maxpool2_bwd_kernel<true>, dim3(std::min(ceil_div(planes, 4), 8 * kNumCU)), dim3(256), 0, ctx->stream, d_gout, d_argmax,
                       d_y_pooled, d_y_full, d_gin, d_plane_sums, planes, h, w, 1);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_relu_bwd_plane_sums(th_ctx *ctx, const float *d_y, const float *d_gout, float *d_gin, float *d_plane_sums, int n, int c, int hw) {
    TH_REQUIRE(ctx && d_y && d_gout && d_gin && d_plane_sums, "th_relu_bwd_plane_sums: null argument");
    TH_REQUIRE(n >= 0 && c >= 0 && hw > 0 && (long)n * c < (1L << 31), "th_relu_bwd_plane_sums: bad geometry");
    TH_REQUIRE((hw & 3) != 0 || ((((uintptr_t)d_y | (uintptr_t)d_gout | (uintptr_t)d_gin) & 15) == 0), "th_relu_bwd_plane_sums: maps must be 16-byte aligned");
    const int planes = n * c;
    if (planes == 0) return 0;
    hipLaunchKernelGGL(relu_bwd_planes_kernel, dim3(std::min(ceil_div(planes, 4), 8 * kNumCU)), dim3(256), 0, ctx->stream, d_y, d_gout, d_gin,
                       d_plane_sums, planes, hw);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_avgpool2d_global_relu_bwd(th_ctx *ctx, const float *d_gout, const float *d_y, float *d_gin, float *d_plane_sums, int n, int c, int hw) {
    TH_REQUIRE(ctx && d_gout && d_y && d_gin && n >= 0 && c >= 0 && hw > 0 && (long)n * c < (1L << 31), "th_avgpool2d_global_relu_bwd: bad argument");
    const int planes = n * c;
    if (planes == 0) return 0;
    hipLaunchKernelGGL(gap_relu_bwd_planes_kernel, dim3(std::min(ceil_div(planes, 4), 8 * kNumCU)), dim3(256), 0, ctx->stream, d_y, d_gout, d_gin,
                       d_plane_sums, planes, hw);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_bias_grad_plane_sums(th_ctx *ctx, const float *d_plane_sums, float *d_gb, int n, int c, int accumulate) {
    TH_REQUIRE(ctx && d_plane_sums && d_gb && n >= 0 && c > 0, "th_bias_grad_plane_sums: bad argument");
    hipLaunchKernelGGL(bias_from_plane_sums_kernel, dim3(ceil_div(c, 16)), dim3(256), 0, ctx->stream, d_plane_sums, d_gb, n, c, accumulate);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_avgpool2d_fwd(th_ctx *ctx, const float *d_x, float *d_y, int n, int c, int h, int w, int k_h, int k_w, int s_h, int s_w,
                     int pad_h, int pad_w) {
    TH_REQUIRE(ctx && d_x && d_y, "th_avgpool2d_fwd: null argument");
    if (s_h == 0) { s_h = k_h; s_w = k_w; }
    TH_REQUIRE(k_h > 0 && k_w > 0 && s_h > 0 && s_w > 0 && h + 2 * pad_h >= k_h && w + 2 * pad_w >= k_w, "th_avgpool2d_fwd: bad geometry");
    const int h_out = (h + 2 * pad_h - k_h) / s_h + 1, w_out = (w + 2 * pad_w - k_w) / s_w + 1;
    const long total = (long)n * c * h_out * w_out;
    if (total == 0) return 0;
    if (k_h == h && k_w == w && pad_h == 0 && pad_w == 0) {  // global pool: one wave per plane
        th::avgpool_global_launch(ctx, d_x, d_y, nullptr, (long)n * c, h * w);
    } else {
        hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, ctx->stream, d_x, d_y, total, h, w, h_out,
                           w_out, k_h, k_w, s_h, s_w, pad_h, pad_w);
    }
    TH_LAUNCH_CHECK();
    return 0;
}

int th_avgpool2d_global_fwd_counts(th_ctx *ctx, const float *d_x, float *d_y, float *d_cnt, int n, int c, int hw) {
    TH_REQUIRE(ctx && d_x && d_y && d_cnt && hw > 0, "th_avgpool2d_global_fwd_counts: null argument");
    if ((long)n * c == 0) return 0;
    th::avgpool_global_launch(ctx, d_x, d_y, d_cnt, (long)n * c, hw);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_bias_grad_counts_adam(th_ctx *ctx, const float *d_gout_pooled, const float *d_cnt, float *d_gb, int n, int c, int hw,
                             const th_adam_fuse *b_fuse, const th_adam_slice *extra, int n_extra) {
    TH_REQUIRE(ctx && d_gout_pooled && d_cnt && d_gb && n > 0 && c > 0 && hw > 0, "th_bias_grad_counts_adam: null argument / empty tensor");
    TH_REQUIRE(n_extra >= 0 && n_extra <= TH_MAX_ADAM_SLICES && (n_extra == 0 || extra), "th_bias_grad_counts_adam: bad extra slices");
    const AdamSlices x = make_adam_slices(extra, n_extra, ctx);
    hipLaunchKernelGGL(bias_grad_counts_adam_kernel, dim3(ceil_div(c, 16) + x.blocks()), dim3(1024), 0, ctx->stream, d_gout_pooled, d_cnt,
                       d_gb, n, c, hw, make_adam_dev(b_fuse), x);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_avgpool2d_bwd(th_ctx *ctx, const float *d_gout, float *d_gin, int n, int c, int h, int w, int k_h, int k_w, int s_h,
                     int s_w, int pad_h, int pad_w) {
    TH_REQUIRE(ctx && d_gout && d_gin, "th_avgpool2d_bwd: null argument");
    if (s_h == 0) { s_h = k_h; s_w = k_w; }
    const int h_out = (h + 2 * pad_h - k_h) / s_h + 1, w_out = (w + 2 * pad_w - k_w) / s_w + 1;
    const long total = (long)n * c * h * w;
    if (total == 0) return 0;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, ctx->stream, d_gout, d_gin, total, h, w, h_out,
                       w_out, k_h, k_w, s_h, s_w, pad_h, pad_w);
    TH_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

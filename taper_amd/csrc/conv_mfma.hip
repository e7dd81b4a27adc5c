// conv_mfma.hip -- direct 3x3 convolution (stride 1, pad 0/1) on the fp32 matrix cores for layers
// with C_in >= 8 (src/tensor.rs:1221-1285 conv2d, 1728-1780 im2col_3x3, 1972-2076 the layout
// round trips -- none of which is materialised here).
//
// Why MFMA: with C_in >= 32 the layer is a dense contraction over K = 9*C_in = 288..576 at
// 20-130 flop/B -- compute-bound, and v_mfma_f32_16x16x4_f32 is exact fp32 (a k-ordered fmaf
// chain), so the parity bar is the same as for the VALU kernel (conv_pool.hip), which stays the
// path for conv1 (C_in = 1: K = 9, HBM-bound).  It is still a DIRECT convolution: the im2col
// matrix exists only as LDS addresses.
//
//   D[co][px] = sum_k Wc[k][co] * X[px, k],   k = (ci, kh, kw)
//   A operand (16 x 4):  lane l -> Wc[4s + (l>>4)][co0 + (l&15)]          (weights, LDS [k][co])
//   B operand (4 x 16):  lane l -> patch[(ci,kh,kw) of k = 4s + (l>>4)] at pixel px0 + (l&15)
//   D tile (16 x 16):    lane l, i -> co = co0 + 4*(l>>4) + i, px = px0 + (l&15)
//
// A workgroup owns IMG images x R output rows x all W_out columns (<= 128 pixels = 8 pixel tiles,
// two per wave) and CO_B <= 128 output channels; per pass CI_T = 8 input channels are staged:
// the zero-haloed input patch [8][IMG][R+2][W_out+2] and the weight slab [72][CO_B].  The global
// loads of pass p+1 are issued before the MFMAs of pass p and parked in registers (one round trip per
// pass, hidden behind 18 k-steps), then written to the single LDS buffer between two barriers.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace th {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int MF_CI = 8;              // input channels per pass  -> 72 k = 18 MFMA k-steps
constexpr int MF_PX_MAX = 128;        // pixels per workgroup (8 tiles of 16)

constexpr int MF_OOB = 0x7ffffff0;     // byte offset beyond every buffer descriptor's range
constexpr int MF_PPT = 12;            // patch elements per thread per pass (patch <= 3072 floats)

// Exact quotient / remainder of small non-negative ints (e < 2^22, d >= 1) through one float multiply and one
// correction step: the per-thread staging plans below need ~40 divisions by launch-time constants, and a
// 32-bit integer division is ~35 VALU instructions on gfx950.
struct FastDiv {
    int d;
    float inv;
    __host__ __device__ explicit FastDiv(int dd) : d(dd), inv(1.0f / (float)dd) {}
    __device__ __forceinline__ void divmod(int e, int &q, int &r) const {
        q = (int)((float)e * inv);
        r = e - q * d;
        if (r < 0) { --q; r += d; }
        else if (r >= d) { ++q; r -= d; }
    }
};

struct ConvMfmaArgs {
    const float *x, *w, *bias;
    float *y;
    int n, c_in, h, w_in, c_out, pad, h_out, w_out;
    int w_ld, w_cols;      // weights [K = 9*c_in][w_ld] floats, k = ci*9 + kh*3 + kw, columns [0, w_cols) readable;
                           // 16-B aligned rows (w_ld % 4 == 0)
    int img_t, rows_t;     // images / output rows per workgroup
    int bands;             // row bands per image group = ceil(h_out / rows_t)
    int co_b;              // output channels per workgroup (multiple of 16)
    int relu;
};

#ifdef TH_PROFILE
__device__ long long g_conv_prof[8];
__device__ long long g_conv_tl[4 * 4096];   // per workgroup: start, end of k loop, end (100 MHz wall clock), hardware id
__device__ long long g_conv_clk[2 * 4096];  // shader clock (clock64) at start / end
#define CONV_TL(slot) do { const int id_ = blockIdx.y * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && id_ < 4096) { g_conv_tl[4 * id_ + (slot)] = wall_clock64(); if ((slot) != 1) g_conv_clk[2 * id_ + ((slot) >> 1)] = clock64(); } } while (0)
#define CONV_TL_HW() do { const int id_ = blockIdx.y * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && id_ < 4096) g_conv_tl[4 * id_ + 3] = ((long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); } while (0)
#define CONV_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 300 && blockIdx.y == 0) g_conv_prof[i] = wall_clock64(); } while (0)
#define IMG_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 100) { g_conv_prof[i] = wall_clock64(); if ((i) == 2 || (i) == 3) g_conv_prof[3 + (i)] = clock64(); } } while (0)
#else
#define IMG_STAMP(i) do { } while (0)
#define CONV_STAMP(i) do { } while (0)
#define CONV_TL(slot) do { } while (0)
#define CONV_TL_HW() do { } while (0)
#endif

// POOL: the epilogue additionally applies a 2x2 / stride-2 max pool (tensor.rs:1391-1470 values; no index output) and
// stores ONLY the pooled tensor [n][c_out][h_out/2][w_out/2] -- the Conv2dReLU -> MaxPool2d pair of the CNNs without
// the full-resolution round trip (needs even rows_t, h_out, w_out).
// WH = 2: 512 threads -- waves 4..7 take the upper half of the channel tiles on the SAME pixels and staged operands, so a
// SIMD holds two of the workgroup's waves: twice the waves per SIMD with no extra staging per MFMA (the layers with 14x14
// maps launch < 2 workgroups per CU and their matrix pipes idled 56 % of the time).
// (<= 32-channel blocks with 4 waves: capped at 128 VGPRs so that 4 workgroups share a CU -- 4 spilled registers buy 66.8 -> 62.9 us
// on the 28x28 layer; a cap of 102 for 5 workgroups spills the plans and costs 50 %)
// DMA: the operands of a pass go global -> LDS directly (`buffer_load ... lds`: no staging registers, no ds_write pass, no
// per-element address / select arithmetic -- halo and out-of-range elements are offsets past the buffer's end, which the
// descriptor's range check turns into zeros), into the other of two LDS stages while the current one feeds the MFMAs: one
// barrier per pass.  Needs whole 8-channel blocks (c_in % 8 == 0).
template <int CT, bool ACCUM, int CIT, bool POOL = false, int WH = 1, bool DMA = false>   // CT = co_b / 16; CIT = input channels per pass: 8, or 1 for single-channel inputs (conv1)
__global__ __launch_bounds__(256 * WH, (CT <= 2 && WH == 1) ? 4 : 1) void conv3x3_mfma_kernel(ConvMfmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CO_B = 16 * CT;
    constexpr int KS = (CIT * 9 + 3) / 4;                      // k-steps per pass: 18, or 3 (k = 9 padded to 12)
    constexpr int WQ = KS * 4 * CO_B / 4;                      // float4 quads in the weight slab (rows past 9*CIT are zero)
    constexpr int NT = 256 * WH, PPT = MF_PPT / WH, CTW = CT / WH;   // threads, patch elements per thread per pass, channel tiles per wave
    constexpr int WPT = (WQ + NT - 1) / NT;                    // quads per thread per pass
    CONV_STAMP(0);
    CONV_TL(0);
    CONV_TL_HW();
    const int t = threadIdx.x, lane = t & 63, wave = (t >> 6) & 3, chalf = t >> 8;   // pixel group, channel half
    const int l16 = lane & 15, g4 = lane >> 4;
    const int wp = a.w_out + 2, rp = a.rows_t + 2;           // patch pitch / rows (input window of the band)
    const int img_stride = rp * wp, ci_stride = a.img_t * img_stride;
    const int patch_n = CIT * ci_stride;
    float *patch = lds;                                        // [CIT][img_t][rp][wp], then one zero float
    float *wsl = lds + ((patch_n + 4) & ~3);                   // [4 KS][CO_B], 16-B aligned
    const int grp = blockIdx.x / a.bands, band = blockIdx.x % a.bands;
    const int img0 = grp * a.img_t, oh0 = band * a.rows_t;
    const int co0 = blockIdx.y * CO_B;
    const int rows_here = min(a.rows_t, a.h_out - oh0);
    const int px_per_img = a.rows_t * a.w_out;
    const int m_wg = a.img_t * px_per_img;                     // <= 128 (rows past rows_here are discarded)

    const FastDiv d_pxi(px_per_img), d_wout(a.w_out), d_cis(ci_stride), d_ims(img_stride), d_wp(wp);
    // this lane's B column in its two pixel tiles: LDS offset of the window's top-left corner
    int pix_off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int p = (wave + 4 * q) * 16 + l16;
        int il, rem, r, c;
        d_pxi.divmod(p < m_wg ? p : 0, il, rem);
        d_wout.divmod(rem, r, c);
        pix_off[q] = il * img_stride + r * wp + c;
    }
    // LDS offset of tap k = 4s + g4 relative to the window corner, for the 18 k-steps of a pass
    int koff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int kk = 4 * s + g4, cl = kk / 9, tap = kk % 9;
        koff[s] = cl * ci_stride + (tap / 3) * wp + (tap % 3);
    }
    // padded k (>= 9*CIT, only when CIT = 1): the weight rows are zero, and the pixel operand must be a finite
    // number too (0 * NaN would poison the sum): point it at the zero float behind the patch
    int pix_off_pad[2] = {pix_off[0], pix_off[1]};
    if (CIT * 9 % 4 != 0 && 4 * (KS - 1) + g4 >= CIT * 9) {
        koff[KS - 1] = 0;
        pix_off_pad[0] = pix_off_pad[1] = patch_n;
    }
    if (threadIdx.x == 0) patch[patch_n] = 0.f;
    // Staging plan, fixed for the whole kernel (only the channel base moves between passes): patch
    // element e = t + 256 j -> global offset within channel block 0 (or -1: halo / tail -> zero) and
    // its local channel; weight quad u = t + 256 j -> row kk and column quad.
    const int shift = 1 - a.pad;                               // pad = 0: the window starts one pixel in
    const long chan = (long)a.h * a.w_in;
    int p_goff[PPT];   // (offset << 3) | local channel, or -1
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int e = t + NT * j;
        p_goff[j] = -1;
        if (e < patch_n) {
            int cl, r1, il, r2, rr, cc;
            d_cis.divmod(e, cl, r1);
            d_ims.divmod(r1, il, r2);
            d_wp.divmod(r2, rr, cc);
            const int img = img0 + il, ih = oh0 + rr - 1 + shift, iw = cc - 1 + shift;
            if (img < a.n && ih >= 0 && ih < a.h && iw >= 0 && iw < a.w_in)
                p_goff[j] = DMA ? (int)(((((long)il * a.c_in + cl) * a.h + ih) * a.w_in + iw) << 2)        // byte offset
                                : (int)((((((long)il * a.c_in + cl) * a.h + ih) * a.w_in + iw) << 3) | cl);   // relative to image img0, channel cb
        }
        if (DMA && p_goff[j] < 0) p_goff[j] = MF_OOB;        // past the end of any buffer: reads as zero
    }
    const float *xbase = a.x + (long)img0 * a.c_in * chan;

    floatx4 acc[2][CTW];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < CTW; ++j) acc[q][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    if constexpr (DMA) {
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-descriptor builtins exist in the device pass only
        typedef __attribute__((address_space(3))) void *lds_ptr_t;
        const int stage_n = ((patch_n + 4) & ~3) + 4 * KS * CO_B;      // floats per stage: patch, then the weight slab
        const int w0 = __builtin_amdgcn_readfirstlane(t >> 6) * 64;     // this wave's first thread (uniform)
        int w_boff[WPT];                                               // weight quad u = t + NT j -> byte offset in channel block 0's slab
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const int u = t + NT * j, kk = u / (CO_B / 4), cq = (u % (CO_B / 4)) * 4;
            w_boff[j] = (u < WQ && co0 + cq + 3 < a.w_cols) ? (kk * a.w_ld + co0 + cq) << 2 : MF_OOB;
        }
        const int imgs_here = min(a.img_t, a.n - img0);
        const int x_bytes = (int)((long)imgs_here * a.c_in * chan * 4), w_bytes = 9 * a.c_in * a.w_ld * 4;
        auto issue = [&](int cb, int stage) {
            // descriptors rebuilt per pass from uniform values: base at channel block cb, range = what is left behind it
            const auto rx = __builtin_amdgcn_make_buffer_rsrc((void *)(xbase + (long)cb * chan), 0, x_bytes - (int)(cb * chan * 4), 0x00020000);
            const auto rw = __builtin_amdgcn_make_buffer_rsrc((void *)(a.w + (long)cb * 9 * a.w_ld), 0, w_bytes - cb * 9 * a.w_ld * 4, 0x00020000);
            float *st = lds + stage * stage_n;
            // whole rounds of NT elements go out unmasked (a uniform test); only the ragged last round is lane-masked
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                if (NT * (j + 1) <= patch_n)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(st + w0 + NT * j), 4, p_goff[j], 0, 0, 0);
                else if (NT * j < patch_n && t + NT * j < patch_n)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(st + w0 + NT * j), 4, p_goff[j], 0, 0, 0);
            }
            float *ws = st + ((patch_n + 4) & ~3);
#pragma unroll
            for (int j = 0; j < WPT; ++j) {
                if (NT * (j + 1) <= WQ)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(ws + 4 * (w0 + NT * j)), 16, w_boff[j], 0, 0, 0);
                else if (t + NT * j < WQ)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(ws + 4 * (w0 + NT * j)), 16, w_boff[j], 0, 0, 0);
            }
        };
        CONV_STAMP(1);
        issue(0, 0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        CONV_STAMP(2);
        int stage = 0;
        for (int cb = 0; cb < a.c_in; cb += CIT, stage ^= 1) {
            const bool more = cb + CIT < a.c_in;
            if (more) issue(cb + CIT, stage ^ 1);     // the other stage was last read before the barrier that ended the previous pass
            const float *pp = lds + stage * stage_n, *ww = pp + ((patch_n + 4) & ~3);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float b0 = pp[pix_off[0] + koff[s]];
                const float b1 = pp[pix_off[1] + koff[s]];
                const float *wk = ww + (4 * s + g4) * CO_B + l16 + 16 * CTW * chalf;
#pragma unroll
                for (int j = 0; j < CTW; ++j) {
                    const float av = wk[16 * j];
                    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[1][j], 0, 0, 0);
                }
            }
            if (more) {
                __builtin_amdgcn_s_waitcnt(0);        // this wave's part of the next stage has landed
                __syncthreads();                      // ... and everybody's; every wave is done reading this stage
            }
        }
#endif
    } else {
    float pv[PPT];
    float4 wv[WPT];
    // all global loads of a pass are issued back to back (one round trip), held in registers while the
    // previous pass computes, then written to LDS
#define TH_MF_LOAD(CB)                                                                                           \
    {                                                                                                            \
        const float *xc = xbase + (long)(CB) * chan;                                                             \
        _Pragma("unroll") for (int j = 0; j < PPT; ++j) {   /* always-valid address + select: no branch per load */   \
            const bool ok = p_goff[j] >= 0 && (CB) + (p_goff[j] & 7) < a.c_in;                                    \
            const float v = xc[ok ? (p_goff[j] >> 3) : 0];                                                        \
            pv[j] = ok ? v : 0.f;                                                                                \
        }                                                                                                        \
        _Pragma("unroll") for (int j = 0; j < WPT; ++j) {                                                        \
            const int u = t + NT * j, kk = u / (CO_B / 4), cq = (u % (CO_B / 4)) * 4;                              \
            const int k = (CB) * 9 + kk, co = co0 + cq;                                                          \
            const bool ok = u < WQ && kk < CIT * 9 && k < a.c_in * 9 && co + 3 < a.w_cols;                       \
            const float4 v = *reinterpret_cast<const float4 *>(a.w + (ok ? (long)k * a.w_ld + co : 0));          \
            wv[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
        }                                                                                                        \
    }
#define TH_MF_STORE()                                                                                            \
    {                                                                                                            \
        _Pragma("unroll") for (int j = 0; j < PPT; ++j) {                                                        \
            const int e = t + NT * j;                                                                            \
            if (e < patch_n) patch[e] = pv[j];                                                                   \
        }                                                                                                        \
        _Pragma("unroll") for (int j = 0; j < WPT; ++j) {                                                        \
            const int u = t + NT * j;                                                                            \
            if (u < WQ) *reinterpret_cast<float4 *>(wsl + 4 * u) = wv[j];                                        \
        }                                                                                                        \
    }
    CONV_STAMP(1);
    TH_MF_LOAD(0)
    TH_MF_STORE()
    __syncthreads();
    CONV_STAMP(2);
    for (int cb = 0; cb < a.c_in; cb += CIT) {
        const bool more = cb + CIT < a.c_in;
        if (more) TH_MF_LOAD(cb + CIT)
        // ---- 18 k-steps: per step 2 pixel fragments and CT weight fragments feed 2*CT MFMAs ----
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float b0 = patch[(s == KS - 1 ? pix_off_pad[0] : pix_off[0]) + koff[s]];
            const float b1 = patch[(s == KS - 1 ? pix_off_pad[1] : pix_off[1]) + koff[s]];
            const float *wk = wsl + (4 * s + g4) * CO_B + l16 + 16 * CTW * chalf;
#pragma unroll
            for (int j = 0; j < CTW; ++j) {
                const float av = wk[16 * j];
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[1][j], 0, 0, 0);
            }
        }
        if (more) {
            __syncthreads();   // every wave is done reading this pass
            TH_MF_STORE()
            __syncthreads();
        }
    }
#undef TH_MF_LOAD
#undef TH_MF_STORE
    }
    CONV_STAMP(3);
    CONV_TL(1);

    if (POOL) {
        // ---- pooled epilogue: bias + ReLU into an LDS tile [CO_B][pixels], then 2x2 maxima straight to the pooled tensor ----
        __syncthreads();                                   // every wave is done with the staged operands
        const int ep_ld = m_wg | 1;                        // odd pitch
        float *ep = lds;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = (wave + 4 * q) * 16 + l16;
            if (p >= m_wg) continue;
#pragma unroll
            for (int j = 0; j < CTW; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cl = 16 * (CTW * chalf + j) + 4 * g4 + i, co = co0 + cl;
                    float v = acc[q][j][i] + ((a.bias && co < a.c_out) ? a.bias[co] : 0.f);
                    if (a.relu) v = v > 0.f ? v : 0.f;
                    ep[cl * ep_ld + p] = v;
                }
        }
        __syncthreads();
        const int pw = a.w_out >> 1, np_img = (a.rows_t >> 1) * pw, np = a.img_t * np_img;
        const FastDiv d_np(np), d_npi(np_img), d_pw(pw);
        const long pchan = (long)(a.h_out >> 1) * pw;
        for (int idx = t; idx < CO_B * np; idx += NT) {
            int cl, rem, il, r2, pr, pc;
            d_np.divmod(idx, cl, rem);
            d_npi.divmod(rem, il, r2);
            d_pw.divmod(r2, pr, pc);
            if (co0 + cl >= a.c_out || img0 + il >= a.n || 2 * pr >= rows_here) continue;
            const float *b = ep + cl * ep_ld + il * px_per_img + 2 * pr * a.w_out + 2 * pc;
            float m = -INFINITY;                           // strict >: NaN never wins (tensor.rs:1449-1461)
            m = b[0] > m ? b[0] : m;
            m = b[1] > m ? b[1] : m;
            m = b[a.w_out] > m ? b[a.w_out] : m;
            m = b[a.w_out + 1] > m ? b[a.w_out + 1] : m;
            a.y[((long)(img0 + il) * a.c_out + co0 + cl) * pchan + (long)((oh0 >> 1) + pr) * pw + pc] = m;
        }
        CONV_STAMP(4);
        CONV_TL(2);
        return;
    }
    // ---- epilogue: bias + ReLU (tensor.rs:2005-2025, nn.rs:433-490), NCHW store straight from the D tiles: lane
    //      (l16, g4) holds pixel px0 + l16 of channels co0 + 4*g4 + i.  (Staging the tiles through LDS for row-
    //      contiguous stores was measured and is no faster: the tail of a workgroup is the MFMA queue draining.) ----
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int p = (wave + 4 * q) * 16 + l16;
        int il = 0, rem = 0, r = 0, c = 0;
        if (p < m_wg) {
            d_pxi.divmod(p, il, rem);
            d_wout.divmod(rem, r, c);
        }
        if (!(p < m_wg && r < rows_here && img0 + il < a.n)) continue;
        float *ypx = a.y + (((long)(img0 + il) * a.c_out) * a.h_out + (oh0 + r)) * a.w_out + c;
        const long ochan = (long)a.h_out * a.w_out;
#pragma unroll
        for (int j = 0; j < CTW; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + 16 * (CTW * chalf + j) + 4 * g4 + i;
                if (co >= a.c_out) continue;
                float v = acc[q][j][i] + (a.bias ? a.bias[co] : 0.f);
                if (a.relu) v = v > 0.f ? v : 0.f;
                float *dst = ypx + (long)co * ochan;
                if (ACCUM) *dst += v;
                else *dst = v;
            }
    }
    CONV_STAMP(4);
    CONV_TL(2);
}

// ------------------------------------------------------------------------------------------------
// conv3x3_img_kernel -- the same contraction with the loops turned inside out, for launches big enough to give every
// CU a whole unit of work (the batch-256 layers of the CNNs): a workgroup owns IMG WHOLE IMAGES x CO_B output
// channels and keeps EVERY output tile of the unit in registers -- wave w holds pixel tiles w, w + 4, ... (<= TPW of
// them) x CT channel tiles -- while the input channels stream through LDS eight at a time.  Per kernel (not per 128
// pixels): one set of staging plans, c_in / 8 barriers, one epilogue; the weight slab of a pass is staged once for
// all of the unit's pixels.  The launch planner picks (IMG, CT) so that the units tile the 256 CUs evenly (one image
// of a 28x28 layer, four of a 14x14 layer, eight of a 7x7 layer = 49 / 49 / 25 pixel tiles), which the
// 128-pixel kernel above cannot do: its 430 workgroups on the 14x14 layers run as 2 + 1 per CU (r01: 0.34-0.44 of the
// fp32 matrix peak, a workgroup's fixed prologue / epilogue ~9 us against 9-18 us of MFMA work).
// Same operand layouts, same k order (exact fp32 fmaf chains in the same order: bit-identical to the kernel above),
// same LDS-DMA staging with zero halos from the buffer descriptor's range check.
constexpr int IM_PPT = 16;            // patch elements per thread per pass (patch <= 8192 floats, 512 threads)

struct ConvImgArgs {
    const float *x, *w, *bias;
    float *y;
    int n, c_in, h, w_in, c_out, pad, h_out, w_out, w_ld, w_cols;
    int img_t;             // images per unit
    int n_groups, n_co;    // image groups, channel blocks
    int relu;
    int tile_store;        // plain instances: go through an LDS tile [CO_B][pixels] and store whole planes as float4 (needs px per image % 4 == 0)
    float *gap_cnt;        // gap != 0: per (image, channel) count of outputs > 0, as floats (nullable)
    int gap;               // POOL instances only: the epilogue is a GLOBAL AVERAGE pool (y = [n][c_out] plane means) instead of the 2x2 max-pool
    const int *goff_tab;   // [512 * IM_PPT] patch element -> byte offset within channel block 0 of the unit's first image (or past-the-end)
    const int *pix_tab;    // [pixel tile slots * 16] pixel -> LDS offset of its window corner
};

// The per-thread staging plans are the same for every workgroup of a launch and for every launch of a geometry: patch element
// e -> (channel, image, row, column) -> global byte offset or "halo", pixel p -> window corner.  Worked out in every workgroup they
// cost ~40 float-reciprocal divisions per thread -- 5.2 us of a 52 us launch with two waves per SIMD; as tables (built on device once
// per geometry, cached in the ctx) they are 16 + TPW coalesced loads.  Images past the batch's end need no entry of their own: their
// offsets lie beyond the range of the unit's buffer descriptor and read as zero.
__global__ void conv_img_tables_kernel(int *goff, int *pix, int n_goff, int n_pix, int c_in, int h, int w_in, int pad, int img_t) {
    const int h_out = h + 2 * pad - 2, w_out = w_in + 2 * pad - 2, wp = w_out + 2, rp = h_out + 2;
    const int img_stride = rp * wp, ci_stride = img_t * img_stride, patch_n = MF_CI * ci_stride, shift = 1 - pad;
    const int px_img = h_out * w_out, m_full = img_t * px_img;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_goff + n_pix; e += gridDim.x * blockDim.x) {
        if (e < n_goff) {
            int v = MF_OOB;
            if (e < patch_n) {
                const int cl = e / ci_stride, r1 = e % ci_stride, il = r1 / img_stride, r2 = r1 % img_stride, rr = r2 / wp, cc = r2 % wp;
                const int ih = rr - 1 + shift, iw = cc - 1 + shift;
                if (ih >= 0 && ih < h && iw >= 0 && iw < w_in) v = (int)(((((long)il * c_in + cl) * h + ih) * w_in + iw) << 2);
            }
            goff[e] = v;
        } else {
            const int p0 = e - n_goff, p = p0 < m_full ? p0 : 0;
            const int il = p / px_img, rem = p % px_img;
            pix[p0] = il * img_stride + (rem / w_out) * wp + rem % w_out;
        }
    }
}

// fp32 MFMA on gfx950 runs on the vector ALUs ("= the FP32 vector rate"): experiments/mfma_overlap.hip shows that NOTHING a wave -- or its
// SIMD partner -- issues to the VALU or the LDS hides under an MFMA: 13 MFMAs + 27 v_add take 1092 cycles for two waves of a SIMD where
// the MFMAs alone take 832, in phase or in anti-phase; a v_add costs ~4.8 cycles of the SIMD, a ds_read_b32 ~9.  The k loop is therefore
// priced per k-step and wave as 32 P Q + 9 (P + Q) + 4.8 P cycles (P pixel tiles x Q channel tiles: P + Q operand reads, P address adds),
// and what pays is operand REUSE: a wave that owns Q = 2 channel tiles feeds two MFMAs from every pixel operand.
//   NW = 8: 8 waves, every wave ONE channel tile (CT = 2: waves 0-3 / 4-7 take tile 0 / 1 on pixel groups 0-3; CT = 1: 8 pixel groups)
//   NW = 4: 4 waves, every wave ALL CT channel tiles on its pixel group (tile = wave + 4 i)
// The launch planner prices both and picks per layer.
// WP > 0: the patch geometry (pitch WP = w_out + 2, channel stride CIS = images x (h_out + 2) x WP) is a compile-time constant and
// the 72 k of a pass are walked as (channel group of 4, tap): k-step s covers tap s % 9 of channels 4 (s / 9) + lane group.  A lane's
// operand address is then (its window corner + its channel offset) + a per-step constant that fits the ds_read offset field -- the P
// address adds per k-step go away (4.8 cycles each of the same ALUs that execute the MFMAs).  The weight slab is read through the same
// permutation.  Sums
// are formed in another order than in the generic instances (WP = 0, k = channel-major like the reference's im2col column): equal within
// fp32 rounding, not bit for bit.
template <int CT, int TPW, bool POOL, int NW, int WP = 0, int CIS = 0>
__global__ __launch_bounds__(64 * NW, 1) void conv3x3_img_kernel(ConvImgArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CO_B = 16 * CT, KS = 18, NT = 64 * NW, WQ = KS * 4 * CO_B / 4, WPT = (WQ + NT - 1) / NT;
    constexpr int CTW = NW == 4 ? CT : 1;                    // channel tiles per wave
    constexpr int PG = NW == 4 ? 4 : (CT == 2 ? 4 : 8);      // pixel groups of waves
    constexpr int PPT = IM_PPT * 512 / NT;                   // patch elements per thread per pass
    // units of one image group sit on one XCD (blocks go round-robin over the 8 XCDs): its channel blocks share the input in that L2
    const int xcd = blockIdx.x & 7, q8 = blockIdx.x >> 3;
    const int grp = xcd + 8 * (q8 / a.n_co), cob = q8 % a.n_co;
    if (grp >= a.n_groups) return;
    IMG_STAMP(0);
    CONV_TL(0);
    CONV_TL_HW();
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l16 = lane & 15, g4 = lane >> 4;
    // pixel group, first channel tile of this wave.  Waves w and w + 4 share a SIMD: with two channel tiles the two waves of a pixel group
    // (the only group that owns a pixel tile more than the others when the tiles do not divide evenly) sit on DIFFERENT SIMDs
    const int pg = NW == 8 && CT == 2 ? wave >> 1 : wave % PG, cj = NW == 4 ? 0 : (CT == 2 ? wave & 1 : wave / PG);
    const int wp = a.w_out + 2, rp = a.h_out + 2;
    const int img_stride = rp * wp, ci_stride = a.img_t * img_stride, patch_n = MF_CI * ci_stride;
    const int img0 = grp * a.img_t, co0 = cob * CO_B;
    const int imgs_here = min(a.img_t, a.n - img0);
    const int px_img = a.h_out * a.w_out, m_unit = imgs_here * px_img;
    const FastDiv d_pxi(px_img);

    int pix_off[TPW];                 // LDS offset of the window corner of this lane's pixel in each of its tiles
#pragma unroll
    for (int i = 0; i < TPW; ++i) pix_off[i] = a.pix_tab[(pg + PG * i) * 16 + l16];
    int p_goff[PPT];                  // byte offset within channel block 0 of image img0, or past-the-end (reads as zero)
#pragma unroll
    for (int j = 0; j < PPT; ++j) p_goff[j] = a.goff_tab[t + NT * j];
    int koff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (WP > 0) {   // compile-time part of the offset (channel group s / 9, tap s % 9); the lane part (g4 * CIS) is folded into pix_off below
            koff[s] = (4 * (s / 9)) * CIS + ((s % 9) / 3) * WP + ((s % 9) % 3);
        } else {
            const int kk = 4 * s + g4, cl = kk / 9, tap = kk % 9;
            koff[s] = cl * ci_stride + (tap / 3) * wp + (tap % 3);
        }
    }
    if (WP > 0) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) pix_off[i] += g4 * CIS;
    }
    const long chan = (long)a.h * a.w_in;
    const float *xbase = a.x + (long)img0 * a.c_in * chan;
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    const int stage_n = ((patch_n + 4) & ~3) + 4 * KS * CO_B;
    const int w0 = wave * 64;
    int w_boff[WPT];
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        const int u = t + NT * j, kk = u / (CO_B / 4), cq = (u % (CO_B / 4)) * 4;
        w_boff[j] = (u < WQ && co0 + cq + 3 < a.w_cols) ? (kk * a.w_ld + co0 + cq) << 2 : MF_OOB;
    }
    const int x_bytes = (int)((long)imgs_here * a.c_in * chan * 4), w_bytes = 9 * a.c_in * a.w_ld * 4;
    auto issue = [&](int cb, int stage) {
        const auto rx = __builtin_amdgcn_make_buffer_rsrc((void *)(xbase + (long)cb * chan), 0, x_bytes - (int)(cb * chan * 4), 0x00020000);
        const auto rw = __builtin_amdgcn_make_buffer_rsrc((void *)(a.w + (long)cb * 9 * a.w_ld), 0, w_bytes - cb * 9 * a.w_ld * 4, 0x00020000);
        float *st = lds + stage * stage_n;
        float *ws = st + ((patch_n + 4) & ~3);
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            if (NT * (j + 1) <= WQ)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(ws + 4 * (w0 + NT * j)), 16, w_boff[j], 0, 0, 0);
            else if (NT * j < WQ && t + NT * j < WQ)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(ws + 4 * (w0 + NT * j)), 16, w_boff[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            if (NT * (j + 1) <= patch_n)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(st + w0 + NT * j), 4, p_goff[j], 0, 0, 0);
            else if (NT * j < patch_n && t + NT * j < patch_n)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(st + w0 + NT * j), 4, p_goff[j], 0, 0, 0);
        }
    };
    const bool last_slot = (pg + PG * (TPW - 1)) * 16 < m_unit;     // wave-uniform
    floatx4 acc[TPW][CTW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int j = 0; j < CTW; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    IMG_STAMP(1);
    issue(0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    IMG_STAMP(2);
    int stage = 0;
    for (int cb = 0; cb < a.c_in; cb += MF_CI, stage ^= 1) {
        const bool more = cb + MF_CI < a.c_in;
        if (more) issue(cb + MF_CI, stage ^ 1);
        const float *pp = lds + stage * stage_n, *ww = pp + ((patch_n + 4) & ~3) + l16 + 16 * cj;
        // The operands of k-step s + 1 are requested before the MFMAs of step s are issued, and pinned there: left alone, the scheduler
        // sinks every read to its use -- one exposed LDS round trip per tile.  (Tried on the 28x28 layer, 8 waves, 50.9 us in this form:
        // one request per MFMA interleaved through sched_group_barrier 62 us; SIMD partners in anti-phase 51.7 us; the next stage's
        // LDS-DMA instructions spread one per k-step 52.2 us -- none of it can win, see the note on fp32 MFMA above.)
        float b0[TPW], b1[TPW], a0[CTW], a1[CTW];
        // slab row of (k-step, lane group): channel-major k = 4 s + g4, or the tap-major permutation (channel 4 (s % 2) + g4, tap s / 2)
        auto wrow = [&](int ks) { return WP > 0 ? (4 * (ks / 9) + g4) * 9 + ks % 9 : 4 * ks + g4; };
#ifdef TH_IMG_NO_READS   /* timing probe: the k loop without its LDS requests (wrong results) */
#define TH_IMG_REQ(B, A, S) { _Pragma("unroll") for (int i = 0; i < TPW; ++i) B[i] = __builtin_bit_cast(float, pix_off[i] + (S)); \
                              _Pragma("unroll") for (int j = 0; j < CTW; ++j) A[j] = __builtin_bit_cast(float, koff[S] + j); }
#else
#define TH_IMG_REQ(B, A, S)                                                                  \
        {                                                                                    \
            _Pragma("unroll") for (int i = 0; i < TPW; ++i) B[i] = pp[pix_off[i] + koff[S]]; \
            _Pragma("unroll") for (int j = 0; j < CTW; ++j) A[j] = ww[wrow(S) * CO_B + 16 * j];        \
        }
#endif
        // (two k-steps per round, so that the requests of consecutive taps pair up into ds_read2_b32 -- 167 instead of 371 non-MFMA
        // instructions per pass in the 13-tile instance -- measured SLOWER: 46.9 vs 43.6 us on the pooled 28x28 layer)
        TH_IMG_REQ(b0, a0, 0)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) TH_IMG_REQ(b1, a1, s + 1)
            __builtin_amdgcn_sched_barrier(0);
            // The LAST tile slot of a wave is skipped (a scalar branch: `wave` is uniform) when it lies past the unit's pixel tiles: 49 tiles
            // over 8 pixel groups are 7 + 6 x 7 -- one wave has 7, the others 6, so the busiest SIMD issues 13 MFMAs per k-step instead of 14.
            // (Earlier slots past the end -- ragged last image group -- compute on pixel 0 and are never stored.)
#pragma unroll
            for (int i = 0; i < TPW - 1; ++i)
#pragma unroll
                for (int j = 0; j < CTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b0[i], acc[i][j], 0, 0, 0);
            if (last_slot) {
#pragma unroll
                for (int j = 0; j < CTW; ++j) acc[TPW - 1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b0[TPW - 1], acc[TPW - 1][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < KS) {
#pragma unroll
                for (int i = 0; i < TPW; ++i) b0[i] = b1[i];
#pragma unroll
                for (int j = 0; j < CTW; ++j) a0[j] = a1[j];
            }
        }
#undef TH_IMG_REQ
        if (more) {
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
        }
    }

    IMG_STAMP(3);
    CONV_TL(1);
    // bias of this lane's channels (co0 + 16 (cj + j) + 4 g4 + e)
    float bv[CTW][4];
#pragma unroll
    for (int j = 0; j < CTW; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = co0 + 16 * (cj + j) + 4 * g4 + e;
            bv[j][e] = (a.bias && co < a.c_out) ? a.bias[co] : 0.f;
        }
    if (POOL) {
        // bias + ReLU into an LDS tile [CO_B][pixels of the unit], then the 2x2 / stride-2 maxima go to the pooled tensor
        __syncthreads();
        const int ep_ld = (a.img_t * px_img) | 1;
        float *ep = lds;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int p = (pg + PG * i) * 16 + l16;
            if (p >= m_unit) continue;
#pragma unroll
            for (int j = 0; j < CTW; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][j][e] + bv[j][e];
                    if (a.relu) v = v > 0.f ? v : 0.f;
                    ep[(16 * (cj + j) + 4 * g4 + e) * ep_ld + p] = v;
                }
        }
        __syncthreads();
        if (a.gap) {
            // Conv2dReLU -> global average pool (tensor.rs:1524-1660 with kernel = plane, nn.rs:670-686): the plane never leaves the CU.
            // 16 lanes per (channel, image) plane, lane l adds elements l, l + 16, ... and a shuffle tree joins them -- the arithmetic of
            // avgpool_global16_kernel on the stored map, so the fused form gives the same bits; the count of outputs > 0 per plane is what
            // the conv's bias gradient needs of the map (th_bias_grad_counts_adam).
            const int l = t & 15;
            for (int pl = t >> 4; pl < CO_B * imgs_here; pl += NT / 16) {
                const int cl = pl / imgs_here, il = pl - cl * imgs_here;
                const float *row = ep + cl * ep_ld + il * px_img;
                float sum = 0.f, k = 0.f;
                for (int i = l; i < px_img; i += 16) {
                    const float v = row[i];
                    sum += v;
                    k += v > 0.f ? 1.f : 0.f;
                }
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) {
                    sum += __shfl_down(sum, off, 16);
                    k += __shfl_down(k, off, 16);
                }
                if (l == 0 && co0 + cl < a.c_out) {
                    const long o = (long)(img0 + il) * a.c_out + co0 + cl;
                    a.y[o] = sum / (float)px_img;
                    if (a.gap_cnt) a.gap_cnt[o] = k;
                }
            }
            IMG_STAMP(4);
            CONV_TL(2);
            return;
        }
        // a thread owns pooled positions q = t, t + NT, ... of the unit (decoded ONCE: image, row, column -> LDS and global offsets) and walks the
        // CO_B channels over them: 4 LDS reads, 3 compares and a store per output.  (Decoding (channel, image, row, column) per output cost
        // three float-reciprocal divisions each: 9.0 us of epilogue on the 28x28 layer.)
        const int pw = a.w_out >> 1, np_img = (a.h_out >> 1) * pw, np = imgs_here * np_img;
        const FastDiv d_npi(np_img), d_pw(pw);
        for (int q = t; q < np; q += NT) {
            int il, r2, pr, pc;
            d_npi.divmod(q, il, r2);
            d_pw.divmod(r2, pr, pc);
            const float *b = ep + il * px_img + 2 * pr * a.w_out + 2 * pc;
            float *yo = a.y + ((long)(img0 + il) * a.c_out + co0) * np_img + r2;
            const int c_here = min(CO_B, a.c_out - co0);
#pragma unroll 4
            for (int cl = 0; cl < c_here; ++cl, b += ep_ld, yo += np_img) {
                float m = -INFINITY;                       // strict >: NaN never wins (tensor.rs:1449-1461)
                m = b[0] > m ? b[0] : m;
                m = b[1] > m ? b[1] : m;
                m = b[a.w_out] > m ? b[a.w_out] : m;
                m = b[a.w_out + 1] > m ? b[a.w_out + 1] : m;
                *yo = m;
            }
        }
        IMG_STAMP(4);
        CONV_TL(2);
        return;
    }
    const long ochan = (long)px_img;
    if (a.tile_store) {
        // bias + ReLU into an LDS tile [CO_B][pixels of the unit] (pitch a multiple of 4 floats + 4: the four row groups of a wave land 16
        // banks apart), then every (channel, image) plane -- contiguous in the tile and in the NCHW tensor -- leaves as float4: 1 KB per wave
        // store instead of 64-byte segments per channel (28x28 layer: 25.7 MB left in 10.5 us of epilogue)
        __syncthreads();
        const int tp = ((a.img_t * px_img + 3) & ~3) + 4;
        float *ep = lds;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int p = (pg + PG * i) * 16 + l16;
            if (p >= m_unit) continue;
#pragma unroll
            for (int j = 0; j < CTW; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][j][e] + bv[j][e];
                    if (a.relu) v = v > 0.f ? v : 0.f;
                    ep[(16 * (cj + j) + 4 * g4 + e) * tp + p] = v;
                }
        }
        __syncthreads();
        const int q_img = px_img >> 2, q_unit = imgs_here * q_img, c_here = min(CO_B, a.c_out - co0);
        const FastDiv d_q(q_img);
        for (int q = t; q < q_unit; q += NT) {             // quad q of the unit's pixels: image il, quad r of its plane
            int il, r;
            d_q.divmod(q, il, r);
            const float *src = ep + il * px_img + 4 * r;
            float *dst = a.y + ((long)(img0 + il) * a.c_out + co0) * ochan + 4 * r;
#pragma unroll 4
            for (int cl = 0; cl < c_here; ++cl, src += tp, dst += ochan) *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(src);
        }
        IMG_STAMP(4);
        CONV_TL(2);
        return;
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int p = (pg + PG * i) * 16 + l16;
        if (p >= m_unit) continue;
        int il, rem;
        d_pxi.divmod(p, il, rem);
        float *ypx = a.y + ((long)(img0 + il) * a.c_out) * ochan + rem;
#pragma unroll
        for (int j = 0; j < CTW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + 16 * (cj + j) + 4 * g4 + e;
                if (co >= a.c_out) continue;
                float v = acc[i][j][e] + bv[j][e];
                if (a.relu) v = v > 0.f ? v : 0.f;
                ypx[(long)co * ochan] = v;
            }
    }
    IMG_STAMP(4);
    CONV_TL(2);
#endif
}

// (IMG, CT, TPW, NW) of the image-resident kernel for a launch, or false: the 128-pixel kernel keeps it.  A candidate costs
// rounds of units over the 256 CUs x waves per SIMD x (32 P Q + 9 (P + Q) + 4.8 P) cycles per k-step (see the kernel's header);
// the instance table fixes P to 4 / 7 / 13.
struct ConvImgPlan { int img_t, ct, tpw, nw, n_groups, n_co; size_t lds; int tile_store; };
static bool conv_img_plan(int n, int c_in, int h, int w_in, int c_out, int pad, bool pool, ConvImgPlan *out, bool gap = false) {
    const int h_out = h + 2 * pad - 2, w_out = w_in + 2 * pad - 2;
    if (c_in % MF_CI != 0 || h_out < 1 || w_out < 1) return false;
    if (pool && !gap && ((h_out | w_out) & 1)) return false;
    static const int nw_env = getenv("TAPER_CONV_IMG_NW") ? atoi(getenv("TAPER_CONV_IMG_NW")) : 0;   // tuning probe: force 4 / 8 waves
    double best = -1;
    for (int nw = 4; nw <= 8; nw += 4) {
        // measured: the 4-wave form (every pixel operand feeds two MFMAs, but one wave per SIMD) is 10-17 % SLOWER on all four batch-256
        // layers (28x28: 59.2 vs 50.4 us): it only runs when asked for
        if (nw != (nw_env ? nw_env : 8)) continue;
        for (int ct = 1; ct <= 2; ++ct) {
            if (ct == 2 && c_out <= 16) continue;
            const int pg = nw == 4 ? 4 : (ct == 2 ? 4 : 8), q = nw == 4 ? ct : 1;
            for (int img = 1; img <= 16; ++img) {
                const long px = (long)img * h_out * w_out;
                const int npt = (int)((px + 15) / 16), tpw_need = (npt + pg - 1) / pg;
                const int tpw = tpw_need <= 4 ? 4 : (tpw_need <= 7 ? 7 : (tpw_need <= 13 ? 13 : 0));
                if (!tpw) break;
                const long patch_n = (long)MF_CI * img * (h_out + 2) * (w_out + 2);
                if (patch_n > (long)IM_PPT * 512) break;
                size_t lds = (size_t)2 * ((((size_t)patch_n + 4) & ~(size_t)3) + (size_t)72 * 16 * ct) * sizeof(float);
                if (pool) lds = std::max(lds, (size_t)16 * ct * ((size_t)px | 1) * sizeof(float));
                // plain outputs of >= 2 MB with 16-byte aligned planes: through an LDS tile, whole planes as float4
                static const int tile_env = getenv("TAPER_CONV_TILE_STORE") ? atoi(getenv("TAPER_CONV_TILE_STORE")) : 1;   // tuning probe
                const size_t tile_lds = (size_t)16 * ct * ((((size_t)px + 3) & ~(size_t)3) + 4) * sizeof(float);
                const int tile_store = tile_env && !pool && (h_out * w_out) % 4 == 0 && (long)n * c_out * h_out * w_out >= (1L << 19) &&
                                       std::max(lds, tile_lds) <= (150u << 10);
                if (tile_store) lds = std::max(lds, tile_lds);
                if (lds > (150u << 10)) continue;
                if ((long)img * c_in * h * w_in >= (1L << 28) || (long)9 * c_in * c_out >= (1L << 28)) continue;
                const int n_groups = ceil_div(n, img), n_co = ceil_div(c_out, 16 * ct);
                const long units = (long)n_groups * n_co, rounds = (units + kNumCU - 1) / kNumCU;
                double cost = (double)rounds * (nw / 4) * (32.0 * tpw * q + 9.0 * (tpw + q) + 4.8 * tpw);
                if (units < kNumCU) cost *= 1.0 + 0.5 * (double)(kNumCU - units) / kNumCU;   // idle CUs: prefer the split that fills the chip
                if (best < 0 || cost < best) {
                    best = cost;
                    *out = ConvImgPlan{img, ct, tpw, nw, n_groups, n_co, lds, tile_store};
                }
            }
        }
    }
    return best >= 0;
}

// images x rows per workgroup: the fullest tiling of <= 128 pixels by whole output rows of one or more images
static void conv_mfma_plan(int h_out, int w_out, int n, int *img_t, int *rows_t, int cit = MF_CI, bool even_rows = false) {
    int best_fill = -1;
    *img_t = 1;
    *rows_t = 1;
    for (int r = 1; r <= h_out; ++r) {
        if (r * w_out > MF_PX_MAX) break;
        if (even_rows && (r & 1)) continue;
        int im = MF_PX_MAX / (r * w_out);
        if (im > n) im = n;
        const int patch_cap = MF_PPT * 256 / (cit * (r + 2) * (w_out + 2));   // the staged patch is <= 12 floats per thread
        if (im > patch_cap) im = patch_cap;
        if (im < 1) continue;
        const int bands = ceil_div(h_out, r);
        // useful pixels per 128-pixel workgroup slot, counting the ragged last band; prefer fewer,
        // taller bands on ties (less halo re-read)
        const int fill = (int)((long)h_out * w_out * im * 1000 / ((long)bands * MF_PX_MAX));
        if (fill >= best_fill) {
            best_fill = fill;
            *img_t = im;
            *rows_t = r;
        }
    }
}

bool conv3x3_mfma_supported(int c_in, int h, int w, int pad) {
    const int w_out = w + 2 * pad - 2, h_out = h + 2 * pad - 2;
    return (c_in >= MF_CI || c_in == 1) && w_out >= 1 && h_out >= 1 && MF_CI * 3 * (w_out + 2) <= MF_PPT * 256;
}

// y (+)= conv3x3(x, w) [+ bias, relu]; weights [9*c_in][w_ld] with columns [0, w_cols) readable, rows 16-B aligned
bool conv3x3_mfma_pool_supported(int c_in, int h, int w, int pad) {   // whole 2x2 windows, at least one even band of <= 128 pixels
    const int w_out = w + 2 * pad - 2, h_out = h + 2 * pad - 2;
    return conv3x3_mfma_supported(c_in, h, w, pad) && h_out % 2 == 0 && w_out % 2 == 0 && 2 * w_out <= MF_PX_MAX &&
           MF_CI * 4 * (w_out + 2) <= MF_PPT * 256;
}

// -1: the image-resident kernel takes launches with >= one unit per two CUs; 0: never; 1: whenever a plan exists
// (TAPER_CONV_IMG at load, th_debug_set_conv_img at run time: the parity tests force both kernels onto the same shapes)
int g_conv_img_mode = getenv("TAPER_CONV_IMG") ? atoi(getenv("TAPER_CONV_IMG")) : -1;
// TAPER_CONV_LAYER_CHAIN=0: the four compiled layer geometries keep the image-resident kernel (measurement probe)
static const bool g_conv_layer_chain = !(getenv("TAPER_CONV_LAYER_CHAIN") && getenv("TAPER_CONV_LAYER_CHAIN")[0] == '0');
int conv_layer_chain_launch(th_ctx *ctx, const float *x, const float *w, const float *bias, float *y, float *cnt, int n, int c_in, int hw, int c_out,
                            int post, bool linear);   // conv_chain.hip

// launch configuration of this thread's most recent matrix-core convolution (th_debug_last_conv_config)
thread_local int t_last_conv_cfg[6] = {0, 0, 0, 0, 0, 0};

bool conv3x3_gap_supported(int n, int c_in, int h, int w_in, int c_out, int pad) {
    ConvImgPlan pl{};
    return c_in >= MF_CI && c_out % 4 == 0 && conv_img_plan(n, c_in, h, w_in, c_out, pad, true, &pl, true);
}

int conv3x3_mfma_launch(th_ctx *ctx, const float *x, const float *w, int w_ld, int w_cols, const float *bias, float *y, int n,
                        int c_in, int h, int w_in, int c_out, int pad, int relu, bool accum, bool pool, float *gap_cnt, bool gap) {
    TH_REQUIRE(w_ld % 4 == 0 && ((uintptr_t)w & 15) == 0, "conv3x3_mfma: weight rows must be 16-byte aligned");
    // the batch-256 layers of the reference CNN as one-stage chains (conv_chain.hip: conv_layer_chain_kernel; same bits): the taper slab read in
    // place, ReLU on, pad 1, default kernel choice only (th_debug_set_conv_img forces the kernels below)
    // (+ the plain sum -- no bias, no ReLU: an input gradient -- for the shapes conv_layer_chain_launch lists)
    const bool lin = !relu && !bias && !pool && !gap;
    if (g_conv_img_mode == -1 && g_conv_layer_chain && !accum && (relu || lin) && pad == 1 && h == w_in && w_ld == c_out && w_cols == c_out) {
        const int rc = conv_layer_chain_launch(ctx, x, w, bias, y, gap ? gap_cnt : nullptr, n, c_in, h, c_out, gap ? 2 : (pool ? 1 : 0), lin);
        if (rc < 0) { th::set_error("conv_layer_chain_kernel: launch failed"); return 1; }
        if (rc == 1) {
            t_last_conv_cfg[0] = c_out / 16; t_last_conv_cfg[1] = 8; t_last_conv_cfg[2] = 8;   // 8: one layer through the chain's compiled mapping
            t_last_conv_cfg[3] = n < kNumCU ? n : kNumCU; t_last_conv_cfg[4] = 1; t_last_conv_cfg[5] = pool ? 1 : 0;
            return 0;
        }
    }
    {
        // launches with at least a unit per two CUs take the image-resident kernel (TAPER_CONV_IMG = 0 / 1: never / whenever it fits)
        const int img_env = g_conv_img_mode;
        ConvImgPlan pl{};
        if ((img_env != 0 || gap) && !accum && c_in >= MF_CI && conv_img_plan(n, c_in, h, w_in, c_out, pad, pool, &pl, gap) &&
            (gap || img_env == 1 || (long)pl.n_groups * pl.n_co >= kNumCU / 2)) {
            ConvImgArgs g{};
            g.x = x; g.w = w; g.bias = bias; g.y = y;
            g.n = n; g.c_in = c_in; g.h = h; g.w_in = w_in; g.c_out = c_out; g.pad = pad;
            g.h_out = h + 2 * pad - 2; g.w_out = w_in + 2 * pad - 2; g.w_ld = w_ld; g.w_cols = w_cols;
            g.img_t = pl.img_t; g.n_groups = pl.n_groups; g.n_co = pl.n_co; g.relu = relu;
            g.gap = gap ? 1 : 0; g.gap_cnt = gap_cnt; g.tile_store = pl.tile_store;
            {   // staging plans of this geometry: built on device the first time, kept with the ctx
                const int pgs = pl.nw == 4 ? 4 : (pl.ct == 2 ? 4 : 8), n_goff = 512 * IM_PPT, n_pix = pgs * pl.tpw * 16;
                const std::array<int, 8> key{h, w_in, pad, c_in, pl.img_t, pgs, pl.tpw, 0};
                auto it = ctx->conv_plans.find(key);
                if (it == ctx->conv_plans.end()) {
                    void *tab = nullptr;
                    TH_HIP(hipMalloc(&tab, (size_t)(n_goff + n_pix) * sizeof(int)));
                    hipLaunchKernelGGL(conv_img_tables_kernel, dim3(ceil_div(n_goff + n_pix, 256)), dim3(256), 0, ctx->stream, (int *)tab,
                                       (int *)tab + n_goff, n_goff, n_pix, c_in, h, w_in, pad, pl.img_t);
                    TH_LAUNCH_CHECK();
                    it = ctx->conv_plans.emplace(key, tab).first;
                }
                g.goff_tab = (const int *)it->second;
                g.pix_tab = g.goff_tab + n_goff;
            }
            const dim3 grid(8 * ceil_div(pl.n_groups, 8) * pl.n_co);
#define TH_IMG2(CTV, TPWV, PL, NWV)                                                                                         \
            { (void)hipFuncSetAttribute((const void *)conv3x3_img_kernel<CTV, TPWV, PL, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
              hipLaunchKernelGGL((conv3x3_img_kernel<CTV, TPWV, PL, NWV>), grid, dim3(64 * NWV), pl.lds, ctx->stream, g); }
#define TH_IMG(CTV, TPWV)                                                                                                   \
            if (pool) { if (pl.nw == 4) TH_IMG2(CTV, TPWV, true, 4) else TH_IMG2(CTV, TPWV, true, 8) }                     \
            else { if (pl.nw == 4) TH_IMG2(CTV, TPWV, false, 4) else TH_IMG2(CTV, TPWV, false, 8) }
            // instances with the patch geometry compiled in (the batch-256 layers of the two CNNs); everything else takes the generic ones
#define TH_IMG_GEO(CTV, TPWV, WPV, CISV)                                                                                    \
            { if (pool) { (void)hipFuncSetAttribute((const void *)conv3x3_img_kernel<CTV, TPWV, true, 8, WPV, CISV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
                          hipLaunchKernelGGL((conv3x3_img_kernel<CTV, TPWV, true, 8, WPV, CISV>), grid, dim3(512), pl.lds, ctx->stream, g); }                                 \
              else { (void)hipFuncSetAttribute((const void *)conv3x3_img_kernel<CTV, TPWV, false, 8, WPV, CISV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);    \
                     hipLaunchKernelGGL((conv3x3_img_kernel<CTV, TPWV, false, 8, WPV, CISV>), grid, dim3(512), pl.lds, ctx->stream, g); } }
            const int wp_ = g.w_out + 2, cis_ = pl.img_t * (g.h_out + 2) * wp_;
            static const int geo_env = getenv("TAPER_CONV_IMG_GEO") ? atoi(getenv("TAPER_CONV_IMG_GEO")) : 1;   // tuning probe: 0 = generic instances only
            bool special = false;
            if (geo_env && pl.nw == 8) {
                special = true;
                if (pl.ct == 2 && pl.tpw == 13 && wp_ == 30 && cis_ == 900) TH_IMG_GEO(2, 13, 30, 900)          // 28x28, one image
                else if (pl.ct == 1 && pl.tpw == 7 && wp_ == 16 && cis_ == 1024) TH_IMG_GEO(1, 7, 16, 1024)      // 14x14, four images
                else if (pl.ct == 1 && pl.tpw == 4 && wp_ == 9 && cis_ == 648) TH_IMG_GEO(1, 4, 9, 648)          // 7x7, eight images
                else special = false;
            }
            if (special) {
            } else if (pl.ct == 2) {
                if (pl.tpw == 4) { TH_IMG(2, 4) } else if (pl.tpw == 7) { TH_IMG(2, 7) } else { TH_IMG(2, 13) }
            } else {
                if (pl.tpw == 4) { TH_IMG(1, 4) } else if (pl.tpw == 7) { TH_IMG(1, 7) } else { TH_IMG(1, 13) }
            }
#undef TH_IMG_GEO
#undef TH_IMG2
#undef TH_IMG
            t_last_conv_cfg[0] = pl.ct; t_last_conv_cfg[1] = 2 + (pl.nw == 4 ? 1 : 0) + (special ? 2 : 0); t_last_conv_cfg[2] = pl.tpw;   // 2 / 3: the image-resident kernel with 8 / 4 waves
            t_last_conv_cfg[3] = (int)grid.x; t_last_conv_cfg[4] = pl.img_t; t_last_conv_cfg[5] = pool ? 1 : 0;
            TH_LAUNCH_CHECK();
            return 0;
        }
    }
    TH_REQUIRE(!gap, "conv3x3 + global average pool: no image-resident plan for this shape (th_conv3x3_gap_supported)");
    ConvMfmaArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    a.n = n; a.c_in = c_in; a.h = h; a.w_in = w_in; a.c_out = c_out; a.pad = pad;
    a.h_out = h + 2 * pad - 2;
    a.w_out = w_in + 2 * pad - 2;
    a.w_ld = w_ld; a.w_cols = w_cols;
    const int cit = c_in == 1 ? 1 : MF_CI;
    conv_mfma_plan(a.h_out, a.w_out, n, &a.img_t, &a.rows_t, cit, pool);
    a.bands = ceil_div(a.h_out, a.rows_t);
    a.relu = relu;
    const int co_tiles = ceil_div(c_out, 16);
    // <= 64 output channels per workgroup: 32 accumulator + 18 prefetch VGPRs keep 3 workgroups per CU,
    // and wide layers get twice the workgroups (conv5 of the reference CNN: 232 instead of 116)
    int ct = co_tiles >= 4 ? 4 : (co_tiles >= 2 ? 2 : 1);
    // fewer than 1.5 workgroups per CU (conv5 of the reference CNN: 7x7 maps, 232 workgroups): nothing overlaps a
    // workgroup's staging / barriers -- halve the channel block instead (measured 45.2 -> 40.5 us; it costs 7-9 us
    // on the 14x14 layers, which have 430 workgroups)
    if (ct == 4 && (long)ceil_div(n, a.img_t) * a.bands * ceil_div(c_out, 64) < 384) ct = 2;
    static const int ct_env = getenv("TAPER_CONV_CT") ? atoi(getenv("TAPER_CONV_CT")) : 0;   // tuning probe
    if (ct_env && ct > ct_env) ct = ct_env;
    a.co_b = ct * 16;
    const size_t patch_n = (size_t)cit * a.img_t * (a.rows_t + 2) * (a.w_out + 2);
    size_t lds = (((patch_n + 4) & ~(size_t)3) + (size_t)((cit * 9 + 3) / 4 * 4) * a.co_b) * sizeof(float);
    // whole 8-channel blocks, plain (non-accumulating) output: operands go global -> LDS directly, two LDS stages
    static const int dma_env = getenv("TAPER_CONV_DMA") ? atoi(getenv("TAPER_CONV_DMA")) : 1;   // tuning probe: 0 = register-staged passes
    const bool dma = dma_env && cit == MF_CI && c_in % MF_CI == 0 && !accum && ct >= 2 &&
                     (long)a.img_t * c_in * h * w_in < (1L << 28) && (long)9 * c_in * w_ld < (1L << 28);
    if (dma) lds *= 2;
    if (pool) lds = std::max(lds, (size_t)a.co_b * ((size_t)(a.img_t * a.rows_t * a.w_out) | 1) * sizeof(float));   // the epilogue tile
    static const int lds_min_kb = getenv("TAPER_CONV_LDS_KB") ? atoi(getenv("TAPER_CONV_LDS_KB")) : 0;   // tuning probe: caps workgroups per CU
    if (lds_min_kb) lds = std::max(lds, (size_t)lds_min_kb * 1024);
    dim3 grid(ceil_div(n, a.img_t) * a.bands, ceil_div(c_out, a.co_b));
#define TH_MF(CTV, ACC, PL, WHV)                                                                                                   \
    if (cit == 1) hipLaunchKernelGGL((conv3x3_mfma_kernel<CTV, ACC, 1, PL, WHV>), grid, dim3(256 * WHV), lds, ctx->stream, a);     \
    else if (dma && !ACC && CTV >= 2) hipLaunchKernelGGL((conv3x3_mfma_kernel<(CTV >= 2 ? CTV : 2), false, MF_CI, PL, WHV, true>), grid, dim3(256 * WHV), lds, ctx->stream, a); \
    else hipLaunchKernelGGL((conv3x3_mfma_kernel<CTV, ACC, MF_CI, PL, WHV>), grid, dim3(256 * WHV), lds, ctx->stream, a);
#define TH_MF_CT(ACC, PL)                                                       \
    switch (ct) {                                                               \
        case 4: if (wh2) { TH_MF(4, ACC, PL, 2) } else { TH_MF(4, ACC, PL, 1) } break;   \
        case 2: if (wh2) { TH_MF(2, ACC, PL, 2) } else { TH_MF(2, ACC, PL, 1) } break;   \
        default: TH_MF(1, ACC, PL, 1) break;                                    \
    }
    // launches with < 3 workgroups per CU: 8 waves per workgroup (see WH; 14x14 layers 38.3 / 62.5 -> 34.4 / 56.5 us, 7x7: 40.0 -> 36.8 us;
    // the 28x28 layer has 1792 workgroups and gains nothing)
    static const int wh_env = getenv("TAPER_CONV_WH") ? atoi(getenv("TAPER_CONV_WH")) : 0;   // tuning probe: 1 forces 4 waves
    // (with LDS-DMA staging there is no per-pass staging work left for extra waves to hide: 4 waves measure 2-6 % faster on those layers)
    const bool wh2 = ct >= 2 && wh_env != 1 && (wh_env == 2 || (!dma && (long)grid.x * grid.y < 768));
    t_last_conv_cfg[0] = ct; t_last_conv_cfg[1] = (dma && !accum && ct >= 2 && cit != 1) ? 1 : 0; t_last_conv_cfg[2] = wh2 ? 2 : 1;
    t_last_conv_cfg[3] = (int)grid.x; t_last_conv_cfg[4] = (int)grid.y; t_last_conv_cfg[5] = pool ? 1 : 0;
    if (pool) TH_MF_CT(false, true) else if (accum) TH_MF_CT(true, false) else TH_MF_CT(false, false)
#undef TH_MF_CT
#undef TH_MF
    TH_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3 convolution on the matrix cores (full_backward extension; the reference
// cuts the tape before it, quirk Q2):
//   gw_eff[k][co] += sum_px patch[px, k] * gy[px, co],   k = (ci, kh, kw),  px = (n, oh, ow)
// The contraction runs over PIXELS.  A workgroup owns a slab of 16 input channels (K = 144 = 9 k-tiles)
// x co_b <= 64 output channels and walks every G-th pixel block (<= 128 pixels: whole rows of 1..9
// images, the forward kernel's tiling), keeping its 9*CT accumulator tiles in registers; it writes ONE
// partial [144][co_b] slab, and wgrad_reduce_kernel adds the G partials in order into the weight
// gradient through the layout map (deterministic, no atomics).
//   A operand (16 k x 4 px): lane l -> patch[pix_off[px0 + (l>>4)] + koff(k0 + (l&15))]   (LDS gather)
//   B operand (4 px x 16 co): lane l -> gyL[co0 + (l&15)][px0 + (l>>4)]
constexpr int WG_CI = 16;                 // input channels per slab
constexpr int WG_KT = WG_CI * 9 / 16;     // 9 k-tiles
constexpr int WG_GP = MF_PX_MAX + 2;      // pitch of the gy tile, 2 (mod 32): a half wave (16 channels x 2 pixels) reads 32 distinct banks
// A channel's LDS pitch: the smallest value >= its chunk that is 2 (mod 32).  The A operand of k-tile `tap` is one dword per lane at
// channel (lane & 15) * pitch + pixel (lane >> 4) + the tap's offset: with the 16 rows of a k-tile being 16 CHANNELS of one tap, a half
// wave's 32 lanes (16 channels x 2 consecutive pixels) land on banks 2 c + p -- all 32 distinct.  (r04, k = (channel, tap) in one index:
// a k-tile's rows mixed taps of two channels, rows 0 and 2 of a 16-wide padded window share their banks: 4-6 cycles per read instead of
// 2, and with 9 such reads per 9 MFMAs from each of 8 waves the LDS was as busy as the matrix pipes.)
__host__ __device__ constexpr int wg_cpitch(int chunk) { return chunk + ((2 - chunk % 32) + 32) % 32; }

struct ConvWgradArgs {
    const float *x, *gy;
    float *part;           // [G][9*c_in][co_ld]
    int n, c_in, h, w_in, c_out, pad, h_out, w_out;
    int img_t, rows_t, bands, n_pb, G, co_ld;
    unsigned x_bytes, gy_bytes;   // extents of x / gy: the ranges of the kernel's buffer descriptors
};

// PQ: patch elements per thread and channel (1 when a channel's chunk is <= 256 floats -- every batch-256 layer of the CNNs --, else 2).
// Two workgroups per CU (launch bounds: <= 256 registers): one's LDS stores and barriers run under the other's MFMAs.
template <int CT, int PQ>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_mfma_kernel(ConvWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CO_B = 16 * CT, NPAIR = WG_KT * CT, SLOTS = (NPAIR + 3) / 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l16 = lane & 15, g4 = lane >> 4;
    const int wp = a.w_out + 2, rp = a.rows_t + 2;
    const int img_stride = rp * wp, ci_stride = a.img_t * img_stride;       // <= 512
    const int cpitch = wg_cpitch(ci_stride);                                   // channel pitch in LDS: == 2 (mod 32), see the A operand below
    float *patch = lds;                                                        // [16][cpitch] >= [16][img_t][rp][wp]
    float *gyl = lds + WG_CI * cpitch;                                         // [CO_B][WG_GP]
    int *pixoff = reinterpret_cast<int *>(gyl + CO_B * WG_GP);                 // [128]
    const int cb = blockIdx.y * WG_CI, co0 = blockIdx.z * CO_B;
    const int px_per_img = a.rows_t * a.w_out, m_wg = a.img_t * px_per_img;
    const int shift = 1 - a.pad;
    const long chan = (long)a.h * a.w_in, ochan = (long)a.h_out * a.w_out;

    // ---- block-invariant plans ----
    // pixel p of a block -> LDS offset of its window corner (same for every block)
    if (t < MF_PX_MAX) {
        const int il = t < m_wg ? t / px_per_img : 0, rem = t < m_wg ? t % px_per_img : 0;
        pixoff[t] = il * img_stride + (rem / a.w_out) * wp + rem % a.w_out;
    }
    // patch element q (within one channel's [img_t][rp][wp] chunk), two per thread
    int pq_rel[PQ], pq_meta[PQ];          // meta = il | rr << 8 | col_ok << 16 | present << 17
#pragma unroll
    for (int u = 0; u < PQ; ++u) {
        const int q = t + 256 * u;
        pq_rel[u] = 0;
        pq_meta[u] = 0;
        if (q < ci_stride) {
            const int il = q / img_stride, r2 = q % img_stride, rr = r2 / wp, cc = r2 % wp;
            const int iw = cc - 1 + shift;
            pq_rel[u] = (int)((long)il * a.c_in * chan + (long)(rr - 1 + shift) * a.w_in + iw);
            pq_meta[u] = il | (rr << 8) | ((iw >= 0 && iw < a.w_in) ? 1 << 16 : 0) | (1 << 17);
        }
    }
    // gy element: pixel t % 128, output-channel parity t / 128
    const int gp = t & 127, gpar = t >> 7;
    const int g_il = gp < m_wg ? gp / px_per_img : 0, g_rem = gp < m_wg ? gp % px_per_img : 0;
    const int g_r = g_rem / a.w_out, g_c = g_rem % a.w_out;
    // this wave's (k-tile, co-tile) pairs: p = wave*SLOTS + j, co-tile-major
    int koffk[SLOTS];
    bool slot_ok[SLOTS], slot_hi[SLOTS];
    const int p_first = wave * SLOTS;
    const int ct_lo = min(p_first / WG_KT, CT - 1), ct_hi = min((p_first + SLOTS - 1) / WG_KT, CT - 1);
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        const int p = p_first + j;
        slot_ok[j] = p < NPAIR;
        const int ct = slot_ok[j] ? p / WG_KT : ct_lo, kt = slot_ok[j] ? p % WG_KT : 0;
        slot_hi[j] = ct != ct_lo;
        koffk[j] = l16 * cpitch + (kt / 3) * wp + kt % 3;     // k-tile kt = tap kt of the slab's 16 channels (row l16 = channel)
    }
    floatx4 acc[SLOTS];
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // The blocks are software-pipelined through registers: block pb + G's input patch and gradient tile are REQUESTED (one batch of buffer
    // loads: an element outside the image, the batch or the channel range gets an offset past the buffer and reads as 0 -- no branch per
    // load) before block pb's reduction steps and stored to LDS after them, so a block's global round trip runs under the previous block's
    // matrix work instead of in front of its own.
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
    const auto rg = __builtin_amdgcn_make_buffer_rsrc((void *)a.gy, 0, a.gy_bytes, 0x00020000);
    constexpr int OOB = 0x7ffffff0;
    float vp[16][PQ], vg[CO_B / 2];
    auto request = [&](int pb) {
        const int grp = pb / a.bands, band = pb % a.bands;
        const int img0 = grp * a.img_t, oh0 = band * a.rows_t;
        const int rows_here = min(a.rows_t, a.h_out - oh0);
        const int xb = (img0 * a.c_in + cb) * (int)chan + oh0 * a.w_in;
        int xo[PQ];
#pragma unroll
        for (int u = 0; u < PQ; ++u) {
            const int meta = pq_meta[u];
            const int il = meta & 255, rr = (meta >> 8) & 255, ih = oh0 + rr - 1 + shift;
            const bool ok = (meta >> 16 & 1) && ih >= 0 && ih < a.h && img0 + il < a.n;
            xo[u] = ok ? (xb + pq_rel[u]) * 4 : OOB;
        }
#pragma unroll
        for (int cl = 0; cl < 16; ++cl)
#pragma unroll
            for (int u = 0; u < PQ; ++u)
                vp[cl][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (cb + cl < a.c_in) ? xo[u] : OOB, cl * (int)chan * 4, 0));
        const bool pok = gp < m_wg && g_r < rows_here && img0 + g_il < a.n;
        const int go = pok ? (((img0 + g_il) * a.c_out + co0 + gpar) * (int)ochan + (oh0 + g_r) * a.w_out + g_c) * 4 : OOB;
#pragma unroll
        for (int i = 0; i < CO_B / 2; ++i)
            vg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, (co0 + 2 * i + gpar < a.c_out) ? go : OOB, 2 * i * (int)ochan * 4, 0));
    };
    const int n_steps = (m_wg + 3) / 4;
    if ((int)blockIdx.x < a.n_pb) request(blockIdx.x);
    for (int pb = blockIdx.x; pb < a.n_pb; pb += a.G) {
        __syncthreads();   // the previous block's MFMAs are done with patch / gyl (and pixoff is written)
#pragma unroll
        for (int cl = 0; cl < 16; ++cl)
#pragma unroll
            for (int u = 0; u < PQ; ++u)
                if (pq_meta[u] >> 17 & 1) patch[cl * cpitch + t + 256 * u] = vp[cl][u];
#pragma unroll
        for (int i = 0; i < CO_B / 2; ++i) gyl[(2 * i + gpar) * WG_GP + gp] = vg[i];
        __syncthreads();
        if (pb + a.G < a.n_pb) request(pb + a.G);
        // ---- reduction steps of 4 pixels (pixels past m_wg are zero columns: their steps would add nothing) ----
#pragma unroll 4
        for (int rs = 0; rs < n_steps; ++rs) {
            const int px = 4 * rs + g4;
            const int po = pixoff[px];
            const float b_lo = gyl[(ct_lo * 16 + l16) * WG_GP + px];
            const float b_hi = gyl[(ct_hi * 16 + l16) * WG_GP + px];
#pragma unroll
            for (int j = 0; j < SLOTS; ++j) {
                const float av = patch[po + koffk[j]];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, slot_hi[j] ? b_hi : b_lo, acc[j], 0, 0, 0);
            }
        }
    }
    // ---- one partial slab per workgroup: part[g][k][co]  (D tile: row = k = 4*(l>>4)+i, col = co = l&15) ----
    float *pp = a.part + (long)blockIdx.x * 9 * a.c_in * a.co_ld;
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        const int p = p_first + j;
        if (p >= NPAIR) continue;
        const int ct = p / WG_KT, kt = p % WG_KT;
        const int co = co0 + ct * 16 + l16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = cb + 4 * g4 + i;                // D row = channel 4 g4 + i of the slab, tile = tap kt
            if (ch < a.c_in && co < a.co_ld) pp[(long)(ch * 9 + kt) * a.co_ld + co] = acc[j][i];
        }
    }
}

// gw (+)= tmp[k][co] through the weight layout map (0: [k][c_out], 1: [co][9 c_in]); tmp is the ordered
// sum of the G partial slabs (th_colsum over the [G, kt*co_ld] matrix: deterministic, no atomics)
__global__ __launch_bounds__(256) void wgrad_scatter_kernel(const float *__restrict__ tmp, float *__restrict__ gw, int kt, int c_out,
                                                            int co_ld, int layout, int accumulate) {
    const long total = (long)kt * c_out;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i / c_out), co = (int)(i % c_out);
        const long idx = layout == 0 ? (long)k * c_out + co : (long)co * kt + k;
        const float v = tmp[(long)k * co_ld + co];
        gw[idx] = accumulate ? gw[idx] + v : v;
    }
}

// gw[c] (+)= sum over the G slabs of part[g][c], c < cols (a multiple of 4): a thread owns four consecutive columns and one of eight row
// classes (slabs r, r + 8, ...: eight 16-byte loads in flight, added in ascending order), the eight classes are added in order through LDS.
// Deterministic.  (th_colsum's one dword per lane and two loads in flight per wave: 15 us for the 38 MB of the 64 -> 64 layer's 256 slabs.)
__global__ __launch_bounds__(256) void slab_sum_kernel(const float *__restrict__ part, float *__restrict__ gw, int G, int cols, int accumulate) {
    __shared__ float4 sh[8][32];
    const int q = threadIdx.x & 31, rc = threadIdx.x >> 5;
    const long c4 = ((long)blockIdx.x * 32 + q) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < cols) {
        int g = rc;
        for (; g + 56 < G; g += 64) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(part + (long)(g + 8 * u) * cols + c4);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; g < G; g += 8) {
            const float4 v = *reinterpret_cast<const float4 *>(part + (long)g * cols + c4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sh[rc][q] = s;
    __syncthreads();
    if (rc == 0 && c4 < cols) {
#pragma unroll
        for (int u = 1; u < 8; ++u) { const float4 v = sh[u][q]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        float4 *o = reinterpret_cast<float4 *>(gw + c4);
        if (accumulate) { const float4 p = *o; s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w; }
        *o = s;
    }
}

// part: [G][kt][co_ld] partial slabs -> gw
int wgrad_reduce(th_ctx *ctx, const float *part, float *gw, int G, int kt, int c_out, int co_ld, int layout, int accumulate) {
    // taper layout and unpadded channel rows: the ordered sum of the slabs IS the gradient as it lies in memory -- no scatter launch
    if (layout == 0 && co_ld == c_out) {
        const int cols = kt * c_out;
        if (cols % 4 == 0 && G >= 16 && ((((uintptr_t)part | (uintptr_t)gw) & 15) == 0)) {
            hipLaunchKernelGGL(slab_sum_kernel, dim3(ceil_div(cols, 128)), dim3(256), 0, ctx->stream, part, gw, G, cols, accumulate);
            TH_LAUNCH_CHECK();
            return 0;
        }
        return accumulate ? th_colsum_accum(ctx, part, gw, G, cols) : th_colsum(ctx, part, gw, G, cols);
    }
    void *tmp = nullptr;
    const int cols = kt * co_ld;
    if (th_malloc(ctx, (size_t)cols * sizeof(float), &tmp)) return 1;
    if (int rc = th_colsum(ctx, part, (float *)tmp, G, cols)) return rc;
    hipLaunchKernelGGL(wgrad_scatter_kernel, dim3(ew_grid((size_t)kt * c_out, 256)), dim3(256), 0, ctx->stream, (const float *)tmp, gw, kt,
                       c_out, co_ld, layout, accumulate);
    TH_LAUNCH_CHECK();
    return th_free(ctx, tmp);
}

// ---- weight gradient, image-resident form (batch >= 128 on the 14 x 14 / 7 x 7 layers) ----------------------------------------------------
// The kernel above walks 128-pixel blocks of 16-channel slabs: every (block, slab) pair re-stages its gradient tile, a wave issues one
// gathered LDS read per MFMA, and a 14 x 14 image is two blocks.  Here a workgroup of eight waves owns IMAGES: the image's C_IN padded input
// planes and its C_OUT gradient planes sit in LDS whole (pitches 2 (mod 32): a half wave = 16 channels x 2 pixels reads 32 distinct banks),
// and a wave owns the 9 taps x CTW channel tiles of one 16-channel group -- per 4-pixel step 9 + CTW operand reads feed 9 CTW MFMAs
// (11 reads / 18 MFMAs at CTW = 2, 13 / 36 at CTW = 4; the slab kernel: 10 / 9).  Waves: (channel group) x (block of CTW channel tiles) x
// (PS pixel-step classes, added through LDS at the end).  A workgroup adds IPW images in its registers and writes ONE slab
// [9 C_IN][C_OUT] (taper layout); wgrad_reduce adds the slabs in order.  Same MFMA, same per-image pixel order within a step class.
__host__ __device__ constexpr int wi_pitch2(int v) { return v + ((2 - v % 32) + 32) % 32; }
// BANDS > 1: the image goes through the LDS in BANDS bands of S / BANDS output rows (+ their two halo rows of the input), one after the
// other, into the same accumulators -- through TWO LDS buffers: band b + 1 is requested (branch-free buffer loads into registers) before
// band b's steps and stored behind them, so only the first band's round trip is exposed (28 x 28 with 32 + 32 channels: four bands of 7
// rows, 2 x 66 KB).
struct WgradImgArgs {
    const float *x, *gy;   // [n][C_IN][S][S], [n][C_OUT][S][S]
    float *part;           // [gridDim.x][9 C_IN][C_OUT]
    int n, ipw;            // images per workgroup
    unsigned x_bytes, gy_bytes;
};
template <int S, int C_IN, int C_OUT, int CTW, int PS, int BANDS = 1>
__global__ __launch_bounds__(512, 1) void conv_wgrad_img_kernel(WgradImgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(S % BANDS == 0, "whole bands");
    constexpr int WP = S + 2, RB = S / BANDS, PX = S * S, PXB = RB * S, NSTEP = (PXB + 3) / 4, XROWS = RB + 2;
    constexpr int XPITCH = wi_pitch2(XROWS * WP), GPITCH = wi_pitch2(4 * NSTEP), BUF = C_IN * XPITCH + C_OUT * GPITCH, NBUF = BANDS > 1 ? 2 : 1;
    constexpr int NG = C_IN / 16, NCB = C_OUT / 16 / CTW;
    static_assert(NG * NCB * PS == 8 && C_IN % 16 == 0 && C_OUT % (16 * CTW) == 0, "eight waves: channel groups x channel-tile blocks x step classes");
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l16 = lane & 15, g4 = lane >> 4;
    const int g = wave % NG, cb = (wave / NG) % NCB, ps = wave / (NG * NCB);
    // the halo columns and the pixel columns past a band are zero for the whole launch: everything else is overwritten band by band
    for (int e = t; e < NBUF * BUF; e += 512) lds[e] = 0.f;
    floatx4 acc[9][CTW];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < CTW; ++j) acc[k][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int aoff = (g * 16 + l16) * XPITCH, boff = C_IN * XPITCH + (cb * CTW * 16 + l16) * GPITCH;
    // a band's elements, thread by thread: input rows band RB - 1 .. band RB + RB of every channel (rows outside the image read as zeros: an
    // offset past the descriptor) and the band's gradient rows
    constexpr int NX = C_IN * XROWS * S, NGY = C_OUT * PXB, UX = (NX + 511) / 512, UG = (NGY + 511) / 512, OOB = 0x7ffffff0;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
    const auto rg = __builtin_amdgcn_make_buffer_rsrc((void *)a.gy, 0, a.gy_bytes, 0x00020000);
    // (one buffer: the whole image at once would be ~50 registers of staging -- it goes through in chunks of seven loads instead)
    float vx[NBUF == 2 ? UX : 1], vg[NBUF == 2 ? UG : 1];
    auto stage_direct = [&](int ib) {                  // (NBUF == 1 means BANDS == 1: the band is the image, its halo rows stay zero)
        const int img = ib;
        constexpr int U = 7;
        const float *xi = a.x + (long)img * C_IN * PX;
        for (int e0 = t; e0 < C_IN * PX; e0 += 512 * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int e = e0 + 512 * u; v[u] = e < C_IN * PX ? xi[e] : 0.f; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + 512 * u;
                if (e < C_IN * PX) { const int c = e / PX, rem = e - c * PX, r = rem / S, q = rem - r * S; lds[c * XPITCH + (r + 1) * WP + q + 1] = v[u]; }
            }
        }
        const float *gi = a.gy + (long)img * C_OUT * PX;
        for (int e0 = t; e0 < C_OUT * PX; e0 += 512 * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int e = e0 + 512 * u; v[u] = e < C_OUT * PX ? gi[e] : 0.f; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + 512 * u;
                if (e < C_OUT * PX) { const int c = e / PX; lds[C_IN * XPITCH + c * GPITCH + e - c * PX] = v[u]; }
            }
        }
    };
    auto request = [&](int ib) {
        if (NBUF == 1) return;
        const int img = ib / BANDS, band = ib - img * BANDS;
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const int e = t + 512 * u, c = e / (XROWS * S), rem = e - c * (XROWS * S), rr = rem / S, q = rem - rr * S;
            const int ir = band * RB + rr - 1;
            const bool ok = e < NX && ir >= 0 && ir < S;
            vx[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? (((img * C_IN + c) * S + ir) * S + q) * 4 : OOB, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int e = t + 512 * u, c = e / PXB;
            vg[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, e < NGY ? ((img * C_OUT + c) * PX + band * PXB + e - c * PXB) * 4 : OOB, 0, 0));
        }
    };
    auto deposit = [&](float *buf) {
        if (NBUF == 1) return;
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const int e = t + 512 * u, c = e / (XROWS * S), rem = e - c * (XROWS * S), rr = rem / S, q = rem - rr * S;
            if (e < NX) buf[c * XPITCH + rr * WP + q + 1] = vx[u];
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int e = t + 512 * u, c = e / PXB;
            if (e < NGY) buf[C_IN * XPITCH + c * GPITCH + e - c * PXB] = vg[u];
        }
    };
    const int img0 = blockIdx.x * a.ipw, img1 = min(a.n, img0 + a.ipw);
    const int ib0 = img0 * BANDS, ib1 = img1 * BANDS;
    int cur = 0;
    if (ib0 < ib1) request(ib0);
    __syncthreads();                                     // the zeros are in place
    if (ib0 < ib1) { if (NBUF == 1) stage_direct(ib0); else deposit(lds); }
    __syncthreads();
    for (int ib = ib0; ib < ib1; ++ib) {
        const bool more = ib + 1 < ib1;
        if (more) request(ib + 1);
        const float *arow = lds + cur * BUF + aoff, *brow = lds + cur * BUF + boff;
#pragma unroll 2
        for (int step = ps; step < NSTEP; step += PS) {
            const int p = 4 * step + g4, pc = p < PXB ? p : PXB - 1;      // (a column past the band: its gradient is 0)
            const int r = pc / S, ao = r * WP + (pc - r * S);
            float av[9], bv[CTW];
#pragma unroll
            for (int k = 0; k < 9; ++k) av[k] = arow[ao + (k / 3) * WP + k % 3];
#pragma unroll
            for (int j = 0; j < CTW; ++j) bv[j] = brow[j * 16 * GPITCH + p];
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int j = 0; j < CTW; ++j) acc[k][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k], bv[j], acc[k][j], 0, 0, 0);
        }
        if (NBUF == 1) __syncthreads();                  // one buffer: every wave is done reading before the next band lands in it
        if (more) { if (NBUF == 1) stage_direct(ib + 1); else deposit(lds + (cur ^ 1) * BUF); }
        __syncthreads();
        if (NBUF == 2) cur ^= 1;
    }
    if (PS > 1) {
        // the step classes' sums through LDS (over the planes: every wave is past its last step), class 1 .. PS - 1 onto class 0 in order
        __syncthreads();
        float *red = lds + (wave % (NG * NCB)) * (9 * CTW * 256);
        for (int q = 1; q < PS; ++q) {
            if (ps == q) {
#pragma unroll
                for (int k = 0; k < 9; ++k)
#pragma unroll
                    for (int j = 0; j < CTW; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) red[((k * CTW + j) * 4 + e) * 64 + lane] = acc[k][j][e];
            }
            __syncthreads();
            if (ps == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k)
#pragma unroll
                    for (int j = 0; j < CTW; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[k][j][e] += red[((k * CTW + j) * 4 + e) * 64 + lane];
            }
            __syncthreads();
        }
    }
    if (ps != 0) return;
    // D tile (tap k, channel tile j): row = channel 4 g4 + e of the group, column = output channel l16 of the tile
    float *pp = a.part + (long)blockIdx.x * 9 * C_IN * C_OUT;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < CTW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                pp[(long)((g * 16 + 4 * g4 + e) * 9 + k) * C_OUT + (cb * CTW + j) * 16 + l16] = acc[k][j][e];
}

// 1: launched (slabs in *part_out, *n_slabs of them); 0: not one of the compiled shapes / batch too small
static int conv_wgrad_img_launch(th_ctx *ctx, const float *x, const float *gy, int n, int c_in, int hw, int c_out, float **part_out, int *n_slabs) {
    static const bool off = getenv("TAPER_WGRAD_IMG") && getenv("TAPER_WGRAD_IMG")[0] == '0';   // measurement / parity knob
    if (off || n < kNumCU / 2) return 0;
    const size_t xb = (size_t)n * c_in * hw * hw * 4, gb = (size_t)n * c_out * hw * hw * 4;
    if (xb >= (1u << 31) || gb >= (1u << 31)) return 0;
    WgradImgArgs a{x, gy, nullptr, n, 1, (unsigned)xb, (unsigned)gb};
#define TH_WI(S_, CI_, CO_, CTW_, PS_, IPW_) TH_WIB(S_, CI_, CO_, CTW_, PS_, IPW_, 1)
#define TH_WIB(S_, CI_, CO_, CTW_, PS_, IPW_, NB_)                                                                                        \
    do {                                                                                                                                 \
        a.ipw = IPW_;                                                                                                                    \
        const int grid = ceil_div(n, IPW_);                                                                                              \
        void *ws = nullptr;                                                                                                              \
        if (th_malloc(ctx, (size_t)grid * 9 * CI_ * CO_ * sizeof(float), &ws)) return -1;                                                \
        a.part = (float *)ws;                                                                                                            \
        constexpr int wp_ = S_ + 2, nst_ = ((S_ / NB_) * S_ + 3) / 4;                                                                    \
        constexpr int stage_ = (NB_ > 1 ? 2 : 1) * (CI_ * wi_pitch2((S_ / NB_ + 2) * wp_) + CO_ * wi_pitch2(4 * nst_)),                     \
                      red_ = (PS_ > 1 ? (8 / PS_) * 9 * CTW_ * 256 : 0);                                                                \
        constexpr int fl_ = stage_ > red_ ? stage_ : red_;                                                                              \
        static_assert(fl_ * 4 <= (160 << 10), "planes fit the LDS");                                                                    \
        auto kern = conv_wgrad_img_kernel<S_, CI_, CO_, CTW_, PS_, NB_>;                                                                 \
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, fl_ * 4);                              \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), fl_ * 4, ctx->stream, a);                                                        \
        if (hipGetLastError() != hipSuccess) return -1;                                                                                  \
        *part_out = a.part;                                                                                                              \
        *n_slabs = grid;                                                                                                                 \
        return 1;                                                                                                                        \
    } while (0)
    if (hw == 14 && c_in == 32 && c_out == 64) TH_WI(14, 32, 64, 2, 2, 1);     // (two pipelined bands: 24.9 vs 24.0 us)
    if (hw == 14 && c_in == 64 && c_out == 64) TH_WI(14, 64, 64, 2, 1, 1);     // (two pipelined bands: 42.5 vs 39.4 us)
    if (hw == 28 && c_in == 32 && c_out == 32) TH_WIB(28, 32, 32, 2, 4, 1, 4);
    // (7 x 7, 64 -> 128, two images per workgroup: built and measured equal to the slab kernel -- 38.1 vs 38.5 us -- with twice the slab
    // bytes for wgrad_reduce: 49 pixels per image are too few steps for a [576][128] slab per workgroup.  Left to the slab kernel.)
#undef TH_WI
#undef TH_WIB
    return 0;
}

// images x rows per pixel block of the weight-gradient kernel: the tiling of <= 128 pixels by whole output rows that takes the fewest
// 4-pixel reduction steps per image (the kernel's steps stop at the block's pixels) -- 14 x 14: two bands of 7 rows = 2 x 25 steps, where
// the forward's fullest-tile plan (9 + 5 rows) takes 2 x 32; taller bands on ties (less halo re-read)
static void conv_wgrad_plan(int h_out, int w_out, int n, int *img_t, int *rows_t) {
    long best = -1;
    *img_t = 1;
    *rows_t = 1;
    for (int r = 1; r <= h_out; ++r) {
        if (r * w_out > MF_PX_MAX) break;
        int im = std::min(n, MF_PX_MAX / (r * w_out));
        im = std::min(im, 512 / ((r + 2) * (w_out + 2)));      // a channel's staged chunk: <= 512 floats (two per thread)
        if (im < 1) continue;
        const long cost = (long)ceil_div(h_out, r) * ceil_div(im * r * w_out, 4) * 1000 / im;
        if (best < 0 || cost <= best) {
            best = cost;
            *img_t = im;
            *rows_t = r;
        }
    }
}

int conv3x3_wgrad_mfma_launch(th_ctx *ctx, const float *x, const float *gy, float *gw, int n, int c_in, int h, int w_in, int c_out,
                              int pad, int layout, int accumulate) {
    if (pad == 1 && h == w_in) {
        float *part = nullptr;
        int n_slabs = 0;
        const int rc = conv_wgrad_img_launch(ctx, x, gy, n, c_in, h, c_out, &part, &n_slabs);
        if (rc < 0) { th::set_error("conv_wgrad_img_kernel: launch failed"); return 1; }
        if (rc == 1) {
            if (int r2 = wgrad_reduce(ctx, part, gw, n_slabs, 9 * c_in, c_out, c_out, layout, accumulate)) return r2;
            return th_free(ctx, part);
        }
    }
    ConvWgradArgs a{};
    a.x = x; a.gy = gy;
    a.n = n; a.c_in = c_in; a.h = h; a.w_in = w_in; a.c_out = c_out; a.pad = pad;
    a.h_out = h + 2 * pad - 2;
    a.w_out = w_in + 2 * pad - 2;
    conv_wgrad_plan(a.h_out, a.w_out, n, &a.img_t, &a.rows_t);
    a.bands = ceil_div(a.h_out, a.rows_t);
    a.n_pb = ceil_div(n, a.img_t) * a.bands;
    const int co_tiles = ceil_div(c_out, 16);
    const int ct = co_tiles >= 4 ? 4 : (co_tiles >= 2 ? 2 : 1);
    const int co_b = ct * 16;
    const int slabs = ceil_div(c_in, WG_CI), co_blocks = ceil_div(c_out, co_b);
    a.co_ld = co_blocks * co_b;
    // workgroups per CU in total (LDS: <= 58 KB each): two -- TAPER_WGRAD_WGS = 1 .. 4 (measurement knob)
    static const int wgs = [] { const char *e = getenv("TAPER_WGRAD_WGS"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 4 ? v : 2; }();
    int G = ceil_div(wgs * kNumCU, slabs * co_blocks);
    if (G > a.n_pb) G = a.n_pb;
    if (G > 512) G = 512;
    a.G = G;
    const int kt = 9 * c_in;
    void *ws = nullptr;
    if (th_malloc(ctx, (size_t)G * kt * a.co_ld * sizeof(float), &ws)) return 1;
    a.part = (float *)ws;
    const size_t ci_stride = (size_t)a.img_t * (a.rows_t + 2) * (a.w_out + 2);
    TH_REQUIRE(ci_stride <= 512, "conv3x3_wgrad: patch chunk too large");
    const size_t xb = (size_t)n * c_in * h * w_in * 4, gb = (size_t)n * c_out * a.h_out * a.w_out * 4;
    TH_REQUIRE(xb < (1u << 31) && gb < (1u << 31), "conv3x3_wgrad: maps of 2 GiB and more are not supported");
    a.x_bytes = (unsigned)xb;
    a.gy_bytes = (unsigned)gb;
    const size_t lds = (WG_CI * (size_t)wg_cpitch((int)ci_stride) + (size_t)co_b * WG_GP) * sizeof(float) + MF_PX_MAX * sizeof(int);
    dim3 grid(G, slabs, co_blocks);
#define TH_WG(CT_, PQ_)                                                                                                                  \
    do {                                                                                                                                \
        auto kern = conv3x3_wgrad_mfma_kernel<CT_, PQ_>;                                                                                \
        if (lds > (64u << 10)) TH_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, a);                                                                 \
    } while (0)
    const bool one = ci_stride <= 256;
    switch (ct) {
        case 4: if (one) TH_WG(4, 1); else TH_WG(4, 2); break;
        case 2: if (one) TH_WG(2, 1); else TH_WG(2, 2); break;
        default: if (one) TH_WG(1, 1); else TH_WG(1, 2); break;
    }
#undef TH_WG
    TH_LAUNCH_CHECK();
    if (int rc = wgrad_reduce(ctx, a.part, gw, G, kt, c_out, a.co_ld, layout, accumulate)) return rc;
    return th_free(ctx, ws);
}

}  // namespace th

#ifdef TH_PROFILE
extern "C" int th_debug_conv_prof(th_ctx *ctx, long long *h_out8) {
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpyFromSymbol(h_out8, HIP_SYMBOL(th::g_conv_prof), 8 * sizeof(long long)));
    return 0;
}
extern "C" int th_debug_conv_timeline(th_ctx *ctx, long long *h_out, int n_wg) {   // tools/conv_timeline.py
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(th::g_conv_tl), (size_t)4 * n_wg * sizeof(long long)));
    TH_HIP(hipMemcpyFromSymbol(h_out + 4 * n_wg, HIP_SYMBOL(th::g_conv_clk), (size_t)2 * n_wg * sizeof(long long)));
    return 0;
}
#endif

extern "C" int th_debug_last_conv_config(th_ctx *ctx, int *out6) {
    TH_REQUIRE(ctx && out6, "th_debug_last_conv_config: null argument");
    for (int i = 0; i < 6; ++i) out6[i] = th::t_last_conv_cfg[i];
    return 0;
}

extern "C" int th_debug_set_conv_img(th_ctx *ctx, int mode) {
    TH_REQUIRE(ctx && mode >= -1 && mode <= 1, "th_debug_set_conv_img: mode must be -1 (auto), 0 (off) or 1 (whenever it fits)");
    th::g_conv_img_mode = mode;
    return 0;
}

// gemm.hip -- fp32 MFMA sgemm for gfx950: th_sgemm (src/gemm.rs semantics),
// th_linear_fwd / th_linear_bwd (src/nn.rs:54-60 and its three tape nodes).
//
// Two kernels:
//  * sgemm_tile<128>: 128x128x32 macro-tile, 4 waves x (2x2) v_mfma_f32_32x32x2_f32 (sgemm_tile<64>: 64x64x32, one MFMA tile per wave),
//    LDS double-buffered with register prefetch, XCD-aware tile order.  Bound:
//    MFMA fp32 peak (157.3 TF).  Used when the output has >= 64 macro-tiles.
//  * sgemm_small16: 16x16 output tile per workgroup, 4 waves split K between
//    them (v_mfma_f32_16x16x4_f32, operands straight from L2, no LDS staging),
//    optional grid-level split-K with a deterministic second-pass reduce.
//    For the latency-bound MNIST-MLP shapes (64x128x784, 128x784x64, ...).
// Operands are described by (row stride, col stride) so NN / NT / TN / TT all
// run the same code: "KC" = k-contiguous in memory, "MC" = m/n-contiguous.
#include "adam_dev.h"

namespace th {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

struct Epilogue {
    float alpha, beta;
    const float *bias;  // per output column (n), nullable
    int relu;
    AdamDev adam;       // adam.p != nullptr: C is a complete gradient and the Adam update of the
                        // parameter at the same index runs right here (th_linear_bwd_adam)
    // sgemm_tile, unsplit launches only (th_linear_bwd_adam_ex2): the backward of the fused Linear + ReLU IN FRONT of a layer, folded into that
    // layer's dX product -- C <- C * [mask > 0] (mask: the layer's input, [m][n] like C; ops.rs:358-369) and colpart [tiles_m][n] (nullable):
    // column sums of the stored tile rows, which th_colsum over the tile rows turns into the bias gradient of the layer in front (tensor.rs:686-691)
    const float *mask;
    float *colpart;
};

static inline Epilogue make_ep(float alpha, float beta, const float *bias = nullptr, int relu = 0) {
    return Epilogue{alpha, beta, bias, relu, AdamDev{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f}, nullptr, nullptr};
}

__device__ __forceinline__ float epilogue_apply(float acc, float c_old, const Epilogue &ep, int col) {
    float v = ep.alpha * acc;
    if (ep.beta != 0.0f) v += ep.beta * c_old;
    if (ep.bias) v += ep.bias[col];
    if (ep.relu) v = v > 0.0f ? v : 0.0f;
    return v;
}

// ------------------------------------------------------------------------
// small / latency-bound kernel
// ------------------------------------------------------------------------
// One workgroup (NW waves) per 16x16 output tile; wave w covers its share of
// [kbeg_slice, kend_slice).  Lane l: r = l & 15 (row of A / col of B),
// g = l >> 4 (which 4-wide k group).  In MFMA step s the lane supplies
// k = kk + 4*g + s for BOTH operands, so every k is used exactly once.
// MASKED: A values are multiplied by (Amask[same index] > 0) on the fly -- the
// ReLU backward (ops.rs:358-369) folded into the operand load of the two
// backward GEMMs of a Linear layer.
struct SmallArgs {
    const float *A, *Amask, *B;
    float *C, *partial;
    int m, n, k;
    long a_rs, a_cs, b_rs, b_cs;
    int kslice, a_vec, b_vec;
    Epilogue ep;
    int xcd_blocks = 0;   // sgemm_small16_tick only: 8 x 4 tiles handed out as one 2 x 2 block per XCD
};

template <bool A_KC, bool B_KC, int NW, bool MASKED>
__device__ __forceinline__ void small16_body(const SmallArgs &p, int tile_row, int tile_col, int zslice, float (*red)[64][4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int row0 = tile_row * 16, col0 = tile_col * 16;
    const int kbeg_slice = zslice * p.kslice;
    const int kend_slice = min(p.k, kbeg_slice + p.kslice);
    // split the slice between the NW waves in multiples of 16
    const int kwave = ((kend_slice - kbeg_slice + 16 * NW - 1) / (16 * NW)) * 16;
    const int kbeg = kbeg_slice + wave * kwave;
    const int kend = min(kend_slice, kbeg + kwave);

    const int arow = row0 + r, bcol = col0 + r;
    const bool a_ok = arow < p.m, b_ok = bcol < p.n;
    const long a_off = (long)(a_ok ? arow : 0) * p.a_rs;
    const float *ap = p.A + a_off;
    const float *mp = MASKED ? p.Amask + a_off : nullptr;
    const float *bp = p.B + (long)(b_ok ? bcol : 0) * p.b_cs;

    // These shapes are latency-bound: after a kernel boundary every dependent
    // global round trip costs ~0.5-1 us.  So (a) everything the epilogue will
    // need from memory -- bias, the old C for beta != 0, the Adam state of a fused
    // update -- is requested by wave 0 NOW, under the operand loads; (b) the
    // operand loads of CH k-steps (64 k per wave) are all issued before the
    // first MFMA consumes any.
    // Epilogue work is spread over waves 0..3: wave e finishes output element e of every lane (row
    // 4*(lane>>4) + e of the tile), so the p / m / v traffic of a fused Adam update and the final adds
    // run on four SIMDs instead of one.
    const int ecol = col0 + (lane & 15);
    const bool fuse_adam = p.ep.adam.p != nullptr && p.partial == nullptr;
    const int erow = row0 + (lane >> 4) * 4 + wave;     // meaningful for wave < 4
    const bool e_ok = wave < 4 && erow < p.m && ecol < p.n;
    const long e_ix = e_ok ? (long)erow * p.n + ecol : 0;
    float e_bias = 0.f, e_cold = 0.f, e_p = 0.f, e_m = 0.f, e_v = 0.f, e_step = 0.f;
    if (e_ok) {
        if (p.ep.beta != 0.0f && !p.partial) e_cold = p.C[e_ix];
        if (fuse_adam) {
            e_p = p.ep.adam.p[e_ix];
            e_m = p.ep.adam.m[e_ix];
            e_v = p.ep.adam.v[e_ix];
            e_step = adam_dev_step(p.ep.adam);
        }
        if (p.ep.bias) e_bias = p.ep.bias[ecol];
    }
    constexpr int CH = 4;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kk0 = kbeg; kk0 < kend; kk0 += 16 * CH) {
        float av[CH][4], bv[CH][4];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int kb = kk0 + 16 * c + g * 4;
            if (A_KC && p.a_vec && kb + 4 <= kend) {
                float4 t = *reinterpret_cast<const float4 *>(ap + kb);
                av[c][0] = t.x; av[c][1] = t.y; av[c][2] = t.z; av[c][3] = t.w;
                if (MASKED) {
                    float4 q = *reinterpret_cast<const float4 *>(mp + kb);
                    av[c][0] = q.x > 0.f ? av[c][0] : 0.f; av[c][1] = q.y > 0.f ? av[c][1] : 0.f;
                    av[c][2] = q.z > 0.f ? av[c][2] : 0.f; av[c][3] = q.w > 0.f ? av[c][3] : 0.f;
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bool in = kb + s < kend;
                    float v = in ? ap[(long)(kb + s) * p.a_cs] : 0.f;
                    if (MASKED) v = (in && mp[(long)(kb + s) * p.a_cs] > 0.f) ? v : 0.f;
                    av[c][s] = v;
                }
            }
            if (B_KC && p.b_vec && kb + 4 <= kend) {
                float4 t = *reinterpret_cast<const float4 *>(bp + kb);
                bv[c][0] = t.x; bv[c][1] = t.y; bv[c][2] = t.z; bv[c][3] = t.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) bv[c][s] = (kb + s < kend) ? bp[(long)(kb + s) * p.b_rs] : 0.f;
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (kk0 + 16 * c >= kend) break;  // wave-uniform
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float a = a_ok ? av[c][s] : 0.f;
                const float b = b_ok ? bv[c][s] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
        }
    }

    // deterministic in-workgroup reduction: every wave parks its 4 partial sums; wave e adds element e
    // of waves 0..NW-1 in wave order
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][lane][i] = acc[i];
    __syncthreads();
    if (wave < 4) {
        float sum = red[0][lane][wave];
        for (int w = 1; w < NW; ++w) sum += red[w][lane][wave];
        // C/D map of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + i  (here i = wave)
        if (e_ok) {
            if (p.partial) {
                p.partial[(long)zslice * p.m * p.n + e_ix] = sum;
            } else {
                float out = p.ep.alpha * sum;
                if (p.ep.beta != 0.0f) out += p.ep.beta * e_cold;
                if (p.ep.bias) out += e_bias;
                if (p.ep.relu) out = out > 0.0f ? out : 0.0f;
                p.C[e_ix] = out;
                if (fuse_adam) {  // th_linear_bwd_adam: C is a complete gradient (optim.rs:99-110)
                    const AdamDev &ad = p.ep.adam;
                    const float gv = out + ad.wd * e_p;
                    const float mn = ad.beta1 * e_m + (1.0f - ad.beta1) * gv;
                    const float vn = ad.beta2 * e_v + (1.0f - ad.beta2) * gv * gv;
                    ad.m[e_ix] = mn;
                    ad.v[e_ix] = vn;
                    ad.p[e_ix] = e_p - e_step * mn / (sqrtf(vn) + ad.eps);
                }
            }
        }
    }
}

// grid = (tiles_n, tiles_m, kz); block = 64 * NW
template <bool A_KC, bool B_KC, int NW>
__global__ __launch_bounds__(64 * NW) void sgemm_small16(SmallArgs p) {
    __shared__ float red[NW][64][4];
    small16_body<A_KC, B_KC, NW, false>(p, blockIdx.y, blockIdx.x, blockIdx.z, red);
}

// th_linear_fwd_ex: the same tiles on rows [0, gridDim.y - 1) of the grid; block (0, gridDim.y - 1) applies the
// carried Adam slices with the step counter as it stands and then opens the next step (t += 1)
template <bool A_KC, bool B_KC, int NW>
__global__ __launch_bounds__(64 * NW) void sgemm_small16_tick(SmallArgs p, AdamSlices x, int32_t *tick) {
    __shared__ float red[NW][64][4];
    if (blockIdx.y + 1 < gridDim.y) {
        int tm = blockIdx.y, tn = blockIdx.x;
        if (p.xcd_blocks) {
            // 8 x 4 tiles (the MNIST MLP's layer 1 at batch 64): block b runs on XCD b % 8; give every XCD a 2 x 2 block of tiles instead of
            // one column of 4, so its L2 fetches 2 row tiles of X and 2 of W (200 KB) instead of all of X and 1 of W (250 KB)
            const int b = blockIdx.x + gridDim.x * blockIdx.y, q = b & 7, j = b >> 3;
            tm = 2 * (q >> 2) + (j >> 1);
            tn = 2 * (q & 3) + (j & 1);
        }
        small16_body<A_KC, B_KC, NW, false>(p, tm, tn, 0, red);
        return;
    }
    if (blockIdx.x == 0) adam_slices_then_tick(x, tick);
}

// second pass of grid-level split-K: fixed slice order -> deterministic.  With ep.adam set, C is a
// complete gradient and the parameter's Adam update (optim.rs:99-110) runs on the same element.
// (the slabs are summed in slice order whatever the instance: same bits)
__global__ __launch_bounds__(256) void splitk_reduce4(const float *__restrict__ partial, float *__restrict__ C, long mn, int n, int kz, Epilogue ep) {
    // four neighbouring elements per thread, 16-byte loads, eight slabs requested together: the scalar form below spent 11 us on the 28 MB of
    // a 784 x 256 gradient's 36 slabs (2.6 TB/s)
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= mn) return;
    const bool fuse = ep.adam.p != nullptr;
    float4 pv = {0.f, 0.f, 0.f, 0.f}, mv = pv, vv = pv, c_old = pv;
    float step = 0.f;
    if (fuse) {
        pv = *reinterpret_cast<const float4 *>(ep.adam.p + i);
        mv = *reinterpret_cast<const float4 *>(ep.adam.m + i);
        vv = *reinterpret_cast<const float4 *>(ep.adam.v + i);
        step = adam_dev_step(ep.adam);
    }
    if (ep.beta != 0.0f) c_old = *reinterpret_cast<const float4 *>(C + i);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 8;
    int z = 0;
    for (; z + U <= kz; z += U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4 *>(partial + (long)(z + u) * mn + i);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w;
        }
    }
    for (; z < kz; ++z) {
        const float4 v = *reinterpret_cast<const float4 *>(partial + (long)z * mn + i);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    }
    const float co[4] = {c_old.x, c_old.y, c_old.z, c_old.w};
    float out[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = epilogue_apply(s[e], co[e], ep, (int)((i + e) % n));
    *reinterpret_cast<float4 *>(C + i) = make_float4(out[0], out[1], out[2], out[3]);
    if (fuse) {
        const AdamDev &ad = ep.adam;
        const float p4[4] = {pv.x, pv.y, pv.z, pv.w}, m4[4] = {mv.x, mv.y, mv.z, mv.w}, v4[4] = {vv.x, vv.y, vv.z, vv.w};
        float pn[4], mo[4], vo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gv = out[e] + ad.wd * p4[e];
            mo[e] = ad.beta1 * m4[e] + (1.0f - ad.beta1) * gv;
            vo[e] = ad.beta2 * v4[e] + (1.0f - ad.beta2) * gv * gv;
            pn[e] = p4[e] - step * mo[e] / (sqrtf(vo[e]) + ad.eps);
        }
        *reinterpret_cast<float4 *>(ad.m + i) = make_float4(mo[0], mo[1], mo[2], mo[3]);
        *reinterpret_cast<float4 *>(ad.v + i) = make_float4(vo[0], vo[1], vo[2], vo[3]);
        *reinterpret_cast<float4 *>(ad.p + i) = make_float4(pn[0], pn[1], pn[2], pn[3]);
    }
}

__global__ __launch_bounds__(256) void splitk_reduce(const float *__restrict__ partial, float *__restrict__ C,
                                                     long mn, int n, int kz, Epilogue ep) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= mn) return;
    const bool fuse = ep.adam.p != nullptr;
    float pv = 0.f, mv = 0.f, vv = 0.f, step = 0.f;
    if (fuse) {
        pv = ep.adam.p[i];
        mv = ep.adam.m[i];
        vv = ep.adam.v[i];
        step = adam_dev_step(ep.adam);
    }
    const float c_old = ep.beta != 0.0f ? C[i] : 0.0f;
    float s = 0.f;
    for (int z = 0; z < kz; ++z) s += partial[(long)z * mn + i];
    const float out = epilogue_apply(s, c_old, ep, (int)(i % n));
    C[i] = out;
    if (fuse) {
        const AdamDev &ad = ep.adam;
        const float gv = out + ad.wd * pv;
        const float mn_ = ad.beta1 * mv + (1.0f - ad.beta1) * gv;
        const float vn = ad.beta2 * vv + (1.0f - ad.beta2) * gv * gv;
        ad.m[i] = mn_;
        ad.v[i] = vn;
        ad.p[i] = pv - step * mn_ / (sqrtf(vn) + ad.eps);
    }
}

static inline bool aligned16p(const void *p) { return ((uintptr_t)p & 15) == 0; }
// the reduce pass: four elements per thread where every pointer and the slab stride allow 16-byte accesses
static inline void launch_splitk_reduce(hipStream_t stream, const float *partial, float *C, long mn, int n, int kz, const Epilogue &ep) {
    const bool quad = mn % 4 == 0 && aligned16p(partial) && aligned16p(C) && !ep.mask && !ep.colpart &&
                      (!ep.adam.p || (aligned16p(ep.adam.p) && aligned16p(ep.adam.m) && aligned16p(ep.adam.v)));
    if (quad) hipLaunchKernelGGL(splitk_reduce4, dim3(ceil_div(mn / 4, 256)), dim3(256), 0, stream, partial, C, mn, n, kz, ep);
    else hipLaunchKernelGGL(splitk_reduce, dim3(ceil_div(mn, 256)), dim3(256), 0, stream, partial, C, mn, n, kz, ep);
}

// Whole backward of one (small) Linear layer in ONE launch (the reference runs
// three tape nodes: ops.rs:238-294, tensor.rs:574-587, 674-694, plus the ReLU
// node ops.rs:358-369 when MASKED).  Workgroup roles by block index:
//   [0, n_dw)            dW[out,in] (+)= dZ^T . X      (TN: A = dZ^T is MC, B = X is MC); n_dw = the tile
//                        count rounded up to 8: block b runs on XCD b % 8, and XCD x takes the x-th
//                        eighth of the tiles in column-major order, so each XCD's L2 fetches only its
//                        own X columns / Adam state lines instead of every XCD fetching all of X
//   [n_dw, n_dw + n_dx)  dX[B,in]   (+)= dZ . W        (NN: A = dZ is KC,  B = W is MC)
//   next n_db            db[out]    (+)= sum_b dZ[b,:] (64 columns per workgroup)
//   the rest             deferred Adam updates of OTHER parameters (th_adam_slice), 1024 elements each
// with dZ = dY (* (Y > 0) when MASKED).
struct LinearBwdArgs {
    SmallArgs dw, dx;
    int n_dw, n_dx, n_db, dw_tiles, dw_tiles_m, dx_tiles_n, dw_kz;   // dw_tiles counts (tile, K slice) pairs
    const float *dy, *ymask;
    float *db;
    int batch, out_f, db_accum;
    AdamDev db_adam;
    AdamSlices extra;
};

// bias gradient role of a Linear layer's backward: db[c] (+)= sum_b dZ[b][c] for 64 columns, Adam in the epilogue when asked for
template <bool MASKED>
__device__ __forceinline__ void linear_db_role(const LinearBwdArgs &q, int blk, float (*red)[64][4]) {
    // bias gradient: 4 waves stride the batch rows, lanes are consecutive columns
    float(*part)[64] = reinterpret_cast<float(*)[64]>(&red[0][0][0]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blk * 64 + lane;
    // epilogue operands are requested before the column sum, not after it (one round trip, not two)
    const bool own = wave == 0 && c < q.out_f, fuse = own && q.db_adam.p;
    float pv = 0.f, mv = 0.f, vv = 0.f, step = 0.f, old = 0.f;
    if (fuse) {
        pv = q.db_adam.p[c];
        mv = q.db_adam.m[c];
        vv = q.db_adam.v[c];
        step = adam_dev_step(q.db_adam);
    }
    if (own && q.db_accum) old = q.db[c];
    float s = 0.f;
    if (c < q.out_f) {
        // 16 independent loads in flight per lane: at batch 1024 this role (256 rows per wave) was the
        // kernel's long pole with 4
        int r = wave;
        for (; r + 60 < q.batch; r += 64) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const long idx = (long)(r + 4 * u) * q.out_f + c;
                v[u] = q.dy[idx];
                if (MASKED) v[u] = q.ymask[idx] > 0.f ? v[u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; r < q.batch; r += 4) {
            const long idx = (long)r * q.out_f + c;
            float v = q.dy[idx];
            if (MASKED) v = q.ymask[idx] > 0.f ? v : 0.f;
            s += v;
        }
    }
    part[wave][lane] = s;
    __syncthreads();
    if (own) {
        const float tot = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
        const float out = q.db_accum ? old + tot : tot;
        q.db[c] = out;
        if (fuse) {   // optim.rs:99-110
            const float gv = out + q.db_adam.wd * pv;
            const float mn = q.db_adam.beta1 * mv + (1.0f - q.db_adam.beta1) * gv;
            const float vn = q.db_adam.beta2 * vv + (1.0f - q.db_adam.beta2) * gv * gv;
            q.db_adam.m[c] = mn;
            q.db_adam.v[c] = vn;
            q.db_adam.p[c] = pv - step * mn / (sqrtf(vn) + q.db_adam.eps);
        }
    }
}

template <bool MASKED>
__global__ __launch_bounds__(256) void linear_bwd_small(LinearBwdArgs q) {
    __shared__ float red[4][64][4];
    const int bid = blockIdx.x;
    if (bid < q.n_dw) {
        const int t = (bid & 7) * (q.n_dw >> 3) + (bid >> 3);
        if (t < q.dw_tiles) {   // K slice innermost: the slices of one tile share its XCD (and its operand columns)
            const int z = t % q.dw_kz, tile = t / q.dw_kz;
            small16_body<false, false, 4, MASKED>(q.dw, tile % q.dw_tiles_m, tile / q.dw_tiles_m, z, red);
        }
    } else if (bid < q.n_dw + q.n_dx) {
        const int t = bid - q.n_dw;
        small16_body<true, false, 4, MASKED>(q.dx, t / q.dx_tiles_n, t % q.dx_tiles_n, 0, red);
    } else if (bid >= q.n_dw + q.n_dx + q.n_db) {
        adam_slices_block(q.extra, bid - q.n_dw - q.n_dx - q.n_db);
    } else {
        linear_db_role<MASKED>(q, bid - q.n_dw - q.n_dx, red);
    }
}

// ---- th_mlp3_xent, launch 2 (mlp3.hip): the parameter gradients of a three-layer classifier in ONE launch -- three dW + db jobs of
// linear_bwd_small (the rows launch wrote the ReLU-masked dZ of every layer: no mask, no dX here) and a lead workgroup that adds the per-block
// loss / hit partial sums in block order and writes the loss, the count and the step log (loss.rs:164, 283; train.rs:117).  Every dW / db
// epilogue may carry its Adam update: nobody reads a parameter in this launch.
struct Mlp3GradArgs {
    LinearBwdArgs job[3];
    int first[5];              // block ranges of the jobs, then of the gap role; block first[4] is the lead
    // gap role (optional): the input rows are the plane means of a bias-only Conv2dReLU + global average pool; its bias gradient
    // db[ch] = sum_n (dX[n][ch] / hw) * cnt[n][ch] (th_bias_grad_counts_adam's formula) and Adam, 16 channels per workgroup
    const float *gap_dx, *gap_cnt;
    float *gap_db;
    int gap_c, gap_hw;
    AdamDev gap_adam;
    const float *part;         // [n_blk][2]: sum of the rows' NLL, hits
    int n_blk, batch;
    float *loss, *ncorrect, *metrics;
    int64_t capacity;
    int64_t *state;
    int64_t advance;
};

__global__ __launch_bounds__(256) void mlp3_grads_kernel(Mlp3GradArgs a) {
    __shared__ float red[4][64][4];
    const int bid = blockIdx.x;
    if (bid >= a.first[4]) {
        float n = 0.f, h = 0.f;
        if (a.n_blk > 32) {
            // one entry per ROW (th_conv_chain_mlp3_xent): thread t adds entries t, t + 256, ... in order, then a fixed tree over the 256 threads
            // (one thread walking hundreds of entries is a chain of dependent round trips)
            for (int b = threadIdx.x; b < a.n_blk; b += 256) {
                n += a.part[2 * b];
                h += a.part[2 * b + 1];
            }
            float *sh = &red[0][0][0];
            sh[threadIdx.x] = n;
            sh[256 + threadIdx.x] = h;
            __syncthreads();
            if (threadIdx.x < 64) {
                const int t = threadIdx.x;
                n = (sh[t] + sh[t + 64]) + (sh[t + 128] + sh[t + 192]);
                h = (sh[256 + t] + sh[320 + t]) + (sh[384 + t] + sh[448 + t]);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    n += __shfl_xor(n, off, 64);
                    h += __shfl_xor(h, off, 64);
                }
            }
        } else if (threadIdx.x == 0) {
            for (int b = 0; b < a.n_blk; ++b) {
                n += a.part[2 * b];
                h += a.part[2 * b + 1];
            }
        }
        if (threadIdx.x == 0) {
            const float l = n / (float)a.batch;   // loss.rs:164
            a.loss[0] = l;
            if (a.ncorrect) a.ncorrect[0] = h;
            if (a.metrics) {                      // the step log of th_log_step
                const int64_t s0 = a.state[0], s1 = a.state[1];
                const int64_t slot = s0 < a.capacity ? s0 : s0 % a.capacity;
                a.metrics[2 * slot] = l;
                a.metrics[2 * slot + 1] = h;
                a.state[0] = s0 + 1;
                a.state[1] = s1 + a.advance;
            }
        }
        return;
    }
    if (bid >= a.first[3]) {
        float(*sh)[17] = reinterpret_cast<float(*)[17]>(&red[0][0][0]);
        const int q = threadIdx.x & 15, r = threadIdx.x >> 4, ch = (bid - a.first[3]) * 16 + q;
        float s = 0.f;
        if (ch < a.gap_c) {
#pragma unroll 4
            for (int b = r; b < a.batch; b += 16) s += a.gap_dx[(long)b * a.gap_c + ch] / (float)a.gap_hw * a.gap_cnt[(long)b * a.gap_c + ch];
        }
        sh[r][q] = s;
        __syncthreads();
        if (threadIdx.x < 16 && ch < a.gap_c) {
            float tot = sh[0][q];
#pragma unroll
            for (int i = 1; i < 16; ++i) tot += sh[i][q];
            a.gap_db[ch] = tot;
            if (a.gap_adam.p) adam_update(a.gap_adam.p, a.gap_adam.m, a.gap_adam.v, ch, tot, adam_dev_step(a.gap_adam), a.gap_adam.beta1,
                                          a.gap_adam.beta2, a.gap_adam.eps, a.gap_adam.wd);
        }
        return;
    }
    const int j = bid >= a.first[2] ? 2 : (bid >= a.first[1] ? 1 : 0);
    const LinearBwdArgs &q = a.job[j];
    const int b = bid - a.first[j];
    if (b < q.n_dw) {
        const int t = (b & 7) * (q.n_dw >> 3) + (b >> 3);
        if (t < q.dw_tiles) small16_body<false, false, 4, false>(q.dw, t % q.dw_tiles_m, t / q.dw_tiles_m, 0, red);
    } else {
        linear_db_role<false>(q, b - q.n_dw, red);
    }
}

// ------------------------------------------------------------------------
// 128x128x32 MFMA kernel
// ------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 32;
// Tile geometry for a TS x TS x BK macro-tile (TS = 128: 4 waves x (2x2) 32x32 MFMA tiles; TS = 64: 4 waves x one 32x32
// tile -- for outputs too small to give the chip enough 128-tiles: tall-skinny MLP products at batch 512..8192).
template <int TS>
struct TileGeo {
    // LDS image of an operand tile: [mn][32 k], 128-byte rows, the eight 16-byte k quads of row r stored at quad ^ ((r >> 1) & 7).
    //  * a lane reads the operands of FOUR k-steps with one ds_read_b128 (the 16 rows of a b128 lane group -- {0-3, 12-15, 20-27} and its
    //    siblings -- map to 16 distinct 16-byte slots of the 256-byte bank row: conflict-free; MI355X_MICROARCH.md, LDS);
    //  * a 128-byte pitch is what LDS-DMA (`buffer_load ... lds`: lane l of a wave lands at base + 16 l) can fill: a k-contiguous operand
    //    goes global -> LDS without staging registers or ds_write -- the lane just FETCHES quad (l & 7) ^ swizzle of its row.
    // An m/n-contiguous operand is staged AS IT LIES in memory -- image [32 k][TS mn], 4 TS-byte rows -- and read with ds_read_b32 (lanes on
    // consecutive m / n: conflict-free; the k of a lane's four operands differ by whole rows: immediates): no transpose anywhere, and
    // LDS-DMA fills this image too (a k row is a whole number of 16-byte lanes).
    static constexpr int LD = BK;          // floats per row of the [mn][k] image
    static constexpr int TILE_MAX = TS * LD;
    static constexpr int R = TS / 32;      // float4 per thread per operand per tile
    static constexpr int QPR = TS / 4;     // float4 quads per k row of an m/n-contiguous tile
    __host__ __device__ static constexpr int swz(int row) { return (row >> 1) & 7; }
};

// Each thread stages 4 R floats per operand per tile, as R float4.
// KC source: the tile is [TS rows][32 k]; float4 along k: 8 per row ->
//   thread t handles rows (t / 8) + 32*j, k quad (t % 8).
// MC source: the tile is [32 k][TS mn]; float4 along mn: QPR per k row ->
//   thread t handles k rows (t / QPR) + (256 / QPR)*j, mn quad (t % QPR).
template <bool KC, bool GUARD, int TS>
__device__ __forceinline__ void load_tile(const float *__restrict__ P, long rs_mn, long rs_k, int mn0, int k0,
                                          int mn_lim, int k_lim, int t, float4 (&reg)[TileGeo<TS>::R], bool vec = false) {
    constexpr int R = TileGeo<TS>::R, QPR = TileGeo<TS>::QPR;
    // element (mn, k) lives at P[mn * rs_mn + k * rs_k]; exactly one stride is 1
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if (KC) {
            const int mn = mn0 + (t >> 3) + 32 * j, kq = k0 + (t & 7) * 4;
            const float *p = P + (long)mn * rs_mn + kq;
            if (!GUARD) {
                reg[j] = *reinterpret_cast<const float4 *>(p);
            } else if (vec) {
                // ragged problem, rows 16-B aligned and k_lim % 4 == 0: a quad is entirely inside or outside.
                // Branch-free: load from the clamped (always valid) address, then zero what lies outside.
                const bool in = mn < mn_lim && kq < k_lim;
                const float4 v = *reinterpret_cast<const float4 *>(P + (long)min(mn, mn_lim - 1) * rs_mn + min(kq, k_lim - 4));
                reg[j] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (mn < mn_lim && kq + e < k_lim) ? p[e] : 0.f;
                reg[j] = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else {
            const int kr = k0 + t / QPR + (256 / QPR) * j, mq = mn0 + (t % QPR) * 4;
            const float *p = P + (long)kr * rs_k + mq;
            if (!GUARD) {
                reg[j] = *reinterpret_cast<const float4 *>(p);
            } else if (vec) {   // mn_lim % 4 == 0
                const bool in = kr < k_lim && mq < mn_lim;
                const float4 v = *reinterpret_cast<const float4 *>(P + (long)min(kr, k_lim - 1) * rs_k + min(mq, mn_lim - 4));
                reg[j] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (kr < k_lim && mq + e < mn_lim) ? p[e] : 0.f;
                reg[j] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <bool KC, int TS>
__device__ __forceinline__ void store_tile(float *__restrict__ S, int t, const float4 (&reg)[TileGeo<TS>::R]) {
    using G = TileGeo<TS>;
    constexpr int R = G::R, QPR = G::QPR, LD = G::LD;
    static_assert(R == 4 || R == 2, "a whole or a half k quad per thread");
    if (KC) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int row = (t >> 3) + 32 * j;
            *reinterpret_cast<float4 *>(S + row * LD + (((t & 7) ^ G::swz(row)) << 2)) = reg[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < R; ++j) *reinterpret_cast<float4 *>(S + (t / QPR + (256 / QPR) * j) * TS + (t % QPR) * 4) = reg[j];
    }
}

// LDS-DMA fill of an operand tile (whole tiles only): wave w, instruction j writes the 1 KB at float offset 256 (4 j + w) of the image with
// one `buffer_load_dwordx4 ... lds` (lane l lands at + 16 l bytes).  [mn][k] image: rows 8 (4 j + w) .. + 7, lane l fetches quad
// (l & 7) ^ swizzle(row) of row + (l >> 3).  [k][mn] image: the image IS the memory order, 16-byte unit u = 64 (4 j + w) + l is k row
// u / (TS / 4), m / n quad u % (TS / 4).
template <int TS>
struct DmaPlan { int voff[TileGeo<TS>::R]; };
template <int TS, bool KC>
__device__ __forceinline__ DmaPlan<TS> dma_plan(long ld_floats, int lane, int wave) {
    DmaPlan<TS> p;
#pragma unroll
    for (int j = 0; j < TileGeo<TS>::R; ++j) {
        if (KC) {
            const int row = 8 * (4 * j + wave) + (lane >> 3);
            p.voff[j] = (int)(((long)row * ld_floats + (((lane & 7) ^ TileGeo<TS>::swz(row)) << 2)) * 4);
        } else {
            const int u = 64 * (4 * j + wave) + lane;
            p.voff[j] = (int)(((long)(u / (TS / 4)) * ld_floats + (u % (TS / 4)) * 4) * 4);
        }
    }
    return p;
}
// KC: element (mn, k) at P[mn ld + k]; MC: at P[k ld + mn]
template <int TS, bool KC>
__device__ __forceinline__ void dma_tile(const float *__restrict__ P, long ld_floats, int mn0, int k0, float *S, const DmaPlan<TS> &p, int wave) {
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    // the descriptor starts at the tile (uniform values only): offsets stay far below 2^31 whatever the matrix size; whole tiles: no range to enforce
    const float *base = KC ? P + (long)mn0 * ld_floats + k0 : P + (long)k0 * ld_floats + mn0;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int j = 0; j < TileGeo<TS>::R; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(S + 256 * (4 * j + wave)), 16, p.voff[j], 0, 0, 0);
}

#ifdef TH_PROFILE
__device__ long long g_gemm_prof[4];   // workgroup 0: wall clock (100 MHz) and shader clock at entry and exit -- the clock the product ran at
#endif

// DXEP: the epilogue with Epilogue::mask / colpart; ADAMEP: C is a complete gradient and the parameter's Adam update (optim.rs:99-110) runs on
// the element in the epilogue, under the matrix work of the CU's other workgroup (their own instances: the plain products keep their registers)
// RAG (r06): a ragged m, n or k (784 = 6 x 128 + 16 = 24 x 32 + 16) no longer sends EVERY tile through the clamped register loads -- the
// operands still arrive by LDS-DMA, a lane whose 16-byte quad lies past the matrix edge asks for an offset past the descriptor's range and
// gets zeros, and only the stores are guarded.  m / n-contiguous operands: quads past m / n (fixed per tile); k-contiguous operands: rows
// past m / n (fixed per tile) and, in the LAST k chunk only, quads past k.  m, n, k multiples of 4 along the contiguous axis (quads in or out).
template <int TS, bool A_KC, bool B_KC, bool GUARD, bool DXEP = false, bool ADAMEP = false, bool RAG = false>
__global__ __launch_bounds__(256, 2) void sgemm_tile(const float *__restrict__ A, const float *__restrict__ B,
                                                     float *__restrict__ C, int m, int n, int k,
                                                     long a_rs, long a_cs, long b_rs, long b_cs,
                                                     int tiles_m, int tiles_n, Epilogue ep, int kslice,
                                                     float *__restrict__ partial, int vec, int raster) {
    // blockIdx.y = K slice [y*kslice, (y+1)*kslice): with `partial` set every slice writes its raw
    // accumulators to partial[y][m*n] and splitk_reduce applies the epilogue in fixed slice order
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TMAX = TileGeo<TS>::TILE_MAX, R = TileGeo<TS>::R, WS = TS / 2, NS = WS / 32;
    // LDS: As[2] then Bs[2], TMAX floats each

    // XCD-aware order: block b runs on XCD b % 8, so give each XCD a
    // contiguous chunk of the tile list (neighbouring tiles share A/B panels
    // in that XCD's L2).  Bijective for any grid size.
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    int tile, zslice = 0;
    const bool slice_xcd = raster >= 0x100;      // (the host's choice rides on the raster argument: sgemm_tile128)
    raster = raster >= 0x100 ? raster - 0x100 : raster;
    if (gridDim.y == 1 || !slice_xcd) {
        const int xcd = bid % kNumXCD, q = nwg / kNumXCD, rmd = nwg % kNumXCD;
        tile = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + bid / kNumXCD;
        zslice = blockIdx.y;
    } else {
        // K slices (grid.y > 1): the launch's workgroups are handed to the XCDs in LINEAR order, x fastest -- the tiles of ONE slice would
        // land on 8 different L2s and each would fetch the slice's rows of A and B for itself (TN 784 x 256 x 16 384, 14 tiles x 36 slices:
        // 188 MB through the fabric for 68 MB of operands).  Give every XCD a contiguous run of (slice, tile) pairs instead: the tiles of a
        // slice run on one XCD, at the same time, and its rows are fetched once: 77 MB, 82.1 -> 79.2 us.  (Slices of fewer than 14 chunks keep
        // the order above -- the host's choice, sgemm_tile128: 14 workgroups asking one L2 for the same rows at once cost more than the second
        // fetch saves -- 784 x 256 x 4 096: 32.9 -> 35.6 us, x 8 192: 57.8 -> 60.1.)
        const int total = nwg * (int)gridDim.y, lin = bid + nwg * (int)blockIdx.y;
        const int xcd = lin % kNumXCD, q = total / kNumXCD, rmd = total % kNumXCD;
        const int s = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + lin / kNumXCD;
        zslice = s / nwg;
        tile = s - zslice * nwg;
    }
    // Within the list the tiles come in GROUPS of 8 tile rows, column-major inside a group: the 64 tiles an XCD has in flight (32 CUs x 2
    // workgroups) then form an 8 x 8 block -- every A / B panel it fetches into its L2 serves 8 tiles -- instead of two whole columns of
    // tiles_m rows, where each A panel served 2 (4096^3 NT: 2.2 GB through the fabric for 0.2 GB of operands).  raster < 0: the r02 order.
    int tm, tn;
    if (raster > 0) {
        const int grp = tile / (raster * tiles_n), first = grp * raster, gh = min(raster, tiles_m - first), in = tile - grp * raster * tiles_n;
        tm = first + in % gh;
        tn = in / gh;
    } else {
        tm = tile % tiles_m;
        tn = tile / tiles_m;
    }
    const int row0 = tm * TS, col0 = tn * TS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#ifdef TH_PROFILE
    if (bid == 0 && zslice == 0 && t == 0) { g_gemm_prof[0] = wall_clock64(); g_gemm_prof[1] = clock64(); }
#endif
    const int wm = (wave >> 1) * WS, wn = (wave & 1) * WS;  // wave's WS x WS sub-tile
    const int li = lane & 31, lk = lane >> 5;

    floatx16 acc[NS][NS];
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // operand staging: whole-tile launches bring both operands global -> LDS by LDS-DMA (no staging registers, no ds_write, no transpose);
    // ragged shapes go through registers (clamped loads, zero fill) into the same images
    constexpr bool A_DMA = !GUARD, B_DMA = !GUARD;
    float4 ra[A_DMA ? 1 : R], rb[B_DMA ? 1 : R];
    const int kbeg = zslice * kslice, kend = min(k, kbeg + kslice);
    const int nt = (kend - kbeg + BK - 1) / BK;
    DmaPlan<TS> pa{}, pb{};
    if constexpr (A_DMA) pa = dma_plan<TS, A_KC>(A_KC ? a_rs : a_cs, lane, wave);
    if constexpr (B_DMA) pb = dma_plan<TS, B_KC>(B_KC ? b_cs : b_rs, lane, wave);
    DmaPlan<TS> pa_last = pa, pb_last = pb;      // RAG: the plan of the slice's last k chunk (k-contiguous operands: quads past k zeroed)
    if constexpr (RAG) {
        static_assert(!RAG || !GUARD, "edge quads by range: operands through LDS-DMA");
        const int k_left = kend - (kbeg + (nt - 1) * BK);          // valid k of the last chunk (a multiple of 4)
#pragma unroll
        for (int j = 0; j < R; ++j) {
            constexpr int OOB = 0x7fffffff;    // past the descriptor's range: the lane's 16 bytes arrive as zeros
            if constexpr (A_KC) {              // [mn][k] image: row 8 (4 j + wave) + (lane >> 3), k quad (lane & 7) ^ swizzle(row) (dma_plan)
                const int row = 8 * (4 * j + wave) + (lane >> 3), kq = ((lane & 7) ^ TileGeo<TS>::swz(row)) << 2;
                if (row0 + row >= m) pa.voff[j] = OOB;
                pa_last.voff[j] = (row0 + row >= m || kq >= k_left) ? OOB : pa.voff[j];
            } else {                           // [k][mn] image: unit u = 64 (4 j + wave) + lane: k row u / (TS / 4), m quad u % (TS / 4)
                const int u = 64 * (4 * j + wave) + lane;
                if (row0 + 4 * (u % (TS / 4)) >= m) pa.voff[j] = OOB;
                pa_last.voff[j] = (u / (TS / 4) >= k_left) ? OOB : pa.voff[j];
            }
            if constexpr (B_KC) {
                const int row = 8 * (4 * j + wave) + (lane >> 3), kq = ((lane & 7) ^ TileGeo<TS>::swz(row)) << 2;
                if (col0 + row >= n) pb.voff[j] = OOB;
                pb_last.voff[j] = (col0 + row >= n || kq >= k_left) ? OOB : pb.voff[j];
            } else {
                const int u = 64 * (4 * j + wave) + lane;
                if (col0 + 4 * (u % (TS / 4)) >= n) pb.voff[j] = OOB;
                pb_last.voff[j] = (u / (TS / 4) >= k_left) ? OOB : pb.voff[j];
            }
        }
    }
    // element (i,k) of op(A) at A[i*a_rs + k*a_cs]; (k,j) of op(B) at B[k*b_rs + j*b_cs]
    auto fetch = [&](int k0, int stage) {
        const bool last = RAG && k0 + BK >= kend;      // (uniform)
        if constexpr (A_DMA) dma_tile<TS, A_KC>(A, A_KC ? a_rs : a_cs, row0, k0, smem + stage * TMAX, last ? pa_last : pa, wave);
        else load_tile<A_KC, GUARD, TS>(A, a_rs, a_cs, row0, k0, m, kend, t, ra, vec);
        if constexpr (B_DMA) dma_tile<TS, B_KC>(B, B_KC ? b_cs : b_rs, col0, k0, smem + (2 + stage) * TMAX, last ? pb_last : pb, wave);
        else load_tile<B_KC, GUARD, TS>(B, b_cs, b_rs, col0, k0, n, kend, t, rb, vec);
    };
    auto stash = [&](int stage) {
        if constexpr (!A_DMA) store_tile<A_KC, TS>(smem + stage * TMAX, t, ra);
        if constexpr (!B_DMA) store_tile<B_KC, TS>(smem + (2 + stage) * TMAX, t, rb);
        if constexpr (A_DMA || B_DMA) __builtin_amdgcn_s_waitcnt(0);   // the DMA writes of this wave have landed (the compiler does not track them)
    };
    fetch(kbeg, 0);
    stash(0);
    __syncthreads();

    // this lane's LDS addresses.  [mn][k] image: row (wave sub-tile + 32 i + li), k quad 2 r + lk of round r, swizzled -- the quads of rounds
    // r = 0..3 differ from quad lk by an XOR of r << 1 on the quad index, r << 3 on the float offset.  [k][mn] image: k row 4 lk, column row;
    // the four operands of round r lie (8 r + e) rows on.
    int ao[NS], bo[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int ar = wm + 32 * i + li, br = wn + 32 * i + li;
        ao[i] = A_KC ? ar * TileGeo<TS>::LD + ((lk ^ TileGeo<TS>::swz(ar)) << 2) : 4 * lk * TS + ar;
        bo[i] = B_KC ? br * TileGeo<TS>::LD + ((lk ^ TileGeo<TS>::swz(br)) << 2) : 4 * lk * TS + br;
    }

    for (int it = 0; it < nt; ++it) {
        const int cur = it & 1;
#ifndef GEMM_PROBE_NOLOAD   /* timing probes (wrong results): tools/prof_gemm_probes.sh */
        if (it + 1 < nt) fetch(kbeg + (it + 1) * BK, cur ^ 1);
#endif
        const float *as = smem + cur * TMAX, *bs = smem + (2 + cur) * TMAX;
        // eight k per round: lane half lk holds k = 8 r + 4 lk + e, e = 0..3, of its row -- MFMA e of round r contracts k = {8 r + e,
        // 8 r + 4 + e} (a permutation of the k order inside the tile; fp32 sums within tolerance).  The operands of round r + 1 are
        // requested before the MFMAs of round r and pinned there.
        float4 af[2][NS], bf[2][NS];
#ifdef GEMM_PROBE_NOREAD
#define TH_GEMM_REQ(SET, RR)                                                          \
    _Pragma("unroll") for (int i = 0; i < NS; ++i) {                                  \
        af[SET][i] = make_float4((float)(RR + it), 1.f, 2.f, 3.f);                    \
        bf[SET][i] = make_float4(1.f, (float)(RR + i), 2.f, 3.f);                     \
    }
#else
#define TH_GEMM_FRAG(KC, S, O, RR)                                                                                                 \
    ((KC) ? *reinterpret_cast<const float4 *>((S) + ((O) ^ ((RR) << 3)))                                                           \
          : make_float4((S)[(O) + (8 * (RR)) * TS], (S)[(O) + (8 * (RR) + 1) * TS], (S)[(O) + (8 * (RR) + 2) * TS], (S)[(O) + (8 * (RR) + 3) * TS]))
#define TH_GEMM_REQ(SET, RR)                                                          \
    _Pragma("unroll") for (int i = 0; i < NS; ++i) {                                  \
        af[SET][i] = TH_GEMM_FRAG(A_KC, as, ao[i], RR);                               \
        bf[SET][i] = TH_GEMM_FRAG(B_KC, bs, bo[i], RR);                               \
    }
#endif
#define TH_GEMM_MFMA(CS, E)                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NS; ++i)                                                                                 \
        _Pragma("unroll") for (int j = 0; j < NS; ++j)                                                                             \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[CS][i].E, bf[CS][j].E, acc[i][j], 0, 0, 0);
        TH_GEMM_REQ(0, 0)
#pragma unroll
        for (int r = 0; r < BK / 8; ++r) {
            const int cs = r & 1;
            if (r + 1 < BK / 8) { TH_GEMM_REQ(cs ^ 1, r + 1) }
            __builtin_amdgcn_sched_barrier(0);
            TH_GEMM_MFMA(cs, x)
            TH_GEMM_MFMA(cs, y)
            TH_GEMM_MFMA(cs, z)
            TH_GEMM_MFMA(cs, w)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef TH_GEMM_MFMA
#undef TH_GEMM_REQ
#undef TH_GEMM_FRAG
#ifndef GEMM_PROBE_NOSTORE
        if (it + 1 < nt) stash(cur ^ 1);
#endif
#ifndef GEMM_PROBE_NOSYNC
        __syncthreads();
#endif
    }

#ifdef TH_PROFILE
    if (bid == 0 && zslice == 0 && t == 0) { g_gemm_prof[2] = wall_clock64(); g_gemm_prof[3] = clock64(); }
#endif
    // epilogue.  C/D map of 32x32x2: col = lane & 31, row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5)
    const bool masked = DXEP && ep.mask != nullptr && !partial;
    float adam_step = 0.f;
    if constexpr (ADAMEP) adam_step = adam_dev_step(ep.adam);
    float csum[NS];                             // this lane's share of the column sums of what it stores (columns wn + 32 j + li)
#pragma unroll
    for (int j = 0; j < NS; ++j) csum[j] = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int col = col0 + wn + 32 * j + li;
            float mk[16];
            if (masked) {                       // the mask's 16 values of this sub-tile are requested together
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lk;
                    mk[e] = (!(GUARD || RAG) || (row < m && col < n)) ? ep.mask[(long)row * n + col] : 0.f;
                }
            }
            // ADAMEP: the 24 operands of eight elements' updates are requested together (a round trip per eight elements, not per element)
            float ap[ADAMEP ? 8 : 1], am[ADAMEP ? 8 : 1], av[ADAMEP ? 8 : 1];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if constexpr (ADAMEP) {
                    if ((e & 7) == 0) {
#pragma unroll
                        for (int f = 0; f < 8; ++f) {
                            const int rowf = row0 + wm + 32 * i + ((e + f) & 3) + 8 * ((e + f) >> 2) + 4 * lk;
                            const long idf = (!(GUARD || RAG) || (rowf < m && col < n)) ? (long)rowf * n + col : 0;
                            ap[f] = ep.adam.p[idf];
                            am[f] = ep.adam.m[idf];
                            av[f] = ep.adam.v[idf];
                        }
                    }
                }
                const int row = row0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (!(GUARD || RAG) || (row < m && col < n)) {
                    const long idx = (long)row * n + col;
                    if (partial) {
                        partial[(long)zslice * m * n + idx] = acc[i][j][e];
                        continue;
                    }
                    const float c_old = ep.beta != 0.0f ? C[idx] : 0.0f;
                    float v = epilogue_apply(acc[i][j][e], c_old, ep, col);
                    if (masked) v = mk[e] > 0.f ? v : 0.f;
                    C[idx] = v;
                    csum[j] += v;
                    if constexpr (ADAMEP) {     // adam_update's arithmetic, element by element
                        const AdamDev &ad = ep.adam;
                        const float gv = v + ad.wd * ap[e & 7];
                        const float mn = ad.beta1 * am[e & 7] + (1.0f - ad.beta1) * gv;
                        const float vn = ad.beta2 * av[e & 7] + (1.0f - ad.beta2) * gv * gv;
                        ad.m[idx] = mn;
                        ad.v[idx] = vn;
                        ad.p[idx] = ap[e & 7] - adam_step * mn / (sqrtf(vn) + ad.eps);
                    }
                }
            }
        }
    if (DXEP && ep.colpart && !partial) {
        // rows in a fixed order: a lane's 16 NS values above, then the two lane halves, then the two waves of a column half through LDS
        // (the staging buffers are dead: every wave is behind the loop's last barrier)
        float *cs = smem;                       // [2 row halves][TS columns]
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            float v = csum[j];
            v += __shfl_xor(v, 32, 64);
            if (lk == 0) cs[(wave >> 1) * TS + wn + 32 * j + li] = v;
        }
        __syncthreads();
        if (t < TS) {
            const int col = col0 + t;
            if (!(GUARD || RAG) || col < n) ep.colpart[(long)tm * n + col] = cs[t] + cs[TS + t];
        }
    }
#endif
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

template <bool A_KC, bool B_KC>
static int launch_small(th_ctx *ctx, const float *A, const float *B, float *C, int m, int n, int k, long a_rs, long a_cs,
                        long b_rs, long b_cs, const Epilogue &ep) {
    const int tiles_m = ceil_div(m, 16), tiles_n = ceil_div(n, 16);
    const long tiles = (long)tiles_m * tiles_n;
    SmallArgs p{A, nullptr, B, C, nullptr, m, n, k, a_rs, a_cs, b_rs, b_cs, 0, 0, 0, ep};
    // float4 operand loads need 16-B aligned rows and slice starts
    p.a_vec = A_KC && aligned16(A) && (a_rs % 4 == 0);
    p.b_vec = B_KC && aligned16(B) && (b_cs % 4 == 0);
    // Deep K on a grid that cannot fill the chip: first widen the workgroup to
    // 16 waves (in-LDS reduction, still one launch); only very deep K goes to a
    // grid-level split with a second reduce pass.
    int kz = 1;
    const bool wide = tiles < 256 && k >= 256;
    if (tiles < 64 && k >= 2048) {   // e.g. the simple CNN's Linear(3136, 10) at batch 256: 16 tiles
        kz = (int)((256 + tiles - 1) / tiles);
        const int kz_max = k / 512;
        if (kz > kz_max) kz = kz_max;
        if (kz < 1) kz = 1;
    }
    int kslice = ceil_div(k, kz);
    kslice = (kslice + 15) / 16 * 16;
    kz = ceil_div(k, kslice);
    p.kslice = kslice;
    if (kz > 1) {
        void *ws = nullptr;
        if (th_malloc(ctx, (size_t)kz * m * n * sizeof(float), &ws)) return 1;
        p.partial = (float *)ws;
    }
    if (wide)
        hipLaunchKernelGGL((sgemm_small16<A_KC, B_KC, 16>), dim3(tiles_n, tiles_m, kz), dim3(1024), 0, ctx->stream, p);
    else
        hipLaunchKernelGGL((sgemm_small16<A_KC, B_KC, 4>), dim3(tiles_n, tiles_m, kz), dim3(256), 0, ctx->stream, p);
    TH_LAUNCH_CHECK();
    if (kz > 1) {
        const long mn = (long)m * n;
        launch_splitk_reduce(ctx->stream, (const float *)p.partial, C, mn, n, kz, ep);
        TH_LAUNCH_CHECK();
        if (th_free(ctx, p.partial)) return 1;
    }
    return 0;
}

int adam_slice(th_ctx *ctx, const AdamDev &a, const float *d_g, int64_t n);  // optim.hip

// K slices for the 128x128 kernel: only when the tile grid cannot fill the chip and K is deep
// enough that every slice still runs >= 8 k-iterations of 32
static inline int tile128_kz(int m, int n, int k) {
    const long tiles = (long)ceil_div(m, BM) * ceil_div(n, BN);
    if (tiles >= 384 || k < 512) return 1;
    static const int kz_env = [] { const char *e = getenv("TAPER_GEMM_KZ"); return e ? atoi(e) : 0; }();   // measurement probe
    if (kz_env > 0) return kz_env;
    // 2 workgroups per CU (__launch_bounds__(256, 2)) = 512 places: as many slices as fit in ONE round of them (r05 rounded up: 14 tiles x 37
    // slices = 518 workgroups, six of which ran a second round on their own)
    int kz = (int)(512 / tiles);
    const int kz_max = k / 256;
    if (kz > kz_max) kz = kz_max;
    return kz < 1 ? 1 : kz;
}

// K slices for the 64x64 kernel: target 2 workgroups per CU, slices of >= 128 k
static inline int tile64_kz(int m, int n, int k) {
    const long tiles = (long)ceil_div(m, 64) * ceil_div(n, 64);
    if (tiles >= 384 || k < 256) return 1;
    int kz = (int)((512 + tiles - 1) / tiles);
    const int kz_max = k / 128;
    if (kz > kz_max) kz = kz_max;
    return kz < 1 ? 1 : kz;
}

template <int TS, bool A_KC, bool B_KC>
static inline bool rag_dma(int m, int n, int k, bool vec, long lda, long ldb) {
    static const bool on = [] { const char *e = getenv("TAPER_GEMM_RAG"); return !e || atoi(e) != 0; }();   // 0: the clamped register loads (A/B probe)
    // (`vec`: 16-byte aligned rows and whole quads in or out of range along each operand's contiguous axis)
    return on && TS == 128 && vec && ((A_KC || B_KC) ? k % 4 == 0 : k % BK == 0) && lda < (1L << 22) && ldb < (1L << 22);
}

template <int TS, bool A_KC, bool B_KC>
static int launch_tile(th_ctx *ctx, const float *A, const float *B, float *C, int m, int n, int k, long a_rs,
                       long a_cs, long b_rs, long b_cs, const Epilogue &ep) {
    const int tiles_m = ceil_div(m, TS), tiles_n = ceil_div(n, TS);
    const size_t lds = 4 * TileGeo<TS>::TILE_MAX * sizeof(float);
    const long lda = A_KC ? a_rs : a_cs, ldb = B_KC ? b_cs : b_rs;
    int kz = TS == 128 ? tile128_kz(m, n, k) : tile64_kz(m, n, k);
    int kslice = ceil_div(ceil_div(k, kz), BK) * BK;
    kz = ceil_div(k, kslice);
    // dwordx4 operand loads: 16-B aligned rows, and whole quads in or out of range along the vector axis
    // (k for a k-contiguous operand, m / n for an m/n-contiguous one); slices start on multiples of 32
    const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0) && ((A_KC ? k : m) % 4 == 0) &&
                     ((B_KC ? k : n) % 4 == 0) && m >= 4 && n >= 4 && k >= 4;
    // whole tiles take the LDS-DMA kernel: its 32-bit lane offsets span a tile's rows (TS x leading dimension x 4 bytes < 2^31)
    const bool exact = (m % TS == 0) && (n % TS == 0) && (k % BK == 0) && vec && lda < (1L << 22) && ldb < (1L << 22);
    float *partial = nullptr;
    if (kz > 1) {
        void *ws = nullptr;
        if (th_malloc(ctx, (size_t)kz * m * n * sizeof(float), &ws)) return 1;
        partial = (float *)ws;
    }
    // a fused Adam update belongs to the pass that completes the gradient: the reduce pass of a split product; the product's own epilogue
    // when it is unsplit and runs on whole 128-tiles in the weight gradient's layout (TN: dW = dZ^T X, th_linear_bwd_adam_ex2's big shapes --
    // 16.7 M parameters of a 4096 x 4096 layer cost a pass of 75 us over p / m / v / g as a launch of their own)
    Epilogue kep = ep;
    kep.adam.p = nullptr;
    static const bool adam_ep_on = [] { const char *e = getenv("TAPER_GEMM_ADAM_EP"); return !e || atoi(e) != 0; }();   // 0: a slice launch behind the product (measurement)
    if constexpr (TS == 128 && !A_KC && !B_KC) {
        if (ep.adam.p && kz == 1 && exact && adam_ep_on && !ep.mask && !ep.colpart) {
            auto kern = sgemm_tile<TS, A_KC, B_KC, false, false, true>;
            TH_SET_MAX_LDS(ctx, kern, lds);
            static const int raster_a = [] { const char *e = getenv("TAPER_GEMM_RASTER"); return e ? atoi(e) : 8; }();
            hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, 1), dim3(256), lds, ctx->stream, A, B, C, m, n, k, a_rs, a_cs, b_rs, b_cs, tiles_m, tiles_n, ep,
                               kslice, (float *)nullptr, 1, raster_a);
            TH_LAUNCH_CHECK();
            return 0;
        }
    }
    // tile order: groups of 8 tile rows (TAPER_GEMM_RASTER = n: groups of n; 0: r02's whole columns)
    static const int raster0 = [] { const char *e = getenv("TAPER_GEMM_RASTER"); return e ? atoi(e) : 8; }();
    // K slices of at least this many chunks are handed to the XCDs slice by slice (sgemm_tile; TAPER_GEMM_SLICE_XCD_CHUNKS, 0 = always, large = never)
    static const int slice_xcd_chunks = [] { const char *e = getenv("TAPER_GEMM_SLICE_XCD_CHUNKS"); return e ? atoi(e) : 14; }();
    const int raster = raster0 + ((raster0 >= 0 && kz > 1 && kslice >= slice_xcd_chunks * BK) ? 0x100 : 0);
    if (ep.mask || ep.colpart) {     // the dX product with the backward of the layer in front in its epilogue (th_linear_bwd_adam_ex2: unsplit by its predicate)
        if (kz != 1) { th::set_error("sgemm_tile: the masked epilogue needs an unsplit product"); return 2; }
        if (exact) {
            auto kern = sgemm_tile<TS, A_KC, B_KC, false, true>;
            TH_SET_MAX_LDS(ctx, kern, lds);
            hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, 1), dim3(256), lds, ctx->stream, A, B, C, m, n, k, a_rs, a_cs, b_rs, b_cs, tiles_m, tiles_n, kep,
                               kslice, (float *)nullptr, 1, raster);
        } else {
            auto kern = sgemm_tile<TS, A_KC, B_KC, true, true>;
            TH_SET_MAX_LDS(ctx, kern, lds);
            hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, 1), dim3(256), lds, ctx->stream, A, B, C, m, n, k, a_rs, a_cs, b_rs, b_cs, tiles_m, tiles_n, kep,
                               kslice, (float *)nullptr, vec ? 1 : 0, raster);
        }
    } else if (exact) {
        auto kern = sgemm_tile<TS, A_KC, B_KC, false>;
        TH_SET_MAX_LDS(ctx, kern, lds);       // (per device: ADVICE r04)
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, kz), dim3(256), lds, ctx->stream, A, B, C, m, n, k, a_rs, a_cs,
                           b_rs, b_cs, tiles_m, tiles_n, kep, kslice, partial, 1, raster);
    } else if (rag_dma<TS, A_KC, B_KC>(m, n, k, vec, lda, ldb)) {
        if constexpr (TS == 128) {      // ragged edges: LDS-DMA with the edge quads zeroed by range, guarded stores
            auto kern = sgemm_tile<TS, A_KC, B_KC, false, false, false, true>;
            TH_SET_MAX_LDS(ctx, kern, lds);
            hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, kz), dim3(256), lds, ctx->stream, A, B, C, m, n, k, a_rs, a_cs,
                               b_rs, b_cs, tiles_m, tiles_n, kep, kslice, partial, 1, raster);
        }
    } else {
        auto kern = sgemm_tile<TS, A_KC, B_KC, true>;
        TH_SET_MAX_LDS(ctx, kern, lds);       // (per device: ADVICE r04)
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, kz), dim3(256), lds, ctx->stream, A, B, C, m, n, k, a_rs, a_cs,
                           b_rs, b_cs, tiles_m, tiles_n, kep, kslice, partial, vec ? 1 : 0, raster);
    }
    TH_LAUNCH_CHECK();
    if (kz > 1) {
        const long mn = (long)m * n;
        launch_splitk_reduce(ctx->stream, (const float *)partial, C, mn, n, kz, ep);
        TH_LAUNCH_CHECK();
        if (th_free(ctx, partial)) return 1;
    } else if (ep.adam.p) {
        return adam_slice(ctx, ep.adam, C, (int64_t)m * n);   // no reduce pass to ride on
    }
    return 0;
}

template <bool A_KC, bool B_KC>
static int launch_tile128(th_ctx *ctx, const float *A, const float *B, float *C, int m, int n, int k, long a_rs,
                          long a_cs, long b_rs, long b_cs, const Epilogue &ep) {
    return launch_tile<128, A_KC, B_KC>(ctx, A, B, C, m, n, k, a_rs, a_cs, b_rs, b_cs, ep);
}


static inline bool gemm_is_big(int m, int n, int k) {
    const long tiles128 = (long)ceil_div(m, BM) * ceil_div(n, BN);
    if (m < BM || n < BN || k < BK) return false;
    return tiles128 >= 64 || tiles128 * tile128_kz(m, n, k) >= 96;   // deep K: split-K slices fill the chip (below: 16x16 tiles with K slices)
}

// 64x64 tiles: outputs with too few 128-tiles for the big kernel but far too much work for 16x16 tiles straight from L2
// (the MLP's tall-skinny products at batch 512..8192: [B,128] = X . W1^T and [128,784] = dZ^T . X)
static inline bool gemm_is_mid(int m, int n, int k) {
    static const long min_macs = getenv("TAPER_GEMM_MID_MACS") ? atol(getenv("TAPER_GEMM_MID_MACS")) : 300000000L;
    if (m < 64 || n < 64 || k < 64) return false;
    const long tiles64 = (long)ceil_div(m, 64) * ceil_div(n, 64);
    return (long)m * n * k >= min_macs && tiles64 * tile64_kz(m, n, k) >= 32;
}

// op(A)[i,k] = A[i*a_rs + k*a_cs], op(B)[k,j] = B[k*b_rs + j*b_cs]
int gemm_dispatch(th_ctx *ctx, int trans_a, int trans_b, int m, int n, int k, const float *A, const float *B, float *C,
                  const Epilogue &ep) {
    if (m == 0 || n == 0) return 0;
    const long a_rs = trans_a ? 1 : k, a_cs = trans_a ? m : 1;  // gemm.rs:88-92
    const long b_rs = trans_b ? 1 : n, b_cs = trans_b ? k : 1;  // gemm.rs:93-97
    const bool a_kc = !trans_a, b_kc = trans_b != 0;
    const bool big = gemm_is_big(m, n, k);
    // Layouts.  r01 ran deep NN / TN / TT products as NT on transposed COPIES (a transpose launch per m/n-contiguous operand); r02 transposed
    // such operands on their way into LDS; r03 stages every operand in the layout it has in memory (sgemm_tile: two LDS images, one MFMA loop).
    // TAPER_GEMM_PRETRANSPOSE=1 still runs the r01 form for comparison.
    static const int pretranspose = getenv("TAPER_GEMM_PRETRANSPOSE") ? atoi(getenv("TAPER_GEMM_PRETRANSPOSE")) : 0;   // measurement probe
    if (pretranspose && big && k >= 1024 && m >= 1024 && n >= 1024 && (!a_kc || !b_kc)) {
        void *at = nullptr, *bt = nullptr;
        const float *A2 = A, *B2 = B;
        if (!a_kc) {  // A stored [k][m] -> [m][k]
            if (th_malloc(ctx, (size_t)m * k * sizeof(float), &at)) return 1;
            if (int rc = th_transpose2d(ctx, A, (float *)at, k, m)) return rc;
            A2 = (const float *)at;
        }
        if (!b_kc) {  // B stored [k][n] -> [n][k]
            if (th_malloc(ctx, (size_t)n * k * sizeof(float), &bt)) return 1;
            if (int rc = th_transpose2d(ctx, B, (float *)bt, k, n)) return rc;
            B2 = (const float *)bt;
        }
        if (int rc = launch_tile128<true, true>(ctx, A2, B2, C, m, n, k, k, 1, 1, k, ep)) return rc;
        if (at && th_free(ctx, at)) return 1;
        if (bt && th_free(ctx, bt)) return 1;
        return 0;
    }
    // a "big" product whose 128-tiles (x K slices) cannot give every CU a workgroup runs on 64-tiles instead
    static const long big_min_wg = getenv("TAPER_GEMM_BIG_WG") ? atol(getenv("TAPER_GEMM_BIG_WG")) : 460;
    const long wg128 = (long)ceil_div(m, BM) * ceil_div(n, BN) * tile128_kz(m, n, k);
    const bool mid = gemm_is_mid(m, n, k) && (!big || wg128 < big_min_wg);
#define TH_GEMM_CASE(AK, BKC)                                                                         \
    if (a_kc == AK && b_kc == BKC)                                                                          \
        return mid ? launch_tile<64, AK, BKC>(ctx, A, B, C, m, n, k, a_rs, a_cs, b_rs, b_cs, ep)            \
             : big ? launch_tile128<AK, BKC>(ctx, A, B, C, m, n, k, a_rs, a_cs, b_rs, b_cs, ep)             \
                   : launch_small<AK, BKC>(ctx, A, B, C, m, n, k, a_rs, a_cs, b_rs, b_cs, ep);
    TH_GEMM_CASE(true, true)
    TH_GEMM_CASE(true, false)
    TH_GEMM_CASE(false, true)
    TH_GEMM_CASE(false, false)
#undef TH_GEMM_CASE
    return 3;
}

// th_mlp3_xent's second launch: layer l: dW_l[out][in] = dZ_l^T . A_l, db_l = column sums of dZ_l (dZ already masked), Adam fused where given
int mlp3_grads_launch(th_ctx *ctx, const float *const dz[3], const float *const act[3], float *const dw[3], float *const db[3],
                      const int out_f[3], const int in_f[3], const th_adam_fuse *const wf[3], const th_adam_fuse *const bf[3], int batch,
                      const float *part, int n_blk, float *loss, float *ncorrect, float *metrics, int64_t capacity, int64_t *state,
                      int64_t advance, const float *gap_dx, const th_mlp3_gap *gap) {
    Mlp3GradArgs a{};
    int blocks = 0;
    for (int l = 0; l < 3; ++l) {
        LinearBwdArgs &q = a.job[l];
        const int tm = ceil_div(out_f[l], 16), tn = ceil_div(in_f[l], 16);
        q.dw_kz = 1;
        q.dw_tiles = dw[l] ? tm * tn : 0;
        q.n_dw = (q.dw_tiles + 7) & ~7;
        q.dw_tiles_m = tm;
        // dW = dZ^T . A : op(A)[i = o, k = b] = dZ[b * out + o] (rs 1, cs out); op(B)[k = b, j] = A[b * in + j]
        q.dw = SmallArgs{dz[l], nullptr, act[l], dw[l], nullptr, out_f[l], in_f[l], batch, 1, out_f[l], in_f[l], 1, (batch + 15) / 16 * 16, 0, 0,
                         make_ep(1.0f, 0.0f)};
        q.dw.ep.adam = make_adam_dev(dw[l] ? wf[l] : nullptr);
        q.db_adam = make_adam_dev(db[l] ? bf[l] : nullptr);
        q.dy = dz[l];
        q.db = db[l];
        q.batch = batch;
        q.out_f = out_f[l];
        q.n_db = db[l] ? ceil_div(out_f[l], 64) : 0;
        a.first[l] = blocks;
        blocks += q.n_dw + q.n_db;
    }
    a.first[3] = blocks;
    if (gap && gap->d_cnt) {
        a.gap_dx = gap_dx; a.gap_cnt = gap->d_cnt; a.gap_db = gap->d_gb; a.gap_c = in_f[0]; a.gap_hw = gap->hw;
        a.gap_adam = make_adam_dev(gap->b_fuse);
        blocks += ceil_div(a.gap_c, 16);
    }
    a.first[4] = blocks;
    a.part = part; a.n_blk = n_blk; a.batch = batch;
    a.loss = loss; a.ncorrect = ncorrect; a.metrics = metrics; a.capacity = capacity; a.state = state; a.advance = advance;
    hipLaunchKernelGGL(mlp3_grads_kernel, dim3(blocks + 1), dim3(256), 0, ctx->stream, a);
    TH_LAUNCH_CHECK();
    return 0;
}

// First half of th_linear_xent_wide: the K slices of logits = X . W^T (no bias, no reduce) into a pool workspace
// partial[kz][m*n]; the caller's kernel adds the slices in order and frees the workspace.
int linear_fwd_partials(th_ctx *ctx, const float *x, const float *w, int m, int n, int k, float **partial_out, int *kz_out) {
    const int tiles_m = ceil_div(m, 16), tiles_n = ceil_div(n, 16);
    const long tiles = (long)tiles_m * tiles_n;
    SmallArgs p{x, nullptr, w, nullptr, nullptr, m, n, k, k, 1, 1, k, 0, 0, 0, make_ep(1.0f, 0.0f)};
    p.a_vec = aligned16(x) && (k % 4 == 0);
    p.b_vec = aligned16(w) && (k % 4 == 0);
    int kz = (int)std::min<long>((256 + tiles - 1) / tiles, k / 512);
    if (kz < 1) kz = 1;
    if (kz > 8) kz = 8;   // wide_head.hip reads all slices of a logit at once (WH_KZ_MAX)
    static const int kz_env = getenv("TAPER_WIDE_KZ") ? atoi(getenv("TAPER_WIDE_KZ")) : 0;   // tuning probe: cap on the K slices
    if (kz_env > 0 && kz > kz_env) kz = kz_env;
    const int kslice = (ceil_div(k, kz) + 15) / 16 * 16;
    kz = ceil_div(k, kslice);
    p.kslice = kslice;
    void *ws = nullptr;
    if (th_malloc(ctx, (size_t)kz * m * n * sizeof(float), &ws)) return 1;
    p.partial = (float *)ws;
    if (tiles < 256 && k >= 256) hipLaunchKernelGGL((sgemm_small16<true, true, 16>), dim3(tiles_n, tiles_m, kz), dim3(1024), 0, ctx->stream, p);
    else hipLaunchKernelGGL((sgemm_small16<true, true, 4>), dim3(tiles_n, tiles_m, kz), dim3(256), 0, ctx->stream, p);
    TH_LAUNCH_CHECK();
    *partial_out = p.partial;
    *kz_out = kz;
    return 0;
}

}  // namespace th

using namespace th;

extern "C" {

#ifdef TH_PROFILE
int th_debug_gemm_prof(th_ctx *ctx, long long *h_out4) {
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpyFromSymbol(h_out4, HIP_SYMBOL(th::g_gemm_prof), 4 * sizeof(long long)));
    return 0;
}
#endif

int th_sgemm(th_ctx *ctx, int trans_a, int trans_b, int m, int n, int k, float alpha, const float *d_a,
             const float *d_b, float beta, float *d_c) {
    TH_REQUIRE(ctx, "th_sgemm: null ctx");
    TH_REQUIRE(m >= 0 && n >= 0 && k >= 0, "th_sgemm: negative dimension");
    TH_REQUIRE((m == 0 || n == 0) || (d_c && (k == 0 || (d_a && d_b))), "th_sgemm: null device pointer");
    Epilogue ep = make_ep(alpha, beta);
    return gemm_dispatch(ctx, trans_a, trans_b, m, n, k, d_a, d_b, d_c, ep);
}

int th_linear_fwd(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_b, float *d_y, int batch,
                  int in_features, int out_features, int relu) {
    TH_REQUIRE(ctx && d_x && d_w && d_y, "th_linear_fwd: null argument");
    // Y = X . W^T: op(B) = W^T with W stored [out, in]  ->  trans_b
    Epilogue ep = make_ep(1.0f, 0.0f, d_b, relu);
    return gemm_dispatch(ctx, 0, 1, batch, out_features, in_features, d_x, d_w, d_y, ep);
}

int th_linear_fwd_ex(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_b, float *d_y, int batch, int in_features,
                     int out_features, int relu, const th_adam_slice *extra, int n_extra, int32_t *d_tick) {
    TH_REQUIRE(ctx && d_x && d_w && d_y, "th_linear_fwd_ex: null argument");
    TH_REQUIRE(n_extra >= 0 && n_extra <= TH_MAX_ADAM_SLICES && (n_extra == 0 || extra), "th_linear_fwd_ex: bad extra slices");
    for (int i = 0; i < n_extra; ++i)
        TH_REQUIRE(extra[i].f.d_p != d_w && extra[i].f.d_p != d_b, "th_linear_fwd_ex: a carried slice must not alias what this launch reads");
    const int m = batch, n = out_features, k = in_features;
    const long tiles = (long)ceil_div(m, 16) * ceil_div(n, 16);
    // the latency-bound case this entry exists for: 16x16 tiles, no grid-level K split -> ONE launch
    if (m > 0 && n > 0 && k > 0 && !gemm_is_big(m, n, k) && !(tiles < 64 && k >= 2048)) {
        SmallArgs p{d_x, nullptr, d_w, d_y, nullptr, m, n, k, k, 1, 1, k, (k + 15) / 16 * 16, 0, 0, make_ep(1.0f, 0.0f, d_b, relu)};
        p.a_vec = aligned16(d_x) && (k % 4 == 0);
        p.b_vec = aligned16(d_w) && (k % 4 == 0);
        const AdamSlices x = make_adam_slices(extra, n_extra, ctx);
        const dim3 grid(ceil_div(n, 16), ceil_div(m, 16) + 1, 1);
        static const int xcd_map = [] { const char *e = std::getenv("TAPER_K1_XCD_MAP"); return e ? atoi(e) : 1; }();
        p.xcd_blocks = xcd_map && grid.x == 8 && grid.y == 5;
        if (tiles < 256 && k >= 256) hipLaunchKernelGGL((sgemm_small16_tick<true, true, 16>), grid, dim3(1024), 0, ctx->stream, p, x, d_tick);
        else hipLaunchKernelGGL((sgemm_small16_tick<true, true, 4>), grid, dim3(256), 0, ctx->stream, p, x, d_tick);
        TH_LAUNCH_CHECK();
        return 0;
    }
    // other shapes: the slices (old counter), the tick, then the product
    if (int rc = th_adam_slices(ctx, extra, n_extra)) return rc;
    if (d_tick)
        if (int rc = th_adam_tick(ctx, d_tick)) return rc;
    return th_linear_fwd(ctx, d_x, d_w, d_b, d_y, batch, in_features, out_features, relu);
}

}  // extern "C" (reopened below)

namespace th {

// dX = dZ . W for a THIN layer (out_features <= 16: a classifier on a wide input -- BASELINE configs[4]'s Linear(4096, 10) at batch 4096):
// 2 B in out flop for 4 B in bytes written, i.e. a streaming kernel, not a product (ops.rs:254-265).  A workgroup owns 32 rows x 1024 columns:
// a thread keeps W[o][its 4 columns] in registers (<= 16 float4), dZ's 32 rows sit in LDS, and each row is one float4 store (+ one float4
// load of the input where the layer in front is a fused Linear + ReLU: dX * [x > 0], ops.rs:358-369, and the column sums of the masked rows,
// colpart [blockIdx.y][in]: tensor.rs:686-691 for that layer's bias).  r04 ran this product on the 16 x 16 tiles of linear_bwd_small:
// 119 us for 67 MB (0.07 of HBM).
constexpr int THIN_ROWS = 32;
__global__ __launch_bounds__(256) void linear_dx_thin_kernel(const float *__restrict__ dz, const float *__restrict__ w, const float *__restrict__ xmask,
                                                             float *__restrict__ dx, float *__restrict__ colpart, int batch, int in_f, int out_f,
                                                             int accumulate) {
    __shared__ float dzs[THIN_ROWS][16];
    const int t = threadIdx.x, r0 = blockIdx.y * THIN_ROWS, col = (blockIdx.x * 256 + t) * 4;
    for (int i = t; i < THIN_ROWS * 16; i += 256) {
        const int r = i >> 4, o = i & 15;
        dzs[r][o] = (r0 + r < batch && o < out_f) ? dz[(long)(r0 + r) * out_f + o] : 0.f;
    }
    float4 wv[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) wv[o] = (o < out_f && col < in_f) ? *reinterpret_cast<const float4 *>(w + (long)o * in_f + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (col >= in_f) return;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int r = 0; r < THIN_ROWS; ++r) {
        if (r0 + r >= batch) break;
        const long idx = (long)(r0 + r) * in_f + col;
        float4 v = accumulate ? *reinterpret_cast<const float4 *>(dx + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < 16; ++o) {                 // k ascending, like the product it replaces
            const float d = dzs[r][o];
            v.x = fmaf(d, wv[o].x, v.x); v.y = fmaf(d, wv[o].y, v.y); v.z = fmaf(d, wv[o].z, v.z); v.w = fmaf(d, wv[o].w, v.w);
        }
        if (xmask) {
            const float4 m = *reinterpret_cast<const float4 *>(xmask + idx);
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
        *reinterpret_cast<float4 *>(dx + idx) = v;
        cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
    }
    if (colpart) *reinterpret_cast<float4 *>(colpart + (long)blockIdx.y * in_f + col) = cs;
}

// rows of the column-partial matrix th_linear_bwd_adam_ex2 writes for this shape (0: the dX epilogue is not available for it)
static int dx_epilogue_rows(int batch, int in_f, int out_f) {
    if (batch <= 0 || in_f <= 0 || out_f <= 0) return 0;
    if (out_f <= 16 && in_f % 4 == 0 && (long)batch * in_f >= (1L << 20)) return ceil_div(batch, THIN_ROWS);          // the thin kernel
    const bool big = gemm_is_big(batch, in_f, out_f);
    const long wg128 = (long)ceil_div(batch, BM) * ceil_div(in_f, BN) * tile128_kz(batch, in_f, out_f);
    static const long big_min_wg = getenv("TAPER_GEMM_BIG_WG") ? atol(getenv("TAPER_GEMM_BIG_WG")) : 460;
    const bool mid = gemm_is_mid(batch, in_f, out_f) && (!big || wg128 < big_min_wg);
    if (big && !mid && tile128_kz(batch, in_f, out_f) == 1) return ceil_div(batch, BM);                                // sgemm_tile<128>, unsplit
    return 0;
}

}  // namespace th

extern "C" {

int th_linear_bwd_dx_epilogue_rows(int batch, int in_features, int out_features) { return th::dx_epilogue_rows(batch, in_features, out_features); }

int th_linear_bwd_separate_products(int batch, int in_features, int out_features, int with_dx, int with_dw, int dx_epilogue) {
    // (th_linear_bwd_adam_ex2's own decision, below)
    const int ep_rows = (dx_epilogue && with_dx) ? th::dx_epilogue_rows(batch, in_features, out_features) : 0;
    const bool thin_dx = ep_rows > 0 && out_features <= 16;
    const bool dw_big = with_dw && gemm_is_big(out_features, in_features, batch);
    const bool dx_big = with_dx && !thin_dx && gemm_is_big(batch, in_features, out_features);
    return (!dw_big && !dx_big && batch <= 4096 && out_features <= 4096) ? 0 : 1;
}

int th_linear_bwd(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_dy, const float *d_relu_y, float *d_dx,
                  float *d_dw, float *d_db, int batch, int in_features, int out_features, int accumulate_mask) {
    return th_linear_bwd_adam(ctx, d_x, d_w, d_dy, d_relu_y, d_dx, d_dw, d_db, batch, in_features, out_features, accumulate_mask,
                              nullptr, nullptr);
}

int th_linear_bwd_adam(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_dy, const float *d_relu_y, float *d_dx,
                       float *d_dw, float *d_db, int batch, int in_features, int out_features, int accumulate_mask,
                       const th_adam_fuse *w_fuse, const th_adam_fuse *b_fuse) {
    return th_linear_bwd_adam_ex(ctx, d_x, d_w, d_dy, d_relu_y, d_dx, d_dw, d_db, batch, in_features, out_features, accumulate_mask,
                                 w_fuse, b_fuse, nullptr, 0);
}

int th_linear_bwd_adam_ex(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_dy, const float *d_relu_y, float *d_dx,
                          float *d_dw, float *d_db, int batch, int in_features, int out_features, int accumulate_mask,
                          const th_adam_fuse *w_fuse, const th_adam_fuse *b_fuse, const th_adam_slice *extra, int n_extra) {
    return th_linear_bwd_adam_ex2(ctx, d_x, d_w, d_dy, d_relu_y, d_dx, d_dw, d_db, batch, in_features, out_features, accumulate_mask, w_fuse,
                                  b_fuse, extra, n_extra, 0, nullptr);
}

int th_linear_bwd_adam_ex2(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_dy, const float *d_relu_y, float *d_dx,
                           float *d_dw, float *d_db, int batch, int in_features, int out_features, int accumulate_mask,
                           const th_adam_fuse *w_fuse, const th_adam_fuse *b_fuse, const th_adam_slice *extra, int n_extra, int mask_dx,
                           float *d_dx_colpart) {
    TH_REQUIRE(ctx && d_dy, "th_linear_bwd: null argument");
    const int ep_rows = (mask_dx || d_dx_colpart) ? th::dx_epilogue_rows(batch, in_features, out_features) : 0;
    TH_REQUIRE(!(mask_dx || d_dx_colpart) || (d_dx && d_x && ep_rows > 0 && !(accumulate_mask & 1)),
               "th_linear_bwd_adam_ex2: the dX epilogue (mask / column partials) needs d_dx written (not accumulated), d_x, and a shape "
               "th_linear_bwd_dx_epilogue_rows accepts (got %d x %d x %d)", batch, in_features, out_features);
    const bool thin_dx = ep_rows > 0 && out_features <= 16;
    TH_REQUIRE(n_extra >= 0 && n_extra <= TH_MAX_ADAM_SLICES && (n_extra == 0 || extra), "th_linear_bwd_adam_ex: bad extra slices");
    for (int i = 0; i < n_extra; ++i)
        TH_REQUIRE(extra[i].f.d_p != d_w || !d_dx, "th_linear_bwd_adam_ex: a carried slice must not alias the weight this launch reads");
    const AdamDev w_adam = make_adam_dev(d_dw ? w_fuse : nullptr), b_adam = make_adam_dev(d_db ? b_fuse : nullptr);
    TH_REQUIRE(!w_adam.p || !(accumulate_mask & 2), "th_linear_bwd_adam: a fused dW must not accumulate (grad slot must be None)");
    TH_REQUIRE(!b_adam.p || !(accumulate_mask & 4), "th_linear_bwd_adam: a fused db must not accumulate (grad slot must be None)");
    TH_REQUIRE(!d_dx || d_w, "th_linear_bwd: d_w required for d_dx");
    TH_REQUIRE(!d_dw || d_x, "th_linear_bwd: d_x required for d_dw");
    if (batch == 0 || out_features == 0 || in_features == 0) return th_adam_slices(ctx, extra, n_extra);
    if (thin_dx) {
        // the thin layer's dX (+ the ReLU mask and column partials of the layer in front) as a streaming launch; its dW / db below
        const float *dzt = d_dy;
        void *tmpz = nullptr;
        if (d_relu_y) {
            const size_t nz = (size_t)batch * out_features;
            if (th_malloc(ctx, nz * sizeof(float), &tmpz)) return 1;
            if (int rc = th_relu_bwd(ctx, d_relu_y, d_dy, (float *)tmpz, nz, 0)) {
                (void)th_free(ctx, tmpz);       // (error paths give their workspace back: ADVICE r05)
                return rc;
            }
            dzt = (const float *)tmpz;
        }
        hipLaunchKernelGGL(th::linear_dx_thin_kernel, dim3(ceil_div(in_features, 1024), ceil_div(batch, th::THIN_ROWS)), dim3(256), 0, ctx->stream, dzt,
                           d_w, mask_dx ? d_x : nullptr, d_dx, d_dx_colpart, batch, in_features, out_features, 0);
        TH_LAUNCH_CHECK();
        if (tmpz && th_free(ctx, tmpz)) return 1;
        d_dx = nullptr;
    }
    const bool dw_big = d_dw && gemm_is_big(out_features, in_features, batch);
    const bool dx_big = d_dx && gemm_is_big(batch, in_features, out_features);
    // latency-bound shapes (the MNIST MLP / classifier heads): the whole layer backward is ONE launch
    if (!dw_big && !dx_big && batch <= 4096 && out_features <= 4096) {
        LinearBwdArgs q{};
        const int dw_tm = ceil_div(out_features, 16), dw_tn = ceil_div(in_features, 16);
        const int dx_tm = ceil_div(batch, 16), dx_tn = ceil_div(in_features, 16);
        // deep batches: K slices of 256 rows per dW tile (a 16x16 tile over 1024 rows was the step's long pole at
        // batch 1024), partial tiles summed in slice order by splitk_reduce, which also carries the fused Adam update
        int dw_kz = 1;
        if (d_dw && batch >= 512) dw_kz = batch / 256;
        q.dw_kz = dw_kz;
        q.dw_tiles = d_dw ? dw_tm * dw_tn * dw_kz : 0;
        q.n_dw = (q.dw_tiles + 7) & ~7;
        q.dw_tiles_m = dw_tm;
        q.n_dx = d_dx ? dx_tm * dx_tn : 0;
        q.dx_tiles_n = dx_tn;
        // dW = dZ^T . X : op(A)[i=o,k=b] = dY[b*out + o] (rs 1, cs out); op(B)[k=b,j] = X[b*in + j]
        q.dw = SmallArgs{d_dy, d_relu_y, d_x, d_dw, nullptr, out_features, in_features, batch, 1, out_features, in_features, 1,
                         (batch + 15) / 16 * 16, 0, 0, make_ep(1.0f, (accumulate_mask & 2) ? 1.0f : 0.0f)};
        // dX = dZ . W : op(A)[i=b,k=o] = dY[b*out + o] (rs out, cs 1); op(B)[k=o,j] = W[o*in + j]
        q.dx = SmallArgs{d_dy, d_relu_y, d_w, d_dx, nullptr, batch, in_features, out_features, out_features, 1, in_features, 1,
                         (out_features + 15) / 16 * 16, 0, 0, make_ep(1.0f, (accumulate_mask & 1) ? 1.0f : 0.0f)};
        q.dx.a_vec = aligned16(d_dy) && (!d_relu_y || aligned16(d_relu_y)) && (out_features % 4 == 0);
        // The dX workgroups of this launch read W: updating W in the dW epilogue would race with
        // them, so with a dX output the W update runs as a slice kernel behind the launch.
        const bool w_in_kernel = w_adam.p && !d_dx && dw_kz == 1;
        if (w_in_kernel) q.dw.ep.adam = w_adam;
        void *dw_part = nullptr;
        if (dw_kz > 1) {
            if (th_malloc(ctx, (size_t)dw_kz * out_features * in_features * sizeof(float), &dw_part)) return 1;
            q.dw.partial = (float *)dw_part;
            q.dw.kslice = ceil_div(ceil_div(batch, dw_kz), 16) * 16;
        }
        q.db_adam = b_adam;
        q.dy = d_dy;
        q.ymask = d_relu_y;
        q.db = d_db;
        q.batch = batch;
        q.out_f = out_features;
        q.db_accum = (accumulate_mask & 4) ? 1 : 0;
        q.n_db = d_db ? ceil_div(out_features, 64) : 0;
        q.extra = make_adam_slices(extra, n_extra, ctx);
        const int grid = q.n_dw + q.n_dx + q.n_db + q.extra.blocks();
        if (grid == 0) return 0;
        if (d_relu_y) hipLaunchKernelGGL(linear_bwd_small<true>, dim3(grid), dim3(256), 0, ctx->stream, q);
        else hipLaunchKernelGGL(linear_bwd_small<false>, dim3(grid), dim3(256), 0, ctx->stream, q);
        TH_LAUNCH_CHECK();
        if (dw_kz > 1) {
            // this launch is behind the dX workgroups that read W (same stream), so the fused update may ride here
            Epilogue ep = make_ep(1.0f, (accumulate_mask & 2) ? 1.0f : 0.0f);
            ep.adam = w_adam;
            const long mn = (long)out_features * in_features;
            launch_splitk_reduce(ctx->stream, (const float *)dw_part, d_dw, mn, in_features, dw_kz, ep);
            TH_LAUNCH_CHECK();
            return th_free(ctx, dw_part);
        }
        if (w_adam.p && !w_in_kernel) return adam_slice(ctx, w_adam, d_dw, (int64_t)out_features * in_features);
        return 0;
    }
    // large shapes: MFMA tile kernels; the ReLU mask is materialised once
    const float *dz = d_dy;
    void *tmp = nullptr;
    if (d_relu_y) {
        const size_t n = (size_t)batch * out_features;
        if (th_malloc(ctx, n * sizeof(float), &tmp)) return 1;
    }
    // (one exit: a failing launch gives the mask's workspace back -- ADVICE r05)
    const int rc = [&]() -> int {
        if (tmp) {
            if (int rc = th_relu_bwd(ctx, d_relu_y, d_dy, (float *)tmp, (size_t)batch * out_features, 0)) return rc;
            dz = (const float *)tmp;
        }
        if (d_dx) {  // dX[B,in] (+)= dZ[B,out] . W[out,in]      (ops.rs:254-265 through the W^T node)
            Epilogue ep = make_ep(1.0f, (accumulate_mask & 1) ? 1.0f : 0.0f);
            ep.mask = mask_dx ? d_x : nullptr;      // (ep_rows > 0: this product takes sgemm_tile<128> unsplit)
            ep.colpart = d_dx_colpart;
            if (int rc = gemm_dispatch(ctx, 0, 0, batch, in_features, out_features, dz, d_w, d_dx, ep)) return rc;
        }
        if (d_dw) {  // dW[out,in] (+)= dZ^T[out,B] . X[B,in]     (ops.rs:280-291 + tensor.rs:574-587)
            Epilogue ep = make_ep(1.0f, (accumulate_mask & 2) ? 1.0f : 0.0f);
            ep.adam = w_adam;   // the dX product above has already consumed W (same stream): update it with the gradient
            if (int rc = gemm_dispatch(ctx, 1, 0, out_features, in_features, batch, dz, d_x, d_dw, ep)) return rc;
        }
        if (d_db) {  // db[out] (+)= sum_b dZ[b,out]               (tensor.rs:686-691)
            if (int rc = (accumulate_mask & 4) ? th_colsum_accum(ctx, dz, d_db, batch, out_features)
                                               : th_colsum(ctx, dz, d_db, batch, out_features))
                return rc;
        }
        // large shapes: W's update rode in the dW product; the bias update runs as a slice kernel
        if (int rc = adam_slice(ctx, b_adam, d_db, out_features)) return rc;
        return th_adam_slices(ctx, extra, n_extra);
    }();
    if (tmp) {
        const int rf = th_free(ctx, tmp);
        return rc ? rc : rf;
    }
    return rc;
}

}  // extern "C"

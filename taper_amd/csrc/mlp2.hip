// mlp2.hip -- th_mlp2_xent: one training step of a two-layer classifier (Linear + ReLU, Linear, softmax cross-entropy: the MNIST MLP of
// BASELINE configs[1] / [3], examples/train_mnist.rs with one hidden layer) at LARGE batch -- 1 024 .. 60 000 rows -- in THREE launches,
// reading the batch's rows where they lie in the dataset (no gathered copy):
//
//   launch 1  mlp2_rows_kernel<RT>  a workgroup owns RT rows.  H = relu(X W1^T + b1) on v_mfma_f32_32x32x2_f32 with X (rows through the
//             loader's index vector) and W1 streamed global -> LDS by LDS-DMA in 32-deep k chunks, double buffered (nn.rs:54-60,
//             activation.rs:10-12); then, with H in LDS, everything that is ROW-parallel: logits, softmax / NLL / argmax / dlogits
//             (loss.rs:101-195, 271-290), dH = dlogits W2 and the ReLU mask (ops.rs:254-265, 358-369).  Writes the masked dZ1 -- the
//             only activation-sized tensor that leaves the CU -- and the workgroup's partial sums of everything that adds over rows but
//             is small: dW2, db1, db2, NLL, hits.  H itself never reaches memory.  Its first thread opens the optimizer step (t += 1).
//   launch 2  mlp2_dw1_kernel       dW1 = dZ1^T X (ops.rs:266-294): 128 x 128 tiles of the [hidden][in] output, the batch split into K
//             slices over the workgroups; both operands by LDS-DMA as they lie in memory (X rows through the index vector again);
//             every slice writes its raw accumulators.
//   launch 3  mlp2_finish_kernel    adds the K slices of dW1 and the row-block partials of launch 1 in fixed order (deterministic),
//             writes the four gradients, the loss, the hit count and the step log, and applies Adam (optim.rs:99-110) to all four
//             parameters in the same threads -- no launch of the step reads a parameter after this one, so nothing is deferred.
//
// Bounds: launches 1 and 2 are 2 B in hid flop each -- MFMA fp32 peak (157.3 TF); launch 3 moves kz x 4 hid in bytes -- HBM.
// fp32 sums are reordered with respect to the reference's k-ordered chains (k quads permuted inside a 32-chunk, K slices, row blocks):
// within the 1e-4 relative bar, like every GEMM here; index work (argmax, hits) is exact.
#include "tail_dev.h"

TH_USES_DEVICE_ERRORS()

namespace th {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

// where the rows of the batch live: a dense [batch][in] block, or rows idx[(cursor + r) % n_idx] of a resident dataset
struct RowSource {
    const float *x, *labels;
    const int32_t *idx;       // nullable
    const int64_t *cursor;    // nullable (0)
    int64_t n_idx;
    unsigned x_bytes;         // extent of x (buffer range of the LDS-DMA descriptor: reads past it return 0)
};

__device__ __forceinline__ int src_row(const RowSource &s, long cur, int r) {
    if (!s.idx) return r;
    long pos = cur + r;
    if (pos >= s.n_idx) pos %= s.n_idx;
    return s.idx[pos];
}

// LDS-DMA as inline assembly.  Through the builtin (raw_ptr_buffer_load_lds) the compiler knows the instruction writes LDS and -- without alias
// information -- parks an `s_waitcnt vmcnt(0)` in front of the next ds_read: every request in flight is drained before the chunk that IS
// there may be read, and a ring of stages degenerates to one.  As an opaque instruction the fetch stays in flight; its landing is waited for
// explicitly (wait_vmcnt: vector-memory operations complete in order) before the workgroup barrier that publishes the stage.
// (Operations the compiler does not count only make ITS vmcnt waits stricter, never weaker: the counter is in order.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void *p, unsigned bytes) {
    const uint64_t a = (uint64_t)p;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));   // stride 0: raw buffer
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);                              // range: offsets >= bytes read as 0
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ unsigned lds_addr(const float *p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
// lane l: 16 bytes from rs.base + voff + soff -> LDS byte address lds_dst + 16 l  (lds_dst, soff wave-uniform)
__device__ __forceinline__ void lds_dma16(i32x4 rs, unsigned lds_dst, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_dst), "v"(voff), "s"(rs), "s"(soff) : "memory");   // (m0 is reserved: it cannot be declared as clobbered; nothing else in these kernels uses it)
}

// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt = simm16[15:14, 3:0], expcnt [6:4], lgkmcnt [11:8]): "at most N vector-memory operations of
// this wave still in flight" -- they complete in order, so the oldest fetches of a ring of LDS-DMA stages have landed
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

#ifdef TH_PROFILE
__device__ long long g_m2_prof[16];     // wall clock (100 MHz) at stage boundaries of the three launches (tools/prof_mlp2_stages.sh)
#define M2_STAMP(i, cond) do { if (threadIdx.x == 0 && (cond)) g_m2_prof[i] = wall_clock64(); } while (0)
#else
#define M2_STAMP(i, cond) do { } while (0)
#endif

constexpr int M2_LDH __attribute__((unused)) = 132;   // row pitch of the H tile in LDS (hidden <= 128): 16-byte aligned rows, rows 4 apart 16 banks apart
constexpr int M2_BK = 32;
#ifndef TH_M2_PAIRS
#define TH_M2_PAIRS 1      // 0: the 16-row tiles take their chunks one by one (r04; measurement)
#endif

__host__ __device__ constexpr int m2_swz(int row) { return (row >> 1) & 7; }

struct Mlp2RowsArgs {
    RowSource src;
    const float *w1, *b1, *w2, *b2;   // DEEP: w2 / b2 are the SECOND HIDDEN layer's [h2][hid] / [h2], w3 / b3 the classifier's [c][h2] / [c]
    const float *w3, *b3;
    int batch, in_f, hid, c, h2;
    float *dz1;          // [gridDim.x * RT][hid]: rows >= batch are written as zeros
    int32_t *rows_res;   // [gridDim.x * RT]: the dataset row of every batch row (rows >= batch: the last row's), for launch 2 (nullable)
    float *part;         // [gridDim.x][part_stride]: dW2 [c][hid], db1 [hid], db2 [16], nll, hits
                         // DEEP: dW3 [c][h2], db2 [h2], db3 [16], nll, hits (the block the shallow form lays out, for the classifier on h2), then
                         // at o_deep: dW2 [h2][hid], db1 [hid]
    int part_stride, o_deep;
    int32_t *tick;       // nullable
    int ksplit;          // 16-row form only: workgroups per row block, each contracting a share of the k chunks (1: none)
    float *kpart;        // [gridDim.x][2048]: a workgroup's accumulators, in register order
    unsigned *karrive;   // [row blocks]: arrival counters (the context's: zero between launches -- the last arrival resets its own)
};

// ---------------------------------------------------------------------------------------------------------------- launch 1
// LDS images of the k chunk: [rows][32 k], 128-byte rows, the eight 16-byte k quads of row r stored at quad ^ swz(r) (gemm.hip's
// k-contiguous image: one ds_read_b128 gives a lane the operands of four k-steps, conflict-free, and an LDS-DMA lane fills one quad).
// The k chunks travel through a ring of NS stages: chunk it + NS - 1 is requested while chunk it is contracted, so a request has NS - 1
// iterations (~0.5-1 us each) to cross the fabric -- with two stages the loop ran at the memory latency, not at the MFMA rate (r04: 52.7 us
// for 3.3 GFLOP at batch 16 384).
// NW = 4 (default) or 8 waves.  With eight, two waves share a SIMD: the waves are two groups of four column quarters -- on 64-row tiles the
// groups are the two 32-row halves (one accumulator tile per wave); on 32-row tiles both groups hold the same 32 x 32 tile and split every
// chunk's four k rounds between them, two accumulators that are added, group 0 + group 1, when H goes to LDS.  Built to test whether one
// wave per SIMD was what held the k loop back; it was not (see the launcher), the form stays as a measurement knob.
// DEEP: TWO hidden layers (examples/train_mnist.rs:40-48: 784-128-64-10).  The second hidden layer is a hid -> h2 contraction on the H tile
// that is in LDS anyway, and everything behind it is still row-parallel: A2 = relu(A1 W2^T + b2) on v_mfma_f32_16x16x4_f32 (a wave owns a
// 16-column tile of h2, W2's operand registers requested when the k loop ends), the classifier and its backward on A2 exactly as the shallow
// form has them on A1 (dZ2 replaces A2 in place), dA1 = dZ2 W2 and the ReLU mask -> the masked dZ1 as before, and the row block's share of
// dW2 = dZ2^T A1 ([h2][hid]: 32 KB per block at 64 x 128) beside the small sums.  Only dW1 needs the batch-wide launch.  hid, h2 multiples of 16.
template <int RT, int NS, int NW, bool DEEP = false>
__global__ __launch_bounds__(64 * NW, (!DEEP && NW == 4 && (RT == 32 || (RT == 64 && NS <= 3))) ? 2 : 1) void mlp2_rows_kernel(Mlp2RowsArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    constexpr bool R16 = RT == 16;              // 16-row tiles: v_mfma_f32_16x16x4_f32, two 16 x 16 accumulator tiles per wave (its 32 columns)
    static_assert(!R16 || NW == 4, "the 16-row form has four waves");
    constexpr int NI = (NW == 4 && RT >= 32) ? RT / 32 : 1;   // 32-row MFMA tiles per wave
    constexpr bool KSPLIT = NW == 8 && RT == 32;   // the two wave groups split a chunk's k rounds
    constexpr int A_T = RT * M2_BK, B_T = 128 * M2_BK;
    constexpr int NRB = RT / 16;                // 16-row blocks of the tile
    constexpr int STG = A_T + B_T;              // floats per stage (X chunk, then W1 chunk)
    constexpr int NA = (RT / 8 + NW - 1) / NW, NB = 16 / NW;   // LDS-DMA instructions per wave and stage (a wave may own fewer X instructions)
    constexpr int LMIN = (RT / 8) / NW + NB;    // ... of the wave that issues the fewest: what the in-order counter may be allowed to hold
    constexpr int NTT = 8 / NW;                 // 16-column tiles per wave in the dH stage
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lk = lane >> 5, l16 = lane & 15, g4 = lane >> 4;
    const int cq = wave & 3, grp = wave >> 2;   // column quarter; wave group (NW == 8)
    // Few row blocks (batch <= 2 048 on the 16-row tiles: <= 128 of them on 256 CUs): `ksplit` workgroups share a block's k chunks, each
    // leaves its accumulators in memory, and the LAST to arrive (an arrival counter per block) adds them in split order -- a fixed order,
    // whoever is last -- and runs the classifier epilogue; the others are done.  The k loop of a block is the longest dependent chain of
    // the step at these sizes (25 chunks: 9.6 us of 17 at batch 1 024); split four ways it is 7 chunks.
    const int ksplit = R16 ? a.ksplit : 1, blk = R16 ? (int)blockIdx.x / ksplit : (int)blockIdx.x, ks = R16 ? (int)blockIdx.x % ksplit : 0;
    const int r0 = blk * RT, B = a.batch, in_f = a.in_f, hid = a.hid, C = a.c;
    static_assert(!DEEP || NW == 4, "the two-hidden-layer form has four waves");
    // the classifier's weights and the width of its input: the last hidden layer's
    const float *wl = DEEP ? a.w3 : a.w2, *bl = DEEP ? a.b3 : a.b2;
    const int kl = DEEP ? a.h2 : hid;
    M2_STAMP(0, blockIdx.x == 0);
    if (a.tick && blockIdx.x == 0 && t == 0) a.tick[0] += 1;                  // optim.rs:84 (the launch that reads t comes later)
    const long cur = (a.src.idx && a.src.cursor) ? a.src.cursor[0] : 0;

    // ---- staging plans: lane -> (row, k quad) of the 1 KB an LDS-DMA instruction fills; instruction i = NW j + wave covers rows 8 i .. 8 i + 7 ----
    int a_voff[NA], b_voff[NB];
    bool a_on[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int row = 8 * (NW * j + wave) + (lane >> 3);
        a_on[j] = NW * j + wave < RT / 8;
        const int srow = src_row(a.src, cur, min(r0 + min(row, RT - 1), B - 1));   // rows beyond the batch: a copy of its last row, zeroed below
        a_voff[j] = (int)((unsigned)srow * (unsigned)in_f * 4u + (unsigned)(((lane & 7) ^ m2_swz(row)) << 4));
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = 8 * (NW * j + wave) + (lane >> 3);
        b_voff[j] = (int)((unsigned)min(row, hid - 1) * (unsigned)in_f * 4u + (unsigned)(((lane & 7) ^ m2_swz(row)) << 4));
    }
    const i32x4 rs_x = make_rsrc(a.src.x, a.src.x_bytes), rs_w = make_rsrc(a.w1, (unsigned)hid * (unsigned)in_f * 4u);
    const unsigned lds0 = lds_addr(smem);
    // what the epilogue needs from memory is requested now, under the whole k loop
    const int hcol = 32 * cq + li;
    const float bias1 = (a.b1 && hcol < hid) ? a.b1[hcol] : 0.f;
    float bias16[2];                                                          // 16-row form: the bias of this lane's two accumulator columns
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) bias16[tt] = (R16 && a.b1 && 32 * cq + 16 * tt + l16 < hid) ? a.b1[32 * cq + 16 * tt + l16] : 0.f;
    const int hrow = 16 * wave + l16;                                         // the row this lane owns in the classifier stage (wave < NRB)
    const int grow = min(r0 + min(hrow, RT - 1), B - 1);
    const int grow_src = (wave < NRB) ? src_row(a.src, cur, grow) : 0;
    const float tf = (wave < NRB) ? a.src.labels[grow_src] : 0.f;
    // launch 2 gathers the same rows: it reads them here instead of walking cursor -> index again (one dependent memory access less on its way in)
    if (a.rows_res && wave < NRB && g4 == 0 && ks == 0) a.rows_res[r0 + hrow] = grow_src;
    float4 w2a[8];                                                            // logits' A operand: W2[class l16][16 u + 4 g4 ..]
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = 16 * u + 4 * g4;
        const float4 v = *reinterpret_cast<const float4 *>(wl + (long)min(l16, C - 1) * kl + min(k, kl - 4));
        w2a[u] = k < kl ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float b2v[4], w2b[NTT][4];                                                // b2[class 4 g4 + e]; dH's B operand: W2[class 4 g4 + s][col]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int cls = min(4 * g4 + e, C - 1);
        b2v[e] = bl ? bl[cls] : 0.f;
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) w2b[tt][e] = wl[(long)cls * kl + min(16 * NTT * wave + 16 * tt + l16, kl - 1)];
    }

    // this workgroup's chunks: [c_lo, c_hi) of the in_f / 32 full chunks (+ the ragged last one for the last split)
    const int nfull_all = in_f / M2_BK, ktail_all = in_f - nfull_all * M2_BK, n_all = nfull_all + (ktail_all ? 1 : 0);
    const int c_lo = (int)((long)n_all * ks / ksplit), c_hi_all = (int)((long)n_all * (ks + 1) / ksplit);
    const int ktail = (ktail_all && c_hi_all == n_all) ? ktail_all : 0, nfull = c_hi_all - c_lo - (ktail ? 1 : 0), kbase = c_lo * M2_BK;
    // the ragged last chunk goes through registers (zero fill beyond in_f), requested first: it is the oldest load in flight
    float4 ta[NA], tb[NB];
    if (ktail) {
        const int k0 = nfull_all * M2_BK;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int row = 8 * (NW * j + wave) + (lane >> 3), mq = (lane & 7) ^ m2_swz(row);
            const bool in = mq * 4 < ktail;
            const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.src.x) + (unsigned)a_voff[j] - (unsigned)(mq << 4) +
                                                               (unsigned)(k0 + (in ? mq * 4 : 0)) * 4u);
            ta[j] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int row = 8 * (NW * j + wave) + (lane >> 3), mq = (lane & 7) ^ m2_swz(row);
            const bool in = mq * 4 < ktail;
            const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.w1) + (unsigned)b_voff[j] - (unsigned)(mq << 4) +
                                                               (unsigned)(k0 + (in ? mq * 4 : 0)) * 4u);
            tb[j] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto fetch = [&](int it, int stage) {
        const int k0 = kbase + it * M2_BK;
        const unsigned st = lds0 + (unsigned)(stage * STG) * 4u;
#ifdef TH_M2_EXP_NOA      // (measurement only: what the k loop costs without its X / W1 traffic -- results are garbage)
        if (it < NS)
#endif
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (NW * j + wave_u < RT / 8) lds_dma16(rs_x, st + 1024u * (unsigned)(NW * j + wave_u), a_voff[j], k0 * 4);
#ifdef TH_M2_EXP_NOB
        if (it < NS)
#endif
#pragma unroll
        for (int j = 0; j < NB; ++j) lds_dma16(rs_w, st + (unsigned)A_T * 4u + 1024u * (unsigned)(NW * j + wave_u), b_voff[j], k0 * 4);
    };

    floatx16 acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    // 16-row form: acc16[tt][e] = H[row 4 g4 + e][column 32 cq + 16 tt + l16].  A round contracts 16 k: lane group g4 reads k quad 4 rd + g4
    // of its row (X row l16; W1 rows 32 cq + 16 tt + l16) with one ds_read_b128, and k-step e of the round uses component e of every group's
    // quad -- k = 16 rd + 4 g4 + e, every k once, the same permutation on both operands; the swizzled image keeps the reads conflict-free.
    floatx4 acc16[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
    int a16[2], b16[2][2];                      // [round]: float offsets of this lane's quads
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        a16[rd] = l16 * M2_BK + (((4 * rd + g4) ^ m2_swz(l16)) << 2);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int brow = 32 * cq + 16 * tt + l16;
            b16[rd][tt] = brow * M2_BK + (((4 * rd + g4) ^ m2_swz(brow)) << 2);
        }
    }
    // this lane's LDS addresses: row (tile row + li) of X, row (32 cq + li) of W1; k quad 2 r + lk of round r, swizzled
    const int row_tile0 = (NW == 8 && RT == 64) ? grp : 0;
    int ao[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int ar = 32 * (row_tile0 + i) + li;
        ao[i] = ar * M2_BK + ((lk ^ m2_swz(ar)) << 2);
    }
    const int br = 32 * cq + li;
    const int bo = br * M2_BK + ((lk ^ m2_swz(br)) << 2);
    const int r_lo = KSPLIT ? 2 * grp : 0, r_n = KSPLIT ? 2 : 4;              // this wave's k rounds of a chunk

    // eight k per round: lane half lk holds k = 8 r + 4 lk + e of its row; the operands of round r + 1 are requested before the MFMAs of round r
    auto contract = [&](const float *as, const float *bs, int nr) {
        if constexpr (R16) {
            // nr counts 8-k rounds of the other forms: 16-k rounds here = (nr + 1) / 2
            float4 xa[2], wb[2][2];
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                xa[rd] = *reinterpret_cast<const float4 *>(as + a16[rd]);
                wb[rd][0] = *reinterpret_cast<const float4 *>(bs + b16[rd][0]);
                wb[rd][1] = *reinterpret_cast<const float4 *>(bs + b16[rd][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                if (2 * rd < nr) {
#define M2_MFMA16(E)                                                                                         \
    acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[rd].E, wb[rd][0].E, acc16[0], 0, 0, 0);                \
    acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[rd].E, wb[rd][1].E, acc16[1], 0, 0, 0);
                    M2_MFMA16(x) M2_MFMA16(y) M2_MFMA16(z) M2_MFMA16(w)
#undef M2_MFMA16
                }
            }
            return;
        }
        float4 af[2][NI], bf[2];
#define M2_REQ(SET, RR)                                                                       \
    {                                                                                         \
        _Pragma("unroll") for (int i = 0; i < NI; ++i) af[SET][i] = *reinterpret_cast<const float4 *>(as + (ao[i] ^ ((RR) << 3))); \
        bf[SET] = *reinterpret_cast<const float4 *>(bs + (bo ^ ((RR) << 3)));                 \
    }
#define M2_MFMA(CS, E) \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[CS][i].E, bf[CS].E, acc[i], 0, 0, 0);
        M2_REQ(0, r_lo)
#pragma unroll
        for (int q = 0; q < r_n; ++q) {
            const int cs = q & 1, r = r_lo + q;
            if (r < nr) {
                if (q + 1 < r_n) M2_REQ(cs ^ 1, r + 1)
                __builtin_amdgcn_sched_barrier(0);
                M2_MFMA(cs, x)
                M2_MFMA(cs, y)
                M2_MFMA(cs, z)
                M2_MFMA(cs, w)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef M2_REQ
#undef M2_MFMA
    };
    // The 16-row form takes its chunks in PAIRS (r05): one wait + one barrier + one burst of requests per two chunks, and the second chunk's six
    // operand reads are in flight under the first chunk's sixteen MFMAs.  A chunk is only 0.23 us of matrix work for a lone wave per SIMD
    // against ~0.16 us of wait / barrier / LDS-DMA issue / LDS read latency around it -- measured per chunk 0.39 us.  Chunk c lives in stage
    // c % NS (NS = 8): the pair (it, it + 1) is waited for while chunks up to it + 5 are in flight, and the stages of the pair before are
    // refilled with chunks it + 6, it + 7 behind the barrier that retires their readers.
    int it0 = 0;
    if constexpr (R16 && NS == 8 && TH_M2_PAIRS) {
        const int npair = nfull / 2;                        // (an odd count's last chunk is requested in turn and taken alone below)
#pragma unroll
        for (int s = 0; s < NS - 2; ++s)
            if (s < nfull) fetch(s, s);
        float4 xq[2] = {}, wq[2][2] = {};                   // two operand sets: X quad, the two W1 quads
        auto rd16 = [&](float4 &x, float4 (&w)[2], const float *as, int rd) {
            x = *reinterpret_cast<const float4 *>(as + a16[rd]);
            w[0] = *reinterpret_cast<const float4 *>(as + A_T + b16[rd][0]);
            w[1] = *reinterpret_cast<const float4 *>(as + A_T + b16[rd][1]);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mm16 = [&](const float4 &x, const float4 (&w)[2]) {
#define M2_MFMA16(E)                                                                         \
    acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.E, w[0].E, acc16[0], 0, 0, 0);          \
    acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.E, w[1].E, acc16[1], 0, 0, 0);
            M2_MFMA16(x) M2_MFMA16(y) M2_MFMA16(z) M2_MFMA16(w)
#undef M2_MFMA16
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int pr = 0; pr < npair; ++pr) {
            const int it = 2 * pr;
            if (it + NS - 2 <= nfull) wait_vmcnt<(NS - 4) * LMIN>();         // (the four chunks behind the pair may still be in flight)
            else wait_vmcnt<0>();
            lds_barrier();
            if (pr == 0) M2_STAMP(1, blockIdx.x == 0);
            if (it + NS - 2 < nfull) fetch(it + NS - 2, (it + NS - 2) % NS);
            if (it + NS - 1 < nfull) fetch(it + NS - 1, (it + NS - 1) % NS);
            // Operand reads run ONE 16-k ROUND AHEAD of the MFMAs, across the barrier too: a round is three ds_read_b128 per lane (12 KB over the
            // four waves: ~100 LDS clocks + latency) against eight MFMAs (256 clocks), and with one wave per SIMD nothing else covers a read
            // that is waited for -- reads-then-MFMAs per pair measured 0.30 us per chunk with the memory traffic switched OFF (0.21 is the
            // matrix pipe alone).  The last round of a pair sits in registers over the next barrier (lds_barrier waits for LDS reads).
            const float *as0 = smem + (it % NS) * STG, *as1 = smem + ((it + 1) % NS) * STG;
            rd16(xq[0], wq[0], as0, 0);
            if (pr > 0) mm16(xq[1], wq[1]);
            rd16(xq[1], wq[1], as0, 1);
            mm16(xq[0], wq[0]);
            rd16(xq[0], wq[0], as1, 0);
            mm16(xq[1], wq[1]);
            rd16(xq[1], wq[1], as1, 1);
            mm16(xq[0], wq[0]);
        }
        if (npair > 0) mm16(xq[1], wq[1]);
        it0 = 2 * npair;
    } else {
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < nfull) fetch(s, s);
    }
    int stage = it0 % NS;
    for (int it = it0; it < nfull; ++it) {
        // chunk `it` has landed once only the NS - 2 fetches issued after it are still in flight (fewer were issued near the end: wait for all)
        if (it + NS - 1 <= nfull && !(R16 && NS == 8 && TH_M2_PAIRS)) wait_vmcnt<(NS - 2) * LMIN>();
        else wait_vmcnt<0>();
        lds_barrier();                                      // ... in every wave; and every wave is done with the stage refilled next
        if (it == 0) M2_STAMP(1, blockIdx.x == 0);
        const int nxt = it + NS - 1;
        if (nxt < nfull && !(R16 && NS == 8 && TH_M2_PAIRS)) fetch(nxt, stage == 0 ? NS - 1 : stage - 1);
        contract(smem + stage * STG, smem + stage * STG + A_T, 4);
        stage = stage + 1 == NS ? 0 : stage + 1;
    }
    if (ktail) {
        lds_barrier();
#pragma unroll
        for (int j = 0; j < NA; ++j)
            if (NW * j + wave < RT / 8) *reinterpret_cast<float4 *>(smem + 256 * (NW * j + wave) + 4 * lane) = ta[j];
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<float4 *>(smem + A_T + 256 * (NW * j + wave) + 4 * lane) = tb[j];
        lds_barrier();
        contract(smem, smem + A_T, (ktail + 7) / 8);
    }
    lds_barrier();

    M2_STAMP(2, blockIdx.x == 0);
    if constexpr (R16) {
        if (ksplit > 1) {
            __shared__ int last_arrival;
            // The partial sums travel through MEMORY, not through this XCD's L2 (the splits of a block sit on different XCDs): system-coherent
            // stores, waited for, then one arrival; the last arrival reads them with system-coherent loads.  (A device-scope __threadfence()
            // instead costs a write-back of the whole L2 per workgroup on this part: measured 57 us for this launch against 16 unsplit.)
            float *mine = a.kpart + (long)blockIdx.x * 2048 + (wave * 128 + lane) * 4;
            // Both stores AND their wait in ONE statement: the compiler does not know an asm statement is a vector-memory store, so its
            // hazard recognizer does not keep the next instruction from rewriting the 128-bit store data -- r04's first form (one statement
            // per store) was compiled to `global_store_dwordx4 v[8:9], v[4:7]` / `s_mov_b64` / `v_accvgpr_read_b32 v7, a3` ...: the second
            // store's data went into the registers the first store was still reading.  With the GPU to itself the store had read them in
            // time, every time (360 runs); with another process's traffic backing up the store path, 5 - 8 of 30 captured runs had a wrong
            // step.  Here the data registers are inputs of the statement that also waits for the stores: nothing can touch them
            // before the stores are done.
            asm volatile("global_store_dwordx4 %0, %2, off sc0 sc1\n\tglobal_store_dwordx4 %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                         ::"v"(mine), "v"(mine + 256), "v"(acc16[0]), "v"(acc16[1]) : "memory");   // ... acknowledged by memory ...
            __syncthreads();                                   // ... every thread's, before the one arrival below
            if (t == 0) last_arrival = __hip_atomic_fetch_add(&a.karrive[blk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ksplit - 1);
            __syncthreads();
            if (!last_arrival) return;                         // (the whole workgroup: uniform)
            const float *all = a.kpart + (long)blk * ksplit * 2048 + (wave * 128 + lane) * 4;
            floatx4 pv[8][2];                                  // all requests out before the first value is used; added in SPLIT ORDER
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    if (q < ksplit) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(pv[q][tt]) : "v"(all + q * 2048 + tt * 256) : "memory");
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    if (q < ksplit) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv[q][tt]) : : "memory");
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                floatx4 v = pv[0][tt];
#pragma unroll
                for (int q = 1; q < 8; ++q)
                    if (q < ksplit) v += pv[q][tt];
                acc16[tt] = v;
            }
            if (t == 0) __hip_atomic_store(&a.karrive[blk], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch (it starts after this one has ended)
        }
    }
    // ---- epilogue: the staging buffers are dead; H, dlogits and two scalars per wave live in their place ----
    float *Hs = smem;                           // [RT][M2_LDH]
    float *H2s = Hs + RT * M2_LDH;              // DEEP: [RT][M2_LDH]: the second hidden layer's activations, later dZ2 in their place
    float *Hc = DEEP ? H2s : Hs;                // the classifier's input tile
    float *D3S = Hc + RT * M2_LDH;              // [RT][20]: dlogits [row][class], zero for classes >= C and rows >= batch
    float *sc = D3S + RT * 20;                  // [4][2]: the waves' NLL / hit sums
    static_assert(!DEEP || (RT * (2 * M2_LDH + 20) + 8) <= NS * STG, "the two-hidden-layer epilogue lives in the ring's space");
    // DEEP: the second hidden layer's operand registers, requested now (they cross the fabric while H goes to LDS).  Wave w owns the
    // 16-column tiles ct = w, w + 4 of h2: B(k, n = l16) = W2[16 ct + l16][k], k = 16 u + 4 g4 + e as component e of quad u (the k order
    // the A operand's float4 reads of the H tile give: every k once, the same permutation on both operands)
    float4 w2f[DEEP ? 2 : 1][8];
    float b2c[2] = {0.f, 0.f};
    if constexpr (DEEP) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ct = wave + 4 * j;
            const bool tile = 16 * ct < a.h2;
            b2c[j] = (tile && a.b2 && 16 * ct + l16 < a.h2) ? a.b2[16 * ct + l16] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = 16 * u + 4 * g4;
                const float4 v = *reinterpret_cast<const float4 *>(a.w2 + (long)min(16 * ct + l16, a.h2 - 1) * hid + min(k, hid - 4));
                w2f[j][u] = (tile && k < hid) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    if (KSPLIT) {                               // the second group's half of the k sum goes through H's own cells: group 0 + group 1
        if (grp == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) Hs[((e & 3) + 8 * (e >> 2) + 4 * lk) * M2_LDH + hcol] = acc[0][e];
        }
        lds_barrier();
    }
    if constexpr (R16) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {           // C/D map of 16x16x4: row 4 g4 + e, column l16
                const int col = 32 * cq + 16 * tt + l16;
                float v = acc16[tt][e] + bias16[tt];
                v = (v > 0.f && col < hid) ? v : 0.f;
                Hs[(4 * g4 + e) * M2_LDH + col] = v;
            }
    } else if (!KSPLIT || grp == 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * (row_tile0 + i) + (e & 3) + 8 * (e >> 2) + 4 * lk;     // C/D map of 32x32x2
                float v = acc[i][e];
                if (KSPLIT) v = v + Hs[row * M2_LDH + hcol];
                v += bias1;
                v = (v > 0.f && hcol < hid) ? v : 0.f;                     // activation.rs:10-12; columns beyond hid hold zeros
                Hs[row * M2_LDH + hcol] = v;
            }
    }
    lds_barrier();
    float w2g[DEEP ? 2 : 1][32];                // DEEP: dA1's B operand, W2[k][16 (2 w + tt) + l16], k = 16 u + 4 g4 + e at [4 u + e]
    if constexpr (DEEP) {
        // A2 = relu(A1 W2^T + b2) (nn.rs:54-60, activation.rs:10-12): D[row 4 g4 + e][column l16] of tile ct, four accumulation chains
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ct = wave + 4 * j;
            const bool tile = 16 * ct < a.h2;          // (wave-uniform; tiles beyond h2 hold zeros: the classifier reads whole quads)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                floatx4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
                if (tile) {
                    const float *ap = Hs + (16 * rb + l16) * M2_LDH + 4 * g4;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float4 h = *reinterpret_cast<const float4 *>(ap + 16 * u);
                        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(h.x, w2f[j][u].x, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(h.y, w2f[j][u].y, c1, 0, 0, 0);
                        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(h.z, w2f[j][u].z, c2, 0, 0, 0);
                        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(h.w, w2f[j][u].w, c3, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = ((c0[e] + c1[e]) + (c2[e] + c3[e])) + b2c[j];
                    H2s[(16 * rb + 4 * g4 + e) * M2_LDH + 16 * ct + l16] = (tile && 16 * ct + l16 < a.h2 && v > 0.f) ? v : 0.f;   // (a ragged last tile: zeros beyond h2)
                }
            }
        }
        // dA1's operand registers: requested here, used behind the classifier
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 16 * u + 4 * g4 + e;
                    const float v = a.w2[(long)min(k, a.h2 - 1) * hid + min(16 * (2 * wave + tt) + l16, hid - 1)];
                    w2g[tt][4 * u + e] = k < a.h2 ? v : 0.f;
                }
        lds_barrier();
    }
    // logits^T[class 4 g4 + e][row l16] = W2 . H^T + b2, the row's softmax cross-entropy (wave w: rows 16 w ..)
    if (wave < NRB) {
        const float *hp = Hc + hrow * M2_LDH + 4 * g4;
        floatx4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 h = *reinterpret_cast<const float4 *>(hp + 16 * u);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[u].x, h.x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[u].y, h.y, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[u].z, h.z, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[u].w, h.w, c3, 0, 0, 0);
        }
        float lg[4], dl[4], nll_row;
        int bi;
#pragma unroll
        for (int e = 0; e < 4; ++e) lg[e] = 4 * g4 + e < C ? ((c0[e] + c1[e]) + (c2[e] + c3[e])) + b2v[e] : -INFINITY;
        tail_row_softmax(lg, g4, C, tf, 1.0f / (float)B, dl, nll_row, bi);
        const bool real = r0 + hrow < B;
        *reinterpret_cast<float4 *>(D3S + hrow * 20 + 4 * g4) = real ? make_float4(dl[0], dl[1], dl[2], dl[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        float ns = real ? nll_row : 0.f, hs = (real && fabsf((float)bi - tf) < 1e-6f) ? 1.f : 0.f;   // loss.rs:283
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {                               // a fixed tree over the 16 rows (lanes 0..15)
            ns += __shfl_down(ns, off, 16);
            hs += __shfl_down(hs, off, 16);
        }
        if (lane == 0) {
            sc[2 * wave] = ns;
            sc[2 * wave + 1] = hs;
        }
    }
    lds_barrier();
    M2_STAMP(3, blockIdx.x == 0);
    float *part = a.part + (long)blk * a.part_stride;
    const int o_db1 = C * kl, o_db2 = o_db1 + kl, o_nll = o_db2 + 16;
    // wave w owns 16 NTT hidden columns of ALL the tile's rows: dZ1 = (dlogits W2) * [H > 0] (ops.rs:254-265, 358-369), its
    // column sums (db1, tensor.rs:686-691) and the tile's share of dW2 = dlogits^T H (ops.rs:266-294).  (DEEP: the same on the second
    // hidden layer -- dZ2, db2, dW3 -- with dZ2 written over A2 in LDS instead of to memory: the element's own lane is its last reader.)
    if (16 * NTT * wave < kl) {
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const int col = 16 * NTT * wave + 16 * tt + l16;
            if (16 * (NTT * wave + tt) >= kl) break;                          // (kl is a multiple of 4, not of 32: a wave's second tile may lie beyond it,
            const bool live = col < kl;                                       //  or end inside it: columns >= kl hold H = 0, so dz = 0, and are not stored)
            float colsum = 0.f;
            floatx4 dw2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const float4 d = *reinterpret_cast<const float4 *>(D3S + (16 * rb + l16) * 20 + 4 * g4);   // A: dl[row l16][class 4 g4 + s]
                floatx4 dh = {0.f, 0.f, 0.f, 0.f};
                dh = __builtin_amdgcn_mfma_f32_16x16x4f32(d.x, w2b[tt][0], dh, 0, 0, 0);
                dh = __builtin_amdgcn_mfma_f32_16x16x4f32(d.y, w2b[tt][1], dh, 0, 0, 0);
                dh = __builtin_amdgcn_mfma_f32_16x16x4f32(d.z, w2b[tt][2], dh, 0, 0, 0);
                dh = __builtin_amdgcn_mfma_f32_16x16x4f32(d.w, w2b[tt][3], dh, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = 16 * rb + 4 * g4 + e;                       // dh[e] = dH[row][col]
                    const float hv = Hc[row * M2_LDH + col];
                    const float dz = hv > 0.f ? dh[e] : 0.f;
                    if constexpr (DEEP) Hc[row * M2_LDH + col] = dz;
                    else if (live) a.dz1[(long)(r0 + row) * hid + col] = dz;    // (zeros for rows >= batch: launch 2 contracts whole 32-row chunks)
                    colsum += dz;
                    // dW2's operands of k-step e: A(class l16, row) and B(row, col) -- the row of this very element
                    dw2 = __builtin_amdgcn_mfma_f32_16x16x4f32(D3S[row * 20 + l16], hv, dw2, 0, 0, 0);
                }
            }
            colsum += __shfl_xor(colsum, 16, 64);
            colsum += __shfl_xor(colsum, 32, 64);
            if (lane < 16 && live) part[o_db1 + col] = colsum;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * g4 + e < C && live) part[(4 * g4 + e) * kl + col] = dw2[e];   // dw2[e] = dW2[class 4 g4 + e][col]
        }
    }
    if constexpr (DEEP) {
        lds_barrier();                                                        // dZ2 is complete (in A2's place)
        const int h2 = a.h2;
        float *pd = part + a.o_deep;                                          // dW2 [h2][hid], then db1 [hid]
        // dA1 = dZ2 W2 (ops.rs:254-265), dZ1 = dA1 * [A1 > 0] (ops.rs:358-369) -> memory, its column sums (db1): wave w owns the
        // 16-column tiles 2 w, 2 w + 1 of hid
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int col = 16 * (2 * wave + tt) + l16;
            if (16 * (2 * wave + tt) >= hid) break;                           // (wave-uniform)
            float colsum = 0.f;
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const float *zp = H2s + (16 * rb + l16) * M2_LDH + 4 * g4;
                floatx4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (16 * u >= h2) break;
                    const float4 z = *reinterpret_cast<const float4 *>(zp + 16 * u);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(z.x, w2g[tt][4 * u], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(z.y, w2g[tt][4 * u + 1], c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(z.z, w2g[tt][4 * u + 2], c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(z.w, w2g[tt][4 * u + 3], c3, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = 16 * rb + 4 * g4 + e;
                    const float dz = Hs[row * M2_LDH + col] > 0.f ? ((c0[e] + c1[e]) + (c2[e] + c3[e])) : 0.f;
                    if (col < hid) a.dz1[(long)(r0 + row) * hid + col] = dz;    // (zeros for rows >= batch: their dlogits are zero; a ragged last tile: A1 = 0 beyond hid)
                    colsum += dz;
                }
            }
            colsum += __shfl_xor(colsum, 16, 64);
            colsum += __shfl_xor(colsum, 32, 64);
            if (lane < 16 && col < hid) pd[h2 * hid + col] = colsum;
        }
        // the row block's share of dW2 = dZ2^T A1 (ops.rs:266-294 through the W^T node): D[j = 16 tj + 4 g4 + e][n = 16 tn + l16], the
        // contraction over the tile's rows -- A(m = j, k = row) = dZ2[row][j], B(k = row, n) = A1[row][n]; wave w owns tn = 2 w, 2 w + 1
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int tn = 2 * wave + tt;
            if (16 * tn >= hid) break;
#pragma unroll
            for (int tj0 = 0; tj0 < 8; tj0 += 4) {                             // four h2 tiles at a time (16 accumulator registers)
                if (16 * tj0 >= h2) break;
                floatx4 dw[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) dw[q] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                for (int s4 = 0; s4 < RT / 4; ++s4) {
                    const int row = 4 * s4 + g4;
                    const float bv = Hs[row * M2_LDH + 16 * tn + l16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float av = H2s[row * M2_LDH + 16 * (tj0 + q) + l16];      // (tiles beyond h2 hold zeros)
                        dw[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, dw[q], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int j = 16 * (tj0 + q) + 4 * g4 + e;
                        if (j < h2 && 16 * tn + l16 < hid) pd[(long)j * hid + 16 * tn + l16] = dw[q][e];
                    }
            }
        }
    }
    if (t < 16) {                                                             // db2 (tensor.rs:686-691): rows in order
        float s = 0.f;
#pragma unroll 8
        for (int r = 0; r < RT; ++r) s += D3S[r * 20 + t];
        part[o_db2 + t] = s;
    }
    if (t == 0) {
        float ns = 0.f, hs = 0.f;
#pragma unroll
        for (int w = 0; w < NRB && w < 4; ++w) {
            ns += sc[2 * w];
            hs += sc[2 * w + 1];
        }
        part[o_nll] = ns;
        part[o_nll + 1] = hs;
    }
    M2_STAMP(4, blockIdx.x == 0);
    M2_STAMP(5, blockIdx.x == gridDim.x - 1);
#endif
}

// ---------------------------------------------------------------------------------------------------------------- launch 2
struct Mlp2DwArgs {
    RowSource src;
    const float *dz1;     // [rows_pad][hid]
    unsigned dz_bytes;
    int rows_pad, batch, in_f, hid;
    float *partial;       // [kz][hid][in_f]
    int tiles_n, kz, kslice;   // kslice a multiple of 32
};

// dW1[m = hidden][n = in] over rows [z kslice, (z + 1) kslice): A(m, k) = dZ1[k][m], B(k, n) = X[row k][n] -- both m / n-contiguous, staged as
// they lie in memory ([32 k][128] images, gemm.hip's m/n-contiguous form: ds_read_b32, lanes on consecutive m / n).  The last n tile reads
// beyond a row's end (into the next row; beyond the buffer: zeros): those columns are never stored.
template <int NS, int WGS, bool INDEXED>
__global__ __launch_bounds__(256, WGS) void mlp2_dw1_kernel(Mlp2DwArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TS = 128, T_T = TS * M2_BK, STG = 2 * T_T, L = 8;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 31, lk = lane >> 5;
    // block b runs on XCD b % 8: every XCD takes a contiguous run of (K slice, n tile) pairs, n innermost -- the tiles of a slice share its dZ1 rows
    const int nwg = a.tiles_n * a.kz, bid = blockIdx.x;
    const int xcd = bid % kNumXCD, q = nwg / kNumXCD, rmd = nwg % kNumXCD;
    const int w = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + bid / kNumXCD;
    const int z = w / a.tiles_n, tn = w % a.tiles_n;
    const int n0 = tn * TS, in_f = a.in_f, hid = a.hid;
    const int kbeg = z * a.kslice, kend = min(a.rows_pad, kbeg + a.kslice), nt = (kend - kbeg) / M2_BK;
    // The dataset rows of this K slice, resolved once into LDS (behind the ring): row r of the batch is idx[(cursor + r) % n].  The cursor was
    // advanced by the PREVIOUS step's finish launch; positions are cursor % n + r < 2 n (batch <= n, host-checked).  In the loop a lane then
    // picks its rows' entries with ds_read -- a scalar or vector load there would sit in the same in-order / unordered counters as the ring's
    // LDS-DMA and the operand reads, and waiting for it would drain them.
    int *rows_l = reinterpret_cast<int *>(smem + NS * STG);
    {
        const int n_idx = INDEXED ? (int)a.src.n_idx : 1;
        const int cur = (INDEXED && a.src.cursor) ? (int)(sload(a.src.cursor) % a.src.n_idx) : 0;
        for (int i = t; i < nt * M2_BK; i += 256) {
            const int row = min(kbeg + i, a.batch - 1);
            const int p = cur + row;
            rows_l[i] = INDEXED ? a.src.idx[p >= n_idx ? p - n_idx : p] : row;
        }
    }
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    // staging: instruction j of wave w fills 16-byte units 64 (4 j + w) .. + 63 of the image = k rows 2 (4 j + w) and + 1, 32 quads each.
    // The two rows' dataset indices are wave-uniform: scalar loads, a chunk ahead of the fetch that needs them.
    int a_voff[4];
    const int quad = lane & 31, upper = lane >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) a_voff[j] = (int)((unsigned)(2 * (4 * j + wave) + upper) * (unsigned)hid * 4u + (unsigned)(quad << 4));
    const i32x4 rs_a = make_rsrc(a.dz1, a.dz_bytes), rs_x = make_rsrc(a.src.x, a.src.x_bytes);
    const unsigned lds0 = lds_addr(smem);
    const unsigned xq = (unsigned)(n0 + quad * 4) * 4u;
    auto fetch = [&](int it, int stage) {
        const int k0 = kbeg + it * M2_BK;
        const unsigned st = lds0 + (unsigned)(stage * STG) * 4u;
        int srow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) srow[j] = rows_l[it * M2_BK + 2 * (4 * j + wave) + upper];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lds_dma16(rs_a, st + 1024u * (unsigned)(4 * j + wave), a_voff[j], (int)((unsigned)k0 * (unsigned)hid * 4u));
            lds_dma16(rs_x, st + (unsigned)T_T * 4u + 1024u * (unsigned)(4 * j + wave), (int)((unsigned)srow[j] * (unsigned)in_f * 4u + xq), 0);
        }
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int ao[2], bo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ao[i] = 4 * lk * TS + wm + 32 * i + li;
        bo[i] = 4 * lk * TS + wn + 32 * i + li;
    }
    M2_STAMP(6, blockIdx.x == 0);
    __syncthreads();                              // rows_l is complete
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nt) fetch(s, s);
    int stage = 0;
    for (int it = 0; it < nt; ++it) {
        if (it + NS - 1 <= nt) wait_vmcnt<(NS - 2) * L>();
        else wait_vmcnt<0>();
        lds_barrier();
        const int nxt = it + NS - 1;
        if (nxt < nt) fetch(nxt, stage == 0 ? NS - 1 : stage - 1);
        const float *as = smem + stage * STG, *bs = as + T_T;
        float4 af[2][2], bf[2][2];
#define M2_FRAG(S, O, RR) make_float4((S)[(O) + (8 * (RR)) * TS], (S)[(O) + (8 * (RR) + 1) * TS], (S)[(O) + (8 * (RR) + 2) * TS], (S)[(O) + (8 * (RR) + 3) * TS])
#define M2_REQ(SET, RR)                                  \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {      \
        af[SET][i] = M2_FRAG(as, ao[i], RR);             \
        bf[SET][i] = M2_FRAG(bs, bo[i], RR);             \
    }
#define M2_MFMA(CS, E)                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)        \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[CS][i].E, bf[CS][j].E, acc[i][j], 0, 0, 0);
        M2_REQ(0, 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cs = r & 1;
            if (r + 1 < 4) { M2_REQ(cs ^ 1, r + 1) }
            __builtin_amdgcn_sched_barrier(0);
            M2_MFMA(cs, x)
            M2_MFMA(cs, y)
            M2_MFMA(cs, z)
            M2_MFMA(cs, w)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef M2_MFMA
#undef M2_REQ
#undef M2_FRAG
        stage = stage + 1 == NS ? 0 : stage + 1;
    }
    M2_STAMP(7, blockIdx.x == 0);
    float *out = a.partial + (long)z * hid * in_f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + 32 * j + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (row < hid && col < in_f) out[(long)row * in_f + col] = acc[i][j][e];
            }
        }
    M2_STAMP(8, blockIdx.x == 0);
    M2_STAMP(9, blockIdx.x == gridDim.x - 1);
#endif
}

// ---- launch 2, second form (default): ONE workgroup of eight waves per CU on a 128 x 112 tile -----------------------------------------
// 784 = 7 x 112: no padded columns (the 128-wide tiles above compute 896).  The K slices hand out whole 32-row chunks, the first
// n_chunks % kz slices one more than the rest, and 7 kz <= 256: every CU holds exactly one workgroup, all of (nearly) equal length -- the
// 2 x 448 layout above leaves a quarter of the CUs with one workgroup and the rest with two.  Half as many slices: half the partial sums
// written here and read by launch 3.  Eight waves keep two waves on every SIMD (one wave alone stalls its matrix pipe at every operand wait),
// and one workgroup per CU leaves LDS for a four-stage ring (4 x 30 KB).
//   wave w: rows m = 16 w .. 16 w + 15 of the tile, all 112 columns: seven v_mfma_f32_16x16x4_f32 accumulators.
//   A image [32 k][128 m], m quad q of row k stored at q ^ ((k & 3) << 2): lane (r, g) of k-step s reads A[k = 4 s + g][m = m0 + r] -- the four
//   g groups land on four different 16-bank groups.  B image [32 k][112 n] as it lies in memory: rows 4 apart are 448 floats = 0 mod 64
//   banks + {0, 48, 32, 16} for g = 0..3: conflict-free too.
struct Mlp2Dw8Args {
    RowSource src;
    const float *dz1;     // [rows_pad][hid]
    unsigned dz_bytes;
    int rows_pad, batch, in_f, hid;
    float *partial;       // [kz][hid][in_f]
    int tiles_n, kz;      // tiles of 112 columns
    const int32_t *rows_res;   // [rows_pad]: launch 1's resolved rows (nullable: resolve through src)
};

template <int NS, bool INDEXED>
__global__ __launch_bounds__(512, 1) void mlp2_dw1_kernel8(Mlp2Dw8Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TN = 112, A_T = 128 * M2_BK, B_T = TN * M2_BK, STG = A_T + B_T;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l16 = lane & 15, g4 = lane >> 4;
    const int nwg = a.tiles_n * a.kz, bid = blockIdx.x;
    const int xcd = bid % kNumXCD, q = nwg / kNumXCD, rmd = nwg % kNumXCD;
    const int w = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + bid / kNumXCD;
    const int z = w / a.tiles_n, tn = w % a.tiles_n;
    const int n0 = tn * TN, in_f = a.in_f, hid = a.hid;
    // whole chunks per slice: the first `extra` slices take one more
    const int n_chunks = a.rows_pad / M2_BK, base = n_chunks / a.kz, extra = n_chunks % a.kz;
    const int nt = base + (z < extra ? 1 : 0), kbeg = (z * base + min(z, extra)) * M2_BK;
    int *rows_l = reinterpret_cast<int *>(smem + NS * STG);
    {
        if (INDEXED && a.rows_res) {
            for (int i = t; i < nt * M2_BK; i += 512) rows_l[i] = a.rows_res[kbeg + i];
        } else {
            const int n_idx = INDEXED ? (int)a.src.n_idx : 1;
            const int cur = (INDEXED && a.src.cursor) ? (int)(sload(a.src.cursor) % a.src.n_idx) : 0;
            for (int i = t; i < nt * M2_BK; i += 512) {
                const int row = min(kbeg + i, a.batch - 1);
                const int p = cur + row;
                rows_l[i] = INDEXED ? a.src.idx[p >= n_idx ? p - n_idx : p] : row;
            }
        }
    }
    // staging plans.  A: 1024 16-byte units, unit u = 64 (2 wave + j) + lane: k row u >> 5, LDS quad u & 31 holds memory quad (u & 31) ^ ((krow & 3) << 2).
    // B: 896 units, unit u = 64 i + lane (instruction i = wave, wave + 8 < 14): k row u / 28, quad u % 28.
    int a_voff[2], b_krow[2];
    unsigned b_q[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int u = 64 * (2 * wave + j) + lane, krow = u >> 5, mq = (u & 31) ^ ((krow & 3) << 2);
        a_voff[j] = (int)((unsigned)krow * (unsigned)hid * 4u + (unsigned)(mq << 4));
        const int ub = 64 * (wave + 8 * j) + lane;
        b_krow[j] = ub / 28;
        b_q[j] = (unsigned)(n0 + (ub % 28) * 4) * 4u;
    }
    const bool b_second = wave + 8 < 14;
    const i32x4 rs_a = make_rsrc(a.dz1, a.dz_bytes), rs_x = make_rsrc(a.src.x, a.src.x_bytes);
    const unsigned lds0 = lds_addr(smem);
    auto fetch = [&](int it, int stage) {
        const int k0 = kbeg + it * M2_BK;
        const unsigned st = lds0 + (unsigned)(stage * STG) * 4u;
        const int s0 = rows_l[it * M2_BK + b_krow[0]], s1 = b_second ? rows_l[it * M2_BK + b_krow[1]] : 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) lds_dma16(rs_a, st + 1024u * (unsigned)(2 * wave + j), a_voff[j], (int)((unsigned)k0 * (unsigned)hid * 4u));
        lds_dma16(rs_x, st + (unsigned)A_T * 4u + 1024u * (unsigned)wave, (int)((unsigned)s0 * (unsigned)in_f * 4u + b_q[0]), 0);
        if (b_second) lds_dma16(rs_x, st + (unsigned)A_T * 4u + 1024u * (unsigned)(wave + 8), (int)((unsigned)s1 * (unsigned)in_f * 4u + b_q[1]), 0);
    };
    floatx4 acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    // operand addresses of k-step s: A[(4 s + g4) * 128 + ((16 wave + l16) ^ (g4 << 4))], B[(4 s + g4) * 112 + 16 i + l16]
    const int ao = g4 * 128 + ((16 * wave + l16) ^ (g4 << 4)), bo = g4 * TN + l16;
    M2_STAMP(6, blockIdx.x == 0);
    __syncthreads();                              // rows_l is complete
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nt) fetch(s, s);
    int stage = 0;
    for (int it = 0; it < nt; ++it) {
        // per wave and chunk: 4 LDS-DMA instructions (3 for waves 6, 7): "at most (NS - 2) x 4 in flight" covers both
        if (it + NS - 1 <= nt) wait_vmcnt<(NS - 2) * 3>();
        else wait_vmcnt<0>();
        lds_barrier();
        const int nxt = it + NS - 1;
        const float *as = smem + stage * STG + ao, *bs = smem + stage * STG + A_T + bo;
        float av[2], bv[2][7];
#define M2_REQ8(SET, S)                                                       \
    {                                                                         \
        av[SET] = as[(S) * 4 * 128];                                          \
        _Pragma("unroll") for (int i = 0; i < 7; ++i) bv[SET][i] = bs[(S) * 4 * TN + 16 * i]; \
    }
        // Eight operand reads per k-step of seven MFMAs: each sits BEHIND an MFMA (its issue is then hidden by the MFMA in flight -- as a group
        // in front of the step's MFMAs the reads cost the matrix pipe their issue time, experiments/mfma_rate.hip), and the next stage's
        // LDS-DMA requests sit behind the first step's MFMAs.
        M2_REQ8(0, 0)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int cs = s & 1;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cs], bv[cs][i], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < 8) {
                    if (i == 0) av[cs ^ 1] = as[(s + 1) * 4 * 128];
                    bv[cs ^ 1][i] = bs[(s + 1) * 4 * TN + 16 * i];
                }
                if (s == 0 && i == 1 && nxt < nt) fetch(nxt, stage == 0 ? NS - 1 : stage - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef M2_REQ8
        stage = stage + 1 == NS ? 0 : stage + 1;
    }
    M2_STAMP(7, blockIdx.x == 0);
    // C/D map of 16x16x4: acc[i][e] = dW1[m = 16 wave + 4 g4 + e][n = n0 + 16 i + l16]
    float *out = a.partial + (long)z * hid * in_f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int col = n0 + 16 * i + l16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 16 * wave + 4 * g4 + e;
            if (row < hid && col < in_f) out[(long)row * in_f + col] = acc[i][e];
        }
    }
    M2_STAMP(8, blockIdx.x == 0);
    M2_STAMP(9, blockIdx.x == gridDim.x - 1);
#endif
}

// ---------------------------------------------------------------------------------------------------------------- launch 3
struct Mlp2Seg {           // a run of a row block's partial sums that is one parameter's gradient
    int start, len;        // elements [start, start + len) of the block's record (start a multiple of 4)
    float *grad;           // nullable: not wanted
    AdamDev ad;
};
constexpr int M2_MAX_SEGS = 5;

struct Mlp2FinishArgs {
    const float *partial;   // [kz][hid * in_f]
    const float *part;      // [n_blk][part_stride]
    int kz, n_blk, part_stride, batch, in_f, hid;
    float *dw1, *loss, *ncorrect, *metrics;
    int64_t capacity;
    int64_t *state;
    int64_t advance;
    AdamDev w1a;
    int w1_blocks;
    Mlp2Seg seg[M2_MAX_SEGS];   // shallow: dW2, db1, db2; two hidden layers: dW3, db2, db3, dW2, db1
    int n_seg, o_nll, part_len; // {nll, hits} at o_nll; part_len: the record's live length
};

__device__ __forceinline__ void m2_apply(const AdamDev &ad, long i, float g) {
    if (ad.p) adam_update(ad.p, ad.m, ad.v, i, g, adam_dev_step(ad), ad.beta1, ad.beta2, ad.eps, ad.wd);
}

// Two roles.  dW1: a workgroup owns 16 float4 of the gradient, 16 threads per float4 add the K slices z = zg, zg + 16, ... and the first
// adds the 16 sums in order.  The row blocks' partial sums: a workgroup owns 16 of the elements, 16 threads per element add the
// blocks b = sub, sub + 16, ..., the first adds the 16 sums in order.  (One thread per element adding 64 slices / 256 blocks one after
// the other is a chain of dependent round trips: 62.6 us at batch 16 384 for 26 MB.)
__global__ __launch_bounds__(256) void mlp2_finish_kernel(Mlp2FinishArgs a) {
    __shared__ float4 sh4[16][16];
    const int bid = blockIdx.x, t = threadIdx.x;
    M2_STAMP(10, blockIdx.x == 0);
    M2_STAMP(11, blockIdx.x == gridDim.x - 1);
    if (bid < a.w1_blocks) {
        // 16 float4 of the gradient per workgroup, 16 threads per float4: thread zg adds slices zg, zg + 16, ...; thread 0 adds the 16 sums in order
        const long mn = (long)a.hid * a.in_f, i0 = ((long)bid * 16 + (t & 15)) * 4;
        const int zg = t >> 4;
        // the owner's Adam operands are requested FIRST: behind the sums they would be a second dependent round trip
        const bool owner = zg == 0 && i0 < mn, fuse = owner && a.w1a.p != nullptr;
        float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), mv = pv, vv = pv;
        float step = 0.f;
        if (fuse) {
            pv = *reinterpret_cast<const float4 *>(a.w1a.p + i0);
            mv = *reinterpret_cast<const float4 *>(a.w1a.m + i0);
            vv = *reinterpret_cast<const float4 *>(a.w1a.v + i0);
            step = adam_dev_step(a.w1a);
        }
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i0 < mn) {
#pragma unroll 4
            for (int z = zg; z < a.kz; z += 16) {
                const float4 v = *reinterpret_cast<const float4 *>(a.partial + (long)z * mn + i0);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        sh4[zg][t & 15] = s;
        __syncthreads();
        if (zg != 0 || i0 >= mn) return;
#pragma unroll
        for (int u = 1; u < 16; ++u) {
            const float4 p = sh4[u][t];
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
        *reinterpret_cast<float4 *>(a.dw1 + i0) = s;
        if (fuse) {      // optim.rs:99-110, element by element as adam_update does
            const AdamDev &ad = a.w1a;
            float4 po, mo, vo;
#define M2_ADAM(k)                                                       \
            {                                                            \
                const float gg = s.k + ad.wd * pv.k;                     \
                mo.k = ad.beta1 * mv.k + (1.0f - ad.beta1) * gg;         \
                vo.k = ad.beta2 * vv.k + (1.0f - ad.beta2) * gg * gg;    \
                po.k = pv.k - step * mo.k / (sqrtf(vo.k) + ad.eps);      \
            }
            M2_ADAM(x) M2_ADAM(y) M2_ADAM(z) M2_ADAM(w)
#undef M2_ADAM
            *reinterpret_cast<float4 *>(ad.m + i0) = mo;
            *reinterpret_cast<float4 *>(ad.v + i0) = vo;
            *reinterpret_cast<float4 *>(ad.p + i0) = po;
        }
        return;
    }
    // the row blocks' partial sums: a WAVE per float4 of the record (four per workgroup) -- lane `sub` adds the blocks sub, sub + 64, ... in
    // order (four loads at 256 blocks, all in flight), the 64 sums meet in a fixed shuffle tree: no LDS, no barrier.  (r04: 16 threads per
    // element, 16 dependent scalar loads each and one thread's chain of Adam updates behind them -- this role, not the 14 MB of K slices,
    // was the launch's critical path: 8.7 - 9.5 us then, 6.3 - 6.6 us with float4 loads and one element per thread, measured.)
    const int qd = (bid - a.w1_blocks) * 4 + (t >> 6), sub = t & 63;
    if (4 * qd >= a.part_len) return;                  // (the whole wave)
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const float *src = a.part + 4 * qd;
#pragma unroll 4
        for (int b = sub; b < a.n_blk; b += 64) {
            const float4 v = *reinterpret_cast<const float4 *>(src + (long)b * a.part_stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s.x += __shfl_down(s.x, off, 64);
        s.y += __shfl_down(s.y, off, 64);
        s.z += __shfl_down(s.z, off, 64);
        s.w += __shfl_down(s.w, off, 64);
    }
    // one element per lane from here (lane j < 4 takes component j): the Adam updates of a float4's four elements are four chains of
    // dependent loads, side by side instead of one after the other
    const float q0 = __shfl(s.x, 0, 64), q1 = __shfl(s.y, 0, 64), q2 = __shfl(s.z, 0, 64), q3 = __shfl(s.w, 0, 64);
    if (sub >= 4) return;
    if (4 * qd == a.o_nll) {
        if (sub != 0) return;
        const float l = q0 / (float)a.batch;          // loss.rs:164
        a.loss[0] = l;
        if (a.ncorrect) a.ncorrect[0] = q1;
        if (a.metrics) {                              // th_log_step
            const int64_t s0 = a.state[0], s1 = a.state[1];
            const int64_t slot = s0 < a.capacity ? s0 : s0 % a.capacity;
            a.metrics[2 * slot] = l;
            a.metrics[2 * slot + 1] = q1;
            a.state[0] = s0 + 1;
            a.state[1] = s1 + a.advance;
        }
        return;
    }
    const float v = sub == 0 ? q0 : (sub == 1 ? q1 : (sub == 2 ? q2 : q3));
    const int e = 4 * qd + sub;
#pragma unroll
    for (int g = 0; g < M2_MAX_SEGS; ++g) {
        if (g < a.n_seg && e >= a.seg[g].start && e < a.seg[g].start + a.seg[g].len && a.seg[g].grad) {
            a.seg[g].grad[e - a.seg[g].start] = v;
            m2_apply(a.seg[g].ad, e - a.seg[g].start, v);
        }
    }
}

thread_local long t_mlp2_calls = 0;
thread_local int t_mlp2_only = 0;      // th_debug_mlp2_only: 0 = the step; 1 / 2 / 3 = that launch alone (per-launch timing; workspace kept from call to call)
thread_local int t_mlp2_ksplit = 0;    // th_debug_mlp2_ksplit: 1 .. 8 = that k split on the 16-row tiles whatever the cap says (tests/test_gpu_repro.py); 0 = default

static int m2_rows_per_block(int batch) {
    static const int forced = [] { const char *e = getenv("TAPER_MLP2_RT"); return e ? atoi(e) : 0; }();
    if (forced == 16 || forced == 32 || forced == 64) return forced;
    // 64-row tiles once they fill the chip (>= 192 workgroups); 32-row tiles down to 128 workgroups; 16-row tiles below (256 workgroups at 4 096
    // rows: with 32-row tiles half the CUs sat idle while the others ran a 23 us k loop)
    return batch >= 12288 ? 64 : (batch > 4096 ? 32 : 16);
}

// One layer of the classifier as the runner sees it
struct M2Layer {
    const float *w, *b;
    float *dw, *db;
    const th_adam_fuse *wf, *bf;
    int out;
};

// the step for n_hidden = 1 (Linear + ReLU, Linear) or 2 (Linear + ReLU, Linear + ReLU, Linear) hidden layers; arguments checked by the callers
static int mlp2_run(th_ctx *ctx, const th_row_source *src, int batch, int in_features, int n_hidden, const M2Layer *L, float *d_loss,
                    float *d_ncorrect, float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance, int32_t *d_tick) {
    const bool deep = n_hidden == 2;
    const int hidden = L[0].out, h2 = deep ? L[1].out : 0, classes = L[n_hidden].out, kl = deep ? h2 : hidden;
    const float *d_w1 = L[0].w, *d_b1 = L[0].b;
    float *d_dw1 = L[0].dw;
    const th_adam_fuse *w1_fuse = L[0].wf;
    const int RT = m2_rows_per_block(batch);
    // (16-row tiles: an even number of row blocks, so that launch 2 sees whole 32-row chunks; a block past the batch writes zeros)
    const int n_blk = RT == 16 ? 2 * ceil_div(batch, 32) : ceil_div(batch, RT), rows_pad = n_blk * RT;
    // a row block's record: the classifier's dW [classes][kl], the bias gradient of the layer in front of it [kl], the classifier's db [16],
    // {nll, hits}; with two hidden layers then dW2 [h2][hidden], db1 [hidden] at o_deep
    const int o_nll = classes * kl + kl + 16, o_deep = (o_nll + 2 + 3) & ~3;
    const int part_len = deep ? o_deep + h2 * hidden + hidden : o_nll + 2, stride = (part_len + 3) & ~3;
    // launch 2's form: 8 (default) = one eight-wave workgroup per CU on 128 x 112 tiles; TAPER_MLP2_DW = 22 | 31 | 32 | 41 = the four-wave
    // 128 x 128 form with that many ring stages x workgroups per CU (measurement knob)
    static const int variant = [] { const char *e = getenv("TAPER_MLP2_DW"); return e ? atoi(e) : 8; }();
    const bool dw8 = variant == 8;
    const int tiles_n = ceil_div(in_features, dw8 ? 112 : 128);
    static const int kz_forced = [] { const char *e = getenv("TAPER_MLP2_KZ"); return e ? atoi(e) : 0; }();
    int kz, kslice = 0;
    if (dw8) {
        // as many slices as fit one workgroup per CU, at least one 32-row chunk each
        kz = kz_forced > 0 ? kz_forced : kNumCU / tiles_n;
        // at least two 32-row chunks per slice unless the knob says otherwise: one-chunk slices (batch 1 024: 32 of them) write twice the
        // partial sums for launch 3 to read and gain nothing -- measured 33.3 -> 31.2 us per step at 1 024 rows, 36.0 -> 35.4 at 2 048
        kz = std::max(1, std::min(kz, kz_forced > 0 ? rows_pad / M2_BK : std::max(1, rows_pad / (2 * M2_BK))));
    } else {
        // two workgroups per CU (two 64 KB double buffers), slices of at least 256 rows
        const int dw_wgs = variant % 10 == 1 ? 1 : 2;
        kz = kz_forced > 0 ? kz_forced : ceil_div(dw_wgs * kNumCU, tiles_n);
        const int kz_max = rows_pad / 256 > 0 ? rows_pad / 256 : 1;
        if (kz > kz_max) kz = kz_max;
        kslice = ceil_div(ceil_div(rows_pad, kz), M2_BK) * M2_BK;
        kz = ceil_div(rows_pad, kslice);
    }
    // launch 1 on the 16-row tiles with few row blocks: as many workgroups per block as fill the CUs (up to 8) share its k chunks
    // (mlp2_rows_kernel): 4 at batch 1 024, 2 at 2 048, none from 4 096 on.  Measured through the C ABI: 31.1 -> 27.3 us per step at 1 024
    // rows (launch 1: 15.6 -> 11.5 us), 35.5 -> 33.1 at 2 048.  TAPER_MLP2_KSPLIT=1 turns it off (measurement knob).
    static const int ksplit_env = [] { const char *e = getenv("TAPER_MLP2_KSPLIT"); return e ? atoi(e) : 0; }();
    const int ksplit_forced = t_mlp2_ksplit ? t_mlp2_ksplit : ksplit_env;
    int ksplit = RT != 16 ? 1 : std::max(1, std::min(8, kNumCU / n_blk));
    if (RT == 16 && ksplit_forced >= 1 && ksplit_forced <= 8 && n_blk <= 512) ksplit = ksplit_forced;   // (the kernel sums up to 8 splits)
    // th_mlp2_set_max_ksplit / TAPER_MLP2_KSPLIT_MAX cap the split (default 8 = the kernel's limit; 1 = off).  (The hand-off's stores are one
    // asm statement with their wait -- mlp2_rows_kernel: as two statements the compiler reused the first store's data registers while it
    // was still reading them, and with other processes on the GPU 5 - 8 of 30 captured runs had a wrong step: tools/dp512_flake_probe.py,
    // tools/mlp2_repro_stress.py)
    static const int cap_env = [] { const char *e = getenv("TAPER_MLP2_KSPLIT_MAX"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 8 ? v : 0; }();
    if (!(RT == 16 && ksplit_forced >= 1 && ksplit_forced <= 8)) ksplit = std::max(1, std::min(ksplit, cap_env ? cap_env : ctx->m2_max_ksplit));   // (TAPER_MLP2_KSPLIT forces a split past the cap: the parity tests)
    const size_t n_dz = (size_t)rows_pad * hidden, n_part = (size_t)n_blk * stride, n_partial = (size_t)kz * hidden * in_features;
    const size_t n_kpart = ksplit > 1 ? (size_t)n_blk * ksplit * 2048 : 0;
    void *ws = nullptr;
    if (th_malloc(ctx, (n_dz + n_part + n_partial + n_kpart + (size_t)rows_pad) * sizeof(float), &ws)) return 1;
    float *dz1 = (float *)ws, *part = dz1 + n_dz, *partial = part + n_part, *kpart = partial + n_partial;
    // (t_mlp2_only: a timing hook that runs one launch alone -- launch 2 then resolves its rows itself)
    int32_t *rows_res = (dw8 && src->d_indices && t_mlp2_only == 0) ? reinterpret_cast<int32_t *>(kpart + n_kpart) : nullptr;

    RowSource rs{src->d_rows, src->d_labels, src->d_indices, src->d_indices ? src->d_cursor : nullptr, src->n_indices,
                 (unsigned)((size_t)src->n_rows * in_features * 4)};
    Mlp2RowsArgs r{};
    r.src = rs;
    r.w1 = d_w1; r.b1 = d_b1; r.w2 = L[1].w; r.b2 = L[1].b;
    r.w3 = deep ? L[2].w : nullptr; r.b3 = deep ? L[2].b : nullptr;
    r.batch = batch; r.in_f = in_features; r.hid = hidden; r.c = classes; r.h2 = h2;
    r.dz1 = dz1; r.part = part; r.part_stride = stride; r.o_deep = o_deep; r.tick = d_tick;
    r.ksplit = ksplit; r.kpart = kpart;
    r.karrive = ctx->m2_arrive;
    r.rows_res = rows_res;
    // ring depth of launch 1: four stages (six for the 32-row tiles, one workgroup per CU, measured no faster: a lone wave per SIMD is bound by
    // its own issue order, not by the requests in flight)
#define M2_ROWS_LAUNCH_(RT_, NS_, NW_, DEEP_)                                                                                      \
    do {                                                                                                                           \
        const size_t lds = (size_t)NS_ * (RT_ + 128) * M2_BK * sizeof(float);                                                      \
        TH_SET_MAX_LDS(ctx, (mlp2_rows_kernel<RT_, NS_, NW_, DEEP_>), lds);                                                        \
        hipLaunchKernelGGL((mlp2_rows_kernel<RT_, NS_, NW_, DEEP_>), dim3(n_blk * (RT_ == 16 ? ksplit : 1)), dim3(64 * NW_), lds, ctx->stream, r); \
    } while (0)
#define M2_ROWS_LAUNCH(RT_, NS_, NW_) do { if (deep) M2_ROWS_LAUNCH_(RT_, NS_, 4, true); else M2_ROWS_LAUNCH_(RT_, NS_, NW_, false); } while (0)
    // waves per workgroup of launch 1: four.  TAPER_MLP2_NW=8 (two waves per SIMD on the same ring; on 32-row tiles the two wave groups split
    // every chunk's k rounds) measured the same to 2 %: 39.7 / 23.7 / 22.7 us against 39.2-40.5 / 23.5 / 22.5 us at 16 384 / 4 096 / 1 024 rows --
    // the k loop already runs at the rate the matrix pipes sustain at the clocks the part holds under this load
    static const int nw = [] { const char *e = getenv("TAPER_MLP2_NW"); return e && atoi(e) == 8 ? 8 : 4; }();
    // 64-row tiles: a THREE-stage ring (72 KB) so that two workgroups share a CU once there are more workgroups than CUs -- one's first-chunk
    // latency and classifier epilogue (8.4 us of a workgroup's 38) run under the other's k loop: launch 1 at 32 768 rows 73.6 -> 60.8 us (0.69 of
    // the matrix peak), at 60 000 rows 142.7 -> 116.4 us; unchanged up to 16 384 rows (<= 256 workgroups).  TAPER_MLP2_NS64=4: r04's first form.
    static const int ns64 = [] { const char *e = getenv("TAPER_MLP2_NS64"); return e ? atoi(e) : 3; }();
    const int only = t_mlp2_only;
    if (only == 0 || only == 1) {
        if (RT == 16) M2_ROWS_LAUNCH(16, 8, 4);          // (an eight-stage ring: a chunk is 0.25 us of MFMA work, a request ~2 us away)
        else if (RT == 64 && nw == 8) M2_ROWS_LAUNCH(64, 4, 8);
        else if (RT == 64 && ns64 == 3) M2_ROWS_LAUNCH(64, 3, 4);
        else if (RT == 64) M2_ROWS_LAUNCH(64, 4, 4);
        else if (nw == 8) M2_ROWS_LAUNCH(32, 4, 8);
        else M2_ROWS_LAUNCH(32, 4, 4);
    }
#undef M2_ROWS_LAUNCH
#undef M2_ROWS_LAUNCH_
    TH_LAUNCH_CHECK();

    if (only != 0 && only != 2) {
    } else if (dw8) {
        Mlp2Dw8Args d{};
        d.src = rs;
        d.dz1 = dz1; d.dz_bytes = (unsigned)(n_dz * 4);
        d.rows_pad = rows_pad; d.batch = batch; d.in_f = in_features; d.hid = hidden;
        d.partial = partial; d.tiles_n = tiles_n; d.kz = kz; d.rows_res = rows_res;
        constexpr int NS8 = 4;
        const int max_chunks = ceil_div(rows_pad / M2_BK, kz);
        const size_t lds = (size_t)NS8 * (128 + 112) * M2_BK * sizeof(float) + (size_t)max_chunks * M2_BK * sizeof(int);
        TH_REQUIRE(lds <= (160u << 10), "th_mlp2_xent: %d rows per K slice do not fit the index table in LDS", max_chunks * M2_BK);
        TH_SET_MAX_LDS(ctx, (mlp2_dw1_kernel8<NS8, true>), 160 << 10);
        TH_SET_MAX_LDS(ctx, (mlp2_dw1_kernel8<NS8, false>), 160 << 10);
        if (rs.idx) hipLaunchKernelGGL((mlp2_dw1_kernel8<NS8, true>), dim3(tiles_n * kz), dim3(512), lds, ctx->stream, d);
        else hipLaunchKernelGGL((mlp2_dw1_kernel8<NS8, false>), dim3(tiles_n * kz), dim3(512), lds, ctx->stream, d);
    } else {
        Mlp2DwArgs d{};
        d.src = rs;
        d.dz1 = dz1; d.dz_bytes = (unsigned)(n_dz * 4);
        d.rows_pad = rows_pad; d.batch = batch; d.in_f = in_features; d.hid = hidden;
        d.partial = partial; d.tiles_n = tiles_n; d.kz = kz; d.kslice = kslice;
        const int ns = variant / 10 >= 2 && variant / 10 <= 4 ? variant / 10 : 2;
        const size_t lds = (size_t)ns * 2 * 128 * M2_BK * sizeof(float) + (size_t)kslice * sizeof(int);
#define M2_DW_LAUNCH(NS_, WGS_)                                                                                                       \
    do {                                                                                                                              \
        TH_SET_MAX_LDS(ctx, (mlp2_dw1_kernel<NS_, WGS_, true>), 160 << 10);                                                          \
        TH_SET_MAX_LDS(ctx, (mlp2_dw1_kernel<NS_, WGS_, false>), 160 << 10);                                                         \
        if (rs.idx) hipLaunchKernelGGL((mlp2_dw1_kernel<NS_, WGS_, true>), dim3(tiles_n * kz), dim3(256), lds, ctx->stream, d);       \
        else hipLaunchKernelGGL((mlp2_dw1_kernel<NS_, WGS_, false>), dim3(tiles_n * kz), dim3(256), lds, ctx->stream, d);             \
    } while (0)
        if (variant == 31) M2_DW_LAUNCH(3, 1);
        else if (variant == 32) M2_DW_LAUNCH(3, 2);
        else if (variant == 41) M2_DW_LAUNCH(4, 1);
        else M2_DW_LAUNCH(2, 2);
#undef M2_DW_LAUNCH
    }
    TH_LAUNCH_CHECK();

    Mlp2FinishArgs f{};
    f.partial = partial; f.part = part; f.kz = kz; f.n_blk = n_blk; f.part_stride = stride; f.batch = batch; f.in_f = in_features;
    f.hid = hidden;
    f.dw1 = d_dw1; f.loss = d_loss; f.ncorrect = d_ncorrect; f.metrics = d_metrics;
    f.capacity = metrics_capacity; f.state = d_state; f.advance = advance;
    f.w1a = make_adam_dev(w1_fuse);
    f.w1_blocks = ceil_div((long)hidden * in_features, 64);
    f.o_nll = o_nll; f.part_len = part_len;
    {
        const M2Layer &last = L[n_hidden], &prev = L[n_hidden - 1];
        int g = 0;
        f.seg[g++] = Mlp2Seg{0, classes * kl, last.dw, make_adam_dev(last.wf)};                              // the classifier's dW
        f.seg[g++] = Mlp2Seg{classes * kl, kl, prev.db, make_adam_dev(prev.bf)};                             // the bias in front of it
        f.seg[g++] = Mlp2Seg{classes * kl + kl, classes, last.db, make_adam_dev(last.bf)};                   // the classifier's db
        if (deep) {
            f.seg[g++] = Mlp2Seg{o_deep, h2 * hidden, L[1].dw, make_adam_dev(L[1].wf)};                      // dW2
            f.seg[g++] = Mlp2Seg{o_deep + h2 * hidden, hidden, L[0].db, make_adam_dev(L[0].bf)};             // db1
        }
        f.n_seg = g;
    }
    const int tail_blocks = ceil_div(ceil_div(part_len, 4), 4);
    if (only == 0 || only == 3) hipLaunchKernelGGL(mlp2_finish_kernel, dim3(f.w1_blocks + tail_blocks), dim3(256), 0, ctx->stream, f);
    TH_LAUNCH_CHECK();
    ++t_mlp2_calls;
    return th_free(ctx, ws);
}

}  // namespace th

using namespace th;

extern "C" {

int th_mlp2_xent_supported(int batch, int in_features, int hidden, int classes, int64_t n_rows) {
    return batch >= 32 && in_features >= 32 && in_features % 4 == 0 && hidden >= 4 && hidden <= 128 && hidden % 4 == 0 && classes >= 1 &&
                   classes <= 16 && n_rows >= 1 && (double)n_rows * in_features * 4.0 < 2147483648.0 && (double)(batch + 64) * hidden * 4.0 < 2147483648.0
               ? 1 : 0;
}

int th_mlp2_xent_deep_supported(int batch, int in_features, int h1, int h2, int classes, int64_t n_rows) {
    return th_mlp2_xent_supported(batch, in_features, h1, classes, n_rows) && h2 >= 4 && h2 <= 128 && h2 % 4 == 0 ? 1 : 0;
}

#define M2_COMMON_CHECKS(NAME)                                                                                                                                  \
    TH_REQUIRE(!src->d_indices || (src->n_indices >= batch && src->n_indices < (1LL << 30)), NAME ": the index vector must hold at least one batch (and fewer than 2^30 entries)"); \
    TH_REQUIRE(src->d_indices || src->n_rows >= batch, NAME ": a dense row block must hold the batch");                                                        \
    TH_REQUIRE(!d_metrics || (d_state && metrics_capacity > 0), NAME ": metrics need d_state and a capacity")

int th_mlp2_xent(th_ctx *ctx, const th_row_source *src, int batch, int in_features, int hidden, int classes, const float *d_w1,
                 const float *d_b1, const float *d_w2, const float *d_b2, float *d_dw1, float *d_db1, float *d_dw2, float *d_db2,
                 float *d_loss, float *d_ncorrect, float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance,
                 int32_t *d_tick, const th_adam_fuse *w1_fuse, const th_adam_fuse *b1_fuse, const th_adam_fuse *w2_fuse,
                 const th_adam_fuse *b2_fuse) {
    TH_REQUIRE(ctx && src && src->d_rows && src->d_labels && d_w1 && d_w2 && d_dw1 && d_dw2 && d_loss, "th_mlp2_xent: null argument");
    TH_REQUIRE(th_mlp2_xent_supported(batch, in_features, hidden, classes, src->n_rows),
               "th_mlp2_xent: needs batch >= 32, in_features a multiple of 4 (>= 32), hidden a multiple of 4 up to 128, classes <= 16, "
               "rows * in_features * 4 < 2^31 (got %d, %d, %d, %d, %ld rows)", batch, in_features, hidden, classes, (long)src->n_rows);
    M2_COMMON_CHECKS("th_mlp2_xent");
    TH_REQUIRE((((uintptr_t)src->d_rows | (uintptr_t)d_w1 | (uintptr_t)d_w2 | (uintptr_t)d_dw1) & 15) == 0, "th_mlp2_xent: rows, W1, W2 and dW1 must be 16-byte aligned");
    TH_REQUIRE(!(b1_fuse && b1_fuse->d_p) || d_db1, "th_mlp2_xent: a fused bias update needs d_db1");
    TH_REQUIRE(!(b2_fuse && b2_fuse->d_p) || d_db2, "th_mlp2_xent: a fused bias update needs d_db2");
    TH_REQUIRE(!(w1_fuse && w1_fuse->d_p) || ((((uintptr_t)w1_fuse->d_p | (uintptr_t)w1_fuse->d_m | (uintptr_t)w1_fuse->d_v) & 15) == 0),
               "th_mlp2_xent: W1's p / m / v slices must be 16-byte aligned");
    const M2Layer L[2] = {{d_w1, d_b1, d_dw1, d_db1, w1_fuse, b1_fuse, hidden}, {d_w2, d_b2, d_dw2, d_db2, w2_fuse, b2_fuse, classes}};
    return mlp2_run(ctx, src, batch, in_features, 1, L, d_loss, d_ncorrect, d_metrics, metrics_capacity, d_state, advance, d_tick);
}

int th_mlp2_xent_deep(th_ctx *ctx, const th_row_source *src, int batch, int in_features, const th_mlp3_layer *layers, float *d_loss,
                      float *d_ncorrect, float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance, int32_t *d_tick) {
    TH_REQUIRE(ctx && src && src->d_rows && src->d_labels && layers && d_loss, "th_mlp2_xent_deep: null argument");
    M2Layer L[3];
    for (int l = 0; l < 3; ++l) {
        const th_mlp3_layer &y = layers[l];
        TH_REQUIRE(y.d_w && y.d_dw && (((uintptr_t)y.d_w | (uintptr_t)y.d_dw) & 15) == 0, "th_mlp2_xent_deep: layer %d needs 16-byte aligned d_w and d_dw", l);
        TH_REQUIRE(!(y.b_fuse && y.b_fuse->d_p) || y.d_db, "th_mlp2_xent_deep: a fused bias update needs d_db (layer %d)", l);
        L[l] = M2Layer{y.d_w, y.d_b, y.d_dw, y.d_db, y.w_fuse, y.b_fuse, y.out_features};
    }
    TH_REQUIRE(th_mlp2_xent_deep_supported(batch, in_features, L[0].out, L[1].out, L[2].out, src->n_rows),
               "th_mlp2_xent_deep: needs batch >= 32, in_features a multiple of 4 (>= 32), both hidden sizes multiples of 4 "
               "(<= 128), classes <= 16, rows * in_features * 4 < 2^31 (got %d, %d, %d, %d, %d, %ld rows)", batch, in_features, L[0].out,
               L[1].out, L[2].out, (long)src->n_rows);
    M2_COMMON_CHECKS("th_mlp2_xent_deep");
    TH_REQUIRE(((uintptr_t)src->d_rows & 15) == 0, "th_mlp2_xent_deep: the rows must be 16-byte aligned");
    TH_REQUIRE(!(L[0].wf && L[0].wf->d_p) || ((((uintptr_t)L[0].wf->d_p | (uintptr_t)L[0].wf->d_m | (uintptr_t)L[0].wf->d_v) & 15) == 0),
               "th_mlp2_xent_deep: W1's p / m / v slices must be 16-byte aligned");
    return mlp2_run(ctx, src, batch, in_features, 2, L, d_loss, d_ncorrect, d_metrics, metrics_capacity, d_state, advance, d_tick);
}
#undef M2_COMMON_CHECKS

int th_mlp2_set_max_ksplit(th_ctx *ctx, int max_ksplit) {
    TH_REQUIRE(ctx && max_ksplit >= 1 && max_ksplit <= 8, "th_mlp2_set_max_ksplit: 1 .. 8");
    ctx->m2_max_ksplit = max_ksplit;
    return 0;
}

int th_debug_mlp2_calls(int64_t *out) {
    if (out) *out = t_mlp2_calls;
    return 0;
}

int th_debug_mlp2_ksplit(int ksplit) {
    t_mlp2_ksplit = (ksplit >= 1 && ksplit <= 8) ? ksplit : 0;
    return 0;
}

int th_debug_mlp2_only(int which) {
    t_mlp2_only = (which >= 1 && which <= 3) ? which : 0;
    return 0;
}

#ifdef TH_PROFILE
int th_debug_mlp2_prof(th_ctx *ctx, long long *h_out16) {
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(th::g_m2_prof), 16 * sizeof(long long)));
    return 0;
}
#endif

}  // extern "C"

// conv_chain.hip -- the convolutional front of the two MNIST CNNs (examples/train_mnist_cnn.rs:35-100: Conv2dReLU / MaxPool2d /
// AdaptiveAvgPool2d rows of a Sequential, nn.rs:433-490, 622-686) as ONE launch: a workgroup owns ONE IMAGE and walks it through every
// stage with the activations never leaving the CU's LDS.
//
// Why: at batch 256 every 3x3 layer is ~14-28 us of matrix-core issue time, and a launch per layer pays, on top of that, its own prologue
// (staging plans, first stage), epilogue (bias / ReLU / pool, the store burst), launch + drain (~3 us) and the HBM round trip of the map:
// 7-12 us per layer of a 171 us step, five times.  One image of every layer fits the 160 KB LDS in the padded [channel][rows + 2][cols + 2]
// form the matrix-core k loop reads (28x28x32: 117 KB; 14x14x64: 70 KB; 7x7x64: 29 KB), 256 images fill the 256 CUs, and with the input
// resident the k loop needs no staging, no barrier per pass and no LDS-DMA at all: the weight operand of a k-step is ONE dword per lane,
// loaded from global memory (L2-resident, the same slab for every workgroup) a whole 8-channel pass ahead.
//
// Arithmetic = the layer-by-layer kernels (conv_mfma.hip's geometry instances, conv_pool.hip's conv1 kernels): the same k order
// (pass of 8 channels; k-step s = tap s % 9 of channels 4 (s / 9) + lane group), bias after the sum, ReLU, strict-> pooling maxima,
// 16-lane plane sums with the same shuffle tree -- the chain's outputs are bit-identical to the layered path's.
#include "tail_dev.h"

TH_USES_DEVICE_ERRORS()

namespace th {

int mlp3_grads_launch(th_ctx *ctx, const float *const dz[3], const float *const act[3], float *const dw[3], float *const db[3],
                      const int out_f[3], const int in_f[3], const th_adam_fuse *const wf[3], const th_adam_fuse *const bf[3], int batch,
                      const float *part, int n_blk, float *loss, float *ncorrect, float *metrics, int64_t capacity, int64_t *state,
                      int64_t advance, const float *gap_dx, const th_mlp3_gap *gap);   // gemm.hip

constexpr int CH_NT = 512;                                                      // threads per workgroup: 8 waves, two per SIMD
__host__ __device__ constexpr int ch_cis(int wp) { return wp * wp + ((16 - (wp * wp) % 32) + 32) % 32; }   // channel stride = 16 mod 32: the
                                                                                // two lane groups of a ds_read_b32 half land 16 banks apart
__host__ __device__ constexpr int ch_tile_ld(int px) { return px + ((4 - px % 8) + 8) % 8; }               // tile pitch = 4 mod 8 (even: b64 reads)

#ifdef TH_PROFILE
__device__ long long g_chain_prof[32];    // wall clock (100 MHz) at phase boundaries of workgroup 100, then clock64 around conv2's k loop
#define CH_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 100) g_chain_prof[i] = wall_clock64(); } while (0)
#define CH_CLK(i) do { if (threadIdx.x == 0 && blockIdx.x == 100) g_chain_prof[i] = clock64(); } while (0)
__device__ long long g_chain_span[2 * 256];   // every workgroup's first and last wall clock of the layer kernel (tools/prof_conv_layer.py: how a launch's workgroups are spread in time)
#define CH_SPAN(k) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_chain_span[2 * blockIdx.x + (k)] = wall_clock64(); } while (0)
// per-pass stamps of the 14x14 layers' k loops (conv3: slots 13-16, conv4: 22-29)
#define CH_PASS(S, C_IN, cb) do { if ((S) == 14) { __builtin_amdgcn_sched_barrier(0); CH_STAMP(((C_IN) == 32 ? 13 : 22) + (cb) / 8); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define CH_PASS(S, C_IN, cb) do { } while (0)
#define CH_SPAN(k) do { } while (0)
#define CH_STAMP(i) do { } while (0)
#define CH_CLK(i) do { } while (0)
#endif

// The classifier behind a chain that ends in a pooled map (th_conv_chain_head_fwd): Linear(k, classes) on the flattened map + softmax
// cross-entropy, one ROW per workgroup -- the image's logits, loss term, dlogits and (cbpart) the per-channel sums of dX * [x > 0], all
// that a bias-only last conv needs of its gradient (nn.rs:54-60, loss.rs:101-195, 271-290, ops.rs:254-265, tensor.rs:1496-1519)
struct ChainHeadArgs {
    const float *w, *bias, *targets;   // [classes][k], [classes] (nullable), [n]
    float *dl, *rowstat, *cbpart;      // [n][16] (zeros past `classes`), [n][2] = {nll, hit}, [n][c_last] (nullable)
    int32_t *tick;                     // nullable: Adam's step counter, += 1 (optim.rs:84; nothing in this launch reads it)
    int classes, k, c_last, hw;
    float inv_b;
    const uint32_t *guard;             // th_ctx_set_update_guard (nullable): non-zero = a data-parallel exchange failed -- no tick
    uint32_t *step_word;               // ... and the exchange's step number, advanced with the tick (th_wide_head_grads_dp reads it; dp_dev.h)
};
__device__ __forceinline__ void chain_head_tick(const ChainHeadArgs &h) {
    if (h.guard && h.guard[0] != 0u) return;
    h.tick[0] += 1;
    if (h.step_word) h.step_word[0] += 1u;
}

// The three-layer classifier behind the reference chain's plane means (th_conv_chain_mlp3_xent): Linear(128, 128) + ReLU, Linear(128, 64) + ReLU,
// Linear(64, classes), softmax cross-entropy -- examples/train_mnist_cnn.rs:53-61 -- ONE ROW per workgroup, in the chain launch's last
// epilogue: forward, loss term, dlogits and the activations' gradients down to dX depend on the row only (nn.rs:54-60, loss.rs:101-195,
// ops.rs:254-265, 358-369); what adds over the batch (dW, db, the conv bias) is th_mlp3_xent's second launch, unchanged.
struct ChainMlp3Args {
    const float *w1, *b1, *w2, *b2, *w3, *b3;   // [128][128], [128], [64][128], [64], [classes][64], [classes] (biases nullable)
    const float *targets;                       // [n]
    float *a1, *a2, *dz1, *dz2, *dz3, *dx;      // [n][128], [n][64], [n][128], [n][64], [n][classes], [n][128] (dx nullable)
    float *part;                                // [n][2]: the row's NLL, hit
    int32_t *tick;                              // nullable: t += 1 (optim.rs:84; nothing in this launch reads it)
    int classes;
    float inv_b;
};

struct ConvChainArgs {
    const float *x;            // [n][1][28][28]
    const float *w[5], *b[5];  // taper layout [9 c_in][c_out] (tensor.rs:1262), bias [c_out]
    float *y;                  // reference chain: [n][128] plane means; simple chain: [n][64][7][7] pooled map
    float *cnt;                // reference chain: [n][128] outputs > 0 per plane (nullable)
    int n;
    ChainHeadArgs head;        // conv_chain_simple_kernel<true, ..> only
    ChainMlp3Args head3;       // conv_chain_reference_kernel<false, true> only
};

// ---- one 3x3 / pad 1 layer on the matrix cores: IN [C_IN][CIS] padded planes in LDS -> NSLOT accumulator tiles per wave ------------------------
// fp32 MFMA shares the vector ALUs' issue slots with everything else a SIMD does (conv_mfma.hip): an LDS operand read costs ~5-6
// cycles of the pipe that a 16x16x4 MFMA holds for 32, so what pays is operand REUSE.  A wave owns a PAIR of channel tiles (chA, chB: weight
// operands in registers) and ND "double" pixel tiles -- every pixel operand read feeds two MFMAs -- and, where the (pixel tile, channel
// tile) count does not divide into doubles evenly over the four SIMDs, one "single" tile (the last pixel tile x chA) on the first waves:
//   28x28, 32 channels: 49 x 2 tiles = 6 doubles per wave (pixel tiles w, w + 8, ...) + singles (48, ch 0 / 1) on waves 0 / 1   -> 25 | 25 | 24 | 24 per SIMD
//   14x14, 64 channels: 13 x 4 tiles = 3 doubles per wave (pair w & 1, pixel tiles w >> 1, + 4, + 8) + singles (12, ch 0..3) on waves 0..3 -> 13 per SIMD
//   7x7, 128 channels:   4 x 8 tiles = 2 doubles per wave (pair w & 3, pixel tiles 2 (w >> 2), + 1)                              ->  8 per SIMD
// (waves w and w + 4 share a SIMD).  Same per-output k order as one tile per wave: the sums do not depend on the mapping.
template <int S, int C_OUT> struct ChainGeo {
    static constexpr int WP = S + 2, CIS = ch_cis(WP), PX = S * S, NPT = (PX + 15) / 16, NCT = C_OUT / 16;
    // (+ two mappings for input gradients, conv_layer_chain_kernel<.., LIN>: 14x14 with 32 channels = 13 x 2 tiles -> 2 doubles per wave
    // (pixel tiles w, w + 8; tiles past 12 are dead: computed on pixel 0, never stored); 7x7 with 64 channels = 4 x 4 tiles -> 1 double per wave)
    static_assert((NCT == 2 && NPT == 49) || (NCT == 4 && NPT == 13) || (NCT == 8 && NPT == 4) || (NCT == 2 && NPT == 13) || (NCT == 4 && NPT == 4),
                  "the compiled tile mappings");
    static constexpr int ND = NCT == 2 ? (NPT == 49 ? 6 : 2) : (NCT == 4 ? (NPT == 13 ? 3 : 1) : 2);                 // doubles per wave
    static constexpr int NS = NCT == 2 ? (NPT == 49 ? 2 : 0) : (NCT == 4 ? (NPT == 13 ? 4 : 0) : 0);                 // waves with a single
    static constexpr int NSLOT = 2 * ND + (NS ? 1 : 0), STILE = NPT - 1;
    __device__ static int chA(int w) { return NCT == 2 ? (w & 1) : (NCT == 4 ? 2 * (w & 1) + ((w >> 1) & 1) : 2 * (w & 3)); }
    __device__ static int chB(int w) { return NCT == 2 ? (w & 1) ^ 1 : (NCT == 4 ? 2 * (w & 1) + (((w >> 1) & 1) ^ 1) : 2 * (w & 3) + 1); }
    __device__ static int dtile(int w, int i) { return NCT == 2 ? w + 8 * i : (NCT == 4 ? (w >> 1) + 4 * i : 2 * (w >> 2) + i); }
    __device__ static bool has_single(int w) { return w < NS; }
    // accumulator slot k of wave w -> (pixel tile, channel tile): slots 2 i, 2 i + 1 = double i x (chA, chB); slot 2 ND = the single
    __device__ static int slot_px(int w, int k) { return k < 2 * ND ? dtile(w, k >> 1) : STILE; }
    __device__ static int slot_ch(int w, int k) { return (k < 2 * ND && (k & 1)) ? chB(w) : chA(w); }
    __device__ static bool slot_live(int w, int k) { return k < 2 * ND || has_single(w); }
};

// Which pixel a lane's column of a pixel tile is.  Plain: 16 consecutive pixels (row-major, wrapping rows).  PM (a layer followed by a 2x2
// max-pool, S even): tile t = pooled positions 4 t .. 4 t + 3 (row-major over the (S/2)^2 pooled map), lane l16 = window position l16 & 3
// (dy = bit 1, dx = bit 0) of pooled position 4 t + (l16 >> 2) -- the four lanes of a quad hold one window, so the pooling is two DPP
// quad-permute maxima on the accumulators and neither the full-resolution tile nor the pooling pass exists.  The tile count is the same
// (28: 196 / 4 = 49; 14: 49 / 4 -> 13), and so is every output's k order: bit-identical results.  -1: no pixel (computed on pixel 0, never stored).
template <int S, bool PM>
__device__ __forceinline__ int chain_lane_pixel(int tile, int l16) {
    if constexpr (PM) {
        constexpr int HP = S / 2;
        const int q = tile * 4 + (l16 >> 2);
        return q < HP * HP ? (2 * (q / HP) + ((l16 >> 1) & 1)) * S + 2 * (q % HP) + (l16 & 1) : -1;
    } else {
        const int p = tile * 16 + l16;
        return p < S * S ? p : -1;
    }
}

// (barriers: lds_barrier() of common.h -- the weight loads in flight, global memory nobody writes, stay in flight across them)
__device__ __forceinline__ void chain_sync() { lds_barrier(); }

// weight operands of pass cb (8 input channels, 18 k-steps) of a layer's [9 C_IN][C_OUT] slab: k-step s wants row (cb + 4 (s / 9) + g4) * 9 + s % 9,
// columns 16 chA + l16 and 16 chB + l16 -- two dwords per lane and k-step, requested a pass ahead (the first pass of a layer: before the
// previous layer's epilogue)
struct ChainW { float a[18], b[18]; };
template <int S, int C_IN, int C_OUT>
__device__ __forceinline__ void chain_weights(const float *__restrict__ w, int cb, ChainW &wr, int wave, int lane) {
    using G = ChainGeo<S, C_OUT>;
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, 9 * C_IN * C_OUT * 4, 0x00020000);
    const int row = ((lane >> 4) * 9 * C_OUT + (lane & 15)) * 4, va = row + 64 * G::chA(wave), vb = row + 64 * G::chB(wave);
#pragma unroll
    for (int s = 0; s < 18; ++s) {
        const int so = ((cb + 4 * (s / 9)) * 9 + s % 9) * C_OUT * 4;
#ifdef CH_NO_WEIGHTS   /* timing probe: no weight loads (wrong results) */
        wr.a[s] = __builtin_bit_cast(float, va + so);
        wr.b[s] = __builtin_bit_cast(float, vb + so);
#else
        wr.a[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, va, so, 0));
        wr.b[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, vb, so, 0));
#endif
    }
}

// wc: the operands of pass 0, already requested (chain_weights).  The k loop's only non-MFMA instructions are its LDS reads and the next
// pass's weight requests: a tile's window-corner address lives in a register that moves on by eight channels per pass (the tap /
// channel-group part of an operand address is the read's immediate offset -- kept below 64 KB by hiding the region's base from constant
// folding), the two weight sets alternate between passes instead of being copied, and the waves without a single tile run their own copy of
// the pass (no branch inside it).
// Where the non-MFMA instructions sit matters more than how many there are (experiments/mfma_rate.hip, this loop's shape on its own: the
// next step's reads issued as a group in front of the step's MFMAs: 38.3 cycles per MFMA per SIMD with two waves, 44.6 with one; ONE read
// behind each pair of MFMAs: 34.3 / 36.0; no reads at all: 33.2 -- an instruction's issue is hidden by the MFMA in flight only when it
// follows one).  So: the read of the next k-step's operand i sits behind MFMA pair i, and the next pass's 36 weight requests sit two per
// k-step behind the first two pairs instead of in a burst at the head of the pass (in-kernel stamps, simple chain: the burst cost every
// wave ~290 cycles of issue per pass with the matrix pipe idle behind it).
// NEXT_* / w_next: the layer that follows (0 / nullptr: none) -- ITS first pass is requested during this layer's last pass, into wc,
// where the caller's next chain_mfma expects it.  (Without a next layer the last pass's requests go to a zero-sized buffer: they return
// zeros without touching memory, and the loop body stays free of branches.)
template <int S, int C_IN, int C_OUT, int NEXT_S = 0, int NEXT_CIN = 0, int NEXT_COUT = 0, bool PM = false>
__device__ __forceinline__ void chain_mfma(const float *in, const float *__restrict__ w, ChainW &wc, floatx4 (&acc)[ChainGeo<S, C_OUT>::NSLOT], int wave,
                                           int lane, const float *__restrict__ w_next = nullptr) {
    using G = ChainGeo<S, C_OUT>;
    constexpr bool HAS_NEXT = NEXT_COUT != 0;
    using G2 = ChainGeo<HAS_NEXT ? NEXT_S : S, HAS_NEXT ? NEXT_COUT : C_OUT>;
    constexpr int WP = G::WP, CIS = G::CIS, ND = G::ND, NP = ND + (G::NS ? 1 : 0), KS = 18, CO2 = HAS_NEXT ? NEXT_COUT : C_OUT;
    static_assert(C_IN % 16 == 0 && ND >= 1, "whole pairs of 8-channel passes");   // (ND == 1: both weight requests of a k-step sit behind its one pair)
    const int l16 = lane & 15, g4 = lane >> 4;
#ifdef CH_PROBE_HS   /* timing probe: every wave (1) / no wave (0) runs the single tile, as a compile-time constant (wrong results) */
    constexpr bool hs = CH_PROBE_HS != 0;
#else
    const bool hs = G::has_single(wave);     // wave-uniform
#endif
    typedef __attribute__((address_space(3))) const float lds_cf;
    lds_cf *pt[NP];              // this lane's window corner in each of its pixel tiles (+ its lane group's channel), current pass
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        int p = chain_lane_pixel<S, PM>(i < ND ? G::dtile(wave, i) : G::STILE, l16);
        if (p < 0) p = 0;        // lanes past the image compute on pixel 0 and are never stored
        pt[i] = (lds_cf *)(in + (p / S) * WP + p % S + g4 * CIS);
        asm volatile("" : "+v"(pt[i]));
    }
#pragma unroll
    for (int k = 0; k < G::NSLOT; ++k) acc[k] = floatx4{0.f, 0.f, 0.f, 0.f};
    ChainW wn;
    // weight addressing (chain_weights): this layer's, and the next layer's first pass
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, 9 * C_IN * C_OUT * 4, 0x00020000);
    const int va = ((lane >> 4) * 9 * C_OUT + (lane & 15)) * 4 + 64 * G::chA(wave), vb = va + 64 * (G::chB(wave) - G::chA(wave));
    const int va2 = ((lane >> 4) * 9 * CO2 + (lane & 15)) * 4 + 64 * G2::chA(wave), vb2 = va2 + 64 * (G2::chB(wave) - G2::chA(wave));
#ifdef CH_NO_READS   /* timing probe: the k loop without its LDS reads (wrong results) */
#define CH_REQ1(B, SS, I) { B[I] = __builtin_bit_cast(float, (int)(size_t)pt[I]); }
#else
#define CH_REQ1(B, SS, I) { B[I] = pt[I][(4 * ((SS) / 9)) * CIS + (((SS) % 9) / 3) * WP + ((SS) % 9) % 3]; }
#endif
#ifdef CH_NO_WEIGHTS   /* timing probe: no weight loads (wrong results) */
#define CH_WREQ(DST, RS, V, SO) { DST = __builtin_bit_cast(float, (V) + (SO)); }
#else
#define CH_WREQ(DST, RS, V, SO) { DST = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(RS, V, SO, 0)); }
#endif
    // one pass: 18 k-steps on WCUR; behind pair 0 / pair 1 of step s, the requests for WNXT.a[s] / .b[s] (row (pass + 4 (s / 9)) * 9 + s % 9
    // of the slab RS, ROWB bytes per row); behind pair i, the read of operand i of step s + 1
#define CH_PASS_BODY_HS(HS, WCUR, WNXT, RS, VA, VB, SOBASE, ROWB)                                                                            \
    {                                                                                                                                        \
        float b0[NP], b1[NP];                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < (HS ? NP : ND); ++i) CH_REQ1(b0, 0, i)                                                         \
        _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                                               \
            _Pragma("unroll") for (int i = 0; i < ND; ++i) {                                                                                 \
                acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(WCUR.a[s], b0[i], acc[2 * i], 0, 0, 0);                                    \
                acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(WCUR.b[s], b0[i], acc[2 * i + 1], 0, 0, 0);                            \
                __builtin_amdgcn_sched_barrier(0);                                                                                           \
                if (s + 1 < KS) CH_REQ1(b1, s + 1, i)                                                                                        \
                if (i == 0) CH_WREQ(WNXT.a[s], RS, VA, (SOBASE) + ((4 * (s / 9)) * 9 + s % 9) * (ROWB))                                      \
                if (i == (ND > 1 ? 1 : 0)) CH_WREQ(WNXT.b[s], RS, VB, (SOBASE) + ((4 * (s / 9)) * 9 + s % 9) * (ROWB))                       \
                __builtin_amdgcn_sched_barrier(0);                                                                                           \
            }                                                                                                                                \
            if (G::NS && HS) {                                                                                                               \
                acc[G::NSLOT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(WCUR.a[s], b0[NP - 1], acc[G::NSLOT - 1], 0, 0, 0);                 \
                __builtin_amdgcn_sched_barrier(0);                                                                                           \
                if (s + 1 < KS) CH_REQ1(b1, s + 1, NP - 1)                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                                                           \
            }                                                                                                                                \
            if (s + 1 < KS) { _Pragma("unroll") for (int i = 0; i < (HS ? NP : ND); ++i) b0[i] = b1[i]; }                                    \
        }                                                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < NP; ++i) pt[i] += 8 * CIS;                                                                     \
    }
    // the waves without a single tile run their own copy of the pass: a wave-uniform branch per PASS -- per k-step, the taken branch cost
    // those waves ~35 of a step's 192 cycles (in-kernel stamps: 0.83 of the MFMA rate for a wave alone against 0.93 for its partner)
#define CH_PASS_BODY(WCUR, WNXT, RS, VA, VB, SOBASE, ROWB)                                                                                   \
    if (G::NS == 0 || hs) CH_PASS_BODY_HS(true, WCUR, WNXT, RS, VA, VB, SOBASE, ROWB) else CH_PASS_BODY_HS(false, WCUR, WNXT, RS, VA, VB, SOBASE, ROWB)
#pragma unroll 1
    for (int cb = 0; cb < C_IN; cb += 16) {
        CH_PASS(S, C_IN, cb);
        CH_PASS_BODY(wc, wn, rw, va, vb, (cb + 8) * 9 * C_OUT * 4, C_OUT * 4)
        CH_PASS(S, C_IN, cb + 8);
        const bool last = cb + 16 >= C_IN;   // (uniform: scalar selects)
        const auto rn = __builtin_amdgcn_make_buffer_rsrc((void *)(last && HAS_NEXT ? w_next : w), 0,
                                                          last ? (HAS_NEXT ? 9 * NEXT_CIN * NEXT_COUT * 4 : 0) : 9 * C_IN * C_OUT * 4, 0x00020000);
        const int van = last ? va2 : va, vbn = last ? vb2 : vb, sob = last ? 0 : (cb + 16) * 9 * C_OUT * 4, rowb = last ? CO2 * 4 : C_OUT * 4;
        CH_PASS_BODY(wn, wc, rn, van, vbn, sob, rowb)
    }
#undef CH_PASS_BODY
#undef CH_PASS_BODY_HS
#undef CH_REQ1
#undef CH_WREQ
}

// ---- the same k loop in HALF passes (r06, conv_chain_simple_kernel<.., LEAN>): the weight operands of 9 k-steps -- one group of 4 input channels
// -- per request set instead of 18, i.e. 36 registers of weights in flight instead of 72, for the instance that must fit 128 registers so that
// two workgroups share a CU (the shorter prefetch distance is the other workgroup's to cover).  Same k order (k-step s of pass cb = tap s % 9 of
// channel group cb + 4 (s / 9): the half passes are the groups in turn), same placement of reads and requests behind the MFMA pairs: the sums
// are the full-pass loop's bit for bit.  No next-layer prefetch (the simple chain's conv2 is its last layer).
struct ChainW9 { float a[9], b[9]; };
template <int S, int C_IN, int C_OUT>
__device__ __forceinline__ void chain_weights9(const float *__restrict__ w, int c4, ChainW9 &wr, int wave, int lane) {
    using G = ChainGeo<S, C_OUT>;
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, 9 * C_IN * C_OUT * 4, 0x00020000);
    const int row = ((lane >> 4) * 9 * C_OUT + (lane & 15)) * 4, va = row + 64 * G::chA(wave), vb = row + 64 * G::chB(wave);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const int so = (c4 * 9 + s) * C_OUT * 4;
        wr.a[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, va, so, 0));
        wr.b[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, vb, so, 0));
    }
}
template <int S, int C_IN, int C_OUT, bool PM>
__device__ __forceinline__ void chain_mfma9(const float *in, const float *__restrict__ w, ChainW9 &wc, floatx4 (&acc)[ChainGeo<S, C_OUT>::NSLOT], int wave,
                                            int lane) {
    using G = ChainGeo<S, C_OUT>;
    constexpr int WP = G::WP, CIS = G::CIS, ND = G::ND, NP = ND + (G::NS ? 1 : 0), KS = 9;
    static_assert(C_IN % 8 == 0 && ND >= 1, "whole pairs of 4-channel half passes");
    const int l16 = lane & 15, g4 = lane >> 4;
    const bool hs = G::has_single(wave);     // wave-uniform
    typedef __attribute__((address_space(3))) const float lds_cf;
    lds_cf *pt[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        int p = chain_lane_pixel<S, PM>(i < ND ? G::dtile(wave, i) : G::STILE, l16);
        if (p < 0) p = 0;
        pt[i] = (lds_cf *)(in + (p / S) * WP + p % S + g4 * CIS);
        asm volatile("" : "+v"(pt[i]));
    }
#pragma unroll
    for (int k = 0; k < G::NSLOT; ++k) acc[k] = floatx4{0.f, 0.f, 0.f, 0.f};
    ChainW9 wn;
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, 9 * C_IN * C_OUT * 4, 0x00020000);
    const int va = ((lane >> 4) * 9 * C_OUT + (lane & 15)) * 4 + 64 * G::chA(wave), vb = va + 64 * (G::chB(wave) - G::chA(wave));
#define CH9_REQ1(B, SS, I) { B[I] = pt[I][((SS) / 3) * WP + (SS) % 3]; }
#define CH9_WREQ(DST, RS, V, SO) { DST = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(RS, V, SO, 0)); }
#define CH9_HALF_HS(HS, WCUR, WNXT, RS, SOBASE)                                                                                              \
    {                                                                                                                                        \
        float b0[NP], b1[NP];                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < (HS ? NP : ND); ++i) CH9_REQ1(b0, 0, i)                                                        \
        _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                                               \
            _Pragma("unroll") for (int i = 0; i < ND; ++i) {                                                                                 \
                acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(WCUR.a[s], b0[i], acc[2 * i], 0, 0, 0);                                    \
                acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(WCUR.b[s], b0[i], acc[2 * i + 1], 0, 0, 0);                            \
                __builtin_amdgcn_sched_barrier(0);                                                                                           \
                if (s + 1 < KS) CH9_REQ1(b1, s + 1, i)                                                                                       \
                if (i == 0) CH9_WREQ(WNXT.a[s], RS, va, (SOBASE) + s * C_OUT * 4)                                                            \
                if (i == (ND > 1 ? 1 : 0)) CH9_WREQ(WNXT.b[s], RS, vb, (SOBASE) + s * C_OUT * 4)                                             \
                __builtin_amdgcn_sched_barrier(0);                                                                                           \
            }                                                                                                                                \
            if (G::NS && HS) {                                                                                                               \
                acc[G::NSLOT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(WCUR.a[s], b0[NP - 1], acc[G::NSLOT - 1], 0, 0, 0);                 \
                __builtin_amdgcn_sched_barrier(0);                                                                                           \
                if (s + 1 < KS) CH9_REQ1(b1, s + 1, NP - 1)                                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                                                           \
            }                                                                                                                                \
            if (s + 1 < KS) { _Pragma("unroll") for (int i = 0; i < (HS ? NP : ND); ++i) b0[i] = b1[i]; }                                    \
        }                                                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < NP; ++i) pt[i] += 4 * CIS;                                                                     \
    }
#define CH9_HALF(WCUR, WNXT, RS, SOBASE) if (G::NS == 0 || hs) CH9_HALF_HS(true, WCUR, WNXT, RS, SOBASE) else CH9_HALF_HS(false, WCUR, WNXT, RS, SOBASE)
#pragma unroll 1
    for (int c4 = 0; c4 < C_IN; c4 += 8) {
        CH9_HALF(wc, wn, rw, (c4 + 4) * 9 * C_OUT * 4)
        const bool last = c4 + 8 >= C_IN;    // (uniform; the last half pass requests from a zero-sized buffer: zeros, no memory touched, no branch)
        const auto rn = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, last ? 0 : 9 * C_IN * C_OUT * 4, 0x00020000);
        CH9_HALF(wn, wc, rn, (c4 + 8) * 9 * C_OUT * 4)
    }
#undef CH9_HALF
#undef CH9_HALF_HS
#undef CH9_REQ1
#undef CH9_WREQ
}

// bias of this lane's channels (16 chA + 4 g4 + e, 16 chB + 4 g4 + e)
struct ChainBias { float a[4], b[4]; };
template <int S, int C_OUT>
__device__ __forceinline__ void chain_bias(const float *__restrict__ bias, ChainBias &bv, int wave, int lane) {
    using G = ChainGeo<S, C_OUT>;
    const int ca = 16 * G::chA(wave) + 4 * (lane >> 4), cb = 16 * G::chB(wave) + 4 * (lane >> 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        bv.a[e] = bias ? bias[ca + e] : 0.f;
        bv.b[e] = bias ? bias[cb + e] : 0.f;
    }
}

// accumulators (+ bias, ReLU) -> PLANES: the next layer's padded planes OUT [C_OUT][CIS] (halo already zero); else a plain tile
// T [C_OUT][ch_tile_ld(pixels)] of the layer's pixels (for the pooling passes)
template <int S, int C_OUT, bool PLANES>
__device__ __forceinline__ void chain_store(const floatx4 (&acc)[ChainGeo<S, C_OUT>::NSLOT], const ChainBias &bv, float *out, int wave, int lane) {
    using G = ChainGeo<S, C_OUT>;
    constexpr int LD = PLANES ? G::CIS : ch_tile_ld(G::PX);
    const int l16 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int k = 0; k < G::NSLOT; ++k) {
        if (!G::slot_live(wave, k)) continue;
        const int p = G::slot_px(wave, k) * 16 + l16;
        if (p >= G::PX) continue;
        const bool second = k < 2 * G::ND && (k & 1);
        float *o = out + (16 * G::slot_ch(wave, k) + 4 * g4) * LD + (PLANES ? (p / S + 1) * G::WP + p % S + 1 : p);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[k][e] + (second ? bv.b[e] : bv.a[e]);
            v = v > 0.f ? v : 0.f;
            o[e * LD] = v;
        }
    }
}

// accumulators of a PM-mapped layer (+ bias, ReLU) -> their 2x2 maxima, straight from the registers: the next layer's padded planes OUT
// [C_OUT][CISO] (TO_GLOBAL false) or the image's pooled NCHW map [C_OUT][HP * HP] in global memory, and (flat) a copy [C_OUT][HP * HP] in
// LDS for the classifier rows.  tensor.rs:1449-1461 scans the window with a strict > from -inf, so NaN never wins -- behind the ReLU
// (v > 0 ? v : 0) no value is NaN or -0, and the maximum of four such values does not depend on the order: v_max over the quad.
template <int S, int C_OUT, bool TO_GLOBAL>
__device__ __forceinline__ void chain_store_pooled(const floatx4 (&acc)[ChainGeo<S, C_OUT>::NSLOT], const ChainBias &bv, float *out, int wave, int lane,
                                                   float *flat = nullptr) {
    using G = ChainGeo<S, C_OUT>;
    constexpr int HP = S / 2, NPO = HP * HP, WPO = HP + 2, CISO = ch_cis(WPO), LD = TO_GLOBAL ? NPO : CISO;
    static_assert(S % 2 == 0, "2x2 windows");
    const int l16 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int k = 0; k < G::NSLOT; ++k) {
        if (!G::slot_live(wave, k)) continue;
        const bool second = k < 2 * G::ND && (k & 1);
        float m[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[k][e] + (second ? bv.b[e] : bv.a[e]);
            v = v > 0.f ? v : 0.f;
            // (values >= +0, never NaN: their order is the order of their bit patterns as integers -- an integer max carries no NaN
            // canonicalisation and takes the quad permute as its DPP operand)
            int iv = __builtin_bit_cast(int, v);
            iv = max(iv, __builtin_amdgcn_mov_dpp(iv, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
            iv = max(iv, __builtin_amdgcn_mov_dpp(iv, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
            m[e] = __builtin_bit_cast(float, iv);
        }
        const int q = G::slot_px(wave, k) * 4 + (l16 >> 2);
        if ((l16 & 3) != 0 || q >= NPO) continue;
        const int c0 = 16 * G::slot_ch(wave, k) + 4 * g4;
        float *o = out + c0 * LD + (TO_GLOBAL ? q : (q / HP + 1) * WPO + q % HP + 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e * LD] = m[e];
            if (flat) flat[(c0 + e) * NPO + q] = m[e];
        }
    }
}

// 2x2 / stride-2 maxima of the tile (strict >: NaN never wins, tensor.rs:1449-1461) -> interior of the next layer's padded planes, or
// (TO_GLOBAL) the pooled NCHW map of this image.  A thread owns one pooled position and a block of channels.
template <int S, int C_OUT, bool TO_GLOBAL, bool COPY = false>
__device__ __forceinline__ void chain_pool(const float *tile, float *out, int t, float *flat = nullptr) {
    constexpr int LD = ch_tile_ld(S * S), HP = S / 2, NP = HP * HP, WPO = HP + 2, CISO = ch_cis(WPO);
    constexpr int CHUNKS = (CH_NT / NP >= 8 ? 8 : (CH_NT / NP >= 4 ? 4 : (CH_NT / NP >= 2 ? 2 : 1))), CPC = C_OUT / CHUNKS;
    static_assert(S % 2 == 0 && NP <= CH_NT && C_OUT % CHUNKS == 0, "pooled plane fits the workgroup");
    const int q = t % NP, chunk = t / NP;
    if (chunk >= CHUNKS) return;
    const int pr = q / HP, pc = q % HP;
    const float *b = tile + (chunk * CPC) * LD + 2 * pr * S + 2 * pc;
    float *o = TO_GLOBAL ? out + (chunk * CPC) * NP + q : out + (chunk * CPC) * CISO + (pr + 1) * WPO + pc + 1;
#pragma unroll 8
    for (int cl = 0; cl < CPC; ++cl, b += LD, o += TO_GLOBAL ? NP : CISO) {
        const float2 r0 = *reinterpret_cast<const float2 *>(b), r1 = *reinterpret_cast<const float2 *>(b + S);
        float m = -INFINITY;
        m = r0.x > m ? r0.x : m;
        m = r0.y > m ? r0.y : m;
        m = r1.x > m ? r1.x : m;
        m = r1.y > m ? r1.y : m;
        *o = m;
        if (COPY) flat[(chunk * CPC + cl) * NP + q] = m;     // the image's flattened [C_OUT][NP] map, for the classifier rows below
    }
}

// the 28x28 image into its zero-haloed 30x30 LDS plane
__device__ __forceinline__ void chain_load_image(const float *__restrict__ xi, float *img, int t) {
    for (int e = t; e < 900; e += CH_NT) {
        const int r = e / 30 - 1, c = e % 30 - 1;
        img[e] = (r >= 0 && r < 28 && c >= 0 && c < 28) ? xi[r * 28 + c] : 0.f;
    }
}

// conv1 (1 -> 32 channels, K = 9) on the matrix cores as well: k = the tap, padded to three k-steps of 4 with zero weights (an fma with a
// zero weight leaves the chain's value unchanged: the sum is conv_pool.hip conv1_kernel's fmaf chain over taps 0..8, bit for bit).  294 MFMAs
// per image against 226 K FMAs on the vector ALUs (which would also wait on 320 scalar weight loads): 10.3 -> ~2 us of the chain.
// IMG = the zero-haloed 30x30 plane; tile mapping = ChainGeo<28, 32> (the epilogues above apply).
struct ChainW1 { float a[3], b[3]; };
__device__ __forceinline__ void chain_conv1_weights(const float *__restrict__ w, ChainW1 &wa, int wave, int lane) {
    using G = ChainGeo<28, 32>;
    const int l16 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        wa.a[s] = 4 * s + g4 < 9 ? w[(4 * s + g4) * 32 + 16 * G::chA(wave) + l16] : 0.f;
        wa.b[s] = 4 * s + g4 < 9 ? w[(4 * s + g4) * 32 + 16 * G::chB(wave) + l16] : 0.f;
    }
}

template <bool PM = false>
__device__ __forceinline__ void chain_conv1_mfma(const float *img, const ChainW1 &wa, floatx4 (&acc)[ChainGeo<28, 32>::NSLOT], int wave, int lane) {
    using G = ChainGeo<28, 32>;
    const int l16 = lane & 15, g4 = lane >> 4;
    int toff[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int tap = 4 * s + g4, tt = tap < 9 ? tap : 0;
        toff[s] = (tt / 3) * 30 + tt % 3;
    }
    const bool pad_lane = 8 + g4 >= 9;     // k-step 2, lane groups 1..3: taps 9..11 carry zero weights
#pragma unroll
    for (int k = 0; k < G::NSLOT; ++k) acc[k] = floatx4{0.f, 0.f, 0.f, 0.f};
    // every tile's three operand reads first, then the MFMAs k-step by k-step across the tiles: an accumulator's three dependent MFMAs are
    // a whole row of independent ones apart, and the LDS latency is paid once (tile by tile -- read, wait, three dependent MFMAs -- this
    // phase took 1.5 us for 39 MFMAs per wave)
    float b[G::ND + 1][3];
#pragma unroll
    for (int i = 0; i < G::ND + 1; ++i) {
        int p = chain_lane_pixel<28, PM>(i < G::ND ? G::dtile(wave, i) : G::STILE, l16);
        if (p < 0) p = 0;
        const float *px = img + (p / 28) * 30 + p % 28;
#pragma unroll
        for (int s = 0; s < 3; ++s) b[i][s] = (s == 2 && pad_lane ? img : px)[toff[s]];   // padded taps read the halo corner (0.0): 0 * 0, never 0 * Inf
    }
    const bool hs = G::has_single(wave);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int i = 0; i < G::ND; ++i) {
            acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.a[s], b[i][s], acc[2 * i], 0, 0, 0);
            acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.b[s], b[i][s], acc[2 * i + 1], 0, 0, 0);
        }
        if (hs) acc[G::NSLOT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.a[s], b[G::ND][s], acc[G::NSLOT - 1], 0, 0, 0);
    }
}

// zeros on the halo ring of C padded planes [C][CIS] of an S x S map (the interior is written by chain_store / chain_pool): a wave per
// channel (no per-element division), lane l takes ring elements l, l + 64
template <int S, int C>
__device__ __forceinline__ void chain_zero_halo(float *planes, int wave, int lane) {
    constexpr int WP = S + 2, CIS = ch_cis(WP), RING = 4 * S + 4;
    int pos[(RING + 63) / 64];
#pragma unroll
    for (int j = 0; j < (RING + 63) / 64; ++j) {
        const int r = lane + 64 * j;
        pos[j] = r < WP ? r                                          // top row
               : r < 2 * WP ? (WP - 1) * WP + (r - WP)               // bottom row
               : r < 2 * WP + S ? (r - 2 * WP + 1) * WP              // left column
               : r < RING ? (r - 2 * WP - S + 1) * WP + WP - 1       // right column
               : -1;
    }
    for (int c = wave; c < C; c += CH_NT / 64) {
#pragma unroll
        for (int j = 0; j < (RING + 63) / 64; ++j)
            if (pos[j] >= 0) planes[c * CIS + pos[j]] = 0.f;
    }
}

// ---- the reference CNN's front (examples/train_mnist_cnn.rs:35-62): 1 -> 32, 32 -> 32 + pool, 32 -> 64, 64 -> 64 + pool, 64 -> 128 + global mean ----
// LDS map (floats); regions reuse each other's space once a barrier has retired their readers:
//   A1 [32][912] @0        conv2's input           T2 [32][788] @0      conv2's outputs (over A1)    A2 [32][272] @25216  conv3's input
//   A3 [64][272] @0        conv4's input           T4 [64][196] @17408  conv4's outputs (over A2)    A4 [64][112] @0      conv5's input
//   T5 [128][52] @7168     conv5's outputs         IMG [900]    @33920
constexpr int CR_A2 = 32 * ch_tile_ld(784), CR_IMG = CR_A2 + 32 * ch_cis(16), CR_LDS = CR_IMG + 900;
constexpr int CR_T4 = 64 * ch_cis(16), CR_T5 = 64 * ch_cis(9);
static_assert(32 * ch_cis(30) <= CR_IMG && CR_T4 + 64 * ch_tile_ld(196) <= CR_LDS && CR_T5 + 128 * ch_tile_ld(49) <= CR_LDS, "LDS map");

// LOOP (batches above one image per CU, r05): min(n, 256) workgroups WALK the images -- image i + gridDim.x's pixels are requested in front of
// conv5's k loop and stored into IMG behind it (IMG is nobody's after conv1), conv2's first weight pass rides in conv5's last pass like any
// next layer's, conv1's weights stay in their registers: a workgroup's next image starts without the launch, the image round trip and the
// weight round trips that a fresh workgroup pays.  Same per-image arithmetic: bit-identical outputs.  LOOP = false is r04's kernel as it was.
// xs: LDS [128], the image's plane means.  sc: LDS scratch, >= 1 200 floats, disjoint from xs.  512 threads; ends without a barrier (global
// stores only).  Every weight is requested up front, in the order of use, as the operand registers of the thread that will use it
// (forward: a thread owns a k range of one output row -- float4 along k; backward: a thread owns a range of the contracted index for one
// output column -- dwords, lanes on consecutive columns): the weights cross the fabric while the means are being formed and while the
// earlier stages run; 197 KB per workgroup from L2.  Sums: a thread's range in ascending index order, then a fixed tree over the threads
// of an output (forward: xor shuffles; backward: four partial sums through LDS, ((0 + 1) + (2 + 3))) -- deterministic, and within fp32
// rounding of the k-ordered chain like every product here (tolerance 1e-4).
struct ChainMlp3W {
    float4 f1[8], f2[4], f3;       // W1[o = t >> 2][32 (t & 3) ..], W2[o = t >> 3][16 (t & 7) ..], W3[o = t >> 4][4 (t & 15) ..]
    float b3c[16], b2c[16], b1c[32];   // W3[c][j = t] (t < 64), W2[16 (t >> 7) + q][i = t & 127], W1[32 (t >> 7) + q][k = t & 127]
};
__device__ __forceinline__ void chain_mlp3_weights(const ChainMlp3Args &h, ChainMlp3W &w, int t) {
    const float4 *r1 = reinterpret_cast<const float4 *>(h.w1 + (t >> 2) * 128 + 32 * (t & 3));
#pragma unroll
    for (int q = 0; q < 8; ++q) w.f1[q] = r1[q];
    const float4 *r2 = reinterpret_cast<const float4 *>(h.w2 + (t >> 3) * 128 + 16 * (t & 7));
#pragma unroll
    for (int q = 0; q < 4; ++q) w.f2[q] = r2[q];
    w.f3 = (t >> 4) < h.classes ? *reinterpret_cast<const float4 *>(h.w3 + (t >> 4) * 64 + 4 * (t & 15)) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < 16; ++c) w.b3c[c] = (t < 64 && c < h.classes) ? h.w3[c * 64 + t] : 0.f;
#ifdef CH_EXP_NO_BWD_LOADS   /* timing probe: the backward operands are not loaded (wrong results) */
#pragma unroll
    for (int q = 0; q < 16; ++q) w.b2c[q] = (float)(t + q);
#pragma unroll
    for (int q = 0; q < 32; ++q) w.b1c[q] = (float)(t - q);
#else
#pragma unroll
    for (int q = 0; q < 16; ++q) w.b2c[q] = h.w2[(16 * (t >> 7) + q) * 128 + (t & 127)];
#pragma unroll
    for (int q = 0; q < 32; ++q) w.b1c[q] = h.w1[(32 * (t >> 7) + q) * 128 + (t & 127)];
#endif
}

__device__ __forceinline__ void chain_mlp3_rows(const ChainMlp3Args &h, const ChainMlp3W &w, const float *xs, float *sc, int img, int t) {
    float *A1 = sc, *A2 = sc + 128, *LG = sc + 192, *D3 = sc + 208, *D2 = sc + 224, *D1 = sc + 288, *RED = sc + 416;   // RED [4][128]
    const int lane = t & 63, r16 = lane & 15, g4 = lane >> 4;
    if (h.tick && img == 0 && t == 0) h.tick[0] += 1;
    {   // A1 = relu(W1 x + b1) (nn.rs:54-60, activation.rs:10-12)
        const int o = t >> 2, p = t & 3;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 x4 = *reinterpret_cast<const float4 *>(xs + 32 * p + 4 * q);
            s = fmaf(w.f1[q].x, x4.x, s); s = fmaf(w.f1[q].y, x4.y, s); s = fmaf(w.f1[q].z, x4.z, s); s = fmaf(w.f1[q].w, x4.w, s);
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += h.b1 ? h.b1[o] : 0.f;
        s = s > 0.f ? s : 0.f;
        if (p == 0) { A1[o] = s; h.a1[(long)img * 128 + o] = s; }
    }
    chain_sync();
    {   // A2 = relu(W2 A1 + b2)
        const int o = t >> 3, p = t & 7;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 x4 = *reinterpret_cast<const float4 *>(A1 + 16 * p + 4 * q);
            s = fmaf(w.f2[q].x, x4.x, s); s = fmaf(w.f2[q].y, x4.y, s); s = fmaf(w.f2[q].z, x4.z, s); s = fmaf(w.f2[q].w, x4.w, s);
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        s += h.b2 ? h.b2[o] : 0.f;
        s = s > 0.f ? s : 0.f;
        if (p == 0) { A2[o] = s; h.a2[(long)img * 64 + o] = s; }
    }
    chain_sync();
    {   // logits = W3 A2 + b3
        const int o = t >> 4, p = t & 15;
        const float4 x4 = *reinterpret_cast<const float4 *>(A2 + 4 * p);
        float s = fmaf(w.f3.x, x4.x, 0.f);
        s = fmaf(w.f3.y, x4.y, s); s = fmaf(w.f3.z, x4.z, s); s = fmaf(w.f3.w, x4.w, s);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        if (p == 0 && o < 16) LG[o] = o < h.classes ? s + (h.b3 ? h.b3[o] : 0.f) : 0.f;
    }
    chain_sync();
    {   // softmax cross-entropy of the row (loss.rs:101-195, 271-290): every wave, redundantly -- no exchange
        float lg[4], dl[4], nll;
        int bi;
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[i] = g4 * 4 + i < h.classes ? LG[g4 * 4 + i] : -INFINITY;
        const float tf = h.targets[img];
        tail_row_softmax(lg, g4, h.classes, tf, h.inv_b, dl, nll, bi);
        if (t < 64 && r16 == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                D3[g4 * 4 + i] = dl[i];
                if (g4 * 4 + i < h.classes) h.dz3[(long)img * h.classes + g4 * 4 + i] = dl[i];
            }
            if (g4 == 0) {
                h.part[(long)img * 2] = nll;
                h.part[(long)img * 2 + 1] = fabsf((float)bi - tf) < 1e-6f ? 1.f : 0.f;   // loss.rs:283
            }
        }
    }
    chain_sync();
    if (t < 64) {   // dZ2 = (dlogits W3) * [A2 > 0] (ops.rs:254-265, 358-369): class order
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) s = fmaf(D3[c], w.b3c[c], s);      // (classes beyond h.classes: dlogits and weights are zero)
        s = A2[t] > 0.f ? s : 0.f;
        D2[t] = s;
        h.dz2[(long)img * 64 + t] = s;
    }
    chain_sync();
    {   // dZ1 = (dZ2 W2) * [A1 > 0]
        const int i = t & 127, jp = t >> 7;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s = fmaf(D2[16 * jp + q], w.b2c[q], s);
        RED[jp * 128 + i] = s;
    }
    chain_sync();
    if (t < 128) {
        float s = (RED[t] + RED[128 + t]) + (RED[256 + t] + RED[384 + t]);
        s = A1[t] > 0.f ? s : 0.f;
        D1[t] = s;
        h.dz1[(long)img * 128 + t] = s;
    }
    if (!h.dx) return;                          // (uniform)
    chain_sync();
    {   // dX = dZ1 W1 (the plane means' gradient: the last conv's bias takes it from there, th_mlp3_gap)
        const int k = t & 127, ip = t >> 7;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) s = fmaf(D1[32 * ip + q], w.b1c[q], s);
        RED[ip * 128 + k] = s;                  // (RED's readers of the stage before are behind the barrier above)
    }
    chain_sync();
    if (t < 128) h.dx[(long)img * 128 + t] = (RED[t] + RED[128 + t]) + (RED[256 + t] + RED[384 + t]);
}

static_assert(CR_T4 + 64 * ch_tile_ld(196) <= CR_IMG && 32 * ch_tile_ld(784) <= CR_IMG && CR_T5 + 128 * ch_tile_ld(49) <= CR_IMG, "IMG is free behind conv1");
template <bool LOOP, bool HEAD3 = false>
__global__ __launch_bounds__(CH_NT, 1) void conv_chain_reference_kernel(ConvChainArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t0_ = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t0_ >> 6);
    ChainBias bv;
    ChainW wc;
    ChainW1 wa;

    CH_STAMP(0);
    CH_CLK(20);
    chain_conv1_weights(a.w[0], wa, wave, t0_ & 63);
    chain_load_image(a.x + (long)blockIdx.x * 784, lds + CR_IMG, t0_);
    chain_weights<28, 32, 32>(a.w[1], 0, wc, wave, t0_ & 63);     // conv2's first pass: in flight under conv1
  int img = blockIdx.x;
  do {      // (LOOP = false: `while (false)` -- no loop in the instance, r04's straight-line code)
    // LOOP: the thread index and the LDS base are made opaque once per image, so that every lane-dependent address is formed where it is
    // used, as in the straight-line kernel -- hoisted out of the image loop they stay live across all five layers and the k loops spill
    // (measured: 113 against 98 us per image)
    int t = t0_;
    float *ldsb = lds;
    if constexpr (LOOP) { asm volatile("" : "+v"(t)); asm volatile("" : "+v"(ldsb)); }
    const int lane = t & 63;
    float *A1 = ldsb, *T2 = ldsb, *A2 = ldsb + CR_A2, *A3 = ldsb, *T4 = ldsb + CR_T4, *A4 = ldsb, *T5 = ldsb + CR_T5, *IMG = ldsb + CR_IMG;
    chain_bias<28, 32>(a.b[0], bv, wave, lane);
    chain_sync();                                              // IMG is complete (LOOP: and the previous image's readers are done)
    CH_STAMP(1);
    {   // conv1 1 -> 32 @28
        floatx4 acc[ChainGeo<28, 32>::NSLOT];
        chain_conv1_mfma(IMG, wa, acc, wave, lane);
        chain_store<28, 32, true>(acc, bv, A1, wave, lane);
        chain_zero_halo<28, 32>(A1, wave, lane);
    }
    chain_sync();
    CH_STAMP(2);
    {   // conv2 32 -> 32 @28 + pool
        floatx4 acc[ChainGeo<28, 32>::NSLOT];
        chain_bias<28, 32>(a.b[1], bv, wave, lane);
        chain_mfma<28, 32, 32, 14, 32, 64>(A1, a.w[1], wc, acc, wave, lane, a.w[2]);   // (+ conv3's first pass, requested during the last pass)
        chain_sync();
        CH_STAMP(3);                                    // every wave is done reading A1
        chain_store<28, 32, false>(acc, bv, T2, wave, lane);
        chain_zero_halo<14, 32>(A2, wave, lane);
        chain_sync();
        CH_STAMP(4);
        chain_pool<28, 32, false>(T2, A2, t);
        chain_sync();
        CH_STAMP(5);
    }
    {   // conv3 32 -> 64 @14
        floatx4 acc[ChainGeo<14, 64>::NSLOT];
        chain_zero_halo<14, 64>(A3, wave, lane);                     // (T2 is dead; A3's interior is written after the k loop's barrier)
        chain_bias<14, 64>(a.b[2], bv, wave, lane);
        chain_mfma<14, 32, 64, 14, 64, 64>(A2, a.w[2], wc, acc, wave, lane, a.w[3]);
        chain_sync();
        CH_STAMP(6);
        chain_store<14, 64, true>(acc, bv, A3, wave, lane);
        chain_sync();
        CH_STAMP(7);
    }
    {   // conv4 64 -> 64 @14 + pool
        floatx4 acc[ChainGeo<14, 64>::NSLOT];
        chain_bias<14, 64>(a.b[3], bv, wave, lane);
        chain_mfma<14, 64, 64, 7, 64, 128>(A3, a.w[3], wc, acc, wave, lane, a.w[4]);
        chain_sync();                                    // every wave is done reading A3
        CH_STAMP(8);
        chain_store<14, 64, false>(acc, bv, T4, wave, lane);
        chain_zero_halo<7, 64>(A4, wave, lane);
        chain_sync();
        CH_STAMP(9);
        chain_pool<14, 64, false>(T4, A4, t);
        chain_sync();
        CH_STAMP(10);
    }
    {   // conv5 64 -> 128 @7 + global average pool
        floatx4 acc[ChainGeo<7, 128>::NSLOT];
        chain_bias<7, 128>(a.b[4], bv, wave, lane);
        float px[2] = {0.f, 0.f};                                // LOOP: this thread's pixels t, t + 512 of the workgroup's next image
        const int nimg = img + (int)gridDim.x;
        if constexpr (LOOP) {
            if (nimg < a.n) {
                const float *xn = a.x + (long)nimg * 784;
                px[0] = xn[t];
                px[1] = t + CH_NT < 784 ? xn[t + CH_NT] : 0.f;
            }
            chain_mfma<7, 64, 128, 28, 32, 32>(A4, a.w[4], wc, acc, wave, lane, a.w[1]);   // (+ conv2's first pass for the next image)
        } else {
            chain_mfma<7, 64, 128>(A4, a.w[4], wc, acc, wave, lane);
        }
        chain_store<7, 128, false>(acc, bv, T5, wave, lane);     // (T5 does not overlap A4)
        ChainMlp3W hw;
        if constexpr (HEAD3) chain_mlp3_weights(a.head3, hw, t);  // the classifier's weights: in flight under the plane means
        if constexpr (LOOP) {
            if (nimg < a.n) {
                IMG[(t / 28 + 1) * 30 + t % 28 + 1] = px[0];
                if (t + CH_NT < 784) IMG[((t + CH_NT) / 28 + 1) * 30 + (t + CH_NT) % 28 + 1] = px[1];
            }
        }
        chain_sync();
        CH_STAMP(11);
        // 16 lanes per channel plane, lane l adds elements l, l + 16, ... and a shuffle tree joins them: avgpool_global16_kernel's arithmetic
        constexpr int LD = ch_tile_ld(49);
        const int l = t & 15;
        for (int c = t >> 4; c < 128; c += CH_NT / 16) {
            const float *row = T5 + c * LD;
            float sum = 0.f, k = 0.f;
            for (int i = l; i < 49; i += 16) {
                const float v = row[i];
                sum += v;
                k += v > 0.f ? 1.f : 0.f;
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                sum += __shfl_down(sum, off, 16);
                k += __shfl_down(k, off, 16);
            }
            if (l == 0) {
                a.y[(long)img * 128 + c] = sum / 49.0f;
                if (a.cnt) a.cnt[(long)img * 128 + c] = k;
                if constexpr (HEAD3) A4[c] = sum / 49.0f;         // (A4 is dead: every wave is behind the barrier above)
            }
        }
        if constexpr (HEAD3) {
            chain_sync();
            CH_STAMP(30);                                    // (13 .. 16, 22 .. 29 are the k loops' pass stamps)
            chain_mlp3_rows(a.head3, hw, A4, A4 + 128, img, t);
        }
#ifdef TH_PROFILE
        chain_sync();
#endif
        CH_STAMP(12);
        CH_CLK(21);
    }
  } while (LOOP && (img += (int)gridDim.x) < a.n);
#endif
}

// ---- classifier rows behind a chain that ends in a pooled map: one image = one row of the batch ------------------------------------------------
// Forward, loss term, dlogits and dX of a row depend on that row only (nn.rs:54-60, loss.rs:101-195, ops.rs:254-265): the workgroup that
// holds the image's flattened map XM [k] in LDS forms its `classes` logits (thread t owns elements t, t + 512, ...: the W reads are
// perfectly coalesced, [classes][k] is 125 KB and L2-resident), the softmax / NLL / hit / dlogits of the row (tail_row_softmax: the
// arithmetic of every fused head of this library), then dX[k] = sum_c dl[c] W[c][k] from the SAME weight registers, masked by x > 0, and
// its per-channel sums -- what a bias-only Conv2dReLU + max-pool in front needs of its gradient (every pooled gradient lands on exactly one
// conv output whose ReLU mask is "pooled value > 0", tensor.rs:1496-1519).  What is left for a second launch are the sums over the
// batch: dW = dl^T X, db, the loss, the conv bias (wide_head.hip: wide_grads_kernel).
// One buffer_load per value and nothing else: the lane's byte offset 4 t is the only vector operand, (c k + 512 j) 4 is a scalar; rows past
// `classes` lie past the descriptor's range and read as zeros; the last round's elements past k are selected to zero (they would read the
// next class's row).
template <int NJ, int NC, int KC>
__device__ __forceinline__ void chain_head_weights(const ChainHeadArgs &h, float (&wv)[NJ][NC], int t) {
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void *)h.w, 0, h.classes * KC * 4, 0x00020000);
    const int vo = t * 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, vo, (c * KC + CH_NT * j) * 4, 0));
            wv[j][c] = (CH_NT * (j + 1) <= KC || t + CH_NT * j < KC) ? v : 0.f;
        }
    }
}

// the same as 16-byte requests: thread t owns elements k = 4 t + 2048 j + v (v = 0..3) -- wv[4 j + v][c]; a quarter of the requests (a
// workgroup's 560 dword requests kept the address unit busy for ~1.2 us in front of conv2's k loop).  Plain global loads: this
// compiler's __builtin_amdgcn_raw_buffer_load_b128 loads ONE dword and splats it (checked in the listing), and an inline-assembly load
// would leave 80 registers written behind the register allocator's back for the length of the k loop.
template <int NJ4, int NC, int KC>
__device__ __forceinline__ void chain_head_weights4(const ChainHeadArgs &h, float (&wv)[4 * NJ4][NC], int t) {
    static_assert(KC % 4 == 0, "whole quads");
#pragma unroll
    for (int j = 0; j < NJ4; ++j) {
        const bool in = 4 * CH_NT * (j + 1) <= KC || 4 * t + 4 * CH_NT * j < KC;
        const float *row = h.w + 4 * t + 4 * CH_NT * j;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 q = (in && c < h.classes) ? *reinterpret_cast<const float4 *>(row + (long)c * KC) : make_float4(0.f, 0.f, 0.f, 0.f);
            wv[4 * j][c] = q.x; wv[4 * j + 1][c] = q.y; wv[4 * j + 2][c] = q.z; wv[4 * j + 3][c] = q.w;
        }
    }
}

template <int NJ, int NC>
__device__ __forceinline__ void chain_head_weights_rt(const ChainHeadArgs &h, float (&wv)[NJ][NC], int t) {   // the same with a run-time k
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void *)h.w, 0, h.classes * h.k * 4, 0x00020000);
    const int vo = t * 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, vo, (c * h.k + CH_NT * j) * 4, 0));
            wv[j][c] = t + CH_NT * j < h.k ? v : 0.f;
        }
    }
}

// xm: LDS [k] (the map; overwritten by the masked dX), red: LDS [NC][CH_NT] + 64 floats.  Ends without a barrier (global stores only).
// V = 1: thread t owns elements t + 512 j (chain_head_weights); V = 4: elements 4 t + 2048 j' + v, wv[4 j' + v] (chain_head_weights4; NJ = 4 NJ4)
template <int NJ, int NC, int V = 1>
__device__ __forceinline__ void chain_head_rows(const ChainHeadArgs &h, const float (&wv)[NJ][NC], float *xm, float *red, int img, int t) {
    static_assert(V == 1 || (V == 4 && NJ % 4 == 0), "dword or quad ownership");
    auto k_of = [&](int j) { return V == 1 ? t + CH_NT * j : 4 * t + 4 * CH_NT * (j >> 2) + (j & 3); };
    static_assert(NC <= 16 && NC * 32 <= CH_NT, "one 32-lane half-wave per class in the reduction");
    const int lane = t & 63, r16 = lane & 15, g4 = lane >> 4;
    float *sc = red + NC * CH_NT;            // [0..15] logits, [16..31] dlogits
    float xv[NJ], part[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) part[c] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int k = k_of(j);
        if (V == 4) {
            if ((j & 3) == 0) {
                const float4 q = k < h.k ? *reinterpret_cast<const float4 *>(xm + k) : make_float4(0.f, 0.f, 0.f, 0.f);   // (h.k % 4 == 0)
                xv[j] = q.x; xv[j + 1] = q.y; xv[j + 2] = q.z; xv[j + 3] = q.w;
            }
        } else {
            xv[j] = k < h.k ? xm[k] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) part[c] = fmaf(xv[j], wv[j][c], part[c]);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) red[c * CH_NT + t] = part[c];
    chain_sync();
    if (t < NC * 32) {                       // class t / 32: lane i adds threads i, i + 32, ... in order, then a fixed tree
        const int c = t >> 5, i = t & 31;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < CH_NT / 32; ++e) s += red[c * CH_NT + i + 32 * e];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 32);
        if (i == 0) sc[c] = s + ((h.bias && c < h.classes) ? h.bias[c] : 0.f);   // nn.rs:54-60: bias after the product
    }
    chain_sync();
    CH_STAMP(7);
    float lg[4], dl[4], nll;
    int bi;
#pragma unroll
    for (int i = 0; i < 4; ++i) lg[i] = (g4 * 4 + i < h.classes && g4 * 4 + i < NC) ? sc[g4 * 4 + i] : -INFINITY;
    const float tf = h.targets[img];
    tail_row_softmax(lg, g4, h.classes, tf, h.inv_b, dl, nll, bi);   // every wave, redundantly: no exchange, no extra barrier
    if (t < 64 && r16 == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sc[16 + g4 * 4 + i] = dl[i];
            h.dl[(long)img * 16 + g4 * 4 + i] = dl[i];
        }
        if (g4 == 0) {
            h.rowstat[(long)img * 2] = nll;
            h.rowstat[(long)img * 2 + 1] = fabsf((float)bi - tf) < 1e-6f ? 1.f : 0.f;   // loss.rs:283
        }
    }
    if (!h.cbpart) return;                   // (uniform)
    chain_sync();
    CH_STAMP(8);
    float d16[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) d16[c] = sc[16 + c];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int k = k_of(j);
        float dx = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) dx = fmaf(d16[c], wv[j][c], dx);      // ops.rs:254-265, class order
        if (k < h.k) xm[k] = xv[j] > 0.f ? dx : 0.f;
    }
    chain_sync();
    CH_STAMP(9);
    for (int ch = t >> 4; ch < h.c_last; ch += CH_NT / 16) {                // 16 lanes per channel
        const float *row = xm + ch * h.hw;
        float p = 0.f;
        for (int q = r16; q < h.hw; q += 16) p += row[q];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) p += __shfl_xor(p, off, 16);
        if (r16 == 0) h.cbpart[(long)img * h.c_last + ch] = p;
    }
    CH_STAMP(10);
}

// ---- the simple CNN's front (examples/train_mnist_cnn.rs:64-100): 1 -> 32 + pool, 32 -> 64 + pool -> [64][7][7] ----
// Both layers are pooled: their pixel tiles are 2x2 windows (chain_lane_pixel<S, true>) and the maxima leave the accumulators directly
// (chain_store_pooled) -- no full-resolution tile in LDS, no pooling pass, two barriers fewer and 53 KB of LDS instead of 140 than
// r03's tile-and-pool form; time-neutral at batch 256 (workgroup 24.5 against 24.6 us: the ~300 VALU instructions per wave it adds
// cost what the tile pass did).
//   A [32][272] @0  conv2's input (conv1's pooled output)    IMG [900] @8704
//   HEAD: XM [3136] @9604 (the flattened pooled map), RED [NC][512] + 64 @0 (over A, dead after conv2's k loop)
constexpr int CS_IMG = 32 * ch_cis(16), CS_XM = CS_IMG + 900, CS_K = 64 * 49, CS_LDS = CS_XM + CS_K;
[[maybe_unused]] constexpr int CS_NJ = (CS_K + CH_NT - 1) / CH_NT;
static_assert(16 * CH_NT + 64 <= CS_IMG, "LDS map: the classifier's reduction fits over A");

// LOOP (batches above one image per CU, r05): as conv_chain_reference_kernel<true> -- the workgroups walk the images, the next image's pixels
// cross the fabric under conv2's k loop, conv2's first weight pass rides in its own last pass, and the classifier's weight registers are
// loaded ONCE per workgroup instead of once per image (125 KB of L2 reads each).
#ifndef CH_LOOP_KEEP_HEAD
#define CH_LOOP_KEEP_HEAD 0      // 1: the walking form keeps the classifier's weight registers across its images (measured: no gain -- 80 more live registers through conv1)
#endif
// LEAN (r06; batches with at least two images per CU): the same arithmetic within 128 registers per lane, so that TWO workgroups share a CU
// -- 51 KB of LDS each -- and one's image load, conv1, pooling and classifier row run under the other's k loop.  What pays for the
// registers: the classifier's weights are requested AFTER conv2's k loop instead of under it (80 registers per lane through the loop; the
// wait they now expose is the other workgroup's to fill).
template <bool HEAD, int NC, bool LOOP = false, bool LEAN = false>
__global__ __launch_bounds__(CH_NT, LEAN ? 4 : 1) void conv_chain_simple_kernel(ConvChainArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t0_ = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t0_ >> 6);
    ChainBias bv;
    ChainW wc;
    ChainW9 wc9;
    ChainW1 wa;
    static_assert(!(LEAN && LOOP), "the two-to-a-CU instance does not walk");
    CH_STAMP(0);
    CH_CLK(20);
    chain_conv1_weights(a.w[0], wa, wave, t0_ & 63);
    chain_load_image(a.x + (long)blockIdx.x * 784, lds + CS_IMG, t0_);
    if constexpr (LEAN) chain_weights9<14, 32, 64>(a.w[1], 0, wc9, wave, t0_ & 63);   // conv2's first half pass: in flight under conv1
    else chain_weights<14, 32, 64>(a.w[1], 0, wc, wave, t0_ & 63);      // conv2's first pass: in flight under conv1
    if (HEAD && a.head.tick && blockIdx.x == 0 && t0_ == 0) chain_head_tick(a.head);
    constexpr int HV = NC <= 10 ? 4 : 1, HNJ = HV == 4 ? 4 * ((CS_K + 4 * CH_NT - 1) / (4 * CH_NT)) : CS_NJ;   // (16 classes: 128 registers as quads)
    float hw_[HNJ][NC];
    if constexpr (HEAD && LOOP && CH_LOOP_KEEP_HEAD) {           // the classifier's weights, once for all of this workgroup's images
        if constexpr (HV == 4) chain_head_weights4<HNJ / 4, NC, CS_K>(a.head, hw_, t0_);
        else chain_head_weights<CS_NJ, NC, CS_K>(a.head, hw_, t0_);
    }
  int img = blockIdx.x;
  do {      // (LOOP = false: `while (false)` -- no loop in the instance, r04's straight-line code)
    int t = t0_;                                                 // LOOP: opaque once per image (conv_chain_reference_kernel)
    float *ldsb = lds;
    if constexpr (LOOP) { asm volatile("" : "+v"(t)); asm volatile("" : "+v"(ldsb)); }
    const int lane = t & 63;
    float *A = ldsb, *IMG = ldsb + CS_IMG, *XM = ldsb + CS_XM;
    chain_bias<28, 32>(a.b[0], bv, wave, lane);
    chain_zero_halo<14, 32>(A, wave, lane);                      // (LOOP: the previous image's classifier reduction lay over A)
    chain_sync();
    CH_STAMP(1);
    {   // conv1 1 -> 32 @28, pooled -> A's interior
        floatx4 acc[ChainGeo<28, 32>::NSLOT];
        chain_conv1_mfma<true>(IMG, wa, acc, wave, lane);
        CH_STAMP(2);
        chain_store_pooled<28, 32, false>(acc, bv, A, wave, lane);
        chain_sync();
        CH_STAMP(3);
    }
    floatx4 acc[ChainGeo<14, 64>::NSLOT];
    chain_bias<14, 64>(a.b[1], bv, wave, lane);
    if constexpr (HEAD && !LEAN && !(LOOP && CH_LOOP_KEEP_HEAD)) {        // the classifier's weights: in flight under conv2's k loop
        if constexpr (HV == 4) chain_head_weights4<HNJ / 4, NC, CS_K>(a.head, hw_, t);
        else chain_head_weights<CS_NJ, NC, CS_K>(a.head, hw_, t);
    }
    float px[2] = {0.f, 0.f};                                    // LOOP: this thread's pixels t, t + 512 of the workgroup's next image
    const int nimg = img + (int)gridDim.x;
    if constexpr (LOOP) {
        if (nimg < a.n) {
            const float *xn = a.x + (long)nimg * 784;
            px[0] = xn[t];
            px[1] = t + CH_NT < 784 ? xn[t + CH_NT] : 0.f;
        }
        chain_mfma<14, 32, 64, 14, 32, 64, true>(A, a.w[1], wc, acc, wave, lane, a.w[1]);   // (+ its own first pass again, for the next image)
        if (nimg < a.n) {                                        // IMG is nobody's behind conv1
            IMG[(t / 28 + 1) * 30 + t % 28 + 1] = px[0];
            if (t + CH_NT < 784) IMG[((t + CH_NT) / 28 + 1) * 30 + (t + CH_NT) % 28 + 1] = px[1];
        }
    } else if constexpr (LEAN) {
        chain_mfma9<14, 32, 64, true>(A, a.w[1], wc9, acc, wave, lane);
    } else {
        chain_mfma<14, 32, 64, 0, 0, 0, true>(A, a.w[1], wc, acc, wave, lane);
    }
    CH_STAMP(4);
    CH_STAMP(5);
    chain_store_pooled<14, 64, true>(acc, bv, a.y + (long)img * 64 * 49, wave, lane, HEAD ? XM : nullptr);   // (XM does not overlap A)
    if constexpr (HEAD && LEAN) {                                // (requested here: the registers were the k loop's until now)
        if constexpr (HV == 4) chain_head_weights4<HNJ / 4, NC, CS_K>(a.head, hw_, t);
        else chain_head_weights<CS_NJ, NC, CS_K>(a.head, hw_, t);
    }
    if constexpr (HEAD) {
        chain_sync();                                            // XM complete; every wave is past conv2's k loop: A is free
        CH_STAMP(6);
        chain_head_rows<HNJ, NC, HV>(a.head, hw_, XM, A, img, t);
    }
#ifdef TH_PROFILE
    chain_sync();
#endif
    CH_STAMP(11);
    CH_CLK(21);
    if constexpr (LOOP) chain_sync();                            // the classifier's reduction (over A) and XM are read: the next image may write them
  } while (LOOP && (img += (int)gridDim.x) < a.n);
#endif
}

// the image's c0 planes [s][s] -> zero-haloed LDS planes [c0][CIS]: a wave per plane row, a lane per column (no per-element division; s + 2 <= 64)
__device__ __forceinline__ void rt_load_planes(const float *__restrict__ xi, float *planes, int c0, int s, int t) {
    const int wp = s + 2, cis = ch_cis(wp), q = (t & 63) - 1, r0 = t >> 6;
    if (q + 1 >= wp) return;
    for (int c = 0; c < c0; ++c)
        for (int r = r0; r < wp; r += CH_NT / 64) {
            const int ri = r - 1;
            planes[c * cis + r * wp + q + 1] = (ri >= 0 && ri < s && q >= 0 && q < s) ? xi[(c * s + ri) * s + q] : 0.f;
        }
}

// ==== ONE Conv2dReLU layer with the chain's compiled tile mappings (r04) ==========================================================================
// The four batch-256 layers of the reference CNN -- 32 -> 32 @28, 32 -> 64 @14, 64 -> 64 @14, 64 -> 128 @7 -- as launches of their own
// (th_conv3x3_fwd / th_conv3x3_pool2_fwd / th_conv3x3_gap_fwd: forward() outside a Trainer step, TAPER_CONV_CHAIN=0, full_backward) ran
// through conv3x3_img_kernel at 0.42-0.57 of the matrix peak while the same layers inside the chain ran at 0.67: same contraction, but there
// a wave owns a PAIR of channel tiles per pixel tile (every pixel operand read feeds two MFMAs), tap offsets are immediates and the weight
// operands come from L2 a pass ahead instead of an LDS slab.  This kernel is the chain's stage on its own: a workgroup loads ONE image's
// planes into the padded LDS form, runs chain_mfma with the compiled mapping, and writes the map (POST 0), its 2x2 maxima (1) or its plane
// means + positive counts (2).  Same k order, bias after the sum, ReLU, strict-> maxima, 16-lane plane sums: the bits of the layered path.
struct LayerChainArgs {
    const float *x, *w, *b;    // [n][C_IN][S][S]; taper layout [9 C_IN][C_OUT]; bias [C_OUT] (nullable)
    float *y, *cnt;            // POST 0: [n][C_OUT][S][S]; 1: [n][C_OUT][S/2][S/2]; 2: [n][C_OUT] (+ cnt [n][C_OUT], nullable)
    int n;
};

// one image's C_IN planes [S][S] -> the interior of the padded LDS planes: consecutive threads take consecutive elements (coalesced), all
// of a thread's loads go out before its first LDS store (a load per loop iteration is a chain of ~1 us round trips: 32 planes of 28 x 28
// took longer than their k loop that way), the (channel, row, column) of an element by compile-time divisions
template <int S, int C_IN>
__device__ __forceinline__ void layer_load_planes(const float *__restrict__ xi, float *planes, int t) {
    constexpr int WP = S + 2, CIS = ch_cis(WP), N = C_IN * S * S, PER = (N + CH_NT - 1) / CH_NT, U = 7;
    constexpr int UB = PER < U ? PER : U;       // (a last, partial batch is covered by the e < N guards)
#pragma unroll 1
    for (int j0 = 0; j0 < PER; j0 += UB) {
        float v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int e = t + CH_NT * (j0 + u);
            v[u] = e < N ? xi[e] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int e = t + CH_NT * (j0 + u);
            if (e < N) {
                const int c = e / (S * S), rem = e - c * (S * S), r = rem / S, q = rem - r * S;
                planes[c * CIS + (r + 1) * WP + q + 1] = v[u];
            }
        }
    }
}

#ifndef TH_LC_STAGE_ALL
#define TH_LC_STAGE_ALL 0     // 1: every map-writing layer goes through the LDS tile (measurement)
#endif
#ifndef TH_LC_STAGE_NONE
#define TH_LC_STAGE_NONE 0    // 1: no layer does (r04's direct stores)
#endif
#ifndef TH_LC_NT
#define TH_LC_NT 0            // 1: the staged map is written with non-temporal stores (measurement)
#endif
#if TH_LC_NT
#define LC_STORE4(p_, v_) __builtin_nontemporal_store((v_), (p_))
#else
#define LC_STORE4(p_, v_) (*(p_) = (v_))
#endif
// LIN (POST 0 only): the map is the plain sum -- no bias, no ReLU: the INPUT GRADIENT of a 3x3 layer, whose mirrored, channel-swapped filter
// th_conv3x3_bwd_input has laid out as a taper slab (full_backward extension)
template <int S, int C_IN, int C_OUT, int POST, bool LIN = false>
__global__ __launch_bounds__(CH_NT, 1) void conv_layer_chain_kernel(LayerChainArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using G = ChainGeo<S, C_OUT>;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l16 = lane & 15, g4 = lane >> 4;
    float *A = lds, *T = lds;                    // the output tile overlays the input planes once every wave is past the k loop
    static_assert(!LIN || POST == 0, "the linear form writes the map");
    constexpr bool LC_STAGED = !LIN && POST == 0 && (TH_LC_STAGE_ALL || (S == 7 && !TH_LC_STAGE_NONE));
    ChainBias bv;
    chain_bias<S, C_OUT>(LIN ? nullptr : a.b, bv, wave, lane);
    if (POST == 0) chain_zero_halo<S, C_IN>(A, wave, lane);             // (the map goes to memory: nothing ever overwrites the halo)
    CH_SPAN(0);
    for (int img = blockIdx.x; img < a.n; img += gridDim.x) {
        ChainW wc;
        CH_STAMP(0);
        chain_weights<S, C_IN, C_OUT>(a.w, 0, wc, wave, lane);          // the first pass: in flight under the image load
        if (POST != 0) chain_zero_halo<S, C_IN>(A, wave, lane);         // (the output tile overlaid the planes)
        layer_load_planes<S, C_IN>(a.x + (long)img * C_IN * S * S, A, t);
        chain_sync();
        CH_STAMP(1);
        floatx4 acc[G::NSLOT];
        chain_mfma<S, C_IN, C_OUT>(A, a.w, wc, acc, wave, lane);
        CH_STAMP(2);
        if (POST == 0 && LC_STAGED) {
            // the NCHW map through the LDS tile: an image's map [C_OUT][PX] is ONE contiguous block, written here as whole float4 lines
            // (16 dword stores per lane straight from the accumulators are 64-byte pieces at a 4 PX-byte pitch -- partial lines, and at
            // 7 x 7 never aligned: that store phase measured 4.6 us of the 18 a workgroup lives, r05)
            chain_sync();                                               // every wave is done reading A
            chain_store<S, C_OUT, false>(acc, bv, T, wave, lane);
            chain_sync();
            constexpr int LD = ch_tile_ld(G::PX), NQ = C_OUT * G::PX / 4;
            static_assert(C_OUT * G::PX % 4 == 0, "whole float4s");
            float4 *yq = reinterpret_cast<float4 *>(a.y + (long)img * C_OUT * G::PX);
#pragma unroll 2
            for (int f = t; f < NQ; f += CH_NT) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int el = 4 * f + e, c = el / G::PX, i = el - c * G::PX;
                    v[e] = T[c * LD + i];
                }
                LC_STORE4(yq + f, make_float4(v[0], v[1], v[2], v[3]));
            }
            if (img + (int)gridDim.x < a.n) {
                chain_sync();                                           // the next image's planes overwrite T
                chain_zero_halo<S, C_IN>(A, wave, lane);
            }
            CH_STAMP(3);
            CH_STAMP(4);
            CH_SPAN(1);
            continue;
        }
        if (POST == 0) {
            // the NCHW map straight from the accumulators
            float *ymap = a.y + (long)img * C_OUT * G::PX;
#pragma unroll
            for (int k = 0; k < G::NSLOT; ++k) {
                if (!G::slot_live(wave, k)) continue;
                const int p = G::slot_px(wave, k) * 16 + l16;
                if (p >= G::PX) continue;
                const bool second = k < 2 * G::ND && (k & 1);
                float *o = ymap + (long)(16 * G::slot_ch(wave, k) + 4 * g4) * G::PX + p;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[k][e] + (second ? bv.b[e] : bv.a[e]);
                    o[e * G::PX] = LIN ? acc[k][e] : (v > 0.f ? v : 0.f);
                }
            }
            chain_sync();                                               // the next image's planes overwrite A
            CH_STAMP(3);
            CH_STAMP(4);
            CH_SPAN(1);
            continue;
        }
        chain_sync();                                                   // every wave is done reading A
        CH_STAMP(3);
        chain_store<S, C_OUT, false>(acc, bv, T, wave, lane);
        chain_sync();
        if constexpr (POST == 1 && S % 2 == 0) {
            chain_pool<S, C_OUT, true>(T, a.y + (long)img * C_OUT * (S / 2) * (S / 2), t);
        } else if constexpr (POST == 2) {
            constexpr int LD = ch_tile_ld(S * S);
            for (int c = t >> 4; c < C_OUT; c += CH_NT / 16) {             // avgpool_global16_kernel's arithmetic
                const float *row = T + c * LD;
                float sum = 0.f, k = 0.f;
                for (int i = l16; i < S * S; i += 16) {
                    const float v = row[i];
                    sum += v;
                    k += v > 0.f ? 1.f : 0.f;
                }
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) {
                    sum += __shfl_down(sum, off, 16);
                    k += __shfl_down(k, off, 16);
                }
                if (l16 == 0) {
                    a.y[(long)img * C_OUT + c] = sum / (float)(S * S);
                    if (a.cnt) a.cnt[(long)img * C_OUT + c] = k;
                }
            }
        }
        chain_sync();
        CH_STAMP(4);
    }
    CH_SPAN(1);
#endif
}

// ==== chains described at run time ==================================================================================================================
// Any run of Conv2dReLU(3x3, stride 1, pad 1) [+ MaxPool2d(2) | + global average pool] stages whose maps fit the 160 KB of LDS, with the
// same structure as the two compiled instances above -- a workgroup carries one image through every stage, the maps live in LDS in the
// padded [channel][rows + 2][cols + 2] form, weights come from L2 a pass ahead -- but with the channel counts, the tile-to-wave mapping and
// the LDS plan as kernel ARGUMENTS (RtStage, filled by rt_plan on the host).  Compiled in: the map sizes a 28 x 28 input and its pools produce
// (S = 28, 14, 7: the tap part of an operand address stays an immediate) -- any other square map up to 32 x 32 takes the instance with the
// size as a run-time value (S = 0: one address add per operand read) --, and the number of "double" pixel tiles a wave
// carries per round (ND = 1, 2, 4, 7).  A wave owns ONE pair of channel tiles per round (chA = 2 q, chB = 2 q + 1: every pixel operand read
// feeds two MFMAs) and ND pixel tiles (piece + m i): chunk (q, piece) = round * 8 + wave.  Same per-output arithmetic as the compiled
// instances and the layer-by-layer kernels (k order, bias after the sum, ReLU, strict-> maxima, the 16-lane plane sums).
constexpr int RT_MAX_STAGES = 8, RT_LDS_FLOATS = 40960;   // 160 KB

struct RtStage {
    const float *w, *b;            // taper layout [9 c_in][c_out], bias [c_out]
    int c_in, c_out, s, post;      // s: the conv's map size (input = output); post: TH_CHAIN_*
    int in_off, out_off, pool_off; // LDS offsets (floats): input planes; output planes (post none) or plain tile (pooled / averaged); pooled planes
    int nd, m, rounds;             // doubles per chunk, chunks per channel pair, rounds of 8 chunks
};

struct RtChainArgs {
    const float *x;                // [n][c0][s0][s0]
    float *y, *cnt;
    int n, n_stages, c0, s0;
    int has_head, xm_off, red_off; // the classifier rows (chain_head_rows): LDS offsets of the flattened map and the reduction scratch
    RtStage st[RT_MAX_STAGES];
    ChainHeadArgs head;
};

// S = 0 in the templates below: the map size is the run-time argument s_rt (any size up to 32 -- the tap part of an operand address is then a
// register, one address add per operand read, instead of the read's immediate offset); S > 0: compiled in, s_rt ignored
template <int S>
__device__ __forceinline__ void rt_zero_halo(float *planes, int c, int wave, int lane, int s_rt) {
    const int SS = S ? S : s_rt, WP = SS + 2, CIS = ch_cis(WP), RING = 4 * SS + 4;
    int pos[3];                                  // RING <= 132
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int r = lane + 64 * j;
        pos[j] = r < WP ? r : r < 2 * WP ? (WP - 1) * WP + (r - WP) : r < 2 * WP + SS ? (r - 2 * WP + 1) * WP : r < RING ? (r - 2 * WP - SS + 1) * WP + WP - 1 : -1;
    }
    for (int ch = wave; ch < c; ch += CH_NT / 64) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (pos[j] >= 0) planes[ch * CIS + pos[j]] = 0.f;
    }
}

// weight operands of pass cb for the pair (chA, chB): one dword per lane and k-step each (chain_weights with run-time sizes)
__device__ __forceinline__ void rt_weights(const float *__restrict__ w, int c_in, int c_out, int cb, int chA, int chB, ChainW &wr, int lane) {
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, 9 * c_in * c_out * 4, 0x00020000);
    const int row = ((lane >> 4) * 9 * c_out + (lane & 15)) * 4, va = row + 64 * chA, vb = row + 64 * chB;
#pragma unroll
    for (int s = 0; s < 18; ++s) {
        const int so = ((cb + 4 * (s / 9)) * 9 + s % 9) * c_out * 4;
        wr.a[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, va, so, 0));
        wr.b[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, vb, so, 0));
    }
}

// the k loop of one chunk: in = padded planes [c_in][CIS]; acc[2 i], acc[2 i + 1] = pixel tile ptile[i] x (chA, chB)
template <int S, int ND>
__device__ __forceinline__ void rt_mfma(const float *in, const float *__restrict__ w, int c_in, int c_out, int chA, int chB, const int (&ptile)[ND],
                                        floatx4 (&acc)[2 * ND], int lane, int s_rt, const ChainW *w0) {
    constexpr int KS = 18;
    const int SS = S ? S : s_rt, WP = SS + 2, CIS = ch_cis(WP), PX = SS * SS, NPT = (PX + 15) / 16;
    // a wave's tiles are piece, piece + m, ...: the waves of a channel pair carry ND or ND - 1 of them -- the last double's MFMAs are
    // skipped (a wave-uniform branch per k-step; its operand read stays unconditional) where the tile does not exist
    const bool last_live = ptile[ND - 1] < NPT;
    const int l16 = lane & 15, g4 = lane >> 4;
    typedef __attribute__((address_space(3))) const float lds_cf;
    lds_cf *pt[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        int p = ptile[i] * 16 + l16;
        if (p >= PX) p = 0;        // lanes / tiles past the image compute on pixel 0 and are never stored
        pt[i] = (lds_cf *)(in + (p / SS) * WP + p % SS + g4 * CIS);
        asm volatile("" : "+v"(pt[i]));
    }
#pragma unroll
    for (int k = 0; k < 2 * ND; ++k) acc[k] = floatx4{0.f, 0.f, 0.f, 0.f};
    ChainW wc, wn;
    if (w0) wc = *w0;                              // requested before the previous stage's epilogue (rt_stage_nd)
    else rt_weights(w, c_in, c_out, 0, chA, chB, wc, lane);
#define RT_REQ(B, SS) { _Pragma("unroll") for (int i = 0; i < ND; ++i) B[i] = pt[i][(4 * ((SS) / 9)) * CIS + (((SS) % 9) / 3) * WP + ((SS) % 9) % 3]; }
#define RT_REQ1(B, SS, I) { B[I] = pt[I][(4 * ((SS) / 9)) * CIS + (((SS) % 9) / 3) * WP + ((SS) % 9) % 3]; }
#define RT_WREQ(DST, RS, V, SO) { DST = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(RS, V, SO, 0)); }
    /* as chain_mfma: one read of the next step behind each pair of MFMAs; the next pass's weight requests two per k-step behind the */
    /* step's first pairs (the pass after the last one: a zero-sized buffer, no branch)                                              */
#define RT_PASS_BODY(WCUR, WNXT, RS, SOBASE)                                                                                        \
    {                                                                                                                               \
        float b0[ND], b1[ND];                                                                                                       \
        RT_REQ(b0, 0)                                                                                                               \
        _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                                            \
            const int so_ = (SOBASE) + ((4 * (s / 9)) * 9 + s % 9) * rowb;                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
            _Pragma("unroll") for (int i = 0; i < ND - 1; ++i) {                                                                    \
                acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(WCUR.a[s], b0[i], acc[2 * i], 0, 0, 0);                           \
                acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(WCUR.b[s], b0[i], acc[2 * i + 1], 0, 0, 0);                   \
                __builtin_amdgcn_sched_barrier(0);                                                                                  \
                if (s + 1 < KS) RT_REQ1(b1, s + 1, i)                                                                               \
                if (i == 0) RT_WREQ(WNXT.a[s], RS, va, so_)                                                                         \
                if (i == 1) RT_WREQ(WNXT.b[s], RS, vb, so_)                                                                         \
                __builtin_amdgcn_sched_barrier(0);                                                                                  \
            }                                                                                                                       \
            if (last_live) {                                                                                                        \
                acc[2 * ND - 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(WCUR.a[s], b0[ND - 1], acc[2 * ND - 2], 0, 0, 0);            \
                acc[2 * ND - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(WCUR.b[s], b0[ND - 1], acc[2 * ND - 1], 0, 0, 0);            \
            }                                                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
            if (s + 1 < KS) RT_REQ1(b1, s + 1, ND - 1)                                                                              \
            if (ND - 1 == 0) RT_WREQ(WNXT.a[s], RS, va, so_)                                                                        \
            if (ND - 1 <= 1) RT_WREQ(WNXT.b[s], RS, vb, so_)                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
            if (s + 1 < KS) { _Pragma("unroll") for (int i = 0; i < ND; ++i) b0[i] = b1[i]; }                                       \
        }                                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < ND; ++i) pt[i] += 8 * CIS;                                                            \
    }
    /* (a copy of the pass per value of last_live, as chain_mfma has for its single tile, was tried: spills -- 114.0 against 113.5 us) */
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, 9 * c_in * c_out * 4, 0x00020000);
    const int rowb = c_out * 4, va = ((lane >> 4) * 9 * c_out + (lane & 15)) * 4 + 64 * chA, vb = va + 64 * (chB - chA);
#pragma unroll 1
    for (int cb = 0; cb < c_in; cb += 16) {
        RT_PASS_BODY(wc, wn, rw, (cb + 8) * 9 * rowb)
        const auto rn = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, cb + 16 < c_in ? 9 * c_in * c_out * 4 : 0, 0x00020000);
        RT_PASS_BODY(wn, wc, rn, (cb + 16) * 9 * rowb)
    }
#undef RT_WREQ
#undef RT_PASS_BODY
#undef RT_REQ
#undef RT_REQ1
}

// a single input channel (the first stage of an image chain): k = the tap, padded to three k-steps of four with zero weights (chain_conv1_mfma)
template <int S, int ND>
__device__ __forceinline__ void rt_conv1(const float *img, const float *__restrict__ w, int c_out, int chA, int chB, const int (&ptile)[ND],
                                         floatx4 (&acc)[2 * ND], int lane, int s_rt) {
    const int SS = S ? S : s_rt, WP = SS + 2, PX = SS * SS;
    const int l16 = lane & 15, g4 = lane >> 4;
    float wa[3], wb[3];
    int toff[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int tap = 4 * s + g4, tt = tap < 9 ? tap : 0;
        wa[s] = tap < 9 ? w[tap * c_out + 16 * chA + l16] : 0.f;
        wb[s] = tap < 9 ? w[tap * c_out + 16 * chB + l16] : 0.f;
        toff[s] = (tt / 3) * WP + tt % 3;
    }
    const bool pad_lane = 8 + g4 >= 9;
#pragma unroll
    for (int k = 0; k < 2 * ND; ++k) acc[k] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        int p = ptile[i] * 16 + l16;
        if (p >= PX) p = 0;
        const float *px = img + (p / SS) * WP + p % SS;
        float b[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) b[s] = (s == 2 && pad_lane ? img : px)[toff[s]];   // padded taps read the halo corner (0.0)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], b[s], acc[2 * i], 0, 0, 0);
            acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[s], b[s], acc[2 * i + 1], 0, 0, 0);
        }
    }
}

// 2x2 / stride-2 maxima of the tile [c][ld(S S)] -> interior of padded planes [c][CIS(S / 2)], or (TO_GLOBAL) the image's pooled NCHW map
// (+ a flat LDS copy for the classifier rows); chain_pool with a run-time channel count
template <int S>
__device__ __forceinline__ void rt_pool(const float *tile, float *out, int c_out, bool to_global, float *flat, int t, int s_rt) {
    const int SS = S ? S : s_rt, LD = ch_tile_ld(SS * SS), HP = SS / 2, NP = HP * HP, WPO = HP + 2, CISO = ch_cis(WPO);
    const int CHUNKS = (CH_NT / NP >= 8 ? 8 : (CH_NT / NP >= 4 ? 4 : (CH_NT / NP >= 2 ? 2 : 1)));
    static_assert(S % 2 == 0 && (S / 2) * (S / 2) <= CH_NT, "pooled plane fits the workgroup");   // (run-time sizes: <= 32, even -- rt_plan)
    const int q = t % NP, chunk = t / NP;
    if (chunk >= CHUNKS) return;
    const int cpc = (c_out + CHUNKS - 1) / CHUNKS, c0 = chunk * cpc, c1 = min(c_out, c0 + cpc);
    const int pr = q / HP, pc = q % HP;
    const float *b = tile + c0 * LD + 2 * pr * SS + 2 * pc;
    for (int c = c0; c < c1; ++c, b += LD) {
        const float2 r0 = *reinterpret_cast<const float2 *>(b), r1 = *reinterpret_cast<const float2 *>(b + SS);
        float m = -INFINITY;
        m = r0.x > m ? r0.x : m;
        m = r0.y > m ? r0.y : m;
        m = r1.x > m ? r1.x : m;
        m = r1.y > m ? r1.y : m;
        if (to_global) {
            out[c * NP + q] = m;
            if (flat) flat[c * NP + q] = m;
        } else {
            out[c * CISO + (pr + 1) * WPO + pc + 1] = m;
        }
    }
}

template <int S, int ND>
__device__ __forceinline__ void rt_stage_nd(const RtChainArgs &a, const RtStage &st, float *lds, int img, int t, bool last) {
    const int SS = S ? S : st.s, WP = SS + 2, CIS = ch_cis(WP), PX = SS * SS, NPT = (PX + 15) / 16, TLD = ch_tile_ld(PX);
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l16 = lane & 15, g4 = lane >> 4;
    const int nct = st.c_out / 16, np2 = (nct + 1) / 2;
    const bool planes = st.post == TH_CHAIN_NONE;
    const bool to_map = planes && last;          // a run that ends in a conv row: its NCHW map goes to memory, nothing is kept in LDS
    const float *in = lds + st.in_off;
    float *out = lds + st.out_off;                // (an LDS pointer on every path: the map that goes to memory has its own store loop)
    float *gmap = a.y + (long)img * st.c_out * PX;
    const int ld = planes ? CIS : TLD;
    for (int r = 0; r < st.rounds; ++r) {
        const int cid = r * (CH_NT / 64) + wave;
        const bool active = cid < np2 * st.m;                    // wave-uniform
        const int q = active ? cid % np2 : 0, piece = cid / np2;  // neighbouring waves take different channel pairs: the waves that carry one
                                                                  // tile more (piece < NPT % m) land on different SIMDs
        const int chA = 2 * q, chB = min(2 * q + 1, nct - 1);
        const bool has_b = 2 * q + 1 < nct;
        int ptile[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) ptile[i] = piece + st.m * i;
        floatx4 acc[2 * ND];
        float ba[4], bb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ba[e] = st.b[16 * chA + 4 * g4 + e];
            bb[e] = st.b[16 * chB + 4 * g4 + e];
        }
        if (active) {
            if (st.c_in == 1) rt_conv1<S, ND>(in, st.w, st.c_out, chA, chB, ptile, acc, lane, st.s);
            else rt_mfma<S, ND>(in, st.w, st.c_in, st.c_out, chA, chB, ptile, acc, lane, st.s, nullptr);
        }
        // (Requesting the NEXT contraction's first weight pass ahead -- as the compiled instances do -- was built twice in r04: under this
        // stage's epilogue (36 more registers: past 256, reference front 116 -> 130 us), and inside the k loop's last pass with the set
        // carried from contraction to contraction through the stage loop (79 - 90 spilled registers: 113.5 -> 204 us).  A stage list
        // walked at run time cannot keep a 36-register set live across its switch; every contraction opens with its own request.)
        if (st.rounds == 1 && !to_map) {                         // every wave is done reading the input: the output may overlay it
            chain_sync();
            if (planes) rt_zero_halo<S>(out, st.c_out, wave, lane, st.s);
            else if (st.post == TH_CHAIN_MAXPOOL2 && !last) rt_zero_halo<S / 2>(lds + st.pool_off, st.c_out, wave, lane, st.s / 2);
        }
        if (active) {
#pragma unroll
            for (int k = 0; k < 2 * ND; ++k) {
                const int p = ptile[k >> 1] * 16 + l16;
                const bool second = k & 1;
                if (ptile[k >> 1] >= NPT || p >= PX || (second && !has_b)) continue;
                const int ch0 = 16 * (second ? chB : chA) + 4 * g4;
                if (to_map) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[k][e] + (second ? bb[e] : ba[e]);
                        gmap[(long)(ch0 + e) * PX + p] = v > 0.f ? v : 0.f;
                    }
                    continue;
                }
                float *o = out + ch0 * ld + (planes ? (p / SS + 1) * WP + p % SS + 1 : p);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[k][e] + (second ? bb[e] : ba[e]);
                    v = v > 0.f ? v : 0.f;
                    o[e * ld] = v;
                }
            }
        }
    }
    if (to_map) {
        chain_sync();                                            // (the next image's planes overwrite this stage's input)
        return;
    }
    if (st.rounds > 1) {                                         // (the output does not overlay the input: rt_plan)
        if (planes) rt_zero_halo<S>(out, st.c_out, wave, lane, st.s);
        else if (st.post == TH_CHAIN_MAXPOOL2 && !last) rt_zero_halo<S / 2>(lds + st.pool_off, st.c_out, wave, lane, st.s / 2);
    }
    chain_sync();
    if (st.post == TH_CHAIN_MAXPOOL2) {
        if constexpr (S % 2 == 0) {
            const int NP = (SS / 2) * (SS / 2);
            if (last) rt_pool<S>(out, a.y + (long)img * st.c_out * NP, st.c_out, true, a.has_head ? lds + a.xm_off : nullptr, t, st.s);
            else rt_pool<S>(out, lds + st.pool_off, st.c_out, false, nullptr, t, st.s);
        }
        chain_sync();
    } else if (st.post == TH_CHAIN_GLOBAL_AVG) {                 // 16 lanes per plane: avgpool_global16_kernel's arithmetic
        for (int c = t >> 4; c < st.c_out; c += CH_NT / 16) {
            const float *row = out + c * TLD;
            float sum = 0.f, k = 0.f;
            for (int i = l16; i < PX; i += 16) {
                const float v = row[i];
                sum += v;
                k += v > 0.f ? 1.f : 0.f;
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                sum += __shfl_down(sum, off, 16);
                k += __shfl_down(k, off, 16);
            }
            if (l16 == 0) {
                a.y[(long)img * st.c_out + c] = sum / (float)PX;
                if (a.cnt) a.cnt[(long)img * st.c_out + c] = k;
            }
        }
        chain_sync();
    }
}

template <int S>
__device__ __forceinline__ void rt_stage(const RtChainArgs &a, const RtStage &st, float *lds, int img, int t, bool last) {
    switch (st.nd) {
    case 1: rt_stage_nd<S, 1>(a, st, lds, img, t, last); break;
    case 2: rt_stage_nd<S, 2>(a, st, lds, img, t, last); break;
    case 4: rt_stage_nd<S, 4>(a, st, lds, img, t, last); break;
    default: rt_stage_nd<S, 7>(a, st, lds, img, t, last); break;
    }
}

// ANYSIZE: every stage takes the instance with the map size as a run-time value.  A chain's sizes are its input's and its halves: all of
// them in {28, 14, 7} (ANYSIZE = false: the sizes compiled in) or none of them -- two kernels, so that neither pays the other's registers.
template <bool ANYSIZE>
__global__ __launch_bounds__(CH_NT, 1) void conv_chain_rt_kernel(RtChainArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    if (a.has_head && a.head.tick && blockIdx.x == 0 && t == 0) chain_head_tick(a.head);
    for (int img = blockIdx.x; img < a.n; img += gridDim.x) {
        const float *xi = a.x + (long)img * a.c0 * a.s0 * a.s0;
        float *in0 = lds + a.st[0].in_off;
        rt_load_planes(xi, in0, a.c0, a.s0, t);
        chain_sync();
        for (int i = 0; i < a.n_stages; ++i) {
            const RtStage &st = a.st[i];
            const bool last = i == a.n_stages - 1;
            if constexpr (ANYSIZE) {
                rt_stage<0>(a, st, lds, img, t, last);
            } else {
                switch (st.s) {
                case 28: rt_stage<28>(a, st, lds, img, t, last); break;
                case 14: rt_stage<14>(a, st, lds, img, t, last); break;
                default: rt_stage<7>(a, st, lds, img, t, last); break;
                }
            }
        }
        if (a.has_head) {                                        // (the last stage's pool left the flattened map at xm_off, behind a barrier)
            float hw_[CS_NJ][16];
            chain_head_weights_rt<CS_NJ, 16>(a.head, hw_, t);
            chain_head_rows<CS_NJ, 16>(a.head, hw_, lds + a.xm_off, lds + a.red_off, img, t);
        }
        chain_sync();                                            // the next image reuses the maps
    }
#endif
}

}  // namespace th

namespace th {
extern thread_local int t_last_conv_cfg[6];   // conv_mfma.hip: th_debug_last_conv_config
}
using namespace th;

// ---- host side: the compiled instances as DATA, the run-time plan ------------------------------------------------------------------------------
namespace {

struct CompiledChain { int id, c_in, hw, n; int c_out[5], post[5]; };
const CompiledChain kCompiled[] = {
    {1, 1, 28, 5, {32, 32, 64, 64, 128}, {TH_CHAIN_NONE, TH_CHAIN_MAXPOOL2, TH_CHAIN_NONE, TH_CHAIN_MAXPOOL2, TH_CHAIN_GLOBAL_AVG}},   // examples/train_mnist_cnn.rs:35-62
    {2, 1, 28, 2, {32, 64, 0, 0, 0}, {TH_CHAIN_MAXPOOL2, TH_CHAIN_MAXPOOL2, 0, 0, 0}},                                                  // BASELINE configs[2]
};
thread_local int t_chain_generic = [] { const char *e = getenv("TAPER_CHAIN_GENERIC"); return (e && e[0] == '1') ? 1 : 0; }();   // 1: never a compiled instance

int compiled_chain(int c_in, int h, int w, const th_conv_stage *st, int n) {
    if (t_chain_generic) return 0;
    for (const CompiledChain &c : kCompiled) {
        if (c.c_in != c_in || c.hw != h || c.hw != w || c.n != n) continue;
        bool same = true;
        for (int i = 0; i < n && same; ++i) same = st[i].c_out == c.c_out[i] && st[i].post == c.post[i];
        if (same) return c.id;
    }
    return 0;
}

struct Span { int lo, hi; };
// lowest 4-aligned offset for `sz` floats inside the LDS that avoids every live span; -1: none
int rt_place(int sz, const Span *live, int n_live) {
    int cand[8] = {0};
    int nc = 1;
    for (int i = 0; i < n_live; ++i) cand[nc++] = (live[i].hi + 3) & ~3;
    int best = -1;
    for (int c = 0; c < nc; ++c) {
        const int lo = cand[c], hi = lo + sz;
        if (hi > RT_LDS_FLOATS) continue;
        bool clash = false;
        for (int i = 0; i < n_live; ++i) clash = clash || (lo < live[i].hi && live[i].lo < hi);
        if (!clash && (best < 0 || lo < best)) best = lo;
    }
    return best;
}

// Tile mapping and LDS plan of a run of stages (RtStage); false: not a chain this kernel runs.  head_classes > 0: with the classifier rows.
bool rt_plan(int c_in, int h, int w, const th_conv_stage *stages, int n_stages, int head_classes, RtChainArgs *out, int *lds_floats) {
    if (!stages || h != w || h < 4 || h > 32 || n_stages < 1 || n_stages > RT_MAX_STAGES) return false;   // (square maps up to 32 x 32: s + 2 <= 64 lanes)
    if (!(c_in == 1 || (c_in % 16 == 0 && c_in >= 16 && c_in <= 256))) return false;
    RtChainArgs a{};
    a.c0 = c_in; a.s0 = h; a.n_stages = n_stages;
    int s = h, ci = c_in, end = 0;
    Span in{0, c_in * ch_cis(h + 2)};
    if (in.hi > RT_LDS_FLOATS) return false;
    end = in.hi;
    for (int i = 0; i < n_stages; ++i) {
        const th_conv_stage &d = stages[i];
        const bool last = i == n_stages - 1;
        if (!d.d_w || !d.d_bias || d.c_out % 16 != 0 || d.c_out < 16 || d.c_out > 512) return false;
        if (i > 0 && ci % 16 != 0) return false;
        if (d.post != TH_CHAIN_NONE && d.post != TH_CHAIN_MAXPOOL2 && d.post != TH_CHAIN_GLOBAL_AVG) return false;
        if (d.post == TH_CHAIN_MAXPOOL2 && s % 2 != 0) return false;
        if (d.post == TH_CHAIN_GLOBAL_AVG && !last) return false;    // the plane means end a run; a run may also end in a pooled map or in a conv row
        RtStage &st = a.st[i];
        st.w = d.d_w; st.b = d.d_bias; st.c_in = ci; st.c_out = d.c_out; st.s = s; st.post = d.post;
        // tile mapping: np2 channel pairs x m chunks of nd pixel tiles, 8 chunks per round; cheapest rounds * nd, then fewest rounds
        const int npt = (s * s + 15) / 16, np2 = (d.c_out / 16 + 1) / 2;
        // one round where it fits: the 8 / np2 waves of a channel pair interleave its pixel tiles (ND or ND - 1 each)
        int best = 1 << 30;
        if (np2 <= 8) {
            const int m = 8 / np2 < npt ? 8 / np2 : npt, need = (npt + m - 1) / m;
            for (int nd : {1, 2, 4, 7})
                if (nd >= need) { best = need * 16 + 1; st.nd = nd; st.m = m; st.rounds = 1; break; }
        }
        for (int nd : {7, 4, 2, 1}) {
            const int m = (npt + nd - 1) / nd, rounds = (np2 * m + 7) / 8, cost = rounds * nd * 16 + rounds;
            if (cost < best) { best = cost; st.nd = nd; st.m = m; st.rounds = rounds; }
        }
        // LDS: the input is live through the k loops; with ONE round every accumulator is in registers when the input dies (a barrier), so
        // the output may overlay it -- taken only when nothing else fits
        st.in_off = in.lo;
        if (last && d.post == TH_CHAIN_NONE) {    // the run's NCHW map goes straight to memory: no LDS region
            if (head_classes > 0) return false;
            st.out_off = 0;
            ci = d.c_out;
            continue;
        }
        const int out_sz = d.c_out * (d.post == TH_CHAIN_NONE ? ch_cis(s + 2) : ch_tile_ld(s * s));
        int o = rt_place(out_sz, &in, 1);
        if (o < 0 && st.rounds == 1) o = rt_place(out_sz, nullptr, 0);
        if (o < 0) return false;
        st.out_off = o;
        Span outs{o, o + out_sz};
        end = end > outs.hi ? end : outs.hi;
        if (d.post == TH_CHAIN_NONE) {
            in = outs;
        } else if (d.post == TH_CHAIN_MAXPOOL2 && !last) {
            const int p_sz = d.c_out * ch_cis(s / 2 + 2);
            const Span live2[2] = {outs, in};
            int po = rt_place(p_sz, live2, 2);
            if (po < 0 && st.rounds == 1) po = rt_place(p_sz, &outs, 1);     // over the (dead) input
            if (po < 0) return false;
            st.pool_off = po;
            in = Span{po, po + p_sz};
            end = end > in.hi ? end : in.hi;
            s /= 2;
        } else if (d.post == TH_CHAIN_MAXPOOL2) {
            s /= 2;
        }
        if (last && head_classes > 0) {
            if (d.post != TH_CHAIN_MAXPOOL2 || head_classes > 16) return false;
            const int k = d.c_out * s * s;
            if (k > CS_NJ * CH_NT) return false;
            const int xm = rt_place(k, &outs, 1);
            if (xm < 0) return false;
            const Span xs{xm, xm + k};
            const int red = rt_place(16 * CH_NT + 64, &xs, 1);
            if (red < 0) return false;
            a.has_head = 1; a.xm_off = xm; a.red_off = red;
            end = end > xs.hi ? end : xs.hi;
            end = end > red + 16 * CH_NT + 64 ? end : red + 16 * CH_NT + 64;
        }
        ci = d.c_out;
    }
    if (out) *out = a;
    if (lds_floats) *lds_floats = end;
    return true;
}

}  // namespace

namespace th {
// th_conv3x3_fwd / _pool2_fwd / _gap_fwd (conv_mfma.hip: conv3x3_mfma_launch) ask here first.  1: launched; 0: not one of the compiled geometries
// (or fewer images than half the CUs: one image per workgroup); -1: error
int conv_layer_chain_launch(th_ctx *ctx, const float *x, const float *w, const float *bias, float *y, float *cnt, int n, int c_in, int hw, int c_out,
                            int post, bool linear) {
    if (n < kNumCU / 2) return 0;
    LayerChainArgs a{x, w, bias, y, cnt, n};
    const dim3 grid(n < kNumCU ? n : kNumCU);
#define TH_LC_(S_, CI_, CO_, P_, L_)                                                                                                     \
    do {                                                                                                                                 \
        constexpr int fl = (CI_ * ch_cis(S_ + 2) > CO_ * ch_tile_ld(S_ * S_) ? CI_ * ch_cis(S_ + 2) : CO_ * ch_tile_ld(S_ * S_));         \
        TH_SET_MAX_LDS(ctx, (conv_layer_chain_kernel<S_, CI_, CO_, P_, L_>), fl * 4); \
        hipLaunchKernelGGL((conv_layer_chain_kernel<S_, CI_, CO_, P_, L_>), grid, dim3(CH_NT), fl * 4, ctx->stream, a);                   \
        if (hipGetLastError() != hipSuccess) return -1;                                                                                  \
        return 1;                                                                                                                        \
    } while (0)
#define TH_LC(S_, CI_, CO_, P_) TH_LC_(S_, CI_, CO_, P_, false)
    if (linear) {
        // input gradients (no bias, no ReLU, the map written) of the four batch-256 layers: 64 -> 32 @14 and 128 -> 64 @7 ran at half their
        // forward rate in the image-resident kernel (36 us each: 23.0 / 21.5 here)
        if (post != 0) return 0;
        if (hw == 14 && c_in == 64 && c_out == 32) TH_LC_(14, 64, 32, 0, true);
        if (hw == 7 && c_in == 128 && c_out == 64) TH_LC_(7, 128, 64, 0, true);
        if (hw == 28 && c_in == 32 && c_out == 32) TH_LC_(28, 32, 32, 0, true);
        if (hw == 14 && c_in == 64 && c_out == 64) TH_LC_(14, 64, 64, 0, true);
        return 0;
    }
    if (hw == 28 && c_in == 32 && c_out == 32) { if (post == 0) TH_LC(28, 32, 32, 0); if (post == 1) TH_LC(28, 32, 32, 1); }
    if (hw == 14 && c_in == 32 && c_out == 64) { if (post == 0) TH_LC(14, 32, 64, 0); if (post == 1) TH_LC(14, 32, 64, 1); }
    if (hw == 14 && c_in == 64 && c_out == 64) { if (post == 0) TH_LC(14, 64, 64, 0); if (post == 1) TH_LC(14, 64, 64, 1); }
    if (hw == 7 && c_in == 64 && c_out == 128) { if (post == 0) TH_LC(7, 64, 128, 0); if (post == 2) TH_LC(7, 64, 128, 2); }
#undef TH_LC
#undef TH_LC_
    return 0;
}
}  // namespace th

extern "C" {

// 0: no chain launch for these stages (the caller launches the layers one by one); 1, 2: a compiled instance (kCompiled); 3: the
// run-time-described kernel
int th_conv_chain_supported(int c_in, int h, int w, const th_conv_stage *stages, int n_stages) {
    if (!stages || n_stages < 1) return 0;
    for (int i = 0; i < n_stages; ++i)
        if (!stages[i].d_w || !stages[i].d_bias) return 0;
    if (const int id = compiled_chain(c_in, h, w, stages, n_stages)) return id;
    return rt_plan(c_in, h, w, stages, n_stages, 0, nullptr, nullptr) ? 3 : 0;
}

thread_local int t_chain_loop = -1;   // th_debug_set_chain_loop
thread_local int t_chain_mlp3_only = 0;   // th_debug_chain_mlp3_only: 1 = th_conv_chain_mlp3_xent runs its first launch alone (per-launch timing)
int th_debug_chain_mlp3_only(int which) {
    t_chain_mlp3_only = which == 1 ? 1 : 0;
    return 0;
}
int th_debug_set_chain_loop(int on) {   // test hook: 1 = batches above 256 images take the walking instances on this thread; 0 = never; -1 = default (off)
    t_chain_loop = on < 0 ? -1 : (on ? 1 : 0);
    return 0;
}

int th_debug_set_chain_generic(int on) {   // test hook: 1 = the compiled instances are not used (their nets take the run-time-described kernel)
    t_chain_generic = on ? 1 : 0;
    return 0;
}

static int rt_launch(th_ctx *ctx, RtChainArgs &a, int lds_floats, const float *d_x, float *d_y, float *d_cnt, int n) {
    a.x = d_x; a.y = d_y; a.cnt = d_cnt; a.n = n;
    const int lds = lds_floats * (int)sizeof(float);
    if (a.s0 == 28 || a.s0 == 14 || a.s0 == 7) {
        TH_SET_MAX_LDS(ctx, (conv_chain_rt_kernel<false>), lds);
        hipLaunchKernelGGL(conv_chain_rt_kernel<false>, dim3(n < kNumCU ? n : kNumCU), dim3(CH_NT), lds, ctx->stream, a);
    } else {
        TH_SET_MAX_LDS(ctx, (conv_chain_rt_kernel<true>), lds);
        hipLaunchKernelGGL(conv_chain_rt_kernel<true>, dim3(n < kNumCU ? n : kNumCU), dim3(CH_NT), lds, ctx->stream, a);
    }
    return 0;
}

// The compiled instances as WALKING workgroups once there are more images than CUs (r05, asked for by the r04 review): built, bit-identical
// per image (tests/test_gpu_chain.py, test_gpu_chain_head.py), and measured 3 - 5 % SLOWER than one workgroup per image at 1 024 and
// 4 096 images (reference front 405 against 393 us per 1 024-image step, simple 118.9 against 113.2): the hardware hands a CU its next
// workgroup within ~1 us, and what the walk saves on top of that (the image and first-weight round trips, ~2 us per image) the loop's code
// gives back -- with the per-image addresses hoisted out of the image loop the k loops spilled (+15 us per image; made opaque per image:
// the figures above).  OFF by default; TAPER_CHAIN_LOOP=1 / th_debug_set_chain_loop(1) turn it on.
static bool chain_loop(int n) {
    static const bool env_on = getenv("TAPER_CHAIN_LOOP") && getenv("TAPER_CHAIN_LOOP")[0] == '1';
    return (t_chain_loop >= 0 ? t_chain_loop != 0 : env_on) && n > kNumCU;
}

int th_conv_chain_fwd(th_ctx *ctx, const float *d_x, const th_conv_stage *stages, int n_stages, float *d_y, float *d_cnt, int n, int c_in,
                      int h, int w) {
    TH_REQUIRE(ctx && d_x && d_y && stages && n > 0, "th_conv_chain_fwd: null argument");
    const int kind = th_conv_chain_supported(c_in, h, w, stages, n_stages);
    TH_REQUIRE(kind != 0, "th_conv_chain_fwd: these stages do not run as a chain (th_conv_chain_supported)");
    if (kind == 3) {
        RtChainArgs ra{};
        int lds_floats = 0;
        TH_REQUIRE(rt_plan(c_in, h, w, stages, n_stages, 0, &ra, &lds_floats), "th_conv_chain_fwd: no plan");
        rt_launch(ctx, ra, lds_floats, d_x, d_y, stages[n_stages - 1].post == TH_CHAIN_GLOBAL_AVG ? d_cnt : nullptr, n);
    } else {
        ConvChainArgs a{};
        a.x = d_x; a.y = d_y; a.cnt = d_cnt; a.n = n;
        for (int i = 0; i < n_stages; ++i) { a.w[i] = stages[i].d_w; a.b[i] = stages[i].d_bias; }
        if (kind == 1) {
            const int lds = CR_LDS * (int)sizeof(float);
            // (more images than CUs: optionally as walking workgroups -- chain_loop)
            if (chain_loop(n)) {
                TH_SET_MAX_LDS(ctx, (conv_chain_reference_kernel<true>), lds);
                hipLaunchKernelGGL(conv_chain_reference_kernel<true>, dim3(kNumCU), dim3(CH_NT), lds, ctx->stream, a);
            } else {
                TH_SET_MAX_LDS(ctx, (conv_chain_reference_kernel<false>), lds);
                hipLaunchKernelGGL(conv_chain_reference_kernel<false>, dim3(n), dim3(CH_NT), lds, ctx->stream, a);
            }
        } else {
            const int lds = CS_LDS * (int)sizeof(float);
            static const int lean_env = [] { const char *e = getenv("TAPER_CHAIN_LEAN"); return e ? atoi(e) : -1; }();
            if (chain_loop(n)) {
                TH_SET_MAX_LDS(ctx, (conv_chain_simple_kernel<false, 1, true>), lds);
                hipLaunchKernelGGL((conv_chain_simple_kernel<false, 1, true>), dim3(kNumCU), dim3(CH_NT), lds, ctx->stream, a);
            } else if (lean_env >= 0 ? lean_env != 0 : n > kNumCU) {      // more than one image per CU: two workgroups to a CU (as th_conv_chain_head_fwd)
                TH_SET_MAX_LDS(ctx, (conv_chain_simple_kernel<false, 1, false, true>), lds);
                hipLaunchKernelGGL((conv_chain_simple_kernel<false, 1, false, true>), dim3(n), dim3(CH_NT), lds, ctx->stream, a);
            } else {
                TH_SET_MAX_LDS(ctx, (conv_chain_simple_kernel<false, 1>), lds);
                hipLaunchKernelGGL((conv_chain_simple_kernel<false, 1>), dim3(n), dim3(CH_NT), lds, ctx->stream, a);
            }
        }
    }
    t_last_conv_cfg[0] = kind; t_last_conv_cfg[1] = 6; t_last_conv_cfg[2] = 0;   // 6: a conv chain (th_debug_last_conv_config)
    t_last_conv_cfg[3] = n; t_last_conv_cfg[4] = 1; t_last_conv_cfg[5] = 0;
    TH_LAUNCH_CHECK();
    return 0;
}

// th_conv_chain_fwd + th_mlp3_xent in TWO launches instead of three: the reference CNN's front with its three-layer classifier's rows in the
// last epilogue (chain_mlp3_rows), then th_mlp3_xent's own second launch (mlp3_grads_launch, gemm.hip).  The classifier's row launch was
// 10.7 us of the 111 us step, 2.7 of them waiting for the plane means it starts from to come back from memory.
int th_conv_chain_mlp3_supported(int c_in, int h, int w, const th_conv_stage *stages, int n_stages, int n, int h1, int h2, int classes) {
    if (!stages || n_stages < 1 || n < 1 || n % 16 != 0 || chain_loop(n)) return 0;
    if (stages[n_stages - 1].post != TH_CHAIN_GLOBAL_AVG || stages[n_stages - 1].c_out != 128) return 0;
    if (h1 != 128 || h2 != 64 || classes < 1 || classes > 16) return 0;
    return th_conv_chain_supported(c_in, h, w, stages, n_stages) == 1 ? 1 : 0;      // the compiled reference instance
}

int th_conv_chain_mlp3_xent(th_ctx *ctx, const float *d_x, const th_conv_stage *stages, int n_stages, float *d_y, float *d_cnt, int n, int c_in,
                            int h, int w, const float *d_targets, const th_mlp3_layer *layers, float *d_dx, float *d_loss, float *d_ncorrect,
                            float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance, int32_t *d_tick, const th_mlp3_gap *gap) {
    TH_REQUIRE(ctx && d_x && d_y && stages && d_targets && layers && d_loss, "th_conv_chain_mlp3_xent: null argument");
    const int h1 = layers[0].out_features, h2 = layers[1].out_features, c = layers[2].out_features;
    TH_REQUIRE(th_conv_chain_mlp3_supported(c_in, h, w, stages, n_stages, n, h1, h2, c),
               "th_conv_chain_mlp3_xent: needs the compiled reference front ending in 128 plane means, a 128-128-64-classes classifier (classes <= 16) "
               "and a batch that is a multiple of 16 (got %d images, %d-%d-%d)", n, h1, h2, c);
    for (int l = 0; l < 3; ++l) {
        TH_REQUIRE(layers[l].d_w && ((uintptr_t)layers[l].d_w & 15) == 0, "th_conv_chain_mlp3_xent: weights must be 16-byte aligned");
        TH_REQUIRE(!(layers[l].w_fuse && layers[l].w_fuse->d_p) || layers[l].d_dw, "th_conv_chain_mlp3_xent: a fused weight update needs d_dw");
        TH_REQUIRE(!(layers[l].b_fuse && layers[l].b_fuse->d_p) || layers[l].d_db, "th_conv_chain_mlp3_xent: a fused bias update needs d_db");
    }
    TH_REQUIRE(!d_metrics || (d_state && metrics_capacity > 0), "th_conv_chain_mlp3_xent: metrics need d_state and a capacity");
    TH_REQUIRE(!(gap && gap->d_cnt) || (d_dx && gap->d_gb && gap->hw > 0 && gap->d_cnt == d_cnt),
               "th_conv_chain_mlp3_xent: the gap bias finish needs d_dx, d_gb, hw > 0 and the counts this launch writes");
    // workspace: A1, A2, dZ1, dZ2, dZ3, the rows' {NLL, hit}
    const size_t n1 = (size_t)n * 128, n2 = (size_t)n * 64, n3 = (size_t)n * c, n3p = (n3 + 3) & ~(size_t)3;
    void *ws = nullptr;
    if (th_malloc(ctx, (2 * n1 + 2 * n2 + n3p + 2 * (size_t)n) * sizeof(float), &ws)) return 1;
    float *a1 = (float *)ws, *a2 = a1 + n1, *dz1 = a2 + n2, *dz2 = dz1 + n1, *dz3 = dz2 + n2, *part = dz3 + n3p;
    ConvChainArgs a{};
    a.x = d_x; a.y = d_y; a.cnt = d_cnt; a.n = n;
    for (int i = 0; i < n_stages; ++i) { a.w[i] = stages[i].d_w; a.b[i] = stages[i].d_bias; }
    a.head3 = ChainMlp3Args{layers[0].d_w, layers[0].d_b, layers[1].d_w, layers[1].d_b, layers[2].d_w, layers[2].d_b, d_targets,
                            a1, a2, dz1, dz2, dz3, d_dx, part, d_tick, c, 1.0f / (float)n};
    const int lds = CR_LDS * (int)sizeof(float);
    TH_SET_MAX_LDS(ctx, (conv_chain_reference_kernel<false, true>), lds);
    hipLaunchKernelGGL((conv_chain_reference_kernel<false, true>), dim3(n), dim3(CH_NT), lds, ctx->stream, a);
    t_last_conv_cfg[0] = 1; t_last_conv_cfg[1] = 9; t_last_conv_cfg[2] = 0;   // 9: a conv chain with the three-layer classifier's rows
    t_last_conv_cfg[3] = n; t_last_conv_cfg[4] = 1; t_last_conv_cfg[5] = 0;
    TH_LAUNCH_CHECK();
    if (t_chain_mlp3_only) return th_free(ctx, ws);
    const float *dz[3] = {dz1, dz2, dz3}, *act[3] = {d_y, a1, a2};
    float *dw[3] = {layers[0].d_dw, layers[1].d_dw, layers[2].d_dw}, *db[3] = {layers[0].d_db, layers[1].d_db, layers[2].d_db};
    const int out_f[3] = {h1, h2, c}, in_f[3] = {128, h1, h2};
    const th_adam_fuse *wf[3] = {layers[0].w_fuse, layers[1].w_fuse, layers[2].w_fuse}, *bf[3] = {layers[0].b_fuse, layers[1].b_fuse, layers[2].b_fuse};
    if (int rc = th::mlp3_grads_launch(ctx, dz, act, dw, db, out_f, in_f, wf, bf, n, part, n, d_loss, d_ncorrect, d_metrics, metrics_capacity, d_state,
                                       advance, d_dx, gap)) {
        (void)th_free(ctx, ws);         // (error paths give their workspace back: ADVICE r05)
        return rc;
    }
    return th_free(ctx, ws);
}

// the classifier rows ride in the chain's last epilogue where the chain ends in a pooled map that fits a thread's registers (<= 7 elements
// per thread): 2 = the compiled simple instance, 3 = the run-time-described kernel
int th_conv_chain_head_supported(int c_in, int h, int w, const th_conv_stage *stages, int n_stages, int classes) {
    if (classes < 1 || classes > 16 || !stages || n_stages < 1) return 0;
    for (int i = 0; i < n_stages; ++i)
        if (!stages[i].d_w || !stages[i].d_bias) return 0;
    if (compiled_chain(c_in, h, w, stages, n_stages) == 2) return 2;
    return rt_plan(c_in, h, w, stages, n_stages, classes, nullptr, nullptr) ? 3 : 0;
}

int th_conv_chain_head_fwd(th_ctx *ctx, const float *d_x, const th_conv_stage *stages, int n_stages, float *d_y, int n, int c_in, int h, int w,
                           const th_chain_head *head) {
    TH_REQUIRE(ctx && d_x && d_y && stages && head && n > 0, "th_conv_chain_head_fwd: null argument");
    TH_REQUIRE(head->d_w && head->d_targets && head->d_dl && head->d_rowstat, "th_conv_chain_head_fwd: the head needs d_w, d_targets, d_dl, d_rowstat");
    const int kind = th_conv_chain_head_supported(c_in, h, w, stages, n_stages, head->classes);
    TH_REQUIRE(kind != 0, "th_conv_chain_head_fwd: these stages do not run as a chain + head (th_conv_chain_head_supported)");
    if (kind == 3) {
        RtChainArgs ra{};
        int lds_floats = 0;
        TH_REQUIRE(rt_plan(c_in, h, w, stages, n_stages, head->classes, &ra, &lds_floats), "th_conv_chain_head_fwd: no plan");
        const RtStage &ls = ra.st[n_stages - 1];
        const int hw = (ls.s / 2) * (ls.s / 2);
        ra.head = ChainHeadArgs{head->d_w, head->d_bias, head->d_targets, head->d_dl, head->d_rowstat, head->d_cbpart, head->d_tick,
                                head->classes, ls.c_out * hw, ls.c_out, hw, 1.0f / (float)n, ctx->update_guard, ctx->update_step_word};
        rt_launch(ctx, ra, lds_floats, d_x, d_y, nullptr, n);
    } else {
        ConvChainArgs a{};
        a.x = d_x; a.y = d_y; a.cnt = nullptr; a.n = n;
        for (int i = 0; i < n_stages; ++i) { a.w[i] = stages[i].d_w; a.b[i] = stages[i].d_bias; }
        a.head = ChainHeadArgs{head->d_w, head->d_bias, head->d_targets, head->d_dl, head->d_rowstat, head->d_cbpart, head->d_tick,
                               head->classes, CS_K, 64, 49, 1.0f / (float)n, ctx->update_guard, ctx->update_step_word};
        const int lds = CS_LDS * (int)sizeof(float);
#define TH_SIMPLE_HEAD(NC_, LOOP_, GRID_)                                                                                                          \
    do {                                                                                                                                          \
        TH_SET_MAX_LDS(ctx, (conv_chain_simple_kernel<true, NC_, LOOP_>), lds);     \
        hipLaunchKernelGGL((conv_chain_simple_kernel<true, NC_, LOOP_>), dim3(GRID_), dim3(CH_NT), lds, ctx->stream, a);                          \
    } while (0)
        // more than one image per CU: the 128-register instance, two workgroups to a CU (TAPER_CHAIN_LEAN = 0 | 1 forces it off / on: A/B probe).
        // From 257 images on, not 512: the one-to-a-CU instance runs a second round for the 257th image (264 images: 48.7 against 43.3 us)
        static const int lean_env = [] { const char *e = getenv("TAPER_CHAIN_LEAN"); return e ? atoi(e) : -1; }();
        const bool lean = lean_env >= 0 ? lean_env != 0 : n > kNumCU;
        if (head->classes <= 10 && lean) {
            TH_SET_MAX_LDS(ctx, (conv_chain_simple_kernel<true, 10, false, true>), lds);
            hipLaunchKernelGGL((conv_chain_simple_kernel<true, 10, false, true>), dim3(n), dim3(CH_NT), lds, ctx->stream, a);
        } else if (head->classes <= 10) {
            if (chain_loop(n)) TH_SIMPLE_HEAD(10, true, kNumCU);
            else TH_SIMPLE_HEAD(10, false, n);
        } else {
            if (chain_loop(n)) TH_SIMPLE_HEAD(16, true, kNumCU);
            else TH_SIMPLE_HEAD(16, false, n);
        }
#undef TH_SIMPLE_HEAD
    }
    t_last_conv_cfg[0] = kind; t_last_conv_cfg[1] = 7; t_last_conv_cfg[2] = 0;   // 7: a conv chain with the classifier rows
    t_last_conv_cfg[3] = n; t_last_conv_cfg[4] = 1; t_last_conv_cfg[5] = 0;
    TH_LAUNCH_CHECK();
    return 0;
}

#ifdef TH_PROFILE
int th_debug_chain_prof(th_ctx *ctx, long long *h_out32) {
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpyFromSymbol(h_out32, HIP_SYMBOL(th::g_chain_prof), 32 * sizeof(long long)));
    return 0;
}
int th_debug_chain_span(th_ctx *ctx, long long *h_out512) {
    TH_HIP(hipStreamSynchronize(ctx->stream));
    TH_HIP(hipMemcpyFromSymbol(h_out512, HIP_SYMBOL(th::g_chain_span), 512 * sizeof(long long)));
    return 0;
}
#endif

}  // extern "C"

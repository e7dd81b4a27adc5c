/*
 * taper_hip.h -- the drop-in boundary: C ABI of the MI355X (gfx950) backend
 * for taper's training hot path.
 *
 * The reference (vaibhawvipul/taper, Rust) defines no FFI for this path; its
 * only foreign seam is cblas_sgemm behind src/gemm.rs:32-47.  This header is
 * the interface a Rust `extern "C"` block in the taper crate would bind (see
 * INTEGRATION.md): the host keeps Tensor / Tape / nn::Module, every Tensor
 * method body that today loops over Vec<f32> becomes one call below.  Each
 * entry point cites the reference code it replaces (file:line under the
 * reference checkout).
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; the message is
 *    in th_last_error() (thread-local).  Nothing panics or aborts: the
 *    reference's assert!/panic! paths map to error returns.
 *  - pointers named d_* are DEVICE pointers (fp32 unless noted) obtained from
 *    th_malloc (or any hipMalloc'd memory); h_* are host pointers.
 *  - every op is enqueued on the context's HIP stream and never synchronises;
 *    only th_ctx_sync, th_memcpy_d2h and th_event_elapsed_ms wait.
 *  - "accumulate" (+=) outputs mirror the reference's lazily-zeroed grad
 *    slots (src/ops.rs:124-137): the caller zero-fills on first touch.
 *  - one th_ctx per host thread / GPU (mirrors the thread-local tape,
 *    src/tape.rs:6-9); a ctx must not be shared between threads.
 */
#ifndef TAPER_HIP_H
#define TAPER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct th_ctx th_ctx;
typedef struct th_graph th_graph;
typedef struct th_event th_event;
typedef struct th_comm th_comm;

/* Adam update applied in the epilogue of the kernel that produces a parameter's
 * gradient (src/optim.rs:99-110 for that tensor, same arithmetic as
 * th_adam_step).  Slices of the flat p / m / v arenas at the parameter's offset;
 * d_t is the step counter ALREADY advanced for this step (th_adam_tick or the
 * tick folded into th_softmax_xent_fwd / th_linear_xent_head). */
typedef struct th_adam_fuse {
    float *d_p, *d_m, *d_v;
    const int32_t *d_t;
    const float *d_lr;
    float beta1, beta2, eps, weight_decay;
} th_adam_fuse;

/* Adam update of a gradient slice that an EARLIER launch of this step already
 * completed; a later launch carries it in spare workgroups instead of paying
 * a kernel boundary for it (th_linear_bwd_adam_ex) -- or th_adam_slices runs
 * the leftovers.  The carrying launch must not read f.d_p. */
typedef struct th_adam_slice {
    const float *d_g;
    int64_t n;
    th_adam_fuse f;
} th_adam_slice;
#define TH_MAX_ADAM_SLICES 4

/* ---- runtime --------------------------------------------------------- */
const char *th_last_error(void);
int th_device_count(int *out);
int th_ctx_create(int device_id, th_ctx **out);
int th_ctx_destroy(th_ctx *ctx);
int th_ctx_sync(th_ctx *ctx);
/* the ctx's hipStream_t, for interop (e.g. torch.cuda.ExternalStream) */
void *th_ctx_stream(th_ctx *ctx);
int th_ctx_device(th_ctx *ctx);

/* Stream-ordered pooled allocator: replaces the Vec<f32> every reference op
 * allocates for its output (e.g. src/ops.rs:22, 212).  th_free returns the
 * block to the pool; it is safe to reuse because all work is on one stream. */
int th_malloc(th_ctx *ctx, size_t bytes, void **d_out);
int th_free(th_ctx *ctx, void *d_ptr);
/* Pinned host memory that kernels can write through the same pointer (hipHostMalloc, mapped + coherent): the per-step
 * {loss, n_correct} log of a replayed epoch lives here, so the host reads it after one stream synchronisation instead of
 * a staged device-to-host copy (examples/train_mnist.rs:110-121 reads both values every step).  Not pooled. */
int th_host_malloc(th_ctx *ctx, size_t bytes, void **h_out);
/* fine-grained (coherent across agents, uncached in remote L2s) device memory, outside the pool: hipExtMallocWithFlags; for buffers peers
 * read while this device keeps writing them between kernel boundaries (th_comm_p2p_export) */
int th_malloc_finegrained(th_ctx *ctx, size_t bytes, void **d_out);
int th_free_finegrained(th_ctx *ctx, void *d_ptr);
int th_host_free(th_ctx *ctx, void *h_ptr);
int th_pool_stats(th_ctx *ctx, size_t *bytes_reserved, size_t *bytes_in_use);
int th_memcpy_h2d(th_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int th_memcpy_d2h(th_ctx *ctx, void *h_dst, const void *d_src, size_t bytes); /* waits */
int th_memcpy_d2d(th_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
int th_fill_f32(th_ctx *ctx, float *d_p, float v, size_t n);

/* events (hipEvent on the ctx stream) -- used by bench.py for the live
 * kernel timing the roofline line needs */
int th_event_create(th_event **out);
int th_event_destroy(th_event *ev);
int th_event_record(th_ctx *ctx, th_event *ev);
int th_event_elapsed_ms(th_event *start, th_event *stop, float *ms_out); /* waits for stop */

/* hipGraph capture of everything enqueued between begin/end on this ctx; the
 * host tape's op list for one training step is replayed with one launch. */
int th_graph_begin(th_ctx *ctx);
int th_graph_end(th_ctx *ctx, th_graph **out);
int th_graph_launch(th_ctx *ctx, th_graph *g);
int th_graph_destroy(th_graph *g);

/* ---- gemm: src/gemm.rs:8-49 / 72-119 `sgemm_rowmajor` ----------------- */
/* C[m,n] = alpha * op(A)[m,k] * op(B)[k,n] + beta * C, row-major;
 * lda = transA ? m : k, ldb = transB ? k : n, ldc = n (gemm.rs:21-29).
 * beta == 0 overwrites C (never reads it).  fp32 MFMA accumulate. */
int th_sgemm(th_ctx *ctx, int trans_a, int trans_b, int m, int n, int k, float alpha,
             const float *d_a, const float *d_b, float beta, float *d_c);

/* ---- fused Linear: src/nn.rs:54-60 (transpose + matmul + add_broadcast) */
/* Y[B,out] = X[B,in] . W[out,in]^T + b[out]  (b nullable); relu != 0 fuses
 * activation.rs:10-12 / ops.rs:312-349 into the epilogue. */
int th_linear_fwd(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_b, float *d_y,
                  int batch, int in_features, int out_features, int relu);
/* same product, and the launch also (a) applies up to TH_MAX_ADAM_SLICES deferred
 * Adam updates (extra nullable) of parameters this layer does not read, with the
 * step counter AS IT STANDS (they belong to the step that just ended), then
 * (b) advances the counter d_tick (nullable; th_adam_step's d_t) by one
 * (optim.rs:84) -- the first launch of a fused training step (th_mlp_tail). */
int th_linear_fwd_ex(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_b, float *d_y,
                     int batch, int in_features, int out_features, int relu,
                     const th_adam_slice *extra, int n_extra, int32_t *d_tick);
/* backward of the three reference nodes at once (ops.rs:238-294,
 * tensor.rs:574-587, 674-694); any of d_dx, d_dw, d_db may be NULL (input
 * without requires_grad: ops.rs:243).  d_relu_y (nullable): the layer's
 * POST-activation output; when given, the ReLU backward (ops.rs:358-369) is
 * folded in: dZ = dY * (Y > 0) (Y > 0 <=> pre-activation > 0, quirk Q15).
 * accumulate_mask bit0/1/2 = dx/dw/db: set -> += into an existing grad;
 * clear -> the grad slot was None, the kernel overwrites (0 + x, saving the
 * zero-fill of ops.rs:247-249).  Latency-bound shapes run as ONE launch. */
int th_linear_bwd(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_dy, const float *d_relu_y,
                  float *d_dx, float *d_dw, float *d_db, int batch, int in_features, int out_features,
                  int accumulate_mask);
/* same, and additionally applies the Adam update of W / b in the epilogue
 * (w_fuse / b_fuse nullable).  A fused tensor's gradient must be complete in
 * this call: its accumulate_mask bit must be clear (grad slot was None). */
int th_linear_bwd_adam(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_dy, const float *d_relu_y,
                       float *d_dx, float *d_dw, float *d_db, int batch, int in_features, int out_features,
                       int accumulate_mask, const th_adam_fuse *w_fuse, const th_adam_fuse *b_fuse);
/* same, plus up to TH_MAX_ADAM_SLICES deferred updates (extra nullable) of
 * OTHER parameters whose gradients are already complete (e.g. the classifier
 * head's W / b, produced by th_linear_xent_head one launch earlier). */
int th_linear_bwd_adam_ex(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_dy, const float *d_relu_y,
                          float *d_dx, float *d_dw, float *d_db, int batch, int in_features, int out_features,
                          int accumulate_mask, const th_adam_fuse *w_fuse, const th_adam_fuse *b_fuse,
                          const th_adam_slice *extra, int n_extra);
/* ... and with the backward of the layer IN FRONT folded into this layer's dX product, when that layer is a fused Linear + ReLU whose output
 * is d_x: mask_dx != 0: d_dx <- d_dx * [d_x > 0] (that ReLU's backward, ops.rs:358-369: no pass over [batch, in_features] for it);
 * d_dx_colpart (nullable) [th_linear_bwd_dx_epilogue_rows(..)][in_features]: column sums of the (masked) d_dx by row block, which
 * th_colsum over its rows turns into that layer's bias gradient (tensor.rs:686-691: no pass for that either).  Needs d_dx written (not
 * accumulated) and a shape for which th_linear_bwd_dx_epilogue_rows is > 0: the unsplit 128 x 128 MFMA tiles (4096-wide layers), or a thin
 * layer (out_features <= 16 on >= 2^20 outputs), whose dX then runs as a streaming launch (67 MB in ~25 us instead of 119). */
int th_linear_bwd_dx_epilogue_rows(int batch, int in_features, int out_features);
/* 1: th_linear_bwd_adam_ex2 runs this shape's products as launches of their own (the MFMA tile kernels: dX first, then dW -- so a fused W
 * update may ride in the dW product's epilogue even when d_dx is asked for, the dX product has consumed W by then); 0: the one-launch form
 * of the latency-bound shapes (there a W update next to d_dx runs as a slice launch behind it: let a later launch carry it instead).
 * with_dx / with_dw: whether d_dx / d_dw will be given; dx_epilogue: whether mask_dx / d_dx_colpart will. */
int th_linear_bwd_separate_products(int batch, int in_features, int out_features, int with_dx, int with_dw, int dx_epilogue);
int th_linear_bwd_adam_ex2(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_dy, const float *d_relu_y, float *d_dx,
                           float *d_dw, float *d_db, int batch, int in_features, int out_features, int accumulate_mask,
                           const th_adam_fuse *w_fuse, const th_adam_fuse *b_fuse, const th_adam_slice *extra, int n_extra, int mask_dx,
                           float *d_dx_colpart);

/* ---- fused classifier head: last Linear + softmax cross-entropy --------- */
/* One workgroup computes logits = H[B,in] . W[C,in]^T + b (nn.rs:54-60), the
 * loss / accuracy count / step log of th_softmax_xent_fwd (loss.rs:101-195,
 * 271-290) AND the backward products for an upstream gradient of exactly 1:
 *   dlogits = (softmax - onehot)/B,  d_dw[C,in] = dlogits^T . H,  d_db[C] = colsum(dlogits),
 *   d_dh[B,in] = dlogits . W   (the ReLU mask of the previous layer is applied by ITS backward).
 * Requires classes <= 16, in_features <= 256.  Up to 64 rows: ONE workgroup, one launch; above: up to 256
 * workgroups each own a row range and write dW / db / loss partials, a finish pass adds them in order.  Nullable:
 * d_bias, d_logits, d_ncorrect, d_dh, d_dw, d_db, metrics/state, d_adam_tick,
 * w_fuse, b_fuse.  d_adam_tick (int32[2], th_adam_step's d_t): t += 1 is done
 * here (optim.rs:84) so that the fused updates of this step see the new t.
 * W / b are updated (fuse) only after every read of W in this launch. */
int th_linear_xent_head(th_ctx *ctx, const float *d_h, const float *d_w, const float *d_bias, const float *d_targets,
                        int batch, int in_features, int classes, float *d_logits, float *d_loss, float *d_ncorrect,
                        float *d_dh, float *d_dw, float *d_db, float *d_metrics, int64_t metrics_capacity,
                        int64_t *d_state, int64_t advance, int32_t *d_adam_tick, const th_adam_fuse *w_fuse,
                        const th_adam_fuse *b_fuse);
/* same; mask_dh_by_h != 0: d_h is a ReLU output and d_dh receives dH * (H > 0), the gradient of the
 * pre-activation (ops.rs:358-369, Q15) -- the previous layer's backward then needs no mask pass of its own
 * (th_linear_bwd with d_relu_y = NULL; the mask is idempotent, so passing it anyway is harmless). */
int th_linear_xent_head_masked(th_ctx *ctx, const float *d_h, const float *d_w, const float *d_bias, const float *d_targets,
                               int batch, int in_features, int classes, float *d_logits, float *d_loss, float *d_ncorrect,
                               float *d_dh, float *d_dw, float *d_db, float *d_metrics, int64_t metrics_capacity,
                               int64_t *d_state, int64_t advance, int32_t *d_adam_tick, const th_adam_fuse *w_fuse,
                               const th_adam_fuse *b_fuse, int mask_dh_by_h);

/* ---- classifier head on a WIDE input (in_features > 256: a CNN's Linear on the flattened feature map) ---- */
/* logits = X[B,in] . W[C,in]^T + b, the loss / accuracy count / step log / Adam tick of th_softmax_xent_fwd and the
 * backward products for an upstream gradient of exactly 1 -- d_dx[B,in] = dlogits . W (nullable), d_dw[C,in] =
 * dlogits^T . X, d_db[C] = colsum(dlogits) (nullable), all overwritten -- in TWO launches (K slices of the logits;
 * then one kernel whose workgroups each own 32 input columns and recompute the softmax of every 16-row block in
 * registers) instead of forward + slice reduce + loss + backward.  No parameter is updated: W is read by every
 * workgroup, so its Adam update is the caller's (th_adam_slice / th_adam_step).  batch <= 4096, classes <= 16. */
int th_linear_xent_wide(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, const float *d_targets,
                        int batch, int in_features, int classes, float *d_loss, float *d_ncorrect,
                        float *d_dx, float *d_dw, float *d_db, float *d_metrics, int64_t metrics_capacity,
                        int64_t *d_state, int64_t advance, int32_t *d_adam_tick);
/* The same two launches with one more output: d_colsum_masked[col] = sum over the rows of dX[row][col] * [x[row][col] > 0]
 * (written, not accumulated; d_dx may then be NULL: dX is not stored).  For an input that is the flattened output of a
 * bias-only Conv2dReLU (+ max-pool) -- faithful mode, Q2 -- these column sums are all its backward needs:
 * th_bias_from_colsum_adam finishes db[ch] = sum_j colsum[ch hw + j] (+ the bias's Adam update, + carried slices). */
int th_linear_xent_wide_ex(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, const float *d_targets, int batch,
                           int in_features, int classes, float *d_loss, float *d_ncorrect, float *d_dx, float *d_dw, float *d_db,
                           float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance, int32_t *d_adam_tick,
                           float *d_colsum_masked);
int th_bias_from_colsum_adam(th_ctx *ctx, const float *d_colsum, float *d_gb, int c, int hw, const th_adam_fuse *b_fuse,
                             const th_adam_slice *extra, int n_extra);
/* ... and the whole tail of such a step in those two launches: in the second one every workgroup also applies Adam (optim.rs:99-110)
 * to the columns of W it owns (no other workgroup reads them: dX is not stored in this form), the LAST workgroup to arrive to b (every workgroup reads b when it starts) -- from the lead's db --, and that same LAST
 * workgroup to arrive sums the conv bias gradient from everybody's column sums (d_conv_gb[conv_c], conv_c * conv_hw == in_features),
 * applies its Adam update and publishes the step counter.  The counter is ticked IN this launch (optim.rs:84): d_adam_tick points at
 * {t, arrival counter (0 between launches)}; the d_t fields of the three th_adam_fuse are ignored, every update uses t + 1.
 * Members with d_p == NULL are skipped; d_conv_gb == NULL: no conv bias in front. */
typedef struct th_wide_fuse {
    th_adam_fuse w, b, conv_b;
    float *d_conv_gb;
    int conv_c, conv_hw;
} th_wide_fuse;
int th_linear_xent_wide_fused(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, const float *d_targets, int batch,
                              int in_features, int classes, float *d_loss, float *d_ncorrect, float *d_dw, float *d_db,
                              float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance, int32_t *d_adam_tick,
                              float *d_colsum_masked, const th_wide_fuse *fuse);

/* ---- fused MLP tail: classifier head + the backward of the hidden layer ---- */
/* For  ... -> H = relu(X . W1^T + b1) -> logits = H . W2^T + b2 -> cross-entropy
 * ONE launch computes what th_linear_xent_head followed by
 * th_linear_bwd_adam_ex(d_relu_y = H) compute (nn.rs:54-60, loss.rs:101-195,
 * 271-290, backward closures ops.rs:238-294, 358-369, tensor.rs:574-587,
 * 674-694) for an upstream gradient of exactly 1:
 *   d_loss[1], d_ncorrect[1] (nullable), the step log (d_metrics / d_state as in
 *   th_softmax_xent_fwd, nullable),
 *   d_dw2[C,hid], d_db2[C]   (nullable; W2 / b2 are read by every workgroup, so their
 *                             Adam update is the caller's: th_adam_slice in a later launch),
 *   d_dw1[hid,in], d_db1[hid] (db1 nullable) -- overwritten (grad slots were None),
 *   + the fused Adam update of W1 / b1 (w1_fuse / b1_fuse nullable; their d_t must
 *   already be advanced for this step, e.g. by th_linear_fwd_ex).
 * Every workgroup recomputes the 16-row logits / softmax it needs in registers;
 * no dH buffer exists.  d_x[B,in] is the hidden layer's input, d_h[B,hid] its
 * post-ReLU output (16-byte aligned, like d_w2).  th_mlp_tail_supported: batch <= 512 (above, the chunks of a batch run
 * as a serial chain per wave and the row-parallel kernels win),
 * hidden <= 256 and a multiple of 4, classes <= 16.
 * d_dx[B,in] (nullable; needs d_w1[hid,in]): additionally dX = dZ1 . W1 for a hidden layer
 * that is not the first (ops.rs:254-265), overwritten; whole tiles only (need_dx: hidden
 * 32 / 64 / 128 / 256, batch and in_features multiples of 16) and, because the launch then
 * reads W1, w1_fuse must be NULL: W1's update is the caller's (th_adam_slice), like W2's. */
int th_mlp_tail_supported(int batch, int in_features, int hidden, int classes, int need_dx);
int th_mlp_tail(th_ctx *ctx, const float *d_x, const float *d_h, const float *d_w2, const float *d_b2,
                const float *d_targets, int batch, int in_features, int hidden, int classes,
                float *d_loss, float *d_ncorrect, float *d_dw1, float *d_db1, float *d_dw2, float *d_db2,
                const float *d_w1, float *d_dx,
                float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance,
                const th_adam_fuse *w1_fuse, const th_adam_fuse *b1_fuse);

/* A three-layer classifier -- Linear + ReLU, Linear + ReLU, Linear, softmax cross-entropy (the tail of examples/train_mnist_cnn.rs:53-61
 * behind the global average pool; examples/train_mnist.rs:40-48 behind the images) -- forward AND backward in TWO launches: everything but
 * the parameter gradients is row-parallel (nn.rs:54-60, loss.rs:101-195, ops.rs:254-265, 358-369), so launch 1 walks 16-row blocks through
 * the three forward and three backward products with the activations in LDS (and opens the optimizer step: *d_tick += 1, optim.rs:84, when
 * d_tick is given), launch 2 forms dW_l = dZ_l^T . A_l and db_l (ops.rs:266-294, tensor.rs:686-691) for all three layers with Adam
 * (optim.rs:99-110) in the epilogues where w_fuse / b_fuse are given -- no workgroup of that launch reads a parameter -- plus the loss
 * (loss.rs:164), the hit count (loss.rs:283) and the step log (th_log_step).  d_dx[B,in] (nullable): the gradient of the input, overwritten.
 * layers[l]: d_w [out][in] (16-byte aligned), d_b [out] (nullable), d_dw / d_db overwritten (nullable: not computed).
 * th_mlp3_supported: batch, in_features and both hidden sizes multiples of 16 (in <= 1024, hidden <= 256), classes <= 16. */
typedef struct th_mlp3_layer {
    const float *d_w, *d_b;
    float *d_dw, *d_db;
    const th_adam_fuse *w_fuse, *b_fuse;
    int out_features;
} th_mlp3_layer;
/* gap (nullable): d_x[B,in] are the plane means of a bias-only Conv2dReLU -> global average pool (th_conv3x3_gap_fwd / th_conv_chain_fwd)
 * with d_cnt[B,in] outputs > 0 per plane of hw elements: launch 2 then also forms that conv's bias gradient
 * d_gb[ch] = sum_n (dX[n][ch] / hw) * cnt[n][ch] (th_bias_grad_counts_adam's formula; needs d_dx) with Adam where b_fuse is given. */
typedef struct th_mlp3_gap {
    const float *d_cnt;
    float *d_gb;
    int hw;
    const th_adam_fuse *b_fuse;
} th_mlp3_gap;
int th_mlp3_supported(int batch, int in_features, int h1, int h2, int classes);
int th_mlp3_xent(th_ctx *ctx, const float *d_x, const float *d_targets, int batch, int in_features, const th_mlp3_layer *layers,
                 float *d_dx, float *d_loss, float *d_ncorrect, float *d_metrics, int64_t metrics_capacity, int64_t *d_state,
                 int64_t advance, int32_t *d_tick, const th_mlp3_gap *gap);

/* ---- th_mlp2_xent: the large-batch step of Linear + ReLU, Linear, softmax cross-entropy in THREE launches --------------------------
 * (nn.rs:54-60, activation.rs:10-12, loss.rs:101-195, 271-290, the backward closures ops.rs:238-294, 358-369, tensor.rs:574-587, 674-694,
 * and optim.rs:83-113 where fuses are given).  Bit-compatible in meaning with th_gather_batch + th_linear_fwd(relu) + th_linear_xent_head_masked
 * + th_linear_bwd + th_adam_step on the same rows; fp32 sums reordered (1e-4 relative), argmax / hit count exact.
 * The batch's rows are read where they lie: th_row_source names either a dense [batch][in_features] block (d_indices NULL; d_labels[r] is
 * row r's target) or a resident dataset (d_rows [n_rows][in_features], d_labels [n_rows]) whose rows d_indices[(d_cursor[0] + r) % n_indices]
 * form the batch -- DataLoader::next / MNISTDataset::get_batch (data/mnist.rs:277-310, 364-386) without the gathered copy.
 *   launch 1: rows (forward, loss terms, masked dZ1, partial dW2 / db1 / db2 / NLL / hits per row block); d_tick (nullable): t += 1 (optim.rs:84)
 *   launch 2: dW1 = dZ1^T X over K slices of the batch
 *   launch 3: fixed-order sums -> d_dw1 [hidden][in], d_db1, d_dw2 [classes][hidden], d_db2, d_loss, d_ncorrect, the step log, and Adam for
 *             every fuse given (complete gradients; no later launch of the step reads a parameter: nothing to defer).
 * Needs hidden a multiple of 4 up to 128 (a width that ends inside a wave's 16-column tile is padded with zero columns in the kernels:
 * 784-100-10 runs at the rate of 784-128-10), classes <= 16, in_features a multiple of 4, batch >= 32, n_rows * in_features * 4 < 2^31.
 * Nullable: d_b1, d_b2, d_db1, d_db2, d_ncorrect, metrics / state, d_tick, the fuses. */
typedef struct th_row_source {
    const float *d_rows;
    const float *d_labels;
    const int32_t *d_indices;   /* nullable */
    const int64_t *d_cursor;    /* nullable: 0 */
    int64_t n_indices;
    int64_t n_rows;             /* rows held by d_rows (the extent reads are clamped to) */
} th_row_source;
int th_mlp2_xent_supported(int batch, int in_features, int hidden, int classes, int64_t n_rows);
/* Cap (1 .. 8, default 8) on the workgroups that share one 16-row block's k chunks in launch 1 at batches below 4096; 1 = no hand-off
 * between workgroups inside the launch (launch 1 at 1 024 rows: 15.6 instead of 11.9 us).  A measurement / fallback switch. */
int th_mlp2_set_max_ksplit(th_ctx *ctx, int max_ksplit);
int th_mlp2_xent(th_ctx *ctx, const th_row_source *src, int batch, int in_features, int hidden, int classes, const float *d_w1,
                 const float *d_b1, const float *d_w2, const float *d_b2, float *d_dw1, float *d_db1, float *d_dw2, float *d_db2,
                 float *d_loss, float *d_ncorrect, float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance,
                 int32_t *d_tick, const th_adam_fuse *w1_fuse, const th_adam_fuse *b1_fuse, const th_adam_fuse *w2_fuse,
                 const th_adam_fuse *b2_fuse);

/* The same step for TWO hidden layers -- Linear + ReLU, Linear + ReLU, Linear, softmax cross-entropy: the model of examples/train_mnist.rs:40-48
 * (784-128-64-10) at large batch, still three launches.  The second hidden layer is a contraction on the hidden tile a row block holds in
 * LDS anyway, and everything behind it stays row-parallel inside launch 1 (A2, the classifier, dZ2, dA1, the masked dZ1, the block's share
 * of dW2 = dZ2^T A1 beside the small sums); only dW1 needs the batch-wide launch; launch 3 adds the blocks' shares and applies Adam for
 * every fuse given.  layers[0..2] as in th_mlp3_xent (d_w [out][in] 16-byte aligned; d_dw required, d_b / d_db nullable); the first
 * hidden size and the second multiples of 4, both <= 128; classes <= 16.  Same row source, step log and tick as th_mlp2_xent. */
int th_mlp2_xent_deep_supported(int batch, int in_features, int h1, int h2, int classes, int64_t n_rows);
int th_mlp2_xent_deep(th_ctx *ctx, const th_row_source *src, int batch, int in_features, const th_mlp3_layer *layers, float *d_loss,
                      float *d_ncorrect, float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance, int32_t *d_tick);

/* ---- element-wise: src/ops.rs:8-120,377-496; src/tensor.rs:36-161 ----- */
int th_add(th_ctx *ctx, const float *d_a, const float *d_b, float *d_out, size_t n);
int th_sub(th_ctx *ctx, const float *d_a, const float *d_b, float *d_out, size_t n);
int th_mul(th_ctx *ctx, const float *d_a, const float *d_b, float *d_out, size_t n);
int th_div(th_ctx *ctx, const float *d_a, const float *d_b, float *d_out, size_t n);
/* y += alpha * x : accumulate_grad (alpha=1) / accumulate_grad_scaled, ops.rs:124-151 */
int th_axpy(th_ctx *ctx, float alpha, const float *d_x, float *d_y, size_t n);
/* g += gout * other : Mul backward, ops.rs:81-116 */
int th_mul_bwd(th_ctx *ctx, const float *d_gout, const float *d_other, float *d_g, size_t n);
/* Div backward, ops.rs:466-492: ga += gout / b ; gb -= gout * a / b^2 (either NULL to skip) */
int th_div_bwd(th_ctx *ctx, const float *d_gout, const float *d_a, const float *d_b, float *d_ga, float *d_gb, size_t n);
/* g += d_scalar[0] / divisor (scalar broadcast): sum(None) backward (divisor 1,
 * tensor.rs:1006-1013) and mean backward (divisor n, tensor.rs:785-796) */
int th_add_scalar_dev(th_ctx *ctx, const float *d_scalar, float divisor, float *d_g, size_t n);

int th_relu_fwd(th_ctx *ctx, const float *d_x, float *d_y, size_t n);                 /* ops.rs:312-349 */
/* ops.rs:358-369: gin (+)= x>0 ? g : 0; accumulate=0 overwrites (grad slot was None) */
int th_relu_bwd(th_ctx *ctx, const float *d_x, const float *d_gout, float *d_gin, size_t n, int accumulate);
int th_sigmoid_fwd(th_ctx *ctx, const float *d_x, float *d_y, size_t n);              /* tensor.rs:594-608 */
int th_sigmoid_bwd(th_ctx *ctx, const float *d_y, const float *d_gout, float *d_gin, size_t n); /* tensor.rs:618-629 */
int th_exp_fwd(th_ctx *ctx, const float *d_x, float *d_y, size_t n);                  /* tensor.rs:1091-1099 */
int th_log_fwd(th_ctx *ctx, const float *d_x, float *d_y, size_t n);                  /* tensor.rs:1136-1142 */
int th_log_bwd(th_ctx *ctx, const float *d_x, const float *d_gout, float *d_gin, size_t n); /* tensor.rs:1151-1165 */
int th_pow_fwd(th_ctx *ctx, const float *d_x, float e, float *d_y, size_t n);         /* tensor.rs:1172-1178 */
int th_pow_bwd(th_ctx *ctx, const float *d_x, float e, const float *d_gout, float *d_gin, size_t n); /* tensor.rs:1188-1202 */

/* ---- post-training-quantization storage codecs: src/tensor.rs:2110-2288 (bit-exact restatements) ---- */
/* IEEE half <-> fp32 as the reference hand-rolls it (tensor.rs:2191-2287): round half UP on the 13 dropped bits with the
 * mantissa carry OR-ed into the exponent field, truncating denormals, NaN -> 0x7E00-style quiet pattern. */
int th_f32_to_f16(th_ctx *ctx, const float *d_x, uint16_t *d_y, size_t n);
int th_f16_to_f32(th_ctx *ctx, const uint16_t *d_x, float *d_y, size_t n);
/* int8 affine quantisation (tensor.rs:2110-2152): min / max over the FINITE elements (+-0.1 when equal),
 * scale = (max - min) / 255, q = clamp(round((x - min) / scale) as i32 - 128, -128, 127).  d_params[2] receives
 * {min_val, scale} on the device; zero_point is -128.  Dequantise (tensor.rs:353-360): (q - zero_point) * scale + min_val. */
int th_quantize_int8(th_ctx *ctx, const float *d_x, int8_t *d_q, size_t n, float *d_params);
int th_dequantize_int8(th_ctx *ctx, const int8_t *d_q, float *d_y, size_t n, float scale, int zero_point, float min_val);

/* ---- broadcast / reduce / layout: src/tensor.rs ---------------------- */
int th_transpose2d(th_ctx *ctx, const float *d_in, float *d_out, int rows, int cols);      /* tensor.rs:544-566 */
/* dst[r*dst_ld + c] = src[r*src_ld + c] for r < rows, c < cols: the strided block copies behind slice_channels /
 * slice_output_channels / cat of a grouped convolution (nn.rs:862-1014) */
int th_copy2d(th_ctx *ctx, const float *d_src, float *d_dst, int64_t rows, int cols, int64_t src_ld, int64_t dst_ld);
int th_transpose2d_bwd(th_ctx *ctx, const float *d_gout, float *d_gin, int rows, int cols); /* tensor.rs:574-587: gin[i,j] += gout[j,i] */
int th_bias_add_rows(th_ctx *ctx, const float *d_x, const float *d_bias, float *d_y, int rows, int cols, int relu); /* tensor.rs:658-663 */
int th_colsum_accum(th_ctx *ctx, const float *d_g, float *d_gb, int rows, int cols);       /* tensor.rs:686-691: gb[f] += sum_b g[b,f] */
int th_sub_rows(th_ctx *ctx, const float *d_x, const float *d_r, float *d_y, int rows, int cols); /* tensor.rs:730-736 */
int th_rowsum_neg_accum(th_ctx *ctx, const float *d_g, float *d_gr, int rows, int cols);   /* tensor.rs:752-764: gr[row] -= sum_c g[row,c] */
int th_rowsum(th_ctx *ctx, const float *d_x, float *d_y, int rows, int cols);              /* tensor.rs:890-939 (2-D, dim=1) */
int th_colsum(th_ctx *ctx, const float *d_x, float *d_y, int rows, int cols);              /* tensor.rs:890-939 (2-D, dim=0) */
int th_rowsum_bwd(th_ctx *ctx, const float *d_gout, float *d_gin, int rows, int cols);     /* tensor.rs:949-991 (dim=1): gin[r,c] += gout[r] */
int th_colsum_bwd(th_ctx *ctx, const float *d_gout, float *d_gin, int rows, int cols);     /* tensor.rs:949-991 (dim=0): gin[r,c] += gout[c] */
int th_sum_all(th_ctx *ctx, const float *d_x, float *d_out1, size_t n, float divisor);     /* out = sum / divisor: tensor.rs:996-998 (1), 772-775 mean (n) */
/* row max + first-max index (strict >, NaN never wins; index stored as f32
 * like the reference): tensor.rs:1021-1071, 1086-1088 */
int th_rowmax(th_ctx *ctx, const float *d_x, float *d_max, float *d_argmax_f32, int rows, int cols);
int th_colmax(th_ctx *ctx, const float *d_x, float *d_max, float *d_argmax_f32, int rows, int cols);
/* sum(dim) / max(dim) over ANY dimension of a tensor of 1..4 dimensions (shape: host array), restating the reference's per-element index
 * arithmetic -- tensor.rs:917-937 (sum: [outer, d, inner] -> [outer, inner], a row's terms in input order: bit-exact), 960-994 (its backward:
 * d_gin[i] += d_gout[the reference's index], quirk included: a dimension whose position is not below the output's element count is skipped),
 * 1042-1066 (max: values + indices along dim as f32; strict >, the first of equals; NaN / -inf never win; quirk Q14 included: on rank > 2
 * the output index only steps for dimensions in front of dim, so several positions share an output element).  Off the training step. */
int th_sum_dim(th_ctx *ctx, const float *d_x, float *d_y, const int64_t *shape, int ndim, int dim);
int th_sum_dim_bwd(th_ctx *ctx, const float *d_gout, float *d_gin, const int64_t *shape, int ndim, int dim, int keepdim);
int th_max_dim(th_ctx *ctx, const float *d_x, float *d_max, float *d_argmax_f32, const int64_t *shape, int ndim, int dim);
/* global max over n elements, tensor.rs:1072-1083: `max_by(partial_cmp)` keeps the LAST of equal maxima; the index is stored as f32
 * (`max_idx as f32`); n == 0 gives (0.0, 0).  d_nan_flag[0] (nullable) = 1 when a NaN is present: the reference's
 * `partial_cmp(..).unwrap()` panics there, the host mirror raises. */
int th_global_max(th_ctx *ctx, const float *d_x, size_t n, float *d_max, float *d_argmax_f32, int *d_nan_flag);

/* ---- fused softmax cross-entropy: src/loss.rs:101-195, 271-290 -------- */
/* logits [B,C], targets [B] fp32 class ids (cast `as usize`).  Writes
 * logp[B,C] (nullable), loss[1] = -(1/B) sum_i logp[i,t_i], argmax[B] as f32
 * (nullable), n_correct[1] as f32 (nullable; accuracy()*B).  Out-of-range
 * target -> NaN loss (the reference panics, loss.rs:161).
 * d_dlogits_unit (nullable) receives (softmax - onehot) * (1/B): the gradient
 * of loss.rs:174-191 for an upstream grad of exactly 1, so that
 * loss.backward() needs no further launch.
 * d_metrics / d_state (nullable): th_log_step folded into this kernel. */
int th_softmax_xent_fwd(th_ctx *ctx, const float *d_logits, const float *d_targets, int batch, int classes,
                        float *d_logp, float *d_loss, float *d_argmax, float *d_ncorrect, float *d_dlogits_unit,
                        float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance,
                        int32_t *d_adam_tick /* nullable: t += 1 here, see th_adam_step pre_ticked */);
/* dlogits (+)= (exp(logp) - onehot) * (g0 / B), g0 read from device (the
 * upstream scalar grad, loss.rs:174-191); accumulate=0 overwrites */
int th_softmax_xent_bwd(th_ctx *ctx, const float *d_logp, const float *d_targets, const float *d_g0,
                        int batch, int classes, float *d_dlogits, int accumulate);
int th_log_softmax_fwd(th_ctx *ctx, const float *d_x, float *d_logp, int rows, int cols); /* loss.rs:101-126 */
int th_accuracy_count(th_ctx *ctx, const float *d_argmax, const float *d_targets, int n, float *d_ncorrect); /* loss.rs:281-287 */

/* ---- other losses / layers: src/loss.rs:6-73, 201-245; src/nn.rs:798-822 */
/* loss[0] = -mean(y ln(p') + (1-y) ln(1-p')), p' = clamp(p, 1e-7, 1-1e-7)  (loss.rs:16-23) */
int th_bce_fwd(th_ctx *ctx, const float *d_pred, const float *d_targets, size_t n, float *d_loss1);
/* loss.rs:41-66; d_gpred / d_gtargets nullable; accumulate_mask bit0 / bit1: += into that grad */
int th_bce_bwd(th_ctx *ctx, const float *d_pred, const float *d_targets, const float *d_g0, size_t n, float *d_gpred,
               float *d_gtargets, int accumulate_mask);
/* cross_entropy_loss_onehot backward, loss.rs:226-240: dlogits (+)= (exp(logp) - targets) * g0 / batch */
int th_xent_onehot_bwd(th_ctx *ctx, const float *d_logp, const float *d_targets, const float *d_g0, int batch, int classes,
                       float *d_dlogits, int accumulate);
/* Dropout mask, nn.rs:808-818: mask[i] = u_i > p ? 1/(1-p) : 0 with u_i = splitmix64(seed, i) in [0,1) */
int th_dropout_mask(th_ctx *ctx, float *d_mask, size_t n, float p, uint64_t seed);

/* ---- conv / pool: src/tensor.rs:1221-2081 ----------------------------- */
/* Direct 3x3 stride-1 convolution, NCHW, fused bias (nullable) and ReLU.
 * weight_layout 0 = taper's reinterpretation (tensor.rs:1262, quirk Q3):
 *   w_eff[co][k] = w_flat[k*C_out + co], k = ci*9 + kh*3 + kw;
 * weight_layout 1 = standard [co][ci][3][3] (tensor.rs:1329,1348).
 * Replaces im2col (1728-1780) + matmul + reshape + transpose_4d (2034-2076)
 * + add_bias_4d (1983-1992) (+ relu). */
int th_conv3x3_fwd(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, float *d_y,
                   int n, int c_in, int h, int w, int c_out, int pad, int weight_layout, int relu);
/* conv3x3 (taper weight layout, Q3) + bias [+ ReLU] + 2x2 / stride-2 max pool in one launch: only the pooled tensor
 * [n][c_out][h_out/2][w_out/2] is written (values of tensor.rs:1391-1470; no index output) -- the Conv2dReLU -> MaxPool2d
 * pair of the CNNs without the full-resolution round trip.  For a conv whose output nobody else reads and whose pool
 * needs no scatter backward (faithful mode, Q2: see th_bias_grad_nchw_masked).  th_conv3x3_pool2_supported: the
 * matrix-core path (c_in >= 8 or == 1), c_out % 4 == 0, even output height and width, output rows of <= 64 pixels. */
int th_conv3x3_pool2_supported(int c_in, int h, int w, int c_out, int pad);
int th_conv3x3_pool2_fwd(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, float *d_y_pooled,
                         int n, int c_in, int h, int w, int c_out, int pad, int relu);
/* Conv2dReLU(3x3, stride 1) -> GLOBAL average pool (tensor.rs:1221-1285 + 1524-1660 with kernel = the whole plane, nn.rs:670-686) as ONE
 * launch of the image-resident kernel: d_y_mean[n][c_out] = plane means of relu(conv + bias), d_cnt[n][c_out] (nullable) = how many
 * outputs of the plane are > 0 -- all a bias-only conv's backward needs of the map (th_bias_grad_counts_adam); the map itself is
 * neither written nor re-read.  Same bits as th_conv3x3_fwd + th_avgpool2d_global_fwd_counts.  Taper weight layout. */
int th_conv3x3_gap_supported(int n, int c_in, int h, int w, int c_out, int pad);
int th_conv3x3_gap_fwd(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, float *d_y_mean, float *d_cnt,
                       int n, int c_in, int h, int w, int c_out, int pad, int relu);
/* A run of Conv2dReLU(3x3, stride 1, pad 1) layers with their pools -- the convolutional front of a Sequential
 * (examples/train_mnist_cnn.rs:35-100; nn.rs:433-490 Conv2dReLU, 622-655 MaxPool2d, 670-686 AdaptiveAvgPool2d) -- as ONE launch: a
 * workgroup carries one image through every stage with the activations resident in LDS; only the last stage's output is written.
 * Stage i: d_w taper layout [9 c_in_i][c_out] (tensor.rs:1262), d_bias [c_out], then `post`.  For a bias-only backward (faithful mode,
 * Q2): intermediate maps are neither written nor available afterwards.  d_y: [n][c_out] plane means when the last stage ends in
 * TH_CHAIN_GLOBAL_AVG (d_cnt [n][c_out], nullable: outputs > 0 per plane, see th_conv3x3_gap_fwd), else the last stage's (pooled) NCHW map.
 * Same bits as the layer-by-layer launches at batch >= 128.  th_conv_chain_supported: 0 = these stages do not run as a chain (the caller
 * launches the layers one by one); 1, 2 = an instance with its sizes compiled in (1: 28x28, 1 -> 32, 32 -> 32 + pool, 32 -> 64, 64 -> 64 + pool,
 * 64 -> 128 + global mean; 2: 28x28, 1 -> 32 + pool, 32 -> 64 + pool); 3 = the kernel that takes its stages as ARGUMENTS: any run of up to
 * 8 stages on a square input of 4 .. 32 pixels a side with 1 or a multiple of 16 (<= 256) channels, c_out a multiple of 16 (<= 512),
 * pools on even maps, ending in a pool, in the global average or in a conv row (d_y: that row's NCHW map), whose maps fit the 160 KB of
 * LDS (the plan -- tile-to-wave mapping, LDS offsets -- is worked out on the host per launch; 28 / 14 / 7 maps run instances with the
 * size compiled in, other sizes one that takes it as an argument); a launch walks min(n, 256) workgroups over the images. */
enum { TH_CHAIN_NONE = 0, TH_CHAIN_MAXPOOL2 = 1, TH_CHAIN_GLOBAL_AVG = 2 };
typedef struct th_conv_stage {
    const float *d_w, *d_bias;
    int c_out;
    int post; /* TH_CHAIN_* */
} th_conv_stage;
int th_conv_chain_supported(int c_in, int h, int w, const th_conv_stage *stages, int n_stages);
int th_conv_chain_fwd(th_ctx *ctx, const float *d_x, const th_conv_stage *stages, int n_stages, float *d_y, float *d_cnt, int n,
                      int c_in, int h, int w);
/* The same launch with the classifier behind it -- Flatten + Linear(c_out h w, classes) + softmax cross-entropy (nn.rs:54-60, 730-756,
 * loss.rs:101-195, 271-290) -- ROW by row in the chain's last epilogue: everything of that head but the sums over the batch.  The workgroup
 * that holds image i's pooled map also writes d_dl[i][16] = dlogits of the row for an upstream gradient of exactly 1 (zeros past `classes`),
 * d_rowstat[i][2] = {-log p[target], 1 if the first arg-max is the target}, and -- d_cbpart [n][c_out], nullable -- the per-channel sums of
 * dX[i] * [x[i] > 0]: all a bias-only last Conv2dReLU + MaxPool2d needs of its gradient (tensor.rs:1496-1519).  d_tick (nullable): Adam's
 * step counter, advanced by one (optim.rs:84).  th_wide_head_grads then forms dW, db, the loss, the hit count and the conv bias gradient
 * (+ their Adam updates) in ONE more launch: the simple CNN's step is two launches.  d_y: the pooled NCHW map as in th_conv_chain_fwd. */
typedef struct th_chain_head {
    const float *d_w, *d_bias, *d_targets; /* [classes][c_out h w], [classes] (nullable), [n] */
    int classes;
    float *d_dl, *d_rowstat, *d_cbpart;
    int32_t *d_tick;
} th_chain_head;
int th_conv_chain_head_supported(int c_in, int h, int w, const th_conv_stage *stages, int n_stages, int classes);
int th_conv_chain_head_fwd(th_ctx *ctx, const float *d_x, const th_conv_stage *stages, int n_stages, float *d_y, int n, int c_in, int h, int w,
                           const th_chain_head *head);
/* The batch sums behind th_conv_chain_head_fwd (ops.rs:280-291 through the W^T node, tensor.rs:686-691, loss.rs:164, 283):
 * d_dw[classes][in] = dl^T . X, d_db[classes] = colsum(dl) (nullable), d_loss[1] = sum(nll) / batch, d_ncorrect[1] (nullable), the step log
 * (th_log_step; d_metrics nullable), and d_conv_gb[conv_c] = sum_i d_cbpart[i][.] (both nullable).  No workgroup reads a parameter, so every
 * Adam update (optim.rs:99-110; fuse descriptors nullable, d_t ALREADY ticked) rides in the epilogue of the workgroup that owns the gradient. */
int th_wide_head_grads(th_ctx *ctx, const float *d_x, const float *d_dl, const float *d_rowstat, const float *d_cbpart, int batch,
                       int in_features, int classes, int conv_c, float *d_dw, float *d_db, float *d_conv_gb, float *d_loss, float *d_ncorrect,
                       float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance, const th_adam_fuse *w_fuse,
                       const th_adam_fuse *b_fuse, const th_adam_fuse *cb_fuse);

/* th_conv_chain_fwd + th_mlp3_xent in TWO launches instead of three -- the front AND the classifier of examples/train_mnist_cnn.rs:35-100
 * (five Conv2dReLU rows with their pools and the global average pool, then Linear(128,128) + ReLU, Linear(128,64) + ReLU, Linear(64,classes),
 * softmax cross-entropy): the classifier is row-parallel except for its parameter gradients (th_mlp3_xent above), and the chain launch
 * holds ONE image per workgroup -- so the image's row of the classifier (forward, loss term, dlogits, the activations' gradients down to the
 * plane means') runs in the chain launch's last epilogue, on the means it has just formed; launch 2 is th_mlp3_xent's (dW / db of the three
 * layers over the batch, Adam in the epilogues, the conv bias finish `gap`, loss, hit count, step log).  The classifier's row launch was
 * 10.7 us of the 111 us step.  d_y [n][128] plane means and d_cnt [n][128] (nullable) as th_conv_chain_fwd; the rest as th_mlp3_xent with
 * d_x = d_y.  _supported: 1 for the compiled reference front (th_conv_chain_supported == 1) ending in 128 plane means, a 128-128-64-classes
 * classifier (classes <= 16), n a multiple of 16. */
int th_conv_chain_mlp3_supported(int c_in, int h, int w, const th_conv_stage *stages, int n_stages, int n, int h1, int h2, int classes);
int th_conv_chain_mlp3_xent(th_ctx *ctx, const float *d_x, const th_conv_stage *stages, int n_stages, float *d_y, float *d_cnt, int n, int c_in,
                            int h, int w, const float *d_targets, const th_mlp3_layer *layers, float *d_dx, float *d_loss, float *d_ncorrect,
                            float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance, int32_t *d_tick, const th_mlp3_gap *gap);

/* 1x1 stride-1 pad-0 convolution as GEMM.  layout 0 = taper (raw NCHW buffer
 * reinterpreted as [N*H*W, C], tensor.rs:1799-1801, Q4 + Q3); 1 = standard. */
int th_conv1x1_fwd(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, float *d_y,
                   int n, int c_in, int h, int w, int c_out, int weight_layout, int relu);
/* General convolution (any kernel / stride / padding / dilation that is neither 3x3-stride-1 nor 1x1): the reference's
 * im2col_general_simd (tensor.rs:1805-1906, copy_consecutive_elements 1910-1969) + matmul against the weight viewed
 * [C_in*K_h*K_w, C_out] (1262, Q3) + NHWC -> NCHW (1275-1276) + add_bias_4d (1279-1282) [+ ReLU].  The gather keeps
 * the reference's arithmetic: taps of one kernel row are copied from consecutive input columns starting at the first
 * in-range tap (dilation is not applied inside the run) and the source plane is batch*ch + ch (Q9). */
int th_conv2d_general_fwd(th_ctx *ctx, const float *d_x, const float *d_w, const float *d_bias, float *d_y,
                          int n, int c_in, int h, int w, int c_out, int k_h, int k_w, int s_h, int s_w,
                          int pad_h, int pad_w, int dil_h, int dil_w, int relu);
/* add_bias_4d fwd/bwd: tensor.rs:1983-1992, 2017-2024 (gb[c] += sum_{n,h,w}) */
int th_bias_add_nchw(th_ctx *ctx, const float *d_x, const float *d_bias, float *d_y, int n, int c, int hw, int relu);
int th_bias_grad_nchw(th_ctx *ctx, const float *d_gout, float *d_gb, int n, int c, int hw);
/* same with the ReLU backward folded in (d_mask_y nullable: only elements whose d_mask_y value is > 0 count,
 * ops.rs:358-369) and an overwrite form (accumulate == 0: the grad slot was None, no zero-fill needed).
 * With d_gout = the gradient of a max-pool's OUTPUT and d_mask_y = that output, this is the bias gradient of a
 * Conv2dReLU -> MaxPool2d pair whose conv output has no other consumer: every pooled gradient lands on exactly
 * one conv output (tensor.rs:1496-1519), whose ReLU mask is "pooled value > 0" -- no scatter, no dZ buffer. */
int th_bias_grad_nchw_masked(th_ctx *ctx, const float *d_gout, const float *d_mask_y, float *d_gb, int n, int c, int hw,
                             int accumulate);
/* the same shortcut behind a GLOBAL average pool (AdaptiveAvgPool2d((1,1)), nn.rs:670-686): d_gout_pooled is [n][c], the
 * gradient of the pool's output; every element of plane (b, ch) receives d_gout_pooled[b][ch] / hw (tensor.rs:1626-1628)
 * and counts where d_mask_y[b][ch][.] > 0.  db[ch] (+)= sum over the batch and the plane. */
int th_bias_grad_avgpool_masked(th_ctx *ctx, const float *d_gout_pooled, const float *d_mask_y, float *d_gb, int n, int c, int hw,
                                int accumulate);
/* either form as the LAST backward launch of a fused step: one workgroup per channel, gradient overwritten (grad slot was None),
 * the bias's Adam update in the epilogue (b_fuse nullable) and up to TH_MAX_ADAM_SLICES deferred updates of other parameters
 * carried in spare workgroups (extra nullable) -- the step then needs no optimizer launch.  pooled_avg: d_gout is [n][c]. */
int th_bias_grad_masked_adam(th_ctx *ctx, const float *d_gout, const float *d_mask_y, float *d_gb, int n, int c, int hw,
                             int pooled_avg, const th_adam_fuse *b_fuse, const th_adam_slice *extra, int n_extra);
/* Global average pool of a post-ReLU map that also counts the elements > 0 of every plane (d_cnt [n][c], as floats), and the
 * bias gradient of the Conv2dReLU in front of it from those counts: db[ch] = sum_n (g[n][ch] / hw) * cnt[n][ch] -- every
 * element of a plane receives g / hw (tensor.rs:1626-1628) and passes the ReLU mask iff it is > 0 (ops.rs:358-369), so the
 * sum over the plane (tensor.rs:2017-2024) is that product; 2 n c floats are read instead of the n c hw conv outputs.
 * b_fuse / extra as in th_bias_grad_masked_adam. */
int th_avgpool2d_global_fwd_counts(th_ctx *ctx, const float *d_x, float *d_y, float *d_cnt, int n, int c, int hw);
int th_bias_grad_counts_adam(th_ctx *ctx, const float *d_gout_pooled, const float *d_cnt, float *d_gb, int n, int c, int hw,
                             const th_adam_fuse *b_fuse, const th_adam_slice *extra, int n_extra);
/* full_backward extension (not in the reference: Q2 cuts these gradients) */
/* accumulate != 0: gx / gw += (a gradient slot that is already Some, ops.rs:124-151); 0: overwrite (the slot was None: no zero fill) */
int th_conv3x3_bwd_input(th_ctx *ctx, const float *d_gy, const float *d_w, float *d_gx,
                         int n, int c_in, int h, int w, int c_out, int pad, int weight_layout, int accumulate);
int th_conv3x3_bwd_weight(th_ctx *ctx, const float *d_x, const float *d_gy, float *d_gw,
                          int n, int c_in, int h, int w, int c_out, int pad, int weight_layout, int accumulate);

/* max-pool: tensor.rs:1391-1521.  s_h == 0 -> stride = kernel (1403).
 * d_argmax: int64 absolute flat input index per output (1449-1461). */
int th_maxpool2d_fwd(th_ctx *ctx, const float *d_x, float *d_y, int64_t *d_argmax, int n, int c, int h, int w,
                     int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w);
/* zero_first=1: zero each (b,c) plane of gin before scatter-add (1496-1500, Q5) */
int th_maxpool2d_bwd(th_ctx *ctx, const float *d_gout, const int64_t *d_argmax, float *d_gin,
                     int n, int c, int h, int w, int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w,
                     int zero_first);
/* The same scatter (zero_first = 1) for a pool whose input is the output of a ReLU (a Conv2dReLU row, nn.rs:433-490), with that ReLU's
 * backward (ops.rs:358-369: gradient kept where the output is > 0) folded in: d_gin = relu_bwd(max_pool2d_bwd(d_gout)), bit for bit,
 * without a pass over the full-resolution map -- the pixel a window's maximum came from has the POOLED value as its output.
 * 2x2 windows, stride 2, no padding, even h and w only (_supported).  d_y_pooled: the pool's output; d_y_full: its input (read for
 * pixel (0,0) of planes with a default-index window only, tensor.rs:1432).  full_backward extension. */
int th_maxpool2d_relu_bwd_supported(int n, int c, int h, int w, int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w);
int th_maxpool2d_relu_bwd(th_ctx *ctx, const float *d_gout, const int64_t *d_argmax, const float *d_y_pooled,
                          const float *d_y_full, float *d_gin, float *d_plane_sums /* nullable: [n][c] */, int n, int c, int h, int w);
/* d_gin = relu_bwd(d_y, d_gout) (ops.rs:358-369) on an NCHW map, and d_plane_sums[b][ch] = the sum of plane (b, ch) of d_gin: the rows of
 * the bias sum of the Conv2dReLU whose output d_y is (tensor.rs:2017-2024).  th_bias_grad_plane_sums: gb[ch] (+)= sum_b plane_sums[b][ch]
 * -- together th_relu_bwd + th_bias_grad_nchw in one pass over the map.  full_backward extension. */
int th_relu_bwd_plane_sums(th_ctx *ctx, const float *d_y, const float *d_gout, float *d_gin, float *d_plane_sums, int n, int c, int hw);
int th_bias_grad_plane_sums(th_ctx *ctx, const float *d_plane_sums, float *d_gb, int n, int c, int accumulate);
/* Backward of a GLOBAL average pool (tensor.rs:1626-1628) over the output d_y of a ReLU, into a gradient slot that is None, with that
 * ReLU's backward folded in: d_gin = relu_bwd(d_y, avg_pool2d_bwd(d_gout [n][c])), and d_plane_sums [n][c] (nullable) as above.
 * full_backward extension. */
int th_avgpool2d_global_relu_bwd(th_ctx *ctx, const float *d_gout, const float *d_y, float *d_gin, float *d_plane_sums,
                                 int n, int c, int hw);
/* avg-pool: tensor.rs:1524-1660 (divisor k_h*k_w incl. padding, Q6) */
int th_avgpool2d_fwd(th_ctx *ctx, const float *d_x, float *d_y, int n, int c, int h, int w,
                     int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w);
int th_avgpool2d_bwd(th_ctx *ctx, const float *d_gout, float *d_gin, int n, int c, int h, int w,
                     int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w); /* gin += */

/* ---- optimizers: src/optim.rs ----------------------------------------- */
/* Fused multi-tensor Adam over flat arenas (optim.rs:83-113).
 * d_offsets[n_tensors+1]: element offsets of each parameter tensor (int64);
 * d_has_grad[n_tensors]: 0 -> the tensor is skipped entirely (grad None, Q8).
 * d_t: int32[2] ON DEVICE: [0] = step counter t, advanced by this call
 * (optim.rs:84) so a captured graph advances it on every replay; [1] = the
 * kernel's arrival counter (must be 0 between calls).  The bias corrections
 * use powi (square-and-multiply) like f32::powi.
 * d_lr[1]: learning rate on device (set_lr, optim.rs:125-127). */
int th_adam_step(th_ctx *ctx, float *d_params, const float *d_grads, float *d_m, float *d_v,
                 const int64_t *d_offsets, const int32_t *d_has_grad, int n_tensors, int64_t total,
                 int32_t *d_t, const float *d_lr, float beta1, float beta2, float eps, float weight_decay,
                 int pre_ticked /* 1: d_t[0] was already advanced for this step; use it as is */);
/* same; d_skip_if_nonzero (nullable): a device word written by an EARLIER launch -- the error word of the peer-to-peer communicator whose
 * all-reduce produced d_grads (th_comm_error_word).  Non-zero: that all-reduce timed out, d_grads still holds this rank's own gradients,
 * and the launch applies nothing (no update, no tick of t). */
int th_adam_step_guarded(th_ctx *ctx, float *d_params, const float *d_grads, float *d_m, float *d_v,
                         const int64_t *d_offsets, const int32_t *d_has_grad, int n_tensors, int64_t total,
                         int32_t *d_t, const float *d_lr, float beta1, float beta2, float eps, float weight_decay,
                         int pre_ticked, const uint32_t *d_skip_if_nonzero);
/* t += 1 alone (optim.rs:84), for steps whose updates all ran fused */
int th_adam_tick(th_ctx *ctx, int32_t *d_t);
/* the deferred updates nobody carried (one launch for all n <= TH_MAX_ADAM_SLICES slices) */
int th_adam_slices(th_ctx *ctx, const th_adam_slice *slices, int n);
/* x *= scale in place: AdamW's decoupled decay `w *= 1 - lr*wd` (optim.rs:150-159) */
int th_scale(th_ctx *ctx, float *d_x, size_t n, float scale);
int th_sgd_step(th_ctx *ctx, float *d_params, const float *d_grads, const int64_t *d_offsets,
                const int32_t *d_has_grad, int n_tensors, int64_t total, const float *d_lr); /* optim.rs:21-33 */

/* ---- data: src/data/mnist.rs:277-310 get_batch ------------------------ */
/* out_images[i,:] = images[idx[i],:], out_labels[i] = labels[idx[i]], with
 * idx = d_indices[(*d_cursor + i) % n_indices] (int32); d_cursor (int64, on
 * device, advanced by th_log_step) lets a captured graph walk the epoch. */
int th_gather_batch(th_ctx *ctx, const float *d_images, const float *d_labels, const int32_t *d_indices,
                    int64_t n_indices, const int64_t *d_cursor, int batch, int row_len,
                    float *d_out_images, float *d_out_labels);
/* u8 -> f32 / 255 (data/mnist.rs:224-229) */
int th_u8_to_unit_f32(th_ctx *ctx, const uint8_t *d_in, float *d_out, size_t n);
/* End-of-step bookkeeping kept on device so a step never synchronises
 * (examples/train_mnist.rs:110-111,121 read loss / accuracy on the host every
 * step): metrics[2*s] = loss[0], metrics[2*s+1] = ncorrect[0] with
 * s = d_state[0] % capacity; then d_state[0] += 1 (step) and
 * d_state[1] += advance (sample cursor used by th_gather_batch). */
int th_log_step(th_ctx *ctx, const float *d_loss, const float *d_ncorrect, float *d_metrics, int64_t capacity,
                int64_t *d_state, int64_t advance);

/* ---- data parallel (new; the reference has no collective) ------------- */
/* One process per GPU.  Rank 0 calls th_comm_unique_id and ships the 128
 * bytes to the other ranks out of band (torch.distributed / a file / MPI). */
int th_comm_unique_id(uint8_t out_id[128]);
int th_comm_init_rank(th_ctx *ctx, int n_ranks, int rank, const uint8_t id[128], th_comm **out);
int th_comm_destroy(th_comm *comm);
/* in-place sum over ranks of d_buf[n] then * scale (1/W): grads of the flat
 * arena between loss.backward() and optim.step() */
int th_allreduce_sum_scale(th_comm *comm, th_ctx *ctx, float *d_buf, size_t n, float scale);
/* Peer-to-peer communicator (one node, <= 8 ranks): a ONE-SHOT all-reduce for latency-bound buffers (the MLP's 407 KB
 * gradient arena).  Every rank maps its peers' buffers through IPC handles and reads them directly over xGMI; sums are
 * formed in rank order, so every rank gets the same bits.  Bootstrap: th_comm_init_p2p, then th_comm_p2p_export (registers
 * the buffer this rank will reduce -- its flat gradient arena -- and fills a blob), ship the TH_P2P_BLOB_BYTES blobs of all
 * ranks to all ranks out of band (rank order), th_comm_p2p_connect.  After that th_allreduce_sum_scale on the registered
 * buffer takes the one-shot path (same signature, same in-place result), and th_allreduce_adam goes one further: the
 * mean gradient is fed straight into Adam (optim.rs:83-113; arguments as th_adam_step) and never written back.
 * Ranks may share a device (the test setup of a 1-GPU box) or own one each.  A peer that never arrives makes the launch give up
 * after the communicator's bound (default 120 s; TAPER_P2P_TIMEOUT_MS at th_comm_init_p2p, th_comm_set_timeout_ms afterwards) and raises
 * the error word.  There is ONE verdict per launch and phase (workgroup 0 watches the flags against the clock and publishes it; every other
 * workgroup follows it), so a launch applies its result in all workgroups or in none:
 *   - peers not READY in time: nothing is read, no reduced gradient is stored, no parameter moves, Adam's counter is not advanced;
 *   - peers not DONE reading in time, in-place form: the same (the reduced gradient is not stored; an Adam step behind it must be
 *     th_adam_step_guarded with th_comm_error_word, which then applies nothing either);
 *   - peers not DONE in time, fused form: the update WAS computed from all W gradients and stays applied, counter included (p / m / v / t
 *     are one consistent step on); the error says a peer may not have finished reading.
 * The error is FINAL: every later launch of the communicator returns at once (all of its workgroups: they consult the word as the previous
 * launch left it), and the host must end the run -- the replicas may be one update apart.  th_comm_error synchronises the stream and reads the
 * word; th_comm_error_peek reads its host-visible copy without touching the stream (call it after any synchronisation the caller does
 * anyway: the end of an epoch's graph replays).  The exported buffer may be fine-grained device memory (th_malloc_finegrained: uncached
 * across agents) where coarse-grained memory cannot be trusted to show a peer's latest writes. */
#define TH_P2P_BLOB_BYTES 192
int th_comm_init_p2p(th_ctx *ctx, int n_ranks, int rank, th_comm **out);
int th_comm_p2p_export(th_comm *comm, float *d_buf, size_t n, uint8_t out_blob[TH_P2P_BLOB_BYTES]);
int th_comm_p2p_connect(th_comm *comm, const uint8_t *blobs /* n_ranks x TH_P2P_BLOB_BYTES, rank order */);
int th_comm_is_p2p(const th_comm *comm);
int th_comm_count(const th_comm *comm, int *out_ranks);   /* ranks the communicator spans (RCCL: ncclCommCount) */
int th_comm_error(th_comm *comm, th_ctx *ctx, int *out_error);
int th_comm_error_peek(const th_comm *comm, int *out_error);
/* where an IN-LAUNCH exchange (csrc/dp_dev.h) gave up, for the error message: out4 = {slot of the first workgroup whose wait ran out (-1: none
 * did -- the time-out was a copy-engine round's), that launch's exchange step, bit s set = rank s's words had not arrived, this rank's
 * exchange step now}.  Synchronises the stream. */
int th_comm_timeout_detail(th_comm *comm, th_ctx *ctx, int out4[4]);
int th_comm_set_timeout_ms(th_comm *comm, int64_t ms);
/* the communicator's sticky error word in device memory (NULL for an RCCL communicator): th_adam_step_guarded's guard */
int th_comm_error_word(const th_comm *comm, const uint32_t **d_out);
/* test hook: {in-place, fused-with-Adam} one-shot launches this communicator has enqueued or captured so far */
int th_comm_stats(const th_comm *comm, int64_t out2[2]);
int th_allreduce_adam(th_comm *comm, th_ctx *ctx, const float *d_grads, size_t n, float scale, float *d_params, float *d_m, float *d_v,
                      const int64_t *d_offsets, const int32_t *d_has_grad, int n_tensors, int32_t *d_t, const float *d_lr,
                      float beta1, float beta2, float eps, float weight_decay, int pre_ticked);

/* ---- the exchange INSIDE the launch that produces the gradients (csrc/dp_dev.h) -------------------------------------------------
 * The same peer-to-peer communicator also carries a receive region per rank (mapped by every peer with its flag block).  A workgroup
 * that has finished a slice of a gradient pushes it into every peer's region as 8-byte words and polls its own region for the same
 * words of every peer -- an empty word is all ones (a NaN no arithmetic produces), so the value is its own flag: one dependent trip through
 * memory, no extra bytes -- adds the W values in rank order (the same bits on every rank), scales by 1/W, puts the empty mark back and goes
 * on to Adam from registers: no extra launch, no arena-wide hand-shake, nothing to do with one rank.  th_mlp_tail_dp is th_mlp_tail with that
 * exchange in its epilogues: dW1 / db1 are reduced and applied in place (w1_fuse / b1_fuse as before; their counter is the one
 * th_linear_fwd_ex ticked), the reduced dW2 / db2 are written for the deferred update exactly as the single-GPU step writes its own.
 * A waiting workgroup holds its place on the device, so ranks that SHARE a device may only take this form while the workgroups of all
 * of them but one leave a place free (th_mlp_tail_dp_supported says so; the three-launch form above is the fallback).
 * A peer whose slice does not arrive within the communicator's bound: the workgroup applies nothing, raises the error word (FINAL, as
 * above) and the lead workgroup takes the counter's tick back; a peer that is absent altogether therefore leaves p / m / v / t untouched.
 * (A peer that dies in the middle of its launch can leave the slices it did push applied: the error is raised all the same, the state is
 * then one PARTIAL step on and only a checkpoint restores it -- the three-launch form's all-or-nothing verdict costs 2.5 us per step.)
 * th_ctx_set_update_guard: while set, the launches that apply deferred updates or tick a counter (th_linear_fwd_ex's spare workgroup,
 * th_adam_slices, th_adam_tick) look at the word first and do nothing once it is non-zero -- the steps behind a failed exchange in a captured
 * graph; and a tick that does happen also advances *d_step_word (nullable; th_comm_step_word): the exchange's step number, which every
 * workgroup of the gradient launch behind it then reads at no cost (it picks the half of the double-buffered receive region). */
int th_comm_init_loopback(th_ctx *ctx, th_comm **out);       /* W = 2 with this rank as its own peer: the whole protocol through local memory; results are the single-GPU step's, bit for bit */
int th_comm_is_loopback(const th_comm *comm);
/* 0: no in-launch exchange (RCCL / not connected); 1: ONE-shot -- every rank reduces every slice, one hop (2 or 3 ranks, loopback);
 * 2: TWO-shot -- slice s is reduced by rank s % W and the mean pushed back, two hops and 2 / W of the bytes per link (4 ranks and more;
 * TAPER_DP_TWO_SHOT = 0 | 1 at th_comm_p2p_connect forces the form on every rank).  Both forms give the same bits. */
int th_comm_exchange_form(const th_comm *comm, int *out_form);
int th_comm_sharing(const th_comm *comm, int *out_ranks_on_this_device);
int th_comm_stats_inkernel(const th_comm *comm, int64_t *out_launches);
/* collective check of the exchange alone: `rounds` launches of `slots` workgroups, every value compared on the device; *out_bad = mismatches + time-outs */
int th_comm_exchange_selftest(th_comm *comm, th_ctx *ctx, int slots, int rounds, int *out_bad);
int th_ctx_set_update_guard(th_ctx *ctx, const uint32_t *d_skip_if_nonzero /* nullable: clears it */, uint32_t *d_step_word /* nullable */);
int th_comm_step_word(const th_comm *comm, uint32_t **d_out);   /* NULL for an RCCL communicator */
/* ... and the simple CNN's step (th_conv_chain_head_fwd, th_wide_head_grads): th_wide_head_grads_dp is th_wide_head_grads with every finished
 * sum -- a block of dW, db, a block of the last conv's bias gradient -- reduced over the ranks before its store and its Adam epilogue; loss
 * and hit count stay this rank's.  d_tick: the counter th_conv_chain_head_fwd ticked (with the guard set it also advances the exchange's
 * step number), taken back by the launch when the exchange fails. */
int th_wide_head_grads_dp_supported(const th_comm *comm, th_ctx *ctx, int batch, int in_features, int classes, int conv_c);
int th_wide_head_grads_dp(th_comm *comm, th_ctx *ctx, const float *d_x, const float *d_dl, const float *d_rowstat, const float *d_cbpart, int batch,
                          int in_features, int classes, int conv_c, float *d_dw, float *d_db, float *d_conv_gb, float *d_loss, float *d_ncorrect,
                          float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance, const th_adam_fuse *w_fuse,
                          const th_adam_fuse *b_fuse, const th_adam_fuse *cb_fuse, int32_t *d_tick);
int th_mlp_tail_dp_supported(const th_comm *comm, th_ctx *ctx, int batch, int in_features, int hidden, int classes);
int th_mlp_tail_dp(th_comm *comm, th_ctx *ctx, const float *d_x, const float *d_h, const float *d_w2, const float *d_b2,
                   const float *d_targets, int batch, int in_features, int hidden, int classes,
                   float *d_loss, float *d_ncorrect, float *d_dw1, float *d_db1, float *d_dw2, float *d_db2,
                   float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance,
                   const th_adam_fuse *w1_fuse, const th_adam_fuse *b1_fuse, int32_t *d_tick);

#ifdef __cplusplus
}
#endif
#endif /* TAPER_HIP_H */

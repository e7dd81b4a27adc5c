/*
 * taper_oracle.h -- CPU restatement of vaibhawvipul/taper's training hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under taper_amd/ may include, link or
 * call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / the timed CPU baseline.
 *
 * Plain C11 restatement (the reference is Rust, which this image cannot
 * build: no cargo/rustc).  Every function cites the reference file:line it
 * follows (paths relative to the reference checkout).
 *
 * Pinning status: pinned against every known-answer test the reference
 * holds for this path (tests/smoke.rs, src/loss.rs:292-374,
 * src/optim.rs:354-423, src/train.rs:387-417) -- see tests/test_kats.py (both backends) and tests/test_golden.py.
 * The reference holds NO vectors for conv2d / pools / transpose values /
 * Adam's exact numbers; those parts are "parity unpinned by reference
 * tests" and are cross-checked against torch-CPU fixtures generated in the
 * authoring container (tests/golden/, script committed) and finite
 * differences.
 */
#ifndef TAPER_ORACLE_H
#define TAPER_ORACLE_H

#include <stddef.h>
#include <stdint.h>

/* The reference runs these loops through rayon (`par_chunks_mut`: tensor.rs:1420,1491,1552,1614,1745) on all cores.
 * The parity oracle is built without OpenMP (the pragma vanishes: one thread, fixed order); the cpu_baseline build
 * (`make fast`, -fopenmp) runs them thread-parallel like the reference.  Every iteration writes its own plane / row. */
#ifdef _OPENMP
#define OT_PAR_FOR _Pragma("omp parallel for schedule(static)")
#else
#define OT_PAR_FOR
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define OT_MAX_DIMS 4

/* Tensor handle.  Mirrors src/tensor.rs:236-244: data / grad / tape_node are
 * shared between clones (Arc), shape and requires_grad are per handle. */
typedef struct ot_core ot_core;
typedef struct ot_tensor {
    ot_core *core;
    size_t shape[OT_MAX_DIMS];
    int ndim;
    int requires_grad;
} ot_tensor;

/* ---- tape (src/tape.rs) ------------------------------------------------ */
void ot_tape_reset(void);                 /* tape.rs:43-49 */
size_t ot_tape_len(void);
/* 1 = literal reference behaviour (tensor.rs:524-528: node id 0 means "no
 * node", so backward() from the first recorded op is a no-op, quirk Q1);
 * 0 = ids still 0-based but backward always runs.  Default 1. */
void ot_tape_set_zero_sentinel(int on);

/* ---- tensor lifetime / access (tensor.rs:470-541) ---------------------- */
ot_tensor *ot_new(const float *data, const size_t *shape, int ndim);
ot_tensor *ot_scalar(float v);
ot_tensor *ot_clone(const ot_tensor *t);
void ot_free(ot_tensor *t);
void ot_set_requires_grad(ot_tensor *t, int on);
size_t ot_len(const ot_tensor *t);
const float *ot_data(const ot_tensor *t);
float *ot_data_mut(ot_tensor *t);
const float *ot_grad(const ot_tensor *t); /* NULL when grad is None */
void ot_set_grad(ot_tensor *t, const float *g); /* NULL -> None */
size_t ot_tape_node(const ot_tensor *t);
void ot_backward(ot_tensor *t);           /* tensor.rs:520-529 */
void ot_zero_grad(ot_tensor *t);          /* tensor.rs:531-533 */

/* ---- gemm (src/gemm.rs:72-119 semantics) ------------------------------- */
/* 0 = plain-loop parity oracle; bit 0 = packed sgemm (cpu_packed_sgemm.c), bit 1 = OpenMP plane loops: the cpu_baseline build */
int ot_baseline_flavour(void);
void ot_sgemm_rowmajor(int trans_a, int trans_b, int m, int n, int k,
                       float alpha, const float *a, const float *b,
                       float beta, float *c);

/* ---- ops (src/ops.rs, src/tensor.rs) ----------------------------------- */
ot_tensor *ot_add(const ot_tensor *a, const ot_tensor *b);     /* ops.rs:8-51 */
ot_tensor *ot_mul(const ot_tensor *a, const ot_tensor *b);     /* ops.rs:53-120 */
ot_tensor *ot_sub(const ot_tensor *a, const ot_tensor *b);     /* ops.rs:377-416 */
ot_tensor *ot_div(const ot_tensor *a, const ot_tensor *b);     /* ops.rs:440-496 */
ot_tensor *ot_matmul(const ot_tensor *a, const ot_tensor *b);  /* ops.rs:200-298 */
ot_tensor *ot_relu(const ot_tensor *x);                        /* ops.rs:312-374 */
ot_tensor *ot_transpose(const ot_tensor *x);                   /* tensor.rs:544-591 */
ot_tensor *ot_sigmoid(const ot_tensor *x);                     /* tensor.rs:594-634 */
ot_tensor *ot_add_broadcast(const ot_tensor *a, const ot_tensor *b); /* tensor.rs:636-704 */
ot_tensor *ot_sub_broadcast_rows(const ot_tensor *a, const ot_tensor *r); /* tensor.rs:707-770 */
ot_tensor *ot_mean(const ot_tensor *x);                        /* tensor.rs:772-800 */
ot_tensor *ot_reshape(const ot_tensor *x, const size_t *shape, int ndim); /* tensor.rs:803-840 */
ot_tensor *ot_flatten(const ot_tensor *x, int start_dim);      /* tensor.rs:843-858 */
ot_tensor *ot_squeeze(const ot_tensor *x, int dim /* -1 = all */); /* tensor.rs:861-877 */
ot_tensor *ot_unsqueeze(const ot_tensor *x, int dim);          /* tensor.rs:880-887 */
ot_tensor *ot_sum(const ot_tensor *x, int dim /* -1 = all */, int keepdim); /* tensor.rs:890-1018 */
/* returns values; *indices_out receives the f32-encoded index tensor */
ot_tensor *ot_max(const ot_tensor *x, int dim /* -1 = all */, ot_tensor **indices_out); /* tensor.rs:1021-1083 */
ot_tensor *ot_argmax(const ot_tensor *x, int dim);             /* tensor.rs:1086-1088 */
ot_tensor *ot_exp(const ot_tensor *x);                         /* tensor.rs:1091-1133 */
ot_tensor *ot_log(const ot_tensor *x);                         /* tensor.rs:1136-1169 */
ot_tensor *ot_pow(const ot_tensor *x, float e);                /* tensor.rs:1172-1206 */
ot_tensor *ot_sqrt(const ot_tensor *x);                        /* tensor.rs:1209-1211 */

/* conv / pool (tensor.rs:1221-2081).  mode 0 = faithful (the reference's
 * chain: im2col -> reinterpreted weight -> matmul -> reshape -> transpose_4d
 * (cuts the tape, Q2) -> add_bias_4d); mode 1 = full_backward extension
 * (same forward numbers, but im2col and transpose_4d propagate gradients). */
ot_tensor *ot_conv2d(const ot_tensor *x, const ot_tensor *w, const ot_tensor *bias /* nullable */,
                     int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int mode);
ot_tensor *ot_conv2d_relu(const ot_tensor *x, const ot_tensor *w, const ot_tensor *bias,
                          int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int mode);
ot_tensor *ot_conv2d_direct_3x3(const ot_tensor *x, const ot_tensor *w, const ot_tensor *bias,
                                int stride_h, int stride_w, int pad_h, int pad_w); /* tensor.rs:1287-1376 */
/* stride_h == 0 means "stride = kernel" (tensor.rs:1403).  zero_first: 1 =
 * reference behaviour Q5 (backward zeroes each plane before scatter). */
ot_tensor *ot_max_pool2d(const ot_tensor *x, int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w,
                         int zero_first, int64_t *argmax_out /* nullable, n*c*ho*wo */);
ot_tensor *ot_avg_pool2d(const ot_tensor *x, int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w);
ot_tensor *ot_adaptive_avg_pool2d(const ot_tensor *x, int h_out, int w_out); /* nn.rs:670-686 */

/* ---- loss (src/loss.rs) ------------------------------------------------ */
ot_tensor *ot_log_softmax(const ot_tensor *x);                 /* loss.rs:101-126 */
ot_tensor *ot_softmax(const ot_tensor *x);                     /* exp(log_softmax): Q12 */
ot_tensor *ot_cross_entropy_loss(const ot_tensor *logits, const ot_tensor *targets); /* loss.rs:136-195 */
float ot_accuracy(const ot_tensor *pred, const ot_tensor *targets); /* loss.rs:271-290 */
ot_tensor *ot_one_hot(const ot_tensor *idx, int num_classes);  /* loss.rs:248-268 */
ot_tensor *ot_bce_loss(const ot_tensor *pred, const ot_tensor *targets); /* loss.rs:6-73 */
ot_tensor *ot_mse_loss(const ot_tensor *pred, const ot_tensor *targets); /* loss.rs:76-80 */

/* ---- linear layer helper (nn.rs:54-60) --------------------------------- */
ot_tensor *ot_linear_forward(const ot_tensor *x, const ot_tensor *w, const ot_tensor *b /* nullable */);

/* ---- optimizers (src/optim.rs) ----------------------------------------- */
typedef struct ot_adam ot_adam;
ot_adam *ot_adam_new(ot_tensor **params, int n, float lr, float beta1, float beta2,
                     float eps, float weight_decay);           /* optim.rs:54-81 */
void ot_adam_step(ot_adam *o);                                 /* optim.rs:83-113 */
void ot_adam_zero_grad(ot_adam *o);                            /* optim.rs:115-119 */
void ot_adam_set_lr(ot_adam *o, float lr);
float ot_adam_get_lr(const ot_adam *o);
int ot_adam_t(const ot_adam *o);
const float *ot_adam_m(const ot_adam *o, int i);
const float *ot_adam_v(const ot_adam *o, int i);
void ot_adam_free(ot_adam *o);
void ot_sgd_step(ot_tensor **params, int n, float lr);         /* optim.rs:21-33 */
/* llvm.powi.f32 as lowered by compiler-rt __powisf2 */
float ot_powi(float a, int b);

/* ---- data (src/data/mnist.rs:277-310) ---------------------------------- */
/* cpu_baseline leg only (see `make fast`) */
int ot_baseline_max_threads(void);
void ot_baseline_set_threads(int n);
void ot_get_batch(const float *images, const float *labels, const size_t *indices, size_t batch,
                  float *out_images /* batch*784 */, float *out_labels);

/* ---- sequential model + step driver (examples/train_mnist*.rs) --------- */
enum {
    OT_L_LINEAR = 0, OT_L_RELU = 1, OT_L_SIGMOID = 2, OT_L_CONV2D_RELU = 3, OT_L_CONV2D = 4,
    OT_L_MAXPOOL = 5, OT_L_AVGPOOL = 6, OT_L_ADAPTIVE_AVGPOOL = 7, OT_L_FLATTEN = 8
};
typedef struct ot_layer {
    int kind;
    ot_tensor *w, *b;          /* Linear [out,in] / Conv [co,ci,kh,kw]; bias nullable */
    int k_h, k_w, s_h, s_w, p_h, p_w; /* conv / pool geometry (pool: s_h==0 -> stride=kernel) */
    int out_h, out_w;          /* adaptive avg pool */
    int start_dim;             /* flatten */
} ot_layer;

typedef struct ot_model {
    ot_layer *layers;
    int n_layers;
    int conv_mode;             /* 0 faithful, 1 full_backward */
} ot_model;

ot_tensor *ot_model_forward(const ot_model *m, const ot_tensor *x); /* nn.rs:149-151 */
/* parameters in Sequential::parameters() order (nn.rs:159-161); returns count */
int ot_model_parameters(const ot_model *m, ot_tensor **out, int cap);

/* One training step exactly as examples/train_mnist.rs:89-121:
 * Tape::reset -> forward -> cross_entropy_loss -> accuracy -> backward ->
 * Adam::step -> zero_grad.  If grads_out != NULL the flat parameter-gradient
 * vector (zeros where grad is None) is copied out BEFORE the optimizer step,
 * and has_grad_out[i] tells which params had Some(grad).
 * x_shape: the shape handed to model.forward ([B,784] or [B,1,28,28]). */
void ot_train_step(const ot_model *m, ot_adam *opt, const float *images, const float *labels,
                   const size_t *x_shape, int x_ndim, float *loss_out, float *acc_out,
                   float *logits_out /* nullable */, float *grads_out /* nullable */,
                   int *has_grad_out /* nullable */);

#ifdef __cplusplus
}
#endif
#endif

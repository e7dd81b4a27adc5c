/*
 * taper_host.h -- C ABI over the C++ host mirror (taper_amd/csrc/host): the
 * Tensor / Tape / nn::Module / loss / optim / data / train surface of the
 * reference (src/lib.rs:1-17 re-exports) as opaque handles, so that tests and
 * bench.py (Python, ctypes) -- or any other language -- can drive exactly the
 * code path a Rust host would.  Device work happens only through
 * include/taper_hip.h; this layer adds no arithmetic.
 *
 * Every function returns 0 on success; on failure the message is in
 * tp_last_error() (the reference panics instead: src/ops.rs:201-208 etc.).
 * Handles are owned by the caller and released with the matching *_free.
 */
#ifndef TAPER_HOST_H
#define TAPER_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tp_tensor tp_tensor;
typedef struct tp_module tp_module;
typedef struct tp_optim tp_optim;
typedef struct tp_dataset tp_dataset;
typedef struct tp_loader tp_loader;
typedef struct tp_trainer tp_trainer;
typedef struct tp_sched tp_sched;
typedef struct tp_comm tp_comm;

const char *tp_last_error(void);

/* ---- device / tape (src/tape.rs) ---- */
int tp_device_set(int device_id);
int tp_device_sync(void);
int tp_device_shutdown(void);
void *tp_device_ctx(void);                 /* the th_ctx* of this thread (for mixing with taper_hip.h calls) */
int tp_tape_reset(void);                   /* tape.rs:43-49 */
int tp_tape_len(size_t *out);
int tp_tape_set_compat_zero_sentinel(int on);   /* quirk Q1 */
int tp_set_full_backward(int on);               /* quirk Q2: 0 = faithful (default) */
int tp_set_conv_chain(int on);                  /* Trainer steps: 1 (default) = the conv front of a Sequential as one launch where compiled (th_conv_chain_fwd), 0 = layer by layer */
int tp_set_conv_chain_head(int on);             /* ... and the classifier behind such a front row by row in the same launch where compiled (th_conv_chain_head_fwd + th_wide_head_grads: the simple CNN's step is two launches); default 1 */

/* ---- Tensor (src/tensor.rs:470-541) ---- */
int tp_tensor_new(const float *h_data, const size_t *shape, int ndim, tp_tensor **out);
int tp_tensor_randn(const size_t *shape, int ndim, uint64_t seed, tp_tensor **out);
int tp_tensor_clone(const tp_tensor *t, tp_tensor **out);     /* Arc clone */
int tp_tensor_free(tp_tensor *t);
int tp_tensor_set_requires_grad(tp_tensor *t, int on);
int tp_tensor_requires_grad(const tp_tensor *t, int *out);
int tp_tensor_ndim(const tp_tensor *t, int *out);
int tp_tensor_shape(const tp_tensor *t, size_t *out4);
int tp_tensor_len(const tp_tensor *t, size_t *out);
int tp_tensor_data(const tp_tensor *t, float *h_out);          /* D2H copy of len floats */
int tp_tensor_set_data(tp_tensor *t, const float *h_in);
int tp_tensor_has_grad(const tp_tensor *t, int *out);
int tp_tensor_grad(const tp_tensor *t, float *h_out);          /* error if grad is None */
int tp_tensor_set_grad(tp_tensor *t, const float *h_in /* NULL -> None */);
int tp_tensor_tape_node(const tp_tensor *t, size_t *out);
int tp_tensor_dptr(const tp_tensor *t, void **d_out);          /* device pointer of the storage */
int tp_tensor_backward(tp_tensor *t);                          /* tensor.rs:520-529 */
int tp_tensor_zero_grad(tp_tensor *t);                         /* tensor.rs:531-533 */

/* ---- ops (src/ops.rs, src/tensor.rs); each returns a new handle ---- */
int tp_add(const tp_tensor *a, const tp_tensor *b, tp_tensor **out);
int tp_sub(const tp_tensor *a, const tp_tensor *b, tp_tensor **out);
int tp_mul(const tp_tensor *a, const tp_tensor *b, tp_tensor **out);
int tp_div(const tp_tensor *a, const tp_tensor *b, tp_tensor **out);
int tp_matmul(const tp_tensor *a, const tp_tensor *b, tp_tensor **out);
int tp_add_broadcast(const tp_tensor *a, const tp_tensor *b, tp_tensor **out);
int tp_sub_broadcast_rows(const tp_tensor *a, const tp_tensor *b, tp_tensor **out);
int tp_relu(const tp_tensor *x, tp_tensor **out);
int tp_sigmoid(const tp_tensor *x, tp_tensor **out);
int tp_transpose(const tp_tensor *x, tp_tensor **out);
int tp_exp(const tp_tensor *x, tp_tensor **out);
int tp_log(const tp_tensor *x, tp_tensor **out);
int tp_pow(const tp_tensor *x, float e, tp_tensor **out);
int tp_mean(const tp_tensor *x, tp_tensor **out);
int tp_sum(const tp_tensor *x, int dim /* -1 = all */, int keepdim, tp_tensor **out);
int tp_max(const tp_tensor *x, int dim /* -1 = all */, tp_tensor **values, tp_tensor **indices);
int tp_reshape(const tp_tensor *x, const size_t *shape, int ndim, tp_tensor **out);
int tp_flatten(const tp_tensor *x, int start_dim, tp_tensor **out);
int tp_squeeze(const tp_tensor *x, int dim /* -1 = all */, tp_tensor **out);
int tp_unsqueeze(const tp_tensor *x, int dim, tp_tensor **out);
int tp_linear(const tp_tensor *x, const tp_tensor *w, const tp_tensor *b /* nullable */, int relu, tp_tensor **out);
int tp_conv2d(const tp_tensor *x, const tp_tensor *w, const tp_tensor *b /* nullable */, int stride_h, int stride_w,
              int pad_h, int pad_w, int dil_h, int dil_w, int relu, tp_tensor **out);
int tp_max_pool2d(const tp_tensor *x, int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w, tp_tensor **out);
int tp_avg_pool2d(const tp_tensor *x, int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w, tp_tensor **out);

/* ---- loss (src/loss.rs) ---- */
int tp_log_softmax(const tp_tensor *x, tp_tensor **out);
int tp_softmax(const tp_tensor *x, tp_tensor **out);
int tp_cross_entropy_loss(const tp_tensor *logits, const tp_tensor *targets, tp_tensor **out);
int tp_accuracy(const tp_tensor *pred, const tp_tensor *targets, float *out);
int tp_one_hot(const tp_tensor *idx, int num_classes, tp_tensor **out);
int tp_mse_loss(const tp_tensor *pred, const tp_tensor *targets, tp_tensor **out);
int tp_bce_loss(const tp_tensor *pred, const tp_tensor *targets, tp_tensor **out);                    /* loss.rs:6-73 */
int tp_cross_entropy_loss_onehot(const tp_tensor *logits, const tp_tensor *targets, tp_tensor **out); /* loss.rs:201-245 */

/* ---- nn (src/nn.rs, src/activation.rs) ---- */
int tp_linear_new(int in_features, int out_features, int with_bias, uint64_t seed, tp_module **out);
int tp_relu_new(tp_module **out);
int tp_sigmoid_new(tp_module **out);
int tp_conv2d_new(int in_ch, int out_ch, int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w, int with_bias,
                  int fuse_relu, uint64_t seed, tp_module **out);
/* nn.rs:180-354 with groups > 1: weight [out, in/groups, k, k]; forward = slice_channels / slice_output_channels /
 * slice_1d, one conv2d per group, cat(dim 1) (nn.rs:289-332, 859-1014).  Like the reference's, the slices carry no tape
 * nodes: a grouped convolution is forward-only (nothing behind it, nor its own parameters, receives a gradient). */
int tp_conv2d_grouped_new(int in_ch, int out_ch, int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w, int groups,
                          int with_bias, int fuse_relu, uint64_t seed, tp_module **out);
int tp_slice_channels(const tp_tensor *x, size_t start, size_t end, tp_tensor **out);   /* nn.rs:862-886 */
int tp_cat(const tp_tensor *const *tensors, int n, size_t dim, tp_tensor **out);        /* nn.rs:928-1014: 2-D dim 0/1, 4-D dim 1 */
int tp_maxpool2d_new(int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w, tp_module **out);
int tp_avgpool2d_new(int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w, tp_module **out);
int tp_adaptive_avgpool2d_new(int out_h, int out_w, tp_module **out);
int tp_flatten_new(int start_dim, tp_module **out);
int tp_dropout_new(float p, uint64_t seed, tp_module **out);                 /* nn.rs:773-827 */
int tp_dropout_set_training(tp_module *m, int training);                     /* train() / eval() */
int tp_dropout_last_mask(const tp_module *m, tp_tensor **out);               /* mask of the latest training forward */
int tp_sequential_new(tp_module *const *layers, int n, int fuse_linear_relu, tp_module **out);
int tp_module_free(tp_module *m);
int tp_module_forward(const tp_module *m, const tp_tensor *x, tp_tensor **out);
int tp_module_num_parameters(const tp_module *m, int *out);
int tp_module_parameter(const tp_module *m, int i, tp_tensor **out);   /* shares storage with the model */

/* ---- optim (src/optim.rs) ---- */
int tp_adam_new(tp_tensor *const *params, int n, float lr, float beta1, float beta2, float eps, float weight_decay,
                tp_optim **out);
int tp_sgd_new(tp_tensor *const *params, int n, float lr, tp_optim **out);
int tp_adamw_new(tp_tensor *const *params, int n, float lr, float beta1, float beta2, float eps, float weight_decay,
                 tp_optim **out);                                            /* optim.rs:130-180 */
int tp_optim_free(tp_optim *o);
int tp_optim_step(tp_optim *o);
int tp_optim_zero_grad(tp_optim *o);
int tp_adam_set_lr(tp_optim *o, float lr);
int tp_adam_get_lr(const tp_optim *o, float *out);
int tp_adam_t(const tp_optim *o, int *out);
int tp_adam_moments(const tp_optim *o, float *h_m, float *h_v);  /* concatenated in parameter order */
int tp_optim_total(const tp_optim *o, int64_t *out);             /* padded arena length (all-reduce size) */
int tp_adam_load_state(tp_optim *o, int t, const float *h_m, const float *h_v);  /* inverse of tp_adam_t / tp_adam_moments */

/* ---- LR schedulers (src/optim.rs:183-352) ---- */
int tp_sched_step_lr(float base_lr, size_t step_size, float gamma, tp_sched **out);
int tp_sched_exponential(float base_lr, float gamma, tp_sched **out);
int tp_sched_cosine(float base_lr, size_t t_max, float min_lr, tp_sched **out);
int tp_sched_plateau(float initial_lr, float factor, size_t patience, float min_lr, int mode_max, tp_sched **out);
int tp_sched_step(tp_sched *s, const float *metric /* NULL = None */);
int tp_sched_get_lr(const tp_sched *s, float *out);
int tp_sched_free(tp_sched *s);

/* ---- data (src/data/mnist.rs) ---- */
int tp_dataset_from_host(const float *h_images, const float *h_labels, size_t n, int train, tp_dataset **out);
int tp_dataset_from_idx(const char *images_path, const char *labels_path, int train, tp_dataset **out);
int tp_dataset_synthetic(size_t n, uint64_t seed, int train, tp_dataset **out);
int tp_dataset_len(const tp_dataset *d, size_t *out);
int tp_dataset_tensors(const tp_dataset *d, tp_tensor **images, tp_tensor **labels);
int tp_dataset_free(tp_dataset *d);
int tp_loader_new(const tp_dataset *d, size_t batch_size, int shuffle, uint64_t seed, tp_loader **out);
int tp_loader_reset(tp_loader *l);
int tp_loader_num_batches(const tp_loader *l, size_t *out);
int tp_loader_next(tp_loader *l, tp_tensor **images, tp_tensor **labels, int *has_batch);
int tp_loader_free(tp_loader *l);

/* ---- data parallel (new) ---- */
int tp_comm_unique_id(uint8_t out_id[128]);
int tp_comm_new(int n_ranks, int rank, const uint8_t id[128], tp_comm **out);
/* peer-to-peer communicator (th_comm_init_p2p): new -> export_arena(optimizer) -> ship all 192-byte blobs to all ranks ->
 * connect(blobs in rank order); the Trainer then runs all-reduce + Adam as one launch */
int tp_comm_new_p2p(int n_ranks, int rank, tp_comm **out);
/* The exchange inside the gradient launch (th_mlp_tail_dp; include/taper_hip.h): a Trainer over a peer-to-peer communicator takes it for
 * the Linear + ReLU + Linear step whenever tp_comm_tail_exchange_ok says the shapes and the placement of the ranks allow it
 * (tp_comm_set_inkernel(c, 0): never -- the three-launch form, for A/B runs).  tp_comm_new_loopback: W = 2 with this process as its own
 * peer -- the whole protocol through local memory, results bit-identical to the single-GPU step; what a 1-GPU box can time of it. */
int tp_comm_new_loopback(tp_comm **out);
int tp_comm_set_inkernel(tp_comm *c, int on);
int tp_comm_inkernel_launches(tp_comm *c, int64_t *out);
int tp_comm_exchange_selftest(tp_comm *c, int slots, int rounds, int *out_bad);   /* collective */
int tp_comm_ranks_on_this_device(tp_comm *c, int *out);
int tp_comm_exchange_form(tp_comm *c, int *out);   /* th_comm_exchange_form: 0 none, 1 one-shot, 2 two-shot */
int tp_comm_tail_exchange_ok(tp_comm *c, int batch, int in_features, int hidden, int classes, int *out);
int tp_comm_export_arena(tp_comm *c, tp_optim *optimizer, uint8_t out_blob[192]);
int tp_comm_connect(tp_comm *c, const uint8_t *blobs, size_t n_bytes);
/* fine_grained != 0: the optimizer's gradient arena is first moved into fine-grained device memory (coherent across agents, never cached
 * in a peer's L2): the fallback when the self-check fails on the pooled, coarse-grained arena.  The communicator keeps the registered
 * arenas alive; free it on every rank before the optimizer. */
int tp_comm_export_arena_ex(tp_comm *c, tp_optim *optimizer, int fine_grained, uint8_t out_blob[192]);
/* collective self-test (every rank calls it, before training): 3 (or `rounds`) all-reduces of patterns that differ per rank, round and
 * element through the SAME addresses of the optimizer's gradient arena, through the in-place kernel and then through the fused
 * all-reduce + Adam kernel (p / m / v against the closed form of optim.rs:83-113; state restored) */
int tp_comm_self_check(tp_comm *c, tp_optim *optimizer, int *ok);
int tp_comm_self_check_rounds(tp_comm *c, tp_optim *optimizer, int rounds, int *ok);
int tp_comm_timed_out(tp_comm *c, int *out);   /* synchronises the stream */
int tp_comm_failed(tp_comm *c, int *out);      /* the same verdict from the host-visible error word, no synchronisation; the Trainer throws on it after every epoch / eager step */
int tp_comm_set_timeout_ms(tp_comm *c, int64_t ms);   /* bound of the in-kernel waits for a peer (default 120 s; TAPER_P2P_TIMEOUT_MS) */
int tp_comm_set_fuse_adam(tp_comm *c, int on);        /* p2p: 1 (default) all-reduce + Adam in one launch, 0 all-reduce in place then Adam::step */
int tp_comm_stats(tp_comm *c, int64_t out2[2]);   /* {in-place, fused} one-shot launches enqueued or captured */
int tp_comm_free(tp_comm *c);
int tp_comm_allreduce_mean(tp_comm *c, void *d_buf, size_t n);
int tp_comm_count(tp_comm *c, int *out_ranks);                     /* th_comm_count: for RCCL, what ncclCommCount reports */
/* `reps` gradient exchanges + Adam as a Trainer step issues them, timed with events on the stream; collective (every rank calls it);
 * the optimizer state moves: for benchmarks, after the timed run */
int tp_comm_time_exchange(tp_comm *c, tp_optim *adam, int reps, float *us_per_exchange);

/* ---- train (src/train.rs, examples/train_mnist*.rs) ---- */
int tp_trainer_new(tp_module *model, tp_optim *adam, tp_trainer **out);
int tp_trainer_set_sample_shape(tp_trainer *t, const size_t *shape, int ndim);  /* e.g. {1,28,28} for the CNN */
int tp_trainer_set_comm(tp_trainer *t, tp_comm *c /* nullable */);
/* graph-path knobs: steps per hipGraph replay; fuse_head 1 = classifier head (last Linear + cross-entropy)
 * as one launch, 2 = additionally the backward of a Linear+ReLU layer in front of it in the same launch
 * (th_mlp_tail; falls back to 1 where unsupported), 0 = off; Adam updates in the epilogues of the
 * gradient kernels (ignored with a communicator) */
int tp_trainer_set_options(tp_trainer *t, int graph_chunk, int fuse_head, int fuse_adam);
int tp_trainer_free(tp_trainer *t);
/* one reference-literal step (examples/train_mnist.rs:89-121); reads loss / accuracy back */
int tp_trainer_train_step(tp_trainer *t, const tp_tensor *images, const tp_tensor *labels, float *loss, float *acc);
/* mode 0 = eager train_epoch (train.rs:98-144), 1 = hipGraph replay, 2 = evaluate (train.rs:147-172).
 * per_step (nullable) receives 2*num_batches floats {loss, n_correct}; max_steps 0 = whole epoch. */
int tp_trainer_run_epoch(tp_trainer *t, tp_loader *l, int mode, size_t max_steps, float *avg_loss, float *accuracy,
                         size_t *total_correct, size_t *total_samples, size_t *num_batches, float *per_step,
                         size_t per_step_cap);

/* train.rs:175-261: fit = epochs of train + evaluate + scheduler.step(Some(val_loss)) + metrics + early
 * stop at val_acc > 0.99; graph != 0 trains through the captured step */
int tp_trainer_set_scheduler(tp_trainer *t, tp_sched *s /* nullable; shared with the caller's handle */);
int tp_trainer_fit(tp_trainer *t, tp_loader *train, tp_loader *val, size_t epochs, int verbose, int graph);
/* Metrics (train.rs:9-71): which = 0 train_loss, 1 train_acc, 2 val_loss, 3 val_acc, 4 epoch_times */
int tp_trainer_metrics(const tp_trainer *t, int which, float *h_out, size_t cap, size_t *n_out);
/* which = 0: the print_last line, 1: the plot_summary text (NUL-terminated, truncated to cap) */
int tp_trainer_metrics_text(const tp_trainer *t, int which, char *buf, size_t cap);
/* train.rs:264-292 text checkpoint and its inverse; the optimizer-state pair is an extension */
int tp_trainer_save_checkpoint(const tp_trainer *t, const char *path);
int tp_trainer_load_checkpoint(tp_trainer *t, const char *path);
int tp_trainer_save_optimizer_state(const tp_trainer *t, const char *path);
int tp_trainer_load_optimizer_state(tp_trainer *t, const char *path);
/* an f32 as Rust's `{}` prints it (the checkpoint's number format) */
int tp_format_f32(float v, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* TAPER_HOST_H */

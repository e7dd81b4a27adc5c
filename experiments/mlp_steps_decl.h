/* Declarations th_mlp2_steps had in include/taper_hip.h while it was compiled into libtaper_hip.so (r02). */
/* EXPERIMENTAL -- measured, NOT on any Trainer path (14.2-15.8 us per step against 11.9 us for the two launches per step).
 * MANY training steps of the two-layer MLP (Linear + ReLU, Linear, softmax cross-entropy, Adam: train.rs:98-144 over the model of
 * examples/train_mnist.rs with one hidden layer -- BASELINE configs[1]) in ONE persistent launch: (batch / 16) x (hidden / 16) <= 32
 * workgroups, all on one XCD (they share its L2: no fences), walk every step as forward tiles -> barrier -> head + dW1 tiles with Adam in
 * the epilogues -> barrier.  d_x [steps][batch][in], d_targets [steps][batch]: the gathered batches of the chunk.  fuse4: W1, b1, W2, b2
 * (p / m / v, ONE shared step counter d_t holding the count BEFORE the launch -- the launch adds `steps` --, d_lr, hyperparameters).
 * d_loss [1]: the last step's loss; the step log as in th_log_step (one slot per step).  h_err: host-visible (th_host_malloc) error word,
 * 0 on entry; 1 = a barrier timed out (bounded spins: the launch ends, the parameters are then undefined), 2 = the workgroups were not
 * placed on one XCD (nothing was updated).  th_mlp2_steps_supported: batch 16 / 32 / 48 / 64, hidden 32 / 64 / 128, in_features % 16 == 0. */
int th_mlp2_steps_supported(int batch, int in_features, int hidden, int classes);
int th_mlp2_steps(th_ctx *ctx, const float *d_x, const float *d_targets, int steps, int batch, int in_features, int hidden, int classes,
                  const th_adam_fuse *fuse4, float *d_loss, float *d_metrics, int64_t metrics_capacity, int64_t *d_state, int64_t advance,
                  int *h_err);


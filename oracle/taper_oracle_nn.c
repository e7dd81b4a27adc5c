/*
 * taper_oracle_nn.c -- CPU restatement of the reference's optimizers, data
 * gather, Sequential model and the train_mnist / train_mnist_cnn step.
 * TEST INFRASTRUCTURE ONLY (see taper_oracle.h).
 */
#include "taper_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------ optimizers */

/* f32::powi lowers to llvm.powi.f32 -> compiler-rt __powisf2: square-and-
 * multiply over the bits of |b|, reciprocal at the end for b < 0. */
float ot_powi(float a, int b) {
    const int recip = b < 0;
    float r = 1.0f;
    while (1) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0f / r : r;
}

struct ot_adam { /* optim.rs:43-52 */
    ot_tensor **params;
    int n;
    float lr, beta1, beta2, eps, weight_decay;
    float **m, **v;
    int t;
};

ot_adam *ot_adam_new(ot_tensor **params, int n, float lr, float beta1, float beta2, float eps,
                     float weight_decay) { /* optim.rs:54-81 */
    ot_adam *o = (ot_adam *)calloc(1, sizeof(ot_adam));
    o->params = (ot_tensor **)calloc((size_t)n, sizeof(ot_tensor *));
    o->m = (float **)calloc((size_t)n, sizeof(float *));
    o->v = (float **)calloc((size_t)n, sizeof(float *));
    for (int i = 0; i < n; ++i) {
        o->params[i] = ot_clone(params[i]);
        o->m[i] = (float *)calloc(ot_len(params[i]), sizeof(float));
        o->v[i] = (float *)calloc(ot_len(params[i]), sizeof(float));
    }
    o->n = n;
    o->lr = lr; o->beta1 = beta1; o->beta2 = beta2; o->eps = eps; o->weight_decay = weight_decay;
    return o;
}

void ot_adam_step(ot_adam *o) { /* optim.rs:83-113 */
    o->t += 1;
    float bc1 = 1.0f - ot_powi(o->beta1, o->t);
    float bc2 = 1.0f - ot_powi(o->beta2, o->t);
    float step_size = o->lr * (sqrtf(bc2) / bc1);
    for (int i = 0; i < o->n; ++i) {
        const float *grad = ot_grad(o->params[i]);
        if (!grad) continue; /* grad None: param skipped entirely (Q8) */
        float *data = ot_data_mut(o->params[i]);
        float *m = o->m[i], *v = o->v[i];
        size_t len = ot_len(o->params[i]);
        for (size_t j = 0; j < len; ++j) {
            float g = grad[j] + o->weight_decay * data[j];
            m[j] = o->beta1 * m[j] + (1.0f - o->beta1) * g;
            v[j] = o->beta2 * v[j] + (1.0f - o->beta2) * g * g;
            data[j] -= step_size * m[j] / (sqrtf(v[j]) + o->eps); /* eps outside sqrt, before bias fold: Q10 */
        }
    }
}

void ot_adam_zero_grad(ot_adam *o) { /* optim.rs:115-119 */
    for (int i = 0; i < o->n; ++i) ot_zero_grad(o->params[i]);
}

void ot_adam_set_lr(ot_adam *o, float lr) { o->lr = lr; }
float ot_adam_get_lr(const ot_adam *o) { return o->lr; }
int ot_adam_t(const ot_adam *o) { return o->t; }
const float *ot_adam_m(const ot_adam *o, int i) { return o->m[i]; }
const float *ot_adam_v(const ot_adam *o, int i) { return o->v[i]; }

void ot_adam_free(ot_adam *o) {
    if (!o) return;
    for (int i = 0; i < o->n; ++i) {
        ot_free(o->params[i]);
        free(o->m[i]);
        free(o->v[i]);
    }
    free(o->params); free(o->m); free(o->v);
    free(o);
}

void ot_sgd_step(ot_tensor **params, int n, float lr) { /* optim.rs:21-33; momentum ignored (14-17) */
    for (int i = 0; i < n; ++i) {
        const float *g = ot_grad(params[i]);
        if (!g) continue;
        float *d = ot_data_mut(params[i]);
        for (size_t j = 0; j < ot_len(params[i]); ++j) d[j] -= lr * g[j];
    }
}

/* ------------------------------------------------------------------ data */

void ot_get_batch(const float *images, const float *labels, const size_t *indices, size_t batch,
                  float *out_images, float *out_labels) { /* data/mnist.rs:277-310 */
    OT_PAR_FOR /* data/mnist.rs:291 */
    for (size_t i = 0; i < batch; ++i) {
        memcpy(out_images + i * 784, images + indices[i] * 784, 784 * sizeof(float));
        out_labels[i] = labels[indices[i]];
    }
}

/* ------------------------------------------------------------ sequential */

static ot_tensor *layer_forward(const ot_layer *l, const ot_tensor *x, int conv_mode) {
    switch (l->kind) {
    case OT_L_LINEAR: return ot_linear_forward(x, l->w, l->b);                  /* nn.rs:54-60 */
    case OT_L_RELU: return ot_relu(x);                                          /* activation.rs:10-12 */
    case OT_L_SIGMOID: return ot_sigmoid(x);                                    /* activation.rs:40-42 */
    case OT_L_CONV2D_RELU:                                                      /* nn.rs:470-479 */
        return ot_conv2d_relu(x, l->w, l->b, l->s_h, l->s_w, l->p_h, l->p_w, 1, 1, conv_mode);
    case OT_L_CONV2D:                                                           /* nn.rs:278-288 */
        return ot_conv2d(x, l->w, l->b, l->s_h, l->s_w, l->p_h, l->p_w, 1, 1, conv_mode);
    case OT_L_MAXPOOL:                                                          /* nn.rs:534-536 */
        return ot_max_pool2d(x, l->k_h, l->k_w, l->s_h, l->s_w, l->p_h, l->p_w, 1, NULL);
    case OT_L_AVGPOOL:                                                          /* nn.rs:593-608 */
        if (l->k_h == 0 && l->k_w == 0)
            return ot_avg_pool2d(x, (int)x->shape[2], (int)x->shape[3], 1, 1, 0, 0);
        return ot_avg_pool2d(x, l->k_h, l->k_w, l->s_h, l->s_w, l->p_h, l->p_w);
    case OT_L_ADAPTIVE_AVGPOOL: return ot_adaptive_avg_pool2d(x, l->out_h, l->out_w); /* nn.rs:670-686 */
    case OT_L_FLATTEN: return ot_flatten(x, l->start_dim);                      /* nn.rs:743-745 */
    default:
        fprintf(stderr, "taper_oracle: unknown layer kind %d\n", l->kind);
        abort();
    }
}

ot_tensor *ot_model_forward(const ot_model *m, const ot_tensor *x) { /* nn.rs:149-151: fold */
    ot_tensor *cur = ot_clone(x);
    for (int i = 0; i < m->n_layers; ++i) {
        ot_tensor *nx = layer_forward(&m->layers[i], cur, m->conv_mode);
        ot_free(cur);
        cur = nx;
    }
    return cur;
}

int ot_model_parameters(const ot_model *m, ot_tensor **out, int cap) { /* nn.rs:159-161 */
    int n = 0;
    for (int i = 0; i < m->n_layers; ++i) {
        const ot_layer *l = &m->layers[i];
        if (l->kind == OT_L_LINEAR || l->kind == OT_L_CONV2D_RELU || l->kind == OT_L_CONV2D) {
            if (n < cap) out[n] = l->w;
            n++;
            if (l->b) {
                if (n < cap) out[n] = l->b;
                n++;
            }
        }
    }
    return n;
}

void ot_train_step(const ot_model *m, ot_adam *opt, const float *images, const float *labels,
                   const size_t *x_shape, int x_ndim, float *loss_out, float *acc_out,
                   float *logits_out, float *grads_out, int *has_grad_out) {
    /* examples/train_mnist.rs:89-121, examples/train_mnist_cnn.rs:154-182 */
    ot_tape_reset();
    size_t batch = x_shape[0];
    ot_tensor *x = ot_new(images, x_shape, x_ndim);
    ot_tensor *y = ot_new(labels, &batch, 1);
    ot_tensor *logits = ot_model_forward(m, x);
    ot_tensor *loss = ot_cross_entropy_loss(logits, y);
    float acc = ot_accuracy(logits, y);
    ot_backward(loss);
    if (logits_out) memcpy(logits_out, ot_data(logits), ot_len(logits) * sizeof(float));
    if (grads_out || has_grad_out) {
        ot_tensor *params[256];
        int np = ot_model_parameters(m, params, 256);
        size_t off = 0;
        for (int i = 0; i < np; ++i) {
            const float *g = ot_grad(params[i]);
            size_t len = ot_len(params[i]);
            if (has_grad_out) has_grad_out[i] = g != NULL;
            if (grads_out) {
                if (g) memcpy(grads_out + off, g, len * sizeof(float));
                else memset(grads_out + off, 0, len * sizeof(float));
            }
            off += len;
        }
    }
    if (opt) {
        ot_adam_step(opt);
        ot_adam_zero_grad(opt);
    }
    if (loss_out) *loss_out = ot_data(loss)[0];
    if (acc_out) *acc_out = acc;
    ot_free(x); ot_free(y); ot_free(logits); ot_free(loss);
    ot_tape_reset(); /* drop captured clones so grads/activations are released */
}

/* ------------------------------------------------- cpu_baseline leg (bench.py) */
#ifdef _OPENMP
#include <omp.h>
#endif
/* threads of the rayon-style loops (rayon's default pool = one thread per logical CPU); 0 on a build without OpenMP */
int ot_baseline_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 0;
#endif
}
void ot_baseline_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* `steps` iterations of the examples' loop body over a resident dataset, batches in index order (wrapping):
 * DataLoader::next -> MNISTDataset::get_batch (data/mnist.rs:373-385,277-310), then the training step of
 * examples/train_mnist.rs:89-121.  One C call, so the timed region holds no interpreter time. */
void ot_baseline_run_steps(const ot_model *m, ot_adam *opt, const float *images, const float *labels, size_t n_rows,
                           const size_t *x_shape, int x_ndim, size_t steps, float *last_loss) {
    const size_t batch = x_shape[0];
    size_t *idx = (size_t *)malloc(batch * sizeof(size_t));
    float *xb = (float *)malloc(batch * 784 * sizeof(float)), *yb = (float *)malloc(batch * sizeof(float));
    size_t cur = 0;
    float loss = 0.0f, acc = 0.0f;
    for (size_t s = 0; s < steps; ++s) {
        if (cur + batch > n_rows) cur = 0;
        for (size_t i = 0; i < batch; ++i) idx[i] = cur + i;
        cur += batch;
        ot_get_batch(images, labels, idx, batch, xb, yb);
        ot_train_step(m, opt, xb, yb, x_shape, x_ndim, &loss, &acc, NULL, NULL, NULL);
    }
    if (last_loss) *last_loss = loss;
    free(idx); free(xb); free(yb);
}


/*
 * taper_oracle.c -- CPU restatement of the reference's tensor/tape/ops layer.
 * TEST INFRASTRUCTURE ONLY (see taper_oracle.h).  Each function cites the
 * reference file:line it follows; quirk ids (Q1..Q15) refer to SURVEY.md A.1.
 */
#include "taper_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define OT_CHECK(cond, ...)                                                   \
    do {                                                                      \
        if (!(cond)) {                                                        \
            fprintf(stderr, "taper_oracle: " __VA_ARGS__);                    \
            fprintf(stderr, " (%s:%d)\n", __FILE__, __LINE__);                \
            abort(); /* the reference panics (assert!/assert_eq!) */          \
        }                                                                     \
    } while (0)

/* ------------------------------------------------------------------ core */

struct ot_core {
    float *data;
    size_t len;
    float *grad; /* NULL == None */
    size_t tape_node;
    int refs;
};

static float *xalloc_f(size_t n) {
    float *p = (float *)calloc(n ? n : 1, sizeof(float));
    OT_CHECK(p, "out of memory (%zu floats)", n);
    return p;
}

static size_t shape_len(const size_t *shape, int ndim) {
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    return n;
}

/* takes ownership of `data` */
static ot_tensor *ot_wrap(float *data, const size_t *shape, int ndim) {
    OT_CHECK(ndim >= 1 && ndim <= OT_MAX_DIMS, "ndim %d unsupported", ndim);
    ot_tensor *t = (ot_tensor *)calloc(1, sizeof(ot_tensor));
    t->core = (ot_core *)calloc(1, sizeof(ot_core));
    t->core->data = data;
    t->core->len = shape_len(shape, ndim);
    t->core->refs = 1;
    t->ndim = ndim;
    for (int i = 0; i < ndim; ++i) t->shape[i] = shape[i];
    return t;
}

ot_tensor *ot_new(const float *data, const size_t *shape, int ndim) { /* tensor.rs:470-478 */
    size_t n = shape_len(shape, ndim);
    float *d = xalloc_f(n);
    if (data) memcpy(d, data, n * sizeof(float));
    return ot_wrap(d, shape, ndim);
}

ot_tensor *ot_scalar(float v) { /* tensor.rs:480-482 */
    size_t one = 1;
    return ot_new(&v, &one, 1);
}

ot_tensor *ot_clone(const ot_tensor *t) { /* #[derive(Clone)] tensor.rs:236 */
    ot_tensor *c = (ot_tensor *)malloc(sizeof(ot_tensor));
    *c = *t;
    c->core->refs++;
    return c;
}

void ot_free(ot_tensor *t) {
    if (!t) return;
    if (--t->core->refs == 0) {
        free(t->core->data);
        free(t->core->grad);
        free(t->core);
    }
    free(t);
}

void ot_set_requires_grad(ot_tensor *t, int on) { t->requires_grad = on; }
size_t ot_len(const ot_tensor *t) { return t->core->len; }
const float *ot_data(const ot_tensor *t) { return t->core->data; }
float *ot_data_mut(ot_tensor *t) { return t->core->data; }
const float *ot_grad(const ot_tensor *t) { return t->core->grad; }
size_t ot_tape_node(const ot_tensor *t) { return t->core->tape_node; }

void ot_set_grad(ot_tensor *t, const float *g) {
    free(t->core->grad);
    t->core->grad = NULL;
    if (g) {
        t->core->grad = xalloc_f(t->core->len);
        memcpy(t->core->grad, g, t->core->len * sizeof(float));
    }
}

void ot_zero_grad(ot_tensor *t) { /* tensor.rs:531-533: grad = None */
    free(t->core->grad);
    t->core->grad = NULL;
}

/* lazily create a zeroed grad slot of n elements (the `slot.is_none()` idiom) */
static float *grad_slot(const ot_tensor *t, size_t n) {
    if (!t->core->grad) t->core->grad = xalloc_f(n);
    return t->core->grad;
}

/* ------------------------------------------------------------------ tape */

enum {
    N_ADD, N_MUL, N_SUB, N_DIV, N_MATMUL, N_RELU, N_TRANSPOSE, N_SIGMOID, N_ADD_BCAST,
    N_SUB_BCAST_ROWS, N_MEAN, N_RESHAPE, N_SUM_DIM, N_SUM_ALL, N_EXP, N_LOG, N_POW,
    N_MAXPOOL, N_AVGPOOL, N_ADD_BIAS_4D, N_CROSS_ENTROPY, N_BCE, N_IM2COL_FULL, N_TRANSPOSE4D_FULL
};

typedef struct ot_node {
    int kind;
    ot_tensor *a, *b, *out;  /* handle clones captured by the closure */
    ot_tensor *aux;          /* extra captured tensor (logp for CE) */
    float *saved;            /* captured Vec<f32> (sigmoid out, exp values) */
    size_t *saved_idx;       /* captured Vec<usize> (max-pool argmax) */
    size_t dims[12];         /* captured geometry */
    float fparam;
    int iparam[4];
} ot_node;

static ot_node *g_nodes = NULL;
static size_t g_n_nodes = 0, g_cap_nodes = 0;
static int g_zero_sentinel = 1;

static void node_release(ot_node *n) {
    ot_free(n->a);
    ot_free(n->b);
    ot_free(n->out);
    ot_free(n->aux);
    free(n->saved);
    free(n->saved_idx);
}

void ot_tape_reset(void) { /* tape.rs:43-49 */
    for (size_t i = 0; i < g_n_nodes; ++i) node_release(&g_nodes[i]);
    g_n_nodes = 0;
}

size_t ot_tape_len(void) { return g_n_nodes; }
void ot_tape_set_zero_sentinel(int on) { g_zero_sentinel = on; }

/* tape.rs:51-101: record only if an input requires grad; id = nodes.len()
 * before the push; stamp the output's tape_node. */
static ot_node *tape_push(int kind, const ot_tensor *a, const ot_tensor *b, const ot_tensor *out) {
    int need = (a && a->requires_grad) || (b && b->requires_grad);
    if (!need) return NULL;
    if (g_n_nodes == g_cap_nodes) {
        g_cap_nodes = g_cap_nodes ? g_cap_nodes * 2 : 64;
        g_nodes = (ot_node *)realloc(g_nodes, g_cap_nodes * sizeof(ot_node));
    }
    size_t id = g_n_nodes++;
    ot_node *n = &g_nodes[id];
    memset(n, 0, sizeof(*n));
    n->kind = kind;
    n->a = a ? ot_clone(a) : NULL;
    n->b = b ? ot_clone(b) : NULL;
    n->out = ot_clone(out);
    out->core->tape_node = id;
    return n;
}

static void node_backward(size_t id);

/* tape.rs:106-127: snapshot closures [0..=min(id,len-1)], run in reverse. */
static void tape_backward(size_t final_id) {
    if (g_n_nodes == 0) return;
    size_t end = final_id < g_n_nodes - 1 ? final_id : g_n_nodes - 1;
    for (size_t i = end + 1; i-- > 0;) node_backward(i);
}

void ot_backward(ot_tensor *t) { /* tensor.rs:520-529 */
    free(t->core->grad);
    t->core->grad = xalloc_f(t->core->len);
    for (size_t i = 0; i < t->core->len; ++i) t->core->grad[i] = 1.0f;
    size_t id = t->core->tape_node;
    if (id != 0 || !g_zero_sentinel) tape_backward(id); /* Q1 */
}

/* ops.rs:124-137: g = g + src through a temp */
static void accumulate_grad(const ot_tensor *t, const float *src) {
    float *g = grad_slot(t, t->core->len);
    for (size_t i = 0; i < t->core->len; ++i) g[i] = g[i] + src[i];
}

/* ops.rs:140-151 */
static void accumulate_grad_scaled(const ot_tensor *t, const float *src, float scale) {
    float *g = grad_slot(t, t->core->len);
    for (size_t i = 0; i < t->core->len; ++i) g[i] += scale * src[i];
}

/* ------------------------------------------------------------------ gemm */

/* gemm.rs:72-119: C[m,n] = alpha * op(A)[m,k] * op(B)[k,n] + beta * C, row
 * major; op(A) strides (k,1) or (1,m); op(B) strides (n,1) or (1,k).
 * The inner product itself lives in the un-vendored crate matrixmultiply
 * 0.3.10 (Cargo.lock); it computes the alpha*A*B panel and then combines it
 * with beta*C (beta == 0 overwrites).  Restated as: per output row an fp32
 * k-ordered accumulation, then c = beta*c + alpha*acc. */
#ifdef OT_PACKED_SGEMM
/* Baseline build only (`make fast`, bench.py's cpu_baseline leg): the GEMM goes to the matrixmultiply-style packed
 * kernel (cpu_packed_sgemm.c) or, once ot_baseline_set_cblas() has been given a vendor cblas_sgemm, through the
 * `--features blas` call of gemm.rs:32-47.  The parity oracle is built WITHOUT this macro. */
void ot_packed_sgemm_rowmajor(int trans_a, int trans_b, int m, int n, int k, float alpha, const float *a, const float *b, float beta,
                              float *c);
typedef void (*ot_cblas32_fn)(int, int, int, int, int, int, float, const float *, int, const float *, int, float, float *, int);
typedef void (*ot_cblas64_fn)(int, int, int, int64_t, int64_t, int64_t, float, const float *, int64_t, const float *, int64_t, float,
                              float *, int64_t);
static void *g_cblas_fn = NULL;
static int g_cblas_ilp64 = 0;
void ot_baseline_set_cblas(void *cblas_sgemm_fn, int ilp64) { g_cblas_fn = cblas_sgemm_fn; g_cblas_ilp64 = ilp64; }
#endif
int ot_baseline_flavour(void) { /* 0 = plain-loop parity oracle; 1 = packed sgemm; +2 = OpenMP plane loops */
    int f = 0;
#ifdef OT_PACKED_SGEMM
    f |= 1;
#endif
#ifdef _OPENMP
    f |= 2;
#endif
    return f;
}

void ot_sgemm_rowmajor(int trans_a, int trans_b, int m, int n, int k, float alpha,
                       const float *a, const float *b, float beta, float *c) {
#ifdef OT_PACKED_SGEMM
    if (g_cblas_fn) { /* gemm.rs:21-47: RowMajor = 101, NoTrans = 111, Trans = 112; lda = m|k, ldb = k|n, ldc = n */
        const int ta = trans_a ? 112 : 111, tb = trans_b ? 112 : 111;
        const int lda = trans_a ? m : k, ldb = trans_b ? k : n;
        if (g_cblas_ilp64) ((ot_cblas64_fn)g_cblas_fn)(101, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, n);
        else ((ot_cblas32_fn)g_cblas_fn)(101, ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, n);
    } else {
        ot_packed_sgemm_rowmajor(trans_a, trans_b, m, n, k, alpha, a, b, beta, c);
    }
    return;
#endif
    const long a_rs = trans_a ? 1 : k, a_cs = trans_a ? m : 1;
    const long b_rs = trans_b ? 1 : n, b_cs = trans_b ? k : 1;
    float *acc = xalloc_f((size_t)n);
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < n; ++j) acc[j] = 0.0f;
        if (!trans_b) {
            for (int p = 0; p < k; ++p) {
                const float av = a[i * a_rs + p * a_cs];
                const float *brow = b + (long)p * b_rs;
                for (int j = 0; j < n; ++j) acc[j] += av * brow[j];
            }
        } else {
            for (int j = 0; j < n; ++j) {
                const float *bcol = b + (long)j * b_cs; /* contiguous in p */
                float s = 0.0f;
                for (int p = 0; p < k; ++p) s += a[i * a_rs + p * a_cs] * bcol[p];
                acc[j] = s;
            }
        }
        float *crow = c + (long)i * n;
        if (beta == 0.0f) {
            for (int j = 0; j < n; ++j) crow[j] = alpha * acc[j];
        } else {
            for (int j = 0; j < n; ++j) crow[j] = beta * crow[j] + alpha * acc[j];
        }
    }
    free(acc);
}

/* --------------------------------------------------------- element-wise */

static ot_tensor *like(const ot_tensor *t, float *data) { return ot_wrap(data, t->shape, t->ndim); }

static void check_same_len(const ot_tensor *a, const ot_tensor *b) {
    /* ops.rs:11-15: only the element COUNT is checked (Q12) */
    OT_CHECK(a->core->len == b->core->len, "Tensor dimensions must match");
}

ot_tensor *ot_add(const ot_tensor *a, const ot_tensor *b) { /* ops.rs:8-51 */
    check_same_len(a, b);
    size_t n = a->core->len;
    float *o = xalloc_f(n);
    for (size_t i = 0; i < n; ++i) o[i] = a->core->data[i] + b->core->data[i];
    ot_tensor *out = like(a, o);
    if (a->requires_grad || b->requires_grad) {
        out->requires_grad = 1;
        tape_push(N_ADD, a, b, out);
    }
    return out;
}

ot_tensor *ot_mul(const ot_tensor *a, const ot_tensor *b) { /* ops.rs:53-120 */
    check_same_len(a, b);
    size_t n = a->core->len;
    float *o = xalloc_f(n);
    for (size_t i = 0; i < n; ++i) o[i] = a->core->data[i] * b->core->data[i];
    ot_tensor *out = like(a, o);
    if (a->requires_grad || b->requires_grad) {
        out->requires_grad = 1;
        tape_push(N_MUL, a, b, out);
    }
    return out;
}

ot_tensor *ot_sub(const ot_tensor *a, const ot_tensor *b) { /* ops.rs:377-416 */
    check_same_len(a, b);
    size_t n = a->core->len;
    float *o = xalloc_f(n);
    for (size_t i = 0; i < n; ++i) o[i] = a->core->data[i] - b->core->data[i];
    ot_tensor *out = like(a, o);
    if (a->requires_grad || b->requires_grad) {
        out->requires_grad = 1;
        tape_push(N_SUB, a, b, out);
    }
    return out;
}

ot_tensor *ot_div(const ot_tensor *a, const ot_tensor *b) { /* ops.rs:440-496 */
    check_same_len(a, b);
    size_t n = a->core->len;
    float *o = xalloc_f(n);
    for (size_t i = 0; i < n; ++i) o[i] = a->core->data[i] / b->core->data[i];
    ot_tensor *out = like(a, o);
    if (a->requires_grad || b->requires_grad) {
        out->requires_grad = 1;
        tape_push(N_DIV, a, b, out);
    }
    return out;
}

ot_tensor *ot_matmul(const ot_tensor *a, const ot_tensor *b) { /* ops.rs:200-298 */
    OT_CHECK(a->ndim == 2, "First tensor must be 2D");
    OT_CHECK(b->ndim == 2, "Second tensor must be 2D");
    int m = (int)a->shape[0], k = (int)a->shape[1], k2 = (int)b->shape[0], n = (int)b->shape[1];
    OT_CHECK(k == k2, "Inner dimensions must match: %d vs %d", k, k2);
    float *c = xalloc_f((size_t)m * n);
    ot_sgemm_rowmajor(0, 0, m, n, k, 1.0f, a->core->data, b->core->data, 0.0f, c);
    size_t shp[2] = {(size_t)m, (size_t)n};
    ot_tensor *out = ot_wrap(c, shp, 2);
    if (a->requires_grad || b->requires_grad) {
        out->requires_grad = 1;
        tape_push(N_MATMUL, a, b, out);
    }
    return out;
}

ot_tensor *ot_relu(const ot_tensor *x) { /* ops.rs:312-374; _mm_max_ps(v, 0) */
    size_t n = x->core->len;
    float *o = xalloc_f(n);
    for (size_t i = 0; i < n; ++i) {
        float v = x->core->data[i];
        o[i] = v > 0.0f ? v : 0.0f; /* maxps(v,0): NaN -> 0 (second operand) */
    }
    ot_tensor *out = like(x, o);
    if (x->requires_grad) {
        out->requires_grad = 1;
        tape_push(N_RELU, x, NULL, out);
    }
    return out;
}

ot_tensor *ot_transpose(const ot_tensor *x) { /* tensor.rs:544-591 */
    OT_CHECK(x->ndim == 2, "Can only transpose 2D tensors");
    size_t rows = x->shape[0], cols = x->shape[1];
    float *o = xalloc_f(rows * cols);
    const size_t block = 16; /* tensor.rs:551: "Optimal for most cache sizes" -- same values, the reference's loop order */
    for (size_t i0 = 0; i0 < rows; i0 += block)
        for (size_t j0 = 0; j0 < cols; j0 += block) {
            size_t i_max = i0 + block < rows ? i0 + block : rows, j_max = j0 + block < cols ? j0 + block : cols;
            for (size_t i = i0; i < i_max; ++i)
                for (size_t j = j0; j < j_max; ++j) o[j * rows + i] = x->core->data[i * cols + j];
        }
    size_t shp[2] = {cols, rows};
    ot_tensor *out = ot_wrap(o, shp, 2);
    if (x->requires_grad) {
        out->requires_grad = 1;
        ot_node *n = tape_push(N_TRANSPOSE, x, NULL, out);
        n->dims[0] = rows;
        n->dims[1] = cols;
    }
    return out;
}

ot_tensor *ot_sigmoid(const ot_tensor *x) { /* tensor.rs:594-634 */
    size_t n = x->core->len;
    float *o = xalloc_f(n);
    for (size_t i = 0; i < n; ++i) {
        float v = x->core->data[i];
        if (v > 0.0f) {
            float e = expf(-v);
            o[i] = 1.0f / (1.0f + e);
        } else {
            float e = expf(v);
            o[i] = e / (1.0f + e);
        }
    }
    ot_tensor *out = like(x, o);
    if (x->requires_grad) {
        out->requires_grad = 1;
        ot_node *nd = tape_push(N_SIGMOID, x, NULL, out);
        nd->saved = xalloc_f(n); /* out_data cloned BEFORE the closure (616) */
        memcpy(nd->saved, o, n * sizeof(float));
    }
    return out;
}

static int same_shape(const ot_tensor *a, const ot_tensor *b) {
    if (a->ndim != b->ndim) return 0;
    for (int i = 0; i < a->ndim; ++i)
        if (a->shape[i] != b->shape[i]) return 0;
    return 1;
}

ot_tensor *ot_add_broadcast(const ot_tensor *a, const ot_tensor *b) { /* tensor.rs:636-704 */
    if (same_shape(a, b)) return ot_add(a, b);
    OT_CHECK(a->ndim == 2 && b->ndim == 1, "Unsupported broadcasting shapes");
    OT_CHECK(a->shape[1] == b->shape[0], "Last dimension must match for broadcasting");
    size_t bs = a->shape[0], f = a->shape[1];
    float *o = xalloc_f(bs * f);
    for (size_t r = 0; r < bs; ++r)
        for (size_t j = 0; j < f; ++j) o[r * f + j] = a->core->data[r * f + j] + b->core->data[j];
    ot_tensor *out = like(a, o);
    if (a->requires_grad || b->requires_grad) {
        out->requires_grad = 1;
        ot_node *n = tape_push(N_ADD_BCAST, a, b, out);
        n->dims[0] = bs;
        n->dims[1] = f;
    }
    return out;
}

ot_tensor *ot_sub_broadcast_rows(const ot_tensor *a, const ot_tensor *r) { /* tensor.rs:707-770 */
    if (same_shape(a, r)) return ot_sub(a, r);
    OT_CHECK(a->ndim == 2 && r->ndim == 2 && a->shape[0] == r->shape[0] && r->shape[1] == 1,
             "Unsupported broadcasting shapes for sub_broadcast_rows");
    size_t b = a->shape[0], c = a->shape[1];
    float *o = xalloc_f(b * c);
    for (size_t row = 0; row < b; ++row) {
        float rv = r->core->data[row];
        for (size_t col = 0; col < c; ++col) o[row * c + col] = a->core->data[row * c + col] - rv;
    }
    ot_tensor *out = like(a, o);
    if (a->requires_grad || r->requires_grad) {
        out->requires_grad = 1;
        tape_push(N_SUB_BCAST_ROWS, a, r, out);
    }
    return out;
}

ot_tensor *ot_mean(const ot_tensor *x) { /* tensor.rs:772-800 */
    size_t n = x->core->len;
    float s = 0.0f;
    for (size_t i = 0; i < n; ++i) s += x->core->data[i];
    ot_tensor *out = ot_scalar(s / (float)n);
    if (x->requires_grad) {
        out->requires_grad = 1;
        ot_node *nd = tape_push(N_MEAN, x, NULL, out);
        nd->fparam = (float)n;
    }
    return out;
}

ot_tensor *ot_reshape(const ot_tensor *x, const size_t *shape, int ndim) { /* tensor.rs:803-840 */
    OT_CHECK(shape_len(shape, ndim) == x->core->len, "Cannot reshape tensor of size %zu", x->core->len);
    ot_tensor *out = ot_new(x->core->data, shape, ndim); /* data copy (814) */
    if (x->requires_grad) {
        out->requires_grad = 1;
        tape_push(N_RESHAPE, x, NULL, out);
    }
    return out;
}

ot_tensor *ot_flatten(const ot_tensor *x, int start_dim) { /* tensor.rs:843-858 */
    OT_CHECK(start_dim < x->ndim, "start_dim out of bounds");
    size_t shp[OT_MAX_DIMS];
    int nd = 0;
    for (int i = 0; i < start_dim; ++i) shp[nd++] = x->shape[i];
    size_t rest = 1;
    for (int i = start_dim; i < x->ndim; ++i) rest *= x->shape[i];
    shp[nd++] = rest;
    return ot_reshape(x, shp, nd);
}

ot_tensor *ot_squeeze(const ot_tensor *x, int dim) { /* tensor.rs:861-877 */
    size_t shp[OT_MAX_DIMS];
    int nd = 0;
    if (dim >= 0) {
        OT_CHECK(dim < x->ndim, "Dimension out of bounds");
        OT_CHECK(x->shape[dim] == 1, "Can only squeeze dimensions of size 1");
        for (int i = 0; i < x->ndim; ++i)
            if (i != dim) shp[nd++] = x->shape[i];
    } else {
        for (int i = 0; i < x->ndim; ++i)
            if (x->shape[i] != 1) shp[nd++] = x->shape[i];
    }
    if (nd == 0) { /* a Rust reshape(&[]) would give a 0-d tensor of product 1 */
        shp[0] = 1;
        nd = 1;
    }
    return ot_reshape(x, shp, nd);
}

ot_tensor *ot_unsqueeze(const ot_tensor *x, int dim) { /* tensor.rs:880-887 */
    OT_CHECK(dim <= x->ndim && x->ndim < OT_MAX_DIMS, "Dimension out of bounds");
    size_t shp[OT_MAX_DIMS];
    int nd = 0;
    for (int i = 0; i < x->ndim; ++i) {
        if (i == dim) shp[nd++] = 1;
        shp[nd++] = x->shape[i];
    }
    if (dim == x->ndim) shp[nd++] = 1;
    return ot_reshape(x, shp, nd);
}

/* tensor.rs:917-937: which output element does input element i feed? */
static size_t sum_out_index(size_t i, const size_t *in_shape, int ndim, int d, int keepdim,
                            const size_t *out_shape, int out_ndim) {
    size_t idx = i, out_idx = 0, multiplier = 1;
    for (int j = ndim - 1; j >= 0; --j) {
        size_t coord = idx % in_shape[j];
        idx /= in_shape[j];
        if (j != d) {
            int out_j = (j > d && !keepdim) ? j - 1 : j;
            if (out_j < out_ndim) {
                out_idx += coord * multiplier;
                multiplier *= out_shape[out_j];
            }
        }
    }
    return out_idx;
}

ot_tensor *ot_sum(const ot_tensor *x, int dim, int keepdim) { /* tensor.rs:890-1018 */
    if (dim >= 0) {
        OT_CHECK(dim < x->ndim, "Dimension %d out of bounds", dim);
        size_t out_shape[OT_MAX_DIMS];
        int out_nd = 0;
        for (int i = 0; i < x->ndim; ++i) {
            if (i == dim) {
                if (keepdim) out_shape[out_nd++] = 1;
            } else {
                out_shape[out_nd++] = x->shape[i];
            }
        }
        int wrap_nd = out_nd;
        size_t wrap_shape[OT_MAX_DIMS];
        memcpy(wrap_shape, out_shape, sizeof(out_shape));
        if (wrap_nd == 0) { /* [n].sum(0,false) -> shape [] in Rust; carry as [1] */
            wrap_shape[0] = 1;
            wrap_nd = 1;
        }
        size_t out_size = shape_len(out_shape, out_nd);
        float *res = xalloc_f(out_size);
        for (size_t i = 0; i < x->core->len; ++i) {
            size_t oi = sum_out_index(i, x->shape, x->ndim, dim, keepdim, out_shape, out_nd);
            res[oi] += x->core->data[i];
        }
        ot_tensor *out = ot_wrap(res, wrap_shape, wrap_nd);
        if (x->requires_grad) {
            out->requires_grad = 1;
            ot_node *n = tape_push(N_SUM_DIM, x, NULL, out);
            n->iparam[0] = dim;
            n->iparam[1] = keepdim;
        }
        return out;
    }
    float s = 0.0f; /* iter().sum(): sequential fp32 */
    for (size_t i = 0; i < x->core->len; ++i) s += x->core->data[i];
    ot_tensor *out = ot_scalar(s);
    if (x->requires_grad) {
        out->requires_grad = 1;
        tape_push(N_SUM_ALL, x, NULL, out);
    }
    return out;
}

ot_tensor *ot_max(const ot_tensor *x, int dim, ot_tensor **indices_out) { /* tensor.rs:1021-1083 */
    if (dim >= 0) {
        OT_CHECK(dim < x->ndim, "Dimension %d out of bounds", dim);
        size_t out_shape[OT_MAX_DIMS];
        for (int i = 0; i < x->ndim; ++i) out_shape[i] = x->shape[i];
        out_shape[dim] = 1;
        size_t out_size = shape_len(out_shape, x->ndim);
        float *mv = xalloc_f(out_size), *mi = xalloc_f(out_size);
        for (size_t i = 0; i < out_size; ++i) mv[i] = -INFINITY;
        for (size_t i = 0; i < x->core->len; ++i) {
            size_t idx = i, out_idx = 0, dim_idx = 0, multiplier = 1;
            for (int j = x->ndim - 1; j >= 0; --j) {
                size_t coord = idx % x->shape[j];
                idx /= x->shape[j];
                if (j == dim) {
                    dim_idx = coord;
                } else {
                    out_idx += coord * multiplier;
                    multiplier *= (j < dim) ? x->shape[j] : 1; /* 1056, Q14 */
                }
            }
            if (out_idx > out_size - 1) out_idx = out_size - 1;
            if (x->core->data[i] > mv[out_idx]) { /* strict >: first max wins, NaN never wins */
                mv[out_idx] = x->core->data[i];
                mi[out_idx] = (float)dim_idx;
            }
        }
        ot_tensor *vals = ot_wrap(mv, out_shape, x->ndim);
        ot_tensor *idxs = ot_wrap(mi, out_shape, x->ndim);
        if (indices_out) *indices_out = idxs; else ot_free(idxs);
        return vals;
    }
    /* global max: max_by(partial_cmp) keeps the LAST maximal element */
    float best = 0.0f;
    size_t best_i = 0;
    for (size_t i = 0; i < x->core->len; ++i) {
        /* partial_cmp(..).unwrap() panics on the first comparison that involves a NaN (every element of a len >= 2 input is compared) */
        OT_CHECK(!(x->core->len >= 2 && isnan(x->core->data[i])), "called `Option::unwrap()` on a `None` value");
        if (i == 0 || x->core->data[i] >= best) {
            best = x->core->data[i];
            best_i = i;
        }
    }
    if (indices_out) *indices_out = ot_scalar((float)best_i);
    return ot_scalar(best);
}

ot_tensor *ot_argmax(const ot_tensor *x, int dim) { /* tensor.rs:1086-1088 */
    ot_tensor *idx = NULL;
    ot_tensor *v = ot_max(x, dim, &idx);
    ot_free(v);
    return idx;
}

ot_tensor *ot_exp(const ot_tensor *x) { /* tensor.rs:1091-1133 */
    size_t n = x->core->len;
    float *o = xalloc_f(n);
    for (size_t i = 0; i < n; ++i) o[i] = expf(x->core->data[i]);
    ot_tensor *out = like(x, o);
    if (x->requires_grad) {
        out->requires_grad = 1;
        ot_node *nd = tape_push(N_EXP, x, NULL, out);
        nd->saved = xalloc_f(n);
        memcpy(nd->saved, o, n * sizeof(float));
    }
    return out;
}

ot_tensor *ot_log(const ot_tensor *x) { /* tensor.rs:1136-1169 */
    size_t n = x->core->len;
    float *o = xalloc_f(n);
    for (size_t i = 0; i < n; ++i) o[i] = logf(x->core->data[i]);
    ot_tensor *out = like(x, o);
    if (x->requires_grad) {
        out->requires_grad = 1;
        tape_push(N_LOG, x, NULL, out);
    }
    return out;
}

ot_tensor *ot_pow(const ot_tensor *x, float e) { /* tensor.rs:1172-1206 */
    size_t n = x->core->len;
    float *o = xalloc_f(n);
    for (size_t i = 0; i < n; ++i) o[i] = powf(x->core->data[i], e);
    ot_tensor *out = like(x, o);
    if (x->requires_grad) {
        out->requires_grad = 1;
        ot_node *nd = tape_push(N_POW, x, NULL, out);
        nd->fparam = e;
    }
    return out;
}

ot_tensor *ot_sqrt(const ot_tensor *x) { return ot_pow(x, 0.5f); } /* tensor.rs:1209-1211 */

/* ------------------------------------------------------------ conv / pool */

/* tensor.rs:1728-1780: im2col for 3x3 stride 1 dilation 1; col[(n,oh,ow), ch*9+kh*3+kw] */
static void im2col_3x3_s1(const float *in, float *col, size_t n, size_t c, size_t h_in, size_t w_in,
                          size_t h_out, size_t w_out, size_t pad_h, size_t pad_w) {
    size_t col_size = c * 9;
    size_t windows = n * h_out * w_out;
    OT_PAR_FOR /* tensor.rs:1745 */
    for (size_t w_idx = 0; w_idx < windows; ++w_idx) {
        size_t batch = w_idx / (h_out * w_out), pos = w_idx % (h_out * w_out);
        size_t oh = pos / w_out, ow = pos % w_out;
        size_t batch_off = batch * c * h_in * w_in;
        float *out = col + w_idx * col_size;
        for (size_t ch = 0; ch < c; ++ch) {
            for (size_t kr = 0; kr < 3; ++kr) {
                size_t ih = oh + kr;
                int hv = ih >= pad_h && ih < h_in + pad_h;
                size_t ihx = hv ? ih - pad_h : 0;
                for (size_t kc = 0; kc < 3; ++kc) {
                    size_t iw = ow + kc;
                    size_t idx = ch * 9 + kr * 3 + kc;
                    if (hv && iw >= pad_w && iw < w_in + pad_w)
                        out[idx] = in[batch_off + ch * h_in * w_in + ihx * w_in + (iw - pad_w)];
                    else
                        out[idx] = 0.0f;
                }
            }
        }
    }
}

/* tensor.rs:1972-2031 */
static ot_tensor *add_bias_4d(const ot_tensor *x, const ot_tensor *bias) {
    OT_CHECK(x->ndim == 4 && bias->ndim == 1 && x->shape[1] == bias->shape[0], "add_bias_4d shapes");
    size_t n = x->shape[0], c = x->shape[1], hw = x->shape[2] * x->shape[3];
    float *o = xalloc_f(x->core->len);
    for (size_t b = 0; b < n; ++b)
        for (size_t ch = 0; ch < c; ++ch) {
            float bv = bias->core->data[ch];
            size_t base = b * c * hw + ch * hw;
            for (size_t s = 0; s < hw; ++s) o[base + s] = x->core->data[base + s] + bv;
        }
    ot_tensor *out = like(x, o);
    if (x->requires_grad || bias->requires_grad) {
        out->requires_grad = 1;
        ot_node *nd = tape_push(N_ADD_BIAS_4D, x, bias, out);
        nd->dims[0] = n;
        nd->dims[1] = c;
        nd->dims[2] = hw;
    }
    return out;
}

/* tensor.rs:2034-2076 with axes [0,3,1,2]: NHWC -> NCHW.  The reference
 * returns a plain Tensor::new (no node, requires_grad=false): Q2. */
static ot_tensor *transpose_nhwc_to_nchw(const ot_tensor *x, int full_backward) {
    size_t d0 = x->shape[0], d1 = x->shape[1], d2 = x->shape[2], d3 = x->shape[3];
    float *o = xalloc_f(x->core->len);
    for (size_t i0 = 0; i0 < d0; ++i0)
        for (size_t i1 = 0; i1 < d1; ++i1)
            for (size_t i2 = 0; i2 < d2; ++i2)
                for (size_t i3 = 0; i3 < d3; ++i3)
                    o[((i0 * d3 + i3) * d1 + i1) * d2 + i2] = x->core->data[((i0 * d1 + i1) * d2 + i2) * d3 + i3];
    size_t shp[4] = {d0, d3, d1, d2};
    ot_tensor *out = ot_wrap(o, shp, 4);
    if (full_backward && x->requires_grad) {
        out->requires_grad = 1;
        tape_push(N_TRANSPOSE4D_FULL, x, NULL, out);
    }
    return out;
}

ot_tensor *ot_conv2d(const ot_tensor *x, const ot_tensor *w, const ot_tensor *bias, int stride_h,
                     int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int mode) {
    /* tensor.rs:1221-1285 */
    OT_CHECK(x->ndim == 4, "Input must be 4D: [N, C_in, H, W]");
    OT_CHECK(w->ndim == 4, "Weight must be 4D: [C_out, C_in, K_h, K_w]");
    size_t n = x->shape[0], c_in = x->shape[1], h_in = x->shape[2], w_in = x->shape[3];
    size_t c_out = w->shape[0], k_h = w->shape[2], k_w = w->shape[3];
    OT_CHECK(c_in == w->shape[1], "Input and weight channel dimensions must match");
    size_t h_out = (h_in + 2 * pad_h - dil_h * (k_h - 1) - 1) / stride_h + 1;
    size_t w_out = (w_in + 2 * pad_w - dil_w * (k_w - 1) - 1) / stride_w + 1;
    size_t k = c_in * k_h * k_w;
    size_t windows = n * h_out * w_out;

    /* im2col_optimized (tensor.rs:1663-1726): fresh tensor, no node (Q2) */
    float *col = xalloc_f(windows * k);
    if (k_h == 3 && k_w == 3 && stride_h == 1 && stride_w == 1 && dil_h == 1 && dil_w == 1) {
        im2col_3x3_s1(x->core->data, col, n, c_in, h_in, w_in, h_out, w_out, pad_h, pad_w);
    } else if (k_h == 1 && k_w == 1) {
        /* tensor.rs:1784-1802: raw memcpy of the NCHW buffer (Q4) */
        OT_CHECK(h_in == h_out && w_in == w_out, "im2col_1x1 requires h_in == h_out");
        memcpy(col, x->core->data, x->core->len * sizeof(float));
    } else {
        OT_CHECK(0, "general-stride im2col is out of scope (reference Q9 indexing bug)");
    }
    size_t col_shape[2] = {windows, k};
    ot_tensor *col_t = ot_wrap(col, col_shape, 2);
    if (mode == 1 && x->requires_grad) {
        col_t->requires_grad = 1;
        ot_node *nd = tape_push(N_IM2COL_FULL, x, NULL, col_t);
        nd->dims[0] = n; nd->dims[1] = c_in; nd->dims[2] = h_in; nd->dims[3] = w_in;
        nd->dims[4] = h_out; nd->dims[5] = w_out; nd->dims[6] = (size_t)pad_h; nd->dims[7] = (size_t)pad_w;
        nd->dims[8] = k_h; /* 3 or 1 */
    }

    size_t w2_shape[2] = {k, c_out};
    ot_tensor *w2 = ot_reshape(w, w2_shape, 2);   /* 1262: reinterpretation, Q3 */
    ot_tensor *out2d = ot_matmul(col_t, w2);      /* 1265 */
    size_t nhwc[4] = {n, h_out, w_out, c_out};
    ot_tensor *o4 = ot_reshape(out2d, nhwc, 4);   /* 1275 */
    ot_tensor *nchw = transpose_nhwc_to_nchw(o4, mode == 1); /* 1276 */
    ot_tensor *res = nchw;
    if (bias) {
        OT_CHECK(bias->ndim == 1 && bias->shape[0] == c_out, "Bias must be 1D with C_out elements");
        res = add_bias_4d(nchw, bias);            /* 1281 */
        ot_free(nchw);
    }
    ot_free(col_t);
    ot_free(w2);
    ot_free(out2d);
    ot_free(o4);
    return res;
}

ot_tensor *ot_conv2d_relu(const ot_tensor *x, const ot_tensor *w, const ot_tensor *bias, int stride_h,
                          int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int mode) {
    /* tensor.rs:1379-1389, 2079-2081 */
    ot_tensor *c = ot_conv2d(x, w, bias, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, mode);
    ot_tensor *r = ot_relu(c);
    ot_free(c);
    return r;
}

ot_tensor *ot_conv2d_direct_3x3(const ot_tensor *x, const ot_tensor *w, const ot_tensor *bias,
                                int stride_h, int stride_w, int pad_h, int pad_w) {
    /* tensor.rs:1287-1376: standard [co][ci][3][3] weights; no tape node */
    size_t n = x->shape[0], c_in = x->shape[1], h_in = x->shape[2], w_in = x->shape[3];
    size_t c_out = w->shape[0];
    size_t h_out = (h_in + 2 * pad_h - 2) / stride_h + 1;
    size_t w_out = (w_in + 2 * pad_w - 2) / stride_w + 1;
    float *o = xalloc_f(n * c_out * h_out * w_out);
    for (size_t idx = 0; idx < n * c_out; ++idx) {
        size_t batch = idx / c_out, oc = idx % c_out;
        float *plane = o + idx * h_out * w_out;
        for (size_t oh = 0; oh < h_out; ++oh)
            for (size_t ow = 0; ow < w_out; ++ow) {
                float sum = 0.0f;
                for (size_t ic = 0; ic < c_in; ++ic) {
                    size_t wb = (oc * c_in + ic) * 9;
                    size_t ib = batch * c_in * h_in * w_in + ic * h_in * w_in;
                    for (size_t kh = 0; kh < 3; ++kh) {
                        size_t ih = oh * stride_h + kh;
                        if (ih < (size_t)pad_h || ih >= h_in + pad_h) continue;
                        for (size_t kw = 0; kw < 3; ++kw) {
                            size_t iw = ow * stride_w + kw;
                            if (iw < (size_t)pad_w || iw >= w_in + pad_w) continue;
                            sum += x->core->data[ib + (ih - pad_h) * w_in + (iw - pad_w)] *
                                   w->core->data[wb + kh * 3 + kw];
                        }
                    }
                }
                plane[oh * w_out + ow] = sum;
            }
        if (bias) {
            float bv = bias->core->data[oc];
            for (size_t s = 0; s < h_out * w_out; ++s) plane[s] += bv;
        }
    }
    size_t shp[4] = {n, c_out, h_out, w_out};
    return ot_wrap(o, shp, 4);
}

ot_tensor *ot_max_pool2d(const ot_tensor *x, int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w,
                         int zero_first, int64_t *argmax_out) {
    /* tensor.rs:1391-1521 */
    OT_CHECK(x->ndim == 4, "Input must be 4D: [N, C, H, W]");
    if (s_h == 0) { s_h = k_h; s_w = k_w; } /* stride.unwrap_or(kernel_size) 1403 */
    size_t n = x->shape[0], c = x->shape[1], h_in = x->shape[2], w_in = x->shape[3];
    size_t h_out = (h_in + 2 * pad_h - k_h) / s_h + 1;
    size_t w_out = (w_in + 2 * pad_w - k_w) / s_w + 1;
    size_t osp = h_out * w_out;
    float *o = xalloc_f(n * c * osp);
    size_t *arg = (size_t *)calloc(n * c * osp > 0 ? n * c * osp : 1, sizeof(size_t));
    OT_PAR_FOR /* tensor.rs:1420 */
    for (size_t bc = 0; bc < n * c; ++bc) {
        size_t in_base = bc * h_in * w_in;
        for (size_t oh = 0; oh < h_out; ++oh)
            for (size_t ow = 0; ow < w_out; ++ow) {
                float best = -INFINITY;
                size_t best_idx = in_base;
                for (size_t kh = 0; kh < (size_t)k_h; ++kh) {
                    size_t ihp = oh * s_h + kh;
                    if (ihp < (size_t)pad_h || ihp >= h_in + pad_h) continue;
                    for (size_t kw = 0; kw < (size_t)k_w; ++kw) {
                        size_t iwp = ow * s_w + kw;
                        if (iwp < (size_t)pad_w || iwp >= w_in + pad_w) continue;
                        size_t idx = in_base + (ihp - pad_h) * w_in + (iwp - pad_w);
                        float v = x->core->data[idx];
                        if (v > best) { best = v; best_idx = idx; }
                    }
                }
                o[bc * osp + oh * w_out + ow] = best;
                arg[bc * osp + oh * w_out + ow] = best_idx;
            }
    }
    if (argmax_out)
        for (size_t i = 0; i < n * c * osp; ++i) argmax_out[i] = (int64_t)arg[i];
    size_t shp[4] = {n, c, h_out, w_out};
    ot_tensor *out = ot_wrap(o, shp, 4);
    if (x->requires_grad) {
        out->requires_grad = 1;
        ot_node *nd = tape_push(N_MAXPOOL, x, NULL, out);
        nd->saved_idx = arg;
        nd->dims[0] = n * c; nd->dims[1] = h_in * w_in; nd->dims[2] = osp;
        nd->iparam[0] = zero_first;
    } else {
        free(arg);
    }
    return out;
}

ot_tensor *ot_avg_pool2d(const ot_tensor *x, int k_h, int k_w, int s_h, int s_w, int pad_h, int pad_w) {
    /* tensor.rs:1524-1660 */
    OT_CHECK(x->ndim == 4, "Input must be 4D: [N, C, H, W]");
    if (s_h == 0) { s_h = k_h; s_w = k_w; }
    size_t n = x->shape[0], c = x->shape[1], h_in = x->shape[2], w_in = x->shape[3];
    size_t h_out = (h_in + 2 * pad_h - k_h) / s_h + 1;
    size_t w_out = (w_in + 2 * pad_w - k_w) / s_w + 1;
    size_t osp = h_out * w_out;
    float pool_size = (float)(k_h * k_w); /* padding counted in the divisor: Q6 */
    float *o = xalloc_f(n * c * osp);
    OT_PAR_FOR /* tensor.rs:1552 */
    for (size_t bc = 0; bc < n * c; ++bc) {
        size_t in_base = bc * h_in * w_in;
        for (size_t oh = 0; oh < h_out; ++oh)
            for (size_t ow = 0; ow < w_out; ++ow) {
                float sum = 0.0f;
                for (size_t kh = 0; kh < (size_t)k_h; ++kh) {
                    size_t ihp = oh * s_h + kh;
                    if (ihp < (size_t)pad_h || ihp >= h_in + pad_h) continue;
                    for (size_t kw = 0; kw < (size_t)k_w; ++kw) {
                        size_t iwp = ow * s_w + kw;
                        if (iwp < (size_t)pad_w || iwp >= w_in + pad_w) continue;
                        sum += x->core->data[in_base + (ihp - pad_h) * w_in + (iwp - pad_w)];
                    }
                }
                o[bc * osp + oh * w_out + ow] = sum / pool_size;
            }
    }
    size_t shp[4] = {n, c, h_out, w_out};
    ot_tensor *out = ot_wrap(o, shp, 4);
    if (x->requires_grad) {
        out->requires_grad = 1;
        ot_node *nd = tape_push(N_AVGPOOL, x, NULL, out);
        nd->dims[0] = n * c; nd->dims[1] = h_in; nd->dims[2] = w_in; nd->dims[3] = h_out; nd->dims[4] = w_out;
        nd->dims[5] = (size_t)k_h; nd->dims[6] = (size_t)k_w; nd->dims[7] = (size_t)s_h; nd->dims[8] = (size_t)s_w;
        nd->dims[9] = (size_t)pad_h; nd->dims[10] = (size_t)pad_w;
        nd->fparam = pool_size;
    }
    return out;
}

ot_tensor *ot_adaptive_avg_pool2d(const ot_tensor *x, int h_out, int w_out) { /* nn.rs:670-686 */
    int kh = (int)(x->shape[2] / (size_t)h_out), kw = (int)(x->shape[3] / (size_t)w_out);
    return ot_avg_pool2d(x, kh, kw, kh, kw, 0, 0);
}

/* ------------------------------------------------------------------ loss */

ot_tensor *ot_log_softmax(const ot_tensor *x) { /* loss.rs:101-126, last dim only */
    int dim = x->ndim - 1;
    ot_tensor *mx = ot_max(x, dim, NULL);
    ot_tensor *shifted = ot_sub_broadcast_rows(x, mx);
    ot_tensor *e = ot_exp(shifted);
    ot_tensor *se = ot_sum(e, dim, 1);
    ot_tensor *ls = ot_log(se);
    ot_tensor *out = ot_sub_broadcast_rows(shifted, ls);
    ot_free(mx); ot_free(shifted); ot_free(e); ot_free(se); ot_free(ls);
    return out;
}

/* loss.rs:82-98 panics for C>1 by inspection (Q12); the build exposes softmax
 * as exp(log_softmax) and keeps the reference tests' PROPERTIES. */
ot_tensor *ot_softmax(const ot_tensor *x) {
    ot_tensor *lp = ot_log_softmax(x);
    ot_tensor *p = ot_exp(lp);
    ot_free(lp);
    return p;
}

ot_tensor *ot_cross_entropy_loss(const ot_tensor *logits, const ot_tensor *targets) { /* loss.rs:136-195 */
    OT_CHECK(targets->ndim == 1 || (targets->ndim == 2 && targets->shape[1] == 1), "Targets must be [B] or [B,1]");
    OT_CHECK(logits->ndim == 2, "Logits must be [B,C]");
    OT_CHECK(logits->shape[0] == targets->shape[0], "Batch sizes must match");
    size_t b = logits->shape[0], c = logits->shape[1];
    ot_tensor *logp = ot_log_softmax(logits);
    float acc = 0.0f;
    for (size_t i = 0; i < b; ++i) {
        size_t cls = (size_t)targets->core->data[i]; /* `as usize` (saturating) */
        OT_CHECK(cls < c, "Target class %zu out of bounds for %zu", cls, c);
        acc -= logp->core->data[i * c + cls];
    }
    ot_tensor *out = ot_scalar(acc / (float)b);
    if (logits->requires_grad) {
        out->requires_grad = 1;
        ot_node *nd = tape_push(N_CROSS_ENTROPY, logits, targets, out);
        /* push_unary_op(logits, ..): gate is logits.requires_grad, true here */
        nd->aux = ot_clone(logp);
    }
    ot_free(logp);
    return out;
}

float ot_accuracy(const ot_tensor *pred, const ot_tensor *targets) { /* loss.rs:271-290 */
    OT_CHECK(pred->shape[0] == targets->shape[0], "Batch sizes must match");
    ot_tensor *cls = ot_argmax(pred, 1);
    size_t n = targets->core->len;
    int correct = 0;
    for (size_t i = 0; i < n; ++i)
        if (fabsf(cls->core->data[i] - targets->core->data[i]) < 1e-6f) correct++;
    ot_free(cls);
    return (float)correct / (float)n;
}

ot_tensor *ot_one_hot(const ot_tensor *idx, int num_classes) { /* loss.rs:248-268 */
    OT_CHECK(idx->ndim == 1, "Indices must be 1D");
    size_t b = idx->shape[0];
    float *o = xalloc_f(b * (size_t)num_classes);
    for (size_t i = 0; i < b; ++i) {
        size_t cls = (size_t)idx->core->data[i];
        OT_CHECK(cls < (size_t)num_classes, "Index %zu out of bounds for %d classes", cls, num_classes);
        o[i * num_classes + cls] = 1.0f;
    }
    size_t shp[2] = {b, (size_t)num_classes};
    return ot_wrap(o, shp, 2);
}

static float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

ot_tensor *ot_bce_loss(const ot_tensor *pred, const ot_tensor *targets) { /* loss.rs:6-73 */
    check_same_len(pred, targets);
    const float eps = 1e-7f;
    size_t n = pred->core->len;
    float acc = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float pi = clampf(pred->core->data[i], eps, 1.0f - eps), yi = targets->core->data[i];
        acc -= yi * logf(pi) + (1.0f - yi) * logf(1.0f - pi);
    }
    ot_tensor *out = ot_scalar(acc / (float)n);
    if (pred->requires_grad || targets->requires_grad) {
        out->requires_grad = 1;
        tape_push(N_BCE, pred, targets, out);
    }
    return out;
}

ot_tensor *ot_mse_loss(const ot_tensor *pred, const ot_tensor *targets) { /* loss.rs:76-80 */
    ot_tensor *d = ot_sub(pred, targets);
    ot_tensor *sq = ot_mul(d, d);
    ot_tensor *m = ot_mean(sq);
    ot_free(d); ot_free(sq);
    return m;
}

ot_tensor *ot_linear_forward(const ot_tensor *x, const ot_tensor *w, const ot_tensor *b) { /* nn.rs:54-60 */
    ot_tensor *wt = ot_transpose(w);
    ot_tensor *out = ot_matmul(x, wt);
    ot_free(wt);
    if (b) {
        ot_tensor *o2 = ot_add_broadcast(out, b);
        ot_free(out);
        out = o2;
    }
    return out;
}

/* -------------------------------------------------------------- backward */

static void node_backward(size_t id) {
    /* copy: closures that record nodes during backward (Q7) may realloc g_nodes */
    ot_node nd = g_nodes[id];
    const float *gout = nd.out->core->grad;
    if (!gout) return; /* every closure: `if let Some(gout) = out.grad...` */
    size_t on = nd.out->core->len;

    switch (nd.kind) {
    case N_ADD: /* ops.rs:38-47 */
        if (nd.a->requires_grad) accumulate_grad(nd.a, gout);
        if (nd.b->requires_grad) accumulate_grad(nd.b, gout);
        break;
    case N_MUL: /* ops.rs:81-116 */
        if (nd.a->requires_grad) {
            float *ga = grad_slot(nd.a, nd.b->core->len);
            for (size_t i = 0; i < nd.a->core->len; ++i) ga[i] = ga[i] + gout[i] * nd.b->core->data[i];
        }
        if (nd.b->requires_grad) {
            float *gb = grad_slot(nd.b, nd.a->core->len);
            for (size_t i = 0; i < nd.b->core->len; ++i) gb[i] = gb[i] + gout[i] * nd.a->core->data[i];
        }
        break;
    case N_SUB: /* ops.rs:403-412 */
        if (nd.a->requires_grad) accumulate_grad(nd.a, gout);
        if (nd.b->requires_grad) accumulate_grad_scaled(nd.b, gout, -1.0f);
        break;
    case N_DIV: /* ops.rs:466-492 */
        if (nd.a->requires_grad) {
            float *ga = grad_slot(nd.a, nd.b->core->len);
            for (size_t i = 0; i < nd.a->core->len; ++i) ga[i] += gout[i] / nd.b->core->data[i];
        }
        if (nd.b->requires_grad) {
            float *gb = grad_slot(nd.b, nd.a->core->len);
            for (size_t i = 0; i < nd.b->core->len; ++i) {
                float bv = nd.b->core->data[i];
                gb[i] -= gout[i] * nd.a->core->data[i] / (bv * bv);
            }
        }
        break;
    case N_MATMUL: { /* ops.rs:238-294 */
        int m = (int)nd.a->shape[0], k = (int)nd.a->shape[1], n = (int)nd.b->shape[1];
        if (nd.a->requires_grad) { /* dA += dC * B^T : sgemm(N,T,m,k,n, beta=1) */
            float *ga = grad_slot(nd.a, (size_t)m * k);
            ot_sgemm_rowmajor(0, 1, m, k, n, 1.0f, gout, nd.b->core->data, 1.0f, ga);
        }
        if (nd.b->requires_grad) { /* dB += A^T * dC : sgemm(T,N,k,n,m, beta=1) */
            float *gb = grad_slot(nd.b, (size_t)k * n);
            ot_sgemm_rowmajor(1, 0, k, n, m, 1.0f, nd.a->core->data, gout, 1.0f, gb);
        }
        break;
    }
    case N_RELU: { /* ops.rs:358-369: mask on the INPUT (Q15) */
        float *gin = grad_slot(nd.a, nd.a->core->len);
        for (size_t i = 0; i < nd.a->core->len; ++i) gin[i] += nd.a->core->data[i] > 0.0f ? gout[i] : 0.0f;
        break;
    }
    case N_TRANSPOSE: { /* tensor.rs:574-587 */
        size_t rows = nd.dims[0], cols = nd.dims[1];
        float *gin = grad_slot(nd.a, rows * cols);
        for (size_t i = 0; i < rows; ++i)
            for (size_t j = 0; j < cols; ++j) gin[i * cols + j] += gout[j * rows + i];
        break;
    }
    case N_SIGMOID: { /* tensor.rs:618-629: from the saved output */
        float *gin = grad_slot(nd.a, on);
        for (size_t i = 0; i < on; ++i) gin[i] += gout[i] * nd.saved[i] * (1.0f - nd.saved[i]);
        break;
    }
    case N_ADD_BCAST: { /* tensor.rs:674-694 */
        size_t bs = nd.dims[0], f = nd.dims[1];
        if (nd.a->requires_grad) accumulate_grad(nd.a, gout);
        if (nd.b->requires_grad) {
            float *gb = grad_slot(nd.b, f);
            for (size_t r = 0; r < bs; ++r)
                for (size_t j = 0; j < f; ++j) gb[j] += gout[r * f + j];
        }
        break;
    }
    case N_SUB_BCAST_ROWS: { /* tensor.rs:745-766 */
        if (nd.a->requires_grad) accumulate_grad(nd.a, gout);
        if (nd.b->requires_grad) {
            size_t b = nd.a->shape[0], c = nd.a->shape[1];
            float *gr = xalloc_f(b);
            for (size_t row = 0; row < b; ++row) {
                float s = 0.0f;
                for (size_t col = 0; col < c; ++col) s += gout[row * c + col];
                gr[row] -= s;
            }
            accumulate_grad(nd.b, gr);
            free(gr);
        }
        break;
    }
    case N_MEAN: { /* tensor.rs:785-796 */
        float g_each = gout[0] / nd.fparam;
        float *gin = grad_slot(nd.a, nd.a->core->len);
        for (size_t i = 0; i < nd.a->core->len; ++i) gin[i] += g_each;
        break;
    }
    case N_RESHAPE: { /* tensor.rs:823-836 */
        float *gin = grad_slot(nd.a, on);
        for (size_t i = 0; i < on; ++i) gin[i] += gout[i];
        break;
    }
    case N_SUM_DIM: { /* tensor.rs:949-991 */
        int d = nd.iparam[0], keepdim = nd.iparam[1];
        const size_t *ish = nd.a->shape;
        int ndim = nd.a->ndim;
        float *gin = grad_slot(nd.a, nd.a->core->len);
        for (size_t i = 0; i < nd.a->core->len; ++i) {
            size_t idx = i, out_idx = 0, multiplier = 1;
            for (int j = ndim - 1; j >= 0; --j) {
                size_t coord = idx % ish[j];
                idx /= ish[j];
                if (j != d) {
                    size_t out_j = (j > d && !keepdim) ? (size_t)(j - 1) : (size_t)j;
                    if (out_j < on) { /* 971: compares against gout.len() */
                        out_idx += coord * multiplier;
                        multiplier *= ish[j]; /* every arm of 973-983 with j != d is in_shape[j] */
                    }
                }
            }
            gin[i] += gout[out_idx < on - 1 ? out_idx : on - 1];
        }
        break;
    }
    case N_SUM_ALL: { /* tensor.rs:1006-1013 */
        float gv = gout[0];
        float *g = grad_slot(nd.a, nd.a->core->len);
        for (size_t i = 0; i < nd.a->core->len; ++i) g[i] = g[i] + gv;
        break;
    }
    case N_EXP: { /* tensor.rs:1109-1129 */
        float *gin = grad_slot(nd.a, on);
        for (size_t i = 0; i < on; ++i) gin[i] = gin[i] + gout[i] * nd.saved[i];
        break;
    }
    case N_LOG: { /* tensor.rs:1151-1165 */
        float *gin = grad_slot(nd.a, on);
        for (size_t i = 0; i < on; ++i) gin[i] += gout[i] / nd.a->core->data[i];
        break;
    }
    case N_POW: { /* tensor.rs:1188-1202 */
        float *gin = grad_slot(nd.a, on);
        for (size_t i = 0; i < on; ++i) gin[i] += gout[i] * nd.fparam * powf(nd.a->core->data[i], nd.fparam - 1.0f);
        break;
    }
    case N_MAXPOOL: { /* tensor.rs:1479-1517 */
        size_t planes = nd.dims[0], isp = nd.dims[1], osp = nd.dims[2];
        float *gin = grad_slot(nd.a, planes * isp);
        OT_PAR_FOR /* tensor.rs:1491 */
        for (size_t bc = 0; bc < planes; ++bc) {
            float *gp = gin + bc * isp;
            if (nd.iparam[0]) /* 1496-1500: zero the plane first (Q5) */
                for (size_t i = 0; i < isp; ++i) gp[i] = 0.0f;
            for (size_t o = 0; o < osp; ++o) gp[nd.saved_idx[bc * osp + o] - bc * isp] += gout[bc * osp + o];
        }
        break;
    }
    case N_AVGPOOL: { /* tensor.rs:1604-1656 */
        size_t planes = nd.dims[0], h_in = nd.dims[1], w_in = nd.dims[2], h_out = nd.dims[3], w_out = nd.dims[4];
        size_t k_h = nd.dims[5], k_w = nd.dims[6], s_h = nd.dims[7], s_w = nd.dims[8], p_h = nd.dims[9], p_w = nd.dims[10];
        float *gin = grad_slot(nd.a, planes * h_in * w_in);
        OT_PAR_FOR /* tensor.rs:1614 */
        for (size_t bc = 0; bc < planes; ++bc) {
            float *gp = gin + bc * h_in * w_in;
            const float *go = gout + bc * h_out * w_out;
            for (size_t oh = 0; oh < h_out; ++oh)
                for (size_t ow = 0; ow < w_out; ++ow) {
                    float gv = go[oh * w_out + ow] / nd.fparam;
                    for (size_t kh = 0; kh < k_h; ++kh) {
                        size_t ihp = oh * s_h + kh;
                        if (ihp < p_h || ihp >= h_in + p_h) continue;
                        for (size_t kw = 0; kw < k_w; ++kw) {
                            size_t iwp = ow * s_w + kw;
                            if (iwp < p_w || iwp >= w_in + p_w) continue;
                            gp[(ihp - p_h) * w_in + (iwp - p_w)] += gv;
                        }
                    }
                }
        }
        break;
    }
    case N_ADD_BIAS_4D: { /* tensor.rs:2003-2027 */
        size_t n = nd.dims[0], c = nd.dims[1], hw = nd.dims[2];
        if (nd.a->requires_grad) accumulate_grad(nd.a, gout);
        if (nd.b->requires_grad) {
            float *gb = grad_slot(nd.b, c);
            for (size_t b = 0; b < n; ++b)
                for (size_t ch = 0; ch < c; ++ch) {
                    size_t base = b * c * hw + ch * hw;
                    for (size_t s = 0; s < hw; ++s) gb[ch] += gout[base + s];
                }
        }
        break;
    }
    case N_CROSS_ENTROPY: { /* loss.rs:174-191 */
        size_t b = nd.a->shape[0], c = nd.a->shape[1];
        ot_tensor *sm = ot_exp(nd.aux); /* records a node DURING backward: Q7 */
        float *grad = xalloc_f(b * c);
        memcpy(grad, sm->core->data, b * c * sizeof(float));
        ot_free(sm);
        /* g_nodes may have been reallocated by the push above; nd is a copy */
        for (size_t i = 0; i < b; ++i) {
            size_t cls = (size_t)nd.b->core->data[i];
            grad[i * c + cls] -= 1.0f;
        }
        float scale = gout[0] / (float)b;
        for (size_t i = 0; i < b * c; ++i) grad[i] *= scale;
        accumulate_grad(nd.a, grad);
        free(grad);
        break;
    }
    case N_BCE: { /* loss.rs:35-68 */
        size_t n = nd.a->core->len;
        float g = gout[0];
        if (nd.a->requires_grad) {
            float *gp = grad_slot(nd.a, n);
            for (size_t i = 0; i < n; ++i) {
                float pi = clampf(nd.a->core->data[i], 1e-7f, 1.0f - 1e-7f), yi = nd.b->core->data[i];
                gp[i] += g * (-(yi / pi - (1.0f - yi) / (1.0f - pi))) / (float)n;
            }
        }
        if (nd.b->requires_grad) {
            float *gy = grad_slot(nd.b, n);
            for (size_t i = 0; i < n; ++i) {
                float pi = clampf(nd.a->core->data[i], 1e-7f, 1.0f - 1e-7f);
                gy[i] += g * (logf(1.0f - pi) - logf(pi)) / (float)n;
            }
        }
        break;
    }
    case N_IM2COL_FULL: { /* full_backward extension: col2im scatter-add */
        size_t n = nd.dims[0], c = nd.dims[1], h_in = nd.dims[2], w_in = nd.dims[3];
        size_t h_out = nd.dims[4], w_out = nd.dims[5], p_h = nd.dims[6], p_w = nd.dims[7], ksz = nd.dims[8];
        float *gin = grad_slot(nd.a, nd.a->core->len);
        if (ksz == 1) {
            for (size_t i = 0; i < nd.a->core->len; ++i) gin[i] += gout[i];
        } else {
            size_t col_size = c * 9;
            for (size_t w_idx = 0; w_idx < n * h_out * w_out; ++w_idx) {
                size_t batch = w_idx / (h_out * w_out), pos = w_idx % (h_out * w_out);
                size_t oh = pos / w_out, ow = pos % w_out;
                for (size_t ch = 0; ch < c; ++ch)
                    for (size_t kr = 0; kr < 3; ++kr) {
                        size_t ih = oh + kr;
                        if (ih < p_h || ih >= h_in + p_h) continue;
                        for (size_t kc = 0; kc < 3; ++kc) {
                            size_t iw = ow + kc;
                            if (iw < p_w || iw >= w_in + p_w) continue;
                            gin[((batch * c + ch) * h_in + (ih - p_h)) * w_in + (iw - p_w)] +=
                                gout[w_idx * col_size + ch * 9 + kr * 3 + kc];
                        }
                    }
            }
        }
        break;
    }
    case N_TRANSPOSE4D_FULL: { /* full_backward extension: inverse permutation */
        size_t d0 = nd.a->shape[0], d1 = nd.a->shape[1], d2 = nd.a->shape[2], d3 = nd.a->shape[3];
        float *gin = grad_slot(nd.a, nd.a->core->len);
        for (size_t i0 = 0; i0 < d0; ++i0)
            for (size_t i1 = 0; i1 < d1; ++i1)
                for (size_t i2 = 0; i2 < d2; ++i2)
                    for (size_t i3 = 0; i3 < d3; ++i3)
                        gin[((i0 * d1 + i1) * d2 + i2) * d3 + i3] += gout[((i0 * d3 + i3) * d1 + i1) * d2 + i2];
        break;
    }
    default:
        OT_CHECK(0, "unknown node kind %d", nd.kind);
    }
}

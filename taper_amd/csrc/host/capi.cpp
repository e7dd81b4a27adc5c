// capi.cpp -- include/taper_host.h: opaque-handle C ABI over the C++ host.
#include <algorithm>
#include <cstring>

#include "../../../include/taper_host.h"
#include "taper.h"

using namespace taper;

struct tp_tensor { Tensor t; };
struct tp_module { std::shared_ptr<Module> m; };
struct tp_optim { std::shared_ptr<Optimizer> o; std::shared_ptr<Adam> adam; };
struct tp_dataset { MNISTDataset d; };
struct tp_loader { std::unique_ptr<DataLoader> l; };
struct tp_trainer { std::unique_ptr<Trainer> t; };
struct tp_comm { std::shared_ptr<Communicator> c; };
struct tp_sched { std::shared_ptr<LRScheduler> s; };

static thread_local std::string g_err;

#define TP_BEGIN try {
#define TP_END                              \
    return 0;                               \
    }                                       \
    catch (const std::exception &e) {       \
        g_err = e.what();                   \
        return 1;                           \
    }                                       \
    catch (...) {                           \
        g_err = "unknown exception";        \
        return 1;                           \
    }

static tp_tensor *wrap(const Tensor &t) { return new tp_tensor{t}; }
static Shape mkshape(const size_t *s, int nd) { return Shape(s, s + nd); }
static const Tensor &opt_t(const tp_tensor *t) {
    static const Tensor none;
    return t ? t->t : none;
}

extern "C" {

const char *tp_last_error(void) { return g_err.c_str(); }

int tp_device_set(int id) { TP_BEGIN Device::set_device(id); TP_END }
int tp_device_sync(void) { TP_BEGIN Device::sync(); TP_END }
int tp_device_shutdown(void) { TP_BEGIN Device::shutdown(); TP_END }
void *tp_device_ctx(void) {
    try { return Device::ctx(); } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
int tp_tape_reset(void) { TP_BEGIN Tape::reset(); TP_END }
int tp_tape_len(size_t *out) { TP_BEGIN *out = Tape::len(); TP_END }
int tp_tape_set_compat_zero_sentinel(int on) { TP_BEGIN Tape::set_compat_zero_sentinel(on != 0); TP_END }
int tp_set_full_backward(int on) { TP_BEGIN set_full_backward(on != 0); TP_END }
int tp_set_conv_chain(int on) { TP_BEGIN set_conv_chain(on != 0); TP_END }
int tp_set_conv_chain_head(int on) { TP_BEGIN set_conv_chain_head(on != 0); TP_END }

int tp_tensor_new(const float *h, const size_t *shape, int nd, tp_tensor **out) {
    TP_BEGIN
    Shape s = mkshape(shape, nd);
    *out = wrap(Tensor(std::vector<float>(h, h + numel(s)), s));
    TP_END
}
int tp_tensor_randn(const size_t *shape, int nd, uint64_t seed, tp_tensor **out) { TP_BEGIN *out = wrap(Tensor::randn(mkshape(shape, nd), seed)); TP_END }
int tp_tensor_clone(const tp_tensor *t, tp_tensor **out) { TP_BEGIN *out = wrap(t->t); TP_END }
int tp_tensor_free(tp_tensor *t) { TP_BEGIN delete t; TP_END }
int tp_tensor_set_requires_grad(tp_tensor *t, int on) { TP_BEGIN t->t.set_requires_grad(on != 0); TP_END }
int tp_tensor_requires_grad(const tp_tensor *t, int *out) { TP_BEGIN *out = t->t.get_requires_grad(); TP_END }
int tp_tensor_ndim(const tp_tensor *t, int *out) { TP_BEGIN *out = (int)t->t.shape().size(); TP_END }
int tp_tensor_shape(const tp_tensor *t, size_t *out4) {
    TP_BEGIN
    for (size_t i = 0; i < t->t.shape().size(); ++i) out4[i] = t->t.shape()[i];
    TP_END
}
int tp_tensor_len(const tp_tensor *t, size_t *out) { TP_BEGIN *out = t->t.len(); TP_END }
int tp_tensor_data(const tp_tensor *t, float *h_out) {
    TP_BEGIN
    auto v = t->t.data();
    std::memcpy(h_out, v.data(), v.size() * sizeof(float));
    TP_END
}
int tp_tensor_set_data(tp_tensor *t, const float *h_in) { TP_BEGIN t->t.set_data(std::vector<float>(h_in, h_in + t->t.len())); TP_END }
int tp_tensor_has_grad(const tp_tensor *t, int *out) { TP_BEGIN *out = t->t.has_grad(); TP_END }
int tp_tensor_grad(const tp_tensor *t, float *h_out) {
    TP_BEGIN
    TAPER_ASSERT(t->t.has_grad(), "grad is None");
    auto v = t->t.grad();
    std::memcpy(h_out, v.data(), v.size() * sizeof(float));
    TP_END
}
int tp_tensor_set_grad(tp_tensor *t, const float *h_in) {
    TP_BEGIN
    if (!h_in) t->t.zero_grad();
    else t->t.set_grad(std::vector<float>(h_in, h_in + t->t.len()));
    TP_END
}
int tp_tensor_tape_node(const tp_tensor *t, size_t *out) { TP_BEGIN *out = t->t.tape_node(); TP_END }
int tp_tensor_dptr(const tp_tensor *t, void **d_out) { TP_BEGIN *d_out = t->t.dptr(); TP_END }
int tp_tensor_backward(tp_tensor *t) { TP_BEGIN t->t.backward(); TP_END }
int tp_tensor_zero_grad(tp_tensor *t) { TP_BEGIN t->t.zero_grad(); TP_END }

#define TP_BIN(name, expr) \
    int name(const tp_tensor *a, const tp_tensor *b, tp_tensor **out) { TP_BEGIN *out = wrap(expr); TP_END }
TP_BIN(tp_add, a->t + b->t)
TP_BIN(tp_sub, a->t - b->t)
TP_BIN(tp_mul, a->t * b->t)
TP_BIN(tp_div, a->t / b->t)
TP_BIN(tp_matmul, a->t.matmul(b->t))
TP_BIN(tp_add_broadcast, a->t.add_broadcast(b->t))
TP_BIN(tp_sub_broadcast_rows, a->t.sub_broadcast_rows(b->t))
#define TP_UN(name, expr) \
    int name(const tp_tensor *x, tp_tensor **out) { TP_BEGIN *out = wrap(expr); TP_END }
TP_UN(tp_relu, x->t.relu())
TP_UN(tp_sigmoid, x->t.sigmoid())
TP_UN(tp_transpose, x->t.transpose())
TP_UN(tp_exp, x->t.exp())
TP_UN(tp_log, x->t.log())
TP_UN(tp_mean, x->t.mean())
TP_UN(tp_log_softmax, log_softmax(x->t))
TP_UN(tp_softmax, softmax(x->t))
int tp_pow(const tp_tensor *x, float e, tp_tensor **out) { TP_BEGIN *out = wrap(x->t.pow(e)); TP_END }
int tp_sum(const tp_tensor *x, int dim, int keepdim, tp_tensor **out) { TP_BEGIN *out = wrap(x->t.sum(dim, keepdim != 0)); TP_END }
int tp_max(const tp_tensor *x, int dim, tp_tensor **values, tp_tensor **indices) {
    TP_BEGIN
    auto r = x->t.max(dim);
    if (values) *values = wrap(r.first);
    if (indices) *indices = wrap(r.second);
    TP_END
}
int tp_reshape(const tp_tensor *x, const size_t *shape, int nd, tp_tensor **out) { TP_BEGIN *out = wrap(x->t.reshape(mkshape(shape, nd))); TP_END }
int tp_flatten(const tp_tensor *x, int sd, tp_tensor **out) { TP_BEGIN *out = wrap(x->t.flatten((size_t)sd)); TP_END }
int tp_squeeze(const tp_tensor *x, int dim, tp_tensor **out) { TP_BEGIN *out = wrap(x->t.squeeze(dim)); TP_END }
int tp_unsqueeze(const tp_tensor *x, int dim, tp_tensor **out) { TP_BEGIN *out = wrap(x->t.unsqueeze((size_t)dim)); TP_END }
int tp_linear(const tp_tensor *x, const tp_tensor *w, const tp_tensor *b, int relu, tp_tensor **out) {
    TP_BEGIN *out = wrap(x->t.linear(w->t, opt_t(b), relu != 0)); TP_END
}
int tp_conv2d(const tp_tensor *x, const tp_tensor *w, const tp_tensor *b, int sh, int sw, int ph, int pw, int dh, int dw, int relu,
              tp_tensor **out) {
    TP_BEGIN *out = wrap(x->t.conv2d(w->t, opt_t(b), {sh, sw}, {ph, pw}, {dh, dw}, relu != 0)); TP_END
}
int tp_max_pool2d(const tp_tensor *x, int kh, int kw, int sh, int sw, int ph, int pw, tp_tensor **out) {
    TP_BEGIN *out = wrap(x->t.max_pool2d({kh, kw}, {sh, sw}, {ph, pw})); TP_END
}
int tp_avg_pool2d(const tp_tensor *x, int kh, int kw, int sh, int sw, int ph, int pw, tp_tensor **out) {
    TP_BEGIN *out = wrap(x->t.avg_pool2d({kh, kw}, {sh, sw}, {ph, pw})); TP_END
}

int tp_cross_entropy_loss(const tp_tensor *lg, const tp_tensor *tg, tp_tensor **out) { TP_BEGIN *out = wrap(cross_entropy_loss(lg->t, tg->t)); TP_END }
int tp_accuracy(const tp_tensor *p, const tp_tensor *tg, float *out) { TP_BEGIN *out = accuracy(p->t, tg->t); TP_END }
int tp_one_hot(const tp_tensor *idx, int nc, tp_tensor **out) { TP_BEGIN *out = wrap(one_hot(idx->t, (size_t)nc)); TP_END }
int tp_mse_loss(const tp_tensor *p, const tp_tensor *tg, tp_tensor **out) { TP_BEGIN *out = wrap(mse_loss(p->t, tg->t)); TP_END }

int tp_bce_loss(const tp_tensor *p, const tp_tensor *tg, tp_tensor **out) { TP_BEGIN *out = wrap(bce_loss(p->t, tg->t)); TP_END }
int tp_cross_entropy_loss_onehot(const tp_tensor *l, const tp_tensor *tg, tp_tensor **out) {
    TP_BEGIN *out = wrap(cross_entropy_loss_onehot(l->t, tg->t)); TP_END
}

static Dropout *as_dropout(const tp_module *m) {
    auto *d = dynamic_cast<Dropout *>(m->m.get());
    TAPER_ASSERT(d, "not a Dropout module");
    return d;
}
int tp_dropout_new(float p, uint64_t seed, tp_module **out) { TP_BEGIN *out = new tp_module{std::make_shared<Dropout>(p, seed)}; TP_END }
int tp_dropout_set_training(tp_module *m, int training) {
    TP_BEGIN
    if (training) as_dropout(m)->train();
    else as_dropout(m)->eval();
    TP_END
}
int tp_dropout_last_mask(const tp_module *m, tp_tensor **out) { TP_BEGIN *out = wrap(as_dropout(m)->last_mask()); TP_END }

int tp_linear_new(int in_f, int out_f, int bias, uint64_t seed, tp_module **out) {
    TP_BEGIN *out = new tp_module{std::make_shared<Linear>((size_t)in_f, (size_t)out_f, bias != 0, seed)}; TP_END
}
int tp_relu_new(tp_module **out) { TP_BEGIN *out = new tp_module{std::make_shared<ReLU>()}; TP_END }
int tp_sigmoid_new(tp_module **out) { TP_BEGIN *out = new tp_module{std::make_shared<Sigmoid>()}; TP_END }
int tp_conv2d_new(int ic, int oc, int kh, int kw, int sh, int sw, int ph, int pw, int bias, int relu, uint64_t seed, tp_module **out) {
    TP_BEGIN
    auto c = std::make_shared<Conv2d>((size_t)ic, (size_t)oc, std::make_pair(kh, kw), std::make_pair(sh, sw), std::make_pair(ph, pw),
                                      bias != 0, seed);
    c->fuse_relu = relu != 0;
    *out = new tp_module{c};
    TP_END
}
int tp_conv2d_grouped_new(int ic, int oc, int kh, int kw, int sh, int sw, int ph, int pw, int groups, int bias, int relu, uint64_t seed,
                          tp_module **out) {
    TP_BEGIN
    TAPER_ASSERT(groups >= 1, "groups must be >= 1");
    auto c = std::make_shared<Conv2d>((size_t)ic, (size_t)oc, std::make_pair(kh, kw), std::make_pair(sh, sw), std::make_pair(ph, pw),
                                      bias != 0, seed, (size_t)groups);
    c->fuse_relu = relu != 0;
    *out = new tp_module{c};
    TP_END
}
int tp_slice_channels(const tp_tensor *x, size_t start, size_t end, tp_tensor **out) {
    TP_BEGIN *out = new tp_tensor{x->t.slice_channels(start, end)}; TP_END
}
int tp_cat(const tp_tensor *const *tensors, int n, size_t dim, tp_tensor **out) {
    TP_BEGIN
    std::vector<Tensor> ts;
    for (int i = 0; i < n; ++i) ts.push_back(tensors[i]->t);
    *out = new tp_tensor{Tensor::cat(ts, dim)};
    TP_END
}
int tp_maxpool2d_new(int kh, int kw, int sh, int sw, int ph, int pw, tp_module **out) {
    TP_BEGIN *out = new tp_module{std::make_shared<MaxPool2d>(std::make_pair(kh, kw), std::make_pair(sh, sw), std::make_pair(ph, pw))}; TP_END
}
int tp_avgpool2d_new(int kh, int kw, int sh, int sw, int ph, int pw, tp_module **out) {
    TP_BEGIN *out = new tp_module{std::make_shared<AvgPool2d>(std::make_pair(kh, kw), std::make_pair(sh, sw), std::make_pair(ph, pw))}; TP_END
}
int tp_adaptive_avgpool2d_new(int oh, int ow, tp_module **out) { TP_BEGIN *out = new tp_module{std::make_shared<AdaptiveAvgPool2d>(std::make_pair(oh, ow))}; TP_END }
int tp_flatten_new(int sd, tp_module **out) { TP_BEGIN *out = new tp_module{std::make_shared<Flatten>((size_t)sd)}; TP_END }
int tp_sequential_new(tp_module *const *layers, int n, int fuse, tp_module **out) {
    TP_BEGIN
    std::vector<std::shared_ptr<Module>> ls;
    for (int i = 0; i < n; ++i) ls.push_back(layers[i]->m);
    auto s = std::make_shared<Sequential>(ls);
    s->fuse = fuse != 0;
    *out = new tp_module{s};
    TP_END
}
int tp_module_free(tp_module *m) { TP_BEGIN delete m; TP_END }
int tp_module_forward(const tp_module *m, const tp_tensor *x, tp_tensor **out) { TP_BEGIN *out = wrap(m->m->forward(x->t)); TP_END }
int tp_module_num_parameters(const tp_module *m, int *out) { TP_BEGIN *out = (int)m->m->parameters().size(); TP_END }
int tp_module_parameter(const tp_module *m, int i, tp_tensor **out) {
    TP_BEGIN
    auto p = m->m->parameters();
    TAPER_ASSERT(i >= 0 && (size_t)i < p.size(), "parameter index out of range");
    *out = wrap(p[i]);
    TP_END
}

static std::vector<Tensor> collect(tp_tensor *const *ps, int n) {
    std::vector<Tensor> v;
    for (int i = 0; i < n; ++i) v.push_back(ps[i]->t);
    return v;
}
int tp_adam_new(tp_tensor *const *ps, int n, float lr, float b1, float b2, float eps, float wd, tp_optim **out) {
    TP_BEGIN
    auto a = std::make_shared<Adam>(collect(ps, n), lr, b1, b2, eps, wd);
    *out = new tp_optim{a, a};
    TP_END
}
int tp_sgd_new(tp_tensor *const *ps, int n, float lr, tp_optim **out) {
    TP_BEGIN *out = new tp_optim{std::make_shared<SGD>(collect(ps, n), lr), nullptr}; TP_END
}
int tp_adamw_new(tp_tensor *const *ps, int n, float lr, float b1, float b2, float eps, float wd, tp_optim **out) {
    TP_BEGIN
    auto w = std::make_shared<AdamW>(collect(ps, n), lr, b1, b2, eps, wd);
    *out = new tp_optim{w, std::shared_ptr<Adam>(w, &w->adam)};
    TP_END
}
int tp_adam_load_state(tp_optim *o, int t, const float *h_m, const float *h_v) {
    TP_BEGIN
    TAPER_ASSERT(o->adam, "not an Adam optimizer");
    size_t n = 0;
    for (const Tensor &p : o->adam->flat().params) n += p.len();
    o->adam->load_state(t, std::vector<float>(h_m, h_m + n), std::vector<float>(h_v, h_v + n));
    TP_END
}
int tp_sched_step_lr(float lr, size_t step, float gamma, tp_sched **out) { TP_BEGIN *out = new tp_sched{std::make_shared<StepLR>(lr, step, gamma)}; TP_END }
int tp_sched_exponential(float lr, float gamma, tp_sched **out) { TP_BEGIN *out = new tp_sched{std::make_shared<ExponentialLR>(lr, gamma)}; TP_END }
int tp_sched_cosine(float lr, size_t t_max, float min_lr, tp_sched **out) {
    TP_BEGIN *out = new tp_sched{std::make_shared<CosineAnnealingLR>(lr, t_max, min_lr)}; TP_END
}
int tp_sched_plateau(float lr, float factor, size_t patience, float min_lr, int mode_max, tp_sched **out) {
    TP_BEGIN
    auto s = std::make_shared<ReduceLROnPlateau>(lr, factor, patience, min_lr, mode_max ? "max" : "min");
    s->verbose = false;
    *out = new tp_sched{s};
    TP_END
}
int tp_sched_step(tp_sched *s, const float *metric) { TP_BEGIN s->s->step(metric); TP_END }
int tp_sched_get_lr(const tp_sched *s, float *out) { TP_BEGIN *out = s->s->get_lr(); TP_END }
int tp_sched_free(tp_sched *s) { TP_BEGIN delete s; TP_END }

int tp_optim_free(tp_optim *o) { TP_BEGIN delete o; TP_END }
int tp_optim_step(tp_optim *o) { TP_BEGIN o->o->step(); TP_END }
int tp_optim_zero_grad(tp_optim *o) { TP_BEGIN o->o->zero_grad(); TP_END }
int tp_adam_set_lr(tp_optim *o, float lr) { TP_BEGIN TAPER_ASSERT(o->adam, "not an Adam optimizer"); o->adam->set_lr(lr); TP_END }
int tp_adam_get_lr(const tp_optim *o, float *out) { TP_BEGIN TAPER_ASSERT(o->adam, "not an Adam optimizer"); *out = o->adam->get_lr(); TP_END }
int tp_adam_t(const tp_optim *o, int *out) { TP_BEGIN TAPER_ASSERT(o->adam, "not an Adam optimizer"); *out = o->adam->t(); TP_END }
int tp_adam_moments(const tp_optim *o, float *h_m, float *h_v) {
    TP_BEGIN
    TAPER_ASSERT(o->adam, "not an Adam optimizer");
    auto m = o->adam->m(), v = o->adam->v();
    if (h_m) std::memcpy(h_m, m.data(), m.size() * sizeof(float));
    if (h_v) std::memcpy(h_v, v.data(), v.size() * sizeof(float));
    TP_END
}
int tp_optim_total(const tp_optim *o, int64_t *out) { TP_BEGIN *out = o->o->flat().total; TP_END }

int tp_dataset_from_host(const float *im, const float *lb, size_t n, int train, tp_dataset **out) {
    TP_BEGIN
    *out = new tp_dataset{MNISTDataset::from_host(std::vector<float>(im, im + n * 784), std::vector<float>(lb, lb + n), train != 0)};
    TP_END
}
int tp_dataset_from_idx(const char *ip, const char *lp, int train, tp_dataset **out) {
    TP_BEGIN *out = new tp_dataset{MNISTDataset::from_idx_files(ip, lp, train != 0)}; TP_END
}
int tp_dataset_synthetic(size_t n, uint64_t seed, int train, tp_dataset **out) {
    TP_BEGIN *out = new tp_dataset{MNISTDataset::synthetic(n, seed, train != 0)}; TP_END
}
int tp_dataset_len(const tp_dataset *d, size_t *out) { TP_BEGIN *out = d->d.len(); TP_END }
int tp_dataset_tensors(const tp_dataset *d, tp_tensor **im, tp_tensor **lb) {
    TP_BEGIN
    if (im) *im = wrap(d->d.images);
    if (lb) *lb = wrap(d->d.labels);
    TP_END
}
int tp_dataset_free(tp_dataset *d) { TP_BEGIN delete d; TP_END }
int tp_loader_new(const tp_dataset *d, size_t bs, int shuffle, uint64_t seed, tp_loader **out) {
    TP_BEGIN *out = new tp_loader{std::make_unique<DataLoader>(d->d, bs, shuffle != 0, seed)}; TP_END
}
int tp_loader_reset(tp_loader *l) { TP_BEGIN l->l->reset(); TP_END }
int tp_loader_num_batches(const tp_loader *l, size_t *out) { TP_BEGIN *out = l->l->num_batches(); TP_END }
int tp_loader_next(tp_loader *l, tp_tensor **im, tp_tensor **lb, int *has) {
    TP_BEGIN
    Tensor a, b;
    *has = l->l->next(&a, &b) ? 1 : 0;
    if (*has) {
        *im = wrap(a);
        *lb = wrap(b);
    }
    TP_END
}
int tp_loader_free(tp_loader *l) { TP_BEGIN delete l; TP_END }

int tp_comm_unique_id(uint8_t out_id[128]) {
    TP_BEGIN
    auto id = Communicator::unique_id();
    std::memcpy(out_id, id.data(), 128);
    TP_END
}
int tp_comm_new(int n, int r, const uint8_t id[128], tp_comm **out) {
    TP_BEGIN *out = new tp_comm{std::make_shared<Communicator>(n, r, std::vector<uint8_t>(id, id + 128))}; TP_END
}
int tp_comm_new_p2p(int n, int r, tp_comm **out) { TP_BEGIN *out = new tp_comm{Communicator::p2p(n, r)}; TP_END }
int tp_comm_new_loopback(tp_comm **out) { TP_BEGIN *out = new tp_comm{Communicator::loopback()}; TP_END }
int tp_comm_set_inkernel(tp_comm *c, int on) { TP_BEGIN c->c->inkernel = on != 0; TP_END }
int tp_comm_inkernel_launches(tp_comm *c, int64_t *out) { TP_BEGIN *out = c->c->inkernel_launches(); TP_END }
int tp_comm_exchange_selftest(tp_comm *c, int slots, int rounds, int *out_bad) { TP_BEGIN *out_bad = c->c->exchange_selftest(slots, rounds); TP_END }
int tp_comm_ranks_on_this_device(tp_comm *c, int *out) { TP_BEGIN *out = c->c->ranks_on_this_device(); TP_END }
int tp_comm_exchange_form(tp_comm *c, int *out) { TP_BEGIN *out = c->c->exchange_form(); TP_END }
int tp_comm_tail_exchange_ok(tp_comm *c, int batch, int in_features, int hidden, int classes, int *out) {
    TP_BEGIN *out = c->c->tail_exchange_ok(batch, in_features, hidden, classes) ? 1 : 0; TP_END
}
int tp_comm_export_arena(tp_comm *c, tp_optim *o, uint8_t out_blob[192]) {
    TP_BEGIN
    auto b = c->c->export_arena(*o->o);
    std::memcpy(out_blob, b.data(), b.size());
    TP_END
}
int tp_comm_connect(tp_comm *c, const uint8_t *blobs, size_t n_bytes) { TP_BEGIN c->c->connect(std::vector<uint8_t>(blobs, blobs + n_bytes)); TP_END }
int tp_comm_stats(tp_comm *c, int64_t out2[2]) { TP_BEGIN th_check(th_comm_stats(c->c->handle(), out2), "th_comm_stats"); TP_END }
int tp_comm_export_arena_ex(tp_comm *c, tp_optim *o, int fine_grained, uint8_t out_blob[192]) {
    TP_BEGIN
    auto b = c->c->export_arena(*o->o, fine_grained != 0);
    std::memcpy(out_blob, b.data(), b.size());
    TP_END
}
int tp_comm_self_check(tp_comm *c, tp_optim *o, int *ok) { TP_BEGIN *ok = c->c->self_check(*o->o) ? 1 : 0; TP_END }
int tp_comm_self_check_rounds(tp_comm *c, tp_optim *o, int rounds, int *ok) { TP_BEGIN *ok = c->c->self_check(*o->o, rounds) ? 1 : 0; TP_END }
int tp_comm_timed_out(tp_comm *c, int *out) { TP_BEGIN *out = c->c->timed_out() ? 1 : 0; TP_END }
int tp_comm_failed(tp_comm *c, int *out) { TP_BEGIN *out = c->c->failed() ? 1 : 0; TP_END }
int tp_comm_set_timeout_ms(tp_comm *c, int64_t ms) { TP_BEGIN c->c->set_timeout_ms(ms); TP_END }
int tp_comm_set_fuse_adam(tp_comm *c, int on) { TP_BEGIN c->c->fuse_adam = on != 0; TP_END }
int tp_comm_free(tp_comm *c) { TP_BEGIN delete c; TP_END }
int tp_comm_allreduce_mean(tp_comm *c, void *d_buf, size_t n) { TP_BEGIN c->c->allreduce_mean((float *)d_buf, n); TP_END }
int tp_comm_count(tp_comm *c, int *out_ranks) { TP_BEGIN th_check(th_comm_count(c->c->handle(), out_ranks), "th_comm_count"); TP_END }
int tp_comm_time_exchange(tp_comm *c, tp_optim *o, int reps, float *us) {
    TP_BEGIN
    auto *adam = dynamic_cast<Adam *>(o->o.get());
    TAPER_ASSERT(adam, "tp_comm_time_exchange: needs an Adam optimizer");
    *us = c->c->time_exchange(*adam, reps);
    TP_END
}

int tp_trainer_new(tp_module *m, tp_optim *o, tp_trainer **out) {
    TP_BEGIN
    TAPER_ASSERT(o->adam, "Trainer takes an Adam optimizer (src/train.rs:76)");
    *out = new tp_trainer{std::make_unique<Trainer>(m->m, o->adam)};
    TP_END
}
int tp_trainer_set_sample_shape(tp_trainer *t, const size_t *shape, int nd) { TP_BEGIN t->t->sample_shape = mkshape(shape, nd); TP_END }
int tp_trainer_set_comm(tp_trainer *t, tp_comm *c) { TP_BEGIN t->t->comm = c ? c->c : nullptr; TP_END }
int tp_trainer_set_options(tp_trainer *t, int graph_chunk, int fuse_head, int fuse_adam) {
    TP_BEGIN
    TAPER_ASSERT(graph_chunk >= 1, "graph_chunk must be >= 1");
    t->t->graph_chunk = (size_t)graph_chunk;
    t->t->fuse_head = fuse_head < 0 ? 0 : (fuse_head > 2 ? 2 : fuse_head);
    t->t->fuse_adam = fuse_adam != 0;
    TP_END
}
int tp_trainer_free(tp_trainer *t) { TP_BEGIN delete t; TP_END }
int tp_trainer_train_step(tp_trainer *t, const tp_tensor *im, const tp_tensor *lb, float *loss, float *acc) {
    TP_BEGIN t->t->train_step(im->t, lb->t, loss, acc); TP_END
}
int tp_trainer_run_epoch(tp_trainer *t, tp_loader *l, int mode, size_t max_steps, float *avg_loss, float *accuracy_out,
                         size_t *total_correct, size_t *total_samples, size_t *num_batches, float *per_step, size_t cap) {
    TP_BEGIN
    EpochResult r;
    if (mode == 0) r = t->t->train_epoch(*l->l);
    else if (mode == 1) r = t->t->train_epoch_graph(*l->l, max_steps);
    else r = t->t->evaluate(*l->l);
    if (avg_loss) *avg_loss = r.avg_loss;
    if (accuracy_out) *accuracy_out = r.accuracy;
    if (total_correct) *total_correct = r.total_correct;
    if (total_samples) *total_samples = r.total_samples;
    if (num_batches) *num_batches = r.num_batches;
    if (per_step)
        for (size_t i = 0; i < r.losses.size() && 2 * i + 1 < cap; ++i) {
            per_step[2 * i] = r.losses[i];
            per_step[2 * i + 1] = r.ncorrect[i];
        }
    TP_END
}


static int copy_text(const std::string &s, char *buf, size_t cap) {
    if (!buf || cap == 0) return 0;
    const size_t n = std::min(s.size(), cap - 1);
    std::memcpy(buf, s.data(), n);
    buf[n] = 0;
    return 0;
}
int tp_trainer_set_scheduler(tp_trainer *t, tp_sched *s) { TP_BEGIN t->t->scheduler = s ? s->s : nullptr; TP_END }
int tp_trainer_fit(tp_trainer *t, tp_loader *train, tp_loader *val, size_t epochs, int verbose, int graph) {
    TP_BEGIN t->t->fit(*train->l, *val->l, epochs, verbose != 0, graph != 0); TP_END
}
int tp_trainer_metrics(const tp_trainer *t, int which, float *h_out, size_t cap, size_t *n_out) {
    TP_BEGIN
    const Metrics &m = t->t->metrics;
    const std::vector<float> *v[5] = {&m.train_loss, &m.train_acc, &m.val_loss, &m.val_acc, &m.epoch_times};
    TAPER_ASSERT(which >= 0 && which < 5, "tp_trainer_metrics: which must be 0..4");
    if (n_out) *n_out = v[which]->size();
    if (h_out) std::memcpy(h_out, v[which]->data(), std::min(cap, v[which]->size()) * sizeof(float));
    TP_END
}
int tp_trainer_metrics_text(const tp_trainer *t, int which, char *buf, size_t cap) {
    TP_BEGIN copy_text(which == 0 ? t->t->metrics.last_line() : t->t->metrics.summary(), buf, cap); TP_END
}
int tp_trainer_save_checkpoint(const tp_trainer *t, const char *path) { TP_BEGIN t->t->save_checkpoint(path); TP_END }
int tp_trainer_load_checkpoint(tp_trainer *t, const char *path) { TP_BEGIN t->t->load_checkpoint(path); TP_END }
int tp_trainer_save_optimizer_state(const tp_trainer *t, const char *path) { TP_BEGIN t->t->save_optimizer_state(path); TP_END }
int tp_trainer_load_optimizer_state(tp_trainer *t, const char *path) { TP_BEGIN t->t->load_optimizer_state(path); TP_END }
int tp_format_f32(float v, char *buf, size_t cap) { TP_BEGIN copy_text(format_f32_display(v), buf, cap); TP_END }

}  // extern "C"

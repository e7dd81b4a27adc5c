// tensor.cpp -- Device, Buffer, Tensor, Tape: host owner of the op-record list
// (mirrors src/tensor.rs, src/ops.rs, src/tape.rs).  Every op = shape checks
// on the host + one call through the C ABI + (optionally) one recorded
// closure.  Nothing here touches the device except via th_*.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "taper.h"

#include <array>

namespace taper {

void th_check(int rc, const char *what) {
    if (rc != 0) throw Error(std::string(what) + ": " + th_last_error());
}
#define TH(call) th_check((call), #call)

size_t numel(const Shape &s) {
    size_t n = 1;
    for (size_t d : s) n *= d;
    return n;
}

// ---------------------------------------------------------------- Device
namespace {
thread_local th_ctx *t_ctx = nullptr;
thread_local int t_device = -1;
thread_local float *t_ones = nullptr;

int default_device() {
    if (t_device >= 0) return t_device;
    if (const char *e = std::getenv("TAPER_DEVICE")) return std::atoi(e);
    if (const char *e = std::getenv("LOCAL_RANK")) return std::atoi(e);
    return 0;
}
}  // namespace

th_ctx *Device::ctx() {
    if (!t_ctx) {
        int n = 0;
        TH(th_device_count(&n));
        TAPER_ASSERT(n > 0, "taper: no MI355X visible -- the HIP backend has no CPU fallback");
        int id = default_device();
        if (id >= n) id = id % n;
        TH(th_ctx_create(id, &t_ctx));
        t_device = id;
    }
    return t_ctx;
}

void Device::set_device(int id) {
    TAPER_ASSERT(!t_ctx || id == t_device, "Device::set_device after the context was created");
    t_device = id;
}
int Device::device_id() { return t_ctx ? t_device : default_device(); }
void Device::sync() { TH(th_ctx_sync(ctx())); }
float *Device::ones1() {
    if (!t_ones) {
        void *p = nullptr;
        TH(th_malloc(ctx(), 256, &p));
        t_ones = (float *)p;
        TH(th_fill_f32(ctx(), t_ones, 1.0f, 4));
    }
    return t_ones;
}

void Device::shutdown() {
    if (t_ctx) {
        Tape::reset();
        t_ones = nullptr;
        th_ctx_destroy(t_ctx);
        t_ctx = nullptr;
    }
}

// ---------------------------------------------------------------- Buffer
Buffer::~Buffer() {
    if (owned && d && t_ctx) {
        if (host_pinned) th_host_free(t_ctx, d);
        else if (fine_grained) th_free_finegrained(t_ctx, d);
        else th_free(t_ctx, d);
    }
}

std::shared_ptr<Buffer> Buffer::alloc_host(size_t n) {
    auto b = std::make_shared<Buffer>();
    void *p = nullptr;
    TH(th_host_malloc(Device::ctx(), std::max<size_t>(n, 1) * sizeof(float), &p));
    b->d = (float *)p;
    b->n = n;
    b->host_pinned = true;
    return b;
}

std::shared_ptr<Buffer> Buffer::alloc_finegrained(size_t n) {
    auto b = std::make_shared<Buffer>();
    void *p = nullptr;
    TH(th_malloc_finegrained(Device::ctx(), std::max<size_t>(n, 1) * sizeof(float), &p));
    b->d = (float *)p;
    b->n = n;
    b->fine_grained = true;
    return b;
}

std::shared_ptr<Buffer> Buffer::alloc(size_t n) {
    auto b = std::make_shared<Buffer>();
    void *p = nullptr;
    TH(th_malloc(Device::ctx(), std::max<size_t>(n, 1) * sizeof(float), &p));
    b->d = (float *)p;
    b->n = n;
    return b;
}

std::shared_ptr<Buffer> Buffer::view(const std::shared_ptr<Buffer> &parent, size_t offset, size_t n) {
    auto b = std::make_shared<Buffer>();
    b->d = parent->d + offset;
    b->n = n;
    b->owned = false;
    b->parent = parent;
    return b;
}

std::shared_ptr<Buffer> Buffer::borrow(float *d, size_t n) {
    auto b = std::make_shared<Buffer>();
    b->d = d;
    b->n = n;
    b->owned = false;
    return b;
}

// ---------------------------------------------------------------- Tape
namespace {
struct TapeState {
    std::vector<std::function<void()>> nodes;
    bool zero_sentinel = false;
};
thread_local TapeState t_tape;
thread_local bool t_full_backward = false;
}  // namespace

void Tape::reset() { t_tape.nodes.clear(); }
size_t Tape::len() { return t_tape.nodes.size(); }
void Tape::set_compat_zero_sentinel(bool on) { t_tape.zero_sentinel = on; }
bool Tape::compat_zero_sentinel() { return t_tape.zero_sentinel; }

void Tape::push(const Tensor &out, bool any_input_requires_grad, std::function<void()> fn) {
    if (!any_input_requires_grad) return;  // tape.rs:55-57, 82-84
    const size_t index = t_tape.nodes.size();
    t_tape.nodes.push_back(std::move(fn));
    *out.tape_node_ = t_tape.zero_sentinel ? index : index + 1;  // tape.rs:65-75 vs Q1
}

void Tape::backward(size_t final_node_id) {  // tape.rs:106-127
    if (t_tape.nodes.empty()) return;
    size_t last;
    if (t_tape.zero_sentinel) {
        last = std::min(final_node_id, t_tape.nodes.size() - 1);
    } else {
        if (final_node_id == 0) return;
        last = std::min(final_node_id - 1, t_tape.nodes.size() - 1);
    }
    for (size_t i = last + 1; i-- > 0;) {
        std::function<void()> fn = t_tape.nodes[i];  // clone: closures may record nodes (Q7)
        fn();
    }
}

void set_full_backward(bool on) { t_full_backward = on; }
bool full_backward() { return t_full_backward; }

// ---------------------------------------------------------------- Tensor basics
static Tensor make(const std::shared_ptr<Buffer> &buf, const Shape &shape) {
    TAPER_ASSERT(shape.size() >= 1 && shape.size() <= 4, "Tensor: 1..4 dimensions supported");
    Tensor t;
    t.data_ = buf;
    t.shape_ = shape;
    t.grad_ = std::make_shared<GradSlot>();
    t.tape_node_ = std::make_shared<size_t>(0);
    return t;
}

Tensor Tensor::empty(const Shape &shape) { return make(Buffer::alloc(numel(shape)), shape); }

Tensor Tensor::zeros(const Shape &shape) {
    Tensor t = empty(shape);
    TH(th_fill_f32(Device::ctx(), t.dptr(), 0.0f, t.len()));
    return t;
}

Tensor::Tensor(const std::vector<float> &data, const Shape &shape) {
    TAPER_ASSERT(data.size() == numel(shape), "Tensor::new: data length does not match shape");
    *this = empty(shape);
    TH(th_memcpy_h2d(Device::ctx(), dptr(), data.data(), data.size() * sizeof(float)));
}

Tensor Tensor::scalar(float v) { return Tensor(std::vector<float>{v}, Shape{1}); }
Tensor Tensor::from_device(float *d, const Shape &shape) { return make(Buffer::borrow(d, numel(shape)), shape); }

Tensor Tensor::randn(const Shape &shape, uint64_t seed) {
    std::mt19937_64 rng(seed);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> v(numel(shape));
    for (auto &x : v) x = nd(rng);
    return Tensor(v, shape);
}

Tensor Tensor::requires_grad() const {
    Tensor t = *this;
    t.requires_grad_ = true;
    return t;
}

std::vector<float> Tensor::data() const {
    std::vector<float> v(len());
    TH(th_memcpy_d2h(Device::ctx(), v.data(), dptr(), v.size() * sizeof(float)));
    return v;
}

void Tensor::set_data(const std::vector<float> &v) {
    TAPER_ASSERT(v.size() == len(), "set_data: length mismatch");
    TH(th_memcpy_h2d(Device::ctx(), dptr(), v.data(), v.size() * sizeof(float)));
}

std::vector<float> Tensor::grad() const {
    if (!has_grad()) return {};
    std::vector<float> v(len());
    TH(th_memcpy_d2h(Device::ctx(), v.data(), grad_->buf->d, v.size() * sizeof(float)));
    return v;
}

void Tensor::set_grad(const std::vector<float> &g) {
    if (g.empty()) {
        zero_grad();
        return;
    }
    TAPER_ASSERT(g.size() == len(), "set_grad: length mismatch");
    bool none;
    float *p = grad_for_write(&none);
    TH(th_memcpy_h2d(Device::ctx(), p, g.data(), g.size() * sizeof(float)));
}

float *Tensor::grad_for_write(bool *was_none) const {
    if (grad_->shared_const) {  // never write through the shared [1.0]: give the slot private storage first
        auto priv = Buffer::alloc(len());
        if (grad_->has) TH(th_memcpy_d2d(Device::ctx(), priv->d, grad_->buf->d, len() * sizeof(float)));
        grad_->buf = priv;
        grad_->shared_const = false;
    }
    if (!grad_->buf) grad_->buf = Buffer::alloc(len());
    if (was_none) *was_none = !grad_->has;
    grad_->has = true;
    grad_->known_zero = false;
    grad_->premasked = false;   // whoever writes next does not know the mask: the producer's node applies it (to 0/1-masked terms: idempotent)
    grad_->plane_sums.reset();
    grad_->dz_colpart.reset();
    return grad_->buf->d;
}

float *Tensor::grad_accum_ptr() const {  // `if slot.is_none() { zeros }` (ops.rs:126-129)
    bool none;
    float *p = grad_for_write(&none);
    if (none) TH(th_fill_f32(Device::ctx(), p, 0.0f, len()));
    return p;
}

void Tensor::backward() const {  // tensor.rs:520-529
    if (len() == 1 && !grad_->buf_is_arena) {
        // scalar root (a loss): grad = [1.0] is the ctx-wide constant, no fill launch
        grad_->buf = Buffer::borrow(Device::ones1(), 1);
        grad_->has = true;
        grad_->known_zero = false;
        grad_->shared_const = true;
    } else {
        bool none;
        float *g = grad_for_write(&none);
        TH(th_fill_f32(Device::ctx(), g, 1.0f, len()));
    }
    if (*tape_node_ != 0) Tape::backward(*tape_node_);  // tensor.rs:526: id 0 = "no node" (in both id schemes)
}

void Tensor::zero_grad() const {  // tensor.rs:531-533
    grad_->has = false;
    if (grad_->shared_const) {
        grad_->buf.reset();
        grad_->shared_const = false;
    }
}

// t.grad (+)= alpha * src  -- accumulate_grad / accumulate_grad_scaled (ops.rs:124-151)
static void accumulate_into(const Tensor &t, const float *src, float alpha = 1.0f) {
    th_ctx *c = Device::ctx();
    bool none;
    float *g = t.grad_for_write(&none);
    if (none && alpha == 1.0f) {
        TH(th_memcpy_d2d(c, g, src, t.len() * sizeof(float)));  // 0 + x
        return;
    }
    if (none) TH(th_fill_f32(c, g, 0.0f, t.len()));
    TH(th_axpy(c, alpha, src, g, t.len()));
}

static void check_same_len(const Tensor &a, const Tensor &b) {
    TAPER_ASSERT(a.len() == b.len(), "Tensor dimensions must match");  // ops.rs:11-15 (count only, Q12)
}

// ---------------------------------------------------------------- element-wise
Tensor Tensor::operator+(const Tensor &o) const {
    check_same_len(*this, o);
    Tensor out = empty(shape_);
    TH(th_add(Device::ctx(), dptr(), o.dptr(), out.dptr(), len()));
    if (requires_grad_ || o.requires_grad_) {
        out.requires_grad_ = true;
        Tensor a = *this, b = o, r = out;
        Tape::push(out, true, [a, b, r]() {
            if (!r.has_grad()) return;
            if (a.get_requires_grad()) accumulate_into(a, r.grad_dptr());
            if (b.get_requires_grad()) accumulate_into(b, r.grad_dptr());
        });
    }
    return out;
}

Tensor Tensor::operator-(const Tensor &o) const {
    check_same_len(*this, o);
    Tensor out = empty(shape_);
    TH(th_sub(Device::ctx(), dptr(), o.dptr(), out.dptr(), len()));
    if (requires_grad_ || o.requires_grad_) {
        out.requires_grad_ = true;
        Tensor a = *this, b = o, r = out;
        Tape::push(out, true, [a, b, r]() {
            if (!r.has_grad()) return;
            if (a.get_requires_grad()) accumulate_into(a, r.grad_dptr());
            if (b.get_requires_grad()) accumulate_into(b, r.grad_dptr(), -1.0f);
        });
    }
    return out;
}

Tensor Tensor::operator*(const Tensor &o) const {
    check_same_len(*this, o);
    Tensor out = empty(shape_);
    TH(th_mul(Device::ctx(), dptr(), o.dptr(), out.dptr(), len()));
    if (requires_grad_ || o.requires_grad_) {
        out.requires_grad_ = true;
        Tensor a = *this, b = o, r = out;
        Tape::push(out, true, [a, b, r]() {
            if (!r.has_grad()) return;
            th_ctx *c = Device::ctx();
            if (a.get_requires_grad()) TH(th_mul_bwd(c, r.grad_dptr(), b.dptr(), a.grad_accum_ptr(), a.len()));
            if (b.get_requires_grad()) TH(th_mul_bwd(c, r.grad_dptr(), a.dptr(), b.grad_accum_ptr(), b.len()));
        });
    }
    return out;
}

Tensor Tensor::operator/(const Tensor &o) const {
    check_same_len(*this, o);
    Tensor out = empty(shape_);
    TH(th_div(Device::ctx(), dptr(), o.dptr(), out.dptr(), len()));
    if (requires_grad_ || o.requires_grad_) {
        out.requires_grad_ = true;
        Tensor a = *this, b = o, r = out;
        Tape::push(out, true, [a, b, r]() {
            if (!r.has_grad()) return;
            float *ga = a.get_requires_grad() ? a.grad_accum_ptr() : nullptr;
            float *gb = b.get_requires_grad() ? b.grad_accum_ptr() : nullptr;
            TH(th_div_bwd(Device::ctx(), r.grad_dptr(), a.dptr(), b.dptr(), ga, gb, a.len()));
        });
    }
    return out;
}

template <typename Fwd, typename Bwd>
static Tensor unary_op(const Tensor &x, Fwd fwd, Bwd bwd) {
    Tensor out = Tensor::empty(x.shape());
    fwd(x, out);
    if (x.get_requires_grad()) {
        out.set_requires_grad(true);
        Tensor in = x, r = out;
        Tape::push(out, true, [in, r, bwd]() {
            if (!r.has_grad()) return;
            bwd(in, r);
        });
    }
    return out;
}

Tensor Tensor::relu() const {
    return unary_op(
        *this, [](const Tensor &x, Tensor &y) { TH(th_relu_fwd(Device::ctx(), x.dptr(), y.dptr(), x.len())); },
        [](const Tensor &x, const Tensor &y) {
            bool none;
            float *gin = x.grad_for_write(&none);
            TH(th_relu_bwd(Device::ctx(), x.dptr(), y.grad_dptr(), gin, x.len(), none ? 0 : 1));  // mask on the INPUT (Q15)
        });
}

Tensor Tensor::sigmoid() const {
    return unary_op(
        *this, [](const Tensor &x, Tensor &y) { TH(th_sigmoid_fwd(Device::ctx(), x.dptr(), y.dptr(), x.len())); },
        [](const Tensor &x, const Tensor &y) {
            TH(th_sigmoid_bwd(Device::ctx(), y.dptr(), y.grad_dptr(), x.grad_accum_ptr(), x.len()));  // saved output
        });
}

Tensor Tensor::exp() const {
    return unary_op(
        *this, [](const Tensor &x, Tensor &y) { TH(th_exp_fwd(Device::ctx(), x.dptr(), y.dptr(), x.len())); },
        [](const Tensor &x, const Tensor &y) {
            TH(th_mul_bwd(Device::ctx(), y.grad_dptr(), y.dptr(), x.grad_accum_ptr(), x.len()));  // gout * e^x
        });
}

Tensor Tensor::log() const {
    return unary_op(
        *this, [](const Tensor &x, Tensor &y) { TH(th_log_fwd(Device::ctx(), x.dptr(), y.dptr(), x.len())); },
        [](const Tensor &x, const Tensor &y) {
            TH(th_log_bwd(Device::ctx(), x.dptr(), y.grad_dptr(), x.grad_accum_ptr(), x.len()));
        });
}

Tensor Tensor::pow(float e) const {
    return unary_op(
        *this, [e](const Tensor &x, Tensor &y) { TH(th_pow_fwd(Device::ctx(), x.dptr(), e, y.dptr(), x.len())); },
        [e](const Tensor &x, const Tensor &y) {
            TH(th_pow_bwd(Device::ctx(), x.dptr(), e, y.grad_dptr(), x.grad_accum_ptr(), x.len()));
        });
}

// ---------------------------------------------------------------- matmul / transpose / linear
Tensor Tensor::matmul(const Tensor &o) const {  // ops.rs:200-298
    TAPER_ASSERT(shape_.size() == 2, "First tensor must be 2D");
    TAPER_ASSERT(o.shape_.size() == 2, "Second tensor must be 2D");
    const int m = (int)shape_[0], k = (int)shape_[1], k2 = (int)o.shape_[0], n = (int)o.shape_[1];
    TAPER_ASSERT(k == k2, "Inner dimensions must match: " + std::to_string(k) + " vs " + std::to_string(k2));
    Tensor out = empty({(size_t)m, (size_t)n});
    TH(th_sgemm(Device::ctx(), 0, 0, m, n, k, 1.0f, dptr(), o.dptr(), 0.0f, out.dptr()));
    if (requires_grad_ || o.requires_grad_) {
        out.requires_grad_ = true;
        Tensor a = *this, b = o, r = out;
        Tape::push(out, true, [a, b, r, m, n, k]() {
            if (!r.has_grad()) return;
            th_ctx *c = Device::ctx();
            bool none;
            if (a.get_requires_grad()) {  // dA += dC * B^T  (ops.rs:254-265)
                float *ga = a.grad_for_write(&none);
                TH(th_sgemm(c, 0, 1, m, k, n, 1.0f, r.grad_dptr(), b.dptr(), none ? 0.0f : 1.0f, ga));
            }
            if (b.get_requires_grad()) {  // dB += A^T * dC  (ops.rs:280-291)
                float *gb = b.grad_for_write(&none);
                TH(th_sgemm(c, 1, 0, k, n, m, 1.0f, a.dptr(), r.grad_dptr(), none ? 0.0f : 1.0f, gb));
            }
        });
    }
    return out;
}

Tensor Tensor::transpose() const {  // tensor.rs:544-591
    TAPER_ASSERT(shape_.size() == 2, "Can only transpose 2D tensors");
    const int rows = (int)shape_[0], cols = (int)shape_[1];
    Tensor out = empty({(size_t)cols, (size_t)rows});
    TH(th_transpose2d(Device::ctx(), dptr(), out.dptr(), rows, cols));
    if (requires_grad_) {
        out.requires_grad_ = true;
        Tensor in = *this, r = out;
        Tape::push(out, true, [in, r, rows, cols]() {
            if (!r.has_grad()) return;
            bool none;
            float *gin = in.grad_for_write(&none);
            if (none) TH(th_transpose2d(Device::ctx(), r.grad_dptr(), gin, cols, rows));
            else TH(th_transpose2d_bwd(Device::ctx(), r.grad_dptr(), gin, rows, cols));
        });
    }
    return out;
}

Tensor Tensor::linear(const Tensor &w, const Tensor &bias, bool relu) const {  // nn.rs:54-60 fused
    TAPER_ASSERT(shape_.size() == 2, "First tensor must be 2D");
    TAPER_ASSERT(w.shape_.size() == 2, "Second tensor must be 2D");
    const int batch = (int)shape_[0], in_f = (int)shape_[1], out_f = (int)w.shape_[0];
    TAPER_ASSERT((size_t)in_f == w.shape_[1],
                 "Inner dimensions must match: " + std::to_string(in_f) + " vs " + std::to_string(w.shape_[1]));
    if (bias.defined()) TAPER_ASSERT(bias.shape_.size() == 1 && bias.shape_[0] == (size_t)out_f, "Last dimension must match for broadcasting");
    Tensor out = empty({(size_t)batch, (size_t)out_f});
    TH(th_linear_fwd(Device::ctx(), dptr(), w.dptr(), bias.defined() ? bias.dptr() : nullptr, out.dptr(), batch, in_f, out_f, relu ? 1 : 0));
    const bool need = requires_grad_ || w.requires_grad_ || (bias.defined() && bias.requires_grad_);
    if (need) {
        out.requires_grad_ = true;
        out.grad_->relu_output = relu;
        Tensor x = *this, wt = w, b = bias, r = out;
        Tape::push(out, true, [x, wt, b, r, batch, in_f, out_f, relu]() {
            if (!r.has_grad()) return;
            th_ctx *c = Device::ctx();
            const float *dy = r.grad_dptr();
            // relu backward through the post-activation mask (y > 0 <=> pre-activation > 0, Q15),
            // folded into the operand loads of the backward GEMMs
            const float *relu_y = (relu && !r.grad_->premasked) ? r.dptr() : nullptr;
            int mask = 0;
            bool none;
            float *dx = nullptr, *dw = nullptr, *db = nullptr;
            th_adam_fuse wf{}, bf{};
            const th_adam_fuse *pw = nullptr, *pb = nullptr;
            Adam *fa = FusedAdamScope::active();  // Trainer: apply this layer's Adam update in the epilogue
            if (x.get_requires_grad()) { dx = x.grad_for_write(&none); if (!none) mask |= 1; }
            // updates of downstream parameters whose gradients are already complete ride along
            th_adam_slice carried[TH_MAX_ADAM_SLICES];
            const int n_carried = fa ? fa->take_deferred(dx ? wt.dptr() : nullptr, carried) : 0;
            bool defer_w = false;
            if (wt.get_requires_grad()) {
                dw = wt.grad_for_write(&none);
                if (!none) mask |= 2;
                else if (fa && dx) {
                    // one-launch form: the dX workgroups read W, the next launch updates it.  Products as launches of their own (the big
                    // shapes): the dX product is through with W when the dW product runs -- the update rides in ITS epilogue, under the
                    // matrix work of the other tiles, instead of costing a pass over p / m / v / g of its own (4096 x 4096: 75 us each)
                    const bool ep_wanted = !(mask & 1) && x.grad_->relu_output;
                    if (th_linear_bwd_separate_products(batch, in_f, out_f, 1, 1, ep_wanted ? 1 : 0) && fa->fuse_for(wt, &wf)) pw = &wf;
                    else defer_w = true;
                }
                else if (fa && fa->fuse_for(wt, &wf)) pw = &wf;
            }
            // the consumer of this layer's output left the column sums of the (masked) gradient by row block beside it: the bias gradient
            // is their sum (tensor.rs:686-691), no pass over [batch, out] for it
            const std::shared_ptr<Buffer> colpart_in = (relu_y == nullptr && relu) ? r.grad_->dz_colpart : nullptr;
            const int colpart_in_rows = r.grad_->dz_colpart_rows;
            bool db_from_colpart = false, db_none = true, defer_b = false;
            if (b.defined() && b.get_requires_grad()) {
                db = b.grad_for_write(&none);
                db_none = none;
                if (!none) mask |= 4;
                else if (fa && colpart_in) defer_b = true;   // the gradient is a sum over a few rows of partials below: its update is carried by a later launch
                else if (fa && fa->fuse_for(b, &bf)) pb = &bf;
                db_from_colpart = colpart_in != nullptr && pb == nullptr;
            }
            // ... and this layer does the same for the fused Linear + ReLU in front of it, in its dX product's epilogue: dX * [x > 0]
            // (ops.rs:358-369) and the row blocks' column sums of that
            const int ep_rows = (dx && !(mask & 1) && x.grad_->relu_output) ? th_linear_bwd_dx_epilogue_rows(batch, in_f, out_f) : 0;
            std::shared_ptr<Buffer> colpart_out = ep_rows > 0 ? Buffer::alloc((size_t)ep_rows * in_f) : nullptr;
            TH(th_linear_bwd_adam_ex2(c, x.dptr(), wt.dptr(), dy, relu_y, dx, dw, db_from_colpart ? nullptr : db, batch, in_f, out_f, mask, pw, pb,
                                      carried, n_carried, ep_rows > 0 ? 1 : 0, colpart_out ? colpart_out->d : nullptr));
            if (db_from_colpart) {
                if (db_none) TH(th_colsum(c, colpart_in->d, db, colpart_in_rows, out_f));
                else TH(th_colsum_accum(c, colpart_in->d, db, colpart_in_rows, out_f));
            }
            if (defer_b) fa->defer_for(b);
            if (ep_rows > 0) {
                x.grad_->premasked = true;
                x.grad_->dz_colpart = colpart_out;
                x.grad_->dz_colpart_rows = ep_rows;
            }
            if (defer_w) fa->defer_for(wt);
        });
    }
    return out;
}

// ---------------------------------------------------------------- broadcast / reduce
Tensor Tensor::add_broadcast(const Tensor &o) const {  // tensor.rs:636-704
    if (shape_ == o.shape_) return *this + o;
    TAPER_ASSERT(shape_.size() == 2 && o.shape_.size() == 1, "Unsupported broadcasting shapes");
    TAPER_ASSERT(shape_[1] == o.shape_[0], "Last dimension must match for broadcasting");
    const int rows = (int)shape_[0], cols = (int)shape_[1];
    Tensor out = empty(shape_);
    TH(th_bias_add_rows(Device::ctx(), dptr(), o.dptr(), out.dptr(), rows, cols, 0));
    if (requires_grad_ || o.requires_grad_) {
        out.requires_grad_ = true;
        Tensor a = *this, b = o, r = out;
        Tape::push(out, true, [a, b, r, rows, cols]() {
            if (!r.has_grad()) return;
            if (a.get_requires_grad()) accumulate_into(a, r.grad_dptr());
            if (b.get_requires_grad()) {
                bool none;
                float *gb = b.grad_for_write(&none);
                if (none) TH(th_colsum(Device::ctx(), r.grad_dptr(), gb, rows, cols));
                else TH(th_colsum_accum(Device::ctx(), r.grad_dptr(), gb, rows, cols));
            }
        });
    }
    return out;
}

Tensor Tensor::sub_broadcast_rows(const Tensor &o) const {  // tensor.rs:707-770
    if (shape_ == o.shape_) return *this - o;
    TAPER_ASSERT(shape_.size() == 2 && o.shape_.size() == 2 && shape_[0] == o.shape_[0] && o.shape_[1] == 1,
                 "Unsupported broadcasting shapes for sub_broadcast_rows");
    const int rows = (int)shape_[0], cols = (int)shape_[1];
    Tensor out = empty(shape_);
    TH(th_sub_rows(Device::ctx(), dptr(), o.dptr(), out.dptr(), rows, cols));
    if (requires_grad_ || o.requires_grad_) {
        out.requires_grad_ = true;
        Tensor a = *this, b = o, r = out;
        Tape::push(out, true, [a, b, r, rows, cols]() {
            if (!r.has_grad()) return;
            if (a.get_requires_grad()) accumulate_into(a, r.grad_dptr());
            if (b.get_requires_grad()) TH(th_rowsum_neg_accum(Device::ctx(), r.grad_dptr(), b.grad_accum_ptr(), rows, cols));
        });
    }
    return out;
}

Tensor Tensor::mean() const {  // tensor.rs:772-800
    Tensor out = empty({1});
    const float n = (float)len();
    TH(th_sum_all(Device::ctx(), dptr(), out.dptr(), len(), n));
    if (requires_grad_) {
        out.requires_grad_ = true;
        Tensor in = *this, r = out;
        Tape::push(out, true, [in, r, n]() {
            if (!r.has_grad()) return;
            TH(th_add_scalar_dev(Device::ctx(), r.grad_dptr(), n, in.grad_accum_ptr(), in.len()));
        });
    }
    return out;
}

Tensor Tensor::reshape(const Shape &s) const {  // tensor.rs:803-840
    TAPER_ASSERT(numel(s) == len(), "Cannot reshape tensor of size " + std::to_string(len()));
    // The reference clones the Vec (814); tensors are immutable once produced,
    // so the device build shares the storage and only gives the view its own
    // grad slot / tape node.
    Tensor out = make(data_, s);
    if (requires_grad_) {
        out.requires_grad_ = true;
        // a view of [n, ...] as [n, rest] keeps the columns of a sample together: the column-sum wish (GradSlot) travels with it
        if (PoolBiasScope::active() && grad_->wants_colsum && !s.empty() && !shape_.empty() && s[0] == shape_[0]) {
            out.grad_->wants_colsum = true;
            out.grad_->colsum_bias = grad_->colsum_bias;
            out.grad_->colsum_c = grad_->colsum_c;
            out.grad_->colsum_hw = grad_->colsum_hw;
        }
        if (PoolBiasScope::active() && grad_->gapfin_cnt && !s.empty() && !shape_.empty() && s[0] == shape_[0]) {
            out.grad_->gapfin_cnt = grad_->gapfin_cnt;
            out.grad_->gapfin_bias = grad_->gapfin_bias;
            out.grad_->gapfin_hw = grad_->gapfin_hw;
        }
        Tensor in = *this, r = out;
        Tape::push(out, true, [in, r]() {
            if (r.grad_->gapfin_done) {   // the classifier finished the conv bias in its own launch (the gradient itself travels on as usual)
                r.grad_->gapfin_done = false;
                in.grad_->gapfin_done = true;
            }
            if (r.grad_->colsum_done) {   // the classifier head finished the conv bias in its own launch
                in.grad_->colsum_done = true;
                return;
            }
            if (r.grad_->colsum) {   // the classifier head left column sums instead of a gradient: hand them on
                in.grad_->colsum = std::move(r.grad_->colsum);
                return;
            }
            if (!r.has_grad()) return;
            if (PoolBiasScope::active() && !in.has_grad() && !in.grad_->buf_is_arena && !r.grad_->shared_const && !r.grad_->buf_is_arena) {
                // Trainer steps over a Sequential (one consumer per tensor): 0 + x is x -- adopt the view's gradient
                in.grad_->buf = r.grad_->buf;
                in.grad_->has = true;
                in.grad_->known_zero = false;
                in.grad_->shared_const = false;
                return;
            }
            accumulate_into(in, r.grad_dptr());
        });
    }
    return out;
}

// ---- grouped-convolution helpers (nn.rs:859-1014): strided block copies, no tape nodes ----
Tensor Tensor::slice_channels(size_t start, size_t end) const {  // nn.rs:862-886
    TAPER_ASSERT(shape_.size() == 4, "slice_channels only works on 4D tensors");
    TAPER_ASSERT(start < end && end <= shape_[1], "Invalid channel range");
    const size_t n = shape_[0], c = shape_[1], hw = shape_[2] * shape_[3], cs = end - start;
    Tensor out = empty({n, cs, shape_[2], shape_[3]});
    TH(th_copy2d(Device::ctx(), dptr() + start * hw, out.dptr(), (int64_t)n, (int)(cs * hw), (int64_t)(c * hw), (int64_t)(cs * hw)));
    return out;
}

Tensor Tensor::slice_output_channels(size_t start, size_t end) const {  // nn.rs:889-914
    TAPER_ASSERT(shape_.size() == 4, "slice_output_channels only works on 4D weight tensors");
    TAPER_ASSERT(start < end && end <= shape_[0], "Invalid output channel range");
    const size_t per = shape_[1] * shape_[2] * shape_[3], rows = end - start;
    Tensor out = empty({rows, shape_[1], shape_[2], shape_[3]});
    TH(th_copy2d(Device::ctx(), dptr() + start * per, out.dptr(), 1, (int)(rows * per), (int64_t)(rows * per), (int64_t)(rows * per)));
    return out;
}

Tensor Tensor::slice_1d(size_t start, size_t end) const {  // nn.rs:917-925
    TAPER_ASSERT(shape_.size() == 1, "slice_1d only works on 1D tensors");
    TAPER_ASSERT(start < end && end <= shape_[0], "Invalid range");
    Tensor out = empty({end - start});
    TH(th_copy2d(Device::ctx(), dptr() + start, out.dptr(), 1, (int)(end - start), (int64_t)(end - start), (int64_t)(end - start)));
    return out;
}

Tensor Tensor::cat(const std::vector<Tensor> &ts, size_t dim) {  // nn.rs:928-1014
    TAPER_ASSERT(!ts.empty(), "Cannot concatenate empty tensor list");
    const Shape &first = ts[0].shape();
    const size_t nd = first.size();
    TAPER_ASSERT(dim < nd, "Concatenation dimension out of bounds");
    size_t total = 0;
    for (const Tensor &t : ts) {
        TAPER_ASSERT(t.shape().size() == nd, "All tensors must have same number of dimensions");
        for (size_t i = 0; i < nd; ++i)
            if (i != dim) TAPER_ASSERT(t.shape()[i] == first[i], "Tensor shapes must match except in concatenation dimension");
        total += t.shape()[dim];
    }
    TAPER_ASSERT(nd == 2 || nd == 4, "Concatenation not implemented for " + std::to_string(nd) + "D tensors");
    TAPER_ASSERT(nd == 2 || dim == 1, "General 4D concatenation not implemented for dim != 1");
    Shape os = first;
    os[dim] = total;
    Tensor out = empty(os);
    // view every operand as [outer][inner_i] rows laid side by side in [outer][sum inner_i]
    size_t outer = 1, tail = 1;
    for (size_t i = 0; i < dim; ++i) outer *= first[i];
    for (size_t i = dim + 1; i < nd; ++i) tail *= first[i];
    const size_t out_ld = total * tail;
    size_t off = 0;
    for (const Tensor &t : ts) {
        const size_t inner = t.shape()[dim] * tail;
        TH(th_copy2d(Device::ctx(), t.dptr(), out.dptr() + off, (int64_t)outer, (int)inner, (int64_t)inner, (int64_t)out_ld));
        off += inner;
    }
    return out;
}

Tensor Tensor::flatten(size_t start_dim) const {  // tensor.rs:843-858
    TAPER_ASSERT(start_dim < shape_.size(), "start_dim out of bounds");
    Shape ns(shape_.begin(), shape_.begin() + start_dim);
    size_t rest = 1;
    for (size_t i = start_dim; i < shape_.size(); ++i) rest *= shape_[i];
    ns.push_back(rest);
    return reshape(ns);
}

Tensor Tensor::squeeze(int dim) const {  // tensor.rs:861-877
    Shape ns;
    if (dim >= 0) {
        TAPER_ASSERT((size_t)dim < shape_.size(), "Dimension out of bounds");
        TAPER_ASSERT(shape_[dim] == 1, "Can only squeeze dimensions of size 1");
        for (size_t i = 0; i < shape_.size(); ++i)
            if ((int)i != dim) ns.push_back(shape_[i]);
    } else {
        for (size_t d : shape_)
            if (d != 1) ns.push_back(d);
    }
    if (ns.empty()) ns.push_back(1);
    return reshape(ns);
}

Tensor Tensor::unsqueeze(size_t dim) const {  // tensor.rs:880-887
    TAPER_ASSERT(dim <= shape_.size() && shape_.size() < 4, "Dimension out of bounds");
    Shape ns = shape_;
    ns.insert(ns.begin() + dim, 1);
    return reshape(ns);
}

Tensor Tensor::sum(int dim, bool keepdim) const {  // tensor.rs:890-1018
    th_ctx *c = Device::ctx();
    if (dim < 0) {
        Tensor out = empty({1});
        TH(th_sum_all(c, dptr(), out.dptr(), len(), 1.0f));
        if (requires_grad_) {
            out.requires_grad_ = true;
            Tensor in = *this, r = out;
            Tape::push(out, true, [in, r]() {
                if (!r.has_grad()) return;
                TH(th_add_scalar_dev(Device::ctx(), r.grad_dptr(), 1.0f, in.grad_accum_ptr(), in.len()));
            });
        }
        return out;
    }
    TAPER_ASSERT((size_t)dim < shape_.size(), "Dimension " + std::to_string(dim) + " out of bounds");
    size_t outer = 1, inner = 1;
    for (int i = 0; i < dim; ++i) outer *= shape_[i];
    for (size_t i = dim + 1; i < shape_.size(); ++i) inner *= shape_[i];
    const size_t d = shape_[dim];
    Shape os;
    for (size_t i = 0; i < shape_.size(); ++i) {
        if ((int)i == dim) { if (keepdim) os.push_back(1); }
        else os.push_back(shape_[i]);
    }
    if (os.empty()) os.push_back(1);
    Tensor out = empty(os);
    // the hot path (loss.rs:108-121) reduces 2-D tensors over their last dimension: wave-shuffle kernels; a first / last dimension of any
    // shape takes the same kernels on the flattened view; a MIDDLE dimension -- which the reference accepts too (tensor.rs:917-937) -- and
    // every output of fewer elements than the tensor has dimensions (where the reference's backward skips a coordinate, tensor.rs:975)
    // take the literal kernels
    const bool literal = (inner != 1 && outer != 1) || out.len() < shape_.size();
    int64_t shp[4] = {1, 1, 1, 1};
    for (size_t i = 0; i < shape_.size(); ++i) shp[i] = (int64_t)shape_[i];
    const int nd = (int)shape_.size();
    const bool by_row = inner == 1;  // [outer, d] -> [outer]
    const int rows = by_row ? (int)outer : (int)d, cols = by_row ? (int)d : (int)inner;
    if (literal) TH(th_sum_dim(c, dptr(), out.dptr(), shp, nd, dim));
    else if (by_row) TH(th_rowsum(c, dptr(), out.dptr(), rows, cols));
    else TH(th_colsum(c, dptr(), out.dptr(), rows, cols));
    if (requires_grad_) {
        out.requires_grad_ = true;
        Tensor in = *this, r = out;
        const std::array<int64_t, 4> sh{shp[0], shp[1], shp[2], shp[3]};
        Tape::push(out, true, [in, r, by_row, rows, cols, literal, sh, nd, dim, keepdim]() {
            if (!r.has_grad()) return;
            if (literal) TH(th_sum_dim_bwd(Device::ctx(), r.grad_dptr(), in.grad_accum_ptr(), sh.data(), nd, dim, keepdim ? 1 : 0));
            else if (by_row) TH(th_rowsum_bwd(Device::ctx(), r.grad_dptr(), in.grad_accum_ptr(), rows, cols));
            else TH(th_colsum_bwd(Device::ctx(), r.grad_dptr(), in.grad_accum_ptr(), rows, cols));
        });
    }
    return out;
}

std::pair<Tensor, Tensor> Tensor::max(int dim) const {  // tensor.rs:1021-1083 (no tape node)
    th_ctx *c = Device::ctx();
    if (dim < 0) {
        // global max (tensor.rs:1072-1083): max_by keeps the LAST of equal maxima; a NaN makes partial_cmp(..).unwrap() panic
        Tensor v = empty({1}), i = empty({1});
        auto flag = Buffer::alloc(1);
        TH(th_global_max(c, dptr(), len(), v.dptr(), i.dptr(), reinterpret_cast<int *>(flag->d)));
        int nan_seen = 0;
        TH(th_memcpy_d2h(c, &nan_seen, flag->d, sizeof(int)));
        TAPER_ASSERT(!(nan_seen && len() >= 2), "called `Option::unwrap()` on a `None` value (max: NaN in the input)");   // (one element: no comparison)
        return {v, i};
    }
    TAPER_ASSERT((size_t)dim < shape_.size(), "Dimension " + std::to_string(dim) + " out of bounds");
    Shape os = shape_;
    os[dim] = 1;
    Tensor v = empty(os), i = empty(os);
    if (shape_.size() == 2) {   // the hot path's shape (loss.rs:108, 271-290)
        const int rows = (int)shape_[0], cols = (int)shape_[1];
        if (dim == 1) TH(th_rowmax(c, dptr(), v.dptr(), i.dptr(), rows, cols));
        else TH(th_colmax(c, dptr(), v.dptr(), i.dptr(), rows, cols));
        return {v, i};
    }
    // any other rank: the reference's index arithmetic as it stands (tensor.rs:1042-1066, quirk Q14 included)
    int64_t shp[4] = {1, 1, 1, 1};
    for (size_t k = 0; k < shape_.size(); ++k) shp[k] = (int64_t)shape_[k];
    TH(th_max_dim(c, dptr(), v.dptr(), i.dptr(), shp, (int)shape_.size(), dim));
    return {v, i};
}

Tensor Tensor::argmax(int dim) const { return max(dim).second; }

// ---------------------------------------------------------------- conv / pool
static void pooled_bias_grad(const Tensor &bias, const float *dy, const float *mask_y, int n, int c, int hw, bool avg, const float *cnt = nullptr);

Tensor Tensor::conv2d(const Tensor &w, const Tensor &bias, std::pair<int, int> stride, std::pair<int, int> padding,
                      std::pair<int, int> dilation, bool relu) const {  // tensor.rs:1221-1285 (+1379-1389)
    TAPER_ASSERT(shape_.size() == 4, "Input must be 4D: [N, C_in, H, W]");
    TAPER_ASSERT(w.shape_.size() == 4, "Weight must be 4D: [C_out, C_in, K_h, K_w]");
    const int n = (int)shape_[0], c_in = (int)shape_[1], h = (int)shape_[2], wd = (int)shape_[3];
    const int c_out = (int)w.shape_[0], k_h = (int)w.shape_[2], k_w = (int)w.shape_[3];
    TAPER_ASSERT((size_t)c_in == w.shape_[1], "Input and weight channel dimensions must match");
    if (bias.defined()) TAPER_ASSERT(bias.shape_ == Shape{(size_t)c_out}, "Bias must be 1D with C_out elements");
    const int h_out = (h + 2 * padding.first - dilation.first * (k_h - 1) - 1) / stride.first + 1;
    const int w_out = (wd + 2 * padding.second - dilation.second * (k_w - 1) - 1) / stride.second + 1;
    th_ctx *c = Device::ctx();
    const bool is3 = k_h == 3 && k_w == 3 && stride == std::make_pair(1, 1) && dilation == std::make_pair(1, 1);
    const bool is1 = k_h == 1 && k_w == 1;
    const float *bp = bias.defined() ? bias.dptr() : nullptr;
    Tensor out = empty({(size_t)n, (size_t)c_out, (size_t)h_out, (size_t)w_out});
    const int pad = padding.first;
    if (is3) {
        TAPER_ASSERT(padding.first == padding.second && (pad == 0 || pad == 1), "conv2d 3x3: padding must be (0,0) or (1,1)");
        TH(th_conv3x3_fwd(c, dptr(), w.dptr(), bp, out.dptr(), n, c_in, h, wd, c_out, pad, /*taper layout*/ 0, relu ? 1 : 0));
    } else if (is1) {
        TAPER_ASSERT(h == h_out && wd == w_out, "im2col_1x1 requires h_in == h_out (tensor.rs:1796-1797)");
        TH(th_conv1x1_fwd(c, dptr(), w.dptr(), bp, out.dptr(), n, c_in, h, wd, c_out, 0, relu ? 1 : 0));
    } else {   // im2col_general_simd (tensor.rs:1703-1722, 1805-1906) with its arithmetic kept as is (Q9)
        TAPER_ASSERT(stride.first > 0 && stride.second > 0 && dilation.first > 0 && dilation.second > 0, "conv2d: stride and dilation must be positive");
        TH(th_conv2d_general_fwd(c, dptr(), w.dptr(), bp, out.dptr(), n, c_in, h, wd, c_out, k_h, k_w, stride.first, stride.second,
                                 padding.first, padding.second, dilation.first, dilation.second, relu ? 1 : 0));
    }
    // Tape.  Faithful mode (Q2): the chain is cut at transpose_4d / im2col, so only
    // the bias (through add_bias_4d, tensor.rs:2003-2027) ever receives a gradient.
    const bool fb = full_backward();
    const bool bias_grad = bias.defined() && bias.requires_grad_;
    const bool w_grad = fb && w.requires_grad_;
    const bool x_grad = fb && requires_grad_ && is3 && pad == 1;
    if (bias_grad || w_grad || x_grad) {
        out.requires_grad_ = true;
        out.grad_->wants_pooled = relu && bias_grad && !w_grad && !x_grad && PoolBiasScope::active();
        out.grad_->relu_output = relu && (w_grad || x_grad);   // full backward: a max-pool behind this layer may fold the ReLU's mask into its scatter
        Tensor x = *this, wt = w, b = bias, r = out;
        Tape::push(out, true, [x, wt, b, r, n, c_in, h, wd, c_out, h_out, w_out, pad, relu, bias_grad, w_grad, x_grad, is3]() {
            th_ctx *c = Device::ctx();
            if (r.grad_->pooled_dy && !r.grad_->has) {
                // the max-pool behind this layer left its OUTPUT's gradient: db = sum of the pooled gradients whose
                // pooled value is > 0 (the one conv output each lands on, and that output's ReLU mask)
                GradSlot &g = *r.grad_;
                pooled_bias_grad(b, g.pooled_dy->d, g.pooled_y->d, g.pooled_n, g.pooled_c, g.pooled_hw, g.pooled_avg,
                                 g.pooled_avg && g.pooled_cnt ? g.pooled_cnt->d : nullptr);
                g.pooled_dy.reset();
                g.pooled_y.reset();
                g.pooled_cnt.reset();
                return;
            }
            if (!r.has_grad()) return;
            const float *gy = r.grad_dptr();
            std::shared_ptr<Buffer> dz;
            // (premasked: the max-pool behind this layer scattered its gradient with this ReLU's mask already applied -- th_maxpool2d_relu_bwd)
            // plane sums [n][c_out] of the masked gradient: all the bias needs of it -- left by the pool's scatter, or formed by the ReLU's backward
            std::shared_ptr<Buffer> ps = r.grad_->premasked ? r.grad_->plane_sums : nullptr;
            r.grad_->plane_sums.reset();
            if (relu && !r.grad_->premasked) {
                dz = Buffer::alloc(r.len());
                if (bias_grad) {
                    ps = Buffer::alloc((size_t)n * c_out);
                    TH(th_relu_bwd_plane_sums(c, r.dptr(), gy, dz->d, ps->d, n, c_out, h_out * w_out));
                } else {
                    TH(th_relu_bwd(c, r.dptr(), gy, dz->d, r.len(), 0));
                }
                gy = dz->d;
            }
            // a slot that is None is overwritten (0 + x): no zero fill in front of the launch
            bool none;
            if (bias_grad) {
                float *db = b.grad_for_write(&none);
                if (ps) TH(th_bias_grad_plane_sums(c, ps->d, db, n, c_out, none ? 0 : 1));
                else TH(th_bias_grad_nchw_masked(c, gy, nullptr, db, n, c_out, h_out * w_out, none ? 0 : 1));
            }
            if (w_grad && is3) {
                float *dw = wt.grad_for_write(&none);
                TH(th_conv3x3_bwd_weight(c, x.dptr(), gy, dw, n, c_in, h, wd, c_out, pad, 0, none ? 0 : 1));
            }
            if (x_grad) {
                float *dx = x.grad_for_write(&none);
                TH(th_conv3x3_bwd_input(c, gy, wt.dptr(), dx, n, c_in, h, wd, c_out, pad, 0, none ? 0 : 1));
            }
        });
    }
    return out;
}

Tensor Tensor::conv2d_relu(const Tensor &w, const Tensor &bias, std::pair<int, int> stride, std::pair<int, int> padding,
                           std::pair<int, int> dilation) const {
    return conv2d(w, bias, stride, padding, dilation, true);
}

// Bias gradient of a bias-only Conv2dReLU from pooled tensors (see GradSlot).  Inside a fused-update step it is the last backward
// launch of the faithful CNNs: the bias's own Adam update and every deferred update still waiting ride in the same launch.
static void pooled_bias_grad(const Tensor &bias, const float *dy, const float *mask_y, int n, int c, int hw, bool avg, const float *cnt) {
    th_ctx *ctx = Device::ctx();
    bool none;
    float *db = bias.grad_for_write(&none);
    Adam *fa = FusedAdamScope::active();
    th_adam_fuse bf{};
    if (avg && cnt && none) {   // the pool's forward counted each plane's positive elements: 2 n c floats instead of n c hw
        th_adam_slice carried[TH_MAX_ADAM_SLICES];
        int n_carried = 0;
        const bool fuse = fa && fa->fuse_for(bias, &bf);
        if (fuse) n_carried = fa->take_deferred(nullptr, carried);
        TH(th_bias_grad_counts_adam(ctx, dy, cnt, db, n, c, hw, fuse ? &bf : nullptr, carried, n_carried));
        return;
    }
    if (fa && none && fa->fuse_for(bias, &bf)) {
        th_adam_slice carried[TH_MAX_ADAM_SLICES];
        const int n_carried = fa->take_deferred(nullptr, carried);
        TH(th_bias_grad_masked_adam(ctx, dy, mask_y, db, n, c, hw, avg ? 1 : 0, &bf, carried, n_carried));
        return;
    }
    if (avg) TH(th_bias_grad_avgpool_masked(ctx, dy, mask_y, db, n, c, hw, none ? 0 : 1));
    else TH(th_bias_grad_nchw_masked(ctx, dy, mask_y, db, n, c, hw, none ? 0 : 1));
}

// tape node of a bias-only Conv2dReLU -> MaxPool2d(2) whose pooled output is `out` (faithful mode, Q2: the bias is the only trainable input)
static void push_pooled_conv_bias_node(Tensor &out, const Tensor &bias, int n, int c_out, int hp, int wp) {
    if (!(bias.defined() && bias.requires_grad_) || NoGradScope::active()) return;
    {
        out.requires_grad_ = true;
        // inside a Trainer step (one consumer per tensor) the sums of dX * [x > 0] per column are all this node needs of its gradient
        out.grad_->wants_colsum = PoolBiasScope::active() && !bias.has_grad();
        if (out.grad_->wants_colsum) {
            out.grad_->colsum_bias = bias.grad_;
            out.grad_->colsum_c = c_out;
            out.grad_->colsum_hw = hp * wp;
        }
        Tensor b = bias, r = out;
        Tape::push(out, true, [b, r, n, c_out, hp, wp]() {
            if (r.grad_->colsum_done) {   // gradient and Adam update of the bias already happened in the head's launch
                r.grad_->colsum_done = false;
                return;
            }
            if (r.grad_->colsum) {
                th_ctx *ctx = Device::ctx();
                bool none;
                float *db = b.grad_for_write(&none);
                TAPER_ASSERT(none, "conv2d_relu_maxpool2: column sums arrived for a bias that already holds a gradient");
                Adam *fa = FusedAdamScope::active();
                th_adam_fuse bf{};
                th_adam_slice carried[TH_MAX_ADAM_SLICES];
                int n_carried = 0;
                const bool fuse = fa && fa->fuse_for(b, &bf);
                if (fuse) n_carried = fa->take_deferred(nullptr, carried);
                TH(th_bias_from_colsum_adam(ctx, r.grad_->colsum->d, db, c_out, hp * wp, fuse ? &bf : nullptr, carried, n_carried));
                r.grad_->colsum.reset();
                return;
            }
            if (!r.has_grad()) return;
            // every pooled gradient lands on exactly one conv output (tensor.rs:1496-1519) whose ReLU mask is "pooled value > 0"
            pooled_bias_grad(b, r.grad_dptr(), r.dptr(), n, c_out, hp * wp, false);
        });
    }
}

// tape node of a bias-only Conv2dReLU -> global average pool: `out` = plane means, cnt = outputs > 0 per plane
static void push_gap_conv_bias_node(Tensor &out, const Tensor &bias, const std::shared_ptr<Buffer> &cnt, int n, int c_out, int hw) {
    {
        out.requires_grad_ = true;
        if (PoolBiasScope::active() && !bias.has_grad()) {   // (GradSlot: a classifier launch may finish this bias)
            out.grad_->gapfin_cnt = cnt;
            out.grad_->gapfin_bias = bias.grad_;
            out.grad_->gapfin_hw = hw;
        }
        Tensor b = bias, r = out;
        Tape::push(out, true, [b, r, cnt, n, c_out, hw]() {
            if (r.grad_->gapfin_done) {   // gradient and Adam update of the bias already happened in the classifier's launch
                r.grad_->gapfin_done = false;
                return;
            }
            if (!r.has_grad()) return;
            // every element of a plane receives g / hw (tensor.rs:1626-1628) and passes the ReLU mask iff it is > 0: db = sum_n g / hw * count
            pooled_bias_grad(b, r.grad_dptr(), nullptr, n, c_out, hw, true, cnt->d);
        });
    }
}

bool Tensor::conv2d_relu_maxpool2_supported(const Tensor &w, const Tensor &bias, std::pair<int, int> padding) const {
    if (full_backward() || shape_.size() != 4 || w.shape_.size() != 4 || w.shape_[2] != 3 || w.shape_[3] != 3) return false;
    if (w.shape_[1] != shape_[1] || padding.first != padding.second) return false;
    if (bias.defined() && bias.shape_ != Shape{w.shape_[0]}) return false;
    if (((uintptr_t)w.dptr() & 15) != 0) return false;
    return th_conv3x3_pool2_supported((int)shape_[1], (int)shape_[2], (int)shape_[3], (int)w.shape_[0], padding.first) != 0;
}

Tensor Tensor::conv2d_relu_maxpool2(const Tensor &w, const Tensor &bias, std::pair<int, int> padding) const {
    // tensor.rs:1221-1285 + nn.rs:433-490 + tensor.rs:1391-1470 (k = s = 2, p = 0), values only
    TAPER_ASSERT(conv2d_relu_maxpool2_supported(w, bias, padding), "conv2d_relu_maxpool2: unsupported shapes / mode");
    const int n = (int)shape_[0], c_in = (int)shape_[1], h = (int)shape_[2], wd = (int)shape_[3], c_out = (int)w.shape_[0];
    const int pad = padding.first, hp = (h + 2 * pad - 2) / 2, wp = (wd + 2 * pad - 2) / 2;
    Tensor out = empty({(size_t)n, (size_t)c_out, (size_t)hp, (size_t)wp});
    TH(th_conv3x3_pool2_fwd(Device::ctx(), dptr(), w.dptr(), bias.defined() ? bias.dptr() : nullptr, out.dptr(), n, c_in, h, wd, c_out,
                            pad, 1));
    push_pooled_conv_bias_node(out, bias, n, c_out, hp, wp);
    return out;
}

bool Tensor::conv2d_relu_gap_supported(const Tensor &w, const Tensor &bias, std::pair<int, int> padding) const {
    if (full_backward() || shape_.size() != 4 || w.shape_.size() != 4 || w.shape_[2] != 3 || w.shape_[3] != 3) return false;
    if (w.shape_[1] != shape_[1] || padding.first != padding.second) return false;
    if (!bias.defined() || bias.shape_ != Shape{w.shape_[0]}) return false;
    if (((uintptr_t)w.dptr() & 15) != 0) return false;
    return th_conv3x3_gap_supported((int)shape_[0], (int)shape_[1], (int)shape_[2], (int)shape_[3], (int)w.shape_[0], padding.first) != 0;
}

Tensor Tensor::conv2d_relu_gap(const Tensor &w, const Tensor &bias, std::pair<int, int> padding) const {
    // tensor.rs:1221-1285 + nn.rs:433-490 + tensor.rs:1524-1660 (kernel = the plane, nn.rs:670-686)
    TAPER_ASSERT(conv2d_relu_gap_supported(w, bias, padding), "conv2d_relu_gap: unsupported shapes / mode");
    const int n = (int)shape_[0], c_in = (int)shape_[1], h = (int)shape_[2], wd = (int)shape_[3], c_out = (int)w.shape_[0];
    const int pad = padding.first, hw = (h + 2 * pad - 2) * (wd + 2 * pad - 2);
    Tensor out = empty({(size_t)n, (size_t)c_out, 1, 1});
    const bool bias_grad = bias.requires_grad_ && !NoGradScope::active();
    std::shared_ptr<Buffer> cnt = bias_grad ? Buffer::alloc((size_t)n * c_out) : nullptr;
    TH(th_conv3x3_gap_fwd(Device::ctx(), dptr(), w.dptr(), bias.dptr(), out.dptr(), cnt ? cnt->d : nullptr, n, c_in, h, wd, c_out, pad, 1));
    if (bias_grad) push_gap_conv_bias_node(out, bias, cnt, n, c_out, hw);
    return out;
}

static bool conv_chain_describe(const Tensor &x, const std::vector<ConvStage> &stages, std::vector<th_conv_stage> *out) {
    if (full_backward() || x.shape().size() != 4 || stages.empty()) return false;
    size_t c_in = x.shape()[1];
    for (const auto &st : stages) {
        const Shape &ws = st.weight.shape();
        if (ws.size() != 4 || ws[1] != c_in || ws[2] != 3 || ws[3] != 3) return false;
        if (!st.bias.defined() || st.bias.shape() != Shape{ws[0]}) return false;
        out->push_back(th_conv_stage{st.weight.dptr(), st.bias.dptr(), (int)ws[0], st.post});
        c_in = ws[0];
    }
    return true;
}

int Tensor::conv_chain_supported(const std::vector<ConvStage> &stages) const {
    std::vector<th_conv_stage> d;
    if (!conv_chain_describe(*this, stages, &d)) return 0;
    // one image per workgroup: a launch takes the same ~100 us for 8 images as for 256 (measured, tools/chain_time.py: the reference front
    // layer by layer 84 / 88 / 97 / 103 / 144 us at batch 8 / 32 / 64 / 96 / 192 against 100-104 us in one launch) -- small batches stay layered
    if (shape_[0] < 96) return 0;
    return th_conv_chain_supported((int)shape_[1], (int)shape_[2], (int)shape_[3], d.data(), (int)d.size());
}

Tensor Tensor::conv_chain(const std::vector<ConvStage> &stages) const {
    // nn.rs:149-151 over Conv2dReLU (433-490) / MaxPool2d (508-549) / AdaptiveAvgPool2d (655-697) rows; tensor.rs:1221-1285, 1391-1470, 1524-1660
    std::vector<th_conv_stage> d;
    TAPER_ASSERT(conv_chain_describe(*this, stages, &d) &&
                     th_conv_chain_supported((int)shape_[1], (int)shape_[2], (int)shape_[3], d.data(), (int)d.size()) != 0,
                 "conv_chain: unsupported stages / mode");
    const int n = (int)shape_[0];
    int h = (int)shape_[2], w = (int)shape_[3];
    for (const auto &st : stages) {
        if (st.post == TH_CHAIN_MAXPOOL2) { h /= 2; w /= 2; }
    }
    const ConvStage &last = stages.back();
    const int c_out = (int)last.weight.shape()[0];
    if (last.post == TH_CHAIN_GLOBAL_AVG) {
        Tensor out = empty({(size_t)n, (size_t)c_out, 1, 1});
        const bool bias_grad = last.bias.requires_grad_ && !NoGradScope::active();
        std::shared_ptr<Buffer> cnt = bias_grad ? Buffer::alloc((size_t)n * c_out) : nullptr;
        TH(th_conv_chain_fwd(Device::ctx(), dptr(), d.data(), (int)d.size(), out.dptr(), cnt ? cnt->d : nullptr, n, (int)shape_[1], (int)shape_[2],
                             (int)shape_[3]));
        if (bias_grad) push_gap_conv_bias_node(out, last.bias, cnt, n, c_out, h * w);
        return out;
    }
    // (a pooled map, or -- a run that ends in a conv row -- that row's map: either way dY * [y > 0] summed per channel is the last conv's
    // bias gradient, tensor.rs:1496-1519, 2017-2024, ops.rs:358-369)
    Tensor out = empty({(size_t)n, (size_t)c_out, (size_t)h, (size_t)w});
    TH(th_conv_chain_fwd(Device::ctx(), dptr(), d.data(), (int)d.size(), out.dptr(), nullptr, n, (int)shape_[1], (int)shape_[2], (int)shape_[3]));
    push_pooled_conv_bias_node(out, last.bias, n, c_out, h, w);
    return out;
}

int Tensor::conv_chain_head_supported(const std::vector<ConvStage> &stages, int classes) const {
    std::vector<th_conv_stage> d;
    if (!conv_chain_describe(*this, stages, &d)) return 0;
    if (shape_[0] < 96) return 0;   // (as conv_chain_supported: one image per workgroup)
    return th_conv_chain_head_supported((int)shape_[1], (int)shape_[2], (int)shape_[3], d.data(), (int)d.size(), classes);
}

Tensor Tensor::conv_chain_head(const std::vector<ConvStage> &stages, const th_chain_head &head) const {
    // nn.rs:149-151 over the Conv2dReLU / MaxPool2d rows, Flatten (730-756) and the last Linear (54-60) + loss.rs:101-195 row by row
    std::vector<th_conv_stage> d;
    TAPER_ASSERT(conv_chain_describe(*this, stages, &d) && stages.back().post == TH_CHAIN_MAXPOOL2, "conv_chain_head: unsupported stages / mode");
    const int n = (int)shape_[0];
    int h = (int)shape_[2], w = (int)shape_[3];
    for (const auto &st : stages)
        if (st.post == TH_CHAIN_MAXPOOL2) { h /= 2; w /= 2; }
    Tensor out = empty({(size_t)n, stages.back().weight.shape()[0], (size_t)h, (size_t)w});
    TH(th_conv_chain_head_fwd(Device::ctx(), dptr(), d.data(), (int)d.size(), out.dptr(), n, (int)shape_[1], (int)shape_[2], (int)shape_[3], &head));
    return out;
}

int Tensor::conv_chain_mlp3_supported(const std::vector<ConvStage> &stages, int h1, int h2, int classes) const {
    std::vector<th_conv_stage> d;
    if (!conv_chain_describe(*this, stages, &d)) return 0;
    if (shape_[0] < 96) return 0;   // (as conv_chain_supported: one image per workgroup)
    return th_conv_chain_mlp3_supported((int)shape_[1], (int)shape_[2], (int)shape_[3], d.data(), (int)d.size(), (int)shape_[0], h1, h2, classes);
}

Tensor Tensor::conv_chain_mlp3(const std::vector<ConvStage> &stages, float *d_cnt, const float *d_targets, const th_mlp3_layer *layers, float *d_dx,
                               float *d_loss, float *d_ncorrect, float *d_metrics, int64_t capacity, int64_t *d_state, int64_t advance,
                               int32_t *d_tick, const th_mlp3_gap *gap) const {
    // nn.rs:149-151 over the Conv2dReLU / MaxPool2d / AdaptiveAvgPool2d rows, Flatten (730-756), the Linear + ReLU rows and the last Linear
    // (54-60), loss.rs:101-195: the rows of the classifier in the chain launch, the sums over the batch in a second one
    std::vector<th_conv_stage> d;
    TAPER_ASSERT(conv_chain_describe(*this, stages, &d) && stages.back().post == TH_CHAIN_GLOBAL_AVG, "conv_chain_mlp3: unsupported stages / mode");
    const int n = (int)shape_[0];
    Tensor out = empty({(size_t)n, stages.back().weight.shape()[0], 1, 1});
    TH(th_conv_chain_mlp3_xent(Device::ctx(), dptr(), d.data(), (int)d.size(), out.dptr(), d_cnt, n, (int)shape_[1], (int)shape_[2], (int)shape_[3],
                               d_targets, layers, d_dx, d_loss, d_ncorrect, d_metrics, capacity, d_state, advance, d_tick, gap));
    return out;
}

Tensor Tensor::max_pool2d(std::pair<int, int> k, std::pair<int, int> s, std::pair<int, int> p) const {  // tensor.rs:1391-1521
    TAPER_ASSERT(shape_.size() == 4, "Input must be 4D: [N, C, H, W]");
    if (s.first == 0) s = k;  // stride.unwrap_or(kernel_size)
    const int n = (int)shape_[0], ch = (int)shape_[1], h = (int)shape_[2], w = (int)shape_[3];
    TAPER_ASSERT(k.first > 0 && k.second > 0 && h + 2 * p.first >= k.first && w + 2 * p.second >= k.second, "max_pool2d: bad geometry");
    const int h_out = (h + 2 * p.first - k.first) / s.first + 1, w_out = (w + 2 * p.second - k.second) / s.second + 1;
    Tensor out = empty({(size_t)n, (size_t)ch, (size_t)h_out, (size_t)w_out});
    auto arg = Buffer::alloc(out.len() * 2);  // int64 per output
    TH(th_maxpool2d_fwd(Device::ctx(), dptr(), out.dptr(), reinterpret_cast<int64_t *>(arg->d), n, ch, h, w, k.first, k.second,
                        s.first, s.second, p.first, p.second));
    if (requires_grad_) {
        out.requires_grad_ = true;
        Tensor in = *this, r = out;
        const int hw_out = h_out * w_out;
        Tape::push(out, true, [in, r, arg, n, ch, h, w, k, s, p, hw_out]() {
            if (!r.has_grad()) return;
            if (in.grad_->wants_pooled && !in.grad_->has && !r.grad_->shared_const) {   // PoolBiasScope: see GradSlot
                GradSlot &g = *in.grad_;
                g.pooled_dy = r.grad_->buf;
                g.pooled_y = r.data_;
                g.pooled_n = n;
                g.pooled_c = ch;
                g.pooled_hw = hw_out;
                g.pooled_avg = false;
                return;
            }
            bool none;
            float *gin = in.grad_for_write(&none);  // zero_first (Q5) overwrites whatever was there
            if (in.grad_->relu_output && th_maxpool2d_relu_bwd_supported(n, ch, h, w, k.first, k.second, s.first, s.second, p.first, p.second)) {
                // the input is a Conv2dReLU's output (full backward): the ReLU's backward rides in the scatter, and the conv's node skips it
                in.grad_->plane_sums = Buffer::alloc((size_t)n * ch);
                TH(th_maxpool2d_relu_bwd(Device::ctx(), r.grad_dptr(), reinterpret_cast<const int64_t *>(arg->d), r.dptr(), in.dptr(), gin,
                                         in.grad_->plane_sums->d, n, ch, h, w));
                in.grad_->premasked = true;
                return;
            }
            TH(th_maxpool2d_bwd(Device::ctx(), r.grad_dptr(), reinterpret_cast<const int64_t *>(arg->d), gin, n, ch, h, w, k.first,
                                k.second, s.first, s.second, p.first, p.second, /*zero_first*/ 1));
        });
    }
    return out;
}

Tensor Tensor::avg_pool2d(std::pair<int, int> k, std::pair<int, int> s, std::pair<int, int> p) const {  // tensor.rs:1524-1660
    TAPER_ASSERT(shape_.size() == 4, "Input must be 4D: [N, C, H, W]");
    if (s.first == 0) s = k;
    const int n = (int)shape_[0], ch = (int)shape_[1], h = (int)shape_[2], w = (int)shape_[3];
    TAPER_ASSERT(k.first > 0 && k.second > 0 && h + 2 * p.first >= k.first && w + 2 * p.second >= k.second, "avg_pool2d: bad geometry");
    const int h_out = (h + 2 * p.first - k.first) / s.first + 1, w_out = (w + 2 * p.second - k.second) / s.second + 1;
    Tensor out = empty({(size_t)n, (size_t)ch, (size_t)h_out, (size_t)w_out});
    if (grad_ && grad_->wants_pooled && k.first == h && k.second == w && p.first == 0 && p.second == 0) {
        // PoolBiasScope: the producer is a bias-only Conv2dReLU -- count each plane's positive elements on the way (see GradSlot)
        grad_->pooled_cnt = Buffer::alloc((size_t)n * ch);
        TH(th_avgpool2d_global_fwd_counts(Device::ctx(), dptr(), out.dptr(), grad_->pooled_cnt->d, n, ch, h * w));
    } else
    TH(th_avgpool2d_fwd(Device::ctx(), dptr(), out.dptr(), n, ch, h, w, k.first, k.second, s.first, s.second, p.first, p.second));
    if (requires_grad_) {
        out.requires_grad_ = true;
        Tensor in = *this, r = out;
        Tape::push(out, true, [in, r, n, ch, h, w, k, s, p]() {
            if (!r.has_grad()) return;
            if (in.grad_->wants_pooled && !in.grad_->has && !r.grad_->shared_const && k.first == h && k.second == w && p.first == 0 &&
                p.second == 0) {   // PoolBiasScope, global pool: see GradSlot
                GradSlot &g = *in.grad_;
                g.pooled_dy = r.grad_->buf;
                g.pooled_y = in.data_;
                g.pooled_n = n;
                g.pooled_c = ch;
                g.pooled_hw = h * w;
                g.pooled_avg = true;
                return;
            }
            bool none;
            float *gin = in.grad_for_write(&none);
            if (none && in.grad_->relu_output && k.first == h && k.second == w && p.first == 0 && p.second == 0) {
                // full backward, a global pool behind a Conv2dReLU: that ReLU's backward and the bias's plane sums ride in this launch
                in.grad_->plane_sums = Buffer::alloc((size_t)n * ch);
                TH(th_avgpool2d_global_relu_bwd(Device::ctx(), r.grad_dptr(), in.dptr(), gin, in.grad_->plane_sums->d, n, ch, h * w));
                in.grad_->premasked = true;
                return;
            }
            if (none) TH(th_fill_f32(Device::ctx(), gin, 0.0f, in.len()));
            TH(th_avgpool2d_bwd(Device::ctx(), r.grad_dptr(), gin, n, ch, h, w, k.first, k.second, s.first, s.second, p.first, p.second));
        });
    }
    return out;
}

}  // namespace taper

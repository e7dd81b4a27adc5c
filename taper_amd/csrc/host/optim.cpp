// optim.cpp -- flat parameter / gradient arenas, SGD / Adam / AdamW and the fused-update scopes (src/optim.rs).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "nn_internal.h"

namespace taper {
// ---------------------------------------------------------------- flat arenas + optimizers
FlatParams::FlatParams(const std::vector<Tensor> &ps) : params(ps) {
    th_ctx *ctx = Device::ctx();
    offsets.resize(ps.size() + 1, 0);
    for (size_t i = 0; i < ps.size(); ++i) {
        // keep every slice 16-byte aligned for the dwordx4 kernels
        offsets[i + 1] = offsets[i] + (int64_t)((ps[i].len() + 3) / 4 * 4);
    }
    total = offsets.back();
    // A parameter list that an earlier optimizer already homed (the reference allows several optimizers over the same
    // tensors: Adam re-created with another lr, SGD then Adam): adopt that optimizer's arenas, so both keep updating the
    // storage every handle points at.  A list that only partly overlaps an existing arena cannot be laid out flat without
    // detaching the earlier optimizer (its captured graphs and fused-update pointers would go stale): refuse it.
    size_t homed = 0;
    for (size_t i = 0; i < ps.size(); ++i) homed += (!ps[i].data_->owned && ps[i].data_->parent && ps[i].grad_->buf_is_arena) ? 1 : 0;
    if (homed) {
        std::shared_ptr<Buffer> P = ps[0].data_->parent, G = ps[0].grad_->buf ? ps[0].grad_->buf->parent : nullptr;
        bool same = homed == ps.size() && P && G && (int64_t)P->n == total && (int64_t)G->n == total;
        for (size_t i = 0; same && i < ps.size(); ++i)
            same = ps[i].data_->parent == P && ps[i].data_->d == P->d + offsets[i] && ps[i].grad_->buf->parent == G &&
                   ps[i].grad_->buf->d == G->d + offsets[i];
        TAPER_ASSERT(same, "optimizer: some of these parameters already live in another optimizer's flat arena; build the new "
                           "optimizer over the same parameter list (same order) or over parameters no optimizer holds yet");
        p_arena = P;
        g_arena = G;
        d_offsets_buf = Buffer::alloc((ps.size() + 1) * 2);
        d_has_grad_buf = Buffer::alloc(ps.size());
        TH(th_memcpy_h2d(ctx, d_offsets_buf->d, offsets.data(), offsets.size() * sizeof(int64_t)));
        uploaded_mask.assign(ps.size(), -1);
        sync_mask();
        return;
    }
    p_arena = Buffer::alloc((size_t)total);
    g_arena = Buffer::alloc((size_t)total);
    TH(th_fill_f32(ctx, p_arena->d, 0.f, (size_t)total));
    TH(th_fill_f32(ctx, g_arena->d, 0.f, (size_t)total));
    for (size_t i = 0; i < ps.size(); ++i) {
        Buffer &b = *ps[i].data_;
        float *dst = p_arena->d + offsets[i];
        TH(th_memcpy_d2d(ctx, dst, b.d, b.n * sizeof(float)));
        // re-home the storage IN PLACE so every handle (model layers, user clones) follows
        if (b.owned && b.d) TH(th_free(ctx, b.d));
        b.d = dst;
        b.owned = false;
        b.parent = p_arena;
        GradSlot &g = *ps[i].grad_;
        const bool had = g.has;
        auto view = Buffer::view(g_arena, (size_t)offsets[i], ps[i].len());
        if (had && g.buf) TH(th_memcpy_d2d(ctx, view->d, g.buf->d, ps[i].len() * sizeof(float)));
        g.buf = view;
        g.buf_is_arena = true;
        g.shared_const = false;
        g.known_zero = !had;
    }
    d_offsets_buf = Buffer::alloc((ps.size() + 1) * 2);
    d_has_grad_buf = Buffer::alloc(ps.size());
    TH(th_memcpy_h2d(ctx, d_offsets_buf->d, offsets.data(), offsets.size() * sizeof(int64_t)));
    uploaded_mask.assign(ps.size(), -1);
    sync_mask();
}

void FlatParams::rehome_grads(const std::shared_ptr<Buffer> &arena) {
    TAPER_ASSERT(arena && arena->n >= (size_t)total, "FlatParams::rehome_grads: the new arena is too small");
    // every slot is checked BEFORE anything moves: a slot without a buffer, or one that has left the arena, is an error, not a null dereference
    for (size_t i = 0; i < params.size(); ++i) {
        const GradSlot &gs = *params[i].grad_;
        TAPER_ASSERT(gs.buf && gs.buf_is_arena && gs.buf->parent == g_arena, "FlatParams::rehome_grads: a grad slot has left the arena");
    }
    th_ctx *ctx = Device::ctx();
    TH(th_memcpy_d2d(ctx, arena->d, g_arena->d, (size_t)total * sizeof(float)));
    for (size_t i = 0; i < params.size(); ++i) {
        Buffer &v = *params[i].grad_->buf;
        v.d = arena->d + offsets[i];
        v.parent = arena;
    }
    Device::sync();   // the copy has run before the old arena can go back to the pool
    g_arena = arena;
}

size_t FlatParams::sync_mask(const std::vector<char> *excluded) {
    std::vector<int32_t> mask(params.size());
    size_t selected = 0;
    for (size_t i = 0; i < params.size(); ++i) {
        mask[i] = (params[i].has_grad() && !(excluded && (*excluded)[i])) ? 1 : 0;
        selected += (size_t)mask[i];
    }
    if (mask == uploaded_mask) return selected;
    // h2d synchronises: illegal inside a graph capture; callers run one eager
    // step first so the (static) mask is already resident
    TH(th_memcpy_h2d(Device::ctx(), d_has_grad_buf->d, mask.data(), mask.size() * sizeof(int32_t)));
    uploaded_mask = mask;
    return selected;
}

void FlatParams::zero_missing() {
    for (size_t i = 0; i < params.size(); ++i) {
        GradSlot &g = *params[i].grad_;
        if (!g.has && !g.known_zero) {
            TH(th_fill_f32(Device::ctx(), g.buf->d, 0.f, params[i].len()));
            g.known_zero = true;
        }
    }
}

void FlatParams::zero_grad() {
    for (auto &p : params) p.zero_grad();
}

SGD::SGD(const std::vector<Tensor> &params, float lr) : fp_(params) {
    lr_buf_ = Buffer::alloc(4);
    TH(th_memcpy_h2d(Device::ctx(), lr_buf_->d, &lr, sizeof(float)));
}

void SGD::step() {  // optim.rs:21-33
    fp_.sync_mask();
    TH(th_sgd_step(Device::ctx(), fp_.p_arena->d, fp_.g_arena->d, fp_.d_offsets(), fp_.d_has_grad(), (int)fp_.params.size(),
                   fp_.total, lr_buf_->d));
}

Adam::Adam(const std::vector<Tensor> &params, float lr, float beta1, float beta2, float eps, float wd)
    : fp_(params), lr_(lr), beta1_(beta1), beta2_(beta2), eps_(eps), wd_(wd) {  // optim.rs:54-81
    th_ctx *ctx = Device::ctx();
    m_ = Buffer::alloc((size_t)fp_.total);
    v_ = Buffer::alloc((size_t)fp_.total);
    TH(th_fill_f32(ctx, m_->d, 0.f, (size_t)fp_.total));
    TH(th_fill_f32(ctx, v_->d, 0.f, (size_t)fp_.total));
    state_ = Buffer::alloc(4);
    TH(th_fill_f32(ctx, state_->d, 0.f, 4));  // t = 0
    fused_.assign(fp_.params.size(), 0);
    set_lr(lr);
}

bool Adam::fuse_for(const Tensor &param, th_adam_fuse *out) { return fuse_for_slot(param.grad_, out); }

bool Adam::fuse_for_slot(const std::shared_ptr<GradSlot> &slot, th_adam_fuse *out) {
    for (size_t i = 0; i < fp_.params.size(); ++i) {
        if (fp_.params[i].grad_ != slot) continue;
        const int64_t off = fp_.offsets[i];
        *out = th_adam_fuse{fp_.p_arena->d + off, m_->d + off, v_->d + off, d_tick(), state_->d + 2, beta1_, beta2_, eps_, wd_};
        fused_[i] = 1;
        return true;
    }
    return false;
}

bool Adam::defer_for(const Tensor &param) {
    th_adam_fuse f{};
    if (!param.grad_->buf || !fuse_for(param, &f)) return false;
    deferred_.push_back(th_adam_slice{param.grad_->buf->d, (int64_t)param.len(), f});
    return true;
}

int Adam::take_deferred(const float *launch_reads, th_adam_slice *out) {
    int n = 0;
    for (size_t i = 0; i < deferred_.size() && n < TH_MAX_ADAM_SLICES;) {
        if (deferred_[i].f.d_p == launch_reads) { ++i; continue; }   // shared weight: that launch reads it
        out[n++] = deferred_[i];
        deferred_.erase(deferred_.begin() + (long)i);
    }
    return n;
}

void Adam::flush_deferred() {
    for (size_t i = 0; i < deferred_.size(); i += TH_MAX_ADAM_SLICES)
        TH(th_adam_slices(Device::ctx(), deferred_.data() + i, (int)std::min<size_t>(TH_MAX_ADAM_SLICES, deferred_.size() - i)));
    deferred_.clear();
}

namespace {
thread_local Adam *t_fused_adam = nullptr;
}
FusedAdamScope::FusedAdamScope(Adam *adam) : prev_(t_fused_adam) {
    t_fused_adam = adam;
    if (adam) adam->set_external_tick(true);
}
FusedAdamScope::~FusedAdamScope() {
    if (t_fused_adam) t_fused_adam->set_external_tick(false);
    t_fused_adam = prev_;
}
Adam *FusedAdamScope::active() { return t_fused_adam; }

namespace {
thread_local const Communicator *t_tail_exchange = nullptr;
}
TailExchangeScope::TailExchangeScope(const Communicator *comm) : prev_(t_tail_exchange) { t_tail_exchange = comm; }
TailExchangeScope::~TailExchangeScope() { t_tail_exchange = prev_; }
const Communicator *TailExchangeScope::active() { return t_tail_exchange; }

namespace {
thread_local bool t_pool_bias = false;
thread_local bool t_no_grad = false;
}
NoGradScope::NoGradScope() : prev_(t_no_grad) { t_no_grad = true; }
NoGradScope::~NoGradScope() { t_no_grad = prev_; }
bool NoGradScope::active() { return t_no_grad; }
PoolBiasScope::PoolBiasScope(bool on) : prev_(t_pool_bias) { t_pool_bias = on; }
PoolBiasScope::~PoolBiasScope() { t_pool_bias = prev_; }
bool PoolBiasScope::active() { return t_pool_bias; }

void Adam::set_lr(float lr) {  // optim.rs:125-127
    lr_ = lr;
    TH(th_memcpy_h2d(Device::ctx(), state_->d + 2, &lr, sizeof(float)));
}

void Adam::step() {  // optim.rs:83-113
    // In carry mode the next step's first launch applies what is left in ONE spare workgroup: fine for a classifier
    // head's W / b, far too slow for a big slice (the reference CNN's Linear(128, 64) weight cost that launch 8 us) --
    // only slices of <= 4096 elements wait, the rest is applied now.
    std::vector<th_adam_slice> wait;
    if (carry_deferred_) {
        std::vector<th_adam_slice> now;
        for (const th_adam_slice &d : deferred_) (d.n <= 4096 ? wait : now).push_back(d);
        deferred_.swap(now);
    }
    if (!deferred_.empty()) {
        // complete gradients no backward launch carried.  When the arena-wide launch below runs anyway (some
        // parameter was not fused), it takes them along: one launch instead of two, same arithmetic, same t.
        bool general = false;
        for (size_t i = 0; i < fp_.params.size(); ++i) general = general || (!fused_[i] && fp_.params[i].has_grad());
        if (general) {
            for (const th_adam_slice &d : deferred_)
                for (size_t i = 0; i < fp_.params.size(); ++i)
                    if (fp_.p_arena->d + fp_.offsets[i] == d.f.d_p) fused_[i] = 0;
            deferred_.clear();
        } else {
            flush_deferred();
        }
    }
    deferred_.swap(wait);
    // parameters whose update already ran in a fused epilogue this step are masked out
    const size_t left = fp_.sync_mask(&fused_);
    std::fill(fused_.begin(), fused_.end(), 0);
    if (external_tick_ && left == 0) return;  // t was ticked by the loss kernel and nothing is left to update
    TH(th_adam_step_guarded(Device::ctx(), fp_.p_arena->d, fp_.g_arena->d, m_->d, v_->d, fp_.d_offsets(), fp_.d_has_grad(),
                            (int)fp_.params.size(), fp_.total, d_tick(), state_->d + 2, beta1_, beta2_, eps_, wd_,
                            external_tick_ ? 1 : 0, step_guard_));
}

bool Adam::step_reduced(const Communicator &comm) {
    if (!comm.is_p2p() || !comm.fuse_adam || external_tick_ || carry_deferred_ || !deferred_.empty()) return false;
    for (char f : fused_)
        if (f) return false;
    fp_.sync_mask(&fused_);   // grad-less tensors are skipped entirely (Q8); the mask is the same on every rank
    TH(th_allreduce_adam(comm.handle(), Device::ctx(), fp_.g_arena->d, (size_t)fp_.total, 1.0f / (float)comm.n_ranks, fp_.p_arena->d, m_->d,
                         v_->d, fp_.d_offsets(), fp_.d_has_grad(), (int)fp_.params.size(), d_tick(), state_->d + 2, beta1_, beta2_, eps_, wd_, 0));
    return true;
}

int Adam::t() const {
    int32_t t = 0;
    TH(th_memcpy_d2h(Device::ctx(), &t, state_->d, sizeof(t)));
    return t;
}

static std::vector<float> gather_unpadded(const FlatParams &fp, const float *arena) {
    std::vector<float> all((size_t)fp.total), out;
    TH(th_memcpy_d2h(Device::ctx(), all.data(), arena, all.size() * sizeof(float)));
    for (size_t i = 0; i < fp.params.size(); ++i)
        out.insert(out.end(), all.begin() + fp.offsets[i], all.begin() + fp.offsets[i] + fp.params[i].len());
    return out;
}
static void scatter_padded(const FlatParams &fp, const std::vector<float> &src, float *arena) {
    std::vector<float> all((size_t)fp.total, 0.f);
    size_t off = 0;
    for (size_t i = 0; i < fp.params.size(); ++i) {
        std::copy(src.begin() + (long)off, src.begin() + (long)(off + fp.params[i].len()), all.begin() + fp.offsets[i]);
        off += fp.params[i].len();
    }
    TH(th_memcpy_h2d(Device::ctx(), arena, all.data(), all.size() * sizeof(float)));
}
void Adam::load_state(int t, const std::vector<float> &m, const std::vector<float> &v) {
    size_t n = 0;
    for (const Tensor &p : fp_.params) n += p.len();
    TAPER_ASSERT(m.size() == n && v.size() == n, "Adam::load_state: moment vectors must cover every parameter");
    scatter_padded(fp_, m, m_->d);
    scatter_padded(fp_, v, v_->d);
    const int32_t st[2] = {t, 0};
    TH(th_memcpy_h2d(Device::ctx(), state_->d, st, sizeof st));
}
std::vector<float> Adam::m() const { return gather_unpadded(fp_, m_->d); }
std::vector<float> Adam::v() const { return gather_unpadded(fp_, v_->d); }

}  // namespace taper

// modules.cpp -- losses (incl. the fused classifier / chain heads) and layers of the host mirror (src/loss.rs, src/nn.rs).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "nn_internal.h"

namespace taper {
// ---------------------------------------------------------------- loss
Tensor log_softmax(const Tensor &x, int dim) {  // loss.rs:101-126
    const int nd = (int)x.shape().size();
    if (dim < 0) dim += nd;
    TAPER_ASSERT(dim == nd - 1, "Only last-dim log_softmax is supported");
    TAPER_ASSERT(nd == 2, "log_softmax: 2-D input expected");
    if (!x.get_requires_grad()) {  // one fused kernel when no tape is needed
        Tensor out = Tensor::empty(x.shape());
        TH(th_log_softmax_fwd(Device::ctx(), x.dptr(), out.dptr(), (int)x.shape()[0], (int)x.shape()[1]));
        return out;
    }
    // with autograd: the reference's own chain of differentiable primitives
    Tensor mx = x.max(dim).first;
    Tensor shifted = x.sub_broadcast_rows(mx);
    Tensor log_sum = shifted.exp().sum(dim, true).log();
    return shifted.sub_broadcast_rows(log_sum);
}

Tensor softmax(const Tensor &x, int dim) { return log_softmax(x, dim).exp(); }  // Q12

Tensor cross_entropy_loss(const Tensor &logits, const Tensor &targets, Tensor *n_correct_out, const StepLogSink *log) {  // loss.rs:136-195
    const Shape &ts = targets.shape();
    TAPER_ASSERT(ts.size() == 1 || (ts.size() == 2 && ts[1] == 1), "Targets must be [B] or [B,1]");
    TAPER_ASSERT(logits.shape().size() == 2, "Logits must be [B,C]");
    TAPER_ASSERT(logits.shape()[0] == ts[0], "Batch sizes must match");
    const int b = (int)logits.shape()[0], c = (int)logits.shape()[1];
    th_ctx *ctx = Device::ctx();
    const bool need_grad = logits.get_requires_grad();
    Tensor logp = Tensor::empty(logits.shape());
    Tensor loss = Tensor::empty({1});
    // the gradient for an upstream grad of exactly 1 comes out of the forward kernel
    std::shared_ptr<Buffer> dunit = need_grad ? Buffer::alloc(logits.len()) : nullptr;
    float *nc = nullptr;
    if (n_correct_out) {
        *n_correct_out = Tensor::empty({1});
        nc = n_correct_out->dptr();
    }
    TH(th_softmax_xent_fwd(ctx, logits.dptr(), targets.dptr(), b, c, logp.dptr(), loss.dptr(), nullptr, nc,
                           dunit ? dunit->d : nullptr, log ? log->d_metrics : nullptr, log ? log->capacity : 0,
                           log ? log->d_state : nullptr, log ? log->advance : 0, log ? log->d_adam_tick : nullptr));
    if (need_grad) {
        loss.set_requires_grad(true);
        Tensor lg = logits, lp = logp, t = targets, out = loss;
        Tape::push(loss, true, [lg, lp, t, out, dunit, b, c]() {
            if (!out.has_grad()) return;
            if (out.grad_->shared_const && !lg.has_grad() && !lg.grad_->buf_is_arena) {
                // loss.backward() on the root: upstream grad is the constant 1 and logits.grad is
                // None -> (softmax - onehot)/B from the forward kernel IS the gradient: adopt it
                lg.grad_->buf = dunit;
                lg.grad_->has = true;
                lg.grad_->known_zero = false;
                lg.grad_->shared_const = false;
                return;
            }
            bool none;
            float *g = lg.grad_for_write(&none);
            TH(th_softmax_xent_bwd(Device::ctx(), lp.dptr(), t.dptr(), out.grad_dptr(), b, c, g, none ? 0 : 1));
        });
    }
    return loss;
}

bool linear_cross_entropy_supported(const Tensor &h, const Tensor &weight) {
    // batch <= 64: one workgroup, one launch; above: up to 256 workgroups + a finish pass
    return h.shape().size() == 2 && weight.shape().size() == 2 && h.shape()[1] == weight.shape()[1] && weight.shape()[0] <= 16 &&
           weight.shape()[1] <= 256 && h.shape()[0] >= 1 && h.shape()[0] <= (1u << 22);
}

Tensor linear_cross_entropy(const Tensor &h, const Tensor &w, const Tensor &bias, const Tensor &targets, Tensor *n_correct_out,
                            const StepLogSink *log) {  // nn.rs:54-60 + loss.rs:136-195 in one launch
    TAPER_ASSERT(linear_cross_entropy_supported(h, w), "linear_cross_entropy: unsupported shapes");
    TAPER_ASSERT(targets.shape()[0] == h.shape()[0], "Batch sizes must match");
    const int b = (int)h.shape()[0], k = (int)h.shape()[1], c = (int)w.shape()[0];
    th_ctx *ctx = Device::ctx();
    Tensor loss = Tensor::empty({1});
    float *nc = nullptr;
    if (n_correct_out) {
        *n_correct_out = Tensor::empty({1});
        nc = n_correct_out->dptr();
    }
    const bool h_grad = h.get_requires_grad();
    const bool w_grad = w.get_requires_grad() && !w.has_grad();
    const bool b_grad = bias.defined() && bias.get_requires_grad() && !bias.has_grad();
    TAPER_ASSERT(!w.get_requires_grad() || w_grad, "linear_cross_entropy: weight already has a gradient (accumulation unsupported)");
    TAPER_ASSERT(!(bias.defined() && bias.get_requires_grad()) || b_grad, "linear_cross_entropy: bias already has a gradient");
    // gradient destinations: parameter grads go straight into their (currently None) slots
    std::shared_ptr<Buffer> dh = h_grad ? Buffer::alloc(h.len()) : nullptr;
    float *dw = nullptr, *db = nullptr;
    if (w_grad) {
        if (!w.grad_->buf) w.grad_->buf = Buffer::alloc(w.len());
        dw = w.grad_->buf->d;
        w.grad_->known_zero = false;
    }
    if (b_grad) {
        if (!bias.grad_->buf) bias.grad_->buf = Buffer::alloc(bias.len());
        db = bias.grad_->buf->d;
        bias.grad_->known_zero = false;
    }
    // The head's own W / b updates ride in the next backward launch (Adam::defer_for): applied in
    // this single-workgroup kernel they cost 2.4 us of extra round trips, there they are free.
    const th_adam_fuse *pw = nullptr, *pb = nullptr;
    if (Adam *fa = FusedAdamScope::active()) {
        if (w_grad) fa->defer_for(w);
        if (b_grad) fa->defer_for(bias);
    }
    // h straight out of a fused Linear+ReLU: dH leaves the kernel already masked (the large-batch backward of that
    // layer would otherwise spend a pass over [B, hidden] on it)
    const bool mask_dh = dh && h.grad_->relu_output;
    TH(th_linear_xent_head_masked(ctx, h.dptr(), w.dptr(), bias.defined() ? bias.dptr() : nullptr, targets.dptr(), b, k, c, nullptr,
                                  loss.dptr(), nc, dh ? dh->d : nullptr, dw, db, log ? log->d_metrics : nullptr, log ? log->capacity : 0,
                                  log ? log->d_state : nullptr, log ? log->advance : 0, log ? log->d_adam_tick : nullptr, pw, pb,
                                  mask_dh ? 1 : 0));
    if (h_grad || w_grad || b_grad) {
        loss.set_requires_grad(true);
        Tensor hh = h, ww = w, bb = bias, out = loss;
        Tape::push(loss, true, [hh, ww, bb, out, dh, w_grad, b_grad, mask_dh]() {
            if (!out.has_grad()) return;
            // the gradients were produced by the forward launch for an upstream grad of exactly 1
            TAPER_ASSERT(out.grad_->shared_const, "linear_cross_entropy: only loss.backward() from the root is supported");
            if (dh) {
                TAPER_ASSERT(!hh.has_grad() && !hh.grad_->buf_is_arena, "linear_cross_entropy: input already has a gradient");
                hh.grad_->buf = dh;
                hh.grad_->has = true;
                hh.grad_->shared_const = false;
                hh.grad_->premasked = mask_dh;
            }
            if (w_grad) ww.grad_->has = true;
            if (b_grad) bb.grad_->has = true;
        });
    }
    return loss;
}

bool linear_cross_entropy_wide_supported(const Tensor &h, const Tensor &w, const Tensor &bias) {
    if (h.shape().size() != 2 || w.shape().size() != 2 || h.shape()[1] != w.shape()[1]) return false;
    if (w.shape()[1] <= 256 || w.shape()[0] > 16 || h.shape()[0] < 1 || h.shape()[0] > 4096) return false;
    if (!w.get_requires_grad() || w.has_grad()) return false;
    if (bias.defined() && (!bias.get_requires_grad() || bias.has_grad())) return false;
    return !(h.get_requires_grad() && (h.has_grad() || h.grad_->buf_is_arena));   // dX is written, never accumulated
}

Tensor linear_cross_entropy_wide(const Tensor &h, const Tensor &w, const Tensor &bias, const Tensor &targets, Tensor *n_correct_out,
                                 const StepLogSink *log) {   // nn.rs:54-60 + loss.rs:136-195 + the Linear's backward closures
    TAPER_ASSERT(linear_cross_entropy_wide_supported(h, w, bias), "linear_cross_entropy_wide: unsupported shapes / gradient state");
    TAPER_ASSERT(targets.shape()[0] == h.shape()[0], "Batch sizes must match");
    const int b = (int)h.shape()[0], k = (int)h.shape()[1], c = (int)w.shape()[0];
    Tensor loss = Tensor::empty({1});
    float *nc = nullptr;
    if (n_correct_out) {
        *n_correct_out = Tensor::empty({1});
        nc = n_correct_out->dptr();
    }
    auto slot = [](const Tensor &p) -> float * {
        if (!p.defined()) return nullptr;
        if (!p.grad_->buf) p.grad_->buf = Buffer::alloc(p.len());
        p.grad_->known_zero = false;
        return p.grad_->buf->d;
    };
    float *dw = slot(w), *db = slot(bias);
    // the input is the flattened output of a bias-only Conv2dReLU + pool (Trainer step): it asked for column sums of dX * [x > 0], not for dX
    const bool colsum_mode = h.get_requires_grad() && h.grad_->wants_colsum && PoolBiasScope::active();
    std::shared_ptr<Buffer> dh = (h.get_requires_grad() && !colsum_mode) ? Buffer::alloc(h.len()) : nullptr;
    std::shared_ptr<Buffer> cs = colsum_mode ? Buffer::alloc((size_t)k) : nullptr;
    // With the Adam fusion on, the whole tail of the step fits these two launches: no dX is stored, so every workgroup owns its columns of W
    // (Adam in its epilogue), the lead owns b, and the last workgroup to arrive finishes the conv bias from the column sums and ticks t.
    Adam *fa = FusedAdamScope::active();
    const std::shared_ptr<GradSlot> cbs = colsum_mode ? h.grad_->colsum_bias : nullptr;
    // (measured on the simple CNN at batch 256: the head grows from 11.9 to 22.5 us -- Adam's p / m / v round trip behind the dW reduction in
    // every workgroup, an agent-scope fence per workgroup, the last arriver's serial finish -- against the 5.0 us finishing launch it replaces:
    // step 57.2 -> 64.1 us.  Off unless TAPER_WIDE_FUSED=1.)
    static const bool wide_fused = std::getenv("TAPER_WIDE_FUSED") && std::getenv("TAPER_WIDE_FUSED")[0] == '1';
    const bool full = wide_fused && colsum_mode && fa && log && log->d_adam_tick && cbs && cbs->buf && cbs->buf_is_arena && !cbs->has &&
                      (long)h.grad_->colsum_c * h.grad_->colsum_hw == k;
    bool fused_done = false;
    if (full) {
        th_wide_fuse f{};
        const bool ok = fa->fuse_for(w, &f.w) && (!bias.defined() || fa->fuse_for(bias, &f.b)) && fa->fuse_for_slot(cbs, &f.conv_b);
        TAPER_ASSERT(ok, "linear_cross_entropy_wide: a parameter of the fused tail is not held by the active optimizer");
        f.d_conv_gb = cbs->buf->d;
        f.conv_c = h.grad_->colsum_c;
        f.conv_hw = h.grad_->colsum_hw;
        TH(th_linear_xent_wide_fused(Device::ctx(), h.dptr(), w.dptr(), bias.defined() ? bias.dptr() : nullptr, targets.dptr(), b, k, c, loss.dptr(),
                                     nc, dw, db, log->d_metrics, log->capacity, log->d_state, log->advance, log->d_adam_tick, cs->d, &f));
        cbs->has = true;
        cbs->known_zero = false;
        fused_done = true;
    } else {
        TH(th_linear_xent_wide_ex(Device::ctx(), h.dptr(), w.dptr(), bias.defined() ? bias.dptr() : nullptr, targets.dptr(), b, k, c, loss.dptr(),
                                  nc, dh ? dh->d : nullptr, dw, db, log ? log->d_metrics : nullptr, log ? log->capacity : 0,
                                  log ? log->d_state : nullptr, log ? log->advance : 0, log ? log->d_adam_tick : nullptr, cs ? cs->d : nullptr));
        if (fa) {   // complete gradients; every workgroup of the launch read W
            fa->defer_for(w);
            if (bias.defined()) fa->defer_for(bias);
        }
    }
    loss.set_requires_grad(true);
    Tensor hh = h, ww = w, bb = bias, out = loss;
    Tape::push(loss, true, [hh, ww, bb, out, dh, cs, fused_done]() {
        if (!out.has_grad()) return;
        TAPER_ASSERT(out.grad_->shared_const, "linear_cross_entropy_wide: only loss.backward() from the root is supported");
        if (fused_done) hh.grad_->colsum_done = true;
        else if (cs) hh.grad_->colsum = cs;
        if (dh) {
            TAPER_ASSERT(!hh.has_grad() && !hh.grad_->buf_is_arena, "linear_cross_entropy_wide: input already has a gradient");
            hh.grad_->buf = dh;
            hh.grad_->has = true;
            hh.grad_->shared_const = false;
        }
        ww.grad_->has = true;
        if (bb.defined()) bb.grad_->has = true;
    });
    return loss;
}

bool conv_chain_head_supported(const Tensor &x, const std::vector<ConvStage> &stages, const Tensor &w, const Tensor &bias) {
    if (stages.empty() || stages.back().post != TH_CHAIN_MAXPOOL2 || x.shape().size() != 4 || w.shape().size() != 2) return false;
    if (!w.get_requires_grad() || w.has_grad()) return false;
    if (bias.defined() && (!bias.get_requires_grad() || bias.has_grad() || bias.shape() != Shape{w.shape()[0]})) return false;
    const Tensor &cb = stages.back().bias;
    if (cb.defined() && cb.get_requires_grad() && cb.has_grad()) return false;   // gradients are written, never accumulated
    size_t h = x.shape()[2], wd = x.shape()[3];
    for (const auto &st : stages)
        if (st.post == TH_CHAIN_MAXPOOL2) { h /= 2; wd /= 2; }
    if (w.shape()[1] != stages.back().weight.shape()[0] * h * wd) return false;
    return x.conv_chain_head_supported(stages, (int)w.shape()[0]) != 0;
}

Tensor conv_chain_head_cross_entropy(const Tensor &x, const std::vector<ConvStage> &stages, const Tensor &w, const Tensor &bias,
                                     const Tensor &targets, Tensor *n_correct_out, const StepLogSink *log) {
    // nn.rs:149-151 (conv rows, Flatten, Linear) + loss.rs:136-195 + the backward closures of the Linear (ops.rs:238-294,
    // tensor.rs:574-587, 674-694) and of the last Conv2dReLU's bias behind its pool (tensor.rs:1496-1519, 2017-2024, ops.rs:358-369)
    TAPER_ASSERT(conv_chain_head_supported(x, stages, w, bias), "conv_chain_head_cross_entropy: unsupported stages / gradient state");
    TAPER_ASSERT(targets.shape()[0] == x.shape()[0], "Batch sizes must match");
    th_ctx *ctx = Device::ctx();
    Adam *fa = FusedAdamScope::active();
    if (fa && fa->has_deferred()) fa->flush_deferred();   // updates an earlier (other) step form left behind: with their own counter
    const int n = (int)x.shape()[0], classes = (int)w.shape()[0], k = (int)w.shape()[1];
    const Tensor &cbias = stages.back().bias;
    const int c_last = (int)stages.back().weight.shape()[0];
    const bool cb_grad = cbias.defined() && cbias.get_requires_grad();
    auto dl = Buffer::alloc((size_t)n * 16), rowstat = Buffer::alloc((size_t)n * 2);
    std::shared_ptr<Buffer> cbpart = cb_grad ? Buffer::alloc((size_t)n * c_last) : nullptr;
    // launch 1: the conv rows with the classifier's row-parallel part in the last epilogue; opens the optimizer step (optim.rs:84)
    th_chain_head head{w.dptr(), bias.defined() ? bias.dptr() : nullptr, targets.dptr(), classes, dl->d, rowstat->d,
                       cbpart ? cbpart->d : nullptr, fa ? fa->d_tick() : nullptr};
    Tensor map = x.conv_chain_head(stages, head);
    // launch 2: the sums over the batch, Adam in the epilogues (nothing there reads a parameter)
    Tensor loss = Tensor::empty({1});
    float *nc = nullptr;
    if (n_correct_out) {
        *n_correct_out = Tensor::empty({1});
        nc = n_correct_out->dptr();
    }
    auto slot = [](const Tensor &p) -> float * {
        if (!p.defined()) return nullptr;
        if (!p.grad_->buf) p.grad_->buf = Buffer::alloc(p.len());
        p.grad_->known_zero = false;
        return p.grad_->buf->d;
    };
    float *dw = slot(w), *db = slot(bias), *gcb = cb_grad ? slot(cbias) : nullptr;
    th_adam_fuse wf{}, bf{}, cf{};
    const bool fw = fa && fa->fuse_for(w, &wf), fb = fa && bias.defined() && fa->fuse_for(bias, &bf), fc = fa && cb_grad && fa->fuse_for(cbias, &cf);
    const Communicator *xc = TailExchangeScope::active();
    if (xc && xc->n_ranks > 1) {
        // data parallel: the launch exchanges every finished sum with the peers; its epilogues apply the mean
        TAPER_ASSERT(fa && xc->wide_exchange_ok(n, k, classes, c_last), "conv_chain_head_cross_entropy: the in-launch exchange does not cover this step");
        TH(th_wide_head_grads_dp(xc->handle(), ctx, map.dptr(), dl->d, rowstat->d, cbpart ? cbpart->d : nullptr, n, k, classes, c_last, dw, db, gcb,
                                 loss.dptr(), nc, log ? log->d_metrics : nullptr, log ? log->capacity : 0, log ? log->d_state : nullptr,
                                 log ? log->advance : 0, fw ? &wf : nullptr, fb ? &bf : nullptr, fc ? &cf : nullptr, fa->d_tick()));
    } else
    TH(th_wide_head_grads(ctx, map.dptr(), dl->d, rowstat->d, cbpart ? cbpart->d : nullptr, n, k, classes, c_last, dw, db, gcb, loss.dptr(), nc,
                          log ? log->d_metrics : nullptr, log ? log->capacity : 0, log ? log->d_state : nullptr, log ? log->advance : 0,
                          fw ? &wf : nullptr, fb ? &bf : nullptr, fc ? &cf : nullptr));
    loss.set_requires_grad(true);
    Tensor ww = w, bb = bias, cc = cb_grad ? cbias : Tensor(), out = loss;
    Tape::push(loss, true, [ww, bb, cc, out]() {
        if (!out.has_grad()) return;
        // the gradients were produced by the forward launches for an upstream grad of exactly 1
        TAPER_ASSERT(out.grad_->shared_const, "conv_chain_head_cross_entropy: only loss.backward() from the root is supported");
        ww.grad_->has = true;
        if (bb.defined()) bb.grad_->has = true;
        if (cc.defined()) cc.grad_->has = true;
    });
    return loss;
}

bool mlp_tail_supported(const Tensor &x, const Tensor &w1, const Tensor &b1, const Tensor &w2, const Tensor &b2) {
    if (x.shape().size() != 2 || w1.shape().size() != 2 || w2.shape().size() != 2) return false;
    if (x.shape()[1] != w1.shape()[1] || w2.shape()[1] != w1.shape()[0]) return false;
    if (x.get_requires_grad() && (x.has_grad() || x.grad_->buf_is_arena)) return false;   // dX is written, never accumulated
    if (!w1.get_requires_grad() || !w2.get_requires_grad() || w1.has_grad() || w2.has_grad()) return false;
    for (const Tensor *b : {&b1, &b2})
        if (b->defined() && (!b->get_requires_grad() || b->has_grad())) return false;
    return th_mlp_tail_supported((int)x.shape()[0], (int)x.shape()[1], (int)w1.shape()[0], (int)w2.shape()[0],
                                 x.get_requires_grad() ? 1 : 0) != 0;
}

Tensor mlp_tail_cross_entropy(const Tensor &x, const Tensor &w1, const Tensor &b1, const Tensor &w2, const Tensor &b2,
                              const Tensor &targets, Tensor *n_correct_out, const StepLogSink *log) {
    TAPER_ASSERT(mlp_tail_supported(x, w1, b1, w2, b2), "mlp_tail_cross_entropy: unsupported shapes / gradient state");
    TAPER_ASSERT(targets.shape()[0] == x.shape()[0], "Batch sizes must match");
    const int b = (int)x.shape()[0], in_f = (int)x.shape()[1], hid = (int)w1.shape()[0], c = (int)w2.shape()[0];
    th_ctx *ctx = Device::ctx();
    Adam *fa = FusedAdamScope::active();
    // launch 1 (nn.rs:54-60 + activation.rs:10-12): H = relu(x . W1^T + b1); the updates the PREVIOUS step
    // deferred (its W2 / b2: read by every workgroup of its tail launch) ride here with that step's counter,
    // then the counter opens this step (optim.rs:84)
    th_adam_slice carried[TH_MAX_ADAM_SLICES];
    const int n_carried = fa ? fa->take_deferred(w1.dptr(), carried) : 0;
    if (fa && fa->has_deferred()) fa->flush_deferred();   // more than one launch can carry (never for a plain MLP): old counter too
    Tensor h = Tensor::empty({(size_t)b, (size_t)hid});
    TH(th_linear_fwd_ex(ctx, x.dptr(), w1.dptr(), b1.defined() ? b1.dptr() : nullptr, h.dptr(), b, in_f, hid, 1, carried, n_carried,
                        fa ? fa->d_tick() : nullptr));
    // launch 2: everything else
    Tensor loss = Tensor::empty({1});
    float *nc = nullptr;
    if (n_correct_out) {
        *n_correct_out = Tensor::empty({1});
        nc = n_correct_out->dptr();
    }
    auto slot = [](const Tensor &p) -> float * {
        if (!p.defined()) return nullptr;
        if (!p.grad_->buf) p.grad_->buf = Buffer::alloc(p.len());
        p.grad_->known_zero = false;
        return p.grad_->buf->d;
    };
    float *dw1 = slot(w1), *db1 = slot(b1), *dw2 = slot(w2), *db2 = slot(b2);
    // a hidden layer that is not the first also hands dX down: the launch then reads W1, whose update is deferred like W2's
    const bool need_dx = x.get_requires_grad();
    std::shared_ptr<Buffer> dx = need_dx ? Buffer::alloc(x.len()) : nullptr;
    th_adam_fuse wf{}, bf{};
    const th_adam_fuse *pw = nullptr, *pb = nullptr;
    if (fa) {
        if (!need_dx && fa->fuse_for(w1, &wf)) pw = &wf;
        if (b1.defined() && fa->fuse_for(b1, &bf)) pb = &bf;
    }
    const Communicator *xc = TailExchangeScope::active();
    if (xc && xc->n_ranks > 1) {
        // data parallel: the launch exchanges every finished slice with the peers; its epilogues apply the mean
        TAPER_ASSERT(fa && !need_dx && xc->tail_exchange_ok(b, in_f, hid, c), "mlp_tail_cross_entropy: the in-launch exchange does not cover this step");
        TH(th_mlp_tail_dp(xc->handle(), ctx, x.dptr(), h.dptr(), w2.dptr(), b2.defined() ? b2.dptr() : nullptr, targets.dptr(), b, in_f, hid, c,
                          loss.dptr(), nc, dw1, db1, dw2, db2, log ? log->d_metrics : nullptr, log ? log->capacity : 0,
                          log ? log->d_state : nullptr, log ? log->advance : 0, pw, pb, fa->d_tick()));
    } else
    TH(th_mlp_tail(ctx, x.dptr(), h.dptr(), w2.dptr(), b2.defined() ? b2.dptr() : nullptr, targets.dptr(), b, in_f, hid, c, loss.dptr(),
                   nc, dw1, db1, dw2, db2, need_dx ? w1.dptr() : nullptr, dx ? dx->d : nullptr, log ? log->d_metrics : nullptr,
                   log ? log->capacity : 0, log ? log->d_state : nullptr, log ? log->advance : 0, pw, pb));
    if (fa) {   // complete gradients of parameters this launch read: updated by the next launch that does not read them
        if (need_dx) fa->defer_for(w1);
        fa->defer_for(w2);
        if (b2.defined()) fa->defer_for(b2);
    }
    loss.set_requires_grad(true);
    Tensor p1 = w1, p2 = b1, p3 = w2, p4 = b2, out = loss, keep = h, xin = x;
    Tape::push(loss, true, [p1, p2, p3, p4, out, keep, xin, dx]() {
        if (!out.has_grad()) return;
        // the gradients were produced by the forward launches for an upstream grad of exactly 1
        TAPER_ASSERT(out.grad_->shared_const, "mlp_tail_cross_entropy: only loss.backward() from the root is supported");
        if (dx) {
            TAPER_ASSERT(!xin.has_grad() && !xin.grad_->buf_is_arena, "mlp_tail_cross_entropy: input already has a gradient");
            xin.grad_->buf = dx;
            xin.grad_->has = true;
            xin.grad_->shared_const = false;
        }
        for (const Tensor *p : {&p1, &p2, &p3, &p4})
            if (p->defined()) p->grad_->has = true;
    });
    return loss;
}

// TAPER_MLP2_MIN_BATCH: from this batch on a Linear + ReLU + Linear classifier steps through th_mlp2_xent (default 480: the crossover
// measured with bench.py --batch B, both settings -- launch-per-layer forms 24.9 us at 448 rows, 30.8 at 512; th_mlp2_xent 25.5 / 28.3)
size_t mlp2_min_batch() {
    static const size_t v = [] { const char *e = std::getenv("TAPER_MLP2_MIN_BATCH"); return e ? (size_t)std::max(32, atoi(e)) : (size_t)480; }();
    return v;
}

// What th_mlp2_xent / th_mlp2_xent_deep need of the parameters themselves, whatever the rows: gradients are written, never accumulated
// (every slot empty), every parameter trains, shapes chain, 16-byte aligned storage.  `w`, `b`: 2 or 3 Linear layers, first to last.
bool mlp2_params_ok(const std::vector<Tensor> &w, const std::vector<Tensor> &b) {
    if ((w.size() != 2 && w.size() != 3) || b.size() != w.size()) return false;
    for (size_t l = 0; l < w.size(); ++l) {
        if (w[l].shape().size() != 2 || (l > 0 && w[l].shape()[1] != w[l - 1].shape()[0])) return false;
        if (!w[l].get_requires_grad() || w[l].has_grad() || ((uintptr_t)w[l].dptr() & 15) != 0) return false;
        if (b[l].defined() && (!b[l].get_requires_grad() || b[l].has_grad() || b[l].shape() != Shape{w[l].shape()[0]})) return false;
    }
    return true;
}

bool mlp2_shapes_ok(size_t batch, const std::vector<Tensor> &w, int64_t n_rows) {
    const int in_f = (int)w[0].shape()[1], h1 = (int)w[0].shape()[0];
    if (w.size() == 2) return th_mlp2_xent_supported((int)batch, in_f, h1, (int)w[1].shape()[0], n_rows) != 0;
    return th_mlp2_xent_deep_supported((int)batch, in_f, h1, (int)w[1].shape()[0], (int)w[2].shape()[0], n_rows) != 0;
}

bool mlp2_supported(const th_row_source &src, size_t batch, const std::vector<Tensor> &w, const std::vector<Tensor> &b) {
    if (!mlp2_params_ok(w, b)) return false;
    if (((uintptr_t)src.d_rows & 15) != 0) return false;
    if (src.d_indices && src.n_indices < (int64_t)batch) return false;
    return mlp2_shapes_ok(batch, w, src.n_rows);
}

bool mlp2_supported(const th_row_source &src, size_t batch, const Tensor &w1, const Tensor &b1, const Tensor &w2, const Tensor &b2) {
    return mlp2_supported(src, batch, std::vector<Tensor>{w1, w2}, std::vector<Tensor>{b1, b2});
}

Tensor mlp2_cross_entropy(const th_row_source &src, size_t batch, const std::vector<Tensor> &w, const std::vector<Tensor> &b,
                          Tensor *n_correct_out, const StepLogSink *log) {
    // nn.rs:54-60, activation.rs:10-12, loss.rs:136-195 and the backward closures of the Linear layers and the ReLU nodes
    // (ops.rs:238-294, 358-369; tensor.rs:574-587, 674-694); data/mnist.rs:277-310 for the rows
    TAPER_ASSERT(mlp2_supported(src, batch, w, b), "mlp2_cross_entropy: unsupported shapes / gradient state");
    th_ctx *ctx = Device::ctx();
    Adam *fa = FusedAdamScope::active();
    if (fa && fa->has_deferred()) fa->flush_deferred();   // updates an earlier (other) step form left behind: with their own counter
    Tensor loss = Tensor::empty({1});
    float *nc = nullptr;
    if (n_correct_out) {
        *n_correct_out = Tensor::empty({1});
        nc = n_correct_out->dptr();
    }
    auto slot = [](const Tensor &p) -> float * {
        if (!p.defined()) return nullptr;
        if (!p.grad_->buf) p.grad_->buf = Buffer::alloc(p.len());
        p.grad_->known_zero = false;
        return p.grad_->buf->d;
    };
    // the finish launch holds every complete gradient and no launch of the step reads a parameter after it: every update rides there
    const size_t nl = w.size();
    th_adam_fuse fw[3], fb[3];
    th_mlp3_layer L[3];
    for (size_t l = 0; l < nl; ++l) {
        L[l].d_w = w[l].dptr();
        L[l].d_b = b[l].defined() ? b[l].dptr() : nullptr;
        L[l].d_dw = slot(w[l]);
        L[l].d_db = slot(b[l]);
        L[l].w_fuse = (fa && fa->fuse_for(w[l], &fw[l])) ? &fw[l] : nullptr;
        L[l].b_fuse = (fa && b[l].defined() && fa->fuse_for(b[l], &fb[l])) ? &fb[l] : nullptr;
        L[l].out_features = (int)w[l].shape()[0];
    }
    if (nl == 2)
        TH(th_mlp2_xent(ctx, &src, (int)batch, (int)w[0].shape()[1], L[0].out_features, L[1].out_features, L[0].d_w, L[0].d_b, L[1].d_w, L[1].d_b,
                        L[0].d_dw, L[0].d_db, L[1].d_dw, L[1].d_db, loss.dptr(), nc, log ? log->d_metrics : nullptr, log ? log->capacity : 0,
                        log ? log->d_state : nullptr, log ? log->advance : 0, fa ? fa->d_tick() : nullptr, L[0].w_fuse, L[0].b_fuse, L[1].w_fuse,
                        L[1].b_fuse));
    else
        TH(th_mlp2_xent_deep(ctx, &src, (int)batch, (int)w[0].shape()[1], L, loss.dptr(), nc, log ? log->d_metrics : nullptr, log ? log->capacity : 0,
                             log ? log->d_state : nullptr, log ? log->advance : 0, fa ? fa->d_tick() : nullptr));
    loss.set_requires_grad(true);
    std::vector<Tensor> ps;
    for (size_t l = 0; l < nl; ++l) {
        ps.push_back(w[l]);
        if (b[l].defined()) ps.push_back(b[l]);
    }
    Tensor out = loss;
    Tape::push(loss, true, [ps, out]() {
        if (!out.has_grad()) return;
        // the gradients were produced by the forward launches for an upstream grad of exactly 1
        TAPER_ASSERT(out.grad_->shared_const, "mlp2_cross_entropy: only loss.backward() from the root is supported");
        for (const Tensor &p : ps) p.grad_->has = true;
    });
    return loss;
}

Tensor mlp2_cross_entropy(const th_row_source &src, size_t batch, const Tensor &w1, const Tensor &b1, const Tensor &w2, const Tensor &b2,
                          Tensor *n_correct_out, const StepLogSink *log) {
    return mlp2_cross_entropy(src, batch, std::vector<Tensor>{w1, w2}, std::vector<Tensor>{b1, b2}, n_correct_out, log);
}

// Linear + ReLU, Linear + ReLU, Linear, softmax cross-entropy -- the classifier of examples/train_mnist_cnn.rs:53-61 and the whole model of
// examples/train_mnist.rs:40-48 -- forward and backward in two launches (th_mlp3_xent): a row-parallel one (forward, loss terms, the
// gradients of the activations down to dX) and one for every parameter gradient with Adam in the epilogues.  No launch reads a
// parameter another one of the step updates, so nothing is deferred; the first launch opens the optimizer step.
bool mlp3_supported(const Tensor &x, const Tensor (&w)[3], const Tensor (&b)[3]) {
    if (x.shape().size() != 2) return false;
    size_t in_f = x.shape()[1];
    for (int l = 0; l < 3; ++l) {
        if (w[l].shape().size() != 2 || w[l].shape()[1] != in_f) return false;
        if (!w[l].get_requires_grad() || w[l].has_grad()) return false;
        if (!b[l].defined() || b[l].shape() != Shape{w[l].shape()[0]} || !b[l].get_requires_grad() || b[l].has_grad()) return false;
        if (((uintptr_t)w[l].dptr() & 15) != 0) return false;
        in_f = w[l].shape()[0];
    }
    if (x.get_requires_grad() && (x.has_grad() || x.grad_->buf_is_arena)) return false;   // dX is written, never accumulated
    if (((uintptr_t)x.dptr() & 15) != 0) return false;
    return th_mlp3_supported((int)x.shape()[0], (int)x.shape()[1], (int)w[0].shape()[0], (int)w[1].shape()[0], (int)w[2].shape()[0]) != 0;
}

Tensor mlp3_cross_entropy(const Tensor &x, const Tensor (&w)[3], const Tensor (&b)[3], const Tensor &targets, Tensor *n_correct_out,
                          const StepLogSink *log) {
    TAPER_ASSERT(mlp3_supported(x, w, b), "mlp3_cross_entropy: unsupported shapes / gradient state");
    TAPER_ASSERT(targets.shape()[0] == x.shape()[0], "Batch sizes must match");
    th_ctx *ctx = Device::ctx();
    Adam *fa = FusedAdamScope::active();
    if (fa && fa->has_deferred()) fa->flush_deferred();   // updates an earlier (other) step form left behind: with their own counter
    Tensor loss = Tensor::empty({1});
    float *nc = nullptr;
    if (n_correct_out) {
        *n_correct_out = Tensor::empty({1});
        nc = n_correct_out->dptr();
    }
    auto slot = [](const Tensor &p) -> float * {
        if (!p.grad_->buf) p.grad_->buf = Buffer::alloc(p.len());
        p.grad_->known_zero = false;
        return p.grad_->buf->d;
    };
    th_mlp3_layer layers[3];
    th_adam_fuse wf[3], bf[3];
    for (int l = 0; l < 3; ++l) {
        layers[l] = th_mlp3_layer{w[l].dptr(), b[l].dptr(), slot(w[l]), slot(b[l]), nullptr, nullptr, (int)w[l].shape()[0]};
        if (fa) {
            if (fa->fuse_for(w[l], &wf[l])) layers[l].w_fuse = &wf[l];
            if (fa->fuse_for(b[l], &bf[l])) layers[l].b_fuse = &bf[l];
        }
    }
    const bool need_dx = x.get_requires_grad();
    std::shared_ptr<Buffer> dx = need_dx ? Buffer::alloc(x.len()) : nullptr;
    // x = the plane means of a bias-only Conv2dReLU + global average pool (GradSlot::gapfin_*): that conv's bias gradient and update
    // ride in the gradient launch -- it needs dX and the counts, nothing else
    th_mlp3_gap gap{};
    th_adam_fuse gf{};
    std::shared_ptr<GradSlot> gap_slot;
    if (need_dx && PoolBiasScope::active() && x.grad_->gapfin_cnt && x.grad_->gapfin_bias && !x.grad_->gapfin_bias->has &&
        x.grad_->gapfin_cnt->n == x.len()) {
        gap_slot = x.grad_->gapfin_bias;
        if (!gap_slot->buf) gap_slot->buf = Buffer::alloc(x.shape()[1]);
        gap_slot->known_zero = false;
        gap.d_cnt = x.grad_->gapfin_cnt->d;
        gap.d_gb = gap_slot->buf->d;
        gap.hw = x.grad_->gapfin_hw;
        if (fa && fa->fuse_for_slot(gap_slot, &gf)) gap.b_fuse = &gf;
    }
    TH(th_mlp3_xent(ctx, x.dptr(), targets.dptr(), (int)x.shape()[0], (int)x.shape()[1], layers, dx ? dx->d : nullptr, loss.dptr(), nc,
                    log ? log->d_metrics : nullptr, log ? log->capacity : 0, log ? log->d_state : nullptr, log ? log->advance : 0,
                    fa ? fa->d_tick() : nullptr, gap_slot ? &gap : nullptr));
    loss.set_requires_grad(true);
    std::vector<Tensor> params{w[0], b[0], w[1], b[1], w[2], b[2]};
    Tensor out = loss, xin = x;
    Tape::push(loss, true, [params, out, xin, dx, gap_slot]() {
        if (!out.has_grad()) return;
        // the gradients were produced by the forward launches for an upstream grad of exactly 1
        TAPER_ASSERT(out.grad_->shared_const, "mlp3_cross_entropy: only loss.backward() from the root is supported");
        if (dx) {
            TAPER_ASSERT(!xin.has_grad() && !xin.grad_->buf_is_arena, "mlp3_cross_entropy: input already has a gradient");
            xin.grad_->buf = dx;
            xin.grad_->has = true;
            xin.grad_->shared_const = false;
        }
        for (const Tensor &p : params) p.grad_->has = true;
        if (gap_slot) {
            gap_slot->has = true;
            xin.grad_->gapfin_done = true;
        }
    });
    return loss;
}

bool conv_chain_mlp3_supported(const Tensor &x, const std::vector<ConvStage> &stages, const Tensor (&w)[3], const Tensor (&b)[3]) {
    if (stages.empty() || stages.back().post != TH_CHAIN_GLOBAL_AVG || x.shape().size() != 4) return false;
    static const size_t max_batch = [] { const char *e = std::getenv("TAPER_CHAIN_MLP3_MAX_BATCH"); return e ? (size_t)std::max(0, atoi(e)) : (size_t)384; }();
    if (x.shape()[0] > max_batch) return false;   // (one image per workgroup: at 1 024 images the classifier's own row launch is the faster form, 393 against 403 us)
    size_t in_f = stages.back().weight.shape()[0];
    for (int l = 0; l < 3; ++l) {
        if (w[l].shape().size() != 2 || w[l].shape()[1] != in_f) return false;
        if (!w[l].get_requires_grad() || w[l].has_grad()) return false;
        if (!b[l].defined() || b[l].shape() != Shape{w[l].shape()[0]} || !b[l].get_requires_grad() || b[l].has_grad()) return false;
        if (((uintptr_t)w[l].dptr() & 15) != 0) return false;
        in_f = w[l].shape()[0];
    }
    const Tensor &cb = stages.back().bias;
    if (cb.defined() && cb.get_requires_grad() && cb.has_grad()) return false;   // gradients are written, never accumulated
    return x.conv_chain_mlp3_supported(stages, (int)w[0].shape()[0], (int)w[1].shape()[0], (int)w[2].shape()[0]) != 0;
}

Tensor conv_chain_mlp3_cross_entropy(const Tensor &x, const std::vector<ConvStage> &stages, const Tensor (&w)[3], const Tensor (&b)[3],
                                     const Tensor &targets, Tensor *n_correct_out, const StepLogSink *log) {
    // nn.rs:149-151 (conv rows, pool rows, Flatten, the classifier) + loss.rs:136-195 + the backward closures of the three Linear layers and
    // their ReLU nodes (ops.rs:238-294, 358-369; tensor.rs:574-587, 674-694) and of the last Conv2dReLU's bias behind the global average pool
    // (tensor.rs:1626-1628, ops.rs:358-369 through the positive counts): mlp3_cross_entropy on the chain's plane means, one launch less
    TAPER_ASSERT(conv_chain_mlp3_supported(x, stages, w, b), "conv_chain_mlp3_cross_entropy: unsupported stages / shapes / gradient state");
    TAPER_ASSERT(targets.shape()[0] == x.shape()[0], "Batch sizes must match");
    Adam *fa = FusedAdamScope::active();
    if (fa && fa->has_deferred()) fa->flush_deferred();   // updates an earlier (other) step form left behind: with their own counter
    const size_t n = x.shape()[0], c_last = stages.back().weight.shape()[0];
    Tensor loss = Tensor::empty({1});
    float *nc = nullptr;
    if (n_correct_out) {
        *n_correct_out = Tensor::empty({1});
        nc = n_correct_out->dptr();
    }
    auto slot = [](const Tensor &p) -> float * {
        if (!p.grad_->buf) p.grad_->buf = Buffer::alloc(p.len());
        p.grad_->known_zero = false;
        return p.grad_->buf->d;
    };
    th_mlp3_layer layers[3];
    th_adam_fuse wf[3], bf[3];
    for (int l = 0; l < 3; ++l) {
        layers[l] = th_mlp3_layer{w[l].dptr(), b[l].dptr(), slot(w[l]), slot(b[l]), nullptr, nullptr, (int)w[l].shape()[0]};
        if (fa) {
            if (fa->fuse_for(w[l], &wf[l])) layers[l].w_fuse = &wf[l];
            if (fa->fuse_for(b[l], &bf[l])) layers[l].b_fuse = &bf[l];
        }
    }
    // the last conv's bias (the only conv parameter the reference's tape reaches, quirk Q2): its gradient is formed by the gradient launch
    // from the plane means' gradient and the positive counts the chain launch leaves
    const Tensor &cbias = stages.back().bias;
    const bool cb_grad = cbias.defined() && cbias.get_requires_grad() && !NoGradScope::active();
    std::shared_ptr<Buffer> cnt = cb_grad ? Buffer::alloc(n * c_last) : nullptr, dx = cb_grad ? Buffer::alloc(n * c_last) : nullptr;
    th_mlp3_gap gap{};
    th_adam_fuse gf{};
    size_t hw = x.shape()[2] * x.shape()[3];
    for (const auto &st : stages)
        if (st.post == TH_CHAIN_MAXPOOL2) hw /= 4;
    if (cb_grad) {
        gap.d_cnt = cnt->d;
        gap.d_gb = slot(cbias);
        gap.hw = (int)hw;
        if (fa && fa->fuse_for(cbias, &gf)) gap.b_fuse = &gf;
    }
    Tensor means = x.conv_chain_mlp3(stages, cnt ? cnt->d : nullptr, targets.dptr(), layers, dx ? dx->d : nullptr, loss.dptr(), nc,
                                     log ? log->d_metrics : nullptr, log ? log->capacity : 0, log ? log->d_state : nullptr, log ? log->advance : 0,
                                     fa ? fa->d_tick() : nullptr, cb_grad ? &gap : nullptr);
    loss.set_requires_grad(true);
    std::vector<Tensor> params{w[0], b[0], w[1], b[1], w[2], b[2]};
    if (cb_grad) params.push_back(cbias);
    Tensor out = loss, keep = means;
    Tape::push(loss, true, [params, out, keep]() {
        if (!out.has_grad()) return;
        // the gradients were produced by the forward launches for an upstream grad of exactly 1
        TAPER_ASSERT(out.grad_->shared_const, "conv_chain_mlp3_cross_entropy: only loss.backward() from the root is supported");
        for (const Tensor &p : params) p.grad_->has = true;
    });
    return loss;
}

float accuracy(const Tensor &pred, const Tensor &targets) {  // loss.rs:271-290
    TAPER_ASSERT(pred.shape()[0] == targets.shape()[0], "Batch sizes must match");
    TAPER_ASSERT(pred.shape().size() == 2, "accuracy: predictions must be [B,C]");
    th_ctx *ctx = Device::ctx();
    const int b = (int)pred.shape()[0], c = (int)pred.shape()[1];
    Tensor am = Tensor::empty({(size_t)b}), cnt = Tensor::empty({1});
    TH(th_rowmax(ctx, pred.dptr(), nullptr, am.dptr(), b, c));
    TH(th_accuracy_count(ctx, am.dptr(), targets.dptr(), b, cnt.dptr()));
    return cnt.data()[0] / (float)targets.len();
}

Tensor one_hot(const Tensor &indices, size_t num_classes) {  // loss.rs:248-268 (host-side, off the hot path)
    TAPER_ASSERT(indices.shape().size() == 1, "Indices must be 1D");
    std::vector<float> idx = indices.data(), oh(idx.size() * num_classes, 0.f);
    for (size_t i = 0; i < idx.size(); ++i) {
        size_t cls = (size_t)idx[i];
        TAPER_ASSERT(cls < num_classes, "Index out of bounds for one_hot");
        oh[i * num_classes + cls] = 1.0f;
    }
    return Tensor(oh, {idx.size(), num_classes});
}

Tensor mse_loss(const Tensor &pred, const Tensor &targets) {  // loss.rs:76-80
    Tensor diff = pred - targets;
    return (diff * diff).mean();
}

// ---------------------------------------------------------------- layers
static std::vector<float> uniform_init(size_t n, float bound, uint64_t seed) {
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<float> d(-bound, bound);
    std::vector<float> v(n);
    for (auto &x : v) x = d(rng);
    return v;
}

Linear::Linear(size_t in_f, size_t out_f, bool with_bias, uint64_t seed) {  // nn.rs:35-50
    const float scale = std::sqrt(2.0f / (float)in_f);
    weight = Tensor(uniform_init(in_f * out_f, scale, seed), {out_f, in_f}).requires_grad();
    if (with_bias) bias = Tensor(std::vector<float>(out_f, 0.f), {out_f}).requires_grad();
}

Tensor Linear::forward(const Tensor &x) const { return x.linear(weight, bias, false); }          // nn.rs:54-60
Tensor Linear::forward_fused_relu(const Tensor &x) const { return x.linear(weight, bias, true); }

std::vector<Tensor> Linear::parameters() const {  // nn.rs:71-77
    std::vector<Tensor> p{weight};
    if (bias.defined()) p.push_back(bias);
    return p;
}

Conv2d::Conv2d(size_t in_ch, size_t out_ch, std::pair<int, int> kernel, std::pair<int, int> s, std::pair<int, int> p,
               bool with_bias, uint64_t seed, size_t g)
    : stride(s), padding(p), groups(g) {  // nn.rs:190-244
    TAPER_ASSERT(groups >= 1 && in_ch % groups == 0, "in_channels must be divisible by groups");      // nn.rs:205-209
    TAPER_ASSERT(out_ch % groups == 0, "out_channels must be divisible by groups");                    // nn.rs:210-214
    const size_t fan_in = in_ch * kernel.first * kernel.second / groups;                               // nn.rs:219
    const float bound = std::sqrt(2.0f / (float)fan_in) * std::sqrt(3.0f);
    weight = Tensor(uniform_init(out_ch * fan_in, bound, seed), {out_ch, in_ch / groups, (size_t)kernel.first, (size_t)kernel.second})
                 .requires_grad();
    if (with_bias) bias = Tensor(std::vector<float>(out_ch, 0.f), {out_ch}).requires_grad();
}

Tensor Conv2d::forward(const Tensor &x) const {
    if (groups == 1) return x.conv2d(weight, bias, stride, padding, dilation, fuse_relu);   // nn.rs:280-288
    // nn.rs:289-332: slices are fresh tensors without tape nodes, so nothing upstream of a grouped
    // convolution (and none of its own parameters) ever receives a gradient -- reproduced as is
    TAPER_ASSERT(x.shape().size() == 4 && x.shape()[1] % groups == 0, "Input channels must be divisible by groups");
    const size_t cin_g = x.shape()[1] / groups, cout_g = weight.shape()[0] / groups;
    std::vector<Tensor> outs;
    for (size_t g = 0; g < groups; ++g) {
        Tensor xs = x.slice_channels(g * cin_g, (g + 1) * cin_g);
        Tensor ws = weight.slice_output_channels(g * cout_g, (g + 1) * cout_g);
        Tensor bs = bias.defined() ? bias.slice_1d(g * cout_g, (g + 1) * cout_g) : Tensor();
        outs.push_back(xs.conv2d(ws, bs, stride, padding, dilation, fuse_relu));
    }
    return Tensor::cat(outs, 1);
}

std::vector<Tensor> Conv2d::parameters() const {
    std::vector<Tensor> p{weight};
    if (bias.defined()) p.push_back(bias);
    return p;
}

Tensor AvgPool2d::forward(const Tensor &x) const {  // nn.rs:593-608
    if (kernel == std::make_pair(0, 0)) return x.avg_pool2d({(int)x.shape()[2], (int)x.shape()[3]}, {1, 1}, {0, 0});
    return x.avg_pool2d(kernel, stride, padding);
}

Tensor AdaptiveAvgPool2d::forward(const Tensor &x) const {  // nn.rs:670-686
    const int kh = (int)x.shape()[2] / output_size.first, kw = (int)x.shape()[3] / output_size.second;
    return x.avg_pool2d({kh, kw}, {kh, kw}, {0, 0});
}

// TAPER_MLP3=0: three-layer classifiers keep the launch-per-layer forms (measurement probe; default: th_mlp3_xent, two launches)
bool mlp3_fuse() {
    static const bool on = [] { const char *e = std::getenv("TAPER_MLP3"); return !(e && e[0] == '0'); }();
    return on;
}

// TAPER_CONV_CHAIN=0: Trainer steps launch the convolutional front layer by layer (measurement probe; default: one launch where compiled)
static bool g_conv_chain = [] { const char *e = std::getenv("TAPER_CONV_CHAIN"); return !(e && e[0] == '0'); }();
void set_conv_chain(bool on) { g_conv_chain = on; }
bool conv_chain_enabled() { return g_conv_chain; }
static bool chain_fuse() { return g_conv_chain; }
// TAPER_CHAIN_HEAD=0: the classifier behind a chain keeps its own launches (th_linear_xent_wide + the bias finish)
static bool g_conv_chain_head = [] { const char *e = std::getenv("TAPER_CHAIN_HEAD"); return !(e && e[0] == '0'); }();
void set_conv_chain_head(bool on) { g_conv_chain_head = on; }
bool conv_chain_head_enabled() { return g_conv_chain_head; }

Tensor Sequential::forward(const Tensor &input) const { return forward_prefix(input, layers.size()); }  // nn.rs:149-151

size_t Sequential::conv_stages_at(size_t i, size_t n_layers, std::vector<ConvStage> *stages) const {
    size_t j = i;
    while (j < n_layers) {
        auto *cv = dynamic_cast<Conv2d *>(layers[j].get());
        if (!(cv && cv->fuse_relu && cv->groups == 1 && cv->stride == std::make_pair(1, 1) && cv->dilation == std::make_pair(1, 1) &&
              cv->padding == std::make_pair(1, 1) && cv->bias.defined()))
            break;
        int post = TH_CHAIN_NONE;
        if (j + 1 < n_layers) {
            auto *mp = dynamic_cast<MaxPool2d *>(layers[j + 1].get());
            auto *gp = dynamic_cast<AdaptiveAvgPool2d *>(layers[j + 1].get());
            if (mp && mp->kernel == std::make_pair(2, 2) && (mp->stride == std::make_pair(0, 0) || mp->stride == std::make_pair(2, 2)) &&
                mp->padding == std::make_pair(0, 0))
                post = TH_CHAIN_MAXPOOL2;
            else if (gp && gp->output_size == std::make_pair(1, 1))
                post = TH_CHAIN_GLOBAL_AVG;
        }
        stages->push_back({cv->weight, cv->bias, post});
        j += post == TH_CHAIN_NONE ? 1 : 2;
        if (post == TH_CHAIN_GLOBAL_AVG) break;
    }
    return j;
}

Tensor Sequential::forward_prefix(const Tensor &input, size_t n_layers) const {
    Tensor x = input;
    for (size_t i = 0; i < n_layers; ++i) {
        if (fuse && i + 1 < n_layers) {
            auto *lin = dynamic_cast<Linear *>(layers[i].get());
            if (lin && dynamic_cast<ReLU *>(layers[i + 1].get())) {
                x = lin->forward_fused_relu(x);  // Linear + ReLU: one kernel, one tape node
                ++i;
                continue;
            }
        }
        if (fuse && chain_fuse() && i + 1 < n_layers && PoolBiasScope::active() && x.shape().size() == 4) {
            // Trainer steps: the whole run of Conv2dReLU(3x3, stride 1, pad 1) [+ MaxPool2d(2) | + global average pool] rows in front of the
            // classifier as ONE launch, when an instance is compiled for it (th_conv_chain_supported): the maps never leave the CU
            std::vector<ConvStage> stages;
            conv_stages_at(i, n_layers, &stages);
            // (a run that ends in conv rows without a pool is taken whole where the chain kernel can write that map; else up to its last pooled stage)
            if (!(stages.size() >= 2 && x.conv_chain_supported(stages)))
                while (!stages.empty() && stages.back().post == TH_CHAIN_NONE) stages.pop_back();
            size_t j2 = i;
            for (const auto &st : stages) j2 += st.post == TH_CHAIN_NONE ? 1 : 2;
            if (stages.size() >= 2 && x.conv_chain_supported(stages)) {   // (a single conv + pool keeps its own launch, below)
                x = x.conv_chain(stages);
                i = j2 - 1;
                continue;
            }
        }
        if (fuse && i + 1 < n_layers && PoolBiasScope::active()) {
            // Trainer steps: Conv2dReLU(3x3, stride 1) + MaxPool2d(2) as one launch that never writes the full-resolution map
            auto *cv = dynamic_cast<Conv2d *>(layers[i].get());
            auto *mp = dynamic_cast<MaxPool2d *>(layers[i + 1].get());
            if (cv && mp && cv->fuse_relu && cv->groups == 1 && cv->stride == std::make_pair(1, 1) && cv->dilation == std::make_pair(1, 1) &&
                mp->kernel == std::make_pair(2, 2) && (mp->stride == std::make_pair(0, 0) || mp->stride == std::make_pair(2, 2)) &&
                mp->padding == std::make_pair(0, 0) &&   // (faithful mode never hands a gradient to the conv's input or weight, Q2)
                x.conv2d_relu_maxpool2_supported(cv->weight, cv->bias, cv->padding)) {
                x = x.conv2d_relu_maxpool2(cv->weight, cv->bias, cv->padding);
                ++i;
                continue;
            }
        }
        if (fuse && i + 1 < n_layers && PoolBiasScope::active()) {
            // Trainer steps: Conv2dReLU(3x3, stride 1) + global average pool as one launch that never writes the map
            auto *cv = dynamic_cast<Conv2d *>(layers[i].get());
            auto *gp = dynamic_cast<AdaptiveAvgPool2d *>(layers[i + 1].get());
            if (cv && gp && gp->output_size == std::make_pair(1, 1) && cv->fuse_relu && cv->groups == 1 && cv->stride == std::make_pair(1, 1) &&
                cv->dilation == std::make_pair(1, 1) && x.conv2d_relu_gap_supported(cv->weight, cv->bias, cv->padding)) {
                x = x.conv2d_relu_gap(cv->weight, cv->bias, cv->padding);
                ++i;
                continue;
            }
        }
        x = layers[i]->forward(x);
    }
    return x;
}

std::vector<Tensor> Sequential::parameters() const {  // nn.rs:159-161
    std::vector<Tensor> p;
    for (auto &l : layers)
        for (auto &t : l->parameters()) p.push_back(t);
    return p;
}

}  // namespace taper

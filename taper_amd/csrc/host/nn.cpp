// nn.cpp -- loss, layers, optimizers, data loader, communicator and trainer of
// the host mirror (src/loss.rs, src/nn.rs, src/optim.rs, src/data/mnist.rs,
// src/train.rs, examples/train_mnist*.rs).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "taper.h"

namespace taper {

#define TH(call) th_check((call), #call)

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
static size_t mlp2_min_batch() {
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

static bool mlp2_shapes_ok(size_t batch, const std::vector<Tensor> &w, int64_t n_rows) {
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
static bool mlp3_fuse() {
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

// ---------------------------------------------------------------- data
MNISTDataset MNISTDataset::from_host(const std::vector<float> &images, const std::vector<float> &labels, bool train) {
    TAPER_ASSERT(images.size() == labels.size() * 784, "MNISTDataset: images must be [N,784]");
    MNISTDataset d;
    d.images = Tensor(images, {labels.size(), 784});
    d.labels = Tensor(labels, {labels.size()});
    d.train = train;
    return d;
}

MNISTDataset MNISTDataset::from_u8(const std::vector<uint8_t> &pixels, const std::vector<uint8_t> &labels, bool train) {
    TAPER_ASSERT(pixels.size() == labels.size() * 784, "MNISTDataset: pixels must be [N,784]");
    th_ctx *ctx = Device::ctx();
    MNISTDataset d;
    d.images = Tensor::empty({labels.size(), 784});
    auto staging = Buffer::alloc((pixels.size() + 3) / 4);
    TH(th_memcpy_h2d(ctx, staging->d, pixels.data(), pixels.size()));
    TH(th_u8_to_unit_f32(ctx, reinterpret_cast<const uint8_t *>(staging->d), d.images.dptr(), pixels.size()));  // mnist.rs:226
    std::vector<float> lf(labels.begin(), labels.end());  // mnist.rs:268
    d.labels = Tensor(lf, {labels.size()});
    d.train = train;
    Device::sync();
    return d;
}

static uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

MNISTDataset MNISTDataset::from_idx_files(const std::string &images_path, const std::string &labels_path, bool train) {
    auto slurp = [](const std::string &path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw Error("Failed to open " + path);
        return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    };
    std::vector<unsigned char> ib = slurp(images_path), lb = slurp(labels_path);
    // mnist.rs:185-233
    TAPER_ASSERT(ib.size() >= 16, "File " + images_path + " is too small");
    TAPER_ASSERT(be32(ib.data()) == 0x00000803, "Invalid magic number for images");
    const size_t n = be32(ib.data() + 4), rows = be32(ib.data() + 8), cols = be32(ib.data() + 12);
    TAPER_ASSERT(rows == 28 && cols == 28, "Unexpected image size");
    TAPER_ASSERT(ib.size() == 16 + n * 784, "File size mismatch (images)");
    // mnist.rs:236-274
    TAPER_ASSERT(lb.size() >= 8, "File " + labels_path + " is too small");
    TAPER_ASSERT(be32(lb.data()) == 0x00000801, "Invalid magic number for labels");
    const size_t nl = be32(lb.data() + 4);
    TAPER_ASSERT(lb.size() == 8 + nl, "File size mismatch (labels)");
    TAPER_ASSERT(nl == n, "image / label count mismatch");
    return from_u8(std::vector<uint8_t>(ib.begin() + 16, ib.end()), std::vector<uint8_t>(lb.begin() + 8, lb.end()), train);
}

static inline uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

MNISTDataset MNISTDataset::synthetic(size_t n, uint64_t seed, bool train) {
    std::vector<uint8_t> px(n * 784), lb(n);
    uint64_t s = seed;
    for (size_t i = 0; i < px.size(); i += 8) {
        uint64_t r = splitmix64(s);
        for (size_t j = 0; j < 8 && i + j < px.size(); ++j) px[i + j] = (uint8_t)(r >> (8 * j));
    }
    for (size_t i = 0; i < n; ++i) lb[i] = (uint8_t)(splitmix64(s) % 10);
    return from_u8(px, lb, train);
}

std::pair<Tensor, Tensor> MNISTDataset::get_batch(const std::vector<size_t> &indices) const {  // mnist.rs:277-310
    std::vector<int32_t> idx(indices.begin(), indices.end());
    th_ctx *ctx = Device::ctx();
    auto d_idx = Buffer::alloc(idx.size());
    TH(th_memcpy_h2d(ctx, d_idx->d, idx.data(), idx.size() * sizeof(int32_t)));
    Tensor xb = Tensor::empty({idx.size(), 784}), yb = Tensor::empty({idx.size()});
    TH(th_gather_batch(ctx, images.dptr(), labels.dptr(), reinterpret_cast<const int32_t *>(d_idx->d), (int64_t)idx.size(), nullptr,
                       (int)idx.size(), 784, xb.dptr(), yb.dptr()));
    return {xb, yb};
}

DataLoader::DataLoader(MNISTDataset dataset, size_t batch_size, bool shuffle, uint64_t seed)
    : ds_(std::move(dataset)), bs_(batch_size), shuffle_(shuffle), rng_(seed) {  // mnist.rs:335-353
    TAPER_ASSERT(batch_size > 0, "DataLoader: batch_size must be positive");
    indices_.resize(ds_.len());
    for (size_t i = 0; i < indices_.size(); ++i) indices_[i] = (int32_t)i;
    if (shuffle_) std::shuffle(indices_.begin(), indices_.end(), rng_);
    d_idx_ = Buffer::alloc(std::max<size_t>(indices_.size(), 1));
    upload_indices();
}

void DataLoader::upload_indices() {
    TH(th_memcpy_h2d(Device::ctx(), d_idx_->d, indices_.data(), indices_.size() * sizeof(int32_t)));
}

void DataLoader::reset() {  // mnist.rs:355-363
    cur_ = 0;
    if (shuffle_) {
        std::shuffle(indices_.begin(), indices_.end(), rng_);
        upload_indices();
    }
}

size_t DataLoader::num_batches() const { return (ds_.len() + bs_ - 1) / bs_; }  // mnist.rs:365-367

bool DataLoader::next(Tensor *images, Tensor *labels) {  // mnist.rs:373-385 (keeps the last partial batch)
    if (cur_ >= ds_.len()) return false;
    const size_t end = std::min(cur_ + bs_, ds_.len()), b = end - cur_;
    *images = Tensor::empty({b, 784});
    *labels = Tensor::empty({b});
    TH(th_gather_batch(Device::ctx(), ds_.images.dptr(), ds_.labels.dptr(), d_indices() + cur_, (int64_t)b, nullptr, (int)b, 784,
                       images->dptr(), labels->dptr()));
    cur_ = end;
    return true;
}

// ---------------------------------------------------------------- communicator
std::vector<uint8_t> Communicator::unique_id() {
    std::vector<uint8_t> id(128);
    TH(th_comm_unique_id(id.data()));
    return id;
}

Communicator::Communicator(int n, int r, const std::vector<uint8_t> &id) : n_ranks(n), rank(r) {
    TAPER_ASSERT(id.size() == 128, "Communicator: unique id must be 128 bytes");
    TH(th_comm_init_rank(Device::ctx(), n, r, id.data(), &comm_));
}

Communicator::~Communicator() { th_comm_destroy(comm_); }

std::shared_ptr<Communicator> Communicator::p2p(int n, int r) {
    std::shared_ptr<Communicator> c(new Communicator());
    c->n_ranks = n;
    c->rank = r;
    c->p2p_ = true;
    TH(th_comm_init_p2p(Device::ctx(), n, r, &c->comm_));
    if (const char *e = std::getenv("TAPER_P2P_FUSE")) c->fuse_adam = e[0] != '0';   // measurement / test probe
    if (const char *e = std::getenv("TAPER_DP_INKERNEL")) c->inkernel = e[0] != '0';  // 0: always the three-launch form
    return c;
}

std::shared_ptr<Communicator> Communicator::loopback() {
    std::shared_ptr<Communicator> c(new Communicator());
    c->n_ranks = 2;
    c->rank = 0;
    c->p2p_ = true;
    c->loopback_ = true;
    TH(th_comm_init_loopback(Device::ctx(), &c->comm_));
    return c;
}

bool Communicator::tail_exchange_ok(int batch, int in_features, int hidden, int classes) const {
    if (!p2p_ || !inkernel) return false;
    // one rank: the mean over the ranks IS this rank's gradient -- the step is the single-GPU step, nothing to exchange
    if (n_ranks == 1) return th_mlp_tail_supported(batch, in_features, hidden, classes, 0) != 0;
    return th_mlp_tail_dp_supported(comm_, Device::ctx(), batch, in_features, hidden, classes) != 0;
}

int64_t Communicator::inkernel_launches() const {
    int64_t n = 0;
    TH(th_comm_stats_inkernel(comm_, &n));
    return n;
}

int Communicator::exchange_selftest(int slots, int rounds) const {
    int bad = 0;
    TH(th_comm_exchange_selftest(comm_, Device::ctx(), slots, rounds, &bad));
    return bad;
}

int Communicator::ranks_on_this_device() const {
    int n = 1;
    TH(th_comm_sharing(comm_, &n));
    return n;
}

std::vector<uint8_t> Communicator::export_arena(Optimizer &opt, bool fine_grained) {
    TAPER_ASSERT(p2p_, "Communicator::export_arena: not a peer-to-peer communicator");
    FlatParams &fp = opt.flat();
    if (fine_grained && !fp.g_arena->fine_grained) fp.rehome_grads(Buffer::alloc_finegrained((size_t)fp.total));
    std::vector<uint8_t> blob(TH_P2P_BLOB_BYTES);
    TH(th_comm_p2p_export(comm_, fp.g_arena->d, (size_t)fp.total, blob.data()));
    g_hold_ = fp.g_arena;   // peers map this allocation: it must outlive the communicator, whatever happens to the optimizer
    p_hold_ = fp.p_arena;
    return blob;
}

void Communicator::connect(const std::vector<uint8_t> &blobs) {
    TAPER_ASSERT(p2p_ && blobs.size() == (size_t)n_ranks * TH_P2P_BLOB_BYTES, "Communicator::connect: expected one blob per rank");
    TH(th_comm_p2p_connect(comm_, blobs.data()));
}

namespace {
// the self-check's gradient of rank r, round k, element i: small integers, so every sum and mean below is exact in fp32
inline float check_pattern(int r, int k, size_t i) { return (float)((r + 1) * (k + 1)) + (float)(i % 7); }
inline float check_mean(int w, int k, size_t i) { return (float)(k + 1) * (float)(w + 1) / 2.0f + (float)(i % 7); }
// llvm.powi.f32 as compiler-rt lowers it (csrc/adam_dev.h powi_f32; optim.rs:87-88)
inline float host_powi(float a, int b) {
    float r = 1.0f;
    while (true) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return r;
}
}  // namespace

bool Communicator::self_check(Optimizer &opt, int rounds) {
    FlatParams &fp = opt.flat();
    th_ctx *ctx = Device::ctx();
    const size_t n = (size_t)fp.total;
    std::vector<float> pat(n), got(n);
    bool ok = true;
    auto reset_grads = [&]() {
        TH(th_fill_f32(ctx, fp.g_arena->d, 0.f, n));
        for (auto &p : fp.params) {
            p.grad_->has = false;
            p.grad_->known_zero = true;
        }
    };
    // (1) the in-place kernel, `rounds` times through the same addresses, a different pattern per rank / round / element.  Every rank runs
    // every round whatever it has seen so far: nobody is left waiting for a peer that has given up.
    for (int k = 0; k < rounds; ++k) {
        for (size_t i = 0; i < n; ++i) pat[i] = check_pattern(rank, k, i);
        TH(th_memcpy_h2d(ctx, fp.g_arena->d, pat.data(), n * sizeof(float)));
        allreduce_mean(fp.g_arena->d, n);
        TH(th_memcpy_d2h(ctx, got.data(), fp.g_arena->d, n * sizeof(float)));
        // (patterns of different rounds / ranks differ by >= 0.5 per element; 1/W is not a power of two for every W)
        for (size_t i = 0; ok && i < n; ++i) ok = std::fabs(got[i] - check_mean(n_ranks, k, i)) <= 1e-3f;
    }
    if (p2p_ && timed_out()) ok = false;
    // (2) the fused all-reduce + Adam kernel on the real p / m / v (saved and restored), against optim.rs:83-113 evaluated on the host
    Adam *adam = dynamic_cast<Adam *>(&opt);
    if (p2p_ && fuse_adam && adam) {
        adam->flush_deferred();   // (nothing is pending at bootstrap; a caller that checks later starts from a clean optimizer)
        std::vector<float> p0(n), pm(n), pv(n), m0, v0;
        TH(th_memcpy_d2h(ctx, p0.data(), fp.p_arena->d, n * sizeof(float)));
        const int t0 = adam->t();
        m0 = adam->m();
        v0 = adam->v();
        std::vector<char> had(fp.params.size());
        for (size_t j = 0; j < fp.params.size(); ++j) {
            had[j] = fp.params[j].grad_->has;
            fp.params[j].grad_->has = true;          // the kernel skips grad-less tensors (Q8): check every slice
        }
        std::vector<float> ph = p0, mh(n, 0.f), vh(n, 0.f);
        adam->load_state(t0, std::vector<float>(m0.size(), 0.f), std::vector<float>(v0.size(), 0.f));
        const float b1 = adam->beta1(), b2 = adam->beta2(), eps = adam->eps(), wd = adam->weight_decay(), lr = adam->get_lr();
        for (int k = 0; k < rounds; ++k) {
            for (size_t i = 0; i < n; ++i) pat[i] = check_pattern(rank, k, i);
            TH(th_memcpy_h2d(ctx, fp.g_arena->d, pat.data(), n * sizeof(float)));
            const bool ran = adam->step_reduced(*this);
            if (!ran) {
                // a host-state precondition of the fused form (deferred / carried updates, an external tick), not a link problem: say so, and
                // keep the peers' launches in step through the in-place form (same flag protocol) instead of leaving them to their time-out
                if (ok) fprintf(stderr, "taper: p2p self-check, rank %d: the optimizer cannot take the fused all-reduce + Adam launch right now "
                                        "(deferred or externally ticked updates pending); reported as a failed check\n", rank);
                ok = false;
                allreduce_mean(fp.g_arena->d, n);
                continue;
            }
            const int t = t0 + k + 1;
            const float step = lr * (std::sqrt(1.0f - host_powi(b2, t)) / (1.0f - host_powi(b1, t)));
            for (size_t j = 0; j < fp.params.size(); ++j)
                for (size_t e = 0; e < fp.params[j].len(); ++e) {
                    const size_t i = (size_t)fp.offsets[j] + e;
                    const float g = check_mean(n_ranks, k, i) + wd * ph[i];
                    mh[i] = b1 * mh[i] + (1.0f - b1) * g;
                    vh[i] = b2 * vh[i] + (1.0f - b2) * g * g;
                    ph[i] = ph[i] - step * mh[i] / (std::sqrt(vh[i]) + eps);
                }
        }
        Device::sync();
        if (ok) {
            TH(th_memcpy_d2h(ctx, pm.data(), fp.p_arena->d, n * sizeof(float)));
            const std::vector<float> mg = adam->m(), vg = adam->v();
            size_t u = 0;
            for (size_t j = 0; ok && j < fp.params.size(); ++j)
                for (size_t e = 0; ok && e < fp.params[j].len(); ++e, ++u) {
                    const size_t i = (size_t)fp.offsets[j] + e;
                    auto near = [](float a, float b) { return std::fabs(a - b) <= 1e-5f * std::max(1.0f, std::fabs(b)); };
                    ok = near(pm[i], ph[i]) && near(mg[u], mh[i]) && near(vg[u], vh[i]);
                }
            ok = ok && adam->t() == t0 + rounds;
        }
        // back to the state training starts from
        TH(th_memcpy_h2d(ctx, fp.p_arena->d, p0.data(), n * sizeof(float)));
        adam->load_state(t0, m0, v0);
        for (size_t j = 0; j < fp.params.size(); ++j) fp.params[j].grad_->has = had[j];
        if (timed_out()) ok = false;
    }
    reset_grads();
    fp.sync_mask();
    Device::sync();
    // (3) the exchange inside the gradient launch (th_mlp_tail_dp), on its own: 16 slots of patterns per round through the same flags, receive
    // slots and step counter a training step uses.  Collective like the rounds above: every rank runs it whatever it has seen so far.
    if (p2p_ && inkernel && rounds > 0) {
        if (exchange_selftest(16, rounds) != 0) ok = false;
        if (timed_out()) ok = false;
    }
    return ok;
}

bool Communicator::timed_out() const {
    int e = 0;
    TH(th_comm_error(comm_, Device::ctx(), &e));
    return e != 0;
}

bool Communicator::failed() const {
    int e = 0;
    TH(th_comm_error_peek(comm_, &e));
    return e != 0;
}

void Communicator::set_timeout_ms(int64_t ms) { TH(th_comm_set_timeout_ms(comm_, ms)); }

uint32_t *Communicator::step_word() const {
    uint32_t *w = nullptr;
    TH(th_comm_step_word(comm_, &w));
    return w;
}

const uint32_t *Communicator::error_word() const {
    const uint32_t *w = nullptr;
    TH(th_comm_error_word(comm_, &w));
    return w;
}

float Communicator::time_exchange(Adam &opt, int reps) {
    th_ctx *ctx = Device::ctx();
    th_event *e0 = nullptr, *e1 = nullptr;
    TH(th_event_create(&e0));
    TH(th_event_create(&e1));
    FlatParams &fp = opt.flat();
    opt.set_step_guard(is_p2p() ? error_word() : nullptr);
    auto once = [&] {
        if (!opt.step_reduced(*this)) {
            fp.zero_missing();
            allreduce_mean(fp.g_arena->d, (size_t)fp.total);
            opt.step();
        }
    };
    for (int i = 0; i < 3; ++i) once();
    TH(th_event_record(ctx, e0));
    for (int i = 0; i < reps; ++i) once();
    TH(th_event_record(ctx, e1));
    Device::sync();
    opt.set_step_guard(nullptr);   // the word lives in this communicator's state block: a later opt.step() must not read it once we are gone
    float ms = 0.f;
    TH(th_event_elapsed_ms(e0, e1, &ms));
    th_event_destroy(e0);
    th_event_destroy(e1);
    return ms * 1e3f / (float)std::max(reps, 1);
}

void Communicator::allreduce_mean(float *d_buf, size_t n) const {
    TH(th_allreduce_sum_scale(comm_, Device::ctx(), d_buf, n, 1.0f / (float)n_ranks));
}

// ---------------------------------------------------------------- trainer
static Tensor shape_input(const Tensor &images, const Shape &sample_shape) {
    if (sample_shape.empty()) return images;
    Shape s{images.shape()[0]};
    s.insert(s.end(), sample_shape.begin(), sample_shape.end());
    return images.reshape(s);  // train_mnist_cnn.rs:161-162
}

static void reduce_grads(Trainer &t) {
    // (the Adam step behind a peer-to-peer all-reduce skips itself on the device when that all-reduce timed out)
    t.optimizer->set_step_guard(t.comm && t.comm->is_p2p() ? t.comm->error_word() : nullptr);
    if (!t.comm) return;
    FlatParams &fp = t.optimizer->flat();
    fp.zero_missing();  // grad None contributes zeros; the has_grad mask is rank-invariant
    t.comm->allreduce_mean(fp.g_arena->d, (size_t)fp.total);
}

void Trainer::train_step(const Tensor &images, const Tensor &labels, float *loss_out, float *acc_out) {
    Tape::reset();                                              // train_mnist.rs:91
    Tensor logits = model->forward(shape_input(images, sample_shape));  // :101
    Tensor loss = cross_entropy_loss(logits, labels);           // :107
    const float acc = accuracy(logits, labels);                 // :110
    loss.backward();                                            // :115
    if (!(comm && optimizer->step_reduced(*comm))) {
        reduce_grads(*this);
        optimizer->step();                                      // :118
    }
    optimizer->zero_grad();                                     // :119
    if (loss_out) *loss_out = loss.data()[0];                   // :121
    if (acc_out) *acc_out = acc;
    check_comm();                                               // (the read-back above synchronised the stream)
}

void Trainer::check_comm() const {
    // a peer that never arrived at an all-reduce: the launch applied nothing and every later one is a no-op (th_comm_error_peek) -- the
    // replicas are no longer in step, and the run must end here, loudly, not train on
    if (comm && comm->is_p2p() && comm->failed())
        throw Error("data-parallel all-reduce timed out waiting for a peer (rank " + std::to_string(comm->rank) + " of " +
                    std::to_string(comm->n_ranks) + "): no update was applied from that step on; the replicas are out of step");
}

EpochResult Trainer::train_epoch(DataLoader &loader) {  // train.rs:98-144
    EpochResult r;
    float total_loss = 0.f;
    loader.reset();
    r.num_batches = loader.num_batches();
    Tensor images, labels;
    while (loader.next(&images, &labels)) {
        float loss, acc;
        train_step(images, labels, &loss, &acc);
        const size_t b = images.shape()[0];
        r.total_correct += (size_t)(acc * (float)b);  // train.rs:117 (truncating cast, Q13)
        r.total_samples += b;
        total_loss += loss;
        r.losses.push_back(loss);
        r.ncorrect.push_back(acc * (float)b);
    }
    r.avg_loss = total_loss / (float)r.num_batches;                    // train.rs:140
    r.accuracy = (float)r.total_correct / (float)r.total_samples;      // train.rs:141
    return r;
}

EpochResult Trainer::evaluate(DataLoader &loader) {  // train.rs:147-172
    // Same formulas, no per-batch read-back: the loss kernel appends {loss, n_correct} of every batch to a device
    // log (as in the graph-replayed training epoch) and the host reads the log once at the end.
    EpochResult r;
    loader.reset();
    const size_t nb = loader.num_batches();
    r.num_batches = nb;
    if (nb == 0) return r;
    th_ctx *ctx = Device::ctx();
    auto log = Buffer::alloc(2 * nb), st = Buffer::alloc(4);
    TH(th_fill_f32(ctx, st->d, 0.f, 4));
    std::vector<size_t> sizes;
    Tensor images, labels;
    // nothing is differentiated here: a Sequential's conv front may take the launches that never write the full-resolution maps
    // (conv + pool pairs, the one-launch conv chain from batch 96 up) exactly as inside a training step
    // (the same condition as a training step's; NoGradScope: no count buffers, no bias tape nodes for gradients nobody will ask for)
    PoolBiasScope pool_scope(fuse_head && dynamic_cast<Sequential *>(model.get()) != nullptr);
    NoGradScope no_grad;
    while (loader.next(&images, &labels)) {
        Tape::reset();
        const size_t b = images.shape()[0];
        StepLogSink sink{log->d, (int64_t)nb, reinterpret_cast<int64_t *>(st->d), (int64_t)b, nullptr};
        Tensor logits = model->forward(shape_input(images, sample_shape));
        Tensor ncorrect;
        cross_entropy_loss(logits, labels, &ncorrect, &sink);   // loss.rs:136-195 + the count of loss.rs:271-290
        sizes.push_back(b);
    }
    Tape::reset();
    std::vector<float> mt(2 * nb);
    TH(th_memcpy_d2h(ctx, mt.data(), log->d, mt.size() * sizeof(float)));
    float total_loss = 0.f;
    for (size_t s = 0; s < nb && s < sizes.size(); ++s) {
        const size_t b = sizes[s];
        const float acc = mt[2 * s + 1] / (float)b;       // loss.rs:289
        r.total_correct += (size_t)(acc * (float)b);      // train.rs:160 (truncating cast, Q13)
        r.total_samples += b;
        total_loss += mt[2 * s];
        r.losses.push_back(mt[2 * s]);
        r.ncorrect.push_back(mt[2 * s + 1]);
    }
    r.avg_loss = total_loss / (float)nb;
    r.accuracy = (float)r.total_correct / (float)r.total_samples;
    return r;
}

// The captured form of a step: identical arithmetic, but nothing is read back.  The batches
// come from the device-resident dataset through the device cursor (ONE gather launch for a
// whole chunk of steps), and the loss kernel itself appends {loss, n_correct} to the device
// log and advances the step / cursor state.
// The Linear layers of a model that is Linear + ReLU (+ Linear + ReLU) + Linear on the loader's 784-wide rows (BASELINE's 784-128-10,
// examples/train_mnist.rs:40-48's 784-128-64-10); empty otherwise.
static bool mlp2_layers(const Module *model, std::vector<Tensor> *w, std::vector<Tensor> *b) {
    auto *seq = dynamic_cast<const Sequential *>(model);
    if (!seq || !seq->fuse || (seq->layers.size() != 3 && seq->layers.size() != 5)) return false;
    w->clear();
    b->clear();
    for (size_t i = 0; i < seq->layers.size(); ++i) {
        if (i % 2 == 1) {
            if (!dynamic_cast<ReLU *>(seq->layers[i].get())) return false;
            continue;
        }
        auto *l = dynamic_cast<Linear *>(seq->layers[i].get());
        if (!l) return false;
        w->push_back(l->weight);
        b->push_back(l->bias);
    }
    return (*w)[0].shape().size() == 2 && (*w)[0].shape()[1] == 784;   // (the loader's rows: data/mnist.rs:16)
}

// this model at this batch takes th_mlp2_xent / th_mlp2_xent_deep, its rows read in place -- decided on EVERYTHING the step will ask for
// (shapes, and the parameters' state: every one trains, no gradient already present, aligned storage), so that a model the large-batch
// step cannot take (a frozen layer, gradients left by a manual backward) falls back to the gathered forms instead of failing inside the
// step (r04 checked the shapes only)
bool Trainer::mlp2_step(size_t batch, int64_t n_rows) const {
    if (fuse_head < 2 || !sample_shape.empty() || batch < mlp2_min_batch()) return false;
    std::vector<Tensor> w, b;
    if (!(mlp2_layers(model.get(), &w, &b) && mlp2_params_ok(w, b) && mlp2_shapes_ok(batch, w, n_rows))) return false;
    // Two hidden layers: below ~1 800 rows the two-launch classifier (th_mlp3_xent behind a gather) is the faster step where it applies --
    // 26.7 / 30.3 / 32.8 / 39.2 us at 512 / 768 / 1 024 / 1 536 rows against 34.4 / 35.8 / 37.7 / 42.0 for th_mlp2_xent_deep, whose 16-row
    // blocks each leave a whole [h2][h1] share of dW2 behind; 47.5 against 43.7 at 2 048 (tools/mlp_min_batch_probe.py, r05)
    static const size_t deep_min = [] { const char *e = std::getenv("TAPER_MLP2_DEEP_MIN_BATCH"); return e ? (size_t)std::max(32, atoi(e)) : (size_t)1792; }();
    if (w.size() == 3 && batch < deep_min && mlp3_fuse() &&
        th_mlp3_supported((int)batch, (int)w[0].shape()[1], (int)w[0].shape()[0], (int)w[1].shape()[0], (int)w[2].shape()[0]))
        return false;
    return true;
}

void Trainer::enqueue_compute(float *d_xb, float *d_yb, size_t batch, const th_row_source *rows) {
    int64_t *state = reinterpret_cast<int64_t *>(state_->d);
    // Adam updates ride in the epilogues of the kernels that produce the gradients -- unless the
    // gradients still have to be all-reduced across ranks first by a launch of their own.  The Linear + ReLU + Linear step over a
    // peer-to-peer communicator reduces them INSIDE its gradient launch (th_mlp_tail_dp) and keeps the fused epilogues.
    const bool dp_tail = comm && !rows && tail_exchange_step(batch);
    FusedAdamScope scope((fuse_adam && (!comm || dp_tail)) ? optimizer.get() : nullptr);
    TailExchangeScope xscope(dp_tail ? comm.get() : nullptr);
    PoolBiasScope pool_scope(fuse_head && dynamic_cast<Sequential *>(model.get()) != nullptr);
    Tape::reset();
    StepLogSink sink{metrics_->d, (int64_t)metrics_cap_, state, (int64_t)batch,
                     FusedAdamScope::active() ? optimizer->d_tick() : nullptr};
    Tensor ncorrect, loss;
    auto *seq = dynamic_cast<Sequential *>(model.get());
    if (rows) {
        // Linear + ReLU (+ Linear + ReLU) + Linear at large batch: three launches for the whole step, the rows read in place (mlp2_step said so,
        // on the same predicate, at the start of this chunk of steps; every step ends with zero_grad)
        std::vector<Tensor> w, b;
        TAPER_ASSERT(mlp2_layers(model.get(), &w, &b) && mlp2_supported(*rows, batch, w, b),
                     "Trainer: the large-batch MLP step met parameters it cannot take (gradients already present?)");
        loss = mlp2_cross_entropy(*rows, batch, w, b, &ncorrect, &sink);
        loss.backward();
        if (!(comm && optimizer->step_reduced(*comm))) {
            reduce_grads(*this);
            optimizer->step();
            optimizer->set_step_guard(nullptr);   // (the guard is the communicator's word: it must not outlive this call in the optimizer -- ADVICE r04)
        }
        optimizer->zero_grad();
        return;
    }
    Tensor x = Tensor::from_device(d_xb, {batch, 784});
    Tensor y = Tensor::from_device(d_yb, {batch});
    Tensor xin = shape_input(x, sample_shape);
    Linear *last = (fuse_head && seq && !seq->layers.empty()) ? dynamic_cast<Linear *>(seq->layers.back().get()) : nullptr;
    bool used_head = false;
    Adam *adam = dynamic_cast<Adam *>(optimizer.get());
    const size_t nl = seq ? seq->layers.size() : 0;
    Linear *hidden = (last && fuse_head >= 2 && seq->fuse && nl >= 3 && dynamic_cast<ReLU *>(seq->layers[nl - 2].get()))
                         ? dynamic_cast<Linear *>(seq->layers[nl - 3].get()) : nullptr;
    // Linear + ReLU + Linear + ReLU + Linear + cross-entropy: two launches for the whole classifier (th_mlp3_xent)
    Linear *hidden0 = (hidden && mlp3_fuse() && nl >= 5 && dynamic_cast<ReLU *>(seq->layers[nl - 4].get()))
                          ? dynamic_cast<Linear *>(seq->layers[nl - 5].get()) : nullptr;
    if (hidden0) {
        const Tensor w3[3] = {hidden0->weight, hidden->weight, last->weight}, b3[3] = {hidden0->bias, hidden->bias, last->bias};
        // (shapes first: the prefix is only run once the form of the step is known)
        const size_t in_f = hidden0->weight.shape()[1];
        // the reference CNN whole (conv rows + pools, global average pool, Flatten, this classifier): its rows in the chain launch
        if (fuse_head >= 2 && seq->fuse && conv_chain_enabled() && conv_chain_head_enabled() && PoolBiasScope::active() && xin.shape().size() == 4 &&
            nl >= 8) {
            auto *fl = dynamic_cast<Flatten *>(seq->layers[nl - 6].get());
            std::vector<ConvStage> stages;
            if (fl && fl->start_dim == 1 && seq->conv_stages_at(0, nl - 6, &stages) == nl - 6 && conv_chain_mlp3_supported(xin, stages, w3, b3)) {
                loss = conv_chain_mlp3_cross_entropy(xin, stages, w3, b3, y, &ncorrect, &sink);
                used_head = true;
            }
        }
        if (!used_head && th_mlp3_supported((int)batch, (int)in_f, (int)w3[0].shape()[0], (int)w3[1].shape()[0], (int)w3[2].shape()[0])) {
            Tensor xh = seq->forward_prefix(xin, nl - 5);
            if (mlp3_supported(xh, w3, b3)) {
                loss = mlp3_cross_entropy(xh, w3, b3, y, &ncorrect, &sink);
                used_head = true;
            } else {
                // (cannot happen for parameters homed in this Trainer's optimizer; keep the step correct anyway)
                if (adam) adam->flush_deferred();
                Tensor h = xh;
                for (size_t i = nl - 5; i < nl; ++i) h = seq->layers[i]->forward(h);
                loss = cross_entropy_loss(h, y, &ncorrect, &sink);
                used_head = true;
            }
        }
    }
    if (last && !hidden && !used_head && fuse_head >= 2 && seq->fuse && conv_chain_enabled() && conv_chain_head_enabled() && nl >= 3 &&
        PoolBiasScope::active() && xin.shape().size() == 4) {
        // conv rows + Flatten + Linear + cross-entropy (the simple CNN): two launches per step
        auto *fl = dynamic_cast<Flatten *>(seq->layers[nl - 2].get());
        std::vector<ConvStage> stages;
        if (fl && fl->start_dim == 1 && seq->conv_stages_at(0, nl - 2, &stages) == nl - 2 &&
            conv_chain_head_supported(xin, stages, last->weight, last->bias)) {
            loss = conv_chain_head_cross_entropy(xin, stages, last->weight, last->bias, y, &ncorrect, &sink);
            used_head = true;
        }
    }
    if (last && !used_head) {
        Tensor h;
        if (hidden) {   // Linear + ReLU + Linear + cross-entropy: two launches per step
            Tensor xh = seq->forward_prefix(xin, nl - 3);
            if (mlp_tail_supported(xh, hidden->weight, hidden->bias, last->weight, last->bias)) {
                if (adam && FusedAdamScope::active()) adam->set_carry_deferred(true);   // flushed by enqueue_steps
                loss = mlp_tail_cross_entropy(xh, hidden->weight, hidden->bias, last->weight, last->bias, y, &ncorrect, &sink);
                used_head = true;
            } else {
                if (adam) adam->flush_deferred();
                h = hidden->forward_fused_relu(xh);
            }
        } else {
            // updates a previous tail step left for its successor must run with ITS counter: before this step's tick
            if (adam) adam->flush_deferred();
            h = seq->forward_prefix(xin, nl - 1);
        }
        if (!used_head) {
            if (linear_cross_entropy_supported(h, last->weight) && !last->weight.has_grad() &&
                !(last->bias.defined() && last->bias.has_grad()))
                loss = linear_cross_entropy(h, last->weight, last->bias, y, &ncorrect, &sink);
            else if (fuse_head >= 2 && linear_cross_entropy_wide_supported(h, last->weight, last->bias))
                loss = linear_cross_entropy_wide(h, last->weight, last->bias, y, &ncorrect, &sink);
            else
                loss = cross_entropy_loss(last->forward(h), y, &ncorrect, &sink);
            used_head = true;
        }
    }
    if (!used_head) {
        if (adam) adam->flush_deferred();
        loss = cross_entropy_loss(model->forward(xin), y, &ncorrect, &sink);
    }
    loss.backward();
    if (dp_tail) {
        optimizer->step();   // (nothing is left: every update ran in an epilogue, on the mean gradient, or waits for the next step's first launch)
    } else if (!(comm && optimizer->step_reduced(*comm))) {   // peer-to-peer communicator: all-reduce + Adam in one launch
        reduce_grads(*this);
        optimizer->step();
        optimizer->set_step_guard(nullptr);
    }
    optimizer->zero_grad();
    if (adam) adam->set_carry_deferred(false);
}

// this step is Linear + ReLU + Linear + cross-entropy straight on the loader's rows, every parameter trains and has no gradient yet, the
// optimizer is Adam with fused updates on, and the communicator can exchange the slices inside the launch: the two-launch step of one GPU,
// with the mean gradient in its epilogues
bool Trainer::tail_exchange_step(size_t batch) const {
    if (!comm || !comm->is_p2p() || !comm->fuse_adam || !fuse_adam || fuse_head < 2 || !sample_shape.empty()) return false;
    auto *seq = dynamic_cast<Sequential *>(model.get());
    if (!seq || !seq->fuse || seq->layers.size() != 3 || !dynamic_cast<Adam *>(optimizer.get())) return false;
    auto *l1 = dynamic_cast<Linear *>(seq->layers[0].get());
    auto *l2 = dynamic_cast<Linear *>(seq->layers[2].get());
    if (!l1 || !l2 || !dynamic_cast<ReLU *>(seq->layers[1].get()) || l1->weight.shape()[1] != 784) return false;
    for (const Tensor *p : {&l1->weight, &l1->bias, &l2->weight, &l2->bias})
        if (p->defined() && (!p->get_requires_grad() || p->has_grad())) return false;
    if (!l1->bias.defined() || !l2->bias.defined()) return false;
    return comm->tail_exchange_ok((int)batch, 784, (int)l1->weight.shape()[0], (int)l2->weight.shape()[0]);
}

void Trainer::enqueue_steps(const float *d_images, const float *d_labels, const int32_t *d_indices, int64_t n_indices,
                            size_t batch, size_t steps) {
    int64_t *state = reinterpret_cast<int64_t *>(state_->d);
    // behind a failed exchange of a peer-to-peer communicator nothing may move: the launches that apply deferred updates or tick Adam's
    // counter (the next step's first launch, the flush below) look at its error word first
    // (and every tick of Adam's counter advances the exchange's step number with it: th_ctx_set_update_guard)
    struct Guard {
        Guard(const uint32_t *w, uint32_t *step) { TH(th_ctx_set_update_guard(Device::ctx(), w, step)); }
        ~Guard() { th_ctx_set_update_guard(Device::ctx(), nullptr, nullptr); }
    } guard(comm && comm->is_p2p() ? comm->error_word() : nullptr, comm && comm->is_p2p() ? comm->step_word() : nullptr);
    if (mlp2_step(batch, d_indices ? n_indices : (int64_t)batch)) {
        // the step reads its rows where they lie: through the index vector at the device cursor (every step's log advances it), or the
        // dataset itself for the one-step epoch in index order -- no gather launch, no staging buffer
        const th_row_source rows{d_images, d_labels, d_indices, d_indices ? state + 1 : nullptr, d_indices ? n_indices : 0,
                                 d_indices ? n_indices : (int64_t)batch};
        for (size_t s = 0; s < steps; ++s) enqueue_compute(nullptr, nullptr, batch, &rows);
    } else if (d_indices == nullptr) {
        // one step over the whole dataset in index order: the "gathered" batch IS the dataset (no copy)
        enqueue_compute(const_cast<float *>(d_images), const_cast<float *>(d_labels), batch);
    } else {
        TH(th_gather_batch(Device::ctx(), d_images, d_labels, d_indices, n_indices, state + 1, (int)(batch * steps), 784, xb_->d,
                           yb_->d));
        for (size_t s = 0; s < steps; ++s) enqueue_compute(xb_->d + s * batch * 784, yb_->d + s * batch, batch);
    }
    // a tail step leaves the head's W / b updates for its successor's first launch; the last one's run here
    if (auto *adam = dynamic_cast<Adam *>(optimizer.get())) adam->flush_deferred();
}

void Trainer::drop_graphs() {
    for (auto &g : graphs_) th_graph_destroy(g.second);
    graphs_.clear();
    for (auto &g : whole_graphs_) th_graph_destroy(g.second);
    whole_graphs_.clear();
}

EpochResult Trainer::train_epoch_graph(DataLoader &loader, size_t max_steps) {
    th_ctx *ctx = Device::ctx();
    static const bool trace = std::getenv("TAPER_TRACE_EPOCH") != nullptr;   // host-side timeline of one call (stderr), a measurement probe
    const auto t_enter = std::chrono::steady_clock::now();
    auto us_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
    loader.reset();
    const MNISTDataset &ds = loader.dataset();
    const size_t n = ds.len(), bs = loader.batch_size();
    size_t nb = loader.num_batches();
    if (max_steps && max_steps < nb) nb = max_steps;
    const size_t n_full = std::min(nb, n / bs);
    // steps per replay: graph_chunk, capped so that the gathered batches of one chunk stay within 256 MB (steps that read their rows in
    // place -- mlp2_step -- gather nothing)
    const bool full_in_place = mlp2_step(bs, (int64_t)n);
    const size_t chunk = full_in_place ? std::max<size_t>(graph_chunk, 1) : std::max<size_t>(std::min<size_t>(graph_chunk, ((size_t)1 << 26) / (bs * 784)), 1);
    const float *d_img = ds.images.dptr(), *d_lab = ds.labels.dptr();
    // full batch, index order (mnist.rs:355-363 without shuffle): the gather would be an identity copy of 188 MB
    const int32_t *d_idx = (!loader.shuffled() && bs >= n && nb == 1) ? nullptr : loader.d_indices();
    // staging rows for the gathered forms: a chunk of full batches, and / or the last partial batch when it is too small for the in-place form
    const size_t rem_rows = nb > n_full ? n - n_full * bs : 0;
    const size_t stage_rows = std::max(full_in_place ? (size_t)0 : chunk * bs, (rem_rows && !mlp2_step(rem_rows, (int64_t)n)) ? rem_rows : (size_t)0);
    if (d_idx && stage_rows && (!xb_ || xb_->n < stage_rows * 784)) {   // (the zero-copy forms read the dataset in place: no staging buffers)
        drop_graphs();
        xb_ = Buffer::alloc(stage_rows * 784);
        yb_ = Buffer::alloc(stage_rows);
    }
    if (!state_) state_ = Buffer::alloc(4);
    // the step log is sized for the loader's whole epoch even when this call stops early (max_steps): the captured graphs
    // hold its address, and a short first call followed by a full epoch would otherwise drop and re-record every graph
    if (metrics_cap_ < loader.num_batches() + 1) {
        drop_graphs();
        metrics_cap_ = loader.num_batches() + 1;
        // pinned host memory: the loss kernels write {loss, n_correct} of every step straight into it (8 B per step, posted writes), and the
        // host reads it after ONE stream synchronisation -- a staged device-to-host copy costs a short run (the contract's 20 steps) ~20 us
        metrics_ = Buffer::alloc_host(2 * metrics_cap_);
    }
    // Everything a captured step bakes in as a kernel argument or as a choice of launch sequence: the dataset and index
    // buffers (two loaders over one dataset share the images but own their index vectors), the epoch length the gather
    // wraps at, the label buffer, the fusion switches, the input reshape, the communicator, the optimizer's arenas.
    // (nullptr indices = the zero-copy full-batch form, a different graph.)  Any mismatch re-records.
    std::vector<uintptr_t> key{(uintptr_t)d_img, (uintptr_t)d_lab, (uintptr_t)d_idx, (uintptr_t)n, (uintptr_t)bs, (uintptr_t)fuse_head,
                               (uintptr_t)fuse_adam, (uintptr_t)comm.get(), (uintptr_t)model.get(), (uintptr_t)optimizer.get(),
                               (uintptr_t)optimizer->flat().p_arena->d, (uintptr_t)optimizer->flat().g_arena->d, (uintptr_t)(xb_ ? xb_->d : nullptr),
                               (uintptr_t)metrics_->d};
    for (size_t d : sample_shape) key.push_back((uintptr_t)d);
    // beta1 / beta2 / eps / weight decay are kernel arguments BY VALUE (only lr and t live in device memory)
    for (float h : {optimizer->beta1(), optimizer->beta2(), optimizer->eps(), optimizer->weight_decay()}) {
        uint32_t bits;
        std::memcpy(&bits, &h, sizeof bits);
        key.push_back(bits);
    }
    key.push_back((uintptr_t)full_backward());
    key.push_back((uintptr_t)conv_chain_enabled());
    key.push_back((uintptr_t)conv_chain_head_enabled());
    if (!graphs_.empty() && graph_key_ != key) drop_graphs();

    size_t done = 0;
    // A call that is ONE graph of full batches (the contract's 20 steps; any short run repeated) replays a graph that also holds the state
    // reset in front of its steps: one host launch instead of two and no stream gap between them (3.5 us of host time in front of the
    // replay + the reset's own launch: ~5 of such a call's 250 us, r05).  Recorded below, by the first call of that length that finds its
    // plain graph in place.
    th_graph *whole = nullptr;
    if (nb == n_full && !graph_capture_failed_ && !(std::getenv("TAPER_NO_GRAPH") && std::getenv("TAPER_NO_GRAPH")[0] == '1'))
        for (auto &g : whole_graphs_)
            if (g.first == n_full) whole = g.second;
    const double us_prep = us_since(t_enter);
    if (!whole) TH(th_fill_f32(ctx, state_->d, 0.f, 4));  // step = 0, cursor = 0
    const double us_fill = us_since(t_enter);
    if (whole) {
        TH(th_graph_launch(ctx, whole));
        done = n_full;
    }
    // Graph sizes still missing for this epoch length (a short first call -- e.g. a 2-step warm-up -- only
    // records the 1-step graph; the chunk graph is added by the first call long enough to use it).
    auto have = [&](size_t steps) {
        for (auto &g : graphs_)
            if (g.first == steps) return true;
        return false;
    };
    // operator override: TAPER_NO_GRAPH=1 enqueues every step's op list eagerly instead of replaying captured graphs -- same kernels, same
    // results, one host launch per kernel (rocprofv3 on ROCm 7.2 crashes inside hipGraphLaunch once several instantiated graphs are replayed
    // back to back: profiling runs use this)
    if (!graph_capture_failed_ && std::getenv("TAPER_NO_GRAPH") && std::getenv("TAPER_NO_GRAPH")[0] == '1') graph_capture_failed_ = true;
    // operator override: TAPER_DP_EAGER=1 keeps data-parallel steps out of hipGraphs (collectives launched eagerly)
    if (comm && !graph_capture_failed_ && std::getenv("TAPER_DP_EAGER") && std::getenv("TAPER_DP_EAGER")[0] == '1')
        graph_capture_failed_ = true;
    // a binary ladder of sizes (chunk, chunk/2, chunk/4, ..., 1): whatever an epoch (or a short run: 20 steps = 16 + 4)
    // leaves over after its whole chunks replays as at most log2(chunk) graphs instead of dozens of single-step
    // launches (~10 us of host time each).  A size is recorded by the first call long enough to use it.
    static const size_t ladder_div = std::getenv("TAPER_GRAPH_LADDER") ? (size_t)std::max(2, atoi(std::getenv("TAPER_GRAPH_LADDER"))) : 2;   // probe: 4 = r01's ladder
    std::vector<size_t> want;
    for (size_t steps = chunk;; steps /= ladder_div) {
        if (steps < 1) steps = 1;
        if (!((steps > 1 && n_full < steps + 1) || have(steps)) && std::find(want.begin(), want.end(), steps) == want.end())
            want.push_back(steps);
        if (steps == 1) break;
    }
    if (!whole && !want.empty() && n_full > 0 && !graph_capture_failed_) {
        // step 0 runs eagerly (pool warm-up, has_grad mask upload); then the SAME host code
        // is run under stream capture to record the op list of 1 step and of a chunk of
        // steps (one hipGraphLaunch per chunk amortises the ~10 us host cost of a replay)
        enqueue_steps(d_img, d_lab, d_idx, (int64_t)n, bs, 1);
        done = 1;
        for (size_t steps : want) {
            TH(th_graph_begin(ctx));
            th_graph *g = nullptr;
            try {
                enqueue_steps(d_img, d_lab, d_idx, (int64_t)n, bs, steps);
                TH(th_graph_end(ctx, &g));
            } catch (const std::exception &e) {
                if (!g) th_graph_end(ctx, &g);
                if (g) th_graph_destroy(g);
                if (!comm) throw;
                // a collective that cannot be captured must not take the run down: the same op list
                // is enqueued eagerly from here on (identical results, one host launch per kernel)
                fprintf(stderr, "taper: step capture with the communicator failed (%s); running data-parallel steps eagerly\n", e.what());
                graph_capture_failed_ = true;
                break;
            }
            graphs_.emplace_back(steps, g);
        }
        std::sort(graphs_.begin(), graphs_.end(), [](const auto &x, const auto &y) { return x.first > y.first; });   // largest first
        graph_key_ = key;
    }
    // What an epoch leaves over after its whole chunks (937 steps = 7 x 128 + 41), or a short run as a whole (20 steps), would replay as one
    // ladder graph per set bit (41 = 32 + 8 + 1; 20 = 16 + 4), ~10 us of host time and a stream gap each: the first call that meets such a
    // remainder records ONE graph of exactly that many steps (the ladder graphs are complete by then; at most four distinct remainders).
    if (!whole) {
        const size_t tail = n_full - done > 0 ? (n_full - done) % chunk : 0;
        size_t exact = 0;
        for (auto &g : graphs_) exact += (g.first & (g.first - 1)) != 0 ? 1 : 0;
        if (tail > 2 && (tail & (tail - 1)) != 0 && ladder_div == 2 && !have(tail) && have(1) && exact < 4 && !graph_capture_failed_ &&
            graph_key_ == key) {
            TH(th_graph_begin(ctx));
            th_graph *g = nullptr;
            try {
                enqueue_steps(d_img, d_lab, d_idx, (int64_t)n, bs, tail);
                TH(th_graph_end(ctx, &g));
                graphs_.emplace_back(tail, g);
                std::sort(graphs_.begin(), graphs_.end(), [](const auto &x, const auto &y) { return x.first > y.first; });
            } catch (const std::exception &e) {
                // enqueue_steps had already changed host state when it failed (deferred / fused Adam bookkeeping, grad slot flags, the
                // tape): put it back to "between steps" before anything replays.  The ladder keeps serving the remainder.
                fprintf(stderr, "taper: capturing a %zu-step remainder graph failed (%s); the ladder graphs serve it\n", tail, e.what());
                if (!g) th_graph_end(ctx, &g);
                if (g) th_graph_destroy(g);
                if (auto *adam = dynamic_cast<Adam *>(optimizer.get())) {
                    adam->set_carry_deferred(false);
                    adam->drop_step_bookkeeping();
                }
                optimizer->zero_grad();
                Tape::reset();
                if (!comm) throw;   // (as the ladder capture: without a communicator nothing here is expected to fail)
            }
        }
    }
    if (!whole && done == 0 && nb == n_full && n_full >= 2 && have(n_full) && whole_graphs_.size() < 4 && !graph_capture_failed_ && graph_key_ == key) {
        TH(th_graph_begin(ctx));
        th_graph *g = nullptr;
        try {
            TH(th_fill_f32(ctx, state_->d, 0.f, 4));
            enqueue_steps(d_img, d_lab, d_idx, (int64_t)n, bs, n_full);
            TH(th_graph_end(ctx, &g));
            whole_graphs_.emplace_back(n_full, g);
            // ... and serves this very call (a second reset in front of the steps changes nothing): a graph's FIRST replay pays its
            // upload, which must not fall into a later, timed call
            TH(th_graph_launch(ctx, g));
            done = n_full;
        } catch (const std::exception &e) {
            fprintf(stderr, "taper: capturing a whole-call graph of %zu steps failed (%s); reset + replay stay two launches\n", n_full, e.what());
            if (!g) th_graph_end(ctx, &g);
            if (g) th_graph_destroy(g);
            if (auto *adam = dynamic_cast<Adam *>(optimizer.get())) {
                adam->set_carry_deferred(false);
                adam->drop_step_bookkeeping();
            }
            optimizer->zero_grad();
            Tape::reset();
            if (!comm) throw;
        }
    }
    while (done < n_full) {
        bool launched = false;
        for (auto &g : graphs_) {
            if (done + g.first <= n_full) {
                TH(th_graph_launch(ctx, g.second));
                done += g.first;
                launched = true;
                break;
            }
        }
        if (!launched) {  // no graph fits (e.g. n_full == 1 on a later call): run the step eagerly
            enqueue_steps(d_img, d_lab, d_idx, (int64_t)n, bs, 1);
            ++done;
        }
    }
    if (nb > n_full) {  // the last, partial batch (mnist.rs:373-385 keeps it)
        const size_t rem = n - n_full * bs;
        enqueue_steps(d_img, d_lab, d_idx, (int64_t)n, rem, 1);
    }
    loader.advance(std::min(n, nb * bs));

    const double us_enqueued = us_since(t_enter);
    Device::sync();
    check_comm();
    if (trace) fprintf(stderr, "taper trace: train_epoch_graph %zu steps: state reset enqueued at %.1f - %.1f us, all enqueued after %.1f us, stream idle after %.1f us\n", nb, us_prep, us_fill, us_enqueued, us_since(t_enter));
    const float *mt = metrics_->d;   // host-visible (th_host_malloc); every step's entry has landed once the stream is idle
    EpochResult r;
    r.num_batches = nb;
    float total_loss = 0.f;
    for (size_t s = 0; s < nb; ++s) {
        const size_t b = (s < n_full) ? bs : n - n_full * bs;
        const float acc = mt[2 * s + 1] / (float)b;       // loss.rs:289
        r.total_correct += (size_t)(acc * (float)b);      // train.rs:117 (Q13)
        r.total_samples += b;
        total_loss += mt[2 * s];
        r.losses.push_back(mt[2 * s]);
        r.ncorrect.push_back(mt[2 * s + 1]);
    }
    r.avg_loss = total_loss / (float)nb;
    r.accuracy = (float)r.total_correct / (float)r.total_samples;
    return r;
}

Trainer::~Trainer() { drop_graphs(); }

}  // namespace taper

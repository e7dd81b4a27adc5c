// train_extra.cpp -- the parts of the reference around the training step that SURVEY.md 8(f) ranks
// "next": bce_loss / cross_entropy_loss_onehot / Dropout (src/loss.rs, src/nn.rs), AdamW and the
// four LR schedulers (src/optim.rs:130-352), Metrics / Trainer::fit / the text checkpoint
// (src/train.rs:9-71, 175-292).  Host logic in f32 like the reference; device work through the C ABI.
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <fstream>
#include <limits>
#include <sstream>

#include "taper.h"

namespace taper {

#define TH(call) th_check((call), #call)

// ---------------------------------------------------------------- losses
Tensor bce_loss(const Tensor &pred, const Tensor &targets) {  // loss.rs:6-73
    TAPER_ASSERT(pred.len() == targets.len(), "bce_loss: predictions and targets must match in length");
    const size_t n = pred.len();
    Tensor out = Tensor::empty({1});   // Tensor::scalar shape (tensor.rs:480)
    TH(th_bce_fwd(Device::ctx(), pred.dptr(), targets.dptr(), n, out.dptr()));
    if (pred.get_requires_grad() || targets.get_requires_grad()) {
        out.set_requires_grad(true);
        Tensor p = pred, t = targets, o = out;
        Tape::push(out, true, [p, t, o, n]() {
            if (!o.has_grad()) return;
            int mask = 0;
            bool none;
            float *gp = nullptr, *gt = nullptr;
            if (p.get_requires_grad()) { gp = p.grad_for_write(&none); if (!none) mask |= 1; }
            if (t.get_requires_grad()) { gt = t.grad_for_write(&none); if (!none) mask |= 2; }
            TH(th_bce_bwd(Device::ctx(), p.dptr(), t.dptr(), o.grad_dptr(), n, gp, gt, mask));
        });
    }
    return out;
}

Tensor cross_entropy_loss_onehot(const Tensor &logits, const Tensor &targets) {  // loss.rs:201-245
    TAPER_ASSERT(logits.shape() == targets.shape(), "Logits and targets shapes must match");
    TAPER_ASSERT(logits.shape().size() == 2, "Must be 2D tensors");
    const int b = (int)logits.shape()[0], c = (int)logits.shape()[1];
    th_ctx *ctx = Device::ctx();
    // forward value: -sum(targets * log_softmax(logits)) / B  (loss.rs:214-222); the reference computes
    // it outside the tape and records ONE node on logits
    Tensor logp = Tensor::empty(logits.shape()), prod = Tensor::empty(logits.shape()), loss = Tensor::empty({1});
    TH(th_log_softmax_fwd(ctx, logits.dptr(), logp.dptr(), b, c));
    TH(th_mul(ctx, targets.dptr(), logp.dptr(), prod.dptr(), prod.len()));
    TH(th_sum_all(ctx, prod.dptr(), loss.dptr(), prod.len(), -(float)b));
    if (logits.get_requires_grad()) {
        loss.set_requires_grad(true);
        Tensor lg = logits, lp = logp, t = targets, out = loss;
        Tape::push(loss, true, [lg, lp, t, out, b, c]() {
            if (!out.has_grad()) return;
            bool none;
            float *g = lg.grad_for_write(&none);
            TH(th_xent_onehot_bwd(Device::ctx(), lp.dptr(), t.dptr(), out.grad_dptr(), b, c, g, none ? 0 : 1));
        });
    }
    return loss;
}

// ---------------------------------------------------------------- Dropout
Dropout::Dropout(float p, uint64_t seed) : p_(p), seed_(seed) {
    TAPER_ASSERT(p >= 0.0f && p <= 1.0f, "Dropout probability must be between 0 and 1");  // nn.rs:782-785
}

Tensor Dropout::forward(const Tensor &x) const {  // nn.rs:799-822
    if (!training_ || p_ == 0.0f) return x;
    if (p_ == 1.0f) return Tensor::zeros(x.shape());
    Tensor mask = Tensor::empty(x.shape());
    // a fresh stream per call, like successive draws from one RNG
    TH(th_dropout_mask(Device::ctx(), mask.dptr(), mask.len(), p_, seed_ + 0x9E3779B97F4A7C15ull * ++calls_));
    last_mask_ = mask;
    return x * mask;
}

// ---------------------------------------------------------------- AdamW / schedulers
void AdamW::step() {  // optim.rs:147-168
    const float wd = adam.weight_decay(), lr = adam.get_lr();
    FlatParams &fp = adam.flat();
    // every parameter decays, grad or not (the loop at optim.rs:153-158 is unconditional); the
    // padding between arena slices is zero and stays zero
    if (wd > 0.0f) TH(th_scale(Device::ctx(), fp.p_arena->d, (size_t)fp.total, 1.0f - lr * wd));
    adam.set_weight_decay(0.0f);
    adam.step();
    adam.set_weight_decay(wd);
}

void StepLR::step(const float *) {  // optim.rs:209-214
    epoch_ += 1;
    if (epoch_ % step_size_ == 0) lr_ *= gamma_;
}

void CosineAnnealingLR::step(const float *) {  // optim.rs:275-283
    epoch_ += 1;
    const float progress = (float)epoch_ / (float)t_max_;
    const float cos_val = (1.0f + std::cos(progress * 3.14159274101257324f)) / 2.0f;   // std::f32::consts::PI
    lr_ = min_ + (base_ - min_) * cos_val;
}

ReduceLROnPlateau::ReduceLROnPlateau(float initial_lr, float factor, size_t patience, float min_lr, const std::string &mode)
    : lr_(initial_lr), factor_(factor), patience_(patience), min_lr_(min_lr), mode_min_(mode == "min"),
      best_(mode == "min" ? std::numeric_limits<float>::infinity() : -std::numeric_limits<float>::infinity()) {}

void ReduceLROnPlateau::step(const float *metric) {  // optim.rs:325-347
    if (!metric) return;
    const bool improved = mode_min_ ? *metric < best_ : *metric > best_;
    if (improved) {
        best_ = *metric;
        counter_ = 0;
        return;
    }
    if (++counter_ >= patience_) {
        lr_ = std::fmax(lr_ * factor_, min_lr_);
        counter_ = 0;
        if (verbose) printf("Reducing learning rate to %.6f\n", lr_);
    }
}

// ---------------------------------------------------------------- Metrics
static std::string fmt(const char *f, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

std::string Metrics::last_line() const {  // train.rs:29-45
    if (train_loss.empty() || train_acc.empty() || val_loss.empty() || val_acc.empty()) return "";
    return fmt("Train Loss: %.4f | Train Acc: %.2f%% | Val Loss: %.4f | Val Acc: %.2f%%", train_loss.back(), train_acc.back() * 100.0f,
               val_loss.back(), val_acc.back() * 100.0f);
}

std::string Metrics::summary() const {  // train.rs:47-70
    const std::string bar(50, '=');
    std::string s = "\nTraining Summary:\n" + bar + "\n";
    if (!train_acc.empty()) {
        float best_t = 0.f, best_v = 0.f;   // fold(0.0, f32::max)
        for (float a : train_acc) best_t = std::fmax(best_t, a);
        for (float a : val_acc) best_v = std::fmax(best_v, a);
        s += fmt("Best Train Accuracy: %.2f%%\nBest Val Accuracy: %.2f%%\nFinal Train Accuracy: %.2f%%\nFinal Val Accuracy: %.2f%%\n",
                 best_t * 100.0f, best_v * 100.0f, train_acc.back() * 100.0f, val_acc.back() * 100.0f);
        if (!epoch_times.empty()) {
            float total = 0.f;
            for (float t : epoch_times) total += t;
            s += fmt("Total Training Time: %.2fs\nAverage Epoch Time: %.2fs\n", total, total / (float)epoch_times.size());
        }
    }
    return s + bar + "\n";
}

void Metrics::print_last() const {
    const std::string l = last_line();
    if (!l.empty()) printf("%s\n", l.c_str());
}
void Metrics::plot_summary() const { fputs(summary().c_str(), stdout); }

// ---------------------------------------------------------------- fit
void Trainer::fit(DataLoader &train_loader, DataLoader &val_loader, size_t epochs, bool verbose, bool graph) {  // train.rs:175-261
    printf("Starting training for %zu epochs\n%s\n", epochs, std::string(60, '=').c_str());
    for (size_t epoch = 0; epoch < epochs; ++epoch) {
        const auto t0 = std::chrono::steady_clock::now();
        if (verbose) printf("\nEpoch %zu/%zu\n", epoch + 1, epochs);
        const EpochResult tr = graph ? train_epoch_graph(train_loader) : train_epoch(train_loader);
        const EpochResult va = evaluate(val_loader);
        if (scheduler) {  // train.rs:209-213
            scheduler->step(&va.avg_loss);
            optimizer->set_lr(scheduler->get_lr());
        }
        metrics.train_loss.push_back(tr.avg_loss);
        metrics.train_acc.push_back(tr.accuracy);
        metrics.val_loss.push_back(va.avg_loss);
        metrics.val_acc.push_back(va.accuracy);
        metrics.epoch_times.push_back(std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count());
        if (verbose) {
            printf("\nEpoch %zu - Train Loss: %.4f | Train Acc: %.2f%% | Val Loss: %.4f | Val Acc: %.2f%% | Time: %.2fs\n", epoch + 1,
                   tr.avg_loss, tr.accuracy * 100.0f, va.avg_loss, va.accuracy * 100.0f, metrics.epoch_times.back());
            if (scheduler) printf("   Learning Rate: %.6f\n", scheduler->get_lr());
        }
        if (va.accuracy > 0.99f) {  // train.rs:247-250
            printf("\nReached 99%% validation accuracy! Stopping early.\n");
            break;
        }
    }
    metrics.plot_summary();
    fflush(stdout);
}

// ---------------------------------------------------------------- checkpoint
std::string format_f32_display(float v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    // shortest round-trip digits (what Rust's Display uses), then laid out positionally: the digits,
    // zero-padded up to the decimal point or behind "0." -- never an exponent
    char buf[64];
    const auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
    std::string sci(buf, r.ptr), out;
    size_t i = 0;
    if (sci[0] == '-') {
        out = "-";
        i = 1;
    }
    const size_t e = sci.find('e');
    std::string digits;
    for (size_t k = i; k < e; ++k)
        if (sci[k] != '.') digits += sci[k];
    const int exp10 = std::stoi(sci.substr(e + 1));   // value = d.ddd x 10^exp10
    const int point = exp10 + 1;                        // digits before the decimal point
    if (point <= 0) {
        out += "0." + std::string((size_t)(-point), '0') + digits;
    } else if ((size_t)point >= digits.size()) {
        out += digits + std::string((size_t)point - digits.size(), '0');
    } else {
        out += digits.substr(0, (size_t)point) + "." + digits.substr((size_t)point);
    }
    return out;
}

static void write_tensor_block(std::ostream &f, const Shape &shape, const std::vector<float> &data) {  // train.rs:273-286
    f << shape.size();
    for (size_t d : shape) f << ' ' << d;
    f << '\n';
    for (float v : data) f << format_f32_display(v) << '\n';
}

static float parse_f32(const std::string &tok) {
    if (tok == "NaN") return std::numeric_limits<float>::quiet_NaN();
    if (tok == "inf") return std::numeric_limits<float>::infinity();
    if (tok == "-inf") return -std::numeric_limits<float>::infinity();
    return std::stof(tok);
}

static std::vector<float> read_tensor_block(std::istream &f, const Shape &expect, const std::string &what) {
    size_t nd = 0;
    TAPER_ASSERT((bool)(f >> nd), what + ": truncated file");
    Shape shape(nd);
    size_t n = 1;
    for (size_t &d : shape) {
        TAPER_ASSERT((bool)(f >> d), what + ": truncated shape");
        n *= d;
    }
    TAPER_ASSERT(shape == expect, what + ": shape mismatch");
    std::vector<float> v(n);
    std::string tok;
    for (float &x : v) {
        TAPER_ASSERT((bool)(f >> tok), what + ": truncated data");
        x = parse_f32(tok);
    }
    return v;
}

void Trainer::save_checkpoint(const std::string &path) const {  // train.rs:264-292
    std::ofstream f(path);
    TAPER_ASSERT(f.good(), "save_checkpoint: cannot create " + path);
    const std::vector<Tensor> params = model->parameters();
    f << params.size() << '\n';
    for (const Tensor &p : params) write_tensor_block(f, p.shape(), p.data());
    TAPER_ASSERT(f.good(), "save_checkpoint: write failed");
}

void Trainer::load_checkpoint(const std::string &path) {
    std::ifstream f(path);
    TAPER_ASSERT(f.good(), "load_checkpoint: cannot open " + path);
    std::vector<Tensor> params = model->parameters();
    size_t n = 0;
    TAPER_ASSERT((bool)(f >> n) && n == params.size(), "load_checkpoint: parameter count mismatch");
    for (size_t i = 0; i < n; ++i) params[i].set_data(read_tensor_block(f, params[i].shape(), "load_checkpoint: parameter " + std::to_string(i)));
}

void Trainer::save_optimizer_state(const std::string &path) const {
    std::ofstream f(path);
    TAPER_ASSERT(f.good(), "save_optimizer_state: cannot create " + path);
    const Adam &o = *optimizer;
    f << "adam " << o.t() << ' ' << format_f32_display(o.get_lr()) << ' ' << format_f32_display(o.beta1()) << ' '
      << format_f32_display(o.beta2()) << ' ' << format_f32_display(o.eps()) << ' ' << format_f32_display(o.weight_decay()) << '\n';
    const std::vector<float> m = o.m(), v = o.v();
    size_t off = 0;
    for (const Tensor &p : optimizer->flat().params) {
        write_tensor_block(f, p.shape(), std::vector<float>(m.begin() + (long)off, m.begin() + (long)(off + p.len())));
        write_tensor_block(f, p.shape(), std::vector<float>(v.begin() + (long)off, v.begin() + (long)(off + p.len())));
        off += p.len();
    }
}

void Trainer::load_optimizer_state(const std::string &path) {
    std::ifstream f(path);
    TAPER_ASSERT(f.good(), "load_optimizer_state: cannot open " + path);
    std::string tag, lr, b1, b2, eps, wd;
    int t = 0;
    TAPER_ASSERT((bool)(f >> tag >> t >> lr >> b1 >> b2 >> eps >> wd) && tag == "adam", "load_optimizer_state: bad header");
    std::vector<float> m, v;
    for (const Tensor &p : optimizer->flat().params) {
        const std::vector<float> pm = read_tensor_block(f, p.shape(), "load_optimizer_state: m"), pv = read_tensor_block(f, p.shape(), "load_optimizer_state: v");
        m.insert(m.end(), pm.begin(), pm.end());
        v.insert(v.end(), pv.begin(), pv.end());
    }
    // beta1 / beta2 / eps are constructor arguments of the optimizer (optim.rs:54-70): a state file written by an optimizer
    // with other values does not describe this one's moments
    TAPER_ASSERT(parse_f32(b1) == optimizer->beta1() && parse_f32(b2) == optimizer->beta2() && parse_f32(eps) == optimizer->eps(),
                 "load_optimizer_state: beta1/beta2/eps in " + path + " (" + b1 + ", " + b2 + ", " + eps + ") differ from the optimizer's");
    optimizer->set_lr(parse_f32(lr));
    optimizer->set_weight_decay(parse_f32(wd));   // a kernel argument by value: captured steps are re-recorded (graph key)
    drop_graphs();
    optimizer->load_state(t, m, v);
}

}  // namespace taper

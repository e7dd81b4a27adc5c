// comm.cpp -- the data-parallel communicator of the host mirror (new: the reference is one process).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "nn_internal.h"

namespace taper {
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

bool Communicator::wide_exchange_ok(int batch, int in_features, int classes, int conv_c) const {
    if (!p2p_ || !inkernel) return false;
    if (n_ranks == 1) return true;     // (the mean over one rank: the single-GPU launch)
    return th_wide_head_grads_dp_supported(comm_, Device::ctx(), batch, in_features, classes, conv_c) != 0;
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

int Communicator::exchange_form() const {
    int f = 0;
    TH(th_comm_exchange_form(comm_, &f));
    return f;
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

}  // namespace taper

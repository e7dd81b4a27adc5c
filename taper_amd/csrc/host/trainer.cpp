// trainer.cpp -- dataset, loader and the step drivers (src/data/mnist.rs, src/train.rs, examples/train_mnist*.rs).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "nn_internal.h"
#include "../../../include/taper_hip_debug.h"   // (th_comm_debug_dump: the post-mortem under TAPER_DP_POSTMORTEM)

namespace taper {
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
    if (comm && comm->is_p2p() && comm->failed()) {
        if (const char *e = std::getenv("TAPER_DP_POSTMORTEM"); e && atoi(e)) th_comm_debug_dump(comm->handle(), Device::ctx());
        int d[4] = {-1, 0, 0, 0};
        th_comm_timeout_detail(comm->handle(), Device::ctx(), d);
        std::string where;
        if (d[0] >= 0) {
            where = "; first wait to run out: slot " + std::to_string(d[0]) + " of exchange step " + std::to_string(d[1]) + ", nothing from rank(s)";
            for (int s = 0; s < comm->n_ranks; ++s)
                if (d[2] >> s & 1) where += " " + std::to_string(s);
        }
        throw Error("data-parallel all-reduce timed out waiting for a peer (rank " + std::to_string(comm->rank) + " of " +
                    std::to_string(comm->n_ranks) + "): no update was applied from that step on; the replicas are out of step" + where);
    }
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
    const bool dp_tail = comm && !rows && (tail_exchange_step(batch) || chain_exchange_step(shape_input(Tensor::from_device(d_xb, {batch, 784}), sample_shape)));
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

// this step is the simple CNN's (conv rows ending in a pooled map, Flatten, Linear, cross-entropy: two launches), the optimizer is Adam with
// fused updates on, and the communicator can exchange the batch sums inside their launch: the single-GPU step with the mean gradient in the
// second launch's epilogues.  The same conditions the step's own choice of form checks (enqueue_compute), decided before the scopes open.
bool Trainer::chain_exchange_step(const Tensor &xin) const {
    if (!comm || !comm->is_p2p() || !comm->fuse_adam || !fuse_adam || fuse_head < 2 || xin.shape().size() != 4) return false;
    auto *seq = dynamic_cast<Sequential *>(model.get());
    if (!seq || !seq->fuse || !conv_chain_enabled() || !conv_chain_head_enabled() || !dynamic_cast<Adam *>(optimizer.get())) return false;
    const size_t nl = seq->layers.size();
    if (nl < 3) return false;
    auto *last = dynamic_cast<Linear *>(seq->layers.back().get());
    auto *fl = dynamic_cast<Flatten *>(seq->layers[nl - 2].get());
    if (!last || !fl || fl->start_dim != 1) return false;
    std::vector<ConvStage> stages;
    if (seq->conv_stages_at(0, nl - 2, &stages) != nl - 2 || !conv_chain_head_supported(xin, stages, last->weight, last->bias)) return false;
    if (xin.conv_chain_head_supported(stages, (int)last->weight.shape()[0]) != 2) return false;      // the compiled simple chain: its tick carries the step number
    return comm->wide_exchange_ok((int)xin.shape()[0], (int)last->weight.shape()[1], (int)last->weight.shape()[0], (int)stages.back().weight.shape()[0]);
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
    whole_capture_failed_.clear();   // (another key: another launch sequence may capture)
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
    // what the choice of launch sequence reads of the parameters themselves (Trainer::mlp2_step, tail_exchange_step: every parameter trains and
    // has no gradient yet): freezing a layer behind a capture must re-record, not replay the other sequence's graphs
    key.push_back((uintptr_t)full_in_place);
    {
        uintptr_t h = 1469598103934665603ull;
        for (const Tensor &p : optimizer->flat().params) h = (h ^ (uintptr_t)(p.get_requires_grad() ? 2 : 1)) * 1099511628211ull;
        key.push_back(h);
    }
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
    if (!whole && done == 0 && nb == n_full && n_full >= 2 && have(n_full) && whole_graphs_.size() < 4 && !graph_capture_failed_ && graph_key_ == key &&
        std::find(whole_capture_failed_.begin(), whole_capture_failed_.end(), n_full) == whole_capture_failed_.end()) {
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
            whole_capture_failed_.push_back(n_full);   // (latched: a later call of this length does not try, print and fail again)
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

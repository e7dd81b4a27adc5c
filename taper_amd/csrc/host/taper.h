// taper.h -- host side of the MI355X backend: a C++ mirror of the reference's
// Rust API surface (Tensor / Tape / nn::Module / loss / optim / data / train)
// that owns the tape and the op-record list and reaches the GPU ONLY through
// the C ABI of include/taper_hip.h -- exactly what a Rust `extern "C"` block
// in the taper crate would bind.  No HIP headers here: this
// library is built with plain g++.
//
// Names, argument meaning and error behaviour follow the reference:
//   src/tensor.rs, src/ops.rs, src/tape.rs, src/nn.rs, src/activation.rs,
//   src/loss.rs, src/optim.rs, src/data/mnist.rs, src/train.rs.
// The reference panics (assert!/panic!) on misuse; here that is taper::Error.
#pragma once

#include <algorithm>
#include <cstdint>
#include <functional>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/taper_hip.h"

namespace taper {

struct Error : std::runtime_error {
    explicit Error(const std::string &m) : std::runtime_error(m) {}
};
void th_check(int rc, const char *what);  // throws Error(th_last_error())
#define TAPER_ASSERT(cond, msg)                       \
    do {                                              \
        if (!(cond)) throw ::taper::Error(msg);       \
    } while (0)

using Shape = std::vector<size_t>;  // <= 4 dims (SmallVec<[usize;4]>, tensor.rs:240)
size_t numel(const Shape &s);

// One context (GPU + stream + pool) per host thread, like the thread-local
// tape (tape.rs:6-9).  Device id: set_device() > $TAPER_DEVICE > $LOCAL_RANK > 0.
class Device {
   public:
    static th_ctx *ctx();
    static void set_device(int id);
    static int device_id();
    static void sync();
    static void shutdown();
    static float *ones1();  // persistent device constant [1.0f]: the implicit upstream grad of a scalar root
};

// Device storage.  `owned` buffers return to the ctx pool on destruction;
// views (optimizer arenas, batch staging) do not.
struct Buffer {
    float *d = nullptr;
    size_t n = 0;
    bool owned = true;
    bool host_pinned = false;        // th_host_malloc: pinned host memory the device writes through the same pointer
    bool fine_grained = false;       // th_malloc_finegrained: device memory that is coherent across agents (peer-readable gradient arenas)
    std::shared_ptr<Buffer> parent;  // keeps an arena alive for views
    Buffer() = default;
    Buffer(const Buffer &) = delete;
    ~Buffer();
    static std::shared_ptr<Buffer> alloc(size_t n);
    static std::shared_ptr<Buffer> alloc_host(size_t n);   // device-visible pinned host memory (read on the host after a stream sync)
    static std::shared_ptr<Buffer> alloc_finegrained(size_t n);
    static std::shared_ptr<Buffer> view(const std::shared_ptr<Buffer> &parent, size_t offset, size_t n);
    static std::shared_ptr<Buffer> borrow(float *d, size_t n);
};

// grad: Arc<RwLock<Option<Vec<f32>>>> (tensor.rs:241)
struct GradSlot {
    std::shared_ptr<Buffer> buf;  // storage; may be a view into an optimizer's flat arena
    bool has = false;             // Some / None
    bool known_zero = false;      // arena slice currently all-zero (DP all-reduce of grad-less params)
    bool shared_const = false;    // buf is the ctx-wide read-only [1.0] (set by backward() on a scalar root)
    bool buf_is_arena = false;    // buf is a view into an optimizer's flat grad arena (never re-pointed)
    // Trainer-internal peephole (PoolBiasScope): the slot belongs to the output of a Conv2dReLU whose only trainable
    // input is its bias (Q2); a max-pool consuming it then leaves {gradient of the POOLED output, pooled output} here
    // instead of scattering, and the conv's backward sums the bias gradient straight from them
    bool relu_output = false;     // the tensor is the output of a fused Linear+ReLU node
    bool premasked = false;       // its gradient already carries that ReLU's mask (th_linear_xent_head_masked; th_maxpool2d_relu_bwd)
    std::shared_ptr<Buffer> dz_colpart;   // premasked output of a fused Linear + ReLU: [dz_colpart_rows][features] column sums of that gradient by row block,
    int dz_colpart_rows = 0;              // left by the consumer's dX product (th_linear_bwd_adam_ex2): th_colsum over the rows is this layer's bias gradient
    std::shared_ptr<Buffer> plane_sums;   // full backward, premasked conv output: [n][c] sums of the planes of that gradient (the bias gradient's rows)
    bool wants_pooled = false;
    std::shared_ptr<Buffer> pooled_dy, pooled_y;
    std::shared_ptr<Buffer> pooled_cnt;   // [n][c] counts of elements > 0 per plane, left by a global average pool's forward
    int pooled_n = 0, pooled_c = 0, pooled_hw = 0;
    bool pooled_avg = false;      // the consumer was a global average pool: pooled_dy is [n][c], pooled_y the conv output itself
    // Trainer-internal peephole (PoolBiasScope): the tensor is the (flattened) output of a bias-only Conv2dReLU + max-pool; all its backward
    // needs of the gradient are the per-column sums of dX * [x > 0], which the wide classifier head leaves here instead of writing dX
    bool wants_colsum = false;
    std::shared_ptr<Buffer> colsum;   // [numel / batch]
    std::shared_ptr<GradSlot> colsum_bias;   // that conv's bias (grad slot) and geometry: the head can finish it in its own launch (th_linear_xent_wide_fused)
    int colsum_c = 0, colsum_hw = 0;
    bool colsum_done = false;                // ... and did
    // Trainer-internal peephole (PoolBiasScope): the tensor is the (flattened) [n, c] output of a bias-only Conv2dReLU + GLOBAL AVERAGE pool:
    // {counts of outputs > 0 per plane, that conv's bias slot, plane size} -- a classifier launch that forms the tensor's gradient anyway
    // (th_mlp3_xent) can finish the conv's bias there and say so
    std::shared_ptr<Buffer> gapfin_cnt;
    std::shared_ptr<GradSlot> gapfin_bias;
    int gapfin_hw = 0;
    bool gapfin_done = false;
};

struct ConvStage;   // one Conv2dReLU row of a conv chain (below)
class Tensor {
   public:
    Tensor() = default;
    Tensor(const std::vector<float> &data, const Shape &shape);  // tensor.rs:470 (uploads)
    static Tensor scalar(float v);                                // tensor.rs:480
    static Tensor empty(const Shape &shape);                      // uninitialised device storage
    static Tensor zeros(const Shape &shape);
    static Tensor from_device(float *d, const Shape &shape);      // non-owning view of device memory
    static Tensor randn(const Shape &shape, uint64_t seed = 0);   // ops.rs:301-309 (seeded)

    Tensor requires_grad() const;        // tensor.rs:484-487 (builder style)
    bool get_requires_grad() const { return requires_grad_; }
    void set_requires_grad(bool on) { requires_grad_ = on; }
    bool defined() const { return (bool)data_; }
    const Shape &shape() const { return shape_; }
    size_t len() const { return data_ ? data_->n : 0; }
    float *dptr() const { return data_->d; }
    std::vector<float> data() const;     // tensor.rs:494 (D2H copy, synchronises)
    void set_data(const std::vector<float> &v);
    bool has_grad() const { return grad_ && grad_->has; }
    std::vector<float> grad() const;     // tensor.rs:505-518; empty vector when None
    float *grad_dptr() const { return has_grad() ? grad_->buf->d : nullptr; }
    void set_grad(const std::vector<float> &g);  // `*w.grad.write().unwrap() = Some(..)` (optim.rs:372)
    size_t tape_node() const { return tape_node_ ? *tape_node_ : 0; }

    void backward() const;               // tensor.rs:520-529
    void zero_grad() const;              // tensor.rs:531-533

    // ops (src/ops.rs, src/tensor.rs)
    Tensor operator+(const Tensor &o) const;   // ops.rs:8-51
    Tensor operator-(const Tensor &o) const;   // ops.rs:377-416
    Tensor operator*(const Tensor &o) const;   // ops.rs:53-120
    Tensor operator/(const Tensor &o) const;   // ops.rs:440-496
    Tensor matmul(const Tensor &o) const;      // ops.rs:200-298
    Tensor relu() const;                       // ops.rs:312-374
    Tensor transpose() const;                  // tensor.rs:544-591
    Tensor sigmoid() const;                    // tensor.rs:594-634
    Tensor add_broadcast(const Tensor &o) const;       // tensor.rs:636-704
    Tensor sub_broadcast_rows(const Tensor &o) const;  // tensor.rs:707-770
    Tensor mean() const;                       // tensor.rs:772-800
    Tensor reshape(const Shape &s) const;      // tensor.rs:803-840
    Tensor view(const Shape &s) const { return reshape(s); }
    Tensor flatten(size_t start_dim) const;    // tensor.rs:843-858
    Tensor squeeze(int dim = -1) const;        // tensor.rs:861-877
    Tensor unsqueeze(size_t dim) const;        // tensor.rs:880-887
    Tensor sum(int dim = -1, bool keepdim = false) const;  // tensor.rs:890-1018 (dim: 2-D only, Q14)
    std::pair<Tensor, Tensor> max(int dim = -1) const;     // tensor.rs:1021-1083
    Tensor argmax(int dim = -1) const;         // tensor.rs:1086-1088
    Tensor exp() const;                        // tensor.rs:1091-1133
    Tensor log() const;                        // tensor.rs:1136-1169
    Tensor pow(float e) const;                 // tensor.rs:1172-1206
    Tensor sqrt() const { return pow(0.5f); }  // tensor.rs:1209-1211
    // helpers of the grouped convolution (nn.rs:859-1014): plain data copies, no tape nodes (like the reference's Tensor::new)
    Tensor slice_channels(size_t start, size_t end) const;         // nn.rs:862-886, 4-D [N,C,H,W] -> [N,end-start,H,W]
    Tensor slice_output_channels(size_t start, size_t end) const;  // nn.rs:889-914, 4-D weight rows
    Tensor slice_1d(size_t start, size_t end) const;               // nn.rs:917-925
    static Tensor cat(const std::vector<Tensor> &tensors, size_t dim);  // nn.rs:928-1014 (2-D dim 0/1, 4-D dim 1)
    // conv / pool (tensor.rs:1221-1660).  bias may be undefined (None).
    Tensor conv2d(const Tensor &weight, const Tensor &bias, std::pair<int, int> stride, std::pair<int, int> padding,
                  std::pair<int, int> dilation, bool relu = false) const;
    Tensor conv2d_relu(const Tensor &weight, const Tensor &bias, std::pair<int, int> stride, std::pair<int, int> padding,
                       std::pair<int, int> dilation) const;
    // Conv2dReLU (3x3, stride 1) + MaxPool2d(2, 2) as ONE launch that writes only the pooled tensor (th_conv3x3_pool2_fwd).
    // Trainer-internal: valid in faithful mode (Q2: the conv's only gradient is its bias, taken from the pooled tensors)
    // when nobody else reads the conv output.  conv2d_relu_maxpool2_supported tells whether the pair qualifies.
    bool conv2d_relu_maxpool2_supported(const Tensor &weight, const Tensor &bias, std::pair<int, int> padding) const;
    Tensor conv2d_relu_maxpool2(const Tensor &weight, const Tensor &bias, std::pair<int, int> padding) const;
    // ... and Conv2dReLU(3x3, stride 1) -> global average pool (AdaptiveAvgPool2d((1, 1))): one launch writes the [n, c, 1, 1] plane means
    // and each plane's count of outputs > 0 (the bias gradient's only need, th_bias_grad_counts_adam); the map is never stored
    bool conv2d_relu_gap_supported(const Tensor &weight, const Tensor &bias, std::pair<int, int> padding) const;
    Tensor conv2d_relu_gap(const Tensor &weight, const Tensor &bias, std::pair<int, int> padding) const;
    // A whole run of Conv2dReLU(3x3, stride 1, pad 1) layers with their pools as ONE launch (th_conv_chain_fwd: one image per workgroup, the
    // maps stay in LDS).  Trainer-internal like the pair fusions above (faithful mode, Q2: only the LAST conv's bias ever receives a
    // gradient -- no conv hands one to its input -- and it is taken from the chain's output).  post: TH_CHAIN_* of include/taper_hip.h.
    int conv_chain_supported(const std::vector<ConvStage> &stages) const;   // 0, or the compiled instance's id
    Tensor conv_chain(const std::vector<ConvStage> &stages) const;
    // ... with the classifier behind it (Flatten + Linear + cross-entropy) done row by row in the same launch (th_conv_chain_head_fwd):
    // returns the pooled map, records NO tape node -- conv_chain_head_cross_entropy (below) owns the step's backward
    int conv_chain_head_supported(const std::vector<ConvStage> &stages, int classes) const;
    Tensor conv_chain_head(const std::vector<ConvStage> &stages, const th_chain_head &head) const;
    // ... and the reference CNN's front with its three-layer classifier's rows in the chain launch (th_conv_chain_mlp3_xent): runs BOTH launches
    // of the step, returns the plane means, records NO tape node -- conv_chain_mlp3_cross_entropy (below) owns the step's backward
    int conv_chain_mlp3_supported(const std::vector<ConvStage> &stages, int h1, int h2, int classes) const;
    Tensor conv_chain_mlp3(const std::vector<ConvStage> &stages, float *d_cnt, const float *d_targets, const th_mlp3_layer *layers, float *d_dx,
                           float *d_loss, float *d_ncorrect, float *d_metrics, int64_t capacity, int64_t *d_state, int64_t advance, int32_t *d_tick,
                           const th_mlp3_gap *gap) const;
    Tensor max_pool2d(std::pair<int, int> kernel, std::pair<int, int> stride /* {0,0} = None */,
                      std::pair<int, int> padding) const;
    Tensor avg_pool2d(std::pair<int, int> kernel, std::pair<int, int> stride, std::pair<int, int> padding) const;
    // fused Linear (nn.rs:54-60): y = x . W^T + b, optional fused ReLU; ONE tape node
    Tensor linear(const Tensor &weight, const Tensor &bias, bool relu = false) const;

    // internals shared with nn/optim
    float *grad_for_write(bool *was_none) const;  // storage of the grad slot; *was_none tells the caller to overwrite
    float *grad_accum_ptr() const;                // lazily zero-filled slot (ops.rs:126-129)
    std::shared_ptr<Buffer> data_;
    Shape shape_;
    std::shared_ptr<GradSlot> grad_;
    bool requires_grad_ = false;
    std::shared_ptr<size_t> tape_node_;
};

struct ConvStage {   // Tensor::conv_chain: weight [c_out, c_in, 3, 3], bias [c_out], then TH_CHAIN_NONE / _MAXPOOL2 / _GLOBAL_AVG
    Tensor weight, bias;
    int post;
};

// ---- tape (src/tape.rs) ----------------------------------------------------
class Tape {
   public:
    static void reset();                                   // tape.rs:43-49
    static size_t len();
    static void push(const Tensor &out, bool any_input_requires_grad, std::function<void()> backward_fn);  // tape.rs:51-101
    static void backward(size_t final_node_id);            // tape.rs:106-127
    // false (default): ids are 1-based so every recorded node can be a root
    // (the reference tests' intent); true: the literal tensor.rs:524-528
    // behaviour where node id 0 doubles as "no node" (quirk Q1).
    static void set_compat_zero_sentinel(bool on);
    static bool compat_zero_sentinel();
};

// conv gradient mode: false = faithful (the reference cuts the tape at
// im2col / transpose_4d, quirk Q2: conv weights and conv inputs never get
// gradients); true = full_backward extension.
void set_full_backward(bool on);
bool full_backward();
// Trainer steps launch the convolutional front of a Sequential as ONE kernel where an instance is compiled for it (Tensor::conv_chain);
// false = layer by layer (the measurement probe TAPER_CONV_CHAIN=0 sets the initial value).
void set_conv_chain(bool on);
bool conv_chain_enabled();
// ... and the classifier's rows inside that launch where compiled (conv_chain_head_cross_entropy; TAPER_CHAIN_HEAD=0 sets the initial value)
void set_conv_chain_head(bool on);
bool conv_chain_head_enabled();

// ---- loss (src/loss.rs) ------------------------------------------------------
Tensor log_softmax(const Tensor &x, int dim = -1);                         // loss.rs:101-126
Tensor softmax(const Tensor &x, int dim = -1);                             // exp(log_softmax), Q12
// Device-side step log folded into the loss kernel (th_softmax_xent_fwd): see Trainer.
struct StepLogSink {
    float *d_metrics = nullptr;
    int64_t capacity = 0;
    int64_t *d_state = nullptr;
    int64_t advance = 0;
    int32_t *d_adam_tick = nullptr;  // Adam's t += 1 folded into the loss kernel (fused-update steps)
};
// Classifier head = last Linear + cross-entropy as ONE launch (th_linear_xent_head), with the
// backward products for loss.backward() from the root computed in the same launch.  Trainer-internal:
// the recorded node only supports that backward (unit upstream grad, grad slots None).
bool linear_cross_entropy_supported(const Tensor &h, const Tensor &weight);
Tensor linear_cross_entropy(const Tensor &h, const Tensor &weight, const Tensor &bias, const Tensor &targets,
                            Tensor *n_correct_out, const StepLogSink *log);
// The same for a WIDE input (in_features > 256, classes <= 16, batch <= 4096): th_linear_xent_wide, two launches; W / b are
// never updated in the launch (deferred like the head's).
bool linear_cross_entropy_wide_supported(const Tensor &h, const Tensor &weight, const Tensor &bias);
Tensor linear_cross_entropy_wide(const Tensor &h, const Tensor &weight, const Tensor &bias, const Tensor &targets,
                                 Tensor *n_correct_out, const StepLogSink *log);
// A convolutional front that ends in a pooled map (Tensor::conv_chain), Flatten, Linear, cross-entropy -- the simple CNN of BASELINE
// configs[2] -- as TWO launches: the chain with the classifier's rows in its last epilogue (th_conv_chain_head_fwd), then every sum over
// the batch with the Adam updates in the epilogues (th_wide_head_grads).  Same contract as linear_cross_entropy.
bool conv_chain_head_supported(const Tensor &x, const std::vector<ConvStage> &stages, const Tensor &weight, const Tensor &bias);
// examples/train_mnist_cnn.rs:35-100 whole -- conv rows, global average pool, Flatten, Linear + ReLU, Linear + ReLU, Linear, cross-entropy -- as
// TWO launches: the chain with the classifier's rows in its last epilogue, then every parameter gradient with Adam in the epilogues
// (th_conv_chain_mlp3_xent).  One image per workgroup: worth it while the batch is about one image per CU (TAPER_CHAIN_MLP3_MAX_BATCH, default 384).
bool conv_chain_mlp3_supported(const Tensor &x, const std::vector<ConvStage> &stages, const Tensor (&w)[3], const Tensor (&b)[3]);
Tensor conv_chain_mlp3_cross_entropy(const Tensor &x, const std::vector<ConvStage> &stages, const Tensor (&w)[3], const Tensor (&b)[3],
                                     const Tensor &targets, Tensor *n_correct_out = nullptr, const StepLogSink *log = nullptr);
Tensor conv_chain_head_cross_entropy(const Tensor &x, const std::vector<ConvStage> &stages, const Tensor &weight, const Tensor &bias,
                                     const Tensor &targets, Tensor *n_correct_out, const StepLogSink *log);
// x -> relu(x . W1^T + b1) -> . W2^T + b2 -> cross-entropy as TWO launches: th_linear_fwd_ex (which also carries the
// previous step's deferred Adam updates and opens this step) and th_mlp_tail (head + the hidden layer's whole
// backward + its Adam update; with an input that requires a gradient also dX, whole tiles only).  Same contract as
// linear_cross_entropy.
// three-layer classifier (Linear + ReLU, Linear + ReLU, Linear) + cross-entropy, forward and backward in two launches (th_mlp3_xent)
bool mlp3_supported(const Tensor &x, const Tensor (&w)[3], const Tensor (&b)[3]);
Tensor mlp3_cross_entropy(const Tensor &x, const Tensor (&w)[3], const Tensor (&b)[3], const Tensor &targets, Tensor *n_correct_out,
                          const StepLogSink *log);
bool mlp_tail_supported(const Tensor &x, const Tensor &w1, const Tensor &b1, const Tensor &w2, const Tensor &b2);
Tensor mlp_tail_cross_entropy(const Tensor &x, const Tensor &w1, const Tensor &b1, const Tensor &w2, const Tensor &b2,
                              const Tensor &targets, Tensor *n_correct_out, const StepLogSink *log);
// Linear + ReLU, Linear, cross-entropy at LARGE batch: three launches (th_mlp2_xent), the batch's rows read where they lie -- a dense block or
// rows of the resident dataset through the loader's index vector (no gathered copy).  Trainer-internal, like the forms above.
bool mlp2_supported(const th_row_source &src, size_t batch, const Tensor &w1, const Tensor &b1, const Tensor &w2, const Tensor &b2);
// ... and for 2 or 3 Linear layers (one or two hidden layers: th_mlp2_xent / th_mlp2_xent_deep), first to last
bool mlp2_params_ok(const std::vector<Tensor> &w, const std::vector<Tensor> &b);
bool mlp2_supported(const th_row_source &src, size_t batch, const std::vector<Tensor> &w, const std::vector<Tensor> &b);
Tensor mlp2_cross_entropy(const th_row_source &src, size_t batch, const std::vector<Tensor> &w, const std::vector<Tensor> &b,
                          Tensor *n_correct_out = nullptr, const StepLogSink *log = nullptr);
Tensor mlp2_cross_entropy(const th_row_source &src, size_t batch, const Tensor &w1, const Tensor &b1, const Tensor &w2, const Tensor &b2,
                          Tensor *n_correct_out, const StepLogSink *log);
// n_correct_out (optional): device scalar receiving accuracy()*B from the fused kernel
Tensor cross_entropy_loss(const Tensor &logits, const Tensor &targets, Tensor *n_correct_out = nullptr,
                          const StepLogSink *log = nullptr);  // loss.rs:136-195
float accuracy(const Tensor &predictions, const Tensor &targets);          // loss.rs:271-290 (synchronises)
Tensor one_hot(const Tensor &indices, size_t num_classes);                 // loss.rs:248-268
Tensor mse_loss(const Tensor &pred, const Tensor &targets);                // loss.rs:76-80
Tensor bce_loss(const Tensor &pred, const Tensor &targets);                // loss.rs:6-73
Tensor cross_entropy_loss_onehot(const Tensor &logits, const Tensor &targets);  // loss.rs:201-245

// ---- nn (src/nn.rs, src/activation.rs) --------------------------------------
class Module {
   public:
    virtual ~Module() = default;
    virtual Tensor forward(const Tensor &input) const = 0;  // nn.rs:11
    virtual std::vector<Tensor> parameters() const = 0;     // nn.rs:12
    virtual const char *name() const = 0;
};
using Layer = Module;  // north_star calls the trait "nn::Layer"

class Linear : public Module {  // nn.rs:28-78
   public:
    Tensor weight, bias;  // [out,in], [out] (bias undefined when with_bias=false)
    Linear(size_t in_features, size_t out_features, bool with_bias, uint64_t seed);
    Tensor forward(const Tensor &input) const override;
    Tensor forward_fused_relu(const Tensor &input) const;
    std::vector<Tensor> parameters() const override;
    const char *name() const override { return "Linear"; }
};

class ReLU : public Module {  // activation.rs:7-21
   public:
    Tensor forward(const Tensor &x) const override { return x.relu(); }
    std::vector<Tensor> parameters() const override { return {}; }
    const char *name() const override { return "ReLU"; }
};

class Sigmoid : public Module {  // activation.rs:37-51
   public:
    Tensor forward(const Tensor &x) const override { return x.sigmoid(); }
    std::vector<Tensor> parameters() const override { return {}; }
    const char *name() const override { return "Sigmoid"; }
};

class Conv2d : public Module {  // nn.rs:180-354
   public:
    Tensor weight, bias;   // [out, in / groups, k_h, k_w], [out]
    std::pair<int, int> stride{1, 1}, padding{0, 0}, dilation{1, 1};
    size_t groups = 1;     // > 1: slice, convolve per group, cat (nn.rs:289-332) -- forward only, like the reference
    bool fuse_relu = false;
    Conv2d(size_t in_ch, size_t out_ch, std::pair<int, int> kernel, std::pair<int, int> stride, std::pair<int, int> padding,
           bool with_bias, uint64_t seed, size_t groups = 1);
    Tensor forward(const Tensor &x) const override;
    std::vector<Tensor> parameters() const override;
    const char *name() const override { return fuse_relu ? "Conv2dReLU" : "Conv2d"; }
};

class Conv2dReLU : public Conv2d {  // nn.rs:433-490
   public:
    Conv2dReLU(size_t in_ch, size_t out_ch, std::pair<int, int> kernel, std::pair<int, int> stride,
               std::pair<int, int> padding, bool with_bias, uint64_t seed)
        : Conv2d(in_ch, out_ch, kernel, stride, padding, with_bias, seed) {
        fuse_relu = true;
    }
};

class MaxPool2d : public Module {  // nn.rs:508-549
   public:
    std::pair<int, int> kernel, stride, padding;
    MaxPool2d(std::pair<int, int> k, std::pair<int, int> s, std::pair<int, int> p) : kernel(k), stride(s), padding(p) {}
    Tensor forward(const Tensor &x) const override { return x.max_pool2d(kernel, stride, padding); }
    std::vector<Tensor> parameters() const override { return {}; }
    const char *name() const override { return "MaxPool2d"; }
};

class AvgPool2d : public Module {  // nn.rs:570-623; kernel {0,0} = global
   public:
    std::pair<int, int> kernel, stride, padding;
    AvgPool2d(std::pair<int, int> k, std::pair<int, int> s, std::pair<int, int> p) : kernel(k), stride(s), padding(p) {}
    Tensor forward(const Tensor &x) const override;
    std::vector<Tensor> parameters() const override { return {}; }
    const char *name() const override { return "AvgPool2d"; }
};

class AdaptiveAvgPool2d : public Module {  // nn.rs:655-697
   public:
    std::pair<int, int> output_size;
    explicit AdaptiveAvgPool2d(std::pair<int, int> o) : output_size(o) {}
    Tensor forward(const Tensor &x) const override;
    std::vector<Tensor> parameters() const override { return {}; }
    const char *name() const override { return "AdaptiveAvgPool2d"; }
};

class Flatten : public Module {  // nn.rs:730-756
   public:
    size_t start_dim;
    explicit Flatten(size_t s = 1) : start_dim(s) {}
    Tensor forward(const Tensor &x) const override { return x.flatten(start_dim); }
    std::vector<Tensor> parameters() const override { return {}; }
    const char *name() const override { return "Flatten"; }
};

class Dropout : public Module {  // nn.rs:773-827
   public:
    explicit Dropout(float p, uint64_t seed = 0x64726f70ull);
    void eval() { training_ = false; }    // nn.rs:789-791
    void train() { training_ = true; }    // nn.rs:793-795
    Tensor forward(const Tensor &x) const override;
    std::vector<Tensor> parameters() const override { return {}; }
    const char *name() const override { return "Dropout"; }
    Tensor last_mask() const { return last_mask_; }   // the mask of the latest training-mode forward (tests)

   private:
    float p_;
    bool training_ = true;
    uint64_t seed_;
    mutable uint64_t calls_ = 0;
    mutable Tensor last_mask_;
};

class Sequential : public Module {  // nn.rs:130-162
   public:
    std::vector<std::shared_ptr<Module>> layers;
    bool fuse = true;  // Linear followed by ReLU runs as one fused kernel + one tape node
    explicit Sequential(std::vector<std::shared_ptr<Module>> l) : layers(std::move(l)) {}
    Tensor forward(const Tensor &input) const override;
    Tensor forward_prefix(const Tensor &input, size_t n_layers) const;  // layers [0, n_layers)
    // the run of Conv2dReLU(3x3, stride 1, pad 1) [+ MaxPool2d(2) | + global average pool] rows that starts at layer i (Tensor::conv_chain's
    // stage list); returns the index of the first layer behind the run
    size_t conv_stages_at(size_t i, size_t n_layers, std::vector<ConvStage> *stages) const;
    std::vector<Tensor> parameters() const override;
    const char *name() const override { return "Sequential"; }
};

// ---- optim (src/optim.rs) ------------------------------------------------------
// Parameters, their grads and the moments live in flat device arenas so that
// (a) Adam is ONE kernel launch and (b) data-parallel training all-reduces ONE
// buffer.  Constructing an optimizer re-homes each parameter's storage and
// grad slot into the arenas (values preserved).
class FlatParams {
   public:
    std::vector<Tensor> params;
    std::shared_ptr<Buffer> p_arena, g_arena;
    std::shared_ptr<Buffer> d_offsets_buf, d_has_grad_buf;  // int64[n+1], int32[n] (stored in float buffers)
    std::vector<int64_t> offsets;
    std::vector<int32_t> uploaded_mask;
    int64_t total = 0;
    explicit FlatParams(const std::vector<Tensor> &ps);
    // move the gradient arena into `arena` (>= total floats; contents and every slot's state preserved): the slots' views are re-pointed in
    // place, so every handle follows.  For the peer-to-peer communicator's fine-grained arena; captured graphs that baked the old address
    // must be dropped by their owner (Trainer keys its graphs on the arena pointer).
    void rehome_grads(const std::shared_ptr<Buffer> &arena);
    const int64_t *d_offsets() const { return reinterpret_cast<const int64_t *>(d_offsets_buf->d); }
    const int32_t *d_has_grad() const { return reinterpret_cast<const int32_t *>(d_has_grad_buf->d); }
    // upload the (has_grad && !excluded) mask iff it changed (not allowed while capturing);
    // returns how many tensors the mask selects
    size_t sync_mask(const std::vector<char> *excluded = nullptr);
    void zero_missing();     // zero-fill arena slices of grad-less params (before an all-reduce)
    void zero_grad();        // optim.rs:115-119
};

class Optimizer {
   public:
    virtual ~Optimizer() = default;
    virtual void step() = 0;
    virtual void zero_grad() = 0;
    virtual FlatParams &flat() = 0;
};

class SGD : public Optimizer {  // optim.rs:8-40 (momentum ignored, 14-17)
   public:
    SGD(const std::vector<Tensor> &params, float lr);
    void step() override;
    void zero_grad() override { fp_.zero_grad(); }
    FlatParams &flat() override { return fp_; }

   private:
    FlatParams fp_;
    std::shared_ptr<Buffer> lr_buf_;
};

class Adam : public Optimizer {  // optim.rs:43-128
   public:
    Adam(const std::vector<Tensor> &params, float lr, float beta1 = 0.9f, float beta2 = 0.999f, float eps = 1e-8f,
         float weight_decay = 0.0f);
    void step() override;                          // optim.rs:83-113
    // data parallel over a peer-to-peer communicator: all-reduce(mean) of the gradient arena and this step's update in ONE
    // launch (th_allreduce_adam); false = not applicable here (the caller all-reduces in place, then step())
    bool step_reduced(const class Communicator &comm);
    // step() behind an all-reduce of `comm` (peer-to-peer form): the update is skipped on the device when that all-reduce timed out
    // (th_adam_step_guarded) -- the arena then still holds this rank's own gradients
    void set_step_guard(const uint32_t *d_error_word) { step_guard_ = d_error_word; }
    void zero_grad() override { fp_.zero_grad(); } // optim.rs:115-119
    float get_lr() const { return lr_; }
    void set_lr(float lr);                         // optim.rs:125-127
    float weight_decay() const { return wd_; }
    float beta1() const { return beta1_; }
    float beta2() const { return beta2_; }
    float eps() const { return eps_; }
    void set_weight_decay(float wd) { wd_ = wd; }
    // optimizer state as host vectors (checkpointing): t, and m / v in parameter order without padding
    void load_state(int t, const std::vector<float> &m, const std::vector<float> &v);
    int t() const;                                 // reads the device counter (synchronises)
    std::vector<float> m() const;
    std::vector<float> v() const;
    FlatParams &flat() override { return fp_; }

    // ---- fused-update steps (Trainer-internal) ----
    // While a FusedAdamScope is active on this thread, the kernels that produce a parameter's
    // gradient apply that parameter's Adam update in their epilogue (same arithmetic, same t);
    // step() then only covers the parameters nobody fused, with the counter already ticked.
    int32_t *d_tick() const { return reinterpret_cast<int32_t *>(state_->d); }
    bool fuse_for(const Tensor &param, th_adam_fuse *out);  // false: not ours / has a grad already
    bool fuse_for_slot(const std::shared_ptr<GradSlot> &slot, th_adam_fuse *out);
    // The gradient of `param` is (or is being) completed by the launch just enqueued, which could not
    // update it in place (the head kernel; a backward whose dX workgroups still read W): a LATER
    // launch of this step carries the update in spare workgroups (th_adam_slice), step() the leftovers.
    bool defer_for(const Tensor &param);
    int take_deferred(const float *launch_reads, th_adam_slice *out);  // <= TH_MAX_ADAM_SLICES, skipping aliases
    void flush_deferred();
    // keep the deferred updates across step(): the NEXT step's first launch carries them (th_linear_fwd_ex);
    // whoever sets this flushes at the end of its run of steps
    void set_carry_deferred(bool on) { carry_deferred_ = on; }
    bool has_deferred() const { return !deferred_.empty(); }
    // a fused-update step that step() has not closed yet: updates already applied in epilogues and / or waiting for a carrier
    bool step_open() const { for (char f : fused_) if (f) return true; return !deferred_.empty(); }
    void set_external_tick(bool on) { external_tick_ = on; }
    // a capture that failed midway: forget the updates queued / marked for the step that never ran
    void drop_step_bookkeeping() { deferred_.clear(); std::fill(fused_.begin(), fused_.end(), 0); }

   private:
    FlatParams fp_;
    std::shared_ptr<Buffer> m_, v_, state_;  // state_: int32 t, int32 arrival counter, float lr
    float lr_, beta1_, beta2_, eps_, wd_;
    bool external_tick_ = false;
    bool carry_deferred_ = false;
    const uint32_t *step_guard_ = nullptr;
    std::vector<char> fused_;                // per parameter: updated by a fused epilogue this step
    std::vector<th_adam_slice> deferred_;    // updates waiting for a carrier launch
};

class AdamW : public Optimizer {  // optim.rs:130-180: decay applied to the weights, then Adam with wd = 0
   public:
    AdamW(const std::vector<Tensor> &params, float lr, float beta1 = 0.9f, float beta2 = 0.999f, float eps = 1e-8f,
          float weight_decay = 0.0f)
        : adam(params, lr, beta1, beta2, eps, weight_decay) {}
    void step() override;
    void zero_grad() override { adam.zero_grad(); }
    FlatParams &flat() override { return adam.flat(); }
    float get_lr() const { return adam.get_lr(); }
    void set_lr(float lr) { adam.set_lr(lr); }
    Adam adam;
};

// ---- learning-rate schedulers (optim.rs:183-352); all arithmetic in f32 like the reference ----
class LRScheduler {
   public:
    virtual ~LRScheduler() = default;
    virtual void step(const float *metric /* nullptr = None */) = 0;
    virtual float get_lr() const = 0;
};
class StepLR : public LRScheduler {  // optim.rs:190-221
   public:
    StepLR(float base_lr, size_t step_size, float gamma) : lr_(base_lr), step_size_(step_size), gamma_(gamma) {}
    void step(const float *) override;
    float get_lr() const override { return lr_; }

   private:
    float lr_;
    size_t step_size_;
    float gamma_;
    size_t epoch_ = 0;
};
class ExponentialLR : public LRScheduler {  // optim.rs:223-249
   public:
    ExponentialLR(float base_lr, float gamma) : lr_(base_lr), gamma_(gamma) {}
    void step(const float *) override { lr_ *= gamma_; }
    float get_lr() const override { return lr_; }

   private:
    float lr_, gamma_;
};
class CosineAnnealingLR : public LRScheduler {  // optim.rs:251-288
   public:
    CosineAnnealingLR(float base_lr, size_t t_max, float min_lr = 0.0f) : base_(base_lr), min_(min_lr), lr_(base_lr), t_max_(t_max) {}
    void step(const float *) override;
    float get_lr() const override { return lr_; }

   private:
    float base_, min_, lr_;
    size_t t_max_, epoch_ = 0;
};
class ReduceLROnPlateau : public LRScheduler {  // optim.rs:290-352
   public:
    ReduceLROnPlateau(float initial_lr, float factor, size_t patience, float min_lr = 1e-6f, const std::string &mode = "min");
    void step(const float *metric) override;
    float get_lr() const override { return lr_; }
    bool verbose = true;   // the reference prints "Reducing learning rate to ..." (optim.rs:342)

   private:
    float lr_, factor_;
    size_t patience_;
    float min_lr_;
    bool mode_min_;
    float best_;
    size_t counter_ = 0;
};

// RAII: marks `adam` as the optimizer whose updates may be fused on this thread.
// While active on this thread (Trainer graph steps over a Sequential model: every tensor has one consumer),
// Conv2dReLU -> MaxPool2d pairs in bias-only mode skip the max-pool scatter and the ReLU-backward pass.
class PoolBiasScope {
   public:
    explicit PoolBiasScope(bool on);
    ~PoolBiasScope();
    static bool active();

   private:
    bool prev_;
};
// Forward passes nobody differentiates (Trainer::evaluate): the Trainer-internal fused conv fronts skip their count buffers and bias tape nodes
class NoGradScope {
   public:
    NoGradScope();
    ~NoGradScope();
    static bool active();

   private:
    bool prev_;
};

class FusedAdamScope {
   public:
    explicit FusedAdamScope(Adam *adam);
    ~FusedAdamScope();
    static Adam *active();

   private:
    Adam *prev_;
};

// A data-parallel step whose gradient launch exchanges its slices itself (Trainer-internal): while active, mlp_tail_cross_entropy
// launches th_mlp_tail_dp on this communicator -- what its fused epilogues apply is then the mean over the ranks.
class TailExchangeScope {
   public:
    explicit TailExchangeScope(const class Communicator *comm);
    ~TailExchangeScope();
    static const class Communicator *active();

   private:
    const class Communicator *prev_;
};

// ---- data (src/data/mnist.rs) -----------------------------------------------------
class MNISTDataset {
   public:
    Tensor images, labels;  // [N,784] in [0,1], [N] class ids as f32; device resident
    bool train = true;
    size_t len() const { return labels.len(); }
    static MNISTDataset from_host(const std::vector<float> &images, const std::vector<float> &labels, bool train);
    static MNISTDataset from_u8(const std::vector<uint8_t> &pixels, const std::vector<uint8_t> &labels, bool train);
    static MNISTDataset from_idx_files(const std::string &images_path, const std::string &labels_path, bool train);  // mnist.rs:185-274
    static MNISTDataset synthetic(size_t n, uint64_t seed, bool train);  // u8 ~ U{0..255}/255, labels U{0..9}
    std::pair<Tensor, Tensor> get_batch(const std::vector<size_t> &indices) const;  // mnist.rs:277-310
};

class DataLoader {  // mnist.rs:327-386
   public:
    DataLoader(MNISTDataset dataset, size_t batch_size, bool shuffle, uint64_t seed = 0x7461706572ull);
    void reset();                 // mnist.rs:355-363 (reshuffles)
    size_t num_batches() const;   // mnist.rs:365-367
    bool next(Tensor *images, Tensor *labels);  // mnist.rs:373-385; false at end
    const MNISTDataset &dataset() const { return ds_; }
    size_t batch_size() const { return bs_; }
    bool shuffled() const { return shuffle_; }
    const int32_t *d_indices() const { return reinterpret_cast<const int32_t *>(d_idx_->d); }
    size_t current() const { return cur_; }
    void advance(size_t n) { cur_ += n; }

   private:
    void upload_indices();
    MNISTDataset ds_;
    size_t bs_;
    bool shuffle_;
    std::vector<int32_t> indices_;
    std::shared_ptr<Buffer> d_idx_;
    size_t cur_ = 0;
    std::mt19937_64 rng_;
};

// ---- data parallel (new) ---------------------------------------------------------------
class Communicator {
   public:
    static std::vector<uint8_t> unique_id();
    Communicator(int n_ranks, int rank, const std::vector<uint8_t> &id);   // RCCL
    // Peer-to-peer form (one node, <= 8 ranks; th_comm_init_p2p): a one-shot all-reduce for latency-bound gradient arenas.
    // p2p() -> export_arena(optimizer) -> [ship every rank's blob to every rank] -> connect(blobs in rank order).
    static std::shared_ptr<Communicator> p2p(int n_ranks, int rank);
    // W = 2 with this process as its own peer (th_comm_init_loopback): every push, flag, poll and load of the in-launch exchange runs, through
    // local memory; results are the single-GPU step's bit for bit.  What a one-GPU box can measure of the protocol's cost.
    static std::shared_ptr<Communicator> loopback();
    bool is_loopback() const { return loopback_; }
    // The exchange inside the gradient launch (th_mlp_tail_dp): may this communicator take it for a Linear + ReLU + Linear classifier of these
    // shapes?  (peer-to-peer, connected, >= 2 ranks, whole tiles, and -- ranks sharing a device -- room for the waiting workgroups)
    bool tail_exchange_ok(int batch, int in_features, int hidden, int classes) const;
    // ... and for the simple CNN's batch sums (th_wide_head_grads_dp): `in_features` = the flattened pooled map, conv_c = the last conv's channels
    bool wide_exchange_ok(int batch, int in_features, int classes, int conv_c) const;
    bool inkernel = true;                                 // TAPER_DP_INKERNEL=0: always the three-launch form (A/B probe)
    int64_t inkernel_launches() const;                    // th_mlp_tail_dp launches enqueued or captured
    int exchange_selftest(int slots, int rounds) const;   // collective; mismatches + time-outs seen by this rank (th_comm_exchange_selftest)
    int ranks_on_this_device() const;
    int exchange_form() const;                            // th_comm_exchange_form: 0 none, 1 one-shot, 2 two-shot (4 ranks and more)
    // fine_grained: first move the optimizer's gradient arena into fine-grained device memory (coherent across agents, never cached in a
    // peer's L2) -- the fallback when the multi-round self-check fails on the pooled, coarse-grained arena.  The communicator keeps the
    // arenas it registered alive; destroy it on every rank before the optimizer goes away (peers hold IPC mappings of the arena).
    std::vector<uint8_t> export_arena(Optimizer &opt, bool fine_grained = false);
    void connect(const std::vector<uint8_t> &blobs);
    bool is_p2p() const { return p2p_; }
    bool timed_out() const;                               // a peer never arrived (synchronises the stream)
    bool failed() const;                                  // the same verdict from the host-visible error word, without synchronising
    void set_timeout_ms(int64_t ms);                      // bound of the in-kernel waits (default 120 s, TAPER_P2P_TIMEOUT_MS)
    // Collective, before training, on every rank: `rounds` all-reduces of patterns that differ per rank, per round AND per element through
    // the SAME addresses of the optimizer's real gradient arena -- a reader that keeps a stale line of a peer's arena from round k fails
    // round k + 1 -- first through the in-place kernel (every element compared), then through the fused all-reduce + Adam kernel
    // (p / m / v compared with the closed form of optim.rs:83-113 on the host; parameters, moments and the counter are restored).
    bool self_check(Optimizer &opt, int rounds = 3);
    ~Communicator();
    void allreduce_mean(float *d_buf, size_t n) const;  // sum over ranks * 1/W on the ctx stream
    th_comm *handle() const { return comm_; }
    const uint32_t *error_word() const;                   // device word, non-zero once an all-reduce timed out (nullptr: RCCL)
    uint32_t *step_word() const;                          // device word: the in-launch exchange's step number (nullptr: RCCL)
    // `reps` exchanges as a Trainer step issues them (all-reduce + Adam: one fused launch, or all-reduce then Adam::step), back to back
    // between two events on the stream, on every rank at once; us per exchange.  The optimizer state moves (call it after the timed run).
    float time_exchange(Adam &opt, int reps);
    int n_ranks, rank;
    bool fuse_adam = true;                                // p2p: feed the mean gradient straight into Adam::step (th_allreduce_adam)

   private:
    Communicator() : n_ranks(1), rank(0) {}
    th_comm *comm_ = nullptr;
    bool p2p_ = false, loopback_ = false;
    std::shared_ptr<Buffer> g_hold_, p_hold_;             // the arenas registered with (and mapped by) the peers
};

// ---- train (src/train.rs, examples/train_mnist*.rs) ------------------------------------
struct EpochResult {
    float avg_loss = 0.f;     // sum(loss) / num_batches   (train.rs:140)
    float accuracy = 0.f;     // total_correct / total_samples with the per-batch truncation of train.rs:117 (Q13)
    size_t total_correct = 0, total_samples = 0, num_batches = 0;
    std::vector<float> losses, ncorrect;  // per step
};

struct Metrics {  // train.rs:9-71
    std::vector<float> train_loss, train_acc, val_loss, val_acc, epoch_times;
    std::string last_line() const;     // text of print_last (train.rs:29-45); empty before the first epoch
    std::string summary() const;       // text of plot_summary (train.rs:47-70)
    void print_last() const;
    void plot_summary() const;
};

// f32 as Rust's `{}` prints it (shortest digits that round-trip, never an exponent): the number
// format of the checkpoint file (train.rs:283-285)
std::string format_f32_display(float v);

class Trainer {  // train.rs:74-172
   public:
    std::shared_ptr<Module> model;
    std::shared_ptr<Adam> optimizer;
    std::shared_ptr<Communicator> comm;   // optional: data-parallel grad all-reduce before step()
    Shape sample_shape;                   // {} -> feed [B,784]; {1,28,28} -> reshape like train_mnist_cnn.rs:161-162
    std::string device = "hip:gfx950";    // train.rs:79 "For future GPU support"
    size_t graph_chunk = 128;             // steps captured per hipGraph replay (plus chunk/2, chunk/4, ..., 1-step graphs for the tail)
    int fuse_head = 2;                    // graph path: 1 = last Linear + cross-entropy as one launch; 2 = additionally the
                                          // backward (+ Adam) of a Linear+ReLU layer in front of it, same launch (th_mlp_tail)
    bool fuse_adam = true;                // Adam updates in the epilogue of the grad-producing kernels (graph path, no DP)
    Trainer(std::shared_ptr<Module> m, std::shared_ptr<Adam> o) : model(std::move(m)), optimizer(std::move(o)) {}

    // one step exactly as examples/train_mnist.rs:89-121 (reads loss + accuracy back every step)
    void train_step(const Tensor &images, const Tensor &labels, float *loss_out, float *acc_out);
    void check_comm() const;   // throws when the peer-to-peer communicator has timed out (after any stream synchronisation)
    EpochResult train_epoch(DataLoader &loader);              // train.rs:98-144 (eager, synchronising)
    EpochResult evaluate(DataLoader &loader);                 // train.rs:147-172
    // same arithmetic, but the step's op list is captured once into a hipGraph
    // and replayed; loss / n_correct stay on device until the epoch ends.
    EpochResult train_epoch_graph(DataLoader &loader, size_t max_steps = 0);
    // train.rs:175-261: epochs of train + evaluate, scheduler.step(Some(val_loss)) -> optimizer.set_lr,
    // metrics, early stop at val_acc > 0.99.  graph = true trains through the captured step.
    std::shared_ptr<LRScheduler> scheduler;
    Metrics metrics;
    void fit(DataLoader &train_loader, DataLoader &val_loader, size_t epochs, bool verbose, bool graph = true);
    // train.rs:264-292 text checkpoint (parameter count; per parameter "ndim d0 d1 ..." then one value per
    // line); load_checkpoint is its inverse (shapes must match the model).  The optimizer-state pair
    // is an extension in the same number format: "adam t lr beta1 beta2 eps wd", then m and v per parameter.
    void save_checkpoint(const std::string &path) const;
    void load_checkpoint(const std::string &path);
    void save_optimizer_state(const std::string &path) const;
    void load_optimizer_state(const std::string &path);
    ~Trainer();

   private:
    // gathers `steps` batches with one launch, then enqueues the compute of each step
    void enqueue_steps(const float *d_images, const float *d_labels, const int32_t *d_indices, int64_t n_indices,
                       size_t batch, size_t steps);
    void enqueue_compute(float *d_xb, float *d_yb, size_t batch, const th_row_source *rows = nullptr);
    bool mlp2_step(size_t batch, int64_t n_rows) const;   // this model at this batch takes th_mlp2_xent (rows read in place)
    bool tail_exchange_step(size_t batch) const;          // data parallel: this step reduces its gradients inside its own launch (th_mlp_tail_dp)
    bool chain_exchange_step(const Tensor &xin) const;    // ... the simple CNN's step does (th_wide_head_grads_dp)
    void drop_graphs();
    std::vector<std::pair<size_t, th_graph *>> graphs_;  // (steps per replay, graph), largest first
    std::vector<std::pair<size_t, th_graph *>> whole_graphs_;  // (steps, graph): state reset + that many full steps -- a whole call in one replay
    bool graph_capture_failed_ = false;
    std::vector<size_t> whole_capture_failed_;                 // call lengths whose whole-call capture failed once: not tried again
    std::vector<uintptr_t> graph_key_;   // what the captured steps bake in (train_epoch_graph); a mismatch drops the graphs
    std::shared_ptr<Buffer> xb_, yb_, state_, metrics_, step_loss_, step_ncorrect_;
    size_t metrics_cap_ = 0;
};

}  // namespace taper

"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy float32 / plain Python) of the reference code
around the training step that SURVEY.md 8(f) ranks "next": LR schedulers and AdamW (src/optim.rs),
cross_entropy_loss_onehot and Dropout (src/loss.rs, src/nn.rs), Metrics text and the text checkpoint
(src/train.rs).  Each function cites the reference lines it follows.  Only tests/ may import this.

Pinning: the reference has no tests for the schedulers, bce_loss or the one-hot cross-entropy -> "parity unpinned by
reference tests"; they are cross-checked against torch-CPU vectors (tests/golden/losses_extra.npz, generator committed:
StepLR / ExponentialLR / CosineAnnealingLR sequences, F.binary_cross_entropy, the one-hot form of F.cross_entropy) and closed
forms in tests/test_train_extra.py;
the number format against Rust's documented `Display for f32` outputs (shortest round-trip digits,
never an exponent: 1 -> "1", f32::MAX -> "340282350000000000000000000000000000000",
f32::MIN_POSITIVE -> "0.000000000000000000000000000000000000011754944")."""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32
PI_F32 = f32(3.14159274101257324)   # std::f32::consts::PI


# ---- src/optim.rs:183-352 ---------------------------------------------------------------------
class StepLR:  # optim.rs:190-221
    def __init__(self, base_lr, step_size, gamma):
        self.current_lr, self.step_size, self.gamma, self.current_epoch = f32(base_lr), int(step_size), f32(gamma), 0

    def step(self, metrics=None):
        self.current_epoch += 1
        if self.current_epoch % self.step_size == 0:
            self.current_lr = f32(self.current_lr * self.gamma)

    def get_lr(self):
        return float(self.current_lr)


class ExponentialLR:  # optim.rs:223-249
    def __init__(self, base_lr, gamma):
        self.current_lr, self.gamma = f32(base_lr), f32(gamma)

    def step(self, metrics=None):
        self.current_lr = f32(self.current_lr * self.gamma)

    def get_lr(self):
        return float(self.current_lr)


class CosineAnnealingLR:  # optim.rs:251-288
    def __init__(self, base_lr, t_max, min_lr=None):
        self.base_lr, self.min_lr = f32(base_lr), f32(0.0 if min_lr is None else min_lr)
        self.current_lr, self.t_max, self.current_epoch = f32(base_lr), int(t_max), 0

    def step(self, metrics=None):
        self.current_epoch += 1
        progress = f32(f32(self.current_epoch) / f32(self.t_max))
        cos_val = f32(f32(f32(1.0) + np.cos(f32(progress * PI_F32), dtype=f32)) / f32(2.0))
        self.current_lr = f32(self.min_lr + f32(f32(self.base_lr - self.min_lr) * cos_val))

    def get_lr(self):
        return float(self.current_lr)


class ReduceLROnPlateau:  # optim.rs:290-352
    def __init__(self, initial_lr, factor, patience, min_lr=None, mode=None):
        self.mode = mode or "min"
        self.current_lr, self.factor, self.patience = f32(initial_lr), f32(factor), int(patience)
        self.min_lr = f32(1e-6 if min_lr is None else min_lr)
        self.best_metric = f32(np.inf if self.mode == "min" else -np.inf)
        self.patience_counter = 0

    def step(self, metrics=None):
        if metrics is None:
            return
        metric = f32(metrics)
        improved = metric < self.best_metric if self.mode == "min" else metric > self.best_metric
        if improved:
            self.best_metric, self.patience_counter = metric, 0
        else:
            self.patience_counter += 1
            if self.patience_counter >= self.patience:
                self.current_lr = max(f32(self.current_lr * self.factor), self.min_lr)
                self.patience_counter = 0

    def get_lr(self):
        return float(self.current_lr)


def adamw_step(oracle_adam, params):
    """optim.rs:147-168 on top of the C oracle's Adam: every weight decays by (1 - lr*wd) in f32 (grad or
    not), then a standard Adam step with weight_decay = 0.  `oracle_adam` must have been built with
    weight_decay=None; returns nothing (params are updated in place)."""
    lr, wd = f32(oracle_adam.get_lr()), f32(oracle_adam.decoupled_wd)
    if wd > 0:
        factor = f32(f32(1.0) - f32(lr * wd))
        for p in params:
            p.set_data((p.data() * factor).astype(f32))
    oracle_adam.step()


# ---- src/loss.rs:201-245 ------------------------------------------------------------------------
def log_softmax_rows(x):
    """loss.rs:101-126 on a [B,C] float32 array"""
    x = np.asarray(x, f32)
    shifted = (x - x.max(axis=1, keepdims=True)).astype(f32)
    lse = np.log(np.exp(shifted, dtype=f32).sum(axis=1, keepdims=True, dtype=f32), dtype=f32)
    return (shifted - lse).astype(f32)


def cross_entropy_loss_onehot(logits, targets, gloss=1.0):
    """-> (loss, dlogits): loss = -sum(t * log_softmax(x)) / B (loss.rs:214-222); the recorded node's
    gradient is (softmax - t) * gloss / B whatever the rows of t sum to (loss.rs:226-240)"""
    logits, targets = np.asarray(logits, f32), np.asarray(targets, f32)
    b = logits.shape[0]
    logp = log_softmax_rows(logits)
    loss = f32(-f32((targets * logp).astype(f32).sum(dtype=f32)) / f32(b))
    grad = ((np.exp(logp, dtype=f32) - targets) * f32(gloss) / f32(b)).astype(f32)
    return float(loss), grad


# ---- src/nn.rs:798-822 ---------------------------------------------------------------------------
def dropout_forward(x, mask=None, p=0.5, training=True):
    """mask: the {0, 1/(1-p)} tensor the layer drew (the reference's RNG is unseeded, so parity is
    'output == input * mask' plus the statistics of the mask)"""
    x = np.asarray(x, f32)
    if not training or p == 0.0:
        return x
    if p == 1.0:
        return np.zeros_like(x)
    return (x * np.asarray(mask, f32)).astype(f32)


# ---- src/nn.rs:289-332, 859-1014: grouped convolution = slices + per-group conv2d + cat --------------------
def slice_channels(x, start, end):
    """nn.rs:862-886 on a [N,C,H,W] array"""
    return np.ascontiguousarray(np.asarray(x, f32)[:, start:end])


def cat(arrays, dim):
    """nn.rs:928-1014: 2-D along dim 0 / 1, 4-D along dim 1 (anything else is unimplemented!() in the reference)"""
    nd = np.asarray(arrays[0]).ndim
    if nd not in (2, 4) or (nd == 4 and dim != 1):
        raise NotImplementedError("cat: the reference implements 2-D (dim 0/1) and 4-D (dim 1) only")
    return np.concatenate([np.asarray(a, f32) for a in arrays], axis=dim)


def grouped_conv2d(O, x, weight, bias, groups, stride=(1, 1), padding=(0, 0), relu=False):
    """nn.rs:289-332 with the C oracle's conv2d (O = oracle.oracle) per group: weight [C_out, C_in/groups, k, k];
    each group's weight slice is handed to conv2d as its own tensor, i.e. the reinterpretation quirk Q3 applies per group"""
    x, weight = np.asarray(x, f32), np.asarray(weight, f32)
    cin_g, cout_g = x.shape[1] // groups, weight.shape[0] // groups
    outs = []
    for g in range(groups):
        xs = O.Tensor(slice_channels(x, g * cin_g, (g + 1) * cin_g))
        ws = O.Tensor(np.ascontiguousarray(weight[g * cout_g:(g + 1) * cout_g]))          # slice_output_channels, nn.rs:889-914
        bs = None if bias is None else O.Tensor(np.asarray(bias, f32)[g * cout_g:(g + 1) * cout_g].copy())   # slice_1d, nn.rs:917-925
        y = (xs.conv2d_relu if relu else xs.conv2d)(ws, bs, stride, padding)
        outs.append(np.asarray(y.data(), f32).reshape(y.shape()))
    return cat(outs, 1)


# ---- src/tensor.rs:1663-1726, 1805-1970: conv2d through the GENERAL im2col (not 3x3-stride-1, not 1x1) ------------------
def _copy_consecutive_elements(inp, out, batch, ch, h_in, w_in, in_h, in_w_start, out_start, count):
    """tensor.rs:1910-1969.  Both branches (the >= 8 block copy and the scalar loop) index the plane as
    batch*ch*h_in*w_in + ch*h_in*w_in (1931, 1964; quirk Q9: `ch`, not the channel count)."""
    if count >= 8 and in_w_start + count <= w_in:                                       # 1928
        in_base = batch * ch * h_in * w_in + ch * h_in * w_in + in_h * w_in + in_w_start   # 1931
        out[out_start:out_start + count] = inp[in_base:in_base + count]                  # 1933-1957
    else:
        for i in range(count):                                                            # 1960
            in_w = in_w_start + i
            if in_w < w_in:                                                               # 1962
                out[out_start + i] = inp[batch * ch * h_in * w_in + ch * h_in * w_in + in_h * w_in + in_w]   # 1964-1965


def im2col_general(x, k_h, k_w, stride, padding, dilation):
    """tensor.rs:1805-1906 loop for loop (pure Python: small cases only) -> col [n*h_out*w_out, c*k_h*k_w] float32"""
    x = np.asarray(x, f32)
    n, c, h_in, w_in = x.shape
    (stride_h, stride_w), (pad_h, pad_w), (dil_h, dil_w) = stride, padding, dilation
    h_out = (h_in + 2 * pad_h - dil_h * (k_h - 1) - 1) // stride_h + 1                   # 1676
    w_out = (w_in + 2 * pad_w - dil_w * (k_w - 1) - 1) // stride_w + 1                   # 1677
    col_size = c * k_h * k_w
    inp, out = x.reshape(-1), np.zeros(n * h_out * w_out * col_size, f32)                # 1681
    for batch in range(n):
        for out_h in range(h_out):
            for out_w in range(w_out):
                col_base = (batch * h_out * w_out + out_h * w_out + out_w) * col_size    # 1829-1830
                for ch in range(c):
                    for k_row in range(k_h):
                        in_h = out_h * stride_h + k_row * dil_h                          # 1834
                        if not (pad_h <= in_h < h_in + pad_h):                           # 1836
                            continue
                        in_h_idx = in_h - pad_h
                        consecutive_count, start_k_col = 0, 0
                        for k_col in range(k_w):
                            in_w = out_w * stride_w + k_col * dil_w                      # 1844
                            if pad_w <= in_w < w_in + pad_w and consecutive_count == k_col - start_k_col:   # 1846-1849
                                consecutive_count += 1
                            else:
                                if consecutive_count > 0:                                 # 1853
                                    _copy_consecutive_elements(inp, out, batch, ch, h_in, w_in, in_h_idx,
                                                               out_w * stride_w + start_k_col * dil_w - pad_w,
                                                               col_base + ch * k_h * k_w + k_row * k_w + start_k_col, consecutive_count)
                                col_idx = col_base + ch * k_h * k_w + k_row * k_w + k_col   # 1871-1872
                                if pad_w <= in_w < w_in + pad_w:                          # 1874
                                    out[col_idx] = inp[batch * c * h_in * w_in + ch * h_in * w_in + in_h_idx * w_in + (in_w - pad_w)]
                                start_k_col, consecutive_count = k_col + 1, 0             # 1883-1884
                        if consecutive_count > 0:                                         # 1889
                            _copy_consecutive_elements(inp, out, batch, ch, h_in, w_in, in_h_idx,
                                                       out_w * stride_w + start_k_col * dil_w - pad_w,
                                                       col_base + ch * k_h * k_w + k_row * k_w + start_k_col, consecutive_count)
    return out.reshape(n * h_out * w_out, col_size), h_out, w_out


def conv2d_general(x, weight, bias, stride, padding, dilation, relu=False):
    """tensor.rs:1221-1285 on the general im2col: col . weight viewed [K, C_out] (1262, Q3) -> [n, h_out, w_out, c_out]
    -> NCHW (1275-1276) + bias (1279-1282) [+ ReLU, nn.rs:433-490]; fp32 accumulation in k order"""
    x, weight = np.asarray(x, f32), np.asarray(weight, f32)
    n, c_out, k_h, k_w = x.shape[0], weight.shape[0], weight.shape[2], weight.shape[3]
    col, h_out, w_out = im2col_general(x, k_h, k_w, stride, padding, dilation)
    w2 = weight.reshape(col.shape[1], c_out)
    out2 = np.zeros((col.shape[0], c_out), f32)
    for k in range(col.shape[1]):                     # k-ordered fp32 chain, like the reference's sgemm inner product
        out2 += col[:, k:k + 1] * w2[k:k + 1, :]
    out = np.ascontiguousarray(out2.reshape(n, h_out, w_out, c_out).transpose(0, 3, 1, 2))
    if bias is not None:
        out = out + np.asarray(bias, f32).reshape(1, c_out, 1, 1)
    return np.maximum(out, 0).astype(f32) if relu else out.astype(f32)


# ---- src/tensor.rs:2110-2288: the PTQ storage codecs that do real work (int8, f16) ------------------------------------
def f32_to_f16_bits(x):
    """tensor.rs:2191-2238, vectorised over a float32 array -> uint16: round half UP on the dropped 13 bits, the mantissa
    carry OR-ed into the exponent field, truncating denormals, overflow -> inf, |x| < 2^-25 -> 0"""
    bits = np.asarray(x, f32).reshape(-1).view(np.uint32).astype(np.uint64)
    sign, exponent, mantissa = (bits >> 31) & 1, (bits >> 23) & 0xFF, bits & 0x7FFFFF
    e16 = exponent.astype(np.int64) - 127 + 15
    out = np.zeros(bits.shape, np.uint64)
    special = exponent == 0xFF
    zero = (exponent == 0) & (mantissa == 0)
    over = ~special & ~zero & (e16 >= 0x1F)
    under = ~special & ~zero & (e16 <= 0)
    tiny = under & (e16 < -10)
    den = under & ~tiny
    normal = ~special & ~zero & ~over & ~under
    out[special] = (sign[special] << 15) | (0x1F << 10) | np.where(mantissa[special] != 0, 0x200, 0).astype(np.uint64)
    out[zero | tiny] = sign[zero | tiny] << 15
    out[over] = (sign[over] << 15) | (0x1F << 10)
    shift = (1 - e16[den] + 13).astype(np.uint64)
    out[den] = (sign[den] << 15) | ((mantissa[den] | 0x800000) >> shift)
    out[normal] = (sign[normal] << 15) | (e16[normal].astype(np.uint64) << 10) | ((mantissa[normal] + 0x1000) >> 13)
    return (out & 0xFFFF).astype(np.uint16).reshape(np.asarray(x).shape)


def f16_bits_to_f32(h):
    """tensor.rs:2241-2287 on a uint16 array -> float32"""
    bits = np.asarray(h, np.uint16).reshape(-1).astype(np.uint32)
    sign, exponent, mantissa = (bits >> 15) & 1, (bits >> 10) & 0x1F, bits & 0x3FF
    out = np.zeros(bits.shape, np.uint32)
    special = exponent == 0x1F
    out[special] = (sign[special] << 31) | (0xFF << 23) | np.where(mantissa[special] != 0, mantissa[special] << 13, 0)
    zero = (exponent == 0) & (mantissa == 0)
    out[zero] = sign[zero] << 31
    den = (exponent == 0) & (mantissa != 0)
    for i in np.nonzero(den)[0]:
        exp, mant = -14, int(mantissa[i])
        while (mant & 0x400) == 0:
            mant <<= 1
            exp -= 1
        out[i] = (int(sign[i]) << 31) | (((exp + 127) & 0xFF) << 23) | ((mant & 0x3FF) << 13)
    normal = ~special & (exponent != 0)
    out[normal] = (sign[normal] << 31) | (((exponent[normal] + 127 - 15) & 0xFF) << 23) | (mantissa[normal] << 13)
    return out.view(f32).reshape(np.asarray(h).shape)


def quantize_int8(x):
    """tensor.rs:2110-2152 -> (q int8, scale, zero_point, min_val)"""
    x = np.asarray(x, f32).reshape(-1)
    fin = x[np.isfinite(x)]
    mn, mx = (f32(fin.min()), f32(fin.max())) if fin.size else (f32(np.inf), f32(-np.inf))
    if mn == mx:
        mn, mx = f32(mn - f32(0.1)), f32(mx + f32(0.1))
    with np.errstate(all="ignore"):
        scale = f32(f32(mx - mn) / f32(255.0))
        t = ((x - mn).astype(f32) / scale).astype(f32)
        r = np.where(np.isnan(t), f32(0), np.sign(t) * np.floor(np.abs(t) + f32(0.5)))          # f32::round: half away from zero
        r = np.clip(np.nan_to_num(r, nan=0.0, posinf=2147483647.0, neginf=-2147483648.0), -2147483648.0, 2147483647.0).astype(np.int64)   # `as i32`
    q = np.clip(r - 128, -128, 127).astype(np.int8)
    return q, float(scale), -128, float(mn)


def dequantize_int8(q, scale, zero_point, min_val):
    """tensor.rs:353-360"""
    return ((np.asarray(q, np.int8).astype(np.int32) - zero_point).astype(f32) * f32(scale)).astype(f32) + f32(min_val)


# ---- src/train.rs -----------------------------------------------------------------------------------
def format_f32_display(v) -> str:
    """Rust `{}` for f32 (train.rs:283-285 `writeln!(file, "{}", value)`): shortest digits that
    round-trip, positional, no trailing ".0" """
    v = f32(v)
    if np.isnan(v):
        return "NaN"
    if np.isinf(v):
        return "-inf" if v < 0 else "inf"
    s = np.format_float_positional(v, unique=True, trim="-")
    return s


def checkpoint_text(params) -> str:
    """train.rs:264-292: params = [(shape tuple, flat float32 array), ...]"""
    lines = [str(len(params))]
    for shape, data in params:
        lines.append(" ".join([str(len(shape))] + [str(d) for d in shape]))
        lines.extend(format_f32_display(x) for x in np.asarray(data, f32).reshape(-1))
    return "\n".join(lines) + "\n"


def metrics_last_line(m) -> str:
    """train.rs:29-45 (m: dict of lists)"""
    if not (m["train_loss"] and m["train_acc"] and m["val_loss"] and m["val_acc"]):
        return ""
    return "Train Loss: %.4f | Train Acc: %.2f%% | Val Loss: %.4f | Val Acc: %.2f%%" % (
        m["train_loss"][-1], float(f32(m["train_acc"][-1]) * f32(100.0)), m["val_loss"][-1], float(f32(m["val_acc"][-1]) * f32(100.0)))


def metrics_summary(m) -> str:
    """train.rs:47-70"""
    bar = "=" * 50
    s = "\nTraining Summary:\n" + bar + "\n"
    if m["train_acc"]:
        pct = lambda a: float(f32(a) * f32(100.0))
        best_t = max([0.0] + [float(a) for a in m["train_acc"]])
        best_v = max([0.0] + [float(a) for a in m["val_acc"]])
        s += "Best Train Accuracy: %.2f%%\nBest Val Accuracy: %.2f%%\nFinal Train Accuracy: %.2f%%\nFinal Val Accuracy: %.2f%%\n" % (
            pct(best_t), pct(best_v), pct(m["train_acc"][-1]), pct(m["val_acc"][-1]))
        if m["epoch_times"]:
            total = f32(0.0)
            for t in m["epoch_times"]:
                total = f32(total + f32(t))
            s += "Total Training Time: %.2fs\nAverage Epoch Time: %.2fs\n" % (float(total), float(f32(total / f32(len(m["epoch_times"])))))
    return s + bar + "\n"


def fit_schedule(val_losses, scheduler, lr0):
    """the learning rate each epoch of Trainer::fit trains with (train.rs:209-213: after epoch e the
    scheduler steps on Some(val_loss) and its lr is handed to the optimizer)"""
    lrs, lr = [], float(f32(lr0))
    for vl in val_losses:
        lrs.append(lr)
        scheduler.step(vl)
        lr = scheduler.get_lr()
    return lrs


# ---- src/data/mnist.rs:184-274 (IDX ingestion) ----------------------------------------------------
def load_idx_images(buffer: bytes):
    """mnist.rs:184-233 load_images: 16-byte big-endian header {0x00000803, count, 28, 28}, then count*784 bytes; each
    pixel becomes `u8 as f32 / 255.0` (:226).  Errors are the reference's `Err(String)` texts, raised as ValueError."""
    if len(buffer) < 16:
        raise ValueError("is too small")                                              # :190-192
    magic = int.from_bytes(buffer[0:4], "big")
    if magic != 0x00000803:
        raise ValueError(f"Invalid magic number for images: {magic:#x}")              # :195-198
    n, rows, cols = (int.from_bytes(buffer[o:o + 4], "big") for o in (4, 8, 12))
    if rows != 28 or cols != 28:
        raise ValueError(f"Unexpected image size: {rows}x{cols}")                      # :205-207
    expected = 16 + n * 784
    if len(buffer) != expected:
        raise ValueError(f"File size mismatch. Expected {expected}, got {len(buffer)}")  # :209-217
    px = np.frombuffer(buffer, dtype=np.uint8, offset=16).astype(f32) / f32(255.0)     # :223-228
    return px.reshape(n, 784)


def load_idx_labels(buffer: bytes):
    """mnist.rs:236-274 load_labels: 8-byte header {0x00000801, count}, then count bytes, each `u8 as f32` (:268)."""
    if len(buffer) < 8:
        raise ValueError("is too small")                                              # :242-244
    magic = int.from_bytes(buffer[0:4], "big")
    if magic != 0x00000801:
        raise ValueError(f"Invalid magic number for labels: {magic:#x}")              # :247-250
    n = int.from_bytes(buffer[4:8], "big")
    expected = 8 + n
    if len(buffer) != expected:
        raise ValueError(f"File size mismatch. Expected {expected}, got {len(buffer)}")  # :254-262
    return np.frombuffer(buffer, dtype=np.uint8, offset=8).astype(f32)

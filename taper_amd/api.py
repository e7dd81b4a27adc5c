"""Python face of the host mirror (include/taper_host.h): the reference's
public surface -- src/lib.rs:1-17: Tensor, Tape, nn, loss, optim, data, train
-- with the reference's names and argument meaning.  Every call crosses the
C ABI into libtaper_host.so (C++), which reaches the GPU only through
libtaper_hip.so (hand-written HIP for gfx950).  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import TaperError, host, tp_check

_p = C.c_void_p


def _shape_arr(shape):
    return (C.c_size_t * len(shape))(*[int(s) for s in shape])


def _new_handle():
    return _p()


class Device:
    @staticmethod
    def set_device(i: int):
        tp_check(host.tp_device_set(int(i)), "tp_device_set")

    @staticmethod
    def sync():
        tp_check(host.tp_device_sync(), "tp_device_sync")

    @staticmethod
    def ctx_handle() -> int:
        h = host.tp_device_ctx()
        if not h:
            raise TaperError(host.tp_last_error().decode())
        return h


class Tape:
    """src/tape.rs"""

    @staticmethod
    def reset():
        tp_check(host.tp_tape_reset(), "tp_tape_reset")

    @staticmethod
    def len() -> int:
        n = C.c_size_t()
        tp_check(host.tp_tape_len(C.byref(n)), "tp_tape_len")
        return n.value

    @staticmethod
    def set_compat_zero_sentinel(on: bool):
        tp_check(host.tp_tape_set_compat_zero_sentinel(1 if on else 0), "tp_tape_set_compat_zero_sentinel")


def set_full_backward(on: bool):
    """False (default) = faithful to the reference (conv weights never get gradients, quirk Q2)."""
    tp_check(host.tp_set_full_backward(1 if on else 0), "tp_set_full_backward")


def set_conv_chain(on: bool):
    """Trainer steps: the convolutional front of a Sequential as ONE launch where an instance is compiled (default), or layer by layer"""
    tp_check(host.tp_set_conv_chain(1 if on else 0), "tp_set_conv_chain")


def set_conv_chain_head(on: bool):
    """Trainer steps: the classifier behind such a front row by row inside the same launch where compiled (default), or as its own launches"""
    tp_check(host.tp_set_conv_chain_head(1 if on else 0), "tp_set_conv_chain_head")


class Tensor:
    """src/tensor.rs Tensor: a shared handle to device storage + grad slot + tape node."""

    def __init__(self, data=None, shape=None, _h=None):
        if _h is not None:
            self._h = _h
            return
        a = np.ascontiguousarray(np.asarray(data, dtype=np.float32))
        if shape is None:
            shape = a.shape if a.ndim else (1,)
        a = a.reshape(-1)
        if a.size != int(np.prod(shape)):
            raise TaperError("Tensor::new: data length does not match shape")
        h = _p()
        tp_check(host.tp_tensor_new(a.ctypes.data, _shape_arr(shape), len(shape), C.byref(h)), "tp_tensor_new")
        self._h = h.value

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                host.tp_tensor_free(h)
            except Exception:
                pass

    @staticmethod
    def scalar(v):
        return Tensor([float(v)], (1,))

    @staticmethod
    def randn(shape, seed=0):
        h = _p()
        tp_check(host.tp_tensor_randn(_shape_arr(shape), len(shape), int(seed), C.byref(h)), "tp_tensor_randn")
        return Tensor(_h=h.value)

    def requires_grad(self):
        tp_check(host.tp_tensor_set_requires_grad(self._h, 1), "set_requires_grad")
        return self

    def clone(self):
        h = _p()
        tp_check(host.tp_tensor_clone(self._h, C.byref(h)), "tp_tensor_clone")
        return Tensor(_h=h.value)

    def shape(self):
        nd = C.c_int()
        tp_check(host.tp_tensor_ndim(self._h, C.byref(nd)), "tp_tensor_ndim")
        s = (C.c_size_t * 4)()
        tp_check(host.tp_tensor_shape(self._h, s), "tp_tensor_shape")
        return tuple(int(s[i]) for i in range(nd.value))

    def numel(self):
        n = C.c_size_t()
        tp_check(host.tp_tensor_len(self._h, C.byref(n)), "tp_tensor_len")
        return n.value

    def data(self) -> np.ndarray:
        out = np.empty(self.numel(), dtype=np.float32)
        tp_check(host.tp_tensor_data(self._h, out.ctypes.data), "tp_tensor_data")
        return out.reshape(self.shape())

    def set_data(self, a):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float32)).reshape(-1)
        assert a.size == self.numel()
        tp_check(host.tp_tensor_set_data(self._h, a.ctypes.data), "tp_tensor_set_data")

    def grad(self):
        has = C.c_int()
        tp_check(host.tp_tensor_has_grad(self._h, C.byref(has)), "tp_tensor_has_grad")
        if not has.value:
            return None
        out = np.empty(self.numel(), dtype=np.float32)
        tp_check(host.tp_tensor_grad(self._h, out.ctypes.data), "tp_tensor_grad")
        return out.reshape(self.shape())

    grad_ref = grad

    def set_grad(self, g):
        if g is None:
            tp_check(host.tp_tensor_set_grad(self._h, None), "tp_tensor_set_grad")
        else:
            a = np.ascontiguousarray(np.asarray(g, dtype=np.float32)).reshape(-1)
            assert a.size == self.numel()
            tp_check(host.tp_tensor_set_grad(self._h, a.ctypes.data), "tp_tensor_set_grad")

    def tape_node(self):
        n = C.c_size_t()
        tp_check(host.tp_tensor_tape_node(self._h, C.byref(n)), "tp_tensor_tape_node")
        return n.value

    def dptr(self) -> int:
        p = _p()
        tp_check(host.tp_tensor_dptr(self._h, C.byref(p)), "tp_tensor_dptr")
        return p.value

    def backward(self):
        tp_check(host.tp_tensor_backward(self._h), "tp_tensor_backward")

    def zero_grad(self):
        tp_check(host.tp_tensor_zero_grad(self._h), "tp_tensor_zero_grad")

    # -- ops ---------------------------------------------------------------
    def _bin(self, fn, o, name):
        h = _p()
        tp_check(fn(self._h, o._h, C.byref(h)), name)
        return Tensor(_h=h.value)

    def _un(self, fn, name, *args):
        h = _p()
        tp_check(fn(self._h, *args, C.byref(h)), name)
        return Tensor(_h=h.value)

    def __add__(self, o): return self._bin(host.tp_add, o, "add")
    def __sub__(self, o): return self._bin(host.tp_sub, o, "sub")
    def __mul__(self, o): return self._bin(host.tp_mul, o, "mul")
    def __truediv__(self, o): return self._bin(host.tp_div, o, "div")
    def matmul(self, o): return self._bin(host.tp_matmul, o, "matmul")
    def add_broadcast(self, o): return self._bin(host.tp_add_broadcast, o, "add_broadcast")
    def sub_broadcast_rows(self, o): return self._bin(host.tp_sub_broadcast_rows, o, "sub_broadcast_rows")
    def relu(self): return self._un(host.tp_relu, "relu")
    def sigmoid(self): return self._un(host.tp_sigmoid, "sigmoid")
    def transpose(self): return self._un(host.tp_transpose, "transpose")
    def exp(self): return self._un(host.tp_exp, "exp")
    def log(self): return self._un(host.tp_log, "log")
    def mean(self): return self._un(host.tp_mean, "mean")
    def pow(self, e): return self._un(host.tp_pow, "pow", float(e))
    def sqrt(self): return self.pow(0.5)
    def sum(self, dim=None, keepdim=False): return self._un(host.tp_sum, "sum", -1 if dim is None else int(dim), 1 if keepdim else 0)
    def reshape(self, shape): return self._un(host.tp_reshape, "reshape", _shape_arr(shape), len(shape))
    view = reshape
    def flatten(self, start_dim): return self._un(host.tp_flatten, "flatten", int(start_dim))
    def squeeze(self, dim=None): return self._un(host.tp_squeeze, "squeeze", -1 if dim is None else int(dim))
    def slice_channels(self, start, end): return self._un(host.tp_slice_channels, "slice_channels", int(start), int(end))   # nn.rs:862-886

    @staticmethod
    def cat(tensors, dim):   # nn.rs:928-1014
        arr = (C.c_void_p * len(tensors))(*[t._h for t in tensors])
        out = _p()
        tp_check(host.tp_cat(arr, len(tensors), int(dim), C.byref(out)), "cat")
        return Tensor(_h=out.value)
    def unsqueeze(self, dim): return self._un(host.tp_unsqueeze, "unsqueeze", int(dim))

    def max(self, dim=None):
        v, i = _p(), _p()
        tp_check(host.tp_max(self._h, -1 if dim is None else int(dim), C.byref(v), C.byref(i)), "max")
        return Tensor(_h=v.value), Tensor(_h=i.value)

    def argmax(self, dim=None):
        return self.max(dim)[1]

    def linear(self, weight, bias=None, relu=False):
        return self._un(host.tp_linear, "linear", weight._h, bias._h if bias is not None else None, 1 if relu else 0)

    def conv2d(self, weight, bias, stride=(1, 1), padding=(0, 0), dilation=(1, 1), relu=False):
        return self._un(host.tp_conv2d, "conv2d", weight._h, bias._h if bias is not None else None, stride[0], stride[1],
                        padding[0], padding[1], dilation[0], dilation[1], 1 if relu else 0)

    def conv2d_relu(self, weight, bias, stride=(1, 1), padding=(0, 0), dilation=(1, 1)):
        return self.conv2d(weight, bias, stride, padding, dilation, relu=True)

    def max_pool2d(self, kernel_size, stride=None, padding=(0, 0)):
        s = stride or (0, 0)
        return self._un(host.tp_max_pool2d, "max_pool2d", kernel_size[0], kernel_size[1], s[0], s[1], padding[0], padding[1])

    def avg_pool2d(self, kernel_size, stride=None, padding=(0, 0)):
        s = stride or (0, 0)
        return self._un(host.tp_avg_pool2d, "avg_pool2d", kernel_size[0], kernel_size[1], s[0], s[1], padding[0], padding[1])


# -- src/loss.rs ------------------------------------------------------------
def _t1(fn, name, x, *args):
    h = _p()
    tp_check(fn(x._h, *args, C.byref(h)), name)
    return Tensor(_h=h.value)


def log_softmax(x, dim=-1): return _t1(host.tp_log_softmax, "log_softmax", x)
def softmax(x, dim=-1): return _t1(host.tp_softmax, "softmax", x)
def cross_entropy_loss(logits, targets): return _t1(host.tp_cross_entropy_loss, "cross_entropy_loss", logits, targets._h)
def one_hot(idx, num_classes): return _t1(host.tp_one_hot, "one_hot", idx, int(num_classes))
def mse_loss(pred, targets): return _t1(host.tp_mse_loss, "mse_loss", pred, targets._h)
def bce_loss(pred, targets): return _t1(host.tp_bce_loss, "bce_loss", pred, targets._h)
def cross_entropy_loss_onehot(logits, targets): return _t1(host.tp_cross_entropy_loss_onehot, "cross_entropy_loss_onehot", logits, targets._h)


def accuracy(pred, targets) -> float:
    out = C.c_float()
    tp_check(host.tp_accuracy(pred._h, targets._h, C.byref(out)), "accuracy")
    return float(out.value)


# -- src/nn.rs, src/activation.rs ------------------------------------------------
class Module:
    def __init__(self, h):
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            host.tp_module_free(h)

    def forward(self, x: Tensor) -> Tensor:
        h = _p()
        tp_check(host.tp_module_forward(self._h, x._h, C.byref(h)), "forward")
        return Tensor(_h=h.value)

    def parameters(self):
        n = C.c_int()
        tp_check(host.tp_module_num_parameters(self._h, C.byref(n)), "num_parameters")
        out = []
        for i in range(n.value):
            h = _p()
            tp_check(host.tp_module_parameter(self._h, i, C.byref(h)), "parameter")
            out.append(Tensor(_h=h.value))
        return out


def _mk(fn, name, *args):
    h = _p()
    tp_check(fn(*args, C.byref(h)), name)
    return h.value


class Linear(Module):
    def __init__(self, in_features, out_features, with_bias=True, seed=1):
        super().__init__(_mk(host.tp_linear_new, "Linear::new", int(in_features), int(out_features), 1 if with_bias else 0, int(seed)))

    @property
    def weight(self): return self.parameters()[0]

    @property
    def bias(self):
        p = self.parameters()
        return p[1] if len(p) > 1 else None


class ReLU(Module):
    def __init__(self): super().__init__(_mk(host.tp_relu_new, "ReLU"))


class Sigmoid(Module):
    def __init__(self): super().__init__(_mk(host.tp_sigmoid_new, "Sigmoid"))


class Conv2d(Module):
    def __init__(self, in_ch, out_ch, kernel_size, stride=None, padding=None, dilation=None, groups=None, bias=True,
                 seed=1, _relu=False):
        if dilation not in (None, (1, 1)):
            raise TaperError("Conv2d: dilation is out of scope (SURVEY.md section 2 row 8)")
        s, p = stride or (1, 1), padding or (0, 0)
        # groups > 1 (nn.rs:289-332): slice / conv per group / cat, forward only like the reference
        super().__init__(_mk(host.tp_conv2d_grouped_new, "Conv2d::new", int(in_ch), int(out_ch), kernel_size[0], kernel_size[1],
                             s[0], s[1], p[0], p[1], int(groups or 1), 1 if bias else 0, 1 if _relu else 0, int(seed)))


class Conv2dReLU(Conv2d):
    def __init__(self, in_ch, out_ch, kernel_size, stride=None, padding=None, dilation=None, groups=None, bias=True, seed=1):
        super().__init__(in_ch, out_ch, kernel_size, stride, padding, dilation, groups, bias, seed, _relu=True)


class MaxPool2d(Module):
    def __init__(self, kernel_size, stride=None, padding=None):
        s, p = stride or (0, 0), padding or (0, 0)
        super().__init__(_mk(host.tp_maxpool2d_new, "MaxPool2d::new", kernel_size[0], kernel_size[1], s[0], s[1], p[0], p[1]))


class AvgPool2d(Module):
    def __init__(self, kernel_size, stride=None, padding=None):
        s, p = stride or (0, 0), padding or (0, 0)
        super().__init__(_mk(host.tp_avgpool2d_new, "AvgPool2d::new", kernel_size[0], kernel_size[1], s[0], s[1], p[0], p[1]))


class AdaptiveAvgPool2d(Module):
    def __init__(self, output_size):
        super().__init__(_mk(host.tp_adaptive_avgpool2d_new, "AdaptiveAvgPool2d::new", output_size[0], output_size[1]))

    @staticmethod
    def global_():
        return AdaptiveAvgPool2d((1, 1))


class Flatten(Module):
    def __init__(self, start_dim=1):
        super().__init__(_mk(host.tp_flatten_new, "Flatten::new", int(start_dim)))


class Dropout(Module):
    """nn.rs:773-827; the mask comes from a counter-based generator (the reference's RNG is unseeded)"""

    def __init__(self, p, seed=0x64726F70):
        super().__init__(_mk(host.tp_dropout_new, "Dropout::new", float(p), int(seed)))

    def eval(self): tp_check(host.tp_dropout_set_training(self._h, 0), "Dropout::eval")
    def train(self): tp_check(host.tp_dropout_set_training(self._h, 1), "Dropout::train")

    def last_mask(self) -> Tensor:
        h = _p()
        tp_check(host.tp_dropout_last_mask(self._h, C.byref(h)), "Dropout::last_mask")
        return Tensor(_h=h.value)


class Sequential(Module):
    def __init__(self, layers, fuse=True):
        self.layers = list(layers)  # keep the children alive
        arr = (_p * len(self.layers))(*[l._h for l in self.layers])
        super().__init__(_mk(host.tp_sequential_new, "Sequential::new", arr, len(self.layers), 1 if fuse else 0))


# -- src/optim.rs -----------------------------------------------------------------
class _Optim:
    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            host.tp_optim_free(h)

    def step(self): tp_check(host.tp_optim_step(self._h), "Optimizer::step")
    def zero_grad(self): tp_check(host.tp_optim_zero_grad(self._h), "Optimizer::zero_grad")

    def total(self) -> int:
        n = C.c_int64()
        tp_check(host.tp_optim_total(self._h, C.byref(n)), "tp_optim_total")
        return n.value


class Adam(_Optim):
    def __init__(self, params, lr, betas=None, eps=None, weight_decay=None):
        betas = betas or (0.9, 0.999)
        self.params = list(params)
        arr = (_p * len(self.params))(*[p._h for p in self.params])
        self._h = _mk(host.tp_adam_new, "Adam::new", arr, len(self.params), float(lr), float(betas[0]), float(betas[1]),
                      float(1e-8 if eps is None else eps), float(0.0 if weight_decay is None else weight_decay))

    def set_lr(self, lr): tp_check(host.tp_adam_set_lr(self._h, float(lr)), "Adam::set_lr")

    def get_lr(self):
        v = C.c_float()
        tp_check(host.tp_adam_get_lr(self._h, C.byref(v)), "Adam::get_lr")
        return float(v.value)

    def t(self):
        v = C.c_int()
        tp_check(host.tp_adam_t(self._h, C.byref(v)), "Adam::t")
        return v.value

    def moments(self):
        n = sum(p.numel() for p in self.params)
        m, v = np.empty(n, np.float32), np.empty(n, np.float32)
        tp_check(host.tp_adam_moments(self._h, m.ctypes.data, v.ctypes.data), "Adam::moments")
        return m, v


    def load_state(self, t, m, v):
        m, v = np.ascontiguousarray(m, np.float32), np.ascontiguousarray(v, np.float32)
        tp_check(host.tp_adam_load_state(self._h, int(t), m.ctypes.data, v.ctypes.data), "Adam::load_state")


class AdamW(Adam):
    """optim.rs:130-180: decoupled decay on every weight, then Adam with weight_decay = 0"""

    def __init__(self, params, lr, betas=None, eps=None, weight_decay=None):
        betas = betas or (0.9, 0.999)
        self.params = list(params)
        arr = (_p * len(self.params))(*[p._h for p in self.params])
        self._h = _mk(host.tp_adamw_new, "AdamW::new", arr, len(self.params), float(lr), float(betas[0]), float(betas[1]),
                      float(1e-8 if eps is None else eps), float(0.0 if weight_decay is None else weight_decay))


class _Scheduler:
    """optim.rs:183-188 LRScheduler: step(metrics: Option<f32>), get_lr()"""

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            host.tp_sched_free(h)

    def step(self, metrics=None):
        m = None if metrics is None else C.byref(C.c_float(float(metrics)))
        tp_check(host.tp_sched_step(self._h, m), "LRScheduler::step")

    def get_lr(self) -> float:
        v = C.c_float()
        tp_check(host.tp_sched_get_lr(self._h, C.byref(v)), "LRScheduler::get_lr")
        return float(v.value)


class StepLR(_Scheduler):
    def __init__(self, base_lr, step_size, gamma):
        self._h = _mk(host.tp_sched_step_lr, "StepLR::new", float(base_lr), int(step_size), float(gamma))


class ExponentialLR(_Scheduler):
    def __init__(self, base_lr, gamma):
        self._h = _mk(host.tp_sched_exponential, "ExponentialLR::new", float(base_lr), float(gamma))


class CosineAnnealingLR(_Scheduler):
    def __init__(self, base_lr, t_max, min_lr=None):
        self._h = _mk(host.tp_sched_cosine, "CosineAnnealingLR::new", float(base_lr), int(t_max), float(min_lr or 0.0))


class ReduceLROnPlateau(_Scheduler):
    def __init__(self, initial_lr, factor, patience, min_lr=None, mode=None):
        self._h = _mk(host.tp_sched_plateau, "ReduceLROnPlateau::new", float(initial_lr), float(factor), int(patience),
                      float(1e-6 if min_lr is None else min_lr), 1 if (mode or "min") == "max" else 0)


def format_f32(v) -> str:
    """an f32 as Rust's `{}` prints it (the checkpoint's number format, train.rs:283-285)"""
    buf = C.create_string_buffer(160)
    tp_check(host.tp_format_f32(float(v), buf, len(buf)), "format_f32")
    return buf.value.decode()


class SGD(_Optim):
    def __init__(self, params, lr, momentum=None):
        self.params = list(params)
        arr = (_p * len(self.params))(*[p._h for p in self.params])
        self._h = _mk(host.tp_sgd_new, "SGD::new", arr, len(self.params), float(lr))


# -- src/data/mnist.rs --------------------------------------------------------------
class MNISTDataset:
    def __init__(self, h):
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            host.tp_dataset_free(h)

    @staticmethod
    def from_host(images, labels, train=True):
        im = np.ascontiguousarray(images, dtype=np.float32).reshape(-1, 784)
        lb = np.ascontiguousarray(labels, dtype=np.float32).reshape(-1)
        return MNISTDataset(_mk(host.tp_dataset_from_host, "MNISTDataset::from_host", im.ctypes.data, lb.ctypes.data, lb.size, 1 if train else 0))

    @staticmethod
    def from_idx(images_path, labels_path, train=True):
        return MNISTDataset(_mk(host.tp_dataset_from_idx, "MNISTDataset::from_idx", str(images_path).encode(), str(labels_path).encode(), 1 if train else 0))

    @staticmethod
    def synthetic(n, seed=0x7461706572, train=True):
        return MNISTDataset(_mk(host.tp_dataset_synthetic, "MNISTDataset::synthetic", int(n), int(seed), 1 if train else 0))

    def len(self):
        n = C.c_size_t()
        tp_check(host.tp_dataset_len(self._h, C.byref(n)), "len")
        return n.value

    def tensors(self):
        a, b = _p(), _p()
        tp_check(host.tp_dataset_tensors(self._h, C.byref(a), C.byref(b)), "tensors")
        return Tensor(_h=a.value), Tensor(_h=b.value)


class DataLoader:
    def __init__(self, dataset, batch_size, shuffle, seed=0x7461706572):
        self.dataset = dataset
        self._h = _mk(host.tp_loader_new, "DataLoader::new", dataset._h, int(batch_size), 1 if shuffle else 0, int(seed))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            host.tp_loader_free(h)

    def reset(self): tp_check(host.tp_loader_reset(self._h), "DataLoader::reset")

    def num_batches(self):
        # (dataset length and batch size are fixed at construction: asked once -- the Trainer wrappers ask on every call)
        n = getattr(self, "_nb", None)
        if n is None:
            c = C.c_size_t()
            tp_check(host.tp_loader_num_batches(self._h, C.byref(c)), "num_batches")
            n = self._nb = c.value
        return n

    def __iter__(self):
        return self

    def __next__(self):
        a, b, has = _p(), _p(), C.c_int()
        tp_check(host.tp_loader_next(self._h, C.byref(a), C.byref(b), C.byref(has)), "DataLoader::next")
        if not has.value:
            raise StopIteration
        return Tensor(_h=a.value), Tensor(_h=b.value)


# -- data parallel --------------------------------------------------------------------
class Communicator:
    """RCCL communicator, one process per GPU; the 128-byte id travels out of band."""

    BLOB_BYTES = 192

    def __init__(self, n_ranks, rank, uid: bytes | None = None, _h=None):
        self.n_ranks, self.rank = n_ranks, rank
        self._p2p = _h is not None
        if _h is not None:
            self._h = _h
            return
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._h = _mk(host.tp_comm_new, "Communicator::new", int(n_ranks), int(rank), buf)

    @staticmethod
    def p2p(n_ranks, rank):
        """peer-to-peer communicator (one node, <= 8 ranks): one-shot all-reduce of the gradient arena fused with Adam"""
        return Communicator(n_ranks, rank, _h=_mk(host.tp_comm_new_p2p, "Communicator::p2p", int(n_ranks), int(rank)))

    @staticmethod
    def loopback():
        """W = 2 with this process as its own peer: the exchange inside the gradient launch (th_mlp_tail_dp) runs every push, flag, poll
        and load of its protocol through local memory; results are the single-GPU step's, bit for bit"""
        return Communicator(2, 0, _h=_mk(host.tp_comm_new_loopback, "Communicator::loopback"))

    def set_inkernel(self, on: bool):
        """False: never the in-launch exchange (the three-launch form: gradient launch, then all-reduce + Adam) -- A/B runs"""
        tp_check(host.tp_comm_set_inkernel(self._h, 1 if on else 0), "Communicator::set_inkernel")

    def inkernel_launches(self) -> int:
        out = C.c_int64()
        tp_check(host.tp_comm_inkernel_launches(self._h, C.byref(out)), "Communicator::inkernel_launches")
        return int(out.value)

    def exchange_selftest(self, slots: int = 16, rounds: int = 3) -> int:
        """collective: the in-launch exchange alone on known patterns; -> mismatches + time-outs seen by this rank"""
        out = C.c_int()
        tp_check(host.tp_comm_exchange_selftest(self._h, int(slots), int(rounds), C.byref(out)), "Communicator::exchange_selftest")
        return int(out.value)

    def exchange_form(self) -> int:
        """0: no in-launch exchange; 1: one-shot (every rank reduces every slice); 2: two-shot (slice s reduced by rank s % W; 4 ranks and more)"""
        out = C.c_int()
        tp_check(host.tp_comm_exchange_form(self._h, C.byref(out)), "Communicator::exchange_form")
        return int(out.value)

    def ranks_on_this_device(self) -> int:
        out = C.c_int()
        tp_check(host.tp_comm_ranks_on_this_device(self._h, C.byref(out)), "Communicator::ranks_on_this_device")
        return int(out.value)

    def tail_exchange_ok(self, batch: int, in_features: int, hidden: int, classes: int) -> bool:
        out = C.c_int()
        tp_check(host.tp_comm_tail_exchange_ok(self._h, int(batch), int(in_features), int(hidden), int(classes), C.byref(out)), "Communicator::tail_exchange_ok")
        return bool(out.value)

    def export_arena(self, optimizer, fine_grained: bool = False) -> bytes:
        """register the optimizer's gradient arena; fine_grained: move it into fine-grained (cross-agent coherent) device memory first"""
        buf = (C.c_uint8 * self.BLOB_BYTES)()
        tp_check(host.tp_comm_export_arena_ex(self._h, optimizer._h, 1 if fine_grained else 0, buf), "Communicator::export_arena")
        return bytes(buf)

    def connect(self, blobs: bytes):
        buf = (C.c_uint8 * len(blobs)).from_buffer_copy(blobs)
        tp_check(host.tp_comm_connect(self._h, buf, len(blobs)), "Communicator::connect")

    def is_p2p(self) -> bool:
        return self._p2p

    def self_check(self, optimizer, rounds: int = 3) -> bool:
        """collective: `rounds` all-reduces of per-rank / per-round / per-element patterns through the SAME arena addresses, through the
        in-place kernel and the fused all-reduce + Adam kernel (p / m / v against the closed form; state restored)"""
        ok = C.c_int()
        tp_check(host.tp_comm_self_check_rounds(self._h, optimizer._h, int(rounds), C.byref(ok)), "Communicator::self_check")
        return bool(ok.value)

    def failed(self) -> bool:
        """a peer never arrived at some all-reduce (host-visible error word; no stream synchronisation)"""
        out = C.c_int()
        tp_check(host.tp_comm_failed(self._h, C.byref(out)), "Communicator::failed")
        return bool(out.value)

    def set_timeout_ms(self, ms: int):
        tp_check(host.tp_comm_set_timeout_ms(self._h, int(ms)), "Communicator::set_timeout_ms")

    def set_fuse_adam(self, on: bool):
        tp_check(host.tp_comm_set_fuse_adam(self._h, 1 if on else 0), "Communicator::set_fuse_adam")

    def stats(self):
        out = (C.c_int64 * 2)()
        tp_check(host.tp_comm_stats(self._h, out), "Communicator::stats")
        return dict(inplace=int(out[0]), fused=int(out[1]))

    def timed_out(self) -> bool:
        out = C.c_int()
        tp_check(host.tp_comm_timed_out(self._h, C.byref(out)), "Communicator::timed_out")
        return bool(out.value)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            host.tp_comm_free(h)

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        tp_check(host.tp_comm_unique_id(buf), "Communicator::unique_id")
        return bytes(buf)

    def allreduce_mean(self, d_ptr: int, n: int):
        tp_check(host.tp_comm_allreduce_mean(self._h, d_ptr, int(n)), "allreduce_mean")

    def count(self) -> int:
        """ranks the communicator spans (RCCL: ncclCommCount)"""
        out = C.c_int()
        tp_check(host.tp_comm_count(self._h, C.byref(out)), "Communicator::count")
        return int(out.value)

    def time_exchange(self, optimizer, reps: int = 200) -> float:
        """collective: us per gradient exchange + Adam as a Trainer step issues it (events on the stream); moves the optimizer state"""
        out = C.c_float()
        tp_check(host.tp_comm_time_exchange(self._h, optimizer._h, int(reps), C.byref(out)), "Communicator::time_exchange")
        return float(out.value)


# -- src/train.rs ---------------------------------------------------------------------
class Trainer:
    """train.rs:74-172 + the captured-graph epoch driver."""

    EAGER, GRAPH, EVAL = 0, 1, 2

    def __init__(self, model: Module, optimizer: Adam, sample_shape=None, comm: Communicator | None = None,
                 graph_chunk: int = 128, fuse_head: bool | int = True, fuse_adam: bool = True, scheduler=None):
        self.model, self.optimizer, self.comm, self.scheduler = model, optimizer, comm, scheduler
        self._h = _mk(host.tp_trainer_new, "Trainer::new", model._h, optimizer._h)
        # fuse_head: False / 0 = off, 1 = classifier head only, True / 2 = head + hidden-layer backward in one launch
        level = 2 if fuse_head is True else int(fuse_head)
        tp_check(host.tp_trainer_set_options(self._h, int(graph_chunk), level, 1 if fuse_adam else 0), "set_options")
        if sample_shape:
            tp_check(host.tp_trainer_set_sample_shape(self._h, _shape_arr(sample_shape), len(sample_shape)), "set_sample_shape")
        if comm is not None:
            tp_check(host.tp_trainer_set_comm(self._h, comm._h), "set_comm")
        if scheduler is not None:
            tp_check(host.tp_trainer_set_scheduler(self._h, scheduler._h), "set_scheduler")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            host.tp_trainer_free(h)

    def train_step(self, images: Tensor, labels: Tensor):
        loss, acc = C.c_float(), C.c_float()
        tp_check(host.tp_trainer_train_step(self._h, images._h, labels._h, C.byref(loss), C.byref(acc)), "train_step")
        return float(loss.value), float(acc.value)

    def run_epoch(self, loader: DataLoader, mode=GRAPH, max_steps=0):
        # (the out-parameters and the per-step buffer are kept between calls: a 20-step call is 250 us, and allocating them was 3 of it)
        nb_max = loader.num_batches()
        st = getattr(self, "_epoch_out", None)
        if st is None or st[0].size < 2 * nb_max:
            st = (np.empty(2 * nb_max, dtype=np.float32), C.c_float(), C.c_float(), C.c_size_t(), C.c_size_t(), C.c_size_t())
            st = st + (st[0].ctypes.data, tuple(C.byref(v) for v in st[1:6]))
            self._epoch_out = st
        per, avg, acc, tc, ts, nb, per_ptr, refs = st
        tp_check(host.tp_trainer_run_epoch(self._h, loader._h, int(mode), int(max_steps), *refs, per_ptr, per.size), "run_epoch")
        n = nb.value
        return dict(avg_loss=avg.value, accuracy=acc.value, total_correct=tc.value, total_samples=ts.value,
                    num_batches=n, losses=per[0:2 * n:2].copy(), ncorrect=per[1:2 * n:2].copy())

    def train_epoch(self, loader): return self.run_epoch(loader, self.EAGER)
    def train_epoch_graph(self, loader, max_steps=0): return self.run_epoch(loader, self.GRAPH, max_steps)
    def evaluate(self, loader): return self.run_epoch(loader, self.EVAL)

    # -- train.rs:175-292 ---------------------------------------------------------
    def fit(self, train_loader: DataLoader, val_loader: DataLoader, epochs: int, verbose: bool = False, graph: bool = True):
        tp_check(host.tp_trainer_fit(self._h, train_loader._h, val_loader._h, int(epochs), 1 if verbose else 0, 1 if graph else 0), "fit")
        return self.metrics()

    def metrics(self):
        out = {}
        for which, name in enumerate(("train_loss", "train_acc", "val_loss", "val_acc", "epoch_times")):
            n = C.c_size_t()
            tp_check(host.tp_trainer_metrics(self._h, which, None, 0, C.byref(n)), "metrics")
            buf = np.zeros(n.value, np.float32)
            tp_check(host.tp_trainer_metrics(self._h, which, buf.ctypes.data, buf.size, None), "metrics")
            out[name] = buf
        return out

    def metrics_text(self, summary=False) -> str:
        buf = C.create_string_buffer(2048)
        tp_check(host.tp_trainer_metrics_text(self._h, 1 if summary else 0, buf, len(buf)), "metrics_text")
        return buf.value.decode()

    def save_checkpoint(self, path): tp_check(host.tp_trainer_save_checkpoint(self._h, str(path).encode()), "save_checkpoint")
    def load_checkpoint(self, path): tp_check(host.tp_trainer_load_checkpoint(self._h, str(path).encode()), "load_checkpoint")
    def save_optimizer_state(self, path): tp_check(host.tp_trainer_save_optimizer_state(self._h, str(path).encode()), "save_optimizer_state")
    def load_optimizer_state(self, path): tp_check(host.tp_trainer_load_optimizer_state(self._h, str(path).encode()), "load_optimizer_state")

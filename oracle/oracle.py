"""ctypes front-end of the CPU oracle (oracle/taper_oracle.{h,c}).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never from taper_amd/.  The class and
method names follow the reference's Rust API (src/tensor.rs, src/ops.rs,
src/loss.rs, src/optim.rs, src/nn.rs) so tests read like tests/smoke.rs.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libtaper_oracle.so"


def build(force: bool = False) -> Path:
    srcs = [_HERE / "taper_oracle.c", _HERE / "taper_oracle_nn.c", _HERE / "taper_oracle.h"]
    stale = (not _SO.exists()) or any(s.stat().st_mtime > _SO.stat().st_mtime for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", str(_HERE), "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


def build_native(outdir: str) -> Path:
    """-march=native copy for the cpu_baseline timing on the box it runs on."""
    subprocess.check_call(["make", "-C", str(_HERE), "-s", "native", f"OUTDIR={outdir}"], stdout=subprocess.DEVNULL)
    return Path(outdir) / "libtaper_oracle_native.so"


def build_fast(outdir: str) -> Path:
    """bench.py's cpu_baseline build (`make fast`): -march=native, the matrixmultiply-style packed sgemm
    (cpu_packed_sgemm.c) and the reference's rayon loops as OpenMP loops.  Not the parity oracle."""
    subprocess.check_call(["make", "-C", str(_HERE), "-s", "fast", f"OUTDIR={outdir}"], stdout=subprocess.DEVNULL)
    return Path(outdir) / "libtaper_oracle_fast.so"


def find_cblas():
    """A vendor cblas_sgemm on this box for the `--features blas` analogue (src/gemm.rs:32-47), or None.
    Looks for the OpenBLAS that numpy's wheel bundles (ILP64, symbol prefix scipy_) and for system libraries.
    -> (ctypes function address, ilp64 flag, description, CDLL kept alive)"""
    import glob
    cands = []
    try:
        import numpy
        base = Path(numpy.__file__).resolve().parent
        cands += sorted(glob.glob(str(base.parent / "numpy.libs" / "lib*openblas*.so*")))
    except Exception:
        pass
    for pat in ("/usr/lib/x86_64-linux-gnu/libopenblas.so*", "/usr/lib/x86_64-linux-gnu/libblas.so*", "/usr/lib64/libopenblas.so*",
                "/opt/intel/oneapi/mkl/latest/lib/intel64/libmkl_rt.so*"):
        cands += sorted(glob.glob(pat))
    for path in cands:
        try:
            dll = C.CDLL(path)
        except OSError:
            continue
        for sym, ilp64 in (("cblas_sgemm", 0), ("scipy_cblas_sgemm64_", 1), ("cblas_sgemm64_", 1)):
            fn = getattr(dll, sym, None)
            if fn is not None:
                return C.cast(fn, C.c_void_p).value, ilp64, f"{Path(path).name}:{sym}", dll
    return None


_p = C.c_void_p
_f32p = C.POINTER(C.c_float)
_szp = C.POINTER(C.c_size_t)


class _TensorStruct(C.Structure):
    _fields_ = [("core", _p), ("shape", C.c_size_t * 4), ("ndim", C.c_int), ("requires_grad", C.c_int)]


class _LayerStruct(C.Structure):
    _fields_ = [("kind", C.c_int), ("w", _p), ("b", _p),
                ("k_h", C.c_int), ("k_w", C.c_int), ("s_h", C.c_int), ("s_w", C.c_int),
                ("p_h", C.c_int), ("p_w", C.c_int), ("out_h", C.c_int), ("out_w", C.c_int),
                ("start_dim", C.c_int)]


class _ModelStruct(C.Structure):
    _fields_ = [("layers", C.POINTER(_LayerStruct)), ("n_layers", C.c_int), ("conv_mode", C.c_int)]


def _load(path: Path | None = None):
    lib = C.CDLL(str(path or build()))
    sig = {
        "ot_tape_reset": (None, []), "ot_tape_len": (C.c_size_t, []),
        "ot_tape_set_zero_sentinel": (None, [C.c_int]),
        "ot_new": (_p, [_f32p, _szp, C.c_int]), "ot_scalar": (_p, [C.c_float]),
        "ot_clone": (_p, [_p]), "ot_free": (None, [_p]),
        "ot_set_requires_grad": (None, [_p, C.c_int]), "ot_len": (C.c_size_t, [_p]),
        "ot_data": (_f32p, [_p]), "ot_data_mut": (_f32p, [_p]), "ot_grad": (_f32p, [_p]),
        "ot_set_grad": (None, [_p, _f32p]), "ot_tape_node": (C.c_size_t, [_p]),
        "ot_backward": (None, [_p]), "ot_zero_grad": (None, [_p]),
        "ot_sgemm_rowmajor": (None, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _f32p, _f32p, C.c_float, _f32p]),
        "ot_sum": (_p, [_p, C.c_int, C.c_int]), "ot_max": (_p, [_p, C.c_int, C.POINTER(_p)]),
        "ot_argmax": (_p, [_p, C.c_int]), "ot_pow": (_p, [_p, C.c_float]),
        "ot_reshape": (_p, [_p, _szp, C.c_int]), "ot_flatten": (_p, [_p, C.c_int]),
        "ot_squeeze": (_p, [_p, C.c_int]), "ot_unsqueeze": (_p, [_p, C.c_int]),
        "ot_conv2d": (_p, [_p, _p, _p] + [C.c_int] * 7),
        "ot_conv2d_relu": (_p, [_p, _p, _p] + [C.c_int] * 7),
        "ot_conv2d_direct_3x3": (_p, [_p, _p, _p] + [C.c_int] * 4),
        "ot_max_pool2d": (_p, [_p] + [C.c_int] * 7 + [C.POINTER(C.c_int64)]),
        "ot_avg_pool2d": (_p, [_p] + [C.c_int] * 6),
        "ot_adaptive_avg_pool2d": (_p, [_p, C.c_int, C.c_int]),
        "ot_cross_entropy_loss": (_p, [_p, _p]), "ot_accuracy": (C.c_float, [_p, _p]),
        "ot_one_hot": (_p, [_p, C.c_int]), "ot_bce_loss": (_p, [_p, _p]), "ot_mse_loss": (_p, [_p, _p]),
        "ot_linear_forward": (_p, [_p, _p, _p]),
        "ot_adam_new": (_p, [C.POINTER(_p), C.c_int] + [C.c_float] * 5),
        "ot_adam_step": (None, [_p]), "ot_adam_zero_grad": (None, [_p]),
        "ot_adam_set_lr": (None, [_p, C.c_float]), "ot_adam_get_lr": (C.c_float, [_p]),
        "ot_adam_t": (C.c_int, [_p]), "ot_adam_m": (_f32p, [_p, C.c_int]), "ot_adam_v": (_f32p, [_p, C.c_int]),
        "ot_adam_free": (None, [_p]), "ot_sgd_step": (None, [C.POINTER(_p), C.c_int, C.c_float]),
        "ot_powi": (C.c_float, [C.c_float, C.c_int]),
        "ot_get_batch": (None, [_f32p, _f32p, _szp, C.c_size_t, _f32p, _f32p]),
        "ot_model_forward": (_p, [C.POINTER(_ModelStruct), _p]),
        "ot_train_step": (None, [C.POINTER(_ModelStruct), _p, _f32p, _f32p, _szp, C.c_int,
                                 _f32p, _f32p, _f32p, _f32p, C.POINTER(C.c_int)]),
    }
    # only in the cpu_baseline build (`make fast`)
    for name, proto in (("ot_baseline_flavour", (C.c_int, [])), ("ot_baseline_max_threads", (C.c_int, [])),
                        ("ot_baseline_set_threads", (None, [C.c_int])), ("ot_baseline_set_cblas", (None, [_p, C.c_int])),
                        ("ot_baseline_run_steps", (None, [C.POINTER(_ModelStruct), _p, _f32p, _f32p, C.c_size_t, _szp, C.c_int, C.c_size_t, _f32p]))):
        if hasattr(lib, name):
            sig[name] = proto
    for name in ("ot_add", "ot_mul", "ot_sub", "ot_div", "ot_matmul", "ot_add_broadcast", "ot_sub_broadcast_rows"):
        sig[name] = (_p, [_p, _p])
    for name in ("ot_relu", "ot_transpose", "ot_sigmoid", "ot_mean", "ot_exp", "ot_log", "ot_sqrt",
                 "ot_log_softmax", "ot_softmax"):
        sig[name] = (_p, [_p])
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def use_library(path) -> None:
    """Switch to another build of the same sources (e.g. the -march=native one)."""
    global lib
    lib = _load(Path(path))


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def _shape_arr(shape):
    return (C.c_size_t * len(shape))(*[int(s) for s in shape])


class Tape:
    """src/tape.rs"""

    @staticmethod
    def reset():
        lib.ot_tape_reset()

    @staticmethod
    def len():
        return lib.ot_tape_len()

    @staticmethod
    def set_zero_sentinel(on: bool):
        lib.ot_tape_set_zero_sentinel(1 if on else 0)


class Tensor:
    """src/tensor.rs Tensor (handle semantics: data/grad shared by clones)."""

    def __init__(self, data=None, shape=None, _h=None):
        if _h is not None:
            self._h = _h
            return
        a = np.ascontiguousarray(np.asarray(data, dtype=np.float32))
        if shape is None:
            shape = a.shape if a.ndim else (1,)
        a = a.reshape(-1)
        assert a.size == int(np.prod(shape)), "data/shape mismatch"
        self._h = lib.ot_new(_fp(a), _shape_arr(shape), len(shape))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            try:
                lib.ot_free(h)
            except Exception:
                pass

    # -- constructors ----------------------------------------------------
    @staticmethod
    def scalar(v):
        return Tensor(_h=lib.ot_scalar(float(v)))

    def requires_grad(self):
        lib.ot_set_requires_grad(self._h, 1)
        return self

    def clone(self):
        return Tensor(_h=lib.ot_clone(self._h))

    # -- accessors -------------------------------------------------------
    @property
    def _s(self):
        return C.cast(self._h, C.POINTER(_TensorStruct)).contents

    def shape(self):
        s = self._s
        return tuple(int(s.shape[i]) for i in range(s.ndim))

    def numel(self):
        return int(lib.ot_len(self._h))

    def data(self) -> np.ndarray:
        n = self.numel()
        return np.ctypeslib.as_array(lib.ot_data(self._h), shape=(n,)).copy().reshape(self.shape())

    def set_data(self, a):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float32)).reshape(-1)
        assert a.size == self.numel()
        C.memmove(lib.ot_data_mut(self._h), a.ctypes.data, a.nbytes)

    def grad(self):
        g = lib.ot_grad(self._h)
        if not g:
            return None
        return np.ctypeslib.as_array(g, shape=(self.numel(),)).copy().reshape(self.shape())

    grad_ref = grad

    def set_grad(self, g):
        if g is None:
            lib.ot_set_grad(self._h, None)
        else:
            a = np.ascontiguousarray(np.asarray(g, dtype=np.float32)).reshape(-1)
            assert a.size == self.numel()
            lib.ot_set_grad(self._h, _fp(a))

    def tape_node(self):
        return int(lib.ot_tape_node(self._h))

    def backward(self):
        lib.ot_backward(self._h)

    def zero_grad(self):
        lib.ot_zero_grad(self._h)

    # -- ops -------------------------------------------------------------
    def _bin(self, fn, o):
        return Tensor(_h=fn(self._h, o._h))

    def __add__(self, o): return self._bin(lib.ot_add, o)
    def __sub__(self, o): return self._bin(lib.ot_sub, o)
    def __mul__(self, o): return self._bin(lib.ot_mul, o)
    def __truediv__(self, o): return self._bin(lib.ot_div, o)
    def matmul(self, o): return self._bin(lib.ot_matmul, o)
    def add_broadcast(self, o): return self._bin(lib.ot_add_broadcast, o)
    def sub_broadcast_rows(self, o): return self._bin(lib.ot_sub_broadcast_rows, o)
    def relu(self): return Tensor(_h=lib.ot_relu(self._h))
    def transpose(self): return Tensor(_h=lib.ot_transpose(self._h))
    def sigmoid(self): return Tensor(_h=lib.ot_sigmoid(self._h))
    def mean(self): return Tensor(_h=lib.ot_mean(self._h))
    def exp(self): return Tensor(_h=lib.ot_exp(self._h))
    def log(self): return Tensor(_h=lib.ot_log(self._h))
    def sqrt(self): return Tensor(_h=lib.ot_sqrt(self._h))
    def pow(self, e): return Tensor(_h=lib.ot_pow(self._h, float(e)))
    def reshape(self, shape): return Tensor(_h=lib.ot_reshape(self._h, _shape_arr(shape), len(shape)))
    view = reshape
    def flatten(self, start_dim): return Tensor(_h=lib.ot_flatten(self._h, int(start_dim)))
    def squeeze(self, dim=None): return Tensor(_h=lib.ot_squeeze(self._h, -1 if dim is None else int(dim)))
    def unsqueeze(self, dim): return Tensor(_h=lib.ot_unsqueeze(self._h, int(dim)))

    def sum(self, dim=None, keepdim=False):
        return Tensor(_h=lib.ot_sum(self._h, -1 if dim is None else int(dim), 1 if keepdim else 0))

    def max(self, dim=None):
        idx = _p()
        v = lib.ot_max(self._h, -1 if dim is None else int(dim), C.byref(idx))
        return Tensor(_h=v), Tensor(_h=idx.value)

    def argmax(self, dim=None):
        return Tensor(_h=lib.ot_argmax(self._h, -1 if dim is None else int(dim)))

    def conv2d(self, weight, bias, stride=(1, 1), padding=(0, 0), dilation=(1, 1), mode=0):
        return Tensor(_h=lib.ot_conv2d(self._h, weight._h, bias._h if bias is not None else None,
                                       stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1], mode))

    def conv2d_relu(self, weight, bias, stride=(1, 1), padding=(0, 0), dilation=(1, 1), mode=0):
        return Tensor(_h=lib.ot_conv2d_relu(self._h, weight._h, bias._h if bias is not None else None,
                                            stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1], mode))

    def conv2d_direct_3x3(self, weight, bias, stride=(1, 1), padding=(1, 1)):
        return Tensor(_h=lib.ot_conv2d_direct_3x3(self._h, weight._h, bias._h if bias is not None else None,
                                                  stride[0], stride[1], padding[0], padding[1]))

    def max_pool2d(self, kernel_size, stride=None, padding=(0, 0), zero_first=True, return_indices=False):
        s = stride or (0, 0)
        n, c, h, w = self.shape()
        sh, sw = stride or kernel_size
        ho = (h + 2 * padding[0] - kernel_size[0]) // sh + 1
        wo = (w + 2 * padding[1] - kernel_size[1]) // sw + 1
        idx = np.zeros(n * c * ho * wo, dtype=np.int64)
        out = Tensor(_h=lib.ot_max_pool2d(self._h, kernel_size[0], kernel_size[1], s[0], s[1], padding[0], padding[1],
                                          1 if zero_first else 0, idx.ctypes.data_as(C.POINTER(C.c_int64))))
        return (out, idx.reshape(n, c, ho, wo)) if return_indices else out

    def avg_pool2d(self, kernel_size, stride=None, padding=(0, 0)):
        s = stride or (0, 0)
        return Tensor(_h=lib.ot_avg_pool2d(self._h, kernel_size[0], kernel_size[1], s[0], s[1], padding[0], padding[1]))

    def adaptive_avg_pool2d(self, output_size):
        return Tensor(_h=lib.ot_adaptive_avg_pool2d(self._h, output_size[0], output_size[1]))


# -- src/gemm.rs ---------------------------------------------------------
def sgemm_rowmajor(trans_a, trans_b, m, n, k, alpha, a, b, beta, c):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1)
    assert c.dtype == np.float32 and c.flags.c_contiguous
    lib.ot_sgemm_rowmajor(int(trans_a), int(trans_b), m, n, k, float(alpha), _fp(a), _fp(b), float(beta), _fp(c))
    return c


# -- src/loss.rs ---------------------------------------------------------
def log_softmax(x, dim=-1): return Tensor(_h=lib.ot_log_softmax(x._h))
def softmax(x, dim=-1): return Tensor(_h=lib.ot_softmax(x._h))
def cross_entropy_loss(logits, targets): return Tensor(_h=lib.ot_cross_entropy_loss(logits._h, targets._h))
def accuracy(pred, targets): return float(lib.ot_accuracy(pred._h, targets._h))
def one_hot(idx, num_classes): return Tensor(_h=lib.ot_one_hot(idx._h, int(num_classes)))
def bce_loss(pred, targets): return Tensor(_h=lib.ot_bce_loss(pred._h, targets._h))
def mse_loss(pred, targets): return Tensor(_h=lib.ot_mse_loss(pred._h, targets._h))
def powi(a, b): return float(lib.ot_powi(float(a), int(b)))


def get_batch(images, labels, indices):
    """data/mnist.rs:277-310"""
    images = np.ascontiguousarray(images, dtype=np.float32)
    labels = np.ascontiguousarray(labels, dtype=np.float32)
    idx = np.ascontiguousarray(indices, dtype=np.uintp)
    xi = np.empty((idx.size, 784), dtype=np.float32)
    yi = np.empty(idx.size, dtype=np.float32)
    lib.ot_get_batch(_fp(images.reshape(-1)), _fp(labels), idx.ctypes.data_as(_szp), idx.size, _fp(xi.reshape(-1)), _fp(yi))
    return xi, yi


# -- src/optim.rs --------------------------------------------------------
class Adam:
    def __init__(self, params, lr, betas=None, eps=None, weight_decay=None):
        betas = betas or (0.9, 0.999)
        self.params = list(params)
        arr = (_p * len(self.params))(*[p._h for p in self.params])
        self._h = lib.ot_adam_new(arr, len(self.params), float(lr), float(betas[0]), float(betas[1]),
                                  float(1e-8 if eps is None else eps), float(0.0 if weight_decay is None else weight_decay))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:
            lib.ot_adam_free(h)

    def step(self): lib.ot_adam_step(self._h)
    def zero_grad(self): lib.ot_adam_zero_grad(self._h)
    def set_lr(self, lr): lib.ot_adam_set_lr(self._h, float(lr))
    def get_lr(self): return float(lib.ot_adam_get_lr(self._h))
    def t(self): return int(lib.ot_adam_t(self._h))
    def m(self, i): return np.ctypeslib.as_array(lib.ot_adam_m(self._h, i), shape=(self.params[i].numel(),)).copy()
    def v(self, i): return np.ctypeslib.as_array(lib.ot_adam_v(self._h, i), shape=(self.params[i].numel(),)).copy()


class SGD:
    def __init__(self, params, lr, momentum=None):
        self.params, self.lr = list(params), float(lr)

    def step(self):
        arr = (_p * len(self.params))(*[p._h for p in self.params])
        lib.ot_sgd_step(arr, len(self.params), self.lr)

    def zero_grad(self):
        for p in self.params:
            p.zero_grad()


# -- src/nn.rs -----------------------------------------------------------
KIND = dict(linear=0, relu=1, sigmoid=2, conv2d_relu=3, conv2d=4, maxpool=5, avgpool=6, adaptive_avgpool=7, flatten=8)


class Sequential:
    """nn.rs:130-162.  `spec` is a list of dicts, e.g.
    {"kind": "linear", "w": Tensor[out,in], "b": Tensor[out]} /
    {"kind": "conv2d_relu", "w": .., "b": .., "stride": (1,1), "padding": (1,1)} /
    {"kind": "maxpool", "kernel": (2,2), "stride": (2,2)} / {"kind": "adaptive_avgpool", "out": (1,1)} /
    {"kind": "flatten", "start_dim": 1} / {"kind": "relu"}"""

    def __init__(self, spec, conv_mode=0):
        self.spec = spec
        self._layers = (_LayerStruct * len(spec))()
        for i, s in enumerate(spec):
            L = self._layers[i]
            L.kind = KIND[s["kind"]]
            L.w = s["w"]._h if "w" in s else None
            L.b = s["b"]._h if s.get("b") is not None else None
            k = s.get("kernel", (0, 0))
            L.k_h, L.k_w = k
            st = s.get("stride", (1, 1) if s["kind"].startswith("conv") else None) or (0, 0)
            L.s_h, L.s_w = st
            L.p_h, L.p_w = s.get("padding", (0, 0))
            L.out_h, L.out_w = s.get("out", (1, 1))
            L.start_dim = s.get("start_dim", 1)
        self._m = _ModelStruct(self._layers, len(spec), conv_mode)

    def forward(self, x):
        return Tensor(_h=lib.ot_model_forward(C.byref(self._m), x._h))

    def parameters(self):
        out = []
        for s in self.spec:
            if "w" in s:
                out.append(s["w"])
                if s.get("b") is not None:
                    out.append(s["b"])
        return out

    def run_steps(self, opt, images, labels, x_shape, steps):
        """cpu_baseline leg: `steps` x {get_batch -> training step} in ONE C call over a resident dataset; -> last loss"""
        images = np.ascontiguousarray(images, dtype=np.float32).reshape(-1, 784)
        labels = np.ascontiguousarray(labels, dtype=np.float32).reshape(-1)
        loss = C.c_float()
        lib.ot_baseline_run_steps(C.byref(self._m), opt._h, _fp(images.reshape(-1)), _fp(labels), labels.size, _shape_arr(x_shape),
                                  len(x_shape), int(steps), C.cast(C.byref(loss), _f32p))
        return loss.value

    def train_step(self, opt, images, labels, x_shape, want_logits=False, want_grads=False):
        """examples/train_mnist.rs:89-121 in one C call; returns dict."""
        images = np.ascontiguousarray(images, dtype=np.float32).reshape(-1)
        labels = np.ascontiguousarray(labels, dtype=np.float32).reshape(-1)
        loss, acc = C.c_float(), C.c_float()
        params = self.parameters()
        n_cls = None
        logits = grads = has = None
        if want_logits:
            n_cls = params[-1].shape()[0]
            logits = np.empty((x_shape[0], n_cls), dtype=np.float32)
        if want_grads:
            grads = np.empty(sum(p.numel() for p in params), dtype=np.float32)
            has = (C.c_int * len(params))()
        lib.ot_train_step(C.byref(self._m), opt._h if opt is not None else None, _fp(images), _fp(labels),
                          _shape_arr(x_shape), len(x_shape), C.byref(loss), C.byref(acc),
                          _fp(logits.reshape(-1)) if logits is not None else None,
                          _fp(grads) if grads is not None else None, has)
        return dict(loss=loss.value, acc=acc.value, logits=logits, grads=grads,
                    has_grad=[bool(h) for h in has] if has is not None else None)

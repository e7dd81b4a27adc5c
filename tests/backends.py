"""Uniform face over the two implementations the parity tests compare:
`oracle` (CPU restatement of the reference, test infrastructure) and `hip`
(taper_amd: the MI355X product path through the C ABI)."""
from __future__ import annotations

import numpy as np


class OracleBackend:
    name = "oracle"

    def __init__(self):
        from oracle import oracle as O
        self.m = O
        self.Tensor, self.Tape = O.Tensor, O.Tape
        self.cross_entropy_loss, self.accuracy = O.cross_entropy_loss, O.accuracy
        self.softmax, self.log_softmax, self.one_hot, self.mse_loss = O.softmax, O.log_softmax, O.one_hot, O.mse_loss
        self.Adam, self.SGD = O.Adam, O.SGD

    def set_zero_sentinel(self, on):
        self.m.Tape.set_zero_sentinel(on)

    def sequential(self, spec, full_backward=False):
        """spec: list of dicts with numpy weights (see build_model)."""
        O = self.m
        layers = []
        for s in spec:
            d = dict(s)
            if "w" in d:
                d["w"] = O.Tensor(d["w"]).requires_grad()
                d["b"] = O.Tensor(d["b"]).requires_grad() if d.get("b") is not None else None
            layers.append(d)
        return O.Sequential(layers, conv_mode=1 if full_backward else 0)

    def forward_backward(self, model, x, y, x_shape):
        """returns loss, acc, logits, grads(list or None per param)"""
        O = self.m
        O.Tape.reset()
        xt, yt = O.Tensor(x, x_shape), O.Tensor(y)
        logits = model.forward(xt)
        loss = O.cross_entropy_loss(logits, yt)
        acc = O.accuracy(logits, yt)
        loss.backward()
        grads = [p.grad() for p in model.parameters()]
        out = float(loss.data()[0]), acc, logits.data(), grads
        for p in model.parameters():
            p.zero_grad()
        O.Tape.reset()
        return out


class HipBackend:
    name = "hip"

    def __init__(self):
        import taper_amd as T
        self.m = T
        self.Tensor, self.Tape = T.Tensor, T.Tape
        self.cross_entropy_loss, self.accuracy = T.cross_entropy_loss, T.accuracy
        self.softmax, self.log_softmax, self.one_hot, self.mse_loss = T.softmax, T.log_softmax, T.one_hot, T.mse_loss
        self.Adam, self.SGD = T.Adam, T.SGD

    def set_zero_sentinel(self, on):
        self.m.Tape.set_compat_zero_sentinel(on)

    def sequential(self, spec, full_backward=False, fuse=True):
        T = self.m
        T.set_full_backward(full_backward)
        layers = []
        for s in spec:
            k = s["kind"]
            if k == "linear":
                o, i = s["w"].shape
                l = T.Linear(i, o, s.get("b") is not None)
            elif k in ("conv2d_relu", "conv2d"):
                co, ci, kh, kw = s["w"].shape
                cls = T.Conv2dReLU if k == "conv2d_relu" else T.Conv2d
                l = cls(ci, co, (kh, kw), s.get("stride", (1, 1)), s.get("padding", (0, 0)), None, None, s.get("b") is not None)
            elif k == "relu":
                l = T.ReLU()
            elif k == "sigmoid":
                l = T.Sigmoid()
            elif k == "maxpool":
                l = T.MaxPool2d(s["kernel"], s.get("stride"), s.get("padding"))
            elif k == "avgpool":
                l = T.AvgPool2d(s["kernel"], s.get("stride"), s.get("padding"))
            elif k == "adaptive_avgpool":
                l = T.AdaptiveAvgPool2d(s.get("out", (1, 1)))
            elif k == "flatten":
                l = T.Flatten(s.get("start_dim", 1))
            else:
                raise ValueError(k)
            if "w" in s:
                ps = l.parameters()
                ps[0].set_data(s["w"])
                if s.get("b") is not None:
                    ps[1].set_data(s["b"])
            layers.append(l)
        return T.Sequential(layers, fuse=fuse)

    def forward_backward(self, model, x, y, x_shape):
        T = self.m
        T.Tape.reset()
        xt, yt = T.Tensor(x, x_shape), T.Tensor(y)
        logits = model.forward(xt)
        loss = T.cross_entropy_loss(logits, yt)
        acc = T.accuracy(logits, yt)
        loss.backward()
        grads = [p.grad() for p in model.parameters()]
        out = float(loss.data()[0]), acc, logits.data(), grads
        for p in model.parameters():
            p.zero_grad()
        T.Tape.reset()
        return out


def get(name):
    return OracleBackend() if name == "oracle" else HipBackend()


# ---- model zoo (numpy weights with the reference's init distributions) --------
def _lin(rng, i, o):
    s = np.sqrt(2.0 / i)  # nn.rs:36-37
    return dict(kind="linear", w=rng.uniform(-s, s, (o, i)).astype(np.float32), b=np.zeros(o, np.float32))


def _conv(rng, ci, co, relu=True):
    bound = np.sqrt(2.0 / (ci * 9)) * np.sqrt(3.0)  # nn.rs:219-222
    return dict(kind="conv2d_relu" if relu else "conv2d", w=rng.uniform(-bound, bound, (co, ci, 3, 3)).astype(np.float32),
                b=np.zeros(co, np.float32), stride=(1, 1), padding=(1, 1))


def mlp_baseline(rng):  # BASELINE.json configs[0/1]: 784-128-10 (src/train.rs:390-394)
    return [_lin(rng, 784, 128), dict(kind="relu"), _lin(rng, 128, 10)]


def mlp_64(rng):  # 784-64-10: the tail launch is 108 workgroups -- four ranks' worth of them fit ONE device together (tests/test_gpu_dp.py)
    return [_lin(rng, 784, 64), dict(kind="relu"), _lin(rng, 64, 10)]


def mlp_100(rng):  # a hidden width that is no multiple of 32 (the fused large-batch step pads its tiles: r05)
    return [_lin(rng, 784, 100), dict(kind="relu"), _lin(rng, 100, 10)]


def mlp_100_52(rng):  # two hidden layers, both ragged
    return [_lin(rng, 784, 100), dict(kind="relu"), _lin(rng, 100, 52), dict(kind="relu"), _lin(rng, 52, 10)]


def mlp_example(rng):  # examples/train_mnist.rs:34-40: 784-128-64-10
    return [_lin(rng, 784, 128), dict(kind="relu"), _lin(rng, 128, 64), dict(kind="relu"), _lin(rng, 64, 10)]


def cnn_reference(rng):  # examples/train_mnist_cnn.rs:35-100
    return [_conv(rng, 1, 32), _conv(rng, 32, 32), dict(kind="maxpool", kernel=(2, 2), stride=(2, 2)),
            _conv(rng, 32, 64), _conv(rng, 64, 64), dict(kind="maxpool", kernel=(2, 2), stride=(2, 2)),
            _conv(rng, 64, 128), dict(kind="adaptive_avgpool", out=(1, 1)), dict(kind="flatten", start_dim=1),
            _lin(rng, 128, 128), dict(kind="relu"), _lin(rng, 128, 64), dict(kind="relu"), _lin(rng, 64, 10)]


def cnn_simple(rng):  # BASELINE.json configs[2]: Conv3x3->ReLU->MaxPool x2 -> Linear
    return [_conv(rng, 1, 32), dict(kind="maxpool", kernel=(2, 2), stride=(2, 2)),
            _conv(rng, 32, 64), dict(kind="maxpool", kernel=(2, 2), stride=(2, 2)),
            dict(kind="flatten", start_dim=1), _lin(rng, 3136, 10)]


def nonzero_biases(spec, rng):
    """the reference initialises biases to 0; parity tests perturb them so bias paths are exercised"""
    for s in spec:
        if s.get("b") is not None:
            s["b"] = rng.uniform(-0.1, 0.1, s["b"].shape).astype(np.float32)
    return spec


def mnist_like(rng, n):
    x = rng.integers(0, 256, (n, 784)).astype(np.float32) / np.float32(255.0)  # data/mnist.rs:226
    y = rng.integers(0, 10, n).astype(np.float32)
    return x, y

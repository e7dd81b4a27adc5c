"""th_mlp3_xent -- a three-layer classifier (Linear + ReLU, Linear + ReLU, Linear, softmax cross-entropy) forward and backward in two
launches -- through the C ABI against the oracle's tape (`matmul` / `transpose` / `add_broadcast` / `relu` / `cross_entropy_loss` /
`accuracy`: /root/reference/src/nn.rs:54-60, ops.rs:200-298,312-374, tensor.rs:544-704, loss.rs:101-195,271-290), with and without the Adam
updates in the gradient epilogues (optim.rs:83-113)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.fixture(scope="module")
def ctx():
    from taper_amd import hip
    c = hip.Ctx(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _net(rng, in_f, h1, h2, c):
    dims = [(h1, in_f), (h2, h1), (c, h2)]
    return [(rng.uniform(-1, 1, d).astype(np.float32) * np.float32(np.sqrt(2.0 / d[1])), rng.uniform(-0.2, 0.2, d[0]).astype(np.float32)) for d in dims]


def _oracle(O, x, y, net, need_dx):
    O.Tape.reset()
    xt = O.Tensor(x)
    if need_dx:
        xt = xt.requires_grad()
    params = [(O.Tensor(w).requires_grad(), O.Tensor(b).requires_grad()) for w, b in net]
    h = xt
    for i, (w, b) in enumerate(params):
        h = h.matmul(w.transpose()).add_broadcast(b)
        if i < 2:
            h = h.relu()
    loss = O.cross_entropy_loss(h, O.Tensor(y))
    acc = O.accuracy(h, O.Tensor(y))
    loss.backward()
    return float(loss.data()[0]), acc, [(w.grad(), b.grad()) for w, b in params], (xt.grad() if need_dx else None)


def _run(ctx, x, y, net, need_dx, fuse=None, tick=None, metrics=False):
    from taper_amd import hip
    B, in_f = x.shape
    dx_, dy_ = ctx.upload(x), ctx.upload(y)
    bufs, layers = [], (hip.Mlp3Layer * 3)()
    for l, (w, b) in enumerate(net):
        dw, db = ctx.upload(w), ctx.upload(b)
        gw, gb = ctx.empty(w.size), ctx.empty(b.size)
        bufs.append((dw, db, gw, gb))
        wf = C.cast(C.pointer(fuse[l][0]), C.c_void_p) if fuse else None
        bf = C.cast(C.pointer(fuse[l][1]), C.c_void_p) if fuse else None
        layers[l] = hip.Mlp3Layer(int(dw), int(db), int(gw), int(gb), wf, bf, w.shape[0])
    gx = ctx.empty(x.size) if need_dx else None
    loss, nc = ctx.empty(1), ctx.empty(1)
    met = ctx.zeros(8) if metrics else None
    st = ctx.upload(np.zeros(2, np.int64)) if metrics else None
    ctx.call("th_mlp3_xent", dx_, dy_, B, in_f, C.cast(layers, C.c_void_p), gx, loss, nc, met, 4 if metrics else 0, st, B if metrics else 0, tick, None)
    ctx.sync()
    out = dict(loss=float(ctx.download(loss, (1,))[0]), nc=float(ctx.download(nc, (1,))[0]),
               grads=[(ctx.download(gw, w.shape), ctx.download(gb, b.shape)) for (w, b), (_, _, gw, gb) in zip(net, bufs)],
               dx=ctx.download(gx, x.shape) if need_dx else None, bufs=bufs)
    if metrics:
        out["metrics"] = ctx.download(met, (8,))
        out["state"] = ctx.download(st, (2,), np.int64)
    return out


CASES = [  # batch, in, h1, h2, classes, need_dx
    (256, 128, 128, 64, 10, True),      # the reference CNN's classifier at its batch (examples/train_mnist_cnn.rs:53-61)
    (256, 784, 128, 64, 10, False),     # examples/train_mnist.rs:40-48
    (16, 16, 16, 16, 1, True),
    (48, 32, 48, 16, 16, True),
    (96, 128, 128, 64, 10, True),
    (32, 1024, 256, 256, 7, False),
]


@pytest.mark.parametrize("B,in_f,h1,h2,c,need_dx", CASES)
def test_mlp3_matches_the_oracle_tape(ctx, O, B, in_f, h1, h2, c, need_dx):
    from taper_amd import hip
    assert hip.hip.th_mlp3_supported(B, in_f, h1, h2, c) == 1
    rng = np.random.default_rng(B * 7 + in_f + c)
    net = _net(rng, in_f, h1, h2, c)
    x = rng.uniform(-1, 1, (B, in_f)).astype(np.float32)
    y = rng.integers(0, c, B).astype(np.float32)
    ref_loss, ref_acc, ref_grads, ref_dx = _oracle(O, x, y, net, need_dx)
    got = _run(ctx, x, y, net, need_dx, metrics=True)
    assert abs(got["loss"] - ref_loss) <= RTOL * max(1.0, abs(ref_loss))
    assert abs(got["nc"] - round(ref_acc * B)) <= 1
    for l, ((gw, gb), (rw, rb)) in enumerate(zip(got["grads"], ref_grads)):
        np.testing.assert_allclose(gw, rw.reshape(gw.shape), rtol=RTOL, atol=2e-4 * float(np.abs(rw).max()) + 1e-9, err_msg=f"dW{l + 1}")
        np.testing.assert_allclose(gb, rb.reshape(gb.shape), rtol=RTOL, atol=2e-4 * float(np.abs(rb).max()) + 1e-9, err_msg=f"db{l + 1}")
    if need_dx:
        np.testing.assert_allclose(got["dx"], ref_dx.reshape(x.shape), rtol=RTOL, atol=2e-4 * float(np.abs(ref_dx).max()) + 1e-9)
    # the step log: slot 0 = {loss, hits}, the step count and the cursor advanced
    assert got["metrics"][0] == np.float32(got["loss"]) and got["metrics"][1] == got["nc"]
    assert list(got["state"]) == [1, B]


def test_mlp3_fused_adam_and_tick(ctx, O):
    """Adam in the gradient epilogues (optim.rs:83-113) of all six parameters, the step counter opened by the first launch: parameters
    after the call = the oracle's Adam step on the oracle's gradients"""
    from taper_amd import hip
    B, in_f, h1, h2, c = 64, 128, 128, 64, 10
    rng = np.random.default_rng(3)
    net = _net(rng, in_f, h1, h2, c)
    x = rng.uniform(-1, 1, (B, in_f)).astype(np.float32)
    y = rng.integers(0, c, B).astype(np.float32)
    _, _, ref_grads, _ = _oracle(O, x, y, net, False)
    lr, b1, b2, eps, wd = 1e-2, 0.9, 0.999, 1e-8, 1e-4
    t = ctx.upload(np.zeros(1, np.int32))
    dlr = ctx.upload(np.array([lr], np.float32))
    state, fuse = [], []
    # parameters live in the buffers _run uploads: the fuse structs must point at them, so upload here and hand the same buffers over
    dxb, dyb = ctx.upload(x), ctx.upload(y)
    layers = (hip.Mlp3Layer * 3)()
    keep = []
    for l, (w, b) in enumerate(net):
        ent = []
        for arr in (w, b):
            p, m, v, g = ctx.upload(arr), ctx.zeros(arr.size), ctx.zeros(arr.size), ctx.empty(arr.size)
            f = hip.AdamFuse(int(p), int(m), int(v), int(t), int(dlr), b1, b2, eps, wd)
            ent.append((p, m, v, g, f))
        keep.append(ent)
        (pw, _, _, gw, fw), (pb, _, _, gb, fb) = ent
        layers[l] = hip.Mlp3Layer(int(pw), int(pb), int(gw), int(gb), C.cast(C.pointer(fw), C.c_void_p), C.cast(C.pointer(fb), C.c_void_p), w.shape[0])
    loss, nc = ctx.empty(1), ctx.empty(1)
    ctx.call("th_mlp3_xent", dxb, dyb, B, in_f, C.cast(layers, C.c_void_p), None, loss, nc, None, 0, None, 0, t, None)
    ctx.sync()
    assert int(ctx.download(t, (1,), np.int32)[0]) == 1
    step = lr * np.sqrt(1 - b2) / (1 - b1)                      # optim.rs:87-90 at t = 1
    for l, ((w, b), (rw, rb)) in enumerate(zip(net, ref_grads)):
        for arr, rg, ent in ((w, rw, keep[l][0]), (b, rb, keep[l][1])):
            g = rg.reshape(arr.shape).astype(np.float64) + wd * arr
            m = (1 - b1) * g
            v = (1 - b2) * g * g
            want = arr - step * m / (np.sqrt(v) + eps)
            np.testing.assert_allclose(ctx.download(ent[0], arr.shape), want, rtol=1e-4, atol=lr * 2e-2, err_msg=f"layer {l + 1}")


def test_mlp3_unsupported_shapes():
    from taper_amd import hip
    f = hip.hip.th_mlp3_supported
    assert f(256, 128, 128, 64, 10) == 1
    assert f(250, 128, 128, 64, 10) == 0        # batch not a multiple of 16
    assert f(256, 100, 128, 64, 10) == 0
    assert f(256, 128, 120, 64, 10) == 0
    assert f(256, 128, 128, 64, 17) == 0
    assert f(256, 2048, 128, 64, 10) == 0


def test_mlp3_finishes_the_bias_of_the_conv_in_front(ctx):
    """gap: the input rows are the plane means of a bias-only Conv2dReLU + global average pool; the gradient launch also forms that conv's
    bias gradient db[ch] = sum_n (dX[n][ch] / hw) * cnt[n][ch] (tensor.rs:1626-1628 + ops.rs:358-369 through the counts, the formula of
    th_bias_grad_counts_adam) from the dX the first launch wrote"""
    from taper_amd import hip
    B, in_f, h1, h2, c, hw = 256, 128, 128, 64, 10, 49
    rng = np.random.default_rng(12)
    net = _net(rng, in_f, h1, h2, c)
    x = rng.uniform(0, 1, (B, in_f)).astype(np.float32)
    y = rng.integers(0, c, B).astype(np.float32)
    cnt = rng.integers(0, hw + 1, (B, in_f)).astype(np.float32)
    dx_, dy_, dcnt = ctx.upload(x), ctx.upload(y), ctx.upload(cnt)
    layers, keep = (hip.Mlp3Layer * 3)(), []
    for l, (w, b) in enumerate(net):
        bufs = (ctx.upload(w), ctx.upload(b), ctx.empty(w.size), ctx.empty(b.size))
        keep.append(bufs)
        layers[l] = hip.Mlp3Layer(int(bufs[0]), int(bufs[1]), int(bufs[2]), int(bufs[3]), None, None, w.shape[0])
    gx, gb, loss, nc = ctx.empty(x.size), ctx.empty(in_f), ctx.empty(1), ctx.empty(1)
    gap = hip.Mlp3Gap(int(dcnt), int(gb), hw, None)
    ctx.call("th_mlp3_xent", dx_, dy_, B, in_f, C.cast(layers, C.c_void_p), gx, loss, nc, None, 0, None, 0, None, C.cast(C.pointer(gap), C.c_void_p))
    ctx.sync()
    dx = ctx.download(gx, x.shape).astype(np.float64)
    want = (dx / hw * cnt).sum(axis=0)
    np.testing.assert_allclose(ctx.download(gb, (in_f,)), want, rtol=1e-4, atol=1e-6 * float(np.abs(want).max()) + 1e-9)

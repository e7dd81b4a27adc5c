"""th_mlp_tail (classifier head + the hidden layer's whole backward + its Adam update in ONE launch) and
th_linear_fwd_ex (layer forward that also carries the previous step's deferred updates and opens the
next step) against the oracle's unfused chain."""
import ctypes as C

import numpy as np
import pytest

from tests import margins

from tests import backends
from tests.test_gpu_fused import _adam_ref, close, RTOL
from taper_amd.hip import AdamFuse, AdamSlice

pytestmark = pytest.mark.gpu
BOUND_M = 1.9e-6   # observed 9.5e-7 (2x the r04 observation, profiles/r04_parity_margins.json)


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


def oracle_tail(O, x, w1, b1, w2, b2, y):
    """reference chain: Linear + ReLU + Linear + cross_entropy_loss, backward from the loss"""
    O.Tape.reset()
    O.Tape.set_zero_sentinel(True)
    w1t, b1t = O.Tensor(w1).requires_grad(), O.Tensor(b1).requires_grad()
    w2t, b2t = O.Tensor(w2).requires_grad(), O.Tensor(b2).requires_grad()
    xt = O.Tensor(x).requires_grad()
    h = xt.matmul(w1t.transpose()).add_broadcast(b1t).relu()
    logits = h.matmul(w2t.transpose()).add_broadcast(b2t)
    yt = O.Tensor(y)
    loss = O.cross_entropy_loss(logits, yt)
    acc = O.accuracy(logits, yt)
    loss.backward()
    out = dict(h=h.data(), loss=float(loss.data()[0]), ncorrect=round(acc * len(y)), dw1=w1t.grad(), db1=b1t.grad(),
               dw2=w2t.grad(), db2=b2t.grad(), dx=xt.grad())
    O.Tape.reset()
    return out


SHAPES = [(64, 784, 128, 10), (32, 784, 128, 10), (128, 784, 128, 10), (1, 5, 4, 2), (70, 37, 20, 5), (256, 100, 256, 16),
          (200, 50, 64, 10), (17, 784, 36, 3),
          (256, 784, 128, 10), (192, 64, 64, 10), (48, 784, 128, 10), (16, 16, 256, 16), (240, 48, 256, 3), (64, 64, 32, 10), (512, 784, 128, 10), (400, 48, 64, 5), (300, 33, 20, 4)]   # whole tiles, several / partial chunks


@pytest.mark.parametrize("batch,inf,hid,c", SHAPES)
@pytest.mark.parametrize("fuse", [False, True], ids=["grads", "adam"])
def test_mlp_tail(ctx, O, batch, inf, hid, c, fuse):
    rng = np.random.default_rng(batch * 13 + inf + hid + c)
    x = rng.uniform(0, 1, (batch, inf)).astype(np.float32)
    w1 = rng.uniform(-1, 1, (hid, inf)).astype(np.float32) * np.float32(np.sqrt(2.0 / inf))
    b1 = rng.uniform(-0.1, 0.1, hid).astype(np.float32)
    w2 = rng.uniform(-0.3, 0.3, (c, hid)).astype(np.float32)
    b2 = rng.uniform(-0.1, 0.1, c).astype(np.float32)
    y = rng.integers(0, c, batch).astype(np.float32)
    ref = oracle_tail(O, x, w1, b1, w2, b2, y)
    h = ref["h"].reshape(batch, hid).astype(np.float32)   # the forward launch's output is this kernel's input
    lr, t = 1e-3, 4
    dw1, db1, dw2, db2 = ctx.empty(w1.size), ctx.empty(hid), ctx.empty(w2.size), ctx.empty(c)
    loss, nc = ctx.empty(1), ctx.empty(1)
    state, metrics = ctx.upload(np.array([5, 640], np.int64)), ctx.zeros(2 * 16)
    pw, pb = ctx.upload(w1), ctx.upload(b1)
    mw, vw, mb, vb = ctx.zeros(w1.size), ctx.zeros(w1.size), ctx.zeros(hid), ctx.zeros(hid)
    tick, dlr = ctx.upload(np.array([t, 0], np.int32)), ctx.upload(np.array([lr], np.float32))
    wf = AdamFuse(int(pw), int(mw), int(vw), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4)
    bf = AdamFuse(int(pb), int(mb), int(vb), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4)
    dw2_in = ctx.upload(w2)
    ctx.call("th_mlp_tail", ctx.upload(x), ctx.upload(h), dw2_in, ctx.upload(b2), ctx.upload(y), batch, inf, hid, c, loss, nc,
             dw1, db1, dw2, db2, None, None, metrics, 16, state, batch, C.byref(wf) if fuse else None, C.byref(bf) if fuse else None)
    assert ctx.download(loss, 1)[0] == pytest.approx(ref["loss"], rel=RTOL, abs=1e-6)
    assert ctx.download(nc, 1)[0] == ref["ncorrect"]
    close(ctx.download(dw1, w1.shape), ref["dw1"], atol=1e-6)
    close(ctx.download(db1, hid), ref["db1"], atol=1e-6)
    close(ctx.download(dw2, w2.shape), ref["dw2"], atol=1e-6)
    close(ctx.download(db2, c), ref["db2"], atol=1e-6)
    np.testing.assert_array_equal(ctx.download(state, 2, np.int64), [6, 640 + batch])
    assert ctx.download(metrics, (16, 2))[5, 0] == ctx.download(loss, 1)[0]
    assert ctx.download(metrics, (16, 2))[5, 1] == ref["ncorrect"]
    assert ctx.download(tick, 2, np.int32)[0] == t                       # never ticks
    np.testing.assert_array_equal(ctx.download(dw2_in, w2.shape), w2)    # W2 is only read
    if fuse:
        w_ref, wm_ref, wv_ref = _adam_ref(O, w1, np.asarray(ref["dw1"], np.float32).reshape(w1.shape), lr, t)
        b_ref, _, _ = _adam_ref(O, b1, np.asarray(ref["db1"], np.float32), lr, t)
        np.testing.assert_allclose(ctx.download(pw, w1.shape), w_ref, rtol=RTOL, atol=lr * 2e-2)
        np.testing.assert_allclose(ctx.download(pb, hid), b_ref, rtol=RTOL, atol=lr * 2e-2)
        margins.check("w1_m", ctx.download(mw, w1.shape), wm_ref.reshape(w1.shape), BOUND_M)
    else:
        np.testing.assert_array_equal(ctx.download(pw, w1.shape), w1)


@pytest.mark.parametrize("batch,inf,hid,c", [(64, 128, 64, 10), (256, 128, 64, 10), (48, 784, 128, 10), (32, 48, 256, 16), (16, 16, 64, 2),
                                             (256, 64, 32, 10), (16, 32, 32, 3)])   # hidden 32: the reference CNN's classifier
def test_mlp_tail_dx(ctx, O, batch, inf, hid, c):
    """a hidden layer that is not the first: the same launch also hands dX = dZ1 . W1 down (whole tiles only)"""
    rng = np.random.default_rng(batch * 17 + inf + hid + c)
    x = np.maximum(rng.standard_normal((batch, inf)), 0).astype(np.float32)     # a ReLU output
    w1 = rng.uniform(-1, 1, (hid, inf)).astype(np.float32) * np.float32(np.sqrt(2.0 / inf))
    b1 = rng.uniform(-0.1, 0.1, hid).astype(np.float32)
    w2 = rng.uniform(-0.3, 0.3, (c, hid)).astype(np.float32)
    b2 = rng.uniform(-0.1, 0.1, c).astype(np.float32)
    y = rng.integers(0, c, batch).astype(np.float32)
    ref = oracle_tail(O, x, w1, b1, w2, b2, y)
    h = ref["h"].reshape(batch, hid).astype(np.float32)
    dw1, db1, dw2, db2, dx = ctx.empty(w1.size), ctx.empty(hid), ctx.empty(w2.size), ctx.empty(c), ctx.empty(x.size)
    loss, nc = ctx.empty(1), ctx.empty(1)
    w1d = ctx.upload(w1)
    ctx.call("th_mlp_tail", ctx.upload(x), ctx.upload(h), ctx.upload(w2), ctx.upload(b2), ctx.upload(y), batch, inf, hid, c, loss, nc,
             dw1, db1, dw2, db2, w1d, dx, None, 0, None, 0, None, None)
    assert ctx.download(loss, 1)[0] == pytest.approx(ref["loss"], rel=RTOL, abs=1e-6)
    close(ctx.download(dx, x.shape), ref["dx"], atol=1e-6)
    close(ctx.download(dw1, w1.shape), ref["dw1"], atol=1e-6)
    close(ctx.download(db1, hid), ref["db1"], atol=1e-6)
    close(ctx.download(dw2, w2.shape), ref["dw2"], atol=1e-6)
    np.testing.assert_array_equal(ctx.download(w1d, w1.shape), w1)


def test_mlp_tail_dx_limits(ctx):
    from taper_amd._lib import TaperError
    from taper_amd.hip import AdamFuse
    z = ctx.zeros(256 * 300)
    with pytest.raises(TaperError, match="whole tiles"):
        ctx.call("th_mlp_tail", z, z, z, None, z, 20, 48, 64, 3, z, None, z, None, None, None, z, z, None, 0, None, 0, None, None)
    f = AdamFuse(int(z), int(z), int(z), int(z), int(z), 0.9, 0.999, 1e-8, 0.0)
    with pytest.raises(TaperError, match="must be deferred"):
        ctx.call("th_mlp_tail", z, z, z, None, z, 16, 48, 64, 3, z, None, z, None, None, None, z, z, None, 0, None, 0, C.byref(f), None)


def test_mlp_tail_optional_outputs_and_limits(ctx):
    from taper_amd._lib import TaperError
    rng = np.random.default_rng(3)
    x, h = rng.uniform(0, 1, (8, 20)).astype(np.float32), rng.uniform(0, 1, (8, 16)).astype(np.float32)
    w2, y = rng.uniform(-1, 1, (3, 16)).astype(np.float32), rng.integers(0, 3, 8).astype(np.float32)
    loss, dw1 = ctx.empty(1), ctx.empty(16 * 20)
    ctx.call("th_mlp_tail", ctx.upload(x), ctx.upload(h), ctx.upload(w2), None, ctx.upload(y), 8, 20, 16, 3, loss, None, dw1, None,
             None, None, None, None, None, 0, None, 0, None, None)
    assert np.isfinite(ctx.download(loss, 1)[0])
    z = ctx.zeros(1024 * 600)
    for shape, pat in [((600, 8, 16, 3), "batch <= 512"), ((8, 8, 300, 3), "hidden <= 256"), ((8, 8, 18, 3), "multiple of 4"),
                       ((8, 8, 16, 17), "classes <= 16")]:
        with pytest.raises(TaperError, match=pat):
            ctx.call("th_mlp_tail", z, z, z, None, z, *shape, z, None, z, None, None, None, None, None, None, 0, None, 0, None, None)


@pytest.mark.parametrize("batch,inf,outf", [(64, 784, 128), (32, 784, 128), (7, 20, 9), (64, 3136, 10), (512, 512, 512)])
@pytest.mark.parametrize("sizes", [(), (1280, 10), (5000, 3)])
def test_linear_fwd_ex(ctx, O, batch, inf, outf, sizes):
    """Y = relu(X W^T + b); carried slices are updated with the counter as it stood, then the counter advances"""
    rng = np.random.default_rng(batch + inf + outf + len(sizes))
    x = rng.uniform(0, 1, (batch, inf)).astype(np.float32)
    w = rng.uniform(-0.1, 0.1, (outf, inf)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, outf).astype(np.float32)
    lr, t = 1e-3, 5
    tick, dlr = ctx.upload(np.array([t, 0], np.int32)), ctx.upload(np.array([lr], np.float32))
    slices, refs, keep = (AdamSlice * max(len(sizes), 1))(), [], []
    for i, n in enumerate(sizes):
        p0 = rng.uniform(-0.1, 0.1, n).astype(np.float32)
        g = (rng.standard_normal(n) * 0.01).astype(np.float32)
        refs.append(_adam_ref(O, p0, g, lr, t))
        bufs = [ctx.upload(p0), ctx.zeros(n), ctx.zeros(n), ctx.upload(g)]
        keep.append(bufs)
        slices[i] = AdamSlice(int(bufs[3]), n, AdamFuse(int(bufs[0]), int(bufs[1]), int(bufs[2]), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4))
    yb = ctx.empty(batch * outf)
    ctx.call("th_linear_fwd_ex", ctx.upload(x), ctx.upload(w), ctx.upload(b), yb, batch, inf, outf, 1, slices if sizes else None,
             len(sizes), tick)
    close(ctx.download(yb, (batch, outf)), np.maximum(x.astype(np.float64) @ w.T.astype(np.float64) + b, 0), atol=1e-5)
    for bufs, (p_ref, m_ref, v_ref), n in zip(keep, refs, sizes):
        np.testing.assert_allclose(ctx.download(bufs[0], (n,)), p_ref, rtol=RTOL, atol=lr * 2e-2)
        margins.check("carried_m", ctx.download(bufs[1], (n,)), m_ref.reshape(n), BOUND_M)
    assert ctx.download(tick, 2, np.int32)[0] == t + 1
    # no tick requested
    ctx.call("th_linear_fwd_ex", ctx.upload(x), ctx.upload(w), None, yb, batch, inf, outf, 0, None, 0, None)
    close(ctx.download(yb, (batch, outf)), x.astype(np.float64) @ w.T.astype(np.float64), atol=1e-5)
    assert ctx.download(tick, 2, np.int32)[0] == t + 1

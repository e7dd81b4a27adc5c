"""th_conv_chain_mlp3_xent -- the reference CNN's front (examples/train_mnist_cnn.rs:35-100) with its three-layer classifier's ROW in the chain
launch's last epilogue, then th_mlp3_xent's gradient launch -- against the two calls it replaces (th_conv_chain_fwd, then th_mlp3_xent on the
plane means: each held to the oracle by tests/test_gpu_chain.py and tests/test_gpu_mlp3.py): the chain's outputs bit for bit (same conv code),
the classifier's within fp32 reordering (a row's products run on the vector ALU in another summation order than the MFMA tiles' --
/root/reference/src/nn.rs:54-60, src/loss.rs:101-195, 271-290, src/ops.rs:238-294, 358-369, src/optim.rs:83-113)."""
import ctypes as C

import numpy as np
import pytest

from tests import margins
from tests.test_gpu_chain import REFERENCE, _images, _params
from tests.test_gpu_fused import _adam_ref

pytestmark = pytest.mark.gpu


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


def _classifier(rng, classes):
    dims = [(128, 128), (64, 128), (classes, 64)]
    return [((rng.uniform(-1, 1, d) * np.sqrt(2.0 / d[1])).astype(np.float32), rng.uniform(-0.1, 0.1, d[0]).astype(np.float32)) for d in dims]


def _run(ctx, x, y, conv, net, fused, adam=None):
    """-> dict of downloaded outputs.  adam: (lr, t0) -> every classifier parameter and the last conv's bias carry a fused Adam update"""
    from taper_amd import hip
    from taper_amd.hip import AdamFuse
    n, classes = x.shape[0], net[2][0].shape[0]
    cbufs = [(ctx.upload(w), ctx.upload(b)) for w, b in conv]
    stages, ns = hip.conv_stages([(dw, db, c_out, post) for (dw, db), (_, c_out, post) in zip(cbufs, REFERENCE)])
    dx_, dy_ = ctx.upload(x), ctx.upload(y)
    means, cnt = ctx.empty(n * 128), ctx.empty(n * 128)
    layers, keep, fuses = (hip.Mlp3Layer * 3)(), [], []
    tick = dlr = None
    if adam:
        tick, dlr = ctx.upload(np.array([adam[1], 0], np.int32)), ctx.upload(np.array([adam[0]], np.float32))
    for l, (w, b) in enumerate(net):
        bufs = [ctx.upload(w), ctx.upload(b), ctx.empty(w.size), ctx.empty(b.size)]
        wf = bf = None
        if adam:
            mom = [ctx.zeros(w.size), ctx.zeros(w.size), ctx.zeros(b.size), ctx.zeros(b.size)]
            wf = AdamFuse(int(bufs[0]), int(mom[0]), int(mom[1]), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4)
            bf = AdamFuse(int(bufs[1]), int(mom[2]), int(mom[3]), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4)
            fuses += [wf, bf]
            keep.append(mom)
        keep.append(bufs)
        layers[l] = hip.Mlp3Layer(int(bufs[0]), int(bufs[1]), int(bufs[2]), int(bufs[3]), C.cast(C.pointer(wf), C.c_void_p) if wf else None,
                                  C.cast(C.pointer(bf), C.c_void_p) if bf else None, w.shape[0])
    gx, gb, loss, nc = ctx.empty(n * 128), ctx.empty(128), ctx.empty(1), ctx.empty(1)
    gap = hip.Mlp3Gap(int(cnt), int(gb), 49, None)
    gp = C.cast(C.pointer(gap), C.c_void_p)
    lp = C.cast(layers, C.c_void_p)
    if fused:
        ctx.call("th_conv_chain_mlp3_xent", dx_, C.cast(stages, C.c_void_p), ns, means, cnt, n, 1, 28, 28, dy_, lp, gx, loss, nc, None, 0, None, 0, tick, gp)
    else:
        ctx.call("th_conv_chain_fwd", dx_, C.cast(stages, C.c_void_p), ns, means, cnt, n, 1, 28, 28)
        ctx.call("th_mlp3_xent", means, dy_, n, 128, lp, gx, loss, nc, None, 0, None, 0, tick, gp)
    ctx.sync()
    out = dict(means=ctx.download(means, (n, 128)), cnt=ctx.download(cnt, (n, 128)), dx=ctx.download(gx, (n, 128)), gb=ctx.download(gb, (128,)),
               loss=ctx.download(loss, 1)[0], nc=ctx.download(nc, 1)[0], t=None if tick is None else int(ctx.download(tick, 2, np.int32)[0]))
    out["grads"], out["params"] = [], []
    for l, (w, b) in enumerate(net):
        bufs = [k for k in keep if int(k[0]) == layers[l].d_w][0]
        out["grads"].append((ctx.download(bufs[2], w.shape), ctx.download(bufs[3], b.shape)))
        out["params"].append((ctx.download(bufs[0], w.shape), ctx.download(bufs[1], b.shape)))
    return out


@pytest.mark.parametrize("n,classes", [(256, 10), (96, 10), (512, 16), (112, 3)])
def test_chain_mlp3_equals_the_two_calls_it_replaces(ctx, n, classes):
    from taper_amd._lib import hip as lib
    rng = np.random.default_rng(n + classes)
    conv = _params(REFERENCE, 5 + n)
    net = _classifier(rng, classes)
    x = _images(n, 3 * n)
    y = rng.integers(0, classes, n).astype(np.float32)
    a = _run(ctx, x, y, conv, net, fused=False)
    b = _run(ctx, x, y, conv, net, fused=True)
    np.testing.assert_array_equal(a["means"], b["means"])          # the same conv code: bit for bit
    np.testing.assert_array_equal(a["cnt"], b["cnt"])
    assert b["nc"] == a["nc"] or abs(b["nc"] - a["nc"]) <= 1        # (an argmax between two logits within rounding)
    assert abs(a["loss"] - b["loss"]) <= 2e-6 * max(1.0, abs(a["loss"]))
    name = "test_chain_mlp3_equals_the_two_calls_it_replaces"
    margins.check("dx", b["dx"], a["dx"], 1e-4, test=name)
    margins.check("conv_bias_grad", b["gb"], a["gb"], 1e-4, test=name)
    for l, ((gw_a, gb_a), (gw_b, gb_b)) in enumerate(zip(a["grads"], b["grads"])):
        margins.check(f"dw{l + 1}", gw_b, gw_a, 1e-4, test=name)
        margins.check(f"db{l + 1}", gb_b, gb_a, 1e-4, test=name)
    # deterministic: fixed-order sums
    b2 = _run(ctx, x, y, conv, net, fused=True)
    for k in ("dx", "gb", "means"):
        np.testing.assert_array_equal(b[k], b2[k])
    assert b["loss"] == b2["loss"]


def test_chain_mlp3_fused_adam_and_tick(ctx, O):
    """the chain launch opens the optimizer step (t += 1, optim.rs:84), the gradient launch applies every classifier parameter's update
    (optim.rs:99-110) with that counter: against _adam_ref on the gradients the same call returns"""
    n, classes, lr, t0 = 256, 10, 1e-3, 4
    rng = np.random.default_rng(77)
    conv, net = _params(REFERENCE, 9), _classifier(rng, classes)
    x, y = _images(n, 21), rng.integers(0, classes, n).astype(np.float32)
    out = _run(ctx, x, y, conv, net, fused=True, adam=(lr, t0))
    assert out["t"] == t0 + 1
    for (w, b), (gw, gb), (pw, pb) in zip(net, out["grads"], out["params"]):
        for p0, g, p1 in ((w, gw, pw), (b, gb, pb)):
            want, _, _ = _adam_ref(O, p0, g, lr, t0 + 1)
            np.testing.assert_allclose(p1, np.asarray(want).reshape(p1.shape), rtol=1e-6, atol=1e-9)


def test_chain_mlp3_limits(ctx):
    from taper_amd import hip
    conv = _params(REFERENCE, 1)
    cbufs = [(ctx.upload(w), ctx.upload(b)) for w, b in conv]
    stages, ns = hip.conv_stages([(dw, db, c_out, post) for (dw, db), (_, c_out, post) in zip(cbufs, REFERENCE)])
    f = lambda n, h1, h2, c: hip.hip.th_conv_chain_mlp3_supported(1, 28, 28, C.cast(stages, C.c_void_p), ns, n, h1, h2, c)
    assert f(256, 128, 64, 10) == 1 and f(1024, 128, 64, 16) == 1
    assert f(250, 128, 64, 10) == 0          # a multiple of 16 rows (the gradient launch's tiles)
    assert f(256, 128, 32, 10) == 0 and f(256, 64, 64, 10) == 0 and f(256, 128, 64, 17) == 0
    stages2, ns2 = hip.conv_stages([(dw, db, c_out, post) for (dw, db), (_, c_out, post) in list(zip(cbufs, REFERENCE))[:4]])
    assert hip.hip.th_conv_chain_mlp3_supported(1, 28, 28, C.cast(stages2, C.c_void_p), ns2, 256, 128, 64, 10) == 0   # no plane means at its end


@pytest.mark.parametrize("n,classes", [(256, 10), (96, 10), (112, 3)])
def test_chain_mlp3_matches_the_oracle_tape(ctx, O, n, classes):
    """the ONE call against the ORACLE, not against other HIP calls: the oracle's conv chain on the images (tensor.rs:1221-1285, 1391-1470 and
    the global mean) gives the plane means, the oracle's tape on them (nn.rs:54-60, loss.rs:101-195, 271-290, the backward closures
    ops.rs:238-294, 358-369, tensor.rs:574-587, 674-694) gives loss, hit count, the six classifier gradients and d(loss)/d(means) --
    which th_conv_chain_mlp3_xent returns as d_dx and, summed over the images through the positive-count of each plane, as the last
    conv's bias gradient"""
    from tests.test_gpu_chain import _oracle_chain
    from tests.test_gpu_mlp3 import _oracle as oracle_classifier
    rng = np.random.default_rng(1000 + n + classes)
    conv, net = _params(REFERENCE, 31 + n), _classifier(rng, classes)
    x, y = _images(n, 7 * n), rng.integers(0, classes, n).astype(np.float32)
    got = _run(ctx, x, y, conv, net, fused=True)
    name = "test_chain_mlp3_matches_the_oracle_tape"
    # the front: plane means within the conv tolerance of the oracle chain
    ref_means, ref_cnt = _oracle_chain(O, x, REFERENCE, conv)
    ref_means = np.asarray(ref_means, np.float32).reshape(n, 128)
    np.testing.assert_allclose(got["means"], ref_means, rtol=1e-4, atol=1e-5)
    # the classifier on the ORACLE's means: everything the call returns about it
    ref_loss, ref_acc, ref_grads, ref_dx = oracle_classifier(O, ref_means, y, net, True)
    assert abs(got["loss"] - ref_loss) <= 1e-4 * max(1.0, abs(ref_loss)), (got["loss"], ref_loss)
    assert abs(got["nc"] - round(ref_acc * n)) <= 1
    for l, ((gw, gb), (rw, rb)) in enumerate(zip(got["grads"], ref_grads)):
        margins.check(f"dw{l + 1}_vs_oracle", gw, np.asarray(rw).reshape(gw.shape), 2e-6, test=name)   # observed <= 8.5e-7 (profiles/r06_parity_margins.json)
        margins.check(f"db{l + 1}_vs_oracle", gb, np.asarray(rb).reshape(gb.shape), 2e-6, test=name)
    ref_dx = np.asarray(ref_dx, np.float32).reshape(n, 128)
    margins.check("dx_vs_oracle", got["dx"], ref_dx, 2e-6, test=name)
    # the last conv's bias gradient (tensor.rs:1276-1283 through the mean's backward: every positive element of a plane receives
    # d(mean) / 49): sum over images of dx * count / 49, with the oracle's own counts
    ref_gb = (ref_dx.astype(np.float64) * ref_cnt.astype(np.float64) / 49.0).sum(axis=0)
    margins.check("conv_bias_grad_vs_oracle", got["gb"], ref_gb.astype(np.float32), 2e-6, test=name)

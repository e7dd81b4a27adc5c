"""th_mlp2_train_steps: many steps of the 784-128-10 style MLP (forward, cross-entropy, backward, Adam) in ONE
launch, parameters and Adam moments resident on chip -- against the oracle's step-by-step training loop."""
import ctypes as C

import numpy as np
import pytest

from tests import backends

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.fixture(scope="module")
def ctx():
    from taper_amd import hip
    c = hip.Ctx(0)
    yield c
    c.close()


def _oracle_run(spec, x, y, batch, order, n_steps, lr=1e-3, wd=1e-4):
    Orc = backends.get("oracle")
    Orc.set_zero_sentinel(True)
    om = Orc.sequential(spec)
    oopt = Orc.m.Adam(om.parameters(), lr, None, None, wd)
    losses, ncorrect = [], []
    for s in range(n_steps):
        idx = order[s * batch:(s + 1) * batch]
        r = om.train_step(oopt, x[idx], y[idx], (len(idx), x.shape[1]))
        losses.append(r["loss"])
        ncorrect.append(round(r["acc"] * len(idx)))
    return losses, ncorrect, [p.data() for p in om.parameters()], oopt


def _device_run(ctx, spec, x, y, batch, order, n_steps, first_pos=0, lr=1e-3, wd=1e-4, t0=0, status=None):
    w1, b1, w2, b2 = spec[0]["w"], spec[0]["b"], spec[2]["w"], spec[2]["b"]
    hid, inf = w1.shape
    cls = w2.shape[0]
    P = [ctx.upload(a.copy()) for a in (w1, b1, w2, b2)]
    M = [ctx.zeros(a.size) for a in (w1, b1, w2, b2)]
    V = [ctx.zeros(a.size) for a in (w1, b1, w2, b2)]
    tick, dlr = ctx.upload(np.array([t0, 0], np.int32)), ctx.upload(np.array([lr], np.float32))
    metrics, state = ctx.zeros(2 * (n_steps + 2)), ctx.upload(np.zeros(2, np.int64))
    st = ctx.upload(np.zeros(1, np.int32))
    ctx.call("th_mlp2_train_steps", ctx.upload(x), ctx.upload(y), ctx.upload(order.astype(np.int32)), len(order), first_pos, batch, n_steps,
             inf, hid, cls, P[0], P[1], P[2], P[3], M[0], V[0], M[1], V[1], M[2], V[2], M[3], V[3], tick, dlr, 0.9, 0.999, 1e-8, wd,
             metrics, n_steps + 2, state, st)
    assert ctx.download(st, 1, np.int32)[0] == 0, "in-launch exchange timed out"
    mt = ctx.download(metrics, (n_steps + 2, 2))[:n_steps]
    params = [ctx.download(p, a.shape) for p, a in zip(P, (w1, b1, w2, b2))]
    moments = [ctx.download(m, a.shape) for m, a in zip(M, (w1, b1, w2, b2))]
    return mt[:, 0], mt[:, 1], params, moments, ctx.download(tick, 2, np.int32)[0], ctx.download(state, 2, np.int64)


@pytest.mark.parametrize("batch,n,inf,hid,cls,steps", [(64, 640, 784, 128, 10, 10), (64, 1000, 784, 128, 10, 16), (32, 200, 64, 16, 3, 7),
                                                       (50, 500, 112, 48, 16, 10), (1, 9, 16, 32, 2, 9), (64, 300, 256, 256, 10, 5)])
def test_mlp2_train_steps_matches_oracle(ctx, batch, n, inf, hid, cls, steps):
    rng = np.random.default_rng(batch + n + hid)
    s1, s2 = np.sqrt(2.0 / inf), np.sqrt(2.0 / hid)
    spec = [dict(kind="linear", w=rng.uniform(-s1, s1, (hid, inf)).astype(np.float32), b=rng.uniform(-0.1, 0.1, hid).astype(np.float32)),
            dict(kind="relu"),
            dict(kind="linear", w=rng.uniform(-s2, s2, (cls, hid)).astype(np.float32), b=rng.uniform(-0.1, 0.1, cls).astype(np.float32))]
    x = (rng.integers(0, 256, (n, inf)).astype(np.float32) / np.float32(255.0))
    y = rng.integers(0, cls, n).astype(np.float32)
    order = rng.permutation(n)
    o_loss, o_nc, o_params, oopt = _oracle_run(spec, x, y, batch, order, steps)
    d_loss, d_nc, d_params, d_m, d_t, d_state = _device_run(ctx, spec, x, y, batch, order, steps)
    np.testing.assert_allclose(d_loss, o_loss, rtol=3e-4, atol=1e-5)
    assert np.abs(d_nc - np.array(o_nc)).max() <= 1
    for i, (dp, op) in enumerate(zip(d_params, o_params)):
        np.testing.assert_allclose(dp, op.reshape(dp.shape), rtol=RTOL, atol=1e-3 * 5e-2, err_msg=f"param {i}")
    for i, dm in enumerate(d_m):
        np.testing.assert_allclose(dm.reshape(-1), oopt.m(i), rtol=2e-3, atol=1e-7, err_msg=f"m {i}")
    assert d_t == steps and list(d_state) == [steps, min(steps * batch, n)]

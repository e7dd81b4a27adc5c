"""th_mlp2_steps -- many training steps of the two-layer MLP in ONE persistent launch (workgroups of one XCD, barriers through its L2) --
against the oracle's step loop (examples/train_mnist.rs:89-121: forward, cross_entropy_loss, backward, Adam::step, zero_grad;
/root/reference/src/train.rs:98-144, optim.rs:83-113): every step's loss and hit count, the parameters and the step counter afterwards."""
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


def _run(ctx, spec, x, y, steps, batch, lr=1e-3, wd=1e-4, t_before=0):
    from taper_amd import hip
    from taper_amd._lib import hip as lib
    ws = [spec[0]["w"], spec[0]["b"], spec[2]["w"], spec[2]["b"]]
    hid, in_f, c = ws[0].shape[0], ws[0].shape[1], ws[2].shape[0]
    t = ctx.upload(np.array([t_before], np.int32))
    dlr = ctx.upload(np.array([lr], np.float32))
    P = [ctx.upload(w) for w in ws]
    M = [ctx.zeros(w.size) for w in ws]
    V = [ctx.zeros(w.size) for w in ws]
    fuse = (hip.AdamFuse * 4)(*[hip.AdamFuse(int(P[i]), int(M[i]), int(V[i]), int(t), int(dlr), 0.9, 0.999, 1e-8, wd) for i in range(4)])
    dx, dy = ctx.upload(x[:steps * batch]), ctx.upload(y[:steps * batch])
    loss, met = ctx.empty(1), ctx.zeros(2 * steps)
    st = ctx.upload(np.zeros(2, np.int64))
    herr = C.c_void_p()
    assert lib.th_host_malloc(ctx.h, 64, C.byref(herr)) == 0
    C.cast(herr, C.POINTER(C.c_int))[0] = 0
    ctx.call("th_mlp2_steps", dx, dy, steps, batch, in_f, hid, c, C.cast(fuse, C.c_void_p), loss, met, steps, st, batch, herr)
    ctx.sync()
    err = C.cast(herr, C.POINTER(C.c_int))[0]
    out = dict(err=err, metrics=ctx.download(met, (steps, 2)), params=[ctx.download(P[i], ws[i].shape) for i in range(4)],
               t=int(ctx.download(t, (1,), np.int32)[0]), state=ctx.download(st, (2,), np.int64), loss=float(ctx.download(loss, (1,))[0]))
    lib.th_host_free(ctx.h, herr)
    return out


def _oracle(spec, x, y, steps, batch, lr=1e-3, wd=1e-4):
    Orc = backends.get("oracle")
    om = Orc.sequential(spec)
    opt = Orc.m.Adam(om.parameters(), lr, None, None, wd)
    in_f = spec[0]["w"].shape[1]
    res = [om.train_step(opt, x[s * batch:(s + 1) * batch], y[s * batch:(s + 1) * batch], (batch, in_f)) for s in range(steps)]
    return res, [p.data() for p in om.parameters()]


@pytest.mark.parametrize("batch,in_f,hid,c,steps", [(64, 784, 128, 10, 1), (64, 784, 128, 10, 12), (32, 784, 128, 10, 5), (16, 64, 32, 3, 7),
                                                    (48, 256, 64, 16, 6), (64, 1024, 128, 10, 3)])
def test_mlp2_steps_match_the_oracle_loop(ctx, batch, in_f, hid, c, steps):
    from taper_amd import hip
    assert hip.hip.th_mlp2_steps_supported(batch, in_f, hid, c) == 1
    rng = np.random.default_rng(batch + in_f + steps)
    s0 = np.sqrt(2.0 / in_f)
    spec = [dict(kind="linear", w=rng.uniform(-s0, s0, (hid, in_f)).astype(np.float32), b=rng.uniform(-.1, .1, hid).astype(np.float32)),
            dict(kind="relu"),
            dict(kind="linear", w=rng.uniform(-.2, .2, (c, hid)).astype(np.float32), b=rng.uniform(-.1, .1, c).astype(np.float32))]
    x = rng.uniform(0, 1, (steps * batch, in_f)).astype(np.float32)
    y = rng.integers(0, c, steps * batch).astype(np.float32)
    ref, ref_params = _oracle(spec, x, y, steps, batch)
    got = _run(ctx, spec, x, y, steps, batch)
    if got["err"] == 2:
        pytest.skip("this box does not place workgroup b on XCD b % 8 (another partition mode?): the launch refused to run, nothing was updated")
    assert got["err"] == 0, "a barrier timed out"
    assert got["t"] == steps and list(got["state"]) == [steps, steps * batch]
    np.testing.assert_allclose(got["metrics"][:, 0], [r["loss"] for r in ref], rtol=3e-4, atol=1e-5)
    assert np.abs(got["metrics"][:, 1] - np.asarray([round(r["acc"] * batch) for r in ref])).max() <= 1
    assert got["loss"] == got["metrics"][-1, 0]
    for i, (hp, op) in enumerate(zip(got["params"], ref_params)):
        np.testing.assert_allclose(hp, op.reshape(hp.shape), rtol=RTOL, atol=1e-3 * 2e-2 * max(1, steps // 4), err_msg=f"param {i}")


def test_mlp2_steps_unsupported():
    from taper_amd import hip
    f = hip.hip.th_mlp2_steps_supported
    assert f(64, 784, 128, 10) == 1 and f(128, 784, 128, 10) == 0 and f(64, 780, 128, 10) == 0 and f(64, 784, 256, 10) == 0 and f(64, 784, 128, 17) == 0

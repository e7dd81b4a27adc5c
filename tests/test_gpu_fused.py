"""Fused launches of the graph-replayed step, each checked against the oracle's
unfused chain: the classifier head (last Linear + cross-entropy + its backward
products in one workgroup launch), the Linear backward with the Adam update in
its epilogue, and the Trainer configurations that combine them."""
import ctypes as C

import numpy as np
import pytest

from tests import margins

from tests import backends
from taper_amd.hip import AdamFuse, AdamSlice   # ctypes mirrors of th_adam_fuse / th_adam_slice

pytestmark = pytest.mark.gpu
RTOL = 1e-4
BOUND_M = 1e-6   # first moments after one fused update, of the tensor's scale: observed <= 4.6e-7 (2x the r04 observation, profiles/r04_parity_margins.json)
BOUND_V = 1e-6   # second moments: observed 0 (bit-identical); one part in 1e6 of the scale
BOUND_EPOCH_LOSS = 1e-6   # per-step losses of an epoch, of the largest: observed <= 4.0e-7 (2x the r04 observation, profiles/r04_parity_margins.json)
BOUND_M15 = 7.9e-6   # first moments after 15 steps: observed 3.9e-6 (2x the r04 observation, profiles/r04_parity_margins.json)
BOUND_CNN_EPOCH_LR = 3.4e-3   # weights after 4 Adam steps, in units of lr: observed 1.66e-3 (2x the r04 observation, profiles/r04_parity_margins.json)
BOUND_GRAPH_EPOCH_LOSS = 4.5e-6   # per-step losses of two epochs, of the largest: observed 2.24e-6 (2x the r04 observation, profiles/r04_parity_margins.json)
BOUND_GRAPH_EPOCH_LR = 2.5e-3   # weights after 8 Adam steps, in units of lr: observed 1.24e-3 (2x the r04 observation, profiles/r04_parity_margins.json)




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


def close(a, b, rtol=RTOL, atol=1e-7):
    b = np.asarray(b)
    scale = float(np.abs(b).max()) if b.size else 0.0
    np.testing.assert_allclose(np.asarray(a).reshape(b.shape), b, rtol=rtol, atol=atol + rtol * 1e-2 * scale)


def oracle_head(O, h, w, b, y):
    """reference chain: Linear (3 nodes) + cross_entropy_loss (6 nodes), backward from the loss"""
    O.Tape.reset()
    O.Tape.set_zero_sentinel(True)   # the loss is never node 0 here
    ht, wt, bt = O.Tensor(h).requires_grad(), O.Tensor(w).requires_grad(), O.Tensor(b).requires_grad()
    logits = ht.matmul(wt.transpose()).add_broadcast(bt)
    yt = O.Tensor(y)
    loss = O.cross_entropy_loss(logits, yt)
    acc = O.accuracy(logits, yt)
    loss.backward()
    out = dict(logits=logits.data(), loss=float(loss.data()[0]), ncorrect=round(acc * len(y)), dh=ht.grad(), dw=wt.grad(), db=bt.grad())
    O.Tape.reset()
    return out


@pytest.mark.parametrize("batch,k,c", [(64, 128, 10), (128, 128, 10), (32, 128, 10), (1, 3, 2), (200, 64, 10), (256, 256, 16), (70, 37, 5), (1024, 64, 10),
                                       (257, 128, 10), (4096, 128, 10), (16500, 128, 10), (33000, 48, 7)])   # > 256: multi-workgroup + finish pass
def test_linear_xent_head(ctx, O, batch, k, c):
    rng = np.random.default_rng(batch * 7 + k + c)
    h = np.maximum(rng.standard_normal((batch, k)), 0).astype(np.float32)        # post-ReLU activations
    w = rng.uniform(-0.3, 0.3, (c, k)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, c).astype(np.float32)
    y = rng.integers(0, c, batch).astype(np.float32)
    ref = oracle_head(O, h, w, b, y)
    dh_, dw_, db_ = ctx.empty(batch * k), ctx.empty(c * k), ctx.empty(c)
    logits, loss, nc = ctx.empty(batch * c), ctx.empty(1), ctx.empty(1)
    state, metrics = ctx.upload(np.array([5, 640], np.int64)), ctx.zeros(2 * 16)
    tick = ctx.upload(np.array([41, 0], np.int32))
    ctx.call("th_linear_xent_head", ctx.upload(h), ctx.upload(w), ctx.upload(b), ctx.upload(y), batch, k, c, logits, loss, nc,
             dh_, dw_, db_, metrics, 16, state, batch, tick, None, None)
    close(ctx.download(logits, (batch, c)), ref["logits"], atol=1e-6)
    assert ctx.download(loss, 1)[0] == pytest.approx(ref["loss"], rel=RTOL, abs=1e-6)
    assert ctx.download(nc, 1)[0] == ref["ncorrect"]
    close(ctx.download(dh_, (batch, k)), ref["dh"])
    close(ctx.download(dw_, (c, k)), ref["dw"])
    close(ctx.download(db_, c), ref["db"])
    np.testing.assert_array_equal(ctx.download(state, 2, np.int64), [6, 640 + batch])
    assert ctx.download(metrics, (16, 2))[5, 0] == ctx.download(loss, 1)[0]
    assert ctx.download(tick, 2, np.int32)[0] == 42                                      # optim.rs:84 folded in
    # every optional output off
    ctx.call("th_linear_xent_head", ctx.upload(h), ctx.upload(w), None, ctx.upload(y), batch, k, c, None, loss, None,
             None, None, None, None, 0, None, 0, None, None, None)
    assert np.isfinite(ctx.download(loss, 1)[0])


@pytest.mark.parametrize("batch", [48, 300])
def test_linear_xent_head_masked(ctx, O, batch):
    """mask_dh_by_h: dH leaves the head already multiplied by (H > 0) -- the ReLU backward of the layer in front"""
    k, c = 128, 10
    rng = np.random.default_rng(batch)
    h = np.maximum(rng.standard_normal((batch, k)), 0).astype(np.float32)
    w = rng.uniform(-0.3, 0.3, (c, k)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, c).astype(np.float32)
    y = rng.integers(0, c, batch).astype(np.float32)
    ref = oracle_head(O, h, w, b, y)
    dh_, loss = ctx.empty(batch * k), ctx.empty(1)
    ctx.call("th_linear_xent_head_masked", ctx.upload(h), ctx.upload(w), ctx.upload(b), ctx.upload(y), batch, k, c, None, loss, None,
             dh_, None, None, None, 0, None, 0, None, None, None, 1)
    close(ctx.download(dh_, (batch, k)), np.asarray(ref["dh"]).reshape(batch, k) * (h > 0))


@pytest.mark.parametrize("batch,k,c", [(256, 3136, 10), (64, 300, 10), (100, 513, 7), (16, 260, 16), (300, 1000, 3), (1, 33, 2)])
def test_linear_xent_wide(ctx, O, batch, k, c):
    """the classifier head on a wide input (two launches: K slices of the logits, then softmax + every backward product)
    against the oracle's Linear + cross_entropy_loss chain"""
    rng = np.random.default_rng(batch + k + c)
    h = np.maximum(rng.standard_normal((batch, k)), 0).astype(np.float32)
    w = (rng.uniform(-1, 1, (c, k)) * np.sqrt(2.0 / k)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, c).astype(np.float32)
    y = rng.integers(0, c, batch).astype(np.float32)
    ref = oracle_head(O, h, w, b, y)
    dx_, dw_, db_ = ctx.empty(batch * k), ctx.empty(c * k), ctx.empty(c)
    loss, nc = ctx.empty(1), ctx.empty(1)
    state, metrics = ctx.upload(np.array([3, 100], np.int64)), ctx.zeros(2 * 16)
    tick = ctx.upload(np.array([7, 0], np.int32))
    wd = ctx.upload(w)
    ctx.call("th_linear_xent_wide", ctx.upload(h), wd, ctx.upload(b), ctx.upload(y), batch, k, c, loss, nc, dx_, dw_, db_, metrics, 16,
             state, batch, tick)
    assert ctx.download(loss, 1)[0] == pytest.approx(ref["loss"], rel=RTOL, abs=1e-6)
    assert ctx.download(nc, 1)[0] == ref["ncorrect"]
    close(ctx.download(dx_, (batch, k)), ref["dh"])
    close(ctx.download(dw_, (c, k)), ref["dw"])
    close(ctx.download(db_, c), ref["db"])
    np.testing.assert_array_equal(ctx.download(state, 2, np.int64), [4, 100 + batch])
    assert ctx.download(metrics, (16, 2))[3, 0] == ctx.download(loss, 1)[0]
    assert ctx.download(tick, 2, np.int32)[0] == 8
    np.testing.assert_array_equal(ctx.download(wd, w.shape), w)
    ctx.call("th_linear_xent_wide", ctx.upload(h), wd, None, ctx.upload(y), batch, k, c, loss, None, None, dw_, None, None, 0, None, 0, None)
    assert np.isfinite(ctx.download(loss, 1)[0])


@pytest.mark.parametrize("batch,c_conv,hw,c", [(256, 64, 49, 10), (96, 64, 49, 10), (37, 5, 9, 7), (300, 8, 36, 3)])
def test_linear_xent_wide_column_sums_and_bias_finish(ctx, O, batch, c_conv, hw, c):
    """th_linear_xent_wide_ex: the masked column sums of dX (what a bias-only Conv2dReLU + pool in front of flatten -> Linear needs,
    Q2) without storing dX, then th_bias_from_colsum_adam: db[ch] = sum of its hw columns (+ Adam) -- against the oracle's dH and the
    pooled-bias formula sum_n sum_sp dH * [x > 0]"""
    k = c_conv * hw
    rng = np.random.default_rng(batch + k + c)
    h = np.maximum(rng.standard_normal((batch, k)), 0).astype(np.float32)      # a ReLU + max-pool output: zeros are masked positions
    w = (rng.uniform(-1, 1, (c, k)) * np.sqrt(2.0 / k)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, c).astype(np.float32)
    y = rng.integers(0, c, batch).astype(np.float32)
    ref = oracle_head(O, h, w, b, y)
    dh = np.asarray(ref["dh"]).reshape(batch, k)
    col_ref = (dh * (h > 0)).sum(axis=0, dtype=np.float64)
    dw_, db_, cs_ = ctx.empty(c * k), ctx.empty(c), ctx.upload(np.full(k, 7.0, np.float32))     # overwritten, not accumulated
    loss = ctx.empty(1)
    ctx.call("th_linear_xent_wide_ex", ctx.upload(h), ctx.upload(w), ctx.upload(b), ctx.upload(y), batch, k, c, loss, None, None, dw_, db_, None, 0,
             None, 0, None, cs_)
    assert ctx.download(loss, 1)[0] == pytest.approx(ref["loss"], rel=RTOL, abs=1e-6)
    close(ctx.download(dw_, (c, k)), ref["dw"])
    got = ctx.download(cs_, k)
    np.testing.assert_allclose(got, col_ref, rtol=RTOL, atol=RTOL * float(np.abs(col_ref).max()) + 1e-9)
    # with dX requested as well: the same sums, and dX as before
    dx_, cs2 = ctx.empty(batch * k), ctx.empty(k)
    ctx.call("th_linear_xent_wide_ex", ctx.upload(h), ctx.upload(w), ctx.upload(b), ctx.upload(y), batch, k, c, loss, None, dx_, dw_, db_, None, 0,
             None, 0, None, cs2)
    close(ctx.download(dx_, (batch, k)), dh)
    np.testing.assert_array_equal(ctx.download(cs2, k), got)                       # deterministic
    # the finishing launch: channel sums, then the same with the bias's Adam update fused
    db_ref = col_ref.reshape(c_conv, hw).sum(axis=1)
    gb = ctx.empty(c_conv)
    ctx.call("th_bias_from_colsum_adam", cs_, gb, c_conv, hw, None, None, 0)
    np.testing.assert_allclose(ctx.download(gb, c_conv), db_ref, rtol=RTOL, atol=RTOL * float(np.abs(db_ref).max()) + 1e-9)
    from taper_amd import hip
    p0 = rng.uniform(-0.5, 0.5, c_conv).astype(np.float32)
    pd, md, vd = ctx.upload(p0), ctx.zeros(c_conv), ctx.zeros(c_conv)
    tick, lr = ctx.upload(np.array([3, 0], np.int32)), ctx.upload(np.array([1e-2], np.float32))
    fuse = hip.AdamFuse(int(pd), int(md), int(vd), int(tick), int(lr), 0.9, 0.999, 1e-8, 1e-4)
    ctx.call("th_bias_from_colsum_adam", cs_, gb, c_conv, hw, C.byref(fuse), None, 0)
    p_ref, m_ref, v_ref = _adam_ref(O, p0, ctx.download(gb, c_conv), 1e-2, 3)
    np.testing.assert_allclose(ctx.download(pd, c_conv), p_ref, rtol=RTOL, atol=1e-2 * 2e-2)
    margins.check("conv_bias_m", ctx.download(md, c_conv), m_ref, BOUND_M)


@pytest.mark.parametrize("batch,c_conv,hw,c", [(256, 64, 49, 10), (96, 64, 49, 10), (40, 6, 50, 7)])
def test_linear_xent_wide_fused_tail(ctx, O, batch, c_conv, hw, c):
    """th_linear_xent_wide_fused: the whole tail of a Conv2dReLU(+pool) -> flatten -> Linear -> cross-entropy step in the head's two
    launches -- Adam on W (every workgroup its own columns), on b (lead), the conv bias finished by the last workgroup to arrive, the
    step counter ticked in the launch -- against oracle gradients followed by oracle Adam at t = old + 1; two consecutive launches
    (the arrival counter must come back to 0)"""
    from taper_amd.hip import WideFuse
    k = c_conv * hw
    rng = np.random.default_rng(batch * 3 + k + c)
    h = np.maximum(rng.standard_normal((batch, k)), 0).astype(np.float32)
    w0 = (rng.uniform(-1, 1, (c, k)) * np.sqrt(2.0 / k)).astype(np.float32)
    b0 = rng.uniform(-0.1, 0.1, c).astype(np.float32)
    cb0 = rng.uniform(-0.3, 0.3, c_conv).astype(np.float32)
    y = rng.integers(0, c, batch).astype(np.float32)
    lr = 1e-2
    wd_, bd_, cbd_ = ctx.upload(w0), ctx.upload(b0), ctx.upload(cb0)
    mom = {n: (ctx.zeros(sz), ctx.zeros(sz)) for n, sz in (("w", c * k), ("b", c), ("cb", c_conv))}
    tick, lrd = ctx.upload(np.array([4, 0, 0, 0], np.int32)), ctx.upload(np.array([lr], np.float32))
    fz = lambda p, n: AdamFuse(int(p), int(mom[n][0]), int(mom[n][1]), int(tick), int(lrd), 0.9, 0.999, 1e-8, 1e-4)
    gcb = ctx.empty(c_conv)
    f = WideFuse(fz(wd_, "w"), fz(bd_, "b"), fz(cbd_, "cb"), int(gcb), c_conv, hw)
    dw_, db_, cs_, loss = ctx.empty(c * k), ctx.empty(c), ctx.empty(k), ctx.empty(1)
    hd, yd = ctx.upload(h), ctx.upload(y)
    w_ref, b_ref, cb_ref = w0, b0, cb0
    state = {n: None for n in ("w", "b", "cb")}
    for step in range(2):
        ref = oracle_head(O, h, w_ref, b_ref, y)
        dh = np.asarray(ref["dh"]).reshape(batch, k)
        gcb_ref = (dh * (h > 0)).sum(axis=0, dtype=np.float64).reshape(c_conv, hw).sum(axis=1).astype(np.float32)
        ctx.call("th_linear_xent_wide_fused", hd, wd_, bd_, yd, batch, k, c, loss, None, dw_, db_, None, 0, None, 0, tick, cs_, C.byref(f))
        assert ctx.download(loss, 1)[0] == pytest.approx(ref["loss"], rel=RTOL, abs=1e-6)
        close(ctx.download(dw_, (c, k)), ref["dw"])
        close(ctx.download(db_, c), ref["db"])
        np.testing.assert_allclose(ctx.download(gcb, c_conv), gcb_ref, rtol=RTOL, atol=RTOL * float(np.abs(gcb_ref).max()) + 1e-9)
        np.testing.assert_array_equal(ctx.download(tick, 4, np.int32)[:2], [5 + step, 0])      # ticked once, arrival counter back at 0
        if step == 0:      # one Adam step from zero moments at t = 5 (the oracle helper starts from m = v = 0)
            for name, p0, g, dev in (("w", w0, np.asarray(ref["dw"]).reshape(c, k), wd_), ("b", b0, np.asarray(ref["db"]), bd_), ("cb", cb0, gcb_ref, cbd_)):
                p_ref, m_ref, _ = _adam_ref(O, p0.reshape(-1), g.reshape(-1).astype(np.float32), lr, 5)
                np.testing.assert_allclose(ctx.download(dev, p0.size), p_ref, rtol=RTOL, atol=lr * 2e-2, err_msg=name)
                margins.check(f"{name}_m", ctx.download(mom[name][0], p0.size), m_ref, BOUND_M)
        # the second launch runs on the updated parameters
        w_ref, b_ref = ctx.download(wd_, (c, k)), ctx.download(bd_, c)


def test_head_limits_are_errors(ctx):
    from taper_amd._lib import TaperError
    x = ctx.zeros(64 * 300)
    with pytest.raises(TaperError, match="classes <= 16"):
        ctx.call("th_linear_xent_head", x, x, None, x, 4, 300, 10, None, x, None, None, None, None, None, 0, None, 0, None, None, None)
    with pytest.raises(TaperError, match="classes <= 16"):
        ctx.call("th_linear_xent_head", x, x, None, x, 4, 64, 17, None, x, None, None, None, None, None, 0, None, 0, None, None, None)


def _adam_ref(O, p0, g, lr, t, wd=1e-4):
    """one oracle Adam step at step counter t (m = v = 0 before)"""
    pt = O.Tensor(p0).requires_grad()
    opt = O.Adam([pt], lr, None, None, wd)
    for _ in range(t - 1):      # advance the counter with zero-effect steps (grad None: skipped, t still ticks)
        opt.step()
    pt.set_grad(g)
    opt.step()
    return pt.data(), opt.m(0), opt.v(0)


@pytest.mark.parametrize("batch,inf,outf,with_dx", [(64, 784, 128, False), (64, 128, 64, True), (16, 40, 24, False),
                                                    (4096, 784, 128, False), (3000, 256, 130, True),   # large: Adam rides in the split-K reduce
                                                    (1024, 784, 128, False), (600, 64, 48, True)])      # mid: K slices + reduce carry the update
def test_linear_bwd_adam_epilogue(ctx, O, batch, inf, outf, with_dx):
    """dW/db from the fused kernel + Adam applied in the epilogue == oracle grads followed by oracle Adam"""
    rng = np.random.default_rng(batch + inf + outf)
    x = rng.uniform(0, 1, (batch, inf)).astype(np.float32)
    w = rng.uniform(-0.1, 0.1, (outf, inf)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, outf).astype(np.float32)
    dy = (rng.standard_normal((batch, outf)) * 0.01).astype(np.float32)
    yact = np.maximum(rng.standard_normal((batch, outf)), 0).astype(np.float32)          # post-ReLU output -> mask
    dz = dy * (yact > 0)
    gw, gb, gx = dz.T @ x, dz.sum(0), dz @ w
    lr, t = 1e-3, 3
    w_ref, wm_ref, wv_ref = _adam_ref(O, w, gw.astype(np.float32), lr, t)
    b_ref, _, _ = _adam_ref(O, b, gb.astype(np.float32), lr, t)
    dw_, db_, dx_ = ctx.empty(w.size), ctx.empty(b.size), (ctx.empty(x.size) if with_dx else None)
    pw, pb = ctx.upload(w), ctx.upload(b)
    mw, vw, mb, vb = ctx.zeros(w.size), ctx.zeros(w.size), ctx.zeros(b.size), ctx.zeros(b.size)
    tick, dlr = ctx.upload(np.array([t, 0], np.int32)), ctx.upload(np.array([lr], np.float32))   # pre-ticked
    wf = AdamFuse(int(pw), int(mw), int(vw), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4)
    bf = AdamFuse(int(pb), int(mb), int(vb), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4)
    w_in = ctx.upload(w)   # the kernel reads W for dX; keep an untouched copy as its operand when W itself is updated
    ctx.call("th_linear_bwd_adam", ctx.upload(x), pw if with_dx else w_in, ctx.upload(dy), ctx.upload(yact), dx_, dw_, db_, batch, inf,
             outf, 0, C.byref(wf), C.byref(bf))
    close(ctx.download(dw_, w.shape), gw, atol=1e-6)
    close(ctx.download(db_, b.shape), gb, atol=1e-6)
    if with_dx:
        close(ctx.download(dx_, x.shape), gx, atol=1e-6)      # dX used the PRE-update W (no race with the fused update)
    np.testing.assert_allclose(ctx.download(pw, w.shape), w_ref, rtol=RTOL, atol=lr * 2e-2)
    np.testing.assert_allclose(ctx.download(pb, b.shape), b_ref, rtol=RTOL, atol=lr * 2e-2)
    margins.check("w_m", ctx.download(mw, w.shape), wm_ref.reshape(w.shape), BOUND_M)
    assert ctx.download(tick, 2, np.int32)[0] == t            # fused epilogues never tick


def test_fused_bwd_rejects_accumulating_into_fused_grad(ctx):
    from taper_amd._lib import TaperError
    z = ctx.zeros(64 * 64)
    f = AdamFuse(int(z), int(z), int(z), int(z), int(z), 0.9, 0.999, 1e-8, 0.0)
    with pytest.raises(TaperError, match="must not accumulate"):
        ctx.call("th_linear_bwd_adam", z, z, z, None, None, z, None, 8, 8, 8, 2, C.byref(f), None)




@pytest.mark.parametrize("carrier", ["linear_bwd", "standalone"])
@pytest.mark.parametrize("sizes", [(1280, 10), (1, 1023, 1025, 4099)])
def test_deferred_adam_slices(ctx, O, carrier, sizes):
    """updates of already-complete gradients carried by another launch (or by th_adam_slices) == oracle Adam;
    ragged lengths cover the float4 body, the scalar tail and slices spanning several workgroups"""
    rng = np.random.default_rng(sum(sizes))
    lr, t = 1e-3, 5
    tick, dlr = ctx.upload(np.array([t, 0], np.int32)), ctx.upload(np.array([lr], np.float32))
    slices, refs, keep = (AdamSlice * len(sizes))(), [], []
    for i, n in enumerate(sizes):
        p0 = rng.uniform(-0.1, 0.1, n).astype(np.float32)
        g = (rng.standard_normal(n) * 0.01).astype(np.float32)
        refs.append(_adam_ref(O, p0, g, lr, t))
        bufs = [ctx.upload(p0), ctx.zeros(n), ctx.zeros(n), ctx.upload(g)]
        keep.append(bufs)
        slices[i] = AdamSlice(int(bufs[3]), n, AdamFuse(int(bufs[0]), int(bufs[1]), int(bufs[2]), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4))
    if carrier == "standalone":
        ctx.call("th_adam_slices", slices, len(sizes))
    else:
        batch, inf, outf = 32, 48, 24
        x = rng.uniform(0, 1, (batch, inf)).astype(np.float32)
        dy = (rng.standard_normal((batch, outf)) * 0.01).astype(np.float32)
        dw_, db_ = ctx.empty(outf * inf), ctx.empty(outf)
        ctx.call("th_linear_bwd_adam_ex", ctx.upload(x), None, ctx.upload(dy), None, None, dw_, db_, batch, inf, outf, 0, None, None,
                 slices, len(sizes))
        close(ctx.download(dw_, (outf, inf)), dy.T @ x, atol=1e-6)       # the carrier's own products are untouched
        close(ctx.download(db_, (outf,)), dy.sum(0), atol=1e-6)
    for bufs, (p_ref, m_ref, v_ref), n in zip(keep, refs, sizes):
        np.testing.assert_allclose(ctx.download(bufs[0], (n,)), p_ref, rtol=RTOL, atol=lr * 2e-2)
        margins.check("slice_m", ctx.download(bufs[1], (n,)), m_ref.reshape(n), BOUND_M)
        margins.check("slice_v", ctx.download(bufs[2], (n,)), v_ref.reshape(n), BOUND_V)
    assert ctx.download(tick, 2, np.int32)[0] == t


@pytest.mark.parametrize("n,c,hw", [(256, 64, 49), (256, 32, 196), (6, 5, 9), (40, 128, 49), (300, 3, 1)])
@pytest.mark.parametrize("pooled_avg", [0, 1])
@pytest.mark.parametrize("carry", [(), (640, 10, 4099)])
def test_bias_grad_masked_adam(ctx, O, n, c, hw, pooled_avg, carry):
    """th_bias_grad_masked_adam: the bias gradient summed from the pool's tensors, Adam on the bias in the same workgroup,
    other parameters' complete gradients updated by the launch's extra workgroups == numpy sums + oracle Adam"""
    rng = np.random.default_rng(n + c + hw + pooled_avg + len(carry))
    yp = np.maximum(rng.standard_normal((n, c, hw)), 0).astype(np.float32)
    if pooled_avg:
        g = (rng.standard_normal((n, c)) * 0.01).astype(np.float32)
        gb = (g.astype(np.float64)[:, :, None] / hw * (yp > 0)).sum((0, 2))
    else:
        g = (rng.standard_normal((n, c, hw)) * 0.01).astype(np.float32)
        gb = (g.astype(np.float64) * (yp > 0)).sum((0, 2))
    lr, t = 1e-3, 4
    b0 = rng.uniform(-0.1, 0.1, c).astype(np.float32)
    b_ref, m_ref, _ = _adam_ref(O, b0, gb.astype(np.float32), lr, t)
    tick, dlr = ctx.upload(np.array([t, 0], np.int32)), ctx.upload(np.array([lr], np.float32))
    pb, mb, vb, out = ctx.upload(b0), ctx.zeros(c), ctx.zeros(c), ctx.empty(c)
    bf = AdamFuse(int(pb), int(mb), int(vb), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4)
    slices, refs, keep = (AdamSlice * max(len(carry), 1))(), [], []
    for i, k in enumerate(carry):
        p0 = rng.uniform(-0.1, 0.1, k).astype(np.float32)
        gk = (rng.standard_normal(k) * 0.01).astype(np.float32)
        refs.append(_adam_ref(O, p0, gk, lr, t))
        bufs = [ctx.upload(p0), ctx.zeros(k), ctx.zeros(k), ctx.upload(gk)]
        keep.append(bufs)
        slices[i] = AdamSlice(int(bufs[3]), k, AdamFuse(int(bufs[0]), int(bufs[1]), int(bufs[2]), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4))
    ctx.call("th_bias_grad_masked_adam", ctx.upload(g), ctx.upload(yp), out, n, c, hw, pooled_avg, C.byref(bf),
             slices if carry else None, len(carry))
    np.testing.assert_allclose(ctx.download(out, c), gb, rtol=1e-4, atol=1e-4 * float(np.abs(gb).max()) + 1e-7)
    np.testing.assert_allclose(ctx.download(pb, c), b_ref, rtol=RTOL, atol=lr * 2e-2)
    margins.check("bias_m", ctx.download(mb, c), m_ref.reshape(c), BOUND_M)
    for bufs, (p_ref, mk_ref, _), k in zip(keep, refs, carry):
        np.testing.assert_allclose(ctx.download(bufs[0], (k,)), p_ref, rtol=RTOL, atol=lr * 2e-2)
        margins.check("carried_m", ctx.download(bufs[1], (k,)), mk_ref.reshape(k), BOUND_M)
    assert ctx.download(tick, 2, np.int32)[0] == t
    # without a fuse descriptor the launch is the plain overwrite-form gradient
    out2 = ctx.upload(np.full(c, 7.0, np.float32))
    ctx.call("th_bias_grad_masked_adam", ctx.upload(g), ctx.upload(yp), out2, n, c, hw, pooled_avg, None, None, 0)
    np.testing.assert_allclose(ctx.download(out2, c), gb, rtol=1e-4, atol=1e-4 * float(np.abs(gb).max()) + 1e-7)


@pytest.mark.parametrize("n,c,hw", [(256, 128, 49), (6, 5, 9), (40, 33, 49), (300, 3, 1), (17, 64, 196)])
@pytest.mark.parametrize("fused", [False, True])
def test_bias_grad_from_plane_counts(ctx, O, n, c, hw, fused):
    """th_avgpool2d_global_fwd_counts (plane means + counts of elements > 0) and th_bias_grad_counts_adam: the bias gradient of a
    Conv2dReLU in front of a global average pool from 2 n c floats == the sum over all n c hw masked elements (+ oracle Adam)"""
    rng = np.random.default_rng(n + c + hw)
    yp = np.maximum(rng.standard_normal((n, c, hw)), 0).astype(np.float32)
    g = (rng.standard_normal((n, c)) * 0.01).astype(np.float32)
    mean, cnt = ctx.empty(n * c), ctx.empty(n * c)
    ctx.call("th_avgpool2d_global_fwd_counts", ctx.upload(yp), mean, cnt, n, c, hw)
    np.testing.assert_allclose(ctx.download(mean, (n, c)), yp.astype(np.float64).mean(2), rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(ctx.download(cnt, (n, c)), (yp > 0).sum(2).astype(np.float32))
    gb = (g.astype(np.float64)[:, :, None] / hw * (yp > 0)).sum((0, 2))
    lr, t = 1e-3, 4
    b0 = rng.uniform(-0.1, 0.1, c).astype(np.float32)
    tick, dlr = ctx.upload(np.array([t, 0], np.int32)), ctx.upload(np.array([lr], np.float32))
    pb, mb, vb, out = ctx.upload(b0), ctx.zeros(c), ctx.zeros(c), ctx.empty(c)
    bf = AdamFuse(int(pb), int(mb), int(vb), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4)
    k = 1500
    p0, gk = rng.uniform(-0.1, 0.1, k).astype(np.float32), (rng.standard_normal(k) * 0.01).astype(np.float32)
    bufs = [ctx.upload(p0), ctx.zeros(k), ctx.zeros(k), ctx.upload(gk)]
    sl = (AdamSlice * 1)(AdamSlice(int(bufs[3]), k, AdamFuse(int(bufs[0]), int(bufs[1]), int(bufs[2]), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4)))
    ctx.call("th_bias_grad_counts_adam", ctx.upload(g), cnt, out, n, c, hw, C.byref(bf) if fused else None, sl if fused else None, 1 if fused else 0)
    np.testing.assert_allclose(ctx.download(out, c), gb, rtol=1e-4, atol=1e-4 * float(np.abs(gb).max()) + 1e-7)
    if fused:
        b_ref, m_ref, _ = _adam_ref(O, b0, gb.astype(np.float32), lr, t)
        np.testing.assert_allclose(ctx.download(pb, c), b_ref, rtol=RTOL, atol=lr * 2e-2)
        margins.check("bias_m", ctx.download(mb, c), m_ref.reshape(c), BOUND_M)
        p_ref, _, _ = _adam_ref(O, p0, gk, lr, t)
        np.testing.assert_allclose(ctx.download(bufs[0], (k,)), p_ref, rtol=RTOL, atol=lr * 2e-2)
    else:
        np.testing.assert_array_equal(ctx.download(pb, c), b0)


def test_carried_slice_must_not_alias_the_weight_read_for_dx(ctx):
    from taper_amd._lib import TaperError
    z = ctx.zeros(64 * 64)
    sl = (AdamSlice * 1)(AdamSlice(int(z), 64, AdamFuse(int(z), int(z), int(z), int(z), int(z), 0.9, 0.999, 1e-8, 0.0)))
    with pytest.raises(TaperError, match="must not alias"):
        ctx.call("th_linear_bwd_adam_ex", z, z, z, None, z, None, None, 8, 8, 8, 0, None, None, sl, 1)


CONFIGS = [dict(graph_chunk=1, fuse_head=False, fuse_adam=False), dict(graph_chunk=8, fuse_head=True, fuse_adam=False),
           dict(graph_chunk=8, fuse_head=False, fuse_adam=True), dict(graph_chunk=32, fuse_head=True, fuse_adam=True),
           dict(graph_chunk=32, fuse_head=1, fuse_adam=True),    # head-only fusion (the three-launch step)
           dict(graph_chunk=3, fuse_head=2, fuse_adam=True)]     # tail fusion with short chunks: carried updates + flushes


@pytest.mark.parametrize("model_name,batch", [("mlp_baseline", 64), ("mlp_example", 96)])
@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"chunk{c['graph_chunk']}-head{int(c['fuse_head'])}-adam{int(c['fuse_adam'])}")
def test_fused_epoch_matches_oracle(model_name, batch, cfg):
    """every fusion configuration of the graph path against the oracle driven with the same batches
    (3 epochs of 5 steps incl. a partial batch; Adam's t must tick once per step in every config)"""
    import taper_amd as T
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(11)
    n = 4 * batch + batch // 2
    spec = backends.nonzero_biases(getattr(backends, model_name)(rng), rng)
    x, y = backends.mnist_like(rng, n)
    hm, om = H.sequential(spec), Orc.sequential(spec)
    hopt, oopt = T.Adam(hm.parameters(), 1e-3, None, None, 1e-4), Orc.m.Adam(om.parameters(), 1e-3, None, None, 1e-4)
    tr = T.Trainer(hm, hopt, **cfg)
    loader = T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False)
    for epoch in range(3):
        ep = tr.run_epoch(loader, T.Trainer.GRAPH)
        ref_losses, ref_nc = [], []
        for s in range(0, n, batch):
            xb, yb = x[s:s + batch], y[s:s + batch]
            r = om.train_step(oopt, xb, yb, (len(xb), 784))
            ref_losses.append(r["loss"])
            ref_nc.append(round(r["acc"] * len(xb)))
        margins.check(f"losses_epoch{epoch}", ep["losses"], ref_losses, BOUND_EPOCH_LOSS)
        assert np.abs(ep["ncorrect"] - np.array(ref_nc)).max() <= 1
    assert hopt.t() == oopt.t() == 15
    for i, (hp, op) in enumerate(zip(hm.parameters(), om.parameters())):
        np.testing.assert_allclose(hp.data(), op.data(), rtol=RTOL, atol=1e-3 * 5e-2, err_msg=f"param {i}")
    m, v = hopt.moments()
    margins.check("adam_m_after_15_steps", m, np.concatenate([oopt.m(i) for i in range(len(om.parameters()))]), BOUND_M15)


def test_cnn_epoch_graph_matches_oracle():
    """reference CNN (faithful mode) through the graph path: conv stack generic, classifier head fused"""
    import taper_amd as T
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(13)
    batch, n = 8, 28
    spec = backends.nonzero_biases(backends.cnn_reference(rng), rng)
    x, y = backends.mnist_like(rng, n)
    hm, om = H.sequential(spec), Orc.sequential(spec)
    hopt, oopt = T.Adam(hm.parameters(), 1e-2, None, None, 1e-4), Orc.m.Adam(om.parameters(), 1e-2, None, None, 1e-4)
    tr = T.Trainer(hm, hopt, sample_shape=(1, 28, 28), graph_chunk=2)
    loader = T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False)
    ep = tr.run_epoch(loader, T.Trainer.GRAPH)
    ref = [om.train_step(oopt, x[s:s + batch], y[s:s + batch], (len(x[s:s + batch]), 1, 28, 28))["loss"] for s in range(0, n, batch)]
    margins.check("losses", ep["losses"], ref, BOUND_EPOCH_LOSS)
    assert hopt.t() == 4
    for i, (hp, op) in enumerate(zip(hm.parameters(), om.parameters())):
        margins.check(f"param{i}", hp.data(), op.data(), BOUND_CNN_EPOCH_LR, lr=1e-2)
    # quirk Q2 survives the fused path: conv weights untouched by the optimizer
    np.testing.assert_array_equal(hm.parameters()[0].data(), spec[0]["w"])


def test_data_parallel_step_with_single_rank_communicator_matches_plain_step():
    """the data-parallel op list (backward -> RCCL mean all-reduce of the flat grad arena -> unfused Adam),
    captured into the step graph with a 1-rank communicator, gives the single-GPU results"""
    import taper_amd as T
    H = backends.get("hip")
    rng = np.random.default_rng(13)
    spec = backends.nonzero_biases(backends.mlp_baseline(rng), rng)
    x, y = backends.mnist_like(rng, 64 * 40 + 17)
    out = []
    for with_comm in (False, True):
        model = H.sequential(spec)
        opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
        comm = T.Communicator(1, 0, T.Communicator.unique_id()) if with_comm else None
        tr = T.Trainer(model, opt, comm=comm, fuse_adam=False)
        ep = tr.run_epoch(T.DataLoader(T.MNISTDataset.from_host(x, y), 64, False), T.Trainer.GRAPH)
        out.append((ep["losses"], [p.data() for p in model.parameters()], opt.t()))
        del tr, comm
    np.testing.assert_array_equal(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        np.testing.assert_array_equal(a, b)
    assert out[0][2] == out[1][2] == 41


@pytest.mark.parametrize("model_name,batch", [("cnn_simple", 16), ("cnn_reference", 8)])
def test_cnn_graph_epoch_matches_oracle(model_name, batch):
    """the graph path of the CNNs (faithful mode, Q2: only the last conv's bias and the Linears train) against the
    oracle: the Conv2dReLU -> MaxPool2d pair of the simple CNN takes the pooled bias-gradient shortcut
    (th_bias_grad_nchw_masked on the pool's OUTPUT gradient, no scatter, no ReLU-backward pass)"""
    import taper_amd as T
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(17)
    n = 3 * batch + batch // 2
    spec = backends.nonzero_biases(getattr(backends, model_name)(rng), rng)
    x, y = backends.mnist_like(rng, n)
    hm, om = H.sequential(spec), Orc.sequential(spec)
    lr = 1e-2
    hopt, oopt = T.Adam(hm.parameters(), lr, None, None, 1e-4), Orc.m.Adam(om.parameters(), lr, None, None, 1e-4)
    tr = T.Trainer(hm, hopt, sample_shape=(1, 28, 28))
    loader = T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False)
    for epoch in range(2):
        ep = tr.run_epoch(loader, T.Trainer.GRAPH)
        ref_losses = []
        for s in range(0, n, batch):
            xb, yb = x[s:s + batch], y[s:s + batch]
            ref_losses.append(om.train_step(oopt, xb, yb, (len(xb), 1, 28, 28))["loss"])
        margins.check(f"losses_epoch{epoch}", ep["losses"], ref_losses, BOUND_GRAPH_EPOCH_LOSS)
    for i, (hp, op) in enumerate(zip(hm.parameters(), om.parameters())):
        margins.check(f"param{i}", hp.data(), op.data(), BOUND_GRAPH_EPOCH_LR, lr=lr)
    assert hopt.t() == oopt.t() == 8


@pytest.mark.parametrize("case", ["no_bias", "ragged", "deep_ragged", "no_decay"])
def test_tail_path_edge_models_match_oracle(case):
    """graph path (two-launch tail where it applies) against the oracle for models off the MNIST shapes: Linears without
    bias, sizes that take the general (non whole-tile) tail kernel, a deeper ragged MLP (no dX role -> head-only form),
    and Adam without weight decay"""
    import taper_amd as T
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng({"no_bias": 1, "ragged": 2, "deep_ragged": 3, "no_decay": 4}[case])
    lin = lambda i, o, b=True: dict(kind="linear", w=(rng.uniform(-1, 1, (o, i)) * np.sqrt(2.0 / i)).astype(np.float32),
                                    b=(rng.uniform(-0.1, 0.1, o).astype(np.float32) if b else None))
    batch = 48
    if case == "no_bias":
        spec = [lin(784, 128, False), dict(kind="relu"), lin(128, 10, False)]
    elif case == "ragged":
        spec, batch = [lin(784, 100), dict(kind="relu"), lin(100, 7)], 50
    elif case == "deep_ragged":
        spec, batch = [lin(784, 60), dict(kind="relu"), lin(60, 36), dict(kind="relu"), lin(36, 10)], 40
    else:
        spec = [lin(784, 128), dict(kind="relu"), lin(128, 10)]
    n = 3 * batch + batch // 2
    x, y = backends.mnist_like(rng, n)
    if case == "ragged":
        y = (y % 7).astype(np.float32)
    hm, om = H.sequential(spec), Orc.sequential(spec)
    wd = None if case == "no_decay" else 1e-4
    hopt, oopt = T.Adam(hm.parameters(), 1e-3, None, None, wd), Orc.m.Adam(om.parameters(), 1e-3, None, None, wd)
    tr = T.Trainer(hm, hopt)
    loader = T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False)
    for epoch in range(2):
        ep = tr.run_epoch(loader, T.Trainer.GRAPH)
        ref = [om.train_step(oopt, x[s:s + batch], y[s:s + batch], (len(x[s:s + batch]), 784))["loss"] for s in range(0, n, batch)]
        margins.check(f"losses_epoch{epoch}", ep["losses"], ref, BOUND_EPOCH_LOSS)
    for i, (hp, op) in enumerate(zip(hm.parameters(), om.parameters())):
        np.testing.assert_allclose(hp.data(), op.data(), rtol=1e-4, atol=2e-5, err_msg=f"param {i}")


@pytest.mark.parametrize("model_name,batch,shape", [("mlp_baseline", 64, None), ("mlp_example", 256, None), ("cnn_simple", 32, (1, 28, 28))])
def test_graph_path_is_deterministic(model_name, batch, shape):
    """no atomics anywhere on the path (fixed-order reductions, slot sums): two runs from the same weights and batches
    give bit-identical per-step losses and weights"""
    import taper_amd as T
    H = backends.get("hip")
    rng = np.random.default_rng(99)
    spec = backends.nonzero_biases(getattr(backends, model_name)(rng), rng)
    n = 5 * batch + 7
    x, y = backends.mnist_like(rng, n)
    runs = []
    for _ in range(2):
        model = H.sequential(spec)
        tr = T.Trainer(model, T.Adam(model.parameters(), 1e-3, None, None, 1e-4), sample_shape=shape)
        loader = T.DataLoader(T.MNISTDataset.from_host(x, y), batch, True, seed=5)
        losses = [np.asarray(tr.run_epoch(loader, T.Trainer.GRAPH)["losses"], np.float32) for _ in range(2)]
        runs.append((np.concatenate(losses), [np.asarray(p.data(), np.float32) for p in model.parameters()]))
    np.testing.assert_array_equal(runs[0][0].view(np.uint32), runs[1][0].view(np.uint32))
    for a, b in zip(runs[0][1], runs[1][1]):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))

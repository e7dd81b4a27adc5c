"""The simple CNN's step (BASELINE configs[2]) as TWO launches: th_conv_chain_head_fwd -- the convolutional front with the classifier's
rows (Flatten + Linear + softmax cross-entropy: logits, NLL, hit, dlogits, the per-channel sums of dX * [x > 0]) in its last epilogue --
and th_wide_head_grads -- dW, db, loss, hit count, the last conv's bias gradient, Adam in the epilogues.  Against the oracle's tape over the
same model (/root/reference/src/nn.rs:54-60, 433-490, 508-549, 730-756; src/loss.rs:101-195, 271-290; src/ops.rs:238-294;
src/tensor.rs:1221-1285, 1391-1521, 2017-2024; src/optim.rs:83-113), at batch 256, ragged batches and 16 classes; then the Trainer's
captured step against the oracle's training loop and against the same step with the head as its own launches."""
import ctypes as C

import numpy as np
import pytest

from tests import backends, margins

pytestmark = pytest.mark.gpu
RTOL = 1e-4
BOUND_M = 4.5e-6   # observed 2.2e-6 (conv bias) (2x the r04 observation, profiles/r04_parity_margins.json)
BOUND_V = 8.1e-6   # observed 4.0e-6 (conv bias) (2x the r04 observation, profiles/r04_parity_margins.json)
SIMPLE = [(1, 32, 1), (32, 64, 1)]
K = 64 * 49


@pytest.fixture(scope="module")
def ctx():
    from taper_amd import hip
    c = hip.Ctx(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    oracle.Tape.set_zero_sentinel(True)
    return oracle


class ChainHead(C.Structure):
    """include/taper_hip.h: th_chain_head"""
    _fields_ = [("d_w", C.c_void_p), ("d_bias", C.c_void_p), ("d_targets", C.c_void_p), ("classes", C.c_int), ("d_dl", C.c_void_p),
                ("d_rowstat", C.c_void_p), ("d_cbpart", C.c_void_p), ("d_tick", C.c_void_p)]


def _k_of(spec):
    hw = 28
    for _, _, post in spec:
        hw = hw // 2 if post == 1 else hw
    return spec[-1][1] * hw * hw, spec[-1][1], hw


def _model(rng, classes, spec=SIMPLE):
    K = _k_of(spec)[0]
    conv = []
    for c_in, c_out, _ in spec:
        bound = np.sqrt(6.0 / (c_in * 9))
        conv.append((rng.uniform(-bound, bound, (c_out, c_in, 3, 3)).astype(np.float32), rng.uniform(-0.1, 0.1, c_out).astype(np.float32)))
    w = (rng.uniform(-1, 1, (classes, K)) * np.sqrt(2.0 / K)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, classes).astype(np.float32)
    return conv, w, b


def _oracle_step(O, conv, w, b, x, y, spec=SIMPLE):
    """the oracle's tape over conv rows -> flatten -> linear -> cross-entropy; -> everything the two launches produce"""
    O.Tape.reset()
    cw = [O.Tensor(a).requires_grad() for a, _ in conv]
    cb = [O.Tensor(a).requires_grad() for _, a in conv]
    wt, bt = O.Tensor(w).requires_grad(), O.Tensor(b).requires_grad()
    t = O.Tensor(x)
    for i in range(len(conv)):
        t = t.conv2d_relu(cw[i], cb[i], (1, 1), (1, 1), (1, 1))
        if spec[i][2] == 1:
            t = t.max_pool2d((2, 2), (2, 2), (0, 0))
    pooled = t.data().copy()
    logits = t.flatten(1).matmul(wt.transpose()).add_broadcast(bt)
    lg = logits.data().copy()
    loss = O.cross_entropy_loss(logits, O.Tensor(y))
    loss.backward()
    n = x.shape[0]
    m = lg - lg.max(axis=1, keepdims=True)
    logp = m - np.log(np.exp(m).sum(axis=1, keepdims=True))
    nll = -logp[np.arange(n), y.astype(int)]
    hit = (lg.argmax(axis=1) == y.astype(int)).astype(np.float32)
    dl = (np.exp(logp) - np.eye(lg.shape[1], dtype=np.float32)[y.astype(int)]) / n
    assert all(c.grad() is None for c in cw) and all(c.grad() is None for c in cb[:-1])      # Q2: the tape is cut at every conv
    return dict(pooled=pooled, logits=lg, loss=float(loss.data()[0]), nll=nll, hit=hit, dl=dl, dw=wt.grad(), db=bt.grad(), gcb=cb[-1].grad())


def _launch(ctx, conv, wd, bd, x, y, classes, want_cb=True, tick=None, fuses=(None, None, None), spec=SIMPLE, kind=2):
    from taper_amd import hip
    n = x.shape[0]
    K, c_last, hw = _k_of(spec)
    bufs = [(ctx.upload(cw), ctx.upload(cb)) for cw, cb in conv]
    stages, ns = hip.conv_stages([(dw, db, c_out, post) for (dw, db), (_, c_out, post) in zip(bufs, spec)])
    assert hip.hip.th_conv_chain_head_supported(1, 28, 28, C.cast(stages, C.c_void_p), ns, classes) == kind
    ymap, dl, rs = ctx.empty(n * K), ctx.empty(n * 16), ctx.empty(n * 2)
    cbp = ctx.empty(n * c_last) if want_cb else None
    yd = ctx.upload(y)
    head = ChainHead(int(wd), int(bd) if bd is not None else None, int(yd), classes, int(dl), int(rs), int(cbp) if cbp is not None else None,
                     int(tick) if tick is not None else None)
    ctx.call("th_conv_chain_head_fwd", ctx.upload(x), C.cast(stages, C.c_void_p), ns, ymap, n, 1, 28, 28, C.byref(head))
    dw, db, gcb, loss, nc = ctx.empty(classes * K), ctx.empty(classes), (ctx.empty(c_last) if want_cb else None), ctx.empty(1), ctx.empty(1)
    f = [C.byref(v) if v is not None else None for v in fuses]
    ctx.call("th_wide_head_grads", ymap, dl, rs, cbp, n, K, classes, c_last, dw, db, gcb, loss, nc, None, 0, None, 0, f[0], f[1], f[2])
    ctx.sync()
    return dict(pooled=ctx.download(ymap, (n, c_last, hw, hw)), dl=ctx.download(dl, (n, 16)), rs=ctx.download(rs, (n, 2)),
                cbp=ctx.download(cbp, (n, c_last)) if want_cb else None, dw=ctx.download(dw, (classes, K)), db=ctx.download(db, classes),
                gcb=ctx.download(gcb, c_last) if want_cb else None, loss=float(ctx.download(loss, 1)[0]), nc=float(ctx.download(nc, 1)[0]))


def _close_with_mask_flips(got, ref, dx_max, what):
    """the conv bias gradient sums dX over the pooled outputs that are > 0: where a pre-activation sits within rounding of zero the two
    implementations may disagree about ONE element's mask, i.e. by one |dX| -- allow a handful of channels to be off by a few such elements"""
    ref = np.asarray(ref, np.float32).reshape(np.shape(got))
    diff = np.abs(got - ref)
    off = diff > RTOL * np.abs(ref) + RTOL * float(np.abs(ref).max()) + 1e-9
    assert off.sum() <= max(3, off.size // 20) and diff.max() <= 4 * dx_max, (what, int(off.sum()), float(diff.max()), dx_max)


def _close(got, ref, what):
    ref = np.asarray(ref, np.float32).reshape(np.shape(got))
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=RTOL * float(np.abs(ref).max()) + 1e-9, err_msg=what)


@pytest.mark.parametrize("n,classes", [(256, 10), (96, 10), (1, 10), (37, 10), (300, 10), (256, 16), (50, 3), (130, 12)])
def test_chain_head_launches_match_the_oracle_tape(ctx, O, n, classes):
    rng = np.random.default_rng(1000 * classes + n)
    conv, w, b = _model(rng, classes)
    x = rng.integers(0, 256, (n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)
    y = rng.integers(0, classes, n).astype(np.float32)
    ref = _oracle_step(O, conv, w, b, x, y)
    got = _launch(ctx, conv, ctx.upload(w), ctx.upload(b), x, y, classes)
    _close(got["pooled"], ref["pooled"], "pooled map")
    _close(got["dl"][:, :classes], ref["dl"], "dlogits")
    assert not got["dl"][:, classes:].any()
    _close(got["rs"][:, 0], ref["nll"], "nll per row")
    # a hit can flip where two logits sit within rounding of each other: allow one row per 256
    assert np.abs(got["rs"][:, 1] - ref["hit"]).sum() <= max(1, n // 256)
    assert got["loss"] == pytest.approx(ref["loss"], rel=RTOL, abs=1e-6)
    assert abs(got["nc"] - ref["hit"].sum()) <= max(1, n // 256)
    _close(got["dw"], ref["dw"], "dW")
    _close(got["db"], ref["db"], "db")
    dx_max = float(np.abs(ref["dl"] @ w).max())
    _close_with_mask_flips(got["gcb"], ref["gcb"], dx_max, "conv bias gradient")
    _close_with_mask_flips(got["cbp"].sum(axis=0), ref["gcb"], dx_max, "per-image channel sums")


GENERIC_NETS = [[(1, 32, 1), (32, 64, 1)],                  # the simple CNN's front on the run-time-described kernel (compiled instances off)
                [(1, 16, 1), (16, 32, 1)],                  # k = 32 * 49
                [(1, 16, 0), (16, 32, 1), (32, 48, 1)],     # three stages, k = 48 * 49
                [(1, 32, 1), (32, 16, 0), (16, 16, 0)] ]    # ends in a conv without a pool: no chain + head (0)


@pytest.mark.parametrize("net", range(4))
@pytest.mark.parametrize("n,classes", [(256, 10), (37, 16), (300, 3)])
def test_chain_head_on_the_run_time_described_kernel(ctx, O, net, n, classes):
    """th_conv_chain_head_fwd where no instance is compiled: the kernel that takes its stages as arguments runs the same classifier rows"""
    from taper_amd import hip
    spec = GENERIC_NETS[net]
    rng = np.random.default_rng(77 * net + n + classes)
    hip.hip.th_debug_set_chain_generic(1)
    try:
        if net == 3:
            conv, w, b = _model(rng, classes, spec[:1])
            bufs = [(ctx.upload(np.zeros((co, ci, 3, 3), np.float32)), ctx.upload(np.zeros(co, np.float32))) for ci, co, _ in spec]
            stages, ns = hip.conv_stages([(dw, db, c_out, post) for (dw, db), (_, c_out, post) in zip(bufs, spec)])
            assert hip.hip.th_conv_chain_head_supported(1, 28, 28, C.cast(stages, C.c_void_p), ns, classes) == 0
            return
        conv, w, b = _model(rng, classes, spec)
        x = rng.integers(0, 256, (n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)
        y = rng.integers(0, classes, n).astype(np.float32)
        ref = _oracle_step(O, conv, w, b, x, y, spec)
        got = _launch(ctx, conv, ctx.upload(w), ctx.upload(b), x, y, classes, spec=spec, kind=3)
    finally:
        hip.hip.th_debug_set_chain_generic(0)
    _close(got["pooled"], ref["pooled"], "pooled map")
    _close(got["dl"][:, :classes], ref["dl"], "dlogits")
    assert not got["dl"][:, classes:].any()
    _close(got["rs"][:, 0], ref["nll"], "nll per row")
    assert np.abs(got["rs"][:, 1] - ref["hit"]).sum() <= max(1, n // 256)
    assert got["loss"] == pytest.approx(ref["loss"], rel=RTOL, abs=1e-6)
    _close(got["dw"], ref["dw"], "dW")
    _close(got["db"], ref["db"], "db")
    dx_max = float(np.abs(ref["dl"] @ w).max())
    _close_with_mask_flips(got["gcb"], ref["gcb"], dx_max, "conv bias gradient")


def test_chain_head_pooled_map_is_bit_identical_to_the_plain_chain(ctx):
    """the conv rows of the launch are th_conv_chain_fwd's: same map, bit for bit"""
    from taper_amd import hip
    rng = np.random.default_rng(7)
    n = 256
    conv, w, b = _model(rng, 10)
    x = rng.integers(0, 256, (n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)
    y = rng.integers(0, 10, n).astype(np.float32)
    got = _launch(ctx, conv, ctx.upload(w), ctx.upload(b), x, y, 10, want_cb=False)
    bufs = [(ctx.upload(cw), ctx.upload(cb)) for cw, cb in conv]
    stages, ns = hip.conv_stages([(dw, db, c_out, post) for (dw, db), (_, c_out, post) in zip(bufs, SIMPLE)])
    ymap = ctx.empty(n * K)
    ctx.call("th_conv_chain_fwd", ctx.upload(x), C.cast(stages, C.c_void_p), ns, ymap, None, n, 1, 28, 28)
    ctx.sync()
    np.testing.assert_array_equal(got["pooled"], ctx.download(ymap, (n, 64, 7, 7)))
    assert got["gcb"] is None and got["cbp"] is None


def _adam_ref(O, p0, g, lr, t, wd=1e-4):
    pt = O.Tensor(p0).requires_grad()
    opt = O.Adam([pt], lr, None, None, wd)
    for _ in range(t - 1):
        opt.step()
    pt.set_grad(g)
    opt.step()
    return pt.data(), opt.m(0), opt.v(0)


@pytest.mark.parametrize("n", [256, 100])
def test_chain_head_adam_epilogues_and_tick(ctx, O, n):
    """launch 1 opens the optimizer step (t += 1, optim.rs:84); launch 2's epilogues apply optim.rs:99-110 at that t to W, b and the conv bias"""
    from taper_amd.hip import AdamFuse
    rng = np.random.default_rng(n)
    conv, w, b = _model(rng, 10)
    x = rng.integers(0, 256, (n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)
    y = rng.integers(0, 10, n).astype(np.float32)
    ref = _oracle_step(O, conv, w, b, x, y)
    lr = 1e-2
    wd_, bd_, cbd_ = ctx.upload(w), ctx.upload(b), ctx.upload(conv[1][1])
    mom = {k: (ctx.zeros(sz), ctx.zeros(sz)) for k, sz in (("w", w.size), ("b", b.size), ("cb", 64))}
    tick, lrd = ctx.upload(np.array([4, 0], np.int32)), ctx.upload(np.array([lr], np.float32))
    fz = lambda p, k: AdamFuse(int(p), int(mom[k][0]), int(mom[k][1]), int(tick), int(lrd), 0.9, 0.999, 1e-8, 1e-4)
    # (the launches read W / b from the optimizer's storage; the conv rows read the conv bias from their own upload: it is only UPDATED here)
    got = _launch(ctx, conv, wd_, bd_, x, y, 10, tick=tick, fuses=(fz(wd_, "w"), fz(bd_, "b"), fz(cbd_, "cb")))
    assert ctx.download(tick, 2, np.int32)[0] == 5
    _close(got["dw"], ref["dw"], "dW")
    for name, p0, g, dev in (("w", w, ref["dw"], wd_), ("b", b, ref["db"], bd_), ("cb", conv[1][1], ref["gcb"], cbd_)):
        p_ref, m_ref, v_ref = _adam_ref(O, p0.reshape(-1), np.asarray(g, np.float32).reshape(-1), lr, 5)
        np.testing.assert_allclose(ctx.download(dev, p0.size), p_ref, rtol=RTOL, atol=lr * 2e-2, err_msg=name)
        margins.check(f"{name}_m", ctx.download(mom[name][0], p0.size), m_ref, BOUND_M)
        margins.check(f"{name}_v", ctx.download(mom[name][1], p0.size), v_ref, BOUND_V)


def test_chain_head_unsupported_and_errors(ctx):
    from taper_amd import hip
    from taper_amd._lib import TaperError
    rng = np.random.default_rng(3)
    conv, w, b = _model(rng, 10)
    bufs = [(ctx.upload(cw), ctx.upload(cb)) for cw, cb in conv]
    st, ns = hip.conv_stages([(dw, db, c_out, post) for (dw, db), (_, c_out, post) in zip(bufs, SIMPLE)])
    f = hip.hip.th_conv_chain_head_supported
    assert f(1, 28, 28, C.cast(st, C.c_void_p), ns, 10) == 2 and f(1, 28, 28, C.cast(st, C.c_void_p), ns, 17) == 0
    assert f(1, 32, 32, C.cast(st, C.c_void_p), ns, 10) == 0
    z = ctx.zeros(64)
    with pytest.raises(TaperError, match="classes <= 16"):
        ctx.call("th_wide_head_grads", z, z, z, None, 4, 16, 17, 0, z, z, None, z, None, None, 0, None, 0, None, None, None)
    with pytest.raises(TaperError, match="together"):
        ctx.call("th_wide_head_grads", z, z, z, z, 4, 16, 10, 4, z, z, None, z, None, None, 0, None, 0, None, None, None)


def _trainer_run(T, H, spec, x, y, batch, head, mode="graph"):
    T.set_conv_chain_head(head)
    try:
        hm = H.sequential(spec)
        opt = T.Adam(hm.parameters(), 1e-2, None, None, 1e-4)
        tr = T.Trainer(hm, opt, sample_shape=(1, 28, 28))
        if mode == "graph":
            ep = tr.run_epoch(T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False), T.Trainer.GRAPH)
            losses, nc = ep["losses"], ep["ncorrect"]
        else:
            losses, nc = [], []
            for s in range(len(y) // batch):
                l, a = tr.train_step(T.Tensor(x[s * batch:(s + 1) * batch]), T.Tensor(y[s * batch:(s + 1) * batch]))
                losses.append(l)
                nc.append(a * batch)
        from tests.test_gpu_full_size import last_conv_config_host
        cfg = last_conv_config_host()
        return losses, nc, [p.data() for p in hm.parameters()], opt.t(), cfg
    finally:
        T.set_conv_chain_head(True)


@pytest.mark.parametrize("batch", [256, 96, 128])
def test_simple_cnn_trainer_steps_take_the_two_launch_form_and_match_the_oracle(batch, mode="graph"):
    import taper_amd as T
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(batch)
    steps, lr = 3, 1e-2
    spec = backends.nonzero_biases(backends.cnn_simple(rng), rng)
    x, y = backends.mnist_like(rng, steps * batch)
    om = Orc.sequential(spec)
    oopt = Orc.m.Adam(om.parameters(), lr, None, None, 1e-4)
    ref = [om.train_step(oopt, x[s * batch:(s + 1) * batch], y[s * batch:(s + 1) * batch], (batch, 1, 28, 28)) for s in range(steps)]
    losses, nc, params, t, cfg = _trainer_run(T, H, spec, x, y, batch, True, mode)
    assert cfg["dma"] == 7 and cfg["ct"] == 2, cfg          # th_conv_chain_head_fwd ran in this process's step
    assert t == steps
    from tests import margins
    tag = f"cnn_simple_two_launch_step_b{batch}_3_adam_steps"
    margins.record(tag, "losses", losses, [r["loss"] for r in ref])
    # bounds = 2x the largest margin observed on MI355X (profiles/r03_parity_margins.json: losses 3.5e-6 of the largest, weights 4.0e-3 lr)
    np.testing.assert_allclose(losses, [r["loss"] for r in ref], rtol=0, atol=7e-6 * max(abs(r["loss"]) for r in ref))
    assert np.abs(np.asarray(nc) - np.asarray([r["acc"] * batch for r in ref])).max() <= 1.5
    for i, (hp, op) in enumerate(zip(params, om.parameters())):
        m = margins.record(tag, f"param{i}", hp, op.data(), lr=lr)
        assert m["err_over_lr"] <= 8e-3, (i, m)
    # ... and the same step with the classifier as its own launches (th_linear_xent_wide + the bias finish): same results within rounding
    l2, nc2, p2, t2, cfg2 = _trainer_run(T, H, spec, x, y, batch, False, mode)
    assert cfg2["dma"] == 6 and t2 == steps
    np.testing.assert_allclose(losses, l2, rtol=1e-5, atol=1e-6)
    for i, (a, b) in enumerate(zip(params, p2)):
        np.testing.assert_allclose(a, b, rtol=RTOL, atol=lr * 2e-2, err_msg=f"param {i}")


def _custom_cnn(rng, kind):
    c, lin = backends._conv, backends._lin
    pool = dict(kind="maxpool", kernel=(2, 2), stride=(2, 2))
    if kind == "pool_head":          # conv rows end in a pooled map, Flatten, Linear: generic chain + classifier rows, two launches
        return [c(rng, 1, 16), pool, c(rng, 16, 32), pool, dict(kind="flatten", start_dim=1), lin(rng, 32 * 49, 10)]
    if kind == "pool_mlp":           # ... followed by Linear + ReLU + Linear: generic chain, then the two-launch MLP tail
        return [c(rng, 1, 16), c(rng, 16, 16), pool, c(rng, 16, 48), pool, dict(kind="flatten", start_dim=1), lin(rng, 48 * 49, 64),
                dict(kind="relu"), lin(rng, 64, 10)]
    if kind == "conv_row_end":       # the conv rows END in a conv (no pool behind it): the chain writes that row's map; Flatten, Linear
        return [c(rng, 1, 16), pool, c(rng, 16, 32), pool, c(rng, 32, 16), dict(kind="flatten", start_dim=1), lin(rng, 16 * 49, 10)]
    # global average + three-layer classifier, other widths than the reference CNN's
    return [c(rng, 1, 16), pool, c(rng, 16, 32), c(rng, 32, 64), pool, c(rng, 64, 96), dict(kind="adaptive_avgpool", out=(1, 1)),
            dict(kind="flatten", start_dim=1), lin(rng, 96, 64), dict(kind="relu"), lin(rng, 64, 32), dict(kind="relu"), lin(rng, 32, 10)]


@pytest.mark.parametrize("kind,want_dma", [("pool_head", 7), ("pool_mlp", 6), ("gap_mlp3", 6), ("conv_row_end", 6)])
@pytest.mark.parametrize("batch", [256, 100])
def test_other_cnns_take_the_run_time_described_chain_and_match_the_oracle(kind, want_dma, batch):
    """Sequentials that are neither of the two compiled nets: the Trainer's captured step launches their conv rows as ONE kernel all the same
    (th_conv_chain_supported == 3) -- 3 Adam steps against the oracle's training loop"""
    import taper_amd as T
    from tests import margins
    from tests.test_gpu_full_size import last_conv_config_host
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(batch + len(kind))
    steps, lr = 3, 1e-2
    spec = backends.nonzero_biases(_custom_cnn(rng, kind), rng)
    x, y = backends.mnist_like(rng, steps * batch)
    om = Orc.sequential(spec)
    oopt = Orc.m.Adam(om.parameters(), lr, None, None, 1e-4)
    ref = [om.train_step(oopt, x[s * batch:(s + 1) * batch], y[s * batch:(s + 1) * batch], (batch, 1, 28, 28)) for s in range(steps)]
    hm = H.sequential(spec)
    opt = T.Adam(hm.parameters(), lr, None, None, 1e-4)
    tr = T.Trainer(hm, opt, sample_shape=(1, 28, 28))
    ep = tr.run_epoch(T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False), T.Trainer.GRAPH)
    cfg = last_conv_config_host()
    assert cfg["dma"] == want_dma and cfg["ct"] == 3, cfg          # a conv chain, id 3: the kernel that takes its stages as arguments
    assert opt.t() == steps
    tag = f"custom_cnn_{kind}_b{batch}_3_adam_steps"
    margins.record(tag, "losses", ep["losses"], [r["loss"] for r in ref])
    np.testing.assert_allclose(ep["losses"], [r["loss"] for r in ref], rtol=0, atol=1e-5 * max(abs(r["loss"]) for r in ref))
    assert np.abs(np.asarray(ep["ncorrect"]) - np.asarray([r["acc"] * batch for r in ref])).max() <= 1.5
    for i, (hp, op) in enumerate(zip(hm.parameters(), om.parameters())):
        m = margins.record(tag, f"param{i}", hp.data(), op.data(), lr=lr)
        assert m["err_over_lr"] <= 2e-2, (i, m)


@pytest.mark.parametrize("n,k,classes,conv_c", [(1, 16, 1, 4), (37, 100, 7, 5), (256, 3136, 10, 64), (513, 1000, 16, 40), (700, 48, 3, 16),
                                                 (4100, 130, 10, 13)])
def test_wide_head_grads_ragged_shapes(ctx, n, k, classes, conv_c):
    """th_wide_head_grads on its own: dW = dl^T X, db, loss, hit count, conv bias sums for batch / width / class / channel counts that are not
    multiples of any tile, against float64 sums (ops.rs:280-291, tensor.rs:686-691, loss.rs:164, 283)"""
    rng = np.random.default_rng(n + k + classes)
    x = rng.standard_normal((n, k)).astype(np.float32)
    dl = np.zeros((n, 16), np.float32)
    dl[:, :classes] = (rng.standard_normal((n, classes)) / n).astype(np.float32)
    rs = np.stack([rng.uniform(0, 3, n), rng.integers(0, 2, n)], axis=1).astype(np.float32)
    cbp = rng.standard_normal((n, conv_c)).astype(np.float32)
    dw, db, gcb, loss, nc = ctx.empty(classes * k), ctx.empty(classes), ctx.empty(conv_c), ctx.empty(1), ctx.empty(1)
    ctx.call("th_wide_head_grads", ctx.upload(x), ctx.upload(dl), ctx.upload(rs), ctx.upload(cbp), n, k, classes, conv_c, dw, db, gcb, loss, nc,
             None, 0, None, 0, None, None, None)
    ctx.sync()
    ref_dw = dl[:, :classes].astype(np.float64).T @ x.astype(np.float64)
    _close(ctx.download(dw, (classes, k)), ref_dw, "dW")
    _close(ctx.download(db, classes), dl[:, :classes].astype(np.float64).sum(axis=0), "db")
    _close(ctx.download(gcb, conv_c), cbp.astype(np.float64).sum(axis=0), "conv bias")
    assert float(ctx.download(loss, 1)[0]) == pytest.approx(float(rs[:, 0].astype(np.float64).sum() / n), rel=1e-5)
    assert float(ctx.download(nc, 1)[0]) == float(rs[:, 1].sum())


@pytest.mark.parametrize("n,classes", [(300, 10), (1024, 10), (700, 16)])
def test_walking_chain_with_the_classifier_rows_is_bit_identical_per_image(ctx, n, classes):
    """th_conv_chain_head_fwd on more images than CUs (r05: the workgroups walk the images, the classifier's weight registers are loaded once
    per workgroup): per image -- pooled map, dlogits row, {nll, hit}, the conv bias partials -- the bits of the one-workgroup-per-image
    launch.  dl and the row statistics carry 1 / n: the batches of <= 200 images are compared after rescaling by an exact power-of-two-free
    ratio only where n / m is exact in fp32, so the row records are compared through the same launch size instead: two walks of the same
    batch in another image order (reversed) must agree image by image."""
    rng = np.random.default_rng(n + classes)
    conv, w, b = _model(rng, classes)
    x = (rng.integers(0, 256, (n, 1, 28, 28))).astype(np.float32) / np.float32(255.0)
    y = rng.integers(0, classes, n).astype(np.float32)
    from taper_amd._lib import hip as lib
    wd, bd = ctx.upload(w), ctx.upload(b)
    lib.th_debug_set_chain_loop(1)              # (off by default: measured 3 - 5 % slower than one workgroup per image)
    try:
        a = _launch(ctx, conv, wd, bd, x, y, classes)
        r = _launch(ctx, conv, wd, bd, x[::-1].copy(), y[::-1].copy(), classes)      # image i here is image n - 1 - i there: another workgroup, another turn
    finally:
        lib.th_debug_set_chain_loop(-1)
    plain = _launch(ctx, conv, wd, bd, x, y, classes)                                # one workgroup per image, the same launch size: the same bits
    for k in ("pooled", "dl", "rs", "cbp"):
        np.testing.assert_array_equal(a[k], plain[k], err_msg=k)
    for k in ("pooled", "dl", "rs", "cbp"):
        np.testing.assert_array_equal(a[k], r[k][::-1], err_msg=k)
    # ... and the pooled maps against the one-workgroup-per-image launch (no 1 / n in them)
    for lo in range(0, n, 200):
        hi = min(n, lo + 200)
        s = _launch(ctx, conv, wd, bd, x[lo:hi], y[lo:hi], classes)
        np.testing.assert_array_equal(a["pooled"][lo:hi], s["pooled"])
    assert a["loss"] == pytest.approx(r["loss"], rel=1e-6)

"""LR schedulers, AdamW, Trainer::fit / Metrics, the text checkpoint,
bce_loss / cross_entropy_loss_onehot / Dropout and the XOR demo -- the host library (and its kernels)
against the oracle's restatement (oracle/train_extra.py, oracle's C bce_loss / Adam / SGD)."""
import math
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import train_extra as OX
from tests import backends

ROOT = Path(__file__).resolve().parent.parent
RTOL = 1e-4


# ------------------------------------------------------------------ host logic (no GPU needed)
RUST_DISPLAY_KATS = [  # `println!("{}", x)` for f32, Rust's documented shortest-round-trip positional form
    (1.0, "1"), (0.1, "0.1"), (-2.5, "-2.5"), (1e-7, "0.0000001"), (16777216.0, "16777216"), (0.3, "0.3"),
    (3.4028235e38, "340282350000000000000000000000000000000"), (1.17549435e-38, "0.000000000000000000000000000000000000011754944"),
    (1e-45, "0.000000000000000000000000000000000000000000001"), (-0.0, "-0"), (0.0, "0"), (123456.79, "123456.79"),
    (float("inf"), "inf"), (float("-inf"), "-inf"), (float("nan"), "NaN"), (1e10, "10000000000"), (0.001, "0.001"),
]


@pytest.mark.parametrize("value,text", RUST_DISPLAY_KATS)
def test_f32_display_known_answers(value, text):
    """pins the oracle's number format (and the host's) on Rust's own outputs"""
    import taper_amd as T
    assert OX.format_f32_display(value) == text
    assert T.format_f32(value) == text


def test_f32_display_host_equals_oracle_and_round_trips():
    import taper_amd as T
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2**32, 4000, dtype=np.uint64).astype(np.uint32)
    vals = bits.view(np.float32)
    vals = np.concatenate([vals[np.isfinite(vals)], np.float32([1e-30, 7e22, 0.5, 65504.0, 9.999999e-5])])
    for v in vals:
        s = T.format_f32(v)
        assert s == OX.format_f32_display(v), repr(v)
        assert "e" not in s and np.float32(s) == v


SCHEDULERS = [
    ("StepLR", (0.1, 3, 0.5)), ("StepLR", (0.01, 1, 0.9)), ("ExponentialLR", (0.05, 0.95)),
    ("CosineAnnealingLR", (0.1, 10, None)), ("CosineAnnealingLR", (0.01, 7, 1e-4)),
    ("ReduceLROnPlateau", (0.1, 0.5, 2, None, None)), ("ReduceLROnPlateau", (0.1, 0.1, 1, 1e-3, "max")),
]


@pytest.mark.parametrize("name,args", SCHEDULERS, ids=[f"{n}{i}" for i, (n, _) in enumerate(SCHEDULERS)])
def test_schedulers_match_oracle(name, args):
    """optim.rs:183-352: same f32 sequence as the restatement over 25 epochs of a wobbling metric"""
    import taper_amd as T
    h, o = getattr(T, name)(*args), getattr(OX, name)(*args)
    rng = np.random.default_rng(3)
    metric = 1.0
    assert h.get_lr() == o.get_lr()
    for epoch in range(25):
        metric = metric * 0.9 if epoch % 4 else metric * 1.05 + rng.uniform(0, 0.01)   # improves, stalls, regresses
        h.step(metric)
        o.step(metric)
        # exact for the multiplicative schedules; cosf (libm) vs numpy's float32 cos may differ by an ulp
        assert h.get_lr() == pytest.approx(o.get_lr(), rel=2e-6 if name.startswith("Cosine") else 2e-7, abs=0), f"epoch {epoch}"


def test_scheduler_closed_forms():
    """what the reference's formulas amount to (optim.rs:209-214, 275-283, 325-347)"""
    import taper_amd as T
    s = T.StepLR(0.1, 2, 0.5)
    lrs = []
    for _ in range(6):
        s.step()
        lrs.append(s.get_lr())
    np.testing.assert_allclose(lrs, [0.1, 0.05, 0.05, 0.025, 0.025, 0.0125], rtol=1e-6)
    c = T.CosineAnnealingLR(0.2, 4, 0.0)
    got = []
    for _ in range(4):
        c.step()
        got.append(c.get_lr())
    np.testing.assert_allclose(got, [0.2 * (1 + math.cos(math.pi * e / 4)) / 2 for e in (1, 2, 3, 4)], rtol=1e-5, atol=1e-8)
    p = T.ReduceLROnPlateau(0.1, 0.5, 2, 0.03)
    p.step(None)                       # None: nothing happens (optim.rs:326)
    for m in (1.0, 1.0, 1.0):          # first improves on inf, then two stalls -> one reduction
        p.step(m)
    assert p.get_lr() == pytest.approx(0.05)
    for m in (2.0, 2.0):
        p.step(m)
    assert p.get_lr() == pytest.approx(0.03)   # clamped at min_lr (optim.rs:340)


# ------------------------------------------------------------------ device paths
gpu = pytest.mark.gpu


def _close(a, b, rtol=RTOL, atol=1e-7):
    b = np.asarray(b)
    scale = float(np.abs(b).max()) if b.size else 0.0
    np.testing.assert_allclose(np.asarray(a).reshape(b.shape), b, rtol=rtol, atol=atol + rtol * 1e-2 * scale)


@gpu
@pytest.mark.parametrize("n,targets_grad", [(4, False), (1000, False), (257, True), (70000, False)])
def test_bce_loss_matches_oracle(n, targets_grad):
    """loss.rs:6-73 incl. the clamp at both ends and (optionally) the gradient towards the targets"""
    import taper_amd as T
    from oracle import oracle as O
    rng = np.random.default_rng(n)
    p = rng.uniform(0.0, 1.0, n).astype(np.float32)
    p[:4] = [0.0, 1.0, 1e-9, 1.0 - 1e-9][: min(4, n)]          # hit the clamp
    y = (rng.uniform(0, 1, n) > 0.5).astype(np.float32)
    if targets_grad:
        y = rng.uniform(0.05, 0.95, n).astype(np.float32)
    res = []
    for M in (O, T):
        M.Tape.reset()
        pt = M.Tensor(p, (n, 1)).requires_grad()
        yt = M.Tensor(y, (n, 1))
        if targets_grad:
            yt = yt.requires_grad()
        loss = M.bce_loss(pt, yt)
        (loss * M.Tensor(np.float32([1.7]), (1,))).backward()       # a non-unit upstream gradient
        res.append((loss.data()[0], pt.grad(), yt.grad() if targets_grad else None))
    (lo, gpo, gyo), (lh, gph, gyh) = res
    assert lh == pytest.approx(lo, rel=RTOL)
    _close(gph, gpo, rtol=2e-4)
    if targets_grad:
        _close(gyh, gyo, rtol=2e-4)


@gpu
def test_bce_loss_accumulates_into_existing_grad():
    import taper_amd as T
    rng = np.random.default_rng(0)
    p, y = rng.uniform(0.1, 0.9, 64).astype(np.float32), (rng.uniform(0, 1, 64) > 0.5).astype(np.float32)
    T.Tape.reset()
    pt, yt = T.Tensor(p, (64,)).requires_grad(), T.Tensor(y, (64,))
    l1 = T.bce_loss(pt, yt)
    l2 = T.bce_loss(pt, yt)
    (l1 + l2).backward()
    g2 = pt.grad().copy()
    pt.zero_grad()
    T.Tape.reset()
    T.bce_loss(pt, yt).backward()
    _close(g2, 2 * pt.grad())


@gpu
@pytest.mark.parametrize("b,c,normalised", [(8, 10, True), (33, 7, False), (256, 10, True)])
def test_cross_entropy_loss_onehot_matches_oracle(b, c, normalised):
    """loss.rs:201-245; with un-normalised targets the recorded gradient is still (softmax - t)/B"""
    import taper_amd as T
    rng = np.random.default_rng(b * c)
    x = (rng.standard_normal((b, c)) * 3).astype(np.float32)
    t = np.eye(c, dtype=np.float32)[rng.integers(0, c, b)] if normalised else rng.uniform(0, 1, (b, c)).astype(np.float32)
    T.Tape.reset()
    xt = T.Tensor(x, (b, c)).requires_grad()
    loss = T.cross_entropy_loss_onehot(xt, T.Tensor(t, (b, c)))
    (loss * T.Tensor(np.float32([0.5]), (1,))).backward()
    lo, go = OX.cross_entropy_loss_onehot(x, t, gloss=0.5)
    assert loss.data()[0] == pytest.approx(lo, rel=RTOL, abs=1e-6)
    _close(xt.grad(), go, rtol=2e-4)
    if normalised:   # same value as the index form (loss.rs:136-195)
        T.Tape.reset()
        li = T.cross_entropy_loss(T.Tensor(x, (b, c)), T.Tensor(t.argmax(1).astype(np.float32), (b,)))
        assert li.data()[0] == pytest.approx(lo, rel=RTOL, abs=1e-6)


@gpu
def test_onehot_shape_checks_are_errors():
    import taper_amd as T
    with pytest.raises(T.TaperError, match="shapes must match"):
        T.cross_entropy_loss_onehot(T.Tensor(np.zeros((2, 3), np.float32), (2, 3)), T.Tensor(np.zeros((2, 4), np.float32), (2, 4)))
    with pytest.raises(T.TaperError, match="must match in length"):
        T.bce_loss(T.Tensor(np.zeros(3, np.float32), (3,)), T.Tensor(np.zeros(4, np.float32), (4,)))


@gpu
def test_dropout_semantics():
    """nn.rs:798-822: identity in eval / p = 0, zeros at p = 1, otherwise input * mask with mask in
    {0, 1/(1-p)}, keep rate ~ 1-p, a fresh mask per call, gradient = upstream * mask"""
    import taper_amd as T
    rng = np.random.default_rng(2)
    x = rng.standard_normal((64, 512)).astype(np.float32)
    xt = T.Tensor(x, x.shape)
    for p in (0.0, 0.3, 0.5, 0.9, 1.0):
        d = T.Dropout(p)
        y = d.forward(xt).data()
        if p == 0.0:
            np.testing.assert_array_equal(y, x)
        elif p == 1.0:
            np.testing.assert_array_equal(y, np.zeros_like(x))
        else:
            m = d.last_mask().data()
            scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
            assert set(np.unique(m)) <= {np.float32(0.0), scale}
            keep = float((m > 0).mean())
            assert abs(keep - (1 - p)) < 4 * math.sqrt(p * (1 - p) / m.size) + 1e-3
            np.testing.assert_array_equal(y, OX.dropout_forward(x, m, p))
            m2 = (d.forward(xt), d.last_mask().data())[1]
            assert (m2 != m).mean() > 0.05                         # successive draws differ
        d.eval()
        np.testing.assert_array_equal(d.forward(xt).data(), x)   # eval(): identity (nn.rs:800-802)
    with pytest.raises(T.TaperError, match="between 0 and 1"):
        T.Dropout(1.5)
    T.Tape.reset()
    d = T.Dropout(0.4)
    xg = T.Tensor(x, x.shape).requires_grad()
    d.forward(xg).mean().backward()
    _close(xg.grad(), d.last_mask().data() / np.float32(x.size))


@gpu
def test_adamw_matches_oracle():
    """optim.rs:130-180: decay hits every weight (also one without a gradient), Adam runs with wd = 0"""
    import taper_amd as T
    from oracle import oracle as O
    rng = np.random.default_rng(9)
    shapes = [(16, 8), (16,), (5, 5)]
    init = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    hp = [T.Tensor(a, a.shape).requires_grad() for a in init]
    op = [O.Tensor(a, a.shape).requires_grad() for a in init]
    hopt = T.AdamW(hp, 1e-2, None, None, 0.05)
    oopt = O.Adam(op, 1e-2, None, None, None)
    oopt.decoupled_wd = 0.05
    for step in range(6):
        for i, (h, o) in enumerate(zip(hp, op)):
            if i == 2 and step % 2 == 0:
                continue                                            # grad None on some steps (Q8)
            g = rng.standard_normal(shapes[i]).astype(np.float32)
            h.set_grad(g)
            o.set_grad(g)
        hopt.step()
        OX.adamw_step(oopt, op)
        hopt.zero_grad()
        oopt.zero_grad()
        for h, o in zip(hp, op):
            np.testing.assert_allclose(h.data(), o.data(), rtol=RTOL, atol=1e-6, err_msg=f"step {step}")
    assert hopt.t() == oopt.t() == 6


def _small_problem(rng, n_train=1024, n_val=512):
    """a learnable 10-class problem in MNIST's shape: a fixed random template per class plus noise"""
    templates = (rng.uniform(0, 1, (10, 784)) > 0.7).astype(np.float32)

    def make(n):
        y = rng.integers(0, 10, n)
        x = np.clip(0.6 * templates[y] + rng.uniform(0, 0.4, (n, 784)), 0, 1).astype(np.float32)
        flip = rng.uniform(0, 1, n) < 0.1                      # 10 % label noise: accuracy tops out near 0.9,
        y = np.where(flip, rng.integers(0, 10, n), y)          # so fit()'s 99 % early stop stays out of the way
        return x, y.astype(np.float32)
    return make(n_train), make(n_val)


@gpu
@pytest.mark.parametrize("graph", [True, False], ids=["graph", "eager"])
def test_fit_with_scheduler_and_metrics(graph, capfd):
    """train.rs:175-261: per-epoch train + evaluate, scheduler.step(Some(val_loss)) -> optimizer lr,
    Metrics vectors and their printed forms"""
    import taper_amd as T
    rng = np.random.default_rng(21)
    (xtr, ytr), (xva, yva) = _small_problem(rng)
    H = backends.get("hip")
    model = H.sequential(backends.mlp_baseline(rng))
    opt = T.Adam(model.parameters(), 2e-3, None, None, 1e-4)
    sched = T.StepLR(2e-3, 2, 0.5)
    tr = T.Trainer(model, opt, scheduler=sched)
    train, val = T.DataLoader(T.MNISTDataset.from_host(xtr, ytr), 128, True), T.DataLoader(T.MNISTDataset.from_host(xva, yva, False), 128, False)
    m = tr.fit(train, val, 5, verbose=True, graph=graph)
    out = capfd.readouterr().out
    assert all(len(m[k]) == 5 for k in m)
    assert m["train_loss"][-1] < m["train_loss"][0] and m["val_acc"][-1] > 0.3          # it learns
    ref = OX.StepLR(2e-3, 2, 0.5)
    OX.fit_schedule(list(m["val_loss"]), ref, 2e-3)
    assert opt.get_lr() == pytest.approx(ref.get_lr(), rel=1e-6) == pytest.approx(2e-3 * 0.25, rel=1e-6)
    md = {k: [float(v) for v in m[k]] for k in m}
    assert tr.metrics_text() == OX.metrics_last_line(md)
    assert tr.metrics_text(summary=True) == OX.metrics_summary(md)
    assert "Starting training for 5 epochs" in out and "Epoch 5 - Train Loss:" in out and "Learning Rate: 0.000500" in out
    assert "Training Summary:" in out
    # evaluate() agrees with the metrics fit recorded for the last epoch
    ev = tr.run_epoch(val, T.Trainer.EVAL)
    assert ev["avg_loss"] == pytest.approx(m["val_loss"][-1], rel=1e-6)


@gpu
def test_fit_stops_early_at_99_percent(capfd):
    """train.rs:247-250"""
    import taper_amd as T
    rng = np.random.default_rng(4)
    x = np.zeros((256, 784), np.float32)
    y = rng.integers(0, 10, 256).astype(np.float32)
    x[np.arange(256), (y * 70).astype(int)] = 1.0                                   # one hot pixel per class: trivially separable
    H = backends.get("hip")
    model = H.sequential(backends.mlp_baseline(rng))
    tr = T.Trainer(model, T.Adam(model.parameters(), 1e-2, None, None, None))
    ds = T.MNISTDataset.from_host(x, y)
    m = tr.fit(T.DataLoader(ds, 64, True), T.DataLoader(ds, 64, False), 40, verbose=False)
    assert 1 <= len(m["val_acc"]) < 40 and m["val_acc"][-1] > 0.99
    assert "Reached 99% validation accuracy! Stopping early." in capfd.readouterr().out


@gpu
def test_checkpoint_text_format_and_round_trip(tmp_path):
    """train.rs:264-292 byte for byte (vs the restated writer), then load -> identical parameters;
    with the optimizer state restored, training continues exactly as if it had never stopped"""
    import taper_amd as T
    rng = np.random.default_rng(8)
    H = backends.get("hip")
    spec = backends.nonzero_biases(backends.mlp_example(rng), rng)
    (xtr, ytr), _ = _small_problem(rng, 512, 8)

    def make():
        model = H.sequential(spec)
        opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
        return model, opt, T.Trainer(model, opt), T.DataLoader(T.MNISTDataset.from_host(xtr, ytr), 64, False)

    model, opt, tr, loader = make()
    tr.run_epoch(loader, T.Trainer.GRAPH)
    ck, st = tmp_path / "model.ckpt", tmp_path / "adam.ckpt"
    tr.save_checkpoint(ck)
    tr.save_optimizer_state(st)
    params = [(tuple(p.shape()), p.data()) for p in model.parameters()]
    assert ck.read_text() == OX.checkpoint_text(params)
    head = st.read_text().split("\n", 1)[0].split()
    assert head[:2] == ["adam", "8"] and head[2:] == [OX.format_f32_display(v) for v in (1e-3, 0.9, 0.999, 1e-8, 1e-4)]
    ref = tr.run_epoch(loader, T.Trainer.GRAPH)                      # the uninterrupted run: one more epoch

    model2, opt2, tr2, loader2 = make()                              # fresh process-equivalent: same init, no training
    tr2.load_checkpoint(ck)
    tr2.load_optimizer_state(st)
    for (shape, data), p in zip(params, model2.parameters()):
        np.testing.assert_array_equal(p.data(), data)               # text round trip is exact
    assert opt2.t() == 8
    res = tr2.run_epoch(loader2, T.Trainer.GRAPH)
    np.testing.assert_array_equal(res["losses"], ref["losses"])      # bit-identical continuation
    for a, b in zip(model.parameters(), model2.parameters()):
        np.testing.assert_array_equal(a.data(), b.data())
    with pytest.raises(T.TaperError, match="shape mismatch|count mismatch"):
        other = H.sequential(backends.mlp_baseline(rng))
        T.Trainer(other, T.Adam(other.parameters(), 1e-3)).load_checkpoint(ck)


@gpu
def test_xor_training_trajectory_matches_oracle():
    """src/main.rs: Linear(2,4) - Sigmoid - Linear(4,1) - Sigmoid, bce_loss, SGD(0.1): same losses as the
    oracle step by step from the same initial weights"""
    import taper_amd as T
    from oracle import oracle as O
    rng = np.random.default_rng(1)
    x = np.float32([[0, 0], [0, 1], [1, 0], [1, 1]])
    y = np.float32([[0], [1], [1], [0]])
    w1, b1 = rng.uniform(-1, 1, (4, 2)).astype(np.float32), np.zeros(4, np.float32)
    w2, b2 = rng.uniform(-0.7, 0.7, (1, 4)).astype(np.float32), np.zeros(1, np.float32)
    losses = []
    for M in (O, T):
        ps = [M.Tensor(a, a.shape).requires_grad() for a in (w1, b1, w2, b2)]
        opt = M.SGD(ps, 0.10, None)
        ls = []
        for it in range(300):
            M.Tape.reset()
            xt, yt = M.Tensor(x, (4, 2)), M.Tensor(y, (4, 1))
            h = xt.matmul(ps[0].transpose()).add_broadcast(ps[1]).sigmoid()       # Linear::forward (nn.rs:54-60) + Sigmoid
            yhat = h.matmul(ps[2].transpose()).add_broadcast(ps[3]).sigmoid()
            loss = M.bce_loss(yhat, yt)
            loss.backward()
            opt.step()
            opt.zero_grad()
            ls.append(float(loss.data()[0]))
        losses.append(ls)
    np.testing.assert_allclose(losses[1], losses[0], rtol=2e-4)
    assert losses[1][-1] < losses[1][0]


@gpu
def test_xor_example_learns_xor():
    """the C++ counterpart of src/main.rs, all 50 000 iterations"""
    exe = ROOT / "examples" / "_build" / "xor"
    subprocess.check_call(["make", "-s", "-C", str(ROOT / "examples")])   # no-op when current; never run a stale binary
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "iteration    0: Loss = " in out.stdout and "learned XOR" in out.stdout


# ---- PTQ storage codecs (src/tensor.rs:2110-2288) ------------------------------------------------------------------
def _codec_inputs():
    rng = np.random.default_rng(77)
    special = np.array([0.0, -0.0, 1.0, -1.0, 65504.0, 65520.0, 70000.0, -70000.0, 6.1e-5, 6.0e-5, 5.96e-8, 2.9e-8, 1e-10, np.inf, -np.inf,
                        np.nan, 1.0009765625, 1.00048828125, 1.000732421875, 2047.5, 2047.75, 0.333333343, 3.14159274], np.float32)
    # values just below a power of two: the half-up rounding carries out of the mantissa (the OR-ed carry quirk)
    carry = np.array([1.9998779296875, 3.99993896484375, 0.99993896484375, 1023.99993896484375], np.float32)
    return np.concatenate([special, carry, rng.standard_normal(5000).astype(np.float32) * 10, rng.uniform(-7e4, 7e4, 2000).astype(np.float32),
                           (rng.standard_normal(2000) * 1e-5).astype(np.float32)])


def test_f16_codec_oracle_against_numpy_half():
    """the restatement agrees with IEEE round-to-nearest-even wherever the reference's half-up / truncating shortcuts do
    (everything but exact ties, denormal results and mantissa carries), and round-trips every half exactly"""
    x = _codec_inputs()
    h = OX.f32_to_f16_bits(x)
    fin = np.isfinite(x) & (np.abs(x) >= 6.2e-5) & (np.abs(x) < 65504)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).view(np.uint16)
    low13 = x.view(np.uint32) & 0x1FFF
    carry = ((x.view(np.uint32) & 0x7FFFFF) + 0x1000) >> 23 != 0
    plain = fin & (low13 != 0x1000) & ~carry
    np.testing.assert_array_equal(h[plain], ref[plain])
    allh = np.arange(65536, dtype=np.uint16)
    back = OX.f16_bits_to_f32(allh)
    ieee = allh.view(np.float16).astype(np.float32)
    nan = np.isnan(ieee)
    np.testing.assert_array_equal(back[~nan].view(np.uint32), ieee[~nan].view(np.uint32))     # every non-NaN half decodes exactly
    assert np.isnan(back[nan]).all()
    # the quirk: 1.99988 rounds up out of the mantissa; the carry (0x400) is OR-ed into exponent field 15 << 10, where that bit is
    # already set -> the result is 0x3C00 = 1.0 instead of 2.0 (tensor.rs:2234-2237)
    assert OX.f32_to_f16_bits(np.float32(1.9998779296875)) == 0x3C00
    assert OX.f32_to_f16_bits(np.float32(3.99993896484375)) == 0x4400              # exponent 16: bit 10 clear, the OR happens to carry correctly
    normals = allh[((allh >> 10) & 0x1F != 0) & ((allh >> 10) & 0x1F != 0x1F)]
    np.testing.assert_array_equal(OX.f32_to_f16_bits(OX.f16_bits_to_f32(normals)), normals)  # exact halves survive the round trip


def test_int8_codec_oracle_properties():
    rng = np.random.default_rng(5)
    x = rng.standard_normal(4000).astype(np.float32) * 3
    q, scale, zp, mn = OX.quantize_int8(x)
    assert zp == -128 and q.min() == -128 and q.max() == 127 and mn == pytest.approx(float(x.min()))
    assert scale == pytest.approx((float(x.max()) - float(x.min())) / 255, rel=1e-6)
    assert np.abs(OX.dequantize_int8(q, scale, zp, mn) - x).max() <= scale * 0.5001
    q2, s2, _, m2 = OX.quantize_int8(np.full(7, 2.5, np.float32))                       # constant tensor: range widened by 0.1 either way
    assert m2 == pytest.approx(2.4) and s2 == pytest.approx(0.2 / 255, rel=1e-5) and (q2 == q2[0]).all()


@gpu
def test_ptq_codecs_match_oracle_bit_for_bit():
    from taper_amd import hip
    ctx = hip.Ctx(0)
    x = _codec_inputs()
    n = x.size
    h = ctx.empty((n + 1) // 2)                                                          # n uint16 in float-sized storage
    ctx.call("th_f32_to_f16", ctx.upload(x), h, n)
    hb = ctx.download(h, ((n + 1) // 2,), np.float32).view(np.uint16)[:n]
    np.testing.assert_array_equal(hb, OX.f32_to_f16_bits(x))
    allh = np.arange(65536, dtype=np.uint16)
    back = ctx.empty(65536)
    ctx.call("th_f16_to_f32", ctx.upload(allh.view(np.float32)), back, 65536)
    got, ref = ctx.download(back, (65536,)), OX.f16_bits_to_f32(allh)
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    for data in (x[np.isfinite(x)], x, np.full(9, -1.25, np.float32), (np.random.default_rng(3).standard_normal(100000) * 4).astype(np.float32)):
        m = data.size
        q, params = ctx.empty((m + 3) // 4), ctx.empty(2)
        ctx.call("th_quantize_int8", ctx.upload(data), q, m, params)
        qb = ctx.download(q, ((m + 3) // 4,), np.float32).view(np.int8)[:m]
        mn, scale = ctx.download(params, (2,))
        rq, rs, rzp, rmn = OX.quantize_int8(data)
        assert float(mn) == rmn and float(scale) == rs
        np.testing.assert_array_equal(qb, rq)
        y = ctx.empty(m)
        ctx.call("th_dequantize_int8", q, y, m, float(scale), -128, float(mn))
        np.testing.assert_array_equal(ctx.download(y, (m,)).view(np.uint32), OX.dequantize_int8(rq, rs, rzp, rmn).view(np.uint32))
    ctx.close()


@gpu
def test_two_loaders_over_one_dataset_do_not_share_captured_graphs():
    """(r01 advisor) the captured steps bake in the loader's device index vector: a second DataLoader over the SAME dataset with the same
    batch size has its own shuffle order and must not replay the first loader's graphs -- graph epochs through alternating loaders
    equal the eager epochs through the same loaders"""
    import taper_amd as T
    rng = np.random.default_rng(12)
    H = backends.get("hip")
    spec = backends.nonzero_biases(backends.mlp_baseline(rng), rng)
    (x, y), _ = _small_problem(rng, 640, 8)
    out = []
    for mode in (T.Trainer.EAGER, T.Trainer.GRAPH):
        model = H.sequential(spec)
        opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
        tr = T.Trainer(model, opt)
        ds = T.MNISTDataset.from_host(x, y)
        la, lb = T.DataLoader(ds, 64, True, seed=1), T.DataLoader(ds, 64, True, seed=2)       # different orders
        losses = [tr.run_epoch(l, mode)["losses"] for l in (la, lb, la, lb)]
        out.append((np.concatenate(losses), [p.data() for p in model.parameters()]))
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=3e-4, atol=1e-5)
    assert not np.allclose(out[0][0][:10], out[0][0][10:20])          # the two loaders really see different batches
    for a, b in zip(out[0][1], out[1][1]):
        np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-3 * 5e-2)


@gpu
def test_optimizer_state_load_checks_hyperparameters_and_rerecords(tmp_path):
    """(r01 advisor) beta1 / beta2 / eps of a state file must match the optimizer; weight decay is a kernel argument by value, so loading a
    state with another decay re-records the captured steps -- the continuation equals an optimizer built with that decay"""
    import taper_amd as T
    rng = np.random.default_rng(13)
    H = backends.get("hip")
    spec = backends.nonzero_biases(backends.mlp_baseline(rng), rng)
    (x, y), _ = _small_problem(rng, 256, 8)

    def make(wd, beta1=None):
        model = H.sequential(spec)
        opt = T.Adam(model.parameters(), 1e-3, (beta1, 0.999) if beta1 else None, None, wd)
        return model, opt, T.Trainer(model, opt), T.DataLoader(T.MNISTDataset.from_host(x, y), 64, False)

    # a run with decay 1e-2 from the start, saved after one epoch
    m1, o1, t1, l1 = make(1e-2)
    t1.run_epoch(l1, T.Trainer.GRAPH)
    ck, st = tmp_path / "m.ckpt", tmp_path / "o.ckpt"
    t1.save_checkpoint(ck)
    t1.save_optimizer_state(st)
    ref = t1.run_epoch(l1, T.Trainer.GRAPH)["losses"]
    # a trainer built with decay 1e-4 that has already captured its graphs, then loads that state: must continue with 1e-2
    m2, o2, t2, l2 = make(1e-4)
    t2.run_epoch(l2, T.Trainer.GRAPH)
    t2.load_checkpoint(ck)
    t2.load_optimizer_state(st)
    got = t2.run_epoch(l2, T.Trainer.GRAPH)["losses"]
    np.testing.assert_array_equal(got, ref)
    for a, b in zip(m1.parameters(), m2.parameters()):
        np.testing.assert_array_equal(a.data(), b.data())
    # other betas: refused
    m3, o3, t3, l3 = make(1e-2, beta1=0.8)
    with pytest.raises(T.TaperError, match="beta1/beta2/eps"):
        t3.load_optimizer_state(st)


@gpu
def test_second_optimizer_over_the_same_parameters_shares_the_arenas():
    """(r01 advisor) the reference lets several optimizers hold the same tensors (Adam re-created with another lr): the second one adopts
    the first one's flat arenas -- both keep stepping the storage every handle points at; a partly overlapping list is refused"""
    import taper_amd as T
    rng = np.random.default_rng(14)
    H = backends.get("hip")
    spec = backends.nonzero_biases(backends.mlp_baseline(rng), rng)
    x, y = backends.mnist_like(rng, 64)
    model = H.sequential(spec)
    a = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
    w0 = [p.data() for p in model.parameters()]
    b = T.Adam(model.parameters(), 1e-2, None, None, 0.0)            # re-created optimizer: same list, same order
    tr_a, tr_b = T.Trainer(model, a), T.Trainer(model, b)
    tr_a.train_step(T.Tensor(x), T.Tensor(y))
    w1 = [p.data() for p in model.parameters()]
    assert all(not np.array_equal(u, v) for u, v in zip(w0, w1))     # the FIRST optimizer still moves the model's storage
    tr_b.train_step(T.Tensor(x), T.Tensor(y))
    w2 = [p.data() for p in model.parameters()]
    assert all(not np.array_equal(u, v) for u, v in zip(w1, w2))     # ... and so does the second
    assert a.t() == 1 and b.t() == 1                                 # each keeps its own step counter / moments
    with pytest.raises(T.TaperError, match="another optimizer's flat arena"):
        T.Adam(model.parameters()[:2] + [T.Tensor(np.zeros(4, np.float32)).requires_grad()], 1e-3)

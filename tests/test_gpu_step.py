"""Whole-step parity: identical weights + batch ->
loss, logits, every parameter gradient (incl. which ones are None, quirk Q2)
and the post-Adam weights must match the CPU oracle; the hipGraph-replayed
epoch must match the eager epoch; size-independent properties at full size."""
import zlib

import numpy as np
import pytest

from tests import margins

from tests import backends

pytestmark = pytest.mark.gpu

RTOL = 1e-4
BOUND_FULL_BWD_GRAD = 4.3e-6   # gradients, of the tensor's scale: observed <= 2.1e-6 (2x the r04 observation, profiles/r04_parity_margins.json)
BOUND_FULL_BWD_GRAD_LARGE = 2.8e-5   # batches 40 / 136 (sums of more terms that cancel: smaller scale): observed <= 1.38e-5 of the tensor's scale (2x, gpurun_out/parity_margins.json)
BOUND_EPOCH_LOSS = 2.7e-6   # per-step losses, of the largest: observed 1.3e-6 (2x the r04 observation, profiles/r04_parity_margins.json)
BOUND_M = 2.2e-6   # max |m - m_oracle| / max |m_oracle| after 3 steps: observed 1.06e-6 (2x the r04 observation, profiles/r04_parity_margins.json)
BOUND_V = 1.8e-6   # observed 8.9e-7 (2x the r04 observation, profiles/r04_parity_margins.json)


def _compare_grads(h_grads, o_grads, names=None):
    assert len(h_grads) == len(o_grads)
    for i, (hg, og) in enumerate(zip(h_grads, o_grads)):
        assert (hg is None) == (og is None), f"param {i}: grad None-ness differs (hip {hg is None}, oracle {og is None})"
        if og is not None:
            scale = float(np.abs(og).max())
            np.testing.assert_allclose(hg, og, rtol=RTOL, atol=RTOL * 1e-1 * scale + 1e-8, err_msg=f"param {i}")


MODELS = {
    "mlp_baseline": (backends.mlp_baseline, None),       # BASELINE configs[0/1]
    "mlp_example": (backends.mlp_example, None),         # examples/train_mnist.rs
    "cnn_simple": (backends.cnn_simple, (1, 28, 28)),    # BASELINE configs[2]
    "cnn_reference": (backends.cnn_reference, (1, 28, 28)),  # examples/train_mnist_cnn.rs
}


@pytest.mark.parametrize("name,batch", [("mlp_baseline", 64), ("mlp_baseline", 32), ("mlp_baseline", 1), ("mlp_example", 256),
                                        ("mlp_example", 96), ("cnn_simple", 16), ("cnn_reference", 8), ("cnn_reference", 3)])
@pytest.mark.parametrize("fuse", [True, False])
def test_forward_backward_parity(name, batch, fuse):
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    # a stable seed (str hashes change per process): the inputs must be the same on every run -- a pre-activation
    # within rounding of 0 flips its ReLU mask between two summation orders, a discontinuity no tolerance covers
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000 + batch)
    builder, sample = MODELS[name]
    spec = backends.nonzero_biases(builder(rng), rng)
    x, y = backends.mnist_like(rng, batch)
    x_shape = (batch, 784) if sample is None else (batch,) + sample
    hm, om = H.sequential(spec, fuse=fuse), Orc.sequential(spec)
    h_loss, h_acc, h_logits, h_grads = H.forward_backward(hm, x, y, x_shape)
    o_loss, o_acc, o_logits, o_grads = Orc.forward_backward(om, x, y, x_shape)
    np.testing.assert_allclose(h_logits, o_logits, rtol=RTOL, atol=RTOL * float(np.abs(o_logits).max()))
    assert abs(h_loss - o_loss) <= RTOL * max(1.0, abs(o_loss))
    assert h_acc == pytest.approx(o_acc, abs=1e-6)
    _compare_grads(h_grads, o_grads)
    if name.startswith("cnn"):
        # Q2: faithful mode -- conv weights never get gradients; only the LAST conv's bias does
        assert h_grads[0] is None and o_grads[0] is None


@pytest.mark.parametrize("name,batch", [("cnn_simple", 6), ("cnn_reference", 4)])
def test_full_backward_extension(name, batch):
    """full_backward mode (extension, not reference behaviour): every parameter gets a
    gradient and it matches the oracle's differentiable im2col chain."""
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(77 + batch)
    builder, sample = MODELS[name]
    spec = backends.nonzero_biases(builder(rng), rng)
    x, y = backends.mnist_like(rng, batch)
    x_shape = (batch,) + sample
    hm, om = H.sequential(spec, full_backward=True), Orc.sequential(spec, full_backward=True)
    try:
        h_loss, _, h_logits, h_grads = H.forward_backward(hm, x, y, x_shape)
    finally:
        H.m.set_full_backward(False)
    o_loss, _, o_logits, o_grads = Orc.forward_backward(om, x, y, x_shape)
    assert all(g is not None for g in o_grads)
    assert abs(h_loss - o_loss) <= RTOL * max(1.0, abs(o_loss))
    for i, (hg, og) in enumerate(zip(h_grads, o_grads)):
        assert hg is not None, f"param {i} has no grad in full_backward mode"
        margins.check(f"grad{i}", hg, og, BOUND_FULL_BWD_GRAD)


@pytest.mark.parametrize("name,batch", [("cnn_simple", 40), ("cnn_reference", 136)])
def test_full_backward_extension_large_batch(name, batch):
    """The same comparison at batches that take the kernels gated on the batch size: the one-image-per-workgroup layer kernels
    (>= 128 images), conv1's image-per-workgroup weight gradient (>= 32), the matrix-core weight gradient (>= 2048 pixels), the pool's
    scatter with the ReLU mask and the plane sums folded in (th_maxpool2d_relu_bwd, th_relu_bwd_plane_sums)."""
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(1234 + batch)
    builder, sample = MODELS[name]
    spec = backends.nonzero_biases(builder(rng), rng)
    x, y = backends.mnist_like(rng, batch)
    x_shape = (batch,) + sample
    hm, om = H.sequential(spec, full_backward=True), Orc.sequential(spec, full_backward=True)
    try:
        h_loss, _, h_logits, h_grads = H.forward_backward(hm, x, y, x_shape)
    finally:
        H.m.set_full_backward(False)
    o_loss, _, o_logits, o_grads = Orc.forward_backward(om, x, y, x_shape)
    assert abs(h_loss - o_loss) <= RTOL * max(1.0, abs(o_loss))
    for i, (hg, og) in enumerate(zip(h_grads, o_grads)):
        assert hg is not None and og is not None, f"param {i} has no grad in full_backward mode"
        margins.check(f"grad{i}", hg, og, BOUND_FULL_BWD_GRAD_LARGE)


@pytest.mark.parametrize("name,batch,lr", [("mlp_baseline", 64, 1e-3), ("mlp_example", 256, 1e-3), ("cnn_reference", 8, 1e-2)])
def test_training_steps_parity(name, batch, lr):
    """5 consecutive steps of examples/train_mnist.rs:89-121 (Adam, wd 1e-4): per-step
    loss / accuracy and the weights after every step."""
    import taper_amd as T
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(5 + batch)
    builder, sample = MODELS[name]
    spec = backends.nonzero_biases(builder(rng), rng)
    hm, om = H.sequential(spec), Orc.sequential(spec)
    hopt = T.Adam(hm.parameters(), lr, None, None, 1e-4)
    oopt = Orc.m.Adam(om.parameters(), lr, None, None, 1e-4)
    tr = T.Trainer(hm, hopt, sample_shape=sample)
    x_all, y_all = backends.mnist_like(rng, 5 * batch)
    for s in range(5):
        x, y = x_all[s * batch:(s + 1) * batch], y_all[s * batch:(s + 1) * batch]
        loss, acc = tr.train_step(T.Tensor(x), T.Tensor(y))
        r = om.train_step(oopt, x, y, (batch, 784) if sample is None else (batch,) + sample)
        assert abs(loss - r["loss"]) <= 2 * RTOL * max(1.0, abs(r["loss"])), f"step {s}"
        assert acc == pytest.approx(r["acc"], abs=1.5 / batch), f"step {s}"
        for i, (hp, op) in enumerate(zip(hm.parameters(), om.parameters())):
            od = op.data()
            # Adam normalises the step to ~lr per element, so |dw| ~ lr: compare on that scale
            np.testing.assert_allclose(hp.data(), od, rtol=RTOL, atol=lr * 2e-2, err_msg=f"step {s} param {i}")
    assert hopt.t() == 5


def test_adam_state_parity_after_steps():
    import taper_amd as T
    H, Orc = backends.get("hip"), backends.get("oracle")
    rng = np.random.default_rng(21)
    spec = backends.mlp_baseline(rng)
    hm, om = H.sequential(spec), Orc.sequential(spec)
    hopt, oopt = T.Adam(hm.parameters(), 1e-3, None, None, 1e-4), Orc.m.Adam(om.parameters(), 1e-3, None, None, 1e-4)
    tr = T.Trainer(hm, hopt)
    for s in range(3):
        x, y = backends.mnist_like(rng, 64)
        tr.train_step(T.Tensor(x), T.Tensor(y))
        om.train_step(oopt, x, y, (64, 784))
    m, v = hopt.moments()
    om_ = np.concatenate([oopt.m(i) for i in range(4)])
    ov_ = np.concatenate([oopt.v(i) for i in range(4)])
    margins.check("adam_m", m, om_, BOUND_M)
    margins.check("adam_v", v, ov_, BOUND_V)


@pytest.mark.parametrize("n,batch", [(640, 64), (1000, 64), (300, 128)])
def test_graph_epoch_equals_eager_epoch(n, batch):
    """The hipGraph-replayed epoch (train_epoch_graph) must produce the same per-step
    losses, counts and final weights as the reference-literal eager loop, including
    the last partial batch (data/mnist.rs:373-385)."""
    import taper_amd as T
    H = backends.get("hip")
    rng = np.random.default_rng(n + batch)
    spec = backends.mlp_baseline(rng)
    x, y = backends.mnist_like(rng, n)
    results, finals = [], []
    for mode in (T.Trainer.EAGER, T.Trainer.GRAPH):
        model = H.sequential(spec)
        opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
        tr = T.Trainer(model, opt)
        ds = T.MNISTDataset.from_host(x, y)
        loader = T.DataLoader(ds, batch, True, seed=123)
        ep = [tr.run_epoch(loader, mode) for _ in range(2)]      # two epochs: the graph is reused, data reshuffled
        results.append(ep)
        finals.append([p.data() for p in model.parameters()])
        assert opt.t() == 2 * ((n + batch - 1) // batch)
    for e in range(2):
        a, b = results[0][e], results[1][e]
        assert a["num_batches"] == b["num_batches"] == (n + batch - 1) // batch
        assert a["total_samples"] == b["total_samples"] == n
        # the graph path runs fused launches (classifier head, Adam in the epilogues): same arithmetic,
        # different summation order than the per-op eager path
        np.testing.assert_allclose(b["losses"], a["losses"], rtol=3e-4, atol=1e-5)
        assert np.abs(b["ncorrect"] - a["ncorrect"]).max() <= 1
        assert abs(int(a["total_correct"]) - int(b["total_correct"])) <= 3
    for pa, pb in zip(*finals):
        np.testing.assert_allclose(pb, pa, rtol=1e-4, atol=1e-3 * 5e-2)


def test_remainder_graph_equals_the_ladder_graphs():
    """an epoch of 20 steps replays as ONE graph of exactly 20 steps (recorded by the first call that meets that remainder: 20 is no ladder
    size); with 4 steps per replay the same epoch is five chunk graphs: same op list, so every loss and weight agrees bit for bit -- also
    in the second epoch, which reuses the recorded graphs"""
    import taper_amd as T
    H = backends.get("hip")
    rng = np.random.default_rng(2020)
    spec = backends.mlp_baseline(rng)
    batch = 32
    x, y = backends.mnist_like(rng, 20 * batch)
    out = []
    for chunk in (128, 4):
        model = H.sequential(spec)
        opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
        tr = T.Trainer(model, opt, graph_chunk=chunk)
        loader = T.DataLoader(T.MNISTDataset.from_host(x, y), batch, True, seed=5)
        losses = []
        for _ in range(2):
            ep = tr.run_epoch(loader, T.Trainer.GRAPH)
            assert ep["num_batches"] == 20
            losses += list(ep["losses"])
        assert opt.t() == 40
        out.append((np.asarray(losses), [p.data() for p in model.parameters()]))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("n,batch", [(96, 96), (80, 128)])
def test_full_batch_in_index_order_reads_the_dataset_in_place(n, batch):
    """one step per epoch over the whole dataset without shuffling: the graph path skips the (identity) batch
    gather and trains straight from the resident dataset -- same losses / weights as the eager loop"""
    import taper_amd as T
    H = backends.get("hip")
    rng = np.random.default_rng(n)
    spec = backends.mlp_baseline(rng)
    x, y = backends.mnist_like(rng, n)
    out = []
    for mode in (T.Trainer.EAGER, T.Trainer.GRAPH):
        model = H.sequential(spec)
        opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
        tr = T.Trainer(model, opt)
        loader = T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False)
        eps = [tr.run_epoch(loader, mode) for _ in range(3)]
        out.append(([e["losses"][0] for e in eps], [p.data() for p in model.parameters()]))
        assert opt.t() == 3
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=3e-4, atol=1e-5)
    for pa, pb in zip(out[0][1], out[1][1]):
        np.testing.assert_allclose(pb, pa, rtol=1e-4, atol=1e-3 * 5e-2)


def test_epoch_against_oracle_with_loader_semantics():
    """DataLoader order (index order, last partial batch kept) + Trainer metrics formulas
    (train.rs:117,140-141, Q13) against the oracle driven with the same batches."""
    import taper_amd as T
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(3)
    n, batch = 200, 64
    spec = backends.mlp_baseline(rng)
    x, y = backends.mnist_like(rng, n)
    hm, om = H.sequential(spec), Orc.sequential(spec)
    hopt, oopt = T.Adam(hm.parameters(), 1e-3, None, None, 1e-4), Orc.m.Adam(om.parameters(), 1e-3, None, None, 1e-4)
    tr = T.Trainer(hm, hopt)
    loader = T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False)
    assert loader.num_batches() == 4
    ep = tr.run_epoch(loader, T.Trainer.GRAPH)
    tot_loss, tot_correct = np.float32(0), 0
    for s in range(0, n, batch):
        xb, yb = x[s:s + batch], y[s:s + batch]
        r = om.train_step(oopt, xb, yb, (len(xb), 784))
        tot_loss += np.float32(r["loss"])
        tot_correct += int(np.float32(r["acc"]) * np.float32(len(xb)))
    assert ep["total_samples"] == n
    assert abs(ep["avg_loss"] - tot_loss / 4) <= 2 * RTOL * max(1.0, abs(tot_loss / 4))
    assert abs(ep["total_correct"] - tot_correct) <= 2


def test_full_size_properties_config1():
    """BASELINE configs[1] at full size (60 000 samples, batch 64): size-independent
    properties -- every step logged, counts bounded, loss finite and decreasing on a
    learnable synthetic task, identical reruns are bit-identical (determinism)."""
    import taper_amd as T
    H = backends.get("hip")
    rng = np.random.default_rng(60000)
    n = 60000
    y = rng.integers(0, 10, n).astype(np.float32)
    x = rng.integers(0, 256, (n, 784)).astype(np.float32) / np.float32(255.0)
    for c in range(10):                                 # a learnable signal: a bright 40-pixel band per class
        x[y == c, c * 78:c * 78 + 40] = 1.0
    spec = backends.mlp_baseline(np.random.default_rng(1))
    runs = []
    for _ in range(2):
        model = H.sequential(spec)
        opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
        tr = T.Trainer(model, opt)
        loader = T.DataLoader(T.MNISTDataset.from_host(x, y), 64, False)
        ep = tr.run_epoch(loader, T.Trainer.GRAPH)
        runs.append((ep, [p.data() for p in model.parameters()]))
    ep = runs[0][0]
    assert ep["num_batches"] == 938 and ep["total_samples"] == n     # 60000/64 -> 938, last batch 32
    assert np.isfinite(ep["losses"]).all()
    assert (ep["ncorrect"] >= 0).all() and (ep["ncorrect"][:-1] <= 64).all() and ep["ncorrect"][-1] <= 32
    assert ep["ncorrect"].sum() == pytest.approx(ep["total_correct"], abs=938)
    assert ep["losses"][-50:].mean() < 0.5 * ep["losses"][:10].mean()
    assert ep["accuracy"] > 0.5
    np.testing.assert_array_equal(runs[0][0]["losses"], runs[1][0]["losses"])
    for a, b in zip(runs[0][1], runs[1][1]):
        np.testing.assert_array_equal(a, b)


def test_large_batch_step_matches_oracle():
    """the throughput-bound path (split-K tile GEMMs, multi-workgroup head, slabbed column sums, Adam in the
    split-K reduce) on one 16 384-row batch: loss, gradients and two Adam steps against the oracle"""
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(16384)
    batch = 16384
    spec = backends.nonzero_biases(backends.mlp_baseline(rng), rng)
    x, y = backends.mnist_like(rng, batch)
    hm, om = H.sequential(spec), Orc.sequential(spec)
    h_loss, h_acc, h_logits, h_grads = H.forward_backward(hm, x, y, (batch, 784))
    o_loss, o_acc, o_logits, o_grads = Orc.forward_backward(om, x, y, (batch, 784))
    np.testing.assert_allclose(h_logits, o_logits, rtol=RTOL, atol=RTOL * float(np.abs(o_logits).max()))
    assert abs(h_loss - o_loss) <= RTOL * max(1.0, abs(o_loss))
    assert abs(h_acc - o_acc) <= 2.0 / batch          # an argmax between two logits within rounding may flip
    for i, (hg, og) in enumerate(zip(h_grads, o_grads)):
        scale = float(np.abs(og).max())               # sums of 16 384 terms: absolute error scales with the largest entry
        np.testing.assert_allclose(hg, og, rtol=RTOL, atol=2e-4 * scale, err_msg=f"param {i}")
    import taper_amd as T
    hopt, oopt = T.Adam(hm.parameters(), 1e-3, None, None, 1e-4), Orc.m.Adam(om.parameters(), 1e-3, None, None, 1e-4)
    tr = T.Trainer(hm, hopt)
    loader = T.DataLoader(T.MNISTDataset.from_host(np.concatenate([x, x]), np.concatenate([y, y])), batch, False)
    ep = tr.run_epoch(loader, T.Trainer.GRAPH)
    ref = [om.train_step(oopt, x, y, (batch, 784))["loss"] for _ in range(2)]
    margins.check("losses", ep["losses"], ref, BOUND_EPOCH_LOSS)
    for i, (hp, op) in enumerate(zip(hm.parameters(), om.parameters())):
        np.testing.assert_allclose(hp.data(), op.data(), rtol=RTOL, atol=1e-3 * 5e-2, err_msg=f"param {i}")


def test_full_batch_gradient_is_the_mean_of_shard_gradients():
    """size-independent property at BASELINE's full size: the gradient of ONE 60 000-row step equals the
    row-weighted mean of the gradients of four 15 000-row steps (what data parallelism relies on)"""
    H = backends.get("hip")
    rng = np.random.default_rng(3)
    n = 60000
    spec = backends.nonzero_biases(backends.mlp_baseline(rng), rng)
    x, y = backends.mnist_like(rng, n)
    model = H.sequential(spec)
    full_loss, _, _, full = H.forward_backward(model, x, y, (n, 784))
    acc = [np.zeros_like(g) for g in full]
    loss_acc = 0.0
    for s in range(4):
        sl = slice(s * 15000, (s + 1) * 15000)
        l, _, _, g = H.forward_backward(model, x[sl], y[sl], (15000, 784))
        loss_acc += l / 4
        for a, gi in zip(acc, g):
            a += np.asarray(gi) / 4
    assert abs(full_loss - loss_acc) <= 1e-5 * max(1.0, abs(full_loss))
    for i, (f, a) in enumerate(zip(full, acc)):
        # 1e-3 of the scale: among 7.7 M hidden activations a couple sit within rounding of 0, and the two GEMM paths
        # (one 60 000-row problem vs four 15 000-row ones) round them to different sides -- each flipped ReLU mask moves
        # one row of dW1 by ~1e-4 of its scale; a dropped tile or slice would be an O(1) error
        np.testing.assert_allclose(f, a, rtol=RTOL, atol=1e-3 * float(np.abs(f).max()), err_msg=f"param {i}")


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.parametrize("name,batch", [("mlp_baseline", 64), ("cnn_simple", 16), ("cnn_simple", 128), ("cnn_reference", 96)])
def test_evaluate_matches_oracle(name, batch):
    """Trainer::evaluate (train.rs:147-172): per-batch loss / hit count from the device log (read once per pass)
    against the oracle's forward + cross_entropy_loss + accuracy on the same batches, last partial batch included (CNNs: the whole
    batches of 96+ go through the one-launch conv chain, the partial one layer by layer)"""
    import taper_amd as T
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(31)
    builder, sample = MODELS[name]
    spec = backends.nonzero_biases(builder(rng), rng)
    n = 3 * batch + batch // 2
    x, y = backends.mnist_like(rng, n)
    hm, om = H.sequential(spec), Orc.sequential(spec)
    tr = T.Trainer(hm, T.Adam(hm.parameters(), 1e-3, None, None, 1e-4), sample_shape=sample)
    ev = tr.run_epoch(T.DataLoader(T.MNISTDataset.from_host(x, y, False), batch, False), T.Trainer.EVAL)
    O = Orc.m
    ref_loss, ref_nc = [], []
    for s in range(0, n, batch):
        xb, yb = x[s:s + batch], y[s:s + batch]
        O.Tape.reset()
        logits = om.forward(O.Tensor(xb.reshape((len(xb), 784) if sample is None else (len(xb),) + sample)))
        ref_loss.append(float(O.cross_entropy_loss(logits, O.Tensor(yb)).data()[0]))
        ref_nc.append(round(O.accuracy(logits, O.Tensor(yb)) * len(xb)))
    np.testing.assert_allclose(ev["losses"], ref_loss, rtol=2e-4, atol=1e-5)
    assert np.abs(np.asarray(ev["ncorrect"]) - np.asarray(ref_nc)).max() <= 1
    assert ev["total_samples"] == n and ev["num_batches"] == 4
    assert ev["avg_loss"] == pytest.approx(np.mean(ref_loss), rel=2e-4)

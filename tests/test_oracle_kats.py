"""Pins the CPU oracle against every known-answer test the reference holds
for the hot path: tests/smoke.rs, src/loss.rs:292-374,
src/optim.rs:354-423, src/train.rs:387-417.  Each test cites its source."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle.oracle import Tape, Tensor


@pytest.fixture(autouse=True)
def _fresh_tape():
    Tape.reset()
    Tape.set_zero_sentinel(False)   # the reference tests' INTENT (see test_q1_* for the literal quirk)
    yield
    Tape.reset()
    Tape.set_zero_sentinel(True)


def g(t):
    gr = t.grad()
    return 0.0 if gr is None else float(gr.reshape(-1)[0])


def test_mul_grads():  # tests/smoke.rs:19-30
    x = Tensor.scalar(2.0).requires_grad()
    y = Tensor.scalar(3.0).requires_grad()
    z = x * y
    z.backward()
    assert abs(z.data()[0] - 6.0) < 1e-6
    assert abs(g(x) - 3.0) < 1e-6
    assert abs(g(y) - 2.0) < 1e-6


def test_q1_zero_sentinel_literal():
    """Q1: with the literal sentinel (tensor.rs:524-528) backward() from the
    FIRST recorded node is a no-op, so tests/smoke.rs:20-30 cannot pass."""
    Tape.set_zero_sentinel(True)
    x = Tensor.scalar(2.0).requires_grad()
    y = Tensor.scalar(3.0).requires_grad()
    z = x * y
    assert z.tape_node() == 0
    z.backward()
    assert x.grad() is None and y.grad() is None


def test_compound_affine():  # tests/smoke.rs:32-43
    a = Tensor.scalar(2.0).requires_grad()
    b = Tensor.scalar(3.0).requires_grad()
    c = a * b + a
    c.backward()
    assert abs(c.data()[0] - 8.0) < 1e-6
    assert abs(g(a) - 4.0) < 1e-6
    assert abs(g(b) - 2.0) < 1e-6


def test_matmul_shapes_and_grads():  # tests/smoke.rs:45-70
    a = Tensor([1., 2., 3., 4., 5., 6.], (2, 3)).requires_grad()
    b = Tensor([7., 8., 9., 10., 11., 12.], (3, 2)).requires_grad()
    c = a.matmul(b)
    assert c.shape() == (2, 2)
    c.backward()
    assert a.grad().shape == (2, 3) and b.grad().shape == (3, 2)
    cd = c.data().reshape(-1)
    assert abs(cd[0] - 58.0) < 1e-4 and abs(cd[3] - 154.0) < 1e-4
    np.testing.assert_allclose(c.data(), [[58, 64], [139, 154]], atol=1e-4)
    # closed form: dA = 1 * B^T, dB = A^T * 1
    np.testing.assert_allclose(a.grad(), [[15, 19, 23], [15, 19, 23]], atol=1e-4)
    np.testing.assert_allclose(b.grad(), [[5, 5], [7, 7], [9, 9]], atol=1e-4)


def test_reshape_operations():  # tests/smoke.rs:262-290
    x = Tensor(np.arange(12), (3, 4))
    assert x.reshape((2, 6)).shape() == (2, 6)
    assert x.flatten(0).shape() == (12,)
    assert Tensor(np.arange(24), (2, 3, 4)).flatten(1).shape() == (2, 12)
    assert Tensor([1.0, 2.0], (1, 2, 1)).squeeze(None).shape() == (2,)
    x1 = Tensor([1.0, 2.0, 3.0], (3,))
    assert x1.unsqueeze(0).shape() == (1, 3)
    assert x1.unsqueeze(1).shape() == (3, 1)


def test_reshape_gradients():  # tests/smoke.rs:292-307
    x = Tensor([1.0, 2.0, 3.0, 4.0], (2, 2)).requires_grad()
    s = x.reshape((4,)).sum(None, False)
    s.backward()
    np.testing.assert_allclose(x.grad(), np.ones((2, 2)), atol=1e-6)


def test_sum_operations():  # tests/smoke.rs:309-336
    x = Tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0], (2, 3))
    sa = x.sum(None, False)
    assert sa.shape() == (1,) and abs(sa.data()[0] - 21.0) < 1e-6
    s0 = x.sum(0, False)
    assert s0.shape() == (3,)
    np.testing.assert_allclose(s0.data(), [5, 7, 9], atol=1e-6)
    s1 = x.sum(1, False)
    assert s1.shape() == (2,)
    np.testing.assert_allclose(s1.data(), [6, 15], atol=1e-6)
    assert x.sum(1, True).shape() == (2, 1)


def test_sum_gradients():  # tests/smoke.rs:338-354
    x = Tensor([1.0, 2.0, 3.0, 4.0], (2, 2)).requires_grad()
    loss = x.sum(1, False).sum(None, False)
    loss.backward()
    np.testing.assert_allclose(x.grad(), np.ones((2, 2)), atol=1e-6)


def test_max_operations():  # tests/smoke.rs:356-377
    x = Tensor([1.0, 3.0, 2.0, 4.0, 6.0, 5.0], (2, 3))
    mv, mi = x.max(0)
    assert mv.shape() == (1, 3)
    np.testing.assert_allclose(mv.data().reshape(-1), [4, 6, 5], atol=1e-6)
    np.testing.assert_allclose(mi.data().reshape(-1), [1, 1, 1], atol=1e-6)
    am = x.argmax(1)
    assert am.shape() == (2, 1)
    np.testing.assert_allclose(am.data().reshape(-1), [1, 1], atol=1e-6)


def test_exp_log_operations():  # tests/smoke.rs:379-406
    x = Tensor([0.0, 1.0, 2.0], (3,))
    e = x.exp()
    assert abs(e.data()[0] - 1.0) < 1e-6
    assert abs(e.data()[1] - 2.71828) < 1e-2
    assert abs(e.data()[2] - 7.38906) < 1e-2
    np.testing.assert_allclose(e.log().data(), x.data(), atol=1e-5)
    x2 = Tensor([1.0, 4.0, 9.0], (3,))
    sq = x2.sqrt()
    np.testing.assert_allclose(sq.data(), [1, 2, 3], atol=1e-6)
    np.testing.assert_allclose(sq.pow(2.0).data(), x2.data(), atol=1e-5)


def test_exp_log_gradients():  # tests/smoke.rs:408-435
    x = Tensor([1.0, 2.0], (2,)).requires_grad()
    x.exp().sum(None, False).backward()
    np.testing.assert_allclose(x.grad(), np.exp([1.0, 2.0]), atol=1e-5)
    Tape.reset()
    x = Tensor([1.0, 2.0, 3.0], (3,)).requires_grad()
    x.log().sum(None, False).backward()
    np.testing.assert_allclose(x.grad(), [1.0, 0.5, 1 / 3.0], atol=1e-5)


def test_softmax_cross_entropy():  # tests/smoke.rs:437-459 (+Q12: softmax = exp(log_softmax))
    logits = Tensor([1.0, 2.0, 3.0, 4.0, 1.0, 2.0], (2, 3))
    probs = O.softmax(logits, -1)
    np.testing.assert_allclose(probs.sum(1, False).data(), [1, 1], atol=1e-6)
    assert (probs.data() > 0).all()
    logits = Tensor([2.0, 1.0, 0.0, 0.0, 1.0, 2.0], (2, 3)).requires_grad()
    targets = Tensor([0.0, 2.0], (2,))
    loss = O.cross_entropy_loss(logits, targets)
    assert loss.data()[0] > 0.0
    loss.backward()
    assert logits.grad() is not None
    # closed form: both rows are [2,1,0] up to permutation -> loss = log(1+e^-1+e^-2)
    assert abs(loss.data()[0] - np.log(1 + np.exp(-1) + np.exp(-2))) < 1e-6


def test_loss_rs_softmax():  # src/loss.rs:297-312
    x = Tensor([1.0, 2.0, 3.0, 4.0, 1.0, 2.0], (2, 3))
    y = O.softmax(x, -1).data()
    assert abs(y[0].sum() - 1.0) < 1e-6 and abs(y[1].sum() - 1.0) < 1e-6
    assert (y > 0).all()


def test_cross_entropy_grad_sign():  # src/loss.rs:314-340
    logits = Tensor([2.0, 1.0, -1.0, 3.0], (2, 2)).requires_grad()
    targets = Tensor([0.0, 1.0], (2,))
    loss = O.cross_entropy_loss(logits, targets)
    assert loss.data()[0] > 0.0
    loss.backward()
    gr = logits.grad().reshape(-1)
    assert gr[0] < 0.0 and gr[3] < 0.0


def test_one_hot():  # src/loss.rs:342-356
    oh = O.one_hot(Tensor([0.0, 2.0, 1.0], (3,)), 3)
    assert oh.shape() == (3, 3)
    np.testing.assert_array_equal(oh.data(), [[1, 0, 0], [0, 0, 1], [0, 1, 0]])


def test_accuracy():  # src/loss.rs:358-373
    preds = Tensor([0.1, 0.9, 0.8, 0.2, 0.3, 0.7], (3, 2))
    targets = Tensor([1.0, 0.0, 0.0], (3,))
    assert abs(O.accuracy(preds, targets) - 2.0 / 3.0) < 1e-6


def test_numerical_stability():  # tests/smoke.rs:504-523
    x = Tensor([1000.0, 1001.0, 1002.0], (1, 3))
    p = O.softmax(x, -1).data()
    assert np.isfinite(p).all() and (p >= 0).all() and (p <= 1).all()
    assert np.isfinite(O.log_softmax(x, -1).data()).all()


def test_mnist_simulation():  # tests/smoke.rs:461-502
    rng = np.random.default_rng(0)
    x = Tensor(rng.standard_normal((4, 784)), (4, 784))
    w = Tensor(rng.standard_normal((10, 784)), (10, 784)).requires_grad()
    b = Tensor(rng.standard_normal(10), (10,)).requires_grad()
    logits = x.matmul(w.transpose()).add_broadcast(b)
    targets = Tensor([3.0, 7.0, 1.0, 9.0], (4,))
    loss = O.cross_entropy_loss(logits, targets)
    loss.backward()
    assert w.grad() is not None and b.grad() is not None
    assert 0.0 <= O.accuracy(logits, targets) <= 1.0


def test_adam_optimizer():  # src/optim.rs:359-389
    rng = np.random.default_rng(1)
    w = Tensor(rng.standard_normal((10, 10)), (10, 10)).requires_grad()
    b = Tensor(rng.standard_normal(10), (10,)).requires_grad()
    opt = O.Adam([w, b], 0.001)
    w.set_grad(np.full(100, 0.1))
    b.set_grad(np.full(10, 0.1))
    before = w.data().copy()
    opt.step()
    assert (np.abs(before - w.data()) > 1e-6).all()
    # first Adam step with constant grad moves every weight by ~lr
    np.testing.assert_allclose(before - w.data(), 0.001, rtol=1e-3)
    opt.zero_grad()
    assert w.grad() is None and b.grad() is None


def test_trainer_basic():  # src/train.rs:387-417: 784-128-10 MLP, randn[100,784], batch 32, Adam 1e-3
    rng = np.random.default_rng(2)
    def lin(i, o):
        s = np.sqrt(2.0 / i)
        return (Tensor(rng.uniform(-s, s, (o, i)), (o, i)).requires_grad(), Tensor(np.zeros(o), (o,)).requires_grad())
    w1, b1 = lin(784, 128)
    w2, b2 = lin(128, 10)
    model = O.Sequential([dict(kind="linear", w=w1, b=b1), dict(kind="relu"), dict(kind="linear", w=w2, b=b2)])
    images = rng.standard_normal((100, 784)).astype(np.float32)
    labels = (np.arange(100) % 10).astype(np.float32)
    opt = O.Adam(model.parameters(), 0.001)
    total_loss, total_correct, total = 0.0, 0, 0
    for s in range(0, 100, 32):
        xb, yb = images[s:s + 32], labels[s:s + 32]
        r = model.train_step(opt, xb, yb, (len(xb), 784))
        total_loss += r["loss"]
        total_correct += int(r["acc"] * len(xb))
        total += len(xb)
    loss, acc = total_loss / 4, total_correct / total
    assert loss > 0.0 and 0.0 <= acc <= 1.0
    assert opt.t() == 4

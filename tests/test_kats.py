"""Known-answer tests of the reference (tests/smoke.rs,
src/loss.rs:292-374, src/optim.rs:354-423, src/train.rs:387-417), run against
BOTH implementations: the CPU oracle (pins the oracle to the reference's own
vectors; runs everywhere) and the HIP product path (-m gpu)."""
import numpy as np
import pytest

from tests import backends


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def B(request):
    b = backends.get(request.param)
    b.Tape.reset()
    b.set_zero_sentinel(False)  # the reference tests' INTENT; the literal quirk Q1 is test_q1_*
    yield b
    b.Tape.reset()


def g(t):
    gr = t.grad()
    return 0.0 if gr is None else float(np.asarray(gr).reshape(-1)[0])


def test_mul_grads(B):  # tests/smoke.rs:19-30
    x = B.Tensor.scalar(2.0).requires_grad()
    y = B.Tensor.scalar(3.0).requires_grad()
    z = x * y
    z.backward()
    assert abs(z.data()[0] - 6.0) < 1e-6
    assert abs(g(x) - 3.0) < 1e-6
    assert abs(g(y) - 2.0) < 1e-6


def test_q1_zero_sentinel_literal(B):
    """Q1: with the literal sentinel (tensor.rs:524-528) backward() from the FIRST
    recorded node is a no-op, so tests/smoke.rs:20-30 cannot pass as written."""
    B.set_zero_sentinel(True)
    x = B.Tensor.scalar(2.0).requires_grad()
    y = B.Tensor.scalar(3.0).requires_grad()
    z = x * y
    assert z.tape_node() == 0
    z.backward()
    assert x.grad() is None and y.grad() is None
    B.set_zero_sentinel(False)


def test_compound_affine(B):  # tests/smoke.rs:32-43
    a = B.Tensor.scalar(2.0).requires_grad()
    b = B.Tensor.scalar(3.0).requires_grad()
    c = a * b + a
    c.backward()
    assert abs(c.data()[0] - 8.0) < 1e-6
    assert abs(g(a) - 4.0) < 1e-6
    assert abs(g(b) - 2.0) < 1e-6


def test_matmul_shapes_and_grads(B):  # tests/smoke.rs:45-70
    a = B.Tensor([1., 2., 3., 4., 5., 6.], (2, 3)).requires_grad()
    b = B.Tensor([7., 8., 9., 10., 11., 12.], (3, 2)).requires_grad()
    c = a.matmul(b)
    assert c.shape() == (2, 2)
    c.backward()
    assert a.grad().shape == (2, 3) and b.grad().shape == (3, 2)
    np.testing.assert_allclose(c.data(), [[58, 64], [139, 154]], atol=1e-4)
    np.testing.assert_allclose(a.grad(), [[15, 19, 23], [15, 19, 23]], atol=1e-4)  # 1 . B^T
    np.testing.assert_allclose(b.grad(), [[5, 5], [7, 7], [9, 9]], atol=1e-4)      # A^T . 1


def test_reshape_operations(B):  # tests/smoke.rs:262-290
    x = B.Tensor(np.arange(12), (3, 4))
    assert x.reshape((2, 6)).shape() == (2, 6)
    assert x.flatten(0).shape() == (12,)
    assert B.Tensor(np.arange(24), (2, 3, 4)).flatten(1).shape() == (2, 12)
    assert B.Tensor([1.0, 2.0], (1, 2, 1)).squeeze(None).shape() == (2,)
    x1 = B.Tensor([1.0, 2.0, 3.0], (3,))
    assert x1.unsqueeze(0).shape() == (1, 3)
    assert x1.unsqueeze(1).shape() == (3, 1)


def test_reshape_gradients(B):  # tests/smoke.rs:292-307
    x = B.Tensor([1.0, 2.0, 3.0, 4.0], (2, 2)).requires_grad()
    s = x.reshape((4,)).sum(None, False)
    s.backward()
    np.testing.assert_allclose(x.grad(), np.ones((2, 2)), atol=1e-6)


def test_sum_operations(B):  # tests/smoke.rs:309-336
    x = B.Tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0], (2, 3))
    sa = x.sum(None, False)
    assert sa.shape() == (1,) and abs(sa.data()[0] - 21.0) < 1e-6
    s0 = x.sum(0, False)
    assert s0.shape() == (3,)
    np.testing.assert_allclose(s0.data(), [5, 7, 9], atol=1e-6)
    s1 = x.sum(1, False)
    assert s1.shape() == (2,)
    np.testing.assert_allclose(s1.data(), [6, 15], atol=1e-6)
    assert x.sum(1, True).shape() == (2, 1)


def test_sum_gradients(B):  # tests/smoke.rs:338-354
    x = B.Tensor([1.0, 2.0, 3.0, 4.0], (2, 2)).requires_grad()
    loss = x.sum(1, False).sum(None, False)
    loss.backward()
    np.testing.assert_allclose(x.grad(), np.ones((2, 2)), atol=1e-6)


def test_max_operations(B):  # tests/smoke.rs:356-377
    x = B.Tensor([1.0, 3.0, 2.0, 4.0, 6.0, 5.0], (2, 3))
    mv, mi = x.max(0)
    assert mv.shape() == (1, 3)
    np.testing.assert_allclose(mv.data().reshape(-1), [4, 6, 5], atol=1e-6)
    np.testing.assert_array_equal(mi.data().reshape(-1), [1, 1, 1])
    am = x.argmax(1)
    assert am.shape() == (2, 1)
    np.testing.assert_array_equal(am.data().reshape(-1), [1, 1])


def test_argmax_first_max_wins(B):  # tensor.rs:1062 strict '>' -> ties resolve to the first index
    x = B.Tensor([[1, 5, 5, 2], [7, 7, 7, 7], [0, -1, 0, -1]], (3, 4))
    np.testing.assert_array_equal(x.argmax(1).data().reshape(-1), [1, 0, 0])


def test_exp_log_operations(B):  # tests/smoke.rs:379-406
    x = B.Tensor([0.0, 1.0, 2.0], (3,))
    e = x.exp()
    assert abs(e.data()[0] - 1.0) < 1e-6
    assert abs(e.data()[1] - 2.71828) < 1e-2
    assert abs(e.data()[2] - 7.38906) < 1e-2
    np.testing.assert_allclose(e.log().data(), x.data(), atol=1e-5)
    x2 = B.Tensor([1.0, 4.0, 9.0], (3,))
    sq = x2.sqrt()
    np.testing.assert_allclose(sq.data(), [1, 2, 3], atol=1e-6)
    np.testing.assert_allclose(sq.pow(2.0).data(), x2.data(), atol=1e-5)


def test_exp_log_gradients(B):  # tests/smoke.rs:408-435
    x = B.Tensor([1.0, 2.0], (2,)).requires_grad()
    x.exp().sum(None, False).backward()
    np.testing.assert_allclose(x.grad(), np.exp([1.0, 2.0]), atol=1e-5)
    B.Tape.reset()
    x = B.Tensor([1.0, 2.0, 3.0], (3,)).requires_grad()
    x.log().sum(None, False).backward()
    np.testing.assert_allclose(x.grad(), [1.0, 0.5, 1 / 3.0], atol=1e-5)


def test_softmax_cross_entropy(B):  # tests/smoke.rs:437-459 (Q12: softmax = exp(log_softmax))
    logits = B.Tensor([1.0, 2.0, 3.0, 4.0, 1.0, 2.0], (2, 3))
    probs = B.softmax(logits, -1)
    np.testing.assert_allclose(probs.sum(1, False).data(), [1, 1], atol=1e-6)
    assert (probs.data() > 0).all()
    logits = B.Tensor([2.0, 1.0, 0.0, 0.0, 1.0, 2.0], (2, 3)).requires_grad()
    targets = B.Tensor([0.0, 2.0], (2,))
    loss = B.cross_entropy_loss(logits, targets)
    assert loss.data()[0] > 0.0
    loss.backward()
    assert logits.grad() is not None
    assert abs(loss.data()[0] - np.log(1 + np.exp(-1) + np.exp(-2))) < 1e-6  # closed form
    p = np.exp([2.0, 1.0, 0.0]); p /= p.sum()
    np.testing.assert_allclose(logits.grad(), np.array([[p[0] - 1, p[1], p[2]], [p[2], p[1], p[0] - 1]]) / 2, atol=1e-6)


def test_log_softmax_autograd(B):  # loss.rs:101-126 as a differentiable chain
    x = B.Tensor([[1.0, 2.0, 3.0], [0.5, -1.0, 2.0]], (2, 3)).requires_grad()
    lp = B.log_softmax(x, -1)
    w = B.Tensor([[1.0, 0.0, 2.0], [0.0, 3.0, 0.0]], (2, 3))
    (lp * w).sum(None, False).backward()
    xs = x.data().astype(np.float64)
    p = np.exp(xs - xs.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
    wn = w.data().astype(np.float64)
    np.testing.assert_allclose(x.grad(), wn - p * wn.sum(1, keepdims=True), atol=1e-5)


def test_cross_entropy_grad_sign(B):  # src/loss.rs:314-340
    logits = B.Tensor([2.0, 1.0, -1.0, 3.0], (2, 2)).requires_grad()
    targets = B.Tensor([0.0, 1.0], (2,))
    loss = B.cross_entropy_loss(logits, targets)
    assert loss.data()[0] > 0.0
    loss.backward()
    gr = logits.grad().reshape(-1)
    assert gr[0] < 0.0 and gr[3] < 0.0


def test_one_hot(B):  # src/loss.rs:342-356
    oh = B.one_hot(B.Tensor([0.0, 2.0, 1.0], (3,)), 3)
    assert oh.shape() == (3, 3)
    np.testing.assert_array_equal(oh.data(), [[1, 0, 0], [0, 0, 1], [0, 1, 0]])


def test_accuracy(B):  # src/loss.rs:358-373
    preds = B.Tensor([0.1, 0.9, 0.8, 0.2, 0.3, 0.7], (3, 2))
    targets = B.Tensor([1.0, 0.0, 0.0], (3,))
    assert abs(B.accuracy(preds, targets) - 2.0 / 3.0) < 1e-6


def test_numerical_stability(B):  # tests/smoke.rs:504-523
    x = B.Tensor([1000.0, 1001.0, 1002.0], (1, 3))
    p = B.softmax(x, -1).data()
    assert np.isfinite(p).all() and (p >= 0).all() and (p <= 1).all()
    assert np.isfinite(B.log_softmax(x, -1).data()).all()


def test_mnist_simulation(B):  # tests/smoke.rs:461-502
    rng = np.random.default_rng(0)
    x = B.Tensor(rng.standard_normal((4, 784)), (4, 784))
    w = B.Tensor(rng.standard_normal((10, 784)), (10, 784)).requires_grad()
    b = B.Tensor(rng.standard_normal(10), (10,)).requires_grad()
    logits = x.matmul(w.transpose()).add_broadcast(b)
    targets = B.Tensor([3.0, 7.0, 1.0, 9.0], (4,))
    loss = B.cross_entropy_loss(logits, targets)
    loss.backward()
    assert w.grad() is not None and b.grad() is not None
    assert 0.0 <= B.accuracy(logits, targets) <= 1.0


def test_adam_optimizer(B):  # src/optim.rs:359-389
    rng = np.random.default_rng(1)
    w = B.Tensor(rng.standard_normal((10, 10)), (10, 10)).requires_grad()
    b = B.Tensor(rng.standard_normal(10), (10,)).requires_grad()
    opt = B.Adam([w, b], 0.001)
    w.set_grad(np.full(100, 0.1))
    b.set_grad(np.full(10, 0.1))
    before = w.data().copy()
    opt.step()
    assert (np.abs(before - w.data()) > 1e-6).all()
    np.testing.assert_allclose(before - w.data(), 0.001, rtol=1e-3)  # first Adam step with constant grad ~ lr
    opt.zero_grad()
    assert w.grad() is None and b.grad() is None


def test_adam_skips_gradless_params(B):  # optim.rs:93: `if let Some(grad)` -- no weight decay either (Q8)
    w = B.Tensor(np.ones(8), (8,)).requires_grad()
    v = B.Tensor(np.ones(8), (8,)).requires_grad()
    opt = B.Adam([w, v], 0.1, None, None, 0.5)
    w.set_grad(np.full(8, 0.25))
    opt.step()
    assert (w.data() < 1.0).all()
    np.testing.assert_array_equal(v.data(), np.ones(8, np.float32))
    assert opt.t() == 1


def test_trainer_basic(B):  # src/train.rs:387-417: 784-128-10, randn[100,784], batch 32, Adam 1e-3
    rng = np.random.default_rng(2)
    model = B.sequential(backends.mlp_baseline(rng))
    images = rng.standard_normal((100, 784)).astype(np.float32)
    labels = (np.arange(100) % 10).astype(np.float32)
    opt = B.Adam(model.parameters(), 0.001)
    total_loss, total_correct, total = 0.0, 0, 0
    for s in range(0, 100, 32):
        xb, yb = images[s:s + 32], labels[s:s + 32]
        loss, acc, _, grads = B.forward_backward(model, xb, yb, (len(xb), 784))
        for p, gr in zip(model.parameters(), grads):
            p.set_grad(gr)
        opt.step()
        opt.zero_grad()
        total_loss += loss
        total_correct += int(acc * len(xb))
        total += len(xb)
    assert total_loss / 4 > 0.0 and 0.0 <= total_correct / total <= 1.0
    assert opt.t() == 4

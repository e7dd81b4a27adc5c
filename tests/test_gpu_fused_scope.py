"""`Adam.fused_step()` (include/taper_host.h: tp_adam_fused_begin / _end): the Trainer's fused-update mode for a hand-written loop -- inside it
`loss.backward()` applies each parameter's Adam update (/root/reference/src/optim.rs:83-113, same arithmetic, same t) in the epilogue of
the kernel that completes its gradient, `step()` covers what nobody fused.  Held against the same loop without the scope (the reference-literal
order: backward, then one arena-wide Adam launch) and against the oracle's loop, on shapes of both backward forms: the one-launch form of the
latency-bound shapes (the update is carried by a later launch) and products as launches of their own (th_linear_bwd_separate_products: the
update rides in the dW product's epilogue although d_dx is asked for)."""
import numpy as np
import pytest

from tests import backends, margins

pytestmark = pytest.mark.gpu


def _spec(rng, dims):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(backends._lin(rng, dims[i], dims[i + 1]))
        if i + 2 < len(dims):
            layers.append(dict(kind="relu"))
    return backends.nonzero_biases(layers, rng)


# (batch, widths): the last one's hidden products are "big" (>= 64 tiles of 128 x 128 / deep K): separate launches
@pytest.mark.parametrize("batch,dims", [(64, (96, 48, 32, 10)), (256, (784, 128, 64, 10)), (1024, (1024, 1024, 1024, 10))])
def test_fused_scope_equals_the_plain_loop_and_the_oracle(batch, dims):
    import taper_amd as T
    from taper_amd._lib import hip as lib
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(batch + dims[0])
    spec = _spec(rng, dims)
    steps, lr = 3, 1e-3
    xs = [rng.uniform(0, 1, (batch, dims[0])).astype(np.float32) for _ in range(steps)]
    ys = [rng.integers(0, dims[-1], batch).astype(np.float32) for _ in range(steps)]
    if dims[1] >= 1024:
        assert lib.th_linear_bwd_separate_products(batch, dims[1], dims[2], 1, 1, 1) == 1        # the middle layer: dX first, then dW + Adam
    else:
        assert lib.th_linear_bwd_separate_products(batch, dims[1], dims[2], 1, 1, 0) == 0

    def run(fused):
        model = H.sequential(spec)
        opt = T.Adam(model.parameters(), lr, None, None, 1e-4)
        losses = []
        for x, y in zip(xs, ys):
            T.Tape.reset()
            opt.zero_grad()
            if fused:
                with opt.fused_step():
                    loss = T.cross_entropy_loss(model.forward(T.Tensor(x)), T.Tensor(y))
                    loss.backward()
                    opt.step()
            else:
                loss = T.cross_entropy_loss(model.forward(T.Tensor(x)), T.Tensor(y))
                loss.backward()
                opt.step()
            losses.append(float(loss.data()[0]))
        T.Tape.reset()
        assert opt.t() == steps
        m, v = opt.moments()
        return losses, [p.data().copy() for p in model.parameters()], m, v

    l_plain, p_plain, m_plain, v_plain = run(False)
    l_fused, p_fused, m_fused, v_fused = run(True)
    if dims[1] < 1024:
        # the same kernels produce the gradients; only WHERE the update runs differs: identical arithmetic per element
        np.testing.assert_array_equal(np.array(l_plain), np.array(l_fused))
        for a, b in zip(p_plain, p_fused):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(m_plain, m_fused)
        np.testing.assert_array_equal(v_plain, v_fused)
    else:
        # big shapes: a hidden layer's bias gradient is the sum of the row blocks' column sums its consumer left (plain loop) or a column-sum
        # pass of its own (a fused bias update wants the gradient in the launch that owns it): two summation orders, the last bit of ~600 of
        # 1 024 sums after one step (tools/fused_scope_probe.py) -- both are held to the oracle below, and to each other within rounding here
        assert np.abs(np.array(l_plain) - np.array(l_fused)).max() <= 1e-6
        for a, b in zip(p_plain, p_fused):
            assert np.abs(a - b).max() <= 2e-2 * lr
        # (the first step's weights are bit-identical where the gradient is: W1, W2, W3 -- one step in the probe)

    om = Orc.sequential(spec)
    oopt = Orc.m.Adam(om.parameters(), lr, None, None, 1e-4)
    for s, (x, y) in enumerate(zip(xs, ys)):
        r = om.train_step(oopt, x, y, x.shape)
        assert abs(l_fused[s] - r["loss"]) <= 2e-4 * max(1.0, abs(r["loss"])), f"step {s}"
    for i, (hp, pp, op) in enumerate(zip(p_fused, p_plain, om.parameters())):
        margins.check_adam_weights(f"fused_scope_param{i}", hp, op.data(), oopt.v(i), lr, steps, 2e-2 * steps, test="test_fused_scope")
        margins.check_adam_weights(f"plain_loop_param{i}", pp, op.data(), oopt.v(i), lr, steps, 2e-2 * steps, test="test_fused_scope")


def test_fused_scope_refuses_to_nest():
    import taper_amd as T
    from taper_amd._lib import TaperError
    H = backends.get("hip")
    rng = np.random.default_rng(3)
    model = H.sequential(_spec(rng, (32, 16, 4)))
    opt = T.Adam(model.parameters(), 1e-3, None, None, 0.0)
    with opt.fused_step():
        with pytest.raises(TaperError, match="already open"):
            with opt.fused_step():
                pass
    # the failed inner begin left the outer scope's bookkeeping alone: a plain step still works afterwards
    T.Tape.reset()
    opt.zero_grad()
    x, y = T.Tensor(rng.uniform(0, 1, (8, 32)).astype(np.float32)), T.Tensor(rng.integers(0, 4, 8).astype(np.float32))
    T.cross_entropy_loss(model.forward(x), y).backward()
    opt.step()
    T.Tape.reset()


def test_fused_scope_without_step_inside_completes_the_step_and_says_so():
    """backward() inside the scope, step() forgotten: the parameters fused in the backward kernels are already updated -- the scope's end
    completes the step (same weights as the correct loop) and raises, instead of leaving half a step behind"""
    import taper_amd as T
    H = backends.get("hip")
    rng = np.random.default_rng(5)
    spec = _spec(rng, (96, 48, 32, 10))
    x, y = rng.uniform(0, 1, (64, 96)).astype(np.float32), rng.integers(0, 10, 64).astype(np.float32)

    def run(forget):
        model = H.sequential(spec)
        opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
        T.Tape.reset()
        opt.zero_grad()
        if forget:
            with pytest.raises(RuntimeError, match="step\\(\\) did not"):
                with opt.fused_step():
                    T.cross_entropy_loss(model.forward(T.Tensor(x)), T.Tensor(y)).backward()
        else:
            with opt.fused_step():
                T.cross_entropy_loss(model.forward(T.Tensor(x)), T.Tensor(y)).backward()
                opt.step()
        T.Tape.reset()
        assert opt.t() == 1
        return [p.data().copy() for p in model.parameters()]

    for a, b in zip(run(False), run(True)):
        np.testing.assert_array_equal(a, b)

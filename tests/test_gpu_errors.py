"""Error paths the reference takes by panicking inside an op, here raised at the first wait after the offending launch
(th_ctx_sync / th_memcpy_d2h return non-zero, the host mirror throws): loss.rs:160-161 `Target class {} out of bounds for {}`."""
import numpy as np
import pytest

from taper_amd._lib import TaperError

pytestmark = pytest.mark.gpu

MSG = "Target class 10 out of bounds for 10"


@pytest.fixture(scope="module")
def ctx():
    from taper_amd import hip
    c = hip.Ctx(0)
    yield c
    c.close()


@pytest.mark.parametrize("batch", [7, 64, 1500])      # one launch (<= 1024 rows) and the rows + finish pair
def test_softmax_xent_target_out_of_range(ctx, batch):
    rng = np.random.default_rng(batch)
    logits = rng.standard_normal((batch, 10)).astype(np.float32)
    y = rng.integers(0, 10, batch).astype(np.float32)
    logp, loss = ctx.empty(batch * 10), ctx.empty(1)
    args = (batch, 10, logp, loss, None, None, None, None, 0, None, 0, None)
    ctx.call("th_softmax_xent_fwd", ctx.upload(logits), ctx.upload(y), *args)
    ctx.sync()                                           # valid targets: nothing raised
    y[batch // 2] = 10.0
    ctx.call("th_softmax_xent_fwd", ctx.upload(logits), ctx.upload(y), *args)
    with pytest.raises(TaperError, match=MSG):
        ctx.sync()
    ctx.sync()                                           # the note is consumed by the wait that reported it
    assert np.isnan(ctx.download(loss, 1)[0])            # and the loss of that launch is NaN


def test_eager_cross_entropy_loss_raises_at_the_first_read():
    import taper_amd as T
    rng = np.random.default_rng(1)
    logits = T.Tensor(rng.standard_normal((8, 10)).astype(np.float32)).requires_grad()
    y = np.arange(8, dtype=np.float32)
    y[3] = 10.0
    loss = T.cross_entropy_loss(logits, T.Tensor(y))
    with pytest.raises(TaperError, match=MSG):
        loss.data()
    T.Tape.reset()


def _mlp(T, sizes):
    mods = []
    for i in range(len(sizes) - 1):
        mods.append(T.Linear(sizes[i], sizes[i + 1], True, seed=1 + i))
        if i + 2 < len(sizes):
            mods.append(T.ReLU())
    return T.Sequential(mods)


def _cnn(T):
    return T.Sequential([T.Conv2dReLU(1, 32, (3, 3), (1, 1), (1, 1), bias=True, seed=1), T.MaxPool2d((2, 2)),
                         T.Conv2dReLU(32, 64, (3, 3), (1, 1), (1, 1), bias=True, seed=2), T.MaxPool2d((2, 2)),
                         T.Flatten(), T.Linear(3136, 10, True, seed=3)])


@pytest.mark.parametrize("kind,batch", [("mlp2", 64), ("mlp2", 256), ("mlp2", 2048), ("mlp3", 256), ("cnn", 128)])
def test_trainer_step_raises_on_out_of_range_target(kind, batch):
    """every fused step form (th_mlp_tail, th_linear_xent_head, th_mlp3_xent, the conv chain's classifier rows) leaves the note"""
    import taper_amd as T
    rng = np.random.default_rng(batch)
    if kind == "cnn":
        model, x = _cnn(T), rng.uniform(0, 1, (batch, 1, 28, 28)).astype(np.float32)
    else:
        model = _mlp(T, [784, 128, 10] if kind == "mlp2" else [784, 128, 64, 10])
        x = rng.uniform(0, 1, (batch, 784)).astype(np.float32)
    y = rng.integers(0, 10, batch).astype(np.float32)
    opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
    tr = T.Trainer(model, opt)
    loss, _ = tr.train_step(T.Tensor(x), T.Tensor(y))
    assert np.isfinite(loss)
    y[batch - 1] = 10.0
    with pytest.raises(TaperError, match=MSG):
        tr.train_step(T.Tensor(x), T.Tensor(y))
    T.Tape.reset()

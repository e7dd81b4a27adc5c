"""Trainer steps of an MLP whose hidden layers are WIDE (784-1024-1024-10 at 1 024 rows): no fused step form covers it, so every layer runs
its own launches -- th_linear_fwd, the classifier head, th_linear_bwd_adam_ex2 -- and with the Trainer's fused updates on, the middle
layer's products are launches of their own (th_linear_bwd_separate_products) whose dW product carries Adam(W) in its epilogue although
dX is asked for as well (sgemm_tile<.., ADAMEP>).  Against the oracle's training loop (/root/reference/src/train.rs:98-144 over nn.rs:28-78,
optim.rs:83-113); r05 held this path through the public Adam.fused_step() scope, which is gone."""
import numpy as np
import pytest

from tests import backends, margins

pytestmark = pytest.mark.gpu


def test_wide_mlp_trainer_steps_match_the_oracle_loop():
    import taper_amd as T
    from taper_amd._lib import hip as lib
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    # (two steps at lr 1e-4: from the third step on a handful of elements whose gradient sits at eps move by a visible fraction of lr in ANY
    # two summation orders -- the hand-written HIP loop lands 0.30 lr from the oracle there, this Trainer 6e-4 lr from that loop)
    batch, dims, steps, lr = 1024, (784, 1024, 1024, 10), 2, 1e-4
    rng = np.random.default_rng(41)
    spec = []
    for i in range(len(dims) - 1):
        spec.append(backends._lin(rng, dims[i], dims[i + 1]))
        if i + 2 < len(dims):
            spec.append(dict(kind="relu"))
    spec = backends.nonzero_biases(spec, rng)
    x, y = backends.mnist_like(rng, steps * batch)
    assert lib.th_linear_bwd_separate_products(batch, dims[1], dims[2], 1, 1, 1) == 1        # the middle layer: dX first, then dW + Adam

    model = H.sequential(spec)
    opt = T.Adam(model.parameters(), lr, None, None, 1e-4)
    tr = T.Trainer(model, opt)
    loader = T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False)
    losses = tr.run_epoch(loader, T.Trainer.GRAPH)["losses"]
    assert opt.t() == steps

    om = Orc.sequential(spec)
    oopt = Orc.m.Adam(om.parameters(), lr, None, None, 1e-4)
    olosses = [om.train_step(oopt, x[s * batch:(s + 1) * batch], y[s * batch:(s + 1) * batch], (batch, 784))["loss"] for s in range(steps)]
    name = "test_wide_mlp_trainer_steps_match_the_oracle_loop"
    margins.check("losses", np.asarray(losses), np.asarray(olosses), 1e-5, test=name)
    for i, (hp, op) in enumerate(zip(model.parameters(), om.parameters())):
        margins.check_adam_weights(f"param{i}", hp.data(), op.data(), oopt.v(i), lr, steps, 2e-2 * steps, test=name)

"""The data-parallel exchange INSIDE the gradient launch (csrc/dp_dev.h, th_mlp_tail_dp) in ONE process: a loopback communicator is
W = 2 with this process as its own peer -- every push, flag, poll and load of the protocol runs, through local memory, and because
(g + g) * 0.5 == g the step must be the single-GPU step bit for bit.  The W > 1 runs across processes are in tests/test_gpu_dp.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mlp(T, hidden=128):
    return T.Sequential([T.Linear(784, hidden, True, seed=1), T.ReLU(), T.Linear(hidden, 10, True, seed=2)])


def _data(n, seed=5):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 256, (n, 784)).astype(np.float32) / 255.0
    y = rng.integers(0, 10, n).astype(np.float32)
    return x, y


def _train(T, batch, steps, mode, hidden, comm_kind):
    model = _mlp(T, hidden)
    opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
    comm = None
    if comm_kind == "loopback":
        comm = T.Communicator.loopback()
    elif comm_kind == "one_rank":
        comm = T.Communicator.p2p(1, 0)
        comm.connect(comm.export_arena(opt))
    tr = T.Trainer(model, opt, comm=comm)
    x, y = _data(batch * steps)
    loader = T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False)
    losses = np.concatenate([tr.run_epoch(loader, mode)["losses"] for _ in range(2)])
    return losses, [p.data() for p in model.parameters()], opt.t(), opt.moments()[0], opt.moments()[1], comm


def test_loopback_exchange_selftest():
    import taper_amd as T
    comm = T.Communicator.loopback()
    assert comm.is_p2p() and comm.ranks_on_this_device() == 1
    assert comm.exchange_selftest(64, 5) == 0
    assert comm.exchange_selftest(512, 3) == 0      # every slot of the region, both parities, the step counter carried on
    assert comm.exchange_selftest(208, 2000) == 0   # a training run's worth of steps through the same words: every value, every time
    assert not comm.timed_out()


@pytest.mark.parametrize("batch,hidden", [(64, 128), (128, 128), (32, 128), (256, 128), (64, 64), (128, 64)])
@pytest.mark.parametrize("mode", ["graph", "enqueued"])
def test_loopback_step_is_the_single_gpu_step_bit_for_bit(batch, hidden, mode, monkeypatch):
    import taper_amd as T
    if mode == "enqueued":
        monkeypatch.setenv("TAPER_NO_GRAPH", "1")      # the same op list, every launch enqueued by the host instead of replayed
    m = T.Trainer.GRAPH
    steps = 7
    ref = _train(T, batch, steps, m, hidden, None)
    got = _train(T, batch, steps, m, hidden, "loopback")
    comm = got[5]
    assert comm.tail_exchange_ok(batch, 784, hidden, 10)
    assert comm.inkernel_launches() >= steps, comm.inkernel_launches()   # the path under test is the one that ran
    assert not comm.timed_out()
    np.testing.assert_array_equal(got[0], ref[0])
    for a, b in zip(got[1], ref[1]):
        np.testing.assert_array_equal(a, b)
    assert got[2] == ref[2] == 2 * steps
    np.testing.assert_array_equal(got[3], ref[3])
    np.testing.assert_array_equal(got[4], ref[4])


def test_one_rank_communicator_takes_the_single_gpu_step():
    """W = 1: the mean over the ranks is this rank's gradient -- no exchange launch, no exchange code: the two-launch step of one GPU"""
    import taper_amd as T
    ref = _train(T, 128, 5, T.Trainer.GRAPH, 128, None)
    got = _train(T, 128, 5, T.Trainer.GRAPH, 128, "one_rank")
    comm = got[5]
    assert comm.stats()["fused"] == 0 and comm.stats()["inplace"] == 0 and comm.inkernel_launches() == 0
    np.testing.assert_array_equal(got[0], ref[0])
    for a, b in zip(got[1], ref[1]):
        np.testing.assert_array_equal(a, b)
    assert got[2] == ref[2]


def test_shapes_the_exchange_does_not_cover_take_the_three_launch_step():
    import taper_amd as T
    comm = T.Communicator.loopback()
    assert comm.tail_exchange_ok(64, 784, 128, 10) and comm.tail_exchange_ok(512, 784, 64, 10)
    assert not comm.tail_exchange_ok(72, 784, 128, 10)       # not whole 16-row tiles
    assert not comm.tail_exchange_ok(64, 784, 100, 10)       # ragged hidden width
    assert not comm.tail_exchange_ok(64, 784, 256, 10)       # no instance
    assert not comm.tail_exchange_ok(1024, 784, 128, 10)     # th_mlp2_xent's territory
    comm.set_inkernel(False)
    assert not comm.tail_exchange_ok(64, 784, 128, 10)


def _simple_cnn(T):
    C = lambda i, o, s: T.Conv2dReLU(i, o, (3, 3), (1, 1), (1, 1), None, None, True, seed=s)
    return T.Sequential([C(1, 32, 1), T.MaxPool2d((2, 2), (2, 2)), C(32, 64, 2), T.MaxPool2d((2, 2), (2, 2)), T.Flatten(1),
                         T.Linear(3136, 10, True, 3)])


def _train_cnn(T, batch, steps, comm_kind):
    model = _simple_cnn(T)
    opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
    comm = T.Communicator.loopback() if comm_kind == "loopback" else None
    tr = T.Trainer(model, opt, comm=comm, sample_shape=(1, 28, 28))
    x, y = _data(batch * steps, seed=9)
    loader = T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False)
    losses = np.concatenate([tr.run_epoch(loader, T.Trainer.GRAPH)["losses"] for _ in range(2)])
    return losses, [p.data() for p in model.parameters()], opt.t(), opt.moments()[0], opt.moments()[1], comm


@pytest.mark.parametrize("batch", [96, 128, 256, 1024])
def test_loopback_simple_cnn_step_is_the_single_gpu_step_bit_for_bit(batch):
    """the simple CNN (BASELINE configs[2]'s model): th_wide_head_grads_dp through a loopback communicator -- the batch-sums launch pushes,
    polls, adds and resets every one of its 103 slices -- must give the single-GPU step's bits: losses, every parameter, Adam's moments
    and counter (1 024 images: the chain launch is the two-to-a-CU instance)"""
    import taper_amd as T
    steps = 4
    ref = _train_cnn(T, batch, steps, None)
    got = _train_cnn(T, batch, steps, "loopback")
    comm = got[5]
    assert comm.inkernel_launches() >= steps, comm.inkernel_launches()
    assert not comm.timed_out()
    assert comm.stats()["fused"] == 0 and comm.stats()["inplace"] == 0      # no all-reduce launch ran
    np.testing.assert_array_equal(got[0], ref[0])
    for a, b in zip(got[1], ref[1]):
        np.testing.assert_array_equal(a, b)
    assert got[2] == ref[2] == 2 * steps
    np.testing.assert_array_equal(got[3], ref[3])
    np.testing.assert_array_equal(got[4], ref[4])

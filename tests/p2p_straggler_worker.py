"""Two ranks bootstrap a peer-to-peer communicator on GPU 0; rank 1 then never launches the collective.  Rank 0's launch must give
up after its bounded spin (set to 3 s here) instead of hanging the GPU, apply NOTHING (no parameter moves, Adam's counter stays), and the
Trainer must raise -- a second step after the error is a no-op that raises again (tests/test_gpu_dp.py).  TAPER_STRAGGLER_FORM=inplace:
the same through the in-place all-reduce followed by Adam::step (th_adam_step_guarded skips on the communicator's error word)."""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
rank, world = int(os.environ["RANK"]), 2
out = Path(os.environ["TAPER_DP_OUT"])
import taper_amd as T  # noqa: E402
from taper_amd import hip  # noqa: E402
from taper_amd.dist import FileRendezvous, init_data_parallel  # noqa: E402

T.Device.set_device(0)
rdzv = FileRendezvous(rank, world, key=os.environ["TAPER_DP_KEY"], root=str(out), timeout_s=60)
form = os.environ.get("TAPER_STRAGGLER_FORM", "fused")
if form == "inkernel_cnn":     # the simple CNN: the exchange inside its batch-sums launch (th_wide_head_grads_dp)
    Cv = lambda i, o, s: T.Conv2dReLU(i, o, (3, 3), (1, 1), (1, 1), None, None, True, seed=s)
    model = T.Sequential([Cv(1, 32, 1), T.MaxPool2d((2, 2), (2, 2)), Cv(32, 64, 2), T.MaxPool2d((2, 2), (2, 2)), T.Flatten(1),
                          T.Linear(3136, 10, True, 3)])
else:
    model = T.Sequential([T.Linear(784, 128, True, seed=1), T.ReLU(), T.Linear(128, 10, True, seed=2)])
opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
comm = init_data_parallel(T, rdzv, backend="p2p", optimizer=opt)      # includes the collective self-check: both ranks take part
assert not comm.timed_out()
t0 = time.time()
if rank == 0:
    from taper_amd._lib import TaperError
    comm.set_timeout_ms(3000)
    if os.environ.get("TAPER_STRAGGLER_FORM") == "inplace":
        comm.set_fuse_adam(False)       # in-place all-reduce, then Adam::step -- guarded by the communicator's error word
    cnn = form == "inkernel_cnn"
    rows = 128 if cnn else 64
    tr = T.Trainer(model, opt, comm=comm, **({"sample_shape": (1, 28, 28)} if cnn else {}))
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, (rows, 784)).astype(np.float32)
    y = rng.integers(0, 10, rows).astype(np.float32)
    before = [p.data() for p in model.parameters()]
    raised = []
    inkernel = form in ("inkernel", "inkernel_cnn")
    if inkernel:
        # the exchange inside the gradient launch (th_mlp_tail_dp), as an epoch of captured steps runs it: the first step's workgroups all
        # give up after the bound and the lead takes the tick back; every later launch of the epoch -- the next steps' first launches with
        # their deferred updates and ticks, the gradient launches, the final flush -- finds the word up and does nothing
        x4, y4 = np.tile(x, (4, 1)), np.tile(y, 4)
        loader = T.DataLoader(T.MNISTDataset.from_host(x4, y4), rows, False)
        assert cnn or comm.tail_exchange_ok(64, 784, 128, 10)
    for _ in range(2):
        try:
            if inkernel:
                tr.run_epoch(loader, T.Trainer.GRAPH)
            else:
                tr.train_step(T.Tensor(x), T.Tensor(y))          # rank 1 never arrives at this all-reduce
            raised.append("")
        except TaperError as e:
            raised.append(str(e))
    if inkernel:
        assert comm.inkernel_launches() >= 1
    first = time.time() - t0
    unchanged = all(np.array_equal(a, p.data()) for a, p in zip(before, model.parameters()))
    ok = all("timed out waiting for a peer" in r for r in raised) and comm.failed() and comm.timed_out()
    (out / "straggler_result.txt").write_text(f"{int(ok)} {first:.2f} {int(unchanged)} {opt.t()}")
rdzv.barrier()
rdzv.close()

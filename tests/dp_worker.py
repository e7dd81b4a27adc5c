"""One rank of a data-parallel run of the PRODUCT Trainer (tests/test_gpu_dp.py spawns W of these).
env: RANK / LOCAL_RANK / WORLD_SIZE (torchrun's contract), TAPER_DP_OUT (directory), TAPER_DP_STEPS, TAPER_DP_GLOBAL_BATCH,
TAPER_DP_MODE (graph | eager), TAPER_DP_DEVICE (optional: every rank on this device -- the peer-to-peer communicator can share
one GPU, RCCL cannot), TAPER_DP_BACKEND (rccl | p2p), TAPER_DP_MODEL (tests/backends.py builder: mlp_baseline | cnn_reference | ...), TAPER_DP_REPEAT (optional: the
two-epoch optimisation R times over from the same start -- parameters, Adam's moments and t restored in place, the captured graphs and the
communicator kept -- every run compared with the first bit for bit (runs_same / runs_maxdiff / a byte checksum per run for the replica comparison):
tests/test_gpu_repro.py)."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def make_problem(steps, global_batch, seed=11, model="mlp_baseline"):
    from tests import backends
    rng = np.random.default_rng(seed)
    spec = backends.nonzero_biases(getattr(backends, model)(rng), rng)
    x, y = backends.mnist_like(rng, steps * global_batch)
    return spec, x, y


def sample_shape(model):
    return (1, 28, 28) if model.startswith("cnn") else None


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    steps, gb = int(os.environ["TAPER_DP_STEPS"]), int(os.environ["TAPER_DP_GLOBAL_BATCH"])
    out = Path(os.environ["TAPER_DP_OUT"])
    import taper_amd as T
    from taper_amd.dist import FileRendezvous, init_data_parallel
    from tests import backends
    dev = os.environ.get("TAPER_DP_DEVICE")
    T.Device.set_device(int(dev) if dev is not None else int(os.environ.get("LOCAL_RANK", rank)))
    rdzv = FileRendezvous(rank, world, key=os.environ["TAPER_DP_KEY"], root=str(out), timeout_s=120)
    model_name = os.environ.get("TAPER_DP_MODEL", "mlp_baseline")
    spec, x, y = make_problem(steps, gb, model=model_name)
    per = gb // world
    rows = np.concatenate([np.arange(s * gb + rank * per, s * gb + (rank + 1) * per) for s in range(steps)])   # SURVEY 8e partitioning
    H = backends.get("hip")
    model = H.sequential(spec)
    opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
    fine = os.environ.get("TAPER_DP_FINE", "0") == "1"
    comm = init_data_parallel(T, rdzv, backend=os.environ.get("TAPER_DP_BACKEND", "rccl"), optimizer=opt, fine_grained=fine)
    extra_rounds = int(os.environ.get("TAPER_DP_SELFTEST_ROUNDS", "0"))
    selftest_bad = comm.exchange_selftest(16, extra_rounds) if extra_rounds and comm is not None and comm.is_p2p() else 0   # collective
    tr = T.Trainer(model, opt, comm=comm, **({"sample_shape": sample_shape(model_name)} if sample_shape(model_name) else {}))
    loader = T.DataLoader(T.MNISTDataset.from_host(x[rows], y[rows]), per, False)
    mode = T.Trainer.GRAPH if os.environ.get("TAPER_DP_MODE", "graph") == "graph" else T.Trainer.EAGER
    repeat = int(os.environ.get("TAPER_DP_REPEAT", "1"))
    start = [p.data() for p in model.parameters()]
    runs = {}
    for run in range(repeat):
        if run:                                # the same optimisation again from the same start
            for p, a in zip(model.parameters(), start):
                p.set_data(a)
            n_all = sum(a.size for a in start)
            opt.load_state(0, np.zeros(n_all, np.float32), np.zeros(n_all, np.float32))
            loader.reset()
        ep = tr.run_epoch(loader, mode)
        ep2 = tr.run_epoch(loader, mode)          # a second epoch over the same rows: the captured graphs are replayed
        if repeat > 1:
            cur = [np.concatenate([ep["losses"], ep2["losses"]])] + [p.data() for p in model.parameters()]
            if run == 0:
                first = cur
                runs["run0_losses"] = cur[0]
                for i, a in enumerate(cur[1:]):
                    runs[f"run0_p{i}"] = a
            # per run: does anything differ from the first run, bit for bit?  (losses, then each parameter) -- and by how much
            runs.setdefault("runs_same", []).append([bool(np.array_equal(a, b)) for a, b in zip(cur, first)])
            runs.setdefault("runs_maxdiff", []).append([float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()) for a, b in zip(cur, first)])
            runs.setdefault("runs_crc", []).append([int(np.frombuffer(np.ascontiguousarray(a).tobytes(), np.uint8).astype(np.uint64).sum()) for a in cur])
    if comm is not None and comm.is_p2p() and comm.timed_out():
        raise SystemExit(f"rank {rank}: a peer never arrived at the all-reduce")
    st = comm.stats() if comm is not None and comm.is_p2p() else dict(inplace=-1, fused=-1)
    st["inkernel"] = comm.inkernel_launches() if comm is not None and comm.is_p2p() else -1
    st["form"] = comm.exchange_form() if comm is not None and comm.is_p2p() else 0
    import ctypes as C
    from taper_amd._lib import hip as lib
    mlp2_calls = C.c_int64()
    lib.th_debug_mlp2_calls(C.byref(mlp2_calls))   # host-side calls of th_mlp2_xent on this thread (eager steps + captures)
    np.savez(out / f"rank{rank}.npz", fine=int(fine), mlp2_calls=mlp2_calls.value, launches_inplace=st["inplace"], launches_fused=st["fused"], launches_inkernel=st["inkernel"], exchange_form=st["form"], selftest_bad=selftest_bad, losses=np.concatenate([ep["losses"], ep2["losses"]]), t=opt.t(),
             **{f"p{i}": p.data() for i, p in enumerate(model.parameters())}, **{k: np.asarray(v) for k, v in runs.items()})
    rdzv.barrier()
    rdzv.close()


if __name__ == "__main__":
    main()

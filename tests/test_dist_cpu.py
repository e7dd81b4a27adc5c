"""N > 1 path on CPU (world_size 2): (1) the data-parallel arithmetic --
rank-sharded rows, all-reduce(sum)/W of the flat parameter-gradient vector
equals the single-process full-batch gradient, and replicated Adam keeps the
ranks bit-identical -- exercised with torch.distributed's gloo backend and the
oracle as the compute engine (the HIP path needs a GPU); (2) the file
rendezvous bench.py uses for its control plane."""
import multiprocessing as mp
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from tests import backends
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ob = backends.get("oracle")
    ob.set_zero_sentinel(True)
    rng = np.random.default_rng(0)                       # identical init + data on every rank
    spec = backends.nonzero_biases(backends.mlp_baseline(rng), rng)
    model = ob.sequential(spec)
    opt = ob.m.Adam(model.parameters(), 1e-3, None, None, 1e-4)
    B = 64
    per = B // world
    out = []
    for step in range(3):
        x, y = backends.mnist_like(rng, B)
        xs, ys = x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]   # SURVEY.md 8e partitioning
        loss, _, _, grads = ob.forward_backward(model, xs, ys, (per, 784))
        flat = np.concatenate([g.reshape(-1) for g in grads])
        t = torch.from_numpy(flat.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)          # what th_allreduce_sum_scale does over RCCL
        flat = (t.numpy() * np.float32(1.0 / world)).astype(np.float32)
        off = 0
        for p in model.parameters():
            n = p.numel()
            p.set_grad(flat[off:off + n])
            off += n
        opt.step()
        opt.zero_grad()
        out.append((float(loss), flat.copy(), [p.data().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


def test_data_parallel_grads_equal_full_batch_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    # single-process reference on the full batch
    from tests import backends
    ob = backends.get("oracle")
    ob.set_zero_sentinel(True)
    rng = np.random.default_rng(0)
    spec = backends.nonzero_biases(backends.mlp_baseline(rng), rng)
    model = ob.sequential(spec)
    opt = ob.m.Adam(model.parameters(), 1e-3, None, None, 1e-4)
    for step in range(3):
        x, y = backends.mnist_like(rng, 64)
        loss, _, _, grads = ob.forward_backward(model, x, y, (64, 784))
        flat = np.concatenate([g.reshape(-1) for g in grads])
        for p, g in zip(model.parameters(), grads):
            p.set_grad(g)
        opt.step()
        opt.zero_grad()
        l0, f0, w0 = res[0][step]
        l1, f1, w1 = res[1][step]
        np.testing.assert_array_equal(f0, f1)                      # all-reduce returns identical bits on all ranks
        for a, b in zip(w0, w1):
            np.testing.assert_array_equal(a, b)                    # replicas stay bit-identical
        np.testing.assert_allclose(f0, flat, rtol=1e-4, atol=1e-7)  # mean of shard-mean grads == full-batch grad
        assert (l0 + l1) / 2 == pytest.approx(loss, rel=1e-5)
        for a, p in zip(w0, model.parameters()):
            np.testing.assert_allclose(a, p.data(), rtol=1e-4, atol=2e-5)


def _rdzv_worker(rank, world, key, root, q):
    sys.path.insert(0, str(ROOT))
    from taper_amd.dist import FileRendezvous
    r = FileRendezvous(rank, world, key=key, root=root, timeout_s=60)
    got = r.broadcast_bytes(b"x" * 128 if rank == 0 else None)
    r.barrier()
    mx = r.all_reduce_max(1.5 + rank)
    sm = r.all_reduce_sum(10.0 * (rank + 1))
    ga = r.all_gather_bytes(bytes([rank]))
    r.close()
    q.put((rank, got, mx, sm, ga))


@pytest.mark.parametrize("world", [2, 4])
def test_file_rendezvous(world, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rdzv_worker, args=(r, world, "t", str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, got, mx, sm, ga in res:
        assert got == b"x" * 128
        assert mx == 1.5 + world - 1
        assert sm == 10.0 * world * (world + 1) / 2
        assert ga == [bytes([r]) for r in range(world)]
    assert not (tmp_path / "taper_rdzv_t").exists()

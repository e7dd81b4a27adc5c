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
    assert not list(tmp_path.glob("taper_rdzv_*"))          # last one out removed the directory


def test_rendezvous_directory_is_private_and_owned(tmp_path):
    """/dev/shm is world-writable: the directory must be 0700, a real directory and ours"""
    from taper_amd.dist import FileRendezvous
    r = FileRendezvous(0, 1, key="own", root=str(tmp_path))
    st = os.lstat(r.dir)
    assert st.st_mode & 0o777 == 0o700 and st.st_uid == os.getuid()
    r.close()
    # a directory someone else could write into is refused
    loose = tmp_path / f"taper_rdzv_{os.getuid()}_loose"
    loose.mkdir(mode=0o777)
    os.chmod(loose, 0o777)
    with pytest.raises(PermissionError):
        FileRendezvous(0, 1, key="loose", root=str(tmp_path))
    # so is a symlink planted under the predictable name
    target = tmp_path / "elsewhere"
    target.mkdir(mode=0o700)
    (tmp_path / f"taper_rdzv_{os.getuid()}_link").symlink_to(target)
    with pytest.raises(PermissionError):
        FileRendezvous(0, 1, key="link", root=str(tmp_path))


class _FakeCommunicator:
    """stands in for taper_amd.Communicator (RCCL needs GPUs): records what init_data_parallel hands it"""
    made = []

    def __init__(self, world, rank, uid):
        self.world, self.rank, self.uid = world, rank, uid
        _FakeCommunicator.made.append(self)

    @staticmethod
    def unique_id():
        return bytes([os.getpid() % 251]) * 128          # differs per process: only rank 0's may be used


class _FakeT:
    Communicator = _FakeCommunicator


def _init_dp_worker(rank, world, root, q):
    sys.path.insert(0, str(ROOT))
    # what torch.distributed.run / bench.py's self-spawn put in the environment
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29555")
    from taper_amd.dist import FileRendezvous, env_rank_world, init_data_parallel
    r, w, lr = env_rank_world()
    rdzv = FileRendezvous(r, w, key="dp", root=root, timeout_s=60)
    comm = init_data_parallel(_FakeT, rdzv)
    # the shard of a 1024-row global batch this rank trains on (rows [r*B/W, (r+1)*B/W))
    per = 1024 // w
    rows = (r * per, (r + 1) * per)
    mx = rdzv.all_reduce_max(float(r))
    rdzv.close()
    q.put((rank, (r, w, lr), comm.world, comm.rank, comm.uid, _FakeCommunicator.unique_id(), rows, mx))


@pytest.mark.parametrize("world", [2, 4])
def test_init_data_parallel_control_flow(world, tmp_path):
    """the product's launch-side logic on CPU: env -> (rank, world), rank 0's RCCL unique id reaches every rank, every rank builds
    its communicator as (world, rank, that id), shards tile the global batch"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_init_dp_worker, args=(r, world, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    uid0 = res[0][5]                                     # what rank 0's unique_id() returns
    covered = []
    for rank, env, cw, cr, uid, _own, rows, mx in res:
        assert env == (rank, world, rank)
        assert (cw, cr) == (world, rank)
        assert uid == uid0 and len(uid) == 128           # broadcast from rank 0, never a rank's own
        assert mx == float(world - 1)
        covered.append(rows)
    assert covered == [(r * 1024 // world, (r + 1) * 1024 // world) for r in range(world)]
    from taper_amd.dist import init_data_parallel
    assert init_data_parallel(_FakeT, None) is None      # a single rank has no communicator


class _FakeP2P:
    """stands in for the peer-to-peer Communicator: the self-check's verdict is scripted per attempt (TAPER_FAKE_P2P = e.g. "fail,ok":
    the pooled arena fails on rank 1 only, the fine-grained one passes)"""
    attempts = 0

    def __init__(self, world, rank, fine=None):
        self.world, self.rank, self.fine, self.timeouts = world, rank, fine, []

    def export_arena(self, optimizer, fine_grained=False):
        self.fine = fine_grained
        return bytes([self.rank + 1]) * 192

    def connect(self, blobs):
        assert len(blobs) == 192 * self.world and all(blobs[192 * r] == r + 1 for r in range(self.world))

    def set_timeout_ms(self, ms):
        self.timeouts.append(ms)

    def self_check(self, optimizer, rounds=3):
        plan = os.environ["TAPER_FAKE_P2P"].split(",")
        verdict = plan[min(_FakeP2P.attempts, len(plan) - 1)]
        _FakeP2P.attempts += 1
        return verdict == "ok" or self.rank == 0          # a failure is seen by rank 1 only: rank 0 must still learn of it


class _FakeTP2P:
    class Communicator(_FakeCommunicator):
        @staticmethod
        def p2p(world, rank):
            return _FakeP2P(world, rank)


def _auto_worker(rank, world, root, plan, q):
    sys.path.insert(0, str(ROOT))
    os.environ["TAPER_FAKE_P2P"] = plan
    from taper_amd.dist import FileRendezvous, init_data_parallel
    rdzv = FileRendezvous(rank, world, key="auto" + plan.replace(",", ""), root=root, timeout_s=60)
    info = {}
    comm = init_data_parallel(_FakeTP2P, rdzv, backend="auto", optimizer=object(), info=info)
    rdzv.close()
    q.put((rank, type(comm).__name__, getattr(comm, "fine", None), getattr(comm, "timeouts", None), info))


@pytest.mark.parametrize("plan,want", [("ok", ("_FakeP2P", False, "p2p")), ("fail,ok", ("_FakeP2P", True, "p2p-finegrained")),
                                       ("fail,fail", ("Communicator", None, "rccl"))])
def test_auto_backend_falls_back_p2p_then_finegrained_then_rccl(plan, want, tmp_path):
    """world 2: a self-check that fails on ONE rank moves EVERY rank on -- first to the fine-grained gradient arena, then to RCCL -- and the
    bootstrap's short wait bound is replaced by the training bound once the check has passed"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_auto_worker, args=(r, 2, str(tmp_path), plan, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, kind, fine, timeouts, info in res:
        assert (kind, fine, info["backend"]) == want, (rank, kind, fine, info)
        if kind == "_FakeP2P":
            assert timeouts == [20000, 120000]
        if want[2] != "p2p":
            assert "self-check failed" in info["why"]


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` without a launcher self-spawns N ranks -- and must not quietly run a smaller job when the box
    has fewer GPUs (it used to fall back to 1 rank)"""
    import subprocess
    from taper_amd import hip
    if hip.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "visible" in r.stderr
    assert r.stdout.strip() == ""

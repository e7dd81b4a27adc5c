"""Single-node rendezvous for one-process-per-GPU data parallelism.

The data path of DP training is native: RCCL all-reduce on the th_ctx stream
(include/taper_hip.h: th_comm_*).  What is left for the host is control
plane only -- shipping RCCL's 128-byte unique id from rank 0, barriers around
the timed region and a max/sum of a few floats.  This module does that through
a directory in /dev/shm (or /tmp), keyed by the launcher's PID + MASTER_PORT,
so the bench process never has to import torch: PyTorch bundles its own copy
of the HIP runtime and RCCL, and two HIP runtimes in one process corrupt each
other's state at exit.

Launch contract (same as torch.distributed.run provides): RANK, WORLD_SIZE,
LOCAL_RANK, MASTER_PORT in the environment, all ranks on ONE node.
"""
from __future__ import annotations

import os
import struct
import time
from pathlib import Path


def _start_ticks(pid: int) -> str:
    """start time of a process (clock ticks since boot, /proc/<pid>/stat field 22): with the PID it names one launcher
    for good, so a directory left behind by a crashed run of a recycled PID is never mistaken for this run's"""
    try:
        return Path(f"/proc/{pid}/stat").read_text().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        return "0"


class FileRendezvous:
    def __init__(self, rank: int, world: int, key: str | None = None, root: str | None = None, timeout_s: float = 600.0):
        self.rank, self.world, self.timeout_s = int(rank), int(world), timeout_s
        if key is None:
            # all workers of one torchrun share the agent as parent and the master port
            key = f"{os.getppid()}_{_start_ticks(os.getppid())}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}"
        base = Path(root or ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"))
        self.dir = base / f"taper_rdzv_{os.getuid()}_{key}"
        # /dev/shm and /tmp are world-writable and the name is predictable: the directory is private (0700), must be a real
        # directory (not a symlink someone planted) and must belong to this user -- otherwise another local user could feed
        # the ranks a forged RCCL unique id or hold the barriers
        os.makedirs(base, exist_ok=True)          # a caller-supplied root that does not exist yet (the check below is on OUR directory)
        try:
            os.mkdir(self.dir, 0o700)
        except FileExistsError:
            pass
        st = os.lstat(self.dir)
        import stat as _stat
        if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise PermissionError(f"rendezvous directory {self.dir} is not a private directory of uid {os.getuid()} "
                                  f"(mode {st.st_mode & 0o7777:o}, uid {st.st_uid}); refusing to use it")
        self._seq = 0

    # -- primitives ---------------------------------------------------------
    def _path(self, name: str, rank: int | None = None) -> Path:
        return self.dir / (f"{name}.{rank}" if rank is not None else name)

    def _publish(self, path: Path, payload: bytes) -> None:
        tmp = path.with_suffix(path.suffix + f".tmp{os.getpid()}")
        tmp.write_bytes(payload)
        os.replace(tmp, path)  # atomic: readers never see a partial file

    def _wait_for(self, path: Path) -> bytes:
        deadline = time.monotonic() + self.timeout_s
        spins = 0
        while True:
            try:
                return path.read_bytes()
            except FileNotFoundError:
                pass
            spins += 1
            if spins > 2000:
                time.sleep(0.0002)
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous: rank {self.rank} timed out waiting for {path}")

    def _next(self, tag: str) -> str:
        self._seq += 1
        return f"{tag}{self._seq:06d}"

    # -- collectives on tiny host values ----------------------------------------
    def broadcast_bytes(self, payload: bytes | None, src: int = 0) -> bytes:
        name = self._next("bcast")
        if self.rank == src:
            assert payload is not None
            self._publish(self._path(name), payload)
            return payload
        return self._wait_for(self._path(name))

    def all_gather_bytes(self, payload: bytes) -> list[bytes]:
        name = self._next("gather")
        self._publish(self._path(name, self.rank), payload)
        return [self._wait_for(self._path(name, r)) for r in range(self.world)]

    def barrier(self) -> None:
        self.all_gather_bytes(b"1")

    def all_reduce_max(self, v: float) -> float:
        return max(struct.unpack("d", b)[0] for b in self.all_gather_bytes(struct.pack("d", float(v))))

    def all_reduce_sum(self, v: float) -> float:
        return sum(struct.unpack("d", b)[0] for b in self.all_gather_bytes(struct.pack("d", float(v))))

    def close(self) -> None:
        """last one out removes the directory"""
        try:
            self.barrier()
            if self.rank == 0:
                time.sleep(0.05)
                for p in self.dir.iterdir():
                    try:
                        p.unlink()
                    except OSError:
                        pass
                self.dir.rmdir()
        except (OSError, TimeoutError) as e:     # best effort: a peer that died leaves the directory behind; say so
            import sys
            print(f"taper rendezvous: could not clean up {self.dir}: {e}", file=sys.stderr)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def _env_ms(name: str, default: int) -> int:
    """a positive millisecond bound from the environment; anything else (unset, 0, negative, not a number) is the default -- what
    th_comm_init_p2p does with the same variable, so that a rank with a malformed value does not raise after the collective self-check"""
    try:
        v = int(os.environ.get(name, ""))
    except ValueError:
        return default
    return v if v > 0 else default


def init_data_parallel(T, rdzv: FileRendezvous | None, backend: str = "rccl", optimizer=None, fine_grained: bool = False, info: dict | None = None):
    """-> taper_amd.Communicator (or None for a single rank).
    backend "rccl": RCCL all-reduce (ring / tree over xGMI) -- rank 0's unique id is broadcast.
    backend "p2p":  the one-shot peer-to-peer all-reduce fused with Adam (th_allreduce_adam): every rank registers the
                    gradient arena of `optimizer`, the IPC blobs are all-gathered, every rank maps its peers.  fine_grained: the arena is
                    moved into fine-grained device memory first (never cached in a peer's L2).
    backend "auto": p2p on the pooled arena; if its multi-round self-check fails, p2p on a fine-grained arena; then RCCL.  `info` (a
                    dict) receives {"backend", "why"}."""
    if rdzv is None or rdzv.world == 1:
        return None
    if backend == "auto":
        why = []
        for fine in (False, True):
            try:
                comm = init_data_parallel(T, rdzv, "p2p", optimizer, fine_grained=fine)
                if info is not None:
                    info.update(backend="p2p-finegrained" if fine else "p2p", why="; ".join(why))
                return comm
            except RuntimeError as e:
                why.append(f"p2p{' (fine-grained arena)' if fine else ''}: {e}")
        if info is not None:
            info.update(backend="rccl", why="; ".join(why))
        return init_data_parallel(T, rdzv, "rccl")
    if backend == "p2p":
        if optimizer is None:
            raise ValueError("init_data_parallel(backend='p2p') needs the optimizer whose gradient arena is reduced")
        # Every step below is collective: a rank that fails (no IPC, no peer access) still takes part in the exchanges, and
        # every rank learns the verdict -- nobody is left waiting in a barrier for a peer that has given up.
        comm, err = None, ""
        try:
            comm = T.Communicator.p2p(rdzv.world, rdzv.rank)
            blob = comm.export_arena(optimizer, fine_grained)
        except Exception as e:   # noqa: BLE001 -- reported to every rank below
            blob, err = b"\0" * 192, f"export: {e}"
        blobs = rdzv.all_gather_bytes(blob)
        if not err and any(b == b"\0" * 192 for b in blobs):
            err = "a peer could not export its arena"
        if not err:
            try:
                comm.connect(b"".join(blobs))
            except Exception as e:   # noqa: BLE001
                err = f"connect: {e}"
        if rdzv.all_reduce_sum(1.0 if err else 0.0) > 0:
            del comm
            rdzv.barrier()        # every rank has dropped its mappings before anybody frees or re-homes an arena (as below)
            raise RuntimeError(f"peer-to-peer communicator unavailable (rank {rdzv.rank}: {err or 'a peer failed'})")
        # Known patterns through the real arena on the real links, several rounds through the SAME addresses and through BOTH kernels: a
        # rank that sees stale or no peer data must not train.  The waits inside are short here (a bootstrap peer is either there or
        # gone) and go back to the training bound afterwards.
        comm.set_timeout_ms(_env_ms("TAPER_P2P_BOOT_TIMEOUT_MS", 20000))
        try:
            ok = comm.self_check(optimizer, int(os.environ.get("TAPER_P2P_SELFCHECK_ROUNDS", "3")))   # (0: skipped -- a measurement knob)
        except Exception as e:   # noqa: BLE001
            ok, err = False, f"self-check: {e}"
        bad = rdzv.all_reduce_sum(0.0 if ok else 1.0)
        if bad > 0:
            del comm
            rdzv.barrier()        # every rank has dropped its mappings before anybody frees or re-homes an arena
            raise RuntimeError(f"peer-to-peer all-reduce self-check failed on {int(bad)} rank(s) (rank {rdzv.rank}: "
                               f"{'ok' if ok else (err or 'mismatch or timeout')})")
        comm.set_timeout_ms(_env_ms("TAPER_P2P_TIMEOUT_MS", 120000))
        return comm
    if backend != "rccl":
        raise ValueError(f"unknown data-parallel backend {backend!r}")
    uid = rdzv.broadcast_bytes(T.Communicator.unique_id() if rdzv.rank == 0 else None, src=0)
    return T.Communicator(rdzv.world, rdzv.rank, uid)

#!/usr/bin/env python3
"""bench.py -- throughput of taper's training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, or self-spawned)

A "step" is one pass of the hot path over one batch of synthetic MNIST-shaped
input already resident in HBM: gather batch -> forward -> softmax cross-entropy
-> backward -> (N > 1: RCCL all-reduce of the flat grad arena) -> Adam -> log,
replayed as one hipGraph (host: C++ tape in libtaper_host.so; device: the HIP
kernels of libtaper_hip.so through the C ABI).  Headline workload at N = 1 =
BASELINE.json configs[1]: MLP 784-128-10, batch 64, Adam(1e-3, wd 1e-4); at N > 1 = configs[3]: the same MLP at
128 rows per GPU (global 1024 at N = 8), with the single-GPU figure of the same per-GPU batch printed beside it.

Prints ONE JSON line (rank 0) with the BASELINE metric plus `roofline` (the
dominant kernel timed live with HIP events), `cpu_baseline` (the C restatement of the reference's CPU path, timed on
this box's host cores) and, at N = 1, `workloads`: the other BASELINE configs (both CNNs at batch 256, the example
MLP, the 4096-wide Linear stack and the 4096^3 sgemm variants), each with its step time, per-kernel roofline
fractions and its own CPU baseline.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3    # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_32x32x2_f32)

WORKLOADS = {
    # name: (model builder key, per-GPU batch, sample_shape, lr)
    "mlp_784-128-10_b64": ("mlp_baseline", 64, None, 1e-3),           # BASELINE configs[1]
    "mlp_784-128-10_b128": ("mlp_baseline", 128, None, 1e-3),         # configs[3]: 1024 over 8 GPUs
    "mlp_784-128-64-10_b256": ("mlp_example", 256, None, 1e-3),       # examples/train_mnist.rs
    "cnn_reference_b256": ("cnn_reference", 256, (1, 28, 28), 1e-2),  # examples/train_mnist_cnn.rs (configs[2] family)
    "cnn_simple_b256": ("cnn_simple", 256, (1, 28, 28), 1e-2),        # BASELINE configs[2]
}


def build_model(T, key, seed=1):
    L, R = T.Linear, T.ReLU
    if key == "mlp_baseline":
        return T.Sequential([L(784, 128, True, seed), R(), L(128, 10, True, seed + 1)])
    if key == "mlp_example":
        return T.Sequential([L(784, 128, True, seed), R(), L(128, 64, True, seed + 1), R(), L(64, 10, True, seed + 2)])
    C = lambda i, o, s: T.Conv2dReLU(i, o, (3, 3), (1, 1), (1, 1), None, None, True, seed=s)
    if key == "cnn_reference":
        return T.Sequential([C(1, 32, 1), C(32, 32, 2), T.MaxPool2d((2, 2), (2, 2)), C(32, 64, 3), C(64, 64, 4),
                             T.MaxPool2d((2, 2), (2, 2)), C(64, 128, 5), T.AdaptiveAvgPool2d((1, 1)), T.Flatten(1),
                             L(128, 128, True, 6), R(), L(128, 64, True, 7), R(), L(64, 10, True, 8)])
    if key == "cnn_simple":
        return T.Sequential([C(1, 32, 1), T.MaxPool2d((2, 2), (2, 2)), C(32, 64, 2), T.MaxPool2d((2, 2), (2, 2)),
                             T.Flatten(1), L(3136, 10, True, 3)])
    raise ValueError(key)


def algorithmic_step(key, batch):
    """algorithmic flops / HBM bytes of one step (each tensor touched once)."""
    if key == "mlp_baseline":
        p = 101_770
        return 409_088 * batch + 14 * p, 3_140 * batch + 28 * p
    if key == "mlp_example":
        p = 109_386
        return 454_400 * batch + 14 * p, 3_140 * batch + 28 * p
    return None, None


# ---------------------------------------------------------------------------- distributed plumbing
# One process per GPU (launched by torch.distributed.run, which only sets RANK /
# LOCAL_RANK / WORLD_SIZE / MASTER_*).  The data path is RCCL on our own HIP
# stream; the control plane (unique-id broadcast, barriers, max over ranks) is
# taper_amd.dist.FileRendezvous.  torch is deliberately NOT imported here: it
# bundles a second HIP runtime + RCCL, and two runtimes in one process corrupt
# each other (observed: "double free or corruption" at exit).
def init_dist(n_gpus):
    from taper_amd.dist import FileRendezvous, env_rank_world
    rank, world, _ = env_rank_world()
    if world == 1:
        return None, 0, 1
    assert world == n_gpus, f"--gpus {n_gpus} but WORLD_SIZE={world}"
    return FileRendezvous(rank, world), rank, world


def barrier_sync(rdzv, T):
    T.Device.sync()      # hipStreamSynchronize on the stream every kernel / collective is enqueued on
    if rdzv is not None:
        rdzv.barrier()


def make_comm(rdzv, T, backend, optimizer, shared_device=False):
    """backend auto: the one-shot peer-to-peer all-reduce fused with Adam when every rank can map its peers and the multi-round
    self-check passes (first on the pooled gradient arena, then on a fine-grained one), RCCL otherwise (the reason is reported)"""
    from taper_amd.dist import init_data_parallel
    if rdzv is None:
        return None, "none"
    info = {}
    comm = init_data_parallel(T, rdzv, backend=backend, optimizer=optimizer, info=info)
    kind = info.get("backend", backend)
    why = f" ({info['why']})" if info.get("why") else ""
    if kind.startswith("p2p"):
        # (ranks sharing ONE device -- TAPER_BENCH_SHARE_DEVICE, the one-GPU harness -- map each other's arenas through IPC: no link is crossed)
        link = "between processes on ONE shared device (IPC mappings, no xGMI link crossed)" if shared_device else "over xGMI"
        form = {1: "one-shot", 2: "two-shot"}.get(comm.exchange_form(), "none")
        return comm, (f"p2p: {form} exchange inside the gradient launch (th_mlp_tail_dp) where the step allows, else th_allreduce_adam; {link}"
                      + (", fine-grained gradient arena" if kind.endswith("finegrained") else "") + why)
    return comm, "rccl ncclAllReduce(avg)" + why


def dp_on_device(T, key, lr, ds, per_gpu_batch=128, steps=4000, rounds=3, sample_shape=None):
    """N = 1: what the data-parallel step of BASELINE configs[3] (128 rows per GPU) costs ON the device, before a link is crossed.  Four
    Trainers on the same rows, timed alternately (min of `rounds`):
      single      no communicator: the two-launch step of one GPU
      one_rank    a 1-rank peer-to-peer communicator: nothing to exchange, the same two launches
      loopback    W = 2 with this process as its own peer: the exchange inside the gradient launch (th_mlp_tail_dp) runs every push, flag,
                  poll and load of its protocol, through local memory; results are the single-GPU step's bit for bit
      three_launch  the r05 form with one rank: gradient launch WITHOUT fused updates, then the one-shot all-reduce + Adam launch
    on_device_efficiency_ceiling = single / loopback: the weak-scaling efficiency the step could reach if links cost nothing."""
    def make(kind):
        m = build_model(T, key)
        o = T.Adam(m.parameters(), lr, None, None, 1e-4)
        c = None
        if kind == "loopback":
            c = T.Communicator.loopback()
        elif kind in ("one_rank", "three_launch"):
            c = T.Communicator.p2p(1, 0)
            c.connect(c.export_arena(o))
            if kind == "three_launch":
                c.set_inkernel(False)
        t = T.Trainer(m, o, comm=c, **({"sample_shape": sample_shape} if sample_shape else {}))
        l = T.DataLoader(ds, per_gpu_batch, False)
        run_steps(T, t, l, 300)
        _KEEP_ALIVE.append((t, o, m, l, c))
        return t, l, c
    kinds = ("single", "one_rank", "loopback", "three_launch")
    runs = {k: make(k) for k in kinds}
    best = {k: None for k in kinds}
    for _ in range(rounds):
        for k in kinds:
            t, l, _c = runs[k]
            T.Device.sync()
            t0 = time.perf_counter()
            run_steps(T, t, l, steps)
            T.Device.sync()
            us = (time.perf_counter() - t0) / steps * 1e6
            best[k] = us if best[k] is None else min(best[k], us)
    lb = runs["loopback"][2]
    out = dict(per_gpu_batch=per_gpu_batch, steps=steps, rounds=rounds,
               single_gpu_step_us=round(best["single"], 3), one_rank_step_us=round(best["one_rank"], 3),
               loopback_two_rank_step_us=round(best["loopback"], 3), three_launch_one_rank_step_us=round(best["three_launch"], 3),
               on_device_efficiency_ceiling=round(best["single"] / best["loopback"], 4),
               one_rank_ceiling=round(best["single"] / best["one_rank"], 4),
               three_launch_ceiling=round(best["single"] / best["three_launch"], 4),
               loopback_inkernel_launches=lb.inkernel_launches(), loopback_timed_out=lb.timed_out(),
               note="one device: every word of the exchange goes through local memory; a link adds its latency per exchange and the gradient arena "
                    "(0.4 MB for the MLP, 0.2 MB for the simple CNN) per peer of transfer")
    return out


def opt_total(T, opt):
    """padded length of the optimizer's flat arenas = the all-reduce size in floats (include/taper_host.h: tp_optim_total)"""
    from taper_amd._lib import host, tp_check
    n = C.c_int64()
    tp_check(host.tp_optim_total(opt._h, C.byref(n)), "tp_optim_total")
    return int(n.value)


def replicas_identical(rdzv, model):
    """crc32 of every rank's weights, compared on every rank (data-parallel replicas must stay bit-identical: SURVEY 8e)"""
    import zlib
    crc = 0
    for prm in model.parameters():
        crc = zlib.crc32(prm.data().tobytes(), crc)
    return len(set(rdzv.all_gather_bytes(crc.to_bytes(4, "little")))) == 1


# ---------------------------------------------------------------------------- the timed loop
def run_steps(T, trainer, loader, steps):
    """exactly `steps` graph-replayed steps; returns samples processed"""
    done, samples = 0, 0
    while done < steps:
        ep = trainer.run_epoch(loader, T.Trainer.GRAPH, max_steps=min(steps - done, loader.num_batches()))
        done += ep["num_batches"]
        samples += ep["total_samples"]
    return samples






def hip_structs():
    from taper_amd import hip   # imported lazily: bench.py must not load the native library before the rendezvous is up
    return hip


class StepKernels:
    """The two launches of the fused MLP 784-128-10 step, issued through the C ABI on raw
    buffers exactly as the Trainer's graph issues them, replayed as graph chains with / without
    each launch: every kernel is timed in situ (its operands were just written by the previous
    launch, like in the real step), on the stream the kernels run on.

    Algorithmic bytes per launch (every tensor touched once, fp32), B = batch:
      K1 sgemm_small16_tick<true,true,16>  H = relu(X.W1^T + b1), the carried Adam(W2, b2) of the
                                      previous step and the step counter:
                                      4*(784B + 128*784 + 128 + 128B) + 28*1290 [g, p, m, v read; p, m, v written]
      K2 mlp_tail_exact_kernel<8,2>   logits / loss / dlogits / dH (registers only), dW2, db2, dW1 = dZ1^T.X, db1,
                                      Adam(W1, b1) in the epilogue, step log:
                                      4*(784B + 128B + B) [X, H, targets] + 4*1290 [W2, b2] + 4*1290 [dW2, db2]
                                      + 4*100480 [dW1, db1 written] + 24*100480 [p, m, v read + written] + 16
    """
    IN, HID, OUT = 784, 128, 10

    def __init__(self, ctx, batch):
        self.ctx, self.B = ctx, batch
        B, IN, HID, OUT = batch, self.IN, self.HID, self.OUT
        rng = np.random.default_rng(0)
        f = lambda *shape: ctx.upload(rng.uniform(-0.05, 0.05, shape).astype(np.float32))
        self.x, self.y = ctx.upload(rng.uniform(0, 1, (B, IN)).astype(np.float32)), ctx.upload(rng.integers(0, OUT, B).astype(np.float32))
        n1, n2 = HID * IN + HID, OUT * HID + OUT
        self.p1, self.g1, self.m1, self.v1 = f(n1), ctx.zeros(n1), ctx.zeros(n1), ctx.zeros(n1)
        self.p2, self.g2, self.m2, self.v2 = f(n2), ctx.zeros(n2), ctx.zeros(n2), ctx.zeros(n2)
        self.h = ctx.empty(B * HID)
        self.loss, self.nc = ctx.empty(1), ctx.empty(1)
        self.metrics, self.state = ctx.zeros(2 * 4096), ctx.upload(np.zeros(2, np.int64))
        self.tick, self.lr = ctx.upload(np.array([0, 0], np.int32)), ctx.upload(np.array([1e-3], np.float32))
        adam = lambda p, m, v, off: hip_structs().AdamFuse(int(p) + 4 * off, int(m) + 4 * off, int(v) + 4 * off, int(self.tick), int(self.lr),
                                               0.9, 0.999, 1e-8, 1e-4)
        self.w1f, self.b1f = adam(self.p1, self.m1, self.v1, 0), adam(self.p1, self.m1, self.v1, HID * IN)
        self.carried = (hip_structs().AdamSlice * 2)(
            hip_structs().AdamSlice(int(self.g2), OUT * HID, adam(self.p2, self.m2, self.v2, 0)),
            hip_structs().AdamSlice(int(self.g2) + 4 * OUT * HID, OUT, adam(self.p2, self.m2, self.v2, OUT * HID)))
        self.kernels = [
            ("sgemm_small16_tick<true, true, 16>", "K1 layer-1 forward (+bias, ReLU) + carried Adam(W2,b2) of the previous step + step counter",
             4 * (IN * B + HID * IN + HID + HID * B) + 28 * n2, 2 * B * IN * HID + 14 * n2),
            ("mlp_tail_exact_kernel<8, 2, false, 4>", "K2 head (Linear(128,10) + softmax-xent + dW2/db2 + step log) + layer-1 backward (dW1, db1) + Adam(W1,b1) epilogue",
             4 * (IN * B + HID * B + B) + 8 * n2 + 4 * n1 + 24 * n1 + 16, 3 * 2 * B * HID * OUT + 2 * B * IN * HID + 14 * n1),
        ]

    def _k1(self):
        c, B = self.ctx, self.B
        c.call("th_linear_fwd_ex", self.x, self.p1, int(self.p1) + 4 * self.HID * self.IN, self.h, B, self.IN, self.HID, 1, self.carried, 2,
               self.tick)

    def _k2(self):
        c, B, n1, n2 = self.ctx, self.B, self.HID * self.IN, self.OUT * self.HID
        c.call("th_mlp_tail", self.x, self.h, self.p2, int(self.p2) + 4 * n2, self.y, B, self.IN, self.HID, self.OUT, self.loss, self.nc,
               self.g1, int(self.g1) + 4 * n1, self.g2, int(self.g2) + 4 * n2, None, None, self.metrics, 4096, self.state, 1,
               C.byref(self.w1f), C.byref(self.b1f))

    def _capture(self, launches, steps=16):
        """`steps` back-to-back steps as one hipGraph (<= 48 kernel nodes)"""
        c = self.ctx
        c.graph_begin()
        try:
            for _ in range(steps):
                for k in launches:
                    k()
        finally:
            g = c.graph_end()
        for _ in range(3):
            c.graph_launch(g)
        c.sync()
        return g

    def _replay_us(self, g, steps=16, reps=120, inner=12):
        """us per step (HIP events on the ctx stream), at most `inner` replays in flight"""
        from taper_amd import hip
        c = self.ctx
        e0, e1 = hip.Event(), hip.Event()
        ms = 0.0
        for _ in range(reps // inner):
            c.record(e0)
            for _ in range(inner):
                c.graph_launch(g)
            c.record(e1)
            ms += hip.Ctx.elapsed_ms(e0, e1)     # synchronises on e1
        return ms * 1e3 / (reps // inner * inner * steps)

    def measure(self):
        """In-situ duration of each launch = (time of the 2-launch step) - (time of the step with that
        launch left out), both replayed as graph chains: the launch keeps its real neighbours, and the
        figure includes the dependent-launch boundary it adds -- which is also what rocprofv3's
        kernel-trace duration covers here (its per-kernel averages sum to the step time).
        All graphs stay alive until the end: under rocprofv3 --kernel-trace a replay issued after a
        hipGraphExecDestroy crashes in the profiler on this ROCm."""
        ks = [self._k1, self._k2]
        graphs = [self._capture(ks)] + [self._capture([k for j, k in enumerate(ks) if j != i]) for i in range(len(ks))]
        full = self._replay_us(graphs[0])
        out = []
        for i, (name, what, nbytes, flops) in enumerate(self.kernels):
            t = full - self._replay_us(graphs[1 + i])
            out.append(dict(kernel=name, role=what, us_per_launch=round(t, 3), alg_bytes_per_launch=nbytes,
                            achieved_GBps=round(nbytes / (t * 1e-6) / 1e9, 2), hbm_frac=round(nbytes / (t * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                            mfma_tflops=round(flops / (t * 1e-6) / 1e12, 3)))
        for g in graphs:
            self.ctx.graph_destroy(g)
        return dict(step_us=round(full, 3), kernels=out)


def batch_sweep(T, build_model, key, lr, dataset, batches=(64, 256, 1024, 4096, 16384, 60000)):
    """SURVEY.md 8(d) batch sweep of the same model / optimizer / step on one GPU (fresh model per point; the
    headline `value` stays BASELINE configs[1], batch 64, at the contract's K steps; these are long runs):
    where the path stops being launch-bound."""
    out = []
    for b in batches:
        if b > dataset.len():
            continue
        model = build_model(T, key)
        opt = T.Adam(model.parameters(), lr, None, None, 1e-4)
        trainer = T.Trainer(model, opt)
        loader = T.DataLoader(dataset, b, False)
        steps = min(80_000, max(200, 20_000_000 // b))   # >= 150 ms of work per point: short bursts run below the steady clocks
        run_steps(T, trainer, loader, max(steps // 8, 3))
        T.Device.sync()
        t0 = time.perf_counter()
        samples = run_steps(T, trainer, loader, steps)
        T.Device.sync()
        dt = time.perf_counter() - t0
        flops, nbytes = algorithmic_step(key, b)
        out.append(dict(batch=b, steps=steps, ms_per_step=round(dt / steps * 1e3, 5), samples_per_s=round(samples / dt, 1),
                        epochs_per_s=round(samples / dt / 60000.0, 2),
                        mfma_frac=None if flops is None else round(flops / (dt / steps) / 1e12 / MFMA_F32_PEAK_TF, 4),
                        hbm_frac=None if nbytes is None else round(nbytes / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4)))
        del trainer, opt, model, loader
    return out


def pmc_traffic(workload, kernel):
    """bytes per launch from the committed rocprofv3 PMC passes of this command (FETCH_SIZE x2 + WRITE_SIZE,
    collected separately: tools/profile_bench.sh); counters cannot be read from inside the process."""
    if workload != "mlp_784-128-10_b64":
        return None, None
    files = sorted((ROOT / "profiles").glob("r*_mlp_b64_pmc_traffic.json"))
    if not files:
        return None, None
    kernels = json.loads(files[-1].read_text())["kernels"]
    for exact in (True, False):      # the same template instance first (the trace of the default run holds other workloads' instances too)
        for name, rec in kernels.items():
            if (kernel in name) if exact else (kernel.split("<")[0] + "<" in name):
                return rec["traffic_bytes_per_launch"], f"profiles/{files[-1].name} (a committed rocprofv3 --pmc pass of this command; NOT measured in this run)"
    return None, None


# ---------------------------------------------------------------------------- CPU baseline (reported, never the target)
_FAST_ORACLE = {}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(key, batch, sample_shape, lr, budget_s=8.0):
    """The reference's CPU path on THIS box's host cores, on a bounded sample of the same workload (about budget_s
    seconds of steps).  kind "port": the Rust crate cannot be built here (no cargo/rustc), so this is the C
    restatement of the reference (oracle/) in its cpu_baseline build (`make -C oracle fast`, gcc -O3 -march=native =
    .cargo/config.toml's target-cpu=native):
      sgemm            packed, register-blocked FMA micro-kernel in matrixmultiply's style, ONE thread -- the reference's
                       default feature set has no `threading` (Cargo.toml:16; src/gemm.rs:102-117)
      im2col windows, max/avg-pool planes (fwd + bwd), batch gather
                       thread-parallel like the reference's rayon loops (src/tensor.rs:1420,1491,1552,1614,1745;
                       src/data/mnist.rs:291); thread count = the best of a short trial over {all logical CPUs, 16, 1}
      everything else  the reference's scalar / auto-vectorised loops, one thread (Adam, softmax chain, transposes, bias)
    plus, when a vendor CBLAS exists on the box (numpy's bundled OpenBLAS), the `--features blas` analogue with every
    sgemm routed through cblas_sgemm (src/gemm.rs:32-47) at the vendor's default thread count."""
    from oracle import oracle as O
    from tests import backends
    if "so" not in _FAST_ORACLE:
        _FAST_ORACLE["so"] = O.build_fast(tempfile.mkdtemp(prefix="taper_oracle_fast_"))   # -march=native for THIS box
        O.use_library(_FAST_ORACLE["so"])
        _FAST_ORACLE["cblas"] = O.find_cblas()
    lib = O.lib
    ncpu = os.cpu_count() or 1
    shape = (batch, 784) if sample_shape is None else (batch,) + tuple(sample_shape)
    rng = np.random.default_rng(1)
    x, y = backends.mnist_like(rng, batch * 8)

    def fresh():
        ob = backends.get("oracle")
        ob.set_zero_sentinel(True)
        model = ob.sequential(getattr(backends, key)(np.random.default_rng(1)))
        return model, O.Adam(model.parameters(), lr, None, None, 1e-4)

    def timed(threads, seconds, cblas=None):
        lib.ot_baseline_set_threads(int(threads))
        lib.ot_baseline_set_cblas(cblas[0] if cblas else None, cblas[1] if cblas else 0)
        model, opt = fresh()
        model.run_steps(opt, x, y, shape, 1)      # warm-up (allocations, thread pool)
        n, chunk, t0 = 0, 1, time.perf_counter()
        while True:
            model.run_steps(opt, x, y, shape, chunk)
            n += chunk
            dt = time.perf_counter() - t0
            if dt > seconds:
                break
            chunk = max(1, min(int(n / dt * 0.25) or 1, 4096))   # ~0.25 s of steps per C call
        lib.ot_baseline_set_cblas(None, 0)
        return n, dt

    trials = {}
    for th in sorted({ncpu, min(16, ncpu), 1}, reverse=True):
        n, dt = timed(th, min(1.0, budget_s / 8))
        trials[th] = round(n * batch / dt, 1)
    best = max(trials, key=trials.get)
    n, dt = timed(best, budget_s * 0.55)
    out = dict(value=n * batch / dt, unit="samples/s", cores=best, kind="port", cpu_model=_cpu_model(), nproc=ncpu,
               threads={"sgemm": 1, "im2col/pool/gather loops": best, "adam/softmax/transpose/bias": 1},
               threads_tried_samples_per_s={str(k): v for k, v in trials.items()},
               sample=f"{n} steps of {key} batch {batch} (get_batch + fwd + xent + bwd + Adam) in {dt:.1f}s; C restatement of the reference "
                      f"CPU path, gcc -O3 -march=native -fopenmp: packed FMA sgemm (matrixmultiply-style, 1 thread) + rayon-style "
                      f"plane loops on {best} thread(s)")
    if _FAST_ORACLE["cblas"] is not None:
        addr, ilp64, what, _keep = _FAST_ORACLE["cblas"]
        try:
            n2, dt2 = timed(best, budget_s * 0.25, (addr, ilp64))
            out["blas_feature"] = dict(value=n2 * batch / dt2, unit="samples/s", lib=what, sgemm_threads="vendor default",
                                       sample=f"{n2} steps in {dt2:.1f}s with every sgemm through cblas_sgemm (src/gemm.rs:32-47)")
        except Exception as e:   # reported, never required
            out["blas_feature"] = dict(value=None, error=str(e))
    return out


# ---------------------------------------------------------------------------- the other BASELINE configs (N = 1)
def _time_launches(ctx, call, reps, warm=5):
    """us per launch: `reps` back-to-back launches on the ctx stream between two HIP events"""
    from taper_amd import hip
    for _ in range(warm):
        call()
    e0, e1 = hip.Event(), hip.Event()
    ctx.record(e0)
    for _ in range(reps):
        call()
    ctx.record(e1)
    return hip.Ctx.elapsed_ms(e0, e1) * 1e3 / reps


def conv_layer_kernels(ctx, layers, reps=60):
    """The conv launches of a CNN step at batch 256, each timed on its own (back-to-back launches of the layer through the
    C ABI, HIP events on the ctx stream).  Algorithmic work per launch: 18*C_in*C_out*H*W flop per sample;
    bytes = input + output (+ weights once), the pooled form writes a quarter of the output.  Bound: fp32 MFMA
    (157.3 TF) for C_in >= 8; the single-channel conv1 (4.4 flop/B) is bound by HBM."""
    rng = np.random.default_rng(0)
    out = []
    for (name, n, ci, hw, co, pool) in layers:
        x = ctx.upload(rng.uniform(0, 1, (n, ci, hw, hw)).astype(np.float32))
        w = ctx.upload(rng.uniform(-0.1, 0.1, (co, ci, 3, 3)).astype(np.float32))
        b = ctx.upload(rng.uniform(-0.1, 0.1, co).astype(np.float32))
        ho = hw // 2 if pool else hw
        y = ctx.empty(n * co * ho * ho)
        if pool:
            call = lambda: ctx.call("th_conv3x3_pool2_fwd", x, w, b, y, n, ci, hw, hw, co, 1, 1)
        else:
            call = lambda: ctx.call("th_conv3x3_fwd", x, w, b, y, n, ci, hw, hw, co, 1, 0, 1)
        us = _time_launches(ctx, call, reps)
        cfg = (C.c_int * 6)()
        ctx.call("th_debug_last_conv_config", C.cast(cfg, C.c_void_p))
        flops = 18.0 * ci * co * hw * hw * n
        nbytes = 4.0 * (n * ci * hw * hw + n * co * ho * ho + 9 * ci * co + co)
        tf, gbs = flops / (us * 1e-6) / 1e12, nbytes / (us * 1e-6) / 1e9
        if ci == 1:
            kern = "conv1_pool2_kernel" if pool else "conv1_kernel"
        elif cfg[1] == 8:               # one layer through the chain's compiled tile mapping (conv_chain.hip)
            kern = "conv_layer_chain_kernel<%d, %d, %d, %d>" % (hw, ci, co, 1 if pool else 0)
        elif cfg[1] in (2, 3, 4, 5):    # image-resident: <channel tiles, pixel tiles per wave, pooled epilogue, waves>, cfg[4] whole images per workgroup
            kern = "conv3x3_img_kernel<%d, %d, %s, %d%s>" % (cfg[0], cfg[2], "true" if pool else "false", 4 if cfg[1] in (3, 5) else 8,
                                                                 ", patch geometry compiled in" if cfg[1] >= 4 else "")
        else:
            kern = f"conv3x3_mfma_kernel<{cfg[0]}, false, 8, {'true' if pool else 'false'}, {cfg[2]}, {'true' if cfg[1] else 'false'}>"
        bound = "hbm" if ci == 1 else "mfma"
        out.append(dict(kernel=kern, layer=name, us_per_launch=round(us, 2), alg_flops_per_launch=flops, alg_bytes_per_launch=nbytes,
                        bound=bound, achieved=round(gbs if bound == "hbm" else tf, 2), peak=HBM_PEAK_GBS if bound == "hbm" else MFMA_F32_PEAK_TF,
                        unit="GB/s" if bound == "hbm" else "TFLOP/s",
                        frac=round((gbs / HBM_PEAK_GBS) if bound == "hbm" else (tf / MFMA_F32_PEAK_TF), 4)))
        del x, w, b, y
    return out


CNN_CHAINS = {   # (c_in, c_out, post) per stage -- include/taper_hip.h TH_CHAIN_*: 0 none, 1 max-pool 2x2, 2 global mean
    "cnn_simple": ("conv_chain_simple_kernel", [(1, 32, 1), (32, 64, 1)]),
    "cnn_reference": ("conv_chain_reference_kernel", [(1, 32, 0), (32, 32, 1), (32, 64, 0), (64, 64, 1), (64, 128, 2)]),
}


def conv_chain_kernel(ctx, key, n=256, reps=100):
    """The ONE conv launch the Trainer's captured CNN step issues (th_conv_chain_fwd: every Conv2dReLU / pool row in front of the
    classifier, one image per workgroup, maps resident in LDS), timed like the layers below.  Algorithmic work = the sum of the
    layers' 18*C_in*C_out*H*W flop per sample; bytes = the images in, the last stage's output out, the weights once.
    Bound: fp32 MFMA."""
    from taper_amd import hip
    name, spec = CNN_CHAINS[key]
    rng = np.random.default_rng(0)
    x = ctx.upload(rng.uniform(0, 1, (n, 1, 28, 28)).astype(np.float32))
    bufs = [(ctx.upload(rng.uniform(-0.1, 0.1, (co, ci, 3, 3)).astype(np.float32)), ctx.upload(rng.uniform(-0.1, 0.1, co).astype(np.float32)))
            for ci, co, _ in spec]
    stages, ns = hip.conv_stages([(w, b, co, post) for (w, b), (_, co, post) in zip(bufs, spec)])
    sp = C.cast(stages, C.c_void_p)
    hw, flops, wbytes = 28, 0.0, 0.0
    for ci, co, post in spec:
        flops += 18.0 * ci * co * hw * hw * n
        wbytes += 4.0 * (9 * ci * co + co)
        hw = hw // 2 if post == 1 else (1 if post == 2 else hw)
    co_last = spec[-1][1]
    y, cnt = ctx.empty(n * co_last * hw * hw), ctx.empty(n * co_last)
    us = _time_launches(ctx, lambda: ctx.call("th_conv_chain_fwd", x, sp, ns, y, cnt, n, 1, 28, 28), reps, warm=10)
    ctx.sync()
    nbytes = 4.0 * (n * 784 + n * co_last * hw * hw) + wbytes
    tf = flops / (us * 1e-6) / 1e12
    return dict(kernel=name, layer="conv chain: " + ", ".join("%d->%d%s" % (ci, co, ("", "+pool", "+mean")[post]) for ci, co, post in spec),
                us_per_launch=round(us, 2), alg_flops_per_launch=flops, alg_bytes_per_launch=nbytes, bound="mfma", achieved=round(tf, 2),
                peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=round(tf / MFMA_F32_PEAK_TF, 4))


class _ChainHead(C.Structure):
    """include/taper_hip.h: th_chain_head"""
    _fields_ = [("d_w", C.c_void_p), ("d_bias", C.c_void_p), ("d_targets", C.c_void_p), ("classes", C.c_int), ("d_dl", C.c_void_p),
                ("d_rowstat", C.c_void_p), ("d_cbpart", C.c_void_p), ("d_tick", C.c_void_p)]


def chain_head_kernels(ctx, n=256, classes=10, reps=100):
    """The TWO launches of the simple CNN's captured step (BASELINE configs[2]): th_conv_chain_head_fwd -- the conv rows with the classifier's
    row-parallel part in the last epilogue; bound fp32 MFMA, work = the conv layers' flops + 4*K*classes per image for logits and dX -- and
    th_wide_head_grads -- dW = dl^T X, db, loss, the conv bias, Adam in the epilogues; bound HBM, bytes = X + the row records in, the
    gradients out, 24 B/parameter of Adam state."""
    from taper_amd import hip
    from taper_amd.hip import AdamFuse
    _, spec = CNN_CHAINS["cnn_simple"]
    rng = np.random.default_rng(0)
    k, c_last = 64 * 49, 64
    x = ctx.upload(rng.uniform(0, 1, (n, 1, 28, 28)).astype(np.float32))
    bufs = [(ctx.upload(rng.uniform(-0.1, 0.1, (co, ci, 3, 3)).astype(np.float32)), ctx.upload(rng.uniform(-0.1, 0.1, co).astype(np.float32)))
            for ci, co, _ in spec]
    stages, ns = hip.conv_stages([(w, b, co, post) for (w, b), (_, co, post) in zip(bufs, spec)])
    sp = C.cast(stages, C.c_void_p)
    w, b = ctx.upload(rng.uniform(-0.02, 0.02, (classes, k)).astype(np.float32)), ctx.upload(rng.uniform(-0.1, 0.1, classes).astype(np.float32))
    yt = ctx.upload(rng.integers(0, classes, n).astype(np.float32))
    ymap, dl, rs, cbp = ctx.empty(n * k), ctx.empty(n * 16), ctx.empty(n * 2), ctx.empty(n * c_last)
    tick, lrd = ctx.upload(np.array([1, 0], np.int32)), ctx.upload(np.array([1e-2], np.float32))
    head = _ChainHead(int(w), int(b), int(yt), classes, int(dl), int(rs), int(cbp), None)
    us1 = _time_launches(ctx, lambda: ctx.call("th_conv_chain_head_fwd", x, sp, ns, ymap, n, 1, 28, 28, C.byref(head)), reps, warm=10)
    mom = [(ctx.zeros(sz), ctx.zeros(sz)) for sz in (classes * k, classes, c_last)]
    fz = lambda p, i: AdamFuse(int(p), int(mom[i][0]), int(mom[i][1]), int(tick), int(lrd), 0.9, 0.999, 1e-8, 1e-4)
    f = [fz(w, 0), fz(b, 1), fz(bufs[1][1], 2)]
    dw, db, gcb, loss = ctx.empty(classes * k), ctx.empty(classes), ctx.empty(c_last), ctx.empty(1)
    us2 = _time_launches(ctx, lambda: ctx.call("th_wide_head_grads", ymap, dl, rs, cbp, n, k, classes, c_last, dw, db, gcb, loss, None, None, 0,
                                                None, 0, C.byref(f[0]), C.byref(f[1]), C.byref(f[2])), reps, warm=10)
    ctx.sync()
    hw, flops, wbytes = 28, 4.0 * k * classes * n, 4.0 * classes * k
    for ci, co, post in spec:
        flops += 18.0 * ci * co * hw * hw * n
        wbytes += 4.0 * (9 * ci * co + co)
        hw = hw // 2 if post == 1 else hw
    nb1 = 4.0 * (n * 784 + n * k + n * (16 + 2 + c_last)) + wbytes
    tf = flops / (us1 * 1e-6) / 1e12
    params = classes * k + classes + c_last
    nb2 = 4.0 * (n * k + n * (16 + 2 + c_last)) + 28.0 * params
    gbs = nb2 / (us2 * 1e-6) / 1e9
    return [dict(kernel=f"conv_chain_simple_kernel<true, {10 if classes <= 10 else 16}>", layer="conv chain 1->32+pool, 32->64+pool + classifier rows",
                 us_per_launch=round(us1, 2), alg_flops_per_launch=flops, alg_bytes_per_launch=nb1, bound="mfma", achieved=round(tf, 2),
                 peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=round(tf / MFMA_F32_PEAK_TF, 4), in_step=True),
            dict(kernel="wide_grads_kernel", layer="dW, db, loss, conv bias + Adam", us_per_launch=round(us2, 2), alg_flops_per_launch=2.0 * n * k * classes,
                 alg_bytes_per_launch=nb2, bound="hbm", achieved=round(gbs, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                 in_step=True)]


def chain_mlp3_kernels(ctx, n=256, classes=10, reps=100):
    """The TWO launches of the reference CNN's captured step (examples/train_mnist_cnn.rs): th_conv_chain_mlp3_xent's first -- the five conv
    rows with their pools and the three-layer classifier's row in the last epilogue; bound fp32 MFMA, work = the conv layers' flops + the row's
    six matrix-vector products -- and its second (th_mlp3_xent's gradient launch: dW / db of the three layers over the batch, the conv bias,
    loss, Adam in the epilogues; latency-bound: 0.1 MB of parameters)."""
    from taper_amd import hip
    _, spec = CNN_CHAINS["cnn_reference"]
    rng = np.random.default_rng(0)
    x = ctx.upload(rng.uniform(0, 1, (n, 1, 28, 28)).astype(np.float32))
    yt = ctx.upload(rng.integers(0, classes, n).astype(np.float32))
    bufs = [(ctx.upload(rng.uniform(-0.1, 0.1, (co, ci, 3, 3)).astype(np.float32)), ctx.upload(rng.uniform(-0.1, 0.1, co).astype(np.float32)))
            for ci, co, _ in spec]
    stages, ns = hip.conv_stages([(w, b, co, post) for (w, b), (_, co, post) in zip(bufs, spec)])
    sp = C.cast(stages, C.c_void_p)
    dims = ((128, 128), (64, 128), (classes, 64))
    layers, keep = (hip.Mlp3Layer * 3)(), []
    for l, (o, i) in enumerate(dims):
        b = (ctx.upload(rng.uniform(-0.1, 0.1, (o, i)).astype(np.float32)), ctx.zeros(o), ctx.empty(o * i), ctx.empty(o))
        keep.append(b)
        layers[l] = hip.Mlp3Layer(int(b[0]), int(b[1]), int(b[2]), int(b[3]), None, None, o)
    lp = C.cast(layers, C.c_void_p)
    means, cnt, gx, gb, loss, nc = ctx.empty(n * 128), ctx.empty(n * 128), ctx.empty(n * 128), ctx.empty(128), ctx.empty(1), ctx.empty(1)
    gap = hip.Mlp3Gap(int(cnt), int(gb), 49, None)
    gp = C.cast(C.pointer(gap), C.c_void_p)
    call = lambda: ctx.call("th_conv_chain_mlp3_xent", x, sp, ns, means, cnt, n, 1, 28, 28, yt, lp, gx, loss, nc, None, 0, None, 0, None, gp)
    us_both = _time_launches(ctx, call, reps, warm=10)
    try:
        hip.hip.th_debug_chain_mlp3_only(1)
        us1 = _time_launches(ctx, call, reps, warm=10)
    finally:
        hip.hip.th_debug_chain_mlp3_only(0)
    ctx.sync()
    hw, flops, wbytes = 28, 0.0, 0.0
    for ci, co, post in spec:
        flops += 18.0 * ci * co * hw * hw * n
        wbytes += 4.0 * (9 * ci * co + co)
        hw = hw // 2 if post == 1 else (1 if post == 2 else hw)
    p_cls = sum(o * i + o for o, i in dims)
    flops1 = flops + 6.0 * n * sum(o * i for o, i in dims)                       # forward + the two backward products of every layer, per row
    nb1 = 4.0 * (n * 784 + n * (2 * 128 + 2 * 128 + 2 * 64 + classes + 2)) + wbytes + 4.0 * p_cls
    tf = flops1 / (us1 * 1e-6) / 1e12
    us2 = max(us_both - us1, 1e-3)
    nb2 = 4.0 * n * (3 * 128 + 2 * 64 + classes + 128) + 28.0 * (p_cls + 128)
    gbs = nb2 / (us2 * 1e-6) / 1e9
    return [dict(kernel="conv_chain_reference_kernel<false, true>", layer="conv chain 1->32, 32->32+pool, 32->64, 64->64+pool, 64->128+mean + classifier rows",
                 us_per_launch=round(us1, 2), alg_flops_per_launch=flops1, alg_bytes_per_launch=nb1, bound="mfma", achieved=round(tf, 2),
                 peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=round(tf / MFMA_F32_PEAK_TF, 4), in_step=True),
            dict(kernel="mlp3_grads_kernel", layer="dW / db of the classifier, conv bias, loss (+ Adam in the step)", us_per_launch=round(us2, 2),
                 alg_flops_per_launch=2.0 * n * sum(o * i for o, i in dims), alg_bytes_per_launch=nb2, bound="hbm", achieved=round(gbs, 2),
                 peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), in_step=True,
                 note="MARGINAL cost in the step: both launches back to back minus the first alone -- the kernel's own duration (rocprofv3, "
                      "profiles/r06_cnn_kernel_stats.txt: ~6 us) plus the launch boundary it adds")]


CNN_LAYERS = {
    # (layer, batch, C_in, H = W, C_out, fused 2x2 max-pool epilogue) -- the launches the Trainer's captured step issues
    "cnn_simple": [("conv1+pool 1->32 @28", 256, 1, 28, 32, True), ("conv2+pool 32->64 @14", 256, 32, 14, 64, True)],
    "cnn_reference": [("conv1 1->32 @28", 256, 1, 28, 32, False), ("conv2+pool 32->32 @28", 256, 32, 28, 32, True),
                      ("conv3 32->64 @14", 256, 32, 14, 64, False), ("conv4+pool 64->64 @14", 256, 64, 14, 64, True),
                      ("conv5 64->128 @7", 256, 64, 7, 128, False)],
}


_KEEP_ALIVE = []


def trainer_workload(T, name, dataset, steps=None):
    """one of WORKLOADS through the Trainer's captured step (what the headline measures, another model / batch)"""
    key, batch, sample_shape, lr = WORKLOADS[name]
    model = build_model(T, key)
    opt = T.Adam(model.parameters(), lr, None, None, 1e-4)
    trainer = T.Trainer(model, opt, sample_shape=sample_shape)
    loader = T.DataLoader(dataset, batch, False)
    steps = steps or 2000
    run_steps(T, trainer, loader, 260)              # untimed: records every graph size of the replay ladder
    T.Device.sync()
    t_w = time.perf_counter()                       # ... and >= 0.25 s of stepping: the CPU legs in between leave the part idle,
    while time.perf_counter() - t_w < 0.25:         # and it takes tens of ms of load to come back to its running clocks
        run_steps(T, trainer, loader, 200)
        T.Device.sync()
    t0 = time.perf_counter()
    samples = run_steps(T, trainer, loader, steps)
    T.Device.sync()
    dt = time.perf_counter() - t0
    flops, nbytes = algorithmic_step(key, batch)
    eager = os.environ.get("TAPER_NO_GRAPH") == "1"
    rec = dict(workload=name, per_gpu_batch=batch, optimizer=f"Adam(lr={lr}, wd=1e-4)", steps=steps, ms_per_step=round(dt / steps * 1e3, 5),
               samples_per_s=round(samples / dt, 1),
               step="eager enqueue, one host launch per kernel (under rocprofv3: NOT the measurement)" if eager else "hipGraph replay of gather+fwd+xent+bwd+adam+log")
    if flops:
        rec["step_roofline"] = {"alg_flops_per_step": flops, "alg_bytes_per_step": nbytes,
                                "hbm_frac": round(nbytes / (dt / steps) / 1e9 / HBM_PEAK_GBS, 6),
                                "mfma_frac": round(flops / (dt / steps) / 1e12 / MFMA_F32_PEAK_TF, 6)}
    _KEEP_ALIVE.append((trainer, opt, model, loader))   # (rocprofv3 on ROCm 7.2 crashes in the next hipGraphLaunch after a hipGraphExecDestroy)
    return rec, key, batch, sample_shape, lr


CNN_FWD_FLOPS = {"cnn_reference": 43.85e6, "cnn_simple": 7.74e6}   # SURVEY 8d, per sample


def cnn_batch_sweep(T, key, sample_shape, lr, dataset, batches=(1024, 4096)):
    """SURVEY 8d's sweep for the CNN steps (r04 review: a single point at batch 256): the same captured Trainer step at larger batches.
    One image per workgroup in the chain launch, so a batch of 1 024 is four rounds of workgroups per CU: launch, drain and the image
    load amortise, the per-image phases do not overlap (no persistent form yet).  `mfma_frac_fwd` prices the forward conv flops only
    (faithful mode: the conv weights do not train, quirk Q2)."""
    out = []
    for b in batches:
        model = build_model(T, key)
        opt = T.Adam(model.parameters(), lr, None, None, 1e-4)
        trainer = T.Trainer(model, opt, sample_shape=sample_shape)
        loader = T.DataLoader(dataset, b, False)
        steps = max(40, 200_000 // b)
        run_steps(T, trainer, loader, max(steps // 4, 20))
        T.Device.sync()
        t0 = time.perf_counter()
        samples = run_steps(T, trainer, loader, steps)
        T.Device.sync()
        dt = time.perf_counter() - t0
        out.append(dict(batch=b, steps=steps, ms_per_step=round(dt / steps * 1e3, 5), samples_per_s=round(samples / dt, 1),
                        mfma_frac_fwd=round(CNN_FWD_FLOPS[key] * samples / dt / 1e12 / MFMA_F32_PEAK_TF, 4)))
        _KEEP_ALIVE.append((trainer, opt, model, loader))
    return out


def sgemm_kernels(ctx, size=4096, reps=120):
    """BASELINE configs[4] / north_star "Linear-layer GEMM at >= 60 % MFMA peak on 4096^3 fp32": the three products of a
    Linear layer's step -- forward X.W^T (NT), dX = dZ.W (NN), dW = dZ^T.X (TN, accumulating: beta = 1) -- through
    th_sgemm (= src/gemm.rs:72-119 argument semantics), >= 100 back-to-back launches each (the part needs ~100 ms of
    matrix load to settle its clocks).  2mnk flop, 4(mk + kn + mn) B (+ 4mn when beta != 0) per launch."""
    rng = np.random.default_rng(0)
    m = n = k = size
    a = ctx.upload(rng.uniform(-1, 1, m * k).astype(np.float32))
    b = ctx.upload(rng.uniform(-1, 1, k * n).astype(np.float32))
    c = ctx.zeros(m * n)
    out = []
    for name, ta, tb, beta in (("NT (forward X.W^T)", 0, 1, 0.0), ("NN (dX = dZ.W)", 0, 0, 0.0), ("TN (dW += dZ^T.X)", 1, 0, 1.0)):
        us = _time_launches(ctx, lambda: ctx.call("th_sgemm", ta, tb, m, n, k, 1.0, a, b, beta, c), reps, warm=10)
        flops, nbytes = 2.0 * m * n * k, 4.0 * (m * k + k * n + m * n * (2 if beta else 1))
        tf = flops / (us * 1e-6) / 1e12
        out.append(dict(kernel="sgemm_tile<128, ...>", variant=name, m=m, n=n, k=k, beta=beta, us_per_launch=round(us, 1), alg_flops_per_launch=flops,
                        alg_bytes_per_launch=nbytes, bound="mfma", achieved=round(tf, 2), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s",
                        frac=round(tf / MFMA_F32_PEAK_TF, 4)))
    return out


def linear_stack_workload(T, layers=4, width=4096, batch=4096, steps=12, warmup=4):
    """BASELINE configs[4]: `layers` x (Linear(4096, 4096) + ReLU) + Linear(4096, 10), batch 4096, fp32: Tape::reset ->
    forward -> cross_entropy_loss -> backward -> Adam::step through the op-by-op host API (every product is ~1 ms: no
    step graph).  Algorithmic flops: (3L - 1) products of 2*B*W*W (the first layer has no dX) + the
    classifier's three + 14 / parameter for Adam."""
    L, W, B = layers, width, batch
    mods = []
    for i in range(L):
        mods += [T.Linear(W, W, True, 1 + i), T.ReLU()]
    mods.append(T.Linear(W, 10, True, 1 + L))
    model = T.Sequential(mods)
    opt = T.Adam(model.parameters(), 1e-4, None, None, 1e-4)
    rng = np.random.default_rng(7)
    x = T.Tensor(rng.uniform(0, 1, (B, W)).astype(np.float32))
    y = T.Tensor(rng.integers(0, 10, B).astype(np.float32))

    def step():          # the reference-literal order (examples/train_mnist.rs:91-119): backward, then one arena-wide Adam launch
        T.Tape.reset()
        opt.zero_grad()
        loss = T.cross_entropy_loss(model.forward(x), y)
        loss.backward()
        opt.step()
        return loss

    for _ in range(warmup):
        step()
    T.Device.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    T.Device.sync()
    dt = (time.perf_counter() - t0) / steps
    params = L * (W * W + W) + W * 10 + 10
    flops = (3 * L - 1) * 2.0 * B * W * W + 3 * 2.0 * B * W * 10 + 14.0 * params
    rec = dict(workload=f"linear_stack_{W}x{L}_b{B}", per_gpu_batch=B, steps=steps, ms_per_step=round(dt * 1e3, 4), samples_per_s=round(B / dt, 1),
               step="eager op-by-op host API, the reference's loop: Tape::reset, forward, cross_entropy_loss, backward, Adam::step",
               alg_flops_per_step=flops,
               tflops=round(flops / dt / 1e12, 2), frac_of_mfma_peak=round(flops / dt / 1e12 / MFMA_F32_PEAK_TF, 4),
               loss_last=round(float(loss.data()[0]), 5))
    del model, opt, x, y
    T.Tape.reset()
    return rec


def mlp2_workload(ctx, batch=16384, reps=150):
    """The large-batch MLP step (th_mlp2_xent, SURVEY 8d's sweep top) launch by launch: each of its three kernels alone between HIP events on
    the ctx stream (th_debug_mlp2_only), the rows read through a shuffled index vector of a resident 60 000-row set like a Trainer step's.
    Algorithmic work: rows and dW1 2 B 784 128 flop each (+ the classifier's 3 x 2 B 128 10 in the rows launch); finish: the K slices of dW1
    (one workgroup per CU: 256 // 7 slices of 128 x 784 floats) + the row blocks' partials + 28 B per parameter of Adam traffic."""
    from taper_amd import hip
    from taper_amd.hip import AdamFuse, RowSource
    rng = np.random.default_rng(3)
    inf, hid, c, n_rows = 784, 128, 10, 60000
    data = ctx.upload((rng.integers(0, 256, (n_rows, inf))).astype(np.float32) / np.float32(255.0))
    labels = ctx.upload(rng.integers(0, c, n_rows).astype(np.float32))
    idx = ctx.upload(rng.permutation(n_rows).astype(np.int32))
    dev = dict(w1=ctx.upload((rng.uniform(-1, 1, (hid, inf)) * np.sqrt(2.0 / inf)).astype(np.float32)), b1=ctx.zeros(hid),
               w2=ctx.upload(rng.uniform(-0.3, 0.3, (c, hid)).astype(np.float32)), b2=ctx.zeros(c))
    mom = {k: (ctx.zeros(n), ctx.zeros(n)) for k, n in (("w1", hid * inf), ("b1", hid), ("w2", c * hid), ("b2", c))}
    tick, dlr = ctx.upload(np.array([0, 0], np.int32)), ctx.upload(np.array([1e-3], np.float32))
    fuses = [AdamFuse(int(dev[k]), int(mom[k][0]), int(mom[k][1]), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4) for k in ("w1", "b1", "w2", "b2")]
    out = dict(dw1=ctx.empty(hid * inf), db1=ctx.empty(hid), dw2=ctx.empty(c * hid), db2=ctx.empty(c), loss=ctx.empty(1), nc=ctx.empty(1))
    state = ctx.upload(np.array([0, 0], np.int64))
    src = RowSource(int(data), int(labels), int(idx), state.offset(8), n_rows, n_rows)

    def step():
        ctx.call("th_mlp2_xent", C.byref(src), batch, inf, hid, c, dev["w1"], dev["b1"], dev["w2"], dev["b2"], out["dw1"], out["db1"], out["dw2"],
                 out["db2"], out["loss"], out["nc"], None, 0, None, 0, tick, *[C.byref(f) for f in fuses])
    P = hid * inf + hid + c * hid + c
    rt = 64 if batch >= 12288 else (32 if batch > 4096 else 16)
    n_blk = 2 * -(-batch // 32) if rt == 16 else -(-batch // rt)
    kz = max(1, min(256 // 7, max(1, n_blk * rt // 64)))   # two 32-row chunks per slice at least (mlp2.hip)
    gemm = 2.0 * batch * inf * hid
    specs = [("mlp2_rows_kernel<%d, %d, 4>" % (rt, 8 if rt == 16 else 4), "rows: X W1^T + b1, ReLU, classifier, masked dZ1", 1, "mfma", gemm + 3 * 2.0 * batch * hid * c,
              4.0 * (batch * inf + hid * inf + batch * hid)),
             ("mlp2_dw1_kernel8<4, true>", "dW1 = dZ1^T X over %d K slices" % kz, 2, "mfma", gemm, 4.0 * (batch * inf + batch * hid + kz * hid * inf)),
             ("mlp2_finish_kernel", "fixed-order sums + Adam", 3, "hbm", 14.0 * P, 4.0 * (kz * hid * inf + n_blk * (c * hid + hid + 18)) + 28.0 * P)]
    us_step = _time_launches(ctx, step, reps, warm=20)
    kernels = []
    try:
        for name, role, which, bound, flops, nbytes in specs:
            hip.hip.th_debug_mlp2_only(which)
            us = _time_launches(ctx, step, reps, warm=10)
            tf, gbs = flops / (us * 1e-6) / 1e12, nbytes / (us * 1e-6) / 1e9
            kernels.append(dict(kernel=name, role=role, in_step=True, us_per_launch=round(us, 2), alg_flops_per_launch=flops, alg_bytes_per_launch=nbytes,
                                bound=bound, achieved=round(tf if bound == "mfma" else gbs, 2), peak=MFMA_F32_PEAK_TF if bound == "mfma" else HBM_PEAK_GBS,
                                unit="TFLOP/s" if bound == "mfma" else "GB/s",
                                frac=round(tf / MFMA_F32_PEAK_TF if bound == "mfma" else gbs / HBM_PEAK_GBS, 4)))
    finally:
        hip.hip.th_debug_mlp2_only(0)
    flops = 409088.0 * batch + 14.0 * P
    return dict(workload=f"mlp_784-128-10_b{batch}", per_gpu_batch=batch, ms_per_step=round(us_step * 1e-3, 5), samples_per_s=round(batch / (us_step * 1e-6), 1),
                step="th_mlp2_xent through the C ABI, three eager launches back to back (the Trainer's replayed step: batch_sweep)",
                alg_flops_per_step=flops, frac_of_mfma_peak_step=round(flops / (us_step * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, 4), kernels=kernels)


def extra_workloads(T, dataset, with_cpu, only=None):
    """the BASELINE configs besides the headline, each on the same line (N = 1)"""
    from taper_amd import hip
    ctx = hip.Ctx(handle=T.Device.ctx_handle())
    out = []
    for name in ("cnn_simple_b256", "cnn_reference_b256", "mlp_784-128-64-10_b256"):
        if only and name not in only:
            continue
        try:
            rec, key, batch, sample_shape, lr = trainer_workload(T, name, dataset)
            if key in CNN_LAYERS:
                # what the captured step launches (one chain kernel), then the layer-by-layer launches it replaced (still the path of
                # forward() outside a Trainer step, of full_backward mode and of TAPER_CONV_CHAIN=0)
                chain = os.environ.get("TAPER_CONV_CHAIN", "1") != "0"
                layers = conv_layer_kernels(ctx, CNN_LAYERS[key])
                for k in layers:
                    k["in_step"] = not chain
                head = chain and os.environ.get("TAPER_CHAIN_HEAD", "1") != "0"     # the classifier's rows ride in the chain launch (both CNNs)
                rec["kernels"] = ((chain_head_kernels(ctx) if key == "cnn_simple" else chain_mlp3_kernels(ctx)) if head else []) + \
                                 ([dict(conv_chain_kernel(ctx, key), in_step=not head)] if chain else []) + layers
                rec["conv_us_per_step"] = round(sum(k["us_per_launch"] for k in rec["kernels"] if k["in_step"]), 1)
                try:
                    rec["batch_sweep"] = cnn_batch_sweep(T, key, sample_shape, lr, dataset)
                except Exception as e:
                    rec["batch_sweep"] = dict(error=str(e))
                if key == "cnn_simple" and os.environ.get("TAPER_NO_GRAPH") != "1":      # (not under rocprofv3: its trace is of the batch-256 step)
                    # the data-parallel step of this model on the device (th_wide_head_grads_dp: the exchange inside the batch-sums launch), as
                    # dp_on_device measures the MLP's: 128 images per GPU
                    try:
                        rec["data_parallel"] = dp_on_device(T, key, lr, dataset, steps=1500, sample_shape=sample_shape)
                    except Exception as e:
                        rec["data_parallel"] = dict(error=str(e))
                # the same model with every conv weight training (full_backward: an extension -- the reference cuts the tape at im2col /
                # transpose_4d, quirk Q2 -- through the layer-by-layer forward and the conv backward kernels)
                try:
                    T.set_full_backward(True)
                    fb = trainer_workload(T, name, dataset, steps=300)[0]
                    rec["full_backward"] = dict(ms_per_step=fb["ms_per_step"], samples_per_s=fb["samples_per_s"], steps=300,
                                                note="extension: conv weights train too (not the reference's behaviour, quirk Q2)")
                except Exception as e:
                    rec["full_backward"] = dict(error=str(e))
                finally:
                    T.set_full_backward(False)
            if with_cpu and key != "mlp_example":
                try:
                    rec["cpu_baseline"] = cpu_baseline(key, batch, sample_shape, lr, budget_s=7.0)
                except Exception as e:
                    rec["cpu_baseline"] = dict(value=None, kind="port", sample=f"failed: {e}")
        except Exception as e:   # one workload failing must not lose the line
            rec = dict(workload=name, error=str(e))
        out.append(rec)
    if not only or "mlp_784-128-10_b16384" in only:
        try:
            out.append(mlp2_workload(ctx))
        except Exception as e:
            out.append(dict(workload="mlp_784-128-10_b16384", error=str(e)))
    if not only or "linear_stack_4096x4_b4096" in only:
        try:
            rec = linear_stack_workload(T)
            rec["kernels"] = sgemm_kernels(ctx)
        except Exception as e:
            rec = dict(workload="linear_stack_4096x4_b4096", error=str(e))
        out.append(rec)
    return out



# ---------------------------------------------------------------------------- what is printed
def write_details(full):
    """the full record (every workload's kernels[], the per-thread CPU trials, the data-parallel side runs ...) goes to a side file; the
    printed line stays short enough for a 2 000-character tail to hold all of it"""
    name = "bench_details.json" if full.get("n_gpus", 1) == 1 else f"bench_details_n{full['n_gpus']}.json"
    for d in (ROOT / "gpurun_out", Path.cwd()):
        try:
            if d.is_dir():
                (d / name).write_text(json.dumps(full, indent=1))
                return str((d / name).relative_to(ROOT)) if str(d).startswith(str(ROOT)) else str(d / name)
        except OSError:
            continue
    return None


def _short(x, n):
    x = str(x)
    return x if len(x) <= n else x[: n - 3] + "..."


def compact_line(full, details_path):
    """ONE line: the contract's keys, `roofline`, `cpu_baseline`, and one short entry per other BASELINE config / sweep point"""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: full[k] for k in keep}
    cfg = full["config"]
    out["config"] = {k: (_short(cfg[k], 60) if isinstance(cfg[k], str) else cfg[k])
                     for k in ("workload", "per_gpu_batch", "global_batch", "parallelism", "comm", "graph_record_steps", "ranks_share_one_gpu") if k in cfg}
    out["epochs_per_s"] = full["epochs_per_s"]
    if full.get("sustained"):
        out["sustained_ms_per_step"] = full["sustained"]["ms_per_step"]
    sr = full.get("step_roofline")
    if sr:
        out["step_roofline"] = {"hbm_frac": round(sr["hbm_frac"], 4), "mfma_frac": round(sr["mfma_frac"], 4)}
    if full.get("data_parallel"):
        dp = full["data_parallel"]
        # (N = 1: three_launch_one_rank_step_us and the ceilings of the other forms are in the details file)
        d = {k: dp[k] for k in ("weak_scaling_efficiency", "replicas_bit_identical", "per_gpu_batch", "single_gpu_step_us", "one_rank_step_us",
                                "loopback_two_rank_step_us", "on_device_efficiency_ceiling", "error") if k in dp}
        if "single_gpu_same_per_gpu_batch" in dp:
            d["single_gpu_ms_per_step"] = dp["single_gpu_same_per_gpu_batch"]["ms_per_step"]
        if "same_job_over_rccl" in dp:
            r = dp["same_job_over_rccl"]
            d["same_job_over_rccl"] = {k: r[k] for k in ("value", "ms_per_step", "rccl_ranks", "skipped", "error") if k in r}
        if "dp_at_64_rows_per_gpu" in dp:
            d["b64_per_gpu_ms_per_step"] = dp["dp_at_64_rows_per_gpu"]["ms_per_step"]
        out["data_parallel"] = d
    if full.get("batch_sweep"):
        # {batch: [us/step, epochs/s, fraction of the fp32 MFMA peak]}
        out["batch_sweep_us_epochs_frac"] = {str(e["batch"]): [round(e["ms_per_step"] * 1e3, 1), round(e["epochs_per_s"], 1), round(e["mfma_frac"], 3)] for e in full["batch_sweep"]}
    if full.get("batch_sweep_mlp_784-128-64-10"):
        # the reference's own model (examples/train_mnist.rs): {batch: [us/step, epochs/s]}
        out["sweep_784-128-64-10"] = {str(e["batch"]): [round(e["ms_per_step"] * 1e3, 1), round(e["epochs_per_s"])] for e in full["batch_sweep_mlp_784-128-64-10"]}
    if full.get("workloads"):
        w = {}
        for rec in full["workloads"]:
            if "error" in rec:
                w[rec["workload"]] = {"error": _short(rec["error"], 60)}
                continue
            if rec["workload"] == "mlp_784-128-64-10_b256" and full.get("batch_sweep_mlp_784-128-64-10"):
                continue      # (the same point is the first entry of sweep_784-128-64-10)
            e = {"ms_per_step": rec["ms_per_step"]}      # (samples/s = per_gpu_batch / ms_per_step: in the details file)
            ks = [k for k in rec.get("kernels", []) if k.get("in_step", True)]
            if rec["workload"].startswith("mlp_784-128-10_b") and ks:      # th_mlp2_xent launch by launch: [us, fraction of its roofline]
                e = {"ms_per_step": rec["ms_per_step"], "kernels": {k["kernel"].split("<")[0].replace("mlp2_", "").replace("_kernel8", "").replace("_kernel", ""): [k["us_per_launch"], k["frac"]] for k in ks}}
            elif ks:
                k = max(ks, key=lambda r: r["us_per_launch"])
                e["kernel"] = [_short(k["kernel"].split("<")[0].split("(")[0].replace("_kernel", ""), 20), k["us_per_launch"], k["bound"], k["frac"]]
            if isinstance(rec.get("batch_sweep"), list):      # CNN steps at larger batches: {batch: [ms/step, forward-conv fraction of the MFMA peak]}
                e["sweep"] = {str(x["batch"]): [round(x["ms_per_step"], 4), round(x["mfma_frac_fwd"], 3)] for x in rec["batch_sweep"]}
            if "frac_of_mfma_peak" in rec:
                e["mfma_frac"] = rec["frac_of_mfma_peak"]
                e["sgemm_4096_frac"] = [k["frac"] for k in rec.get("kernels", [])]
            if rec.get("full_backward", {}).get("ms_per_step"):
                e["full_bwd_ms"] = round(rec["full_backward"]["ms_per_step"], 4)      # (extension: conv weights train too)
            cb = rec.get("cpu_baseline")
            if cb and cb.get("value"):
                e["cpu_sps"] = round(cb["value"], 1)      # CPU baseline, samples/s
            w[rec["workload"]] = e
        out["workloads"] = w
    rf = full.get("roofline")
    if rf:
        out["roofline"] = ({k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "us_per_launch", "peak_basis", "alg_bytes_per_launch") if k in rf}
                           if "skipped" not in rf else rf)
        if "kernel" in out["roofline"]:
            out["roofline"]["kernel"] = _short(out["roofline"]["kernel"], 48)
    else:
        out["roofline"] = None
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: (round(cb[k], 1) if k == "value" and cb[k] else cb[k]) for k in ("value", "unit", "cores", "kind") if k in cb}
        out["cpu_baseline"]["sample"] = _short(cb.get("sample", ""), 44)
        if cb.get("blas_feature", {}).get("value"):
            out["cpu_baseline"]["blas_feature_value"] = round(cb["blas_feature"]["value"], 1)
        if "see" in cb:
            out["cpu_baseline"]["see"] = cb["see"]
    else:
        out["cpu_baseline"] = None
    out["details"] = details_path
    # the driver keeps the last 2 000 characters of the output: shed the least essential extras until the line fits (everything is in `details`)
    for path in (("workloads", "*", "cpu_sps"), ("workloads", "*", "full_bwd_ms"), ("sustained_ms_per_step",), ("sweep_784-128-64-10",),
                 ("cpu_baseline", "blas_feature_value"), ("cpu_baseline", "sample"), ("step_roofline",), ("data_parallel", "one_rank_step_us"),
                 ("workloads", "*", "sweep")):
        if len(json.dumps(out)) <= 1960:
            break
        if len(path) == 1:
            out.pop(path[0], None)
        elif path[1] != "*":
            (out.get(path[0]) or {}).pop(path[1], None)
        else:
            for e in (out.get(path[0]) or {}).values():
                if isinstance(e, dict):
                    e.pop(path[2], None)
    return out


# ---------------------------------------------------------------------------- launcher
def self_spawn(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* exactly as torch.distributed.run would set them), one per GPU; rank 0 prints the JSON line."""
    import socket
    import subprocess
    from taper_amd import hip
    have = hip.device_count()
    if have < n and os.environ.get("TAPER_BENCH_SHARE_DEVICE") != "1":   # (test hook: every rank on GPU 0, p2p backend only)
        sys.exit(f"bench.py: --gpus {n} but only {have} GPU(s) visible; refusing to run a smaller job under that name")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", TAPER_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def timed_run(T, dist, trainer, loader, steps):
    """barrier + stream sync, exactly `steps` steps, barrier + stream sync; MAX over ranks, samples summed over ranks"""
    barrier_sync(dist, T)
    t0 = time.perf_counter()
    samples = run_steps(T, trainer, loader, steps)
    barrier_sync(dist, T)
    dt = time.perf_counter() - t0
    if dist is not None:
        dt = dist.all_reduce_max(dt)
        samples = int(dist.all_reduce_sum(samples))
    return samples, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: mlp_784-128-10_b64 (BASELINE configs[1]) at N = 1, mlp_784-128-10_b128 (configs[3]) at N > 1")
    ap.add_argument("--dataset-size", type=int, default=60000)
    ap.add_argument("--batch", type=int, default=0, help="override the workload's per-GPU batch (SURVEY 8d batch sweep)")
    ap.add_argument("--graph-chunk", type=int, default=0, help="steps per hipGraph replay (0: the Trainer's default)")
    ap.add_argument("--settle-seconds", type=float, default=0.0, help="untimed extra stepping before the timed region (sustained-clock state); 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-backward", action="store_true",
                    help="CNN workloads: train the conv weights too (extension; the reference cuts the tape there, quirk Q2)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel timing after the timed region")
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch sweep (SURVEY 8d) reported beside the headline value")
    ap.add_argument("--dp-backend", default="auto", choices=["auto", "p2p", "rccl"],
                    help="N > 1 gradient exchange: p2p = one-shot peer-to-peer all-reduce fused with Adam, rccl = ncclAllReduce; auto = p2p "
                         "when its self-check passes, else rccl")
    ap.add_argument("--workloads", default="all", help="N = 1: the other BASELINE configs reported beside the headline: all | none | comma list")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    under_profiler = "rocprofiler-sdk" in os.environ.get("LD_PRELOAD", "") or bool(os.environ.get("ROCP_TOOL_LIBRARIES"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args.gpus)          # never returns
    import taper_amd as T
    dist, rank, world = init_dist(args.gpus)
    share = os.environ.get("TAPER_BENCH_SHARE_DEVICE") == "1"
    T.Device.set_device(0 if share else int(os.environ.get("LOCAL_RANK", "0")))

    if args.workload is None:
        args.workload = "mlp_784-128-10_b64" if world == 1 else "mlp_784-128-10_b128"
    key, batch, sample_shape, lr = WORKLOADS[args.workload]
    if args.full_backward:
        T.set_full_backward(True)
    if args.batch:
        batch = args.batch
        args.workload = args.workload.rsplit("_b", 1)[0] + f"_b{batch}"
    model = build_model(T, key)
    opt = T.Adam(model.parameters(), lr, None, None, 1e-4)          # examples/train_mnist.rs:50-51
    comm, comm_kind = make_comm(dist, T, args.dp_backend, opt, share and world > 1)
    trainer = T.Trainer(model, opt, sample_shape=sample_shape, comm=comm, **({"graph_chunk": args.graph_chunk} if args.graph_chunk else {}))
    # every rank owns its shard of the synthetic epoch (rows are independent: SURVEY.md 8e)
    ds = T.MNISTDataset.synthetic(args.dataset_size, seed=0x7461706572 + rank)
    loader = T.DataLoader(ds, batch, False)

    run_steps(T, trainer, loader, max(args.warmup, 2))              # untimed; also captures the graph
    # Every graph size of the replay ladder (128, 64, ..., 2, 1 steps) is recorded by the first call long enough to use it
    # (128 + 1 steps): a warm-up shorter than that is topped up, untimed, so that no recording falls into the timed region.
    # Reported as config.graph_record_steps; the W steps above and the K timed steps below are exactly what was asked for.
    record_steps = max(0, 257 - max(args.warmup, 2))
    if record_steps:
        run_steps(T, trainer, loader, record_steps)
    # ... and a run length that is no single ladder size (the contract's K = 20 = 16 + 4) gets ONE graph of exactly K steps, recorded by
    # the first call that asks for K steps: that call is made here, untimed, and counted in graph_record_steps as well
    k_tail = args.steps % (args.graph_chunk or 128)
    if k_tail > 2 and k_tail & (k_tail - 1):
        run_steps(T, trainer, loader, args.steps)
        record_steps += args.steps
    settle_steps = 0
    if args.settle_seconds > 0:      # optional: measure the sustained-clock state (see `sustained` below for the default run)
        T.Device.sync()
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < args.settle_seconds:
            settle_steps += run_steps(T, trainer, loader, 256) // batch
            T.Device.sync()
    samples, dt = timed_run(T, dist, trainer, loader, args.steps)

    # N > 1: the single-GPU figure of the SAME per-GPU batch (no communicator), measured by rank 0 in the same run with the
    # same W / K -- the denominator of SURVEY 8e's weak-scaling efficiency -- and the data-parallel figure at the N = 1
    # headline's 64 rows per GPU (comparable with a separate N = 1 run of this script)
    dp_extra = None
    if world > 1 and not args.batch:
        dp_extra = {}

        dp_extra["replicas_bit_identical"] = replicas_identical(dist, model)

        def side_run(b, backend):
            m2 = build_model(T, key)
            o2 = T.Adam(m2.parameters(), lr, None, None, 1e-4)
            c2, kind2 = make_comm(dist, T, backend, o2, share and world > 1) if backend else (None, "none")   # (a p2p communicator is bound to one optimizer's arena)
            t2 = T.Trainer(m2, o2, sample_shape=sample_shape, comm=c2)
            l2 = T.DataLoader(ds, b, False)
            run_steps(T, t2, l2, max(args.warmup, 2) + record_steps)
            if k_tail > 2 and k_tail & (k_tail - 1):
                run_steps(T, t2, l2, args.steps)
            s_, d_ = timed_run(T, dist if backend else None, t2, l2, args.steps)
            same = replicas_identical(dist, m2) if backend else None
            _KEEP_ALIVE.append((t2, o2, m2, l2, c2))
            return s_, d_, kind2, same
        if batch != 64 and key == "mlp_baseline":
            s64, d64, k64, same64 = side_run(64, args.dp_backend)
            dp_extra["dp_at_64_rows_per_gpu"] = dict(workload="mlp_784-128-10_b64", per_gpu_batch=64, global_batch=64 * world, value=round(s64 / d64, 1),
                                                     unit="samples/s", ms_per_step=round(d64 / args.steps * 1e3, 5), comm=k64, replicas_bit_identical=same64)
        # the same job over RCCL -- north_star's transport -- whenever every rank has a GPU of its own (RCCL cannot put two ranks on one
        # device); rccl_ranks is what ncclCommCount reports for the communicator that ran it
        from taper_amd import hip as _hip
        if comm is not None and comm.is_p2p():
            if not share and _hip.device_count() >= world:
                try:      # reported, never required: a failure here must not cost the run its line
                    sr, dr, kr, samer = side_run(batch, "rccl")
                    c_rccl = _KEEP_ALIVE[-1][4]
                    dp_extra["same_job_over_rccl"] = dict(value=round(sr / dr, 1), unit="samples/s", ms_per_step=round(dr / args.steps * 1e3, 5), comm=kr,
                                                          rccl_ranks=c_rccl.count() if c_rccl is not None else None, replicas_bit_identical=samer)
                except Exception as e:   # noqa: BLE001
                    dp_extra["same_job_over_rccl"] = dict(error=str(e)[:200])
            else:
                dp_extra["same_job_over_rccl"] = dict(skipped=f"{world} ranks share {_hip.device_count()} GPU(s): RCCL needs one device per rank")
        elif comm is not None:
            dp_extra["rccl_ranks"] = comm.count()
        if rank == 0:
            s1, d1, _, _ = side_run(batch, None)
            dp_extra["single_gpu_same_per_gpu_batch"] = dict(workload=args.workload, per_gpu_batch=batch, n_gpus=1, value=round(s1 / d1, 1), unit="samples/s",
                                                             ms_per_step=round(d1 / args.steps * 1e3, 5), steps=args.steps, warmup=args.warmup)
            # eff(W) = T_step(1 GPU, B/W rows) / T_step(W GPUs, B/W rows each), same W / K, same run
            # (ranks that SHARE one device time-share it: their ratio is an artefact of the harness, not an efficiency -- null)
            dp_extra["weak_scaling_efficiency"] = None if share else round((d1 / args.steps) / (dt / args.steps), 4)
        barrier_sync(dist, T)
        # the exchange launch on its own: `reps` gradient exchanges + Adam as the step issues them, back to back between two events on
        # every rank's stream (collective).  Moves the optimizer state: the timed run and the replica check are behind us.
        inkernel = comm.is_p2p() and comm.inkernel_launches() > 0
        dp_extra["exchange_form"] = ("inside the gradient launch (th_mlp_tail_dp): two launches per step" if inkernel else
                                     "its own launch behind the gradient launch: three launches per step")
        try:
            ex_us = comm.time_exchange(opt, 200)
            ex_us = dist.all_reduce_max(ex_us)
        except Exception as e:   # reported, never required
            ex_us, dp_extra["exchange_error"] = None, str(e)
        if inkernel:
            # no exchange launch exists: the roofline line is the whole step against the bytes it moves -- the single-GPU step's algorithmic
            # bytes + every peer's gradient slices pushed out and the same amount read back from the receive region
            P = opt_total(T, opt)
            _, step_bytes = algorithmic_step(key, batch)
            ex_bytes = float(step_bytes) + 2.0 * (world - 1) * 4.0 * P
            us = dt / args.steps * 1e6
            links = 7 * 153.0
            peak, basis = (HBM_PEAK_GBS, "HBM (the ranks share one GPU: pushes are local)") if share else \
                          (HBM_PEAK_GBS, "HBM 8000 GB/s (the pushed slices cross xGMI: 7 links x 153 GB/s per GPU)")
            dp_extra["exchange"] = dict(kernel="sgemm_small16_tick + mlp_tail_exact_kernel<DP> (th_mlp_tail_dp)", us_per_launch=round(us, 2),
                                        alg_bytes_per_launch=ex_bytes, bound="hbm", achieved=round(ex_bytes / (us * 1e-6) / 1e9, 2), peak=round(peak, 1),
                                        unit="GB/s", frac=round(ex_bytes / (us * 1e-6) / 1e9 / peak, 5), peak_basis=basis, traffic=None,
                                        three_launch_exchange_us=None if not ex_us else round(ex_us, 2),
                                        note="latency-bound: the whole two-launch step; the exchange is %d x 0.4 MB pushed per rank" % (world - 1))
        elif ex_us:
            P = opt_total(T, opt)
            ex_bytes = (world - 1) * 4.0 * P + 4.0 * P + 24.0 * P        # peers' gradients over the links + own + Adam's p / m / v read and written
            links = 7 * 153.0
            peak, basis = (HBM_PEAK_GBS, "HBM (the ranks share one GPU: peer reads are local)") if share else \
                          (min(HBM_PEAK_GBS, links), "xGMI, 7 links x 153 GB/s per GPU (< HBM 8000 GB/s)")
            dp_extra["exchange"] = dict(kernel="p2p_allreduce_adam_kernel" if comm.is_p2p() else "ncclAllReduce + adam_kernel", us_per_launch=round(ex_us, 2),
                                        alg_bytes_per_launch=ex_bytes, bound="hbm", achieved=round(ex_bytes / (ex_us * 1e-6) / 1e9, 2), peak=round(peak, 1),
                                        unit="GB/s", frac=round(ex_bytes / (ex_us * 1e-6) / 1e9 / peak, 5), peak_basis=basis, traffic=None,
                                        note="latency-bound: 0.4 MB per rank; max over ranks of 200 back-to-back exchanges")

    sustained = None
    if world == 1 and args.settle_seconds == 0 and not args.no_roofline and not under_profiler:   # (the trace is of the W + K run only)
        # the same K steps again after 0.25 s of continuous stepping: under sustained load the part settles at lower clocks
        # than it holds through the first ~50 ms of a run; both states are reported, `value` is the contract's W + K run
        T.Device.sync()
        t_s = time.perf_counter()
        while time.perf_counter() - t_s < 0.25:
            run_steps(T, trainer, loader, 256)
            T.Device.sync()
        t_s = time.perf_counter()
        s_samples = run_steps(T, trainer, loader, args.steps)
        T.Device.sync()
        s_dt = time.perf_counter() - t_s
        sustained = {"value": round(s_samples / s_dt, 1), "unit": "samples/s", "ms_per_step": round(s_dt / args.steps * 1e3, 5),
                     "after": "0.25 s of untimed stepping"}
    if rank == 0:
        flops, nbytes = algorithmic_step(key, batch)
        roof = None
        if key == "mlp_baseline" and under_profiler:
            # rocprofv3 (ROCm 7.2) segfaults in hipGraphLaunch once a process replays more than one
            # instantiated graph back to back; the Trainer's replays above are what the trace is for.
            roof = dict(skipped="per-kernel timers are not run under rocprofv3; see profiles/ for the trace of this command")
        elif key == "mlp_baseline" and not args.no_roofline and batch <= 512 and world == 1:
            # per-launch durations of the step's two kernels, measured live (HIP events on the ctx
            # stream, graph chains with / without each launch).  `roofline` is the kernel that carries
            # the step's HBM traffic (83% of its algorithmic bytes); the full list is in `kernels`.
            from taper_amd import hip
            sk = StepKernels(hip.Ctx(handle=T.Device.ctx_handle()), batch).measure()
            k = max(sk["kernels"], key=lambda r: r["alg_bytes_per_launch"])
            by_time = max(sk["kernels"], key=lambda r: r["us_per_launch"])
            traffic, traffic_src = pmc_traffic(args.workload, k["kernel"])
            roof = dict(bound="hbm", achieved=k["achieved_GBps"], peak=HBM_PEAK_GBS, unit="GB/s", frac=k["hbm_frac"], traffic=traffic,
                        traffic_source=traffic_src,
                        kernel=k["kernel"], role=k["role"], us_per_launch=k["us_per_launch"],
                        alg_bytes_per_launch=k["alg_bytes_per_launch"], mfma_tflops=k["mfma_tflops"],
                        dominant_by="algorithmic bytes; by time the leader is %s (%.1f us)" % (by_time["kernel"], by_time["us_per_launch"]),
                        step_us_two_launch_chain=sk["step_us"], kernels=sk["kernels"])
        if world > 1 and dp_extra and dp_extra.get("exchange"):
            roof = dict(dp_extra["exchange"])           # N > 1: the kernel the job adds to the N = 1 step is the gradient exchange
        sweep = sweep_example = None
        # (not under rocprofv3: the trace of this command is for the headline workload's kernels only)
        if key == "mlp_baseline" and world == 1 and not args.batch and not args.no_sweep and not under_profiler:
            sweep = batch_sweep(T, build_model, key, lr, ds)
            # the reference's OWN example model (examples/train_mnist.rs:40-48: two hidden layers) over the same batches: from 480 rows on its
            # step is th_mlp2_xent_deep (three launches)
            sweep_example = batch_sweep(T, build_model, "mlp_example", lr, ds, batches=(256, 1024, 4096, 16384, 60000))
        cpu = None
        with_cpu = not args.no_cpu_baseline and world == 1          # the CPU leg is timed on rank 0 at N = 1 only
        if with_cpu:
            try:
                cpu = cpu_baseline(key, batch, sample_shape, lr)
            except Exception as e:  # the baseline is reported, never required
                cpu = dict(value=None, unit="samples/s", cores=1, kind="port", sample=f"failed: {e}")
        if world > 1:   # the CPU leg is timed at N = 1 only (rank 0's host cores are shared by the N ranks here)
            cpu = dict(value=None, unit="samples/s", cores=None, kind="port", sample="not timed at N > 1",
                       see="cpu_baseline of `python bench.py --gpus 1` (same box, same workload family)")
        dp_ceiling = None
        if world == 1 and key == "mlp_baseline" and not args.batch and not under_profiler and args.workloads != "none":
            try:
                dp_ceiling = dp_on_device(T, key, lr, ds)
            except Exception as e:   # reported, never required
                dp_ceiling = dict(error=str(e))
        workloads = None
        if under_profiler:
            # rocprofv3 (ROCm 7.2) crashes inside hipGraphLaunch once a process replays the graphs of a SECOND Trainer: the headline above ran as
            # graph replays (its trace is the one compared with `roofline`); the other workloads enqueue every step's op list eagerly -- the
            # same kernels with the same arguments, one host launch each -- so the trace holds their per-kernel durations, while their step
            # times in THIS run are host-launch-bound and are not the measurement
            os.environ["TAPER_NO_GRAPH"] = "1"
        if world == 1 and not args.batch and args.workloads != "none" and args.workload == "mlp_784-128-10_b64":
            only = None if args.workloads == "all" else set(args.workloads.split(","))
            workloads = extra_workloads(T, ds, with_cpu, only)
        out = {
            "metric": "MNIST samples/sec fwd+bwd+step", "value": round(samples / dt, 1), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "per_gpu_batch": batch, "global_batch": batch * world,
                       "optimizer": f"Adam(lr={lr}, wd=1e-4)", "parallelism": f"dp{world}" if world > 1 else "single",
                       "comm": comm_kind, **({"ranks_share_one_gpu": True} if share and world > 1 else {}),
                       "step": "hipGraph replay of gather+fwd+xent+bwd+adam+log", "graph_record_steps": record_steps, "clock_settle_steps": settle_steps,
                       **({"conv_gradients": "full_backward (extension)"} if args.full_backward else {})},
            "epochs_per_s": round(samples / dt / 60000.0, 3),
            "step_roofline": None if flops is None else {
                "alg_flops_per_step": flops, "alg_bytes_per_step": nbytes,
                "hbm_frac": round(nbytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 6),
                "mfma_frac": round(flops / (dt / args.steps) / 1e12 / MFMA_F32_PEAK_TF, 6)},
            "sustained": sustained, "batch_sweep": sweep, "batch_sweep_mlp_784-128-64-10": sweep_example, "roofline": roof, "cpu_baseline": cpu,
            **({"data_parallel": dp_extra} if dp_extra is not None else {}),
            **({"data_parallel": dp_ceiling} if dp_ceiling is not None else {}),
            **({"workloads": workloads} if workloads is not None else {}),
        }
        print(json.dumps(compact_line(out, write_details(out))), flush=True)
    if dist is not None:
        dist.close()


if __name__ == "__main__":
    main()

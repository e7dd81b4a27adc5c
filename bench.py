#!/usr/bin/env python3
"""bench.py -- throughput of taper's training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of synthetic MNIST-shaped
input already resident in HBM: gather batch -> forward -> softmax cross-entropy
-> backward -> (N > 1: RCCL all-reduce of the flat grad arena) -> Adam -> log,
replayed as one hipGraph (host: C++ tape in libtaper_host.so; device: the HIP
kernels of libtaper_hip.so through the C ABI).  Default workload =
BASELINE.json configs[1]: MLP 784-128-10, batch 64 per GPU, Adam(1e-3, wd 1e-4).

Prints ONE JSON line (rank 0) with the BASELINE metric plus `roofline` (the
dominant kernel timed live with HIP events) and `cpu_baseline` (the C
restatement of the reference's CPU path, timed on this box's host cores).
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


def make_comm(rdzv, T):
    from taper_amd.dist import init_data_parallel
    comm = init_data_parallel(T, rdzv)
    return comm, ("rccl" if comm is not None else "none")


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


def batch_sweep(T, build_model, key, lr, dataset_size, batches=(256, 1024, 4096, 16384, 60000)):
    """SURVEY.md 8(d) batch sweep of the same model / optimizer / step on one GPU (fresh model per point; the
    headline `value` stays BASELINE configs[1], batch 64): where the path stops being launch-bound."""
    out = []
    for b in batches:
        if b > dataset_size:
            continue
        model = build_model(T, key)
        opt = T.Adam(model.parameters(), lr, None, None, 1e-4)
        trainer = T.Trainer(model, opt)
        loader = T.DataLoader(T.MNISTDataset.synthetic(dataset_size, seed=0x7461706572), b, False)
        steps = min(80_000, max(200, 20_000_000 // b))   # >= 150 ms of work per point: short bursts run below the steady clocks
        run_steps(T, trainer, loader, max(steps // 8, 3))
        T.Device.sync()
        t0 = time.perf_counter()
        samples = run_steps(T, trainer, loader, steps)
        T.Device.sync()
        dt = time.perf_counter() - t0
        out.append(dict(batch=b, steps=steps, ms_per_step=round(dt / steps * 1e3, 5), samples_per_s=round(samples / dt, 1),
                        epochs_per_s=round(samples / dt / 60000.0, 2)))
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
    for name, rec in json.loads(files[-1].read_text())["kernels"].items():
        if kernel.split("<")[0] + "<" in name:
            return rec["traffic_bytes_per_launch"], f"profiles/{files[-1].name}"
    return None, None


def cpu_baseline(key, batch, sample_shape, lr, budget_s=12.0):
    """The reference's CPU path (C restatement: oracle/, kind 'port'), 1 thread, timed on
    this host on a bounded sample of the same workload (~budget_s seconds of CPU work)."""
    from oracle import oracle as O
    try:
        so = O.build_native(tempfile.mkdtemp(prefix="taper_oracle_native_"))   # -march=native for THIS box
        O.use_library(so)
        flavour = "-O3 -march=native"
    except Exception:
        flavour = "-O3 -mavx -mfma (prebuilt)"
    from tests import backends
    rng = np.random.default_rng(1)
    spec = getattr(backends, key)(rng)
    ob = backends.get("oracle")
    ob.set_zero_sentinel(True)
    model = ob.sequential(spec)
    opt = O.Adam(model.parameters(), lr, None, None, 1e-4)
    x, y = backends.mnist_like(rng, batch)
    shape = (batch, 784) if sample_shape is None else (batch,) + tuple(sample_shape)
    model.train_step(opt, x, y, shape)   # warm-up
    t0, n = time.perf_counter(), 0
    while True:
        model.train_step(opt, x, y, shape)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return dict(value=n * batch / dt, unit="samples/s", cores=1, kind="port",
                sample=f"{n} steps of {key} batch {batch} in {dt:.1f}s; C restatement of the reference CPU tape "
                       f"(gcc {flavour}, 1 thread = matrixmultiply without its threading feature); host has {os.cpu_count()} cpus")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--workload", default="mlp_784-128-10_b64", choices=sorted(WORKLOADS))
    ap.add_argument("--dataset-size", type=int, default=60000)
    ap.add_argument("--batch", type=int, default=0, help="override the workload's per-GPU batch (SURVEY 8d batch sweep)")
    ap.add_argument("--graph-chunk", type=int, default=0, help="steps per hipGraph replay (0: the Trainer's default)")
    ap.add_argument("--settle-seconds", type=float, default=0.0, help="untimed extra stepping before the timed region (sustained-clock state); 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-backward", action="store_true",
                    help="CNN workloads: train the conv weights too (extension; the reference cuts the tape there, quirk Q2)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel timing after the timed region")
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch sweep (SURVEY 8d) reported beside the headline value")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import taper_amd as T
    dist, rank, world = init_dist(args.gpus)
    T.Device.set_device(int(os.environ.get("LOCAL_RANK", "0")))

    key, batch, sample_shape, lr = WORKLOADS[args.workload]
    if args.full_backward:
        T.set_full_backward(True)
    if args.batch:
        batch = args.batch
        args.workload = args.workload.rsplit("_b", 1)[0] + f"_b{batch}"
    model = build_model(T, key)
    opt = T.Adam(model.parameters(), lr, None, None, 1e-4)          # examples/train_mnist.rs:50-51
    comm, comm_kind = make_comm(dist, T)
    trainer = T.Trainer(model, opt, sample_shape=sample_shape, comm=comm, **({"graph_chunk": args.graph_chunk} if args.graph_chunk else {}))
    # every rank owns its shard of the synthetic epoch (rows are independent: SURVEY.md 8e)
    ds = T.MNISTDataset.synthetic(args.dataset_size, seed=0x7461706572 + rank)
    loader = T.DataLoader(ds, batch, False)

    run_steps(T, trainer, loader, max(args.warmup, 2))              # untimed; also captures the graph
    # Every graph size of the replay ladder (128, 32, 8, 2, 1 steps) is recorded by the first call long enough to use it
    # (2 x 128 steps): a warm-up shorter than that is topped up, untimed, so that no recording falls into the timed region.
    # Reported as config.graph_record_steps; the W steps above and the K timed steps below are exactly what was asked for.
    record_steps = max(0, 257 - max(args.warmup, 2))
    if record_steps:
        run_steps(T, trainer, loader, record_steps)
    settle_steps = 0
    if args.settle_seconds > 0:      # optional: measure the sustained-clock state (see `sustained` below for the default run)
        T.Device.sync()
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < args.settle_seconds:
            settle_steps += run_steps(T, trainer, loader, 256) // batch
            T.Device.sync()
    barrier_sync(dist, T)
    t0 = time.perf_counter()
    samples = run_steps(T, trainer, loader, args.steps)
    barrier_sync(dist, T)
    dt = time.perf_counter() - t0

    if dist is not None:
        dt = dist.all_reduce_max(dt)                 # MAX over ranks
        samples = int(dist.all_reduce_sum(samples))  # whole-job aggregate

    sustained = None
    under_profiler = "rocprofiler-sdk" in os.environ.get("LD_PRELOAD", "") or bool(os.environ.get("ROCP_TOOL_LIBRARIES"))
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
        elif key == "mlp_baseline" and not args.no_roofline:
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
        sweep = None
        # (not under rocprofv3: the trace of this command is for the headline workload's kernels only)
        if key == "mlp_baseline" and world == 1 and not args.batch and not args.no_sweep and not under_profiler:
            sweep = batch_sweep(T, build_model, key, lr, args.dataset_size)
        cpu = None
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N = 1 only
            try:
                cpu = cpu_baseline(key, batch, sample_shape, lr)
            except Exception as e:  # the baseline is reported, never required
                cpu = dict(value=None, unit="samples/s", cores=1, kind="port", sample=f"failed: {e}")
        out = {
            "metric": "MNIST samples/sec fwd+bwd+step", "value": round(samples / dt, 1), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "per_gpu_batch": batch, "global_batch": batch * world,
                       "optimizer": f"Adam(lr={lr}, wd=1e-4)", "parallelism": f"dp{world}" if world > 1 else "single",
                       "comm": comm_kind, "step": "hipGraph replay of gather+fwd+xent+bwd+adam+log", "graph_record_steps": record_steps, "clock_settle_steps": settle_steps,
                       **({"conv_gradients": "full_backward (extension)"} if args.full_backward else {})},
            "epochs_per_s": round(samples / dt / 60000.0, 3),
            "step_roofline": None if flops is None else {
                "alg_flops_per_step": flops, "alg_bytes_per_step": nbytes,
                "hbm_frac": round(nbytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 6),
                "mfma_frac": round(flops / (dt / args.steps) / 1e12 / MFMA_F32_PEAK_TF, 6)},
            "sustained": sustained, "batch_sweep": sweep, "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.close()


if __name__ == "__main__":
    main()

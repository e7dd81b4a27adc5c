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


def time_dominant_kernel(T, key, batch, reps=400):
    """Live HIP-event timing (on the ctx stream) of the step's dominant kernel: the
    layer-1 weight-gradient GEMM dW1[128,784] (+)= dZ1^T[128,B] . X[B,784] (th_linear_bwd ->
    sgemm TN), algorithmic bytes = 4*(B*128 + B*784 + 128*784)."""
    from taper_amd import hip
    ctx = hip.Ctx(handle=T.Device.ctx_handle())
    out_f, in_f = 128, 784
    rng = np.random.default_rng(0)
    x = ctx.upload(rng.uniform(0, 1, (batch, in_f)).astype(np.float32))
    dz = ctx.upload(rng.uniform(-1, 1, (batch, out_f)).astype(np.float32))
    dw = ctx.zeros(out_f * in_f)
    for _ in range(20):
        ctx.call("th_linear_bwd", x, None, dz, None, None, dw, None, batch, in_f, out_f, 0)
    e0, e1 = hip.Event(), hip.Event()
    ctx.record(e0)
    for _ in range(reps):
        ctx.call("th_linear_bwd", x, None, dz, None, None, dw, None, batch, in_f, out_f, 0)
    ctx.record(e1)
    us = hip.Ctx.elapsed_ms(e0, e1) * 1e3 / reps
    alg_bytes = 4 * (batch * out_f + batch * in_f + out_f * in_f)
    return dict(kernel="linear_bwd_small (dW1 = dZ1^T.X, 128x784x%d)" % batch, us_per_launch=us, alg_bytes=alg_bytes,
                alg_flops=2 * out_f * in_f * batch)


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import taper_amd as T
    dist, rank, world = init_dist(args.gpus)
    T.Device.set_device(int(os.environ.get("LOCAL_RANK", "0")))

    key, batch, sample_shape, lr = WORKLOADS[args.workload]
    model = build_model(T, key)
    opt = T.Adam(model.parameters(), lr, None, None, 1e-4)          # examples/train_mnist.rs:50-51
    comm, comm_kind = make_comm(dist, T)
    trainer = T.Trainer(model, opt, sample_shape=sample_shape, comm=comm)
    # every rank owns its shard of the synthetic epoch (rows are independent: SURVEY.md 8e)
    ds = T.MNISTDataset.synthetic(args.dataset_size, seed=0x7461706572 + rank)
    loader = T.DataLoader(ds, batch, False)

    run_steps(T, trainer, loader, max(args.warmup, 2))              # untimed; also captures the graph
    barrier_sync(dist, T)
    t0 = time.perf_counter()
    samples = run_steps(T, trainer, loader, args.steps)
    barrier_sync(dist, T)
    dt = time.perf_counter() - t0

    if dist is not None:
        dt = dist.all_reduce_max(dt)                 # MAX over ranks
        samples = int(dist.all_reduce_sum(samples))  # whole-job aggregate

    if rank == 0:
        flops, nbytes = algorithmic_step(key, batch)
        roof = None
        if key.startswith("mlp"):
            k = time_dominant_kernel(T, key, batch)
            gbs = k["alg_bytes"] / (k["us_per_launch"] * 1e-6) / 1e9
            roof = dict(bound="hbm", achieved=round(gbs, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 5),
                        traffic=None, kernel=k["kernel"], us_per_launch=round(k["us_per_launch"], 3),
                        alg_bytes_per_launch=k["alg_bytes"],
                        mfma_tflops=round(k["alg_flops"] / (k["us_per_launch"] * 1e-6) / 1e12, 3))
        cpu = None
        if not args.no_cpu_baseline:
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
                       "comm": comm_kind, "step": "hipGraph replay of gather+fwd+xent+bwd+adam+log"},
            "epochs_per_s": round(samples / dt / 60000.0, 3),
            "step_roofline": None if flops is None else {
                "alg_flops_per_step": flops, "alg_bytes_per_step": nbytes,
                "hbm_frac": round(nbytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 6),
                "mfma_frac": round(flops / (dt / args.steps) / 1e12 / MFMA_F32_PEAK_TF, 6)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.close()


if __name__ == "__main__":
    main()

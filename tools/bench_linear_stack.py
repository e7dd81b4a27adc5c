#!/usr/bin/env python3
"""BASELINE configs[4] / SURVEY.md 8(d) config 5: a stack of L Linear(4096, 4096)+ReLU layers and a
Linear(4096, 10) classifier, batch 4096, fp32 -- Tape::reset -> forward -> cross_entropy_loss ->
backward -> Adam::step through the public op-by-op host API (no step graph: every product is ~1 ms).

Algorithmic flops per step (2mnk per product):
  per 4096 layer: forward X.W^T, dW = dZ^T.X, dX = dZ.W   -> 3 x 137.44 GFLOP
  (the first layer has no dX: the input carries no gradient)   -> (3L - 1) x 137.44 GFLOP
  classifier 4096 -> 10: 3 x 2*4096*4096*10 = 1.0 GFLOP;  Adam 14 flop / parameter.
Prints one JSON line: ms per step, TFLOP/s of the whole step and its fraction of the dense fp32 MFMA
peak (157.3 TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MFMA_F32_PEAK_TF = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-adam", action="store_true", help="forward + backward only")
    args = ap.parse_args()
    import taper_amd as T
    T.Device.set_device(0)
    L, W, B = args.layers, args.width, args.batch
    layers = []
    for i in range(L):
        layers += [T.Linear(W, W, True, 1 + i), T.ReLU()]
    layers.append(T.Linear(W, 10, True, 1 + L))
    model = T.Sequential(layers)
    opt = T.Adam(model.parameters(), 1e-4, None, None, 1e-4)
    rng = np.random.default_rng(0x7461706572 & 0xFFFFFFFF)
    x = T.Tensor(rng.uniform(0, 1, (B, W)).astype(np.float32))
    y = T.Tensor(rng.integers(0, 10, B).astype(np.float32))

    def step():
        T.Tape.reset()
        opt.zero_grad()
        loss = T.cross_entropy_loss(model.forward(x), y)
        loss.backward()
        if not args.no_adam:
            opt.step()
        return loss

    for _ in range(args.warmup):
        loss = step()
    first = float(loss.data()[0])
    T.Device.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    T.Device.sync()
    dt = (time.perf_counter() - t0) / args.steps
    last = float(loss.data()[0])
    params = L * (W * W + W) + W * 10 + 10
    flops = (3 * L - 1) * 2.0 * B * W * W + 3 * 2.0 * B * W * 10 + (0 if args.no_adam else 14.0 * params)
    alg_bytes = 28.0 * params + 4.0 * B * W * (2 * L + 1)    # Adam streams + every activation written once and read once
    print(json.dumps({
        "workload": f"linear_stack_{W}x{L}_b{B}", "dtype": "f32", "steps": args.steps, "ms_per_step": round(dt * 1e3, 4),
        "samples_per_s": round(B / dt, 1), "tflops": round(flops / dt / 1e12, 2), "frac_of_mfma_peak": round(flops / dt / 1e12 / MFMA_F32_PEAK_TF, 4),
        "alg_flops_per_step": flops, "alg_bytes_per_step": alg_bytes, "alg_GBps": round(alg_bytes / dt / 1e9, 1),
        "adam": not args.no_adam, "loss_first": round(first, 5), "loss_last": round(last, 5)}))


if __name__ == "__main__":
    main()

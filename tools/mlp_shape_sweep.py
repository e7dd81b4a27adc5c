#!/usr/bin/env python3
"""Which step form an MLP of another shape takes, and what it costs: Trainer steps (graph replays) of 784-H-10 / 784-H1-H2-10 for hidden
sizes the fused forms were not written around, at several batches.  usage: mlp_shape_sweep.py [H or H1xH2 ...]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import taper_amd as T  # noqa: E402
import bench  # noqa: E402

shapes = sys.argv[1:] or ["128", "100", "96", "64", "256", "512", "128x64", "100x50", "256x128"]
ds = T.MNISTDataset.synthetic(60000, seed=7)
for sh in shapes:
    hs = [int(v) for v in sh.split("x")]
    line = []
    for b in (64, 256, 1024, 4096, 16384):
        dims = [784] + hs + [10]
        layers = []
        for i in range(len(dims) - 1):
            layers.append(T.Linear(dims[i], dims[i + 1], True, 1 + i))
            if i + 2 < len(dims):
                layers.append(T.ReLU())
        model = T.Sequential(layers)
        opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
        tr = T.Trainer(model, opt)
        loader = T.DataLoader(ds, b, False)
        steps = min(20000, max(100, 4_000_000 // b))
        bench.run_steps(T, tr, loader, max(steps // 8, 3))
        T.Device.sync()
        t0 = time.perf_counter()
        bench.run_steps(T, tr, loader, steps)
        T.Device.sync()
        us = (time.perf_counter() - t0) / steps * 1e6
        flops = sum(6.0 * b * dims[i] * dims[i + 1] for i in range(len(dims) - 1)) - 2.0 * b * dims[0] * dims[1]   # (no dX of the first layer)
        line.append(f"b{b}: {us:7.1f} us ({flops / us / 1e6 / 157.3:.3f})")
        del tr, opt, model, loader
    print(f"784-{'-'.join(map(str, hs))}-10   " + "   ".join(line), flush=True)

#!/usr/bin/env python3
"""Where the large-batch MLP step (th_mlp2_xent / _deep) starts to pay: Trainer steps of 784-128-10 and 784-128-64-10 at small batches, run once
per TAPER_MLP2_MIN_BATCH setting (the choice is read once per process).  usage: TAPER_MLP2_MIN_BATCH=32 mlp_min_batch_probe.py"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import taper_amd as T  # noqa: E402
import bench  # noqa: E402

ds = T.MNISTDataset.synthetic(60000, seed=7)
for key in ("mlp_baseline", "mlp_example"):
    line = []
    for b in [int(v) for v in os.environ.get("BATCHES", "64,128,256,384,512,768").split(",")]:
        model = bench.build_model(T, key)
        opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
        tr = T.Trainer(model, opt)
        loader = T.DataLoader(ds, b, False)
        steps = 20000
        bench.run_steps(T, tr, loader, 2000)
        T.Device.sync()
        t0 = time.perf_counter()
        bench.run_steps(T, tr, loader, steps)
        T.Device.sync()
        line.append(f"b{b}: {(time.perf_counter() - t0) / steps * 1e6:6.2f}")
        del tr, opt, model, loader
    print(f"min_batch={os.environ.get('TAPER_MLP2_MIN_BATCH', 'default(480)')} {key}: " + "  ".join(line), flush=True)

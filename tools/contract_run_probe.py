#!/usr/bin/env python3
"""Where the contract run's 20 steps spend their wall time (bench.py --steps 20 --warmup 5): Python-side stamps around the pieces of the timed
region; TAPER_TRACE_EPOCH=1 adds the host library's own timeline of the same call."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import taper_amd as T
import bench

model = bench.build_model(T, "mlp_baseline")
opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
tr = T.Trainer(model, opt)
ds = T.MNISTDataset.synthetic(60000, seed=1)
loader = T.DataLoader(ds, 64, False)
bench.run_steps(T, tr, loader, 300)
bench.run_steps(T, tr, loader, 20)
res = []
for rep in range(30):
    T.Device.sync()
    t0 = time.perf_counter()
    T.Device.sync()
    t1 = time.perf_counter()
    ep = tr.run_epoch(loader, T.Trainer.GRAPH, max_steps=20)
    t2 = time.perf_counter()
    T.Device.sync()
    t3 = time.perf_counter()
    res.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
import numpy as np
r = np.array(res[5:])
print("idle Device.sync %.1f us; run_epoch(20 steps) %.1f us (min %.1f); Device.sync behind it %.1f us" % (np.median(r[:, 0]), np.median(r[:, 1]), r[:, 1].min(), np.median(r[:, 2])))

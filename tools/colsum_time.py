#!/usr/bin/env python3
"""th_colsum ([rows][cols] -> [cols], the bias gradient of a Linear layer outside the fused steps: tensor.rs:686-691) timed with events.
usage: colsum_time.py [rows cols ...]     default: the shapes of 784-256-10 / 784-512-10 at 16 384 rows and of a 4 096-wide layer"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402

ctx = hip.Ctx(0)
args = [int(v) for v in sys.argv[1:]] or [16384, 256, 16384, 512, 16384, 10, 4096, 4096, 60000, 128, 1024, 128]
rng = np.random.default_rng(0)
for rows, cols in zip(args[0::2], args[1::2]):
    x = rng.standard_normal((rows, cols)).astype(np.float32)
    dx, dy = ctx.upload(x), ctx.empty(cols)
    for _ in range(5):
        ctx.call("th_colsum", dx, dy, rows, cols)
    e0, e1 = hip.Event(), hip.Event()
    n = 200
    ctx.record(e0)
    for _ in range(n):
        ctx.call("th_colsum", dx, dy, rows, cols)
    ctx.record(e1)
    us = hip.Ctx.elapsed_ms(e0, e1) / n * 1e3
    got = ctx.download(dy, (cols,))
    ref = x.astype(np.float64).sum(0)
    err = float(np.max(np.abs(got - ref)) / max(1e-30, np.max(np.abs(ref))))
    print(f"th_colsum [{rows}][{cols}]: {us:7.2f} us per call (launch gaps included), {rows * cols * 4 / us / 1e6:6.2f} TB/s, max rel err {err:.1e}")

#!/usr/bin/env python3
"""Per-layer time of th_conv3x3_fwd (bias + ReLU fused) for the conv shapes of the two CNNs, back to back in one stream.
usage: bench_conv.py [reps]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402
from taper_amd._lib import hip as lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
LAYERS = [(256, 1, 28, 28, 32), (256, 32, 28, 28, 32), (256, 32, 14, 14, 64), (256, 64, 14, 14, 64), (256, 64, 7, 7, 128)]
for n, ci, h, w, co in LAYERS:
    x = ctx.upload(rng.standard_normal((n, ci, h, w)).astype(np.float32))
    wt = ctx.upload(rng.standard_normal((co, ci, 3, 3)).astype(np.float32))
    b = ctx.upload(rng.standard_normal(co).astype(np.float32))
    y = ctx.empty(n * co * h * w)
    call = lambda: ctx.call("th_conv3x3_fwd", x, wt, b, y, n, ci, h, w, co, 1, 0, 1)
    for _ in range(5):
        call()
    e0, e1 = hip.Event(), hip.Event()
    ctx.record(e0)
    for _ in range(reps):
        call()
    ctx.record(e1)
    us = hip.Ctx.elapsed_ms(e0, e1) * 1e3 / reps
    gf = 2 * 9 * ci * co * h * w * n / 1e9
    line = f"conv {ci:3d}->{co:3d} {h:2d}x{w:2d} batch {n}: {us:7.2f} us  {gf / (us * 1e-6) / 1e3:6.1f} TFLOP/s"
    if h % 2 == 0 and lib.th_conv3x3_pool2_supported(ci, h, w, co, 1):
        yp, arg = ctx.empty(n * co * h * w // 4), ctx.empty(n * co * h * w // 4, np.int64)
        pool = lambda: ctx.call("th_maxpool2d_fwd", y, yp, arg, n, co, h, w, 2, 2, 2, 2, 0, 0)
        fused = lambda: ctx.call("th_conv3x3_pool2_fwd", x, wt, b, yp, n, ci, h, w, co, 1, 1)
        t = []
        for f in (pool, fused):
            for _ in range(5):
                f()
            ctx.record(e0)
            for _ in range(reps):
                f()
            ctx.record(e1)
            t.append(hip.Ctx.elapsed_ms(e0, e1) * 1e3 / reps)
        line += f"   + 2x2 max pool {t[0]:6.2f} us; fused conv+pool {t[1]:6.2f} us"
    print(line)

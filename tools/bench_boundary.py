#!/usr/bin/env python3
"""Per-kernel fixed cost on this box: a chain of N trivial dependent kernels
(th_fill_f32 of 4 floats) replayed as one hipGraph vs launched eagerly."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402

ctx = hip.Ctx(0)
buf = ctx.zeros(1024)
N = 200
for label in ("graph", "eager"):
    if label == "graph":
        ctx.graph_begin()
        for _ in range(N):
            ctx.call("th_fill_f32", buf, 1.0, 4)
        g = ctx.graph_end()
        run = lambda: ctx.graph_launch(g)
    else:
        def run():
            for _ in range(N):
                ctx.call("th_fill_f32", buf, 1.0, 4)
    for _ in range(5):
        run()
    ctx.sync()
    e0, e1 = hip.Event(), hip.Event()
    t0 = time.perf_counter()
    ctx.record(e0)
    R = 20
    for _ in range(R):
        run()
    ctx.record(e1)
    ms = hip.Ctx.elapsed_ms(e0, e1)
    wall = time.perf_counter() - t0
    print(f"{label}: {ms * 1e3 / (R * N):.3f} us per kernel (events), host wall {wall * 1e6 / (R * N):.3f} us per kernel")

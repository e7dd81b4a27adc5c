#!/usr/bin/env python3
"""Event-timed th_mlp3_xent (two launches: rows + gradients) at the reference CNN's classifier shape and the example MLP's."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402

ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
for B, in_f, h1, h2, c, need_dx in ((256, 128, 128, 64, 10, True), (256, 784, 128, 64, 10, False), (64, 128, 128, 64, 10, True)):
    x, y = ctx.upload(rng.uniform(-1, 1, (B, in_f)).astype(np.float32)), ctx.upload(rng.integers(0, c, B).astype(np.float32))
    layers, keep = (hip.Mlp3Layer * 3)(), []
    for l, (o, i) in enumerate(((h1, in_f), (h2, h1), (c, h2))):
        w, b, gw, gb = ctx.upload(rng.uniform(-.1, .1, (o, i)).astype(np.float32)), ctx.zeros(o), ctx.empty(o * i), ctx.empty(o)
        keep.append((w, b, gw, gb))
        layers[l] = hip.Mlp3Layer(int(w), int(b), int(gw), int(gb), None, None, o)
    gx = ctx.empty(B * in_f) if need_dx else None
    loss, nc = ctx.empty(1), ctx.empty(1)
    lp = C.cast(layers, C.c_void_p)
    call = lambda: ctx.call("th_mlp3_xent", x, y, B, in_f, lp, gx, loss, nc, None, 0, None, 0, None, None)
    for _ in range(20):
        call()
    e0, e1 = hip.Event(), hip.Event()
    ctx.record(e0)
    for _ in range(300):
        call()
    ctx.record(e1)
    ctx.sync()
    print(f"B={B} {in_f}-{h1}-{h2}-{c} dx={need_dx}: {hip.Ctx.elapsed_ms(e0, e1) * 1e3 / 300:.2f} us per call (2 launches, eager back to back)")

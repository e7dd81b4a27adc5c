#!/usr/bin/env python3
"""The reference CNN's front + classifier at batch N (default 256), event-timed back to back: th_conv_chain_fwd + th_mlp3_xent (three launches)
against th_conv_chain_mlp3_xent (two: the classifier's rows in the chain launch's last epilogue)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402

REFERENCE = [(1, 32, 0), (32, 32, 1), (32, 64, 0), (64, 64, 1), (64, 128, 2)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
x = ctx.upload(rng.random((n, 1, 28, 28), dtype=np.float32))
y = ctx.upload(rng.integers(0, 10, n).astype(np.float32))
bufs = [(ctx.upload((rng.standard_normal((co, ci, 3, 3)) * 0.1).astype(np.float32)), ctx.upload(rng.standard_normal(co).astype(np.float32) * 0.1))
        for ci, co, _ in REFERENCE]
stages, ns = hip.conv_stages([(w, b, co, post) for (w, b), (_, co, post) in zip(bufs, REFERENCE)])
sp = C.cast(stages, C.c_void_p)
layers, keep = (hip.Mlp3Layer * 3)(), []
for l, (o, i) in enumerate(((128, 128), (64, 128), (10, 64))):
    b = (ctx.upload(rng.uniform(-.1, .1, (o, i)).astype(np.float32)), ctx.zeros(o), ctx.empty(o * i), ctx.empty(o))
    keep.append(b)
    layers[l] = hip.Mlp3Layer(int(b[0]), int(b[1]), int(b[2]), int(b[3]), None, None, o)
lp = C.cast(layers, C.c_void_p)
means, cnt, gx, gb, loss, nc = ctx.empty(n * 128), ctx.empty(n * 128), ctx.empty(n * 128), ctx.empty(128), ctx.empty(1), ctx.empty(1)
gap = hip.Mlp3Gap(int(cnt), int(gb), 49, None)
gp = C.cast(C.pointer(gap), C.c_void_p)


def three():
    ctx.call("th_conv_chain_fwd", x, sp, ns, means, cnt, n, 1, 28, 28)
    ctx.call("th_mlp3_xent", means, y, n, 128, lp, gx, loss, nc, None, 0, None, 0, None, gp)


def two():
    ctx.call("th_conv_chain_mlp3_xent", x, sp, ns, means, cnt, n, 1, 28, 28, y, lp, gx, loss, nc, None, 0, None, 0, None, gp)


def chain():
    ctx.call("th_conv_chain_fwd", x, sp, ns, means, cnt, n, 1, 28, 28)


def timed(fn, reps=300):
    for _ in range(30):
        fn()
    e0, e1 = hip.Event(), hip.Event()
    ctx.record(e0)
    for _ in range(reps):
        fn()
    ctx.record(e1)
    ctx.sync()
    return hip.Ctx.elapsed_ms(e0, e1) * 1e3 / reps


for _ in range(2):
    print(f"batch {n}: chain alone {timed(chain):.1f} us; chain + th_mlp3_xent (3 launches) {timed(three):.1f} us; th_conv_chain_mlp3_xent (2 launches) {timed(two):.1f} us")

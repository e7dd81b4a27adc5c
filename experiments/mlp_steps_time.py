#!/usr/bin/env python3
"""Event-timed th_mlp2_steps (one persistent launch for many steps of the 784-128-10 MLP) -- us per step."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402
from taper_amd._lib import hip as lib  # noqa: E402

ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
B, in_f, hid, c = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 784, 128, 10)
ws = [rng.uniform(-.05, .05, (hid, in_f)).astype(np.float32), np.zeros(hid, np.float32), rng.uniform(-.2, .2, (c, hid)).astype(np.float32), np.zeros(c, np.float32)]
t, dlr = ctx.upload(np.zeros(1, np.int32)), ctx.upload(np.array([1e-3], np.float32))
P, M, V = [ctx.upload(w) for w in ws], [ctx.zeros(w.size) for w in ws], [ctx.zeros(w.size) for w in ws]
fuse = (hip.AdamFuse * 4)(*[hip.AdamFuse(int(P[i]), int(M[i]), int(V[i]), int(t), int(dlr), 0.9, 0.999, 1e-8, 1e-4) for i in range(4)])
herr = C.c_void_p()
lib.th_host_malloc(ctx.h, 64, C.byref(herr))
C.cast(herr, C.POINTER(C.c_int))[0] = 0
for steps in (16, 128, 512):
    x, y = ctx.upload(rng.uniform(0, 1, (steps * B, in_f)).astype(np.float32)), ctx.upload(rng.integers(0, c, steps * B).astype(np.float32))
    loss, met, st = ctx.empty(1), ctx.zeros(2 * steps), ctx.upload(np.zeros(2, np.int64))
    call = lambda: ctx.call("th_mlp2_steps", x, y, steps, B, in_f, hid, c, C.cast(fuse, C.c_void_p), loss, met, steps, st, B, herr)
    for _ in range(3):
        call()
    e0, e1 = hip.Event(), hip.Event()
    ctx.record(e0)
    reps = 10
    for _ in range(reps):
        call()
    ctx.record(e1)
    ctx.sync()
    print(f"B={B} {in_f}-{hid}-{c}: {steps} steps per launch: {hip.Ctx.elapsed_ms(e0, e1) * 1e3 / (reps * steps):.2f} us per step (err {C.cast(herr, C.POINTER(C.c_int))[0]})")

#!/usr/bin/env python3
"""us per step of th_mlp2_train_steps (whole steps of the 784-128-10 MLP in one launch)."""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--hidden", type=int, default=128)
ap.add_argument("--steps", type=int, default=937)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
n, inf, hid, cls = 60000, 784, a.hidden, 10
x = ctx.upload(rng.uniform(0, 1, (n, inf)).astype(np.float32))
y = ctx.upload(rng.integers(0, cls, n).astype(np.float32))
idx = ctx.upload(rng.permutation(n).astype(np.int32))
shapes = [(hid, inf), (hid,), (cls, hid), (cls,)]
P = [ctx.upload(rng.uniform(-0.05, 0.05, s).astype(np.float32)) for s in shapes]
M = [ctx.zeros(int(np.prod(s))) for s in shapes]
V = [ctx.zeros(int(np.prod(s))) for s in shapes]
tick, lr = ctx.upload(np.array([0, 0], np.int32)), ctx.upload(np.array([1e-3], np.float32))
metrics, state, st = ctx.zeros(2 * (a.steps + 2)), ctx.upload(np.zeros(2, np.int64)), ctx.upload(np.zeros(1, np.int32))


def run():
    ctx.call("th_mlp2_train_steps", x, y, idx, n, 0, a.batch, a.steps, inf, hid, cls, P[0], P[1], P[2], P[3], M[0], V[0], M[1], V[1], M[2],
             V[2], M[3], V[3], tick, lr, 0.9, 0.999, 1e-8, 1e-4, metrics, a.steps + 2, state, st)


run()
ctx.sync()
e0, e1 = hip.Event(), hip.Event()
ctx.record(e0)
for _ in range(a.reps):
    run()
ctx.record(e1)
ms = hip.Ctx.elapsed_ms(e0, e1) / a.reps
print(f"{ms * 1e3 / a.steps:.3f} us/step  ({a.batch * a.steps / ms / 1e3:.2f} M samples/s)  status={ctx.download(st, 1, np.int32)[0]}  "
      f"loss[0]={ctx.download(metrics, 2)[0]:.4f}")

from taper_amd._lib import hip as lib  # noqa: E402
if hasattr(lib, "th_debug_mlp_epoch_prof"):
    import ctypes as C
    lib.th_debug_mlp_epoch_prof.argtypes = [C.c_void_p, C.c_void_p]
    out = (C.c_longlong * 8)()
    lib.th_debug_mlp_epoch_prof(ctx.h, out)
    names = ["1 forward slab + publish", "2 all-gather", "3a logits", "3b softmax", "3c dZ slab + dW2 + log", "5a Adam W2/b2",
             "4+5b dW1 slab + Adam + end barrier", "stage rows"]
    for nm, v in zip(names, out):
        print(f"{v * 0.01 / a.steps:8.3f} us/step  {nm}")

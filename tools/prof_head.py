#!/usr/bin/env python3
"""In-kernel phase timing of the classifier-head kernel (needs a PROFILE=1 build:
make -C taper_amd/csrc clean all PROFILE=1).  wall_clock64 ticks at 100 MHz."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402
from taper_amd.hip import AdamFuse  # noqa: E402
from taper_amd._lib import hip as lib  # noqa: E402

ctx = hip.Ctx(0)
lib.th_debug_head_prof.argtypes = [C.c_void_p, C.c_void_p]
lib.th_debug_head_prof.restype = C.c_int
B, K, Cc = 64, 128, 10
rng = np.random.default_rng(0)
h, w, b = ctx.upload(rng.random((B, K), np.float32)), ctx.upload(rng.random((Cc, K), np.float32)), ctx.upload(rng.random(Cc, np.float32))
y = ctx.upload(rng.integers(0, Cc, B).astype(np.float32))
loss, nc, dh, dw, db = ctx.empty(1), ctx.empty(1), ctx.empty(B * K), ctx.empty(Cc * K), ctx.empty(Cc)


FUSE = "--fuse" in sys.argv
tick, lr = ctx.upload(np.array([1, 0], np.int32)), ctx.upload(np.array([1e-3], np.float32))
m2, v2, mb2, vb2 = ctx.zeros(Cc * K), ctx.zeros(Cc * K), ctx.zeros(Cc), ctx.zeros(Cc)
wf = AdamFuse(int(w), int(m2), int(v2), int(tick), int(lr), 0.9, 0.999, 1e-8, 1e-4)
bf = AdamFuse(int(b), int(mb2), int(vb2), int(tick), int(lr), 0.9, 0.999, 1e-8, 1e-4)
fuse_args = (tick, C.byref(wf), C.byref(bf)) if FUSE else (None, None, None)
names = ["entry->prefetch+Wstage", "Hstage+sync", "logits", "softmax", "dH", "dW+db", "reduce", "grads out", "exit->next kernel start"]
acc = np.zeros(9)
N = 50
for it in range(N + 5):
    ctx.call("th_fill_f32", loss, 0.0, 4)
    ctx.call("th_linear_xent_head", h, w, b, y, B, K, Cc, None, loss, nc, dh, dw, db, None, 0, None, 0, *fuse_args)
    out = (C.c_longlong * 16)()
    lib.th_debug_head_prof(ctx.h, out)
    ts = [out[i] for i in range(9)] + [out[15]]
    if it >= 5:
        acc += np.diff(ts) * 0.01
for n, v in zip(names, acc / N):
    print(f"{v:7.3f} us  {n}")
print(f"{acc.sum() / N:7.3f} us  total inside + tail")

#!/usr/bin/env python3
"""In-kernel phase timing of one workgroup of the matrix-core conv kernel (needs `make PROFILE=1`).
usage: prof_conv.py n c_in h w c_out"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402
from taper_amd._lib import hip as lib  # noqa: E402

n, c_in, h, w, c_out = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (256, 32, 28, 28, 32)
mode = sys.argv[6] if len(sys.argv) > 6 else "plain"       # plain | pool (fused 2x2 max-pool) | gap (fused global average pool)
ctx = hip.Ctx(0)
lib.th_debug_conv_prof.argtypes = [C.c_void_p, C.c_void_p]
lib.th_debug_conv_prof.restype = C.c_int
rng = np.random.default_rng(0)
x = ctx.upload(rng.standard_normal((n, c_in, h, w)).astype(np.float32))
wt = ctx.upload(rng.standard_normal((c_out, c_in, 3, 3)).astype(np.float32))
b = ctx.upload(rng.standard_normal(c_out).astype(np.float32))
y = ctx.empty(n * c_out * h * w)
cnt = ctx.empty(n * c_out)
names = ["entry -> plans done", "first stage (load, store, sync)", "k loop (all passes)", "epilogue stores"]
acc = np.zeros(4)
N = 20
e0, e1 = hip.Event(), hip.Event()
for it in range(N + 3):
    ctx.record(e0)
    if mode == "pool":
        ctx.call("th_conv3x3_pool2_fwd", x, wt, b, y, n, c_in, h, w, c_out, 1, 1)
    elif mode == "gap":
        ctx.call("th_conv3x3_gap_fwd", x, wt, b, y, cnt, n, c_in, h, w, c_out, 1, 1)
    else:
        ctx.call("th_conv3x3_fwd", x, wt, b, y, n, c_in, h, w, c_out, 1, 0, 1)
    ctx.record(e1)
    ms = hip.Ctx.elapsed_ms(e0, e1)
    out = (C.c_longlong * 8)()
    lib.th_debug_conv_prof(ctx.h, out)
    if it >= 3:
        acc += np.diff([out[i] for i in range(5)]) * 0.01
        if out[6] > out[5] > 0 and out[3] > out[2]:
            ghz = (out[6] - out[5]) / ((out[3] - out[2]) * 10.0)
for nm, v in zip(names, acc / N):
    print(f"{v:8.3f} us  {nm}")
print(f"{acc.sum() / N:8.3f} us  workgroup total;  kernel {ms * 1e3:.1f} us" + (f";  shader clock through the k loop {ghz:.2f} GHz" if "ghz" in dir() else ""))

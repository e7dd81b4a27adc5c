#!/usr/bin/env python3
"""Phase timing of one workgroup of the reference conv chain (needs the TH_PROFILE build of tools/prof_chain.sh)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402
from taper_amd._lib import hip as lib  # noqa: E402

REFERENCE = [(1, 32, 0), (32, 32, 1), (32, 64, 0), (64, 64, 1), (64, 128, 2)]
import os
n = int(os.environ.get("N_IMG", 256))
ctx = hip.Ctx(0)
lib.th_debug_chain_prof.argtypes = [C.c_void_p, C.c_void_p]
lib.th_debug_chain_prof.restype = C.c_int
rng = np.random.default_rng(0)
x = ctx.upload(rng.random((n, 1, 28, 28), dtype=np.float32))
bufs = []
for c_in, c_out, post in REFERENCE:
    b = np.sqrt(6.0 / (c_in * 9))
    bufs.append((ctx.upload(rng.uniform(-b, b, (c_out, c_in, 3, 3)).astype(np.float32)), ctx.upload(rng.uniform(-.1, .1, c_out).astype(np.float32))))
stages, ns = hip.conv_stages([(w, b, c_out, post) for (w, b), (_, c_out, post) in zip(bufs, REFERENCE)])
sp = C.cast(stages, C.c_void_p)
y, cnt = ctx.empty(n * 128), ctx.empty(n * 128)
names = ["image -> LDS", "conv1 (VALU) -> A1", "conv2 k loop", "conv2 -> tile, zero A2", "pool -> A2", "conv3 k loop (+ zero A3)", "conv3 -> A3",
         "conv4 k loop", "conv4 -> tile, zero A4", "pool -> A4", "conv5 k loop -> tile", "global mean -> y"]
acc = np.zeros(12)
N = 20
e0, e1 = hip.Event(), hip.Event()
tot = 0.0
BURST = int(sys.argv[1]) if len(sys.argv) > 1 else 1      # launches back to back before the stamps are read (sustained-clock state)
for it in range(N + 3):
    for _ in range(BURST - 1):
        ctx.call("th_conv_chain_fwd", x, sp, ns, y, cnt, n, 1, 28, 28)
    ctx.record(e0)
    ctx.call("th_conv_chain_fwd", x, sp, ns, y, cnt, n, 1, 28, 28)
    ctx.record(e1)
    ms = hip.Ctx.elapsed_ms(e0, e1)
    out = (C.c_longlong * 32)()
    lib.th_debug_chain_prof(ctx.h, out)
    if it >= 3:
        acc += np.diff([out[i] for i in range(13)]) * 0.01
        tot += ms
        ghz = (out[21] - out[20]) / ((out[12] - out[0]) * 10.0)
print("conv3 passes from the stage's start (us):", [round((out[13 + i] - out[5]) * 0.01, 2) for i in range(4)], "k loop ends", round((out[6] - out[5]) * 0.01, 2))
print("conv4 passes from the stage's start (us):", [round((out[22 + i] - out[7]) * 0.01, 2) for i in range(8)], "k loop ends", round((out[8] - out[7]) * 0.01, 2))
for nm, v in zip(names, acc / N):
    print(f"{v:8.3f} us  {nm}")
print(f"{acc.sum() / N:8.3f} us  workgroup total;  kernel (events, eager) {tot / N * 1e3:.1f} us;  shader clock over the workgroup's life {ghz:.2f} GHz (burst of {BURST})")

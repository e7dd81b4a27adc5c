#!/usr/bin/env python3
"""Phase timing of one workgroup of the simple CNN's chain + classifier rows (th_conv_chain_head_fwd; needs the TH_PROFILE build:
tools/prof_chain.sh builds it, then run this instead of prof_chain.py)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402
from taper_amd._lib import hip as lib  # noqa: E402


class ChainHead(C.Structure):
    _fields_ = [("d_w", C.c_void_p), ("d_bias", C.c_void_p), ("d_targets", C.c_void_p), ("classes", C.c_int), ("d_dl", C.c_void_p),
                ("d_rowstat", C.c_void_p), ("d_cbpart", C.c_void_p), ("d_tick", C.c_void_p)]


SIMPLE = [(1, 32, 1), (32, 64, 1)]
import os
n, K = int(os.environ.get("CHAIN_N", "256")), 3136
ctx = hip.Ctx(0)
lib.th_debug_chain_prof.argtypes = [C.c_void_p, C.c_void_p]
lib.th_debug_chain_prof.restype = C.c_int
rng = np.random.default_rng(0)
x = ctx.upload(rng.random((n, 1, 28, 28), dtype=np.float32))
bufs = []
for c_in, c_out, post in SIMPLE:
    b = np.sqrt(6.0 / (c_in * 9))
    bufs.append((ctx.upload(rng.uniform(-b, b, (c_out, c_in, 3, 3)).astype(np.float32)), ctx.upload(rng.uniform(-.1, .1, c_out).astype(np.float32))))
stages, ns = hip.conv_stages([(w, b, c_out, post) for (w, b), (_, c_out, post) in zip(bufs, SIMPLE)])
sp = C.cast(stages, C.c_void_p)
w = ctx.upload((rng.uniform(-1, 1, (10, K)) * np.sqrt(2.0 / K)).astype(np.float32))
bias, yt = ctx.upload(rng.uniform(-.1, .1, 10).astype(np.float32)), ctx.upload(rng.integers(0, 10, n).astype(np.float32))
y, dl, rs, cbp = ctx.empty(n * K), ctx.empty(n * 16), ctx.empty(n * 2), ctx.empty(n * 64)
head = ChainHead(int(w), int(bias), int(yt), 10, int(dl), int(rs), int(cbp), None)
names = ["image -> LDS, weights requested, zero A", "conv1 MFMAs", "conv1 pooled -> A", "conv2 k loop (first wave)", "-", "conv2 pooled -> y, XM (+ the other waves' k loops)",
         "logits (partials, reduce)", "softmax, dlogits", "masked dX -> XM", "channel sums", "drain"]
acc = np.zeros(11)
N, tot = 20, 0.0
e0, e1 = hip.Event(), hip.Event()
BURST = int(sys.argv[1]) if len(sys.argv) > 1 else 1
call = lambda: ctx.call("th_conv_chain_head_fwd", x, sp, ns, y, n, 1, 28, 28, C.byref(head))
for it in range(N + 3):
    for _ in range(BURST - 1):
        call()
    ctx.record(e0)
    call()
    ctx.record(e1)
    ms = hip.Ctx.elapsed_ms(e0, e1)
    out = (C.c_longlong * 32)()
    lib.th_debug_chain_prof(ctx.h, out)
    if it >= 3:
        acc += np.diff([out[i] for i in range(12)]) * 0.01
        tot += ms
        ghz = (out[21] - out[20]) / ((out[11] - out[0]) * 10.0)
for nm, v in zip(names, acc / N):
    print(f"{v:8.3f} us  {nm}")
print("conv2 passes as wave 0 sees them (us from kernel entry):", [round((out[i] - out[0]) * 0.01, 2) for i in range(13, 17)], "k loop ends", round((out[4] - out[0]) * 0.01, 2))
print(f"{acc.sum() / N:8.3f} us  workgroup total;  kernel (events, eager) {tot / N * 1e3:.1f} us;  shader clock {ghz:.2f} GHz (burst of {BURST})")

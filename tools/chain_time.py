#!/usr/bin/env python3
"""Event-timed launches of the conv chains (th_conv_chain_fwd) at batch 256 beside the layer-by-layer launches they replace."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402

REFERENCE = [(1, 32, 0), (32, 32, 1), (32, 64, 0), (64, 64, 1), (64, 128, 2)]
SIMPLE = [(1, 32, 1), (32, 64, 1)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
x = ctx.upload(rng.random((n, 1, 28, 28), dtype=np.float32))


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    e0, e1 = hip.Event(), hip.Event()
    ctx.record(e0)
    for _ in range(reps):
        fn()
    ctx.record(e1)
    ctx.sync()
    return hip.Ctx.elapsed_ms(e0, e1) * 1e3 / reps


for name, spec in (("reference", REFERENCE), ("simple", SIMPLE)):
    bufs = []
    for c_in, c_out, post in spec:
        b = np.sqrt(6.0 / (c_in * 9))
        bufs.append((ctx.upload(rng.uniform(-b, b, (c_out, c_in, 3, 3)).astype(np.float32)), ctx.upload(rng.uniform(-.1, .1, c_out).astype(np.float32))))
    stages, ns = hip.conv_stages([(w, b, c_out, post) for (w, b), (_, c_out, post) in zip(bufs, spec)])
    y = ctx.empty(n * 128 * 49)
    cnt = ctx.empty(n * 128)
    sp = C.cast(stages, C.c_void_p)
    t_chain = timed(lambda: ctx.call("th_conv_chain_fwd", x, sp, ns, y, cnt, n, 1, 28, 28))
    maps = [ctx.empty(n * 32 * 784) for _ in spec]

    def layered():
        cur, hw = x, 28
        for (c_in, c_out, post), (w, b), out in zip(spec, bufs, maps):
            if post == 0:
                ctx.call("th_conv3x3_fwd", cur, w, b, out, n, c_in, hw, hw, c_out, 1, 0, 1)
            elif post == 1:
                ctx.call("th_conv3x3_pool2_fwd", cur, w, b, out, n, c_in, hw, hw, c_out, 1, 1)
                hw //= 2
            else:
                ctx.call("th_conv3x3_gap_fwd", cur, w, b, out, cnt, n, c_in, hw, hw, c_out, 1, 1)
            cur = out
    t_lay = timed(layered)
    flops = 0
    hw = 28
    for c_in, c_out, post in spec:
        flops += 2 * n * hw * hw * c_out * c_in * 9
        hw = hw // 2 if post == 1 else hw
    hip.hip.th_debug_set_chain_generic(1)     # the same net through the kernel that takes its stages as arguments
    t_rt = timed(lambda: ctx.call("th_conv_chain_fwd", x, sp, ns, y, cnt, n, 1, 28, 28))
    hip.hip.th_debug_set_chain_generic(0)
    print(f"{name} batch {n}: compiled chain {t_chain:.1f} us ({flops / t_chain * 1e-6:.1f} TF), run-time-described chain {t_rt:.1f} us "
          f"({flops / t_rt * 1e-6:.1f} TF), layered {t_lay:.1f} us (eager launches back to back)")


# maps that are not MNIST-shaped (r04): the instance that takes the map size as an argument, beside the layer-by-layer launches
for name, c0, s0, spec in (("32x32x1 -> 16+pool -> 32+pool -> 64+mean", 1, 32, [(1, 16, 1), (16, 32, 1), (32, 64, 2)]),
                           ("24x24x16 -> 32 -> 32+pool -> 64 (conv row end)", 16, 24, [(16, 32, 0), (32, 32, 1), (32, 64, 0)]),
                           ("16x16x32 -> 64+pool -> 128+mean", 32, 16, [(32, 64, 1), (64, 128, 2)])):
    xg = ctx.upload(rng.random((n, c0, s0, s0), dtype=np.float32))
    bufs = []
    for c_in, c_out, post in spec:
        b = np.sqrt(6.0 / (c_in * 9))
        bufs.append((ctx.upload(rng.uniform(-b, b, (c_out, c_in, 3, 3)).astype(np.float32)), ctx.upload(rng.uniform(-.1, .1, c_out).astype(np.float32))))
    stages, ns = hip.conv_stages([(w, b, c_out, post) for (w, b), (_, c_out, post) in zip(bufs, spec)])
    sp = C.cast(stages, C.c_void_p)
    kind = hip.hip.th_conv_chain_supported(c0, s0, s0, sp, ns)
    y = ctx.empty(n * 128 * s0 * s0)
    cnt = ctx.empty(n * 128)
    maps = [ctx.empty(n * 64 * s0 * s0) for _ in spec]

    def layered_g():
        cur, hw = xg, s0
        for (c_in, c_out, post), (w, b), out in zip(spec, bufs, maps):
            if post == 0:
                ctx.call("th_conv3x3_fwd", cur, w, b, out, n, c_in, hw, hw, c_out, 1, 0, 1)
            elif post == 1:
                ctx.call("th_conv3x3_fwd", cur, w, b, out, n, c_in, hw, hw, c_out, 1, 0, 1)
                ctx.call("th_maxpool2d_fwd", out, y, None, n, c_out, hw, hw, 2, 2, 2, 2, 0, 0)
                hw //= 2
            else:
                ctx.call("th_conv3x3_fwd", cur, w, b, out, n, c_in, hw, hw, c_out, 1, 0, 1)
                ctx.call("th_avgpool2d_fwd", out, y, n, c_out, hw, hw, hw, hw, hw, hw, 0, 0)
            cur = out
    flops, hw = 0, s0
    for c_in, c_out, post in spec:
        flops += 2 * n * hw * hw * c_out * c_in * 9
        hw = hw // 2 if post == 1 else hw
    if kind:
        t_c = timed(lambda: ctx.call("th_conv_chain_fwd", xg, sp, ns, y, cnt, n, c0, s0, s0))
        t_l = timed(layered_g)
        print(f"{name} batch {n}: chain (kind {kind}) {t_c:.1f} us ({flops / t_c * 1e-6:.1f} TF = {flops / t_c * 1e-6 / 157.3:.2f}), layer by layer {t_l:.1f} us")
    else:
        print(f"{name}: no chain")

# the simple CNN's two-launch step: the chain with the classifier rows in its last epilogue + the batch sums (th_wide_head_grads)
class _ChainHead(C.Structure):
    _fields_ = [("d_w", C.c_void_p), ("d_bias", C.c_void_p), ("d_targets", C.c_void_p), ("classes", C.c_int), ("d_dl", C.c_void_p),
                ("d_rowstat", C.c_void_p), ("d_cbpart", C.c_void_p), ("d_tick", C.c_void_p)]


K = 64 * 49
bufs = []
for c_in, c_out, post in SIMPLE:
    b = np.sqrt(6.0 / (c_in * 9))
    bufs.append((ctx.upload(rng.uniform(-b, b, (c_out, c_in, 3, 3)).astype(np.float32)), ctx.upload(rng.uniform(-.1, .1, c_out).astype(np.float32))))
stages, ns = hip.conv_stages([(w, b, c_out, post) for (w, b), (_, c_out, post) in zip(bufs, SIMPLE)])
sp = C.cast(stages, C.c_void_p)
w = ctx.upload((rng.uniform(-1, 1, (10, K)) * np.sqrt(2.0 / K)).astype(np.float32))
bias, yt = ctx.upload(rng.uniform(-.1, .1, 10).astype(np.float32)), ctx.upload(rng.integers(0, 10, n).astype(np.float32))
ymap, dl, rs, cbp = ctx.empty(n * K), ctx.empty(n * 16), ctx.empty(n * 2), ctx.empty(n * 64)
head = _ChainHead(int(w), int(bias), int(yt), 10, int(dl), int(rs), int(cbp), None)
dw, db, gcb, loss = ctx.empty(10 * K), ctx.empty(10), ctx.empty(64), ctx.empty(1)
t_head = timed(lambda: ctx.call("th_conv_chain_head_fwd", x, sp, ns, ymap, n, 1, 28, 28, C.byref(head)))
t_grads = timed(lambda: ctx.call("th_wide_head_grads", ymap, dl, rs, cbp, n, K, 10, 64, dw, db, gcb, loss, None, None, 0, None, 0, None, None, None))
print(f"simple batch {n}: chain + classifier rows {t_head:.1f} us, batch sums (no Adam) {t_grads:.1f} us")

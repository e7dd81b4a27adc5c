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

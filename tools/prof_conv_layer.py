#!/usr/bin/env python3
"""Phase timing of one workgroup of conv_layer_chain_kernel (th_conv3x3_fwd / _pool2_fwd / _gap_fwd on the four compiled layer geometries at
batch 256; needs the TH_PROFILE build: PROF_SCRIPT=tools/prof_conv_layer.py tools/prof_chain.sh)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402
from taper_amd._lib import hip as lib  # noqa: E402

ctx = hip.Ctx(0)
lib.th_debug_chain_prof.argtypes = [C.c_void_p, C.c_void_p]
lib.th_debug_chain_prof.restype = C.c_int
rng = np.random.default_rng(0)
n = 256
for ci, hw, co, post in [(32, 28, 32, 1), (32, 14, 64, 0), (64, 14, 64, 1), (64, 7, 128, 0)]:
    x = ctx.upload(rng.standard_normal((n, ci, hw, hw)).astype(np.float32))
    wt = ctx.upload((rng.standard_normal((co, ci, 3, 3)) * 0.05).astype(np.float32))
    b = ctx.upload(rng.standard_normal(co).astype(np.float32))
    y = ctx.empty(n * co * hw * hw)
    if post == 1:
        call = lambda: ctx.call("th_conv3x3_pool2_fwd", x, wt, b, y, n, ci, hw, hw, co, 1, 1)
    else:
        call = lambda: ctx.call("th_conv3x3_fwd", x, wt, b, y, n, ci, hw, hw, co, 1, 0, 1)
    e0, e1 = hip.Event(), hip.Event()
    acc, tot, N = np.zeros(4), 0.0, 20
    for it in range(N + 3):
        for _ in range(50):
            call()
        ctx.record(e0)
        call()
        ctx.record(e1)
        ms = hip.Ctx.elapsed_ms(e0, e1)
        out = (C.c_longlong * 32)()
        lib.th_debug_chain_prof(ctx.h, out)
        if it >= 3:
            acc += np.diff([out[i] for i in range(5)]) * 0.01
            tot += ms
    sp = (C.c_longlong * 512)()
    lib.th_debug_chain_span.argtypes = [C.c_void_p, C.c_void_p]
    lib.th_debug_chain_span(ctx.h, sp)
    st, en = np.array(sp[0::2], float) * 0.01, np.array(sp[1::2], float) * 0.01
    t0 = st.min()
    print(f"    workgroup starts after the first: median {np.median(st - t0):.2f}, p90 {np.quantile(st - t0, .9):.2f}, last {(st - t0).max():.2f} us; "
          f"durations median {np.median(en - st):.2f}, max {(en - st).max():.2f}; last end {(en - t0).max():.2f} us after the first start")
    names = ["weights requested, planes -> LDS", "k loop (first wave)", "map stores / wait for the other waves", "tile, pool / means"]
    print(f"{hw}x{hw} {ci}->{co} post {post}: kernel {tot / N * 1e3:.1f} us; workgroup 100: " + "; ".join(f"{nm} {v:.2f}" for nm, v in zip(names, acc / N)))

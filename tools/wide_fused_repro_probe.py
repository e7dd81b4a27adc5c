#!/usr/bin/env python3
"""Probe: th_linear_xent_wide_fused captured `steps` times in one graph, replayed from a restored state: which outputs move between replays?
usage: wide_fused_repro_probe.py [steps] [replays]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402
from taper_amd._lib import hip as lib, th_check  # noqa: E402
from taper_amd.hip import AdamFuse, WideFuse  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
replays = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = hip.Ctx(0)
batch, c_conv, hw, c, lr = 256, 64, 49, 10, 1e-2
k = c_conv * hw
rng = np.random.default_rng(5)
h = np.maximum(rng.standard_normal((batch, k)), 0).astype(np.float32)
p0 = dict(w=(rng.uniform(-1, 1, (c, k)) * np.sqrt(2.0 / k)).astype(np.float32), b=rng.uniform(-0.1, 0.1, c).astype(np.float32),
          cb=rng.uniform(-0.3, 0.3, c_conv).astype(np.float32))
y = rng.integers(0, c, batch).astype(np.float32)
dev = {n: ctx.upload(v) for n, v in p0.items()}
mom = {n: (ctx.zeros(v.size), ctx.zeros(v.size)) for n, v in p0.items()}
tick, lrd = ctx.upload(np.array([4, 0, 0, 0], np.int32)), ctx.upload(np.array([lr], np.float32))
fz = lambda n: AdamFuse(int(dev[n]), int(mom[n][0]), int(mom[n][1]), int(tick), int(lrd), 0.9, 0.999, 1e-8, 1e-4)
gcb = [ctx.empty(c_conv) for _ in range(steps)]
dw = [ctx.empty(c * k) for _ in range(steps)]
db_, cs_, loss = ctx.empty(c), ctx.empty(k), [ctx.empty(1) for _ in range(steps)]
hd, yd = ctx.upload(h), ctx.upload(y)
fs = [WideFuse(fz("w"), fz("b"), fz("cb"), int(gcb[s]), c_conv, hw) for s in range(steps)]


def enqueue():
    for s in range(steps):
        ctx.call("th_linear_xent_wide_fused", hd, dev["w"], dev["b"], yd, batch, k, c, loss[s], None, dw[s], db_, None, 0, None, 0, tick, cs_, C.byref(fs[s]))


def put(buf, a):
    a = np.ascontiguousarray(a)
    th_check(lib.th_memcpy_h2d(ctx.h, int(buf), a.ctypes.data, a.nbytes), "h2d")


def restore():
    for n, v in p0.items():
        put(dev[n], v)
        put(mom[n][0], np.zeros(v.size, np.float32))
        put(mom[n][1], np.zeros(v.size, np.float32))
    put(tick, np.array([4, 0, 0, 0], np.int32))


def collect():
    got = {n: ctx.download(dev[n], v.size) for n, v in p0.items()}
    got["gcb"] = np.stack([ctx.download(b, c_conv) for b in gcb])
    got["losses"] = np.array([ctx.download(l, 1)[0] for l in loss])
    got["dw"] = np.stack([ctx.download(d, c * k) for d in dw])
    got["t"] = ctx.download(tick, 4, np.int32)
    return got


for mode in ("eager", "graph"):
    if mode == "graph":
        ctx.graph_begin()
        enqueue()
        g = ctx.graph_end()
    runs = []
    for r in range(replays):
        restore()
        if mode == "graph":
            ctx.graph_launch(g)
        else:
            enqueue()
        runs.append(collect())
    for r in range(1, replays):
        diffs = {kk: float(np.abs(runs[r][kk].astype(np.float64) - runs[0][kk]).max()) for kk in runs[0]}
        prev = {kk: float(np.abs(runs[r][kk].astype(np.float64) - runs[r - 1][kk]).max()) for kk in runs[0]}
        per_step_dw = [float(np.abs(runs[r]["dw"][s] - runs[0]["dw"][s]).max()) for s in range(steps)]
        print(mode, "replay", r, "vs 0:", {kk: f"{v:.2e}" for kk, v in diffs.items() if v > 0}, "| vs previous:", {kk: f"{v:.2e}" for kk, v in prev.items() if v > 0},
              "| dw per step vs 0:", [f"{v:.1e}" for v in per_step_dw], "| losses", runs[r]["losses"], "t", runs[r]["t"])
    print(mode, "replay 0 losses", runs[0]["losses"], "t", runs[0]["t"])

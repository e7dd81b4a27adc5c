#!/usr/bin/env python3
"""Timeline of every workgroup of one matrix-core conv launch (needs `make PROFILE=1`): start / k-loop end / end stamps
(100 MHz wall clock) and the hardware id of each workgroup -> residency per CU, start spread, life.
usage: conv_timeline.py n c_in h w c_out"""
import ctypes as C
import sys
from collections import Counter
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402
from taper_amd._lib import hip as lib  # noqa: E402

n, c_in, h, w, c_out = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (256, 32, 28, 28, 32)
ctx = hip.Ctx(0)
lib.th_debug_conv_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.th_debug_conv_timeline.restype = C.c_int
rng = np.random.default_rng(0)
x = ctx.upload(rng.standard_normal((n, c_in, h, w)).astype(np.float32))
wt = ctx.upload(rng.standard_normal((c_out, c_in, 3, 3)).astype(np.float32))
b = ctx.upload(rng.standard_normal(c_out).astype(np.float32))
y = ctx.empty(n * c_out * h * w)
for _ in range(5):
    ctx.call("th_conv3x3_fwd", x, wt, b, y, n, c_in, h, w, c_out, 1, 0, 1)
ctx.sync()
NWG = 4096
buf = np.zeros(6 * NWG, np.int64)
lib.th_debug_conv_timeline(ctx.h, buf.ctypes.data_as(C.c_void_p), NWG)
t = buf[:4 * NWG].reshape(NWG, 4)
clk = buf[4 * NWG:].reshape(NWG, 2)
live = t[:, 2] > 0
t, clk = t[live], clk[live]
mhz = (clk[:, 1] - clk[:, 0]) / ((t[:, 2] - t[:, 0]) * 0.01)
print(f'shader clock over a workgroup life: mean {mhz.mean():.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f})')
t0 = t[:, 0].min()
start, kend, end = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01, (t[:, 2] - t0) * 0.01
hw = t[:, 3]
xcc, hwid = (hw >> 32) & 0xF, hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 0x7
place = xcc * 1000 + se * 100 + sh * 16 + cu
print(f"{len(t)} workgroups; launch span {end.max():.1f} us; life mean {np.mean(end - start):.1f} us (k loop {np.mean(kend - start):.1f}, epilogue {np.mean(end - kend):.1f})")
print("start time percentiles (us): " + ", ".join(f"p{p}={np.percentile(start, p):.1f}" for p in (1, 25, 50, 75, 99)))
print("distinct CUs used:", len(set(place.tolist())), " workgroups per CU (min/mean/max):", min(Counter(place.tolist()).values()),
      round(len(t) / len(set(place.tolist())), 2), max(Counter(place.tolist()).values()))
# concurrency on one CU over time
events = sorted([(s, 1) for s in start] + [(e, -1) for e in end])
cur, peak, area, last = 0, 0, 0.0, 0.0
for tt, d in events:
    area += cur * (tt - last); last = tt; cur += d; peak = max(peak, cur)
print(f"chip-wide resident workgroups: peak {peak}, time-average {area / end.max():.0f}")
one = place == place[0]
print("one CU's workgroups (start, k-loop end, end):", sorted((round(float(s), 1), round(float(k), 1), round(float(e), 1)) for s, k, e in zip(start[one], kend[one], end[one])))

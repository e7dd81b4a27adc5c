"""th_mlp2_xent, whole steps only (A/B of two builds of the library): python tools/mlp2_step_time.py [batch ...] -> us/step, eager launches and a
captured graph of 20 steps"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402
from taper_amd.hip import AdamFuse, RowSource  # noqa: E402

ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
inf, hid, c, n_rows = 784, 128, 10, 60000
data = ctx.upload((rng.integers(0, 256, (n_rows, inf))).astype(np.float32) / np.float32(255.0))
labels = ctx.upload(rng.integers(0, c, n_rows).astype(np.float32))
idx = ctx.upload(rng.permutation(n_rows).astype(np.int32))
dev = dict(w1=ctx.upload((rng.uniform(-1, 1, (hid, inf)) * np.sqrt(2.0 / inf)).astype(np.float32)), b1=ctx.zeros(hid),
           w2=ctx.upload(rng.uniform(-0.3, 0.3, (c, hid)).astype(np.float32)), b2=ctx.zeros(c))
mom = {k: (ctx.zeros(n), ctx.zeros(n)) for k, n in (("w1", hid * inf), ("b1", hid), ("w2", c * hid), ("b2", c))}
tick, dlr = ctx.upload(np.array([0, 0], np.int32)), ctx.upload(np.array([1e-3], np.float32))
fuses = [AdamFuse(int(dev[k]), int(mom[k][0]), int(mom[k][1]), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4) for k in ("w1", "b1", "w2", "b2")]
out = dict(dw1=ctx.empty(hid * inf), db1=ctx.empty(hid), dw2=ctx.empty(c * hid), db2=ctx.empty(c), loss=ctx.empty(1), nc=ctx.empty(1))
res = []
for batch in [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384, 60000]:
    state = ctx.upload(np.array([0, 0], np.int64))
    src = RowSource(int(data), int(labels), int(idx), state.offset(8), n_rows, n_rows)

    def step():
        ctx.call("th_mlp2_xent", C.byref(src), batch, inf, hid, c, dev["w1"], dev["b1"], dev["w2"], dev["b2"], out["dw1"], out["db1"], out["dw2"],
                 out["db2"], out["loss"], out["nc"], None, 0, None, 0, tick, *[C.byref(f) for f in fuses])
    for _ in range(20):
        step()
    ctx.sync()
    e0, e1 = hip.Event(), hip.Event()
    best = []
    for _ in range(3):
        ctx.record(e0)
        for _ in range(200):
            step()
        ctx.record(e1)
        ctx.sync()
        best.append(ctx.elapsed_ms(e0, e1) * 1e3 / 200)
    ctx.graph_begin()
    for _ in range(20):
        step()
    g = ctx.graph_end()
    for _ in range(3):
        ctx.graph_launch(g)
    ctx.sync()
    gb = []
    for _ in range(3):
        ctx.record(e0)
        for _ in range(10):
            ctx.graph_launch(g)
        ctx.record(e1)
        ctx.sync()
        gb.append(ctx.elapsed_ms(e0, e1) * 1e3 / 200)
    ctx.graph_destroy(g)
    res.append(f"b{batch}: eager {min(best):.1f} graph {min(gb):.1f}")
print("   ".join(res))

"""Times th_mlp2_xent (three launches) through the C ABI with HIP events: python tools/mlp2_time.py [batch ...]
Env: TAPER_MLP2_RT (32 | 64), TAPER_MLP2_KZ (K slices of launch 2)."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402
from taper_amd.hip import AdamFuse, RowSource  # noqa: E402

ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
inf, hid, c, n_rows = int(os.environ.get('INF', 784)), 128, 10, int(os.environ.get('NROWS', 60000))
data = ctx.upload((rng.integers(0, 256, (n_rows, inf))).astype(np.float32) / np.float32(255.0))
labels = ctx.upload(rng.integers(0, c, n_rows).astype(np.float32))
idx = ctx.upload(rng.permutation(n_rows).astype(np.int32))
w1 = rng.uniform(-1, 1, (hid, inf)).astype(np.float32) * np.float32(np.sqrt(2.0 / inf))
w2 = rng.uniform(-0.3, 0.3, (c, hid)).astype(np.float32)
dev = dict(w1=ctx.upload(w1), b1=ctx.zeros(hid), w2=ctx.upload(w2), b2=ctx.zeros(c))
mom = {k: (ctx.zeros(n), ctx.zeros(n)) for k, n in (("w1", hid * inf), ("b1", hid), ("w2", c * hid), ("b2", c))}
tick, dlr = ctx.upload(np.array([0, 0], np.int32)), ctx.upload(np.array([1e-3], np.float32))
fuses = [AdamFuse(int(dev[k]), int(mom[k][0]), int(mom[k][1]), int(tick), int(dlr), 0.9, 0.999, 1e-8, 1e-4) for k in ("w1", "b1", "w2", "b2")]
out = dict(dw1=ctx.empty(hid * inf), db1=ctx.empty(hid), dw2=ctx.empty(c * hid), db2=ctx.empty(c), loss=ctx.empty(1), nc=ctx.empty(1))
for batch in [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384]:
    state = ctx.upload(np.array([0, 0], np.int64))
    src = RowSource(int(data), int(labels), int(idx), state.offset(8), n_rows, n_rows)

    def step():
        ctx.call("th_mlp2_xent", C.byref(src), batch, inf, hid, c, dev["w1"], dev["b1"], dev["w2"], dev["b2"], out["dw1"], out["db1"], out["dw2"],
                 out["db2"], out["loss"], out["nc"], None, 0, None, 0, tick, *[C.byref(f) for f in fuses])
    for _ in range(20):
        step()
    ctx.sync()
    reps = 200
    e0, e1 = hip.Event(), hip.Event()
    ctx.record(e0)
    for _ in range(reps):
        step()
    ctx.record(e1)
    ctx.sync()
    us = ctx.elapsed_ms(e0, e1) * 1e3 / reps
    flops = 2.0 * batch * inf * hid * 2 + 2.0 * batch * hid * c * 3
    per = []
    for which in (1, 2, 3):                      # each launch alone (th_debug_mlp2_only)
        hip.hip.th_debug_mlp2_only(which)
        for _ in range(10):
            step()
        ctx.record(e0)
        for _ in range(reps):
            step()
        ctx.record(e1)
        ctx.sync()
        per.append(ctx.elapsed_ms(e0, e1) * 1e3 / reps)
    hip.hip.th_debug_mlp2_only(0)
    print(f"    rows {per[0]:.1f} us ({2.0 * batch * inf * hid / per[0] / 1e6 / 157.3:.3f})  dw1 {per[1]:.1f} us ({2.0 * batch * inf * hid / per[1] / 1e6 / 157.3:.3f})  finish {per[2]:.1f} us")
    print(f"batch {batch}: {us:.1f} us/step  {batch / us:.2f} M samples/s  {60000 / batch * us and 1e6 / (60000 / batch * us):.0f} epochs/s  {flops / us / 1e6:.1f} TF ({flops / us / 1e6 / 157.3:.3f} of peak)  loss {ctx.download(out['loss'], 1)[0]:.4f}")

#!/usr/bin/env python3
"""Per-launch cost of each kernel of the MLP step, measured as a chain of N
dependent launches replayed as one hipGraph (so it includes the ~1.55 us
dependent-launch boundary of this box, like the real step does)."""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402
from taper_amd.hip import AdamFuse  # noqa: E402




def chain_us(ctx, fn, n=100, reps=20):
    ctx.graph_begin()
    for _ in range(n):
        fn()
    g = ctx.graph_end()
    for _ in range(3):
        ctx.graph_launch(g)
    ctx.sync()
    e0, e1 = hip.Event(), hip.Event()
    ctx.record(e0)
    for _ in range(reps):
        ctx.graph_launch(g)
    ctx.record(e1)
    us = hip.Ctx.elapsed_ms(e0, e1) * 1e3 / (reps * n)
    ctx.graph_destroy(g)
    return us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    B, IN, HID, OUT = args.batch, 784, 128, 10
    ctx = hip.Ctx(0)
    rng = np.random.default_rng(0)
    f = lambda *s: ctx.upload(rng.uniform(-0.1, 0.1, s).astype(np.float32))
    x, w1, b1, h = f(B, IN), f(HID, IN), f(HID), ctx.empty(B * HID)
    w2, b2, y = f(OUT, HID), f(OUT), ctx.upload(rng.integers(0, OUT, B).astype(np.float32))
    dh, dw2, db2, dw1, db1 = ctx.empty(B * HID), ctx.empty(OUT * HID), ctx.empty(OUT), ctx.empty(HID * IN), ctx.empty(HID)
    loss, nc = ctx.empty(1), ctx.empty(1)
    m1, v1, mb1, vb1 = ctx.zeros(HID * IN), ctx.zeros(HID * IN), ctx.zeros(HID), ctx.zeros(HID)
    tick, lr = ctx.upload(np.array([1, 0], np.int32)), ctx.upload(np.array([1e-3], np.float32))
    wf = AdamFuse(int(w1), int(m1), int(v1), int(tick), int(lr), 0.9, 0.999, 1e-8, 1e-4)
    bf = AdamFuse(int(b1), int(mb1), int(vb1), int(tick), int(lr), 0.9, 0.999, 1e-8, 1e-4)
    imgs = f(4096, IN)
    labels = ctx.upload(rng.integers(0, OUT, 4096).astype(np.float32))
    idx = ctx.upload(rng.permutation(4096).astype(np.int32))
    xb, yb = ctx.empty(32 * B * IN), ctx.empty(32 * B)
    logits, logp, dun = ctx.empty(B * OUT), ctx.empty(B * OUT), ctx.empty(B * OUT)
    offs, has = ctx.upload(np.array([0, HID * IN, HID * IN + HID], np.int64)), ctx.upload(np.array([1, 1], np.int32))
    pflat, gflat, mflat, vflat = ctx.zeros(HID * IN + HID), ctx.zeros(HID * IN + HID), ctx.zeros(HID * IN + HID), ctx.zeros(HID * IN + HID)
    cases = {
        "boundary (fill 4 floats)": lambda: ctx.call("th_fill_f32", loss, 1.0, 4),
        "gather_batch x1": lambda: ctx.call("th_gather_batch", imgs, labels, idx, 4096, None, B, IN, xb, yb),
        "gather_batch x32 (per chunk)": lambda: ctx.call("th_gather_batch", imgs, labels, idx, 4096, None, 32 * B, IN, xb, yb),
        "linear_fwd L1 (B x784->128, relu)": lambda: ctx.call("th_linear_fwd", x, w1, b1, h, B, IN, HID, 1),
        "linear_fwd L2 (B x128->10)": lambda: ctx.call("th_linear_fwd", h, w2, b2, logits, B, HID, OUT, 0),
        "softmax_xent_fwd": lambda: ctx.call("th_softmax_xent_fwd", logits, y, B, OUT, logp, loss, None, nc, dun, None, 0, None, 0, None),
        "linear_xent_head (L2+xent+bwd)": lambda: ctx.call("th_linear_xent_head", h, w2, b2, y, B, HID, OUT, None, loss, nc, dh, dw2, db2,
                                                           None, 0, None, 0, None, None, None),
        "linear_bwd L2 (dX,dW,db)": lambda: ctx.call("th_linear_bwd", h, w2, dun, None, dh, dw2, db2, B, HID, OUT, 0),
        "linear_bwd L1 (dW,db, relu mask)": lambda: ctx.call("th_linear_bwd", x, None, dh, h, None, dw1, db1, B, IN, HID, 0),
        "linear_bwd_adam L1 (+Adam epilogue)": lambda: ctx.call("th_linear_bwd_adam", x, None, dh, h, None, dw1, db1, B, IN, HID, 0,
                                                                C.byref(wf), C.byref(bf)),
        "adam_step (100 480 params)": lambda: ctx.call("th_adam_step", pflat, gflat, mflat, vflat, offs, has, 2, HID * IN + HID, tick, lr,
                                                        0.9, 0.999, 1e-8, 1e-4, 1),
    }
    metrics, state = ctx.zeros(2 * 4096), ctx.upload(np.array([0, 0], np.int64))
    m2, v2, mb2, vb2 = ctx.zeros(OUT * HID), ctx.zeros(OUT * HID), ctx.zeros(OUT), ctx.zeros(OUT)
    wf2 = AdamFuse(int(w2), int(m2), int(v2), int(tick), int(lr), 0.9, 0.999, 1e-8, 1e-4)
    bf2 = AdamFuse(int(b2), int(mb2), int(vb2), int(tick), int(lr), 0.9, 0.999, 1e-8, 1e-4)
    cases["head + step log"] = lambda: ctx.call("th_linear_xent_head", h, w2, b2, y, B, HID, OUT, None, loss, nc, dh, dw2, db2,
                                                metrics, 4096, state, 1, None, None, None)
    cases["head + step log + tick"] = lambda: ctx.call("th_linear_xent_head", h, w2, b2, y, B, HID, OUT, None, loss, nc, dh, dw2, db2,
                                                       metrics, 4096, state, 1, tick, None, None)
    cases["head + step log + tick + fused Adam"] = lambda: ctx.call("th_linear_xent_head", h, w2, b2, y, B, HID, OUT, None, loss, nc, dh,
                                                                    dw2, db2, metrics, 4096, state, 1, tick, C.byref(wf2), C.byref(bf2))
    cases["head + fused Adam only"] = lambda: ctx.call("th_linear_xent_head", h, w2, b2, y, B, HID, OUT, None, loss, nc, dh,
                                                       dw2, db2, None, 0, None, 0, None, C.byref(wf2), C.byref(bf2))

    def step_like():
        cases["linear_fwd L1 (B x784->128, relu)"]()
        cases["linear_xent_head (L2+xent+bwd)"]()
        cases["linear_bwd_adam L1 (+Adam epilogue)"]()
    cases["step-like chain (L1 fwd, head, L1 bwd+Adam) /3 launches"] = step_like
    out = {}
    for name, fn in cases.items():
        out[name] = round(chain_us(ctx, fn), 3)
        print(f"{out[name]:8.3f} us  {name}", flush=True)
    print(json.dumps({"batch": B, "us_per_launch_in_graph_chain": out}))
    ctx.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A/B of the simple CNN's chain launch (th_conv_chain_head_fwd, BASELINE configs[2]'s front + classifier rows): the one-workgroup-per-CU
instance (244 registers) against the two-to-a-CU instance (128 registers, half-pass k loop: conv_chain_simple_kernel<.., LEAN>), selected by
TAPER_CHAIN_LEAN in a child process each; outputs compared bit for bit, launches timed with events.  usage: chain_lean_ab.py [n ...]"""
import ctypes as C
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def child(n, out):
    from taper_amd import hip
    from tests.test_gpu_chain_head import SIMPLE, ChainHead, _k_of, _model
    ctx = hip.Ctx(0)
    rng = np.random.default_rng(5)
    conv, w, b = _model(rng, 10)
    x = (rng.integers(0, 256, (n, 1, 28, 28)).astype(np.float32) / 255.0)
    y = rng.integers(0, 10, n).astype(np.float32)
    K, c_last, hw = _k_of(SIMPLE)
    bufs = [(ctx.upload(cw), ctx.upload(cb)) for cw, cb in conv]
    stages, ns = hip.conv_stages([(dw, db, c_out, post) for (dw, db), (_, c_out, post) in zip(bufs, SIMPLE)])
    ymap, dl, rs, cbp = ctx.empty(n * K), ctx.empty(n * 16), ctx.empty(n * 2), ctx.empty(n * c_last)
    head = ChainHead(int(ctx.upload(w)), int(ctx.upload(b)), int(ctx.upload(y)), 10, int(dl), int(rs), int(cbp), None)
    xd, sp = ctx.upload(x), C.cast(stages, C.c_void_p)
    call = lambda: ctx.call("th_conv_chain_head_fwd", xd, sp, ns, ymap, n, 1, 28, 28, C.byref(head))
    for _ in range(30):
        call()
    e0, e1 = hip.Event(), hip.Event()
    best = 1e9
    for _ in range(5):
        ctx.record(e0)
        for _ in range(100):
            call()
        ctx.record(e1)
        ctx.sync()
        best = min(best, hip.Ctx.elapsed_ms(e0, e1) * 1e3 / 100)
    np.savez(out, us=best, pooled=ctx.download(ymap, (n, K)), dl=ctx.download(dl, (n, 16)), rs=ctx.download(rs, (n, 2)), cbp=ctx.download(cbp, (n, c_last)))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 4096]
    flops = lambda n: 2.0 * n * (784 * 9 * 32 + 196 * 9 * 32 * 64)          # conv1 + conv2
    for n in sizes:
        res = {}
        for lean in (0, 1):
            out = f"/tmp/chain_lean_{n}_{lean}.npz"
            subprocess.check_call([sys.executable, __file__, "--child", str(n), out], env=dict(os.environ, TAPER_CHAIN_LEAN=str(lean)))
            res[lean] = np.load(out)
        same = all(np.array_equal(res[0][k], res[1][k]) for k in ("pooled", "dl", "rs", "cbp"))
        print(json.dumps(dict(n=n, us_one_per_cu=round(float(res[0]["us"]), 2), us_two_per_cu=round(float(res[1]["us"]), 2),
                              frac_one=round(flops(n) / float(res[0]["us"]) / 1e6 / 157.3, 3), frac_two=round(flops(n) / float(res[1]["us"]) / 1e6 / 157.3, 3),
                              bit_identical=bool(same))))

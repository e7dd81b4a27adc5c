#!/usr/bin/env python3
"""fp32 MFMA sgemm micro-benchmark (BASELINE configs[4]: 4096^3 Linear stack):
TFLOP/s of th_sgemm NN / NT / TN against the 157.3 TF fp32 matrix peak, timed
with HIP events on the ctx stream, uniform random [-1,1) operands.  Steady state: the default
200 repetitions per variant let the clocks settle (short runs read 10-15 % low)."""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip  # noqa: E402

PEAK = 157.3


def bench(ctx, ta, tb, m, n, k, reps, beta=0.0):
    rng = np.random.default_rng(0)
    a = ctx.upload(rng.uniform(-1, 1, m * k).astype(np.float32))
    b = ctx.upload(rng.uniform(-1, 1, k * n).astype(np.float32))
    c = ctx.zeros(m * n)
    for _ in range(3):
        ctx.call("th_sgemm", ta, tb, m, n, k, 1.0, a, b, beta, c)
    e0, e1 = hip.Event(), hip.Event()
    ctx.record(e0)
    for _ in range(reps):
        ctx.call("th_sgemm", ta, tb, m, n, k, 1.0, a, b, beta, c)
    ctx.record(e1)
    ms = hip.Ctx.elapsed_ms(e0, e1) / reps
    return ms, 2.0 * m * n * k / (ms * 1e-3) / 1e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="4096")
    # >= 100 back-to-back products: the chip needs ~100 ms of sustained MFMA load to reach its steady clocks
    # (4096^3 NT: 108 TF over 5 reps, 116 over 20, 125 over 100, 127 over 400)
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    ctx = hip.Ctx(0)
    rows = []
    for s in args.sizes.split(","):
        dims = [int(v) for v in s.split("x")]
        m, n, k = dims * 3 if len(dims) == 1 else dims
        for name, ta, tb, beta in [("NN", 0, 0, 0.0), ("NT", 0, 1, 0.0), ("TN", 1, 0, 1.0), ("TN(beta=0)", 1, 0, 0.0), ("NN(beta=1)", 0, 0, 1.0), ("TT", 1, 1, 0.0)]:
            ms, tf = bench(ctx, ta, tb, m, n, k, args.reps, beta)
            rows.append(dict(variant=name, m=m, n=n, k=k, beta=beta, ms=round(ms, 4), tflops=round(tf, 2), frac_of_peak=round(tf / PEAK, 4)))
            print(json.dumps(rows[-1]), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""run ONE sgemm shape a number of times (for rocprofv3 --pmc / --kernel-trace runs): gemm_mnk.py <ta> <tb> <m> <n> <k> [reps] [beta]"""
import sys
import time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip
ta, tb, m, n, k = (int(x) for x in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
beta = float(sys.argv[7]) if len(sys.argv) > 7 else 0.0
ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
a, b, c = ctx.upload(rng.uniform(-1, 1, m * k).astype(np.float32)), ctx.upload(rng.uniform(-1, 1, k * n).astype(np.float32)), ctx.zeros(m * n)
for _ in range(3):
    ctx.call("th_sgemm", ta, tb, m, n, k, 1.0, a, b, beta, c)
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.call("th_sgemm", ta, tb, m, n, k, 1.0, a, b, beta, c)
ctx.sync()
us = (time.perf_counter() - t0) / reps * 1e6
print(f"th_sgemm ta={ta} tb={tb} m={m} n={n} k={k} beta={beta}: {us:.1f} us per launch, {2.0 * m * n * k / us / 1e6:.1f} TFLOP/s = {2.0 * m * n * k / us / 1e6 / 157.3:.3f} of the fp32 MFMA peak")

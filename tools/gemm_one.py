#!/usr/bin/env python3
"""run ONE sgemm variant a few times (for rocprofv3 --pmc runs): gemm_one.py <ta> <tb> [n]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taper_amd import hip
ta, tb = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
ctx = hip.Ctx(0)
rng = np.random.default_rng(0)
a, b, c = ctx.upload(rng.uniform(-1, 1, n * n).astype(np.float32)), ctx.upload(rng.uniform(-1, 1, n * n).astype(np.float32)), ctx.zeros(n * n)
for _ in range(5):
    ctx.call("th_sgemm", ta, tb, n, n, n, 1.0, a, b, 0.0, c)
ctx.sync()

#!/usr/bin/env python3
"""In-kernel phase timing of th_mlp_tail behind th_linear_fwd_ex (needs a PROFILE=1 build:
make -C taper_amd/csrc clean all PROFILE=1).  wall_clock64 ticks at 100 MHz."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from taper_amd import hip  # noqa: E402
from taper_amd._lib import hip as lib  # noqa: E402

ctx = hip.Ctx(0)
lib.th_debug_tail_prof.argtypes = [C.c_void_p, C.c_void_p]
lib.th_debug_tail_mark.argtypes = [C.c_void_p, C.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sk = bench.StepKernels(ctx, B)
names = ["mark -> entry", "entry -> loads issued", "step-size math", "wait for loads + zeroing", "logits + softmax", "dH + dW1 MFMA (head: db1, dW2, db2)",
         "LDS reduce + barrier", "sum + Adam + stores issued", "exit -> next kernel start"]
acc = np.zeros((2, 9))
N = 50
for it in range(N + 5):
    sk._k1()
    lib.th_debug_tail_mark(ctx.h, 14)
    sk._k2()
    lib.th_debug_tail_mark(ctx.h, 15)
    out = (C.c_longlong * 32)()
    lib.th_debug_tail_prof(ctx.h, out)
    for r in range(2):
        ts = [out[16 * r + 14]] + [out[16 * r + i] for i in range(8)] + [out[16 * r + 15]]
        if it >= 5:
            acc[r] += np.diff(ts) * 0.01
for r, role in enumerate(["lead head workgroup (block 0)", "first dW1 workgroup"]):
    print(role)
    for n, v in zip(names, acc[r] / N):
        print(f"  {v:7.3f} us  {n}")
    print(f"  {acc[r].sum() / N:7.3f} us  mark to mark")

# the same inside a hipGraph replay (kernel arguments live in device memory there): stamps of the LAST tail launch
g = sk._capture([sk._k1, sk._k2], steps=8)
acc = np.zeros((2, 7))
for it in range(N):
    ctx.graph_launch(g)
    out = (C.c_longlong * 32)()
    lib.th_debug_tail_prof(ctx.h, out)
    for r in range(2):
        acc[r] += np.diff([out[16 * r + i] for i in range(8)]) * 0.01
print("inside a graph replay (last of 8 steps)")
for r, role in enumerate(["lead head workgroup (block 0)", "first dW1 workgroup"]):
    print(role)
    for n, v in zip(names[1:8], acc[r] / N):
        print(f"  {v:7.3f} us  {n}")
    print(f"  {acc[r].sum() / N:7.3f} us  entry to exit")

#!/bin/bash
# Runs on the GPU box: the shader clock the 4096^3 sgemm actually runs at (TH_PROFILE stamps of workgroup 0: clock64 / wall_clock64),
# after N back-to-back products.  usage: tools/prof_gemm_clock.sh
set -e
cd $GRAFT_REPO_ROOT/taper_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off -DTH_PROFILE -c gemm.hip -o /tmp/gemm_prof.o
OBJS=$(ls _build/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtaper_hip.so $OBJS /tmp/gemm_prof.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
cd $GRAFT_REPO_ROOT
python - <<'PY'
import ctypes as C, numpy as np
from taper_amd import hip
from taper_amd._lib import hip as lib
lib.th_debug_gemm_prof.argtypes = [C.c_void_p, C.c_void_p]; lib.th_debug_gemm_prof.restype = C.c_int
ctx = hip.Ctx(0)
n = 4096
rng = np.random.default_rng(0)
a, b, c = ctx.upload(rng.uniform(-1, 1, n * n).astype(np.float32)), ctx.upload(rng.uniform(-1, 1, n * n).astype(np.float32)), ctx.zeros(n * n)
for name, ta, tb in (("NN", 0, 0), ("NT", 0, 1), ("TN", 1, 0)):
    for reps in (1, 20, 200):
        e0, e1 = hip.Event(), hip.Event()
        ctx.record(e0)
        for _ in range(reps):
            ctx.call("th_sgemm", ta, tb, n, n, n, 1.0, a, b, 0.0, c)
        ctx.record(e1)
        ms = hip.Ctx.elapsed_ms(e0, e1) / reps
        out = (C.c_longlong * 4)(); lib.th_debug_gemm_prof(ctx.h, out)
        ghz = (out[3] - out[1]) / ((out[2] - out[0]) * 10.0)
        tf = 2.0 * n ** 3 / (ms * 1e-3) / 1e12
        print(f"{name} x{reps}: {ms*1e3:.0f} us/product {tf:.1f} TF; workgroup 0 of the last product lived {(out[2]-out[0])*0.01:.0f} us at {ghz:.2f} GHz -> fp32 MFMA peak at that clock {256*4*512*ghz/1e3:.1f} TF, achieved {tf/(256*4*512*ghz/1e3):.3f} of it")
PY

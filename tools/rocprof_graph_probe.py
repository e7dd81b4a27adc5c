"""Which combination crashes rocprofv3 --kernel-trace: graph replays with events / without / backlog depth."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import StepKernels
from taper_amd import hip

variant = sys.argv[1]
ctx = hip.Ctx(0)
sk = StepKernels(ctx, 64)
ks = [sk._k1, sk._k2]
if variant == "k1only":
    ks = [sk._k1]
if variant == "k2only":
    ks = [sk._k2]
if variant == "k3only":
    ks = [sk._k3]
ctx.graph_begin()
for _ in range(16):
    for k in ks:
        k()
g = ctx.graph_end()
for _ in range(3):
    ctx.graph_launch(g)
ctx.sync()
print(variant, "warm ok", flush=True)
if variant == "noevents":
    t0 = time.perf_counter()
    for _ in range(12):
        ctx.graph_launch(g)
    ctx.sync()
    print("noevents ok", (time.perf_counter() - t0) * 1e6 / (12 * 16), flush=True)
elif variant == "sync_each":
    for _ in range(12):
        ctx.graph_launch(g)
        ctx.sync()
    print("sync_each ok", flush=True)
else:
    e0, e1 = hip.Event(), hip.Event()
    ctx.record(e0)
    for _ in range(12):
        ctx.graph_launch(g)
    ctx.record(e1)
    print(variant, "events ok", hip.Ctx.elapsed_ms(e0, e1) * 1e3 / (12 * 16), flush=True)
ctx.graph_destroy(g)
ctx.close()

import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import taper_amd as T
rng = np.random.default_rng(0)
n = 60000
x = rng.integers(0, 256, (n, 784)).astype(np.float32) / 255.0
y = rng.integers(0, 10, n).astype(np.float32)
model = T.Sequential([T.Linear(784, 128, True, seed=1), T.ReLU(), T.Linear(128, 10, True, seed=2)])
opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
tr = T.Trainer(model, opt)
loader = T.DataLoader(T.MNISTDataset.from_host(x, y), 64, False)
for _ in range(3):
    tr.run_epoch(loader, T.Trainer.GRAPH, max_steps=300)
for _ in range(30):
    tr.run_epoch(loader, T.Trainer.GRAPH, max_steps=20)
ts = []
for _ in range(20):
    T.Device.sync()
    t0 = time.perf_counter()
    tr.run_epoch(loader, T.Trainer.GRAPH, max_steps=20)
    T.Device.sync()
    ts.append((time.perf_counter() - t0) * 1e6)
print("python-timed 20-step call: median %.1f us, min %.1f us" % (np.median(ts), min(ts)))
os.environ["TAPER_TRACE_EPOCH"] = "1"

"""debug probe: where do the plain loop and Adam.fused_step() differ (tests/test_gpu_fused_scope.py)"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import taper_amd as T
from tests import backends
from tests.test_gpu_fused_scope import _spec

batch, dims = 1024, (1024, 1024, 1024, 10)
H = backends.get("hip")
rng = np.random.default_rng(batch + dims[0])
spec = _spec(rng, dims)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
xs = [rng.uniform(0, 1, (batch, dims[0])).astype(np.float32) for _ in range(steps)]
ys = [rng.integers(0, dims[-1], batch).astype(np.float32) for _ in range(steps)]


def run(fused):
    model = H.sequential(spec)
    opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
    grads = None
    for x, y in zip(xs, ys):
        T.Tape.reset(); opt.zero_grad()
        if fused:
            with opt.fused_step():
                loss = T.cross_entropy_loss(model.forward(T.Tensor(x)), T.Tensor(y)); loss.backward()
                grads = [p.grad().copy() if p.grad() is not None else None for p in model.parameters()]
                opt.step()
        else:
            loss = T.cross_entropy_loss(model.forward(T.Tensor(x)), T.Tensor(y)); loss.backward()
            grads = [p.grad().copy() if p.grad() is not None else None for p in model.parameters()]
            opt.step()
    T.Tape.reset()
    m, v = opt.moments()
    return [p.data().copy() for p in model.parameters()], m, v, grads


pa, ma, va, ga = run(False)
pb, mb, vb, gb = run(True)
for i, (a, b) in enumerate(zip(pa, pb)):
    print("param", i, a.shape, "differ", int((a != b).sum()), "max", float(np.abs(a - b).max()))
for i, (a, b) in enumerate(zip(ga, gb)):
    if a is not None and b is not None:
        print("grad", i, "differ", int((a != b).sum()))
print("m differ", int((ma != mb).sum()), "v differ", int((va != vb).sum()))

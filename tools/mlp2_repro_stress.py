#!/usr/bin/env python3
"""Run-to-run reproducibility of the Trainer's large-batch MLP step (th_mlp2_xent): the same 4-step, two-epoch optimisation R times in one
process; every run must give bit-identical losses and weights (the step has no atomics on its data path and adds in fixed orders).  A run
that differs points at the k split's hand-off between workgroups (mlp2_rows_kernel, batch < 4096) -- the only place where one workgroup
reads what another wrote inside a launch.  usage: mlp2_repro_stress.py [batch] [runs]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import taper_amd as T  # noqa: E402
from tests import backends  # noqa: E402
from tests.dp_worker import make_problem  # noqa: E402


def one_run(steps, batch, mode):
    spec, x, y = make_problem(steps, batch, model="mlp_baseline")
    H = backends.get("hip")
    model = H.sequential(spec)
    opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
    tr = T.Trainer(model, opt)
    loader = T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False)
    losses = np.concatenate([tr.run_epoch(loader, mode)["losses"] for _ in range(2)])
    return losses, [p.data() for p in model.parameters()]


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    for mode_name in ("GRAPH", "EAGER"):
        mode = getattr(T.Trainer, mode_name)
        ref = one_run(4, batch, mode)
        bad = 0
        for r in range(runs):
            got = one_run(4, batch, mode)
            same = np.array_equal(got[0], ref[0]) and all(np.array_equal(a, b) for a, b in zip(got[1], ref[1]))
            if not same:
                bad += 1
                d = np.abs(got[0] - ref[0])
                print(f"  {mode_name} run {r}: losses differ at steps {np.nonzero(d)[0].tolist()} (max {d.max():.3e}); "
                      f"weights max diff {max(float(np.abs(a - b).max()) for a, b in zip(got[1], ref[1])):.3e}")
        print(f"batch {batch} {mode_name}: {bad} of {runs} runs differ from the first")


if __name__ == "__main__":
    main()

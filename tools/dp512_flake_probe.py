#!/usr/bin/env python3
"""Repeat tests/test_gpu_dp.py::test_p2p_two_ranks_of_512_rows_take_the_three_launch_step's run R times and say, per run, whose numbers moved:
each rank's per-step losses against the first run's, the replicas against each other.  usage: dp512_flake_probe.py [runs] [steps] [mode]"""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests.test_gpu_dp import _run_ranks  # noqa: E402
from tests.dp_worker import make_problem  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mode = sys.argv[3] if len(sys.argv) > 3 else "graph"
spec, _, _ = make_problem(steps, 1024)
init = [np.asarray(l[k], np.float32).ravel() for l in spec if "w" in l for k in ("w", "b")]
all_runs = []
for i in range(runs):
    with tempfile.TemporaryDirectory() as d:
        ranks = _run_ranks(Path(d), 2, "p2p", mode, steps=steps, global_batch=1024, same_device=True, fuse=os.environ.get("PROBE_FUSE", "1") == "1", fine=os.environ.get("PROBE_FINE", "0") == "1")
        all_runs.append([dict(losses=np.array(r["losses"]), p=[np.array(r[f"p{k}"]).ravel() for k in range(4)]) for r in ranks])
# the reference is the MOST COMMON outcome (a wrong run is wrong in its own way)
sig = [tuple(np.concatenate([c["losses"] for c in run]).tolist()) for run in all_runs]
best = max(set(sig), key=sig.count)
ref = all_runs[sig.index(best)]
bad = 0
for i, cur in enumerate(all_runs):
    msgs = []
    for r in range(2):
        d = np.abs(cur[r]["losses"] - ref[r]["losses"])
        if d.max() > 0:
            msgs.append(f"rank {r} losses differ at steps {np.nonzero(d)[0].tolist()}")
    for k in range(4):
        dp = cur[0]["p"][k] - ref[0]["p"][k]
        if np.abs(dp).max() > 0:
            nz = np.abs(dp) > 0
            msgs.append(f"param {k}: {int(nz.sum())}/{dp.size} differ, max {np.abs(dp).max():.2e}")
    same_rep = all(np.array_equal(cur[0]["p"][k], cur[1]["p"][k]) for k in range(4))
    if msgs:
        bad += 1
        print(f"run {i}: replicas identical: {same_rep};", "; ".join(msgs))
print(f"{mode}, {steps} step(s) per epoch: {bad} of {runs} runs differ from the most common outcome ({sig.count(best)} runs)")

#!/usr/bin/env python3
"""Per-kernel share of the MLP step, measured IN SITU: the step's two launches
(layer-1 forward + carried updates + tick, fused head + layer-1 backward + Adam) are replayed as hipGraph chains with and without
each launch (bench.StepKernels.measure), so every kernel keeps the data flow of
the real step (operands freshly written by the previous launch)."""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import StepKernels  # noqa: E402
from taper_amd import hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    ctx = hip.Ctx(0)
    sk = StepKernels(ctx, args.batch)
    print(json.dumps(sk.measure(), indent=1))
    ctx.close()


if __name__ == "__main__":
    main()

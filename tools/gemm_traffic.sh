#!/bin/bash
# Runs on the GPU box: fabric-side bytes per launch of th_sgemm's kernels for one shape (FETCH_SIZE / WRITE_SIZE in passes of their own; KiB; gfx950:
# FETCH_SIZE x 2 -- MI355X_MICROARCH.md, HBM section), beside the un-profiled time.   usage: gemm_traffic.sh ta tb m n k [ENV=VAL ...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
shape="$1 $2 $3 $4 $5"; shift 5
for e in "$@"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
echo "## th_sgemm (ta tb m n k) = $shape $*"
python $ROOT/tools/gemm_mnk.py $shape 150
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/gt_$c
    timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d /tmp/gt_$c -- python $ROOT/tools/gemm_mnk.py $shape 10 > /dev/null 2>&1
done
python - <<'PY'
import csv
from collections import defaultdict
from pathlib import Path
acc = defaultdict(lambda: defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in Path("/tmp/gt_" + c).rglob("*counter_collection.csv"):
        for row in csv.DictReader(open(f, newline="")):
            if "sgemm" in row["Kernel_Name"] or "splitk" in row["Kernel_Name"]:
                acc[row["Kernel_Name"].split("(")[0][:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(f"{k}: read {2 * m.get('FETCH_SIZE', 0) * 1024 / 1e6:.1f} MB, written {m.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB per launch ({len(next(iter(c.values())))} dispatches)")
PY

#!/bin/bash
# GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of th_mlp2_xent's three launches at BATCH (default 16384).  Out: gpurun_out/mlp2_pmc/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/mlp2_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python $ROOT/tools/mlp2_time.py ${BATCH:-16384} > "$OUT/pmc_$c.txt" 2> "$OUT/pmc_$c.err"
    python $ROOT/tools/summarize_pmc.py "$OUT/pmc_$c" $c > "$OUT/pmc_$c.summary.csv"
    rm -rf "$OUT/pmc_$c"
done
cat "$OUT"/pmc_*.summary.csv | grep -i 'mlp2\|name'

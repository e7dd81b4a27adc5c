#!/bin/bash
# Runs on the GPU box: BASELINE configs[4] (4 x Linear(4096,4096)+ReLU + classifier, batch 4096) -- the bench line,
# the rocprofv3 kernel-trace stats of the same command and three separate PMC passes (MFMA busy cycles, FETCH_SIZE,
# WRITE_SIZE).  Outputs in gpurun_out/linear_stack/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/linear_stack
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_linear_stack.py"
{ timeout -s KILL 300 $CMD --steps 30 --warmup 8; timeout -s KILL 300 $CMD --steps 30 --warmup 8; timeout -s KILL 300 $CMD --steps 30 --warmup 8 --no-adam; } > "$OUT/bench.jsonl"
cat "$OUT/bench.jsonl" | cut -c1-220
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ls_trace -- $CMD --steps 12 --warmup 4 > /dev/null 2>&1
python $ROOT/tools/kstats.py /tmp/ls_trace/*/*kernel_stats.csv | head -16 > "$OUT/kernel_stats.txt"
rm -rf /tmp/ls_trace
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" FETCH_SIZE WRITE_SIZE; do
    tag=${c%% *}
    timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d /tmp/ls_pmc_$tag -- $CMD --steps 3 --warmup 1 > /dev/null 2>&1
    for one in $c; do python $ROOT/tools/summarize_pmc.py /tmp/ls_pmc_$tag $one | head -8; done > "$OUT/pmc_$tag.csv"
    rm -rf /tmp/ls_pmc_$tag
done
cat "$OUT/kernel_stats.txt" "$OUT"/pmc_*.csv

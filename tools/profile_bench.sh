#!/bin/bash
# Runs on the GPU box (gpurun): the default bench line, the rocprofv3 kernel-trace
# stats of the SAME command, and two separate PMC passes (FETCH_SIZE / WRITE_SIZE:
# they do not fit one pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Outputs land in gpurun_out/profile_bench/; copy what is to be kept into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profile_bench
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --gpus 1 ${BENCH_ARGS:-}"

timeout -s KILL 400 $BENCH > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 400 "$OUT/bench.json"
cp "$ROOT/gpurun_out/bench_details.json" "$OUT/bench_details.json" 2>/dev/null   # (the later passes overwrite it)

timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/trace"          # the raw trace is large; the stats summary is what is kept

for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- \
        python $ROOT/bench.py --gpus 1 --steps 192 --warmup 32 --no-cpu-baseline --no-sweep --workloads none > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err"
    python $ROOT/tools/summarize_pmc.py "$OUT/pmc_$c" $c > "$OUT/pmc_$c.summary.csv"
    rm -rf "$OUT/pmc_$c"
done
cat "$OUT/kernel_stats.csv" "$OUT"/pmc_*.summary.csv

#!/bin/bash
# Runs on the GPU box: VERDICT r01 item 6a -- the MLP batch-64 layer-1 kernel (sgemm_small16_tick) with and without the
# 2 x 2 tile-block-per-XCD map: step time (K=20 and 4000 steps), kernel-trace averages, FETCH_SIZE per launch.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/k1_xcd
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --gpus 1 --no-cpu-baseline --workloads none"
for m in 1 0 1 0; do
    for a in "--steps 20 --warmup 5" "--steps 4000 --warmup 200"; do
        echo "map=$m $a: $(TAPER_K1_XCD_MAP=$m timeout -s KILL 200 $B $a 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"]*1000, "us", d["roofline"].get("kernel"), d["roofline"].get("us_per_launch"))')"
    done
done
for m in 1 0; do
    TAPER_K1_XCD_MAP=$m timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace$m" -- $B --steps 2000 --warmup 100 > /dev/null 2> "$OUT/trace$m.err"
    find "$OUT/trace$m" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_map$m.csv" \;
    rm -rf "$OUT/trace$m"
    echo "== map=$m kernel stats"; head -5 "$OUT/kernel_stats_map$m.csv"
    TAPER_K1_XCD_MAP=$m timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc$m" -- $B --steps 192 --warmup 32 > /dev/null 2> "$OUT/pmc$m.err"
    python $ROOT/tools/summarize_pmc.py "$OUT/pmc$m" FETCH_SIZE > "$OUT/pmc_fetch_map$m.csv"
    rm -rf "$OUT/pmc$m"
    echo "== map=$m FETCH_SIZE"; cat "$OUT/pmc_fetch_map$m.csv"
done

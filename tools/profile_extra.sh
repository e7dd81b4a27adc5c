#!/bin/bash
# Runs on the GPU box: per-kernel rocprofv3 stats of the non-default workloads (CNNs, the MLP at large
# batch) and the GEMM shapes behind them.  Outputs in gpurun_out/profile_extra/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profile_extra
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args...
    local name=$1; shift
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$name -- \
        python $ROOT/bench.py --no-cpu-baseline --no-roofline --no-sweep --steps 96 --warmup 32 "$@" 2> /dev/null | tail -1 | cut -c1-200
    python $ROOT/tools/kstats.py /tmp/pe_$name/*/*kernel_stats.csv | head -20
    rm -rf /tmp/pe_$name
}
{
    for w in cnn_reference_b256 cnn_simple_b256; do echo "### $w"; stats $w --workload $w; done
} > "$OUT/cnn_kernel_stats.txt"
{
    for b in 4096 16384 60000; do echo "### mlp_784-128-10 batch $b"; stats b$b --batch $b; done
} > "$OUT/mlp_large_batch_kernel_stats.txt"
timeout -s KILL 300 python $ROOT/tools/bench_gemm.py --sizes 4096,16384x128x784,60000x128x784,128x784x16384,128x784x60000 --reps 10 \
    > "$OUT/gemm_mlp_shapes.txt" 2>&1
for w in cnn_reference_b256 cnn_simple_b256 mlp_784-128-64-10_b256; do
    timeout -s KILL 300 python $ROOT/bench.py --no-roofline --no-sweep --workload $w --steps 400 --warmup 40 2> /dev/null | tail -1
done > "$OUT/bench_other_workloads.jsonl"
tail -3 "$OUT/bench_other_workloads.jsonl" | cut -c1-400

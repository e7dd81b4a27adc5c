#!/bin/bash
# GPU box: th_mlp2_xent timing (HIP events) + rocprofv3 kernel stats per batch.  Out: gpurun_out/mlp2/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/mlp2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python $ROOT/tools/mlp2_time.py ${BATCHES:-1024 4096 16384} | tee "$OUT/time${TAG:-}.txt"
for B in ${BATCHES:-1024 4096 16384}; do
    timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$B" -- python $ROOT/tools/mlp2_time.py $B > /dev/null 2> "$OUT/trace_$B.err"
    find "$OUT/trace_$B" -name "*kernel_stats.csv" -exec cp {} "$OUT/b${B}${TAG:-}_kernel_stats.csv" \;
    rm -rf "$OUT/trace_$B"
    echo "== batch $B"; head -5 "$OUT/b${B}${TAG:-}_kernel_stats.csv" | cut -c1-60,150-260
done

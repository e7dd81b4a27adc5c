#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of the MLP 784-128-10 step at the SURVEY 8(d) sweep's
# mid / large batches (one bench.py run per batch; the CNN / GEMM workloads and the CPU leg are off).
# Outputs: gpurun_out/mlp_batches/b<B>_kernel_stats.csv + b<B>.json (the bench line).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/mlp_batches
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for B in ${BATCHES:-1024 4096 16384}; do
    ARGS="--gpus 1 --batch $B --steps ${STEPS:-400} --warmup 40 --workloads none --no-sweep --no-cpu-baseline ${EXTRA:-}"
    timeout -s KILL 200 python $ROOT/bench.py $ARGS > "$OUT/b$B.json" 2> "$OUT/b$B.err"
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$B" -- python $ROOT/bench.py $ARGS > /dev/null 2> "$OUT/trace_$B.err"
    find "$OUT/trace_$B" -name "*kernel_stats.csv" -exec cp {} "$OUT/b${B}_kernel_stats.csv" \;
    rm -rf "$OUT/trace_$B"
    echo "== batch $B"; python -c "import json,sys; d=json.loads(open('$OUT/b$B.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
    head -14 "$OUT/b${B}_kernel_stats.csv"
done

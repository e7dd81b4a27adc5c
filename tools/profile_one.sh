#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of ONE bench workload (graph replay): tools/profile_one.sh <workload> [extra bench flags]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
w=$1; shift
OUT=$ROOT/gpurun_out/profile_one
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc_$w
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$w -- \
    python $ROOT/bench.py --no-cpu-baseline --no-roofline --no-sweep --workloads none --steps 470 --warmup 260 --workload $w "$@" 2> /dev/null | tail -1 | cut -c1-260 > "$OUT/$w.json"
{ echo "### $w (470 timed + 260 warm-up steps, graph replay)"; cat "$OUT/$w.json"; echo; python $ROOT/tools/kstats.py /tmp/pc_$w/*/*kernel_stats.csv | head -24; } > "$OUT/$w.txt"
rm -rf /tmp/pc_$w
cat "$OUT/$w.txt"

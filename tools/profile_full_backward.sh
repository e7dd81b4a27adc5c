#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of the CNN workloads with --full-backward (every conv weight trains: the extension
# beyond the reference's cut tape, quirk Q2), eager enqueue (TAPER_NO_GRAPH=1: rocprofv3 on ROCm 7.2 crashes in a hipGraphLaunch that follows a hipGraphExecDestroy); outputs in gpurun_out/profile_fb/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profile_fb
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in ${WORKLOADS:-cnn_reference_b256 cnn_simple_b256}; do
    rm -rf /tmp/fb_$w
    TAPER_NO_GRAPH=1 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fb_$w -- \
        python $ROOT/bench.py --full-backward --no-cpu-baseline --no-roofline --no-sweep --workloads none --steps 100 --warmup 20 --workload $w 2> "$OUT/$w.err" | tail -1 | cut -c1-400 > "$OUT/$w.json"
    { echo "### $w --full-backward (100 timed + 20 warm-up steps; per-step time = sum over kernels of calls x avg / steps)"; cat "$OUT/$w.json"; echo; python $ROOT/tools/kstats.py /tmp/fb_$w/*/*kernel_stats.csv | head -40; } > "$OUT/$w.txt"
    rm -rf /tmp/fb_$w
done
cat "$OUT"/*.txt

set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for b in 1024 4096; do
rm -rf /tmp/pc_s$b
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_s$b -- python $ROOT/bench.py --no-cpu-baseline --no-roofline --no-sweep --workloads none --steps 200 --warmup 260 --workload cnn_simple_b256 --batch $b 2> /dev/null | tail -1 | cut -c1-200
python $ROOT/tools/kstats.py /tmp/pc_s$b/*/*kernel_stats.csv | head -8
done

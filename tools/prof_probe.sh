cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "TAPER_NO_GRAPH=0" "TAPER_NO_GRAPH=0 TAPER_GRAPH_LADDER=4"; do
  echo "== $cfg"
  env $cfg timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python $R/bench.py --steps 192 --warmup 32 --no-cpu-baseline --workloads none > /tmp/pp.json 2> /tmp/pp.err; echo rc=$?
  tail -c 200 /tmp/pp.json; echo
  python $R/tools/kstats.py /tmp/pp/*/*kernel_stats.csv 2>/dev/null | head -6
  rm -rf /tmp/pp
done

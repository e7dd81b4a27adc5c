cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/mlp2x; mkdir -p $OUT
run() { # tag, env...
  tag=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$tag -- python $R/tools/mlp2_time.py ${BATCH:-16384} > $OUT/$tag.txt 2>/dev/null
  find $OUT/t_$tag -name "*kernel_stats.csv" -exec cp {} $OUT/$tag.csv \; ; rm -rf $OUT/t_$tag
  echo "== $tag: $(cat $OUT/$tag.txt)"; $R/tools/kstats.sh $OUT/$tag.csv | grep mlp2
}
run base TAPER_MLP2_DW=22
run inf1568 TAPER_MLP2_DW=22 INF=1568 NROWS=30000
run inf392 TAPER_MLP2_DW=22 INF=392

cd /tmp
R=$GRAFT_REPO_ROOT
for ns in 4 6; do echo "== NS=$ns"; TAPER_MLP2_NS=$ns python $R/tools/mlp2_time.py 1024 2048 4096 8192; done

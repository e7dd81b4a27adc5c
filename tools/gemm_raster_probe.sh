cd $GRAFT_REPO_ROOT
for r in 8 0 4 16; do echo "raster=$r"; TAPER_GEMM_RASTER=$r python tools/bench_gemm.py --sizes 4096 --reps 120 2>&1 | grep -E '"NN"|"NT"|"TN"' | cut -c1-140; done

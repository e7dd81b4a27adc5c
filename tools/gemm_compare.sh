cd $GRAFT_REPO_ROOT
echo "in-kernel transposing stage:"; python tools/bench_gemm.py --sizes 4096 --reps 120
echo "separate transpose launch:"; TAPER_GEMM_PRETRANSPOSE=1 python tools/bench_gemm.py --sizes 4096 --reps 120
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "sgemm or linear" 2>&1 | tail -3

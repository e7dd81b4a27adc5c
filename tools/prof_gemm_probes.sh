#!/bin/bash
# Runs on the GPU box: 4096^3 sgemm NT with parts of the tile loop removed (WRONG results, timing only): what the loop waits for.
# usage: tools/prof_gemm_probes.sh ["-DPROBE1 -DPROBE2" ...]   (default: the r03 set)
set -e
cd $GRAFT_REPO_ROOT/taper_amd/csrc
OBJS=$(ls _build/*.o | grep -v "/gemm.o")
if [ $# -eq 0 ]; then set -- "" "-DGEMM_PROBE_NOLOAD" "-DGEMM_PROBE_NOSTORE" "-DGEMM_PROBE_NOLOAD -DGEMM_PROBE_NOSTORE" "-DGEMM_PROBE_NOLOAD -DGEMM_PROBE_NOSTORE -DGEMM_PROBE_NOSYNC -DGEMM_PROBE_NOREAD"; fi
for probe in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-unused-variable -ffp-contract=off $probe -c gemm.hip -o /tmp/gemm_probe.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtaper_hip.so $OBJS /tmp/gemm_probe.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
    echo "probe: ${probe:-none}"
    (cd $GRAFT_REPO_ROOT && python tools/bench_gemm.py --sizes 4096 --reps 100 2>&1 | grep -E '"NT"|"NN"|"TN"' | cut -c1-130)
done

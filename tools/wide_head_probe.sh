#!/bin/bash
# Runs on the GPU box: the simple CNN's step with the wide classifier head built for 1 / 2 / 4 column tiles per workgroup (-DTH_WH_TX)
set -e
cd $GRAFT_REPO_ROOT/taper_amd/csrc
for tx in 2 1 4; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off -DTH_WH_TX=$tx -c wide_head.hip -o /tmp/wide_head_p.o
  OBJS=$(ls _build/*.o | grep -v wide_head.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtaper_hip.so $OBJS /tmp/wide_head_p.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  echo "== WH_TX=$tx"
  (cd $GRAFT_REPO_ROOT && timeout 200 python bench.py --no-cpu-baseline --workloads cnn_simple_b256 --steps 200 --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
for w in d['workloads']: print(w.get('workload'), w.get('ms_per_step'))")
  (cd $GRAFT_REPO_ROOT && timeout 200 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k wide 2>&1 | tail -1)
done

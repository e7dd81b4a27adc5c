#!/bin/bash
# Runs on the GPU box: A/B of conv_chain.hip builds (extra -D flags per variant, ';'-separated in VARIANTS) on tools/bench_conv.py
# usage: VARIANTS="-DTH_LC_STAGE_ALL=1;-DTH_LC_NT=1" tools/conv_layer_ab.sh
set -e
cd $GRAFT_REPO_ROOT/taper_amd/csrc
cp ../lib/libtaper_hip.so /tmp/lib_default.so
OBJS=$(ls _build/*.o | grep -v conv_chain.o)
IFS=';' read -ra VS <<< "${VARIANTS:-}"
run() { cd $GRAFT_REPO_ROOT; for i in 1 2; do python tools/bench_conv.py 200 | grep -E "32-> 64 14|64->128  7|64-> 64 14|32-> 32 28"; done; cd $GRAFT_REPO_ROOT/taper_amd/csrc; }
echo "== default"; run
for v in "${VS[@]}"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off $v -c conv_chain.hip -o /tmp/conv_chain_v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtaper_hip.so $OBJS /tmp/conv_chain_v.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
    echo "== $v"; run
done
cp /tmp/lib_default.so ../lib/libtaper_hip.so

#!/bin/bash
# GPU box: tools/chain_mlp3_time.py on builds of conv_chain.hip with extra -D flags (';'-separated in VARIANTS)
set -e
cd $GRAFT_REPO_ROOT/taper_amd/csrc
cp ../lib/libtaper_hip.so /tmp/lib_default.so
OBJS=$(ls _build/*.o | grep -v conv_chain.o)
IFS=';' read -ra VS <<< "${VARIANTS:-}"
run() { cd $GRAFT_REPO_ROOT; python tools/chain_mlp3_time.py 256 | tail -1; cd $GRAFT_REPO_ROOT/taper_amd/csrc; }
echo "== default"; run
for v in "${VS[@]}"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off $v -c conv_chain.hip -o /tmp/conv_chain_v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtaper_hip.so $OBJS /tmp/conv_chain_v.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
    echo "== $v"; run
done
cp /tmp/lib_default.so ../lib/libtaper_hip.so

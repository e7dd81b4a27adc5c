#!/bin/bash
# Runs on the GPU box: in-kernel phase timing (TH_PROFILE stamps of workgroup 100) of the reference CNN's conv chain at batch 256.
# usage: tools/prof_chain.sh [extra -D flags]
set -e
cd $GRAFT_REPO_ROOT/taper_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off -DTH_PROFILE "$@" -c conv_chain.hip -o /tmp/conv_chain_prof.o
OBJS=$(ls _build/*.o | grep -v conv_chain.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtaper_hip.so $OBJS /tmp/conv_chain_prof.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
cd $GRAFT_REPO_ROOT
python ${PROF_SCRIPT:-tools/prof_chain.py} 100

#!/bin/bash
# Runs on the GPU box: in-kernel phase timing (TH_PROFILE stamps) of the image-resident conv kernel on the batch-256 layers.
# usage: tools/prof_conv_img.sh [extra -D flags, e.g. -DTH_IMG_NO_READS]
set -e
cd $GRAFT_REPO_ROOT/taper_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off -DTH_PROFILE "$@" -c conv_mfma.hip -o /tmp/conv_mfma_prof.o 2>/dev/null
OBJS=$(ls _build/*.o | grep -v conv_mfma.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtaper_hip.so $OBJS /tmp/conv_mfma_prof.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
cd $GRAFT_REPO_ROOT
for args in "256 32 28 28 32 plain" "256 32 28 28 32 pool" "256 64 14 14 64 pool" "256 32 14 14 64 plain" "256 64 7 7 128 gap"; do echo "== $args $*"; python tools/prof_conv.py $args; done

#!/bin/bash
# Runs on the GPU box: per-workgroup timeline of the image-resident conv kernel (PROFILE build of conv_mfma.hip only).
set -e
cd $GRAFT_REPO_ROOT/taper_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off -DTH_PROFILE -c conv_mfma.hip -o /tmp/conv_mfma_prof.o 2>/dev/null
OBJS=$(ls _build/*.o | grep -v conv_mfma.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtaper_hip.so $OBJS /tmp/conv_mfma_prof.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
cd $GRAFT_REPO_ROOT
for args in "256 32 28 28 32" "256 64 14 14 64" "256 64 7 7 128"; do echo "== $args"; python tools/conv_timeline.py $args; done

cd $GRAFT_REPO_ROOT/taper_amd/csrc
OBJS=$(ls _build/*.o | grep -v "/comm.o")
for probe in "" "-DP2P_PROBE_NOACQ" "-DP2P_PROBE_NOREL" "-DP2P_PROBE_NOACQ -DP2P_PROBE_NOREL"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off $probe -c comm.hip -o /tmp/comm_probe.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtaper_hip.so $OBJS /tmp/comm_probe.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  echo "probe: ${probe:-none}"
  (cd $GRAFT_REPO_ROOT && bash tools/profile_dp.sh 2 2>&1 | grep -A3 "W = 1" | grep -E "p2p_|adam_kernel")
done

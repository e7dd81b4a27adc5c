#!/bin/bash
# Runs on the GPU box: the data-parallel step's all-reduce kernels under rocprofv3 with W ranks SHARING GPU 0 (the box has one GPU): the
# one-shot peer-to-peer all-reduce fused with Adam and the in-place form -- their durations here are the LATENCY FLOOR of the exchange
# (flag handshake + reading W arenas through IPC mappings of the same device); over xGMI the reads cross links instead.  usage: profile_dp.sh [W]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${1:-2}
OUT=$ROOT/gpurun_out/profile_dp
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for fuse in 1 0; do
    KEY=dp$RANDOM
    rm -rf /tmp/dp_out /tmp/dp_tr_*; mkdir -p /tmp/dp_out
    for r in $(seq 0 $((W-1))); do
        RANK=$r LOCAL_RANK=$r WORLD_SIZE=$W MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 TAPER_DP_OUT=/tmp/dp_out TAPER_DP_STEPS=200 TAPER_DP_GLOBAL_BATCH=$((128*W)) \
        TAPER_DP_MODE=eager TAPER_DP_BACKEND=p2p TAPER_DP_KEY=$KEY TAPER_DP_DEVICE=0 TAPER_P2P_FUSE=$fuse TAPER_NO_GRAPH=1 \
        timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp_tr_$r -- python $ROOT/tests/dp_worker.py > /tmp/dp_out/log_$r.txt 2>&1 &
    done
    wait
    { echo "### W = $W ranks on one GPU, 128 rows per rank, MLP 784-128-10, p2p communicator, fuse_adam = $fuse (rank 0's trace; 400 steps + the self-check's launches)"
      python $ROOT/tools/kstats.py /tmp/dp_tr_0/*/*kernel_stats.csv | grep -E "p2p_|adam_kernel|mlp_tail|sgemm_small16" ; } > "$OUT/w${W}_fuse$fuse.txt"
    cat "$OUT/w${W}_fuse$fuse.txt"; tail -2 /tmp/dp_out/log_0.txt
done
# W = 1: the kernels' own cost with nobody to wait for (flag push + poll on the local block, one arena read, Adam / in-place store)
cat > /tmp/dp_w1.py <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import taper_amd as T
fuse = os.environ.get("TAPER_P2P_FUSE", "1") == "1"
model = T.Sequential([T.Linear(784, 128, True, seed=1), T.ReLU(), T.Linear(128, 10, True, seed=2)])
opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
comm = T.Communicator.p2p(1, 0)
comm.connect(comm.export_arena(opt))
comm.set_fuse_adam(fuse)
assert comm.self_check(opt)
tr = T.Trainer(model, opt, comm=comm)
rng = np.random.default_rng(0)
x, y = T.Tensor(rng.uniform(0, 1, (128, 784)).astype(np.float32)), T.Tensor(rng.integers(0, 10, 128).astype(np.float32))
for _ in range(300):
    tr.train_step(x, y)
print("ok", comm.stats())
PY
for fuse in 1 0; do
    rm -rf /tmp/dp_tr_w1
    TAPER_P2P_FUSE=$fuse timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp_tr_w1 -- python /tmp/dp_w1.py > /tmp/dp_w1.log 2>&1
    { echo "### W = 1 (one rank, its own arena: nobody to wait for), fuse_adam = $fuse, 300 eager steps"
      python $ROOT/tools/kstats.py /tmp/dp_tr_w1/*/*kernel_stats.csv | grep -E "p2p_|adam_kernel" ; } > "$OUT/w1_fuse$fuse.txt"
    cat "$OUT/w1_fuse$fuse.txt"; grep -E "^ok|Error|error" /tmp/dp_w1.log | head -3
done

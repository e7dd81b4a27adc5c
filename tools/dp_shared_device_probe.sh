# W processes on GPU 0 train MODEL data-parallel through the in-launch exchange, TRIALS times; prints how many ranks finished each trial and,
# for one that timed out, where the first wait ran out (and, with TAPER_DP_POSTMORTEM=1, what the receive regions held).  DESIGN 6e /
# profiles/r06_dp_three_ranks_one_device.txt: mlp_64 at W = 3 needs TAPER_DP_SHARED_RULE=places to take the in-launch form at all; with the
# event-ring build (experiments/dp_event_ring.patch) a time-out also prints the `[dp ring ...]` lines tools/dp_placement.py reads.
# usage: gpurun -- bash tools/dp_shared_device_probe.sh MODEL W GLOBAL_BATCH TRIALS [ENV=VAL ...]     (TMO=ms: the wait bound, default 5000)
cd $GRAFT_REPO_ROOT
MODEL=$1; W=$2; GB=$3; TRIALS=$4; shift 4
for e in "$@"; do export "$e"; done
for trial in $(seq 1 $TRIALS); do
T0=$(date +%s)
KEY=w3$RANDOM; rm -rf /tmp/w3out; mkdir -p /tmp/w3out
for r in $(seq 0 $((W-1))); do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=$W MASTER_ADDR=127.0.0.1 MASTER_PORT=29581 TAPER_DP_OUT=/tmp/w3out TAPER_DP_STEPS=3 TAPER_DP_GLOBAL_BATCH=$GB TAPER_DP_MODE=graph \
  TAPER_DP_BACKEND=p2p TAPER_DP_KEY=$KEY TAPER_DP_DEVICE=0 TAPER_DP_MODEL=$MODEL TAPER_P2P_TIMEOUT_MS=${TMO:-5000} HSA_ENABLE_IPC_MODE_LEGACY=0 \
  timeout 100 python tests/dp_worker.py > /tmp/w3out/log_$r.txt 2>&1 &
done
wait
echo "$MODEL W=$W $* trial $trial [$(( $(date +%s) - T0 )) s]: $(ls /tmp/w3out/*.npz 2>/dev/null | wc -l) ranks finished"; for r in $(seq 0 $((W-1))); do tail -1 /tmp/w3out/log_$r.txt | grep -o "first wait.*" | cut -c1-200; grep "post-mortem" /tmp/w3out/log_$r.txt | cut -c1-200; done
done

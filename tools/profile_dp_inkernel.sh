#!/bin/bash
# Runs on the GPU box: the data-parallel MLP step (128 rows per rank, BASELINE configs[3]'s shard) under rocprofv3, per form, one trace each:
#   single        no communicator: sgemm_small16_tick + mlp_tail_exact_kernel<8,2,false,8>
#   loopback      W = 2 with the process as its own peer: the gradient launch is mlp_tail_exact_kernel<8,2,false,8,2> -- the exchange inside
#   three_launch  one rank, the r05 form: the gradient launch without fused updates + p2p_allreduce_adam_kernel
#   w2            two processes on GPU 0, the exchange inside the gradient launch (rank 0's trace)
#   cnn_*         the same three forms for the simple CNN at 128 images: conv_chain_simple_kernel + wide_grads_kernel<0 | 2> (| + p2p_allreduce_adam)
# Steps are enqueued launch by launch (TAPER_NO_GRAPH=1: rocprofv3 on ROCm 7.2 crashes in back-to-back graph replays), so the kernel
# durations are the measurement, not the step time.  -> gpurun_out/profile_dp_inkernel/*.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profile_dp_inkernel
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 TAPER_NO_GRAPH=1 GRAFT_REPO_ROOT=$ROOT
cat > /tmp/dp_forms.py <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import taper_amd as T
form = os.environ["DP_FORM"]
cnn = form.startswith("cnn_")       # cnn_single / cnn_loopback / cnn_three_launch: the simple CNN, 128 images (th_wide_head_grads[_dp])
form = form[4:] if cnn else form
if cnn:
    Cv = lambda i, o, s: T.Conv2dReLU(i, o, (3, 3), (1, 1), (1, 1), None, None, True, seed=s)
    model = T.Sequential([Cv(1, 32, 1), T.MaxPool2d((2, 2), (2, 2)), Cv(32, 64, 2), T.MaxPool2d((2, 2), (2, 2)), T.Flatten(1), T.Linear(3136, 10, True, 3)])
else:
    model = T.Sequential([T.Linear(784, 128, True, seed=1), T.ReLU(), T.Linear(128, 10, True, seed=2)])
opt = T.Adam(model.parameters(), 1e-3, None, None, 1e-4)
comm = None
if form == "loopback":
    comm = T.Communicator.loopback()
elif form == "three_launch":
    comm = T.Communicator.p2p(1, 0)
    comm.connect(comm.export_arena(opt))
    comm.set_inkernel(False)
tr = T.Trainer(model, opt, comm=comm, **({"sample_shape": (1, 28, 28)} if cnn else {}))
ds = T.MNISTDataset.synthetic(128 * 100, seed=3)
loader = T.DataLoader(ds, 128, False)
for _ in range(5):
    tr.run_epoch(loader, T.Trainer.GRAPH)
print("ok", form, comm.inkernel_launches() if comm else None)
PY
for form in ${FORMS:-single loopback three_launch cnn_single cnn_loopback cnn_three_launch}; do
    rm -rf /tmp/dp_tr_$form
    DP_FORM=$form timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp_tr_$form -- python /tmp/dp_forms.py > /tmp/dp_$form.log 2>&1
    { echo "### $form: $([ "${form#cnn_}" != "$form" ] && echo "the simple CNN, 128 images" || echo "MLP 784-128-10, 128 rows"), 500 steps enqueued launch by launch (us: avg per launch)"
      python $ROOT/tools/kstats.py /tmp/dp_tr_$form/*/*kernel_stats.csv | grep -E "p2p_|adam|mlp_tail|sgemm_small16|conv_chain|wide_grads|Name" ; } > "$OUT/$form.txt"
    cat "$OUT/$form.txt"; grep -E "^ok|Error|error" /tmp/dp_$form.log | head -3
done
# two processes on GPU 0
KEY=dp$RANDOM
rm -rf /tmp/dp_out /tmp/dp_tr_w2_*; mkdir -p /tmp/dp_out
for r in 0 1; do
    RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29573 TAPER_DP_OUT=/tmp/dp_out TAPER_DP_STEPS=200 TAPER_DP_GLOBAL_BATCH=256 \
    TAPER_DP_MODE=graph TAPER_DP_BACKEND=p2p TAPER_DP_KEY=$KEY TAPER_DP_DEVICE=0 \
    timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp_tr_w2_$r -- python $ROOT/tests/dp_worker.py > /tmp/dp_out/log_$r.txt 2>&1 &
done
wait
{ echo "### w2: two processes on ONE GPU, 128 rows per rank, the exchange inside the gradient launch (rank 0's trace; 400 steps + the self-check's launches)"
  python $ROOT/tools/kstats.py /tmp/dp_tr_w2_0/*/*kernel_stats.csv | grep -E "p2p_|adam|mlp_tail|sgemm_small16|dp_|Name" ; } > "$OUT/w2.txt"
cat "$OUT/w2.txt"; tail -2 /tmp/dp_out/log_0.txt

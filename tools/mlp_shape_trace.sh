#!/bin/bash
# GPU box: rocprofv3 kernel stats of Trainer steps of one MLP shape (tools/mlp_shape_sweep.py SHAPE at one batch).  usage: mlp_shape_trace.sh 256 16384
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
SHAPE=$1 BATCH=$2
cat > /tmp/one_shape.py <<PY
import sys, time
sys.path.insert(0, "$ROOT")
import taper_amd as T, bench
hs=[int(v) for v in "$SHAPE".split("x")]; b=$BATCH
ds=T.MNISTDataset.synthetic(60000, seed=7)
dims=[784]+hs+[10]; layers=[]
for i in range(len(dims)-1):
    layers.append(T.Linear(dims[i],dims[i+1],True,1+i))
    if i+2<len(dims): layers.append(T.ReLU())
model=T.Sequential(layers); opt=T.Adam(model.parameters(),1e-3,None,None,1e-4); tr=T.Trainer(model,opt); loader=T.DataLoader(ds,b,False)
import os
os.environ["TAPER_NO_GRAPH"]="1"
bench.run_steps(T,tr,loader,60); T.Device.sync()
PY
rm -rf /tmp/tr_shape
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_shape -- python /tmp/one_shape.py > /dev/null 2>/tmp/tr_shape.err
python - <<PY
import csv, glob
for f in glob.glob("/tmp/tr_shape/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < ${3:-16}: print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']:>6s} %")
PY

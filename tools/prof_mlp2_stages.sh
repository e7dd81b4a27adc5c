#!/bin/bash
# GPU box: in-kernel stage timing (TH_PROFILE stamps, 100 MHz wall clock) of th_mlp2_xent's three launches.  BATCH=16384 by default.
set -e
cd $GRAFT_REPO_ROOT/taper_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off -DTH_PROFILE "$@" -c mlp2.hip -o /tmp/mlp2_prof.o
OBJS=$(ls _build/*.o | grep -v mlp2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtaper_hip.so $OBJS /tmp/mlp2_prof.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
cd $GRAFT_REPO_ROOT
python - <<'PY'
import ctypes as C, os, numpy as np
from taper_amd import hip
from taper_amd.hip import AdamFuse, RowSource
from taper_amd._lib import hip as lib
lib.th_debug_mlp2_prof.argtypes=[C.c_void_p,C.c_void_p]; lib.th_debug_mlp2_prof.restype=C.c_int
ctx=hip.Ctx(0); rng=np.random.default_rng(0)
inf,hid,c,n_rows=784,128,10,60000
batch=int(os.environ.get("BATCH",16384))
data=ctx.upload(rng.integers(0,256,(n_rows,inf)).astype(np.float32)/np.float32(255)); labels=ctx.upload(rng.integers(0,c,n_rows).astype(np.float32))
idx=ctx.upload(rng.permutation(n_rows).astype(np.int32))
dev=dict(w1=ctx.upload((rng.uniform(-1,1,(hid,inf))*np.sqrt(2/inf)).astype(np.float32)),b1=ctx.zeros(hid),w2=ctx.upload(rng.uniform(-.3,.3,(c,hid)).astype(np.float32)),b2=ctx.zeros(c))
mom={k:(ctx.zeros(n),ctx.zeros(n)) for k,n in (("w1",hid*inf),("b1",hid),("w2",c*hid),("b2",c))}
tick,dlr=ctx.upload(np.array([0,0],np.int32)),ctx.upload(np.array([1e-3],np.float32))
fuses=[AdamFuse(int(dev[k]),int(mom[k][0]),int(mom[k][1]),int(tick),int(dlr),0.9,0.999,1e-8,1e-4) for k in ("w1","b1","w2","b2")]
out=dict(dw1=ctx.empty(hid*inf),db1=ctx.empty(hid),dw2=ctx.empty(c*hid),db2=ctx.empty(c),loss=ctx.empty(1),nc=ctx.empty(1))
state=ctx.upload(np.array([0,0],np.int64)); src=RowSource(int(data),int(labels),int(idx),state.offset(8),n_rows,n_rows)
def step():
    ctx.call("th_mlp2_xent",C.byref(src),batch,inf,hid,c,dev["w1"],dev["b1"],dev["w2"],dev["b2"],out["dw1"],out["db1"],out["dw2"],out["db2"],out["loss"],out["nc"],None,0,None,0,tick,*[C.byref(f) for f in fuses])
names=["rows: entry -> first chunk landed (block 0)","rows: k loop","rows: H -> LDS, logits, softmax","rows: dZ1, dW2 / db partials (block 0 end)","rows: block 0 end -> last block end",
       "last rows block end -> dW1 block 0 has its row indices","dW1: k loop (block 0)","dW1: partial stores (block 0)","dW1: block 0 end -> last block end","last dW1 block end -> finish block 0 entry","finish: block 0 entry -> last block entry"]
acc=np.zeros(len(names)); N=30
for it in range(N+5):
    for _ in range(10): step()
    o=(C.c_longlong*16)(); lib.th_debug_mlp2_prof(ctx.h,o)
    ts=[o[i] for i in (0,1,2,3,4,5,6,7,8,9,10,11)]
    if it>=5: acc+=np.diff(ts)*0.01
for n,v in zip(names,acc/N): print(f"{v:8.2f} us  {n}")
print(f"{acc.sum()/N:8.2f} us  rows entry -> finish's last block entry")
PY

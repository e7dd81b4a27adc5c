#!/bin/bash
# Runs on the GPU box: in-kernel stage timing (TH_PROFILE stamps of workgroup 3) of th_mlp3_xent's rows launch at the reference CNN's classifier shape
set -e
cd $GRAFT_REPO_ROOT/taper_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off -DTH_PROFILE "$@" -c mlp3.hip -o /tmp/mlp3_prof.o
OBJS=$(ls _build/*.o | grep -v mlp3.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtaper_hip.so $OBJS /tmp/mlp3_prof.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
cd $GRAFT_REPO_ROOT
python - <<'PY'
import ctypes as C, numpy as np
from taper_amd import hip
from taper_amd._lib import hip as lib
lib.th_debug_mlp3_prof.argtypes=[C.c_void_p,C.c_void_p]; lib.th_debug_mlp3_prof.restype=C.c_int
ctx=hip.Ctx(0); rng=np.random.default_rng(0)
B,in_f,h1,h2,c=256,128,128,64,10
x,y=ctx.upload(rng.uniform(-1,1,(B,in_f)).astype(np.float32)),ctx.upload(rng.integers(0,c,B).astype(np.float32))
layers,keep=(hip.Mlp3Layer*3)(),[]
for l,(o,i) in enumerate(((h1,in_f),(h2,h1),(c,h2))):
    w,b,gw,gb=ctx.upload(rng.uniform(-.1,.1,(o,i)).astype(np.float32)),ctx.zeros(o),ctx.empty(o*i),ctx.empty(o)
    keep.append((w,b,gw,gb)); layers[l]=hip.Mlp3Layer(int(w),int(b),int(gw),int(gb),None,None,o)
gx=ctx.empty(B*in_f); loss,nc=ctx.empty(1),ctx.empty(1); lp=C.cast(layers,C.c_void_p)
acc=np.zeros(9); N=30
names=["X -> LDS (+ L1, L2 weights requested)","L1 forward","L2 forward","logits + softmax","dZ2","dZ1","(stamp 7)","","dX"]
for it in range(N+5):
    for _ in range(20): ctx.call("th_mlp3_xent",x,y,B,in_f,lp,gx,loss,nc,None,0,None,0,None,None)
    out=(C.c_longlong*16)(); lib.th_debug_mlp3_prof(ctx.h,out)
    ts=[out[i] for i in (0,1,2,3,4,5,6,7,9)]
    if it>=5: acc[:8]+=np.diff(ts)*0.01
for n,v in zip(["X -> LDS","L1 forward","L2 forward","logits + softmax","dZ2","dZ1","-","dX"],acc/N): print(f"{v:7.3f} us  {n}")
print(f"{acc.sum()/N:7.3f} us  first stamp to last")
PY

#!/bin/bash
# Runs on the GPU box: in-kernel phase timing (TH_PROFILE stamps of workgroup 5) of the wide classifier head (simple CNN: Linear(3136, 10), batch 256)
set -e
cd $GRAFT_REPO_ROOT/taper_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off -DTH_PROFILE "$@" -c wide_head.hip -o /tmp/wh_prof.o
OBJS=$(ls _build/*.o | grep -v wide_head.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtaper_hip.so $OBJS /tmp/wh_prof.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
cd $GRAFT_REPO_ROOT
python - <<'PY'
import ctypes as C, numpy as np
from taper_amd import hip
from taper_amd._lib import hip as lib
lib.th_debug_wide_prof.argtypes=[C.c_void_p,C.c_void_p]; lib.th_debug_wide_prof.restype=C.c_int
ctx=hip.Ctx(0); rng=np.random.default_rng(0)
B,K,c=256,3136,10
x=ctx.upload(rng.uniform(0,1,(B,K)).astype(np.float32)); w=ctx.upload(rng.uniform(-.05,.05,(c,K)).astype(np.float32)); b=ctx.zeros(c)
y=ctx.upload(rng.integers(0,c,B).astype(np.float32)); loss,nc=ctx.empty(1),ctx.empty(1); dw,db,cs=ctx.empty(c*K),ctx.empty(c),ctx.empty(K)
acc=np.zeros(5); N=30
for it in range(N+5):
    for _ in range(10): ctx.call("th_linear_xent_wide_ex",x,w,b,y,B,K,c,loss,nc,None,dw,db,None,0,None,0,None,cs)
    out=(C.c_longlong*16)(); lib.th_debug_wide_prof(ctx.h,out)
    if it>=5: acc+=np.diff([out[i] for i in range(6)])*0.01
for n,v in zip(["entry -> logits summed (operand loads)","softmax","transpose + dX / colsum + dW MFMAs","column-sum reduce","dW reduce + stores"],acc/N): print(f"{v:7.3f} us  {n}")
print(f"{acc.sum()/N:7.3f} us  entry to exit")
PY

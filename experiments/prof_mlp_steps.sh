#!/bin/bash
# Runs on the GPU box: in-kernel phase timing (TH_PROFILE stamps, last step, workgroups 0 and 5) of the persistent MLP step launch
set -e
cd $GRAFT_REPO_ROOT/taper_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off -DTH_PROFILE "$@" -c mlp_steps.hip -o /tmp/ms_prof.o
OBJS=$(ls _build/*.o | grep -v mlp_steps.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libtaper_hip.so $OBJS /tmp/ms_prof.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
cd $GRAFT_REPO_ROOT
python - <<'PY'
import ctypes as C, numpy as np
from taper_amd import hip
from taper_amd._lib import hip as lib
lib.th_debug_mlp_steps_prof.argtypes=[C.c_void_p,C.c_void_p]; lib.th_debug_mlp_steps_prof.restype=C.c_int
ctx=hip.Ctx(0); rng=np.random.default_rng(0)
B,in_f,hid,c,steps=64,784,128,10,64
ws=[rng.uniform(-.05,.05,(hid,in_f)).astype(np.float32),np.zeros(hid,np.float32),rng.uniform(-.2,.2,(c,hid)).astype(np.float32),np.zeros(c,np.float32)]
t,dlr=ctx.upload(np.zeros(1,np.int32)),ctx.upload(np.array([1e-3],np.float32))
P,M,V=[ctx.upload(w) for w in ws],[ctx.zeros(w.size) for w in ws],[ctx.zeros(w.size) for w in ws]
fuse=(hip.AdamFuse*4)(*[hip.AdamFuse(int(P[i]),int(M[i]),int(V[i]),int(t),int(dlr),0.9,0.999,1e-8,1e-4) for i in range(4)])
herr=C.c_void_p(); lib.th_host_malloc(ctx.h,64,C.byref(herr)); C.cast(herr,C.POINTER(C.c_int))[0]=0
x,y=ctx.upload(rng.uniform(0,1,(steps*B,in_f)).astype(np.float32)),ctx.upload(rng.integers(0,c,steps*B).astype(np.float32))
loss,met,st=ctx.empty(1),ctx.zeros(2*steps),ctx.upload(np.zeros(2,np.int64))
acc=np.zeros((2,7)); N=10
for it in range(N+2):
    ctx.call("th_mlp2_steps",x,y,steps,B,in_f,hid,c,C.cast(fuse,C.c_void_p),loss,met,steps,st,B,herr)
    out=(C.c_longlong*32)(); lib.th_debug_mlp_steps_prof(ctx.h,out)
    if it>=2:
        for r in range(2): acc[r]+=np.diff([out[16*r+i] for i in range(8)])*0.01
wg=(C.c_longlong*128)(); lib.th_debug_mlp_steps_wg.argtypes=[C.c_void_p,C.c_void_p]; lib.th_debug_mlp_steps_wg(ctx.h,wg)
w=np.array(list(wg),dtype=np.int64).reshape(32,4); base=w[:,0].min()
print('per workgroup (last step, us from the earliest phase-A start): start, arrives at barrier A, released, arrives at barrier B')
for i in range(32): print(f'  wg {i:2d} (row tile {i//8}, hidden tile {i%8}): ' + ' '.join(f'{(v-base)*0.01:6.2f}' for v in w[i]))
names=["W2 / b2 update (workgroup 0)","H tile","barrier A","logits, softmax, dZ1 tile","db1, head gradients, log","dW1 tiles + Adam","barrier B"]
for r,role in enumerate(["workgroup 0","workgroup 5"]):
    print(role)
    for n,v in zip(names,acc[r]/N): print(f"  {v:7.3f} us  {n}")
    print(f"  {acc[r].sum()/N:7.3f} us  per step")
PY

#!/usr/bin/env python3
"""Both matrix-core 3x3 kernels (mode 0: 128-pixel blocks; -1: image-resident) on the batch-256 layers of the CNNs, plain and pooled.
usage: conv_compare.py [reps]"""
import sys, numpy as np
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 50
sys.path.insert(0,'/root/repo')
from taper_amd import hip
import ctypes as C
ctx=hip.Ctx(0); rng=np.random.default_rng(0)
LAYERS=[(256,32,28,28,32),(256,32,14,14,64),(256,64,14,14,64),(256,64,7,7,128)]
for mode in (0,-1):
    ctx.call("th_debug_set_conv_img", mode)
    for n,ci,h,w,co in LAYERS:
        x=ctx.upload(rng.standard_normal((n,ci,h,w)).astype(np.float32)); wt=ctx.upload(rng.standard_normal((co,ci,3,3)).astype(np.float32)); b=ctx.upload(rng.standard_normal(co).astype(np.float32))
        y=ctx.empty(n*co*h*w); yp=ctx.empty(n*co*h*w//4)
        res=[]
        for name,call in (("plain",lambda: ctx.call("th_conv3x3_fwd",x,wt,b,y,n,ci,h,w,co,1,0,1)),("pool",lambda: ctx.call("th_conv3x3_pool2_fwd",x,wt,b,yp,n,ci,h,w,co,1,1))):
            if name=="pool" and h%2: continue
            for _ in range(5): call()
            e0,e1=hip.Event(),hip.Event(); ctx.record(e0)
            for _ in range(REPS): call()
            ctx.record(e1); us=hip.Ctx.elapsed_ms(e0,e1)*1e3/REPS
            cfg=(C.c_int*6)(); ctx.call("th_debug_last_conv_config",C.cast(cfg,C.c_void_p))
            res.append(f"{name} {us:6.2f} us {2*9*ci*co*h*w*n/us/1e6:6.1f} TF cfg={list(cfg)}")
        print("mode",mode,f"{ci}->{co}@{h}", " | ".join(res))

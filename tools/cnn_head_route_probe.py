#!/usr/bin/env python3
"""Which step form the CNNs should take above one image per CU: the classifier's rows inside the chain launch (default up to its cap) or as
launches of their own.  usage: [TAPER_CHAIN_HEAD=0] cnn_head_route_probe.py  (the switch is read once per process)"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import taper_amd as T
import bench

ds = T.MNISTDataset.synthetic(60000, seed=7)
for key in ("cnn_simple", "cnn_reference"):
    out = bench.cnn_batch_sweep(T, key, (1, 28, 28), 1e-2, ds, batches=(256, 512, 1024, 4096))
    print(f"TAPER_CHAIN_HEAD={os.environ.get('TAPER_CHAIN_HEAD', '1')} {key}: " + "  ".join(f"b{r['batch']}: {r['ms_per_step'] * 1e3:.1f} us" for r in out), flush=True)
